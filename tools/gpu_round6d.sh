#!/bin/bash
# Round-6 GPU pass D: whole GPU suite, VQ tail A/B + VQ timeline, default bench.
tag=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 1500 "$@" > gpurun_out/tests_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_$tag.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/tests_$tag.log | cut -c1-300 | tail -20
for i in 1 2; do for v in 1 0; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --opt vq_tail_fused=$v > gpurun_out/ab_vqtail${v}_${i}_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_vqtail${v}_${i}_$tag.json"))
print("vq_tail_fused=$v run $i: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 2) for k, v in d["phase_ms"].items()})
PY
done; done
bash tools/vq_timeline.sh vqtl_$tag > /dev/null 2>&1
python - <<PY
import re, collections
t = open("gpurun_out/vqtl_${tag}_timeline.md").read()
print(t.splitlines()[0])
agg = collections.defaultdict(lambda: [0, 0.0])
for m in re.finditer(r"\| \d+ \| \`([^\`]+)\` \| ([\d.]+) \|", t):
    k = m.group(1)[:44]; agg[k][0] += 1; agg[k][1] += float(m.group(2))
for k, (n, us) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]: print(f"{us / 1000:8.2f} ms {n:4d}  {k}")
PY
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("value", d["value"], "phase_ms", {k: round(v["mean"], 1) for k, v in d["phase_ms"].items()})
rf = d.get("roofline", {})
print({k: round(v.get("frac", 0), 3) for k, v in rf.get("classes", {}).items()})
print("secondary", {k: (v.get("images_per_s") or v.get("samples_per_s")) for k, v in d.get("secondary", {}).items() if isinstance(v, dict)})
PY
