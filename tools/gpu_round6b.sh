#!/bin/bash
# Round-6 GPU pass B: the whole GPU suite (no -x: every failure in one pass), VQ tail / GroupNorm-epilogue A/B, default bench line.
tag=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 1500 "$@" > gpurun_out/tests_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_$tag.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/tests_$tag.log | cut -c1-300 | tail -30
for i in 1 2; do for v in 1 0; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --opt vq_tail_fused=$v > gpurun_out/ab_vqtail${v}_${i}_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_vqtail${v}_${i}_$tag.json"))
print("vq_tail_fused=$v run $i: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 2) for k, v in d["phase_ms"].items()})
PY
done; done
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("value", d["value"], "phase_ms", {k: round(v["mean"], 1) for k, v in d["phase_ms"].items()})
print("host", json.dumps(d.get("host")))
rf = d.get("roofline", {})
print({k: round(v.get("frac", 0), 3) for k, v in rf.get("classes", {}).items()})
PY
tail -3 gpurun_out/bench_$tag.err
