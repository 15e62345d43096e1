#!/bin/bash
# final pass of round 6: class traces on the final sources (for profiles/kernel_classes_b*.json), smoke, the whole GPU suite, the default bench line (twice)
tag=${1:-final6}
mkdir -p gpurun_out
for b in 64 32 8; do bash tools/trace_batch.sh $b r06_b$b > /dev/null 2>&1; done
for b in 64 32 8; do python tools/trace_classes.py gpurun_out/trace_r06_b${b}_summary.md $b gpurun_out/kernel_classes_b$b.json; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/smoke_$tag.log
python -m pytest tests -m gpu -q --timeout 1500 > gpurun_out/tests_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_$tag.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/tests_$tag.log | cut -c1-300 | tail -12
for i in 1 2; do
  /usr/bin/time -f "bench wall %e s" python bench.py > gpurun_out/bench_${tag}_$i.json 2> gpurun_out/bench_${tag}_$i.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_${tag}_$i.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_${tag}_$i.json"))
print("value", d["value"], "phase_ms", {k: round(v["mean"], 1) for k, v in d["phase_ms"].items()}, "loop_roofline", round(d["loop_roofline"]["frac"], 3), round(d["loop_roofline"]["moved_frac"], 3))
rf = d.get("roofline", {})
print("roofline", rf.get("frac"), "traffic", rf.get("traffic"), "rocprof_frac", rf.get("rocprof_frac"), "gemm_norm_phase", (rf.get("decode_gemm_norm_phase") or {}).get("frac"))
print({k: round(v.get("frac", 0), 3) for k, v in rf.get("classes", {}).items()})
print("secondary", {k: (v.get("images_per_s") or v.get("samples_per_s")) for k, v in d.get("secondary", {}).items() if isinstance(v, dict)}, "cpu", d.get("cpu_baseline", {}).get("value"))
print("host", json.dumps(d.get("host"))[:400])
PY
done
