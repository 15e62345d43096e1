#!/bin/bash
# the default bench line twice, with its wall time
tag=${1:-final6}
mkdir -p gpurun_out
for i in 1 2; do
  t0=$(date +%s); python bench.py > gpurun_out/bench_${tag}_$i.json 2> gpurun_out/bench_${tag}_$i.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
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
