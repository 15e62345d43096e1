#!/bin/bash
# Round-6 GPU pass F: x prefetch two chunks ahead (sk3 XA = 2): microbenchmark, stamps, in-loop A/B.
tag=${1:-x}
mkdir -p gpurun_out
python tools/xa_bench.py 2>&1 | tee gpurun_out/xa_bench_$tag.log
python tools/sk3_profile.py 503 2>&1 | tee gpurun_out/sk3_profile_xa_$tag.log
for i in 1 2; do for v in 2 1; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --diag-opt sk3_xa=$v > gpurun_out/ab_xa${v}_${i}_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_xa${v}_${i}_$tag.json"))
print("sk3_xa=$v run $i: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 2) for k, v in d["phase_ms"].items()})
PY
done; done
