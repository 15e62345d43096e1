#!/bin/bash
tag=${1:-h}
mkdir -p gpurun_out
python tools/xa_bench.py 2>&1 | grep -E "509|530|400|531" | tee gpurun_out/xa_bench_$tag.log
for i in 1 2; do for v in 3 1; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --diag-opt sk3_xa=$v > gpurun_out/ab_nt${v}_${i}_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_nt${v}_${i}_$tag.json"))
print("sk3_xa=$v (3 = nt W loads) run $i: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 2) for k, v in d["phase_ms"].items()})
PY
done; done
