#!/bin/bash
# Round-6 GPU pass C: the v5 decode GEMM -- correctness files first, then microbenchmarks / stamps / launch floor, then the in-loop A/B.
tag=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullwidth.py tests/test_gpu_fulldepth.py tests/test_gpu_full.py tests/test_gpu_smallbatch.py tests/test_gpu_fullvocab.py -q -x --timeout 1500 > gpurun_out/tests_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_$tag.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/tests_$tag.log | cut -c1-300 | tail -12
python tools/sk5_bench.py 2>&1 | tee gpurun_out/sk5_bench_$tag.log
for v in 500 501 502; do python tools/sk3_profile.py $v 2>&1 | tee -a gpurun_out/sk3_profile_$tag.log; done
python tools/launch_floor.py o 2>&1 | tee gpurun_out/launch_floor_$tag.log
for i in 1 2; do for v in 1 0; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --diag-opt sk5=$v > gpurun_out/ab_sk5${v}_${i}_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_sk5${v}_${i}_$tag.json"))
print("sk5=$v run $i: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 2) for k, v in d["phase_ms"].items()})
PY
done; done
for v in 1 0; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --opt vq_tail_fused=$v > gpurun_out/ab_vqtail${v}_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_vqtail${v}_$tag.json"))
print("vq_tail_fused=$v: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 2) for k, v in d["phase_ms"].items()})
PY
done
