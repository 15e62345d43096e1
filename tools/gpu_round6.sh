#!/bin/bash
# Round-6 GPU pass A: the reference-relative bf16 parity files first (their own log, -s for the measured ratios), then an A/B of the
# deferred-1/rms decode norm on the diagnostics library (same box, alternating), then the default bench line (with the `secondary` block).
#   usage: gpurun --timeout 3000 -- 'bash tools/gpu_round6.sh <tag> [ab|tests|bench ...]'
tag=${1:-x}; shift
what=${@:-tests ab bench}
mkdir -p gpurun_out
for w in $what; do
  case $w in
  tests)
    python -m pytest tests/test_gpu_fullconfig.py tests/test_gpu_fulldepth.py tests/test_gpu_fullconfig_text.py tests/test_gpu_fullvocab.py tests/test_gpu_fullwidth.py \
        tests/test_gpu_smallbatch.py tests/test_gpu_path.py tests/test_gpu_vision_full.py tests/test_gpu_vision_fulldepth.py tests/test_gpu_full.py tests/test_gpu_ops.py \
        -q -s --timeout 1500 > gpurun_out/anchored_$tag.log 2>&1; echo "anchored tests rc=$?" | tee -a gpurun_out/anchored_$tag.log
    grep -E "passed|failed|reference-relative|E_hip exceeds|Error|error" gpurun_out/anchored_$tag.log | cut -c1-700 | tail -60 ;;
  ab)
    for i in 1 2; do for v in 1 0; do
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-rccl-selftest --no-gemm-phase --no-roofline --diag-opt defer_norm=$v > gpurun_out/ab_defer${v}_${i}_$tag.json 2> gpurun_out/ab_$tag.err
      python - <<PY
import json
d = json.load(open("gpurun_out/ab_defer${v}_${i}_$tag.json"))
print("defer_norm=$v run $i: %.2f images/s, phases" % d["value"], {k: round(v["mean"], 1) for k, v in d["phase_ms"].items()})
PY
    done; done ;;
  bench)
    python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
    python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("value", d["value"], "phase_ms", {k: round(v["mean"], 1) for k, v in d["phase_ms"].items()})
print("host", json.dumps(d.get("host")))
print("secondary", json.dumps(d.get("secondary"))[:3000])
rf = d.get("roofline", {})
print("roofline frac", rf.get("frac"), "gemm_norm_phase", json.dumps(rf.get("decode_gemm_norm_phase"))[:600])
print({k: round(v.get("frac", 0), 3) for k, v in rf.get("classes", {}).items()})
PY
    tail -5 gpurun_out/bench_$tag.err ;;
  esac
done
