#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_full.py tests/test_gpu_fullwidth.py tests/test_gpu_vision_full.py -q 2>&1 | tail -4
for o in "" "--opt gemm256=9"; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-rccl-selftest $o 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$o', 'img/s %.2f step %.1f loop %.1f prefill %.2f vq %.2f' % (j['value'], j['ms_per_step'], j['last_step_ms']['decode_loop'], j['last_step_ms']['prefill'], j['last_step_ms']['vq_decode']))"; done
for o in "" "--opt gemm256=9"; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-rccl-selftest $o 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$o', 'img/s %.2f step %.1f loop %.1f prefill %.2f vq %.2f' % (j['value'], j['ms_per_step'], j['last_step_ms']['decode_loop'], j['last_step_ms']['prefill'], j['last_step_ms']['vq_decode']))"; done
