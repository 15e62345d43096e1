#!/bin/bash
# In-loop A/B of engine options on the bs=64 bench workload: decode-loop ms per option set.
# usage: gpurun -- 'bash tools/ab_loop.sh "stream_gemm=0" "stream_gemm=1" "diag:attn_variant=100" ...'   (each arg = space-separated key=value list;
# a diag: prefix = pg_diag_set_option of libplangen_diag.so, the whole run then uses the diagnostics library)
for o in "$@"; do
  args=""; for kv in $o; do case $kv in diag:*) args="$args --diag-opt ${kv#diag:}";; *) args="$args --opt $kv";; esac; done
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-rccl-selftest ${BATCH:+--batch $BATCH} $args 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('%-40s img/s %.2f  step %.1f ms  loop %.1f  prefill %.1f  vq %.1f' % ('$o', j['value'], j['ms_per_step'], j['last_step_ms']['decode_loop'], j['last_step_ms']['prefill'], j['last_step_ms']['vq_decode']))"
done
