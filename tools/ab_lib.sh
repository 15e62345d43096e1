#!/bin/bash
# Same-box A/B of two builds of the library: plangen_amd/lib/libplangen_hip.so (new) vs libplangen_hip_old.so.bin (old), alternating.
# usage: gpurun -- 'bash tools/ab_lib.sh [rounds]'   (BATCH=8 for another batch size)
L=plangen_amd/lib
cp $L/libplangen_hip.so /tmp/new.so; cp $L/libplangen_hip_old.so.bin /tmp/old.so
for r in $(seq 1 ${1:-2}); do
  for v in new old; do
    cp /tmp/$v.so $L/libplangen_hip.so
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-rccl-selftest ${BATCH:+--batch $BATCH} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('%-4s img/s %.2f  step %.1f ms  loop %.1f  prefill %.1f  vq %.1f' % ('$v', j['value'], j['ms_per_step'], j['last_step_ms']['decode_loop'], j['last_step_ms']['prefill'], j['last_step_ms']['vq_decode']))"
  done
done
cp /tmp/new.so $L/libplangen_hip.so
