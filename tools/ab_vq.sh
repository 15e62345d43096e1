#!/bin/bash
# Same-box A/B of the VQ decode: the tree's library vs ab_old/old.so.bin (a build of an earlier commit), alternating.  usage: gpurun -- 'bash tools/ab_vq.sh [rounds]'
L=plangen_amd/lib
cp $L/libplangen_hip.so /tmp/new.so; cp ab_old/old.so.bin /tmp/old.so
for r in $(seq 1 ${1:-3}); do
  for v in new old; do
    cp /tmp/$v.so $L/libplangen_hip.so
    echo "$v $(python3 tools/vq_only.py 64 4 2>/dev/null | tail -2 | tr '\n' ' ')"
  done
done
cp /tmp/new.so $L/libplangen_hip.so
