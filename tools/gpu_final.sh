#!/bin/bash
# final pass of a round: class traces on the final sources (for profiles/kernel_classes_b*.json), then the whole GPU suite + the default bench line
for b in 64 32 8; do bash tools/trace_batch.sh $b r05_b$b > /dev/null 2>&1; done
bash tools/gpu_round5.sh ${1:-final}
