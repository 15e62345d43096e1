#!/bin/bash
# final round-5 pass: class traces on the final sources, then the whole GPU suite + bench line
for b in 64 32 8; do bash tools/trace_batch.sh $b r05_b$b > /dev/null 2>&1; done
bash tools/gpu_round5.sh r05g
