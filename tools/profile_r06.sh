#!/bin/bash
# Round-6 evidence pass on the GPU box (summaries only come back):
#   kernel traces of one bench step at bs = 64 / 32 / 8, MFMA-utilisation counters of the MFMA kernels (prefill, VQ, SigLIP),
#   the HBM-traffic passes of the dominant kernel (pmc_attn.json), secondary-workload lines.
#   + SQ wave-state / LDS counters of the MFMA kernels the verdict names (where do their wave cycles go).
# usage: gpurun --timeout 2400 -- 'bash tools/profile_r06.sh r06'
tag=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for b in 64 32 8; do bash $ROOT/tools/trace_batch.sh $b ${tag}_b$b > /dev/null; done
cd /tmp && export TMPDIR=/tmp
pmc() {   # name, program args...
  n=$1; shift
  rm -rf $OUT/pmc_${tag}_$n
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_${tag}_$n -o p -- python3 "$@" > $OUT/pmc_${tag}_$n.log 2>&1
  find $OUT/pmc_${tag}_$n -name '*results.db' | head -1
}
{
  echo "## prefill of the bench batch (B=64, L=256): rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --batch 64 --steps 1 --warmup 0 --tokens 1"
  db=$(pmc prefill $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --batch 64 --steps 1 --warmup 0 --tokens 1)
  python3 $ROOT/tools/pmc_mfma.py $db gemm256_kernelI12PlainLoader attn_prefill_flash2_kernel gemm_big_kernelI12PlainLoader
  echo; echo "## VQ-16 decode + encode of 64 images: ... -- python3 tools/vq_only.py 64 1"
  db=$(pmc vq $ROOT/tools/vq_only.py 64 1)
  python3 $ROOT/tools/pmc_mfma.py $db conv3x3_halo_kernelI3EpiIDF16bELb1ELb0 conv3x3_halo_kernelI3EpiIDF16bELb1ELb1 gemm256_kernelI11ConvLoader gemm_big_kernelI11ConvLoader conv3x3_out_halo
  echo; echo "## SigLIP-L tower + aligner, 64 images: ... -- python3 tools/vit_only.py 64 1"
  db=$(pmc vit $ROOT/tools/vit_only.py 64 1)
  python3 $ROOT/tools/pmc_mfma.py $db attn_vit_resident_kernel gemm256_kernelI12PlainLoader gemm_big_kernelI12PlainLoader
} > $OUT/${tag}_mfma_counters.md 2>&1
rm -rf $OUT/pmc_${tag}_prefill $OUT/pmc_${tag}_vq $OUT/pmc_${tag}_vit
cat $OUT/${tag}_mfma_counters.md
{
  echo "## SQ wave-state / LDS counters (tools/pmc_sq.sh; share of SQ_WAVE_CYCLES)"
  echo "### conv3x3_halo_kernel (VQ decode, tools/vq_only.py 64 1)"; bash $ROOT/tools/pmc_sq.sh ${tag}_halo conv3x3_halo_kernel tools/vq_only.py 64 1
  echo "### gemm256_kernel<ConvLoader> (VQ decode)"; bash $ROOT/tools/pmc_sq.sh ${tag}_g256c gemm256_kernelI11ConvLoader tools/vq_only.py 64 1
  echo "### gn_apply_kernel (VQ decode)"; bash $ROOT/tools/pmc_sq.sh ${tag}_gn gn_apply_kernel tools/vq_only.py 64 1
  echo "### gemm256_kernel<PlainLoader> (prefill of the bench batch)"; bash $ROOT/tools/pmc_sq.sh ${tag}_g256p gemm256_kernelI12PlainLoader bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --batch 64 --steps 1 --warmup 0 --tokens 1
  echo "### attn_prefill_flash2_kernel (prefill)"; bash $ROOT/tools/pmc_sq.sh ${tag}_fl2 attn_prefill_flash2_kernel bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --batch 64 --steps 1 --warmup 0 --tokens 1
} > $OUT/${tag}_sq_counters.md 2>&1
cat $OUT/${tag}_sq_counters.md
cd /tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --no-secondary --batch 64 --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_${tag}_$c -o p -- $BENCH --tokens 24 > $OUT/pmc_${tag}_$c.log 2>&1
done
F=$(find $OUT/pmc_${tag}_FETCH_SIZE -name '*results.db' | head -1); W=$(find $OUT/pmc_${tag}_WRITE_SIZE -name '*results.db' | head -1)
python3 $ROOT/tools/pmc_traffic.py $F $W --tokens 24 --json $OUT/pmc_attn_$tag.json \
  --cmd "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- $BENCH --tokens 24" | tee $OUT/${tag}_pmc_attn_traffic.txt
rm -rf $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE
cd $ROOT
python3 bench.py --no-cpu-baseline --no-shard-check --prompt-len 512 --steps 2 --warmup 1 > $OUT/${tag}_bench_L512.json 2> $OUT/${tag}_bench_L512.err
for b in 8 32; do python3 bench.py --no-cpu-baseline --no-shard-check --batch $b --steps 3 --warmup 1 > $OUT/${tag}_bench_b$b.json 2>/dev/null; done
python3 tools/bench_configs.py uni_2stage > $OUT/${tag}_cfg_uni_2stage.json 2>/dev/null
python3 tools/bench_configs.py mmu > $OUT/${tag}_cfg_mmu.json 2>/dev/null
for f in $OUT/${tag}_bench_L512.json $OUT/${tag}_bench_b8.json $OUT/${tag}_bench_b32.json $OUT/${tag}_cfg_uni_2stage.json $OUT/${tag}_cfg_mmu.json; do echo "== $f"; cut -c1-700 $f; done
