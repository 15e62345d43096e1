// Internal declarations of libplangen_diag.so: measurement / forensics entry points and variant launchers that are NOT part of
// libplangen_hip.so (include/plangen_hip.h is the product's whole surface).  tools/, bench.py's instrumented pass and a few GPU tests load
// the diagnostics library through plangen_amd/_lib.py::load_diag().
#pragma once
#include "kernels.h"

// gemm.hip (production functions the variant table re-uses)
void launch_gemm_skinny_v1(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S);
bool launch_gemm_skinny_tiled_only(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S);
// diag_gemm.hip: variant table for the microbenchmarks; returns BK (0 = unsupported)
int launch_gemm_skinny_variant(hipStream_t s, int variant, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S);
struct SkFuse;
bool launch_gemm_skinny_fused_norm(hipStream_t s, const bf16* x, const bf16* Wt, float* slabs, int M, int N, int K, int S, const SkFuse* site_dev);
// diag_attn.hip: PgDiagHooks::attn_decode -- attn_variant 100 = the round-2 non-pipelined 7-deep kernel, 101-107 = timing ablations of the
// production kernel (results wrong by construction); anything else returns false (production kernel runs)
bool diag_attn_decode(hipStream_t s, bool is_bf16, const float* qkv, int S, long slab, void* obuf, void* kc, void* vc, const float* cos_t, const float* sin_t,
                      const SeqState& st, int M, int nh, int slots, int max_pos, float scale);
const PgDiagHooks* diag_hooks();
// diag_gemm_bw.hip: the "big-wave" 256x256 GEMM experiment (PgDiagHooks::gemm_big_wave; taken when pg_tune->gemm256 has bit 4 set)
bool gemm_bw_try(hipStream_t s, const GemmA& a, const bf16* W, long ldb, const GemmEpi& e, int M, int N, int K, int batch, int batch2);

// bench_kernels.hip: free-running weight-stream kernel (round 4's run-ahead prefetcher, measured 5-11 % SLOWER beside the decode loop,
// profiles/r04_b; kept as the background memory-load stressor of the LDS-DMA hazard screens)
struct PfMat { const void* base; uint32_t kib; uint32_t regions; };          // kib: size in KiB; regions: consumer blocks (order interleaves them)
struct PfLayer { PfMat m[4]; };
void launch_weight_prefetch(hipStream_t s, const PfLayer* plan_dev, int n_layers, const uint32_t* prog, int steps, int per_step,
                            int first, int stride, int blocks, int depth, int nt, uint32_t* stats);
