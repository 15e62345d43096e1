// Host-side launcher declarations for the HIP kernels (internal; the public surface is
// include/plangen_hip.h).  T = float (PG_F32) or bf16 (PG_BF16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

// ---------------------------------------------------------------- GEMM  C = A . W^T
// A operand description.  kind 0: row-major [M,K] with leading dim lda.
// kind 1: implicit im2col of a 3x3 / pad 1 convolution over an NHWC tensor
//         [B, Hi, Wi, Cin] (optionally nearest-2x upsampled on the fly: up=1), M = B*Ho*Wo,
//         K = 9*Cin ordered (tap, ci); Cin % 64 == 0.
// kind 2: same with stride 2 and the reference's asymmetric (0,1,0,1) padding
//         (Downsample, vq_model.py:440-447): Ho = Hi/2.
// Tuning knobs (pg_set_option) are PER HANDLE: every C-ABI entry point points the calling thread's ``pg_tune``
// at its handle's copy before it launches anything, so two handles in one process do not interfere.
struct PgTune {
    int split_small = 128, split_mid = 256, split_big = 128;   // decode split-K block-count targets at M < 48 / < 96 / >= 96
    int gemm256 = 1;                                            // 256x256 eight-phase MFMA GEMM for large shapes
    int conv_halo = 1;                                          // direct halo-tile 3x3 convolution (2: lock-step variant)
    int attn_waves = 0;                                         // 4 / 8 pin the decode-attention block size
    int attn_variant = 0;                                       // libplangen_diag.so only: which fused decode-attention form PgDiagHooks::attn_decode launches (0: production)
    int prefill_attn = 2;                                       // MFMA prefill attention: 2 = 128-query LDS-DMA / transpose-read kernel, 1 = 64-query kernel
    int gn_epilogue256 = 0;                                     // GroupNorm partial sums from the 256x256 kernel's conv epilogue (round 6; libplangen_diag.so switch).  Correct (GPU test) but its
                                                                // extra live registers spill 48 VGPRs into the K loop: +3.0 ms of convolutions for -2.7 ms of statistics passes.  Off.
    int vq_tail_fused = 0;                                      // 1: VQ decoder tail norm_out + swish + conv_out in one pass over the fp32 skip stream (bit-identical to gn_apply + conv_out).  Measured
                                                                // in round 6: 4.05 ms against 1.44 + 1.95 ms for the two passes (profiles/r06_c) -- off by default, kept for A/B
    int vq_argmin_multi = 1;                                    // VQ nearest-code search: 8 latent vectors per block (0: one per block, rounds 1-3)
    int vit_attn = 2;                                           // SigLIP attention: 2 = K / V^T of a head resident in LDS (round 4), 1 = 64-key tile kernel
    int ln_wave = 1;                                            // SigLIP LayerNorm: wave-per-row register kernel (0: generic block-per-row kernel)
    int wt_store = 0;                                           // v3 decode GEMM slabs with write-through (sc1) stores (libplangen_diag.so only)
    int sk3_xa = 1;                                             // wide-N decode GEMMs at 65..128 rows: x prefetch distance of the v3 8-wave block (2 = two chunks ahead + W ring 3: the x wait no
                                                                // longer retires the W ring; bit-identical results)
    int sk5 = 0;                                                // 1 (libplangen_diag.so only: the bf16 sums round differently): the v5 kernel for the wide-N decode GEMMs at 65..128 rows (n-tile pairs x
                                                                // two K halves, x by LDS-DMA; gemm_skinny.h).  Measured in round 6: on par in the microbenchmark, +80 ms per bs=64 step in the loop
                                                                // (profiles/r06_b) -- NOT the default; kept selectable for A/B
    int stream_gemm = -1;                                       // -1 auto, else bit mask: which decode GEMM classes run on the v4 LDS-DMA kernel (gemm.hip sk4_prod)
    const struct PgDiagHooks* diag = nullptr;                   // null in libplangen_hip.so; libplangen_diag.so (diag_api.hip) points it at its variant launchers
};
struct SeqState; struct GemmA; struct GemmEpi;
// Launch hooks only libplangen_diag.so fills in (measurement forms of production kernels; some produce WRONG results by construction).
// A hook returns true when it launched something in place of the production kernel.
struct PgDiagHooks {
    bool (*attn_decode)(hipStream_t s, bool is_bf16, const float* qkv, int S, long slab, void* obuf, void* kc, void* vc, const float* cos_t, const float* sin_t,
                        const SeqState& st, int M, int nh, int slots, int max_pos, float scale);
    bool (*gemm_big_wave)(hipStream_t s, const GemmA& a, const bf16* W, long ldb, const GemmEpi& e, int M, int N, int K, int batch, int batch2);
};
extern thread_local const PgTune* pg_tune;
struct GemmA {
    int kind = 0;
    const void* ptr = nullptr;
    long lda = 0, strideA = 0;          // strideA: per-batch (blockIdx.y) element stride
    long strideA2 = 0;                  // second batch dimension (e.g. attention heads)
    int Hi = 0, Wi = 0, Cin = 0, up = 0;
    const void* zeros = nullptr;        // >= 256 B of zeros (conv halo source)
    void* gn_part = nullptr;            // optional: GroupNorm partials of the OUTPUT (kernels that support it set *gn_nsplit > 0)
    int* gn_nsplit = nullptr;
    int gn_hw = 0;                      // pixels per image of the OUTPUT for kind == 0 launches that ask for partials (1x1 convolutions); convolutions derive it
};
// act == 3 (256x256 kernel, prefill QKV projection): RoPE(q), RoPE(k) and the KV-cache write straight from the accumulators.
// W must be the [8 | 8]-interleaved copy of Wqkv (launch_interleave_qk): n-tile t of a q / k head holds rotary columns 8t..8t+7 and
// 64+8t..64+8t+7, so lanes l and l ^ 32 hold x[j] and x[j+64] of the same token; v heads keep their column order.
struct RopeEpi {
    void* qbuf = nullptr; void* kc = nullptr; void* vc = nullptr;      // bf16: q [M][nh*128]; caches [R][nh][slots][128]
    const float* cos_t = nullptr; const float* sin_t = nullptr;        // [max_pos][64]
    const int32_t* tok_row = nullptr; const int32_t* tok_j = nullptr;  // packed token m -> (row, slot)
    const int32_t* pos_off = nullptr;                                  // [R] RoPE position of slot 0
    int nh = 0, slots = 0, max_pos = 0;
};
// Epilogue: v = acc*scale + bias_n[col] + bias_m[row] + residual[row,col]; act; store.
struct GemmEpi {
    void* out = nullptr;
    int out_f32 = 1;                    // 1: float* out, 0: T* out
    long ldc = 0, strideC = 0, strideC2 = 0;
    const float* bias_n = nullptr;
    const float* bias_m = nullptr;
    const void* residual = nullptr;     // T (or fp32 when res_f32), same ldc/stride layout as out unless ldr set
    int res_f32 = 0;
    long ldr = 0, strideR = 0;
    float scale = 1.f;
    int act = 0;                        // 0 none, 1 gelu(erf); 2 (256x256 kernel only) SwiGLU over [8 gate | 8 up] column blocks -> bf16 [M, N/2]; 3 (256x256 only) RoPE + KV write (rope)
    RopeEpi rope;
    // 256x256 kernel only (round 6; set by gemm256_try, never by callers): GroupNorm(32) partial sums of the stored tensor, one (sum, sum of squares)
    // per (image, 64-row chunk, group) in the layout gn_finalize_kernel reads: [B][gn_hw / 64][32][2].  gn_cpg = channels per group (4 / 8 / 16).
    float* gn_part = nullptr; int gn_hw = 0, gn_cpg = 0;
};
template <typename T>
void launch_gemm(hipStream_t s, const GemmA& a, const T* W, long ldb, long strideB,
                 const GemmEpi& e, int M, int N, int K, int batch, int batch2 = 1, long strideB2 = 0);

// 256x256 eight-phase kernel (gemm256.hip); returns false when the shape should stay on the 128x128 kernel.
bool gemm256_try(hipStream_t s, const GemmA& a, const bf16* W, long ldb, long strideB, const GemmEpi& e, int M, int N, int K,
                 int batch, int batch2, long strideB2);
void launch_attn_vit_flash(hipStream_t s, const bf16* qk, const bf16* vt, bf16* o, int B, int P, int C, int NH, float scale);
// direct 3x3 convolution with an LDS-resident input halo tile (conv_halo.hip), Cin = Cout = 128
// gn_part (optional, fp32 output only): per-(image, tile, group) GroupNorm partial sums of the stored tensor,
// layout [B][*gn_nsplit][32][2] -- the input gn_finalize_kernel expects.
bool conv_halo_try(hipStream_t s, const GemmA& a, const bf16* W, const GemmEpi& e, int M, int N, int K, float* gn_part = nullptr,
                   int* gn_nsplit = nullptr);
void launch_gn_finalize(hipStream_t s, const float* ws, float* stats, float* coef, const float* gamma, const float* beta, int B,
                        int nsplit, int HW, int C, float eps);
bool conv_out_halo_try(hipStream_t s, const bf16* x, const bf16* w, const float* bias, const bf16* zeros, void* out, int out_bf16,
                       int B, int H, int Wd, int Cin, int Cout);
// round 6: norm_out + swish + conv_out in one pass over the fp32 skip stream (coef = gn_finalize_kernel's per-(image, channel) affine coefficients)
bool conv_out_gn_try(hipStream_t s, const float* x_f32, const float* coef, const bf16* w, const float* bias, void* out, int out_bf16,
                     int B, int H, int Wd, int Cin, int Cout, int swish);

// Skinny weight-streaming GEMM (decode): x [M,K] bf16 (M <= 128 per launch block-row),
// W [N,K] bf16, out fp32 [S,M,N] split-K partial slabs (consumer kernels reduce over S).
// Returns S.  N % 16 == 0, K % 128 == 0.
int skinny_pick_splits(int N, int K);
int skinny_pick_splits(int N, int K, int M);
bool launch_gemm_skinny_swiglu(hipStream_t s, const bf16* x, const bf16* W, bf16* h, int M, int N, int K, const bf16* Wt = nullptr);
// Wt: optional tiled decode copy of W (launch_tile_weights); preferred when present
void launch_gemm_skinny(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, const bf16* Wt = nullptr);
// Deferred-1/rms RMSNorm (round 6, decode at 65..128 rows, bf16): the norm launch is a barrier-free elementwise pass that writes the residual
// stream, xw = bf16(x . w_norm) and 8 partial sums of squares per row (launch_rmsnorm_defer); the consumer GEMM applies 1/rms to its fp32 result.
bool deferred_norm_ok(int M, int N, int K, int S);
bool launch_gemm_skinny_deferred(hipStream_t s, const bf16* xw, const bf16* Wt, float* out, int M, int N, int K, int S, const float* ssq, float eps);
bool launch_gemm_skinny_swiglu_deferred(hipStream_t s, const bf16* xw, const bf16* Wt, bf16* h, int M, int N, int K, const float* ssq, float eps);
void launch_rmsnorm_defer(hipStream_t s, float* x, const float* partial, int S, long slab, const bf16* w, bf16* xw, float* ssq, int M, int H);
void launch_tile_weights(hipStream_t s, const bf16* src, bf16* dst, int N, int K);
// dst = Wqkv [3*nh*128, K] with the rows of every q / k head re-ordered [8 | 8] per 16-row tile (RopeEpi); v rows copied
void launch_interleave_qk(hipStream_t s, const bf16* src, bf16* dst, int nh, int K);

// ---------------------------------------------------------------- LLM elementwise / attention
// x (fp32 residual stream, in/out) += sum_s partial[s];  xn = w * (x * rsqrt(mean(x^2)+eps)).
// partial may be null (S=0).  xn may be null (residual update only).
template <typename T>
void launch_rmsnorm(hipStream_t s, float* x, const float* partial, int S, long slab, const T* w, T* xn,
                    int M, int H, float eps, int32_t* advance = nullptr);     // advance: *advance += 1 by one thread (decode step counter)

// gather rows: dst fp32 [n,H] = table[ids[idx]]  (table fp32)
void launch_embed_gather(hipStream_t s, const float* table, const int32_t* ids, const int32_t* src_idx, float* dst,
                         int n, int H, int vocab);
// byte-row copy with optional gather (src_idx) / scatter (dst_idx)
void launch_copy_rows(hipStream_t s, const void* src, const int32_t* src_idx, void* dst, const int32_t* dst_idx,
                      int n, long row_bytes);
// src (any of f32/bf16) rows -> fp32 dst with optional row index list (src_row[i] or i)
void launch_rows_to_f32(hipStream_t s, const void* src, int src_bf16, const int32_t* src_row, float* dst, int n, int H);
void launch_f32_to_rows(hipStream_t s, const float* src, void* dst, int dst_bf16, const int32_t* dst_row, int n, int H);
template <typename T>
void launch_t_to_rows(hipStream_t s, const T* src, void* dst, int dst_bf16, const int32_t* dst_row, int n, int H);

struct SeqState {              // device arrays describing the rows of the current batch
    const int32_t* len;        // [R] real prompt tokens of row r (KV slots filled by prefill)
    const int32_t* pos_off;    // [R] RoPE position of slot 0 (pad_len in position_mode 0, 0 in mode 1)
    const int32_t* n_dec;      // [1] decode steps completed so far
    const int32_t* tok_row;    // [Ntok] prefill: row of packed token t
    const int32_t* tok_j;      // [Ntok] prefill: slot of packed token t
    int shared_len;            // > 0: odd (uncond CFG) rows share one prompt; its K/V (slots [0, shared_len)) live in row shared_row only
    int shared_row;
    const int32_t* row_order;  // decode attention: blockIdx.y -> row, longest rows first (may be null)
};
// qkv partial fp32 [S, M, 3*nh*128] -> RoPE(q), RoPE(k); q -> qbuf T [M, nh*128];
// k,v -> caches [R][nh][slots][128].  mode 0: decode (token m = row m, slot = len+n_dec);
// mode 1: prefill (token m -> tok_row/tok_j).
template <typename T>
void launch_rope_kv(hipStream_t s, const float* qkv, int S, long slab, T* qbuf, T* kc, T* vc,
                    const float* cos_t, const float* sin_t, SeqState st, int mode, int M, int nh,
                    int slots, int max_pos);
// softmax(q.K^T * scale) V per (query, head).  mode as above: decode klen = len+n_dec+1;
// prefill klen = tok_j+1.
template <typename T>
void launch_attn(hipStream_t s, const T* qbuf, T* obuf, const T* kc, const T* vc, SeqState st, int mode,
                 int M, int nh, int slots, float scale);
// prefill: causal varlen flash attention on MFMA (bf16, head_dim 128); row_off[r] = first packed token of row r or -1
void launch_attn_prefill_flash(hipStream_t s, const bf16* qbuf, bf16* obuf, const bf16* kc, const bf16* vc,
                               const int32_t* row_off, const int32_t* len, int R, int max_len, int nh, int slots, float scale);
// decode step: RoPE + KV append + attention in one kernel (reads the QKV split-K slabs)
template <typename T>
void launch_attn_decode_fused(hipStream_t s, const float* qkv, int S, long slab, T* obuf, T* kc, T* vc,
                              const float* cos_t, const float* sin_t, SeqState st, int M, int nh, int slots,
                              int max_pos, float scale);
// h = silu(g) * u from gate-up partial fp32 [S, M, 2I] whose columns are interleaved in
// blocks of 8 (8 gate, 8 up, ...: every 16-column MFMA n-tile holds matching gate/up columns)
template <typename T>
void launch_silu_mul(hipStream_t s, const float* gu, int S, long slab, T* h, int M, int I);
// out T [M,N] = act(sum_s partial + bias)
template <typename T>
void launch_bias_act(hipStream_t s, const float* partial, int S, long slab, const float* bias, T* out,
                     int M, int N, int act);
// logits fp32 [M,N] = sum_s partial + bias
void launch_bias_f32(hipStream_t s, const float* partial, int S, long slab, const float* bias, float* out, int M, int N);

// Per-call sampling parameters live in DEVICE memory (written by a 1-thread kernel at the top of
// pg_decode_image_tokens), so the captured decode-step graph depends on shapes only and is replayed
// across calls with different seeds / temperatures / caller buffers.
struct SampleParams { float cfg_weight, temperature; uint64_t seed; int32_t T, has_force, has_mask, img_off; };   // img_off: global index of this engine's image 0 (prompt-sharded runs draw the same noise as one big batch)
struct SampleArgs {
    const float* logits_partial; int S; long slab; const float* bias; int V;
    const SampleParams* p;
    const int32_t* force_tok; const uint8_t* force_mask;          // [B,T] library-owned copies (valid when p->has_force / has_mask)
    int32_t* out_tok;                                             // [B,T] library-owned; copied to the caller at the end of the loop
    float* logits_out;                                            // [T,B,V] or null (tests)
    const float* embed_table;                                     // [V, H] fp32 (gen_embed->gen_aligner)
    float* x; int H;                                              // residual stream rows [2B, H]
    const int32_t* n_dec;
    int b_off, B_total;                                           // this launch covers images [b_off, b_off + gridDim) of B_total
};
void launch_set_sample_params(hipStream_t s, SampleParams* dst, SampleParams v);
// scratch: >= 16*B floats and ints
void launch_cfg_sample(hipStream_t s, const SampleArgs& a, int B, float* scratch_v, int* scratch_i);
// greedy text: argmax over vocab of (sum_s partial) per row + EOS bookkeeping, writes
// out[b, step] (int64) and next-token embedding into x.
struct TextParams { int32_t eos, min_new, max_new, pad; };
struct TextArgs {
    const float* logits_partial; int S; long slab; int V;
    const TextParams* p;
    int64_t* out;                                                 // [B, max_new] library-owned
    int32_t* unfinished; int32_t* any_unfinished;
    const float* embed_table; float* x; int H; const int32_t* n_dec;
};
void launch_set_text_params(hipStream_t s, TextParams* dst, TextParams v);
void launch_text_argmax(hipStream_t s, const TextArgs& a, int B, float* scratch_v, int* scratch_i);   // scratch: B * 16 entries each
void launch_advance(hipStream_t s, int32_t* n_dec);
void launch_rows_differ(hipStream_t s, const int32_t* ids, int L, int first, int stride, int ref, int n, int from, int32_t* flag);
void launch_uniform_from_bits(hipStream_t s, const uint64_t* z, float* out, int n);

// ---------------------------------------------------------------- VQ
// NHWC activations of type T.
template <typename T>
void launch_vq_gather(hipStream_t s, const T* table, const int32_t* codes, T* out, int n, int C, int vocab);
// GroupNorm(32, eps): statistics -> stats fp32 [B,32,2] = (mean, rstd) and, when coef != null,
// per-(image, channel) affine coefficients coef fp32 [B,C,2] = (rstd*gamma, beta - mean*rstd*gamma).
// ws: fp32 scratch >= B*32*2*nsplit
void launch_gn_stats(hipStream_t s, const void* x, int is_bf16, float* stats, float* ws, int B, int HW, int C, float eps,
                     float* coef, const float* gamma, const float* beta);
// y = x*a + sh, optional swish
template <typename TI, typename TO>
void launch_gn_apply(hipStream_t s, const TI* x, const float* coef, TO* y, int B, int HW, int C, int swish);
// softmax over rows of fp32 [rows, n] * scale -> T
template <typename T>
void launch_softmax_rows(hipStream_t s, const float* x, T* y, int rows, int n, float scale);
// 3x3 conv with tiny Cout (conv_out 128->3): NHWC T in -> NCHW out (fp32 or bf16)
template <typename T>
void launch_conv3x3_small(hipStream_t s, const T* x, const T* w /*[Cout][9][Cin]*/, const float* bias,
                          void* out, int out_bf16, int B, int H, int W, int Cin, int Cout);
// 3x3 conv with tiny Cin (encoder conv_in 3->128): NCHW in (f32/bf16) -> NHWC T
template <typename T>
void launch_conv3x3_in(hipStream_t s, const void* x, int x_bf16, const float* w /*[Cout][Cin][3][3]*/,
                       const float* bias, T* out, int B, int H, int W, int Cin, int Cout);
// nearest-code argmin: z fp32 [n, D] (D<=8) vs L2-normalised codebook fp32 [V, D] -> int64 idx
void launch_vq_argmin(hipStream_t s, const float* z, const float* codebook, int64_t* idx, int n, int D, int V);

// ---------------------------------------------------------------- SigLIP (a13)
// LayerNorm over the last dim of fp32 rows -> T
template <typename T>
void launch_layernorm(hipStream_t s, const float* x, const float* gamma, const float* beta, T* y, int M, int C, float eps);
// non-overlapping patches of NCHW images -> T [B*P, 3*ps*ps] with k = (c, py, px) (Conv2d weight order)
template <typename T>
void launch_patchify(hipStream_t s, const void* img, int img_bf16, T* out, int B, int S, int ps);
// x[b, p, :] += pos[p, :]
void launch_add_pos(hipStream_t s, float* x, const float* pos, int B, int P, int C);

// ---------------------------------------------------------------- weight conversion
// dst (T) [rows, cols] at row offset <- src fp32/bf16 [rows, cols] (device staging)
template <typename T>
void launch_convert(hipStream_t s, const void* src, int src_bf16, T* dst, long n);
// conv weight [Cout][Cin][kh][kw] -> [Cout][kh*kw][Cin]
template <typename T>
void launch_convert_conv(hipStream_t s, const void* src, int src_bf16, T* dst, int Cout, int Cin, int kk);
// gate/up rows interleaved in blocks of 8: src [I,H] -> dst rows (n/8)*16 + which*8 + n%8
template <typename T>
void launch_convert_interleave16(hipStream_t s, const void* src, int src_bf16, T* dst, int I, int H, int which);
void launch_to_f32(hipStream_t s, const void* src, int src_bf16, float* dst, long n);
void launch_l2norm_rows(hipStream_t s, const float* src, float* dst, int n, int D);
void launch_fill_zero(hipStream_t s, void* p, long bytes);
