#include <cstdlib>
// Language-model side kernels (HBM-bound, wave64): fused split-K-reduce + residual + RMSNorm,
// RoPE + KV append, decode attention with per-lane-group online softmax, SwiGLU gate,
// gen_head activation, fused CFG-mix + argmax / Gumbel-max sampling + next-embedding gather.
// References: SURVEY.md K1-K9, a6.1-a6.4, a7-a9 (third-party transformers Llama formulas).
#include <type_traits>
#include "kernels.h"
#include "attn_decode.h"      // sum_slabs + the fused decode-attention kernel template


// ------------------------------------------------------------------------------- RMSNorm
// One 256-thread block per row.  x += sum_s partial[s]; xn = w * (x * rsqrt(mean(x^2)+eps)).
// LlamaRMSNorm: stats in fp32, normalised value cast to the input dtype (fp32 residual
// stream here) before the weight multiply; the product is rounded once to T.
// Register-resident fast path: H <= NV*1024, thread holds NV float4 (no LDS row buffer), all
// (S+1)*NV loads are independent and issued up front, weights / outputs moved as 8- or 16-byte
// vectors.
template <typename T, int NV, int SB = 4>
__global__ __launch_bounds__(256) void rmsnorm_kernel(float* __restrict__ x, const float* __restrict__ partial,
                                                     int S, long slab, const T* __restrict__ w,
                                                     T* __restrict__ xn, int H, float eps, int32_t* __restrict__ advance) {
    __shared__ float red[4];
    rmsnorm_row<T, NV, false, SB>(blockIdx.x, x, partial, S, slab, w, xn, H, eps, red);
    // the decode step's LAST kernel also advances the device step counter (no kernel of the step reads it after this
    // point; saves the 1-thread advance launch of every step)
    if (advance && blockIdx.x == 0 && threadIdx.x == 0) *advance += 1;
}
// 512 threads per row (H = 2048: one f32x4 per thread and slab): twice the waves issue the row's loads
template <typename T, int SB>
__global__ __launch_bounds__(512) void rmsnorm512_kernel(float* __restrict__ x, const float* __restrict__ partial, const T* __restrict__ w,
                                                        T* __restrict__ xn, int32_t* __restrict__ advance, int S, int slab, int H, float eps) {
    // 5 pointers + 4 x 32 bits = 56 bytes: the whole kernarg block is preloaded into SGPRs at wave launch (see attn_decode.h); slab (elements
    // of one split-K slab, M x N <= 2^31) travels as int for that
    __shared__ float red[8];
    rmsnorm_row<T, 1, false, SB, 512>(blockIdx.x, x, partial, S, (long)slab, w, xn, H, eps, red);
    if (advance && blockIdx.x == 0 && threadIdx.x == 0) *advance += 1;
}
template <typename T>
void launch_rmsnorm(hipStream_t s, float* x, const float* partial, int S, long slab, const T* w, T* xn,
                    int M, int H, float eps, int32_t* advance) {
    if (M <= 0) return;
    // rmsnorm512_kernel carries the slab stride as a 32-bit int (56-byte preloaded kernarg block): a stride that does not fit -- M x H >= 2^31,
    // i.e. >= 2^20 rows at H = 2048, only reachable through pg_op_rmsnorm -- takes the long-stride kernel below
    if (H == 2048 && (S == 0 || slab <= 0x7fffffffL)) {   // every row count (the reduction order must not depend on M: sharded == unsharded tokens); decode loop -8 ms at bs=64, -7 ms at bs=8 vs 256 threads per row
        // S == 0 (prefill: o / down add into the residual stream in their GEMM epilogues, 13 k rows per launch): no slab loads at all -- the
        // branch-free decode form would read every row four more times (-6 us of 34 per launch at the bench's packed batch)
        if (S == 0) hipLaunchKernelGGL((rmsnorm512_kernel<T, 0>), dim3(M), dim3(512), 0, s, x, partial, w, xn, advance, S, (int)slab, H, eps);
        else if (S > 4) hipLaunchKernelGGL((rmsnorm512_kernel<T, 8>), dim3(M), dim3(512), 0, s, x, partial, w, xn, advance, S, (int)slab, H, eps);
        else hipLaunchKernelGGL((rmsnorm512_kernel<T, 4>), dim3(M), dim3(512), 0, s, x, partial, w, xn, advance, S, (int)slab, H, eps);
        return;
    }
    if (H <= 1024) hipLaunchKernelGGL((rmsnorm_kernel<T, 1>), dim3(M), dim3(256), 0, s, x, partial, S, slab, w, xn, H, eps, advance);
    else if (H <= 2048 && S > 4) hipLaunchKernelGGL((rmsnorm_kernel<T, 2, 8>), dim3(M), dim3(256), 0, s, x, partial, S, slab, w, xn, H, eps, advance);   // 5-8 slabs (small row counts): still one round trip
    else if (H <= 2048) hipLaunchKernelGGL((rmsnorm_kernel<T, 2>), dim3(M), dim3(256), 0, s, x, partial, S, slab, w, xn, H, eps, advance);
    else if (H <= 4096) hipLaunchKernelGGL((rmsnorm_kernel<T, 4>), dim3(M), dim3(256), 0, s, x, partial, S, slab, w, xn, H, eps, advance);
    else hipLaunchKernelGGL((rmsnorm_kernel<T, 8>), dim3(M), dim3(256), 0, s, x, partial, S, slab, w, xn, H, eps, advance);
}
// Deferred-1/rms form of the decode norm (round 6; H = 2048, bf16): NO block reduction, no barrier.  Block = 256 threads = half a row, every thread
// one f32x4 of x and of each slab (all loads up front, added in the SAME order as rmsnorm_row: the residual stream keeps its bits), stores the
// updated residual, xw = bf16(x . w) -- the consumer GEMM's A operand WITHOUT the 1/rms -- and each WAVE leaves the sum of squares of its 256
// columns in ssq[row][half * 4 + wave] (8 partials per row, summed in fixed order by the consumer: deterministic, no atomics).  2 M blocks instead
// of M: twice the waves have the row's loads in flight, and the kernel's tail is a store, not a reduction.
template <int SB>
__global__ __launch_bounds__(256) void rmsnorm_defer_kernel(float* __restrict__ x, const float* __restrict__ partial, const bf16* __restrict__ w,
                                                           bf16* __restrict__ xw, float* __restrict__ ssq, int S, int slab, int H) {
    const int tid = threadIdx.x, m = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int i = half * 1024 + tid * 4;
    float* xr = x + (long)m * H + i;
    const u32x2 wv = *(const u32x2*)(w + i);
    f32x4 v = *(const f32x4*)xr;
    f32x4 t[SB];
    const float* pp = partial + (long)m * H + i;
    const int smax = S - 1;
#pragma unroll
    for (int u = 0; u < SB; ++u) t[u] = *(const f32x4*)(pp + (long)(u < smax ? u : smax) * slab);
#pragma unroll
    for (int u = 0; u < SB; ++u) if (u < S) v += t[u];
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    *(f32x4*)xr = v;
    u32x2 ov; ov.x = pack_bf16x2(bf16_lo(wv.x) * v.x, bf16_hi(wv.x) * v.y); ov.y = pack_bf16x2(bf16_lo(wv.y) * v.z, bf16_hi(wv.y) * v.w);
    *(u32x2*)(xw + (long)m * H + i) = ov;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) ssq[m * 8 + half * 4 + (tid >> 6)] = ss;
}
void launch_rmsnorm_defer(hipStream_t s, float* x, const float* partial, int S, long slab, const bf16* w, bf16* xw, float* ssq, int M, int H) {
    // contract (checked by the caller through deferred_norm_ok + the engine): H == 2048, 1 <= S <= 8, slab fits 32 bits
    if (S > 4) hipLaunchKernelGGL((rmsnorm_defer_kernel<8>), dim3(2 * M), dim3(256), 0, s, x, partial, w, xw, ssq, S, (int)slab, H);
    else hipLaunchKernelGGL((rmsnorm_defer_kernel<4>), dim3(2 * M), dim3(256), 0, s, x, partial, w, xw, ssq, S, (int)slab, H);
}
template void launch_rmsnorm<float>(hipStream_t, float*, const float*, int, long, const float*, float*, int, int, float, int32_t*);
template void launch_rmsnorm<bf16>(hipStream_t, float*, const float*, int, long, const bf16*, bf16*, int, int, float, int32_t*);

// ------------------------------------------------------------------------------- row movers
// dst[t] = table[ids[src_idx ? src_idx[t] : t]]   (K1; packed left-pad-free gather)
__global__ void embed_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                    const int32_t* __restrict__ src_idx, float* __restrict__ dst, int H, int vocab) {
    const int t = blockIdx.x;
    int id = ids[src_idx ? src_idx[t] : t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* src = table + (long)id * H;
    for (int i = threadIdx.x * 4; i < H; i += blockDim.x * 4)
        *(f32x4*)(dst + (long)t * H + i) = *(const f32x4*)(src + i);
}
void launch_embed_gather(hipStream_t s, const float* table, const int32_t* ids, const int32_t* src_idx, float* dst,
                         int n, int H, int vocab) {
    if (n <= 0) return;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(n), dim3(256), 0, s, table, ids, src_idx, dst, H, vocab);
}
// byte-row copy with optional gather (src_idx) / scatter (dst_idx); row_bytes % 4 == 0
__global__ void copy_rows_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ src_idx,
                                 uint32_t* __restrict__ dst, const int32_t* __restrict__ dst_idx, int row_words) {
    const int t = blockIdx.x;
    const long rs = src_idx ? src_idx[t] : t, rd = dst_idx ? dst_idx[t] : t;
    for (int i = threadIdx.x; i < row_words; i += blockDim.x) dst[rd * row_words + i] = src[rs * row_words + i];
}
void launch_copy_rows(hipStream_t s, const void* src, const int32_t* src_idx, void* dst, const int32_t* dst_idx,
                      int n, long row_bytes) {
    if (n <= 0) return;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(n), dim3(256), 0, s, (const uint32_t*)src, src_idx, (uint32_t*)dst, dst_idx, (int)(row_bytes / 4));
}

__global__ void rows_to_f32_kernel(const void* __restrict__ src, int src_bf16, const int32_t* __restrict__ src_row,
                                   float* __restrict__ dst, int H) {
    const int t = blockIdx.x;
    const long r = src_row ? src_row[t] : t;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        float v = src_bf16 ? ET<bf16>::ld((const bf16*)src + r * H + i) : ((const float*)src)[r * H + i];
        dst[(long)t * H + i] = v;
    }
}
void launch_rows_to_f32(hipStream_t s, const void* src, int src_bf16, const int32_t* src_row, float* dst, int n, int H) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rows_to_f32_kernel, dim3(n), dim3(256), 0, s, src, src_bf16, src_row, dst, H);
}
template <typename TS>
__global__ void t_to_rows_kernel(const TS* __restrict__ src, void* __restrict__ dst, int dst_bf16,
                                 const int32_t* __restrict__ dst_row, int H) {
    const int t = blockIdx.x;
    const long r = dst_row ? dst_row[t] : t;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        const float v = ET<TS>::ld(src + (long)t * H + i);
        if (dst_bf16) ET<bf16>::st((bf16*)dst + r * H + i, v); else ((float*)dst)[r * H + i] = v;
    }
}
void launch_f32_to_rows(hipStream_t s, const float* src, void* dst, int dst_bf16, const int32_t* dst_row, int n, int H) {
    if (n <= 0) return;
    hipLaunchKernelGGL(t_to_rows_kernel<float>, dim3(n), dim3(256), 0, s, src, dst, dst_bf16, dst_row, H);
}
template <typename T>
void launch_t_to_rows(hipStream_t s, const T* src, void* dst, int dst_bf16, const int32_t* dst_row, int n, int H) {
    if (n <= 0) return;
    hipLaunchKernelGGL(t_to_rows_kernel<T>, dim3(n), dim3(256), 0, s, src, dst, dst_bf16, dst_row, H);
}
template void launch_t_to_rows<float>(hipStream_t, const float*, void*, int, const int32_t*, int, int);
template void launch_t_to_rows<bf16>(hipStream_t, const bf16*, void*, int, const int32_t*, int, int);

// flag |= any(ids[row][from..L) != ids[ref][from..L)) over rows first, first+stride, ... (n rows): the
// batch-constant negative prompt probe of pg_prefill (plangen_base.py:673-686 builds every uncond row from
// the same neg_prompt; verified, not assumed).
__global__ void rows_differ_kernel(const int32_t* __restrict__ ids, int L, int first, int stride, int ref, int from, int32_t* __restrict__ flag) {
    const int32_t* a = ids + (long)(first + blockIdx.x * stride) * L;
    const int32_t* b = ids + (long)ref * L;
    int diff = 0;
    for (int j = from + threadIdx.x; j < L; j += blockDim.x) diff |= a[j] != b[j];
    if (diff) atomicOr(flag, 1);
}
void launch_rows_differ(hipStream_t s, const int32_t* ids, int L, int first, int stride, int ref, int n, int from, int32_t* flag) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rows_differ_kernel, dim3(n), dim3(256), 0, s, ids, L, first, stride, ref, from, flag);
}

// ------------------------------------------------------------------------------- RoPE + KV append
// grid (M tokens, nh/4), 256 threads = 4 heads x 64 rotation pairs.  rotate_half form
// (half-split, not interleaved): out[j] = x[j]cos - x[j+64]sin, out[j+64] = x[j+64]cos + x[j]sin.
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(const float* __restrict__ qkv, int S, long slab,
                                                     T* __restrict__ qbuf, T* __restrict__ kc, T* __restrict__ vc,
                                                     const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                     SeqState st, int mode, int nh, int slots, int max_pos) {
    const int m = blockIdx.x, head = blockIdx.y * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (head >= nh) return;
    int row, slot;
    if (mode == 0) { row = m; slot = st.len[m] + *st.n_dec; }
    else { row = st.tok_row[m]; slot = st.tok_j[m]; }
    int pos = st.pos_off[row] + slot;
    if (pos >= max_pos) pos = max_pos - 1;
    if (slot >= slots) return;                                     // capacity guard (host checks too)
    const int HD = nh * 128;
    const long base = (long)m * 3 * HD + head * 128 + j;
    float a6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int o6[6] = {0, 64, HD, HD + 64, 2 * HD, 2 * HD + 64};
    sum_slabs<6>(qkv + base, slab, S, o6, a6);
    const float q0 = a6[0], q1 = a6[1], k0 = a6[2], k1 = a6[3], v0 = a6[4], v1 = a6[5];
    const float c = cos_t[(long)pos * 64 + j], sn = sin_t[(long)pos * 64 + j];
    T* qo = qbuf + (long)m * HD + head * 128 + j;
    ET<T>::st(qo, rope_lo(q0, q1, c, sn));
    ET<T>::st(qo + 64, rope_hi(q0, q1, c, sn));
    const long co = (((long)row * nh + head) * slots + slot) * 128 + j;
    ET<T>::st(kc + co, rope_lo(k0, k1, c, sn));
    ET<T>::st(kc + co + 64, rope_hi(k0, k1, c, sn));
    ET<T>::st(vc + co, v0);
    ET<T>::st(vc + co + 64, v1);
}
template <typename T>
void launch_rope_kv(hipStream_t s, const float* qkv, int S, long slab, T* qbuf, T* kc, T* vc,
                    const float* cos_t, const float* sin_t, SeqState st, int mode, int M, int nh,
                    int slots, int max_pos) {
    if (M <= 0) return;
    hipLaunchKernelGGL(rope_kv_kernel<T>, dim3(M, (nh + 3) / 4), dim3(256), 0, s, qkv, S, slab, qbuf, kc, vc,
                       cos_t, sin_t, st, mode, nh, slots, max_pos);
}
template void launch_rope_kv<float>(hipStream_t, const float*, int, long, float*, float*, float*, const float*, const float*, SeqState, int, int, int, int, int);
template void launch_rope_kv<bf16>(hipStream_t, const float*, int, long, bf16*, bf16*, bf16*, const float*, const float*, SeqState, int, int, int, int, int);

// ------------------------------------------------------------------------------- attention
// One 256-thread block per (query, head); head_dim = 128.  HBM-bound streaming of that
// (row, head)'s K and V (contiguous [slot][128]).  Each 16-byte vector load covers EPV
// elements of one key; LPK = 128/EPV lanes cover a key, so one wave load instruction covers
// KPI = 64/LPK keys (bf16: 4 keys, fp32: 2).  Every LPK-lane group runs its OWN online
// softmax over the keys it sees (no cross-lane max exchange inside the loop); the
// 4 waves x KPI groups partial states (m, l, o[128]) are merged once through LDS.
template <typename T, int UN, bool NT>
__global__ __launch_bounds__(256) void attn_kernel(const T* __restrict__ qbuf, T* __restrict__ obuf,
                                                  const T* __restrict__ kc, const T* __restrict__ vc,
                                                  SeqState st, int mode, int nh, int slots, float scale) {
    constexpr int EPV = ET<T>::EPV, LPK = 128 / EPV, KPI = 64 / LPK, NST = 4 * KPI;
    __shared__ float s_o[NST][128];
    __shared__ float s_m[NST], s_l[NST];
    const int qi = blockIdx.x, head = blockIdx.y;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int grp = l / LPK, lk = l % LPK;                // key group in wave, lane within key
    int row, klen;
    if (mode == 0) { row = qi; klen = st.len[qi] + *st.n_dec + 1; }
    else { row = st.tok_row[qi]; klen = st.tok_j[qi] + 1; }
    if (klen > slots) klen = slots;
    const int HD = nh * 128;
    float q[EPV];
    {
        const u32x4 qv = *(const u32x4*)(qbuf + (long)qi * HD + head * 128 + lk * EPV);
        ET<T>::unpack(qv, q);
#pragma unroll
        for (int e = 0; e < EPV; ++e) q[e] *= scale;
    }
    const T* kb = kc + ((long)row * nh + head) * slots * 128 + lk * EPV;
    const T* vb = vc + ((long)row * nh + head) * slots * 128 + lk * EPV;
    float m_run = -INFINITY, l_run = 0.f, o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = 0.f;

    constexpr int KPW = KPI * UN;                          // keys per wave iteration
    for (int base = w * KPW; base < klen; base += 4 * KPW) {
        u32x4 kv[UN], vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = base + u * KPI + grp;
            key = key < klen ? key : klen - 1;             // clamp: load stays in bounds
            kv[u] = NT ? __builtin_nontemporal_load((const u32x4*)(kb + (long)key * 128)) : *(const u32x4*)(kb + (long)key * 128);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = base + u * KPI + grp;
            key = key < klen ? key : klen - 1;
            vv[u] = NT ? __builtin_nontemporal_load((const u32x4*)(vb + (long)key * 128)) : *(const u32x4*)(vb + (long)key * 128);
        }
        float sc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float kf[EPV]; ET<T>::unpack(kv[u], kf);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) d = fmaf(q[e], kf[e], d);
#pragma unroll
            for (int o_ = LPK / 2; o_ > 0; o_ >>= 1) d += __shfl_xor(d, o_, 64);
            sc[u] = (base + u * KPI + grp < klen) ? d : -INFINITY;
        }
        float mx = m_run;
#pragma unroll
        for (int u = 0; u < UN; ++u) mx = fmaxf(mx, sc[u]);
        if (mx > -INFINITY) {
            const float alpha = __expf(m_run - mx);        // m_run=-inf -> 0
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < EPV; ++e) o[e] *= alpha;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float p = __expf(sc[u] - mx);        // -inf -> 0
                l_run += p;
                float vf[EPV]; ET<T>::unpack(vv[u], vf);
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = fmaf(p, vf[e], o[e]);
            }
            m_run = mx;
        }
    }
    const int stt = w * KPI + grp;
#pragma unroll
    for (int e = 0; e < EPV; ++e) s_o[stt][lk * EPV + e] = o[e];
    if (lk == 0) { s_m[stt] = m_run; s_l[stt] = l_run; }
    __syncthreads();
    if (tid < 128) {
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < NST; ++i) M = fmaxf(M, s_m[i]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const float f = (s_m[i] > -INFINITY) ? __expf(s_m[i] - M) : 0.f;
            num = fmaf(f, s_o[i][tid], num);
            den = fmaf(f, s_l[i], den);
        }
        ET<T>::st(obuf + (long)qi * HD + head * 128 + tid, num / den);
    }
}
template <typename T>
void launch_attn_decode_fused(hipStream_t s, const float* qkv, int S, long slab, T* obuf, T* kc, T* vc,
                              const float* cos_t, const float* sin_t, SeqState st, int M, int nh, int slots,
                              int max_pos, float scale) {
    if (M <= 0) return;
    // The software-pipelined loop (template argument ABL = 16): 8-wave blocks 5 deep when the launch has few (row, head) blocks (small batch:
    // 2x the K/V bytes in flight per CU), 4-wave blocks 6 deep otherwise (7 deep spills in the pipelined form).  Only these two forms are
    // instantiated in libplangen_hip.so; the older non-pipelined kernel and the timing ablations live in libplangen_diag.so (diag_attn.hip),
    // which registers itself in PgTune::diag.
    if (pg_tune->diag && pg_tune->diag->attn_decode && pg_tune->diag->attn_decode(s, std::is_same<T, bf16>::value, qkv, S, slab, obuf, kc, vc, cos_t, sin_t, st, M, nh, slots, max_pos, scale)) return;
#define ATT_LAUNCH(U, W, A) hipLaunchKernelGGL((attn_decode_fused_kernel<T, U, W, A>), dim3(nh, M), dim3(64 * W), 0, s, st.row_order, st.len, st.n_dec, kc, vc, nh, slots, st.shared_len, st.shared_row, qkv, slab, obuf, cos_t, sin_t, st.pos_off, S, max_pos, scale)
    const bool small = (M * nh <= 512 && pg_tune->attn_waves != 4) || pg_tune->attn_waves == 8;
    if (small) ATT_LAUNCH(5, 8, 16);
    else ATT_LAUNCH(6, 4, 16);
#undef ATT_LAUNCH
}
template void launch_attn_decode_fused<float>(hipStream_t, const float*, int, long, float*, float*, float*, const float*, const float*, SeqState, int, int, int, int, float);
template void launch_attn_decode_fused<bf16>(hipStream_t, const float*, int, long, bf16*, bf16*, bf16*, const float*, const float*, SeqState, int, int, int, int, float);

template <typename T>
void launch_attn(hipStream_t s, const T* qbuf, T* obuf, const T* kc, const T* vc, SeqState st, int mode,
                 int M, int nh, int slots, float scale) {
    if (M <= 0) return;
    dim3 grid(M, nh), block(256);
    // decode (mode 0): 8 keys per lane group in flight, non-temporal K/V loads (read once per step); prefill: cached loads
    if (mode == 0) hipLaunchKernelGGL((attn_kernel<T, 8, true>), grid, block, 0, s, qbuf, obuf, kc, vc, st, mode, nh, slots, scale);
    else hipLaunchKernelGGL((attn_kernel<T, 8, false>), grid, block, 0, s, qbuf, obuf, kc, vc, st, mode, nh, slots, scale);
}
template void launch_attn<float>(hipStream_t, const float*, float*, const float*, const float*, SeqState, int, int, int, int, float);
template void launch_attn<bf16>(hipStream_t, const bf16*, bf16*, const bf16*, const bf16*, SeqState, int, int, int, int, float);

// ------------------------------------------------------------------------------- SwiGLU gate
// gu columns are interleaved in blocks of 8: [8 gate | 8 up] per 16 columns (= one MFMA n-tile).
template <typename T>
__global__ void silu_mul_kernel(const float* __restrict__ gu, int S, long slab, T* __restrict__ h, int I) {
    const int m = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= I) return;
    const long cg = (long)m * 2 * I + (n >> 3) * 16 + (n & 7);
    float a2[2] = {0.f, 0.f};
    const int o2[2] = {0, 8};
    sum_slabs<2>(gu + cg, slab, S, o2, a2);
    const float g = a2[0], u = a2[1];
    ET<T>::st(h + (long)m * I + n, (g / (1.f + expf(-g))) * u);
}
template <typename T>
void launch_silu_mul(hipStream_t s, const float* gu, int S, long slab, T* h, int M, int I) {
    if (M <= 0) return;
    hipLaunchKernelGGL(silu_mul_kernel<T>, dim3((I + 255) / 256, M), dim3(256), 0, s, gu, S, slab, h, I);
}
template void launch_silu_mul<float>(hipStream_t, const float*, int, long, float*, int, int);
template void launch_silu_mul<bf16>(hipStream_t, const float*, int, long, bf16*, int, int);

template <typename T>
__global__ void bias_act_kernel(const float* __restrict__ partial, int S, long slab, const float* __restrict__ bias,
                                T* __restrict__ out, int N, int act) {
    const int m = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float a1[1] = {bias ? bias[n] : 0.f};
    const int o1[1] = {0};
    sum_slabs<1>(partial + (long)m * N + n, slab, S, o1, a1);
    float v = a1[0];
    if (act == 1) v = gelu_erf(v);
    ET<T>::st(out + (long)m * N + n, v);
}
template <typename T>
void launch_bias_act(hipStream_t s, const float* partial, int S, long slab, const float* bias, T* out,
                     int M, int N, int act) {
    if (M <= 0) return;
    hipLaunchKernelGGL(bias_act_kernel<T>, dim3((N + 255) / 256, M), dim3(256), 0, s, partial, S, slab, bias, out, N, act);
}
template void launch_bias_act<float>(hipStream_t, const float*, int, long, const float*, float*, int, int, int);
template void launch_bias_act<bf16>(hipStream_t, const float*, int, long, const float*, bf16*, int, int, int);
void launch_bias_f32(hipStream_t s, const float* partial, int S, long slab, const float* bias, float* out, int M, int N) {
    launch_bias_act<float>(s, partial, S, slab, bias, out, M, N, 0);
}

// ------------------------------------------------------------------------------- CFG + sample
// One block per image b (rows 2b = cond, 2b+1 = uncond).  mixed = u + w (c - u)
// (plangen_base.py:587); greedy: first index of the max (torch.argmax tie rule);
// temperature > 0: Gumbel-max == multinomial(softmax(mixed / T)).  Then the chosen (or
// forced) token's precomputed gen_aligner(gen_embed(tok)) row is written to both CFG rows
// of the residual stream (plangen_base.py:602-604).
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
// Stage 1: grid (CFG_CHUNKS, B): each block scans V/CFG_CHUNKS logits of image b and leaves its
// local (value, index) winner; stage 2 (one block per image) combines the winners, applies the
// forcing rules and gathers the next embedding.
#define CFG_CHUNKS 16
__global__ __launch_bounds__(256) void cfg_scan_kernel(SampleArgs a, float* __restrict__ pv, int* __restrict__ pi) {
    __shared__ float sv[4]; __shared__ int si[4];
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, step = *a.n_dec;
    const int bg = b + a.b_off, B = a.B_total;            // global image index (lane-independent results)
    const int per = (a.V + CFG_CHUNKS - 1) / CFG_CHUNKS, v0 = ch * per, v1 = min(a.V, v0 + per);
    const long rc = (long)(2 * b) * a.V, ru = (long)(2 * b + 1) * a.V;
    float best = -INFINITY; int bi = 0x7fffffff;
    const float temperature = a.p->temperature, cfg_weight = a.p->cfg_weight;
    const uint64_t seed = a.p->seed;
    const float invT = temperature > 0.f ? 1.f / temperature : 1.f;
    // four vocabulary entries per thread per pass, ALL their loads (bias, cond / uncond row of every slab) issued before the first use
    // (indices clamped into the chunk): one memory round trip per pass instead of one per entry
    constexpr int IT = 4;
    const float* bias0 = a.bias ? a.bias : a.logits_partial + rc;     // no bias: any mapped address, value discarded
    const int ocu[2] = {0, a.V};                                      // cond row, uncond row (adjacent rows)
    for (int vb = v0 + tid; vb < v1; vb += IT * 256) {
        float bb[IT], cu[IT][2];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = vb + it * 256, vc = v < v1 ? v : v1 - 1;
            bb[it] = bias0[vc];
        }
        float t[IT][4][2];
        const int smax = a.S > 0 ? a.S - 1 : 0;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = vb + it * 256, vc = v < v1 ? v : v1 - 1;
            const float* pp = a.logits_partial + rc + vc;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* q = pp + (long)(u < smax ? u : smax) * a.slab;       // slab index clamped: branch-free loads
                t[it][u][0] = q[0]; t[it][u][1] = q[a.V];
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = vb + it * 256, vc = v < v1 ? v : v1 - 1;
            cu[it][0] = a.bias ? bb[it] : 0.f; cu[it][1] = cu[it][0];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (u < a.S) { cu[it][0] += t[it][u][0]; cu[it][1] += t[it][u][1]; }
            if (a.S > 4) sum_slabs<2>(a.logits_partial + rc + vc + 4 * a.slab, a.slab, a.S - 4, ocu, cu[it]);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = vb + it * 256;
            if (v < v1) {
                const float c = cu[it][0], u = cu[it][1];
                float mixed = u + cfg_weight * (c - u);
                if (a.logits_out) a.logits_out[((long)step * B + bg) * a.V + v] = mixed;
                if (temperature > 0.f) {
                    const float uu = rng_uniform(seed, (uint64_t)(bg + a.p->img_off) * 1000003ull + step, v);
                    mixed = mixed * invT - __logf(-__logf(uu));
                }
                if (mixed > best) { best = mixed; bi = v; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        argmax_combine(best, bi, ov, oi);
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        float v = sv[0]; int i = si[0];
        for (int k = 1; k < 4; ++k) argmax_combine(v, i, sv[k], si[k]);
        pv[b * CFG_CHUNKS + ch] = v; pi[b * CFG_CHUNKS + ch] = i;
    }
}
__global__ __launch_bounds__(256) void cfg_pick_kernel(SampleArgs a, const float* __restrict__ pv, const int* __restrict__ pi) {
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, step = *a.n_dec;
    if (tid == 0) {
        float v = pv[b * CFG_CHUNKS]; int i = pi[b * CFG_CHUNKS];
        for (int k = 1; k < CFG_CHUNKS; ++k) argmax_combine(v, i, pv[b * CFG_CHUNKS + k], pi[b * CFG_CHUNKS + k]);
        i = i < 0 ? 0 : (i >= a.V ? a.V - 1 : i);
        int emit = i, feed = i;
        const long bg = b + a.b_off;
        const int T = a.p->T;
        if (step < T && a.p->has_force) {
            const int f = a.force_tok[bg * T + step];
            if (a.p->has_mask) { if (a.force_mask[bg * T + step] == 0) { emit = f; feed = f; } }
            else feed = f;
        }
        if (step < T) a.out_tok[bg * T + step] = emit;
        feed = feed < 0 ? 0 : (feed >= a.V ? a.V - 1 : feed);
        s_tok = feed;
    }
    __syncthreads();
    const float* src = a.embed_table + (long)s_tok * a.H;
    float* x0 = a.x + (long)(2 * b) * a.H;
    for (int i = tid * 4; i < a.H; i += 1024) {
        const f32x4 v = *(const f32x4*)(src + i);
        *(f32x4*)(x0 + i) = v;
        *(f32x4*)(x0 + a.H + i) = v;
    }
}
__global__ void set_sample_params_kernel(SampleParams* dst, SampleParams v) { *dst = v; }
void launch_set_sample_params(hipStream_t s, SampleParams* dst, SampleParams v) { hipLaunchKernelGGL(set_sample_params_kernel, dim3(1), dim3(1), 0, s, dst, v); }
__global__ void set_text_params_kernel(TextParams* dst, TextParams v) { *dst = v; }
void launch_set_text_params(hipStream_t s, TextParams* dst, TextParams v) { hipLaunchKernelGGL(set_text_params_kernel, dim3(1), dim3(1), 0, s, dst, v); }
void launch_cfg_sample(hipStream_t s, const SampleArgs& a, int B, float* scratch_v, int* scratch_i) {
    hipLaunchKernelGGL(cfg_scan_kernel, dim3(CFG_CHUNKS, B), dim3(256), 0, s, a, scratch_v, scratch_i);
    hipLaunchKernelGGL(cfg_pick_kernel, dim3(B), dim3(256), 0, s, a, scratch_v, scratch_i);
}

// greedy text token (HF generate, do_sample=False): argmax over vocab, finished rows emit
// eos, unfinished &= (tok != eos).  Two stages like the image sampler: grid (CFG_CHUNKS, B) scans V/CFG_CHUNKS logits with
// 16-byte loads (split-K slabs summed in slab order), one block per row combines the winners (lowest index on ties =
// torch.argmax), does the EOS bookkeeping and gathers the next embedding.  (One block per row over 102 400 logits with
// scalar loads was 160 us per step at 64 rows; this pair is ~10.)
__global__ __launch_bounds__(256) void text_scan_kernel(TextArgs a, float* __restrict__ pv, int* __restrict__ pi) {
    __shared__ float sv[4]; __shared__ int si[4];
    const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, step = *a.n_dec;
    const int ban = step < a.p->min_new ? a.p->eos : -1;
    const int chunk = ((a.V + CFG_CHUNKS * 4 - 1) / (CFG_CHUNKS * 4)) * 4;
    const int v0 = ch * chunk, v1 = v0 + chunk < a.V ? v0 + chunk : a.V;
    const float* lp = a.logits_partial + (long)b * a.V;
    float best = -INFINITY; int bi = 0x7fffffff;
    if (((a.V | (int)(a.slab & 3)) & 3) == 0) {
        // eight independent 16-byte loads per slab in flight per thread (addresses clamped into the chunk, validity applied to the
        // compare): a plain `for v` / `for s` nest is one dependent round trip per vector and slab
        constexpr int IT = 8;
        for (int vb = v0 + tid * 4; vb < v1; vb += IT * 1024) {
            f32x4 c[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int v = vb + it * 1024;
                c[it] = *(const f32x4*)(lp + (v < v1 ? v : v1 - 4));
            }
            for (int s = 1; s < a.S; ++s) {
                f32x4 t[IT];
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int v = vb + it * 1024;
                    t[it] = *(const f32x4*)(lp + (long)s * a.slab + (v < v1 ? v : v1 - 4));
                }
#pragma unroll
                for (int it = 0; it < IT; ++it) c[it] += t[it];
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int v = vb + it * 1024;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float cj = (v + j == ban || v >= v1) ? -INFINITY : c[it][j];
                    if (cj > best) { best = cj; bi = v + j; }
                }
            }
        }
    } else {
        for (int v = v0 + tid; v < v1; v += 256) {
            float c = 0.f;
            for (int s = 0; s < a.S; ++s) c += lp[(long)s * a.slab + v];
            if (v == ban) c = -INFINITY;
            if (c > best) { best = c; bi = v; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        argmax_combine(best, bi, ov, oi);
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        float v = sv[0]; int i = si[0];
        for (int k = 1; k < 4; ++k) argmax_combine(v, i, sv[k], si[k]);
        pv[b * CFG_CHUNKS + ch] = v; pi[b * CFG_CHUNKS + ch] = i;
    }
}
__global__ __launch_bounds__(256) void text_argmax_kernel(TextArgs a, const float* __restrict__ pv, const int* __restrict__ pi) {
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, step = *a.n_dec;
    if (tid == 0) {
        const int eos = a.p->eos, max_new = a.p->max_new;
        float v = pv[b * CFG_CHUNKS]; int i = pi[b * CFG_CHUNKS];
        for (int k = 1; k < CFG_CHUNKS; ++k) argmax_combine(v, i, pv[b * CFG_CHUNKS + k], pi[b * CFG_CHUNKS + k]);
        if (i == 0x7fffffff) i = 0;                                   // every logit -inf / NaN: index 0 (torch.argmax)
        const int unf = a.unfinished[b];
        const int tok = unf ? i : eos;
        if (step < max_new) a.out[(long)b * max_new + step] = tok;
        const int still = unf && (tok != eos);
        a.unfinished[b] = still;
        if (still) atomicOr(a.any_unfinished + ((step + 1) & 1023), 1);
        s_tok = tok < 0 ? 0 : (tok >= a.V ? a.V - 1 : tok);
    }
    __syncthreads();
    const float* src = a.embed_table + (long)s_tok * a.H;
    float* x0 = a.x + (long)b * a.H;
    for (int i = tid * 4; i < a.H; i += 1024) *(f32x4*)(x0 + i) = *(const f32x4*)(src + i);
}
void launch_text_argmax(hipStream_t s, const TextArgs& a, int B, float* scratch_v, int* scratch_i) {
    hipLaunchKernelGGL(text_scan_kernel, dim3(CFG_CHUNKS, B), dim3(256), 0, s, a, scratch_v, scratch_i);
    hipLaunchKernelGGL(text_argmax_kernel, dim3(B), dim3(256), 0, s, a, scratch_v, scratch_i);
}

// test tap: the sampler's uniform / Gumbel transform of raw 64-bit RNG outputs: out[i] = u, out[n+i] = -log(-log(u))
__global__ void uniform_from_bits_kernel(const uint64_t* __restrict__ z, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = uniform_from_bits(z[i]);
    out[i] = u;
    out[n + i] = -__logf(-__logf(u));
}
void launch_uniform_from_bits(hipStream_t s, const uint64_t* z, float* out, int n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(uniform_from_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, s, z, out, n);
}

__global__ void advance_kernel(int32_t* n) { *n += 1; }
void launch_advance(hipStream_t s, int32_t* n_dec) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, s, n_dec); }

// ------------------------------------------------------------------------------- weight conversion
template <typename T>
__global__ void convert_kernel(const void* __restrict__ src, int src_bf16, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = src_bf16 ? ET<bf16>::ld((const bf16*)src + i) : ((const float*)src)[i];
        ET<T>::st(dst + i, v);
    }
}
// fp32 -> bf16, 8 elements per thread per pass (two 16-byte loads, one 16-byte store); n % 8 == 0, 16-byte aligned pointers
__global__ __launch_bounds__(256) void convert_f32_bf16_vec_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 a = *(const u32x4*)(src + i * 8), b = *(const u32x4*)(src + i * 8 + 4);
        float f[8];
        ET<float>::unpack(a, f); ET<float>::unpack(b, f + 4);
        *(u32x4*)(dst + i * 8) = ET<bf16>::pack(f);
    }
}
template <typename T>
void launch_convert(hipStream_t s, const void* src, int src_bf16, T* dst, long n) {
    if (n <= 0) return;
    if constexpr (sizeof(T) == 2) {
        if (!src_bf16 && n % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && n >= 65536) {      // the VQ upsample inputs
            const long n8 = n / 8;
            const int vb = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
            hipLaunchKernelGGL(convert_f32_bf16_vec_kernel, dim3(vb), dim3(256), 0, s, (const float*)src, (bf16*)dst, n8);
            return;
        }
    }
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convert_kernel<T>, dim3(blocks), dim3(256), 0, s, src, src_bf16, dst, n);
}
template void launch_convert<float>(hipStream_t, const void*, int, float*, long);
template void launch_convert<bf16>(hipStream_t, const void*, int, bf16*, long);
void launch_to_f32(hipStream_t s, const void* src, int src_bf16, float* dst, long n) { launch_convert<float>(s, src, src_bf16, dst, n); }

template <typename T>
__global__ void convert_conv_kernel(const void* __restrict__ src, int src_bf16, T* __restrict__ dst, int Cout, int Cin, int kk) {
    const long n = (long)Cout * Cin * kk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // dst index i = (co*kk + tap)*Cin + ci ; src index = (co*Cin + ci)*kk + tap
        const int ci = (int)(i % Cin); const long t = i / Cin; const int tap = (int)(t % kk); const long co = t / kk;
        const long si = (co * Cin + ci) * kk + tap;
        const float v = src_bf16 ? ET<bf16>::ld((const bf16*)src + si) : ((const float*)src)[si];
        ET<T>::st(dst + i, v);
    }
}
template <typename T>
void launch_convert_conv(hipStream_t s, const void* src, int src_bf16, T* dst, int Cout, int Cin, int kk) {
    const long n = (long)Cout * Cin * kk;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convert_conv_kernel<T>, dim3(blocks), dim3(256), 0, s, src, src_bf16, dst, Cout, Cin, kk);
}
template void launch_convert_conv<float>(hipStream_t, const void*, int, float*, int, int, int);
template void launch_convert_conv<bf16>(hipStream_t, const void*, int, bf16*, int, int, int);

template <typename T>
__global__ void convert_il16_kernel(const void* __restrict__ src, int src_bf16, T* __restrict__ dst, int I, int H, int which) {
    const long n = (long)I * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / H; const int c = (int)(i % H);
        const long dr = (r >> 3) * 16 + which * 8 + (r & 7);
        const float v = src_bf16 ? ET<bf16>::ld((const bf16*)src + i) : ((const float*)src)[i];
        ET<T>::st(dst + dr * H + c, v);
    }
}
template <typename T>
void launch_convert_interleave16(hipStream_t s, const void* src, int src_bf16, T* dst, int I, int H, int which) {
    const long n = (long)I * H;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convert_il16_kernel<T>, dim3(blocks), dim3(256), 0, s, src, src_bf16, dst, I, H, which);
}
template void launch_convert_interleave16<float>(hipStream_t, const void*, int, float*, int, int, int);
template void launch_convert_interleave16<bf16>(hipStream_t, const void*, int, bf16*, int, int, int);

__global__ void l2norm_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float ss = 0.f;
    for (int d = 0; d < D; ++d) ss += src[(long)r * D + d] * src[(long)r * D + d];
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);                   // F.normalize eps
    for (int d = 0; d < D; ++d) dst[(long)r * D + d] = src[(long)r * D + d] / nrm;
}
void launch_l2norm_rows(hipStream_t s, const float* src, float* dst, int n, int D) {
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n, D);
}
void launch_fill_zero(hipStream_t s, void* p, long bytes) { (void)hipMemsetAsync(p, 0, bytes, s); }
