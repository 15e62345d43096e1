// Fused decode attention kernel template (RoPE + KV append + decode attention): included by llm_kernels.hip, which instantiates the
// PRODUCTION forms only, and by diag_attn.hip (libplangen_diag.so), which instantiates the timing ablations / older forms.
#pragma once
#include <type_traits>
#include "kernels.h"

// acc[v] += sum_s p[s*slab + offs[v]] with the loads of 4 slabs x NV values issued together
// (hipcc does not unroll a runtime-S loop: a plain loop costs S dependent memory round trips).
template <int NV>
__device__ __forceinline__ void sum_slabs(const float* __restrict__ p, long slab, int S, const int (&offs)[NV], float (&acc)[NV]) {
    for (int s0 = 0; s0 < S; s0 += 4) {
        float t[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* q = p + (long)(s0 + u < S ? s0 + u : S - 1) * slab;
#pragma unroll
            for (int v = 0; v < NV; ++v) t[u][v] = q[offs[v]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s0 + u < S) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += t[u][v];
            }
    }
}

// ------------------------------------------------------------------------------- fused decode attention
// Decode step only: one block per (row, head) does RoPE(q), RoPE(k), appends K/V at slot
// len+n_dec (reading the QKV GEMM's fp32 split-K slabs directly), then streams that row-head's
// cached K/V (non-temporal 16-byte loads: the cache is read exactly once per step) and merges
// the new key, which never leaves LDS, as one more online-softmax state.  Replaces
// rope_kv_kernel + attn_kernel (one launch and the q round trip less per layer).
// Round 2 (in-loop ablation, profiles/r02_c_attention_prologue.md): the prologue costs 6.6 us of the 69.6 us launch -- 3.6 us the
// slab loads, 2.4 us the K/V append stores (8 192 scattered 128-byte line writes per launch), 0.6 us the rest -- because at kernel
// start (and when the second round of blocks starts) every resident block is in its prologue and nothing streams.  So: the FIRST
// K/V chunk of every wave is issued BEFORE the prologue's dependent chain (slab loads -> RoPE -> LDS -> barrier) and consumed right
// after the barrier, in the loop's own kv[] / vv[] registers (no double buffer, still 4 waves per SIMD); the append is ONE
// 8-byte-per-lane store instruction per block, issued after the barrier.  Loop 1800 -> 1783 ms at bs=64.
template <typename T, int UN, int NW, int ABL = 0>      // ABL (timing ablations, WRONG results): 1 no K/V append store, 2 no slab / cos / sin loads, 4 no merge epilogue
// Argument order (round 5): gfx950 preloads the first 14 kernarg dwords into SGPRs at wave launch (-amdgpu-kernarg-preload-count); the rest
// arrive through an s_load that misses every cache (the host wrote the kernarg block for this launch).  The first 56 bytes are therefore
// exactly what the FIRST K/V chunk's addresses need -- row order, lengths, step counter, cache bases, geometry, shared-prompt alias -- and
// what the RoPE prologue needs (slabs, cos / sin, positions, output) comes behind them, under the chunk's flight.
__global__ __launch_bounds__(64 * NW) void attn_decode_fused_kernel(const int32_t* __restrict__ row_order, const int32_t* __restrict__ len_p,
                                                              const int32_t* __restrict__ n_dec_p, T* __restrict__ kc, T* __restrict__ vc,
                                                              int nh, int slots, int shared_len, int shared_row,
                                                              const float* __restrict__ qkv, long slab, T* __restrict__ obuf,
                                                              const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                              const int32_t* __restrict__ pos_off_p, int S, int max_pos, float scale) {
    constexpr int EPV = ET<T>::EPV, LPK = 128 / EPV, KPI = 64 / LPK, NST = NW * KPI;
    __shared__ float s_o[NST][128];
    __shared__ float s_m[NST], s_l[NST];
    __shared__ __attribute__((aligned(16))) float s_q[128];
    __shared__ float s_k[128], s_v[128];
    __shared__ float s_new;
    // ABL bit 32 (round 4, `attn_pair`): the grid has M / 2 rows of blocks and every block processes TWO (row, head) items -- rank y of the
    // longest-first order, then rank M - 1 - y -- so all blocks carry (longest + shortest) ~ the same number of keys and the second item's
    // prologue runs while the CU's other blocks stream (the one-item launch has every block in its prologue at once and a tail of short rows)
    constexpr int NIT = (ABL & 32) ? 2 : 1;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int head = blockIdx.x;
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
    const int yi = it == 0 ? (int)blockIdx.y : (int)(2 * gridDim.y - 1 - blockIdx.y);
    const int row = row_order ? row_order[yi] : yi;
    const int grp = l / LPK, lk = l % LPK;
    const int slot = len_p[row] + *n_dec_p;
    const int nprev = slot < slots ? slot : slots - 1;
    const int HD = nh * 128;
    const long cbase = ((long)row * nh + head) * slots * 128;
    constexpr int KPW = KPI * UN;
    const bool sh = shared_len > 0 && (row & 1);
    const int kstart = sh ? (shared_len < nprev ? shared_len : nprev) : 0;
    const long sbase = ((long)shared_row * nh + head) * (long)slots * 128 + lk * EPV;
    const T* const kpriv = kc + cbase + lk * EPV; const T* const vpriv = vc + cbase + lk * EPV;
    const T* const kshr = kc + sbase; const T* const vshr = vc + sbase;

    u32x4 kv[UN], vv[UN];
    auto issue = [&](const T* kb, const T* vb, int base, int k1, auto ntl) {
        constexpr bool NTL = decltype(ntl)::value;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = base + u * KPI + grp;
            key = key < k1 ? key : k1 - 1;
            key = key < 0 ? 0 : key;                                 // k1 == 0 (peeled issue of an empty segment): slot 0 is always mapped
            kv[u] = NTL ? __builtin_nontemporal_load((const u32x4*)(kb + (long)key * 128)) : *(const u32x4*)(kb + (long)key * 128);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = base + u * KPI + grp;
            key = key < k1 ? key : k1 - 1;
            key = key < 0 ? 0 : key;
            vv[u] = NTL ? __builtin_nontemporal_load((const u32x4*)(vb + (long)key * 128)) : *(const u32x4*)(vb + (long)key * 128);
        }
    };
    // first chunk of the first segment (shared prefix for uncond rows, private stream otherwise): no dependence on q
    const int seg_k1 = sh ? kstart : nprev;
    const int base0 = w * KPW;
    const bool have0 = base0 < seg_k1;
    // wave 0: the prologue's slab / cos / sin loads go out FIRST and straight-line (a runtime-S loop makes the compiler drain vmcnt
    // at its header, which serialised the peeled chunk in front of the slab loads), then every wave's first K/V chunk; the slab
    // values are waited for with the K/V chunk still in flight behind them.
    const int o6[6] = {0, 64, HD, HD + 64, 2 * HD, 2 * HD + 64};
    float t4[4][6], cs = 0.f, sn = 0.f;
    int pos = pos_off_p[row] + slot;
    if (pos >= max_pos) pos = max_pos - 1;
    const float* const qrow = qkv + (long)row * 3 * HD + head * 128 + (tid & 63);
    if (tid < 64) {
        if constexpr (ABL & 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 6; ++v) t4[u][v] = 0.01f * (float)(tid + v);
            cs = 1.f; sn = 0.f;
        } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* qq = qrow + (long)(u < S ? u : S - 1) * slab;
#pragma unroll
            for (int v = 0; v < 6; ++v) t4[u][v] = qq[o6[v]];
        }
        cs = cos_t[(long)pos * 64 + tid]; sn = sin_t[(long)pos * 64 + tid];
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // the slab sums must not be scheduled (with their vmcnt waits) in front of the K/V issue
    // UNCONDITIONAL (addresses clamped into the segment): behind a branch the compiler must count wave 0's slab waits for the path
    // that issued nothing, i.e. 14 ops too strict on the path that did -- the RoPE prologue then waited for 12 of the 14 K/V loads
    // (3.5 us per launch in the in-loop ablation)
    if (sh) issue(kshr, vshr, base0, seg_k1, std::false_type{}); else issue(kpriv, vpriv, base0, seg_k1, std::true_type{});
    __builtin_amdgcn_sched_barrier(0);

    if (tid < 64) {
        const int j = tid;
        float a6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // opaque touch: the loaded slab values may not be consumed (and waited for) before this point in program order
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 6; ++v) asm volatile("" : "+v"(t4[u][v])::"memory");
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < S) {
#pragma unroll
                for (int v = 0; v < 6; ++v) a6[v] += t4[u][v];
            }
        if (S > 4) sum_slabs<6>(qrow + 4 * slab, slab, S - 4, o6, a6);
        const float q0 = a6[0], q1 = a6[1], k0 = a6[2], k1 = a6[3], v0 = a6[4], v1 = a6[5];
        const float c = cs;
        s_q[j] = ET<T>::round(q0 * c - q1 * sn) * scale;
        s_q[j + 64] = ET<T>::round(q1 * c + q0 * sn) * scale;
        const float kr0 = ET<T>::round(k0 * c - k1 * sn), kr1 = ET<T>::round(k1 * c + k0 * sn);
        const float vr0 = ET<T>::round(v0), vr1 = ET<T>::round(v1);
        s_k[j] = kr0; s_k[j + 64] = kr1; s_v[j] = vr0; s_v[j + 64] = vr1;
    }
    __syncthreads();
    if (w == 0) {
        float d = s_q[l] * s_k[l] + s_q[l + 64] * s_k[l + 64];
        d = wave_sum(d);
        if (l == 0) s_new = d;
    }
    float q[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) q[e] = s_q[lk * EPV + e];
    float m_run = -INFINITY, l_run = 0.f, o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = 0.f;
    auto consume = [&](int base, int k1) {
        float sc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float kf[EPV]; ET<T>::unpack(kv[u], kf);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) d = fmaf(q[e], kf[e], d);
#pragma unroll
            for (int o_ = LPK / 2; o_ > 0; o_ >>= 1) d += __shfl_xor(d, o_, 64);
            sc[u] = (base + u * KPI + grp < k1) ? d : -INFINITY;
        }
        float mx = m_run;
#pragma unroll
        for (int u = 0; u < UN; ++u) mx = fmaxf(mx, sc[u]);
        if (mx > -INFINITY) {
            const float alpha = __expf(m_run - mx);
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < EPV; ++e) o[e] *= alpha;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float p = __expf(sc[u] - mx);
                l_run += p;
                float vf[EPV]; ET<T>::unpack(vv[u], vf);
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = fmaf(p, vf[e], o[e]);
            }
            m_run = mx;
        }
    };
    // explicitly software-pipelined form of the loop (ABL bit 16, experiment): K(i+1) goes out BEFORE the wait for V(i), so one of the
    // two round trips of an iteration runs under the other's arithmetic; loads unconditional (clamped) so the counted waits stay exact
    auto issueK = [&](const T* kb, int base, int k1) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = base + u * KPI + grp;
            key = key < k1 ? key : k1 - 1;
            kv[u] = __builtin_nontemporal_load((const u32x4*)(kb + (long)key * 128));
        }
    };
    auto issueV = [&](const T* vb, int base, int k1) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = base + u * KPI + grp;
            key = key < k1 ? key : k1 - 1;
            vv[u] = __builtin_nontemporal_load((const u32x4*)(vb + (long)key * 128));
        }
    };
    auto run_pipe = [&](const T* kb, const T* vb, int kfirst, int k1) {
        if (kfirst >= k1) return;
        issueK(kb, kfirst, k1);
        for (int base = kfirst; base < k1; base += NW * KPW) {
            issueV(vb, base, k1);
            float sc[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float kf[EPV]; ET<T>::unpack(kv[u], kf);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < EPV; ++e) d = fmaf(q[e], kf[e], d);
#pragma unroll
                for (int o_ = LPK / 2; o_ > 0; o_ >>= 1) d += __shfl_xor(d, o_, 64);
                sc[u] = (base + u * KPI + grp < k1) ? d : -INFINITY;
            }
            __builtin_amdgcn_sched_barrier(0);
            issueK(kb, base + NW * KPW, k1);                         // next chunk's K (clamped to the last key past the end: one cache line)
            __builtin_amdgcn_sched_barrier(0);
            float mx = m_run;
#pragma unroll
            for (int u = 0; u < UN; ++u) mx = fmaxf(mx, sc[u]);
            const float mxs = (mx > -INFINITY) ? mx : 0.f;
            const float alpha = __expf(m_run - mxs);
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < EPV; ++e) o[e] *= alpha;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float p = __expf(sc[u] - mxs);
                l_run += p;
                float vf[EPV]; ET<T>::unpack(vv[u], vf);
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = fmaf(p, vf[e], o[e]);
            }
            m_run = mx;
        }
    };
    auto run = [&](const T* kb, const T* vb, int kfirst, int k1, auto ntl) {
        if constexpr ((ABL & 16) != 0 && decltype(ntl)::value) { run_pipe(kb, vb, kfirst, k1); return; }
        for (int base = kfirst; base < k1; base += NW * KPW) { issue(kb, vb, base, k1, ntl); consume(base, k1); }
    };
    if (have0) consume(base0, seg_k1);
    if (sh) {
        run(kshr, vshr, base0 + NW * KPW, kstart, std::false_type{});
        run(kpriv, vpriv, kstart + w * KPW, nprev, std::true_type{});
    } else {
        run(kpriv, vpriv, base0 + NW * KPW, nprev, std::true_type{});
    }
    if constexpr (ABL & 4) {
        float acc = l_run + m_run;
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc += o[e];
        if (acc == 123.456f) ET<T>::st(obuf + (long)row * HD + head * 128 + tid % 128, acc);
        return;
    }
    const int stt = w * KPI + grp;
#pragma unroll
    for (int e = 0; e < EPV; ++e) s_o[stt][lk * EPV + e] = o[e];
    if (lk == 0) { s_m[stt] = m_run; s_l[stt] = l_run; }
    __syncthreads();
    if (tid < 128) {
        float Mx = s_new;
#pragma unroll
        for (int i = 0; i < NST; ++i) Mx = fmaxf(Mx, s_m[i]);
        const float fn = __expf(s_new - Mx);
        float num = fn * s_v[tid], den = fn;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const float f = (s_m[i] > -INFINITY) ? __expf(s_m[i] - Mx) : 0.f;
            num = fmaf(f, s_o[i][tid], num);
            den = fmaf(f, s_l[i], den);
        }
        ET<T>::st(obuf + (long)row * HD + head * 128 + tid, num / den);
    }
    // K/V append, LAST thing the block does: ONE store instruction (lanes 0-31 the K row, 32-63 the V row, 4 elements each) from the
    // RoPE'd row still sitting in LDS.  Nothing waits behind it: issued right after the prologue it sat in front of wave 0's first
    // `vmcnt` wait (counted in order), and the store's acknowledgement cost 2.4 us of every launch (in-loop ablation, 34 ms per loop).
    if (!(ABL & 1) && tid < 64 && slot < slots) {
        const float* src = (l < 32 ? s_k : s_v) + (l & 31) * 4;
        T* dst = (l < 32 ? kc : vc) + cbase + (long)slot * 128 + (l & 31) * 4;
        if constexpr (sizeof(T) == 2) {
            u32x2 pk; pk.x = pack_bf16x2(src[0], src[1]); pk.y = pack_bf16x2(src[2], src[3]);
            *(u32x2*)dst = pk;
        } else {
            *(f32x4*)dst = *(const f32x4*)src;
        }
    }
    if constexpr (NIT > 1) __syncthreads();          // the append and the merge have read this item's LDS state
    }
}
