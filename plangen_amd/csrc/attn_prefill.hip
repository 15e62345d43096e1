// Prefill (causal, variable-length, packed) flash attention on MFMA for head_dim 128, bf16.
//
// One 256-thread block = 64 consecutive queries of one (row, head); wave w owns 16 of them.
// Per 64-key tile: K [64][128] and V^T [128][64] are staged in LDS (coalesced 16-byte global
// loads; V is transposed on the way in so the PV MFMA's B operand is key-contiguous).
// S^T = K.Q^T is computed with K as the A operand ("swapped" QK^T): the C layout then gives
// lane (lr, g) the scores of QUERY lr for keys {16*kt + 4*g + r}, which is exactly the A-operand
// layout of the PV MFMA under a consistent key permutation -- P never goes through LDS -- and a
// query's softmax statistics need only two cross-lane steps (xor 16, 32).  O accumulates in the
// C layout (row = query 4*g+r), so the per-query rescale factors are fetched with 4 shuffles.
// K/V come from the KV cache [row][head][slot][128] that rope_kv_kernel just filled.
#include "kernels.h"
#include "gemm_common.h"

#define FA_KROW 272          // bytes per K row in LDS (256 + 16 pad)
#define FA_VROW 144          // bytes per V^T row in LDS (64 keys * 2 + 16 pad)

__global__ __launch_bounds__(256) void attn_prefill_flash_kernel(const bf16* __restrict__ qbuf, bf16* __restrict__ obuf,
                                                                const bf16* __restrict__ kc, const bf16* __restrict__ vc,
                                                                const int32_t* __restrict__ row_off, const int32_t* __restrict__ len,
                                                                int nh, int slots, float scale) {
    __shared__ __attribute__((aligned(16))) char sK[64 * FA_KROW];
    __shared__ __attribute__((aligned(16))) char sV[128 * FA_VROW];
    const int qt = blockIdx.x, head = blockIdx.y, row = blockIdx.z;
    const int off = row_off[row];
    const int L = len[row];
    if (off < 0 || qt * 64 >= L) return;                       // row has no packed tokens here / tile beyond the prompt
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lr = l & 15;
    const int HD = nh * 128;
    const int q0 = qt * 64 + w * 16;                            // first query (slot index) of this wave
    const int myq = q0 + lr;                                    // the query whose softmax state this lane carries
    // Q fragments (B operand): lane (q = lr, g) holds Q[q][ds*32 + g*8 .. +8], pre-scaled
    bf16x8 qf[4];
    {
        const int qq = myq < L ? myq : L - 1;
        const bf16* qp = qbuf + (long)(off + qq) * HD + head * 128 + g * 8;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const u32x4 v = *(const u32x4*)(qp + ds * 32);
            float f[8]; ET<bf16>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= scale;
            const u32x4 pv = ET<bf16>::pack(f);
            qf[ds] = *(const bf16x8*)&pv;
        }
    }
    const bf16* kbase = kc + ((long)row * nh + head) * slots * 128;
    const bf16* vbase = vc + ((long)row * nh + head) * slots * 128;
    f32x4 oacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int qmax = min(qt * 64 + 63, L - 1);                  // last query of the block
    const int ntiles = qmax / 64 + 1;                           // causal: keys 0 .. qmax
    for (int kt0 = 0; kt0 < ntiles; ++kt0) {
        const int kb = kt0 * 64;
        __syncthreads();                                        // previous tile fully consumed
        // ---- stage K (row-major) and V (transposed) for keys kb .. kb+63
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = tid + j * 256, key = v >> 4, cv = v & 15;
            const int ks = (kb + key < L) ? kb + key : L - 1;
            const u32x4 kv = *(const u32x4*)(kbase + (long)ks * 128 + cv * 8);
            *(u32x4*)(sK + key * FA_KROW + cv * 16) = kv;
            const u32x4 vv = *(const u32x4*)(vbase + (long)ks * 128 + cv * 8);
            const uint16_t* ve = (const uint16_t*)&vv;
#pragma unroll
            for (int e = 0; e < 8; ++e) *(uint16_t*)(sV + (cv * 8 + e) * FA_VROW + key * 2) = ve[e];
        }
        __syncthreads();
        // ---- S^T[key][q] = sum_d K[key][d] Q[q][d] : 4 key sub-tiles x 4 d-steps
        f32x4 sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sacc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const bf16x8 ka = *(const bf16x8*)(sK + (kt * 16 + lr) * FA_KROW + (ds * 32 + g * 8) * 2);
                sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ds], sacc[kt], 0, 0, 0);
            }
        }
        // lane (lr, g): query myq, keys kb + kt*16 + g*4 + r
        float mx = m_run;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + kt * 16 + g * 4 + r;
                if (key > myq || key >= L) sacc[kt][r] = -INFINITY;        // causal + length mask
                mx = fmaxf(mx, sacc[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // key 0 <= every query, so mx is finite from the first tile on
        const float alpha = __expf(m_run - mx);
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float p[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __expf(sacc[2 * s][r] - mx);
                p[4 + r] = __expf(sacc[2 * s + 1][r] - mx);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) psum += p[e];
            const u32x4 pv = ET<bf16>::pack(p);
            pf[s] = *(const bf16x8*)&pv;
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = mx;
        // ---- rescale O rows (row = query g*4+r of this wave) by that query's alpha
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) oacc[dt][r] *= a;
        }
        // ---- O[q][d] += P[q][keys] V[keys][d] : k-step s covers key sub-tiles 2s, 2s+1;
        //      B operand lane (d = dt*16+lr, g): keys {2s*16+g*4..+3, (2s+1)*16+g*4..+3}
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const char* vr = sV + (dt * 16 + lr) * FA_VROW;
                const u32x2 lo = *(const u32x2*)(vr + ((2 * s) * 16 + g * 4) * 2);
                const u32x2 hi = *(const u32x2*)(vr + ((2 * s + 1) * 16 + g * 4) * 2);
                u32x4 vb; vb.x = lo.x; vb.y = lo.y; vb.z = hi.x; vb.w = hi.y;
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[s], *(const bf16x8*)&vb, oacc[dt], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store: row q = q0 + g*4 + r, column d = dt*16 + lr
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lq = __shfl(l_run, g * 4 + r, 64);
        const int q = q0 + g * 4 + r;
        if (q < L) {
            const float inv = 1.f / lq;
            bf16* op = obuf + (long)(off + q) * HD + head * 128 + lr;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) ET<bf16>::st(op + dt * 16, oacc[dt][r] * inv);
        }
    }
}

// ---- v2: 128 queries per block, K/V tiles by LDS-DMA (double-buffered), V consumed through the LDS transpose read -----------------
// The first kernel spends most of a tile on staging (V transposed with eight 2-byte LDS writes per 16-byte load, two block
// barriers, 32 MFMAs per wave per 64-key tile).  Here:
//   * a wave owns 32 queries (two 16-query groups): 64 MFMAs per wave per tile on the same K / V fragments;
//   * K [64][128] and V [64][128] tiles go global -> LDS by global_load_lds straight from the KV cache (key-major rows of 256 B,
//     no transpose pass), double-buffered per operand, each tile retired one barrier before the phase that reads it (two raw
//     s_barrier per tile, counted vmcnt(4); protocol at the loop);
//   * 16-byte chunk c of row r sits in slot c ^ (r & 15) for K (conflict-free ds_read_b128 fragments) and c ^ ((r & 3) << 1) for V,
//     applied through the DMA's source address (the DMA writes lane-linear);
//   * the PV MFMA's B operand (8 keys of one d column per lane) comes from ds_read_b64_tr_b16: a 16-lane group fetches a
//     [4 keys][16 d] block (lane i: row i >> 2, four d at (i & 3) * 4) and lane j receives column j -- the V swizzle puts the four
//     rows of a block in four different 32-byte bank segments;
//   * query tiles are launched longest first (causal: tile qt reads 2 qt + 2 key tiles); a wave skips key tiles that lie entirely
//     in its queries' future.
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define FB_TILE 16384
__device__ __forceinline__ u32x2 lds_tr16(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(u32x2, v);
}
__global__ __launch_bounds__(256, 2) void attn_prefill_flash2_kernel(const bf16* __restrict__ qbuf, bf16* __restrict__ obuf,
                                                                    const bf16* __restrict__ kc, const bf16* __restrict__ vc,
                                                                    const int32_t* __restrict__ row_off, const int32_t* __restrict__ len,
                                                                    int nh, int slots, float scale, int nqt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];                    // K[2][16 KiB] | V[2][16 KiB]
    const int qt = nqt - 1 - (int)blockIdx.x, head = blockIdx.y, row = blockIdx.z;
    const int off = row_off[row];
    const int L = len[row];
    if (off < 0 || qt * 128 >= L) return;
    const int tid = threadIdx.x, l = tid & 63, g = l >> 4, lr = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HD = nh * 128;
    const int q0w = qt * 128 + w * 32;                                              // first query of this wave
    bf16x8 qf[2][4];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
        const int myq = q0w + qg * 16 + lr;
        const int qq = myq < L ? myq : L - 1;
        const bf16* qp = qbuf + (long)(off + qq) * HD + head * 128 + g * 8;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const u32x4 v = *(const u32x4*)(qp + ds * 32);
            float f[8]; ET<bf16>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= scale;
            const u32x4 pv = ET<bf16>::pack(f);
            qf[qg][ds] = *(const bf16x8*)&pv;
        }
    }
    const bf16* kbase = kc + ((long)row * nh + head) * slots * 128;
    const bf16* vbase = vc + ((long)row * nh + head) * slots * 128;
    // DMA: wave w, instruction i covers tile rows (w*4 + i)*4 .. +3 (lane: row l >> 4, slot l & 15)
    int rr[4], ksw[4], vsw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (w * 4 + i) * 4 + (l >> 4);
        rr[i] = r; ksw[i] = ((l & 15) ^ (r & 15)) * 8; vsw[i] = ((l & 15) ^ ((r & 3) << 1)) * 8;
    }
    // K tile t -> K slot t & 1, V tile t -> V slot t & 1 (4 DMA instructions per wave each)
    auto issueK = [&](int t) __attribute__((always_inline)) {
        const int kb = t * 64;
        char* kd = smem + (t & 1) * FB_TILE + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = kb + rr[i];
            key = key < L ? key : L - 1;
            glds16(kbase + (long)key * 128 + ksw[i], kd + i * 1024);
        }
    };
    auto issueV = [&](int t) __attribute__((always_inline)) {
        const int kb = t * 64;
        char* vd = smem + 2 * FB_TILE + (t & 1) * FB_TILE + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = kb + rr[i];
            key = key < L ? key : L - 1;
            glds16(vbase + (long)key * 128 + vsw[i], vd + i * 1024);
        }
    };
    f32x4 oacc[2][8];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
        for (int i = 0; i < 8; ++i) oacc[qg][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int qlast = min(qt * 128 + 127, L - 1);
    const int ntiles = qlast / 64 + 1;                                             // causal: keys 0 .. qlast
    // V transpose-read addressing: rows key0 + (lr >> 2), logical chunk 2 dt + ((lr & 3) >> 1), 8-byte half lr & 1
    const int vlane = (4 * g + (lr >> 2)) * 256 + (lr & 1) * 8;
    const int vcb = (lr & 3) >> 1, vsz = (lr >> 2) << 1;
    // Staging protocol (round 3): a tile is RETIRED (its issuing waves' vmcnt wait + a block barrier) one barrier BEFORE the phase
    // that first reads it, never in the same phase (CDNA guide: "read a staged buffer one phase after the wait that retires it";
    // the same-phase form of the decode GEMM read a stale DMA piece in 0.5 % of cold launches).  Two barriers per tile, still
    // two slots per operand (64 KiB, two blocks per CU):
    //   A(t): retires V(t)   [issued after A(t-1)];  everyone is past PV(t-1)  -> V(t+1) is issued into the slot of V(t-1)
    //         ... QK^T(t) + softmax read K(t), retired at B(t-1) ...
    //   B(t): retires K(t+1) [issued after PV(t-1)]; everyone is past QK^T(t)
    //         ... PV(t) reads V(t), retired at A(t) ...  then K(t+2) is issued into the slot of K(t)
    // In flight behind each wait is exactly one younger tile (4 DMA instructions per wave): counted vmcnt(4), vmcnt(0) at the tail.
    issueK(0); issueV(0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                // K(0) landed (V(0) may still be in flight)
    __builtin_amdgcn_s_barrier();                                                   // B(-1): retires K(0)
    if (1 < ntiles) issueK(1);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // V(t) landed; K(t+1) behind it
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                               // A(t)
        if (t + 1 < ntiles) issueV(t + 1);
        const int kb = t * 64;
        const bool skip = kb > q0w + 31;                                            // the whole tile is in the future of this wave's queries
        const char* sK = smem + (t & 1) * FB_TILE;
        const char* sV = sK + 2 * FB_TILE;
        bf16x8 pf[2][2];
        if (!skip) {
        f32x4 sacc[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sacc[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; sacc[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const bf16x8 ka = *(const bf16x8*)(sK + (kt * 16 + lr) * 256 + (((ds * 4 + g) ^ lr) << 4));
                sacc[0][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[0][ds], sacc[0][kt], 0, 0, 0);
                sacc[1][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[1][ds], sacc[1][kt], 0, 0, 0);
            }
        }
        const bool need_mask = (kb + 63 > q0w) || (kb + 64 > L);
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) {
            const int myq = q0w + qg * 16 + lr;
            float mx = m_run[qg];
            if (need_mask) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb + kt * 16 + g * 4 + r;
                        if (key > myq || key >= L) sacc[qg][kt][r] = -INFINITY;
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qg][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float alpha = __expf(m_run[qg] - mx);                              // key 0 is visible to every query: mx finite from tile 0
            float psum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float p[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __expf(sacc[qg][2 * s2][r] - mx);
                    p[4 + r] = __expf(sacc[qg][2 * s2 + 1][r] - mx);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) psum += p[e];
                const u32x4 pv = ET<bf16>::pack(p);
                pf[qg][s2] = *(const bf16x8*)&pv;
            }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l_run[qg] = l_run[qg] * alpha + psum;
            m_run[qg] = mx;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) oacc[qg][dt][r] *= a;
            }
        }
        }
        if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // K(t+1) landed; V(t+1) behind it
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                               // B(t)
        if (!skip) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const int co = ((2 * dt + vcb) ^ vsz) << 4;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* va = sV + (32 * s2) * 256 + vlane + co;
                const u32x2 lo = lds_tr16(va), hi = lds_tr16(va + 16 * 256);
                u32x4 vb; vb.x = lo.x; vb.y = lo.y; vb.z = hi.x; vb.w = hi.y;
                const bf16x8 vf = *(const bf16x8*)&vb;
                oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[0][s2], vf, oacc[0][dt], 0, 0, 0);
                oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[1][s2], vf, oacc[1][dt], 0, 0, 0);
            }
        }
        }
        // issued AFTER PV(t): hipcc drains vmcnt in front of the transpose reads (it cannot disambiguate them from LDS-DMA writes),
        // so anything issued before PV(t) would be waited for there
        if (t + 2 < ntiles) issueK(t + 2);
    }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lq = __shfl(l_run[qg], g * 4 + r, 64);
            const int q = q0w + qg * 16 + g * 4 + r;
            if (q < L) {
                const float inv = 1.f / lq;
                bf16* op = obuf + (long)(off + q) * HD + head * 128 + lr;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) ET<bf16>::st(op + dt * 16, oacc[qg][dt][r] * inv);
            }
        }
}

void launch_attn_prefill_flash(hipStream_t s, const bf16* qbuf, bf16* obuf, const bf16* kc, const bf16* vc,
                               const int32_t* row_off, const int32_t* len, int R, int max_len, int nh, int slots, float scale) {
    if (R <= 0 || max_len <= 0) return;
    if (pg_tune->prefill_attn != 1) {
        (void)PG_DYN_LDS(attn_prefill_flash2_kernel, 4 * FB_TILE);
        const int nqt = (max_len + 127) / 128;
        hipLaunchKernelGGL(attn_prefill_flash2_kernel, dim3(nqt, nh, R), dim3(256), 4 * FB_TILE, s, qbuf, obuf, kc, vc, row_off, len, nh, slots, scale, nqt);
        return;
    }
    hipLaunchKernelGGL(attn_prefill_flash_kernel, dim3((max_len + 63) / 64, nh, R), dim3(256), 0, s, qbuf, obuf, kc, vc, row_off, len,
                       nh, slots, scale);
}

// ------------------------------------------------------------------------------- SigLIP (ViT) attention
// Non-causal flash attention for the understanding encoder (siglip_vit.py:164-192): head_dim 64, P tokens per
// image (P % 64 == 0), bf16.  Same MFMA scheme as the prefill kernel above (swapped QK^T so P stays in
// registers, per-query statistics with two cross-lane steps); q | k come from the fused [M, 2C] projection,
// V^T from the [B][C][P] tensor the encoder's V projection already writes (key-contiguous rows: staged
// without a transpose).  Scores are scaled in fp32 after the MFMA, like the softmax kernel it replaces.
#define VA_ROW 144           // bytes per LDS row (64 x bf16 + 16 pad)
// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) on the VALU: v_permlane16_swap / v_permlane32_swap (gfx950) exchange
// rows between two copies of the value -- with both operands the same register the pair it returns is (even rows duplicated, odd rows duplicated)
// resp. (low half duplicated, high half duplicated) -- instead of two ds_bpermute round trips through the LDS crossbar per reduction
// (the flash kernels' per-tile softmax chain is latency-bound).  Same operands, commutative operation: bit-identical to the __shfl_xor form.
__device__ __forceinline__ float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__global__ __launch_bounds__(256) void attn_vit_flash_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt,
                                                            bf16* __restrict__ o, int P, int C, float scale) {
    __shared__ __attribute__((aligned(16))) char sK[64 * VA_ROW];
    __shared__ __attribute__((aligned(16))) char sV[64 * VA_ROW];
    const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lr = l & 15;
    const int q0 = qt * 64 + w * 16, myq = q0 + lr;
    bf16x8 qf[2];
    {
        const bf16* qp = qk + ((long)b * P + myq) * 2 * C + head * 64 + g * 8;
        qf[0] = *(const bf16x8*)qp; qf[1] = *(const bf16x8*)(qp + 32);
    }
    const bf16* kbase = qk + (long)b * P * 2 * C + C + head * 64;
    const bf16* vbase = vt + ((long)b * C + head * 64) * P;
    f32x4 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = scale * 1.44269504088896340736f;
    for (int kb = 0; kb < P; kb += 64) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int v = tid + j * 256, r = v >> 3, cv = v & 7;            // 64 rows x 8 chunks of 16 B
            *(u32x4*)(sK + r * VA_ROW + cv * 16) = *(const u32x4*)(kbase + (long)(kb + r) * 2 * C + cv * 8);
            *(u32x4*)(sV + r * VA_ROW + cv * 16) = *(const u32x4*)(vbase + (long)r * P + kb + cv * 8);
        }
        __syncthreads();
        f32x4 sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sacc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const bf16x8 ka = *(const bf16x8*)(sK + (kt * 16 + lr) * VA_ROW + (ds * 32 + g * 8) * 2);
                sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ds], sacc[kt], 0, 0, 0);
            }
        }
        // scores stay UNSCALED in the accumulators: the running maximum is kept in units of log2 (m_run = max(s) * scale * log2 e) and every
        // probability is ONE fma + ONE v_exp_f32: 2^(s * c - m) with c = scale * log2 e (round 4: was scale-multiply, subtract, log2e-multiply, exp)
        float mraw = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mraw = fmaxf(mraw, sacc[kt][r]);
        mraw = fmaxf(mraw, __shfl_xor(mraw, 16, 64));
        mraw = fmaxf(mraw, __shfl_xor(mraw, 32, 64));
        const float mx = fmaxf(m_run, mraw * c2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - mx);
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float p[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * s2][r], c2, -mx));
                p[4 + r] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * s2 + 1][r], c2, -mx));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) psum += p[e];
            const u32x4 pv = ET<bf16>::pack(p);
            pf[s2] = *(const bf16x8*)&pv;
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = mx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt][r] *= a;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* vr = sV + (dt * 16 + lr) * VA_ROW;
                const u32x2 lo = *(const u32x2*)(vr + ((2 * s2) * 16 + g * 4) * 2);
                const u32x2 hi = *(const u32x2*)(vr + ((2 * s2 + 1) * 16 + g * 4) * 2);
                u32x4 vb; vb.x = lo.x; vb.y = lo.y; vb.z = hi.x; vb.w = hi.y;
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[s2], *(const bf16x8*)&vb, oacc[dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float inv = 1.f / __shfl(l_run, g * 4 + r, 64);
        bf16* op = o + ((long)b * P + q0 + g * 4 + r) * C + head * 64 + lr;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ET<bf16>::st(op + dt * 16, oacc[dt][r] * inv);
    }
}
// Round 4: K / V^T RESIDENT form.  One block per (image, head) keeps the head's whole K [P][64] and V^T [64][P] in LDS (P = 576: 81 + 73 KiB of
// the CU's 160 KiB) and walks ALL P queries over it (passes of NW waves x 16 queries; NW = 16: four waves per SIMD hide the per-tile softmax chain): no per-tile staging, no barrier inside the loops --
// the 64-key tile kernel above staged every K / V tile 9 times per head behind two __syncthreads() each and kept the MFMA pipe busy 11 % of the
// time.  Grid = heads x images = 1024 blocks = exactly 4 per CU.  Same per-tile arithmetic (order of the online-softmax updates, bf16 P, fp32
// statistics) as the tile kernel, so the two agree to rounding of identical operations (tests/test_gpu_vision_full.py).
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_vit_resident_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt,
                                                               bf16* __restrict__ o, int P, int C, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VROW = P * 2 + 16;                      // bytes per V^T row (P keys + 16 pad: row stride = 36 dwords mod 64 like VA_ROW)
    char* sK = smem;                                   // [P][VA_ROW]
    char* sV = smem + (long)P * VA_ROW;                // [64][VROW]
    const int head = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lr = l & 15;
    const bf16* kbase = qk + (long)b * P * 2 * C + C + head * 64;
    const bf16* vbase = vt + ((long)b * C + head * 64) * P;
    // one-time fill: K rows are 128 B in global (8 x 16 B), V^T rows P * 2 B (P / 8 x 16 B); all loads of a round issued before the LDS writes
    for (int v0 = 0; v0 < P * 8; v0 += 64 * NW * 4) {
        u32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int v = v0 + j * 64 * NW + tid; if (v < P * 8) t[j] = *(const u32x4*)(kbase + (long)(v >> 3) * 2 * C + (v & 7) * 8); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int v = v0 + j * 64 * NW + tid; if (v < P * 8) *(u32x4*)(sK + (v >> 3) * VA_ROW + (v & 7) * 16) = t[j]; }
    }
    const int vpr = P / 8;                             // 16-byte vectors per V^T row
    for (int v0 = 0; v0 < 64 * vpr; v0 += 64 * NW * 4) {
        u32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int v = v0 + j * 64 * NW + tid; if (v < 64 * vpr) t[j] = *(const u32x4*)(vbase + (long)(v / vpr) * P + (v % vpr) * 8); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int v = v0 + j * 64 * NW + tid; if (v < 64 * vpr) *(u32x4*)(sV + (v / vpr) * VROW + (v % vpr) * 16) = t[j]; }
    }
    __syncthreads();
    for (int qt = 0; qt * 16 * NW < P; ++qt) {
        const int q0 = qt * 16 * NW + w * 16, myq = q0 + lr;
        if (q0 >= P) break;                                     // wave-uniform: a partial last pass (P not a multiple of 16 * NW)
        bf16x8 qf[2];
        {
            const bf16* qp = qk + ((long)b * P + myq) * 2 * C + head * 64 + g * 8;
            qf[0] = *(const bf16x8*)qp; qf[1] = *(const bf16x8*)(qp + 32);
        }
        f32x4 oacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        const float c2 = scale * 1.44269504088896340736f;
        for (int kb = 0; kb < P; kb += 64) {
            f32x4 sacc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sacc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ds = 0; ds < 2; ++ds) {
                    const bf16x8 ka = *(const bf16x8*)(sK + (kb + kt * 16 + lr) * VA_ROW + (ds * 32 + g * 8) * 2);
                    sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ds], sacc[kt], 0, 0, 0);
                }
            }
            float mraw = -INFINITY;                        // same arithmetic as the tile kernel: unscaled scores, log2-domain running maximum
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mraw = fmaxf(mraw, sacc[kt][r]);
            mraw = rows_max(mraw);
            const float mx = fmaxf(m_run, mraw * c2);
            const float alpha = __builtin_amdgcn_exp2f(m_run - mx);
            float psum = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float p[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * s2][r], c2, -mx));
                    p[4 + r] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * s2 + 1][r], c2, -mx));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) psum += p[e];
                const u32x4 pv = ET<bf16>::pack(p);
                pf[s2] = *(const bf16x8*)&pv;
            }
            psum = rows_sum(psum);
            l_run = l_run * alpha + psum;
            m_run = mx;
            if (!__all(alpha == 1.0f)) {                       // the running maximum moved for some query of this wave (x 1.0f is a bitwise no-op: skipping it changes nothing)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oacc[dt][r] *= a;
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const char* vr = sV + (dt * 16 + lr) * VROW + kb * 2;
                    const u32x2 lo = *(const u32x2*)(vr + ((2 * s2) * 16 + g * 4) * 2);
                    const u32x2 hi = *(const u32x2*)(vr + ((2 * s2 + 1) * 16 + g * 4) * 2);
                    u32x4 vb; vb.x = lo.x; vb.y = lo.y; vb.z = hi.x; vb.w = hi.y;
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[s2], *(const bf16x8*)&vb, oacc[dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv = 1.f / __shfl(l_run, g * 4 + r, 64);
            bf16* op = o + ((long)b * P + q0 + g * 4 + r) * C + head * 64 + lr;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ET<bf16>::st(op + dt * 16, oacc[dt][r] * inv);
        }
    }
}
// qk [B*P, 2C] (q | k), vt [B][C][P] (V transposed), o [B*P, C]; heads of 64; P % 64 == 0.
void launch_attn_vit_flash(hipStream_t s, const bf16* qk, const bf16* vt, bf16* o, int B, int P, int C, int NH, float scale) {
    const long lds = (long)P * VA_ROW + 64L * (P * 2 + 16);
    if (pg_tune->vit_attn != 1 && lds <= 160 * 1024) {          // K / V^T of one head resident in LDS (vit_attn = 1 selects the 64-key tile kernel for A/B)
        const int nw = pg_tune->vit_attn >= 4 ? pg_tune->vit_attn : 16;      // measured on MI355X (64 images): tower 41.8 / 36.7 / 35.5 / 34.9 ms at 4 / 8 / 12 / 16 waves, tile kernel 37.1
        // ~154 KB of dynamic LDS: needs the per-device attribute; when it cannot be set (or the device offers less LDS) the tile kernel below runs
#define VIT_RES(NW) if (PG_DYN_LDS(attn_vit_resident_kernel<NW>, 160 * 1024)) { hipLaunchKernelGGL(attn_vit_resident_kernel<NW>, dim3(NH, B), dim3(64 * NW), (size_t)lds, s, qk, vt, o, P, C, scale); return; }
        if (nw == 4) { VIT_RES(4) }
        else if (nw == 8) { VIT_RES(8) }
        else if (nw == 12) { VIT_RES(12) }
        else { VIT_RES(16) }
#undef VIT_RES
    }
    hipLaunchKernelGGL(attn_vit_flash_kernel, dim3(P / 64, NH, B), dim3(256), 0, s, qk, vt, o, P, C, scale);
}
