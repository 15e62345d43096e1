// Skinny (decode) weight-streaming GEMM kernel templates v1 / v3 / v4 and their launch templates: included by gemm.hip, which instantiates
// the PRODUCTION dispatch only, and by diag_gemm.hip (libplangen_diag.so: sweep / ablation / hazard-forensics instantiations).
#pragma once
#include "kernels.h"
#include "gemm_common.h"

// ------------------------------------------------------------------------------- skinny GEMM
#define SK_BK 128
#define SK_ROWB 288           // LDS row stride (256 B of k + 32 B pad: conflict-free ds_read_b128 for the (lr, g) fragment order)


// Block epilogue of the skinny kernels: the NW waves' accumulators (MFMA C layout: lane holds
// 4 rows x 1 column) are transposed through LDS so every lane stores 16 contiguous bytes
// instead of 32 scattered dword stores.  smem must hold MT*16 rows x (NW*16+4) floats.
// rs (round 6, deferred-1/rms RMSNorm): per-row scale of the block's MT*16 rows in LDS (outside the transposition tile), or nullptr.  The A operand
// then was bf16(x . w_norm) WITHOUT the 1/rms; rs[row] = rsqrt(mean(x^2) + eps) commutes with the GEMM and is applied to the fp32 result here.
template <int MT, int NW>
__device__ __forceinline__ void skinny_store_tile(char* smem, const f32x4 (&acc)[MT], float* __restrict__ o, int M, int N,
                                                  int mbase, int nbase, int w, int g, int lr, int tid, int wt = 0, const float* rs = nullptr) {
    constexpr int LD = NW * 16 + 4, NTH = NW * 64, V4 = NW * 4;        // float4 per tile row
    float* t = (float*)smem;
    __syncthreads();                                   // every wave is done reading the x tiles
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[(mt * 16 + g * 4 + r) * LD + w * 16 + lr] = acc[mt][r];
    __syncthreads();
    for (int v = tid; v < MT * 16 * V4; v += NTH) {
        const int row = v / V4, c4 = (v % V4) * 4;
        const int m = mbase + row, n = nbase + c4;
        if (m < M && n < N) {
            f32x4 v4 = *(const f32x4*)(t + row * LD + c4);
            if (rs) { const float r_ = rs[row]; v4.x *= r_; v4.y *= r_; v4.z *= r_; v4.w *= r_; }
            float* p = o + (long)m * N + n;
            // wt: write-through (sc1) -- the slab leaves this XCD's L2 while the kernel runs, not as dirty lines at the boundary
            // `s_nop 1` INSIDE the statement: hipcc does not know this is a 128-bit VMEM store, so it neither keeps the data registers
            // alive nor pads the store-data hazard (a VALU write to them needs 2 wait states); see sk4_store_direct
            if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v4) : "memory");
            else __builtin_nontemporal_store(v4, (f32x4*)p);
        }
    }
}

// SwiGLU epilogue (S == 1 only): every 16-column n-tile is [8 gate | 8 up] (weights interleaved
// in blocks of 8 at load time), so h = silu(g) * u for 8 output columns per tile comes straight
// out of the transposed LDS tile; bf16 h [M, I] is written, no fp32 slab, no extra kernel.
template <int MT, int NW>
__device__ __forceinline__ void skinny_store_swiglu(char* smem, const f32x4 (&acc)[MT], bf16* __restrict__ h, int M, int I,
                                                    int mbase, int nblk, int w, int g, int lr, int tid, const float* rs = nullptr) {
    constexpr int LD = NW * 16 + 4, NTH = NW * 64, OC = NW * 8;        // outputs per tile row
    float* t = (float*)smem;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[(mt * 16 + g * 4 + r) * LD + w * 16 + lr] = acc[mt][r];
    __syncthreads();
    for (int v = tid; v < MT * 16 * (OC / 2); v += NTH) {
        const int row = v / (OC / 2), c2 = (v % (OC / 2)) * 2;          // c2 in [0, OC), even
        const int m = mbase + row, col = nblk * OC + c2;
        if (m < M && col < I) {
            const int tc = (c2 >> 3) * 16 + (c2 & 7);                   // gate column in the tile
            const float r_ = rs ? rs[row] : 1.f;                        // deferred 1/rms of the row (x 1.0f is exact: the undeferred form keeps its bits)
            const float g0 = t[row * LD + tc] * r_, g1 = t[row * LD + tc + 1] * r_;
            const float u0 = t[row * LD + tc + 8] * r_, u1 = t[row * LD + tc + 9] * r_;
            const float h0 = (g0 / (1.f + expf(-g0))) * u0, h1 = (g1 / (1.f + expf(-g1))) * u1;
            *(uint32_t*)(h + (long)m * I + col) = pack_bf16x2(h0, h1);
        }
    }
}

// MFMA phase of one 128-wide K chunk: A fragments (x tile in LDS, row stride SK_ROWB) are read
// one m-tile PAIR ahead of the MFMAs that consume them (double-buffered registers), and the
// two m-tiles of a pair alternate accumulators so no MFMA waits on the previous one's result.
// Without this hipcc serialises ds_read -> s_waitcnt lgkmcnt(0) -> mfma through one fragment
// register (measured ~80 clk per MFMA per SIMD instead of ~17).
template <int MT>
__device__ __forceinline__ void skinny_mfma_chunk(const char* xt, int lr, int g, const bf16x8 (&wc)[4], f32x4 (&acc)[MT]) {
    const char* rp = xt + lr * SK_ROWB + g * 16;
    if constexpr (MT == 1) {
        bf16x8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(rp + i * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], wc[i], acc[0], 0, 0, 0);
    } else {
        constexpr int NP = MT / 2;
        bf16x8 af[2][2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][h][i] = *(const bf16x8*)(rp + h * 16 * SK_ROWB + i * 64);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p + 1 < NP) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        af[(p + 1) & 1][h][i] = *(const bf16x8*)(rp + ((p + 1) * 2 + h) * 16 * SK_ROWB + i * 64);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the next pair's 8 reads ahead of this pair's MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p & 1][0][i], wc[i], acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p & 1][1][i], wc[i], acc[2 * p + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MT>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W,
                                                         float* __restrict__ out, int M, int N, int K, int nck) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 * MT*16*SK_ROWB
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lr = l & 15;
    const int split = blockIdx.y, mblk = blockIdx.z;
    const int mbase = mblk * 128;
    const int n = blockIdx.x * 64 + w * 16 + lr;
    // fragment order: k-step i, lane (lr, g) holds k = i*32 + g*8 .. +8, so the 4 lanes of a W row
    // read one full 64-byte sector per load instruction
    const bf16* wp = W + (long)(n < N ? n : N - 1) * K + g * 8;
    const int kbeg = split * nck * SK_BK;
    constexpr int XB = MT * 16 * SK_ROWB;

    u32x4 xs[MT];                     // staging registers for the x tile: MT x 16 B per thread
    auto xload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int v = tid + j * 256, row = v >> 4, cv = v & 15;
            const int m = mbase + row;
            if (m < M) xs[j] = *(const u32x4*)(x + (long)m * K + k0 + cv * 8);
            else xs[j] = (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto xstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int v = tid + j * 256, row = v >> 4, cv = v & 15;
            *(u32x4*)(smem + buf * XB + row * SK_ROWB + cv * 16) = xs[j];
        }
    };
    bf16x8 wc[4], wn[4];
    auto wload = [&](bf16x8* dst, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *(const bf16x8*)(wp + k0 + i * 32);
    };
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    wload(wc, kbeg);
    xload(kbeg);
    xstore(0);
    __syncthreads();
    for (int c = 0; c < nck; ++c) {
        const int knext = kbeg + (c + 1) * SK_BK;
        const bool more = c + 1 < nck;
        if (more) { wload(wn, knext); xload(knext); }
        const char* xt = smem + (c & 1) * XB;
        skinny_mfma_chunk<MT>(xt, lr, g, wc, acc);
        if (more) {
            xstore((c + 1) & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = wn[i];
        }
        __syncthreads();
    }
    skinny_store_tile<MT, 4>(smem, acc, out + (long)split * M * N, M, N, mbase, blockIdx.x * 64, w, g, lr, tid);
}

// ------------------------------------------------------------------------------- skinny GEMM v3
// Same geometry as v1 (BN = 64: 4 waves x 16 columns, BK = 128) but the W stream is decoupled
// from the per-chunk barrier: a register ring of depth D keeps D chunks of every wave's W
// rows in flight (the HBM requests of chunk c+D are issued while chunk c computes), the chunk
// loop is fully unrolled (NCK = chunks per block, compile time) so the ring is statically
// indexed.  XDB: x tile double-buffered (2 blocks/CU at MT=8) or single-buffered (4 blocks/CU).
constexpr int sk3_main_lds(int MT, int NW, bool XDB) {
    const int XL = (XDB ? 2 : 1) * MT * 16 * SK_ROWB, TL = MT * 16 * (NW * 16 + 4) * 4;
    return XL > TL ? XL : TL;
}
// XA (round 6): x prefetch distance in chunks (1 = rounds 1-5).  XA = 2: the staging registers of x(c+2) are loaded at the top of chunk c, so the wait for x(c+1) at the
// end of chunk c finds it issued a whole chunk earlier AND -- loads return in order -- no longer retires W(c+1), x(c+2), W(c+2), which are all younger.
// NTW (round 6 experiment): the weight fragments with the non-temporal hint (the v4 kernels' `nt`); every W tile here is read by TWO row blocks.
template <int MT, int NCK, int D, bool XDB, int EPI, int NW, bool TILED = false, bool PROF = false, int XA = 1, bool NTW = false>      // PROF (libplangen_diag.so only): per-wave cycle stamps through the ssq pointer
__global__ __launch_bounds__(64 * NW, 2) void gemm_skinny3_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W,
                                                               float* __restrict__ out, const float* __restrict__ ssq_, int M, int N, int K, int wt, float eps) {
    // ssq (round 6): deferred-1/rms RMSNorm -- x is bf16(residual . w_norm), ssq [M][8] holds 8 partial sums of squares of every residual row
    // (rmsnorm_defer_kernel); the block turns its rows' partials into 1/rms while the first weight chunks are in flight and scales its fp32 result.
    // nullptr: x is the normalised activation (every other caller).  4 pointers + 4 x 32 bits + eps = 52 bytes: still one preloaded kernarg block.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XB = MT * 16 * SK_ROWB, NTH = 64 * NW, BN = 16 * NW;
    constexpr int RS_OFF = sk3_main_lds(MT, NW, XDB);                   // the row scales live behind the x tiles / the transposition tile
    const float* const ssq = PROF ? nullptr : ssq_;
    // PROF: 64 stamp slots per wave, wave index = linear block index * NW + wave (tools/sk3_profile.py)
    int pslot = 0;
    unsigned long long* pw = PROF && ssq_ ? (unsigned long long*)ssq_ + ((long)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + (threadIdx.x >> 6)) * 64 : nullptr;
    auto stamp = [&]() { if constexpr (PROF) { if (pw && (threadIdx.x & 63) == 0 && pslot < 64) pw[pslot] = __builtin_readcyclecounter(); ++pslot; } };
    stamp();                                                            // 0: wave start
    constexpr int XV = (MT * 256 + NTH - 1) / NTH;                     // x vectors per thread per chunk
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lr = l & 15;
    const int split = blockIdx.y, mbase = blockIdx.z * (MT * 16);
    const int bx = blockIdx.x;
    const int n = bx * BN + w * 16 + lr;
    const int kbeg = split * NCK * SK_BK;
    // Row-major W: lane (lr, g) reads W[n][k + i*32 + g*8 ..+8].  TILED W (decode copy, built at load
    // time): [n-tile][k-chunk][i][lane][8] -- every wave load instruction is one contiguous 1 KiB and
    // an n-tile's whole K stream is contiguous in HBM (DRAM-page friendly).
    const int ntile = bx * NW + w, ntiles = (N + 15) / 16;
    const bf16* wp = TILED ? W + ((long)(ntile < ntiles ? ntile : ntiles - 1) * (K / SK_BK) + split * NCK) * 2048 + l * 8
                           : W + (long)(n < N ? n : N - 1) * K + kbeg + g * 8;
    constexpr int WCH = TILED ? 2048 : SK_BK, WI = TILED ? 512 : 32;       // element strides per chunk / per k-step
    const bf16* xp = x + kbeg;

    static_assert(XA == 1 || XA == 2, "x prefetch distance");
    u32x4 xsr[XA][XV];
    auto xload = [&](int c) {
        u32x4 (&xs)[XV] = xsr[c % XA];
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int v = tid + j * NTH, row = v >> 4, cv = v & 15;
            const int m = mbase + row;
            // BRANCH-FREE (row clamped into the matrix; rows >= M are never stored): behind `if (m < M)` hipcc put every x load
            // in its own exec-masked block with `s_waitcnt vmcnt(0)` at the joins -- four serialised memory round trips (and a drain
            // of the W ring's first loads) before the first MFMA
            if constexpr ((MT * 256) % NTH == 0) {
                xs[j] = *(const u32x4*)(xp + (long)(m < M ? m : M - 1) * K + c * SK_BK + cv * 8);
            } else {
                // tile does not divide over the block (NW = 6: 1 024 vectors over 384 threads): the surplus threads re-read the tile's last row -- an
                // UNCONDITIONAL load (round 6; a conditional one costs a vmcnt(0) at the join), dropped by the guarded xstore
                const int mc = mbase + (row < MT * 16 ? row : MT * 16 - 1);
                xs[j] = *(const u32x4*)(xp + (long)(mc < M ? mc : M - 1) * K + c * SK_BK + cv * 8);
            }
        }
    };
    auto xstore = [&](int buf, int c) {
        u32x4 (&xs)[XV] = xsr[c % XA];
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int v = tid + j * NTH, row = v >> 4, cv = v & 15;
            // unconditional when the tile divides over the block (hipcc cannot prove tid < NTH, keeps the guard, and SINKS the matching
            // x load into it: a conditional load with a `vmcnt(0)` at the join)
            if ((MT * 256) % NTH == 0 || row < MT * 16) *(u32x4*)(smem + buf * XB + row * SK_ROWB + cv * 16) = xs[j];
        }
    };
    auto wld = [&](const bf16* p) __attribute__((always_inline)) -> bf16x8 {
        if constexpr (NTW) return __builtin_nontemporal_load((const bf16x8*)p);
        else return *(const bf16x8*)p;
    };
    bf16x8 wr[D][4];
#pragma unroll
    for (int c = 0; c < D && c < NCK; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[c][i] = wld(wp + c * WCH + i * WI);
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xload(0);
    if constexpr (XA == 2 && NCK > 1) xload(1);
    stamp();                                                            // 1: W ring prologue + x chunk 0 issued
    float* const rs = ssq ? (float*)(smem + RS_OFF) : nullptr;
    if (ssq && tid < MT * 16) {                                        // fixed summation order: the scale of a row does not depend on the block that computes it
        const int m = mbase + tid;
        const float* pp = ssq + (long)(m < M ? m : M - 1) * 8;
        const f32x4 a = *(const f32x4*)pp, b = *(const f32x4*)(pp + 4);
        rs[tid] = rsqrtf((((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) / (float)K + eps);
    }
    xstore(0, 0);
    stamp();                                                            // 2: x chunk 0 arrived and written to LDS
    __syncthreads();
    stamp();                                                            // 3: block barrier
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
        if (c + XA < NCK) xload(c + XA);
        const char* xt = smem + (XDB ? (c & 1) * XB : 0);
        skinny_mfma_chunk<MT>(xt, lr, g, wr[c % D], acc);
        if constexpr (PROF) stamp();                                    // 4 + 3c: fragment reads + MFMAs of chunk c issued (W(c) was already in registers: the x wait below retires it, loads return in order)
        if (c + D < NCK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[c % D][i] = wld(wp + (c + D) * WCH + i * WI);
        }
        if (c + 1 < NCK) {
            if (!XDB) __syncthreads();
            xstore(XDB ? ((c + 1) & 1) : 0, c + 1);
            if constexpr (PROF) stamp();                                // 5 + 3c: x(c+1) -- and with it every older load: W(c+1) -- landed, tile written
            __syncthreads();
        } else if constexpr (PROF) stamp();
        if constexpr (PROF) stamp();                                    // 6 + 3c: block barrier
    }
    if constexpr (EPI == 1) skinny_store_swiglu<MT, NW>(smem, acc, (bf16*)out, M, N / 2, mbase, bx, w, g, lr, tid, rs);
    else skinny_store_tile<MT, NW>(smem, acc, out + (long)split * M * N, M, N, mbase, bx * BN, w, g, lr, tid, wt, rs);
    if constexpr (PROF) { stamp(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); }      // epilogue issued / stores acknowledged
}
// Deferred-1/rms RMSNorm site handed down the decode GEMM dispatch (round 6): ssq [M][8] partial sums of squares + eps; null = off.
struct SkRowScale { const float* ssq = nullptr; float eps = 0.f; };
template <int MT, int NCK, int D, bool XDB, int EPI = 0, int NW = 4, bool TILED = false, bool PROF = false, int XA = 1, bool NTW = false>
static void launch_sk3(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, SkRowScale rsc = {}) {
    constexpr int LDS = sk3_main_lds(MT, NW, XDB) + MT * 16 * 4;      // + the row scales (used only with rsc.ssq)
    auto kfn = gemm_skinny3_kernel<MT, NCK, D, XDB, EPI, NW, TILED, PROF, XA, NTW>;
    (void)PG_DYN_LDS(kfn, LDS);
    dim3 grid((N + 16 * NW - 1) / (16 * NW), S, (M + MT * 16 - 1) / (MT * 16)), block(64 * NW);
    hipLaunchKernelGGL(kfn, grid, block, LDS, s, x, W, out, rsc.ssq, M, N, K, pg_tune->wt_store & 1, rsc.eps);
}
template <int D, bool XDB>
static int sk3_dispatch(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S) {
    const int nck = K / SK_BK / S;
    if (nck * S * SK_BK != K) return 0;
    switch (nck) {
        case 1: launch_sk3<8, 1, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        case 2: launch_sk3<8, 2, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        case 4: launch_sk3<8, 4, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        case 8: launch_sk3<8, 8, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        case 11: launch_sk3<8, 11, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        case 16: launch_sk3<8, 16, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        case 22: launch_sk3<8, 22, D, XDB>(s, x, W, out, M, N, K, S); return 1;
        default: return 0;
    }
}


// ---- production dispatch: v3 (W register ring depth 2, double-buffered x tile) ----
// x prefetched two chunks ahead, W ring 3 (round 6): the 64-row x 128-column wide-N block whose x wait no longer drains the W ring; same K order = same bits
template <int EPI>
static bool sk3_nt_nck(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, int nck, SkRowScale rsc = {}) {      // production block, W with the nt hint
    switch (nck) {
        case 4: launch_sk3<4, 4, 2, true, EPI, 8, true, false, 1, true>(s, x, W, out, M, N, K, S, rsc); return true;
        case 8: launch_sk3<4, 8, 2, true, EPI, 8, true, false, 1, true>(s, x, W, out, M, N, K, S, rsc); return true;
        case 16: launch_sk3<4, 16, 2, true, EPI, 8, true, false, 1, true>(s, x, W, out, M, N, K, S, rsc); return true;
        default: return false;
    }
}
template <int EPI>
static bool sk3_xa2_nck(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, int nck, SkRowScale rsc = {}) {
    switch (nck) {
        case 4: launch_sk3<4, 4, 3, true, EPI, 8, true, false, 2>(s, x, W, out, M, N, K, S, rsc); return true;
        case 8: launch_sk3<4, 8, 3, true, EPI, 8, true, false, 2>(s, x, W, out, M, N, K, S, rsc); return true;
        case 16: launch_sk3<4, 16, 3, true, EPI, 8, true, false, 2>(s, x, W, out, M, N, K, S, rsc); return true;
        default: return false;
    }
}
template <int MT, int EPI, int NW = 4, bool TILED = false, int D = 2>
static bool sk3_prod_nck(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, int nck, SkRowScale rsc = {}) {
    switch (nck) {
        case 1: launch_sk3<MT, 1, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        case 2: launch_sk3<MT, 2, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        case 4: launch_sk3<MT, 4, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        case 8: launch_sk3<MT, 8, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        case 11: launch_sk3<MT, 11, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        case 16: launch_sk3<MT, 16, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        case 22: launch_sk3<MT, 22, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S, rsc); return true;
        default: return false;
    }
}
// ------------------------------------------------------------------------------- skinny GEMM v4 ("stream")
// Same block tile as v3 (BN = 16*NW columns, MT*16 rows, BK = 128, tiled W, W register ring of depth WD) but
// the x tile never touches a VGPR: every 128-wide K chunk of the block's MT*16 rows is brought HBM/L2 -> LDS by
// LDS-DMA (global_load_lds, 1 KiB per wave instruction) into a ring of XD chunk slots, XD-1 chunks AHEAD of
// the MFMAs that read it.  v3 staged x through registers one chunk ahead: a block's every chunk then waited
// for an L2 round trip (measured: x-tile loads = 3.6 of the qkv kernel's 15 us); here that latency sits under
// XD-1 chunks of MFMA + W streaming, the xstore LDS writes and their staging registers are gone, and there is
// ONE s_barrier per chunk (no vmcnt(0)/lgkmcnt(0) drain: raw s_barrier + counted vmcnt).
//   LDS chunk slot: MT*16 rows x 256 B, no padding (the DMA writes 64 consecutive 16-byte slots per
//   instruction = 4 rows); bank conflicts are avoided by permuting the SOURCE: slot s of row r holds logical
//   16-byte chunk s ^ (r & 15), so the fragment read of lane (lr, g), k-step i is at
//   row*256 + (((4i + g) ^ lr) << 4): 16 distinct slots per 16-lane group.
//   Program order of VMEM ops per wave (all counted by hand, W loads by the compiler):
//     prologue  X(0) .. X(XD-2), W(0) .. W(WD-1)
//     chunk c   s_waitcnt vmcnt(A(c))  -> this wave's X(c) pieces have landed        (A = ops issued after X(c))
//               s_barrier              -> every wave's X(c) pieces have landed, every wave is done reading X(c-1)
//               issue X(c+XD-1) into the slot of X(c-1)
//               MFMA on X(c), W(c)     (the compiler waits for W(c))
//               issue W(c+WD) into W(c)'s registers
constexpr int sk4_wait_count(int c, int NCK, int XD, int WD, int MT, int ahead = 0) {      // ahead = 1: chunk c+1 (not only c) retired at chunk c's barrier
    int ops = 0, lastX[64] = {};
    for (int p = 0; p < XD - 1 && p < NCK; ++p) { ops += MT; lastX[p] = ops; }
    for (int p = 0; p < WD && p < NCK; ++p) ops += 4;
    for (int it = 0; it < NCK; ++it) {
        if (it == c) return ops - lastX[(c + ahead < NCK) ? c + ahead : NCK - 1];
        if (it + XD - 1 < NCK) { ops += MT; lastX[it + XD - 1] = ops; }
        if (it + WD < NCK) ops += 4;
    }
    return 0;
}
// ops issued after W(c)'s last load at the point of chunk c where the MFMAs start (after X(c+XD-1) was issued)
constexpr int sk4_wait_count_w(int c, int NCK, int XD, int WD, int MT) {
    int ops = 0, lastW[64] = {};
    for (int p = 0; p < XD - 1 && p < NCK; ++p) ops += MT;
    for (int p = 0; p < WD && p < NCK; ++p) { ops += 4; lastW[p] = ops; }
    for (int it = 0; it < NCK; ++it) {
        if (it + XD - 1 < NCK) ops += MT;
        if (it == c) return ops - lastW[c];
        if (it + WD < NCK) { ops += 4; lastW[it + WD] = ops; }
    }
    return 0;
}
// W fragment load as OPAQUE asm: the compiler may schedule a plain load from a const __restrict__ pointer across an
// `asm volatile("" ::: "memory")` fence (nothing can alias it), which silently changes the VMEM issue order the hand-counted vmcnt
// waits of the v4 kernel assume.  asm volatile statements keep their program order, so the simulated order of sk4_wait_count() is the
// order in the instruction stream.  (Round 2 attributed a "stale 4-row x piece on a cold first launch" to this; round 4 found that symptom's
// real cause in the epilogue's asm store, see sk4_store_direct.  The ordering argument stands on its own.)
__device__ __forceinline__ void sk4_wload(bf16x8& dst, const bf16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory");
}
// wait until at most N_ VMEM ops are outstanding; the W registers are in/out operands so no consumer of them can be scheduled above
template <int N_> __device__ __forceinline__ void wait_vmcnt_w(bf16x8 (&wv)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]) : "n"(N_ > 63 ? 63 : N_) : "memory");
}
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_ > 63 ? 63 : N_) : "memory"); }   // 6-bit field: clamping only waits longer

template <bool SWAP>
__device__ __forceinline__ f32x4 sk4_mfma(const bf16x8& a, const bf16x8& wv, const f32x4& c) {
    // SWAP: D = W . x^T -- the lane then holds 4 consecutive output COLUMNS of one row (16-byte epilogue stores)
    if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, a, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wv, c, 0, 0, 0);
}
template <int MT, bool SWAP = false, int ROWX = 0, bool ROT = false>
__device__ __forceinline__ void sk4_mfma_chunk(const char* xt, int lr, int g, const bf16x8 (&wc)[4], f32x4 (&acc)[MT]) {
    // A fragment (m-tile mt, k-step i): row mt*16 + lr, logical 16-byte chunk 4i + g, swizzled by lr
    const char* rp = xt + (lr ^ ROWX) * 256;
    int so[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) so[i] = (ROT ? ((i * 4 + g + lr) & 15) : ((i * 4 + g) ^ lr)) << 4;
    if constexpr (MT == 1) {
        bf16x8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(rp + so[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = sk4_mfma<SWAP>(a[i], wc[i], acc[0]);
        __builtin_amdgcn_sched_barrier(0);      // the ds_reads / MFMAs of a chunk stay between its barrier and the next one (the DMA that re-stages the slot follows that barrier)
    } else {
        constexpr int NP = MT / 2;
        bf16x8 af[2][2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][h][i] = *(const bf16x8*)(rp + h * 4096 + so[i]);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p + 1 < NP) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        af[(p + 1) & 1][h][i] = *(const bf16x8*)(rp + ((p + 1) * 2 + h) * 4096 + so[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * p] = sk4_mfma<SWAP>(af[p & 1][0][i], wc[i], acc[2 * p]);
                acc[2 * p + 1] = sk4_mfma<SWAP>(af[p & 1][1][i], wc[i], acc[2 * p + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// Epilogues for the swapped accumulator layout: lane (lr, g) holds out[row mt*16 + lr][n-tile column 4g .. 4g+3].
// WT: write-through (sc1) stores -- the slab leaves the XCD's L2 while the kernel still runs instead of as dirty lines
// at the kernel boundary (the boundary pays ~0.2-0.4 us per dirty MB).
template <int MT, bool WT = false>
__device__ __forceinline__ void sk4_store_direct(const f32x4 (&acc)[MT], float* __restrict__ o, int M, int N, int mbase, int ncol0, int g, int lr) {
    const int n = ncol0 + g * 4;
    if (n >= N) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mbase + mt * 16 + lr;
        if (m < M) {
            float* p = o + (long)m * N + n;
            // ROOT CAUSE of the "stale x piece" failures of rounds 2-3 (found in round 4 by running the kernel under a concurrent memory
            // load, tools/sk4_load_stress.py): this statement used to be the bare store.  hipcc treats an asm statement as opaque -- it does
            // not know a 128-bit VMEM store reads its data registers for two more issue slots -- and re-used acc[mt]'s first register for the
            // NEXT m-tile's row index one instruction later (`v_or_b32 v12, 16, v18` behind `global_store_dwordx4 .., v[12:15]`).  When the
            // memory pipeline is back-pressured (cold launch, another stream streaming) the store then wrote the clobbered dword for the
            // lanes it reads last: rows 12-15 of every m-tile except the block's last one (whose store is followed by s_endpgm) lost a
            // whole split's contribution -- exactly the "rows 12-15 / 28-31 / 44-47" signature that was blamed on the LDS-DMA staging.
            // With accumulators in AGPRs (another register budget) the data goes through fresh VGPRs and the fault disappears, which is
            // how it was isolated.  Fix = the ISA's required wait states inside the statement (CDNA guide 5.7: an asm store_dwordx3/x4
            // ends with `s_nop 1`).
            if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(acc[mt]) : "memory");
            else __builtin_nontemporal_store(acc[mt], (f32x4*)p);
        }
    }
}
// SwiGLU: an n-tile is [8 gate | 8 up]: lanes g = 0,1 hold gate columns 4g..4g+3, lanes g + 2 the matching up columns
template <int MT>
__device__ __forceinline__ void sk4_store_swiglu_direct(const f32x4 (&acc)[MT], bf16* __restrict__ h, int M, int I, int mbase, int ntile, int g, int lr) {
    const int col = ntile * 8 + (g & 1) * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 u;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = __shfl_xor(acc[mt][r], 32, 64);
        const int m = mbase + mt * 16 + lr;
        if (g < 2 && m < M && col < I) {
            float hv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float gt = acc[mt][r]; hv[r] = (gt / (1.f + expf(-gt))) * u[r]; }
            u32x2 pk; pk.x = pack_bf16x2(hv[0], hv[1]); pk.y = pack_bf16x2(hv[2], hv[3]);
            *(u32x2*)(h + (long)m * I + col) = pk;
        }
    }
}

template <int C, int NCK, int XD, int WD, int MT, class F> __device__ __forceinline__ void sk4_static_for(F&& f) {
    if constexpr (C < NCK) { f(std::integral_constant<int, C>{}); sk4_static_for<C + 1, NCK, XD, WD, MT>(f); }
}

// ---- EPI 5 (round 5 experiment): split-K producer with the consumer norm's reduction folded into its TAIL.
// Every split block stores its slab tile write-through, waits for the acknowledgements and takes a ticket of its (row block, column block)
// tile; the block that draws the LAST ticket re-reads all S slab tiles (fixed order s = 0 .. S-1, loads that bypass this XCD's L2: the other
// splits ran on other XCDs), adds the residual and writes x, xw = bf16(x . w_norm) -- the consumer GEMM's A operand WITHOUT the 1/rms, which
// commutes with the GEMM and is applied in the consumer's epilogue -- and adds the tile's per-row sums of squares to ssq[row] as 2^-28
// fixed point (integer atomics: the order of arrival cannot change the sum).  No grid barrier, no polling: the other S - 1 blocks exit.
struct SkFuse {                        // device-resident, one per norm site
    float* x;                          // residual stream [M][N] fp32, in / out
    const bf16* w;                     // the consumer norm's weight [N]
    bf16* xw;                          // out [M][N]: bf16(x_new * w)
    unsigned long long* ssq;           // [M] += sum_n x_new^2 * 2^28 (zeroed once per decode step)
    unsigned* ticket;                  // [row blocks][column blocks], self-resetting (atomicInc wraps at S - 1)
    int32_t* advance;                  // optional decode step counter, bumped by tile (0, 0)'s finisher
};
#define SK_SSQ_SCALE 268435456.f       // 2^28
__device__ __forceinline__ f32x4 sk_load_sc1(const float* p) {          // device-scope load: never served from this XCD's (non-coherent) L2
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MTW>
__device__ __forceinline__ void sk4_finish_tile(const SkFuse* __restrict__ fsp, const float* __restrict__ slabs, int S, int M, int N, int mb, int ncol0,
                                                int g, int lr, int w, int tid, float* red /* >= 4 * MTW * 16 floats of LDS */) {
    const SkFuse fs = *fsp;
    const int n = ncol0 + g * 4;
    f32x4 v[MTW]; float ss[MTW];
    const u32x2 wv = *(const u32x2*)(fs.w + (n < N ? n : 0));
    long o[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = mb + mt * 16 + lr, mc = m < M ? m : M - 1;
        o[mt] = (long)mc * N + (n < N ? n : 0);
        v[mt] = *(const f32x4*)(fs.x + o[mt]);                          // x was written by an earlier kernel: ordinary load
    }
    // every slab request of the tile (MTW m-tiles x up to 4 slabs per pass) in flight together: ONE memory round trip per pass, added in slab order
    for (int s0 = 0; s0 < S; s0 += 4) {
        f32x4 t[MTW][4];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int u = 0; u < 4; ++u) t[mt][u] = sk_load_sc1(slabs + (long)(s0 + u < S ? s0 + u : S - 1) * M * N + o[mt]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int u = 0; u < 4; ++u) { asm volatile("" : "+v"(t[mt][u])); if (s0 + u < S) v[mt] += t[mt][u]; }
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = mb + mt * 16 + lr;
        float q = v[mt][0] * v[mt][0] + v[mt][1] * v[mt][1] + v[mt][2] * v[mt][2] + v[mt][3] * v[mt][3];
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);         // the n-tile's 16 columns (4 lane groups)
        ss[mt] = q;
        if (m < M && n < N) {
            const long o = (long)m * N + n;
            *(f32x4*)(fs.x + o) = v[mt];
            u32x2 ov;
            ov.x = pack_bf16x2(bf16_lo(wv.x) * v[mt][0], bf16_hi(wv.x) * v[mt][1]);
            ov.y = pack_bf16x2(bf16_lo(wv.y) * v[mt][2], bf16_hi(wv.y) * v[mt][3]);
            *(u32x2*)(fs.xw + o) = ov;
        }
    }
    // the block's four n-tile waves -> one fixed-point atomic per row
    if (g == 0) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) red[(w & 3) * (MTW * 16) + mt * 16 + lr] = ss[mt];
    }
    __syncthreads();
    if (tid < MTW * 16) {
        const int m = mb + tid;
        const float q = (red[tid] + red[MTW * 16 + tid]) + (red[2 * MTW * 16 + tid] + red[3 * MTW * 16 + tid]);
        if (m < M) atomicAdd(fs.ssq + m, (unsigned long long)__float2ull_rn(q * SK_SSQ_SCALE));
    }
    if (fs.advance && tid == 0 && blockIdx.x == 0 && blockIdx.z == 0) *fs.advance += 1;
}

// MS = 2: 8 waves per block, wave w = (n-tile w & 3, row half w >> 2): two waves per SIMD, so one wave's LDS fragment
// reads run under the other's MFMAs (measured with 4 lock-stepped waves: reads and MFMAs of a chunk serialise,
// ~1050 cycles per chunk instead of ~550).  Both row halves load the same W fragments (second one hits L1/L2).
template <int MT, int NCK, int XD, int WD, int EPI, int OCC, int ABL = 0, int MS = 1>      // ABL (bench only): 1 no x DMA / barriers, 2 no MFMA, 4 no stores, 32 per-wave s_memtime stamps
__global__ __launch_bounds__(256 * MS, OCC) void gemm_sk4_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W,
                                                               float* __restrict__ out, int M, int N, int K, unsigned long long* prof) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MTW = MT / MS;                                     // m-tiles per wave
    // tuning aid: per-wave s_memtime stamps (prof != nullptr only from the microbenchmark): 64 slots per wave
    constexpr bool PROF = (ABL & 32) != 0;
    int pslot = 0;
    unsigned long long* pw = (PROF && prof) ? prof + ((long)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (4 * MS) + (threadIdx.x >> 6)) * 64 : nullptr;
    auto stamp = [&]() { if constexpr (PROF) { if (pw && (threadIdx.x & 63) == 0 && pslot < 64) pw[pslot] = __builtin_readcyclecounter(); ++pslot; } };
    stamp();
    if constexpr (ABL & 8192) return;                                // ABL 8192 (bench only): an EMPTY kernel of the same geometry -- what the launch itself costs
    constexpr int XB = MT * 16 * 256;                                // bytes per x chunk slot
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, lr = l & 15;
    const int wn = w & 3, wm = w >> 2;
    const int split = blockIdx.y, mbase = blockIdx.z * (MT * 16);
    const int ntile = blockIdx.x * 4 + wn, ntiles = (N + 15) / 16;
    const bf16* wp = W + ((long)(ntile < ntiles ? ntile : ntiles - 1) * (K / SK_BK) + split * NCK) * 2048 + l * 8;
    // x DMA: wave w owns pieces w*MTW .. w*MTW+MTW-1 of every chunk; piece q = rows 4q .. 4q+3
    const bf16* xsrc[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int r = ((ABL & 1024) ? (j * 4 + w) : (w * MTW + j)) * 4 + (l >> 4);       // ABL 1024 (hazard screen): wave w owns pieces w, w+4, w+8, ... (one per 4 KiB LDS page)
        const int m = mbase + r;
        // ABL 4096 (hazard screen): rotation swizzle -- LDS position p of row r holds logical chunk (p - r) & 15 -- instead of the XOR swizzle
        xsrc[j] = x + (long)(m < M ? m : M - 1) * K + split * NCK * SK_BK + (((ABL & 4096) ? (((l & 15) - (r & 15)) & 15) : ((l & 15) ^ (r & 15))) << 3);
    }
    auto issueX = [&](int c) {
        char* slot = smem + (c % XD) * XB + w * (MTW * 1024);
#pragma unroll
        for (int jj = 0; jj < MTW; ++jj) {
            const int j = (ABL & 128) ? MTW - 1 - jj : jj;                       // ABL 128 (hazard screen): pieces issued in reverse order
            if constexpr (ABL & 1024) glds16(xsrc[j] + c * SK_BK, smem + (c % XD) * XB + (j * 4 + w) * 1024);
            else if constexpr (ABL & 2048) glds16(xsrc[j] + c * SK_BK, slot + (j ^ 3) * 1024);     // ABL 2048 (hazard screen): rows 4j..4j+3 of a 16-row group stored at LDS rows (4j..4j+3) ^ 12
            else
            glds16(xsrc[j] + c * SK_BK, slot + j * 1024);
            if constexpr (ABL & 256) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // ABL 256 (hazard screen): idle issue slots behind every DMA
        }
        if constexpr (ABL & 512) glds16(xsrc[0] + c * SK_BK, smem + XD * XB + w * 1024);   // ABL 512 (hazard screen): one more DMA into a dummy 1 KiB per wave behind the group
    };
    bf16x8 wr[WD][4];
    f32x4 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (!(ABL & 1)) {
#pragma unroll
        for (int c = 0; c < XD - 1 && c < NCK; ++c) issueX(c);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < WD && c < NCK; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) sk4_wload(wr[c][i], wp + c * 2048 + i * 512);
    stamp();
    sk4_static_for<0, NCK, XD, WD, MTW>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (!(ABL & 1)) {
            if constexpr (ABL & 8) wait_vmcnt<0>(); else wait_vmcnt<sk4_wait_count(c, NCK, XD, WD, MTW, (ABL & 64) ? 1 : 0)>();      // ABL 8 (bench): drain everything
            stamp();
            __builtin_amdgcn_s_barrier();
            stamp();
            if constexpr (c + XD - 1 < NCK) issueX(c + XD - 1);
        }
        wait_vmcnt_w<(ABL & 1) ? 0 : sk4_wait_count_w(c, NCK, XD, WD, MTW)>(wr[c % WD]);      // W(c) landed (younger loads stay in flight)
        stamp();
        if constexpr (!(ABL & 2)) sk4_mfma_chunk<MTW, (EPI >= 2), (ABL & 2048) ? 12 : 0, (ABL & 4096) != 0>(smem + (c % XD) * XB + wm * (MTW * 4096), lr, g, wr[c % WD], acc);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[0][0] += (float)wr[c % WD][i][0]; }
        }
        if constexpr (c + WD < NCK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sk4_wload(wr[c % WD][i], wp + (c + WD) * 2048 + i * 512);
        }
        stamp();
    });
    if constexpr (ABL & 4) { if (acc[0][0] == 123.456f) out[tid] = acc[0][0]; return; }
    const int mb = mbase + wm * MTW * 16;
    if constexpr (EPI == 3) sk4_store_swiglu_direct<MTW>(acc, (bf16*)out, M, N / 2, mb, ntile, g, lr);
    else if constexpr (EPI == 2) sk4_store_direct<MTW>(acc, out + (long)split * M * N, M, N, mb, ntile * 16, g, lr);
    else if constexpr (EPI == 4) sk4_store_direct<MTW, true>(acc, out + (long)split * M * N, M, N, mb, ntile * 16, g, lr);
    else if constexpr (EPI == 5) {
        static_assert(EPI != 5 || MS == 1, "fused-norm producer: MS = 1");
        const int S = gridDim.y;
        const SkFuse* fsp = (const SkFuse*)prof;                          // EPI 5: the last kernel argument is the norm site
        sk4_store_direct<MTW, true>(acc, out + (long)split * M * N, M, N, mb, ntile * 16, g, lr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's write-through stores are acknowledged
        __syncthreads();                                                  // ... and every wave's of the block
        unsigned* s_old = (unsigned*)smem;                               // the x ring is dead: every wave is past its last LDS read (barrier above)
        if (tid == 0) *s_old = S > 1 ? atomicInc(fsp->ticket + blockIdx.z * gridDim.x + blockIdx.x, (unsigned)(S - 1)) : 0u;
        __syncthreads();
        if (*s_old != (unsigned)(S - 1)) return;
        __syncthreads();                                                  // s_old read by everyone before red[] (same LDS) is written
        sk4_finish_tile<MTW>(fsp, out, S, M, N, mb, ntile * 16, g, lr, w, tid, (float*)smem);
    }
    else if constexpr (EPI == 1) { static_assert(EPI != 1 || MS == 1, "transposed epilogues: MS = 1"); skinny_store_swiglu<MT, 4>(smem, acc, (bf16*)out, M, N / 2, mbase, blockIdx.x, w, g, lr, tid); }
    else { static_assert(EPI != 0 || MS == 1, "transposed epilogues: MS = 1"); skinny_store_tile<MT, 4>(smem, acc, out + (long)split * M * N, M, N, mbase, blockIdx.x * 64, w, g, lr, tid); }
    if constexpr (PROF) { if (pw) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); } }
}
// ------------------------------------------------------------------------------- skinny GEMM v5 (round 6): the wide-N GEMMs at 65..128 rows
// Why: at 128 rows qkv and gate|up ran on the v3 8-wave blocks (64 rows x 128 columns, x tile through registers).  Three things held them at
// 2.4-2.7 TB/s (profiles/r06_*): (1) every wave reads the block's whole x chunk from LDS for ONE 16-column n-tile -- 1 KiB of fragment reads per MFMA,
// twice what the LDS delivers while the MFMAs run (gate|up: 6.8 us of LDS reads for 3.9 us of MFMAs); (2) the x tile's staging registers are waited for
// with the W ring on the same in-order vmcnt counter: the wait for x(c+1) also retires every older W load, so the ring never holds more than ONE
// chunk in flight (32 KiB of W per CU = one memory round trip per chunk, whatever its depth); (3) x costs a VGPR round trip and an LDS write.
// v5: the same 64-row x 128-column block and grid (so the second row block's weights still come from L2), 8 waves = 4 n-tile PAIRS x 2 K halves:
//   * a wave owns 32 columns and HALF of the block's K range: one x fragment feeds two MFMAs (half the LDS reads per MFMA) and the two K halves
//     stream two independent parts of the weight matrix (twice the requests in flight per column);
//   * x by LDS-DMA (v4's ring, XOR-swizzled source, no VGPRs), W through an asm-ordered register ring of WD chunks (8 x 1 KiB per chunk per wave),
//     ONE hand-counted vmcnt per wait (sk5_wait_*: the issue order is simulated at compile time, tools/sk5_isa_check.py replays the compiled code):
//     the x wait no longer drains the ring -- WD x 8 KiB of W per wave stay in flight (192 KiB per CU at WD = 3);
//   * x chunk c+1 is retired at chunk c's barrier and read one barrier later (the staging rule of tools/dma_isa_check.py, strict form);
//   * the two K halves meet once, through LDS, after the loop: every wave hands the n-tile it does not finalise to its partner and finalises the
//     other one (sum = half 0 + half 1, fixed order), then the v4 direct epilogues (16-byte write-through slab stores / SwiGLU).
constexpr int sk5_wait_x(int c, int NH, int XD, int WD, int pre = 0) {     // ops that may stay outstanding so that X(min(c + 1, NH - 1)) has landed; pre: the prologue's wait for X(0)
    int ops = 0, lastX[64] = {};
    for (int p = 0; p < XD - 1 && p < NH; ++p) { ops += 4; lastX[p] = ops; }
    for (int p = 0; p < WD && p < NH; ++p) ops += 8;
    if (pre) return ops - lastX[0];
    for (int it = 0; it < NH; ++it) {
        if (it == c) return ops - lastX[(c + 1 < NH) ? c + 1 : NH - 1];
        if (it + XD - 1 < NH) { ops += 4; lastX[it + XD - 1] = ops; }
        if (it + WD < NH) ops += 8;
    }
    return 0;
}
constexpr int sk5_wait_w(int c, int NH, int XD, int WD) {                 // ops that may stay outstanding so that W(c) has landed, evaluated behind the issue of X(c + XD - 1)
    int ops = 0, lastW[64] = {};
    for (int p = 0; p < XD - 1 && p < NH; ++p) ops += 4;
    for (int p = 0; p < WD && p < NH; ++p) { ops += 8; lastW[p] = ops; }
    for (int it = 0; it < NH; ++it) {
        if (it + XD - 1 < NH) ops += 4;
        if (it == c) return ops - lastW[c];
        if (it + WD < NH) { ops += 8; lastW[it + WD] = ops; }
    }
    return 0;
}
template <int N_> __device__ __forceinline__ void wait_vmcnt_w2(bf16x8 (&wv)[2][4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(wv[0][0]), "+v"(wv[0][1]), "+v"(wv[0][2]), "+v"(wv[0][3]), "+v"(wv[1][0]), "+v"(wv[1][1]), "+v"(wv[1][2]), "+v"(wv[1][3])
                 : "n"(N_ > 63 ? 63 : N_) : "memory");
}
template <int C, int NH, class F> __device__ __forceinline__ void sk5_static_for(F&& f) {
    if constexpr (C < NH) { f(std::integral_constant<int, C>{}); sk5_static_for<C + 1, NH>(f); }
}
// EPI 4: fp32 split-K slab, write-through stores; EPI 3: SwiGLU -> bf16 h [M][N/2].  NCK = chunks of the block's K range (even).
template <int NCK, int XD, int WD, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_sk5_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W, float* __restrict__ out,
                                                       const float* __restrict__ ssq, int M, int N, int K, float eps) {
    static_assert(NCK % 2 == 0 && XD >= 3 && WD >= 1, "two K halves; the x ring holds the chunk being read, the retired next one and one in flight");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NH = NCK / 2;                                       // chunks per K half
    constexpr int XH = 64 * 256, XS = 2 * XH;                         // bytes per (half, chunk) x tile / per ring slot (both halves)
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 4, lr = l & 15;
    const int pr = w & 3, h = w >> 2;                                 // n-tile pair of the block, K half
    const int split = blockIdx.y, mbase = blockIdx.z * 64;
    const int nt0 = blockIdx.x * 8 + pr * 2, ntiles = N >> 4;
    const int kc0 = split * NCK + h * NH;                             // first 128-wide K chunk of this wave's half
    const bf16* wp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) wp[q] = W + ((long)(nt0 + q < ntiles ? nt0 + q : ntiles - 1) * (K / SK_BK) + kc0) * 2048 + l * 8;
    // x DMA: the 4 waves of half h stage that half's 16 pieces (4 rows each) of every chunk; wave (pr, h) owns pieces 4 pr .. 4 pr + 3
    const bf16* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (pr * 4 + j) * 4 + (l >> 4), m = mbase + r;
        xsrc[j] = x + (long)(m < M ? m : M - 1) * K + (long)kc0 * SK_BK + (((l & 15) ^ (r & 15)) << 3);
    }
    auto issueX = [&](int c) __attribute__((always_inline)) {
        char* slot = smem + (c % XD) * XS + h * XH + pr * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(xsrc[j] + c * SK_BK, slot + j * 1024);
    };
    // deferred-1/rms RMSNorm (kernels.h): lane (lr, g) fetches the 8 partial sums of squares of row mbase + 16 g + lr -- the OLDEST VMEM operations of
    // the wave, so every counted wait below (they all retire something younger) has them landed and the counts need not know them; the scale is
    // formed in the epilogue and handed to the lanes that finalise the row by shuffles.  ssq == nullptr: x is already normalised.
    f32x4 sqa = {0.f, 0.f, 0.f, 0.f}, sqb = {0.f, 0.f, 0.f, 0.f};
    {
        const int m = mbase + g * 16 + lr;
        const float* pp = (ssq ? ssq : (const float*)x) + (ssq ? (long)(m < M ? m : M - 1) * 8 : 0);      // branch-free: without ssq a harmless read of x[0..7]
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(sqa), "=&v"(sqb) : "v"(pp) : "memory");
    }
    bf16x8 wr[WD][2][4];
    f32x4 acc[2][4];                                                  // [n-tile of the pair][m-tile]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < XD - 1 && c < NH; ++c) issueX(c);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < WD && c < NH; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) sk4_wload(wr[c][q][i], wp[q] + c * 2048 + i * 512);
    // X(0) retired one barrier before its first read (strict staging rule)
    wait_vmcnt<sk5_wait_x(0, NH, XD, WD, 1)>();
    __builtin_amdgcn_s_barrier();
    sk5_static_for<0, NH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        wait_vmcnt<sk5_wait_x(c, NH, XD, WD)>();                     // this wave's pieces of X(c + 1) have landed
        __builtin_amdgcn_s_barrier();                                 // ... every wave's have; every wave is done reading X(c - 1)
        if constexpr (c + XD - 1 < NH) issueX(c + XD - 1);            // into the slot of X(c - 1)
        wait_vmcnt_w2<sk5_wait_w(c, NH, XD, WD)>(wr[c % WD]);         // W(c) landed (younger loads stay in flight)
        {
            const char* rp = smem + (c % XD) * XS + h * XH + lr * 256;
            int so[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) so[i] = ((i * 4 + g) ^ lr) << 4;
            bf16x8 af[2][2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) af[0][hh][i] = *(const bf16x8*)(rp + hh * 4096 + so[i]);
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                if (p2 == 0) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[1][hh][i] = *(const bf16x8*)(rp + (2 + hh) * 4096 + so[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[q][2 * p2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[c % WD][q][i], af[p2][0][i], acc[q][2 * p2], 0, 0, 0);
                        acc[q][2 * p2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[c % WD][q][i], af[p2][1][i], acc[q][2 * p2 + 1], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (c + WD < NH) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) sk4_wload(wr[c % WD][q][i], wp[q] + (c + WD) * 2048 + i * 512);
        }
    });
    // the two K halves meet: wave (pr, h) hands the partial sums of n-tile 1 - h to its partner (pr, 1 - h) and finalises n-tile h
    __syncthreads();                                                  // every wave is past its last x fragment read: the ring becomes the exchange buffer
    f32x4* xch = (f32x4*)smem;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xch[((w * 4 + mt) << 6) + l] = acc[1 - h][mt];
    __syncthreads();
    // the loads of sqa / sqb are older than everything the loop waited for: landed (asm outputs: the compiler must not use them before this point's
    // explicit wait -- nothing is outstanding any more except the last W chunks' loads, all consumed)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sqa), "+v"(sqb)::"memory");
    const float myrs = ssq ? rsqrtf((((sqa.x + sqa.y) + (sqa.z + sqa.w)) + ((sqb.x + sqb.y) + (sqb.z + sqb.w))) / (float)K + eps) : 1.f;
    float rsc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rsc[mt] = __shfl(myrs, mt * 16 + lr, 64);
    f32x4 fin[4];
    const int pw = (w ^ 4);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 o = xch[((pw * 4 + mt) << 6) + l];
        const f32x4 a0 = h == 0 ? acc[0][mt] : o, a1 = h == 0 ? o : acc[1][mt];     // half 0 + half 1, in that order for both n-tiles
        fin[mt] = a0 + a1;
        fin[mt] *= rsc[mt];
    }
    const int ntile = nt0 + h;
    if (ntile < ntiles) {
        if constexpr (EPI == 3) sk4_store_swiglu_direct<4>(fin, (bf16*)out, M, N / 2, mbase, ntile, g, lr);
        else sk4_store_direct<4, true>(fin, out + (long)split * M * N, M, N, mbase, ntile * 16, g, lr);
    }
}
template <int NCK, int EPI>
static void launch_sk5(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S, SkRowScale rsc) {
    constexpr int XD = 4, WD = 3;
    constexpr int LDS = XD * 2 * 64 * 256;                            // 128 KiB: the x ring (the 32 KiB exchange buffer aliases it)
    auto kfn = gemm_sk5_kernel<NCK, XD, WD, EPI>;
    (void)PG_DYN_LDS(kfn, LDS);
    dim3 grid(N / 128, S, (M + 63) / 64), block(512);
    hipLaunchKernelGGL(kfn, grid, block, LDS, s, x, Wt, out, rsc.ssq, M, N, K, rsc.eps);
}
// true when the shape has a v5 instantiation: N a multiple of 128, K range of the block 8 or 16 chunks
template <int EPI>
static bool sk5_try(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S, SkRowScale rsc = {}) {
    if (!Wt || (N % 128) || M > 128 || S < 1 || K % (SK_BK * S)) return false;
    const int nck = K / SK_BK / S;
    if (nck == 16) { launch_sk5<16, EPI>(s, x, Wt, out, M, N, K, S, rsc); return true; }
    if (nck == 8) { launch_sk5<8, EPI>(s, x, Wt, out, M, N, K, S, rsc); return true; }
    return false;
}

extern unsigned long long* g_sk4_prof;          // stamp buffer of the PROF instantiations (libplangen_diag.so: tools/sk4_profile.py); null in production
template <int MT, int NCK, int XD, int WD, int EPI, int OCC, int ABL = 0, int MS = 1>
static void launch_sk4(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S, const void* tail = nullptr) {      // tail: EPI 5's SkFuse site
    constexpr int XL = XD * MT * 16 * 256 + ((ABL & 512) ? 4096 * MS : 0), TL = MT * 16 * (4 * 16 + 4) * 4;
    constexpr int LDS = XL > TL ? XL : TL;
    auto kfn = gemm_sk4_kernel<MT, NCK, XD, WD, EPI, OCC, ABL, MS>;
    (void)PG_DYN_LDS(kfn, LDS);
    dim3 grid((N + 63) / 64, S, (M + MT * 16 - 1) / (MT * 16)), block(256 * MS);
    hipLaunchKernelGGL(kfn, grid, block, LDS, s, x, Wt, out, M, N, K, tail ? (unsigned long long*)tail : g_sk4_prof);
}
template <int MT, int XD, int WD, int EPI, int OCC, int ABL = 0, int MS = 1>
static bool sk4_nck(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S, const void* tail = nullptr) {
    const int nck = K / SK_BK / S;
    if (nck * S * SK_BK != K || (N & 15)) return false;
    switch (nck) {
        case 2: launch_sk4<MT, 2, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S, tail); return true;
        case 4: launch_sk4<MT, 4, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S, tail); return true;
        case 8: launch_sk4<MT, 8, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S, tail); return true;
        case 11: launch_sk4<MT, 11, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S, tail); return true;
        case 16: launch_sk4<MT, 16, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S, tail); return true;
        case 22: launch_sk4<MT, 22, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S, tail); return true;
        default: return false;
    }
}

