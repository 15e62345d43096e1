// GEMM kernels for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  (both operands K-contiguous).
//
//  gemm_big   : bf16 MFMA (v_mfma_f32_16x16x32_bf16), 128x128x64 block tile, 4 waves (2x2,
//               64x64 per wave = 4x4 MFMA tiles), operands staged HBM->LDS with
//               global_load_lds_dwordx4 (no VGPR round trip), XOR-swizzled through the
//               SOURCE address so ds_read_b128 fragment reads are bank-conflict free,
//               double-buffered with one barrier per K tile.  A-operand loaders: plain
//               row-major, or implicit im2col of a 3x3 conv over NHWC (optionally with the
//               nearest-2x upsample folded into the address).  Used for prefill (MFMA-bound,
//               M = packed prompt tokens) and the VQ-16 decoder convolutions.
//  gemm_skinny: decode-step GEMM, M <= 128 rows: HBM-bound weight streaming.  Each wave streams
//               its own 16 weight rows straight into VGPRs (64 contiguous bytes per lane per
//               128-wide K chunk), the small x tile is shared through LDS, split-K over
//               blockIdx.y fills the 256 CUs; fp32 partial slabs are reduced by the consumer
//               kernel (deterministic, no atomics).
//  gemm_f32   : plain fp32 tiled GEMM (PG_F32 parity mode); same loaders/epilogue.
#include "kernels.h"

#include "gemm_common.h"
#include <type_traits>

// ------------------------------------------------------------------------------- big MFMA GEMM
#define BIG_BM 128
#define BIG_BN 128
#define BIG_BK 64
#define BIG_TILE_BYTES (128 * 64 * 2)   // 16 KiB per operand tile
#define BIG_LDS (4 * BIG_TILE_BYTES)    // A[2] + B[2]

template <class AL, class EP>
__global__ __launch_bounds__(256, 2) void gemm_big_kernel(AL al, const bf16* __restrict__ W, long ldb,
                                                         long strideA, long strideB, long strideA2, long strideB2, EP ep,
                                                         int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int batch = blockIdx.y, batch2 = blockIdx.z;
    const long coff = (long)batch * ep.e.strideC + (long)batch2 * ep.e.strideC2;
    const long roff = (long)batch * ep.e.strideR;
    // XCD-aware tile order: each XCD (blockIdx % 8) owns a contiguous run of tiles; inside
    // the run tiles are grouped 8(M) x 8(N) so concurrently resident blocks share A and W panels in L2.
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m, tile_n;
    if (((ntm | ntn) & 7) == 0) {
        const int sup = t >> 6, in = t & 63, gm = ntm >> 3;
        tile_m = (sup % gm) * 8 + (in & 7);
        tile_n = (sup / gm) * 8 + (in >> 3);
    } else { tile_m = t % ntm; tile_n = t / ntm; }
    const int m0 = tile_m * BIG_BM, n0 = tile_n * BIG_BN;

    al.A_offset(strideA * batch + strideA2 * batch2);
    const bf16* Wb = W + strideB * batch + strideB2 * batch2;
    // staging assignment: wave w, instruction i covers tile rows (w*4+i)*8 .. +7, lane -> (row l>>3, chunk l&7)
    const bf16* wrow[4]; int sc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (w * 4 + i) * 8 + (l >> 3);
        sc[i] = ((l & 7) ^ ((r >> 1) & 7)) * 8;          // swizzled SOURCE chunk (elements)
        al.init(i, m0 + r);
        const int n = n0 + r;
        wrow[i] = Wb + (long)(n < N ? n : N - 1) * ldb;
    }
    char* sA = smem; char* sB = smem + 2 * BIG_TILE_BYTES;
    const int wbase = __builtin_amdgcn_readfirstlane(w) * 4 * 1024;

    auto stage = [&](int buf, int kt) {
        const int k0 = kt * BIG_BK;
        al.set_ktile(k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(al.ptr(i, k0 + sc[i]), sA + buf * BIG_TILE_BYTES + wbase + i * 1024);
            glds16(wrow[i] + k0 + sc[i], sB + buf * BIG_TILE_BYTES + wbase + i * 1024);
        }
    };

    const int wm = w >> 1, wn = w & 1, g = l >> 4, lr = l & 15;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BIG_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* a_t = sA + cur * BIG_TILE_BYTES;
        const char* b_t = sB + cur * BIG_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wm * 64 + mt * 16 + lr;
                af[mt] = *(const bf16x8*)(a_t + row * 128 + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int row = wn * 64 + nt * 16 + lr;
                bfr[nt] = *(const bf16x8*)(b_t + row * 128 + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // Round 3: tile kt+1 was retired by the wait + barrier above; a second barrier separates that retirement from the phase that reads it
        // (the staging rule of tools/dma_isa_check.py, strict form -- a third LDS slot would halve this kernel's residency instead).
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    // C/D layout of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {                                          // 16 elements per batch: their bias / residual loads go out together
        int rows[16], cols[16]; float av[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { rows[i] = m0 + wm * 64 + mt * 16 + g * 4 + (i & 3); cols[i] = n0 + wn * 64 + (i >> 2) * 16 + lr; av[i] = acc[mt][i >> 2][i & 3]; }
        ep.template store1_batch<16>(coff, roff, rows, cols, av);
    }
}

// ------------------------------------------------------------------------------- fp32 GEMM
template <class AL, class EP>
__global__ __launch_bounds__(256) void gemm_f32_kernel(AL al, const float* __restrict__ W, long ldb,
                                                      long strideA, long strideB, long strideA2, long strideB2, int nbatch, EP ep,
                                                      int M, int N, int K) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int batch = blockIdx.z % nbatch, batch2 = blockIdx.z / nbatch;
    const long coff = (long)batch * ep.e.strideC + (long)batch2 * ep.e.strideC2;
    const long roff = (long)batch * ep.e.strideR;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    al.A_offset(strideA * batch + strideA2 * batch2);
    const float* Wb = W + strideB * batch + strideB2 * batch2;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * 256, r = e >> 4, kk = e & 15;   // r in [0,64), kk in [0,16)
            As[kk][r] = (k0 + kk < K) ? al.elem(m0 + r, k0 + kk) : 0.f;
            const int n = n0 + r;
            Bs[kk][r] = (n < N && k0 + kk < K) ? Wb[(long)n * ldb + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ep(coff, roff, m0 + ty * 4 + i, n0 + tx * 4 + j, acc[i][j]);
}

// ------------------------------------------------------------------------------- launchers
template <typename T> struct BigDispatch;

template <> struct BigDispatch<bf16> {
    static void run(hipStream_t s, const GemmA& a, const bf16* W, long ldb, long strideB, const GemmEpi& e,
                    int M, int N, int K, int batch, int batch2, long strideB2) {
        if (batch == 1 && batch2 == 1 && pg_tune->gemm256 && conv_halo_try(s, a, W, e, M, N, K, (float*)a.gn_part, a.gn_nsplit)) return;
        // libplangen_diag.so only (round 5 experiment, profiles/r05_c): one wave per SIMD, 128 x 128 of C per wave, 32x32x16 MFMAs
        if (pg_tune->diag && pg_tune->diag->gemm_big_wave && pg_tune->diag->gemm_big_wave(s, a, W, ldb, e, M, N, K, batch, batch2)) return;
        if (gemm256_try(s, a, W, ldb, strideB, e, M, N, K, batch, batch2, strideB2)) return;
        Epi<bf16> ep{e, M, N};
        const int ntm = (M + BIG_BM - 1) / BIG_BM, ntn = (N + BIG_BN - 1) / BIG_BN;
        dim3 grid(ntm * ntn, batch, batch2), block(256);
        if (a.kind == 0) {
            PlainLoaderB<bf16> al; al.A = (const bf16*)a.ptr; al.lda = a.lda; al.M = M;
            auto kfn = gemm_big_kernel<PlainLoaderB<bf16>, Epi<bf16>>;
            (void)PG_DYN_LDS(kfn, BIG_LDS);
            hipLaunchKernelGGL(kfn, grid, block, BIG_LDS, s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, ntm, ntn);
        } else {
            ConvLoaderB<bf16> al; al.X = (const bf16*)a.ptr; al.zeros = (const bf16*)a.zeros;
            al.Hi = a.Hi; al.Wi = a.Wi; al.Cin = a.Cin; al.up = a.up; al.stride2 = (a.kind == 2);
            al.Ho = al.stride2 ? a.Hi / 2 : (a.Hi << a.up); al.Wo = al.stride2 ? a.Wi / 2 : (a.Wi << a.up); al.M = M;
            auto kfn = gemm_big_kernel<ConvLoaderB<bf16>, Epi<bf16>>;
            (void)PG_DYN_LDS(kfn, BIG_LDS);
            hipLaunchKernelGGL(kfn, grid, block, BIG_LDS, s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, ntm, ntn);
        }
    }
};

template <> struct BigDispatch<float> {
    static void run(hipStream_t s, const GemmA& a, const float* W, long ldb, long strideB, const GemmEpi& e,
                    int M, int N, int K, int batch, int batch2, long strideB2) {
        Epi<float> ep{e, M, N};
        dim3 grid((N + 63) / 64, (M + 63) / 64, batch * batch2), block(256);
        if (a.kind == 0) {
            PlainLoaderB<float> al; al.A = (const float*)a.ptr; al.lda = a.lda; al.M = M;
            hipLaunchKernelGGL((gemm_f32_kernel<PlainLoaderB<float>, Epi<float>>), grid, block, 0, s, al, W, ldb,
                               a.strideA, strideB, a.strideA2, strideB2, batch, ep, M, N, K);
        } else {
            ConvLoaderB<float> al; al.X = (const float*)a.ptr; al.zeros = (const float*)a.zeros;
            al.Hi = a.Hi; al.Wi = a.Wi; al.Cin = a.Cin; al.up = a.up; al.stride2 = (a.kind == 2);
            al.Ho = al.stride2 ? a.Hi / 2 : (a.Hi << a.up); al.Wo = al.stride2 ? a.Wi / 2 : (a.Wi << a.up); al.M = M;
            hipLaunchKernelGGL((gemm_f32_kernel<ConvLoaderB<float>, Epi<float>>), grid, block, 0, s, al, W, ldb,
                               a.strideA, strideB, a.strideA2, strideB2, batch, ep, M, N, K);
        }
    }
};

template <typename T>
void launch_gemm(hipStream_t s, const GemmA& a, const T* W, long ldb, long strideB, const GemmEpi& e,
                 int M, int N, int K, int batch, int batch2, long strideB2) {
    if (M <= 0 || N <= 0) return;
    BigDispatch<T>::run(s, a, W, ldb, strideB, e, M, N, K, batch, batch2, strideB2);
}
template void launch_gemm<float>(hipStream_t, const GemmA&, const float*, long, long, const GemmEpi&, int, int, int, int, int, long);
template void launch_gemm<bf16>(hipStream_t, const GemmA&, const bf16*, long, long, const GemmEpi&, int, int, int, int, int, long);

#include "gemm_skinny.h"

// Split-K count: smallest divisor S of the chunk count that gives enough blocks to keep the
// 256 CUs streaming.  Measured on MI355X (tools/skinny_sweep.py): at M = 128 fewer, fatter
// blocks win (slab traffic grows with S), at M <= 32 more, thinner ones do; S is restricted to
// chunk counts the unrolled kernel is instantiated for.
unsigned long long* g_sk4_prof = nullptr;
static const PgTune g_default_tune{};
thread_local const PgTune* pg_tune = &g_default_tune;
int skinny_pick_splits(int N, int K, int M) {
    const int nblk = (N + 63) / 64, nchunks = K / SK_BK;
    const int target = M >= 96 ? pg_tune->split_big : (M >= 48 ? pg_tune->split_mid : pg_tune->split_small);   // GEMM+consumer optimum (sweep "+n" columns)
    int best = 1;
    for (int S = 1; S <= nchunks; ++S) {
        if (nchunks % S) continue;
        const int nck = nchunks / S;
        if (!(nck == 1 || nck == 2 || nck == 4 || nck == 8 || nck == 11 || nck == 16 || nck == 22) && S != nchunks) continue;
        best = S;
        if (nblk * S >= target) break;
    }
    return best;
}
int skinny_pick_splits(int N, int K) { return skinny_pick_splits(N, K, 128); }

void launch_gemm_skinny_v1(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S) {
    if (M <= 0) return;
    const int nck = K / SK_BK / S;
    const int mblocks = (M + 127) / 128;
    const int mrows = M < 128 ? M : 128;
    dim3 grid((N + 63) / 64, S, mblocks), block(256);
    if (mrows <= 16) {
        hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, block, 2 * 1 * 16 * SK_ROWB, s, x, W, out, M, N, K, nck);
    } else if (mrows <= 32) {
        hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, block, 2 * 2 * 16 * SK_ROWB, s, x, W, out, M, N, K, nck);
    } else if (mrows <= 64) {
        hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, block, 2 * 4 * 16 * SK_ROWB, s, x, W, out, M, N, K, nck);
    } else {
        auto kfn = gemm_skinny_kernel<8>;
        (void)PG_DYN_LDS(kfn, 2 * 8 * 16 * SK_ROWB);
        hipLaunchKernelGGL(kfn, grid, block, 2 * 8 * 16 * SK_ROWB, s, x, W, out, M, N, K, nck);
    }
}

// Build the decode ("tiled") copy of a row-major bf16 weight [N][K]: [n-tile 16][k-chunk 128][k-step i][lane][8].
__global__ void tile_weights_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int N, int K) {
    const long nvec = (long)N * K / 8;
    const int nchunks = K / SK_BK;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(v & 63), i = (int)((v >> 6) & 3);
        const long t = v >> 8;                                  // tile index = ntile * nchunks + chunk
        const int chunk = (int)(t % nchunks); const long ntile = t / nchunks;
        const long n = ntile * 16 + (lane & 15);
        const int k = chunk * SK_BK + i * 32 + (lane >> 4) * 8;
        *(u32x4*)(dst + v * 8) = *(const u32x4*)(src + n * K + k);
    }
}
void launch_tile_weights(hipStream_t s, const bf16* src, bf16* dst, int N, int K) {
    const long nvec = (long)N * K / 8;
    const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    hipLaunchKernelGGL(tile_weights_kernel, dim3(blocks), dim3(256), 0, s, src, dst, N, K);
}

// Row re-ordering of Wqkv for the prefill RoPE epilogue (RopeEpi): new row 16t + p of a q / k head = old row 8t + p (p < 8) or 64 + 8t + p - 8.
__global__ void interleave_qk_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int nh, int K) {
    const int n = blockIdx.x, HD = nh * 128;
    const int sec = n / HD, hc = n - sec * HD, head = hc >> 7, c = hc & 127;
    int so = c;
    if (sec < 2) { const int t = c >> 4, p = c & 15; so = p < 8 ? 8 * t + p : 64 + 8 * t + p - 8; }
    const u32x4* sp = (const u32x4*)(src + ((long)sec * HD + head * 128 + so) * K);
    u32x4* dp = (u32x4*)(dst + (long)n * K);
    for (int i = threadIdx.x; i < K / 8; i += blockDim.x) dp[i] = sp[i];
}
void launch_interleave_qk(hipStream_t s, const bf16* src, bf16* dst, int nh, int K) {
    hipLaunchKernelGGL(interleave_qk_kernel, dim3(3 * nh * 128), dim3(256), 0, s, src, dst, nh, K);
}

template <int EPI, bool TILED>
static bool sk3_prod_tiled(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S, SkRowScale rsc = {}) {
    const int nck = K / SK_BK / S;
    if (nck * S * SK_BK != K || (N & 15)) return false;
    const int mrows = M < 128 ? M : 128;
    if (mrows <= 16) return sk3_prod_nck<1, EPI, 4, TILED>(s, x, Wt, out, M, N, K, S, nck, rsc);
    if (mrows <= 32) return sk3_prod_nck<2, EPI, 4, TILED>(s, x, Wt, out, M, N, K, S, nck, rsc);
    // tiled weights: 64-row M blocks for every shape (measured best at M = 128: the second reader of a
    // W tile hits L2, LDS per block halves -> more blocks per CU)
    // wide N (qkv, gate|up, gen_head at 65..128 rows): 128-column blocks of 8 waves -- the x tile is fetched and staged once per 128
    // columns instead of once per 64 (loop -11 ms at bs=64); stream_gemm bit 128 keeps the 4-wave block for A/B
    if (N % 128 == 0 && N >= 4096 && !(pg_tune->stream_gemm >= 0 && (pg_tune->stream_gemm & 128))) {
        if constexpr (TILED) {
            if (pg_tune->sk3_xa == 2 && sk3_xa2_nck<EPI>(s, x, Wt, out, M, N, K, S, nck, rsc)) return true;
            if (pg_tune->sk3_xa == 3 && sk3_nt_nck<EPI>(s, x, Wt, out, M, N, K, S, nck, rsc)) return true;       // round 6 experiment: nt hint on the weight fragments
        }
        return sk3_prod_nck<4, EPI, 8, TILED>(s, x, Wt, out, M, N, K, S, nck, rsc);
    }
    return sk3_prod_nck<4, EPI, 4, TILED>(s, x, Wt, out, M, N, K, S, nck, rsc);
}

template <int EPI>
static bool sk3_prod(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S) {
    const int nck = K / SK_BK / S;
    if (nck * S * SK_BK != K) return false;
    const int mrows = M < 128 ? M : 128;
    if (mrows <= 16) return sk3_prod_nck<1, EPI>(s, x, W, out, M, N, K, S, nck);
    if (mrows <= 32) return sk3_prod_nck<2, EPI>(s, x, W, out, M, N, K, S, nck);
    if (mrows <= 64) return sk3_prod_nck<4, EPI>(s, x, W, out, M, N, K, S, nck);
    // narrow outputs (N = 2048: o_proj, down_proj): too few column blocks to fill 256 CUs, so the
    // rows are split into 64-row blocks as well (the second reader of a W slab hits L2);
    // measured 8.0 -> 6.7 us (o) and 15.3 -> 12.4 us (down) at M = 128, slower for wide N.
    if ((N + 63) / 64 < 64 && EPI == 0) return sk3_prod_nck<4, EPI>(s, x, W, out, M, N, K, S, nck);
    return sk3_prod_nck<8, EPI>(s, x, W, out, M, N, K, S, nck);
}



// v4 ("stream", x by LDS-DMA, swapped-operand 16-byte write-through stores) where it measured faster than v3 on MI355X
// (tools/sk4_sweep.py, profiles/r02_*): 64 < M <= 128 -> 128-row blocks for wide N, 64-row blocks for N < 4096;
// 16 < M <= 64 -> one 64-row block; M <= 16 -> one 16-row block.  EPI 4: slab, 3: SwiGLU.
// pg_tune->stream_gemm bits: 1 wide-N slabs (qkv, gen_head, lm_head), 2 narrow-N slabs (o, down), 4 SwiGLU gate|up,
// 8 the M <= 16 kernels, 16 SwiGLU through the LDS-transposed epilogue instead of the direct one.
template <int EPI>
// Every production instantiation retires an x chunk ONE BARRIER BEFORE its first read (ABL bit 64, ring one slot deeper to keep the
// prefetch distance): the CDNA guide's staging rule, kept because it is sound and free.  It was introduced in round 2 against wrong rows
// 12-15 / 28-31 / 44-47 in ~0.5 % of COLD launches; round 4 showed those came from the epilogue's asm store (sk4_store_direct), not from
// the staging -- the rule never was the fix.
static bool sk4_prod(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S) {
    // -1 (default): what measured faster IN the decode loop on MI355X (tools/ab_loop.sh, profiles/r02_b_decode_gemm_investigation.md): every class at
    // M <= 64 (48 KiB blocks, 2-3 per CU: loop -3.3 % at bs=32, -6.2 % at bs=8), only the narrow-N slabs at M = 128 (-1.5 %;
    // the 128-row blocks own 96 KiB of LDS = one block per CU and lose 0.7-2.6 % in the loop although they win the microbenchmark)
    const int sg = pg_tune->stream_gemm >= 0 ? pg_tune->stream_gemm : (M > 64 ? 2 : 15);
    if (!sg || !Wt || M > 128) return false;
    constexpr bool SW = EPI == 3;
    const bool wide = N >= 4096;
    if (SW ? !(sg & 4) : !(sg & (wide ? 1 : 2))) return false;
    if (M > 64) {
        if constexpr (SW) {
            if (sg & 16) return sk4_nck<8, 4, 3, 1, 1, 64>(s, x, Wt, out, M, N, K, S);
            return sk4_nck<8, 4, 3, 3, 1, 64>(s, x, Wt, out, M, N, K, S);
        } else {
            if (wide) return sk4_nck<8, 4, 3, EPI, 1, 64>(s, x, Wt, out, M, N, K, S);
            return sk4_nck<4, 4, 3, EPI, 2, 64>(s, x, Wt, out, M, N, K, S);
        }
    }
    if (M > 16) {                               // 17..64 rows: one 64-row block (rows beyond M clamped; bs=16 loop -1.6 %, bs=32 -3.3 %)
        if constexpr (SW) { if (sg & 16) return sk4_nck<4, 4, 3, 1, 2, 64>(s, x, Wt, out, M, N, K, S); }
        return sk4_nck<4, 4, 3, EPI, 2, 64>(s, x, Wt, out, M, N, K, S);
    }
    if (sg & 8) {
        if constexpr (SW) { if (sg & 16) return sk4_nck<1, 5, 3, 1, 4, 64>(s, x, Wt, out, M, N, K, S); }
        return sk4_nck<1, 5, 3, EPI, 4, 64>(s, x, Wt, out, M, N, K, S);
    }
    return false;
}
// Deferred-1/rms forms (round 6; decode, > 64 rows, wide N on the tiled copy): x = bf16(residual . w_norm), ssq = rmsnorm_defer_kernel's partial
// sums of squares; the v3 kernel scales its fp32 result by the row's 1/rms.  False when the shape has no such instantiation (the caller then
// runs the ordinary norm).  deferred_norm_ok() is the ONE predicate both the norm launch and the GEMM launch consult.
// v5 takes the wide-N GEMMs at 65..128 rows whose block K range is 8 or 16 chunks (qkv at S = 2, gate|up / gen_head / lm_head at S = 1)
static bool sk5_shape(int M, int N, int K, int S) {
    const int nck = S > 0 ? K / SK_BK / S : 0;
    return pg_tune->sk5 && M > 64 && M <= 128 && N >= 4096 && (N % 128) == 0 && S > 0 && nck * S * SK_BK == K && (nck == 8 || nck == 16);
}
bool deferred_norm_ok(int M, int N, int K, int S) {
    if (K != 2048) return false;
    if (sk5_shape(M, N, K, S)) return true;
    const int sg = pg_tune->stream_gemm >= 0 ? pg_tune->stream_gemm : (M > 64 ? 2 : 15);
    const int nck = S > 0 ? K / SK_BK / S : 0;
    return M > 64 && M <= 128 && N >= 4096 && (N % 128) == 0 && !(sg & 1) && !(sg & 4) && S > 0 && nck * S * SK_BK == K
           && (nck == 1 || nck == 2 || nck == 4 || nck == 8 || nck == 16);
}
bool launch_gemm_skinny_deferred(hipStream_t s, const bf16* xw, const bf16* Wt, float* out, int M, int N, int K, int S, const float* ssq, float eps) {
    if (!Wt || !ssq || !deferred_norm_ok(M, N, K, S)) return false;
    if (sk5_shape(M, N, K, S)) return sk5_try<4>(s, xw, Wt, out, M, N, K, S, SkRowScale{ssq, eps});
    return sk3_prod_tiled<0, true>(s, xw, Wt, out, M, N, K, S, SkRowScale{ssq, eps});
}
bool launch_gemm_skinny_swiglu_deferred(hipStream_t s, const bf16* xw, const bf16* Wt, bf16* h, int M, int N, int K, const float* ssq, float eps) {
    if (!Wt || !ssq || !deferred_norm_ok(M, N, K, 1)) return false;
    if (sk5_shape(M, N, K, 1)) return sk5_try<3>(s, xw, Wt, (float*)h, M, N, K, 1, SkRowScale{ssq, eps});
    return sk3_prod_tiled<1, true>(s, xw, Wt, (float*)h, M, N, K, 1, SkRowScale{ssq, eps});
}
void launch_gemm_skinny(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, const bf16* Wt) {
    if (M <= 0) return;
    if (Wt && sk5_shape(M, N, K, S) && sk5_try<4>(s, x, Wt, out, M, N, K, S)) return;
    if (sk4_prod<4>(s, x, Wt, out, M, N, K, S)) return;
    if (Wt && sk3_prod_tiled<0, true>(s, x, Wt, out, M, N, K, S)) return;
    if (!sk3_prod<0>(s, x, W, out, M, N, K, S)) launch_gemm_skinny_v1(s, x, W, out, M, N, K, S);
}
bool launch_gemm_skinny_tiled_only(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S) {      // the production dispatch on the tiled copy alone (diag: verification against v1)
    if (sk5_shape(M, N, K, S) && sk5_try<4>(s, x, Wt, out, M, N, K, S)) return true;
    return sk4_prod<4>(s, x, Wt, out, M, N, K, S) || sk3_prod_tiled<0, true>(s, x, Wt, out, M, N, K, S);
}
// gate|up GEMM with the SwiGLU gate fused (S = 1): h bf16 [M, N/2].  Returns false when the
// shape has no fused instantiation (caller falls back to slabs + silu_mul kernel).
bool launch_gemm_skinny_swiglu(hipStream_t s, const bf16* x, const bf16* W, bf16* h, int M, int N, int K, const bf16* Wt) {
    if (M <= 0) return true;
    if (Wt && sk5_shape(M, N, K, 1) && sk5_try<3>(s, x, Wt, (float*)h, M, N, K, 1)) return true;
    if (sk4_prod<3>(s, x, Wt, (float*)h, M, N, K, 1)) return true;
    if (Wt && sk3_prod_tiled<1, true>(s, x, Wt, (float*)h, M, N, K, 1)) return true;
    return sk3_prod<1>(s, x, W, (float*)h, M, N, K, 1);
}
