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
        if (gemm256_try(s, a, W, ldb, strideB, e, M, N, K, batch, batch2, strideB2)) return;
        Epi<bf16> ep{e, M, N};
        const int ntm = (M + BIG_BM - 1) / BIG_BM, ntn = (N + BIG_BN - 1) / BIG_BN;
        dim3 grid(ntm * ntn, batch, batch2), block(256);
        if (a.kind == 0) {
            PlainLoaderB<bf16> al; al.A = (const bf16*)a.ptr; al.lda = a.lda; al.M = M;
            auto kfn = gemm_big_kernel<PlainLoaderB<bf16>, Epi<bf16>>;
            static bool attr = false;
            if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS); attr = true; }
            hipLaunchKernelGGL(kfn, grid, block, BIG_LDS, s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, ntm, ntn);
        } else {
            ConvLoaderB<bf16> al; al.X = (const bf16*)a.ptr; al.zeros = (const bf16*)a.zeros;
            al.Hi = a.Hi; al.Wi = a.Wi; al.Cin = a.Cin; al.up = a.up; al.stride2 = (a.kind == 2);
            al.Ho = al.stride2 ? a.Hi / 2 : (a.Hi << a.up); al.Wo = al.stride2 ? a.Wi / 2 : (a.Wi << a.up); al.M = M;
            auto kfn = gemm_big_kernel<ConvLoaderB<bf16>, Epi<bf16>>;
            static bool attr = false;
            if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS); attr = true; }
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

// ------------------------------------------------------------------------------- skinny GEMM
#define SK_BK 128
#define SK_ROWB 288           // LDS row stride (256 B of k + 32 B pad: conflict-free ds_read_b128 for the (lr, g) fragment order)


// Block epilogue of the skinny kernels: the NW waves' accumulators (MFMA C layout: lane holds
// 4 rows x 1 column) are transposed through LDS so every lane stores 16 contiguous bytes
// instead of 32 scattered dword stores.  smem must hold MT*16 rows x (NW*16+4) floats.
template <int MT, int NW>
__device__ __forceinline__ void skinny_store_tile(char* smem, const f32x4 (&acc)[MT], float* __restrict__ o, int M, int N,
                                                  int mbase, int nbase, int w, int g, int lr, int tid, int wt = 0) {
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
            const f32x4 v4 = *(const f32x4*)(t + row * LD + c4);
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
                                                    int mbase, int nblk, int w, int g, int lr, int tid) {
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
            const float g0 = t[row * LD + tc], g1 = t[row * LD + tc + 1];
            const float u0 = t[row * LD + tc + 8], u1 = t[row * LD + tc + 9];
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

// Split-K count: smallest divisor S of the chunk count that gives enough blocks to keep the
// 256 CUs streaming.  Measured on MI355X (tools/skinny_sweep.py): at M = 128 fewer, fatter
// blocks win (slab traffic grows with S), at M <= 32 more, thinner ones do; S is restricted to
// chunk counts the unrolled kernel is instantiated for.
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
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 16 * SK_ROWB); attr = true; }
        hipLaunchKernelGGL(kfn, grid, block, 2 * 8 * 16 * SK_ROWB, s, x, W, out, M, N, K, nck);
    }
}

// ------------------------------------------------------------------------------- skinny GEMM v3
// Same geometry as v1 (BN = 64: 4 waves x 16 columns, BK = 128) but the W stream is decoupled
// from the per-chunk barrier: a register ring of depth D keeps D chunks of every wave's W
// rows in flight (the HBM requests of chunk c+D are issued while chunk c computes), the chunk
// loop is fully unrolled (NCK = chunks per block, compile time) so the ring is statically
// indexed.  XDB: x tile double-buffered (2 blocks/CU at MT=8) or single-buffered (4 blocks/CU).
template <int MT, int NCK, int D, bool XDB, int EPI, int NW, bool TILED = false>
__global__ __launch_bounds__(64 * NW, 2) void gemm_skinny3_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W,
                                                               float* __restrict__ out, int M, int N, int K, int wt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XB = MT * 16 * SK_ROWB, NTH = 64 * NW, BN = 16 * NW;
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

    u32x4 xs[XV];
    auto xload = [&](int c) {
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
                if (row < MT * 16) xs[j] = *(const u32x4*)(xp + (long)(m < M ? m : M - 1) * K + c * SK_BK + cv * 8);
                else xs[j] = (u32x4){0u, 0u, 0u, 0u};
            }
        }
    };
    auto xstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int v = tid + j * NTH, row = v >> 4, cv = v & 15;
            // unconditional when the tile divides over the block (hipcc cannot prove tid < NTH, keeps the guard, and SINKS the matching
            // x load into it: a conditional load with a `vmcnt(0)` at the join)
            if ((MT * 256) % NTH == 0 || row < MT * 16) *(u32x4*)(smem + buf * XB + row * SK_ROWB + cv * 16) = xs[j];
        }
    };
    bf16x8 wr[D][4];
#pragma unroll
    for (int c = 0; c < D && c < NCK; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[c][i] = *(const bf16x8*)(wp + c * WCH + i * WI);
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xload(0);
    xstore(0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
        if (c + 1 < NCK) xload(c + 1);
        const char* xt = smem + (XDB ? (c & 1) * XB : 0);
        skinny_mfma_chunk<MT>(xt, lr, g, wr[c % D], acc);
        if (c + D < NCK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[c % D][i] = *(const bf16x8*)(wp + (c + D) * WCH + i * WI);
        }
        if (c + 1 < NCK) {
            if (!XDB) __syncthreads();
            xstore(XDB ? ((c + 1) & 1) : 0);
            __syncthreads();
        }
    }
    if constexpr (EPI == 1) skinny_store_swiglu<MT, NW>(smem, acc, (bf16*)out, M, N / 2, mbase, bx, w, g, lr, tid);
    else skinny_store_tile<MT, NW>(smem, acc, out + (long)split * M * N, M, N, mbase, bx * BN, w, g, lr, tid, wt);
}
template <int MT, int NCK, int D, bool XDB, int EPI = 0, int NW = 4, bool TILED = false>
static void launch_sk3(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S) {
    constexpr int XL = (XDB ? 2 : 1) * MT * 16 * SK_ROWB, TL = MT * 16 * (NW * 16 + 4) * 4;
    constexpr int LDS = XL > TL ? XL : TL;
    auto kfn = gemm_skinny3_kernel<MT, NCK, D, XDB, EPI, NW, TILED>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    dim3 grid((N + 16 * NW - 1) / (16 * NW), S, (M + MT * 16 - 1) / (MT * 16)), block(64 * NW);
    hipLaunchKernelGGL(kfn, grid, block, LDS, s, x, W, out, M, N, K, pg_tune->wt_store & 1);
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
template <int MT, int EPI, int NW = 4, bool TILED = false, int D = 2>
static bool sk3_prod_nck(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, int nck) {
    switch (nck) {
        case 1: launch_sk3<MT, 1, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        case 2: launch_sk3<MT, 2, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        case 4: launch_sk3<MT, 4, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        case 8: launch_sk3<MT, 8, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        case 11: launch_sk3<MT, 11, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        case 16: launch_sk3<MT, 16, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        case 22: launch_sk3<MT, 22, D, true, EPI, NW, TILED>(s, x, W, out, M, N, K, S); return true;
        default: return false;
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
static bool sk3_prod_tiled(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S) {
    const int nck = K / SK_BK / S;
    if (nck * S * SK_BK != K || (N & 15)) return false;
    const int mrows = M < 128 ? M : 128;
    if (mrows <= 16) return sk3_prod_nck<1, EPI, 4, TILED>(s, x, Wt, out, M, N, K, S, nck);
    if (mrows <= 32) return sk3_prod_nck<2, EPI, 4, TILED>(s, x, Wt, out, M, N, K, S, nck);
    // tiled weights: 64-row M blocks for every shape (measured best at M = 128: the second reader of a
    // W tile hits L2, LDS per block halves -> more blocks per CU)
    // wide N (qkv, gate|up, gen_head at 65..128 rows): 128-column blocks of 8 waves -- the x tile is fetched and staged once per 128
    // columns instead of once per 64 (loop -11 ms at bs=64); stream_gemm bit 128 keeps the 4-wave block for A/B
    if (N % 128 == 0 && N >= 4096 && !(pg_tune->stream_gemm >= 0 && (pg_tune->stream_gemm & 128)))
        return sk3_prod_nck<4, EPI, 8, TILED>(s, x, Wt, out, M, N, K, S, nck);
    return sk3_prod_nck<4, EPI, 4, TILED>(s, x, Wt, out, M, N, K, S, nck);
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
    else if constexpr (EPI == 1) { static_assert(EPI != 1 || MS == 1, "transposed epilogues: MS = 1"); skinny_store_swiglu<MT, 4>(smem, acc, (bf16*)out, M, N / 2, mbase, blockIdx.x, w, g, lr, tid); }
    else { static_assert(EPI != 0 || MS == 1, "transposed epilogues: MS = 1"); skinny_store_tile<MT, 4>(smem, acc, out + (long)split * M * N, M, N, mbase, blockIdx.x * 64, w, g, lr, tid); }
    if constexpr (PROF) { if (pw) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); } }
}
unsigned long long* g_sk4_prof = nullptr;       // microbenchmark only
template <int MT, int NCK, int XD, int WD, int EPI, int OCC, int ABL = 0, int MS = 1>
static void launch_sk4(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S) {
    constexpr int XL = XD * MT * 16 * 256 + ((ABL & 512) ? 4096 * MS : 0), TL = MT * 16 * (4 * 16 + 4) * 4;
    constexpr int LDS = XL > TL ? XL : TL;
    auto kfn = gemm_sk4_kernel<MT, NCK, XD, WD, EPI, OCC, ABL, MS>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    dim3 grid((N + 63) / 64, S, (M + MT * 16 - 1) / (MT * 16)), block(256 * MS);
    hipLaunchKernelGGL(kfn, grid, block, LDS, s, x, Wt, out, M, N, K, g_sk4_prof);
}
template <int MT, int XD, int WD, int EPI, int OCC, int ABL = 0, int MS = 1>
static bool sk4_nck(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S) {
    const int nck = K / SK_BK / S;
    if (nck * S * SK_BK != K || (N & 15)) return false;
    switch (nck) {
        case 2: launch_sk4<MT, 2, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S); return true;
        case 4: launch_sk4<MT, 4, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S); return true;
        case 8: launch_sk4<MT, 8, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S); return true;
        case 11: launch_sk4<MT, 11, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S); return true;
        case 16: launch_sk4<MT, 16, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S); return true;
        case 22: launch_sk4<MT, 22, XD, WD, EPI, OCC, ABL, MS>(s, x, Wt, out, M, N, K, S); return true;
        default: return false;
    }
}

void launch_gemm_skinny_v1(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S);
// variant table for the microbenchmark (tools/skinny_sweep.py): returns BK (0 = unsupported)
static bool launch_gemm_skinny_tiled_only(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S);
int launch_gemm_skinny_variant(hipStream_t s, int variant, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S) {
    switch (variant) {
        case 0: launch_gemm_skinny(s, x, W, out, M, N, K, S, nullptr); return 128;         // production, row-major W (v3, falls back to v1)
        case 1: launch_gemm_skinny_v1(s, x, W, out, M, N, K, S); return 128;               // v1: 1-deep prefetch
        case 2: return launch_gemm_skinny_tiled_only(s, x, W, out, M, N, K, S) ? 128 : 0;   // production dispatch on the TILED decode copy
        case 20: return sk3_dispatch<2, true>(s, x, W, out, M, N, K, S) ? 128 : 0;         // v3 ring 2, x double-buffered
        case 24: return sk3_prod_nck<4, 0>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;   // 64-row M blocks (grid.z = M/64)
        case 40: {   // MFMA tile kernel (128x128x64, glds) with split-K expressed through the batch strides
            if (K % (64 * S)) return 0;
            GemmA a; a.ptr = x; a.lda = K; a.strideA = K / S;
            GemmEpi e; e.out = out; e.out_f32 = 1; e.ldc = N; e.strideC = (long)M * N;
            launch_gemm<bf16>(s, a, W, K, K / S, e, M, N, K / S, S);
            return 64;
        }
        case 50: return sk3_prod_nck<8, 0, 4, true>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;   // tiled W layout
        case 51: return sk3_prod_nck<4, 0, 4, true>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;   // tiled W, 64-row M blocks
        // v4 (x by LDS-DMA): MT / x ring depth / W ring depth / min blocks per CU
        case 271: return sk4_nck<4, 3, 3, 4, 2, 32>(s, x, W, out, M, N, K, S) ? 128 : 0;    // stamped (tools/sk4_profile.py): 64-row blocks
        case 274: return sk4_nck<8, 3, 3, 4, 1, 32>(s, x, W, out, M, N, K, S) ? 128 : 0;    // stamped: 128-row blocks
        case 71: return sk4_nck<4, 3, 3, 0, 2>(s, x, W, out, M, N, K, S) ? 128 : 0;
        case 74: return sk4_nck<8, 3, 3, 0, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;
        case 160: return sk4_nck<2, 4, 3, 4, 4>(s, x, W, out, M, N, K, S) ? 128 : 0;        // 32-row blocks
        case 161: return sk4_nck<1, 4, 3, 4, 4>(s, x, W, out, M, N, K, S) ? 128 : 0;        // 16-row blocks
        case 150: return sk4_nck<8, 3, 3, 4, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;        // v123 with write-through stores
        case 151: return sk4_nck<4, 3, 3, 4, 2>(s, x, W, out, M, N, K, S) ? 128 : 0;        // 64-row blocks, write-through stores
        // ablations of v123 (MT 8, XD 3, WD 3, direct stores)
        case 141: return sk4_nck<8, 3, 3, 2, 1, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;     // no x
        case 142: return sk4_nck<8, 3, 3, 2, 1, 2>(s, x, W, out, M, N, K, S) ? 128 : 0;     // no MFMA
        case 143: return sk4_nck<8, 3, 3, 2, 1, 3>(s, x, W, out, M, N, K, S) ? 128 : 0;     // no x, no MFMA
        case 144: return sk4_nck<8, 3, 3, 2, 1, 4>(s, x, W, out, M, N, K, S) ? 128 : 0;     // no stores
        case 145: return sk4_nck<8, 3, 3, 2, 1, 5>(s, x, W, out, M, N, K, S) ? 128 : 0;     // no x, no stores
        case 146: return sk4_nck<8, 3, 3, 2, 1, 6>(s, x, W, out, M, N, K, S) ? 128 : 0;     // no MFMA, no stores
        case 147: return sk4_nck<8, 3, 3, 2, 1, 7>(s, x, W, out, M, N, K, S) ? 128 : 0;     // W only
        // 8 waves per block (two row halves): LDS reads of one wave under the MFMAs of the other
        case 130: return sk4_nck<8, 3, 3, 2, 1, 0, 2>(s, x, W, out, M, N, K, S) ? 128 : 0;
        case 134: return sk4_nck<4, 3, 3, 2, 2, 0, 2>(s, x, W, out, M, N, K, S) ? 128 : 0;    // 64-row blocks, 8 waves of 2 m-tiles
        // round 4, hazard screen under background memory load (tools/sk4_load_stress.py): the 64-row production block and protocol variants
        case 300: return sk4_nck<4, 4, 3, 4, 2, 64>(s, x, W, out, M, N, K, S) ? 128 : 0;         // production copy: chunk c+1 retired at chunk c's barrier
        case 301: return sk4_nck<4, 4, 3, 4, 2, 64 | 8>(s, x, W, out, M, N, K, S) ? 128 : 0;     // vmcnt(0) in front of every barrier (no counted wait at all)
        case 302: return sk4_nck<4, 4, 3, 4, 2, 0>(s, x, W, out, M, N, K, S) ? 128 : 0;          // round-2 form: chunk c retired at its own barrier
        case 303: return sk4_nck<4, 4, 3, 4, 2, 64 | 128>(s, x, W, out, M, N, K, S) ? 128 : 0;   // pieces issued in reverse order
        case 304: return sk4_nck<4, 4, 3, 4, 2, 64 | 256>(s, x, W, out, M, N, K, S) ? 128 : 0;   // s_nop padding behind every DMA
        case 305: return sk4_nck<4, 4, 3, 4, 2, 64 | 8 | 512>(s, x, W, out, M, N, K, S) ? 128 : 0;   // a dummy fifth DMA behind every group (+ vmcnt(0) waits: the counted waits do not know it)
        case 306: return sk4_nck<4, 4, 3, 4, 1, 64>(s, x, W, out, M, N, K, S) ? 128 : 0;         // production copy, one block per CU
        case 310: return sk4_nck<4, 4, 3, 4, 2, 64 | 4096>(s, x, W, out, M, N, K, S) ? 128 : 0;  // rotation swizzle instead of XOR
        case 308: return sk4_nck<4, 4, 3, 4, 2, 64 | 1024>(s, x, W, out, M, N, K, S) ? 128 : 0;  // wave w issues pieces w, w+4, w+8, w+12
        case 309: return sk4_nck<4, 4, 3, 4, 2, 64 | 2048>(s, x, W, out, M, N, K, S) ? 128 : 0;  // LDS row placement permuted (rows ^ 12 within a 16-row group)
        case 307: return sk4_nck<2, 4, 3, 4, 4, 64>(s, x, W, out, M, N, K, S) ? 128 : 0;         // 32-row blocks (two pieces per wave)
        // round 4: W register-ring depth of the wide-N production block (64 rows x 128 columns, 8 waves, tiled W) -- bytes in flight per block
        case 400: return sk3_prod_nck<4, 0, 8, true, 2>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 401: return sk3_prod_nck<4, 0, 8, true, 3>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 402: return sk3_prod_nck<4, 0, 8, true, 4>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 403: return sk3_prod_nck<4, 0, 8, true, 6>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 121: return sk4_nck<8, 3, 8, 2, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;     // deep W ring, direct 16-byte stores
        case 123: return sk4_nck<8, 3, 3, 2, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;     // shallow W ring, direct stores
        default: return 0;
    }
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
void launch_gemm_skinny(hipStream_t s, const bf16* x, const bf16* W, float* out, int M, int N, int K, int S, const bf16* Wt) {
    if (M <= 0) return;
    if (sk4_prod<4>(s, x, Wt, out, M, N, K, S)) return;
    if (Wt && sk3_prod_tiled<0, true>(s, x, Wt, out, M, N, K, S)) return;
    if (!sk3_prod<0>(s, x, W, out, M, N, K, S)) launch_gemm_skinny_v1(s, x, W, out, M, N, K, S);
}
static bool launch_gemm_skinny_tiled_only(hipStream_t s, const bf16* x, const bf16* Wt, float* out, int M, int N, int K, int S) {
    return sk4_prod<4>(s, x, Wt, out, M, N, K, S) || sk3_prod_tiled<0, true>(s, x, Wt, out, M, N, K, S);
}
// gate|up GEMM with the SwiGLU gate fused (S = 1): h bf16 [M, N/2].  Returns false when the
// shape has no fused instantiation (caller falls back to slabs + silu_mul kernel).
bool launch_gemm_skinny_swiglu(hipStream_t s, const bf16* x, const bf16* W, bf16* h, int M, int N, int K, const bf16* Wt) {
    if (M <= 0) return true;
    if (sk4_prod<3>(s, x, Wt, (float*)h, M, N, K, 1)) return true;
    if (Wt && sk3_prod_tiled<1, true>(s, x, Wt, (float*)h, M, N, K, 1)) return true;
    return sk3_prod<1>(s, x, W, (float*)h, M, N, K, 1);
}
