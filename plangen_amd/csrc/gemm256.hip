// 256x256x64 bf16 MFMA GEMM for gfx950, eight waves (2 x 4), 128 KiB of LDS, one block per CU.
//
//   C[M,N] = A[M,K] . W[N,K]^T      (same loaders / epilogue as gemm.hip's 128x128 kernel)
//
// Schedule (designed for this kernel; the numbers below are what the synchronisation relies on):
//
// * A K tile (64 wide) is staged as FOUR 16 KiB half-tiles named after the C quadrant that
//   consumes them, not after their position in memory:
//       HA0 = A rows {wr*128 + [0,64)}      (m-tiles 0-3 of every wave)      read in phase 1
//       HB0 = W rows {wc*64  + [0,32)}      (n-tiles 0-1 of every wave)      read in phase 1
//       HB1 = W rows {wc*64  + 32 + [0,32)} (n-tiles 2-3)                    read in phase 2
//       HA1 = A rows {wr*128 + 64 + [0,64)} (m-tiles 4-7)                    read in phase 3
//   Every wave owns a 128x64 piece of C = 8 x 4 MFMA tiles and walks its four 64x32 quadrants in
//   the order (0,0) (0,1) (1,1) (1,0): one quadrant = 16 MFMAs (4 m x 2 n x 2 k-steps) per phase,
//   each phase needs only ONE new half-tile's fragments (phase 4 none: B0 stays in registers).
// * Two LDS buffers (even / odd K tile).  Because a half-tile is dead as soon as its fragments
//   are in registers, its slot is re-staged for tile kt+2 long before the tile ends:
//       phase 1 of tile kt stages HB1(kt+1), phase 2 HA1(kt+1), phase 3 HA0(kt+2), phase 4 HB0(kt+2)
//   i.e. the global_load_lds stream is simply "tile after tile, halves in the order HA0 HB0 HB1
//   HA1", running 5-6 phases (more than one whole K tile) ahead of its consumer.  Four half-tiles
//   (64 KiB per CU) are in flight at any time; the loop never waits for vmcnt(0).
// * WAR: a slot is re-staged >= 2 phases after the phase that read it (HA0/HB0: read P1, staged
//   P3/P4; HB1: read P2, staged P1 of the next tile; HA1: read P3, staged P2 of the next tile).
//   RAW: the phase BEFORE a half-tile is read ends its load section with s_waitcnt vmcnt(8)
//   (8 = 2 loads x the 4 half-tiles issued after the one needed), then the phase barriers.
//   Both margins hold with the two wave groups running one barrier apart (below).
// * Every phase = [ds_read fragments, issue 2 global_load_lds, counted vmcnt] s_barrier
//   [lgkmcnt(0), 16 MFMAs under s_setprio 1] s_barrier.  Waves with wr == 1 pass one extra barrier
//   before the loop, so on every SIMD one wave is in its MFMA section while the other one is in
//   its load section: LDS reads and address arithmetic hide under the other wave's MFMAs.
// * K tiles past the end are staged from the last valid tile (never read): the loop is
//   branch-free and the vmcnt counts stay exact.
#include <type_traits>
#include "gemm_common.h"

#define G2_BK 64
#define G2_HALF 16384
#define G2_BUF (4 * G2_HALF)
#define G2_LDS (2 * G2_BUF)

__device__ __forceinline__ void g2_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <class AL, class EP, bool STAGGER, int ABL = 0, int SCHED = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(AL al, const bf16* __restrict__ W, long ldb, long strideA, long strideB,
                                                     long strideA2, long strideB2, EP ep, int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int batch = blockIdx.y, batch2 = blockIdx.z;
    const long coff = (long)batch * ep.e.strideC + (long)batch2 * ep.e.strideC2;
    const long roff = (long)batch * ep.e.strideR;
    al.A_offset(strideA * batch + strideA2 * batch2);
    const bf16* Wb = W + strideB * batch + strideB2 * batch2;
    const int nk = K / G2_BK;
    const int NT = ntm * ntn, G = gridDim.x;
    const bool vec = ep.vec_ok(coff, roff);

    // staging: wave w, instruction i covers LDS rows (w*2+i)*8 .. +7 of a half-tile
    const int sr0 = (w * 2) * 8 + (l >> 3), sr1 = sr0 + 8;
    const int sc[2] = {((l & 7) ^ ((sr0 >> 1) & 7)) * 8, ((l & 7) ^ ((sr1 >> 1) & 7)) * 8};
    char* const swave = smem + w * 2048;

    const int wr = w >> 2, wc = w & 3, g = l >> 4, lr = l & 15;
    // fragment addresses: row*128 + ((ks*4+g) ^ ((row>>1)&7))*16; (row>>1)&7 == (lr>>1)&7 for every tile of this lane
    const int swz = (lr >> 1) & 7;
    const int aoff0 = (wr * 64 + lr) * 128 + ((g ^ swz) << 4), aoff1 = aoff0 ^ 64;
    const int boff0 = 2 * G2_HALF + (wc * 32 + lr) * 128 + ((g ^ swz) << 4), boff1 = boff0 ^ 64;

    const bf16* wrow[4];
    int m0 = 0, n0 = 0;
    // persistent tile loop: round `base` handles tiles [base, base+G); inside a round block b takes the
    // XCD-grouped position, and consecutive tile indices walk 4 tile rows x all tile columns, so the 32
    // tiles resident on one XCD share 4 A panels and 8 W panels in its L2.
    auto setup = [&](int base) -> bool {
        const int nr = min(G, NT - base);
        if ((int)blockIdx.x >= nr) return false;
        const int t = base + xcd_remap(blockIdx.x, nr);
        const int per = 4 * ntn, mg = t / per, rem = t - mg * per;
        const int gm = min(4, ntm - mg * 4);
        m0 = (mg * 4 + rem % gm) * 256; n0 = (rem / gm) * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i ? sr1 : sr0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                al.init(h * 2 + i, m0 + (r >> 6) * 128 + h * 64 + (r & 63));
                const int n = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
                wrow[h * 2 + i] = Wb + (long)(n < N ? n : N - 1) * ldb;
            }
        }
        return true;
    };
    auto stage_A = [&](int h, int T) {
        const int k0 = (T < nk ? T : nk - 1) * G2_BK;
        char* d = swave + (T & 1) * G2_BUF + h * G2_HALF;
        al.set_ktile(k0);
        glds16(al.ptr(h * 2 + 0, k0 + sc[0]), d);
        glds16(al.ptr(h * 2 + 1, k0 + sc[1]), d + 1024);
    };
    auto stage_B = [&](int h, int T) {
        const int k0 = (T < nk ? T : nk - 1) * G2_BK;
        char* d = swave + (T & 1) * G2_BUF + 2 * G2_HALF + h * G2_HALF;
        glds16(wrow[h * 2 + 0] + k0 + sc[0], d);
        glds16(wrow[h * 2 + 1] + k0 + sc[1], d + 1024);
    };
    // tile 0 complete + the first two halves of tile 1, in stream order
    auto prologue = [&]() { stage_A(0, 0); stage_B(0, 0); stage_B(1, 0); stage_A(1, 0); stage_A(0, 1); stage_B(0, 1); };

    f32x4 acc[8][4];
    bf16x8 a[4][2], b0[2][2], b1[2][2];

    // operands swapped (W fragment first): D rows = n, D cols = m, so a lane holds 4 CONSECUTIVE columns
    // n = g*4 .. +3 of row m = lr -> one 16-byte store per MFMA tile in the epilogue
#define G2_MFMA_SECTION(MH, BF, NH)                                                                         \
    g2_barrier();                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                    \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                \
                acc[MH * 4 + mt][NH * 2 + nt] =                                                             \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[nt][ks], a[mt][ks], acc[MH * 4 + mt][NH * 2 + nt], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    g2_barrier();

    bool have = setup(0);
    if (have) prologue();
    for (int base = 0; have; ) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        g2_barrier();
        if constexpr (SCHED == 1) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                b0[nt][0] = *(const bf16x8*)(smem + boff0 + nt * 2048);
                b0[nt][1] = *(const bf16x8*)(smem + boff1 + nt * 2048);
            }
        }
        if (STAGGER && wr == 1) g2_barrier();

        if constexpr (SCHED == 1) {
            // Variant: the 4 HB0 fragment reads move from phase 1 (12 ds_reads) to phase 4 of the PREVIOUS
            // K tile (0 reads) into the register set B1 just vacated -> 8/4/8/4 reads per phase; the two
            // B register sets swap roles every K tile.  Counted waits: P1 vmcnt(8), P2 vmcnt(8),
            // P3 vmcnt(6) (retires HB0(kt+1), issued 3 half-tiles earlier, and HA0(kt+1) before it), P4 none.
#define G2_TILE(KT, BX, BY)                                                                                 \
            {                                                                                               \
                const char* buf = smem + ((KT) & 1) * G2_BUF;                                               \
                const char* nbuf = smem + (((KT) + 1) & 1) * G2_BUF;                                        \
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                          \
                    a[mt][0] = *(const bf16x8*)(buf + aoff0 + mt * 2048);                                   \
                    a[mt][1] = *(const bf16x8*)(buf + aoff1 + mt * 2048);                                   \
                }                                                                                           \
                stage_B(1, (KT) + 1);                                                                       \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                            \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                G2_MFMA_SECTION(0, BX, 0)                                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                          \
                    BY[nt][0] = *(const bf16x8*)(buf + G2_HALF + boff0 + nt * 2048);                        \
                    BY[nt][1] = *(const bf16x8*)(buf + G2_HALF + boff1 + nt * 2048);                        \
                }                                                                                           \
                stage_A(1, (KT) + 1);                                                                       \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                            \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                G2_MFMA_SECTION(0, BY, 1)                                                                   \
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                          \
                    a[mt][0] = *(const bf16x8*)(buf + G2_HALF + aoff0 + mt * 2048);                         \
                    a[mt][1] = *(const bf16x8*)(buf + G2_HALF + aoff1 + mt * 2048);                         \
                }                                                                                           \
                stage_A(0, (KT) + 2);                                                                       \
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                            \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                G2_MFMA_SECTION(1, BY, 1)                                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                          \
                    BY[nt][0] = *(const bf16x8*)(nbuf + boff0 + nt * 2048);                                 \
                    BY[nt][1] = *(const bf16x8*)(nbuf + boff1 + nt * 2048);                                 \
                }                                                                                           \
                stage_B(0, (KT) + 2);                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                G2_MFMA_SECTION(1, BX, 0)                                                                   \
            }
            for (int kt = 0; kt < nk; kt += 2) {
                G2_TILE(kt, b0, b1)
                if (kt + 1 < nk) G2_TILE(kt + 1, b1, b0)
            }
#undef G2_TILE
        } else
        for (int kt = 0; kt < nk; ++kt) {
            const char* buf = smem + (kt & 1) * G2_BUF;
            // ---- phase 1: quadrant (0,0): fragments of HB0 and HA0; stage HB1(kt+1)
            if (!(ABL & 2) || kt == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    b0[nt][0] = *(const bf16x8*)(buf + boff0 + nt * 2048);
                    b0[nt][1] = *(const bf16x8*)(buf + boff1 + nt * 2048);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    a[mt][0] = *(const bf16x8*)(buf + aoff0 + mt * 2048);
                    a[mt][1] = *(const bf16x8*)(buf + aoff1 + mt * 2048);
                }
            }
            if (!(ABL & 1)) stage_B(1, kt + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(0, b0, 0)
            // ---- phase 2: quadrant (0,1): fragments of HB1; stage HA1(kt+1)
            if (!(ABL & 2) || kt == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    b1[nt][0] = *(const bf16x8*)(buf + G2_HALF + boff0 + nt * 2048);
                    b1[nt][1] = *(const bf16x8*)(buf + G2_HALF + boff1 + nt * 2048);
                }
            }
            if (!(ABL & 1)) stage_A(1, kt + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(0, b1, 1)
            // ---- phase 3: quadrant (1,1): fragments of HA1; stage HA0(kt+2)
            if (!(ABL & 2) || kt == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    a[mt][0] = *(const bf16x8*)(buf + G2_HALF + aoff0 + mt * 2048);
                    a[mt][1] = *(const bf16x8*)(buf + G2_HALF + aoff1 + mt * 2048);
                }
            }
            if (!(ABL & 1)) stage_A(0, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(1, b1, 1)
            // ---- phase 4: quadrant (1,0): B0 still in registers; stage HB0(kt+2)
            if (!(ABL & 1)) stage_B(0, kt + 2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(1, b0, 0)
        }
        if (STAGGER && wr == 0) g2_barrier();
        // every wave is past its last ds_read: the next tile's first loads go out before this tile's
        // stores (same vmcnt counter, but loads return in order among themselves and the clamped tail
        // stages of this tile were issued earlier by the same wave to the same LDS bytes)
        const int em0 = m0 + wr * 128 + lr, en0 = n0 + wc * 64 + g * 4;
        base += G;
        have = base < NT && setup(base);
        if (have) prologue();
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ep.store4(coff, roff, em0 + mt * 16, en0 + nt * 16, acc[mt][nt], vec);
    }
#undef G2_MFMA_SECTION
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // clamped tail stages must not outlive the block's LDS
}

int g_gemm256 = 1;          // pg_set_option("gemm256", 0/1/2): 0 off, 1 staggered, 2 lock-step (debug)

template <class AL>
static void launch256(hipStream_t s, AL al, const bf16* W, long ldb, long strideA, long strideB, long strideA2, long strideB2,
                      const Epi<bf16>& ep, int M, int N, int K, int batch, int batch2) {
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    const int per_batch = 256 / (batch * batch2) > 8 ? 256 / (batch * batch2) : 8;   // blocks per (batch) slice: one per CU overall
    dim3 grid(ntm * ntn < per_batch ? ntm * ntn : per_batch, batch, batch2), block(512);
    if (g_gemm256 >= 11 && g_gemm256 <= 13) {
        if constexpr (std::is_same<AL, PlainLoaderB<bf16>>::value) {
            void (*kfn)(AL, const bf16*, long, long, long, long, long, Epi<bf16>, int, int, int, int, int) =
                g_gemm256 == 11 ? gemm256_kernel<AL, Epi<bf16>, true, 1> : g_gemm256 == 12 ? gemm256_kernel<AL, Epi<bf16>, true, 2> : gemm256_kernel<AL, Epi<bf16>, true, 3>;
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
            hipLaunchKernelGGL(kfn, grid, block, G2_LDS, s, al, W, ldb, strideA, strideB, strideA2, strideB2, ep, M, N, K, ntm, ntn);
        }
        return;
    }
    if (g_gemm256 == 3) {
        auto kfn = gemm256_kernel<AL, Epi<bf16>, true, 0, 1>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS); attr = true; }
        hipLaunchKernelGGL(kfn, grid, block, G2_LDS, s, al, W, ldb, strideA, strideB, strideA2, strideB2, ep, M, N, K, ntm, ntn);
        return;
    }
    if (g_gemm256 == 2) {
        auto kfn = gemm256_kernel<AL, Epi<bf16>, false>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS); attr = true; }
        hipLaunchKernelGGL(kfn, grid, block, G2_LDS, s, al, W, ldb, strideA, strideB, strideA2, strideB2, ep, M, N, K, ntm, ntn);
    } else {
        auto kfn = gemm256_kernel<AL, Epi<bf16>, true>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS); attr = true; }
        hipLaunchKernelGGL(kfn, grid, block, G2_LDS, s, al, W, ldb, strideA, strideB, strideA2, strideB2, ep, M, N, K, ntm, ntn);
    }
}

// Takes the shapes the 256^2 tile fills well; everything else stays on the 128^2 kernel.
bool gemm256_try(hipStream_t s, const GemmA& a, const bf16* W, long ldb, long strideB, const GemmEpi& e, int M, int N, int K,
                 int batch, int batch2, long strideB2) {
    if (!g_gemm256 || K % G2_BK || K < 2 * G2_BK) return false;
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    // padding waste of the last tile row / column, and enough tiles to fill the chip
    if ((long)ntm * 256 * ntn * 256 > (long)M * N * 5 / 4) return false;
    if ((long)ntm * ntn * batch * batch2 < 200) return false;
    Epi<bf16> ep{e, M, N};
    if (a.kind == 0) {
        PlainLoaderB<bf16> al; al.A = (const bf16*)a.ptr; al.lda = a.lda; al.M = M;
        launch256(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2);
    } else {
        ConvLoaderB<bf16> al; al.X = (const bf16*)a.ptr; al.zeros = (const bf16*)a.zeros;
        al.Hi = a.Hi; al.Wi = a.Wi; al.Cin = a.Cin; al.up = a.up; al.stride2 = (a.kind == 2);
        al.Ho = al.stride2 ? a.Hi / 2 : (a.Hi << a.up); al.Wo = al.stride2 ? a.Wi / 2 : (a.Wi << a.up); al.M = M;
        launch256(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2);
    }
    return true;
}
