// "Big-wave" 256x256 bf16 MFMA GEMM for gfx950 (round 5 experiment):  C[M,N] = A[M,K] . W[N,K]^T.
//
// Why: the MFMA-only probe (bench_kernels.hip::mfma_lds_fed_kernel, profiles/r05_c) shows that ONE wave per SIMD owning a 128 x 128 piece of C
// (4 x 4 tiles of v_mfma_f32_32x32x16_bf16, 256 accumulator registers in AGPRs) and reading its fragments from LDS with ds_read_b128
// sustains 1.9 PF on random operands (2.44 PF on constant ones) -- the chip's power-limited rate -- while gemm256_kernel (two waves per
// SIMD, 128 x 64 per wave, 16x16x32 MFMAs, two barriers per phase) reaches 1.0-1.09 PF.  This kernel is that main loop with a staging
// stream around it:
//
//   block = 4 waves (2 x 2), 256 x 256 of C, one persistent block per CU, 128 KiB of LDS = 4 stages x (A 256 rows x 64 B | W 256 rows x 64 B)
//   K tile = 32 (two k-steps of 16).  Staging: global_load_lds, 8 wave-instructions (1 KiB each) per wave per K tile; LDS rows are 64 B with
//   the 16-byte chunk position XORed with (row >> 2) & 3 through the SOURCE address, so a ds_read_b128 lane group (rows {0-3, 12-15, 20-27}
//   of one logical chunk) touches every bank once.
//   Loop, per K tile t (stage t & 3):   [reads of (t, k-step 1)] 16 MFMAs on (t, 0);  vmcnt(8) -> tile t+1 landed;  s_barrier -> everyone's
//   tile t+1 landed and everyone is past tile t-1;  stage tile t+3 into the slot of t-1;  [reads of (t+1, 0)] 16 MFMAs on (t, 1).
//   One barrier per 32 MFMAs per wave; a staged tile is read one barrier after the barrier that retires it + 1 (tile t+1 is retired in
//   iteration t and first read behind that barrier -- the fragments of (t+1, 0) are only CONSUMED in iteration t+1).
//   Operands swapped in the MFMA (D = W . A^T): acc element r of tile (i, j) is C[m = i*32 + lane % 32][n = j*32 + 8 (r / 4) + 4 (lane / 32) + r % 4]:
//   four consecutive columns per register quad -> the shared 16-byte epilogue (gemm_common.h store4_batch).
//
// RESULT (MI355X, `tools/tile_height_sweep.sh 13285 "1 17"`): bit-identical to the 16x16x32 kernels (maxdiff 0 on all four prefill shapes) but
// SLOWER: 727 / 631 / 771 / 794 TF/s on qkv / o / gate|up / down against 997 / 942 / 1082 / 1093 for gemm256_kernel.  With one wave per
// SIMD nothing runs under the 8 global_load_lds issues per K tile (~60-100 cycles of issue each, MI355X_MICROARCH.md price list) and under the
// barrier: gemm256_kernel's second wave per SIMD is what hides exactly that.  Kept in libplangen_diag.so only (PgDiagHooks::gemm_big_wave,
// gemm256 option bit 4) as the record; the probe that motivated it is bench_kernels.hip::mfma_lds_fed_kernel.
#include <type_traits>
#include "gemm_common.h"
#include "diag.h"

#define BW_BK 32
#define BW_STAGES 4
#define BW_OP_BYTES (256 * 64)                 // one operand tile: 256 rows x 64 B
#define BW_STAGE_BYTES (2 * BW_OP_BYTES)
#define BW_LDS (BW_STAGES * BW_STAGE_BYTES)    // 128 KiB

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <class EP>
__global__ __launch_bounds__(256, 1) void gemm_bw_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ W, long ldb, EP ep,
                                                        int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1, lr = l & 31, kg = l >> 5;
    const int nk = K / BW_BK;
    const int NT = ntm * ntn, G = gridDim.x;
    const bool vec = ep.vec_ok(0, 0);

    // staging: wave w, instruction i in [0, 4) (A) / [4, 8) (W) covers operand rows (i4 * 4 + w) * 16 .. +15; lane: row l >> 2, LDS chunk position l & 3
    const int srow = l >> 2, spos = l & 3;
    const bf16* arow[4]; const bf16* brow[4];
    int m0 = 0, n0 = 0;
    auto setup = [&](int base) __attribute__((always_inline)) -> bool {
        const int nr = min(G, NT - base);
        if ((int)blockIdx.x >= nr) return false;
        const int t = base + xcd_remap(blockIdx.x, nr);
        const int per = 4 * ntn, mg = t / per, rem = t - mg * per;
        const int gm = min(4, ntm - mg * 4);
        m0 = (mg * 4 + rem % gm) * 256; n0 = (rem / gm) * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (i * 4 + w) * 16 + srow;                       // operand row 0 .. 255
            const int c = (spos ^ ((r >> 2) & 3)) * 8;                   // swizzled SOURCE chunk (elements)
            const int m = m0 + r, n = n0 + r;
            arow[i] = A + (long)(m < M ? m : M - 1) * lda + c;
            brow[i] = W + (long)(n < N ? n : N - 1) * ldb + c;
        }
        return true;
    };
    auto stage = [&](int T) __attribute__((always_inline)) {
        const int k0 = (T < nk ? T : nk - 1) * BW_BK;                    // tiles past the end re-stage the last one (never read): counts stay exact
        char* d = smem + (T & (BW_STAGES - 1)) * BW_STAGE_BYTES + w * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(arow[i] + k0, d + i * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(brow[i] + k0, d + BW_OP_BYTES + i * 4096);
    };
    // fragment addresses inside a stage: row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); (row >> 2) & 3 == (lr >> 2) & 3 for every tile of this lane
    const int swz = (lr >> 2) & 3;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int pos = ((ks * 2 + kg) ^ swz) << 4;
        aoff[ks] = (wr * 128 + lr) * 64 + pos;
        boff[ks] = BW_OP_BYTES + (wc * 128 + lr) * 64 + pos;
    }
    bf16x8 af[2][4], bfr[2][4];
    auto rd = [&](int buf, int T, int ks) __attribute__((always_inline)) {
        const char* st = smem + (T & (BW_STAGES - 1)) * BW_STAGE_BYTES;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[buf][t] = *(const bf16x8*)(st + aoff[ks] + t * 2048);
            bfr[buf][t] = *(const bf16x8*)(st + boff[ks] + t * 2048);
        }
    };
    f32x16 acc[4][4];
#define BW_MFMA(BUF)                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                            \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[BUF][j], af[BUF][i], acc[i][j], 0, 0, 0);

    bool have = setup(0);
    if (have) { stage(0); stage(1); stage(2); }
    for (int base = 0; have; ) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                // tile 0 landed (tiles 1, 2 may be in flight)
        __builtin_amdgcn_s_barrier();
        rd(0, 0, 0);
        for (int t = 0; t < nk; ++t) {
            // part A: the reads of (t, k-step 1) go out between the MFMAs of (t, 0) -- one wave per SIMD: every issue slot that is not under an
            // executing MFMA idles the matrix core
            rd(1, t, 1);
            BW_MFMA(0)
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // tile t+1 landed (tile t+2 may be in flight)
            __builtin_amdgcn_s_barrier();                                // ... for every wave; every wave is past its reads of tile t-1
            // part B: stage tile t+3 into the slot of tile t-1 and read (t+1, 0), both between the MFMAs of (t, 1)
            stage(t + 3);
            rd(0, t + 1, 0);                                             // (t + 1 == nk: reads the re-staged last tile, never consumed)
            BW_MFMA(1)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        // the clamped tail stages (tiles nk, nk+1, nk+2) are still in flight into slots nobody reads any more; the next tile's stream starts behind them
        const int em0 = m0 + wr * 128 + lr, en0 = n0 + wc * 128 + kg * 4;
        base += G;
        have = base < NT && setup(base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // every wave has consumed its last fragments before the slots are re-staged
        if (have) { stage(0); stage(1); stage(2); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {                             // 8 fragments per batch: two 32-column tiles x four column quads
                int rows[8], cols[8]; f32x4 av[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const int j = jp * 2 + (f >> 2), q = f & 3;
                    rows[f] = em0 + i * 32; cols[f] = en0 + j * 32 + q * 8;
                    av[f] = (f32x4){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
                ep.template store4_batch<8>(0, 0, rows, cols, av, vec);
            }
    }
#undef BW_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Plain row-major A, single batch, act 0 / 1 epilogues (bias, residual); returns false for anything else.
bool gemm_bw_try(hipStream_t s, const GemmA& a, const bf16* W, long ldb, const GemmEpi& e, int M, int N, int K, int batch, int batch2) {
    if (!(pg_tune->gemm256 & 16) || a.kind != 0 || batch != 1 || batch2 != 1 || e.act > 1 || K % BW_BK || K < 4 * BW_BK) return false;
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    if ((long)ntm * ntn < 200 || (long)ntm * 256 * ntn * 256 > (long)M * N * 5 / 4) return false;
    Epi<bf16> ep{e, M, N};
    auto kfn = gemm_bw_kernel<Epi<bf16>>;
    if (!PG_DYN_LDS(kfn, BW_LDS)) return false;
    hipLaunchKernelGGL(kfn, dim3(ntm * ntn < 256 ? ntm * ntn : 256), dim3(256), BW_LDS, s, (const bf16*)a.ptr, a.lda, W, ldb, ep, M, N, K, ntm, ntn);
    return true;
}
