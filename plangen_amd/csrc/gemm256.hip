// 256x256x64 (and 512x128x64) bf16 MFMA GEMM for gfx950, eight waves, 128 / 160 KiB of LDS, one persistent
// block per CU.
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
//   TWO-PHASE FORM (PH == 2, the default since round 5; invariants restated in round 6 after ADVICE r5): phase A reads HB0, HA0, HB1 and
//   stages HB1(kt+1), HA1(kt+1); phase B reads HA1 and stages HA0(kt+2), HB0(kt+2).  A slot is therefore re-staged in the phase RIGHT
//   AFTER the one that read it (HA0 / HB0: read A(kt), staged B(kt); HA1: read B(kt), staged A(kt+1); HB1: read A(kt), staged A(kt+1)), and
//   with the groups one barrier apart the LEADING group issues that re-stage right behind the very barrier that ends the LAGGING group's
//   load section.  WAR rule of this form: every wave executes `s_waitcnt lgkmcnt(0)` BEFORE the barrier that ends its load section (its
//   fragments are in registers when it arrives), so one barrier separates the last read of a slot from any global_load_lds into it.  (Round 5
//   had the lgkmcnt(0) BEHIND that barrier: only the DMA's memory latency separated the two.  tools/dma_isa_check.py checks the new rule:
//   a re-stage needs >= 2 barriers since the slot's last read, or 1 barrier with the read drained in front of it.)
//   RAW of the two-phase form: phase B's load section ends with vmcnt(6) (HB1(kt+1) and everything older landed), phase A's with vmcnt(8)
//   (HA1(kt) landed); both one full phase (two barriers) before the fragments are read.
// * Every phase = [ds_read fragments, issue 2 global_load_lds, counted vmcnt] s_barrier
//   [lgkmcnt(0), 16 MFMAs under s_setprio 1] s_barrier.  Waves with wr == 1 pass one extra barrier
//   before the loop, so on every SIMD one wave is in its MFMA section while the other one is in
//   its load section: LDS reads and address arithmetic hide under the other wave's MFMAs.
// * K tiles past the end are staged from the last valid tile (never read): the loop is
//   branch-free and the vmcnt counts stay exact.
// * Geometry is a template (WM x WN waves, 128x64 of C per wave): 2x4 = 256x256 (A half-tile 16 KiB = 2 loads
//   per thread, W half-tile 16 KiB = 2 loads, counted wait vmcnt(8)) is what runs; 4x2 = 512x128 (A half-tile
//   32 KiB = 4 loads, W half-tile 8 KiB = 1 load, vmcnt(10), all 160 KiB of LDS) is correct but slower (below).
//   Any 4 consecutive half-tiles of the stream are two A and two W halves, hence one count per geometry.
#include <type_traits>
#include "gemm_common.h"

#define G2_BK 64

__device__ __forceinline__ void g2_barrier() { asm volatile("s_barrier" ::: "memory"); }

// MT1 (round 5): m-tiles of a wave's SECOND row half (4 = the 256-row tile; 3 / 2 = 224 / 192 rows per block at WM = 2).  The tile height is
// picked per launch so that ceil(tiles / 256 CUs) x height is smallest: at the bench's packed prompt batch (13.4 k tokens) o_proj / down_proj are
// 424 tiles of 256 rows = 1.66 rounds on 256 CUs (17 % of the launch idle) but 480 tiles of 224 rows = 1.875 rounds of 7/8 the work.  Only
// the fragment reads and MFMAs of the second half shrink (phases 3 / 4: 4 x MT1 MFMAs... MT1 m-tiles x 2 n-tiles x 2 k-steps); the staging
// stream, its counted waits and the LDS layout are unchanged (the half-tile HA1 still brings 64 rows per wave row, 64 - 16 MT1 of them unused).
template <class AL, class EP, int WM, int WN, bool ROPE = false, int MT1 = 4, int PH = 4, bool GN = false>      // GN (round 6): the epilogue also emits GroupNorm partial sums -- its own instantiation: in the common one its
                                                                                                              // extra live registers spilled 27-61 VGPRs and cost the prefill GEMMs 1.5 %.  ROPE: the prefill QKV instantiation (RoPE + KV-write epilogue, act 3) -- its own kernel so its
__global__ __launch_bounds__(512) void gemm256_kernel(                 // register needs do not reach the other users of this template
AL al, const bf16* __restrict__ W, long ldb, long strideA, long strideB,
                                                     long strideA2, long strideB2, EP ep, int M, int N, int K, int ntm, int ntn) {
    static_assert(WM * WN == 8 && (WN == 2 || WN == 4), "eight waves, 128x64 of C each");
    static_assert(MT1 >= 1 && MT1 <= 4, "second row half: 1..4 m-tiles");
    static_assert(PH == 4 || (PH == 2 && WM == 2 && WN == 4), "two-phase schedule: 2x4 waves only (its counted waits assume 2 loads per half-tile)");
    constexpr int WROWS = 64 + 16 * MT1, NMT = 4 + MT1;          // rows / m-tiles of C per wave row
    constexpr int BM = WM * WROWS, BN = WN * 64;
    constexpr int AH = WM * 64 * 128, BH = WN * 32 * 128;        // bytes per A / W half-tile
    constexpr int BUF = 2 * AH + 2 * BH;                         // one K tile: [HA0 | HA1 | HB0 | HB1]
    constexpr int NA = WM, NB = WN / 2;                          // global_load_lds per thread per half-tile
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

    // staging: wave w, instruction i covers LDS rows (i*8 + w)*8 .. +7 of a half-tile; the swizzle term
    // (row>>1)&7 = (w&1)*4 + (l>>4) does not depend on i
    const int srow = w * 8 + (l >> 3);                           // + i*64
    const int sc = ((l & 7) ^ (((w & 1) << 2) + (l >> 4))) * 8;  // swizzled SOURCE chunk (elements)
    char* const swave = smem + w * 1024;

    const int wr = w / WN, wc = w % WN, g = l >> 4, lr = l & 15;
    // fragment addresses: row*128 + ((ks*4+g) ^ ((row>>1)&7))*16; (row>>1)&7 == (lr>>1)&7 for every tile of this lane
    const int swz = (lr >> 1) & 7;
    const int aoff0 = (wr * 64 + lr) * 128 + ((g ^ swz) << 4), aoff1 = aoff0 ^ 64;
    const int boff0 = 2 * AH + (wc * 32 + lr) * 128 + ((g ^ swz) << 4), boff1 = boff0 ^ 64;

    const bf16* wrow[2 * NB];
    int m0 = 0, n0 = 0;
    // persistent tile loop: round `base` handles tiles [base, base+G); inside a round block b takes the
    // XCD-grouped position, and consecutive tile indices walk 4 tile rows x all tile columns, so the 32
    // tiles resident on one XCD share 4 A panels and 8 W panels in its L2.
    auto setup = [&](int base) __attribute__((always_inline)) -> bool {
        const int nr = min(G, NT - base);
        if ((int)blockIdx.x >= nr) return false;
        const int t = base + xcd_remap(blockIdx.x, nr);
        const int per = 4 * ntn, mg = t / per, rem = t - mg * per;
        const int gm = min(4, ntm - mg * 4);
        m0 = (mg * 4 + rem % gm) * BM; n0 = (rem / gm) * BN;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = i * 64 + srow;                     // LDS row of the half-tile -> (wave row r>>6, row r&63)
                al.init(h * NA + i, m0 + (r >> 6) * WROWS + h * 64 + (r & 63));   // (MT1 < 4: rows 16 MT1 .. 63 of HA1 are staged but never read)
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = i * 64 + srow;
                const int n = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
                wrow[h * NB + i] = Wb + (long)(n < N ? n : N - 1) * ldb;
            }
        }
        return true;
    };
    auto stage_A = [&](auto hc, int T) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        const int k0 = (T < nk ? T : nk - 1) * G2_BK;
        char* d = swave + (T & 1) * BUF + h * AH;
        al.set_ktile(k0);
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(al.ptr(h * NA + i, k0 + sc), d + i * 8192);
    };
    auto stage_B = [&](auto hc, int T) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        const int k0 = (T < nk ? T : nk - 1) * G2_BK;
        char* d = swave + (T & 1) * BUF + 2 * AH + h * BH;
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(wrow[h * NB + i] + k0 + sc, d + i * 8192);
    };
    // tile 0 complete + the first two halves of tile 1, in stream order
    constexpr std::integral_constant<int, 0> H0{}; constexpr std::integral_constant<int, 1> H1{};
    auto prologue = [&]() __attribute__((always_inline)) { stage_A(H0, 0); stage_B(H0, 0); stage_B(H1, 0); stage_A(H1, 0); stage_A(H0, 1); stage_B(H0, 1); };

    f32x4 acc[NMT][4];
    bf16x8 a[4][2], b0[2][2], b1[2][2];

    // counted wait: the four half-tiles issued after the one needed next = 2 A + 2 W halves
#define G2_WAIT()                                                                                           \
    if constexpr (2 * NA + 2 * NB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   \
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    // operands swapped (W fragment first): D rows = n, D cols = m, so a lane holds 4 CONSECUTIVE columns
    // n = g*4 .. +3 of row m = lr -> one 16-byte store per MFMA tile in the epilogue
#define G2_MFMA_SECTION(MH, BF, NH)                                                                         \
    g2_barrier();                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
        _Pragma("unroll") for (int mt = 0; mt < (MH ? MT1 : 4); ++mt)                                       \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                \
                acc[MH * 4 + mt][NH * 2 + nt] =                                                             \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[nt][ks], a[mt][ks], acc[MH * 4 + mt][NH * 2 + nt], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    g2_barrier();

    // PH == 2 (round 5): the SAME stream and LDS layout with TWO phases per K tile instead of four -- phase A = quadrants (0,0) (0,1) on the
    // fragments of HB0, HA0, HB1 (32 MFMAs), phase B = quadrants (1,1) (1,0) on HA1 (8 MT1 MFMAs); each phase re-stages two half-tiles.  Half the
    // barriers per MFMA: the four-phase form measured only -3.5 % when a quarter of the MFMAs of phases 3 / 4 was removed (224-row tiles), i.e. its
    // phases are bound by their fixed part (two barriers, fragment reads, staging, counted wait), not by their 16 MFMAs.
#define G2_MFMA_SECTION2(MH, BFA, NHA, BFB, NHB)                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* WAR (round 6): fragments in registers BEFORE the barrier, see the header */ \
    g2_barrier();                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
        _Pragma("unroll") for (int mt = 0; mt < (MH ? MT1 : 4); ++mt)                                       \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                              \
                acc[MH * 4 + mt][NHA * 2 + nt] =                                                            \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(BFA[nt][ks], a[mt][ks], acc[MH * 4 + mt][NHA * 2 + nt], 0, 0, 0); \
                acc[MH * 4 + mt][NHB * 2 + nt] =                                                            \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(BFB[nt][ks], a[mt][ks], acc[MH * 4 + mt][NHB * 2 + nt], 0, 0, 0); \
            }                                                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    g2_barrier();

    bool have = setup(0);
    if (have) prologue();
    for (int base = 0; have; ) {
#pragma unroll
        for (int i = 0; i < NMT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // HB1(0) landed (younger: HA1(0) HA0(1) HB0(1))
        else { G2_WAIT() }
        g2_barrier();
        if (wr >= WM / 2) g2_barrier();                   // second wave group runs one barrier behind

        if constexpr (PH == 2) {
        for (int kt = 0; kt < nk; ++kt) {
            const char* buf = smem + (kt & 1) * BUF;
            // ---- phase A: quadrants (0,0) (0,1): fragments of HB0, HA0, HB1; stage HB1(kt+1), HA1(kt+1)
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
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                b1[nt][0] = *(const bf16x8*)(buf + BH + boff0 + nt * 2048);
                b1[nt][1] = *(const bf16x8*)(buf + BH + boff1 + nt * 2048);
            }
            stage_B(H1, kt + 1);
            stage_A(H1, kt + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // HA1(kt) landed (younger: HA0 HB0 HB1 HA1 of kt+1)
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION2(0, b0, 0, b1, 1)
            // ---- phase B: quadrants (1,1) (1,0): fragments of HA1; stage HA0(kt+2), HB0(kt+2)
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) {
                a[mt][0] = *(const bf16x8*)(buf + AH + aoff0 + mt * 2048);
                a[mt][1] = *(const bf16x8*)(buf + AH + aoff1 + mt * 2048);
            }
            stage_A(H0, kt + 2);
            stage_B(H0, kt + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // HB1(kt+1) (and the older HA0 / HB0(kt+1)) landed (younger: HA1(kt+1) HA0 HB0(kt+2))
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION2(1, b1, 1, b0, 0)
        }
        } else
        for (int kt = 0; kt < nk; ++kt) {
            const char* buf = smem + (kt & 1) * BUF;
            // ---- phase 1: quadrant (0,0): fragments of HB0 and HA0; stage HB1(kt+1)
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
            stage_B(H1, kt + 1);
            G2_WAIT()
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(0, b0, 0)
            // ---- phase 2: quadrant (0,1): fragments of HB1; stage HA1(kt+1)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                b1[nt][0] = *(const bf16x8*)(buf + BH + boff0 + nt * 2048);
                b1[nt][1] = *(const bf16x8*)(buf + BH + boff1 + nt * 2048);
            }
            stage_A(H1, kt + 1);
            G2_WAIT()
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(0, b1, 1)
            // ---- phase 3: quadrant (1,1): fragments of HA1; stage HA0(kt+2)
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) {
                a[mt][0] = *(const bf16x8*)(buf + AH + aoff0 + mt * 2048);
                a[mt][1] = *(const bf16x8*)(buf + AH + aoff1 + mt * 2048);
            }
            stage_A(H0, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(1, b1, 1)
            // ---- phase 4: quadrant (1,0): B0 still in registers; stage HB0(kt+2)
            stage_B(H0, kt + 2);
            G2_WAIT()
            __builtin_amdgcn_sched_barrier(0);
            G2_MFMA_SECTION(1, b0, 0)
        }
        if (wr < WM / 2) g2_barrier();
        // every wave is past its last ds_read: the next tile's first loads go out before this tile's
        // stores (same vmcnt counter, but loads return in order among themselves and the clamped tail
        // stages of this tile were issued earlier by the same wave to the same LDS bytes)
        const int em0 = m0 + wr * WROWS + lr, en0 = n0 + wc * 64 + g * 4, ecol0 = n0 + wc * 64;   // setup() below moves m0 / n0 on
        base += G;
        have = base < NT && setup(base);
        if (have) prologue();
        if constexpr (ROPE) {
            // Prefill QKV projection: RoPE(q), RoPE(k) and the KV-cache write from the accumulators.  W rows of the q / k heads
            // are interleaved [8 | 8] per n-tile (launch_interleave_qk): lanes g = 0,1 hold rotary columns j = 8t + 4g .. +3, lanes g = 2,3
            // (lane ^ 32) the partners j + 64 of the same token -- one cross-half exchange, then out[j] = x[j] cos - x[j+64] sin,
            // out[j+64] = x[j+64] cos + x[j] sin (rotate_half form, the arithmetic of rope_kv_kernel), 4 bf16 = 8 bytes per store.
            // The fp32 q|k|v tensor and the separate RoPE / KV-fill pass disappear.
            const RopeEpi& R = ep.e.rope;
            const int HDm = R.nh * 128;
            // per-row metadata of the lane's 8 rows first (two dependent round trips for all of them); the cos / sin rows of m-tile mt+1 are
            // requested under the stores of m-tile mt (single-buffered: the accumulators leave ~60 VGPRs, a double buffer spilled)
            int rowv[NMT], slotv[NMT], posv[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int m = em0 + mt * 16, mc = m < M ? m : M - 1;
                rowv[mt] = R.tok_row[mc]; slotv[mt] = R.tok_j[mc];
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int pos = R.pos_off[rowv[mt]] + slotv[mt];
                posv[mt] = pos < R.max_pos ? pos : R.max_pos - 1;
            }
            int secv[4], headv[4], tv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int c0 = ecol0 + nt * 16;                                       // wave-uniform: the n-tile's first column
                secv[nt] = c0 < N ? c0 / HDm : 3;                               // 3: past the matrix (nothing stored)
                c0 = c0 < N ? c0 : 0;
                const int hc = c0 - (c0 / HDm) * HDm;
                headv[nt] = hc >> 7; tv[nt] = (hc & 127) >> 4;
            }
            f32x4 cs[4], sn[4];
            auto fetch = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const long o = (long)posv[mt] * 64 + 8 * tv[nt] + 4 * (g & 1);
                    cs[nt] = *(const f32x4*)(R.cos_t + o); sn[nt] = *(const f32x4*)(R.sin_t + o);
                }
            };
            fetch(0);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int m = em0 + mt * 16, row = rowv[mt], slot = slotv[mt];
                const bool ok = m < M && slot < R.slots;
                uint2 q[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 v = acc[mt][nt];
                    float o[4] = {v[0], v[1], v[2], v[3]};
                    if (secv[nt] < 2) {
                        f32x4 u;
#pragma unroll
                        for (int j = 0; j < 4; ++j) u[j] = __shfl_xor(v[j], 32, 64);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = g < 2 ? rope_lo(v[j], u[j], cs[nt][j], sn[nt][j]) : rope_hi(u[j], v[j], cs[nt][j], sn[nt][j]);   // lanes g >= 2 hold x[j+64] (v), receive x[j] (u)
                    }
                    q[nt].x = pack_bf16x2(o[0], o[1]); q[nt].y = pack_bf16x2(o[2], o[3]);
                }
                if (mt + 1 < NMT) fetch(mt + 1);                                  // in flight under this m-tile's stores
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int sec = secv[nt], head = headv[nt], t = tv[nt];
                    if (sec > 2 || !ok) continue;
                    bf16* dst;
                    if (sec == 0) dst = (bf16*)R.qbuf + (long)m * HDm + head * 128 + (g < 2 ? 0 : 64) + 8 * t + 4 * (g & 1);
                    else if (sec == 1) dst = (bf16*)R.kc + (((long)row * R.nh + head) * R.slots + slot) * 128 + (g < 2 ? 0 : 64) + 8 * t + 4 * (g & 1);
                    else dst = (bf16*)R.vc + (((long)row * R.nh + head) * R.slots + slot) * 128 + 16 * t + 4 * g;
                    *(uint2*)dst = q[nt];
                }
            }
        } else if (ep.e.act == 2) {
            // SwiGLU epilogue: W rows are interleaved [8 gate | 8 up] per 16-column n-tile (the engine's gate|up
            // layout), so lanes g = 0,1 hold 4 gate columns and lanes g = 2,3 (lane ^ 32) the matching 4 up columns
            // of the same row: one cross-half exchange, then h = silu(gate) * up goes out as 4 bf16 (8 bytes)
            // at column n/2 -- the fp32 gate|up tensor is never written.  out: bf16 [M, N/2], ldc = N/2.
            bf16* hout = (bf16*)ep.e.out + coff;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 v = acc[mt][nt];
                    f32x4 u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = __shfl_xor(v[j], 32, 64);
                    const int row = em0 + mt * 16, col = (ecol0 + nt * 16) / 2 + g * 4;
                    if (g < 2 && row < M && col < N / 2) {
                        float hh[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) hh[j] = (v[j] / (1.f + expf(-v[j]))) * u[j];
                        uint2 q; q.x = pack_bf16x2(hh[0], hh[1]); q.y = pack_bf16x2(hh[2], hh[3]);
                        *(uint2*)(hout + (long)row * ep.e.ldc + col) = q;
                    }
                }
        } else if (GN && NMT == 8 && vec) {
            // GroupNorm(32) statistics of the values being stored (round 6; conv_halo.hip does the same for the 384^2 / 192^2 levels): a wave owns two
            // 64-row chunks x 64 columns; a chunk never straddles an image (HW % 64 == 0, checked by gemm256_try), so every (image, chunk, group)
            // slot has exactly ONE writer -- fixed-order reduction (4 m-tiles in registers, 16 row lanes + the group's column lanes by shuffles), no
            // atomics; gn_finalize_kernel adds an image's chunks in double.  Replaces the separate gn_stats pass over the stored tensor.
            const int cpg = ep.e.gn_cpg, hw = ep.e.gn_hw, nsplit = hw >> 6;
            // image / chunk of the tile's first row: ONE wave-uniform division per tile (hw >= 256: a 256-row tile crosses at most one image boundary)
            const int tm0 = __builtin_amdgcn_readfirstlane(em0 - lr) - wr * WROWS;
            const int tb0 = tm0 / hw, trem = tm0 - tb0 * hw;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mq = 0; mq < 4; ++mq) {                          // one m-tile row (4 fragments) per batch, stored values returned IN PLACE: no second fragment array
                    const int mt = h * 4 + mq;
                    int rows[4], cols[4]; f32x4 av[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { rows[i] = em0 + mt * 16; cols[i] = en0 + i * 16; av[i] = acc[mt][i]; }
                    ep.template store4_batch<4>(coff, roff, rows, cols, av, true, av);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 v = av[i];
                        s1[i] += (v[0] + v[1]) + (v[2] + v[3]);
                        s2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                    __builtin_amdgcn_sched_barrier(0);                    // one batch at a time: hoisting the later batches' bias / residual loads up here is what spilled
                }
                const int row0 = em0 - lr + h * 64;                    // first row of the chunk
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { s1[nt] += __shfl_xor(s1[nt], o, 64); s2[nt] += __shfl_xor(s2[nt], o, 64); }
                    if (cpg >= 8) { s1[nt] += __shfl_xor(s1[nt], 16, 64); s2[nt] += __shfl_xor(s2[nt], 16, 64); }
                    if (cpg >= 16) { s1[nt] += __shfl_xor(s1[nt], 32, 64); s2[nt] += __shfl_xor(s2[nt], 32, 64); }
                    const int col = en0 + nt * 16;
                    const bool writer = lr == 0 && (cpg == 4 || (cpg == 8 && (g & 1) == 0) || (cpg == 16 && g == 0));
                    if (writer && row0 < M && col < N) {
                        int off = trem + wr * WROWS + h * 64, b = tb0;
                        if (off >= hw) { off -= hw; ++b; }
                        const int split = off >> 6;
                        float* o = ep.e.gn_part + (((long)b * nsplit + split) * 32 + col / cpg) * 2;
                        o[0] = s1[nt]; o[1] = s2[nt];
                    }
                }
            }
        } else {
            // two m-tile rows (8 fragments) per batch: their bias / residual loads go out together (gemm_common.h store4_batch)
#pragma unroll
            for (int mp = 0; mp < NMT / 2; ++mp) {
                int rows[8], cols[8]; f32x4 av[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { rows[i] = em0 + (mp * 2 + (i >> 2)) * 16; cols[i] = en0 + (i & 3) * 16; av[i] = acc[mp * 2 + (i >> 2)][i & 3]; }
                ep.template store4_batch<8>(coff, roff, rows, cols, av, vec);
            }
            if constexpr (NMT & 1) {                              // odd tile heights: the last m-tile row on its own
                int rows[4], cols[4]; f32x4 av[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { rows[i] = em0 + (NMT - 1) * 16; cols[i] = en0 + i * 16; av[i] = acc[NMT - 1][i]; }
                ep.template store4_batch<4>(coff, roff, rows, cols, av, vec);
            }
        }
    }
#undef G2_MFMA_SECTION
#undef G2_MFMA_SECTION2
#undef G2_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // clamped tail stages must not outlive the block's LDS
}


template <class AL, int WM, int WN, bool ROPE = false, int MT1 = 4, int PH = 4, bool GN = false>
static void launch256(hipStream_t s, AL al, const bf16* W, long ldb, long strideA, long strideB, long strideA2, long strideB2,
                      const Epi<bf16>& ep, int M, int N, int K, int batch, int batch2) {
    constexpr int BM = WM * (64 + 16 * MT1), BN = WN * 64, LDS = 2 * (2 * WM * 64 * 128 + 2 * WN * 32 * 128);
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    const int ncu = pg_cu_count();
    const int per_batch = ncu / (batch * batch2) > 8 ? ncu / (batch * batch2) : 8;   // blocks per (batch) slice: one per CU overall
    dim3 grid(ntm * ntn < per_batch ? ntm * ntn : per_batch, batch, batch2), block(512);
    auto kfn = gemm256_kernel<AL, Epi<bf16>, WM, WN, ROPE, MT1, PH, GN>;
    (void)PG_DYN_LDS(kfn, LDS);
    hipLaunchKernelGGL(kfn, grid, block, LDS, s, al, W, ldb, strideA, strideB, strideA2, strideB2, ep, M, N, K, ntm, ntn);
}

// Tile height of a plain-A launch (round 5): rounds of one resident block per CU x rows per tile (+ a fixed per-tile share for the epilogue
// and the pipeline fill, in row equivalents), smallest wins; ties keep the taller tile.  Returns MT1 (4 / 3 / 2 = 256 / 224 / 192 rows).
// Option encoding (the ONE place it is documented for the code; include/plangen_hip.h for callers; pg_set_option rejects anything else):
//   gemm256 = 0 off | 1 auto tile height | 4 / 5 / 6 pin 256 / 224 / 192 rows;  + 8 = four phases per K tile instead of two.
static int pick_tile_height(int M, int N, int batches) {
    const int pin = pg_tune->gemm256 & 7;
    if (pin >= 4) return pin == 4 ? 4 : pin == 5 ? 3 : 2;      // A/B: 4 / 5 / 6 pin 256 / 224 / 192 rows
    if (batches != 1) return 4;                                 // batched launches: launch256 gives each slice its share of the CUs, tall tiles
    const int ntn = (N + 255) / 256;
    const int ncu = pg_cu_count();
    int best = 4; long best_cost = -1;
    for (int mt1 = 4; mt1 >= 2; --mt1) {
        const int bm = 2 * (64 + 16 * mt1);
        const long tiles = (long)((M + bm - 1) / bm) * ntn;
        const long cost = ((tiles + ncu - 1) / ncu) * (bm + 16);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt1; }
    }
    return best;
}

// Takes the shapes the big tiles fill well; everything else stays on the 128^2 kernel.
bool gemm256_try(hipStream_t s, const GemmA& a, const bf16* W, long ldb, long strideB, const GemmEpi& e, int M, int N, int K,
                 int batch, int batch2, long strideB2) {
    if (!pg_tune->gemm256 || K % G2_BK || K < 2 * G2_BK) return false;
    // (a 4x2-wave 512x128 instantiation for the Cout = 128 convolutions was measured: 470-560 TFLOP/s against
    // 640-690 for the 128x128 kernel -- K = 9*Cin is only 18-36 K tiles, so the fill / drain of one persistent
    // block per CU and the 8-slot im2col address state outweigh the deeper pipeline; not instantiated)
    const int BM = 256, BN = 256;
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    // padding waste of the last tile row / column, and enough tiles to fill the chip
    if ((long)ntm * BM * ntn * BN > (long)M * N * 5 / 4) return false;
    if ((long)ntm * ntn * batch * batch2 < 200) return false;
    // GroupNorm partials of the output from the epilogue (round 6): N = 32 groups of 4 / 8 / 16 channels, images of a multiple of 64 pixels (a wave's
    // 64-row chunk then never straddles two images), dense [M][N] output, 256-row tiles (the chunk arithmetic assumes them)
    GemmEpi eg = e;
    const long out_hw = a.kind == 0 ? a.gn_hw : (a.kind == 2 ? (long)(a.Hi / 2) * (a.Wi / 2) : (long)(a.Hi << a.up) * (a.Wi << a.up));
    const bool want_gn = pg_tune->gn_epilogue256 && a.gn_part && a.gn_nsplit && batch == 1 && batch2 == 1 && e.act == 0 && (N == 128 || N == 256 || N == 512) && e.ldc == N
                         && out_hw >= 256 && (out_hw % 64) == 0 && (M % out_hw) == 0 && out_hw / 64 <= 1024 && !e.strideC
                         && (((e.ldr ? e.ldr : e.ldc) | e.ldc) & 3) == 0;          // the kernel's vec_ok() for coff = roff = 0: the partials come out of the vector epilogue only
    if (want_gn) { eg.gn_part = (float*)a.gn_part; eg.gn_hw = (int)out_hw; eg.gn_cpg = N / 32; *a.gn_nsplit = (int)(out_hw / 64); }
    Epi<bf16> ep{eg, M, N};
    if (a.kind == 0) {
        PlainLoaderB<bf16> al; al.A = (const bf16*)a.ptr; al.lda = a.lda; al.M = M;
        const int mt1 = want_gn ? 4 : pick_tile_height(M, N, batch * batch2);
#define G2_LAUNCH(ROPE_, MT1_, PH_) launch256<PlainLoaderB<bf16>, 2, 4, ROPE_, MT1_, PH_>(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2)
        if (want_gn) {        // 1x1 convolution whose output feeds a GroupNorm (AttnBlock.proj_out): the GN instantiation, 256-row tiles, two phases
            launch256<PlainLoaderB<bf16>, 2, 4, false, 4, 2, true>(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2);
            return true;
        }
        // two phases per K tile by default (round 5: +2-3 % on every prefill shape, bit-identical); gemm256 bit 3 (value 8) selects the four-phase schedule for A/B
        const bool four = (pg_tune->gemm256 & 8) != 0;
        if (!four) {
            if (e.act == 3) { if (mt1 == 3) G2_LAUNCH(true, 3, 2); else if (mt1 == 2) G2_LAUNCH(true, 2, 2); else G2_LAUNCH(true, 4, 2); }
            else { if (mt1 == 3) G2_LAUNCH(false, 3, 2); else if (mt1 == 2) G2_LAUNCH(false, 2, 2); else G2_LAUNCH(false, 4, 2); }
        } else
        if (e.act == 3) { if (mt1 == 3) G2_LAUNCH(true, 3, 4); else if (mt1 == 2) G2_LAUNCH(true, 2, 4); else G2_LAUNCH(true, 4, 4); }
        else { if (mt1 == 3) G2_LAUNCH(false, 3, 4); else if (mt1 == 2) G2_LAUNCH(false, 2, 4); else G2_LAUNCH(false, 4, 4); }
#undef G2_LAUNCH
        return true;
    }
    if (e.act == 3) return false;
    const int Ho = a.kind == 2 ? a.Hi / 2 : (a.Hi << a.up), Wo = a.kind == 2 ? a.Wi / 2 : (a.Wi << a.up);
    const long in_elems = ((long)M / ((long)Ho * Wo)) * a.Hi * a.Wi * a.Cin;
    if (in_elems >= (1L << 31)) return false;             // the slim loader keeps 32-bit element offsets
    ConvLoaderS<4> al; al.setup(a, M);
    if (want_gn) launch256<ConvLoaderS<4>, 2, 4, false, 4, 2, true>(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2);      // GroupNorm partials from the epilogue (two-phase form only)
    else if (pg_tune->gemm256 & 8) launch256<ConvLoaderS<4>, 2, 4, false, 4, 4>(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2);
    else launch256<ConvLoaderS<4>, 2, 4, false, 4, 2>(s, al, W, ldb, a.strideA, strideB, a.strideA2, strideB2, ep, M, N, K, batch, batch2);
    return true;
}
