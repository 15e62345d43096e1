// 3x3 / pad 1 / stride 1 convolution, Cin = Cout = 128, NHWC bf16 in, fp32 or bf16 out, for gfx950:
// the VQ-16 decoder's 384^2 and 192^2 ResBlock convolutions (11 of its 13 Cout=128 convolutions, half of the
// decoder's time on the im2col-per-K-tile GEMM kernel).
//
// Direct convolution with an LDS-resident INPUT HALO TILE: a block owns 8 x 32 output pixels x all 128 output
// channels.  The (8+2) x (32+2) x 128-channel input patch (85 KiB, 90 KiB as laid out) is brought into LDS ONCE with
// global_load_lds (out-of-image pixels come from a zero page) and serves all 9 taps x 128 channels = 18 K
// tiles of 64: the A operand is never re-fetched per tap, only the weights stream (16 KiB per K tile, a
// 4-slot LDS ring, counted vmcnt, the two wave groups one barrier apart).
//
//   waves: 8 = 4 (pixel rows pairs) x 2 (64 output channels); a wave owns 2 rows x 32 px = 4 MFMA m-tiles of
//          16 consecutive pixels, and 4 n-tiles: acc 4x4 f32x4, 32 MFMAs (16x16x32 bf16) per K tile.
//   halo LDS layout (round 6): [patch row][36 pixels][16 chunks of 16 B] -- the 34 halo pixels of a row padded to nine 4-pixel LDS-DMA instructions, so an
//          instruction's row and segment are wave-uniform -- with the chunk position XORed with 2 * (pixel-in-row & 7) through the SOURCE address of the
//          LDS-DMA (the DMA writes lane-linear).  A ds_read_b128 lane group is 8 pixels of one k-group and 8 of the next: with this map the 16 lanes
//          touch 16 distinct 16-byte slots for EVERY first-pixel alignment (XOR (pixel & 15), rounds 2-5, collided 2-way on every dx = 1 tap).
//          The four k-steps of a tap are address ^ {0, 64, 128, 192}.  90 KiB.
//   weights: [Cout][tap][Cin] (K = tap*128 + ci, the engine's conv layout); LDS rows of 128 B with the
//          (row>>1)&7 XOR swizzle of the GEMM kernels.
//   operands swapped in the MFMA (D = W . A^T): a lane holds 4 consecutive output channels of one pixel ->
//          16-byte epilogue accesses (bias, fp32 residual, store) through the shared Epi.
//   persistent: one block per CU walks tiles in image-row order (neighbouring tiles share halo rows in L2).
//   UP: the nearest-2x upsample in front of the convolution (vq_model.py:417-427) folded into the addressing:
//          H x W is the OUTPUT size, the input is H/2 x W/2, the halo patch is the 6 x 18 SOURCE pixels an
//          8 x 32 output tile touches, and output pixel (py, px), tap (dy, dx) reads source halo pixel
//          (((py+dy-1)>>1)+1, ((px+dx-1)>>1)+1); zero padding of the upsampled image == zero padding of the source.
//
// Round 6, per-tile stamps (s_memrealtime, block 7, 64 x 384^2, profiles/r06_c_vq_decode.md): a tile was 20.2 us = top wait 1.5 + 18 K tiles
// 12.4 + epilogue 6.0-7.2, against 7.7 us of MFMA at full rate.  What was kept: the epilogue without a wait behind its first store (below), the padded patch rows
// (scalar row / segment decode), the conflict-free chunk map, W(t+2) staged first in its phase: 3x3 kernels -10 %, VQ decode -2.0 ms on one box.  What was built,
// measured and removed (commit 2af565a has it): the patch as two CHANNEL halves filled ping-pong under the other half's nine K tiles, wave roles split (weights /
// halo) -- top wait + epilogue 7.5 -> 2.3 us but the K loop 12.4 -> 15.7 us, the decode 1.4 ms slower; prefetch distance 3 on the weight ring (neutral); half the
// weight stream (K loop 12.47 -> 12.14 us: the loop is not ingest-bound); stores allowed to stay in flight across the tile boundary (neutral).  A phase is
// [16 fragment reads: 0.2 us of LDS time] beside [the other wave group's 32 MFMAs: 0.25-0.3 us], strictly alternating: ~10 us per tile is the schedule's own bound.
#include "gemm_common.h"

#define CH_TH 8
#define CH_TW 32
#define CH_HW (CH_TW + 2)                       // halo width 34
#define CH_HP ((CH_TH + 2) * CH_HW)             // 340 halo pixels
#define CH_HALO_BYTES (88 * 1024)               // 88 wave-instructions x 1 KiB (>= 340 x 256 B)
#define CH_WSLOT 16384                          // one K tile of weights: 128 rows x 128 B
#define CH_HWP 36                               // round 6: halo rows padded to 9 LDS-DMA instructions of 4 pixels (the 3x3 kernel's own patch: 10 x 36 pixels = 90 KiB)
#define CH_PATCH_BYTES (90 * 1024)
#define CH_RED (CH_PATCH_BYTES + 4 * CH_WSLOT)      // 1 KiB: per-wave GroupNorm partials of the epilogue
#define CH_BIAS (CH_RED + 1024)                 // 512 B: the 128 per-channel biases (fast epilogue), staged once per block
#define CH_LDS (CH_BIAS + 512)
#define CH_NKT 18                               // 9 taps x (128 / 64)

// Epilogue of one 8 x 32 tile, shared by the two halo kernels: a wave's 4 x 4 f32x4 accumulators (m-tile mt = pixel row pair / 16-pixel half, n-tile nt = 16
// output channels) through the shared Epi in two batches of 8 fragments; with gn_part the GroupNorm(32 groups of 4 channels) partial sums of the STORED values.
template <class EP>
__device__ __forceinline__ void halo_epilogue(const EP& ep, f32x4 (&acc)[4][4], long mrow, int Wd, int tile_id, bool vec, float* __restrict__ gn_part,
                                              float* red, int tid, int w, int wc, int g, int lr) {
    if (gn_part && vec) {
        // GroupNorm(32 groups of 4 channels) statistics of the values being stored: a lane's f32x4 is one
        // group of one pixel.  Fixed-order reduction: 4 m-tiles in registers, 16 pixel lanes by shuffles,
        // the 4 pixel-row waves through LDS -> one (sum, sum of squares) per (tile, group); the finalize
        // kernel adds the tiles of an image in double.  Replaces the separate statistics pass.
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {                                // 8 fragments per batch: all bias / residual loads up front
            int rows[8], cols[8]; f32x4 av[8], vo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mt = mp * 2 + (i >> 2), nt = i & 3;
                rows[i] = (int)(mrow + (long)(mt >> 1) * Wd + (mt & 1) * 16); cols[i] = wc * 64 + nt * 16 + g * 4; av[i] = acc[mt][nt];
            }
            ep.template store4_batch<8>(0, 0, rows, cols, av, true, vo);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int nt = i & 3;
                const f32x4 v = vo[i];
                s1[nt] += (v[0] + v[1]) + (v[2] + v[3]);
                s2[nt] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1[nt] += __shfl_xor(s1[nt], o, 64); s2[nt] += __shfl_xor(s2[nt], o, 64); }
        if (lr == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { red[(w * 16 + nt * 4 + g) * 2] = s1[nt]; red[(w * 16 + nt * 4 + g) * 2 + 1] = s2[nt]; }
        }
        __syncthreads();
        if (tid < 32) {
            const int gwc = tid >> 4, gi = tid & 15;                    // group = gwc*16 + gi
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) { a += red[((r4 * 2 + gwc) * 16 + gi) * 2]; q += red[((r4 * 2 + gwc) * 16 + gi) * 2 + 1]; }
            gn_part[((long)tile_id * 32 + tid) * 2] = a; gn_part[((long)tile_id * 32 + tid) * 2 + 1] = q;
        }
        __syncthreads();                                            // red is rewritten by the next tile
    } else {
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            int rows[8], cols[8]; f32x4 av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mt = mp * 2 + (i >> 2), nt = i & 3;
                rows[i] = (int)(mrow + (long)(mt >> 1) * Wd + (mt & 1) * 16); cols[i] = wc * 64 + nt * 16 + g * 4; av[i] = acc[mt][nt];
            }
            ep.template store4_batch<8>(0, 0, rows, cols, av, vec);
        }
    }
}


// Round 6: the epilogue without a wait behind its first store.  One s_waitcnt vmcnt(N) covers loads AND stores of a wave in issue order, so the generic epilogue above
// -- two batches of [bias loads, residual loads, wait, 8 stores] -- waited, in its second batch, for the write acknowledgements of the first (and in its first, for the
// next tile's halo fill issued in front of it): 6-10 us of a 20-24 us tile with the MFMA pipe idle (stamps: profiles/r06_c_vq_decode.md).  Here the bias (4 x f32x4 per
// lane, the same for every tile) is loaded once per kernel, all 16 residual loads of a tile are issued first (halo_epi_loads, BEFORE the next tile's fill goes out) and
// the 16 stores follow with nothing to wait for.  Same arithmetic and order as Epi::store4 (scale, bias, residual), same GroupNorm partial sums.
// Taken when the layout is the vector one and there is no bias_m / activation (every convolution of the decoder); halo_epilogue otherwise.

// Addressing: one wave-uniform 64-bit base per tile (its first pixel) + a 32-bit per-lane byte offset per m-tile (a tile spans 8 image rows: < 2^31 bytes), so the 16
// accesses of a tile need 4 offset registers, not 16 64-bit pointers.  trow0 = index of the tile's first pixel, lrow = (wr*2)*Wd + lr.
template <int RES, class EP>
__device__ __forceinline__ void halo_epi_loads(const EP& ep, long trow0, int lrow, int Wd, int wc, int g, f32x4 (&r)[4][4]) {
    const auto& e = ep.e;
    if constexpr (RES == 0) return;
    const int ldr = (int)(e.ldr ? e.ldr : e.ldc);
    constexpr int esz = RES == 1 ? 4 : 2;
    const char* const base = (const char*)e.residual + trow0 * ldr * esz;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const unsigned off = (unsigned)(((lrow + (mt >> 1) * Wd + (mt & 1) * 16) * ldr + wc * 64 + g * 4) * esz);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if constexpr (RES == 1) r[mt][nt] = *(const f32x4*)(base + off + nt * 64);
            else {
                const uint2 r2 = *(const uint2*)(base + off + nt * 32);
                r[mt][nt] = (f32x4){__uint_as_float(r2.x), __uint_as_float(r2.y), 0.f, 0.f};        // bf16 pairs, decoded at the use
            }
        }
    }
}

template <bool GN, int RES, bool OF32, class EP>
__device__ __forceinline__ void halo_epi_stores_impl(const EP& ep, f32x4 (&acc)[4][4], const f32x4 (&bias)[4], f32x4 (&r)[4][4], long trow0, int lrow, int Wd, int tile_id,
                                                     float* __restrict__ gn_part, float* red, int tid, int w, int wc, int g, int lr) {
    const auto& e = ep.e;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int ldc = (int)e.ldc;
    constexpr int esz = OF32 ? 4 : 2;
    char* const base = (char*)e.out + trow0 * ldc * esz;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const unsigned off = (unsigned)(((lrow + (mt >> 1) * Wd + (mt & 1) * 16) * ldc + wc * 64 + g * 4) * esz);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[mt][nt];
            v *= e.scale;
            v += bias[nt];
            if constexpr (RES == 1) v += r[mt][nt];
            else if constexpr (RES == 2) {
                const unsigned a = __float_as_uint(r[mt][nt][0]), b2 = __float_as_uint(r[mt][nt][1]);
                v += (f32x4){bf16_lo(a), bf16_hi(a), bf16_lo(b2), bf16_hi(b2)};
            }
            if constexpr (OF32) *(f32x4*)(base + off + nt * 64) = v;
            else { uint2 q; q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); *(uint2*)(base + off + nt * 32) = q; }
            if constexpr (GN) {
                s1[nt] += (v[0] + v[1]) + (v[2] + v[3]);
                s2[nt] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
    }
    if constexpr (GN) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1[nt] += __shfl_xor(s1[nt], o, 64); s2[nt] += __shfl_xor(s2[nt], o, 64); }
        if (lr == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { red[(w * 16 + nt * 4 + g) * 2] = s1[nt]; red[(w * 16 + nt * 4 + g) * 2 + 1] = s2[nt]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // LDS only: __syncthreads() would also wait for this tile's stores and the next tile's fill (vmcnt(0))
        if (tid < 32) {
            const int gwc = tid >> 4, gi = tid & 15;
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) { a += red[((r4 * 2 + gwc) * 16 + gi) * 2]; q += red[((r4 * 2 + gwc) * 16 + gi) * 2 + 1]; }
            gn_part[((long)tile_id * 32 + tid) * 2] = a; gn_part[((long)tile_id * 32 + tid) * 2 + 1] = q;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // red is rewritten by the next tile
    }
}
template <int RES, bool OF32, class EP>
__device__ __forceinline__ void halo_epi_stores(const EP& ep, f32x4 (&acc)[4][4], const f32x4 (&bias)[4], f32x4 (&r)[4][4], long trow0, int lrow, int Wd, int tile_id,
                                                float* __restrict__ gn_part, float* red, int tid, int w, int wc, int g, int lr) {
    if (gn_part) halo_epi_stores_impl<true, RES, OF32>(ep, acc, bias, r, trow0, lrow, Wd, tile_id, gn_part, red, tid, w, wc, g, lr);
    else halo_epi_stores_impl<false, RES, OF32>(ep, acc, bias, r, trow0, lrow, Wd, tile_id, gn_part, red, tid, w, wc, g, lr);
}
// the 128 biases to LDS once per block (thread c < 128 stores bias[c]); a tile's epilogue reads its 4 x f32x4 back (4 ds_read_b128) instead of holding 16 registers
// through the MFMA loop.  Visible to every wave after the first block barrier of the kernel.
template <class EP>
__device__ __forceinline__ void halo_bias_stage(const EP& ep, bool fast, int tid, float* lds_bias) {
    if (fast && tid < 128) lds_bias[tid] = ep.e.bias_n[tid];
}
__device__ __forceinline__ void halo_bias_read(const float* lds_bias, int wc, int g, f32x4 (&bias)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = *(const f32x4*)(lds_bias + wc * 64 + nt * 16 + g * 4);
}

template <class EP, bool STAG, bool UP, int EPK = -1>      // EPK: -1 generic epilogue; else residual kind (0 none, 1 fp32, 2 bf16) * 2 + (fp32 output)
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wt,
                                                          const bf16* __restrict__ zeros, EP ep, int B, int H, int Wd,
                                                          float* __restrict__ gn_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wlds = smem + CH_PATCH_BYTES;
    float* const red = (float*)(smem + CH_RED);
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1, g = l >> 4, lr = l & 15;
    const int tiles_x = Wd / CH_TW, tiles_y = H / CH_TH, tiles_img = tiles_x * tiles_y;
    const int NT = tiles_img * B, G = gridDim.x;
    const bool vec = ep.vec_ok(0, 0);
    constexpr bool fast = EPK >= 0;                                 // host-checked: vector layout, bias_n, no bias_m / activation (conv_halo_try)
    constexpr int RES = EPK >= 0 ? EPK / 2 : 0; constexpr bool OF32 = EPK >= 0 && (EPK & 1);
    float* const lds_bias = (float*)(smem + CH_BIAS);
    halo_bias_stage(ep, fast, tid, lds_bias);

    // weight staging: wave w, instruction i covers LDS rows (i*8 + w)*8 .. +7 (row = output channel)
    const int wsrow = w * 8 + (l >> 3);
    const int wsc = ((l & 7) ^ (((w & 1) << 2) + (l >> 4))) * 8;               // swizzled source chunk (elements)
    const bf16* wsrc0 = Wt + (long)wsrow * (9 * 128) + wsc;
    const bf16* wsrc1 = wsrc0 + (long)64 * (9 * 128);
    auto stage_w = [&](int t) __attribute__((always_inline)) {
        const int tt = t < CH_NKT ? t : CH_NKT - 1;                            // clamped tail: keeps the counts exact
        char* d = wlds + (t & 3) * CH_WSLOT + w * 1024;
        glds16(wsrc0 + tt * 64, d);
        glds16(wsrc1 + tt * 64, d + 8192);
    };
    // B (weight) fragment addresses inside a slot
    const int swz = (lr >> 1) & 7;
    const int boff0 = (wc * 64 + lr) * 128 + ((g ^ swz) << 4), boff1 = boff0 ^ 64;

    f32x4 acc[4][4];
    bf16x8 af[4][2], bfr[4][2];

    int b = 0, y0 = 0, x0 = 0;
    // halo fill (12 LDS-DMA instructions per wave, 4 halo pixels x 16 chunks each) + the first weight tiles
    auto fill = [&](int tix) __attribute__((always_inline)) {
        b = tix / tiles_img;
        const int r = tix - b * tiles_img;
        y0 = (r / tiles_x) * CH_TH; x0 = (r % tiles_x) * CH_TW;
        // Round 6: a halo row is padded to a whole number of 4-pixel LDS-DMA instructions (36 / 20 pixels for 34 / 18), so an instruction's row and segment
        // are wave-uniform (scalar unit) and a lane only adds its pixel-in-segment: ~8 vector instructions per DMA instead of ~35 (division by 34, 64-bit
        // multiply-adds, an exec-masked branch).
        constexpr int HW = UP ? CH_TW / 2 + 2 : CH_HW, SPR = UP ? 5 : 9, NROW = UP ? CH_TH / 2 + 2 : CH_TH + 2, NQ = NROW * SPR, NJ = (NQ + 7) / 8;
        const int Hs = UP ? H / 2 : H, Ws = UP ? Wd / 2 : Wd;                    // source image
        const int sy0 = (UP ? y0 / 2 : y0) - 1, sx0 = (UP ? x0 / 2 : x0) - 1;      // source coords of halo pixel (0, 0)
        const bf16* img = X + (long)b * Hs * Ws * 128;
        int ll = l;
        asm volatile("" : "+v"(ll));                                           // the per-lane parts are recomputed per tile (a handful of instructions), not held in registers through the MFMA loop
        const int lp = ll >> 4, pos = ll & 15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int q = j * 8 + w;                                                 // wave-instruction index, 1 KiB = 4 halo pixels x 16 chunks
            q = q < NQ ? q : NQ - 1;                                           // surplus instructions repeat the last one (same bytes to the same place)
            const int hy = q / SPR, seg = q - hy * SPR;                        // scalar
            const int y = sy0 + hy;
            const bool yok = y >= 0 && y < Hs;
            const int hx = seg * 4 + lp;
            const bool ok = yok && hx < HW && (unsigned)(sx0 + hx) < (unsigned)Ws;
            const unsigned sch16 = (unsigned)((pos ^ ((hx & 7) << 1)) << 4);      // chunk position XOR 2*(hx & 7): see the fragment read
            const char* const rowp = (const char*)(img + ((long)y * Ws + sx0) * 128);      // scalar; not dereferenced when the row is outside the image
            const char* const base = ok ? rowp : (const char*)zeros;
            const unsigned off = ok ? (unsigned)(hx << 8) + sch16 : sch16;
            glds16(base + off, halo + q * 1024);
        }
        stage_w(0); stage_w(1);
        if (!STAG) stage_w(2);
    };
    int tix = blockIdx.x;
    if (tix < NT) fill(tix);
    while (tix < NT) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if constexpr (STAG) {
            // staggered wave groups (as in gemm256.hip): tile 0 + halo landed (W1 may still fly), then the
            // second group runs one barrier behind, so one wave per SIMD reads LDS while the other runs MFMAs.
            // Ring discipline under the stagger: W(t+2) is staged in phase t into the slot read in phase t-2;
            // W(t+1) is retired by vmcnt(2) in phase t and read in phase t+1.
            // Round 3: the halo and W(0) are retired here and read by the first wave group right behind the barrier -- the same-phase
            // form the staging rule forbids (tools/dma_isa_check.py) -- so a second barrier separates retirement from the first read
            // (once per 8x32 tile; both groups pass it, the barrier counts stay equal).
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (wr >= 2) asm volatile("s_barrier" ::: "memory");
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            int abase[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int hrow, hx;                                               // patch row and pixel-in-row; LDS pixel slot = row * (padded width) + hx, chunk XOR (hx & 15)
                if constexpr (UP) { hrow = ((wr * 2 + (mt >> 1) + dy - 1) >> 1) + 1; hx = (((mt & 1) * 16 + lr + dx - 1) >> 1) + 1; }
                else { hrow = wr * 2 + (mt >> 1) + dy; hx = (mt & 1) * 16 + dx + lr; }
                // chunk position XOR 2*(hx & 7): a ds_read_b128 lane group is 8 pixels of one k-group + 8 of the next (MI355X_MICROARCH.md LDS table), and XOR (hx & 15)
                // collided 2-way whenever the first pixel was odd (every dx = 1 tap: a third of the A reads at twice the LDS cycles); this map is conflict-free for
                // every pixel alignment and k-step (exhaustive check over the four lane groups).
                abase[mt] = (hrow * (UP ? 20 : CH_HWP) + hx) * 256 + ((((hx & 7) << 1) ^ g) << 4);
            }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const int t = tap * 2 + kh;
                if constexpr (!STAG) {
                    // W tile t (and, for t == 0, the halo) has landed: the loads issued after it are tiles t+1, t+2
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");                    // round 3: retirement and first read one barrier apart (staging rule, strict form)
                    stage_w(t + 3);                                            // slot of tile t-1: every wave is past its reads
                }
                if constexpr (STAG) stage_w(t + 2);                           // round 6: first thing in the phase (it was behind the 16 fragment reads): W(t+2) gets ~0.2 us more to land before phase t+1 waits for it
                const char* ws = wlds + (t & 3) * CH_WSLOT;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bfr[nt][0] = *(const bf16x8*)(ws + boff0 + nt * 2048);
                    bfr[nt][1] = *(const bf16x8*)(ws + boff1 + nt * 2048);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    af[mt][0] = *(const bf16x8*)(halo + (abase[mt] ^ ((kh * 2 + 0) << 6)));
                    af[mt][1] = *(const bf16x8*)(halo + (abase[mt] ^ ((kh * 2 + 1) << 6)));
                }
                if constexpr (STAG) {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");            // W(t+1) landed
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STAG) asm volatile("s_barrier" ::: "memory");
            }
        }
        if (STAG && wr < 2) asm volatile("s_barrier" ::: "memory");
        // every wave must be done with the halo before the next tile's fill overwrites it; the fill (and the
        // first weight tiles) then go out BEFORE this tile's stores so they fly during the epilogue
        if (!STAG) asm volatile("s_barrier" ::: "memory");
        const long mrow = ((long)b * H + y0 + wr * 2) * Wd + x0 + lr;
        const int tile_id = tix;                                        // (image, tile) index of THIS tile
        tix += G;
        if constexpr (fast) {
            const long trow0 = ((long)b * H + y0) * Wd + x0;            // first pixel of the tile: wave-uniform (b, y0, x0 come from the tile index)
            int lrow = wr * 2 * Wd + lr;
            asm volatile("" : "+v"(lrow));                            // the 8 per-lane byte offsets are recomputed per tile instead of living (as 64-bit values) through the MFMA loop
            f32x4 bias[4], rres[4][4];
            // the next tile's fill (and first weight tiles) go out first and fly under the residual loads and the stores; with a residual its wait covers the fill
            // too (one counter, issue order), which costs little: both come from HBM at the same time.  Nothing is waited for behind the first store.
            if (tix < NT) fill(tix);
            __builtin_amdgcn_sched_barrier(0);                          // the fill's address registers die here, before 64 residual registers come alive
            halo_epi_loads<RES>(ep, trow0, lrow, Wd, wc, g, rres);
            if constexpr (RES != 0) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0) the compiler can see on EVERY path: otherwise it guards the MFMA loop's first register write with its own (and waits there for the stores)
            halo_bias_read(lds_bias, wc, g, bias);
            halo_epi_stores<RES, OF32>(ep, acc, bias, rres, trow0, lrow, Wd, tile_id, gn_part, red, tid, w, wc, g, lr);
        } else {
            if (tix < NT) fill(tix);
            halo_epilogue(ep, acc, mrow, Wd, tile_id, vec, gn_part, red, tid, w, wc, g, lr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// Takes 3x3 / pad 1 / stride 1 convolutions with Cin = Cout = 128 on images whose sides are multiples of the
// 8 x 32 tile; returns false otherwise (the implicit-GEMM kernels handle the rest).
bool conv_halo_try(hipStream_t s, const GemmA& a, const bf16* W, const GemmEpi& e, int M, int N, int K, float* gn_part, int* gn_nsplit) {
    if (!pg_tune->conv_halo || a.kind != 1 || a.up > 1 || a.Cin != 128 || N != 128 || K != 9 * 128) return false;
    const int H = a.Hi << a.up, Wd = a.Wi << a.up;                 // output size
    if (H % CH_TH || Wd % CH_TW || a.strideA || e.strideC) return false;
    const long px = (long)H * Wd;
    if (M % px) return false;
    const int B = (int)(M / px);
    const int tiles = B * (H / CH_TH) * (Wd / CH_TW);
    if (tiles < 128) return false;
    Epi<bf16> ep{e, M, N};
    const dim3 grid(tiles < 256 ? tiles : 256), block(512);
    if (gn_part && (e.ldc & 3) != 0) gn_part = nullptr;
    if (gn_nsplit) *gn_nsplit = gn_part ? (H / CH_TH) * (Wd / CH_TW) : 0;
    // fast epilogue (round 6): vector layout, per-channel bias, no per-row bias / activation, and one of the residual / output combinations instantiated below;
    // anything else, and the lock-step variant (conv_halo = 2), keeps the generic one (same values: tests/test_gpu_full.py compares whole decodes)
    int epk = -1;
    {
        const long ldr = e.ldr ? e.ldr : e.ldc;
        const long ldmax = e.ldc > ldr ? e.ldc : ldr;
        const bool vec_host = ((e.ldc | ldr) & 3) == 0 && (N & 3) == 0 && (long)(CH_TH * Wd + 2 * CH_TW) * ldmax * 4 < (1L << 31);      // the fast epilogue's 32-bit byte offsets inside a tile
        const int res = !e.residual ? 0 : e.res_f32 ? 1 : 2;
        if (vec_host && e.bias_n && !e.bias_m && e.act == 0) {
            const int k = res * 2 + (e.out_f32 ? 1 : 0);
            if (k == 0 || k == 1 || k == 3 || k == 4) epk = k;          // none->bf16, none->fp32, fp32->fp32 (the decoder's three), bf16->bf16
            if (a.up && res) epk = -1;
        }
    }
#define CH_GO(...)                                                                                                            \
    {                                                                                                                         \
        auto kfn = __VA_ARGS__;                                                                                               \
        (void)PG_DYN_LDS(kfn, CH_LDS);                                                                                        \
        hipLaunchKernelGGL(kfn, grid, block, CH_LDS, s, (const bf16*)a.ptr, W, (const bf16*)a.zeros, ep, B, H, Wd, gn_part);  \
    }
#define CH_EPK(UP)                                                                                                            \
    switch (epk) {                                                                                                            \
        case 0: CH_GO(conv3x3_halo_kernel<Epi<bf16>, true, UP, 0>) break;                                                      \
        case 1: CH_GO(conv3x3_halo_kernel<Epi<bf16>, true, UP, 1>) break;                                                      \
        case 3: if constexpr (!UP) { CH_GO(conv3x3_halo_kernel<Epi<bf16>, true, false, 3>) } break;                            \
        case 4: if constexpr (!UP) { CH_GO(conv3x3_halo_kernel<Epi<bf16>, true, false, 4>) } break;                            \
        default: CH_GO(conv3x3_halo_kernel<Epi<bf16>, true, UP, -1>) break;                                                    \
    }
    if (pg_tune->conv_halo == 2) { if (a.up) CH_GO(conv3x3_halo_kernel<Epi<bf16>, false, true>) else CH_GO(conv3x3_halo_kernel<Epi<bf16>, false, false>) }
    else { if (a.up) { CH_EPK(true) } else { CH_EPK(false) } }
#undef CH_EPK
#undef CH_GO
    return true;
}

// ------------------------------------------------------------------------------- conv_out (128 -> 3)
// Same halo tile, but Cout <= 4 is padded to ONE 16-wide MFMA n-tile whose weight fragments (36 k-steps x 4
// registers, zero for the padding rows) live in registers for the whole persistent block: no weight traffic,
// no barrier inside a tile.  Wave w owns output row w of the 8 x 32 tile (2 m-tiles).  NCHW output
// (vq_model.py:213-214: the decoder's image), fp32 or bf16.
__global__ __launch_bounds__(512) void conv3x3_out_halo_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wt,
                                                              const float* __restrict__ bias, const bf16* __restrict__ zeros,
                                                              void* __restrict__ out, int out_bf16, int B, int H, int Wd, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = l >> 4, lr = l & 15;
    const int tiles_x = Wd / CH_TW, tiles_y = H / CH_TH, tiles_img = tiles_x * tiles_y;
    const int NT = tiles_img * B, G = gridDim.x;
    // weight fragments: lane (lr = output channel, g) holds k = g*8 .. +7 of every 32-wide k-step
    bf16x8 wf[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        wf[q] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (lr < Cout) wf[q] = *(const bf16x8*)(Wt + (long)lr * (9 * 128) + q * 32 + g * 8);
    }
    float bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = (g * 4 + j < Cout) ? bias[g * 4 + j] : 0.f;
    const int hp0 = w * CH_HW + lr;
    for (int tix = blockIdx.x; tix < NT; tix += G) {
        const int b = tix / tiles_img, r = tix - b * tiles_img;
        const int y0 = (r / tiles_x) * CH_TH, x0 = (r % tiles_x) * CH_TW;
        const bf16* img = X + (long)b * H * Wd * 128;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int q = j * 8 + w;
            int hp = q * 4 + (l >> 4);
            const int pos = l & 15;
            const bool inr = hp < CH_HP;
            hp = inr ? hp : CH_HP - 1;
            const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = inr && y >= 0 && y < H && x >= 0 && x < Wd;
            const int sch = pos ^ (hp & 15);
            const bf16* src = ok ? img + ((long)y * Wd + x) * 128 + sch * 8 : zeros + sch * 8;
            glds16(src, halo + q * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_s_barrier();                      // round 3: the patch is retired by the barrier above and read one barrier later (staging rule, strict form)
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int hp = hp0 + dy * CH_HW + mt * 16 + dx;
                const int ab = hp * 256 + (((hp & 15) ^ g) << 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 af = *(const bf16x8*)(halo + (ab ^ (q << 6)));
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap * 4 + q], af, acc[mt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);              // keep the fragment reads of later taps from piling up in registers
        }
        __syncthreads();                                   // halo free for the next tile
        // D rows = output channel (g*4 + reg), D cols = pixel lr: only g == 0 lanes hold channels 0..3
        if (g == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < Cout) {
                        const long o = (((long)b * Cout + j) * H + y0 + w) * Wd + x0 + mt * 16 + lr;
                        const float v = acc[mt][j] + bs[j];
                        if (out_bf16) ET<bf16>::st((bf16*)out + o, v); else ((float*)out)[o] = v;
                    }
        }
    }
}

// ------------------------------------------------------------------------------- conv_out, ping-pong patches (round 6)
// conv3x3_out_halo_kernel above serialises fill -> 72 MFMAs -> stores per 8 x 32 tile: 13.5 us per tile, 1.95 ms for the 2.4 GB it reads (1.2 TB/s) -- one
// 85 KiB patch per CU, 144 registers of weight fragments (51 of them spilled) and the division-by-34 fill.  Here the tile is 4 x 32, its (4+2) x 36-pixel patch
// 54 KiB, and TWO patches sit in LDS: the next tile's fill flies under this tile's MFMAs.  The padded weights (16 rows x 1152) live in LDS (36 KiB,
// [k-step][row][4 chunks], chunk XOR 2 * ((row >> 2) & 1): conflict-free 16-row fragment reads) instead of registers.  Wave w owns output row w & 3, pixels
// (w >> 2) * 16 .. +15: one m-tile, 36 MFMAs in the order of the kernel above (tap-major, 4 k-steps) -> identical pixels.  Patch layout = conv3x3_halo_kernel's
// (rows padded to nine 4-pixel DMA instructions, chunk XOR 2 * (pixel-in-row & 7)).
// Per tile: fill(next) | vmcnt(7 + Cout): this tile's patch landed, the next fill and this wave's Cout stores of the previous tile may fly | barrier, barrier |
// 36 x (A read, B read, MFMA) | barrier (patch free) | stores.
#define CO2_TH 4
#define CO2_NQ ((CO2_TH + 2) * 9)                // 54 wave-instructions of 4 halo pixels
#define CO2_BUF (CO2_NQ * 1024)
#define CO2_W (2 * CO2_BUF)
#define CO2_LDS (CO2_W + 36 * 1024)
__global__ __launch_bounds__(512) void conv3x3_out2_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wt, const float* __restrict__ bias,
                                                          const bf16* __restrict__ zeros, void* __restrict__ out, int out_bf16, int B, int H, int Wd, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = l >> 4, lr = l & 15;
    const int orow = w & 3, ohalf = w >> 2;
    const int tiles_x = Wd / CH_TW, tiles_y = H / CO2_TH, tiles_img = tiles_x * tiles_y;
    const int NT = tiles_img * B, G = gridDim.x;
    // padded weights -> LDS, once per block: 36 k-steps x 16 rows x 4 chunks of 16 B
    for (int c = tid; c < 36 * 64; c += 512) {
        const int ks = c >> 6, row = (c >> 2) & 15, pos = c & 3, gq = pos ^ (((row >> 2) & 1) << 1);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < Cout) v = *(const u32x4*)(Wt + (long)row * (9 * 128) + ks * 32 + gq * 8);
        *(u32x4*)(smem + CO2_W + c * 16) = v;
    }
    float bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = (g * 4 + j < Cout) ? bias[g * 4 + j] : 0.f;
    auto fill = [&](int tx, int buf) __attribute__((always_inline)) {
        const int fb = tx / tiles_img, r = tx - fb * tiles_img;
        const int sy0 = (r / tiles_x) * CO2_TH - 1, sx0 = (r % tiles_x) * CH_TW - 1;
        const bf16* img = X + (long)fb * H * Wd * 128;
        int ll = l;
        asm volatile("" : "+v"(ll));
        const int lp = ll >> 4, pos = ll & 15;
        char* const dst = smem + buf * CO2_BUF;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int q = j * 8 + w;
            q = q < CO2_NQ ? q : CO2_NQ - 1;                                    // surplus instructions repeat the last one
            const int hy = q / 9, seg = q - hy * 9;                             // scalar
            const int y = sy0 + hy;
            const bool yok = y >= 0 && y < H;
            const int hx = seg * 4 + lp;
            const bool ok = yok && hx < CH_HW && (unsigned)(sx0 + hx) < (unsigned)Wd;
            const unsigned sch16 = (unsigned)((pos ^ ((hx & 7) << 1)) << 4);
            const char* const rowp = (const char*)(img + ((long)y * Wd + sx0) * 128);
            const char* const base = ok ? rowp : (const char*)zeros;
            const unsigned off = ok ? (unsigned)(hx << 8) + sch16 : sch16;
            glds16(base + off, dst + q * 1024);
        }
    };
    int tix = blockIdx.x, cur = 0;
    if (tix < NT) fill(tix, 0);
    __syncthreads();                                                            // weights visible (and the first patch landed: __syncthreads waits vmcnt(0))
    const int boff = CO2_W + lr * 64 + ((g ^ (((lr >> 2) & 1) << 1)) << 4);
    while (tix < NT) {
        const int nxt = tix + G;
        if (nxt < NT) {
            fill(nxt, cur ^ 1);                                                 // the other patch: every wave is behind the barrier that followed its last read of it
            if (Cout == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // this tile's patch landed; younger: Cout stores of the previous tile + the 7 fill instructions
            else if (Cout == 4) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");                                 // retirement and first read one barrier apart (staging rule, strict form)
        const int b = tix / tiles_img, r = tix - b * tiles_img;
        const int y0 = (r / tiles_x) * CO2_TH, x0 = (r % tiles_x) * CH_TW;
        const char* const halo = smem + cur * CO2_BUF;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int hx = ohalf * 16 + dx + lr;
            const int ab = ((orow + dy) * CH_HWP + hx) * 256 + ((((hx & 7) << 1) ^ g) << 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 af = *(const bf16x8*)(halo + (ab ^ (q << 6)));
                const bf16x8 wf = *(const bf16x8*)(smem + boff + (tap * 4 + q) * 1024);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");                                 // patch `cur` free for the fill after next
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < Cout) {
                    const long o = (((long)b * Cout + j) * H + y0 + orow) * Wd + x0 + ohalf * 16 + lr;
                    const float v = acc[j] + bs[j];
                    if (out_bf16) ET<bf16>::st((bf16*)out + o, v); else ((float*)out)[o] = v;
                }
        }
        tix = nxt; cur ^= 1;
    }
}

// ------------------------------------------------------------------------------- norm_out + swish + conv_out in ONE pass (round 6)
// Decoder.forward's tail (vq_model.py:210-214): h = conv_out(nonlinearity(norm_out(h))).  The unfused tail reads the fp32 skip stream (4.8 GB at
// 64 x 384^2 x 128), writes the normalised bf16 tensor (2.4 GB), and conv_out reads that again (2.4 GB at 1.2 TB/s: its LDS-DMA fill and its MFMA
// phase alternate on one 85 KiB patch per CU).  Here the block reads the fp32 skip stream DIRECTLY: the (4+2) x (32+2) x 128-channel patch of a
// 4 x 32 output tile comes through registers -- y = swish(x a[b][c] + sh[b][c]) with the coefficients gn_finalize_kernel wrote, exactly
// gn_apply_kernel's arithmetic -- is packed to bf16 and written into the same XOR-swizzled LDS layout the MFMA phase of conv3x3_out_halo_kernel
// reads.  The NEXT tile's 13 loads per thread are issued before the current tile's MFMA phase and fly under it.  Same accumulation order (tap-major,
// one chain per m-tile) as the unfused kernel: the two tails are bit-identical (tests/test_gpu_ops.py).  Out-of-image halo pixels are zeros of the
// NORMALISED tensor (conv_out pads its own input).
#define CO_TH 4
#define CO_HP ((CO_TH + 2) * CH_HW)             // 204 halo pixels
#define CO_NV ((CO_HP * 32 + 511) / 512)        // f32x4 vectors per thread per tile: 13
#define CO_LDS (CO_HP * 256)
__global__ __launch_bounds__(512) void conv3x3_out_gn_kernel(const float* __restrict__ X, const float* __restrict__ coef, const bf16* __restrict__ Wt,
                                                            const float* __restrict__ bias, void* __restrict__ out, int out_bf16, int B, int H, int Wd,
                                                            int Cout, int swish) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = l >> 4, lr = l & 15;
    const int tiles_x = Wd / CH_TW, tiles_y = H / CO_TH, tiles_img = tiles_x * tiles_y;
    const int NT = tiles_img * B, G = gridDim.x;
    bf16x8 wf[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        wf[q] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (lr < Cout) wf[q] = *(const bf16x8*)(Wt + (long)lr * (9 * 128) + q * 32 + g * 8);
    }
    float bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = (g * 4 + j < Cout) ? bias[g * 4 + j] : 0.f;
    // staging: vector v = it * 512 + tid -> halo pixel v >> 5, channels (v & 31) * 4 .. +3: the thread's channel quad is the same for every vector
    const int c4 = tid & 31;
    f32x4 xv[CO_NV];
    unsigned okmask = 0;
    // address + in-image flag of vector ``it`` of the tile at (image base, y0, x0) (clamped: the load is unconditional, the value masked).  The tile ->
    // (image, y0, x0) divisions are done ONCE per tile by the caller: inside this per-vector helper they cost more than the normalisation itself.
    auto src_of = [&](const float* img, int y0, int x0, int it, bool& ok) __attribute__((always_inline)) -> const f32x4* {
        int hp = it * 16 + (tid >> 5);
        const bool inr = hp < CO_HP;
        hp = inr ? hp : CO_HP - 1;
        const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        ok = inr && y >= 0 && y < H && x >= 0 && x < Wd;
        const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= Wd ? Wd - 1 : x);
        return (const f32x4*)(img + ((long)yc * Wd + xc) * 128 + c4 * 4);
    };
    auto tile_of = [&](int tix, const float*& img, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        b = tix / tiles_img;
        const int r = tix - b * tiles_img, ty = r / tiles_x;
        y0 = ty * CO_TH; x0 = (r - ty * tiles_x) * CH_TW;
        img = X + (long)b * H * Wd * 128;
    };
    const int orow = w >> 1, omt = w & 1;                         // wave w: output row w >> 1 of the 4 x 32 tile, pixels (w & 1) * 16 .. +15
    const int hp0 = orow * CH_HW + omt * 16 + lr;
    int tix = blockIdx.x;
    const float* img_n = X; int b_n = 0, y0_n = 0, x0_n = 0;
    if (tix < NT) {
        tile_of(tix, img_n, b_n, y0_n, x0_n);
#pragma unroll
        for (int it = 0; it < CO_NV; ++it) { bool ok; xv[it] = *src_of(img_n, y0_n, x0_n, it, ok); okmask |= ok ? (1u << it) : 0u; }
    }
    for (; tix < NT; tix += G) {
        const int b = b_n, y0 = y0_n, x0 = x0_n;                    // this tile (computed when its loads were issued)
        const f32x4 ca = *(const f32x4*)(coef + ((long)b * 128 + c4 * 4) * 2), cb = *(const f32x4*)(coef + ((long)b * 128 + c4 * 4) * 2 + 4);   // a0 sh0 a1 sh1 | a2 sh2 a3 sh3
        if (tix + G < NT) tile_of(tix + G, img_n, b_n, y0_n, x0_n);   // the next tile (the last one re-reads itself and drops the data)
        unsigned oknext = 0;
        // vector by vector: normalise + swish + pack, then the SAME register is re-armed with the next tile's vector (its load flies under the rest of
        // this loop, the MFMA phase and the stores), then the packed value goes to LDS.  In-order returns: the next tile consumes xv[0] first.
#pragma unroll
        for (int it = 0; it < CO_NV; ++it) {
            const int hp = it * 16 + (tid >> 5);
            float t0 = fmaf(xv[it][0], ca[0], ca[1]), t1 = fmaf(xv[it][1], ca[2], ca[3]), t2 = fmaf(xv[it][2], cb[0], cb[1]), t3 = fmaf(xv[it][3], cb[2], cb[3]);
            bool okn;
            const f32x4* nsrc = src_of(img_n, y0_n, x0_n, it, okn);
            xv[it] = *nsrc; oknext |= okn ? (1u << it) : 0u;      // unconditional (the last tile re-reads itself and drops it): a conditional load costs a vmcnt(0) at the join
            if (swish) {                                          // gn_apply_kernel's bf16-output form
                t0 = t0 * __frcp_rn(1.f + __expf(-t0)); t1 = t1 * __frcp_rn(1.f + __expf(-t1));
                t2 = t2 * __frcp_rn(1.f + __expf(-t2)); t3 = t3 * __frcp_rn(1.f + __expf(-t3));
            }
            u32x2 pk; pk.x = pack_bf16x2(t0, t1); pk.y = pack_bf16x2(t2, t3);
            if (!((okmask >> it) & 1u)) { pk.x = 0u; pk.y = 0u; }
            // logical 16-byte chunk j = c4 >> 1 of pixel hp lives at slot j ^ (hp & 15); this thread owns its low / high 8 bytes
            if (hp < CO_HP) *(u32x2*)(halo + hp * 256 + ((((c4 >> 1) ^ (hp & 15))) << 4) + (c4 & 1) * 8) = pk;
        }
        okmask = oknext;
        __syncthreads();                                          // the patch is complete
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int hp = hp0 + dy * CH_HW + dx;
            const int ab = hp * 256 + (((hp & 15) ^ g) << 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 af = *(const bf16x8*)(halo + (ab ^ (q << 6)));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap * 4 + q], af, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                          // every wave is done reading: the next tile may overwrite the patch
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < Cout) {
                    const long o = (((long)b * Cout + j) * H + y0 + orow) * Wd + x0 + omt * 16 + lr;
                    const float v = acc[j] + bs[j];
                    if (out_bf16) ET<bf16>::st((bf16*)out + o, v); else ((float*)out)[o] = v;
                }
        }
    }
}
// norm_out + swish + conv_out from the fp32 skip stream; false when the shape is not the decoder tail's (the caller runs gn_apply + conv_out_halo_try)
bool conv_out_gn_try(hipStream_t s, const float* x_f32, const float* coef, const bf16* w, const float* bias, void* out, int out_bf16,
                     int B, int H, int Wd, int Cin, int Cout, int swish) {
    if (!pg_tune->conv_halo || Cin != 128 || Cout > 4 || H % CO_TH || Wd % CH_TW) return false;
    const int tiles = B * (H / CO_TH) * (Wd / CH_TW);
    if (tiles < 64) return false;
    auto kfn = conv3x3_out_gn_kernel;
    if (!PG_DYN_LDS(kfn, CO_LDS)) return false;
    hipLaunchKernelGGL(kfn, dim3(tiles < pg_cu_count() ? tiles : pg_cu_count()), dim3(512), CO_LDS, s, x_f32, coef, w, bias, out, out_bf16, B, H, Wd, Cout, swish);
    return true;
}

bool conv_out_halo_try(hipStream_t s, const bf16* x, const bf16* w, const float* bias, const bf16* zeros, void* out, int out_bf16,
                       int B, int H, int Wd, int Cin, int Cout) {
    if (!pg_tune->conv_halo || Cin != 128 || Cout > 4 || H % CH_TH || Wd % CH_TW) return false;
    const int tiles = B * (H / CH_TH) * (Wd / CH_TW);
    if (tiles < 64) return false;
    if (pg_tune->conv_halo != 2) {                                     // round 6: 4 x 32 tiles, two patches in LDS (conv_halo = 2 keeps the one-patch kernel: the reference of the equality test)
        const int t2 = tiles * (CH_TH / CO2_TH);
        auto k2 = conv3x3_out2_kernel;
        (void)PG_DYN_LDS(k2, CO2_LDS);
        hipLaunchKernelGGL(k2, dim3(t2 < 256 ? t2 : 256), dim3(512), CO2_LDS, s, x, w, bias, zeros, out, out_bf16, B, H, Wd, Cout);
        return true;
    }
    auto kfn = conv3x3_out_halo_kernel;
    (void)PG_DYN_LDS(kfn, CH_HALO_BYTES);
    hipLaunchKernelGGL(kfn, dim3(tiles < 256 ? tiles : 256), dim3(512), CH_HALO_BYTES, s, x, w, bias, zeros, out, out_bf16, B, H, Wd, Cout);
    return true;
}
