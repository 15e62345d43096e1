// libplangen_diag.so only: the skinny-GEMM VARIANT TABLE of the microbenchmarks (tools/skinny_sweep.py, sk4_sweep.py, sk4_profile.py,
// sk4_load_stress.py): older kernel generations, ring-depth / block-shape sweeps, timing ablations (results wrong by construction) and the
// round-4 hazard-forensics protocol variants.  None of these instantiations is linked into libplangen_hip.so.
#include "gemm_skinny.h"
#include "diag.h"

// variant table for the microbenchmark (tools/skinny_sweep.py): returns BK (0 = unsupported)
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
        // round 6: what the o_proj launch (5.2 us for 8.4 MB) is made of -- the production block (variant 300) against an empty kernel of
        // the same geometry, the weight stream alone, and the kernel without its stores / without its x tile (tools/launch_floor.py)
        case 320: return sk4_nck<4, 4, 3, 4, 2, 64 | 8192>(s, x, W, out, M, N, K, S) ? 128 : 0;     // empty kernel, same grid / block / LDS / kernarg block
        case 321: return sk4_nck<4, 4, 3, 4, 2, 64 | 7>(s, x, W, out, M, N, K, S) ? 128 : 0;        // W stream only (no x DMA / barriers, no MFMA, no stores)
        case 322: return sk4_nck<4, 4, 3, 4, 2, 64 | 4>(s, x, W, out, M, N, K, S) ? 128 : 0;        // no stores
        case 323: return sk4_nck<4, 4, 3, 4, 2, 64 | 1>(s, x, W, out, M, N, K, S) ? 128 : 0;        // no x tile
        case 324: return sk4_nck<4, 4, 3, 4, 2, 64 | 2>(s, x, W, out, M, N, K, S) ? 128 : 0;        // no MFMA
        case 307: return sk4_nck<2, 4, 3, 4, 4, 64>(s, x, W, out, M, N, K, S) ? 128 : 0;         // 32-row blocks (two pieces per wave)
        // round 4: W register-ring depth of the wide-N production block (64 rows x 128 columns, 8 waves, tiled W) -- bytes in flight per block
        case 400: return sk3_prod_nck<4, 0, 8, true, 2>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 401: return sk3_prod_nck<4, 0, 8, true, 3>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 402: return sk3_prod_nck<4, 0, 8, true, 4>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        case 403: return sk3_prod_nck<4, 0, 8, true, 6>(s, x, W, out, M, N, K, S, K / SK_BK / S) ? 128 : 0;
        // round 6: the bs=64 PRODUCTION wide-N blocks (64 rows x 128 columns, 8 waves, tiled W, ring 2) with per-wave cycle stamps
        // (tools/sk3_profile.py): 500 gate|up + SwiGLU (S = 1, 16 chunks), 501 qkv (S = 2, 8 chunks), 502 / 503 the same with a ring of 4
        case 500: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 8, true, true>(s, x, W, out, M, N, K, S, SkRowScale{(const float*)g_sk4_prof, 0.f}); return 128;
        case 501: if (S != 2 || K != 2048) return 0; launch_sk3<4, 8, 2, true, 0, 8, true, true>(s, x, W, out, M, N, K, S, SkRowScale{(const float*)g_sk4_prof, 0.f}); return 128;
        // x prefetched TWO chunks ahead (XA = 2): 503 stamped gate|up, 504 / 505 plain gate|up + SwiGLU (ring 2 / ring 3), 506 / 507 qkv slabs S = 2 (ring 2 / 3), 508 slab form S = 1
        case 503: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 8, true, true, 2>(s, x, W, out, M, N, K, S, SkRowScale{(const float*)g_sk4_prof, 0.f}); return 128;
        case 504: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 8, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 505: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 3, true, 1, 8, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 506: if (S != 2 || K != 2048) return 0; launch_sk3<4, 8, 2, true, 0, 8, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 507: if (S != 2 || K != 2048) return 0; launch_sk3<4, 8, 3, true, 0, 8, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 508: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 0, 8, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 509: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 8, true, false, 1>(s, x, W, out, M, N, K, S); return 128;      // production gate|up + SwiGLU, unstamped (reference for 504 / 505)
        // 96-column blocks (6 waves): gate|up 236 blocks, qkv (S = 2) 256 blocks instead of 176 / 192 -- per-CU ingest 640 / 320 KiB instead of 768 / 384
        case 520: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 6, true, false, 1>(s, x, W, out, M, N, K, S); return 128;
        case 521: if (S != 2 || K != 2048) return 0; launch_sk3<4, 8, 2, true, 0, 6, true, false, 1>(s, x, W, out, M, N, K, S); return 128;
        case 522: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 3, true, 1, 6, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 523: if (S != 2 || K != 2048) return 0; launch_sk3<4, 8, 3, true, 0, 6, true, false, 2>(s, x, W, out, M, N, K, S); return 128;
        case 530: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 8, true, false, 1, true>(s, x, W, out, M, N, K, S); return 128;      // production gate|up + SwiGLU with nt weight loads
        case 531: if (S != 2 || K != 2048) return 0; launch_sk3<4, 8, 2, true, 0, 8, true, false, 1, true>(s, x, W, out, M, N, K, S); return 128;       // qkv with nt weight loads
        case 524: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 2, true, 1, 4, true, false, 1>(s, x, W, out, M, N, K, S); return 128;      // 64-column blocks: 352
        case 502: if (S != 1 || K != 2048) return 0; launch_sk3<4, 16, 4, true, 1, 8, true, true>(s, x, W, out, M, N, K, S, SkRowScale{(const float*)g_sk4_prof, 0.f}); return 128;
        // round 6: v5 (n-tile pairs x two K halves, x by LDS-DMA): 510 gate|up + SwiGLU (S = 1), 511 fp32 slabs (qkv S = 2, gen_head / lm_head S = 1)
        case 510: return sk5_try<3>(s, x, W, out, M, N, K, S) ? 128 : 0;
        case 511: return sk5_try<4>(s, x, W, out, M, N, K, S) ? 128 : 0;
        case 121: return sk4_nck<8, 3, 8, 2, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;     // deep W ring, direct 16-byte stores
        case 123: return sk4_nck<8, 3, 3, 2, 1>(s, x, W, out, M, N, K, S) ? 128 : 0;     // shallow W ring, direct stores
        default: return 0;
    }
}

// Round 5 experiment: the split-K producer with the consumer norm's reduction in its tail (gemm_skinny.h, EPI 5), same block
// shapes as the production o_proj / down_proj dispatch (sk4_prod: 64-row blocks above 16 rows, 16-row blocks up to 16).
bool launch_gemm_skinny_fused_norm(hipStream_t s, const bf16* x, const bf16* Wt, float* slabs, int M, int N, int K, int S, const SkFuse* site_dev) {
    if (M > 128 || !site_dev) return false;
    if (M > 16) return sk4_nck<4, 4, 3, 5, 2, 64>(s, x, Wt, slabs, M, N, K, S, site_dev);
    return sk4_nck<1, 5, 3, 5, 4, 64>(s, x, Wt, slabs, M, N, K, S, site_dev);
}
