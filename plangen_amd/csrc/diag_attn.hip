// libplangen_diag.so only: measurement forms of the fused decode-attention kernel (attn_decode.h), reached through PgDiagHooks::attn_decode
// when pg_diag_set_option(h, "attn_variant", v) selected one.  v = 100: the round-2 non-pipelined 7-deep kernel (correct results);
// 101-107: timing ablations of the production kernel -- no K/V append store (1), no slab / cos / sin loads (2), no merge epilogue (4) and
// their combinations -- whose RESULTS ARE WRONG BY CONSTRUCTION (profiles/r02_c, r04 DESIGN notes).  None of them exists in libplangen_hip.so.
#include "attn_decode.h"
#include "diag.h"

template <typename T>
static bool diag_attn_decode_t(hipStream_t s, const float* qkv, int S, long slab, T* obuf, T* kc, T* vc, const float* cos_t, const float* sin_t,
                               const SeqState& st, int M, int nh, int slots, int max_pos, float scale) {
#define ATT_LAUNCH(U, W, A) hipLaunchKernelGGL((attn_decode_fused_kernel<T, U, W, A>), dim3(nh, M), dim3(64 * W), 0, s, st.row_order, st.len, st.n_dec, kc, vc, nh, slots, st.shared_len, st.shared_row, qkv, slab, obuf, cos_t, sin_t, st.pos_off, S, max_pos, scale)
    const int av = pg_tune->attn_variant;
    const bool small = (M * nh <= 512 && pg_tune->attn_waves != 4) || pg_tune->attn_waves == 8;
    if (av == 100) { if (small) ATT_LAUNCH(7, 8, 0); else ATT_LAUNCH(7, 4, 0); return true; }
    if (av > 100 && av < 108 && !small) {
        switch (av) {
            case 101: ATT_LAUNCH(6, 4, 17); break;
            case 102: ATT_LAUNCH(6, 4, 18); break;
            case 103: ATT_LAUNCH(6, 4, 19); break;
            case 104: ATT_LAUNCH(6, 4, 20); break;
            default: ATT_LAUNCH(6, 4, 23); break;
        }
        return true;
    }
    return false;
#undef ATT_LAUNCH
}
bool diag_attn_decode(hipStream_t s, bool is_bf16, const float* qkv, int S, long slab, void* obuf, void* kc, void* vc, const float* cos_t, const float* sin_t,
                      const SeqState& st, int M, int nh, int slots, int max_pos, float scale) {
    if (M <= 0 || pg_tune->attn_variant < 100) return false;
    return is_bf16 ? diag_attn_decode_t<bf16>(s, qkv, S, slab, (bf16*)obuf, (bf16*)kc, (bf16*)vc, cos_t, sin_t, st, M, nh, slots, max_pos, scale)
                   : diag_attn_decode_t<float>(s, qkv, S, slab, (float*)obuf, (float*)kc, (float*)vc, cos_t, sin_t, st, M, nh, slots, max_pos, scale);
}
const PgDiagHooks* diag_hooks() {          // (a namespace-scope table would also be emitted for the device pass, which cannot see host functions)
    static PgDiagHooks h;
    h.attn_decode = diag_attn_decode;
    h.gemm_big_wave = gemm_bw_try;
    return &h;
}
