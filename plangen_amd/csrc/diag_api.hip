// libplangen_diag.so only: the switches that do NOT belong in the product library -- measurement modes whose results are garbage by
// construction (skip_attn, attn_variant 101-107), kernel-form selectors of the sweeps (attn_variant 100, attn_waves, wt_store,
// split_target_small / _mid), and A/B switches of settled questions (fuse_rope, force_swiglu, lpt_order).  libplangen_diag.so is a
// SUPERSET build: the same object files as libplangen_hip.so plus diag_*.o / bench_kernels.o / chain.o, so a handle created through it
// runs exactly the product's code until one of these is set.  tools/, bench.py's instrumented pass and the hazard-screen tests use it.
#include "engine.h"
#include "diag.h"

extern "C" {

int pg_diag_set_option(pg_handle h, const char* key, int64_t value) {
    if (!h || !key) return PG_ERR_ARG;
    h->tune.diag = diag_hooks();
    if (!strcmp(key, "skip_attn")) { h->skip_attn = value != 0; return PG_OK; }          // MEASUREMENT ONLY: decode steps without their attention launches
    if (!strcmp(key, "attn_variant")) { h->tune.attn_variant = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "attn_waves")) { h->tune.attn_waves = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "wt_store")) { h->tune.wt_store = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "split_target_small")) { h->tune.split_small = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "split_target_mid")) { h->tune.split_mid = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "gn_epilogue256")) { h->tune.gn_epilogue256 = value != 0; return PG_OK; }      // GroupNorm partials from the 256x256 conv epilogue (measured slower, profiles/r06_c)
    if (!strcmp(key, "sk3_xa")) { h->tune.sk3_xa = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "sk5")) { h->tune.sk5 = value != 0; h->tune_epoch++; return PG_OK; }      // 0: the round-5 v3 blocks for the wide-N decode GEMMs at 65..128 rows (bf16 sums round differently)
    if (!strcmp(key, "defer_norm")) { h->defer_norm = value != 0; h->tune_epoch++; return PG_OK; }      // 0: rmsnorm512 + plain GEMMs at 65..128 rows too (round-5 arithmetic; bf16 rounding differs)
    if (!strcmp(key, "fuse_rope")) { h->fuse_rope = value != 0; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "force_swiglu")) { h->force_swiglu = value != 0; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "lpt_order")) { h->lpt_order = value != 0; h->tune_epoch++; return PG_OK; }
    h->err = std::string("unknown diagnostics option ") + key;
    return PG_ERR_ARG;
}

}  // extern "C"
