// The C ABI of libplangen_hip.so (include/plangen_hip.h): thin entry points over pg_engine.  Every symbol exported by the library is
// defined here and listed in plangen_hip.map; nothing else leaves the shared object.
#include "engine.h"

static thread_local std::string g_err;

int pg_engine::fetch_timing() {
    if (have_decode_t) { HIPCHK(hipEventSynchronize(ev_t1)); HIPCHK(hipEventElapsedTime(&timing.decode_ms, ev_t0, ev_t1)); have_decode_t = false; }
    if (have_prefill_t) { HIPCHK(hipEventSynchronize(ev_p1)); HIPCHK(hipEventElapsedTime(&timing.prefill_ms, ev_p0, ev_p1)); have_prefill_t = false; }
    if (have_vq_t) { HIPCHK(hipEventSynchronize(ev_v1)); HIPCHK(hipEventElapsedTime(&timing.vq_ms, ev_v0, ev_v1)); have_vq_t = false; }
    if (tc_used) {
        for (int c = 0; c < TC_N; ++c) { tc_ms[c] = 0; tc_bytes[c] = 0; tc_launches[c] = 0; }
        HIPCHK(hipEventSynchronize(tc_ev[tc_used - 1]));
        for (size_t i = 0; i + 1 < tc_used; i += 2) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, tc_ev[i], tc_ev[i + 1]));
            const int c = tc_meta[i / 2].first;
            tc_ms[c] += ms; tc_bytes[c] += tc_meta[i / 2].second; tc_launches[c]++;
        }
        timing.attn_ms_sum = (float)tc_ms[TC_ATTN]; timing.attn_launches = tc_launches[TC_ATTN]; timing.attn_bytes_sum = tc_bytes[TC_ATTN];
        tc_used = 0;
    }
    return PG_OK;
}

extern "C" {

int pg_create(pg_handle* out, const pg_config* cfg, int device_id) {
    if (!out || !cfg) { g_err = "pg_create: null argument"; return PG_ERR_ARG; }
    pg_engine* e = new pg_engine();
    e->cfg = *cfg; e->dev = device_id;
    const int rc = e->create();
    if (rc != PG_OK) { g_err = e->err; e->destroy(); delete e; *out = nullptr; return rc; }
    *out = e;
    return PG_OK;
}
int pg_destroy(pg_handle h) { if (h) { h->destroy(); delete h; } return PG_OK; }
const char* pg_last_error(pg_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int pg_load_tensor(pg_handle h, const char* name, const void* src, int dtype, const int64_t* shape, int ndim) {
    if (!h || !name || !src) return PG_ERR_ARG;
    return h->load_tensor(name, src, dtype, shape, ndim);
}
int pg_finalize_weights(pg_handle h, int* missing, pg_stream s) { TuneGuard _tg(h); return h ? h->finalize(missing, (hipStream_t)s) : PG_ERR_ARG; }

int pg_prefill(pg_handle h, const int32_t* ids_dev, const int32_t* pad_len_host, int R, int L, int position_mode,
               void* hidden_out_dev, int hidden_dtype, pg_stream s) { TuneGuard _tg(h);
    if (!h || !ids_dev || !pad_len_host) return PG_ERR_ARG;
    return h->prefill(ids_dev, nullptr, 0, pad_len_host, R, L, position_mode, hidden_out_dev, hidden_dtype, (hipStream_t)s);
}
int pg_prefill_embeds(pg_handle h, const void* embeds_dev, int embeds_dtype, const int32_t* pad_len_host, int R, int L,
                      int position_mode, void* hidden_out_dev, int hidden_dtype, pg_stream s) { TuneGuard _tg(h);
    if (!h || !embeds_dev || !pad_len_host) return PG_ERR_ARG;
    return h->prefill(nullptr, embeds_dev, embeds_dtype, pad_len_host, R, L, position_mode, hidden_out_dev, hidden_dtype, (hipStream_t)s);
}
int pg_step(pg_handle h, const void* embeds_dev, int embeds_dtype, void* hidden_out_dev, int hidden_dtype, pg_stream s) { TuneGuard _tg(h);
    if (!h || !embeds_dev) return PG_ERR_ARG;
    return h->step(embeds_dev, embeds_dtype, hidden_out_dev, hidden_dtype, (hipStream_t)s);
}
int pg_gen_head(pg_handle h, const void* h_dev, int h_dtype, float* logits_dev, int R, pg_stream s) { TuneGuard _tg(h);
    if (!h || !h_dev || !logits_dev) return PG_ERR_ARG;
    return h->gen_head(h_dev, h_dtype, logits_dev, R, (hipStream_t)s);
}
int pg_gen_embed(pg_handle h, const int32_t* tok_dev, void* out_dev, int out_dtype, int R, pg_stream s) {
    if (!h || !tok_dev || !out_dev) return PG_ERR_ARG;
    if (!h->finalized) { h->err = "pg_finalize_weights not called"; return PG_ERR_STATE; }
    if (R < 0 || (out_dtype != PG_F32 && (long)R * h->H() > h->part_elems)) { h->err = "pg_gen_embed: too many rows for the bf16 output scratch"; return PG_ERR_CAPACITY; }
    (void)hipSetDevice(h->dev);
    hipStream_t st = (hipStream_t)s;
    if (out_dtype == PG_F32) launch_embed_gather(st, h->gen_table, tok_dev, nullptr, (float*)out_dev, R, h->H(), h->cfg.img_vocab);
    else {
        // gather fp32 rows into x-sized scratch is not safe mid-sequence: convert row by row through part
        launch_embed_gather(st, h->gen_table, tok_dev, nullptr, h->part, R, h->H(), h->cfg.img_vocab);
        launch_f32_to_rows(st, h->part, out_dev, 1, nullptr, R, h->H());
    }
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_embed_tokens(pg_handle h, const int32_t* ids_dev, void* out_dev, int out_dtype, int n, pg_stream s) {
    if (!h || !ids_dev || !out_dev) return PG_ERR_ARG;
    (void)hipSetDevice(h->dev);
    hipStream_t st = (hipStream_t)s;
    if (out_dtype == PG_F32) launch_embed_gather(st, h->embed, ids_dev, nullptr, (float*)out_dev, n, h->H(), h->cfg.vocab);
    else {
        if ((long)n * h->H() > h->part_elems) { h->err = "pg_embed_tokens: too many tokens for bf16 output scratch"; return PG_ERR_CAPACITY; }
        launch_embed_gather(st, h->embed, ids_dev, nullptr, h->part, n, h->H(), h->cfg.vocab);
        launch_f32_to_rows(st, h->part, out_dev, 1, nullptr, n, h->H());
    }
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_decode_image_tokens(pg_handle h, int T, float cfg_weight, float temperature, uint64_t seed,
                           const int32_t* force_tok_dev, const uint8_t* force_mask_dev, int32_t* out_tok_dev,
                           float* logits_out_dev, pg_stream s) { TuneGuard _tg(h);
    if (!h || !out_tok_dev) return PG_ERR_ARG;
    return h->decode_image(T, cfg_weight, temperature, seed, force_tok_dev, force_mask_dev, out_tok_dev, logits_out_dev, (hipStream_t)s);
}
int pg_generate_text_greedy(pg_handle h, int max_new, int min_new, int eos_id, int64_t* out_dev, int* out_len_host, pg_stream s) { TuneGuard _tg(h);
    if (!h || !out_dev) return PG_ERR_ARG;
    return h->text_greedy(max_new, min_new, eos_id, out_dev, out_len_host, (hipStream_t)s);
}
int pg_vq_decode(pg_handle h, const int32_t* codes_dev, void* img_out_dev, int out_dtype, int B, pg_stream s) { TuneGuard _tg(h);
    if (!h || !codes_dev || !img_out_dev) return PG_ERR_ARG;
    return h->bf ? h->vq_decode<bf16>(codes_dev, img_out_dev, out_dtype, B, (hipStream_t)s)
                 : h->vq_decode<float>(codes_dev, img_out_dev, out_dtype, B, (hipStream_t)s);
}
int pg_vq_encode(pg_handle h, const void* img_dev, int img_dtype, int64_t* idx_out_dev, int B, pg_stream s) { TuneGuard _tg(h);
    if (!h || !img_dev || !idx_out_dev) return PG_ERR_ARG;
    return h->bf ? h->vq_encode<bf16>(img_dev, img_dtype, idx_out_dev, B, (hipStream_t)s)
                 : h->vq_encode<float>(img_dev, img_dtype, idx_out_dev, B, (hipStream_t)s);
}
int pg_vision_encode(pg_handle h, const void* img_dev, int img_dtype, void* out_dev, int out_dtype, int B, pg_stream s) { TuneGuard _tg(h);
    if (!h || !img_dev || !out_dev) return PG_ERR_ARG;
    return h->bf ? h->vision_encode<bf16>(img_dev, img_dtype, out_dev, out_dtype, B, (hipStream_t)s)
                 : h->vision_encode<float>(img_dev, img_dtype, out_dev, out_dtype, B, (hipStream_t)s);
}
int pg_get_timing(pg_handle h, pg_timing* out) {
    if (!h || !out) return PG_ERR_ARG;
    const int rc = h->fetch_timing();
    *out = h->timing;
    return rc;
}
int pg_get_class_timing(pg_handle h, int cls, const char** name, double* ms_sum, int* launches, double* bytes_sum) {
    static const char* const names[pg_engine::TC_N] = {"decode_attention", "decode_gemm_qkv", "decode_gemm_o", "decode_gemm_gate_up_swiglu",
                                                       "decode_gemm_down", "decode_rmsnorm", "decode_gen_head", "decode_cfg_sampler", "empty_event_pair"};
    if (!h || cls < 0 || cls >= pg_engine::TC_N) return PG_ERR_ARG;
    if (name) *name = names[cls];
    if (ms_sum) *ms_sum = h->tc_ms[cls];
    if (launches) *launches = h->tc_launches[cls];
    if (bytes_sum) *bytes_sum = h->tc_bytes[cls];
    return PG_OK;
}
int pg_set_option(pg_handle h, const char* key, int64_t value) {
    // The product's switches (include/plangen_hip.h documents each): measurement taps, per-call hints, and A/B fallbacks that produce the same
    // results.  Experiments that lost their measurement and switches that make a handle compute something else live in libplangen_diag.so
    // (pg_diag_set_option, diag_api.hip) -- nothing reachable from this function changes what a handle returns.
    if (!h || !key) return PG_ERR_ARG;
    if (!strcmp(key, "time_attn")) { h->time_attn = value != 0; return PG_OK; }
    if (!strcmp(key, "time_stride")) { h->time_stride = value > 0 ? (int)value : 1; return PG_OK; }
    if (!strcmp(key, "rng_image_offset")) { h->rng_image_offset = (int)value; return PG_OK; }
    if (!strcmp(key, "allow_partial_weights")) { h->allow_partial = value != 0; return PG_OK; }
    if (!strcmp(key, "uncond_shared_hint")) { h->uncond_hint = value < 0 ? -1 : (value != 0); return PG_OK; }
    if (!strcmp(key, "share_uncond")) { h->share_uncond = value != 0; return PG_OK; }
    if (!strcmp(key, "use_graph")) { h->use_graph = value != 0; return PG_OK; }
    if (!strcmp(key, "lanes")) { h->lanes_opt = (int)value; return PG_OK; }
    if (!strcmp(key, "stream_gemm")) { h->tune.stream_gemm = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "split_target_big")) { h->tune.split_big = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "flash_prefill")) { h->flash_prefill = value != 0; return PG_OK; }
    if (!strcmp(key, "prefill_attn")) { h->tune.prefill_attn = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "prefill_rope_epi")) { h->prefill_rope_epi = value != 0; return PG_OK; }
    if (!strcmp(key, "prefill_res_epi")) { h->prefill_res_epi = value != 0; return PG_OK; }
    if (!strcmp(key, "gemm256")) {        // 0 | 1 | 4 | 5 | 6, optionally + 8 (gemm256.hip::pick_tile_height documents the encoding)
        const int v = (int)value, base = v & 7;
        if (value < 0 || value > 15 || (v & ~15) || !(base == 0 || base == 1 || base == 4 || base == 5 || base == 6) || (base == 0 && v != 0)) return PG_ERR_ARG;
        h->tune.gemm256 = v; h->tune_epoch++; return PG_OK;
    }
    if (!strcmp(key, "conv_halo")) { h->tune.conv_halo = (int)value; return PG_OK; }
    if (!strcmp(key, "vq_mid_bf16")) { h->mid_bf16 = value != 0; return PG_OK; }
    if (!strcmp(key, "vq_tail_fused")) { h->tune.vq_tail_fused = value != 0; return PG_OK; }
    if (!strcmp(key, "vq_argmin_multi")) { h->tune.vq_argmin_multi = (int)value; return PG_OK; }
    if (!strcmp(key, "vit_attn")) { h->tune.vit_attn = (int)value; return PG_OK; }
    if (!strcmp(key, "ln_wave")) { h->tune.ln_wave = (int)value; return PG_OK; }
    h->err = std::string("unknown option ") + key;
    return PG_ERR_ARG;
}
int64_t pg_device_bytes(pg_handle h) { return h ? h->bytes : 0; }

int pg_debug_read(pg_handle h, const char* name, int index, void* dst_dev, int64_t max_bytes, pg_stream s) {
    if (!h || !name || !dst_dev) return PG_ERR_ARG;
    const void* src = nullptr; int64_t n = 0;
    const std::string nm = name;
    if (nm == "kcache") { src = h->kc(index); n = (int64_t)h->kv_layer_elems() * h->esz; }
    else if (nm == "vcache") { src = h->vc(index); n = (int64_t)h->kv_layer_elems() * h->esz; }
    else if (nm == "x") { src = h->x; n = (int64_t)h->max_tok * h->H() * 4; }
    else if (nm == "xn") { src = h->xn; n = (int64_t)h->max_tok * h->H() * h->esz; }
    else if (nm == "hfin") { src = h->hfin; n = (int64_t)h->cfg.max_rows * h->H() * h->esz; }
    else if (nm == "gen_table") { src = h->gen_table; n = (int64_t)h->cfg.img_vocab * h->H() * 4; }
    else if (nm == "pq_table") { src = h->pq_table; n = (int64_t)h->cfg.img_vocab * h->cfg.vq_z * h->esz; }
    else if (nm == "qbuf") { src = h->qbuf; n = (int64_t)h->max_tok * h->HD() * h->esz; }
    else if (nm == "obuf") { src = h->obuf; n = (int64_t)h->max_tok * h->HD() * h->esz; }
    else if (nm == "vit_feat" && h->cfg.with_vision) {   // SigLIP features (after the final LayerNorm, compute dtype) of the last pg_vision_encode
        const int64_t P = (h->cfg.vit_img / h->cfg.vit_patch) * (h->cfg.vit_img / h->cfg.vit_patch);
        src = h->vt; n = (int64_t)h->cfg.max_vision_images * P * h->cfg.vit_width * h->esz; }
    else { h->err = "pg_debug_read: unknown buffer " + nm; return PG_ERR_NAME; }
    if (n > max_bytes) n = max_bytes;
    (void)hipSetDevice(h->dev);
    if (hipMemcpyAsync(dst_dev, src, (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess) { h->err = "pg_debug_read: copy failed"; return PG_ERR_HIP; }
    return PG_OK;
}

int pg_op_rmsnorm(pg_handle h, float* x_dev, const float* partial_dev, int S, const void* w_dev, void* out_dev, int M, int H,
                  float eps, pg_stream s) { TuneGuard _tg(h);
    if (!h || !x_dev || !w_dev) return PG_ERR_ARG;
    (void)hipSetDevice(h->dev);
    if (h->bf) launch_rmsnorm<bf16>((hipStream_t)s, x_dev, partial_dev, S, (long)M * H, (const bf16*)w_dev, (bf16*)out_dev, M, H, eps);
    else launch_rmsnorm<float>((hipStream_t)s, x_dev, partial_dev, S, (long)M * H, (const float*)w_dev, (float*)out_dev, M, H, eps);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_op_gemm(pg_handle h, const void* a_dev, const void* w_dev, float* out_dev, int M, int N, int K, int force_kind,
               int* S_out, pg_stream s) { TuneGuard _tg(h);
    if (!h || !a_dev || !w_dev || !out_dev) return PG_ERR_ARG;
    (void)hipSetDevice(h->dev);
    int S = 1;
    if (h->bf && (force_kind == 1 || force_kind == 4 || (force_kind == 0 && M <= 128)) && K % 128 == 0) {
        S = skinny_pick_splits(N, K, M);
        bf16* wt = nullptr;
        if (force_kind == 4) {     // the decode layout: tiled copy of W built exactly like pg_finalize_weights does
            if ((N & 15)) { h->err = "pg_op_gemm: tiled mode needs N % 16 == 0"; return PG_ERR_ARG; }
            if (hipMalloc((void**)&wt, (size_t)N * K * 2) != hipSuccess) { h->err = "pg_op_gemm: hipMalloc failed"; return PG_ERR_HIP; }
            launch_tile_weights((hipStream_t)s, (const bf16*)w_dev, wt, N, K);
        }
        launch_gemm_skinny((hipStream_t)s, (const bf16*)a_dev, (const bf16*)w_dev, out_dev, M, N, K, S, wt);
        if (wt) { (void)hipStreamSynchronize((hipStream_t)s); (void)hipFree(wt); }
    } else {
        GemmA ga; ga.ptr = a_dev; ga.lda = K;
        GemmEpi e; e.out = out_dev; e.out_f32 = 1; e.ldc = N;
        if (h->bf) launch_gemm<bf16>((hipStream_t)s, ga, (const bf16*)w_dev, K, 0, e, M, N, K, 1);
        else launch_gemm<float>((hipStream_t)s, ga, (const float*)w_dev, K, 0, e, M, N, K, 1);
    }
    if (S_out) *S_out = S;
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_op_swiglu_gemm(pg_handle h, const void* a_dev, const void* wgu_dev, void* h_out_dev, int M, int I, int K, pg_stream s) { TuneGuard _tg(h);
    if (!h || !a_dev || !wgu_dev || !h_out_dev) return PG_ERR_ARG;
    if (!h->bf || K % 128 || (I & 7)) { h->err = "pg_op_swiglu_gemm: bf16 engine, K % 128 == 0, I % 8 == 0"; return PG_ERR_ARG; }
    (void)hipSetDevice(h->dev);
    bf16* wt = nullptr;
    if (hipMalloc((void**)&wt, (size_t)2 * I * K * 2) != hipSuccess) { h->err = "pg_op_swiglu_gemm: hipMalloc failed"; return PG_ERR_HIP; }
    launch_tile_weights((hipStream_t)s, (const bf16*)wgu_dev, wt, 2 * I, K);
    const bool ok = launch_gemm_skinny_swiglu((hipStream_t)s, (const bf16*)a_dev, (const bf16*)wgu_dev, (bf16*)h_out_dev, M, 2 * I, K, wt);
    (void)hipStreamSynchronize((hipStream_t)s);
    (void)hipFree(wt);
    if (!ok) { h->err = "pg_op_swiglu_gemm: no fused instantiation for this shape"; return PG_ERR_ARG; }
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_op_uniform(pg_handle h, const uint64_t* bits_dev, float* out_dev, int n, pg_stream s) {
    if (!h || !bits_dev || !out_dev) return PG_ERR_ARG;
    (void)hipSetDevice(h->dev);
    launch_uniform_from_bits((hipStream_t)s, bits_dev, out_dev, n);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_op_conv3x3(pg_handle h, const void* x_dev, const void* w_dev, const float* bias_dev, const void* residual_dev,
                  void* out_dev, int B, int Hi, int Wi, int Cin, int Cout, int up, int stride2, pg_stream s) { TuneGuard _tg(h);
    if (!h || !x_dev || !w_dev || !out_dev) return PG_ERR_ARG;
    (void)hipSetDevice(h->dev);
    ConvW cw; cw.w = (void*)w_dev; cw.b = (float*)bias_dev; cw.cin = Cin; cw.cout = Cout;
    if (h->bf) h->conv3<bf16>((hipStream_t)s, cw, (const bf16*)x_dev, out_dev, 0, residual_dev, 0, B, Hi, Wi, up, stride2);
    else h->conv3<float>((hipStream_t)s, cw, (const float*)x_dev, out_dev, 0, residual_dev, 0, B, Hi, Wi, up, stride2);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_op_groupnorm(pg_handle h, const void* x_dev, const float* gamma_dev, const float* beta_dev, void* out_dev, int B,
                    int HW, int C, int swish, pg_stream s) { TuneGuard _tg(h);
    if (!h || !x_dev || !out_dev) return PG_ERR_ARG;
    if (B > h->cfg.max_images || C > 1024) { h->err = "pg_op_groupnorm: B > max_images or C > 1024"; return PG_ERR_CAPACITY; }
    (void)hipSetDevice(h->dev);
    NormW n; n.g = (float*)gamma_dev; n.b = (float*)beta_dev; n.c = C;
    if (h->bf) h->gn<bf16>((hipStream_t)s, n, (const float*)x_dev, (bf16*)out_dev, B, HW, swish);
    else h->gn<float>((hipStream_t)s, n, (const float*)x_dev, (float*)out_dev, B, HW, swish);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}

}  // extern "C"

