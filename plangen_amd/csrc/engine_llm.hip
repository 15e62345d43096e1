// pg_engine: the language-model side of the path -- packed prefill, the 576-step CFG decode loop (System.sample_image,
// plangen_base.py:567-607), single steps behind language_model.model(...), gen_head, greedy text decode (System.x2t, :513-523).
#include "engine.h"

// =============================================================================== LLM
// C = a . W^T into fp32 split-K slabs ``part`` [S_last][M][N].
template <typename T>
void pg_engine::gemm_llm(hipStream_t s, const T* a, const T* W, int M, int N, int K, bool allow_skinny, const void* Wt) {
    slab_last = (long)M * N;
    if constexpr (std::is_same<T, bf16>::value) {
        if (allow_skinny && M <= 512 && K % 128 == 0) {
            const int S = skinny_pick_splits(N, K, M);
            if ((long)S * M * N <= part_elems) {
                launch_gemm_skinny(s, a, W, part, M, N, K, S, (const bf16*)Wt);
                S_last = S;
                return;
            }
        }
    }
    GemmA ga; ga.ptr = a; ga.lda = K;
    GemmEpi e; e.out = part; e.out_f32 = 1; e.ldc = N;
    launch_gemm<T>(s, ga, W, K, 0, e, M, N, K, 1);
    S_last = 1;
}

// Prefill form of the two projections that end a residual branch: x[M,N] (fp32 residual stream) += a . W^T in the GEMM's own epilogue
// (every element is read and written by the same lane), so the norm kernel that follows has no slab to fold in (S_last = 0): it reads x
// and writes xn only -- 220 MB less traffic per norm at the bench's 13.4 k packed tokens.  Same fp32 sum as the slab form (x + acc), bit for bit.
template <typename T>
void pg_engine::gemm_residual(hipStream_t s, const T* a, const T* W, int M, int N, int K) {
    GemmA ga; ga.ptr = a; ga.lda = K;
    GemmEpi e; e.out = x; e.out_f32 = 1; e.ldc = N; e.residual = x; e.res_f32 = 1;
    launch_gemm<T>(s, ga, W, K, 0, e, M, N, K, 1);
    S_last = 0; slab_last = (long)M * N;
}

// The layer stack on M token rows.  mode 0: decode (row m = batch row, slot len+n_dec);
// mode 1: prefill (packed prompt tokens).  Residual stream x fp32 [M,H]; ends with the final
// RMSNorm written to final_out (T).  Every GEMM leaves fp32 split-K slabs in ``part``; the
// next elementwise kernel folds the reduction in (deterministic, no atomics).
template <typename T>
void pg_engine::run_layers(hipStream_t s, int M, int mode, T* final_out, int32_t* advance) {
    const int Hh = H(), I = cfg.inter, HDm = HD();
    const bool sk = mode == 0;
    int S_pend = 0; long slab_pend = 0;
    const float scale = 1.0f / sqrtf(128.0f);
    tc_on = time_attn && mode == 0 && (n_dec_host % (time_stride > 0 ? time_stride : 1)) == 0;
    const double wb = (double)esz;                                  // weight bytes per element
    auto norm_bytes = [&](int S) { return (double)M * Hh * (4.0 * (S + 2) + wb); };   // x + S slabs read, x written (S > 0), xn written
    for (int li = 0; li < cfg.n_layers; ++li) {
        const Layer& ly = layers[li];
        tic(s); toc(s, TC_EMPTY, 0.0);          // an event pair around nothing: what the instrumentation itself adds to every timed launch
        tic(s);
        // Deferred 1/rms (round 6; decode at 65..128 rows, bf16): the norm launch becomes a barrier-free elementwise pass (residual + xw = bf16(x . w) +
        // 8 partial sums of squares per row) and the consumer GEMM scales its fp32 result by the row's 1/rms (kernels.h).  PG_F32 and every other
        // row count keep rmsnorm512_kernel; layer 0 of a step has no slabs to fold in (S_pend = 0) and keeps it too.
        bool d1 = false, d2 = false;
        int Sq = 0;
        if constexpr (std::is_same<T, bf16>::value) {
            if (sk && defer_norm && Hh == 2048 && S_pend > 0 && S_pend <= 8 && slab_pend <= 0x7fffffffL && ly.wqkv_t) {
                Sq = skinny_pick_splits(3 * HDm, Hh, M);
                d1 = deferred_norm_ok(M, 3 * HDm, Hh, Sq) && (long)Sq * M * 3 * HDm <= part_elems;
            }
            if (d1) launch_rmsnorm_defer(s, x, part, S_pend, slab_pend, (const bf16*)ly.ln1, (bf16*)xn, ssq_part, M, Hh);
        }
        if (!d1) launch_rmsnorm<T>(s, x, part, S_pend, slab_pend, (const T*)ly.ln1, (T*)xn, M, Hh, cfg.rms_eps);
        toc(s, TC_NORM, norm_bytes(S_pend));
        tic(s);
        bool qkv_fused = false;
        if constexpr (std::is_same<T, bf16>::value) {
            // prefill: RoPE(q), RoPE(k) and the KV-cache write in the QKV GEMM's own epilogue when the packed batch is big enough
            // for the 256x256 kernel; smaller batches keep GEMM -> fp32 q|k|v -> rope_kv_kernel on the un-interleaved weights
            if (mode == 1 && prefill_rope_epi && ly.wqkv_p) {
                GemmA ga; ga.ptr = xn; ga.lda = Hh;
                GemmEpi ge; ge.act = 3; ge.out = qbuf; ge.out_f32 = 0; ge.ldc = 3 * HDm;
                ge.rope.qbuf = qbuf; ge.rope.kc = kc(li); ge.rope.vc = vc(li); ge.rope.cos_t = cos_t; ge.rope.sin_t = sin_t;
                ge.rope.tok_row = d_tok_row; ge.rope.tok_j = d_tok_j; ge.rope.pos_off = d_pos_off;
                ge.rope.nh = cfg.n_heads; ge.rope.slots = slots; ge.rope.max_pos = max_pos;
                qkv_fused = gemm256_try(s, ga, (const bf16*)ly.wqkv_p, Hh, 0, ge, M, 3 * HDm, Hh, 1, 1, 0);
            }
        }
        if constexpr (std::is_same<T, bf16>::value) {
            if (d1) {
                d1 = launch_gemm_skinny_deferred(s, (const bf16*)xn, (const bf16*)ly.wqkv_t, part, M, 3 * HDm, Hh, Sq, ssq_part, cfg.rms_eps);
                if (d1) { S_last = Sq; slab_last = (long)M * 3 * HDm; qkv_fused = true; }
                else launch_rmsnorm<T>(s, x, part, 0, 0, (const T*)ly.ln1, (T*)xn, M, Hh, cfg.rms_eps);   // cannot happen (one predicate); the residual is already updated: normalise it the ordinary way
            }
        }
        if (!qkv_fused) gemm_llm<T>(s, (const T*)xn, (const T*)ly.wqkv, M, 3 * HDm, Hh, sk, ly.wqkv_t);
        toc(s, TC_QKV, 3.0 * HDm * Hh * wb);
        if (mode == 0 && skip_attn) {
            // nothing: the GEMM + norm phase alone (outputs are garbage by construction)
        } else if (mode == 0 && fuse_rope) {
            tic(s);
            launch_attn_decode_fused<T>(s, part, S_last, slab_last, (T*)obuf, (T*)kc(li), (T*)vc(li), cos_t, sin_t, seq(), M,
                                        cfg.n_heads, slots, max_pos, scale);
        } else {
            if (!qkv_fused)
                launch_rope_kv<T>(s, part, S_last, slab_last, (T*)qbuf, (T*)kc(li), (T*)vc(li), cos_t, sin_t, seq(), mode, M,
                                  cfg.n_heads, slots, max_pos);
            tic(s);
            bool done = false;
            if constexpr (std::is_same<T, bf16>::value) {
                if (mode == 1 && flash_prefill) {
                    launch_attn_prefill_flash(s, (const bf16*)qbuf, (bf16*)obuf, (const bf16*)kc(li), (const bf16*)vc(li), d_row_off, d_len,
                                              R, max_len_host, cfg.n_heads, slots, scale);
                    done = true;
                }
            }
            if (!done)
                launch_attn<T>(s, (const T*)qbuf, (T*)obuf, (const T*)kc(li), (const T*)vc(li), seq(), mode, M, cfg.n_heads, slots, scale);
        }
        if (tc_on && !(mode == 0 && skip_attn)) {
            double keys = shared_len;       // the shared uncond prompt is read from HBM once per launch
            for (int r = 0; r < R; ++r) keys += (double)(h_len[h_len_off + r] + n_dec_host + 1) - ((shared_len > 0 && (r & 1)) ? shared_len : 0);
            toc(s, TC_ATTN, keys * cfg.n_heads * 128 * 2 * (double)esz);
        }
        tic(s);
        if (!sk && prefill_res_epi) gemm_residual<T>(s, (const T*)obuf, (const T*)ly.wo, M, Hh, HDm);      // prefill: x += o . Wo^T in the GEMM's epilogue
        else gemm_llm<T>(s, (const T*)obuf, (const T*)ly.wo, M, Hh, HDm, sk, ly.wo_t);
        toc(s, TC_O, (double)Hh * HDm * wb);
        tic(s);
        if constexpr (std::is_same<T, bf16>::value) {
            if (sk && defer_norm && Hh == 2048 && S_last > 0 && S_last <= 8 && slab_last <= 0x7fffffffL && ly.wgu_t)
                d2 = deferred_norm_ok(M, 2 * I, Hh, 1) && (force_swiglu || skinny_pick_splits(2 * I, Hh, M) == 1);
            if (d2) launch_rmsnorm_defer(s, x, part, S_last, slab_last, (const bf16*)ly.ln2, (bf16*)xn, ssq_part, M, Hh);
        }
        if (!d2) launch_rmsnorm<T>(s, x, part, S_last, slab_last, (const T*)ly.ln2, (T*)xn, M, Hh, cfg.rms_eps);
        toc(s, TC_NORM, norm_bytes(S_last));
        bool fused = false;
        tic(s);
        if constexpr (std::is_same<T, bf16>::value) {
            if (d2) {
                fused = launch_gemm_skinny_swiglu_deferred(s, (const bf16*)xn, (const bf16*)ly.wgu_t, (bf16*)hbuf, M, 2 * I, Hh, ssq_part, cfg.rms_eps);
                if (!fused) launch_rmsnorm<T>(s, x, part, 0, 0, (const T*)ly.ln2, (T*)xn, M, Hh, cfg.rms_eps);   // cannot happen (one predicate); see the QKV site
            }
            // decode: SwiGLU gate fused into the gate|up GEMM epilogue (S = 1, no slab, no extra kernel)
            if (!fused && sk && M <= 512 && Hh % 128 == 0 && (force_swiglu || skinny_pick_splits(2 * I, Hh, M) == 1))   // fused wins at every M (B=4/8/16: -2..3 % loop time)
                fused = launch_gemm_skinny_swiglu(s, (const bf16*)xn, (const bf16*)ly.wgu, (bf16*)hbuf, M, 2 * I, Hh, (const bf16*)ly.wgu_t);
        }
        if constexpr (std::is_same<T, bf16>::value) {
            // prefill: SwiGLU in the 256x256 GEMM's epilogue (h written as bf16, no fp32 gate|up tensor, no extra pass)
            if (!fused && !sk && (I % 4) == 0) {
                GemmA ga; ga.ptr = xn; ga.lda = Hh;
                GemmEpi ge; ge.out = hbuf; ge.out_f32 = 0; ge.ldc = I; ge.act = 2;
                fused = gemm256_try(s, ga, (const bf16*)ly.wgu, Hh, 0, ge, M, 2 * I, Hh, 1, 1, 0);
            }
        }
        if (!fused) {
            gemm_llm<T>(s, (const T*)xn, (const T*)ly.wgu, M, 2 * I, Hh, sk, ly.wgu_t);
            launch_silu_mul<T>(s, part, S_last, slab_last, (T*)hbuf, M, I);
        }
        toc(s, TC_GU, 2.0 * I * Hh * wb);
        tic(s);
        if (!sk && prefill_res_epi) gemm_residual<T>(s, (const T*)hbuf, (const T*)ly.wd, M, Hh, I);         // prefill: x += h . Wd^T
        else gemm_llm<T>(s, (const T*)hbuf, (const T*)ly.wd, M, Hh, I, sk, ly.wd_t);
        toc(s, TC_DOWN, (double)Hh * I * wb);
        S_pend = S_last; slab_pend = slab_last;
    }
    tic(s);
    launch_rmsnorm<T>(s, x, part, S_pend, slab_pend, (const T*)norm_w, final_out, M, Hh, cfg.rms_eps, advance);
    toc(s, TC_NORM, norm_bytes(S_pend));
    tc_on = false;
}

int pg_engine::prefill(const int32_t* ids_dev, const void* emb_dev, int emb_dtype, const int32_t* pad_len, int R_, int L_,
                       int pmode, void* hidden_out, int hidden_dtype, hipStream_t s) {
    // one-shot: the hint describes the ids of THIS call only -- consumed before anything can fail, so that a rejected call
    // cannot leave it armed for the next batch
    const int hint = uncond_hint; uncond_hint = -1;
    if (!finalized) FAIL(PG_ERR_STATE, "pg_finalize_weights not called");
    if (R_ <= 0 || R_ > cfg.max_rows) FAIL(PG_ERR_CAPACITY, "rows %d > max_rows %d", R_, cfg.max_rows);
    if (L_ + cfg.max_new + 1 > max_pos) FAIL(PG_ERR_CAPACITY, "padded length %d too long for the RoPE table (%d)", L_, max_pos);
    if (!ids_dev && !emb_dev) FAIL(PG_ERR_ARG, "ids or embeds required");
    HIPCHK(hipSetDevice(dev));
    for (int r = 0; r < R_; ++r) {                 // validate every row BEFORE pad_len is used as an offset anywhere
        const int pad = pad_len[r];
        if (pad < 0 || pad >= L_) FAIL(PG_ERR_ARG, "row %d: pad_len %d not in [0,%d)", r, pad, L_);
        if (L_ - pad > cfg.max_prompt) FAIL(PG_ERR_CAPACITY, "row %d: %d prompt tokens > max_prompt %d", r, L_ - pad, cfg.max_prompt);
    }
    // Shared negative prompt (the uncond prompt is batch-constant for non-edit
    // data): when every odd row carries the same ids and padding, its prompt is prefilled and
    // its K/V stored ONCE (row 1); the other uncond rows alias it.  Verified per batch, never
    // assumed: the ids are compared ON THE DEVICE and one 4-byte flag comes back (the only
    // host<->device round trip of pg_prefill: the packed-token count, hence every GEMM shape of
    // the prefill, depends on the answer).  Only on the fused path (no per-position hidden output).
    shared_len = 0;
    if (share_uncond && fuse_rope && ids_dev && !hidden_out && pmode == 0 && R_ >= 4 && (R_ % 2) == 0) {
        bool same = true;
        for (int r = 3; r < R_ && same; r += 2) same = pad_len[r] == pad_len[1];
        if (same && hint == 1) shared_len = L_ - pad_len[1];      // the caller compared the ids on the host (its collate built them): no probe, no sync
        else if (same && hint != 0) {
            HIPCHK(hipMemsetAsync(d_flag, 0, 4, s));
            launch_rows_differ(s, ids_dev, L_, /*first*/ 3, /*stride*/ 2, /*ref row*/ 1, (R_ - 2) / 2, pad_len[1], d_flag);
            HIPCHK(hipMemcpyAsync(h_flag, d_flag, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (*h_flag == 0) shared_len = L_ - pad_len[1];
        }
    }
    // pinned staging is double-buffered: the copies of call n are still in flight while call n+1 fills
    // the other buffer; a buffer is reused only after the event recorded behind its copies has fired
    // (two calls back: in practice never waits), so pg_prefill itself does not synchronise the stream.
    stage_sel ^= 1;
    int32_t* const hs = h_stage2[stage_sel];
    if (stage_used[stage_sel]) HIPCHK(hipEventSynchronize(ev_stage[stage_sel]));
    int ntok = 0;
    h_len.assign(R_, 0);
    int32_t* s_len = hs; int32_t* s_off = hs + cfg.max_rows; int32_t* s_last = hs + 2 * cfg.max_rows;
    int32_t* s_roff = hs + 3 * cfg.max_rows; int32_t* s_ord = hs + 4 * cfg.max_rows; max_len_host = 0;
    int32_t* s_row = hs + 5 * cfg.max_rows; int32_t* s_j = s_row + max_tok; int32_t* s_src = s_j + max_tok;
    for (int r = 0; r < R_; ++r) {
        const int pad = pad_len[r];
        const int len = L_ - pad;
        h_len[r] = len; s_len[r] = len; s_off[r] = pmode == 0 ? pad : 0;
        if (shared_len > 0 && (r & 1) && r != 1) { s_last[r] = s_last[1]; s_roff[r] = -1; continue; }   // aliases row 1's prompt
        s_roff[r] = ntok; if (len > max_len_host) max_len_host = len;
        for (int j = 0; j < len; ++j) { s_row[ntok] = r; s_j[ntok] = j; s_src[ntok] = r * L_ + pad + j; ++ntok; }
        s_last[r] = ntok - 1;
    }
    R = R_; L = L_; Ntok = ntok; pos_mode = pmode; n_dec_host = 0;
    {   // longest-first row order for the decode attention launch (private keys per row)
        for (int r = 0; r < R_; ++r) s_ord[r] = r;
        std::stable_sort(s_ord, s_ord + R_, [&](int a, int b) {
            const int ka = h_len[a] - ((shared_len > 0 && (a & 1)) ? shared_len : 0), kb = h_len[b] - ((shared_len > 0 && (b & 1)) ? shared_len : 0);
            return ka > kb; });
        order_valid = true; order_rows = R_;
    }
    HIPCHK(hipEventRecord(ev_p0, s));
    HIPCHK(hipMemcpyAsync(d_row_order, s_ord, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_len, s_len, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_pos_off, s_off, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_last, s_last, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_row_off, s_roff, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_tok_row, s_row, (size_t)ntok * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_tok_j, s_j, (size_t)ntok * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_tok_src, s_src, (size_t)ntok * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(ev_stage[stage_sel], s));
    stage_used[stage_sel] = true;
    HIPCHK(hipMemsetAsync(d_ndec, 0, 64, s));
    HIPCHK(hipMemsetAsync(d_ndec2, 0, 64, s));
    if (ids_dev) launch_embed_gather(s, embed, ids_dev, d_tok_src, x, ntok, H(), cfg.vocab);
    else launch_rows_to_f32(s, emb_dev, emb_dtype == PG_BF16, d_tok_src, x, ntok, H());
    if (bf) run_layers<bf16>(s, ntok, 1, (bf16*)xn); else run_layers<float>(s, ntok, 1, (float*)xn);
    // last real token of every row -> hfin (input of gen_head / lm_head for the first sample)
    launch_copy_rows(s, xn, d_last, hfin, nullptr, R, (long)H() * esz);
    if (hidden_out) {
        const size_t ob = (size_t)R * L * H() * (hidden_dtype == PG_BF16 ? 2 : 4);
        HIPCHK(hipMemsetAsync(hidden_out, 0, ob, s));
        if (bf) launch_t_to_rows<bf16>(s, (const bf16*)xn, hidden_out, hidden_dtype == PG_BF16, d_tok_src, ntok, H());
        else launch_t_to_rows<float>(s, (const float*)xn, hidden_out, hidden_dtype == PG_BF16, d_tok_src, ntok, H());
    }
    HIPCHK(hipEventRecord(ev_p1, s));
    have_prefill_t = true;
    HIPCHK(hipGetLastError());
    prefilled = true;
    return PG_OK;
}

// gen_head: Linear+b -> GELU(erf) -> Linear (second bias folded into the consumer)
template <typename T>
void pg_engine::head_logits(hipStream_t s, const T* in, int M) {
    const int Hh = H(), G = cfg.gen_head_dim, V = cfg.img_vocab;
    gemm_llm<T>(s, in, (const T*)gh_w1, M, G, Hh, true, gh_w1_t);
    launch_bias_act<T>(s, part, S_last, slab_last, gh_b1, (T*)gh_mid, M, G, 1);
    gemm_llm<T>(s, (const T*)gh_mid, (const T*)gh_w2, M, V, G, true, gh_w2_t);
}

void pg_engine::forward_decode(hipStream_t s) {
    // the final RMSNorm launch also advances the device step counter
    if (bf) run_layers<bf16>(s, R, 0, (bf16*)hfin, d_ndec); else run_layers<float>(s, R, 0, (float*)hfin, d_ndec);
}

int pg_engine::decode_image(int T, float cfgw, float temp, uint64_t seed, const int32_t* force_tok,
                            const uint8_t* force_mask, int32_t* out_tok, float* logits_out, hipStream_t s) {
    if (!prefilled) FAIL(PG_ERR_STATE, "pg_decode_image_tokens before pg_prefill");
    if (R % 2) FAIL(PG_ERR_ARG, "CFG decode needs an even number of rows (got %d)", R);
    if (n_dec_host != 0) FAIL(PG_ERR_STATE, "decode loop needs a fresh prefill");
    if (T < 1 || T - 1 > cfg.max_new) FAIL(PG_ERR_CAPACITY, "T=%d exceeds max_new=%d", T, cfg.max_new);
    HIPCHK(hipSetDevice(dev));
    const int Rtot = R, B = R / 2;
    // Lanes: at large batch the rows are split into two independent chains on two streams so one
    // half's latency-bound GEMM / norm kernels overlap the other half's bandwidth-bound attention.
    // CFG pairs never straddle lanes; results are lane-independent (global image index in the RNG).
    // Measured on MI355X at B=64: two lanes are 6 % SLOWER (kernels of both streams each fill the chip, the
    // weights are read twice) -> one lane unless asked for (pg_set_option "lanes").
    int nl = lanes_opt > 0 ? lanes_opt : 1;
    if (Rtot % 4 || (long)decode_part_elems <= 0) nl = 1;
    const bool graph = use_graph && !time_attn && T > 2;
    struct LaneDef { int r0, nrows; float* part; int32_t* ndec; };
    LaneDef ld[2];
    ld[0] = {0, nl == 2 ? Rtot / 2 : Rtot, part, d_ndec};
    ld[1] = {Rtot / 2, Rtot / 2, part2, d_ndec2};
    // saved whole-batch views (a lane = pointer rebasing of the row-indexed buffers)
    float* const x0 = x; void* const xn0 = xn; void* const q0 = qbuf; void* const o0 = obuf; void* const hb0 = hbuf;
    void* const hf0 = hfin; void* const gm0 = gh_mid; float* const p0 = part; int32_t* const len0 = d_len;
    int32_t* const po0 = d_pos_off; int32_t* const nd0 = d_ndec; float* const pv0 = cfg_pv; int* const pi0 = cfg_pi;
    const int Hh = H(), HDm = HD(), G = cfg.gen_head_dim, I = cfg.inter;
    auto enter = [&](const LaneDef& L) {
        const size_t r = (size_t)L.r0;
        x = x0 + r * Hh; xn = (char*)xn0 + r * Hh * esz; qbuf = (char*)q0 + r * HDm * esz; obuf = (char*)o0 + r * HDm * esz;
        hbuf = (char*)hb0 + r * I * esz; hfin = (char*)hf0 + r * Hh * esz; gh_mid = (char*)gm0 + r * G * esz;
        part = L.part; d_len = len0 + r; d_pos_off = po0 + r; d_ndec = L.ndec; cfg_pv = pv0 + r * 8; cfg_pi = pi0 + r * 8;
        kv_row_off = r * cfg.n_heads * (size_t)slots * 128 * esz; shared_row = 1 - L.r0; h_len_off = L.r0; R = L.nrows;
    };
    auto leave = [&]() {
        x = x0; xn = xn0; qbuf = q0; obuf = o0; hbuf = hb0; hfin = hf0; gh_mid = gm0; part = p0; d_len = len0; d_pos_off = po0;
        d_ndec = nd0; cfg_pv = pv0; cfg_pi = pi0; kv_row_off = 0; shared_row = 1; h_len_off = 0; R = Rtot;
    };
    if (force_mask && !force_tok) FAIL(PG_ERR_ARG, "force_mask needs force_tok");
    SampleArgs sa{};
    sa.bias = gh_b2; sa.V = cfg.img_vocab; sa.p = d_sparams;
    sa.force_tok = d_force_tok; sa.force_mask = d_force_mask; sa.out_tok = d_out_tok; sa.logits_out = logits_out;
    sa.embed_table = gen_table; sa.H = Hh; sa.B_total = B;
    auto sample = [&](hipStream_t st, const LaneDef& L) {
        tc_on = time_attn && (n_dec_host % (time_stride > 0 ? time_stride : 1)) == 0;
        tic(st);
        if (bf) head_logits<bf16>(st, (const bf16*)hfin, R); else head_logits<float>(st, (const float*)hfin, R);
        toc(st, TC_HEAD, ((double)G * Hh + (double)cfg.img_vocab * G) * (double)esz);
        sa.logits_partial = part; sa.S = S_last; sa.slab = slab_last; sa.x = x; sa.n_dec = d_ndec; sa.b_off = L.r0 / 2;
        tic(st);
        launch_cfg_sample(st, sa, R / 2, cfg_pv, cfg_pi);
        toc(st, TC_SAMPLE, (double)S_last * R * cfg.img_vocab * 4.0);
        tc_on = false;
    };
    if (time_attn) { tc_used = 0; tc_meta.clear(); }
    hipStream_t ws = s;
    if (graph || nl == 2) {      // graphs cannot be captured on the legacy default stream: hop to our own
        HIPCHK(hipEventRecord(ev_in, s));
        HIPCHK(hipStreamWaitEvent(istream, ev_in, 0));
        ws = istream;
    }
    // one loop iteration for every lane: sample token i from hfin (step index = n_dec), then
    // (with_forward) run the stack on its embedding (appends KV slot len+n_dec) and advance n_dec.
    // In time_attn mode the lanes run back to back on one stream so the per-launch events are clean.
    const bool two_streams = nl == 2 && !time_attn;
    auto iteration = [&](bool with_forward) -> int {
        if (two_streams) { HIPCHK(hipEventRecord(ev_fork, ws)); HIPCHK(hipStreamWaitEvent(istream2, ev_fork, 0)); }
        for (int li = 0; li < nl; ++li) {
            hipStream_t st = (two_streams && li == 1) ? istream2 : ws;
            enter(ld[li]);
            sample(st, ld[li]);
            if (with_forward) forward_decode(st);
            leave();
        }
        if (two_streams) { HIPCHK(hipEventRecord(ev_join, istream2)); HIPCHK(hipStreamWaitEvent(ws, ev_join, 0)); }
        return PG_OK;
    };
    HIPCHK(hipEventRecord(ev_t0, ws));
    {   // per-call parameters and the caller's forcing tensors -> library-owned device memory (what the graph reads)
        SampleParams sp{}; sp.cfg_weight = cfgw; sp.temperature = temp; sp.seed = seed; sp.T = T;
        sp.has_force = force_tok != nullptr; sp.has_mask = force_mask != nullptr; sp.img_off = rng_image_offset;
        launch_set_sample_params(ws, d_sparams, sp);
        if (force_tok) HIPCHK(hipMemcpyAsync(d_force_tok, force_tok, (size_t)B * T * 4, hipMemcpyDeviceToDevice, ws));
        if (force_mask) HIPCHK(hipMemcpyAsync(d_force_mask, force_mask, (size_t)B * T, hipMemcpyDeviceToDevice, ws));
    }
    TRY(iteration(T > 1));
    if (T > 1) n_dec_host++;
    int i = 1;
    if (graph) {
        // shapes and kernel selection only: seeds, temperatures, T and the caller's buffers reach the kernels through
        // device memory, so a bench / serving loop replays ONE instantiated graph across calls
        std::vector<int64_t> key = {Rtot, (int64_t)bf, (int64_t)logits_out, (int64_t)shared_len, (int64_t)fuse_rope, (int64_t)nl,
                                    (int64_t)(lpt_order && order_valid), (int64_t)tune_epoch, (int64_t)skip_attn};
        if (!gexec || key != gkey) {
            if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(ws, hipStreamCaptureModeThreadLocal));
            const int rc = iteration(true);
            hipError_t ce = hipStreamEndCapture(ws, &g);
            if (rc != PG_OK) return rc;
            HIPCHK(ce);
            HIPCHK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            gkey = key;
        }
        for (; i < T - 1; ++i) { HIPCHK(hipGraphLaunch(gexec, ws)); n_dec_host++; }
    } else {
        for (; i < T - 1; ++i) { TRY(iteration(true)); n_dec_host++; }
    }
    if (T > 1) TRY(iteration(false));
    HIPCHK(hipMemcpyAsync(out_tok, d_out_tok, (size_t)B * T * 4, hipMemcpyDeviceToDevice, ws));
    HIPCHK(hipEventRecord(ev_t1, ws));
    if (ws != s) {
        HIPCHK(hipEventRecord(ev_out, ws));
        HIPCHK(hipStreamWaitEvent(s, ev_out, 0));
    }
    have_decode_t = true;
    HIPCHK(hipGetLastError());
    return PG_OK;
}

int pg_engine::step(const void* emb, int emb_dtype, void* hidden_out, int hidden_dtype, hipStream_t s) {
    if (!prefilled) FAIL(PG_ERR_STATE, "pg_step before pg_prefill");
    if (n_dec_host + 1 > cfg.max_new) FAIL(PG_ERR_CAPACITY, "decode capacity max_new=%d exhausted", cfg.max_new);
    HIPCHK(hipSetDevice(dev));
    launch_rows_to_f32(s, emb, emb_dtype == PG_BF16, nullptr, x, R, H());
    forward_decode(s);
    n_dec_host++;
    if (hidden_out) {
        if (bf) launch_t_to_rows<bf16>(s, (const bf16*)hfin, hidden_out, hidden_dtype == PG_BF16, nullptr, R, H());
        else launch_t_to_rows<float>(s, (const float*)hfin, hidden_out, hidden_dtype == PG_BF16, nullptr, R, H());
    }
    HIPCHK(hipGetLastError());
    return PG_OK;
}

int pg_engine::gen_head(const void* h_dev, int h_dtype, float* logits, int R_, hipStream_t s) {
    if (!finalized) FAIL(PG_ERR_STATE, "pg_finalize_weights not called");
    if (R_ < 1 || R_ > cfg.max_rows) FAIL(PG_ERR_CAPACITY, "rows %d > max_rows %d", R_, cfg.max_rows);
    HIPCHK(hipSetDevice(dev));
    // bring h into the compute dtype
    if (bf) {
        if (h_dtype == PG_BF16) HIPCHK(hipMemcpyAsync(gh_in, h_dev, (size_t)R_ * H() * 2, hipMemcpyDeviceToDevice, s));
        else launch_t_to_rows<float>(s, (const float*)h_dev, gh_in, 1, nullptr, R_, H());
        head_logits<bf16>(s, (const bf16*)gh_in, R_);
    } else {
        launch_rows_to_f32(s, h_dev, h_dtype == PG_BF16, nullptr, (float*)gh_in, R_, H());
        head_logits<float>(s, (const float*)gh_in, R_);
    }
    launch_bias_f32(s, part, S_last, slab_last, gh_b2, logits, R_, cfg.img_vocab);
    HIPCHK(hipGetLastError());
    return PG_OK;
}

int pg_engine::text_greedy(int max_new, int min_new, int eos, int64_t* out, int* out_len, hipStream_t s) {
    if (!prefilled) FAIL(PG_ERR_STATE, "pg_generate_text_greedy before pg_prefill");
    if (!cfg.with_lm_head) FAIL(PG_ERR_STATE, "engine created without lm_head");
    if (n_dec_host != 0) FAIL(PG_ERR_STATE, "text decode needs a fresh prefill");
    if (max_new < 1 || max_new > cfg.max_new || max_new > 1000) FAIL(PG_ERR_CAPACITY, "max_new=%d exceeds capacity %d", max_new, cfg.max_new);
    HIPCHK(hipSetDevice(dev));
    const int B = R;
    std::vector<int32_t> ones(B, 1);
    HIPCHK(hipMemcpyAsync(d_unf, ones.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_anyunf, 0, 1024 * 4, s));
    HIPCHK(hipStreamSynchronize(s));
    TextArgs ta{};
    ta.V = cfg.vocab; ta.p = d_tparams; ta.out = d_text_out; ta.unfinished = d_unf;
    ta.any_unfinished = d_anyunf; ta.embed_table = embed; ta.x = x; ta.H = H(); ta.n_dec = d_ndec;
    std::vector<int32_t> flags(1024);
    int checked = 0, done_len = -1;
    // one step = lm_head GEMM -> argmax / EOS bookkeeping (device step counter) -> the stack on the new
    // token's embedding.  Like the image loop it is captured once and replayed; every 8th step the
    // any-unfinished flags come back to the host (HF generate stops when every row has emitted EOS).
    hipStream_t ws = s;
    if (use_graph) {
        HIPCHK(hipEventRecord(ev_in, s));
        HIPCHK(hipStreamWaitEvent(istream, ev_in, 0));
        ws = istream;
    }
    { TextParams tp{}; tp.eos = eos; tp.min_new = min_new; tp.max_new = max_new; launch_set_text_params(ws, d_tparams, tp); }
    auto iteration = [&](bool with_forward) {
        if (bf) gemm_llm<bf16>(ws, (const bf16*)hfin, (const bf16*)lm_head, B, cfg.vocab, H(), true, lm_head_t);
        else gemm_llm<float>(ws, (const float*)hfin, (const float*)lm_head, B, cfg.vocab, H(), true);
        ta.logits_partial = part; ta.S = S_last; ta.slab = slab_last;
        launch_text_argmax(ws, ta, B, cfg_pv, cfg_pi);
        if (with_forward) forward_decode(ws);
    };
    for (int step_i = 0; step_i < max_new; ++step_i) {
        const bool last = step_i == max_new - 1;
        if (use_graph && !last && step_i > 0) {
            std::vector<int64_t> key = {R, (int64_t)bf, (int64_t)fuse_rope, (int64_t)shared_len, (int64_t)(lpt_order && order_valid), (int64_t)tune_epoch};
            if (!gexec_txt || key != gkey_txt) {
                if (gexec_txt) { (void)hipGraphExecDestroy(gexec_txt); gexec_txt = nullptr; }
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(ws, hipStreamCaptureModeThreadLocal));
                iteration(true);
                HIPCHK(hipStreamEndCapture(ws, &g));
                HIPCHK(hipGraphInstantiate(&gexec_txt, g, nullptr, nullptr, 0));
                (void)hipGraphDestroy(g);
                gkey_txt = key;
            }
            HIPCHK(hipGraphLaunch(gexec_txt, ws));
        } else {
            iteration(!last);
        }
        if (!last) n_dec_host++;
        if (last || (step_i & 7) == 7) {
            HIPCHK(hipMemcpyAsync(flags.data(), d_anyunf, 1024 * 4, hipMemcpyDeviceToHost, ws));
            HIPCHK(hipStreamSynchronize(ws));
            for (; checked <= step_i; ++checked)
                if (flags[(checked + 1) & 1023] == 0) { done_len = checked + 1; break; }
            if (done_len >= 0) break;
        }
    }
    if (done_len < 0) done_len = max_new;
    // only the columns this call produced; the caller's buffer keeps its own fill beyond them
    HIPCHK(hipMemcpy2DAsync(out, (size_t)max_new * 8, d_text_out, (size_t)max_new * 8, (size_t)done_len * 8, B, hipMemcpyDeviceToDevice, ws));
    if (ws != s) {
        HIPCHK(hipEventRecord(ev_out, ws));
        HIPCHK(hipStreamWaitEvent(s, ev_out, 0));
    }
    if (out_len) *out_len = done_len;
    HIPCHK(hipGetLastError());
    return PG_OK;
}

