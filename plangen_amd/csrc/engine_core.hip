// pg_engine: allocation of weights / KV cache / workspaces (create), weight loading and conversion to the kernel layouts (finalize).
#include "engine.h"

// =============================================================================== create
void pg_engine::add_slot(const std::string& name, void* dst, SlotKind k, long n, int a, int b, int c) {
    Slot s; s.dst = dst; s.kind = k; s.n = n; s.a = a; s.b = b; s.c = c;
    slots_map[name] = s;
}
int pg_engine::alloc_conv(const std::string& name, ConvW& cw, int cout, int cin, int k) {
    cw.cin = cin; cw.cout = cout; cw.k = k;
    TRY(dalloc(&cw.w, (size_t)cout * cin * k * k * esz));
    TRY(dalloc(&cw.b, (size_t)cout * 4));
    add_slot(name + ".weight", cw.w, k == 1 ? K_T : K_CONV, (long)cout * cin * k * k, cout, cin, k * k);
    slot_shape(name + ".weight", {cout, cin, k, k});
    add_slot(name + ".bias", cw.b, K_F32, cout);
    return PG_OK;
}
int pg_engine::alloc_norm(const std::string& name, NormW& nw, int c) {
    nw.c = c;
    TRY(dalloc(&nw.g, (size_t)c * 4));
    TRY(dalloc(&nw.b, (size_t)c * 4));
    add_slot(name + ".weight", nw.g, K_F32, c);
    add_slot(name + ".bias", nw.b, K_F32, c);
    return PG_OK;
}
int pg_engine::alloc_res(const std::string& name, ResBlockW& r, int cin, int cout) {
    TRY(alloc_norm(name + ".norm1", r.n1, cin));
    TRY(alloc_conv(name + ".conv1", r.c1, cout, cin, 3));
    TRY(alloc_norm(name + ".norm2", r.n2, cout));
    TRY(alloc_conv(name + ".conv2", r.c2, cout, cout, 3));
    r.has_nin = cin != cout;
    if (r.has_nin) TRY(alloc_conv(name + ".nin_shortcut", r.nin, cout, cin, 1));
    return PG_OK;
}
int pg_engine::alloc_attn(const std::string& name, AttnW& a, int c) {
    TRY(alloc_norm(name + ".norm", a.n, c));
    TRY(alloc_conv(name + ".q", a.q, c, c, 1));
    TRY(alloc_conv(name + ".k", a.k, c, c, 1));
    TRY(alloc_conv(name + ".v", a.v, c, c, 1));
    TRY(alloc_conv(name + ".proj_out", a.p, c, c, 1));
    return PG_OK;
}

// Decoder / Encoder structure: vq_model.py:127-187 / :48-103.
int pg_engine::build_vq() {
    const int nres = cfg.vq_levels, ch = cfg.vq_ch;
    const std::string V = "gen_vision_model.";
    {
        const std::string D = V + "decoder.";
        int block_in = ch * cfg.vq_ch_mult[nres - 1];
        TRY(alloc_conv(D + "conv_in", dec.conv_in, block_in, cfg.vq_z, 3));
        TRY(alloc_res(D + "mid.0", dec.mid0, block_in, block_in));
        TRY(alloc_attn(D + "mid.1", dec.mid1, block_in));
        TRY(alloc_res(D + "mid.2", dec.mid2, block_in, block_in));
        dec.levels.resize(nres);
        for (int bi = 0; bi < nres; ++bi) {
            const int i_level = nres - 1 - bi;
            const int block_out = ch * cfg.vq_ch_mult[i_level];
            VqLevel& lv = dec.levels[bi];
            lv.res.resize(cfg.vq_res_blocks + 1);
            if (i_level == nres - 1) lv.attn.resize(cfg.vq_res_blocks + 1);
            const std::string p = D + "conv_blocks." + std::to_string(bi);
            for (int j = 0; j < cfg.vq_res_blocks + 1; ++j) {
                TRY(alloc_res(p + ".res." + std::to_string(j), lv.res[j], block_in, block_out));
                block_in = block_out;
                if (i_level == nres - 1) TRY(alloc_attn(p + ".attn." + std::to_string(j), lv.attn[j], block_in));
            }
            if (i_level != 0) {
                lv.has_resample = true;
                TRY(alloc_conv(p + ".upsample.conv", lv.resample, block_in, block_in, 3));
            }
        }
        TRY(alloc_norm(D + "norm_out", dec.norm_out, block_in));
        TRY(alloc_conv(D + "conv_out", dec.conv_out, 3, block_in, 3));
    }
    if (cfg.with_vq_encoder) {
        const std::string E = V + "encoder.";
        TRY(dalloc(&enc_in_w, (size_t)ch * 27 * 4));
        TRY(dalloc(&enc_in_b, (size_t)ch * 4));
        add_slot(E + "conv_in.weight", enc_in_w, K_F32, (long)ch * 27);
        slot_shape(E + "conv_in.weight", {ch, 3, 3, 3});
        add_slot(E + "conv_in.bias", enc_in_b, K_F32, ch);
        enc.levels.resize(nres);
        int b_in = ch;
        for (int lvl = 0; lvl < nres; ++lvl) {
            const int b_out = ch * cfg.vq_ch_mult[lvl];
            VqLevel& lv = enc.levels[lvl];
            lv.res.resize(cfg.vq_res_blocks);
            if (lvl == nres - 1) lv.attn.resize(cfg.vq_res_blocks);
            const std::string p = E + "conv_blocks." + std::to_string(lvl);
            for (int j = 0; j < cfg.vq_res_blocks; ++j) {
                TRY(alloc_res(p + ".res." + std::to_string(j), lv.res[j], b_in, b_out));
                b_in = b_out;
                if (lvl == nres - 1) TRY(alloc_attn(p + ".attn." + std::to_string(j), lv.attn[j], b_in));
            }
            if (lvl != nres - 1) {
                lv.has_resample = true;
                TRY(alloc_conv(p + ".downsample.conv", lv.resample, b_in, b_in, 3));
            }
        }
        TRY(alloc_res(E + "mid.0", enc.mid0, b_in, b_in));
        TRY(alloc_attn(E + "mid.1", enc.mid1, b_in));
        TRY(alloc_res(E + "mid.2", enc.mid2, b_in, b_in));
        TRY(alloc_norm(E + "norm_out", enc.norm_out, b_in));
        TRY(alloc_conv(E + "conv_out", enc.conv_out, cfg.vq_z, b_in, 3));
        TRY(dalloc(&qc_w, (size_t)cfg.img_dim * cfg.vq_z * esz));
        TRY(dalloc(&qc_b, (size_t)cfg.img_dim * 4));
        add_slot(V + "quant_conv.weight", qc_w, K_T, (long)cfg.img_dim * cfg.vq_z);
        slot_shape(V + "quant_conv.weight", {cfg.img_dim, cfg.vq_z, 1, 1});
        add_slot(V + "quant_conv.bias", qc_b, K_F32, cfg.img_dim);
    }
    return PG_OK;
}

int pg_engine::alloc_lin(const std::string& name, LinW& l, int out, int in) {
    l.out = out; l.in = in;
    TRY(dalloc(&l.w, (size_t)out * in * esz));
    TRY(dalloc(&l.b, (size_t)out * 4));
    add_slot(name + ".weight", l.w, K_T, (long)out * in);
    slot_shape(name + ".weight", {out, in});
    add_slot(name + ".bias", l.b, K_F32, out);
    return PG_OK;
}
// CLIPVisionTower(siglip_large_patch16_384) + aligner: clip_encoder.py:30-122, siglip_vit.py:262-572,
// modeling_vlm.py:196-202.  State-dict names as in the Janus-Pro checkpoints.
int pg_engine::build_vision() {
    const int C = cfg.vit_width, ps = cfg.vit_patch, P = (cfg.vit_img / ps) * (cfg.vit_img / ps), Hh = H();
    if (cfg.vit_heads < 1 || C != cfg.vit_heads * 64) FAIL(PG_ERR_ARG, "SigLIP head_dim must be 64 (width %d, heads %d)", C, cfg.vit_heads);
    if (C % 64 || cfg.vit_mlp % 64 || (3 * ps * ps) % 16) FAIL(PG_ERR_ARG, "vit_width / vit_mlp must be multiples of 64");
    if (bf && (P % 64 || (3 * ps * ps) % 64)) FAIL(PG_ERR_ARG, "bf16 mode needs patch count and 3*patch^2 to be multiples of 64");
    const std::string VT = "vision_model.vision_tower.";
    TRY(alloc_lin(VT + "patch_embed.proj", vit_patch, C, 3 * ps * ps));
    slot_shape(VT + "patch_embed.proj.weight", {C, 3, ps, ps});          // a Conv2d weight in the checkpoint
    TRY(dalloc(&vit_pos, (size_t)P * C * 4));
    add_slot(VT + "pos_embed", vit_pos, K_F32, (long)P * C);
    vit_blocks.resize(cfg.vit_layers);
    for (int i = 0; i < cfg.vit_layers; ++i) {
        const std::string b = VT + "blocks." + std::to_string(i) + ".";
        VitBlockW& w = vit_blocks[i];
        TRY(alloc_norm(b + "norm1", w.n1, C));
        TRY(alloc_lin(b + "attn.qkv", w.qkv, 3 * C, C));
        TRY(alloc_lin(b + "attn.proj", w.proj, C, C));
        TRY(alloc_norm(b + "norm2", w.n2, C));
        TRY(alloc_lin(b + "mlp.fc1", w.fc1, cfg.vit_mlp, C));
        TRY(alloc_lin(b + "mlp.fc2", w.fc2, C, cfg.vit_mlp));
    }
    TRY(alloc_norm(VT + "norm", vit_norm, C));
    TRY(alloc_lin("aligner.layers.0", al0, Hh, C));
    TRY(alloc_lin("aligner.layers.2", al2, Hh, Hh));
    const long nb = cfg.max_vision_images, M = nb * P;
    TRY(dalloc(&vx, (size_t)M * C * 4));
    TRY(dalloc(&vt, (size_t)M * (C > 3 * ps * ps ? C : 3 * ps * ps) * esz));
    TRY(dalloc(&vqk, (size_t)M * 2 * C * esz));
    TRY(dalloc(&vvt, (size_t)M * C * esz));
    TRY(dalloc(&vo, (size_t)M * C * esz));
    TRY(dalloc(&vh, (size_t)M * cfg.vit_mlp * esz));
    TRY(dalloc(&val, (size_t)M * Hh * esz));
    TRY(dalloc(&vscore, (size_t)nb * cfg.vit_heads * P * P * 4));
    TRY(dalloc(&vp, (size_t)nb * cfg.vit_heads * P * P * esz));
    return PG_OK;
}

int pg_engine::create() {
    bf = cfg.compute_dtype == PG_BF16;
    esz = bf ? 2 : 4;
    if (cfg.head_dim != 128) FAIL(PG_ERR_ARG, "head_dim must be 128 (got %d)", cfg.head_dim);
    if (cfg.hidden % 128 || cfg.inter % 128 || cfg.gen_head_dim % 128)
        FAIL(PG_ERR_ARG, "hidden/inter/gen_head_dim must be multiples of 128");
    if (cfg.vocab < 1 || cfg.img_vocab % 16) FAIL(PG_ERR_ARG, "img_vocab must be a multiple of 16");
    if (cfg.img_dim > 8 || cfg.img_dim < 1) FAIL(PG_ERR_ARG, "img_dim must be in [1,8]");
    if (cfg.vq_levels < 1 || cfg.vq_levels > PG_MAX_VQ_LEVELS) FAIL(PG_ERR_ARG, "vq_levels");
    if (cfg.vq_ch % 64 || cfg.vq_z % 64) FAIL(PG_ERR_ARG, "vq_ch / vq_z must be multiples of 64");
    if (bf && (cfg.grid * cfg.grid) % 64) FAIL(PG_ERR_ARG, "bf16 mode needs grid^2 %% 64 == 0 (AttnBlock GEMM K)");
    if (cfg.max_rows < 1 || cfg.max_prompt < 1 || cfg.max_new < 1 || cfg.max_images < 1) FAIL(PG_ERR_ARG, "capacities must be >= 1");
    if (cfg.max_new > 1000 && cfg.with_lm_head) FAIL(PG_ERR_ARG, "max_new <= 1000 with lm_head");
    HIPCHK(hipSetDevice(dev));
    const int Hh = H(), I = cfg.inter, HDm = HD();
    const std::string LM = "language_model.model.";
    TRY(dalloc(&embed, (size_t)cfg.vocab * Hh * 4));
    add_slot(LM + "embed_tokens.weight", embed, K_F32, (long)cfg.vocab * Hh);
    slot_shape(LM + "embed_tokens.weight", {cfg.vocab, Hh});
    layers.resize(cfg.n_layers);
    for (int i = 0; i < cfg.n_layers; ++i) {
        Layer& ly = layers[i];
        TRY(dalloc(&ly.wqkv, (size_t)3 * HDm * Hh * esz));
        TRY(dalloc(&ly.wo, (size_t)Hh * HDm * esz));
        TRY(dalloc(&ly.wgu, (size_t)2 * I * Hh * esz));
        TRY(dalloc(&ly.wd, (size_t)Hh * I * esz));
        TRY(dalloc(&ly.ln1, (size_t)Hh * esz));
        TRY(dalloc(&ly.ln2, (size_t)Hh * esz));
        const std::string p = LM + "layers." + std::to_string(i) + ".";
        add_slot(p + "self_attn.q_proj.weight", ly.wqkv, K_T, (long)HDm * Hh);
        add_slot(p + "self_attn.k_proj.weight", (char*)ly.wqkv + (size_t)HDm * Hh * esz, K_T, (long)HDm * Hh);
        add_slot(p + "self_attn.v_proj.weight", (char*)ly.wqkv + (size_t)2 * HDm * Hh * esz, K_T, (long)HDm * Hh);
        add_slot(p + "self_attn.o_proj.weight", ly.wo, K_T, (long)Hh * HDm);
        add_slot(p + "mlp.gate_proj.weight", ly.wgu, K_IL16_G, (long)I * Hh, I, Hh);
        add_slot(p + "mlp.up_proj.weight", ly.wgu, K_IL16_U, (long)I * Hh, I, Hh);
        add_slot(p + "mlp.down_proj.weight", ly.wd, K_T, (long)Hh * I);
        for (const char* nm : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight"}) slot_shape(p + nm, {HDm, Hh});
        slot_shape(p + "self_attn.o_proj.weight", {Hh, HDm});
        slot_shape(p + "mlp.gate_proj.weight", {I, Hh}); slot_shape(p + "mlp.up_proj.weight", {I, Hh});
        slot_shape(p + "mlp.down_proj.weight", {Hh, I});
        add_slot(p + "input_layernorm.weight", ly.ln1, K_T, Hh);
        add_slot(p + "post_attention_layernorm.weight", ly.ln2, K_T, Hh);
    }
    TRY(dalloc(&norm_w, (size_t)Hh * esz));
    add_slot(LM + "norm.weight", norm_w, K_T, Hh);
    if (cfg.with_lm_head) {
        TRY(dalloc(&lm_head, (size_t)cfg.vocab * Hh * esz));
        add_slot("language_model.lm_head.weight", lm_head, K_T, (long)cfg.vocab * Hh);
        slot_shape("language_model.lm_head.weight", {cfg.vocab, Hh});
    }
    const int G = cfg.gen_head_dim, V = cfg.img_vocab, Dm = cfg.img_dim;
    TRY(dalloc(&gh_w1, (size_t)G * Hh * esz));
    TRY(dalloc(&gh_b1, (size_t)G * 4));
    TRY(dalloc(&gh_w2, (size_t)V * G * esz));
    TRY(dalloc(&gh_b2, (size_t)V * 4));
    add_slot("gen_head.output_mlp_projector.weight", gh_w1, K_T, (long)G * Hh);
    add_slot("gen_head.output_mlp_projector.bias", gh_b1, K_F32, G);
    add_slot("gen_head.vision_head.weight", gh_w2, K_T, (long)V * G);
    add_slot("gen_head.vision_head.bias", gh_b2, K_F32, V);
    slot_shape("gen_head.output_mlp_projector.weight", {G, Hh}); slot_shape("gen_head.vision_head.weight", {V, G});
    TRY(dalloc(&ge_w, (size_t)V * Dm * 4));
    TRY(dalloc(&al_w0, (size_t)Hh * Dm * 4));
    TRY(dalloc(&al_b0, (size_t)Hh * 4));
    TRY(dalloc(&al_w2, (size_t)Hh * Hh * 4));
    TRY(dalloc(&al_b2, (size_t)Hh * 4));
    add_slot("gen_embed.weight", ge_w, K_F32, (long)V * Dm);
    add_slot("gen_aligner.layers.0.weight", al_w0, K_F32, (long)Hh * Dm);
    add_slot("gen_aligner.layers.0.bias", al_b0, K_F32, Hh);
    add_slot("gen_aligner.layers.2.weight", al_w2, K_F32, (long)Hh * Hh);
    add_slot("gen_aligner.layers.2.bias", al_b2, K_F32, Hh);
    slot_shape("gen_embed.weight", {V, Dm}); slot_shape("gen_aligner.layers.0.weight", {Hh, Dm}); slot_shape("gen_aligner.layers.2.weight", {Hh, Hh});
    TRY(dalloc(&gen_table, (size_t)V * Hh * 4));
    TRY(dalloc(&codebook, (size_t)V * Dm * 4));
    TRY(dalloc(&codebook_n, (size_t)V * Dm * 4));
    TRY(dalloc(&pq_w, (size_t)cfg.vq_z * Dm * 4));
    TRY(dalloc(&pq_b, (size_t)cfg.vq_z * 4));
    TRY(dalloc(&pq_table, (size_t)V * cfg.vq_z * esz));
    add_slot("gen_vision_model.quantize.embedding.weight", codebook, K_F32, (long)V * Dm);
    add_slot("gen_vision_model.post_quant_conv.weight", pq_w, K_F32, (long)cfg.vq_z * Dm);
    add_slot("gen_vision_model.post_quant_conv.bias", pq_b, K_F32, cfg.vq_z);
    slot_shape("gen_vision_model.quantize.embedding.weight", {V, Dm}); slot_shape("gen_vision_model.post_quant_conv.weight", {cfg.vq_z, Dm, 1, 1});
    TRY(build_vq());
    if (cfg.with_vision) TRY(build_vision());

    // ---- state + workspaces
    slots = cfg.max_prompt + cfg.max_new;
    max_pos = 2 * cfg.max_prompt + cfg.max_new + 64;
    max_tok = (long)cfg.max_rows * cfg.max_prompt;
    TRY(dalloc(&kv, (size_t)cfg.n_layers * 2 * kv_layer_elems() * esz, false));
    TRY(dalloc(&cos_t, (size_t)max_pos * 64 * 4));
    TRY(dalloc(&sin_t, (size_t)max_pos * 64 * 4));
    TRY(dalloc(&zeros, 1024));
    HIPCHK(hipMemset(zeros, 0, 1024));
    TRY(dalloc(&d_len, (size_t)cfg.max_rows * 4));
    TRY(dalloc(&d_pos_off, (size_t)cfg.max_rows * 4));
    TRY(dalloc(&d_last, (size_t)cfg.max_rows * 4));
    TRY(dalloc(&d_row_off, (size_t)cfg.max_rows * 4));
    TRY(dalloc(&d_row_order, (size_t)cfg.max_rows * 4));
    TRY(dalloc(&d_unf, (size_t)cfg.max_rows * 4));
    TRY(dalloc(&d_anyunf, 1024 * 4));
    TRY(dalloc(&d_ndec, 64));
    TRY(dalloc(&cfg_pv, (size_t)cfg.max_rows * 16 * 4));
    TRY(dalloc(&cfg_pi, (size_t)cfg.max_rows * 16 * 4));
    HIPCHK(hipMemset(d_ndec, 0, 64));
    TRY(dalloc(&d_sparams, 64)); TRY(dalloc(&d_tparams, 64));
    {
        const size_t nt = (size_t)cfg.max_rows * (cfg.max_new + 1);     // [B, T] with B <= max_rows / 2 ... [R, max_new] for text
        TRY(dalloc(&d_out_tok, nt * 4)); TRY(dalloc(&d_force_tok, nt * 4)); TRY(dalloc(&d_force_mask, nt));
        if (cfg.with_lm_head) TRY(dalloc(&d_text_out, nt * 8));
    }
    TRY(dalloc(&d_tok_row, (size_t)max_tok * 4));
    TRY(dalloc(&d_tok_j, (size_t)max_tok * 4));
    TRY(dalloc(&d_tok_src, (size_t)max_tok * 4));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipHostMalloc((void**)&h_stage2[i], (size_t)(3 * max_tok + 5 * cfg.max_rows + 16) * 4));
        HIPCHK(hipEventCreateWithFlags(&ev_stage[i], hipEventDisableTiming));
    }
    HIPCHK(hipHostMalloc((void**)&h_flag, 64));
    TRY(dalloc(&d_flag, 64));
    TRY(dalloc(&x, (size_t)max_tok * Hh * 4));
    TRY(dalloc(&xn, (size_t)max_tok * Hh * esz));
    long pn = 3L * HDm; if (2L * I > pn) pn = 2L * I; if (Hh > pn) pn = Hh;
    part_elems = max_tok * pn;
    {   // decode-time split-K slabs can exceed the prefill need when max_tok is small
        const long rows = cfg.max_rows;
        long need = 0;
        auto upd = [&](long N, long K) {
            for (long mc : {16L, 32L, 48L, 64L, 96L, 128L, 192L, 256L, rows, rows / 2}) {
                if (mc < 1) continue;
                const long m = mc < rows ? mc : rows;
                const long S = (K % 128 == 0) ? skinny_pick_splits((int)N, (int)K, (int)m) : 1;
                if (S * m * N > need) need = S * m * N;
            }
        };
        upd(3L * HDm, Hh); upd(Hh, HDm); upd(2L * I, Hh); upd(Hh, I); upd(G, Hh); upd(V, G);
        if (cfg.with_lm_head) upd(cfg.vocab, Hh);
        if (need > part_elems) part_elems = need;
        decode_part_elems = need;
    }
    TRY(dalloc(&part, (size_t)part_elems * 4));
    TRY(dalloc(&part2, (size_t)decode_part_elems * 4));
    TRY(dalloc(&ssq_part, (size_t)cfg.max_rows * 8 * 4));
    TRY(dalloc(&d_ndec2, 64));
    HIPCHK(hipMemset(d_ndec2, 0, 64));
    TRY(dalloc(&qbuf, (size_t)max_tok * HDm * esz));
    TRY(dalloc(&obuf, (size_t)max_tok * HDm * esz));
    TRY(dalloc(&hbuf, (size_t)max_tok * I * esz));
    TRY(dalloc(&hfin, (size_t)cfg.max_rows * Hh * esz));
    TRY(dalloc(&gh_in, (size_t)cfg.max_rows * Hh * esz));
    TRY(dalloc(&gh_mid, (size_t)cfg.max_rows * G * esz));
    {   // VQ activations: largest tensor of the decoder / encoder schedule
        long mx = 0;
        const int nres = cfg.vq_levels;
        for (int lvl = 0; lvl < nres; ++lvl) {
            const long side = (long)cfg.grid << (nres - 1 - lvl);
            long c = (long)cfg.vq_ch * cfg.vq_ch_mult[lvl];
            if (lvl + 1 < nres && (long)cfg.vq_ch * cfg.vq_ch_mult[lvl + 1] > c) c = (long)cfg.vq_ch * cfg.vq_ch_mult[lvl + 1];
            if (side * side * c > mx) mx = side * side * c;
        }
        const long g2 = (long)cfg.grid * cfg.grid;
        if (g2 * cfg.vq_z > mx) mx = g2 * cfg.vq_z;
        vbuf_elems = mx * cfg.max_images;
        for (int i = 0; i < 4; ++i) TRY(dalloc(&vbuf[i], (size_t)vbuf_elems * 4));   // fp32 skip stream
        const long cm = (long)cfg.vq_ch * cfg.vq_ch_mult[nres - 1];
        const long ab = (long)cfg.max_images * g2 * cm;
        TRY(dalloc(&aq, (size_t)ab * esz));
        TRY(dalloc(&ak, (size_t)ab * esz));
        TRY(dalloc(&avt, (size_t)ab * esz));
        TRY(dalloc(&ao, (size_t)ab * esz));
        TRY(dalloc(&ap, (size_t)cfg.max_images * g2 * g2 * esz));
        TRY(dalloc(&ascore, (size_t)cfg.max_images * g2 * g2 * 4));
        TRY(dalloc(&gn_stats, (size_t)cfg.max_images * 64 * 4));
        TRY(dalloc(&gn_ws, (size_t)cfg.max_images * 64 * 4 * 1024));     // [image][<= 1024 splits / conv tiles][32 groups][2]
        TRY(dalloc(&gn_coef, (size_t)cfg.max_images * (cm > 1024 ? cm : 1024) * 2 * 4));
        if (cfg.with_vq_encoder) TRY(dalloc(&enc_z, (size_t)cfg.max_images * g2 * 8 * 4));
    }
    HIPCHK(hipStreamCreateWithFlags(&istream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&istream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&ev_t0)); HIPCHK(hipEventCreate(&ev_t1));
    HIPCHK(hipEventCreate(&ev_p0)); HIPCHK(hipEventCreate(&ev_p1));
    HIPCHK(hipEventCreate(&ev_v0)); HIPCHK(hipEventCreate(&ev_v1));
    const char* ng = getenv("PG_NO_GRAPH");
    if (ng && ng[0] == '1') use_graph = false;
    const char* ug = getenv("PG_USE_GRAPH");
    if (ug && ug[0] == '1') use_graph = true;
    // dalloc zero-fills every allocation with hipMemsetAsync on the NULL stream; the caller's streams may be non-blocking ones
    // (PyTorch side streams do not order with the legacy default stream), so the fills must have landed before any of them runs
    // (a still-queued memset could land on a workspace after the first op wrote it).
    HIPCHK(hipStreamSynchronize(nullptr));
    return PG_OK;
}

void pg_engine::destroy() {
    (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    drop_graphs();
    for (void* p : allocs) (void)hipFree(p);
    if (stage_dev) (void)hipFree(stage_dev);
    for (int i = 0; i < 2; ++i) { if (h_stage2[i]) (void)hipHostFree(h_stage2[i]); if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]); }
    if (h_flag) (void)hipHostFree(h_flag);
    for (hipEvent_t e : tc_ev) (void)hipEventDestroy(e);
    hipEvent_t evs[] = {ev_in, ev_out, ev_t0, ev_t1, ev_p0, ev_p1, ev_v0, ev_v1, ev_fork, ev_join};
    if (istream2) (void)hipStreamDestroy(istream2);
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    if (istream) (void)hipStreamDestroy(istream);
}

// =============================================================================== weights
int pg_engine::load_tensor(const char* name_c, const void* src, int dtype, const int64_t* shape, int ndim) {
    std::string name = name_c;
    if (name.rfind("vl_gpt.", 0) == 0) name = name.substr(7);
    auto it = slots_map.find(name);
    if (it == slots_map.end()) FAIL(PG_ERR_NAME, "unknown tensor '%s'", name.c_str());
    Slot& sl = it->second;
    long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != sl.n) FAIL(PG_ERR_ARG, "tensor '%s': %ld elements, expected %ld", name.c_str(), n, sl.n);
    if (!sl.shape.empty() && ndim >= 2) {     // 2-D / 4-D slots: a transposed or re-laid-out tensor has the right count and the wrong shape
        bool ok = ndim == (int)sl.shape.size();
        for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == sl.shape[i];
        if (!ok) {
            std::string got, want;
            for (int i = 0; i < ndim; ++i) got += (i ? "," : "") + std::to_string(shape[i]);
            for (size_t i = 0; i < sl.shape.size(); ++i) want += (i ? "," : "") + std::to_string(sl.shape[i]);
            FAIL(PG_ERR_ARG, "tensor '%s': shape [%s], expected [%s]", name.c_str(), got.c_str(), want.c_str());
        }
    }
    if (dtype != PG_F32 && dtype != PG_BF16) FAIL(PG_ERR_ARG, "tensor '%s': dtype must be f32/bf16", name.c_str());
    HIPCHK(hipSetDevice(dev));
    const long nbytes = n * (dtype == PG_BF16 ? 2 : 4);
    if (nbytes > stage_bytes) {
        if (stage_dev) { (void)hipFree(stage_dev); bytes -= stage_bytes; }
        stage_bytes = nbytes < (64L << 20) ? (64L << 20) : nbytes;
        HIPCHK(hipMalloc(&stage_dev, stage_bytes)); bytes += stage_bytes;
    }
    HIPCHK(hipMemcpy(stage_dev, src, nbytes, hipMemcpyHostToDevice));
    const int sb = dtype == PG_BF16;
    hipStream_t s = nullptr;
    switch (sl.kind) {
        case K_F32: launch_to_f32(s, stage_dev, sb, (float*)sl.dst, n); break;
        case K_T:
            if (bf) launch_convert<bf16>(s, stage_dev, sb, (bf16*)sl.dst, n);
            else launch_convert<float>(s, stage_dev, sb, (float*)sl.dst, n);
            break;
        case K_IL16_G: case K_IL16_U: {
            const int which = sl.kind == K_IL16_U;
            if (bf) launch_convert_interleave16<bf16>(s, stage_dev, sb, (bf16*)sl.dst, sl.a, sl.b, which);
            else launch_convert_interleave16<float>(s, stage_dev, sb, (float*)sl.dst, sl.a, sl.b, which);
            break;
        }
        case K_CONV:
            if (bf) launch_convert_conv<bf16>(s, stage_dev, sb, (bf16*)sl.dst, sl.a, sl.b, sl.c);
            else launch_convert_conv<float>(s, stage_dev, sb, (float*)sl.dst, sl.a, sl.b, sl.c);
            break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    sl.loaded = true;
    finalized = false;
    return PG_OK;
}

int pg_engine::finalize(int* missing, hipStream_t s) {
    int miss = 0;
    for (auto& kvp : slots_map)
        if (!kvp.second.loaded) { if (!miss) err = "missing tensor: " + kvp.first; ++miss; }
    if (missing) *missing = miss;
    HIPCHK(hipSetDevice(dev));
    const int Hh = H(), V = cfg.img_vocab, Dm = cfg.img_dim;
    {   // gen_table[v] = gen_aligner(gen_embed[v])  (modeling_vlm.py:270-271; projector.py:38-44), fp32
        float* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, (size_t)V * Hh * 4));
        GemmA a; a.ptr = ge_w; a.lda = Dm;
        GemmEpi e; e.out = tmp; e.out_f32 = 1; e.ldc = Hh; e.bias_n = al_b0; e.act = 1;
        launch_gemm<float>(s, a, al_w0, Dm, 0, e, V, Hh, Dm, 1);
        GemmA a2; a2.ptr = tmp; a2.lda = Hh;
        GemmEpi e2; e2.out = gen_table; e2.out_f32 = 1; e2.ldc = Hh; e2.bias_n = al_b2;
        launch_gemm<float>(s, a2, al_w2, Hh, 0, e2, V, Hh, Hh, 1);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipFree(tmp));
    }
    {   // pq_table[v] = post_quant_conv(normalize(codebook[v]))  (vq_model.py:284-299, :500-503)
        launch_l2norm_rows(s, codebook, codebook_n, V, Dm);
        float* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, (size_t)V * cfg.vq_z * 4));
        GemmA a; a.ptr = codebook_n; a.lda = Dm;
        GemmEpi e; e.out = tmp; e.out_f32 = 1; e.ldc = cfg.vq_z; e.bias_n = pq_b;
        launch_gemm<float>(s, a, pq_w, Dm, 0, e, V, cfg.vq_z, Dm, 1);
        if (bf) launch_convert<bf16>(s, tmp, 0, (bf16*)pq_table, (long)V * cfg.vq_z);
        else launch_convert<float>(s, tmp, 0, (float*)pq_table, (long)V * cfg.vq_z);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipFree(tmp));
    }
    {   // RoPE tables: inv_freq = theta^(-2j/128); cos/sin(pos * inv_freq) in fp32 (LlamaRotaryEmbedding)
        std::vector<float> c((size_t)max_pos * 64), sn((size_t)max_pos * 64);
        for (int j = 0; j < 64; ++j) {
            const float inv = 1.0f / powf(cfg.rope_theta, (float)(2 * j) / 128.0f);
            for (int p = 0; p < max_pos; ++p) {
                const float f = (float)p * inv;
                c[(size_t)p * 64 + j] = cosf(f); sn[(size_t)p * 64 + j] = sinf(f);
            }
        }
        HIPCHK(hipMemcpy(cos_t, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sin_t, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    }
    if (bf) {   // decode copies of the GEMM weights in the tiled layout (contiguous 1 KiB per wave load)
        const int Hh2 = H(), I = cfg.inter, HDm = HD();
        for (Layer& ly : layers) {
            TRY(tile_one(s, ly.wqkv, &ly.wqkv_t, 3 * HDm, Hh2));
            // prefill copy for the fused RoPE / KV-write epilogue (gemm256 act 3); 25 MB per layer at Janus-Pro-1B size.  Only when this
            // handle's CAPACITY can ever reach the fused path (gemm256_try takes >= 200 tiles of 256 x 256: ~2.1 k packed prompt tokens at
            // N = 6144) -- small-batch / short-prompt engines never use it and no longer pay for it
            const long max_packed = (long)cfg.max_rows * cfg.max_prompt;
            const bool can_fuse = ((max_packed + 255) / 256) * ((3L * HDm + 255) / 256) >= 200;
            if ((Hh2 & 7) == 0 && can_fuse) {
                if (!ly.wqkv_p) TRY(dalloc(&ly.wqkv_p, (size_t)3 * HDm * Hh2 * 2));
                launch_interleave_qk(s, (const bf16*)ly.wqkv, (bf16*)ly.wqkv_p, cfg.n_heads, Hh2);
            }
            TRY(tile_one(s, ly.wo, &ly.wo_t, Hh2, HDm));
            TRY(tile_one(s, ly.wgu, &ly.wgu_t, 2 * I, Hh2));
            TRY(tile_one(s, ly.wd, &ly.wd_t, Hh2, I));
        }
        TRY(tile_one(s, gh_w1, &gh_w1_t, cfg.gen_head_dim, Hh2));
        TRY(tile_one(s, gh_w2, &gh_w2_t, cfg.img_vocab, cfg.gen_head_dim));
        if (lm_head) TRY(tile_one(s, lm_head, &lm_head_t, cfg.vocab, Hh2));
        HIPCHK(hipStreamSynchronize(s));
    }
    HIPCHK(hipGetLastError());
    // missing tensors: the engine refuses to run (prefill / decode / VQ check ``finalized``) unless the caller opted in
    finalized = miss == 0 || allow_partial;
    return PG_OK;
}
int pg_engine::tile_one(hipStream_t s, const void* src, void** dst, int N, int K) {
    if ((N & 15) || (K % 128)) { *dst = nullptr; return PG_OK; }
    if (!*dst) TRY(dalloc(dst, (size_t)N * K * 2));
    launch_tile_weights(s, (const bf16*)src, (bf16*)*dst, N, K);
    return PG_OK;
}

