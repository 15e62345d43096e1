// pg_engine: VQ-16 decoder (VQModel.decode_code, vq_model.py:505-508 -> Decoder.forward :193-214) and encoder + quantiser (:494-498).
#include "engine.h"

// =============================================================================== VQ-16
// Precision layout (bf16 mode): the tensors that carry the resblock skip connections (``cur``,
// conv outputs feeding a GroupNorm, shortcut outputs) are fp32; conv / GEMM inputs (GroupNorm
// outputs, attention operands) are T.  Keeping the skip stream in fp32 removes the largest
// bf16 error term (measured offline: 5.6e-5 of the 1e-4 pixel-MSE budget).
// GroupNorm statistics of ``in`` -> per-(image, channel) affine coefficients in gn_coef (y = x a + sh)
template <typename TI>
void pg_engine::gn_coefs(hipStream_t s, const NormW& n, const TI* in, int B, int HW) {
    // statistics already produced by the convolution that wrote ``in`` (conv_halo / gemm256 epilogue)?
    if (gn_part_of == (const void*)in && gn_part_n > 0 && gn_part_b == B) launch_gn_finalize(s, gn_ws, gn_stats, gn_coef, n.g, n.b, B, gn_part_n, HW, n.c, 1e-6f);
    else launch_gn_stats(s, in, sizeof(TI) == 2, gn_stats, gn_ws, B, HW, n.c, 1e-6f, gn_coef, n.g, n.b);
    gn_part_of = nullptr;
}
template <typename T, typename TI>
void pg_engine::gn(hipStream_t s, const NormW& n, const TI* in, T* out, int B, int HW, int swish) {
    gn_coefs<TI>(s, n, in, B, HW);
    launch_gn_apply<TI, T>(s, in, gn_coef, out, B, HW, n.c, swish);
}
template <typename T>
void pg_engine::conv3(hipStream_t s, const ConvW& cw, const T* in, void* out, int out_f32, const void* residual,
                      int res_f32, int B, int Hi, int Wi, int up, int stride2, int feeds_gn) {
    GemmA a; a.kind = stride2 ? 2 : 1; a.ptr = in; a.Hi = Hi; a.Wi = Wi; a.Cin = cw.cin; a.up = up; a.zeros = zeros;
    const int Ho = stride2 ? Hi / 2 : (Hi << up), Wo = stride2 ? Wi / 2 : (Wi << up);
    GemmEpi e; e.out = out; e.out_f32 = out_f32; e.ldc = cw.cout; e.bias_n = cw.b; e.residual = residual; e.res_f32 = res_f32;
    int nsp = 0;
    gn_part_of = nullptr;
    if (feeds_gn < 0) feeds_gn = out_f32;                     // fp32 outputs are the skip stream / GroupNorm inputs
    if (feeds_gn && (Ho / 8) * (Wo / 32) <= 1024) { a.gn_part = gn_ws; a.gn_nsplit = &nsp; }
    launch_gemm<T>(s, a, (const T*)cw.w, 9L * cw.cin, 0, e, B * Ho * Wo, cw.cout, 9 * cw.cin, 1);
    if (nsp > 0) { gn_part_of = out; gn_part_n = nsp; gn_part_b = B; }
}
template <typename T>
void pg_engine::conv1(hipStream_t s, const ConvW& cw, const T* in, void* out, int out_f32, const void* residual,
                      int res_f32, long M, int gn_hw) {
    GemmA a; a.ptr = in; a.lda = cw.cin;
    GemmEpi e; e.out = out; e.out_f32 = out_f32; e.ldc = cw.cout; e.bias_n = cw.b; e.residual = residual; e.res_f32 = res_f32;
    int nsp = 0;
    // the output feeds a GroupNorm: statistics from the GEMM's epilogue when the kernel can (round 6).  Without the request gn_ws is not written and
    // whatever partials it holds (of the tensor a previous convolution stored) stay valid: gn_part_of is left alone.
    if (gn_hw > 0) { gn_part_of = nullptr; a.gn_part = gn_ws; a.gn_nsplit = &nsp; a.gn_hw = gn_hw; }
    launch_gemm<T>(s, a, (const T*)cw.w, cw.cin, 0, e, (int)M, cw.cout, cw.cin, 1);
    if (nsp > 0) { gn_part_of = out; gn_part_n = nsp; gn_part_b = (int)(M / gn_hw); }
}
// ResnetBlock.forward (vq_model.py:337-352) on ``cur`` (fp32); result becomes the new ``cur``.
template <typename T>
void pg_engine::resblock(hipStream_t s, const ResBlockW& r, int B, int Hs, int Ws) {
    const int HW = Hs * Ws;
    const void* res = cur;
    if (r.has_nin) {                          // 1x1 shortcut needs a T copy of the block input
        launch_convert<T>(s, cur, 0, (T*)t1, (long)B * HW * r.nin.cin);
        conv1<T>(s, r.nin, (const T*)t1, t3, 1, nullptr, 0, (long)B * HW);
        res = t3;
    }
    gn<T>(s, r.n1, (const float*)cur, (T*)t1, B, HW, 1);
    // conv1's output only feeds norm2 (not the skip stream): kept in T when mid_bf16 (its GroupNorm statistics still come from
    // the fp32 accumulators in the epilogue) -- 4 bytes per element less traffic on an HBM-bound pair of kernels
    if (mid_bf16 && sizeof(T) == 2) {
        conv3<T>(s, r.c1, (const T*)t1, t2, 0, nullptr, 0, B, Hs, Ws, 0, 0, 1);
        gn<T, T>(s, r.n2, (const T*)t2, (T*)t1, B, HW, 1);
    } else {
        conv3<T>(s, r.c1, (const T*)t1, t2, 1, nullptr, 0, B, Hs, Ws, 0, 0);
        gn<T>(s, r.n2, (const float*)t2, (T*)t1, B, HW, 1);
    }
    conv3<T>(s, r.c2, (const T*)t1, t2, 1, res, 1, B, Hs, Ws, 0, 0);
    std::swap(cur, t2);
}
// AttnBlock.forward (vq_model.py:366-390): single head over HW tokens, scale C^-0.5.
template <typename T>
void pg_engine::attnblock(hipStream_t s, const AttnW& a, int B, int HW) {
    const int C = a.n.c;
    gn<T>(s, a.n, (const float*)cur, (T*)t1, B, HW, 0);
    conv1<T>(s, a.q, (const T*)t1, aq, 0, nullptr, 0, (long)B * HW);
    conv1<T>(s, a.k, (const T*)t1, ak, 0, nullptr, 0, (long)B * HW);
    {   // V^T[b] = Wv . t1[b]^T + bv  -> [C, HW]  (operands swapped so the PV GEMM sees K-contiguous V)
        GemmA ga; ga.ptr = a.v.w; ga.lda = C; ga.strideA = 0;
        GemmEpi e; e.out = avt; e.out_f32 = 0; e.ldc = HW; e.strideC = (long)C * HW; e.bias_m = a.v.b;
        launch_gemm<T>(s, ga, (const T*)t1, C, (long)HW * C, e, C, HW, C, B);
    }
    {   // scores[b] = q[b] . k[b]^T   fp32 [HW, HW]
        GemmA ga; ga.ptr = aq; ga.lda = C; ga.strideA = (long)HW * C;
        GemmEpi e; e.out = ascore; e.out_f32 = 1; e.ldc = HW; e.strideC = (long)HW * HW;
        launch_gemm<T>(s, ga, (const T*)ak, C, (long)HW * C, e, HW, HW, C, B);
    }
    launch_softmax_rows<T>(s, ascore, (T*)ap, B * HW, HW, 1.0f / sqrtf((float)C));
    {   // o[b] = P[b] . V^T[b]^T  -> [HW, C]
        GemmA ga; ga.ptr = ap; ga.lda = HW; ga.strideA = (long)HW * HW;
        GemmEpi e; e.out = ao; e.out_f32 = 0; e.ldc = C; e.strideC = (long)HW * C;
        launch_gemm<T>(s, ga, (const T*)avt, HW, (long)C * HW, e, HW, C, HW, B);
    }
    conv1<T>(s, a.p, (const T*)ao, t2, 1, cur, 1, (long)B * HW, HW);          // the block's output is the next GroupNorm's input
    std::swap(cur, t2);
}

// VQModel.decode_code (vq_model.py:505-508) -> Decoder.forward (:193-214).
template <typename T>
int pg_engine::vq_decode(const int32_t* codes, void* img_out, int out_dtype, int B, hipStream_t s) {
    if (!finalized) FAIL(PG_ERR_STATE, "pg_finalize_weights not called");
    if (B < 1 || B > cfg.max_images) FAIL(PG_ERR_CAPACITY, "images %d > max_images %d", B, cfg.max_images);
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipEventRecord(ev_v0, s));
    cur = vbuf[0]; t1 = vbuf[1]; t2 = vbuf[2]; t3 = vbuf[3];
    const int g = cfg.grid, nres = cfg.vq_levels;
    launch_vq_gather<T>(s, (const T*)pq_table, codes, (T*)t1, B * g * g, cfg.vq_z, cfg.img_vocab);
    conv3<T>(s, dec.conv_in, (const T*)t1, cur, 1, nullptr, 0, B, g, g, 0, 0);
    resblock<T>(s, dec.mid0, B, g, g);
    attnblock<T>(s, dec.mid1, B, g * g);
    resblock<T>(s, dec.mid2, B, g, g);
    int side = g;
    for (int bi = 0; bi < nres; ++bi) {
        const VqLevel& lv = dec.levels[bi];
        for (size_t j = 0; j < lv.res.size(); ++j) {
            resblock<T>(s, lv.res[j], B, side, side);
            if (j < lv.attn.size()) attnblock<T>(s, lv.attn[j], B, side * side);
        }
        if (lv.has_resample) {   // Upsample.forward (:417-427): nearest 2x folded into the conv's addressing
            launch_convert<T>(s, cur, 0, (T*)t1, (long)B * side * side * lv.resample.cin);
            conv3<T>(s, lv.resample, (const T*)t1, t2, 1, nullptr, 0, B, side, side, 1, 0);
            std::swap(cur, t2);
            side *= 2;
        }
    }
    // tail (vq_model.py:210-214): conv_out(swish(norm_out(h))) = gn_apply + conv_out.  Option vq_tail_fused = 1 (bf16): one pass over the fp32 skip stream
    // (conv3x3_out_gn_kernel, round 6: the normalised tensor is never written; bit-identical, but measured SLOWER -- 4.05 ms against 3.4 ms -- and off by default).
    bool co_done = false, t1_ready = false;
    if constexpr (std::is_same<T, bf16>::value) {
        if (tune.vq_tail_fused && tune.conv_halo && dec.norm_out.c == 128 && dec.conv_out.cin == 128) {
            gn_coefs<float>(s, dec.norm_out, (const float*)cur, B, side * side);
            co_done = conv_out_gn_try(s, (const float*)cur, gn_coef, (const bf16*)dec.conv_out.w, dec.conv_out.b, img_out, out_dtype == PG_BF16, B, side, side,
                                      dec.conv_out.cin, 3, 1);
            if (!co_done) {       // shape not taken: the coefficients are in place, only the apply pass is missing
                launch_gn_apply<float, T>(s, (const float*)cur, gn_coef, (T*)t1, B, side * side, dec.norm_out.c, 1);
                t1_ready = true;
            }
        }
    }
    if (!co_done && !t1_ready) gn<T>(s, dec.norm_out, (const float*)cur, (T*)t1, B, side * side, 1);
    if constexpr (std::is_same<T, bf16>::value)
        if (!co_done)
            co_done = conv_out_halo_try(s, (const bf16*)t1, (const bf16*)dec.conv_out.w, dec.conv_out.b, (const bf16*)zeros, img_out,
                                        out_dtype == PG_BF16, B, side, side, dec.conv_out.cin, 3);
    if (!co_done)
        launch_conv3x3_small<T>(s, (const T*)t1, (const T*)dec.conv_out.w, dec.conv_out.b, img_out, out_dtype == PG_BF16, B, side, side,
                                dec.conv_out.cin, 3);
    HIPCHK(hipEventRecord(ev_v1, s));
    have_vq_t = true;
    HIPCHK(hipGetLastError());
    return PG_OK;
}

// VQModel.encode (vq_model.py:494-498) -> Encoder.forward (:105-124) -> VectorQuantizer (:236-258).
template <typename T>
int pg_engine::vq_encode(const void* img, int img_dtype, int64_t* idx, int B, hipStream_t s) {
    if (!finalized) FAIL(PG_ERR_STATE, "pg_finalize_weights not called");
    if (!cfg.with_vq_encoder) FAIL(PG_ERR_STATE, "engine created without the VQ encoder");
    if (B < 1 || B > cfg.max_images) FAIL(PG_ERR_CAPACITY, "images %d > max_images %d", B, cfg.max_images);
    HIPCHK(hipSetDevice(dev));
    cur = vbuf[0]; t1 = vbuf[1]; t2 = vbuf[2]; t3 = vbuf[3];
    const int nres = cfg.vq_levels;
    int side = img_size();
    launch_conv3x3_in<float>(s, img, img_dtype == PG_BF16, enc_in_w, enc_in_b, (float*)cur, B, side, side, 3, cfg.vq_ch);
    for (int lvl = 0; lvl < nres; ++lvl) {
        const VqLevel& lv = enc.levels[lvl];
        for (size_t j = 0; j < lv.res.size(); ++j) {
            resblock<T>(s, lv.res[j], B, side, side);
            if (j < lv.attn.size()) attnblock<T>(s, lv.attn[j], B, side * side);
        }
        if (lv.has_resample) {
            launch_convert<T>(s, cur, 0, (T*)t1, (long)B * side * side * lv.resample.cin);
            conv3<T>(s, lv.resample, (const T*)t1, t2, 1, nullptr, 0, B, side, side, 0, 1);
            std::swap(cur, t2);
            side /= 2;
        }
    }
    resblock<T>(s, enc.mid0, B, side, side);
    attnblock<T>(s, enc.mid1, B, side * side);
    resblock<T>(s, enc.mid2, B, side, side);
    gn<T>(s, enc.norm_out, (const float*)cur, (T*)t1, B, side * side, 1);
    conv3<T>(s, enc.conv_out, (const T*)t1, t2, 0, nullptr, 0, B, side, side, 0, 0);
    {   // quant_conv 1x1 z -> img_dim, fp32 out
        GemmA a; a.ptr = t2; a.lda = cfg.vq_z;
        GemmEpi e; e.out = enc_z; e.out_f32 = 1; e.ldc = cfg.img_dim; e.bias_n = qc_b;
        launch_gemm<T>(s, a, (const T*)qc_w, cfg.vq_z, 0, e, B * side * side, cfg.img_dim, cfg.vq_z, 1);
    }
    launch_vq_argmin(s, enc_z, codebook_n, idx, B * side * side, cfg.img_dim, cfg.img_vocab);
    HIPCHK(hipGetLastError());
    return PG_OK;
}

template int pg_engine::vq_decode<float>(const int32_t*, void*, int, int, hipStream_t);
template int pg_engine::vq_decode<bf16>(const int32_t*, void*, int, int, hipStream_t);
template int pg_engine::vq_encode<float>(const void*, int, int64_t*, int, hipStream_t);
template int pg_engine::vq_encode<bf16>(const void*, int, int64_t*, int, hipStream_t);
// pg_op_conv3x3 / pg_op_groupnorm (engine_api.hip)
template void pg_engine::conv3<float>(hipStream_t, const ConvW&, const float*, void*, int, const void*, int, int, int, int, int, int, int);
template void pg_engine::conv3<bf16>(hipStream_t, const ConvW&, const bf16*, void*, int, const void*, int, int, int, int, int, int, int);
template void pg_engine::gn<float, float>(hipStream_t, const NormW&, const float*, float*, int, int, int);
template void pg_engine::gn<bf16, float>(hipStream_t, const NormW&, const float*, bf16*, int, int, int);
