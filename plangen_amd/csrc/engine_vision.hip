// pg_engine: SigLIP-L understanding encoder + aligner (CLIPVisionTower.forward, clip_encoder.py:107-122 -> siglip_vit.py:562-572 ->
// modeling_vlm.py:243-250).
#include "engine.h"

// =============================================================================== SigLIP + aligner
template <typename T>
void pg_engine::lin(hipStream_t s, const LinW& l, const T* in, void* out, int out_f32, const void* residual, int res_f32, int act, long M) {
    GemmA a; a.ptr = in; a.lda = l.in;
    GemmEpi e; e.out = out; e.out_f32 = out_f32; e.ldc = l.out; e.bias_n = l.b; e.residual = residual; e.res_f32 = res_f32; e.act = act;
    launch_gemm<T>(s, a, (const T*)l.w, l.in, 0, e, (int)M, l.out, l.in, 1);
}
template <typename T>
int pg_engine::vision_encode(const void* img, int img_dtype, void* out, int out_dtype, int B, hipStream_t s) {
    if (!finalized) FAIL(PG_ERR_STATE, "pg_finalize_weights not called");
    if (!cfg.with_vision) FAIL(PG_ERR_STATE, "engine created without the vision encoder");
    if (B < 1 || B > cfg.max_vision_images) FAIL(PG_ERR_CAPACITY, "images %d > max_vision_images %d", B, cfg.max_vision_images);
    if (out_dtype != PG_F32 && !bf) FAIL(PG_ERR_ARG, "bf16 output needs the bf16 engine");
    HIPCHK(hipSetDevice(dev));
    const int C = cfg.vit_width, ps = cfg.vit_patch, g = cfg.vit_img / ps, P = g * g, NH = cfg.vit_heads;
    const long M = (long)B * P;
    // PatchEmbed conv16x16/s16 as a GEMM over gathered patches, + bias, + learned pos-embed (no cls token)
    launch_patchify<T>(s, img, img_dtype == PG_BF16, (T*)vt, B, cfg.vit_img, ps);
    lin<T>(s, vit_patch, (const T*)vt, vx, 1, nullptr, 0, 0, M);
    launch_add_pos(s, vx, vit_pos, B, P, C);
    const float scale = 1.0f / sqrtf(64.0f);
    for (const VitBlockW& w : vit_blocks) {
        launch_layernorm<T>(s, vx, w.n1.g, w.n1.b, (T*)vt, (int)M, C, 1e-6f);
        {   // q | k = t . Wqk^T + b  ([M, 2C]); V^T[b] = Wv . t[b]^T + bv ([C, P], operands swapped)
            GemmA a; a.ptr = vt; a.lda = C;
            GemmEpi e; e.out = vqk; e.out_f32 = 0; e.ldc = 2 * C; e.bias_n = w.qkv.b;
            launch_gemm<T>(s, a, (const T*)w.qkv.w, C, 0, e, (int)M, 2 * C, C, 1);
            GemmA av; av.ptr = (const T*)w.qkv.w + (long)2 * C * C; av.lda = C;
            GemmEpi ev; ev.out = vvt; ev.out_f32 = 0; ev.ldc = P; ev.strideC = (long)C * P; ev.bias_m = w.qkv.b + 2 * C;
            launch_gemm<T>(s, av, (const T*)vt, C, (long)P * C, ev, C, P, C, B);
        }
        bool vflash = false;
        if constexpr (std::is_same<T, bf16>::value) {
            if (flash_prefill && C / NH == 64 && P % 64 == 0) {      // fused non-causal flash attention (no score tensor)
                launch_attn_vit_flash(s, (const bf16*)vqk, (const bf16*)vvt, (bf16*)vo, B, P, C, NH, scale);
                vflash = true;
            }
        }
        if (!vflash) {
        {   // scores[b,h] = q[b,:,h] . k[b,:,h]^T / sqrt(64)   (non-causal SDPA, siglip_vit.py:178-183)
            GemmA a; a.ptr = vqk; a.lda = 2 * C; a.strideA = (long)P * 2 * C; a.strideA2 = 64;
            GemmEpi e; e.out = vscore; e.out_f32 = 1; e.ldc = P; e.strideC = (long)NH * P * P; e.strideC2 = (long)P * P;
            launch_gemm<T>(s, a, (const T*)vqk + C, 2 * C, (long)P * 2 * C, e, P, P, 64, B, NH, 64);
        }
        launch_softmax_rows<T>(s, vscore, (T*)vp, (int)(B * NH * P), P, scale);
        {   // o[b,:,h] = P[b,h] . V^T[b][h*64..]^T
            GemmA a; a.ptr = vp; a.lda = P; a.strideA = (long)NH * P * P; a.strideA2 = (long)P * P;
            GemmEpi e; e.out = vo; e.out_f32 = 0; e.ldc = C; e.strideC = (long)P * C; e.strideC2 = 64;
            launch_gemm<T>(s, a, (const T*)vvt, P, (long)C * P, e, P, 64, P, B, NH, (long)64 * P);
        }
        }
        lin<T>(s, w.proj, (const T*)vo, vx, 1, vx, 1, 0, M);                      // x += proj(o)
        launch_layernorm<T>(s, vx, w.n2.g, w.n2.b, (T*)vt, (int)M, C, 1e-6f);
        lin<T>(s, w.fc1, (const T*)vt, vh, 0, nullptr, 0, 1, M);                    // GELU(erf)
        lin<T>(s, w.fc2, (const T*)vh, vx, 1, vx, 1, 0, M);                        // x += fc2(.)
    }
    launch_layernorm<T>(s, vx, vit_norm.g, vit_norm.b, (T*)vt, (int)M, C, 1e-6f);
    lin<T>(s, al0, (const T*)vt, val, 0, nullptr, 0, 1, M);                         // aligner: Linear -> GELU -> Linear
    lin<T>(s, al2, (const T*)val, out, out_dtype == PG_F32 ? 1 : 0, nullptr, 0, 0, M);
    HIPCHK(hipGetLastError());
    return PG_OK;
}

template int pg_engine::vision_encode<float>(const void*, int, void*, int, int, hipStream_t);
template int pg_engine::vision_encode<bf16>(const void*, int, void*, int, int, hipStream_t);
