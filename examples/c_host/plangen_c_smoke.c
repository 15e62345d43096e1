/* A PURE C (C99) host of libplangen_hip.so: no Python, no torch, no C++ -- the drop-in boundary of this repository is the C ABI of
 * include/plangen_hip.h, and this is the smallest program that drives the path through it:
 *   pg_create (tiny Janus-shaped config) -> pg_load_tensor (seeded weights by their reference state_dict names, modeling_vlm.py:190-219)
 *   -> pg_finalize_weights -> pg_prefill (CFG-interleaved rows, left-padded) -> pg_decode_image_tokens (greedy, cfg 5; the whole
 *   System.sample_image loop of plangen_base.py:567-607 in one call) -> tokens back to the host.
 * It runs the generation twice and checks that the tokens are identical, in range and not all equal.  The VQ decoder weights are not
 * loaded (allow_partial_weights: they read as zeros), so pg_vq_decode is only checked for returning finite pixels.
 *
 * build:  gcc -std=c99 -O1 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host/plangen_c_smoke.c \
 *             -L plangen_amd/lib -lplangen_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/plangen_amd/lib -Wl,-rpath,/opt/rocm/lib -o plangen_c_smoke
 * (tests/test_c_host.py: compiled here on the CPU box, run on the GPU box.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "plangen_hip.h"

#define CHECK_PG(call) do { int rc_ = (call); if (rc_ != PG_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pg_last_error(h)); return 1; } } while (0)
#define CHECK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static float rnd_normal(void) {               /* sum of 4 uniforms, variance-matched: plenty for a smoke test */
    float s = 0.f;
    for (int i = 0; i < 4; ++i) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; s += (float)((rng_state >> 40) & 0xffffff) / 16777216.f; }
    return (s - 2.f) * 1.7320508f;
}
static int load(pg_handle h, const char* name, float std, float mean, int64_t d0, int64_t d1) {
    const int64_t n = d0 * (d1 > 0 ? d1 : 1);
    float* buf = (float*)malloc((size_t)n * sizeof(float));
    if (!buf) return 1;
    for (int64_t i = 0; i < n; ++i) buf[i] = mean + std * rnd_normal();
    const int64_t shape[2] = {d0, d1};
    const int rc = pg_load_tensor(h, name, buf, PG_F32, shape, d1 > 0 ? 2 : 1);
    free(buf);
    if (rc != PG_OK) { fprintf(stderr, "pg_load_tensor(%s) -> %d: %s\n", name, rc, pg_last_error(h)); return 1; }
    return 0;
}

int main(void) {
    pg_handle h = NULL;
    pg_config c;
    memset(&c, 0, sizeof c);
    c.hidden = 256; c.inter = 512; c.n_layers = 2; c.n_heads = 2; c.head_dim = 128; c.vocab = 512;
    c.img_vocab = 256; c.img_dim = 8; c.grid = 8; c.gen_head_dim = 256;
    c.vq_ch = 64; c.vq_levels = 3; c.vq_ch_mult[0] = 1; c.vq_ch_mult[1] = 2; c.vq_ch_mult[2] = 2; c.vq_z = 64; c.vq_res_blocks = 2;
    c.rms_eps = 1e-6f; c.rope_theta = 10000.f; c.compute_dtype = PG_BF16;
    c.max_rows = 4; c.max_prompt = 16; c.max_new = 64; c.max_images = 2;
    if (pg_create(&h, &c, 0) != PG_OK) { fprintf(stderr, "pg_create: %s\n", pg_last_error(NULL)); return 1; }
    CHECK_PG(pg_set_option(h, "allow_partial_weights", 1));            /* the VQ decoder stays zero in this smoke test */

    const int H = c.hidden, I = c.inter, HD = c.n_heads * c.head_dim, G = c.gen_head_dim, V = c.img_vocab;
    char name[160];
    if (load(h, "language_model.model.embed_tokens.weight", 0.02f, 0.f, c.vocab, H)) return 1;
    for (int l = 0; l < c.n_layers; ++l) {
        const char* lin[7] = {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"};
        const int64_t d0[7] = {HD, HD, HD, H, I, I, H}, d1[7] = {H, H, H, HD, H, H, I};
        for (int k = 0; k < 7; ++k) { snprintf(name, sizeof name, "language_model.model.layers.%d.%s.weight", l, lin[k]); if (load(h, name, 0.05f, 0.f, d0[k], d1[k])) return 1; }
        snprintf(name, sizeof name, "language_model.model.layers.%d.input_layernorm.weight", l); if (load(h, name, 0.05f, 1.f, H, 0)) return 1;
        snprintf(name, sizeof name, "language_model.model.layers.%d.post_attention_layernorm.weight", l); if (load(h, name, 0.05f, 1.f, H, 0)) return 1;
    }
    if (load(h, "language_model.model.norm.weight", 0.05f, 1.f, H, 0)) return 1;
    if (load(h, "gen_head.output_mlp_projector.weight", 0.05f, 0.f, G, H) || load(h, "gen_head.output_mlp_projector.bias", 0.02f, 0.f, G, 0)) return 1;
    if (load(h, "gen_head.vision_head.weight", 0.05f, 0.f, V, G) || load(h, "gen_head.vision_head.bias", 0.02f, 0.f, V, 0)) return 1;
    if (load(h, "gen_embed.weight", 1.0f, 0.f, V, c.img_dim)) return 1;
    if (load(h, "gen_aligner.layers.0.weight", 0.3f, 0.f, H, c.img_dim) || load(h, "gen_aligner.layers.0.bias", 0.02f, 0.f, H, 0)) return 1;
    if (load(h, "gen_aligner.layers.2.weight", 0.05f, 0.f, H, H) || load(h, "gen_aligner.layers.2.bias", 0.02f, 0.f, H, 0)) return 1;
    int missing = -1;
    CHECK_PG(pg_finalize_weights(h, &missing, NULL));

    /* two images = four CFG-interleaved rows (cond, uncond, cond, uncond), left-padded to L = 12 */
    enum { R = 4, L = 12, T = 32 };
    const int32_t pad_len[R] = {0, 7, 3, 7};
    int32_t ids[R][L];
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < L; ++j) ids[r][j] = j < pad_len[r] ? 3 : (int32_t)(8 + ((r & 1) ? 17 * j : 31 * r + 13 * j) % (c.vocab - 8));
    int32_t *ids_dev = NULL, *tok_dev = NULL;
    CHECK_HIP(hipMalloc((void**)&ids_dev, sizeof ids));
    CHECK_HIP(hipMalloc((void**)&tok_dev, (R / 2) * T * sizeof(int32_t)));
    CHECK_HIP(hipMemcpy(ids_dev, ids, sizeof ids, hipMemcpyHostToDevice));
    int32_t tok[2][R / 2][T];
    for (int run = 0; run < 2; ++run) {
        CHECK_PG(pg_prefill(h, ids_dev, pad_len, R, L, 0, NULL, PG_F32, NULL));
        CHECK_PG(pg_decode_image_tokens(h, T, 5.0f, 0.0f, 0, NULL, NULL, tok_dev, NULL, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(tok[run], tok_dev, sizeof tok[run], hipMemcpyDeviceToHost));
    }
    int distinct = 0, bad = 0;
    for (int b = 0; b < R / 2; ++b)
        for (int t = 0; t < T; ++t) {
            if (tok[0][b][t] != tok[1][b][t] || tok[0][b][t] < 0 || tok[0][b][t] >= V) ++bad;
            if (t && tok[0][b][t] != tok[0][b][t - 1]) ++distinct;
        }
    /* full-length decode + VQ decode call (zero decoder weights -> finite pixels) */
    const int TI = c.grid * c.grid, S = c.grid << (c.vq_levels - 1);
    int32_t* code_dev = NULL; float* img_dev = NULL;
    CHECK_HIP(hipMalloc((void**)&code_dev, (R / 2) * TI * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void**)&img_dev, (size_t)(R / 2) * 3 * S * S * sizeof(float)));
    CHECK_PG(pg_prefill(h, ids_dev, pad_len, R, L, 0, NULL, PG_F32, NULL));
    CHECK_PG(pg_decode_image_tokens(h, TI, 5.0f, 0.0f, 0, NULL, NULL, code_dev, NULL, NULL));
    CHECK_PG(pg_vq_decode(h, code_dev, img_dev, PG_F32, R / 2, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* img = (float*)malloc((size_t)(R / 2) * 3 * S * S * sizeof(float));
    CHECK_HIP(hipMemcpy(img, img_dev, (size_t)(R / 2) * 3 * S * S * sizeof(float), hipMemcpyDeviceToHost));
    int nonfinite = 0;
    for (long i = 0; i < (long)(R / 2) * 3 * S * S; ++i) if (!isfinite(img[i])) ++nonfinite;
    free(img);
    printf("{\"c_host\": \"%s\", \"missing_tensors\": %d, \"tokens\": %d, \"mismatch_or_out_of_range\": %d, \"token_changes\": %d, \"first\": [%d, %d, %d, %d], "
           "\"device_mb\": %.1f, \"image\": [%d, 3, %d, %d], \"nonfinite_pixels\": %d}\n",
           (bad == 0 && distinct > 4 && nonfinite == 0) ? "ok" : "FAILED", missing, (R / 2) * T, bad, distinct, tok[0][0][0], tok[0][0][1], tok[0][1][0], tok[0][1][1],
           (double)pg_device_bytes(h) / 1048576.0, R / 2, S, S, nonfinite);
    (void)hipFree(ids_dev); (void)hipFree(tok_dev); (void)hipFree(code_dev); (void)hipFree(img_dev);
    CHECK_PG(pg_destroy(h));
    return (bad == 0 && distinct > 4 && nonfinite == 0) ? 0 : 2;
}
