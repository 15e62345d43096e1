/*
 * plangen_hip.h -- C ABI of the MI355X-native PlanGen layout->image generation path.
 *
 * The reference (360CVGroup/PlanGen) has no FFI of its own: its hot path is Python
 * calling torch/transformers.  The entry points below are the boundary SURVEY.md
 * section 8b defines: each replaces one Python-level call the reference's
 * ``System`` makes on ``self.vl_gpt`` (file:line cited per function, relative to the
 * reference tree).  The Python facade in ``plangen_amd/`` binds them with ctypes and
 * re-exposes the reference's call surface.
 *
 * Conventions
 *   - return 0 on success, negative pg_status on failure; never throws across the ABI;
 *     ``pg_last_error`` gives the message for the handle (or the global one for create).
 *   - "dev" pointers are device (HBM) memory owned by the caller (PyTorch-ROCm tensors);
 *     "host" pointers are small per-row metadata in host memory.
 *   - the library owns weights, KV cache and workspaces (hipMalloc at pg_create).
 *   - every call is asynchronous on the caller's hipStream_t; no hidden device syncs
 *     except where stated (pg_generate_text_greedy polls its finished-flag; pg_prefill reads ONE
 *     4-byte flag back when it probes for a batch-constant negative prompt -- ONLY when the caller
 *     did not supply the answer through the one-shot ``uncond_shared_hint`` option (0 / 1): with
 *     the hint set, pg_prefill performs no device->host read and no stream synchronisation).
 *   - one handle per (process, GPU); a handle is not thread-safe.
 */
#ifndef PLANGEN_HIP_H
#define PLANGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pg_engine* pg_handle;
typedef void* pg_stream;              /* hipStream_t */

typedef enum { PG_F32 = 0, PG_BF16 = 1, PG_I32 = 2, PG_I64 = 3 } pg_dtype;

typedef enum {
    PG_OK = 0,
    PG_ERR_ARG = -1,        /* bad argument / shape */
    PG_ERR_HIP = -2,        /* HIP runtime error (message in pg_last_error) */
    PG_ERR_STATE = -3,      /* call out of order (e.g. decode before prefill) */
    PG_ERR_NAME = -4,       /* unknown tensor name */
    PG_ERR_CAPACITY = -5    /* exceeds rows / KV slots / images configured at create */
} pg_status;

#define PG_MAX_VQ_LEVELS 8

/* Shapes of the path.  Defaults for Janus-Pro-1B as used by PlanGen: SURVEY App. A. */
typedef struct pg_config {
    int32_t hidden, inter, n_layers, n_heads, head_dim;   /* 2048 5632 24 16 128 */
    int32_t vocab;                                        /* 102400 */
    int32_t img_vocab, img_dim, grid, gen_head_dim;       /* 16384 8 24 2048 */
    int32_t vq_ch, vq_levels, vq_ch_mult[PG_MAX_VQ_LEVELS], vq_z, vq_res_blocks; /* 128 5 {1,1,2,2,4} 256 2 */
    float   rms_eps, rope_theta;                          /* 1e-6 10000 */
    int32_t compute_dtype;   /* PG_BF16: bf16 weights/activations/KV, fp32 accumulate, fp32
                                residual stream and norm/softmax statistics (what
                                torch.autocast(bf16) does at plangen_base.py:360).
                                PG_F32 : everything fp32 (BASELINE config 1 / parity mode). */
    int32_t max_rows;        /* R capacity (2 x images for CFG) */
    int32_t max_prompt;      /* longest real prompt (tokens, without padding) */
    int32_t max_new;         /* decode capacity (576 image tokens, or text tokens) */
    int32_t max_images;      /* VQ decode batch capacity */
    int32_t with_lm_head;    /* allocate lm_head (text/layout decode, a11) */
    int32_t with_vq_encoder; /* allocate VQ encoder (a14) */
    /* SigLIP understanding encoder + aligner (a13, task_type='mmu'); siglip_large_patch16_384:
       width 1024, 24 layers, 16 heads (64 each), MLP 4096, patch 16, image 384 (siglip_vit.py:628-637) */
    int32_t with_vision, vit_width, vit_layers, vit_heads, vit_mlp, vit_patch, vit_img;
    int32_t max_vision_images;   /* images per pg_vision_encode call */
} pg_config;

/* -- lifetime ------------------------------------------------------------------------ */
/* Replaces AutoModelForCausalLM.from_pretrained(...).cuda() (plangen_base.py:95). */
int pg_create(pg_handle* out, const pg_config* cfg, int device_id);
int pg_destroy(pg_handle h);
const char* pg_last_error(pg_handle h /* may be NULL */);

/* Load one tensor by its reference state_dict name (MultiModalityCausalLM keys,
 * modeling_vlm.py:190-219; a leading "vl_gpt." from PlanGen's .pth overlay,
 * base_system.py:153-155, is accepted).  ``src`` is HOST memory, dtype PG_F32 or PG_BF16.
 * Unknown-but-irrelevant names (vision_model.*, aligner.*, encoder when not configured)
 * return PG_ERR_NAME so the caller can count what it skipped. */
int pg_load_tensor(pg_handle h, const char* name, const void* src, int dtype,
                   const int64_t* shape, int ndim);
/* After all tensors: build derived tables (gen_embed->gen_aligner table, L2-normalised
 * codebook -> post_quant_conv table, RoPE cos/sin).  ``missing`` (may be NULL) receives the
 * number of required tensors never loaded. */
int pg_finalize_weights(pg_handle h, int* missing, pg_stream s);

/* -- language-model path ---------------------------------------------------------------- */
/* Replaces the first call of  vl_gpt.language_model.model(inputs_embeds=embed(ids),
 * attention_mask=mask, use_cache=True)  (plangen_base.py:548,571-577) for a LEFT-padded
 * batch.  ids_dev int32 [R, L]; pad_len_host int32 [R] = number of leading pad slots of
 * each row (mask==0 prefix, plangen_base.py:711-712).  Pad slots are skipped (SURVEY App.
 * B-9).  position_mode 0: RoPE position of slot j = j (absolute slot index incl. padding,
 * what sample_image gets because it passes no position_ids, App. B-1); 1: position =
 * j - pad_len (HF generate: mask.cumsum-1).  Resets the KV cache. hidden_out_dev (may be
 * NULL) receives last_hidden_state in ``hidden_dtype`` laid out [R, L, hidden] (pad slots
 * zero). */
int pg_prefill(pg_handle h, const int32_t* ids_dev, const int32_t* pad_len_host, int R, int L,
               int position_mode, void* hidden_out_dev, int hidden_dtype, pg_stream s);
/* Same, from caller-provided embeddings [R, L, hidden] (dtype PG_F32/PG_BF16): the
 * ``emb is not None`` branch of t2i (plangen_base.py:543-545) and x2t (:513). */
int pg_prefill_embeds(pg_handle h, const void* embeds_dev, int embeds_dtype,
                      const int32_t* pad_len_host, int R, int L, int position_mode,
                      void* hidden_out_dev, int hidden_dtype, pg_stream s);

/* One decode step behind  language_model.model(inputs_embeds=[R,1,H], past_key_values=...)
 * (plangen_base.py:571-577, i>0): consumes embeds_dev [R, hidden], appends K/V, writes
 * last_hidden_state [R, hidden] (after the final RMSNorm) to hidden_out_dev. */
int pg_step(pg_handle h, const void* embeds_dev, int embeds_dtype, void* hidden_out_dev,
            int hidden_dtype, pg_stream s);

/* vl_gpt.gen_head(h) (modeling_vlm.py:47-51; call site plangen_base.py:579):
 * h_dev [R, hidden] -> logits_dev fp32 [R, img_vocab]. */
int pg_gen_head(pg_handle h, const void* h_dev, int h_dtype, float* logits_dev, int R, pg_stream s);

/* vl_gpt.prepare_gen_img_embeds(tok) (modeling_vlm.py:270-271; plangen_base.py:603):
 * tok_dev int32 [R] -> out_dev [R, hidden]. */
int pg_gen_embed(pg_handle h, const int32_t* tok_dev, void* out_dev, int out_dtype, int R, pg_stream s);

/* language_model.get_input_embeddings()(ids) (plangen_base.py:371,548): ids int32 [n]. */
int pg_embed_tokens(pg_handle h, const int32_t* ids_dev, void* out_dev, int out_dtype, int n, pg_stream s);

/* The whole System.sample_image loop (plangen_base.py:567-607) on device after a
 * pg_prefill of R = 2B CFG-interleaved rows: T steps of {layer stack, gen_head, CFG mix
 * (:580-587), sample (:588-591), gen_embed+gen_aligner feedback (:602-604)}.
 * temperature <= 0: greedy argmax, ties -> lowest index (parity mode);
 * > 0: sample softmax(logits/temperature) by the Gumbel-max trick with a counter-based
 * RNG keyed on (seed, image, step).
 * force_tok_dev  int32 [B, T] or NULL: token fed back instead of the sampled one
 *     (teacher-forced parity protocol); with force_mask_dev uint8 [B, T] != NULL it is the
 *     reference's use_teacher_forcing branch (:593-598): where mask==0 the emitted AND
 *     fed-back token is force_tok (gt label), elsewhere the model's own.
 * out_tok_dev    int32 [B, T]: emitted tokens.
 * logits_out_dev fp32 [T, B, img_vocab] or NULL: CFG-mixed logits per step (tests). */
int pg_decode_image_tokens(pg_handle h, int T, float cfg_weight, float temperature, uint64_t seed,
                           const int32_t* force_tok_dev, const uint8_t* force_mask_dev,
                           int32_t* out_tok_dev, float* logits_out_dev, pg_stream s);

/* language_model.generate(inputs_embeds=..., do_sample=False, max_new_tokens=...,
 * eos_token_id=pad_token_id=eos) (System.x2t, plangen_base.py:513-523) after a
 * pg_prefill*(position_mode=1) of B rows: greedy argmax(lm_head(h[:, -1])), finished rows
 * emit eos, stops when every row is finished.  out_dev int64 [B, max_new]; *out_len_host =
 * number of columns produced (new tokens only).  min_new suppresses EOS for the first
 * min_new steps (benchmark workload).  Synchronises the stream every few steps to read
 * the all-finished flag. */
int pg_generate_text_greedy(pg_handle h, int max_new, int min_new, int eos_id, int64_t* out_dev,
                            int* out_len_host, pg_stream s);

/* -- VQ-16 tokenizer ---------------------------------------------------------------------- */
/* gen_vision_model.decode_code(codes, shape=[B,8,g,g]) (vq_model.py:505-508; call site
 * plangen_base.py:555): codes_dev int32 [B, g*g] -> img_out_dev [B, 3, S, S] (NCHW like the
 * reference), S = g * 2^(levels-1). */
int pg_vq_decode(pg_handle h, const int32_t* codes_dev, void* img_out_dev, int out_dtype, int B, pg_stream s);
/* gen_vision_model.encode(x)[-1][-1] (vq_model.py:494-498; plangen_base.py:532):
 * img_dev [B,3,S,S] -> idx_out_dev int64 [B*g*g]. */
int pg_vq_encode(pg_handle h, const void* img_dev, int img_dtype, int64_t* idx_out_dev, int B, pg_stream s);

/* -- understanding encoder ------------------------------------------------------------------ */
/* aligner(vision_model(images)) of MultiModalityCausalLM.prepare_inputs_embeds
 * (modeling_vlm.py:243-250): CLIPVisionTower.forward (clip_encoder.py:107-122) ->
 * VisionTransformer.forward_features (siglip_vit.py:562-572; no cls token, learned pos-embed,
 * LayerNorm eps 1e-6, non-causal MHA, GELU MLP) -> MlpProjector mlp_gelu (projector.py:38-44).
 * img_dev [B,3,S,S] -> out_dev [B, (S/patch)^2, hidden].  The masked scatter into the text
 * embeddings (modeling_vlm.py:263-266) is data movement done by the caller. */
int pg_vision_encode(pg_handle h, const void* img_dev, int img_dtype, void* out_dev, int out_dtype, int B, pg_stream s);

/* -- introspection (tests / bench) -------------------------------------------------------- */
/* Last-launch timing of the decode loop measured with HIP events on ``s`` inside the
 * library: fills ms for the whole pg_decode_image_tokens call and, when per-kernel
 * timing was enabled with pg_set_option("time_attn", 1), the summed duration and launch
 * count of the decode-attention kernel. */
typedef struct pg_timing {
    float  decode_ms;          /* whole last pg_decode_image_tokens */
    float  attn_ms_sum;        /* sum over timed decode-attention launches */
    int32_t attn_launches;
    double attn_bytes_sum;     /* algorithmic K/V bytes those launches had to read */
    float  prefill_ms;         /* whole last pg_prefill* */
    float  vq_ms;              /* whole last pg_vq_decode */
} pg_timing;
int pg_get_timing(pg_handle h, pg_timing* out);
/* Per-kernel-class sums of the last instrumented decode loop (pg_set_option("time_attn", 1); call pg_get_timing
 * first: it collects the events).  cls 0..8: decode attention, QKV / O / gate|up(+SwiGLU) / down GEMMs, RMSNorm
 * (+ split-K reduce + residual), gen_head, CFG sampler, 8: event pairs around nothing (instrumentation overhead); *bytes_sum = algorithmic HBM bytes of the timed launches
 * (weights once per launch; K/V once per launch).  Returns PG_ERR_ARG past the last class. */
int pg_get_class_timing(pg_handle h, int cls, const char** name, double* ms_sum, int* launches, double* bytes_sum);
/* Switches of the product library (defaults in parentheses), PER HANDLE.  NONE changes what a handle returns: they are measurement taps,
 * per-call hints, and A/B fallbacks that produce the same results (each is equality-tested on the GPU).  Experiments that lost their
 * measurement were removed in round 5; kernel-variant tables, timing ablations and the "decode step without attention" measurement
 * mode live in a SEPARATE library (libplangen_diag.so: pg_diag_set_option, pg_bench_*; csrc/diag_api.hip) that this header does not cover.
 *   measurement
 *     time_attn (0)          per-launch HIP events around every decode kernel class (eager loop) -> pg_get_timing / pg_get_class_timing
 *     time_stride (1)        ... on every n-th decode step only
 *   per-call hints
 *     rng_image_offset (0)   global index of this handle's image 0: prompt-sharded ranks sample exactly what one big batch would
 *     uncond_shared_hint     ONE-SHOT, consumed by the next pg_prefill: 1 = the caller has compared the ids on the host (its collate built
 *                            them, plangen_base.py:672-686 replicates one negative prompt) and every odd row equals row 1 -> no device
 *                            probe, pg_prefill does not synchronise; 0 = they differ; -1 (default) = probe on the device (one 4-byte
 *                            read + stream sync)
 *     allow_partial_weights (0)  run although required tensors were never loaded (they read as zeros)
 *   decode loop
 *     share_uncond (1)       prefill / store a batch-constant negative prompt once (0: every uncond row keeps a private copy)
 *     use_graph (0)          replay the decode step as a hipGraph (one launch per step from the host; env PG_USE_GRAPH=1).  Off by
 *                            default: same-stream launches of the ~175 kernels of a step measure 1-3 % faster than graph replay
 *     lanes (1)              2: two row-range lanes on two streams (measured 6 % slower at bs=64; kept as a documented fallback)
 *     stream_gemm (-1 auto)  bit mask of the decode GEMM classes on the v4 kernel (x tile by LDS-DMA): 1 wide-N slabs, 2 narrow-N slabs,
 *                            4 SwiGLU gate|up, 8 the M <= 16 kernels, 16 SwiGLU through the LDS-transposed epilogue; 0 = v3 everywhere;
 *                            128 = v3 wide-N blocks of 64 columns / 4 waves instead of 128 columns / 8 waves
 *     split_target_big (128) decode split-K block-count target at >= 96 rows (per-handle isolation is tested with it)
 *   prefill
 *     flash_prefill (1)      MFMA flash attention for prefill (0: per-query streaming kernel)
 *     prefill_attn (2)       MFMA prefill attention: 2 = 128 queries per block, K/V by LDS-DMA, V through the LDS transpose read; 1 = 64-query kernel
 *     prefill_rope_epi (1)   QKV projection: RoPE + KV-cache write in the 256x256 GEMM's epilogue when the packed batch takes that kernel
 *                            (plangen_base.py:571 -> LlamaAttention.forward); 0 = GEMM -> fp32 q|k|v -> RoPE / KV-fill kernel.  Same bits.
 *     prefill_res_epi (1)    o_proj / down_proj: residual add in the GEMM epilogue; 0 = fp32 slab folded in by the norm kernel.  Same bits.
 *     gemm256 (1)            256x256 MFMA GEMM for large shapes (0: 128x128 kernel everywhere; 4 / 5 / 6 pin the tile height to 256 / 224 / 192 rows
 *                            instead of choosing it per launch; +8 = four phases per K tile instead of two).  Bit-identical results in every form.
 *   VQ-16
 *     conv_halo (1)          direct halo-tile 3x3 convolution for Cin=Cout=128 (2: lock-step variant with the generic epilogue and the one-patch conv_out, 0: implicit-GEMM kernel)
 *     vq_mid_bf16 (1)        bf16 mode: the tensor between a ResnetBlock's two convolutions is bf16 (statistics from the fp32
 *                            accumulators); 0 keeps it fp32 like the skip stream
 *     vq_tail_fused (0)      1 = decoder tail conv_out(swish(norm_out(h))) (vq_model.py:210-214) in one pass over the fp32 skip stream instead of a GroupNorm
 *                            apply pass + conv_out.  Bit-identical pixels; measured slower in round 6 (4.05 vs 3.4 ms at 64 images), hence off.
 *     vq_argmin_multi (1)    nearest-code search: 8 latent vectors per block (0: one per block); identical indices
 *   SigLIP
 *     vit_attn (2)           2 = K / V^T of a head resident in LDS, 16 waves per block; 4 / 8 / 12 / 16 = that kernel with so many waves;
 *                            1 = the 64-key tile kernel of rounds 2-3 (bit-identical results)
 *     ln_wave (1)            LayerNorm (width 1024): wave-per-row register kernel; 0 = block-per-row kernel
 * Returns PG_ERR_ARG for an unknown key. */
int pg_set_option(pg_handle h, const char* key, int64_t value);
/* Bytes of device memory the handle owns (weights + KV + workspace). */
int64_t pg_device_bytes(pg_handle h);
/* Debug taps for parity tests: copy an internal buffer to dst_dev.
 * name: "kcache"/"vcache" (layer in ``index``), "x" (residual stream), "xn", "hfin", "gen_table", "pq_table", "qbuf", "obuf",
 * "vit_feat" (SigLIP features of the last pg_vision_encode, after the final LayerNorm, compute dtype [B, P, vit_width]). */
int pg_debug_read(pg_handle h, const char* name, int index, void* dst_dev, int64_t max_bytes, pg_stream s);

/* Stand-alone operator entry points (unit parity tests call these through the ABI;
 * the engine uses the same kernels internally). All tensors device memory. */
int pg_op_rmsnorm(pg_handle h, float* x_dev /*[M,H] in/out*/, const float* partial_dev /*[S,M,H]|NULL*/,
                  int S, const void* w_dev /*compute dtype [H]*/, void* out_dev /*compute dtype [M,H]*/,
                  int M, int H, float eps, pg_stream s);
int pg_op_gemm(pg_handle h, const void* a_dev /*[M,K]*/, const void* w_dev /*[N,K]*/, float* out_dev /*[S,M,N]*/,
               int M, int N, int K, int force_kind /*0 auto,1 skinny,2 big,3 f32,4 skinny on the tiled decode copy of W*/,
               int* S_out, pg_stream s);
/* The decode gate|up GEMM with SwiGLU in its epilogue (transformers LlamaMLP: down(silu(gate(x)) * up(x))):
 * a_dev bf16 [M,K]; wgu_dev bf16 [2I,K] with gate/up rows interleaved in blocks of 8 (rows 16j..16j+7 = gate rows
 * 8j..8j+7, rows 16j+8..16j+15 = the matching up rows: the engine's internal layout); h_out_dev bf16 [M,I]. */
int pg_op_swiglu_gemm(pg_handle h, const void* a_dev, const void* wgu_dev, void* h_out_dev, int M, int I, int K, pg_stream s);
/* The sampler's RNG output stage: raw 64-bit generator outputs bits_dev [n] -> out_dev fp32 [2n] =
 * (uniform u in (0,1) | Gumbel noise -log(-log u)) exactly as cfg_scan_kernel computes them. */
int pg_op_uniform(pg_handle h, const uint64_t* bits_dev, float* out_dev, int n, pg_stream s);

/* 3x3 convolution over NHWC activations (compute dtype), the VQ-16 ResnetBlock / Upsample /
 * Downsample conv (vq_model.py:337-352, :417-427, :440-447).  w_dev is [Cout][9][Cin]
 * (tap-major, the engine's internal layout), bias fp32; up=1 folds a nearest-2x upsample of
 * the input in; stride2=1 is the encoder's pad-(0,1,0,1) stride-2 form. */
int pg_op_conv3x3(pg_handle h, const void* x_dev, const void* w_dev, const float* bias_dev, const void* residual_dev,
                  void* out_dev, int B, int Hi, int Wi, int Cin, int Cout, int up, int stride2, pg_stream s);
/* GroupNorm(32, eps=1e-6) (+ optional swish) over NHWC activations (vq_model.py:393-403):
 * x_dev fp32 (the VQ skip stream is kept in fp32), out_dev compute dtype. */
int pg_op_groupnorm(pg_handle h, const void* x_dev, const float* gamma_dev, const float* beta_dev, void* out_dev, int B,
                    int HW, int C, int swish, pg_stream s);

#ifdef __cplusplus
}
#endif
#endif /* PLANGEN_HIP_H */
