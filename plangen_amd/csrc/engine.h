// Host side of libplangen_hip.so: the engine behind include/plangen_hip.h (struct pg_engine; member functions in engine_core.hip =
// create / weights, engine_llm.hip = prefill / decode loop / text decode, engine_vq.hip = VQ-16 decoder + encoder, engine_vision.hip =
// SigLIP + aligner, engine_api.hip = the C ABI).  One engine per (process, GPU).  Owns weights (converted to the compute dtype and to
// the layouts the kernels want), the KV cache [layer][K|V][row][head][slot][128], workspaces, and the decode-step hipGraph.  No torch
// types anywhere: plain pointers and sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/plangen_hip.h"
#include "kernels.h"

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            char _b[512];                                                                                \
            snprintf(_b, sizeof _b, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            this->err = _b;                                                                              \
            return PG_ERR_HIP;                                                                           \
        }                                                                                                \
    } while (0)
#define FAIL(code, ...)                       \
    do {                                      \
        char _b[512];                         \
        snprintf(_b, sizeof _b, __VA_ARGS__); \
        this->err = _b;                       \
        return code;                          \
    } while (0)
#define TRY(expr)                  \
    do {                           \
        int _rc = (expr);          \
        if (_rc != PG_OK) return _rc; \
    } while (0)

enum SlotKind { K_T = 0, K_F32, K_IL16_G, K_IL16_U, K_CONV };
struct Slot {
    void* dst = nullptr; SlotKind kind = K_T; long n = 0;
    int a = 0, b = 0, c = 0;
    bool loaded = false;
    std::vector<int64_t> shape;       // expected state_dict shape (empty: only the element count is checked)
};
struct ConvW { void* w = nullptr; float* b = nullptr; int cin = 0, cout = 0, k = 3; };
struct NormW { float* g = nullptr; float* b = nullptr; int c = 0; };
struct ResBlockW { NormW n1, n2; ConvW c1, c2, nin; bool has_nin = false; };
struct AttnW { NormW n; ConvW q, k, v, p; };
struct VqLevel { std::vector<ResBlockW> res; std::vector<AttnW> attn; ConvW resample; bool has_resample = false; };
struct LinW { void* w = nullptr; float* b = nullptr; int out = 0, in = 0; };
struct VitBlockW { NormW n1, n2; LinW qkv, proj, fc1, fc2; };
struct VqNet { ConvW conv_in; ResBlockW mid0, mid2; AttnW mid1; std::vector<VqLevel> levels; NormW norm_out; ConvW conv_out; };

struct pg_engine {
    pg_config cfg{};
    int dev = 0;
    bool bf = true;
    size_t esz = 2;
    std::string err;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    std::map<std::string, Slot> slots_map;

    // ---- weights
    struct Layer { void *wqkv, *wo, *wgu, *wd, *ln1, *ln2; void *wqkv_t = nullptr, *wo_t = nullptr, *wgu_t = nullptr, *wd_t = nullptr;
                   void* wqkv_p = nullptr; };      // wqkv_p: [8 | 8]-interleaved q / k rows for the prefill RoPE epilogue (bf16)
    void *gh_w1_t = nullptr, *gh_w2_t = nullptr, *lm_head_t = nullptr;
    int tile_one(hipStream_t s, const void* src, void** dst, int N, int K);
    std::vector<Layer> layers;
    void* norm_w = nullptr; float* embed = nullptr; void* lm_head = nullptr;
    void *gh_w1 = nullptr, *gh_w2 = nullptr; float *gh_b1 = nullptr, *gh_b2 = nullptr;
    float *ge_w = nullptr, *al_w0 = nullptr, *al_b0 = nullptr, *al_w2 = nullptr, *al_b2 = nullptr;
    float* gen_table = nullptr;
    float *codebook = nullptr, *codebook_n = nullptr, *pq_w = nullptr, *pq_b = nullptr;
    void* pq_table = nullptr;
    void* qc_w = nullptr; float* qc_b = nullptr;                 // encoder quant_conv (z -> img_dim)
    float *enc_in_w = nullptr, *enc_in_b = nullptr;              // encoder conv_in (3 -> ch), fp32 [Cout][3][3][3]
    VqNet dec, enc;
    // SigLIP + aligner (a13)
    LinW vit_patch, al0, al2; float* vit_pos = nullptr; std::vector<VitBlockW> vit_blocks; NormW vit_norm;
    float* vx = nullptr; void *vt = nullptr, *vqk = nullptr, *vvt = nullptr, *vo = nullptr, *vh = nullptr, *vp = nullptr, *val = nullptr;
    float* vscore = nullptr;
    float *cos_t = nullptr, *sin_t = nullptr; int max_pos = 0;
    void* zeros = nullptr;
    bool finalized = false;
    bool allow_partial = false;       // pg_set_option("allow_partial_weights", 1): run with missing tensors (they read as zeros)

    // ---- sequence state
    int R = 0, L = 0, Ntok = 0, slots = 0; bool prefilled = false; int pos_mode = 0;
    int n_dec_host = 0;
    std::vector<int> h_len;
    int32_t *d_len = nullptr, *d_pos_off = nullptr, *d_ndec = nullptr, *d_tok_row = nullptr, *d_tok_j = nullptr,
            *d_tok_src = nullptr, *d_last = nullptr, *d_unf = nullptr, *d_anyunf = nullptr, *d_row_off = nullptr;
    int max_len_host = 0; bool flash_prefill = true;
    int32_t* h_stage2[2] = {nullptr, nullptr};                   // pinned host staging, double-buffered
    hipEvent_t ev_stage[2] = {nullptr, nullptr}; bool stage_used[2] = {false, false}; int stage_sel = 0;
    int32_t* d_flag = nullptr; int32_t* h_flag = nullptr;        // uncond-sharing probe result
    float* cfg_pv = nullptr; int* cfg_pi = nullptr;              // sampler stage-1 winners
    SampleParams* d_sparams = nullptr; TextParams* d_tparams = nullptr;   // per-call parameters the graphs read from HBM
    int32_t *d_out_tok = nullptr, *d_force_tok = nullptr; uint8_t* d_force_mask = nullptr; int64_t* d_text_out = nullptr;
    int rng_image_offset = 0;                                    // pg_set_option("rng_image_offset", lo): this rank's first image in the global batch
    PgTune tune;                                                 // per-handle tuning knobs (pg_set_option)
    int tune_epoch = 0;                                          // bumped by every option that changes what a captured graph contains
    void* kv = nullptr;
    // ---- workspaces
    long max_tok = 0;
    float* x = nullptr; void* xn = nullptr; float* part = nullptr; long part_elems = 0, decode_part_elems = 0;
    float* ssq_part = nullptr;        // [max_rows][8] partial sums of squares of the residual rows: the deferred-1/rms decode norm (round 6)
    bool defer_norm = true;           // decode at 65..128 rows, bf16: norm = barrier-free elementwise pass, 1/rms applied in the consumer GEMM's epilogue
    void *qbuf = nullptr, *obuf = nullptr, *hbuf = nullptr, *hfin = nullptr, *gh_in = nullptr, *gh_mid = nullptr;
    // VQ: cur / t1 / t2 / t3 rotate through vbuf
    void* vbuf[4] = {nullptr, nullptr, nullptr, nullptr}; long vbuf_elems = 0;
    void *cur = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
    void *aq = nullptr, *ak = nullptr, *avt = nullptr, *ao = nullptr, *ap = nullptr; float* ascore = nullptr;
    float *gn_stats = nullptr, *gn_ws = nullptr, *gn_coef = nullptr;
    const void* gn_part_of = nullptr; int gn_part_n = 0, gn_part_b = 0;   // gn_ws holds conv-epilogue partials of this tensor
    float* enc_z = nullptr;
    void* stage_dev = nullptr; long stage_bytes = 0;
    // ---- streams / graph / timing
    hipStream_t istream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_p0 = nullptr, ev_p1 = nullptr, ev_v0 = nullptr, ev_v1 = nullptr;
    hipGraphExec_t gexec = nullptr; std::vector<int64_t> gkey;
    hipGraphExec_t gexec_txt = nullptr; std::vector<int64_t> gkey_txt;      // text-decode step (lm_head + argmax + stack)
    void drop_graphs() { if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; } if (gexec_txt) { (void)hipGraphExecDestroy(gexec_txt); gexec_txt = nullptr; } }
    bool use_graph = false;   // decode step replayed as a hipGraph; OFF by default: same-stream launches measure 1 % (bs=64) to 3.3 % (bs=8/16) faster than graph replay on ROCm 7.2 and the host loop keeps ahead at every batch size
    bool prefill_rope_epi = true;     // prefill QKV: RoPE + KV write in the 256x256 GEMM's epilogue when the shape takes that kernel (0: GEMM -> fp32 q|k|v -> rope_kv_kernel)
    bool prefill_res_epi = true;      // prefill o / down: residual add in the GEMM epilogue (0: slab + norm-kernel form, for A/B)
    bool time_attn = false; bool fuse_rope = true; bool force_swiglu = true; bool mid_bf16 = true;
    bool skip_attn = false;                                      // libplangen_diag.so only (pg_diag_set_option): the decode step WITHOUT its attention launches -- results are garbage; no entry point of libplangen_hip.so can set it
    // per-kernel-class HIP-event timing of the decode loop (eager instrumented pass, pg_set_option("time_attn", 1)):
    // one event pair per launch group on the launch stream, on every ``time_stride``-th decode step
    enum { TC_ATTN = 0, TC_QKV, TC_O, TC_GU, TC_DOWN, TC_NORM, TC_HEAD, TC_SAMPLE, TC_EMPTY, TC_N };
    std::vector<hipEvent_t> tc_ev; size_t tc_used = 0; std::vector<std::pair<int, double>> tc_meta; int time_stride = 1;
    double tc_ms[TC_N] = {}, tc_bytes[TC_N] = {}; int tc_launches[TC_N] = {};
    bool tc_on = false; hipStream_t tc_stream = nullptr;
    void tic(hipStream_t s) {
        if (!tc_on) return;
        while (tc_ev.size() < tc_used + 2) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { tc_on = false; return; } tc_ev.push_back(e); }
        (void)hipEventRecord(tc_ev[tc_used], s);
    }
    void toc(hipStream_t s, int cls, double bytes) {
        if (!tc_on || tc_ev.size() < tc_used + 2) return;
        (void)hipEventRecord(tc_ev[tc_used + 1], s);
        tc_meta.emplace_back(cls, bytes); tc_used += 2;
    }
    pg_timing timing{};
    bool have_decode_t = false, have_prefill_t = false, have_vq_t = false;
    int S_last = 1; long slab_last = 0;

    template <typename U> int dalloc(U** p, size_t n_bytes, bool zero = true) {
        void* q = nullptr;
        if (n_bytes == 0) n_bytes = 16;
        HIPCHK(hipMalloc(&q, n_bytes));
        if (zero) HIPCHK(hipMemsetAsync(q, 0, n_bytes, nullptr));      // weights never loaded read as zeros, not as stale HBM
        allocs.push_back(q); bytes += (int64_t)n_bytes; *p = (U*)q;
        return PG_OK;
    }
    int H() const { return cfg.hidden; }
    int HD() const { return cfg.n_heads * cfg.head_dim; }
    int img_tokens() const { return cfg.grid * cfg.grid; }
    int img_size() const { return cfg.grid << (cfg.vq_levels - 1); }
    size_t kv_layer_elems() const { return (size_t)cfg.max_rows * cfg.n_heads * slots * 128; }
    void* kc(int layer) const { return (char*)kv + ((size_t)layer * 2 + 0) * kv_layer_elems() * esz + kv_row_off; }
    void* vc(int layer) const { return (char*)kv + ((size_t)layer * 2 + 1) * kv_layer_elems() * esz + kv_row_off; }
    int shared_len = 0, shared_row = 1; bool share_uncond = true;
    int uncond_hint = -1;             // next pg_prefill only: 1 = caller guarantees every odd row carries row 1's ids, 0 = it does not, -1 = probe on the device (one 4-byte read + stream sync)
    // decode lanes: the batch's rows split into independent chains on separate streams
    size_t kv_row_off = 0;            // byte offset of the current lane's first row inside a K or V layer block
    int h_len_off = 0; int lanes_opt = -1;   // -1 auto, 1, 2
    float* part2 = nullptr; hipStream_t istream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int32_t* d_ndec2 = nullptr;
    int32_t* d_row_order = nullptr; bool lpt_order = true; bool order_valid = false; int order_rows = 0;
    SeqState seq() const { return SeqState{d_len, d_pos_off, d_ndec, d_tok_row, d_tok_j, shared_len, shared_row, (lpt_order && order_valid && kv_row_off == 0 && R == order_rows) ? d_row_order : nullptr}; }

    int create();
    void add_slot(const std::string& name, void* dst, SlotKind k, long n, int a = 0, int b = 0, int c = 0);
    void slot_shape(const std::string& name, std::initializer_list<int64_t> shp) { slots_map[name].shape = shp; }
    int alloc_conv(const std::string& name, ConvW& cw, int cout, int cin, int k);
    int alloc_norm(const std::string& name, NormW& nw, int c);
    int alloc_res(const std::string& name, ResBlockW& r, int cin, int cout);
    int alloc_attn(const std::string& name, AttnW& a, int c);
    int build_vq();
    int alloc_lin(const std::string& name, LinW& l, int out, int in);
    int build_vision();
    template <typename T> void lin(hipStream_t s, const LinW& l, const T* in, void* out, int out_f32, const void* residual, int res_f32, int act, long M);
    template <typename T> int vision_encode(const void* img, int img_dtype, void* out, int out_dtype, int B, hipStream_t s);
    int load_tensor(const char* name, const void* src, int dtype, const int64_t* shape, int ndim);
    int finalize(int* missing, hipStream_t s);
    int prefill(const int32_t* ids_dev, const void* emb_dev, int emb_dtype, const int32_t* pad_len, int R_, int L_,
                int pmode, void* hidden_out, int hidden_dtype, hipStream_t s);
    template <typename T> void gemm_residual(hipStream_t s, const T* a, const T* W, int M, int N, int K);
    template <typename T> void gemm_llm(hipStream_t s, const T* a, const T* W, int M, int N, int K, bool allow_skinny, const void* Wt = nullptr);
    template <typename T> void run_layers(hipStream_t s, int M, int mode, T* final_out, int32_t* advance = nullptr);
    template <typename T> void head_logits(hipStream_t s, const T* in, int M);
    void forward_decode(hipStream_t s);
    int decode_image(int T, float cfgw, float temp, uint64_t seed, const int32_t* force_tok, const uint8_t* force_mask,
                     int32_t* out_tok, float* logits_out, hipStream_t s);
    int step(const void* emb, int emb_dtype, void* hidden_out, int hidden_dtype, hipStream_t s);
    int gen_head(const void* h_dev, int h_dtype, float* logits, int R_, hipStream_t s);
    int text_greedy(int max_new, int min_new, int eos, int64_t* out, int* out_len, hipStream_t s);
    template <typename T> int vq_decode(const int32_t* codes, void* img_out, int out_dtype, int B, hipStream_t s);
    template <typename T> int vq_encode(const void* img, int img_dtype, int64_t* idx, int B, hipStream_t s);
    template <typename T> void conv3(hipStream_t s, const ConvW& cw, const T* in, void* out, int out_f32, const void* residual, int res_f32, int B, int Hi, int Wi, int up, int stride2, int feeds_gn = -1);
    template <typename T> void conv1(hipStream_t s, const ConvW& cw, const T* in, void* out, int out_f32, const void* residual, int res_f32, long M, int gn_hw = 0);   // gn_hw > 0: the output feeds a GroupNorm (pixels per image)
    template <typename T> void resblock(hipStream_t s, const ResBlockW& r, int B, int Hs, int Ws);
    template <typename T> void attnblock(hipStream_t s, const AttnW& a, int B, int HW);
    template <typename TI> void gn_coefs(hipStream_t s, const NormW& n, const TI* in, int B, int HW);      // statistics (conv epilogue partials or the stats kernel) -> gn_coef
    template <typename T, typename TI = float> void gn(hipStream_t s, const NormW& n, const TI* in, T* out, int B, int HW, int swish);
    int fetch_timing();
    void destroy();
};

struct TuneGuard {          // points this thread's kernel launchers at the handle's knobs for the duration of one ABI call
    const PgTune* saved;
    explicit TuneGuard(pg_handle h) : saved(pg_tune) { if (h) pg_tune = &h->tune; }
    ~TuneGuard() { pg_tune = saved; }
};

