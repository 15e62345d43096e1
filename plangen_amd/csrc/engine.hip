// Host side of libplangen_hip.so: the engine behind include/plangen_hip.h.
// One engine per (process, GPU).  Owns weights (converted to the compute dtype and to the
// layouts the kernels want), the KV cache [layer][K|V][row][head][slot][128], workspaces,
// and the decode-step hipGraph.  No torch types anywhere: plain pointers and sizes.
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

static thread_local std::string g_err;

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
    bool time_attn = false; bool fuse_rope = true; bool force_swiglu = true; bool gn_fuse = true; bool mid_bf16 = true; int cu_split = 0;
    bool skip_attn = false;                                      // measurement only: the decode step WITHOUT its attention launches (bench.py's graph-replayed GEMM + norm phase time)
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
    // run-ahead weight stream through the Infinity Cache (weight_prefetch_kernel on its own stream, paced by d_prog)
    int mall_prefetch = 0, pf_blocks = 256, pf_depth = 16, pf_nt = 0, pf_first = 2;
    uint32_t* d_prog = nullptr; PfLayer* d_pfplan = nullptr; uint32_t* d_pfstats = nullptr; bool pf_plan_ok = false;
    hipStream_t pf_stream = nullptr; hipEvent_t ev_pf0 = nullptr, ev_pf1 = nullptr; bool pf_live = false;
    int32_t* d_row_order = nullptr; bool lpt_order = true; int lpt_snake = 0; bool order_valid = false; int order_rows = 0;
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
    template <typename T> void conv1(hipStream_t s, const ConvW& cw, const T* in, void* out, int out_f32, const void* residual, int res_f32, long M);
    template <typename T> void resblock(hipStream_t s, const ResBlockW& r, int B, int Hs, int Ws);
    template <typename T> void attnblock(hipStream_t s, const AttnW& a, int B, int HW);
    template <typename T, typename TI = float> void gn(hipStream_t s, const NormW& n, const TI* in, T* out, int B, int HW, int swish);
    int fetch_timing();
    void destroy();
};

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
    {   // lowest priority: HIP maps streams onto a few hardware queues; a prefetcher that lands on the decode stream's queue blocks it until its bounded spin gives up
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&pf_stream, hipStreamNonBlocking, lo));
    }
    HIPCHK(hipEventCreateWithFlags(&ev_pf0, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_pf1, hipEventDisableTiming));
    TRY(dalloc(&d_prog, 64));
    TRY(dalloc(&d_pfstats, 64));
    TRY(dalloc(&d_pfplan, sizeof(PfLayer) * (size_t)(cfg.n_layers > 0 ? cfg.n_layers : 1)));
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
    if (pf_stream) { (void)hipStreamSynchronize(pf_stream); (void)hipStreamDestroy(pf_stream); }
    if (ev_pf0) (void)hipEventDestroy(ev_pf0);
    if (ev_pf1) (void)hipEventDestroy(ev_pf1);
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
        {   // what the run-ahead weight stream reads behind layer l's o_proj: gate|up(l), down(l), then qkv(l+1) -- or, behind the last
            // layer, the two gen_head matrices.  regions = consumer blocks of 128 columns (their K streams are contiguous in the tiled copy)
            std::vector<PfLayer> plan((size_t)cfg.n_layers);
            auto mat = [&](const void* p, long n, long k) { PfMat m{}; m.base = p; m.kib = (uint32_t)(n * k * 2 / 1024); m.regions = (uint32_t)((n + 127) / 128); return m; };
            for (int li = 0; li < cfg.n_layers; ++li) {
                PfLayer& pl = plan[(size_t)li]; pl = PfLayer{};
                pl.m[0] = mat(layers[li].wgu_t, 2L * I, Hh2);
                pl.m[1] = mat(layers[li].wd_t, Hh2, I);
                if (li + 1 < cfg.n_layers) pl.m[2] = mat(layers[li + 1].wqkv_t, 3L * HDm, Hh2);
                else { pl.m[2] = mat(gh_w1_t, cfg.gen_head_dim, Hh2); pl.m[3] = mat(gh_w2_t, cfg.img_vocab, cfg.gen_head_dim); }
            }
            HIPCHK(hipMemcpyAsync(d_pfplan, plan.data(), sizeof(PfLayer) * plan.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            pf_plan_ok = true;
        }
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
        launch_rmsnorm<T>(s, x, part, S_pend, slab_pend, (const T*)ly.ln1, (T*)xn, M, Hh, cfg.rms_eps, nullptr, (sk && mall_prefetch > 0) ? d_prog : nullptr);
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
        launch_rmsnorm<T>(s, x, part, S_last, slab_last, (const T*)ly.ln2, (T*)xn, M, Hh, cfg.rms_eps, nullptr, (sk && mall_prefetch > 0) ? d_prog : nullptr);
        toc(s, TC_NORM, norm_bytes(S_last));
        bool fused = false;
        tic(s);
        if constexpr (std::is_same<T, bf16>::value) {
            // decode: SwiGLU gate fused into the gate|up GEMM epilogue (S = 1, no slab, no extra kernel)
            if (sk && M <= 512 && Hh % 128 == 0 && (force_swiglu || skinny_pick_splits(2 * I, Hh, M) == 1))   // fused wins at every M (B=4/8/16: -2..3 % loop time)
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
    launch_rmsnorm<T>(s, x, part, S_pend, slab_pend, (const T*)norm_w, final_out, M, Hh, cfg.rms_eps, advance, (sk && mall_prefetch > 0) ? d_prog : nullptr);
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
        // snake order (round 4, option lpt_snake = chunk size, 0 = plain longest-first): all 16 * R blocks of the attention launch are resident at
        // once (8 per CU), so "longest first" balances nothing by itself; what matters is which ranks share a CU.  Reversing every second chunk
        // of the sorted list makes the ranks a CU receives (a stride-`chunk` pick) sum to the same length to first order.
        if (lpt_snake > 1)
            for (int c0 = lpt_snake; c0 + lpt_snake <= R_; c0 += 2 * lpt_snake) std::reverse(s_ord + c0, s_ord + c0 + lpt_snake);
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
    // cu_split: the two lanes own disjoint CU sets (streams created with a CU mask) and run their whole loops
    // independently (no per-step join, no graph: graph nodes do not carry the masks)
    const bool free_lanes = nl == 2 && cu_split > 0 && !time_attn;
    const bool graph = use_graph && !time_attn && T > 2 && !free_lanes;
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
        if (two_streams && !free_lanes) { HIPCHK(hipEventRecord(ev_fork, ws)); HIPCHK(hipStreamWaitEvent(istream2, ev_fork, 0)); }
        for (int li = 0; li < nl; ++li) {
            hipStream_t st = (two_streams && li == 1) ? istream2 : ws;
            enter(ld[li]);
            sample(st, ld[li]);
            if (with_forward) forward_decode(st);
            leave();
        }
        if (two_streams && !free_lanes) { HIPCHK(hipEventRecord(ev_join, istream2)); HIPCHK(hipStreamWaitEvent(ws, ev_join, 0)); }
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
    if (free_lanes) { HIPCHK(hipEventRecord(ev_fork, ws)); HIPCHK(hipStreamWaitEvent(istream2, ev_fork, 0)); }
    // run-ahead weight stream (option mall_prefetch, stream launches only): ticket zeroed on the work stream, ONE cross-stream edge
    // per call hands the side stream its start, the prefetcher then paces itself on the tickets of this call's T - 1 forwards
    const bool pf = mall_prefetch > 0 && pf_plan_ok && bf && !graph && nl == 1 && !time_attn && T > 2;
    if (pf) {
        HIPCHK(hipMemsetAsync(d_prog, 0, 4, ws));
        HIPCHK(hipEventRecord(ev_pf0, ws));
        HIPCHK(hipStreamWaitEvent(pf_stream, ev_pf0, 0));
        launch_weight_prefetch(pf_stream, d_pfplan, cfg.n_layers, d_prog, T - 1, 2 * cfg.n_layers + 1, pf_first, 2, pf_blocks, pf_depth, pf_nt, d_pfstats);
        HIPCHK(hipEventRecord(ev_pf1, pf_stream));
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
    if (free_lanes) { HIPCHK(hipEventRecord(ev_join, istream2)); HIPCHK(hipStreamWaitEvent(ws, ev_join, 0)); }
    if (pf) HIPCHK(hipStreamWaitEvent(ws, ev_pf1, 0));       // the prefetcher has left before the next call re-zeroes its ticket (it exits with the last forward's tickets)
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

// =============================================================================== VQ-16
// Precision layout (bf16 mode): the tensors that carry the resblock skip connections (``cur``,
// conv outputs feeding a GroupNorm, shortcut outputs) are fp32; conv / GEMM inputs (GroupNorm
// outputs, attention operands) are T.  Keeping the skip stream in fp32 removes the largest
// bf16 error term (measured offline: 5.6e-5 of the 1e-4 pixel-MSE budget).
template <typename T, typename TI>
void pg_engine::gn(hipStream_t s, const NormW& n, const TI* in, T* out, int B, int HW, int swish) {
    // statistics already produced by the convolution that wrote ``in`` (conv_halo epilogue)?
    if (gn_part_of == (const void*)in && gn_part_n > 0 && gn_part_b == B) launch_gn_finalize(s, gn_ws, gn_stats, gn_coef, n.g, n.b, B, gn_part_n, HW, n.c, 1e-6f);
    else launch_gn_stats(s, in, sizeof(TI) == 2, gn_stats, gn_ws, B, HW, n.c, 1e-6f, gn_coef, n.g, n.b);
    gn_part_of = nullptr;
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
    if (feeds_gn && gn_fuse && (Ho / 8) * (Wo / 32) <= 1024) { a.gn_part = gn_ws; a.gn_nsplit = &nsp; }
    launch_gemm<T>(s, a, (const T*)cw.w, 9L * cw.cin, 0, e, B * Ho * Wo, cw.cout, 9 * cw.cin, 1);
    if (nsp > 0) { gn_part_of = out; gn_part_n = nsp; gn_part_b = B; }
}
template <typename T>
void pg_engine::conv1(hipStream_t s, const ConvW& cw, const T* in, void* out, int out_f32, const void* residual,
                      int res_f32, long M) {
    GemmA a; a.ptr = in; a.lda = cw.cin;
    GemmEpi e; e.out = out; e.out_f32 = out_f32; e.ldc = cw.cout; e.bias_n = cw.b; e.residual = residual; e.res_f32 = res_f32;
    launch_gemm<T>(s, a, (const T*)cw.w, cw.cin, 0, e, (int)M, cw.cout, cw.cin, 1);
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
    conv1<T>(s, a.p, (const T*)ao, t2, 1, cur, 1, (long)B * HW);
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
    gn<T>(s, dec.norm_out, (const float*)cur, (T*)t1, B, side * side, 1);
    bool co_done = false;
    if constexpr (std::is_same<T, bf16>::value)
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

// =============================================================================== C ABI
struct TuneGuard {          // points this thread's kernel launchers at the handle's knobs for the duration of one ABI call
    const PgTune* saved;
    explicit TuneGuard(pg_handle h) : saved(pg_tune) { if (h) pg_tune = &h->tune; }
    ~TuneGuard() { pg_tune = saved; }
};

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
    if (!h || !key) return PG_ERR_ARG;
    if (!strcmp(key, "time_attn")) { h->time_attn = value != 0; return PG_OK; }
    if (!strcmp(key, "skip_attn")) { h->skip_attn = value != 0; return PG_OK; }
    if (!strcmp(key, "rng_image_offset")) { h->rng_image_offset = (int)value; return PG_OK; }
    if (!strcmp(key, "time_stride")) { h->time_stride = value > 0 ? (int)value : 1; return PG_OK; }
    if (!strcmp(key, "allow_partial_weights")) { h->allow_partial = value != 0; return PG_OK; }
    if (!strncmp(key, "split_target_", 13)) {
        (key[13] == 's' ? h->tune.split_small : key[13] == 'm' ? h->tune.split_mid : h->tune.split_big) = (int)value;
        h->tune_epoch++; return PG_OK;
    }
    if (!strcmp(key, "force_swiglu")) { h->force_swiglu = value != 0; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "use_graph")) { h->use_graph = value != 0; return PG_OK; }
    if (!strcmp(key, "share_uncond")) { h->share_uncond = value != 0; return PG_OK; }
    if (!strcmp(key, "uncond_shared_hint")) { h->uncond_hint = value < 0 ? -1 : (value != 0); return PG_OK; }
    if (!strcmp(key, "flash_prefill")) { h->flash_prefill = value != 0; return PG_OK; }
    if (!strcmp(key, "lanes")) { h->lanes_opt = (int)value; return PG_OK; }
    if (!strcmp(key, "mall_prefetch")) { h->mall_prefetch = (int)value; return PG_OK; }
    if (!strcmp(key, "pf_blocks")) { h->pf_blocks = value > 0 ? (int)value : 256; return PG_OK; }
    if (!strcmp(key, "pf_depth")) { h->pf_depth = (int)value; return PG_OK; }
    if (!strcmp(key, "pf_nt")) { h->pf_nt = value != 0; return PG_OK; }
    if (!strcmp(key, "pf_first")) { h->pf_first = (int)value; return PG_OK; }
    if (!strcmp(key, "gemm256")) { h->tune.gemm256 = (int)value; return PG_OK; }
    if (!strcmp(key, "conv_halo")) { h->tune.conv_halo = (int)value; return PG_OK; }
    if (!strcmp(key, "attn_pair")) { h->tune.attn_pair = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "vit_attn")) { h->tune.vit_attn = (int)value; return PG_OK; }
    if (!strcmp(key, "vq_argmin_multi")) { h->tune.vq_argmin_multi = (int)value; return PG_OK; }
    if (!strcmp(key, "wt_store")) { h->tune.wt_store = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "stream_gemm")) { h->tune.stream_gemm = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "cu_split")) {
        // lane streams with complementary CU masks: 1 = low / high half of the mask bits, 2 = even / odd bits,
        // 3 = alternating groups of 32 bits, 4 = alternating groups of 8 bits; 0 = unmasked streams
        (void)hipSetDevice(h->dev);
        h->drop_graphs();
        (void)hipStreamSynchronize(h->istream); (void)hipStreamSynchronize(h->istream2);
        (void)hipStreamDestroy(h->istream); (void)hipStreamDestroy(h->istream2);
        h->cu_split = (int)value;
        if (value == 0) {
            if (hipStreamCreateWithFlags(&h->istream, hipStreamNonBlocking) != hipSuccess) return PG_ERR_HIP;
            if (hipStreamCreateWithFlags(&h->istream2, hipStreamNonBlocking) != hipSuccess) return PG_ERR_HIP;
        } else {
            uint32_t m0[8], m1[8];
            for (int i = 0; i < 8; ++i) {
                uint32_t a;
                if (value == 1) a = i < 4 ? 0xffffffffu : 0u;
                else if (value == 2) a = 0x55555555u;
                else if (value == 3) a = (i & 1) ? 0u : 0xffffffffu;
                else a = 0x00ff00ffu;
                m0[i] = a; m1[i] = ~a;
            }
            if (hipExtStreamCreateWithCUMask(&h->istream, 8, m0) != hipSuccess) return PG_ERR_HIP;
            if (hipExtStreamCreateWithCUMask(&h->istream2, 8, m1) != hipSuccess) return PG_ERR_HIP;
        }
        return PG_OK;
    }
    if (!strcmp(key, "gn_fuse")) { h->gn_fuse = value != 0; return PG_OK; }
    if (!strcmp(key, "vq_mid_bf16")) { h->mid_bf16 = value != 0; return PG_OK; }
    if (!strcmp(key, "attn_waves")) { h->tune.attn_waves = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "lpt_order")) { h->lpt_order = value != 0; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "lpt_snake")) { h->lpt_snake = (int)value; h->tune_epoch++; return PG_OK; }      // takes effect at the next pg_prefill
    if (!strcmp(key, "fuse_rope")) { h->fuse_rope = value != 0; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "prefill_res_epi")) { h->prefill_res_epi = value != 0; return PG_OK; }
    if (!strcmp(key, "prefill_rope_epi")) { h->prefill_rope_epi = value != 0; return PG_OK; }
    if (!strcmp(key, "attn_variant")) { h->tune.attn_variant = (int)value; h->tune_epoch++; return PG_OK; }
    if (!strcmp(key, "prefill_attn")) { h->tune.prefill_attn = (int)value; h->tune_epoch++; return PG_OK; }
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
    else if (nm == "pf_stats") { src = h->d_pfstats; n = 16; }
    else if (nm == "vit_feat" && h->cfg.with_vision) {   // SigLIP features (after the final LayerNorm, compute dtype) of the last pg_vision_encode
        const int64_t P = (h->cfg.vit_img / h->cfg.vit_patch) * (h->cfg.vit_img / h->cfg.vit_patch);
        src = h->vt; n = (int64_t)h->cfg.max_vision_images * P * h->cfg.vit_width * h->esz; }
    else { h->err = "pg_debug_read: unknown buffer " + nm; return PG_ERR_NAME; }
    if (n > max_bytes) n = max_bytes;
    (void)hipSetDevice(h->dev);
    if (hipMemcpyAsync(dst_dev, src, (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess) { h->err = "pg_debug_read: copy failed"; return PG_ERR_HIP; }
    return PG_OK;
}

// Measurement / stress only (tools/sk4_load_stress.py): the run-ahead weight stream kernel FREE-RUNNING over the engine's decode weights on
// the handle's side stream (``passes`` sweeps of every layer's list; depth < 0 = register-destination loads instead of LDS-DMA).
int pg_bench_background_stream(pg_handle h, int passes, int blocks, int depth, int nt) {
    if (!h || !h->pf_plan_ok) return PG_ERR_STATE;
    (void)hipSetDevice(h->dev);
    launch_weight_prefetch(h->pf_stream, h->d_pfplan, h->cfg.n_layers, h->d_prog, passes, 0, 0, 0, blocks > 0 ? blocks : 256, depth, nt, h->d_pfstats);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ERR_HIP;
}
int pg_bench_background_wait(pg_handle h) {
    if (!h) return PG_ERR_ARG;
    (void)hipSetDevice(h->dev);
    return hipStreamSynchronize(h->pf_stream) == hipSuccess ? PG_OK : PG_ERR_HIP;
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
