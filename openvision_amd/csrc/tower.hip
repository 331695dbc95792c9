// tower.hip — host-side launch sequences: the resblock loop and the two encoders (no device code here).
//
//   ov_tower_forward        Transformer.forward / ResidualAttentionBlock.forward   open_clip/transformer.py:355-366, 254-265
//   ov_vision_embed         conv1 + cls + pos-emb (ln_pre = Identity)              transformer.py:610-620
//   ov_vision_head_forward  _global_pool -> ln_post -> @ proj (-> F.normalize)     transformer.py:638-646, model.py:267
//   ov_encode_image         VisionTransformer.forward                              transformer.py:609-651
//   ov_encode_text          CLIP.encode_text                                       model.py:269-284
//
// Per block, seven launches on the caller's stream (all asynchronous, nothing allocated):
//   LN1 -> QKV GEMM(+bias) -> attention -> out-proj GEMM(+bias +residual, in place)
//   LN2 -> FC GEMM(+bias +GELU) -> proj GEMM(+bias +residual, in place)
// Workspace per tower call: h [M, D] | big [M, max(3D, mlp_pad)]  (qkv and the MLP hidden alias).
#include "common.h"
#include <new>
#include <vector>
#include <mutex>
#include <stdlib.h>

// Side stream for the "tail" images of a batch (tile-quantisation fix, see ov_tower_forward): owned by the tower, created on the
// device that is current at the first forward that needs it; a forward on another device does not split.
struct TailCtx {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int state = 0;                         // 0 = not created, 1 = ok, -1 = unavailable
    int device = -1;
    std::mutex mu;                         // held from fork to join: two host threads on one tower must not interleave them
};

struct ov_tower {
    mutable TailCtx tail;
    ov_tower_cfg cfg;
    ov_block_weights* blocks;
    unsigned char* set;
    ov_block_fp8* fp8;            // optional fp8 copies (config #5); the fp8 path runs when every layer has one
    unsigned char* set8;
    float* h_amax;                // fp8 path: [4 * layers]: scales' maxima (hidden | attention out) and the running ones (device, borrowed)
    int h_mode;
    unsigned char* mask8;         // fp8 path: per layer, which of the four GEMMs take e4m3 operands (OV_FP8_QKV | _OUT | _FC | _PROJ)
};

// Row pitch (elements) of the workspace's `big` region (qkv / MLP hidden).  max(3 D, mlp_pad) is a power-of-two number of bytes for
// the usual widths (L/14: 8 KiB), and the 256 rows of a GEMM operand piece then start 8 KiB apart: OVHIP_BIG_PAD elements (a multiple
// of 64 = one 128-byte line; default 64) are added to the pitch to spread them over the memory channels (L/14 step 45.09 -> 44.85 ms).
static inline int big_pitch(const ov_tower_cfg& c) {
    static int pad = -1;
    if (pad < 0) { const char* e = getenv("OVHIP_BIG_PAD"); pad = e ? atoi(e) : 64; if (pad < 0 || pad % 64) pad = 64; }
    return (3 * c.width > c.mlp_pad ? 3 * c.width : c.mlp_pad) + pad;
}

// OVHIP_ROWPARTS=1: LayerNorm row statistics from the residual GEMMs' epilogues (ov_gemm_rowparts + ov_rowstats_finalize) instead of a
// pass over the residual stream in front of every folded GEMM.  OFF by default: on the L/14 step the row passes shrink from 1.61 to
// 0.59 ms and the two residual GEMM classes grow by 0.57 ms, but the step moves by 0.1 ms only (44.58 against 44.66 ms, alternating
// runs) -- the memory-bound row passes were pauses in which the power-bound chip recovered clock for the next GEMM -- and the default
// keeps the two-pass statistics.
static inline bool use_rowparts() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OVHIP_ROWPARTS"); v = (e && e[0] == '1') ? 1 : 0; }
    return v != 0;
}

namespace {
// ---- optional in-situ kernel timing (HIP events on the caller's stream; off by default) -----------------
struct ProfRec { int cls; hipEvent_t e0, e1; int64_t rows; };
struct Profiler {
    std::mutex mu;
    unsigned mask = 0;
    int device = -1;                   // the device the event pool belongs to: launches on other devices are not recorded
    std::vector<hipEvent_t> pool;      // unused events
    std::vector<ProfRec> recs;
} g_prof;

struct ProfScope {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    int cls;
    int64_t rows;
    bool on = false;
    ProfScope(int c, ov_stream_t s, int64_t r) : st((hipStream_t)s), cls(c), rows(r) {
        if (!(g_prof.mask & (1u << c))) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.pool.size() < 2 || g_prof.device != ov_current_device()) return;
        e0 = g_prof.pool.back(); g_prof.pool.pop_back();
        e1 = g_prof.pool.back(); g_prof.pool.pop_back();
        on = true;
        (void)hipEventRecord(e0, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(e1, st);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.recs.push_back({cls, e0, e1, rows});
    }
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int max_i(int a, int b) { return a > b ? a : b; }

struct VisionWs { size_t x, tower, pooled, lnrow, feat, total; };
inline VisionWs vision_ws(const ov_tower* t, const ov_vision_head* h, int B) {
    const int g = h->image_size / h->patch_size, L = g * g + 1, D = t->cfg.width;
    VisionWs w;
    size_t off = 0;
    w.x = off;      off += align_up((size_t)B * L * D * 2, 256);
    w.tower = off;  off += align_up(ov_tower_workspace_bytes(t, B, L), 256);
    w.pooled = off; off += align_up((size_t)B * D * 4, 256);
    w.lnrow = off;  off += align_up((size_t)B * D * 2, 256);
    w.feat = off;   off += align_up((size_t)B * h->embed_pad * 2, 256);
    w.total = off;
    return w;
}
struct TextWs { size_t x, tower, last, lnrow, feat, total; };
inline TextWs text_ws(const ov_tower* t, const ov_text_head* h, int B) {
    const int T = h->context_length, D = t->cfg.width;
    const int epad = (h->embed_dim + 7) / 8 * 8;
    TextWs w;
    size_t off = 0;
    w.x = off;     off += align_up((size_t)B * T * D * 2, 256);
    w.tower = off; off += align_up(ov_tower_workspace_bytes(t, B, T), 256);
    w.last = off;  off += align_up((size_t)B * D * 2, 256);
    w.lnrow = off; off += align_up((size_t)B * D * 2, 256);
    w.feat = off;  off += align_up((size_t)B * epad * 2, 256);
    w.total = off;
    return w;
}
}  // namespace

extern "C" int ov_abi_version(void) { return OV_ABI_VERSION; }

extern "C" const char* ov_error_string(int status) {
    switch (status) {
        case OV_OK: return "ok";
        case OV_ERR_INVALID: return "invalid argument";
        case OV_ERR_UNSUPPORTED: return "shape/dtype not supported by the gfx950 kernels";
        case OV_ERR_WORKSPACE: return "workspace too small";
        case OV_ERR_NO_DEVICE: return "no gfx950 (MI355X) device visible";
        default: break;
    }
    if (status <= OV_ERR_HIP) return hipGetErrorString((hipError_t)(OV_ERR_HIP - status));
    return "unknown error";
}

extern "C" int ov_device_check(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return OV_ERR_HIP - (int)e;
    if (n <= 0) return OV_ERR_NO_DEVICE;
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return OV_ERR_HIP - (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return OV_ERR_HIP - (int)e;
    const char* arch = p.gcnArchName;
    if (!(arch[0] == 'g' && arch[1] == 'f' && arch[2] == 'x' && arch[3] == '9' && arch[4] == '5' && arch[5] == '0'))
        return OV_ERR_NO_DEVICE;
    return OV_OK;
}

extern "C" int ov_profile_enable(unsigned class_mask, int max_records) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& r : g_prof.recs) { g_prof.pool.push_back(r.e0); g_prof.pool.push_back(r.e1); }
    g_prof.recs.clear();
    g_prof.mask = class_mask;
    if (max_records < 0) return OV_ERR_INVALID;
    const int dev = ov_current_device();
    if (g_prof.device != dev) {                       // events are per device: a pool made elsewhere is dropped
        for (hipEvent_t e : g_prof.pool) (void)hipEventDestroy(e);
        g_prof.pool.clear();
        g_prof.device = dev;
    }
    while ((int)g_prof.pool.size() < 2 * max_records) {
        hipEvent_t e;
        hipError_t err = hipEventCreate(&e);
        if (err != hipSuccess) return OV_ERR_HIP - (int)err;
        g_prof.pool.push_back(e);
    }
    return OV_OK;
}

extern "C" int ov_profile_read(int cls, double* total_ms, int* count, double* total_rows) {
    if (!total_ms || !count) return OV_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double tot = 0.0, rows = 0.0;
    int n = 0;
    for (auto& r : g_prof.recs) {
        if (r.cls != cls) continue;
        hipError_t err = hipEventSynchronize(r.e1);
        if (err != hipSuccess) return OV_ERR_HIP - (int)err;
        float ms = 0.f;
        err = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (err != hipSuccess) return OV_ERR_HIP - (int)err;
        tot += ms;
        rows += (double)r.rows;
        ++n;
    }
    *total_ms = tot;
    *count = n;
    if (total_rows) *total_rows = rows;
    return OV_OK;
}

extern "C" ov_tower* ov_tower_create(const ov_tower_cfg* cfg) {
    if (!cfg || cfg->width <= 0 || cfg->layers <= 0 || cfg->heads <= 0 || cfg->mlp <= 0) return nullptr;
    if (cfg->width % cfg->heads || cfg->width % 64 || cfg->mlp_pad % 64 || cfg->mlp_pad < cfg->mlp) return nullptr;
    ov_tower* t = new (std::nothrow) ov_tower;
    if (!t) return nullptr;
    t->cfg = *cfg;
    t->h_amax = nullptr;
    t->h_mode = 0;
    t->blocks = new (std::nothrow) ov_block_weights[cfg->layers]();
    t->set = new (std::nothrow) unsigned char[cfg->layers]();
    t->fp8 = new (std::nothrow) ov_block_fp8[cfg->layers]();
    t->set8 = new (std::nothrow) unsigned char[cfg->layers]();
    t->mask8 = new (std::nothrow) unsigned char[cfg->layers];
    if (!t->blocks || !t->set || !t->fp8 || !t->set8 || !t->mask8) { ov_tower_destroy(t); return nullptr; }
    for (int i = 0; i < cfg->layers; ++i) t->mask8[i] = OV_FP8_ALL;
    return t;
}

extern "C" void ov_tower_destroy(ov_tower* t) {
    if (!t) return;
    if (t->tail.state == 1) {
        (void)hipEventDestroy(t->tail.fork);
        (void)hipEventDestroy(t->tail.join);
        (void)hipStreamDestroy(t->tail.stream);
    }
    delete[] t->blocks;
    delete[] t->set;
    delete[] t->fp8;
    delete[] t->mask8;
    delete[] t->set8;
    delete t;
}

extern "C" int ov_tower_set_block(ov_tower* t, int layer, const ov_block_weights* w) {
    if (!t || !w || layer < 0 || layer >= t->cfg.layers) return OV_ERR_INVALID;
    const void* p[] = {w->ln1_w, w->ln1_b, w->qkv_w, w->qkv_b, w->out_w, w->out_b, w->ln2_w, w->ln2_b,
                       w->fc_w, w->fc_b, w->proj_w, w->proj_b};
    for (const void* q : p)
        if (!q || ((uintptr_t)q & 15)) return OV_ERR_INVALID;
    if ((w->qkv_colsum == nullptr) != (w->fc_colsum == nullptr)) return OV_ERR_INVALID;
    if (((uintptr_t)w->qkv_colsum | (uintptr_t)w->fc_colsum) & 15) return OV_ERR_INVALID;
    t->blocks[layer] = *w;
    t->set[layer] = 1;
    return OV_OK;
}

namespace {
bool tower_fp8(const ov_tower* t) {
    for (int i = 0; i < t->cfg.layers; ++i)
        if (!t->set8[i]) return false;
    return true;
}
}  // namespace

extern "C" int ov_tower_set_block_fp8(ov_tower* t, int layer, const ov_block_fp8* q) {
    if (!t || layer < 0 || layer >= t->cfg.layers) return OV_ERR_INVALID;
    if (!q) { t->set8[layer] = 0; return OV_OK; }                 // NULL clears: back to the bf16 path
    const int D = t->cfg.width, F = t->cfg.mlp_pad;
    if (D % 128 || F % 128 || D < 384 || F < 384) return OV_ERR_UNSUPPORTED;      // K-tiles of 128 fp8 elements, at least three
    const void* p[] = {q->qkv_w8, q->qkv_s, q->qkv_b, q->out_w8, q->out_s, q->fc_w8, q->fc_s, q->fc_b, q->proj_w8, q->proj_s};
    for (const void* v : p)
        if (!v || ((uintptr_t)v & 15)) return OV_ERR_INVALID;
    t->fp8[layer] = *q;
    t->set8[layer] = 1;
    return OV_OK;
}

extern "C" int ov_tower_set_fp8_hidden_scale(ov_tower* t, float* h_amax, int mode) {
    if (!t || mode < 0 || mode > 3 || (mode > 0 && !h_amax)) return OV_ERR_INVALID;
    t->h_amax = h_amax;
    t->h_mode = mode;
    return OV_OK;
}

extern "C" int ov_tower_set_fp8_mask(ov_tower* t, const unsigned char* mask, int n) {
    if (!t || (mask && n != t->cfg.layers)) return OV_ERR_INVALID;
    for (int i = 0; i < t->cfg.layers; ++i) {
        const unsigned char m = mask ? mask[i] : (unsigned char)OV_FP8_ALL;
        if (m & ~OV_FP8_ALL) return OV_ERR_INVALID;
        t->mask8[i] = m;
    }
    return OV_OK;
}

extern "C" size_t ov_tower_workspace_bytes(const ov_tower* t, int B, int L) {
    if (!t || B <= 0 || L <= 0) return 0;
    const size_t M = (size_t)B * L;
    const int D = t->cfg.width;
    size_t n = align_up(M * D * 2, 256) + align_up(M * (size_t)big_pitch(t->cfg) * 2, 256) + align_up(M * 8, 256) +
               align_up(M * (size_t)(D / 32) * 8, 256);            // h | big | row statistics | their partial sums (last)
    if (tower_fp8(t)) n += align_up(M * (size_t)max_i(D, t->cfg.mlp_pad), 256) + align_up(M * 4, 256);   // fp8 activations + row scales
    return n;
}

namespace {
// One ResidualAttentionBlock on rows [0, B*L) of x (in place).  `prof` = record in-situ timings for these launches.
// `parts` (or NULL): partial sums of x's row statistics (ov_rowparts layout).  parts_in: they describe x on entry (left by the previous
// block's c_proj); they always describe x on exit.
int run_block(const ov_tower_cfg& c, const ov_block_weights& w, ov_bf16* x, ov_bf16* h, ov_bf16* big, float* stats, float* parts,
              bool parts_in, int B, int L, ov_stream_t stream, bool prof) {
    const int D = c.width, H = c.heads, hd = D / H;
    const int64_t M = (int64_t)B * L;
    const int ldb = big_pitch(c);                               // row pitch of `big` (shared by qkv and the MLP hidden)
    const float scale = 1.0f / sqrtf((float)hd);
    const int gelu = c.gelu_tanh ? OV_EPI_BIAS_GELU_TANH : OV_EPI_BIAS_GELU_ERF;
    const int fc_cls = c.gelu_tanh ? OV_PROF_GEMM_FC_TANH : OV_PROF_GEMM_FC;
    const bool fold = w.qkv_colsum != nullptr && w.fc_colsum != nullptr;   // LN folded into the QKV / c_fc epilogues
    int rc;
#define OV_STEP(cls, call)                                           \
    do {                                                             \
        if (prof) { ProfScope ps__(cls, stream, M); rc = (call); }   \
        else rc = (call);                                            \
        if (rc) return rc;                                           \
    } while (0)
    const bool rp = fold && parts != nullptr;                   // statistics ride on the residual GEMMs' epilogues
    if (fold) {
        if (rp && parts_in) OV_STEP(OV_PROF_LN, ov_rowstats_finalize(parts, stats, M, D, c.ln_eps, stream));
        else OV_STEP(OV_PROF_LN, ov_rowstats(x, D, stats, M, D, c.ln_eps, stream));
        OV_STEP(OV_PROF_GEMM_QKV, ov_gemm_ln(x, D, w.qkv_w, D, w.qkv_b, w.qkv_colsum, stats, big, ldb, M, 3 * D, D, OV_EPI_BIAS, stream));
    } else {
        OV_STEP(OV_PROF_LN, ov_layernorm(x, OV_BF16, D, w.ln1_w, w.ln1_b, h, OV_BF16, D, M, D, c.ln_eps, stream));
        OV_STEP(OV_PROF_GEMM_QKV, ov_gemm(h, D, w.qkv_w, D, w.qkv_b, big, ldb, M, 3 * D, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream));
    }
    OV_STEP(OV_PROF_ATTN, ov_attention(big, ldb, h, D, B, L, H, hd, scale, stream));
    if (rp) OV_STEP(OV_PROF_GEMM_OUT, ov_gemm_rowparts(h, D, w.out_w, D, w.out_b, x, D, M, D, D, x, D, parts, stream));
    else OV_STEP(OV_PROF_GEMM_OUT, ov_gemm(h, D, w.out_w, D, w.out_b, x, D, M, D, D, OV_EPI_BIAS_RESIDUAL, x, D, 0, 0, 0, stream));
    if (fold) {
        if (rp) OV_STEP(OV_PROF_LN, ov_rowstats_finalize(parts, stats, M, D, c.ln_eps, stream));
        else OV_STEP(OV_PROF_LN, ov_rowstats(x, D, stats, M, D, c.ln_eps, stream));
        OV_STEP(fc_cls, ov_gemm_ln(x, D, w.fc_w, D, w.fc_b, w.fc_colsum, stats, big, ldb, M, c.mlp_pad, D, gelu, stream));
    } else {
        OV_STEP(OV_PROF_LN, ov_layernorm(x, OV_BF16, D, w.ln2_w, w.ln2_b, h, OV_BF16, D, M, D, c.ln_eps, stream));
        OV_STEP(fc_cls, ov_gemm(h, D, w.fc_w, D, w.fc_b, big, ldb, M, c.mlp_pad, D, gelu, nullptr, 0, 0, 0, 0, stream));
    }
    if (rp) OV_STEP(OV_PROF_GEMM_PROJ, ov_gemm_rowparts(big, ldb, w.proj_w, c.mlp_pad, w.proj_b, x, D, M, D, c.mlp_pad, x, D, parts, stream));
    else OV_STEP(OV_PROF_GEMM_PROJ, ov_gemm(big, ldb, w.proj_w, c.mlp_pad, w.proj_b, x, D, M, D, c.mlp_pad, OV_EPI_BIAS_RESIDUAL, x,
                                            D, 0, 0, 0, stream));
#undef OV_STEP
    return OV_OK;
}

// The tower's tail context if it is usable on the CURRENT device (created on first use), else nullptr (= do not split).
TailCtx* tail_ctx(const ov_tower* t) {
    TailCtx& tc = t->tail;
    std::lock_guard<std::mutex> lk(tc.mu);
    if (tc.state == 0) {
        const char* e = getenv("OVHIP_NO_TAIL_SPLIT");
        if (e && e[0] == '1') { tc.state = -1; return nullptr; }
        tc.device = ov_current_device();
        if (hipStreamCreateWithFlags(&tc.stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&tc.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&tc.join, hipEventDisableTiming) == hipSuccess)
            tc.state = 1;
        else
            tc.state = -1;
    }
    return tc.state == 1 && tc.device == ov_current_device() ? &tc : nullptr;
}

// Images to peel off so that the main part's 256-row tile count is a multiple of 64 (x 4 column tiles = whole rounds of
// 256 CUs for the N = width GEMMs).  0 = do not split.
int tail_images(int B, int L) {
    const int64_t T = ((int64_t)B * L + 255) / 256;
    const int r = (int)(T % 64);
    if (T <= 64 || r == 0 || r > 8) return 0;
    const int64_t target = T - r;                                // tiles the main part may use
    int Bm = (int)((target * 256) / L);                           // largest B' with ceil(B'*L/256) <= target
    while (Bm > 0 && ((int64_t)Bm * L + 255) / 256 > target) --Bm;
    const int tail = B - Bm;
    if (Bm <= 0 || tail <= 0 || tail > B / 8) return 0;
    return tail;
}
}  // namespace

namespace {
// The same block with fp8 (e4m3) operands on the GEMMs `mask` names (BASELINE.json config #5; OV_FP8_ALL = all four): in front of an
// fp8 QKV / c_fc the LayerNorm is fused with the row quantisation; in front of an fp8 out_proj / c_proj the attention output / MLP
// hidden is written as e4m3 by its producer where that producer has a static scale (h_mode >= 2: attention epilogue for head_dim 64,
// an fp8 c_fc's epilogue) and re-quantised row by row otherwise.  A GEMM outside the mask runs exactly as in run_block (LN fold
// included), so mask 0 is the bf16 block.
int run_block_fp8(const ov_tower_cfg& c, const ov_block_weights& w, const ov_block_fp8& q, int mask, ov_bf16* x, ov_bf16* h, ov_bf16* big,
                  float* stats, unsigned char* q8, float* qs, float* h_amax, float* a_amax, float* h_next, float* a_next, int h_mode, int B,
                  int L, ov_stream_t stream, bool prof) {
    const int D = c.width, H = c.heads, hd = D / H, F = c.mlp_pad;
    const int64_t M = (int64_t)B * L;
    const int ldb = big_pitch(c);
    const float scale = 1.0f / sqrtf((float)hd);
    const int gelu = c.gelu_tanh ? OV_EPI_BIAS_GELU_TANH : OV_EPI_BIAS_GELU_ERF;
    const int fc_cls = c.gelu_tanh ? OV_PROF_GEMM_FC_TANH : OV_PROF_GEMM_FC;
    const bool fold = w.qkv_colsum != nullptr && w.fc_colsum != nullptr;
    const bool f_qkv = mask & OV_FP8_QKV, f_out = mask & OV_FP8_OUT, f_fc = mask & OV_FP8_FC, f_proj = mask & OV_FP8_PROJ;
    int rc;
#define OV_STEP(cls, call)                                           \
    do {                                                             \
        if (prof) { ProfScope ps__(cls, stream, M); rc = (call); }   \
        else rc = (call);                                            \
        if (rc) return rc;                                           \
    } while (0)
    // ---- attention half ----
    if (f_qkv) {
        OV_STEP(OV_PROF_LN, ov_layernorm_quant_fp8(x, D, w.ln1_w, w.ln1_b, q8, D, qs, M, D, c.ln_eps, stream));
        OV_STEP(OV_PROF_GEMM_QKV, ov_gemm_fp8(q8, D, q.qkv_w8, D, qs, q.qkv_s, q.qkv_b, big, ldb, M, 3 * D, D, OV_EPI_BIAS, nullptr, 0, stream));
    } else if (fold) {
        OV_STEP(OV_PROF_LN, ov_rowstats(x, D, stats, M, D, c.ln_eps, stream));
        OV_STEP(OV_PROF_GEMM_QKV, ov_gemm_ln(x, D, w.qkv_w, D, w.qkv_b, w.qkv_colsum, stats, big, ldb, M, 3 * D, D, OV_EPI_BIAS, stream));
    } else {
        OV_STEP(OV_PROF_LN, ov_layernorm(x, OV_BF16, D, w.ln1_w, w.ln1_b, h, OV_BF16, D, M, D, c.ln_eps, stream));
        OV_STEP(OV_PROF_GEMM_QKV, ov_gemm(h, D, w.qkv_w, D, w.qkv_b, big, ldb, M, 3 * D, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream));
    }
    if (f_out && h_mode >= 2 && hd == 64) {
        // static scale: the attention epilogue writes e4m3 itself (into the fp8 activation buffer, free at this point)
        OV_STEP(OV_PROF_ATTN, ov_attention_fp8out(big, ldb, q8, D, B, L, H, hd, scale, a_amax, a_next, stream));
        OV_STEP(OV_PROF_GEMM_OUT, ov_gemm_fp8_static(q8, D, q.out_w8, D, nullptr, a_amax, q.out_s, w.out_b, x, D, nullptr, nullptr, M, D, D,
                                                     OV_EPI_BIAS_RESIDUAL, x, D, stream));
    } else {
        OV_STEP(OV_PROF_ATTN, ov_attention(big, ldb, h, D, B, L, H, hd, scale, stream));
        if (f_out) {
            OV_STEP(OV_PROF_LN, ov_quant_rows_fp8(h, D, q8, D, qs, M, D, h_mode == 1 ? a_amax : nullptr, stream));
            OV_STEP(OV_PROF_GEMM_OUT, ov_gemm_fp8(q8, D, q.out_w8, D, qs, q.out_s, w.out_b, x, D, M, D, D, OV_EPI_BIAS_RESIDUAL, x, D, stream));
        } else {
            OV_STEP(OV_PROF_GEMM_OUT, ov_gemm(h, D, w.out_w, D, w.out_b, x, D, M, D, D, OV_EPI_BIAS_RESIDUAL, x, D, 0, 0, 0, stream));
        }
    }
    // ---- MLP half ----
    if (f_fc) {
        OV_STEP(OV_PROF_LN, ov_layernorm_quant_fp8(x, D, w.ln2_w, w.ln2_b, q8, D, qs, M, D, c.ln_eps, stream));
    } else if (fold) {
        OV_STEP(OV_PROF_LN, ov_rowstats(x, D, stats, M, D, c.ln_eps, stream));
    } else {
        OV_STEP(OV_PROF_LN, ov_layernorm(x, OV_BF16, D, w.ln2_w, w.ln2_b, h, OV_BF16, D, M, D, c.ln_eps, stream));
    }
    if (f_fc && f_proj && h_mode >= 2) {
        // static hidden scale: c_fc quantises its own output (e4m3 bytes, pitch F, in the `big` region), c_proj reads it as is
        unsigned char* h8 = (unsigned char*)big;
        OV_STEP(fc_cls, ov_gemm_fp8_static(q8, D, q.fc_w8, D, qs, nullptr, q.fc_s, q.fc_b, h8, F, h_amax, h_next, M, F, D, gelu, nullptr, 0, stream));
        OV_STEP(OV_PROF_GEMM_PROJ, ov_gemm_fp8_static(h8, F, q.proj_w8, F, nullptr, h_amax, q.proj_s, w.proj_b, x, D, nullptr, nullptr, M, D, F,
                                                      OV_EPI_BIAS_RESIDUAL, x, D, stream));
    } else {
        if (f_fc) OV_STEP(fc_cls, ov_gemm_fp8(q8, D, q.fc_w8, D, qs, q.fc_s, q.fc_b, big, ldb, M, F, D, gelu, nullptr, 0, stream));
        else if (fold) OV_STEP(fc_cls, ov_gemm_ln(x, D, w.fc_w, D, w.fc_b, w.fc_colsum, stats, big, ldb, M, F, D, gelu, stream));
        else OV_STEP(fc_cls, ov_gemm(h, D, w.fc_w, D, w.fc_b, big, ldb, M, F, D, gelu, nullptr, 0, 0, 0, 0, stream));
        if (f_proj) {
            OV_STEP(OV_PROF_LN, ov_quant_rows_fp8(big, ldb, q8, F, qs, M, F, (h_mode == 1 && f_fc) ? h_amax : nullptr, stream));
            OV_STEP(OV_PROF_GEMM_PROJ, ov_gemm_fp8(q8, F, q.proj_w8, F, qs, q.proj_s, w.proj_b, x, D, M, D, F, OV_EPI_BIAS_RESIDUAL, x, D, stream));
        } else {
            OV_STEP(OV_PROF_GEMM_PROJ, ov_gemm(big, ldb, w.proj_w, F, w.proj_b, x, D, M, D, F, OV_EPI_BIAS_RESIDUAL, x, D, 0, 0, 0, stream));
        }
    }
#undef OV_STEP
    return OV_OK;
}
}  // namespace

// x[B*L, D] is updated in place through all blocks.  When B*L leaves a few 256-row tiles over a whole number of rounds
// (L/14 at B = 256: 257 tiles -> the out-proj / c_proj GEMMs need a 5th round for 4 of their 1028 tiles), the last
// image(s) are peeled off and run, layer by layer, on an internal side stream: rows are independent through LN/GEMM and
// attention never crosses images, so the split is exact; the main part then fills whole rounds and the tail's small
// kernels slot into idle CUs.  Fork/join by events; everything remains ordered with respect to the caller's stream.
extern "C" int ov_tower_forward(const ov_tower* t, ov_bf16* x, int B, int L, void* workspace, size_t workspace_bytes,
                                ov_stream_t stream) {
    if (!t || !x || !workspace || B <= 0 || L <= 0) return OV_ERR_INVALID;
    if (workspace_bytes < ov_tower_workspace_bytes(t, B, L)) return OV_ERR_WORKSPACE;
    if (((uintptr_t)x | (uintptr_t)workspace) & 15) return OV_ERR_INVALID;
    const ov_tower_cfg& c = t->cfg;
    const int D = c.width;
    const int64_t M = (int64_t)B * L;
    const int ldb = big_pitch(c);
    ov_bf16* h = (ov_bf16*)workspace;
    ov_bf16* big = (ov_bf16*)((char*)workspace + align_up((size_t)M * D * 2, 256));
    float* stats = (float*)((char*)big + align_up((size_t)M * ldb * 2, 256));   // {mean, rstd} per row (LN fold)
    for (int i = 0; i < c.layers; ++i)
        if (!t->set[i]) return OV_ERR_INVALID;

    int nt = tail_images(B, L);
    TailCtx* tc = nt > 0 ? tail_ctx(t) : nullptr;
    if (!tc) nt = 0;
    const int Bm = B - nt;
    hipStream_t main_st = (hipStream_t)stream;
    // fork .. join under the tower's mutex: the events are this tower's own, a second host thread enqueues after the join
    std::unique_lock<std::mutex> tail_lock;
    if (nt > 0) {
        tail_lock = std::unique_lock<std::mutex>(tc->mu);
        hipError_t e = hipEventRecord(tc->fork, main_st);
        if (e == hipSuccess) e = hipStreamWaitEvent(tc->stream, tc->fork, 0);
        if (e != hipSuccess) return OV_ERR_HIP - (int)e;
    }
    const int64_t off = (int64_t)Bm * L;
    const bool fp8 = tower_fp8(t);
    int rc = OV_OK;
    if (fp8 && t->h_amax && t->h_mode == 2)            // delayed scaling: last forward's maxima become this forward's scales
        rc = ov_amax_roll(t->h_amax, t->h_amax + 2 * c.layers, 2 * c.layers, stream);
    const int G = D / 32;
    float* parts = use_rowparts() && !fp8 ? (float*)((char*)workspace + ov_tower_workspace_bytes(t, B, L) - align_up((size_t)M * G * 8, 256)) : nullptr;
    const int qw = D > c.mlp_pad ? D : c.mlp_pad;                 // row pitch reserved per token in the fp8 activation buffer
    unsigned char* q8 = (unsigned char*)stats + align_up((size_t)M * 8, 256);
    float* qs = (float*)(q8 + align_up((size_t)M * qw, 256));
    for (int i = 0; i < c.layers && rc == OV_OK; ++i) {
        float* ha = t->h_amax ? t->h_amax + i : nullptr;
        float* aa = t->h_amax ? t->h_amax + c.layers + i : nullptr;
        float* hn = t->h_amax ? t->h_amax + 2 * c.layers + i : nullptr;
        float* an = t->h_amax ? t->h_amax + 3 * c.layers + i : nullptr;
        const int hm = t->h_amax ? t->h_mode : 0;
        rc = fp8 ? run_block_fp8(c, t->blocks[i], t->fp8[i], t->mask8[i], x, h, big, stats, q8, qs, ha, aa, hn, an, hm, Bm, L, stream, true)
                 : run_block(c, t->blocks[i], x, h, big, stats, parts, i > 0, Bm, L, stream, true);
        if (rc == OV_OK && nt > 0) {
            rc = fp8 ? run_block_fp8(c, t->blocks[i], t->fp8[i], t->mask8[i], x + off * D, h + off * D, big + off * ldb, stats + 2 * off,
                                     q8 + off * qw, qs + off, ha, aa, hn, an, hm, nt, L, (ov_stream_t)tc->stream, false)
                     : run_block(c, t->blocks[i], x + off * D, h + off * D, big + off * ldb, stats + 2 * off,
                                 parts ? parts + 2 * off * G : nullptr, i > 0, nt, L, (ov_stream_t)tc->stream, false);
        }
    }
    if (nt > 0) {
        // always join, also after an error between fork and here: whatever the side stream got stays ordered before the
        // caller's next work on `stream` (and before the workspace is reused)
        hipError_t e = hipEventRecord(tc->join, tc->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(main_st, tc->join, 0);
        if (e != hipSuccess && rc == OV_OK) rc = OV_ERR_HIP - (int)e;
    }
    return rc;
}

// ---- training-side entry points: keep every block's input, run the blocks' backward in reverse -------------
extern "C" int ov_block_backward(const ov_tower_cfg* cfg, const ov_block_weights* w, const ov_bf16* x, const ov_block_saved* saved,
                                 const ov_bf16* dy, ov_bf16* dx, const ov_block_grads* g, int B, int L, void* workspace,
                                 size_t workspace_bytes, ov_stream_t stream);
extern "C" size_t ov_block_backward_workspace_bytes(const ov_tower_cfg* cfg, int B, int L);

// saved activations of one layer: [x | qkv | attention out | x1 | ln_1 out | ln_2 out | c_fc pre-activation | c_fc activation],
// 8 D + 2 mlp_pad bf16 per token (288 GB of HBM: keep, do not recompute)
// + per layer the attention's row log-sum-exp, fp32 [B * heads][L rounded up to 32] (ov_attention_lse; used by the backward where the
// resident kernel applies: head_dim 64, L <= 288)
static inline size_t saved_per_token(const ov_tower_cfg& c) { return (size_t)8 * c.width + 2 * (size_t)c.mlp_pad; }
static inline size_t saved_lse_elems(const ov_tower_cfg& c, int B, int L) {      // in bf16 elements, a multiple of 8 (16-byte sections)
    return ((size_t)2 * B * c.heads * ((L + 31) / 32 * 32) + 7) / 8 * 8;
}
static inline size_t saved_per_layer(const ov_tower_cfg& c, int B, int L) { return (size_t)B * L * saved_per_token(c) + saved_lse_elems(c, B, L); }
static inline bool saved_lse_used(const ov_tower_cfg& c, int L) { return c.width / c.heads == 64 && (L + 31) / 32 * 32 <= 288; }
extern "C" size_t ov_tower_saved_bytes(const ov_tower* t, int B, int L) {
    if (!t || B <= 0 || L <= 0) return 0;
    return (size_t)t->cfg.layers * saved_per_layer(t->cfg, B, L) * sizeof(ov_bf16);
}

extern "C" int ov_tower_forward_saving(const ov_tower* t, ov_bf16* x, ov_bf16* saved, int B, int L, void* workspace,
                                       size_t workspace_bytes, ov_stream_t stream) {
    if (!t || !x || !saved || !workspace || B <= 0 || L <= 0) return OV_ERR_INVALID;
    if (tower_fp8(t)) return OV_ERR_UNSUPPORTED;                   // the backward differentiates the bf16 path
    if (workspace_bytes < ov_tower_workspace_bytes(t, B, L)) return OV_ERR_WORKSPACE;
    if (((uintptr_t)x | (uintptr_t)saved | (uintptr_t)workspace) & 15) return OV_ERR_INVALID;
    const ov_tower_cfg& c = t->cfg;
    const int D = c.width, H = c.heads, hd = D / H;
    const int64_t M = (int64_t)B * L;
    const float scale = 1.0f / sqrtf((float)hd);              // (every intermediate lands in `saved`: the workspace stays unused)
    const int gelu = c.gelu_tanh ? OV_EPI_BIAS_GELU_TANH : OV_EPI_BIAS_GELU_ERF;
    const size_t spl = saved_per_layer(c, B, L);
    const bool keep_lse = saved_lse_used(c, L);
    for (int i = 0; i < c.layers; ++i)
        if (!t->set[i] || t->blocks[i].qkv_colsum || t->blocks[i].fc_colsum) return OV_ERR_INVALID;   // the module's own weights
    // layer i reads its input from its own saved slot and writes its output straight into layer i+1's slot (the last one into x):
    // one copy for the whole tower
    hipError_t e = hipMemcpyAsync(saved, x, (size_t)M * D * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return OV_ERR_HIP - (int)e;
    for (int i = 0; i < c.layers; ++i) {
        const ov_block_weights& w = t->blocks[i];
        ov_bf16* sx = saved + (size_t)i * spl;
        ov_bf16* sqkv = sx + (size_t)M * D;
        ov_bf16* so = sqkv + (size_t)M * 3 * D;
        ov_bf16* sx1 = so + (size_t)M * D;
        ov_bf16* sn1 = sx1 + (size_t)M * D;
        ov_bf16* sn2 = sn1 + (size_t)M * D;
        ov_bf16* spre = sn2 + (size_t)M * D;
        ov_bf16* sact = spre + (size_t)M * c.mlp_pad;
        float* slse = (float*)(sact + (size_t)M * c.mlp_pad);
        ov_bf16* y = i + 1 < c.layers ? saved + (size_t)(i + 1) * spl : x;
        int rc;
        // the same operator sequence as run_block, with qkv / attention output / x1 written where the backward will read them
        if ((rc = ov_layernorm(sx, OV_BF16, D, w.ln1_w, w.ln1_b, sn1, OV_BF16, D, M, D, c.ln_eps, stream))) return rc;
        if ((rc = ov_gemm(sn1, D, w.qkv_w, D, w.qkv_b, sqkv, 3 * D, M, 3 * D, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream))) return rc;
        rc = keep_lse ? ov_attention_lse(sqkv, 3 * D, so, D, slse, B, L, H, hd, scale, stream) : ov_attention(sqkv, 3 * D, so, D, B, L, H, hd, scale, stream);
        if (rc) return rc;
        if ((rc = ov_gemm(so, D, w.out_w, D, w.out_b, sx1, D, M, D, D, OV_EPI_BIAS_RESIDUAL, sx, D, 0, 0, 0, stream))) return rc;
        if ((rc = ov_layernorm(sx1, OV_BF16, D, w.ln2_w, w.ln2_b, sn2, OV_BF16, D, M, D, c.ln_eps, stream))) return rc;
        if ((rc = ov_gemm_keep(sn2, D, w.fc_w, D, w.fc_b, sact, c.mlp_pad, spre, c.mlp_pad, M, c.mlp_pad, D, gelu, stream))) return rc;
        if ((rc = ov_gemm(sact, c.mlp_pad, w.proj_w, c.mlp_pad, w.proj_b, y, D, M, D, c.mlp_pad, OV_EPI_BIAS_RESIDUAL, sx1, D, 0, 0, 0, stream)))
            return rc;
    }
    return OV_OK;
}

extern "C" size_t ov_tower_backward_workspace_bytes(const ov_tower* t, int B, int L) {
    if (!t) return 0;
    return ov_block_backward_workspace_bytes(&t->cfg, B, L);
}

extern "C" int ov_tower_backward(const ov_tower* t, const ov_bf16* saved, ov_bf16* dx, const ov_block_grads* grads, int B, int L,
                                 void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!t || !saved || !dx || !grads || !workspace || B <= 0 || L <= 0) return OV_ERR_INVALID;
    const ov_tower_cfg& c = t->cfg;
    const int D = c.width;
    const int64_t M = (int64_t)B * L;
    for (int i = 0; i < c.layers; ++i)
        if (!t->set[i]) return OV_ERR_INVALID;
    for (int i = c.layers - 1; i >= 0; --i) {                     // dx holds d(block output) on entry and d(block input) on exit
        const ov_bf16* sx = saved + (size_t)i * saved_per_layer(c, B, L);
        ov_block_saved sv;
        sv.qkv = sx + (size_t)M * D;
        sv.attn_out = sv.qkv + (size_t)M * 3 * D;
        sv.x1 = sv.attn_out + (size_t)M * D;
        sv.ln1_out = sv.x1 + (size_t)M * D;
        sv.ln2_out = sv.ln1_out + (size_t)M * D;
        sv.fc_pre = sv.ln2_out + (size_t)M * D;
        sv.fc_act = sv.fc_pre + (size_t)M * c.mlp_pad;
        sv.attn_lse = saved_lse_used(c, L) ? (const float*)(sv.fc_act + (size_t)M * c.mlp_pad) : nullptr;
        const int rc = ov_block_backward(&c, &t->blocks[i], sx, &sv, dx, dx, &grads[i], B, L, workspace, workspace_bytes, stream);
        if (rc) return rc;
    }
    return OV_OK;
}

extern "C" size_t ov_vision_workspace_bytes(const ov_tower* t, const ov_vision_head* h, int B) {
    if (!t || !h || B <= 0 || h->patch_size <= 0) return 0;
    return vision_ws(t, h, B).total;
}

extern "C" int ov_vision_embed(const ov_tower* t, const ov_vision_head* h, const void* image, int img_dtype, int B,
                               ov_bf16* x, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!t || !h || !image || !x || !workspace || B <= 0) return OV_ERR_INVALID;
    const int g = h->image_size / h->patch_size, L = g * g + 1, D = t->cfg.width;
    const size_t need = (size_t)B * g * g * h->kpad * 2;
    if (workspace_bytes < need) return OV_ERR_WORKSPACE;
    if (h->kpad % 64 || h->kpad < 3 * h->patch_size * h->patch_size) return OV_ERR_INVALID;
    ov_bf16* patches = (ov_bf16*)workspace;
    int rc;
    if ((rc = ov_im2col_patches(image, img_dtype, patches, B, h->image_size, h->patch_size, h->kpad, stream))) return rc;
    // conv1 as a GEMM; epilogue adds pos-emb rows 1..g*g (broadcast over the batch) and skips one cls row per image
    if ((rc = ov_gemm(patches, h->kpad, h->conv_w, h->kpad, nullptr, x, D, (int64_t)B * g * g, D, h->kpad,
                      OV_EPI_BIAS_RESIDUAL, h->pos, D, g * g, g * g, 1, stream)))
        return rc;
    return ov_cls_rows(x, D, h->cls, h->pos_f32, B, L, D, stream);
}

extern "C" int ov_vision_head_forward(const ov_tower* t, const ov_vision_head* h, const ov_bf16* x, int B, float* features,
                                      int normalize, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!t || !h || !x || !features || !workspace || B <= 0) return OV_ERR_INVALID;
    const int g = h->image_size / h->patch_size, L = g * g + 1, D = t->cfg.width, E = h->embed_dim, EP = h->embed_pad;
    if (EP % 8 || EP < E) return OV_ERR_INVALID;
    if (!h->final_ln_after_pool) return OV_ERR_UNSUPPORTED;       // OpenVision: pool -> LN (transformer.py:638-640)
    const size_t o_pooled = 0, o_ln = align_up((size_t)B * D * 4, 256), o_feat = o_ln + align_up((size_t)B * D * 2, 256);
    if (workspace_bytes < o_feat + (size_t)B * EP * 2) return OV_ERR_WORKSPACE;
    float* pooled = (float*)((char*)workspace + o_pooled);
    ov_bf16* ln = (ov_bf16*)((char*)workspace + o_ln);
    ov_bf16* feat = (ov_bf16*)((char*)workspace + o_feat);
    int rc;
    if (h->pool_avg) {
        if ((rc = ov_mean_pool(x, D, pooled, B, L, D, 1, stream))) return rc;
        if ((rc = ov_layernorm(pooled, OV_F32, D, h->ln_post_w, h->ln_post_b, ln, OV_BF16, D, B, D, t->cfg.ln_eps, stream))) return rc;
    } else {
        if ((rc = ov_layernorm(x, OV_BF16, (int64_t)L * D, h->ln_post_w, h->ln_post_b, ln, OV_BF16, D, B, D, t->cfg.ln_eps, stream)))
            return rc;
    }
    if ((rc = ov_gemm(ln, D, h->proj_t, D, nullptr, feat, EP, B, EP, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream))) return rc;
    if (normalize) return ov_l2norm(feat, OV_BF16, EP, features, E, B, E, stream);
    return ov_convert(feat, OV_BF16, EP, features, OV_F32, E, B, E, stream);
}

extern "C" int ov_encode_image(const ov_tower* t, const ov_vision_head* h, const void* image, int img_dtype, int B,
                               float* features, int normalize, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!t || !h || !image || !features || !workspace || B <= 0) return OV_ERR_INVALID;
    const VisionWs w = vision_ws(t, h, B);
    if (workspace_bytes < w.total) return OV_ERR_WORKSPACE;
    const int g = h->image_size / h->patch_size, L = g * g + 1;
    char* ws = (char*)workspace;
    ov_bf16* x = (ov_bf16*)(ws + w.x);
    const size_t tower_bytes = ov_tower_workspace_bytes(t, B, L);
    int rc;
    // the im2col buffer aliases the (not yet used) tower workspace
    if ((size_t)B * g * g * h->kpad * 2 > tower_bytes) return OV_ERR_WORKSPACE;
    if ((rc = ov_vision_embed(t, h, image, img_dtype, B, x, ws + w.tower, tower_bytes, stream))) return rc;
    if ((rc = ov_tower_forward(t, x, B, L, ws + w.tower, tower_bytes, stream))) return rc;
    return ov_vision_head_forward(t, h, x, B, features, normalize, ws + w.pooled, w.total - w.pooled, stream);
}

extern "C" size_t ov_text_workspace_bytes(const ov_tower* t, const ov_text_head* h, int B) {
    if (!t || !h || B <= 0) return 0;
    return text_ws(t, h, B).total;
}

extern "C" int ov_encode_text(const ov_tower* t, const ov_text_head* h, const int64_t* tokens, int B, float* features,
                              int normalize, int* err_flag, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!t || !h || !tokens || !features || !workspace || B <= 0) return OV_ERR_INVALID;
    const TextWs w = text_ws(t, h, B);
    if (workspace_bytes < w.total) return OV_ERR_WORKSPACE;
    const int T = h->context_length, D = t->cfg.width, E = h->embed_dim, EP = (E + 7) / 8 * 8;
    char* ws = (char*)workspace;
    ov_bf16* x = (ov_bf16*)(ws + w.x);
    ov_bf16* last = (ov_bf16*)(ws + w.last);
    ov_bf16* ln = (ov_bf16*)(ws + w.lnrow);
    ov_bf16* feat = (ov_bf16*)(ws + w.feat);
    int rc;
    if ((rc = ov_text_embed(tokens, h->token_embedding, h->pos, x, D, B, T, D, h->vocab_size, err_flag, stream))) return rc;
    if ((rc = ov_tower_forward(t, x, B, T, ws + w.tower, ov_tower_workspace_bytes(t, B, T), stream))) return rc;
    // ln_final is per-token, so only the pooled row needs it (model.py:276-277)
    if ((rc = ov_gather_rows(x, D, last, D, B, T, h->pool_last ? T - 1 : 0, D, stream))) return rc;
    if ((rc = ov_layernorm(last, OV_BF16, D, h->ln_final_w, h->ln_final_b, ln, OV_BF16, D, B, D, t->cfg.ln_eps, stream))) return rc;
    if ((rc = ov_gemm(ln, D, h->proj_t, D, nullptr, feat, EP, B, EP, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream))) return rc;
    if (normalize) return ov_l2norm(feat, OV_BF16, EP, features, E, B, E, stream);
    return ov_convert(feat, OV_BF16, EP, features, OV_F32, E, B, E, stream);
}
