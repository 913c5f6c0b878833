// Shared declarations of the training step's translation units (train_step.cpp: Swin-V2 encoders, decoder, heads; train_hybrid_step.cpp: the
// ViT-hybrid encoder).
#pragma once
#include <algorithm>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "train.h"

namespace soccdpt {
namespace trn {

struct TArena {
    char* base;
    size_t off = 0;
    explicit TArena(void* p) : base(static_cast<char*>(p)) {}
    float* f(size_t n) {
        off = (off + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += n * sizeof(float);
        return p;
    }
};

static const std::string ENC = "depth_net.pretrained.model.";
static const std::string HYB = "depth_net.pretrained.";
static const std::string SCR = "depth_net.scratch.";
inline std::string blk_key(int s, int j) { return ENC + "layers." + std::to_string(s) + ".blocks." + std::to_string(j) + "."; }

constexpr size_t kTrainSkPartFloats = (size_t)8 << 20;   // 32 MB of f32 split-K partials
constexpr size_t kTrainSkCountWords = 4096;
constexpr size_t kTrainTnArenaFloats = (size_t)160 << 20;   // 640 MB: the weight-gradient partials of one backward pass wait here for the batched sum (train.h TnDefer); flushed when full

struct BlkT {
    float *qkv_bias, *scale, *table, *bias_acc;
    const float* xin;
    float *qkv, *attn, *a_pre, *x1, *hpre, *hact, *m_pre, *xout;
    float* dp;   // stochastic depth: [2][B] per-sample scales of the attention / MLP branch (0 or 1 / (1 - p_block)); see train_forward
};

// ViT-hybrid encoder tape (train_hybrid_step.cpp)
struct RnBlkT {
    int cin, cout, mid, stride, rin, rout;
    bool proj;
    std::string key;
    const float* xin;                           // [B*rin*rin][cin]
    float *w_ds, *w_c1, *w_c2, *w_c3;           // standardised weights, tap-major
    float *ds_raw, *ds_stats, *c1_raw, *c1_stats, *t1 /*halo*/, *c2_raw, *c2_stats, *t2, *c3_raw, *c3_stats, *out;
};
struct VitBlkT {
    const float* xin;
    float *ln1, *qkv, *attn, *x1, *ln2, *hpre, *hact, *xout, *rowstat;
};
struct HyTape {
    float *a0, *w_stem, *stem_raw, *stem_stats, *pool;
    uint8_t* pool_idx;
    std::vector<RnBlkT> blk;
    float *pe_y, *x0;
    std::vector<VitBlkT> vb;
    float *cat[2], *ro_pre[2], *ro_act[2], *pp4_in /*halo*/, *w_pp4;
    float* gn_part[2] = {nullptr, nullptr};   // per-tile GroupNorm partials of the convolution(s) whose output is waiting for its gn_apply: [0] conv1 / conv2 / conv3, [1] the shortcut projection
    int gn_bm[2] = {0, 0};                     // M-tile rows of the launch that last filled each buffer
    size_t gn_part_floats = 0;
    float *GT, *GR, *xg, *attn_part;           // token-stream / residual-stream gradients, strided-shortcut operand, attention segment partials
};

struct Tape {
    HyTape hy;
    const float* xt_tn_src = nullptr;   // the halo image whose 16-bit copy (train_wgrad_tn.hip layout) is in S_T2, or nullptr: conv3_bwd's reuse_xt only holds within one layout
    // encoder
    float *patches, *pe_wpad, *pe_pre, *x0;
    std::vector<BlkT> blk[4];
    float *mg[3], *mr_pre[3], *mx[3];
    float* feat[4];   // zero-halo
    // decoder (level 0 = finest); *_relu / t1 / t2 are zero-halo images
    float *lrn_raw[4], *lrn_relu[4], *t1[4], *out_raw[4], *out_relu[4], *t2[4], *u[4], *oc[4];
    float *w_lrn[4], *w_rcu[4][2][2];
    float *path1, *d1, *d1u, *e, *inv, *seg;
    float *w_d0, *w_d2, *w_s0;
    float *c_raw, *bn_stats, *r, *logits;
    uint8_t* keep;
    // halo zone [halo_lo, halo_hi): zero-filled at the start of every forward
    size_t halo_lo = 0, halo_hi = 0;
    // backward scratch
    float *G[5], *GX, *GP, *DOC, *DF[4];
    float *S_T1, *S_T2, *S_halo, *S_wt, *S_dw, *S_col, *S_vec;
    float *dS, *rowstat, *dscale_part, *dtable, *dt, *S_cpb, *attn_part;
    float* sk_part;
    float* tn_arena;
    size_t S_dw_n = 0;   // floats of S_dw (a weight gradient written THERE is post-processed at once by its caller: never deferred)
    unsigned* sk_count;
    // dgrad weight operands of the whole backward pass, staged by stage_weights() in a few batched launches (4 bytes per element reserved per weight)
    float* WT = nullptr;
    std::vector<long long> wt_off;                        // per Handle::weights index: element offset of its slot in WT, -1 = not staged (odd shapes)
    std::unordered_map<const float*, long long> wt_by_ptr;   // bound weight pointer -> slot offset, filled by stage_weights()
    size_t maxAct = 0;
    float dropout_p = 0.f;
};

struct Ctx {
    Handle& h;
    Tape& T;
    int B;
    hipStream_t st;
    std::string& err;
    TnDefer tn;   // deferred weight-gradient sums of this pass (flushed by train_backward / train_backward_encoder before they return)
    std::unordered_set<const void*> grad_ptrs;   // the bound parameter-gradient buffers: only sums that land THERE may wait (a gradient written into scratch is read by its caller's next launch)
    bool may_defer(const float* dW, const float* db) const { return tn.arena && grad_ptrs.count(dW) && (!db || grad_ptrs.count(db)); }
    void arm_defer(float* arena, size_t cap) { tn.arena = arena; tn.cap = cap; for (const auto& w : h.weights) if (w.grad) grad_ptrs.insert(w.grad); }
    const float* W(const std::string& key) const { return h.weights[h.index.at(key)].ptr; }
    float* Gd(const std::string& key) const { return h.weights[h.index.at(key)].grad; }
};

#define TRY(call) do { if (call) return 1; } while (0)

#define TRY(call) do { if (call) return 1; } while (0)

int gemm(Ctx& c, IgemmDesc d, bool x3 = false);
// a GEMM of the train-mode FORWARD: exact f32, or -- train amp mode 3 -- its operands converted to x3 into scratch first (outputs stay f32)
int gemm_fwd(Ctx& c, IgemmDesc d, size_t x_elems, size_t w_elems);
int gemm16(Ctx& c, IgemmDesc d);   // 16-bit operands of the amp mode (bf16 / fp16), f32 outputs
int gemm_wgrad(Ctx& c, IgemmDesc d, bool bf16_operands, bool x3 = false);   // operands as written by the caller's staging kernels
int copy_d2d(Ctx& c, void* dst, const void* src, size_t bytes, const char* what);
// Transposed (Linear / 1x1) or rotated tap-major (3x3) copies of every regularly shaped bound weight in the amp mode's operand format, for the dgrad GEMMs of
// this backward pass: two or three launches instead of one per layer.  linear_bwd / conv3_bwd use a staged copy when they find one (staged_wt) and stage their
// own otherwise (derived weights: padded patch embedding, standardised ResNetV2 kernels).
int stage_weights(Ctx& c);
const void* staged_wt(const Ctx& c, const float* W);
// y = x W^T + b backward.  dY [M][N], X [M][K], W [N][K].  dX_out = dY W (+ dX_res); dW = dY^T X; db = colsum(dY).
int linear_bwd(Ctx& c, const float* dY, const float* X, const float* W, size_t M, int N, int K, float* dX_out, const float* dX_res, float* dW, float* db);
int conv3_bwd(Ctx& c, const float* dY, const float* Xhalo, const float* W, int r, int N, int C, float* dX_out, const float* dX_res, float* dW, float* db,
              bool reuse_xt = false);   // reuse_xt: the im2col^T of Xhalo is still in S_T2 from the previous call
int ln_bwd(Ctx& c, const float* y, const float* g, const float* dout, float* dy, float* xhat, size_t M, int C, float* dg, float* dbeta, float eps = 1e-5f);
IgemmDesc conv_desc(const void* X, int Cin, const void* Wt, int N, int r, int B);
bool any_grad(const Handle& h, const std::string& prefix);

// train_hybrid_step.cpp
void hy_carve_halo(const Handle& h, int B, TArena& ar, Tape& T);   // inside the zero-filled zone
void hy_carve(const Handle& h, int B, TArena& ar, Tape& T, size_t& maxAct);
int hy_forward(Ctx& c, const float* x);       // fills T.feat[0..3]
int hy_backward(Ctx& c);                       // consumes T.DF[0..3]

}  // namespace trn
}  // namespace soccdpt
