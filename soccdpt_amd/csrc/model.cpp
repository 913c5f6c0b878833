// Host-side orchestration of the SOccDPT_V3 network on gfx950: state-dict table, weight
// preparation (bf16 re-layout, BN fold, CPB bias tables) and the kernel launch sequence of
//   DPTDepthModel.forward  /root/reference/SOccDPT/model/dpt.py:142-232
//   forward_swin + hooks   /root/reference/SOccDPT/model/backbones/swin_common.py:8-54
//   seg_head               /root/reference/SOccDPT/model/SOccDPT.py:660-674,682-683
// All activations are NHWC ("token-major"): the encoder's [B,L,C] tokens ARE the decoder's feature
// maps, so the reference's Transpose+Unflatten never materialises.  3x3-conv inputs are bf16 images
// with a one-pixel zero halo that no kernel ever writes (the workspace is zero-filled once by the
// host), which removes all bounds checks from the convolution main loop.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <unordered_set>

#include "internal.h"
#include "kernels.h"

namespace soccdpt {

struct BlockW {
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;  // bf16 copies, or the bound f32 tensors in SOCCDPT_PREC_F32
    float *qkv_bias, *scale, *table, *bias_acc;
    const float *proj_b, *n1_g, *n1_b, *fc1_b, *fc2_b, *n2_g, *n2_b;
};
struct MergeW {
    const void* red_w;
    const float *g, *b;
};
struct RcuW {
    const void *w1, *w2;
    const float *b1, *b2;
};
// ViT-hybrid encoder (dpt_hybrid_384): ResNetV2 bottleneck, ViT block
struct RnBlockW {
    const void *ds_w = nullptr, *c1_w = nullptr, *c2_w = nullptr, *c3_w = nullptr;   // weight-standardised, tap-major, operand type
    const float *ds_g = nullptr, *ds_b = nullptr, *n1_g = nullptr, *n1_b = nullptr, *n2_g = nullptr, *n2_b = nullptr, *n3_g = nullptr, *n3_b = nullptr;
    int cin = 0, cout = 0, mid = 0, stride = 1;
    bool proj = false;
};
struct VitBlockW {
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
    const float *qkv_b, *proj_b, *fc1_b, *fc2_b, *n1_g, *n1_b, *n2_g, *n2_b;
};
struct HybridW {
    const void* stem_w = nullptr;          // [64][160]
    const float *stem_g = nullptr, *stem_b = nullptr;
    std::vector<std::vector<RnBlockW>> stages;
    const void* pe_w = nullptr;
    const float *pe_b = nullptr, *cls = nullptr;
    float* pos = nullptr;                  // position embedding at the run-time grid ([1 + g*g][768])
    std::vector<VitBlockW> blocks;
    const void *ro_w[2] = {nullptr, nullptr}, *pp_w[2] = {nullptr, nullptr}, *pp4_w = nullptr;
    const float *ro_b[2] = {nullptr, nullptr}, *pp_b[2] = {nullptr, nullptr}, *pp4_b = nullptr;
};
struct Prepared {
    HybridW hy;
    std::vector<std::vector<BlockW>> blocks;  // [stage][block]
    MergeW merge[3];
    const void* layer_rn[4];
    RcuW rcu[4][2];  // [refinenet-1][unit-1]
    const void* oc_w[4];
    const float* oc_b[4];
    const void *d0_w, *d2_w;
    const float *d0_b, *d2_b, *d4_w;
    float d4_b = 0.f;
    const void* s0_w;
    float *bn_scale, *bn_shift;
    const float *s4_w, *s4_b;
    float* patch_wT = nullptr;  // [48][128]
    std::unordered_set<const void*> x2w;   // prepared weights of x2w groups (x3 pairs read beside fp16 activations): gemm() picks the x2w tiles for them
};

Handle::~Handle() {
    model_drop_graph(*this);
    if (graph_stream) (void)hipStreamDestroy(graph_stream);
    if (graph_in) (void)hipEventDestroy(graph_in);
    if (graph_out) (void)hipEventDestroy(graph_out);
    delete prep;
    for (auto s : sub_streams) (void)hipStreamDestroy(s);
    for (auto e : join_events) (void)hipEventDestroy(e);
}

namespace {
inline long long shape_key(int M, int N, int K, int taps) { return (((long long)M * 8192 + N) * 65536 + K) * 16 + taps; }

// Operand format code of a launch-site group: 0 bf16, 1 fp16, 2 f32, 3 x3.  The uniform modes give every group the same code;
// SOCCDPT_PREC_MIXED looks the group up in the handle's precision map (absent = fp16).
int group_fmt(const Handle& h, const std::string& g) {
    switch (h.cfg.precision) {
        case SOCCDPT_PREC_F32: return 2;
        case SOCCDPT_PREC_F16: return 1;
        case SOCCDPT_PREC_F16X3: return 3;
        case SOCCDPT_PREC_MIXED: { auto it = h.prec_map.find(g); return (it != h.prec_map.end() && it->second == 3) ? 3 : 1; }   // an x2w group (4) reads fp16 activations
        default: return 0;
    }
}
// x2w (round 5): the group's ACTIVATIONS stay fp16 -- group_fmt() says 1, every producer / consumer of its operand buffers treats it as an fp16 group --
// and only its WEIGHTS are prepared as x3 pairs; the launch runs the two-MFMA x2w tiles (igemm_kernel.h)
bool group_x2w(const Handle& h, const std::string& g) {
    if (h.cfg.precision != SOCCDPT_PREC_MIXED) return false;
    auto it = h.prec_map.find(g);
    return it != h.prec_map.end() && it->second == 4;
}
inline std::string gname(const char* a, int i) { return std::string(a) + std::to_string(i); }
inline std::string gblk(int s, int j, const char* part) { return "s" + std::to_string(s) + ".b" + std::to_string(j) + "." + part; }
inline std::string gvit(int i, const char* part) { return "vit.b" + std::to_string(i) + "." + part; }
// ResNetV2 stage s of the hybrid: c = 1 the 1x1 reduce convolutions and the shortcut projection (both read the block input), 2 the 3x3, 3 the 1x1 expand
inline int stem_fmt(const Handle& h) { return h.cfg.precision == SOCCDPT_PREC_F32 ? 2 : 3; }   // operand format of the hybrid's stem GEMM (see lay_out)
inline std::string grn(int s, int c) { return "rn.s" + std::to_string(s) + ".c" + std::to_string(c); }


struct Arena {
    char* base;
    size_t off = 0, cap;
    Arena(void* p, size_t c) : base(static_cast<char*>(p)), cap(c) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

void add_w(Handle& h, const std::string& key, std::vector<int64_t> shape) {
    h.index[key] = (int)h.weights.size();
    h.weights.push_back(WeightSlot{key, std::move(shape), nullptr});
}

const std::string ENC = "depth_net.pretrained.model.";
const std::string HYB = "depth_net.pretrained.";
std::string rnblk(int s, int j) { return ENC + "patch_embed.backbone.stages." + std::to_string(s) + ".blocks." + std::to_string(j) + "."; }
std::string vitblk(int i) { return ENC + "blocks." + std::to_string(i) + "."; }
const std::string SCR = "depth_net.scratch.";

std::string blk(int s, int j) { return ENC + "layers." + std::to_string(s) + ".blocks." + std::to_string(j) + "."; }

// Walks the prepared-weights arena; with base == nullptr it only measures.  When `st` runs
// (base != nullptr) it also launches the conversion kernels.
int lay_out(Handle& h, Arena& ar, Prepared* P, hipStream_t st, std::string& err) {
    const Arch& a = h.arch;
    const bool run = ar.base != nullptr;
    auto W = [&](const std::string& key) -> const float* { return h.weights[h.index.at(key)].ptr; };
    // SOCCDPT_PREC_MIXED: every weight copy gets 4 bytes per element whatever its group's format, so the arena layout (and
    // soccdpt_prepared_bytes) does not depend on the precision map
    const bool MIX = h.cfg.precision == SOCCDPT_PREC_MIXED;
    // WEIGHT format code of a group: 0 bf16, 1 fp16, 2 f32, 3 x3 (half16.h); + 0x100 when the group is x2w (x3 weights beside fp16 activations)
    auto GF = [&](const std::string& g) { return group_x2w(h, g) ? (3 | 0x100) : group_fmt(h, g); };
    auto reg = [&](int& fmt, const void* p) { if ((fmt & 0x100) && run && P && p) P->x2w.insert(p); };
    static const char kNoCopy = 0;  // non-null placeholder while measuring
    auto cvt = [&](const std::string& key, size_t n, int fmt_) -> const void* {
        const int fmt = fmt_ & 0xff;
        if (fmt == 2) return run ? static_cast<const void*>(W(key)) : static_cast<const void*>(&kNoCopy);  // [N][K] f32 as bound
        bf16_t* p = (fmt == 3 || MIX) ? reinterpret_cast<bf16_t*>(ar.take<float>(n)) : ar.take<bf16_t>(n);
        if (run && launch_cvt_bf16(W(key), p, n, fmt, st, err)) return nullptr;
        reg(fmt_, p);
        return run ? static_cast<const void*>(p) : static_cast<const void*>(&kNoCopy);
    };
    auto convw = [&](const std::string& key, int Cout, int Cin, const float* scale, int fmt_) -> const void* {
        const int fmt = fmt_ & 0xff;
        const size_t n = (size_t)Cout * Cin * 9;
        void* p = (fmt >= 2 || MIX) ? static_cast<void*>(ar.take<float>(n)) : static_cast<void*>(ar.take<bf16_t>(n));
        if (run && launch_conv_w(W(key), scale, p, fmt == 2 ? 1 : 0, fmt == 2 ? 0 : fmt, Cout, Cin, st, err)) return nullptr;
        reg(fmt_, p);
        return run ? static_cast<const void*>(p) : static_cast<const void*>(&kNoCopy);
    };
    // weight-standardised convolution weight (timm StdConv2dSame, eps 1e-8), tap-major [Cout][Kpad]; fmt = operand format code of hybrid.hip
    auto wsw = [&](const std::string& key, int Cout, int Cin, int k, int Kpad, int fmt_) -> const void* {
        const int fmt = fmt_ & 0xff;
        const size_t n = (size_t)Cout * Kpad;
        void* p = (fmt >= 2 || MIX) ? static_cast<void*>(ar.take<float>(n)) : static_cast<void*>(ar.take<bf16_t>(n));
        if (run && launch_ws_conv_w(W(key), p, fmt, Cout, Cin, k, Kpad, 1e-8f, st, err)) return nullptr;
        reg(fmt_, p);
        return run ? static_cast<const void*>(p) : static_cast<const void*>(&kNoCopy);
    };
    if (a.hybrid) {
        HybridW hw;
        const std::string bb = ENC + "patch_embed.backbone.";
        {   // The stem runs f32-grade in EVERY precision mode: weight standardisation makes each filter zero-mean, so the large DC level
            // of the reference's un-normalised inputs (pixel values up to 509, SURVEY.md 3.4: no /255) cancels exactly in f32 and only the
            // small pixel-to-pixel variation survives -- 16-bit operand rounding of either side (absolute error ~1 on the pixels, a
            // non-zero filter sum after rounding) is as large as that signal (measured: stage-0 features 3.5e-2 off in bf16, 2.4e-3 in fp16).
            // Exact f32 operands in SOCCDPT_PREC_F32; x3 pairs (22 bits: the DC level cancels to 1e-4 absolute) everywhere else since round 5 --
            // the f32 MFMA ran this one GEMM at 27 TFLOP/s, 103 us of the 4.7 ms forward.
            float* p = ar.take<float>((size_t)a.stem_ch * 160);
            if (run && launch_ws_conv_w(W(bb + "stem.conv.weight"), p, stem_fmt(h), a.stem_ch, 3, 7, 160, 1e-8f, st, err)) return 1;
            hw.stem_w = run ? static_cast<const void*>(p) : static_cast<const void*>(&kNoCopy);
        }
        if (run) { if (!hw.stem_w) return 1; hw.stem_g = W(bb + "stem.norm.weight"); hw.stem_b = W(bb + "stem.norm.bias"); }
        int prev = a.stem_ch;
        hw.stages.assign(3, {});
        for (int s3 = 0; s3 < 3; ++s3) {
            const int cout = 256 << s3, mid = cout / 4;
            for (int j = 0; j < a.rn_layers[s3]; ++j) {
                const std::string b = rnblk(s3, j);
                RnBlockW bw;
                const int f1 = GF(grn(s3, 1)), f2 = GF(grn(s3, 2)), f3 = GF(grn(s3, 3));
                bw.cin = prev; bw.cout = cout; bw.mid = mid; bw.proj = (j == 0); bw.stride = (j == 0 && s3 > 0) ? 2 : 1;
                if (bw.proj) bw.ds_w = wsw(b + "downsample.conv.weight", cout, prev, 1, prev, f1);   // reads the block input like conv1: same group
                bw.c1_w = wsw(b + "conv1.weight", mid, prev, 1, prev, f1);
                bw.c2_w = wsw(b + "conv2.weight", mid, mid, 3, 9 * mid, f2);
                bw.c3_w = wsw(b + "conv3.weight", cout, mid, 1, mid, f3);
                if (run) {
                    if ((bw.proj && !bw.ds_w) || !bw.c1_w || !bw.c2_w || !bw.c3_w) return 1;
                    if (bw.proj) { bw.ds_g = W(b + "downsample.norm.weight"); bw.ds_b = W(b + "downsample.norm.bias"); }
                    bw.n1_g = W(b + "norm1.weight"); bw.n1_b = W(b + "norm1.bias");
                    bw.n2_g = W(b + "norm2.weight"); bw.n2_b = W(b + "norm2.bias");
                    bw.n3_g = W(b + "norm3.weight"); bw.n3_b = W(b + "norm3.bias");
                }
                hw.stages[s3].push_back(bw);
                prev = cout;
            }
        }
        const int E = a.vit_dim, g = a.grid();
        hw.pe_w = cvt(ENC + "patch_embed.proj.weight", (size_t)E * prev, GF("pe"));
        hw.pos = ar.take<float>((size_t)(1 + g * g) * E);
        if (run) {
            if (!hw.pe_w) return 1;
            hw.pe_b = W(ENC + "patch_embed.proj.bias");
            hw.cls = W(ENC + "cls_token");
            const int g0 = 24;   // vit_base_resnet50_384: pos_embed is [1, 1 + 24*24, 768]
            if (launch_pos_embed_resize(W(ENC + "pos_embed"), hw.pos, g0, g, E, st, err)) return 1;
        }
        for (int i = 0; i < a.vit_depth; ++i) {
            const std::string b = vitblk(i);
            VitBlockW vb{};
            vb.qkv_w = cvt(b + "attn.qkv.weight", (size_t)3 * E * E, GF(gvit(i, "qkv")));
            vb.proj_w = cvt(b + "attn.proj.weight", (size_t)E * E, GF(gvit(i, "proj")));
            vb.fc1_w = cvt(b + "mlp.fc1.weight", (size_t)4 * E * E, GF(gvit(i, "fc1")));
            vb.fc2_w = cvt(b + "mlp.fc2.weight", (size_t)4 * E * E, GF(gvit(i, "fc2")));
            if (run) {
                if (!vb.qkv_w || !vb.proj_w || !vb.fc1_w || !vb.fc2_w) return 1;
                vb.qkv_b = W(b + "attn.qkv.bias"); vb.proj_b = W(b + "attn.proj.bias");
                vb.fc1_b = W(b + "mlp.fc1.bias"); vb.fc2_b = W(b + "mlp.fc2.bias");
                vb.n1_g = W(b + "norm1.weight"); vb.n1_b = W(b + "norm1.bias");
                vb.n2_g = W(b + "norm2.weight"); vb.n2_b = W(b + "norm2.bias");
            }
            hw.blocks.push_back(vb);
        }
        for (int k = 0; k < 2; ++k) {
            const std::string ap = HYB + "act_postprocess" + std::to_string(3 + k) + ".";
            hw.ro_w[k] = cvt(ap + "0.project.0.weight", (size_t)E * 2 * E, GF(gname("ro", k)));
            hw.pp_w[k] = cvt(ap + "3.weight", (size_t)a.fdim(2 + k) * E, GF(gname("ro", k)));
            if (run) {
                if (!hw.ro_w[k] || !hw.pp_w[k]) return 1;
                hw.ro_b[k] = W(ap + "0.project.0.bias");
                hw.pp_b[k] = W(ap + "3.bias");
            }
        }
        hw.pp4_w = convw(HYB + "act_postprocess4.4.weight", a.fdim(3), a.fdim(3), nullptr, GF("pp4"));
        if (run) {
            if (!hw.pp4_w) return 1;
            hw.pp4_b = W(HYB + "act_postprocess4.4.bias");
            P->hy = hw;
        }
    } else {
        float* pw = ar.take<float>(48 * 128);
        if (run) {
            if (launch_patch_w(W(ENC + "patch_embed.proj.weight"), pw, a.embed, st, err)) return 1;
            P->patch_wT = pw;
        }
    }
    if (run) P->blocks.assign(4, {});
    for (int s = 0; s < 4 && !a.hybrid; ++s) {
        const int C = a.dim(s), H = a.heads[s], ws = a.ws(s);
        for (int j = 0; j < a.depths[s]; ++j) {
            const std::string b = blk(s, j);
            BlockW bw{};
            bw.qkv_w = cvt(b + "attn.qkv.weight", (size_t)3 * C * C, GF(gblk(s, j, "qkv")));
            bw.proj_w = cvt(b + "attn.proj.weight", (size_t)C * C, GF(gblk(s, j, "proj")));
            bw.fc1_w = cvt(b + "mlp.fc1.weight", (size_t)4 * C * C, GF(gblk(s, j, "fc1")));
            bw.fc2_w = cvt(b + "mlp.fc2.weight", (size_t)4 * C * C, GF(gblk(s, j, "fc2")));
            bw.qkv_bias = ar.take<float>(3 * C);
            bw.scale = ar.take<float>(H);
            bw.table = ar.take<float>((size_t)(2 * ws - 1) * (2 * ws - 1) * H);
            bw.bias_acc = ar.take<float>(attn_bias_elems(ws, H));
            if (run) {
                if (!bw.qkv_w || !bw.proj_w || !bw.fc1_w || !bw.fc2_w) return 1;
                if (launch_qkv_bias(W(b + "attn.q_bias"), W(b + "attn.v_bias"), bw.qkv_bias, C, st, err)) return 1;
                if (launch_logit_scale(W(b + "attn.logit_scale"), bw.scale, H, st, err)) return 1;
                if (launch_cpb_table(W(b + "attn.cpb_mlp.0.weight"), W(b + "attn.cpb_mlp.0.bias"), W(b + "attn.cpb_mlp.2.weight"), bw.table,
                                     ws, a.pretrained_window[s], H, st, err))
                    return 1;
                if (launch_attn_bias(bw.table, bw.bias_acc, ws, H, st, err)) return 1;
                bw.proj_b = W(b + "attn.proj.bias");
                bw.n1_g = W(b + "norm1.weight"); bw.n1_b = W(b + "norm1.bias");
                bw.fc1_b = W(b + "mlp.fc1.bias"); bw.fc2_b = W(b + "mlp.fc2.bias");
                bw.n2_g = W(b + "norm2.weight"); bw.n2_b = W(b + "norm2.bias");
                P->blocks[s].push_back(bw);
            }
        }
        if (s < 3) {
            const std::string d = ENC + "layers." + std::to_string(s) + ".downsample.";
            const void* rw = cvt(d + "reduction.weight", (size_t)8 * C * C, GF(gname("merge", s)));
            if (run) {
                if (!rw) return 1;
                P->merge[s] = MergeW{rw, W(d + "norm.weight"), W(d + "norm.bias")};
            }
        }
    }
    const int F = h.cfg.features;
    for (int i = 0; i < 4; ++i) {
        const void* p = convw(SCR + "layer" + std::to_string(i + 1) + "_rn.weight", F, a.fdim(i), nullptr, GF(gname("lrn", i)));
        if (run) { if (!p) return 1; P->layer_rn[i] = p; }
    }
    for (int r = 1; r <= 4; ++r) {
        const std::string b = SCR + "refinenet" + std::to_string(r) + ".";
        const void* ocw = cvt(b + "out_conv.weight", (size_t)F * F, GF(gname("oc", r - 1)));
        if (run) { if (!ocw) return 1; P->oc_w[r - 1] = ocw; P->oc_b[r - 1] = W(b + "out_conv.bias"); }
        for (int u = 1; u <= 2; ++u) {
            if (r == 4 && u == 1) continue;
            const std::string ub = b + "resConfUnit" + std::to_string(u) + ".";
            const void* w1 = convw(ub + "conv1.weight", F, F, nullptr, GF(gname("ref", r - 1)));
            const void* w2 = convw(ub + "conv2.weight", F, F, nullptr, GF(gname("ref", r - 1)));
            if (run) {
                if (!w1 || !w2) return 1;
                P->rcu[r - 1][u - 1] = RcuW{w1, w2, W(ub + "conv1.bias"), W(ub + "conv2.bias")};
            }
        }
    }
    {
        const void* d0 = convw(SCR + "output_conv.0.weight", F / 2, F, nullptr, GF("head"));
        const void* d2 = convw(SCR + "output_conv.2.weight", 32, F / 2, nullptr, GF("head.d2"));
        float* bscale = ar.take<float>(F);
        float* bshift = ar.take<float>(F);
        if (run && launch_bn_fold(W("seg_head.1.weight"), W("seg_head.1.bias"), W("seg_head.1.running_mean"), W("seg_head.1.running_var"),
                                  bscale, bshift, F, st, err))
            return 1;
        const void* s0 = convw("seg_head.0.weight", F, F, bscale, GF("head"));
        if (run) {
            if (!d0 || !d2 || !s0) return 1;
            P->d0_w = d0; P->d2_w = d2; P->s0_w = s0;
            P->d0_b = W(SCR + "output_conv.0.bias"); P->d2_b = W(SCR + "output_conv.2.bias");
            P->d4_w = W(SCR + "output_conv.4.weight");
            P->bn_scale = bscale; P->bn_shift = bshift;
            P->s4_w = W("seg_head.4.weight"); P->s4_b = W("seg_head.4.bias");
        }
    }
    return 0;
}

// "op" buffers hold GEMM/conv operands: bf16, or f32 in SOCCDPT_PREC_F32 (then xb aliases xf)
struct Workspace {
    float *xf, *y;
    void *xb, *qkv, *attn, *hbuf;
    void* feat[4];  // halo
    // decoder, index = level-1 (level 4 = coarsest)
    float *lrn_raw[4], *out_raw[4], *oc[4], *path[4];
    void *lrn_relu[4], *t_relu[4], *out_relu[4], *u[4];
    void *path1, *d1, *d1u, *s1;
    float* s2;
    // ViT-hybrid encoder
    void *hy_a0 = nullptr, *hy_xop = nullptr, *hy_t1[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, *hy_t2 = nullptr;
    float *hy_r[4] = {nullptr, nullptr, nullptr, nullptr}, *hy_xf = nullptr, *hy_stats[4] = {nullptr, nullptr, nullptr, nullptr}, *hy_part[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t hy_part_floats = 0;
    float *vt_y = nullptr, *vt_xf = nullptr;
    void *vt_xb = nullptr, *vt_qkv = nullptr, *vt_attn = nullptr, *vt_h = nullptr, *vt_tok[2] = {nullptr, nullptr}, *vt_ro = nullptr, *vt_pp4 = nullptr;
    float* sk_part;      // split-K partial tiles (igemm.h): kSplitKPartFloats floats
    unsigned* sk_count;  // split-K arrival counters: zero from workspace init, left zero by every launch
};

void carve(const Handle& h, int B, Arena& ar, Workspace& w) {
    const Arch& a = h.arch;
    const int G = a.grid(), C0 = a.embed, F = h.cfg.features;
    const size_t M0 = (size_t)B * G * G;
    const bool F32 = h.cfg.precision == SOCCDPT_PREC_F32;
    // f32 and x3 operands: 4 bytes per element.  SOCCDPT_PREC_MIXED sizes every operand buffer for x3, so the layout does not depend on the
    // precision map (an fp16 group uses the first half of its buffers)
    const size_t es = (F32 || h.cfg.precision == SOCCDPT_PREC_F16X3 || h.cfg.precision == SOCCDPT_PREC_MIXED) ? 4 : 2;
    auto op = [&](size_t elems) -> void* { return ar.take<char>(elems * es); };
    if (a.hybrid) {
        const int S = a.img, H1 = S / 2, H2 = S / 4, E = a.vit_dim, NT = G * G + 1;
        w.hy_a0 = ar.take<float>((size_t)B * H1 * H1 * 160);       // f32 in every mode (see lay_out: the stem)
        const size_t big = (size_t)B * H1 * H1 * a.stem_ch;     // = B * H2*H2 * 256: the largest raw convolution output
        for (int i = 0; i < 4; ++i) w.hy_r[i] = ar.take<float>(big);
        w.hy_xf = ar.take<float>(big);
        w.hy_xop = op(big);
        w.hy_t2 = op((size_t)B * H2 * H2 * 128);                 // widest plain mid tensor: 48^2 x 256 = 96^2 x 64 <= 96^2 x 128
        // one zero-halo image per distinct (resolution, width) of a 3x3 conv input: the halo positions differ, a shared buffer
        // would carry stale interior values of one geometry into the border of another
        const int tr[5] = {H2, H2, H2 / 2, H2 / 2, H2 / 4}, tc[5] = {64, 128, 128, 256, 256};
        for (int i = 0; i < 5; ++i) w.hy_t1[i] = op(Halo{tr[i], tr[i], tc[i]}.elems(B));
        for (int i = 0; i < 4; ++i) w.hy_stats[i] = ar.take<float>((size_t)B * 32 * 2);
        w.hy_part_floats = (size_t)B * H1 * H1 * 2;              // (M / 32 tiles) x 32 groups x 2 at the stem's M = B * H1^2
        for (int i = 0; i < 4; ++i) w.hy_part[i] = ar.take<float>(w.hy_part_floats);   // per statistics slot: the reader of the slot adds the partials
        w.vt_y = ar.take<float>((size_t)B * G * G * E);
        w.vt_xf = ar.take<float>((size_t)B * NT * E);
        w.vt_xb = op((size_t)B * NT * E);
        w.vt_qkv = op((size_t)B * NT * 3 * E);
        w.vt_attn = op((size_t)B * NT * E);
        w.vt_h = op((size_t)B * NT * 4 * E);
        for (int k = 0; k < 2; ++k) w.vt_tok[k] = op((size_t)B * NT * E);
        w.vt_ro = op((size_t)B * G * G * E);
        w.vt_pp4 = op(Halo{G, G, a.fdim(3)}.elems(B));
        w.xf = w.y = nullptr; w.xb = w.qkv = w.attn = w.hbuf = nullptr;
    } else {
        w.xf = ar.take<float>(M0 * C0);
        w.y = ar.take<float>(M0 * C0);
        w.xb = F32 ? static_cast<void*>(w.xf) : op(M0 * C0);
        w.qkv = op(M0 * 3 * C0);
        w.attn = op(M0 * C0);
        w.hbuf = op(M0 * 4 * C0);
    }
    for (int s = 0; s < 4; ++s) {
        Halo hl{a.fres(s), a.fres(s), a.fdim(s)};
        w.feat[s] = op(hl.elems(B));
    }
    for (int l = 0; l < 4; ++l) {
        const int r = a.fres(l);
        const size_t M = (size_t)B * r * r;
        Halo hl{r, r, F};
        w.lrn_raw[l] = ar.take<float>(M * F);
        w.out_raw[l] = ar.take<float>(M * F);
        w.oc[l] = ar.take<float>(M * F);
        w.path[l] = ar.take<float>(M * F);  // path arriving AT this level (from level l+1)
        w.lrn_relu[l] = op(hl.elems(B));
        w.t_relu[l] = op(hl.elems(B));
        w.out_relu[l] = op(hl.elems(B));
        w.u[l] = op(M * F);
    }
    const int r1 = 2 * a.fres(0), r0 = 4 * a.fres(0);
    w.path1 = op(Halo{r1, r1, F}.elems(B));
    w.d1 = op((size_t)B * r1 * r1 * (F / 2));
    w.d1u = op(Halo{r0, r0, F / 2}.elems(B));
    w.s1 = op((size_t)B * r1 * r1 * F);
    w.s2 = ar.take<float>((size_t)B * r1 * r1 * 4);
    w.sk_part = ar.take<float>(kSplitKPartFloats);
    w.sk_count = ar.take<unsigned>(kSplitKCountWords);
}

}  // namespace

static void carve_all(const Handle& h, int B, Arena& ar, std::vector<Workspace>& ws);
static void chunk_range(int B, int n, int i, int& lo, int& hi);

int model_init(Handle& h, std::string& err) {
    Arch a;
    if (h.cfg.backbone == SOCCDPT_BACKBONE_SWIN2T16_256) {
        // defaults
    } else if (h.cfg.backbone == SOCCDPT_BACKBONE_SWIN2B24_384) {
        a.img = 384; a.embed = 128; a.window = 24;
        int d[4] = {2, 2, 18, 2}, hd[4] = {4, 8, 16, 32}, pw[4] = {12, 12, 12, 6}, hk[4] = {1, 1, 17, 1};
        for (int i = 0; i < 4; ++i) { a.depths[i] = d[i]; a.heads[i] = hd[i]; a.pretrained_window[i] = pw[i]; a.hooks[i] = hk[i]; }
    } else if (h.cfg.backbone == SOCCDPT_BACKBONE_VITB_RN50_384) {
        a.hybrid = true; a.img = 384; a.patch = 16;
    } else {
        err = "soccdpt_create: backbone not implemented on the HIP path";
        return 1;
    }
    h.arch = a;
    h.img = a.img;
    if (h.cfg.precision == SOCCDPT_PREC_MIXED) model_prec_default(h);
    const int64_t C0 = a.embed;
    if (a.hybrid) {
        // timm 0.6.12 vit_base_resnet50_384 + the reference's act_postprocess3/4 (backbones/vit.py:183-229): SURVEY.md 8a row a4-H
        const int64_t E = a.vit_dim, NT = (int64_t)a.grid() * a.grid() + 1;
        add_w(h, ENC + "cls_token", {1, 1, E});
        add_w(h, ENC + "pos_embed", {1, NT, E});
        const std::string bb = ENC + "patch_embed.backbone.";
        add_w(h, bb + "stem.conv.weight", {a.stem_ch, 3, 7, 7});
        add_w(h, bb + "stem.norm.weight", {a.stem_ch});
        add_w(h, bb + "stem.norm.bias", {a.stem_ch});
        int64_t prev = a.stem_ch;
        for (int s3 = 0; s3 < 3; ++s3) {
            const int64_t cout = 256 << s3, mid = cout / 4;
            for (int j = 0; j < a.rn_layers[s3]; ++j) {
                const std::string b = rnblk(s3, j);
                if (j == 0) {
                    add_w(h, b + "downsample.conv.weight", {cout, prev, 1, 1});
                    add_w(h, b + "downsample.norm.weight", {cout});
                    add_w(h, b + "downsample.norm.bias", {cout});
                }
                add_w(h, b + "conv1.weight", {mid, prev, 1, 1});
                add_w(h, b + "norm1.weight", {mid});
                add_w(h, b + "norm1.bias", {mid});
                add_w(h, b + "conv2.weight", {mid, mid, 3, 3});
                add_w(h, b + "norm2.weight", {mid});
                add_w(h, b + "norm2.bias", {mid});
                add_w(h, b + "conv3.weight", {cout, mid, 1, 1});
                add_w(h, b + "norm3.weight", {cout});
                add_w(h, b + "norm3.bias", {cout});
                prev = cout;
            }
        }
        add_w(h, ENC + "patch_embed.proj.weight", {E, prev, 1, 1});
        add_w(h, ENC + "patch_embed.proj.bias", {E});
        for (int i = 0; i < a.vit_depth; ++i) {
            const std::string b = vitblk(i);
            add_w(h, b + "norm1.weight", {E});
            add_w(h, b + "norm1.bias", {E});
            add_w(h, b + "attn.qkv.weight", {3 * E, E});
            add_w(h, b + "attn.qkv.bias", {3 * E});
            add_w(h, b + "attn.proj.weight", {E, E});
            add_w(h, b + "attn.proj.bias", {E});
            add_w(h, b + "norm2.weight", {E});
            add_w(h, b + "norm2.bias", {E});
            add_w(h, b + "mlp.fc1.weight", {4 * E, E});
            add_w(h, b + "mlp.fc1.bias", {4 * E});
            add_w(h, b + "mlp.fc2.weight", {E, 4 * E});
            add_w(h, b + "mlp.fc2.bias", {E});
        }
        for (int k = 0; k < 2; ++k) {
            const std::string ap = HYB + "act_postprocess" + std::to_string(3 + k) + ".";
            add_w(h, ap + "0.project.0.weight", {E, 2 * E});
            add_w(h, ap + "0.project.0.bias", {E});
            add_w(h, ap + "3.weight", {a.fdim(2 + k), E, 1, 1});
            add_w(h, ap + "3.bias", {a.fdim(2 + k)});
        }
        add_w(h, HYB + "act_postprocess4.4.weight", {a.fdim(3), a.fdim(3), 3, 3});
        add_w(h, HYB + "act_postprocess4.4.bias", {a.fdim(3)});
    } else {
    add_w(h, ENC + "patch_embed.proj.weight", {C0, 3, a.patch, a.patch});
    add_w(h, ENC + "patch_embed.proj.bias", {C0});
    add_w(h, ENC + "patch_embed.norm.weight", {C0});
    add_w(h, ENC + "patch_embed.norm.bias", {C0});
    }
    for (int s = 0; s < 4 && !a.hybrid; ++s) {
        const int64_t C = a.dim(s), H = a.heads[s];
        for (int j = 0; j < a.depths[s]; ++j) {
            const std::string b = blk(s, j);
            add_w(h, b + "attn.logit_scale", {H, 1, 1});
            add_w(h, b + "attn.q_bias", {C});
            add_w(h, b + "attn.v_bias", {C});
            add_w(h, b + "attn.cpb_mlp.0.weight", {512, 2});
            add_w(h, b + "attn.cpb_mlp.0.bias", {512});
            add_w(h, b + "attn.cpb_mlp.2.weight", {H, 512});
            add_w(h, b + "attn.qkv.weight", {3 * C, C});
            add_w(h, b + "attn.proj.weight", {C, C});
            add_w(h, b + "attn.proj.bias", {C});
            add_w(h, b + "norm1.weight", {C});
            add_w(h, b + "norm1.bias", {C});
            add_w(h, b + "mlp.fc1.weight", {4 * C, C});
            add_w(h, b + "mlp.fc1.bias", {4 * C});
            add_w(h, b + "mlp.fc2.weight", {C, 4 * C});
            add_w(h, b + "mlp.fc2.bias", {C});
            add_w(h, b + "norm2.weight", {C});
            add_w(h, b + "norm2.bias", {C});
        }
        if (s < 3) {
            const std::string d = ENC + "layers." + std::to_string(s) + ".downsample.";
            add_w(h, d + "reduction.weight", {2 * C, 4 * C});
            add_w(h, d + "norm.weight", {2 * C});
            add_w(h, d + "norm.bias", {2 * C});
        }
    }
    const int64_t F = h.cfg.features;
    for (int i = 0; i < 4; ++i) add_w(h, SCR + "layer" + std::to_string(i + 1) + "_rn.weight", {F, a.fdim(i), 3, 3});
    for (int r = 1; r <= 4; ++r) {
        const std::string b = SCR + "refinenet" + std::to_string(r) + ".";
        add_w(h, b + "out_conv.weight", {F, F, 1, 1});
        add_w(h, b + "out_conv.bias", {F});
        for (int u = 1; u <= 2; ++u) {
            if (r == 4 && u == 1) continue;  // refinenet4 gets one input: its RCU1 never runs (model/dpt.py:163-165)
            for (int c = 1; c <= 2; ++c) {
                add_w(h, b + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(c) + ".weight", {F, F, 3, 3});
                add_w(h, b + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(c) + ".bias", {F});
            }
        }
    }
    add_w(h, SCR + "output_conv.0.weight", {F / 2, F, 3, 3});
    add_w(h, SCR + "output_conv.0.bias", {F / 2});
    add_w(h, SCR + "output_conv.2.weight", {32, F / 2, 3, 3});
    add_w(h, SCR + "output_conv.2.bias", {32});
    add_w(h, SCR + "output_conv.4.weight", {1, 32, 1, 1});
    add_w(h, SCR + "output_conv.4.bias", {1});
    add_w(h, "seg_head.0.weight", {F, F, 3, 3});
    add_w(h, "seg_head.1.weight", {F});
    add_w(h, "seg_head.1.bias", {F});
    add_w(h, "seg_head.1.running_mean", {F});
    add_w(h, "seg_head.1.running_var", {F});
    add_w(h, "seg_head.4.weight", {h.cfg.num_classes, F, 1, 1});
    add_w(h, "seg_head.4.bias", {h.cfg.num_classes});

    Arena measure(nullptr, 0);
    if (lay_out(h, measure, nullptr, nullptr, err)) return 1;
    h.prepared_bytes = measure.off + 256;
    return 0;
}

int model_bind(Handle& h, const char* key, const void* ptr, const int64_t* shape, int ndim, std::string& err) {
    auto it = h.index.find(key);
    if (it == h.index.end()) {
        err = std::string("soccdpt_bind_weight: key not consumed by the HIP path: ") + key;
        return 2;
    }
    WeightSlot& w = h.weights[it->second];
    bool ok = (int)w.shape.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = (w.shape[i] == shape[i]);
    if (!ok) {
        err = std::string("soccdpt_bind_weight: shape mismatch for ") + key;
        return 3;
    }
    w.ptr = static_cast<const float*>(ptr);
    h.is_prepared = false;
    return 0;
}


std::vector<std::string> model_prec_groups(const Handle& h) {
    const Arch& a = h.arch;
    std::vector<std::string> g;
    if (a.hybrid) {
        for (int s = 0; s < 3; ++s) for (int c = 1; c <= 3; ++c) g.push_back(grn(s, c));
        g.push_back("pe");
        for (int i = 0; i < a.vit_depth; ++i) for (const char* part : {"qkv", "proj", "fc1", "fc2"}) g.push_back(gvit(i, part));
        g.push_back("ro0"); g.push_back("ro1"); g.push_back("pp4");
    } else {
        for (int s = 0; s < 4; ++s) {
            for (int j = 0; j < a.depths[s]; ++j) for (const char* part : {"qkv", "proj", "fc1", "fc2"}) g.push_back(gblk(s, j, part));
            if (s < 3) g.push_back(gname("merge", s));
        }
    }
    for (int l = 3; l >= 0; --l) { g.push_back(gname("lrn", l)); g.push_back(gname("ref", l)); g.push_back(gname("oc", l)); }
    g.push_back("head"); g.push_back("head.d2"); g.push_back("head.s1");
    return g;
}

// Groups whose weights are read by something else than an igemm launch in the 16-bit formats cannot take x2w (x3 weight pairs beside fp16 activations):
// "head.d2" (the fused depth tail keeps the 32 x 1152 filter in registers, depth_tail.hip), "head.s1" (a storage format, not a launch).  "head" takes x2w since
// round 6: output_conv.0 runs the x2w tile like any convolution and the seg head falls back from the classifier-in-the-epilogue launch (a 16-bit-operand
// instantiation) to convolution -> feature map -> seg_tail, the form an x3 "head" has always used -- on checkpoints whose class logits need the head's
// WEIGHTS split (salt1 of bench.py's other_weights) that is two MFMAs per product instead of three for the two largest convolutions of the forward.
bool model_prec_x2w_ok(const std::string& g) { return g != "head.d2" && g != "head.s1"; }
// the seg head's Conv3x3 + BN + ReLU + Conv1x1 as ONE launch (igemm D3): 16-bit operands of the "head" group, 3 classes, 128-channel tiles
static bool seg_dot3_active(const Handle& h) {
    static const bool no_dot3 = getenv("SOCCDPT_SEG_DOT3_OFF") != nullptr;   // A/B switch: the unfused classifier of rounds 1-3
    return group_fmt(h, "head") <= 1 && h.cfg.features % 128 == 0 && h.cfg.num_classes == 3 && !no_dot3 && !group_x2w(h, "head");
}

int model_prec_set(Handle& h, const char* pattern, int fmt, std::string& err) {
    std::string p(pattern);
    const bool prefix = !p.empty() && p.back() == '*';
    if (prefix) p.pop_back();
    int n = 0;
    for (const auto& g : model_prec_groups(h)) {
        if (prefix ? g.compare(0, p.size(), p) != 0 : g != p) continue;
        if (fmt == 4 && !model_prec_x2w_ok(g)) {
            if (!prefix) { err = std::string("soccdpt_prec_map_set: this group cannot run x2w (its weights are read by a fused 16-bit kernel): ") + g; return -1; }
            h.prec_map[g] = 1;   // a pattern: the group keeps plain fp16
            ++n;
            continue;
        }
        h.prec_map[g] = fmt;
        ++n;
    }
    if (n == 0) { err = std::string("soccdpt_prec_map_set: no such group: ") + pattern; return -1; }
    return n;
}

// The shipped precision maps (Swin-V2 models: re-derived in round 6 with hold-out frames and head-room): what soccdpt_prec_calibrate_ex (calibrate.cpp; tools/derive_shipped_maps.py) derives on the synthetic weights of
// the tests and the benchmark -- three formats per group (fp16 / x2w / x3), one-group-out variances against the library's exact-f32 mode, greedy by
// variance removed per measured microsecond (prec_cost_table.h), measured prune -- with head-room under the bar the tests hold them to: all seven
// quantities within 5e-4 relative L2 of the fp32 reference (dpt_hybrid_384: 1e-3, its fp16 error is 2.5e-2).  profiles/r06_precision_map_*.json hold
// the reports.  On any other weights soccdpt_prepare switches every group to x3 until a calibration has run (capi.cpp).
void model_prec_default(Handle& h) {
    h.prec_map.clear();
    std::string err;
    auto x3 = [&](std::initializer_list<const char*> groups) { for (const char* g : groups) (void)model_prec_set(h, g, 3, err); };
    auto x2w = [&](std::initializer_list<const char*> groups) { for (const char* g : groups) (void)model_prec_set(h, g, 4, err); };
    switch (h.cfg.backbone) {
        case SOCCDPT_BACKBONE_VITB_RN50_384:
            // round 4's map in round 5's group names (B = 4, bar 1e-3): worst of the seven quantities 6.9e-4 (fp16 everywhere: 2.5e-2).  The weight-standardised
            // ResNetV2 stages amplify ACTIVATION rounding (x2w on them leaves 1e-3 ... 2e-2: profiles/r05_hybrid_map_try.txt) and take x3; the ViT blocks and
            // the 3x3 convolutions of the decoder stay fp16.  soccdpt_prec_calibrate's own pick for these weights (22 groups x3, 14 x2w at 8.9e-4) measured
            // SLOWER in round 5 (781 vs 811 frames/s); with round 6's cost table (profiles/r06_prec_costs_hybrid384.json) its pick -- the same nine ResNetV2 groups + oc0 in x3,
            // six x2w -- ties (918.5 vs 919.1 frames/s) at a larger error (7.7e-4 vs 6.8e-4): the hand-checked map stays.
            x3({"rn.s0.*", "rn.s1.*", "rn.s2.*", "ro1", "oc0", "oc1", "oc2", "oc3", "head.s1"});
            break;
        case SOCCDPT_BACKBONE_SWIN2B24_384:
            // dpt_swin2_base_384: budget 0.0005, worst of the seven quantities 4.23e-04 (fp16 everywhere: 1.21e-03); 42 groups x3, 26 x2w of 114; 281 forwards
            // (round 6: tools/derive_shipped_maps.py = soccdpt_prec_calibrate_ex on the synthetic weights, 4 + 2 frames at 256 px / 2 + 1 at 384 px, budget 5e-4: calibration frames <= 0.85 x budget,
            //  hold-out frames <= budget; frames/s: calibrated map 1259.5 -- the round-5 map (budget 4.7e-4 on two frames, 4.55e-4 measured) ran 1.3 % faster with half the margin)
            x3({"lrn2", "lrn3", "merge1", "merge2", "oc0", "oc1", "oc2", "oc3", "s0.b0.fc1", "s0.b1.fc1", "s1.b0.fc1", "s1.b0.proj", "s1.b0.qkv", "s1.b1.fc1", "s1.b1.qkv", "s2.b0.fc1", "s2.b0.proj", "s2.b0.qkv", "s2.b1.fc1", "s2.b1.proj", "s2.b1.qkv", "s2.b10.proj", "s2.b15.qkv", "s2.b2.fc1", "s2.b2.proj", "s2.b2.qkv", "s2.b3.fc1", "s2.b3.proj", "s2.b3.qkv", "s2.b4.proj", "s2.b4.qkv", "s2.b5.proj", "s2.b5.qkv", "s2.b6.proj", "s2.b6.qkv", "s2.b7.proj", "s2.b7.qkv", "s2.b8.proj", "s2.b8.qkv", "s3.b0.fc2", "s3.b0.proj", "s3.b1.proj"});
            x2w({"merge0", "ref2", "s0.b0.proj", "s0.b1.fc2", "s0.b1.proj", "s0.b1.qkv", "s1.b0.fc2", "s1.b1.fc2", "s1.b1.proj", "s2.b0.fc2", "s2.b1.fc2", "s2.b11.proj", "s2.b12.proj", "s2.b13.proj", "s2.b14.proj", "s2.b15.proj", "s2.b16.proj", "s2.b17.proj", "s2.b2.fc2", "s2.b4.fc2", "s2.b5.fc2", "s2.b9.proj", "s3.b0.fc1", "s3.b0.qkv", "s3.b1.fc2", "s3.b1.qkv"});
            break;
        default:
            // dpt_swin2_tiny_256: budget 0.0005, worst of the seven quantities 4.24e-04 (fp16 everywhere: 9.85e-04); 22 groups x3, 21 x2w of 66; 181 forwards
            // (round 6: tools/derive_shipped_maps.py = soccdpt_prec_calibrate_ex on the synthetic weights, 4 + 2 frames at 256 px / 2 + 1 at 384 px, budget 5e-4: calibration frames <= 0.85 x budget,
            //  hold-out frames <= budget; frames/s: calibrated map 4015.9 -- the round-5 map (budget 4.7e-4 on two frames, 4.56e-4 measured) ran at the same speed with half the margin)
            x3({"lrn2", "lrn3", "merge2", "oc0", "oc1", "oc2", "s1.b0.fc1", "s1.b0.proj", "s1.b1.fc1", "s1.b1.proj", "s2.b0.fc1", "s2.b0.proj", "s2.b1.fc1", "s2.b1.proj", "s2.b2.fc1", "s2.b2.proj", "s2.b3.fc1", "s2.b4.fc1", "s2.b4.proj", "s2.b5.fc1", "s3.b0.proj", "s3.b1.proj"});
            x2w({"merge0", "merge1", "oc3", "s0.b0.fc2", "s0.b0.proj", "s0.b0.qkv", "s0.b1.qkv", "s1.b0.fc2", "s1.b0.qkv", "s1.b1.fc2", "s1.b1.qkv", "s2.b0.fc2", "s2.b0.qkv", "s2.b1.fc2", "s2.b1.qkv", "s2.b2.qkv", "s2.b3.proj", "s2.b3.qkv", "s2.b5.proj", "s2.b5.qkv", "s3.b0.qkv"});
            break;
    }
}

int model_workspace_tensor(Handle& h, int B, const char* name, size_t* byte_offset, size_t* elems, int* kind, int* H, int* W, int* C) {
    char* fake = reinterpret_cast<char*>(uintptr_t(1) << 20);  // only offsets are used
    Arena ar(fake, ~size_t(0) >> 1);
    std::vector<Workspace> wsp;
    carve_all(h, B, ar, wsp);
    // "name" addresses the single-stream layout; "name@i" the i-th concurrent sub-batch of a multi-stream layout (diagnostics)
    std::string n(name);
    size_t chunk = 0;
    const size_t at = n.find('@');
    if (at != std::string::npos) { chunk = (size_t)atoi(n.c_str() + at + 1); n.resize(at); }
    else if (wsp.size() != 1) return 2;
    if (chunk >= wsp.size()) return 2;
    const Workspace& w = wsp[chunk];
    const Arch& a = h.arch;
    { int lo, hi; chunk_range(B, (int)wsp.size(), (int)chunk, lo, hi); B = hi - lo; }
    auto set = [&](const void* p, size_t e, int k, int hh, int ww, int cc) {
        *byte_offset = (size_t)(static_cast<const char*>(p) - fake); *elems = e; *kind = k; *H = hh; *W = ww; *C = cc;
        return 0;
    };
    // zero-halo NHWC kinds by the operand format of the group that READS the tensor: 2 bf16, 3 f32, 5 fp16, 7 x3 (plain: kind - 1, f32 = 0)
    auto halo_kind = [&](const std::string& g) { const int f = group_fmt(h, g); return f == 3 ? 7 : (f == 2 ? 3 : (f == 1 ? 5 : 2)); };
    for (int s = 0; s < 4; ++s)
        if (n == "feat" + std::to_string(s)) return set(w.feat[s], Halo{a.fres(s), a.fres(s), a.fdim(s)}.elems(B), halo_kind(gname("lrn", s)), a.fres(s), a.fres(s), a.fdim(s));
    const int r1 = 2 * a.fres(0);
    if (n == "path1") return set(w.path1, Halo{r1, r1, h.cfg.features}.elems(B), halo_kind("head"), r1, r1, h.cfg.features);
    if (n == "seg_feat") {   // seg head conv3x3 + BN + ReLU output
        // ... which does not exist when the 1x1 classifier rides in the convolution's epilogue (16-bit "head" group: the buffer then holds the partial logits
        // [F / tile][M][4]): not a tensor anybody should read as a feature map; SOCCDPT_SEG_DOT3_OFF=1 brings it back
        if (seg_dot3_active(h)) return 2;
        const int fs = h.cfg.precision == SOCCDPT_PREC_MIXED ? group_fmt(h, "head.s1") : group_fmt(h, "head");
        return set(w.s1, (size_t)B * r1 * r1 * h.cfg.features, fs >= 2 ? 0 : (fs == 1 ? 4 : 1), r1, r1, h.cfg.features);
    }
    if (n == "seg_logits") return set(w.s2, (size_t)B * r1 * r1 * 3, 0, r1, r1, 3);  // Conv2d(256,3,1) output before up-sampling / activation
    if (n == "xf" && !a.hybrid) return set(w.xf, (size_t)B * a.res(3) * a.res(3) * a.dim(3), 0, a.res(3), a.res(3), a.dim(3));   // stage 3's residual stream
    if (n == "vit_tokens" && a.hybrid) return set(w.vt_xf, (size_t)B * (a.grid() * a.grid() + 1) * a.vit_dim, 0, 1, a.grid() * a.grid() + 1, a.vit_dim);  // residual stream after the last block
    if (n == "rn_stage2" && a.hybrid) return set(w.hy_xf, (size_t)B * a.grid() * a.grid() * 1024, 0, a.grid(), a.grid(), 1024);                     // ResNetV2 output (stage 2)
    return 1;
}

int model_prepare(Handle& h, void* prepared, size_t bytes, hipStream_t st, std::string& err) {
    for (const auto& w : h.weights)
        if (!w.ptr) { err = "soccdpt_prepare: weight not bound: " + w.key; return 1; }
    if (!prepared || bytes < h.prepared_bytes) { err = "soccdpt_prepare: prepared buffer too small"; return 1; }
    delete h.prep;
    h.prep = new Prepared();
    Arena ar(prepared, bytes);
    if (lay_out(h, ar, h.prep, st, err)) return 1;
    // the depth head's last bias is a kernel argument: fetch the scalar (prepare may synchronise)
    hipError_t e = hipMemcpyAsync(&h.prep->d4_b, h.weights[h.index.at(SCR + "output_conv.4.bias")].ptr, sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { err = std::string("soccdpt_prepare: ") + hipGetErrorString(e); return 1; }
    h.is_prepared = true;
    model_drop_graph(h);  // the captured kernel arguments point into the old prepared arena
    return 0;
}

// The whole network for one contiguous sub-batch on one stream.
static int run_chunk(Handle& h, const Workspace& w, const float* x, int B, float* inv256, float* seg256, hipStream_t st, int& launches,
                     std::string& err) {
    const Arch& a = h.arch;
    const Prepared& P = *h.prep;
    const int F = h.cfg.features;
    const bool MIX = h.cfg.precision == SOCCDPT_PREC_MIXED;  // per-group fp16 / x3 operands (the precision map); 4-byte operand buffers
    auto GF = [&](const std::string& g) { return group_fmt(h, g); };   // operand format code of a launch-site group: 0 bf16, 1 fp16, 2 f32, 3 x3
    // a GEMM output that a NON-GEMM kernel consumes as plain f32 (attention of the 4-byte modes, bilinear, the seg tail): f32 operands ARE plain
    // f32 (out_op of an f32 launch); every other launch type writes it through out_f32 (fmt = operand format code of the launch)
    auto to_plain = [&](IgemmDesc& d, void* buf, int fmt) {
        if (fmt == 3 || (MIX && fmt == 1)) { d.out_f32 = static_cast<float*>(buf); d.act_on_f32 = 1; } else d.out_op = buf; };
#define RUN(call) do { if (call) return 1; ++launches; } while (0)
#define PROF(name, flops, bytes) ProfScope _ps(h.prof, name, flops, bytes, st)
    auto gemm = [&](IgemmDesc d, int fmt) {   // fmt: operand format code of the launch's group
        if (!d.f32 && !d.x3) { d.f32 = fmt == 2; d.f16 = fmt == 1; d.x3 = fmt == 3; }   // a caller may force the exact-f32 / x3 kernels for one launch (hybrid stem)
#ifdef SOCCDPT_ABLATIONS   // timing-only switches that give WRONG results: compiled only into `make ABLATIONS=1` builds (tools/), never into the shipped library
        static const int dbg_skip = getenv("SOCCDPT_DBG_SKIP_OUT_OP") ? atoi(getenv("SOCCDPT_DBG_SKIP_OUT_OP")) : 0;
        static const bool dbg_skip_warned = dbg_skip ? (fprintf(stderr, "soccdpt: SOCCDPT_DBG_SKIP_OUT_OP is set: TIMING-ONLY ablation -- operand copies are not stored, results are WRONG\n"), true) : false;
        (void)dbg_skip_warned;
        d.dbg_skip_out_op = dbg_skip;
#endif
        d.x3_among_f16 = MIX && d.x3;
        if (MIX && fmt == 1 && !P.x2w.empty() && P.x2w.count(d.Wt)) d.x2w = 1;   // the group's weights were prepared as x3 pairs: the two-MFMA x2w tiles
        const long long key = shape_key(d.M, d.N, d.taps * d.Cin, d.taps + ((MIX && fmt == 3) ? 16 : 0) + (d.x2w ? 32 : 0));   // mixed mode: the x3 / x2w launches of a shape are their own sites
        if ((!MIX || (fmt == 1 && !d.x2w)) && !h.tune_by_shape.empty() && d.tune < 0 && !d.ln_g && !d.dot3) {   // in-network tuning override (tools/autotune_network.py; tile ids are per format: in the mixed mode the fp16 launches only)
            auto it = h.tune_by_shape.find(key);
            if (it != h.tune_by_shape.end()) d.tune = it->second;
        }
        d.splitk = igemm_pick_splitk(d, kSplitKPartFloats, kSplitKCountWords);
        if (d.splitk > 1) { d.sk_part = w.sk_part; d.sk_count = w.sk_count; d.sk_part_floats = kSplitKPartFloats; d.sk_count_words = kSplitKCountWords; }
        const char* pname = igemm_family(d);
        if (h.prof_sites && h.prof.on) {
            size_t i = 0;
            for (; i < h.sites.size(); ++i)
                if (shape_key(h.sites[i].M, h.sites[i].N, h.sites[i].K, h.sites[i].taps + (h.sites[i].name[7] == 'x' ? 16 : 0) + (h.sites[i].name[7] == 'w' ? 32 : 0)) == key) break;
            if (i == h.sites.size() && h.sites.size() >= 1024) { err = "soccdpt: too many distinct igemm shapes for site profiling"; return 1; }
            if (i == h.sites.size()) {
                SiteRec r{d.M, d.N, d.taps * d.Cin, d.taps, igemm_config_id(d), 0, {0}};
                snprintf(r.name, sizeof(r.name), (MIX && fmt == 3) ? "site%03zux" : (d.x2w ? "site%03zuw" : "site%03zu"), i);
                h.sites.push_back(r);
            }
            h.sites[i].count++;
            pname = h.sites[i].name;
        }
        PROF(pname, igemm_flops(d), 0.0); return launch_igemm(d, st, err); };

    auto W = [&](const std::string& key) -> const float* { return h.weights[h.index.at(key)].ptr; };
    if (a.hybrid) {
        // ---------------- ViT-hybrid encoder (dpt_hybrid_384) ----------------
        // forward_flex (/root/reference/SOccDPT/model/backbones/vit.py:44-85) + forward_adapted_unflatten (backbones/utils.py:84-133);
        // launch for launch what oracle/soccdpt_ref.py hybrid_encoder() states.
        const HybridW& Y = P.hy;
        const int S = a.img, H1 = S / 2, H2 = S / 4;
        // Per-group operand formats (uniform modes: one code everywhere): "rn.s<k>.c<1|2|3>" (a ResNetV2 stage's bottlenecks share the zero-halo images
        // of their 3x3 inputs, so a convolution TYPE of a stage is the unit), "pe", the ViT blocks' "vit.b<i>.qkv" / ".proj" / ".fc1" / ".fc2", "ro<k>", "pp4".
        // per convolution type of a ResNetV2 stage (round 5; round 4 had one group per stage): [stage][0] 1x1 reduce + shortcut, [1] 3x3, [2] 1x1 expand
        int fRn[3][3];
        for (int s3 = 0; s3 < 3; ++s3) for (int c = 0; c < 3; ++c) fRn[s3][c] = GF(grn(s3, c + 1));
        const int fPe = GF("pe");
        // GroupNorm statistics ride on the producing convolution (igemm ST epilogue)
        // ... as per-tile partials only: the reader of the raw output -- gn_apply, or gn_finish for the stem -- adds them up.  Round 5,
        // tools/rn_stamps.py: the producer's own last-arriver finish put three dependent memory round trips (2-7 us) behind every one of these launches.
        int bm_slot[4] = {0, 0, 0, 0};   // M-tile rows of the launch that last filled each slot
        auto with_stats = [&](IgemmDesc& d, int slot, int cout, int hw) {
            d.gn_stats = w.hy_stats[slot]; d.gn_part = w.hy_part[slot]; d.gn_bm_out = &bm_slot[slot]; d.gn_cpg = cout / 32; d.gn_hw = hw; d.gn_eps = 1e-5f;
            d.gn_part_floats = w.hy_part_floats;
        };
        auto gn = [&](GnApplyArgs g, int om, int slot, int slot2 = -1) {   // om: format of out_op (and of out_halo unless g.halo_mode says otherwise); slot(s): whose partials
            if (bm_slot[slot] <= 0 || (slot2 >= 0 && bm_slot[slot2] <= 0)) { err = "soccdpt: GroupNorm statistics slot read before a convolution filled it"; return 1; }
#ifdef SOCCDPT_ABLATIONS
            static const int dbg_old = getenv("SOCCDPT_DBG_GN_OLDPART") ? atoi(getenv("SOCCDPT_DBG_GN_OLDPART")) : 0;   // timing-only: read another slot's (older) partials
            static const bool dbg_warned = dbg_old ? (fprintf(stderr, "soccdpt: SOCCDPT_DBG_GN_OLDPART is set: TIMING-ONLY ablation -- the GroupNorm statistics are WRONG\n"), true) : false;
            (void)dbg_warned;
#else
            constexpr int dbg_old = 0;
#endif
            g.part = w.hy_part[dbg_old ? (slot + 2) % 4 : slot]; g.tps = g.HW / bm_slot[slot];
            if (slot2 >= 0) { g.part2 = w.hy_part[slot2]; g.tps2 = g.HW / bm_slot[slot2]; }
            const int eo = om >= 2 ? 4 : 2;
            PROF("gn_apply", 0.0, (double)g.M * g.C * (4.0 + (g.raw2 || g.res ? 4.0 : 0.0) + (g.out_f32 ? 4.0 : 0.0) + (g.out_op ? eo : 0) + (g.out_halo ? eo : 0)));
            return launch_gn_apply(g, om, st, err);
        };
        {   // stem: Conv 7x7 / 2 'SAME' (im2col + igemm) -> GroupNorm + ReLU -> MaxPool 3x3 / 2 'SAME'
            { PROF("stem_im2col", 0.0, (double)B * S * S * 12.0 + (double)B * H1 * H1 * 160.0 * 4.0);
              RUN(launch_stem_im2col(x, w.hy_a0, stem_fmt(h), B, S, st, err)); }
            IgemmDesc d;
            d.f32 = stem_fmt(h) == 2; d.x3 = !d.f32;
            d.X = w.hy_a0; d.Wt = Y.stem_w; d.M = B * H1 * H1; d.N = a.stem_ch; d.Cin = 160; d.ldx = 160; d.out_f32 = w.hy_r[0];
            with_stats(d, 0, a.stem_ch, H1 * H1);
            RUN(gemm(d, stem_fmt(h)));
            { PROF("gn_relu_maxpool", 0.0, (double)B * H1 * H1 * a.stem_ch * 4.0 * 2.25 + (double)B * H2 * H2 * a.stem_ch * (fRn[0][0] >= 2 ? 4 : 2));
              RUN(launch_gn_finish(w.hy_part[0], w.hy_stats[0], B, H1 * H1 / bm_slot[0], 32, H1 * H1, a.stem_ch / 32, 1e-5f, st, err));
              RUN(launch_gn_relu_maxpool(w.hy_r[0], w.hy_stats[0], Y.stem_g, Y.stem_b, w.hy_xop, fRn[0][0], B, H1, a.stem_ch, a.stem_ch / 32, st, err)); }
        }
        int rcur = H2;
        for (int s3 = 0; s3 < 3; ++s3) {
            const int nb = (int)Y.stages[s3].size();
            const int f1 = fRn[s3][0], f2 = fRn[s3][1], f3 = fRn[s3][2];
            for (int j = 0; j < nb; ++j) {
                const RnBlockW& bw = Y.stages[s3][j];
                const int rin = rcur, rout = rin / bw.stride;
                const int Min = B * rin * rin, Mout = B * rout * rout;
                const int fnext = j + 1 < nb ? f1 : (s3 < 2 ? fRn[s3 + 1][0] : fPe);   // who reads this bottleneck's output
                // zero-halo image for this (resolution, width): see carve()
                const int ti = s3 == 0 ? 0 : (s3 == 1 ? (j == 0 ? 1 : 2) : (j == 0 ? 3 : 4));
                if (bw.proj) {   // shortcut: GN(1x1 stride-s conv)
                    IgemmDesc d;
                    d.X = w.hy_xop; d.Wt = bw.ds_w; d.M = Mout; d.N = bw.cout; d.Cin = bw.cin; d.out_f32 = w.hy_r[3];
                    if (bw.stride == 1) { d.ldx = bw.cin; }
                    else { d.gather1 = 1; d.stride = bw.stride; d.pad = 0; d.in_halo = 0; d.Hi = rin; d.Wi = rin; d.H = rout; d.W = rout; }
                    with_stats(d, 3, bw.cout, rout * rout);
                    RUN(gemm(d, f1));
                }
                {   // conv1 1x1 -> GN + ReLU -> halo image
                    IgemmDesc d;
                    d.X = w.hy_xop; d.Wt = bw.c1_w; d.M = Min; d.N = bw.mid; d.Cin = bw.cin; d.ldx = bw.cin; d.out_f32 = w.hy_r[0];
                    with_stats(d, 0, bw.mid, rin * rin);
                    RUN(gemm(d, f1));
                    GnApplyArgs g;
                    g.raw = w.hy_r[0]; g.stats = w.hy_stats[0]; g.gamma = bw.n1_g; g.beta = bw.n1_b; g.out_halo = w.hy_t1[ti];
                    g.M = (size_t)Min; g.HW = rin * rin; g.W = rin; g.C = bw.mid; g.cpg = bw.mid / 32;
                    RUN(gn(g, f2, 0));   // written in the format of its reader, the 3x3
                }
                {   // conv2 3x3 (stride on this conv; 'SAME': pad 1 at stride 1, the extra pixel right / bottom at stride 2) -> GN + ReLU
                    IgemmDesc d;
                    d.X = w.hy_t1[ti]; d.Wt = bw.c2_w; d.M = Mout; d.N = bw.mid; d.Cin = bw.mid; d.taps = 9; d.H = rout; d.W = rout; d.Hi = rin; d.Wi = rin;
                    d.stride = bw.stride; d.pad = bw.stride == 1 ? 1 : 0; d.in_halo = 1; d.out_f32 = w.hy_r[1];
                    with_stats(d, 1, bw.mid, rout * rout);
                    RUN(gemm(d, f2));
                    GnApplyArgs g;
                    g.raw = w.hy_r[1]; g.stats = w.hy_stats[1]; g.gamma = bw.n2_g; g.beta = bw.n2_b; g.out_op = w.hy_t2;
                    g.M = (size_t)Mout; g.HW = rout * rout; g.W = rout; g.C = bw.mid; g.cpg = bw.mid / 32;
                    RUN(gn(g, f3, 1));
                }
                {   // conv3 1x1 -> GN, + shortcut, ReLU: the new residual stream (f32) and its operand copy; hooked stages also as a halo image
                    IgemmDesc d;
                    d.X = w.hy_t2; d.Wt = bw.c3_w; d.M = Mout; d.N = bw.cout; d.Cin = bw.mid; d.ldx = bw.mid; d.out_f32 = w.hy_r[2];
                    with_stats(d, 2, bw.cout, rout * rout);
                    RUN(gemm(d, f3));
                    GnApplyArgs g;
                    g.raw = w.hy_r[2]; g.stats = w.hy_stats[2]; g.gamma = bw.n3_g; g.beta = bw.n3_b;
                    if (bw.proj) { g.raw2 = w.hy_r[3]; g.stats2 = w.hy_stats[3]; g.gamma2 = bw.ds_g; g.beta2 = bw.ds_b; }
                    else g.res = w.hy_xf;
                    g.out_f32 = w.hy_xf; g.out_op = w.hy_xop;
                    if (j == nb - 1 && s3 < 2) { g.out_halo = w.feat[s3]; g.halo_mode = GF(gname("lrn", s3)); }   // hooks on patch_embed.backbone.stages[0], [1] (vit.py:164-167)
                    g.M = (size_t)Mout; g.HW = rout * rout; g.W = rout; g.C = bw.cout; g.cpg = bw.cout / 32;
                    RUN(gn(g, fnext, 2, bw.proj ? 3 : -1));
                }
                rcur = rout;
            }
        }
        // ---- ViT-B over g*g + 1 tokens ----
        const int E = a.vit_dim, G = a.grid(), NT = G * G + 1, Mt = B * NT, Mp = B * G * G;
        {
            IgemmDesc d;
            d.X = w.hy_xop; d.Wt = Y.pe_w; d.M = Mp; d.N = E; d.Cin = 1024; d.ldx = 1024; d.bias = Y.pe_b; d.out_f32 = w.vt_y;
            RUN(gemm(d, fPe));
            const int f0 = GF(gvit(0, "qkv"));
            PROF("vit_tokens_ln", 0.0, (double)Mt * E * (8.0 + 4.0 + (f0 >= 2 ? 4 : 2)));
            RUN(launch_vit_tokens_ln(w.vt_y, Y.cls, Y.pos, w.vt_xf, Y.blocks[0].n1_g, Y.blocks[0].n1_b, w.vt_xb, f0, B, NT, E, 1e-6f, st, err));
        }
        for (int i = 0; i < a.vit_depth; ++i) {
            const VitBlockW& vb = Y.blocks[i];
            // one format per GEMM: fq (qkv), fp (proj), f1 (fc1), f2 (fc2); each producer writes for its reader
            const int fq = GF(gvit(i, "qkv")), fp = GF(gvit(i, "proj")), f1 = GF(gvit(i, "fc1")), f2 = GF(gvit(i, "fc2"));
            IgemmDesc d;
            d.X = w.vt_xb; d.Wt = vb.qkv_w; d.M = Mt; d.N = 3 * E; d.Cin = E; d.ldx = E; d.bias = vb.qkv_b;
            if (MIX) { d.out_op = w.vt_qkv; d.out_fmt = 1; }   // mixed mode: fp16 q, k, v for the fp16 attention kernel
            else to_plain(d, w.vt_qkv, fq);
            RUN(gemm(d, fq));
            { PROF("vit_attention", 4.0 * B * (double)NT * NT * E, (double)Mt * E * 4.0 * (fq >= 2 && !MIX ? 4 : 2));
              const int aprec = MIX ? SOCCDPT_PREC_F16 : h.cfg.precision;
              RUN(launch_vit_attention(w.vt_qkv, w.vt_attn, aprec, B, NT, a.vit_heads, st, err, (MIX && fp == 3) ? 1 : 0)); }
            d = IgemmDesc();
            d.X = w.vt_attn; d.Wt = vb.proj_w; d.M = Mt; d.N = E; d.Cin = E; d.ldx = E; d.bias = vb.proj_b; d.res1 = w.vt_xf; d.out_f32 = w.vt_xf;   // x += attn (in place)
            RUN(gemm(d, fp));
            { PROF("ln_rows", 0.0, (double)Mt * E * (4.0 + (f1 >= 2 ? 4 : 2)));
              RUN(launch_ln_rows(w.vt_xf, vb.n2_g, vb.n2_b, w.vt_xb, f1, Mt, E, 1e-6f, st, err)); }
            d = IgemmDesc();
            d.X = w.vt_xb; d.Wt = vb.fc1_w; d.M = Mt; d.N = 4 * E; d.Cin = E; d.ldx = E; d.bias = vb.fc1_b; d.act = ACT_GELU; d.out_op = w.vt_h; d.out_fmt = MIX ? f2 : -1;
            RUN(gemm(d, f1));
            d = IgemmDesc();
            d.X = w.vt_h; d.Wt = vb.fc2_w; d.M = Mt; d.N = E; d.Cin = 4 * E; d.ldx = 4 * E; d.bias = vb.fc2_b; d.res1 = w.vt_xf; d.out_f32 = w.vt_xf;      // x += mlp (in place)
            for (int k = 0; k < 2; ++k)
                if (i == a.vit_hooks[k]) { d.out_op = w.vt_tok[k]; d.out_fmt = MIX ? GF(gname("ro", k)) : -1; }   // hooks on blocks[8], blocks[11] (vit.py:168-171): operand copy for the readout GEMM
            RUN(gemm(d, f2));
            if (i + 1 < a.vit_depth) {
                const int fn = GF(gvit(i + 1, "qkv"));
                PROF("ln_rows", 0.0, (double)Mt * E * (4.0 + (fn >= 2 ? 4 : 2)));
                RUN(launch_ln_rows(w.vt_xf, Y.blocks[i + 1].n1_g, Y.blocks[i + 1].n1_b, w.vt_xb, fn, Mt, E, 1e-6f, st, err));
            }
        }
        // ---- act_postprocess3 / 4: ProjectReadout (cat(token, cls) @ W^T + GELU, the cat never materialises) -> Conv1x1 (-> Conv3x3 / 2) ----
        for (int k = 0; k < 2; ++k) {
            const int fr = GF(gname("ro", k)), fp4 = GF("pp4");
            IgemmDesc d;
            d.X = w.vt_tok[k]; d.Wt = Y.ro_w[k]; d.M = Mp; d.N = E; d.Cin = 2 * E; d.ldx = E; d.bias = Y.ro_b[k]; d.act = ACT_GELU; d.out_op = w.vt_ro;
            d.grp_rows = G * G; d.grp_stride = (long long)NT * E; d.grp_off = E; d.seg2_k = E; d.seg2_off = 0;
            RUN(gemm(d, fr));
            d = IgemmDesc();
            d.X = w.vt_ro; d.Wt = Y.pp_w[k]; d.M = Mp; d.N = a.fdim(2 + k); d.Cin = E; d.ldx = E; d.bias = Y.pp_b[k]; d.H = G; d.W = G;
            d.out_op = k == 0 ? w.feat[2] : w.vt_pp4; d.out_halo = 1; d.out_fmt = MIX ? (k == 0 ? GF("lrn2") : fp4) : -1;
            RUN(gemm(d, fr));
            if (k == 1) {
                d = IgemmDesc();
                d.X = w.vt_pp4; d.Wt = Y.pp4_w; d.M = B * (G / 2) * (G / 2); d.N = a.fdim(3); d.Cin = a.fdim(3); d.taps = 9; d.H = G / 2; d.W = G / 2; d.Hi = G; d.Wi = G;
                d.stride = 2; d.pad = 1; d.in_halo = 1; d.bias = Y.pp4_b; d.out_op = w.feat[3]; d.out_halo = 1; d.out_fmt = MIX ? GF("lrn3") : -1;
                RUN(gemm(d, fp4));
            }
        }
    } else {
    // ---------------- encoder ----------------
    // Every launch-site group carries its own operand format (uniform modes: the same one everywhere).  A producer writes each operand
    // copy in the format of the group that READS it: fa / fm = this block's attention / MLP group, fnext = the group that reads the
    // block's output (the next block's attention, or the PatchMerging reduction), fhook = the decoder's layer_rn conv of this stage.
    { const int f0 = GF(gblk(0, 0, "qkv"));
      PROF("patch_embed_ln", 0.0, (double)B * a.img * a.img * 12.0 + (double)B * a.grid() * a.grid() * a.embed * 6.0);
    RUN(launch_patch_embed(x, P.patch_wT, W(ENC + "patch_embed.proj.bias"), W(ENC + "patch_embed.norm.weight"),
                           W(ENC + "patch_embed.norm.bias"), w.xf, f0 == 2 ? nullptr : static_cast<bf16_t*>(w.xb), f0 == 2 ? 0 : f0, B, a.img, a.embed, st, err)); }
    for (int s = 0; s < 4; ++s) {
        const int C = a.dim(s), res = a.res(s), M = B * res * res, wsz = a.ws(s), H = a.heads[s];
        bool merged = false;   // the stage's last block wrote its operand copy straight into the PatchMerging layout (w.hbuf)
        const int fhook = GF(gname("lrn", s));
        for (int j = 0; j < a.depths[s]; ++j) {
            const BlockW& bw = P.blocks[s][j];
            const bool last = j == a.depths[s] - 1;
            // one format per GEMM: fa (qkv), fp (proj), fm (fc1), f2 (fc2); each producer writes its operand copy for the launch that reads it
            const int fa = GF(gblk(s, j, "qkv")), fp = GF(gblk(s, j, "proj")), fm = GF(gblk(s, j, "fc1")), f2 = GF(gblk(s, j, "fc2"));
            const int fnext = !last ? GF(gblk(s, j + 1, "qkv")) : (s < 3 ? GF(gname("merge", s)) : f2);
            // the operand copy goes straight into the PatchMerging layout where the reduction GEMM reads 16-bit (or, mixed mode, x3) operands
            const bool to_merge = s < 3 && last && (fnext <= 1 || (MIX && fnext == 3));
            const bool hook = (j == a.hooks[s]);
            void *b_qkv = w.qkv, *b_attn = w.attn, *b_hbuf = w.hbuf, *b_xin = w.xb, *b_x1 = w.xb, *b_x2 = to_merge ? w.hbuf : w.xb;
            float *b_y1 = w.y, *b_y2 = w.y, *b_xf = w.xf;
            IgemmDesc d;
            // round 6: qkv projection inside the attention kernel (attention_qkv.hip) where the group's activations are 16-bit (fp16 / bf16 / x2w groups) and
            // the window form is instantiated; x3 groups and the exact-f32 mode keep the two-launch chain
            const bool qkv_x2w = MIX && fa == 1 && group_x2w(h, gblk(s, j, "qkv"));
            const bool fuse_qkv = h.fuse_qkv && fa <= 1 && window_attention_qkv_supported(wsz, C, qkv_x2w ? 1 : 0) && (h.fuse_qkv_mask >> s & 1);
            if (fuse_qkv) {
                PROF("window_attention_qkv", 6.0 * M * (double)C * C + 4.0 * M * (double)(wsz * wsz) * C, (double)M * C * 4.0);
                RUN(launch_window_attention_qkv(static_cast<const bf16_t*>(b_xin), bw.qkv_w, bw.qkv_bias, bw.bias_acc, bw.scale, static_cast<bf16_t*>(b_attn), MIX ? 1 : fa,
                                                qkv_x2w ? 1 : 0, B, res, wsz, a.shift(s, j), H, st, err, (MIX && fp == 3) ? 1 : 0));
            } else {
            d.X = b_xin; d.Wt = bw.qkv_w; d.M = M; d.N = 3 * C; d.Cin = C; d.ldx = C; d.bias = bw.qkv_bias;
            if (MIX) { d.out_op = b_qkv; d.out_fmt = 1; }   // mixed mode: fp16 q, k, v for the fp16 attention kernel whatever the GEMM ran in
            else to_plain(d, b_qkv, fa);
            RUN(gemm(d, fa));
            }
            if (!fuse_qkv) { PROF("window_attention", 4.0 * M * (double)(wsz * wsz) * C, (double)M * C * 8.0);
              if (!MIX && fa >= 2) RUN(launch_window_attention_f32(static_cast<const float*>(b_qkv), bw.bias_acc, bw.table, bw.scale, static_cast<float*>(b_attn), B, res, wsz,
                                                       a.shift(s, j), H, st, err, fa == 3 ? 1 : 0));
              else RUN(launch_window_attention(static_cast<const bf16_t*>(b_qkv), bw.bias_acc, bw.scale, static_cast<bf16_t*>(b_attn), MIX ? 1 : fa, B, res, wsz,
                                               a.shift(s, j), H, st, err, (MIX && fp == 3) ? 1 : 0)); }
            const bool fuse_ln = C <= 128;  // whole rows fit one igemm tile; measured: a win for C = 96, a wash at 192, a loss beyond
            d = IgemmDesc();
            d.X = b_attn; d.Wt = bw.proj_w; d.M = M; d.N = C; d.Cin = C; d.ldx = C; d.bias = bw.proj_b;
            if (fuse_ln) {
                d.ln_g = bw.n1_g; d.ln_b = bw.n1_b; d.ln_xf = w.xf; d.out_op = fm == 2 ? nullptr : w.xb; d.out_fmt = MIX ? fm : -1;
                RUN(gemm(d, fp));
            } else {
                d.out_f32 = b_y1;
                RUN(gemm(d, fp));
                { PROF("ln_residual", 0.0, (double)M * C * 14.0);
                  RUN(launch_ln_residual(b_y1, bw.n1_g, bw.n1_b, b_xf, fm == 2 ? nullptr : static_cast<bf16_t*>(b_x1), nullptr, nullptr, fm == 2 ? 0 : fm, M, C, 1, res, 0, st, err)); }
            }
            if (fm <= 1 && f2 == fm && fnext == fm && (!hook || fhook == fm) && C <= h.mlp_fuse_max && mlp_ln_supported(C) &&
                !group_x2w(h, gblk(s, j, "fc1")) && !group_x2w(h, gblk(s, j, "fc2"))) {   // (the fused kernel reads plain fp16 weights)   // fc1 + GELU + fc2 + LayerNorm + residual as one launch
                PROF("mlp_ln_fused", 16.0 * M * (double)C * C, 0.0);
                RUN(launch_mlp_ln(static_cast<const bf16_t*>(b_x1), b_xf, static_cast<const bf16_t*>(bw.fc1_w), bw.fc1_b, static_cast<const bf16_t*>(bw.fc2_w),
                                  bw.fc2_b, bw.n2_g, bw.n2_b, static_cast<bf16_t*>(b_x2), hook ? static_cast<bf16_t*>(w.feat[s]) : nullptr, fm,
                                  M, C, res, res, to_merge ? 1 : 0, st, err));
                merged = to_merge;
                continue;
            }
            d = IgemmDesc();
            d.X = b_x1; d.Wt = bw.fc1_w; d.M = M; d.N = 4 * C; d.Cin = C; d.ldx = C; d.bias = bw.fc1_b; d.act = ACT_GELU; d.out_op = b_hbuf; d.out_fmt = MIX ? f2 : -1;
            RUN(gemm(d, fm));
            d = IgemmDesc();
            d.X = b_hbuf; d.Wt = bw.fc2_w; d.M = M; d.N = C; d.Cin = 4 * C; d.ldx = 4 * C; d.bias = bw.fc2_b;
            if (fuse_ln) {
                d.ln_g = bw.n2_g; d.ln_b = bw.n2_b; d.ln_xf = w.xf; d.out_op = fnext == 2 ? nullptr : w.xb; d.out_fmt = MIX ? fnext : -1;   // C <= 128: never a persistent-path stage; the LayerNorm epilogue writes plain rows (merge_gather follows)
                if (hook) { d.ln_halo = w.feat[s]; d.H = res; d.W = res; d.halo_fmt = MIX ? fhook : -1; }
                RUN(gemm(d, f2));
            } else {
            d.out_f32 = b_y2;
            RUN(gemm(d, f2));
            { PROF("ln_residual", 0.0, (double)M * C * 14.0);
              RUN(launch_ln_residual(b_y2, bw.n2_g, bw.n2_b, b_xf, fnext == 2 ? nullptr : static_cast<bf16_t*>(b_x2),
                                     (hook && fhook != 2) ? static_cast<bf16_t*>(w.feat[s]) : nullptr, (hook && fhook == 2) ? static_cast<float*>(w.feat[s]) : nullptr,
                                     fnext == 2 ? 0 : fnext, M, C, 1, res, to_merge ? 1 : 0, st, err, nullptr, 1, fhook == 2 ? -1 : fhook));
              merged = to_merge; }
            }
        }
        if (s < 3) {
            const int fg = GF(gname("merge", s)), fn = GF(gblk(s + 1, 0, "qkv"));
            if (!merged) { PROF("merge_gather", 0.0, (double)M * C * 4.0);
              RUN(launch_merge_gather(w.xb, w.hbuf, B, res, C, fg >= 2 ? 4 : 2, st, err)); }
            IgemmDesc d;
            d.X = w.hbuf; d.Wt = P.merge[s].red_w; d.M = M / 4; d.N = 2 * C; d.Cin = 4 * C; d.ldx = 4 * C; d.out_f32 = w.y;
            RUN(gemm(d, fg));
            { PROF("ln_residual", 0.0, (double)(M / 4) * 2 * C * 10.0);
              RUN(launch_ln_residual(w.y, P.merge[s].g, P.merge[s].b, w.xf, fn == 2 ? nullptr : static_cast<bf16_t*>(w.xb), nullptr, nullptr,
                                     fn == 2 ? 0 : fn, M / 4, 2 * C, 0, res / 2, 0, st, err)); }
        }
    }
    }   // Swin-V2 encoder
    // ---------------- decoder: reassemble + RefineNet fusion (coarse -> fine) ----------------
    auto conv = [&](const void* X, int Cin, const void* Wt, int N, int r) {
        IgemmDesc d;
        d.X = X; d.Wt = Wt; d.M = B * r * r; d.N = N; d.Cin = Cin; d.taps = 9; d.H = r; d.W = r;
        return d;
    };
    const int fH = GF("head"), fD2 = GF("head.d2");
    for (int l = 3; l >= 0; --l) {
        const int r = a.fres(l), M = B * r * r;
        const int fL = GF(gname("lrn", l)), fR = GF(gname("ref", l)), fO = GF(gname("oc", l));
        {   // layer{l+1}_rn: 3x3, no bias.  raw f32 (residual) + relu'd operand halo image (RCU conv1 input)
            IgemmDesc d = conv(w.feat[l], a.fdim(l), P.layer_rn[l], F, r);
            d.out_f32 = w.lrn_raw[l]; d.out_op = w.lrn_relu[l]; d.out_halo = 1; d.act = ACT_RELU; d.out_fmt = MIX ? fR : -1;
            RUN(gemm(d, fL));
        }
        const float* fused_raw = w.lrn_raw[l];
        const void* fused_relu = w.lrn_relu[l];
        if (l < 3) {  // output = path + RCU1(layer_rn)
            const RcuW& u1 = P.rcu[l][0];
            IgemmDesc d = conv(w.lrn_relu[l], F, u1.w1, F, r);
            d.bias = u1.b1; d.act = ACT_RELU; d.out_op = w.t_relu[l]; d.out_halo = 1;
            RUN(gemm(d, fR));
            d = conv(w.t_relu[l], F, u1.w2, F, r);
            d.bias = u1.b2; d.res1 = w.lrn_raw[l];
            d.res2 = w.oc[l + 1]; d.res2_h = a.fres(l + 1); d.res2_w = a.fres(l + 1);  // bilinear(out_conv output of the coarser level), on the fly
            d.out_f32 = w.out_raw[l]; d.out_op = w.out_relu[l]; d.out_halo = 1; d.act = ACT_RELU;
            RUN(gemm(d, fR));
            fused_raw = w.out_raw[l];
            fused_relu = w.out_relu[l];
        }
        {   // RCU2
            const RcuW& u2 = P.rcu[l][1];
            IgemmDesc d = conv(fused_relu, F, u2.w1, F, r);
            d.bias = u2.b1; d.act = ACT_RELU; d.out_op = w.t_relu[l]; d.out_halo = 1;
            RUN(gemm(d, fR));
            d = conv(w.t_relu[l], F, u2.w2, F, r);
            d.bias = u2.b2; d.res1 = fused_raw; d.out_op = w.u[l]; d.out_fmt = MIX ? fO : -1;
            RUN(gemm(d, fR));
        }
        {   // out_conv (1x1) BEFORE the bilinear resize: both are linear and the interpolation weights sum to 1
            IgemmDesc d;
            d.X = w.u[l]; d.Wt = P.oc_w[l]; d.M = M; d.N = F; d.Cin = F; d.ldx = F; d.bias = P.oc_b[l]; d.out_f32 = w.oc[l];
            RUN(gemm(d, fO));
        }
        if (l == 0) { PROF("bilinear_resize", 0.0, (double)M * F * (4.0 + 8.0));
               RUN(launch_bilinear(w.oc[0], 0, nullptr, fH == 2 ? nullptr : static_cast<bf16_t*>(w.path1), fH == 2 ? static_cast<float*>(w.path1) : nullptr, 1, fH == 2 ? 0 : fH, B, r, r,
                                   2 * r, 2 * r, F, st, err)); }
    }
    // ---------------- heads ----------------
    const int r1 = 2 * a.fres(0), r0 = 4 * a.fres(0);
    {
        IgemmDesc d = conv(w.path1, F, P.d0_w, F / 2, r1);
        d.bias = P.d0_b;
        if (fD2 <= 1 && F == 256) {
            // fused: up-sample + conv3x3(128->32) + ReLU + 1x1 + ReLU straight from the half-resolution map (16-bit operands)
            d.out_op = w.d1; d.out_fmt = MIX ? fD2 : -1;
            RUN(gemm(d, fH));
            PROF("depth_tail_fused", 2.0 * B * r0 * r0 * 32.0 * 9.0 * (F / 2), 0.0);
            RUN(launch_depth_tail(static_cast<const bf16_t*>(w.d1), static_cast<const bf16_t*>(P.d2_w), P.d2_b, P.d4_w, P.d4_b, inv256, fD2, B, r1, r1, st, err));
        } else {
            const bool d1_16 = fH <= 1 && !MIX;   // uniform 16-bit mode with F != 256: the half-resolution map stays a 16-bit operand
            if (d1_16) d.out_op = w.d1; else to_plain(d, w.d1, fH);
            RUN(gemm(d, fH));
            { PROF("bilinear_resize", 0.0, (double)B * r1 * r1 * (F / 2) * (2.0 + 8.0));
              RUN(launch_bilinear(w.d1, d1_16 ? 1 : 0, nullptr, fD2 == 2 ? nullptr : static_cast<bf16_t*>(w.d1u), fD2 == 2 ? static_cast<float*>(w.d1u) : nullptr, 1, fD2 == 2 ? 0 : fD2, B, r1,
                                  r1, r0, r0, F / 2, st, err)); }
            d = conv(w.d1u, F / 2, P.d2_w, 32, r0);
            d.bias = P.d2_b; d.act = ACT_RELU; d.dot_w = P.d4_w; d.dot_b = P.d4_b; d.out_dot = inv256;
            RUN(gemm(d, fD2));
        }
        // seg head: conv3x3 + folded BN + ReLU, then the 1x1 classifier + up-sampling + activation (f32 VALU).  The feature map between them is
        // a 16-bit operand in the 16-bit modes, plain f32 in the 4-byte modes; in the mixed mode the "head.s1" entry of the map chooses.
        const int fS1 = MIX ? GF("head.s1") : fH;
        const bool s1_f32 = fS1 >= 2;
        d = conv(w.path1, F, P.s0_w, F, r1);
        d.bias = P.bn_shift; d.act = ACT_RELU;
        if (seg_dot3_active(h)) {
            // the 1x1 classifier rides in the convolution's epilogue (igemm D3): the 256-channel feature map -- 67 MB written and read back per
            // forward at B = 8, rounded to 16 bits on the way -- is never stored; the logits come from the f32 accumulators
            d.dot3 = 1; d.dot_w = P.s4_w; d.out_dot = static_cast<float*>(w.s1);   // [F / BN][M][4] partial logits in the feature map's buffer
            d.f16 = fH == 1;
            const int parts = F / igemm_dot3_bn(d);   // one plane per channel tile of the launch (128-wide tiles: 2; the 16-wave 256 x 256 tile: 1)
            RUN(gemm(d, fH));
            PROF("seg_tail", 0.0, (double)B * r1 * r1 * (parts * 16.0 + 12.0 + 12.0 + 48.0));
            RUN(launch_seg_tail_parts(static_cast<const float*>(w.s1), parts, P.s4_b, w.s2, seg256, B, r1, r1, h.cfg.sigmoid, st, err));
        } else {
        if (s1_f32) to_plain(d, w.s1, fH); else { d.out_op = w.s1; d.out_fmt = MIX ? 1 : -1; }
        RUN(gemm(d, fH));
        { PROF("seg_tail", 0.0, (double)B * r1 * r1 * (F * 2.0 + 12.0 + 48.0));
          RUN(launch_seg_tail(w.s1, s1_f32 ? 1 : 0, fS1 == 1, P.s4_w, P.s4_b, w.s2, seg256, B, r1, r1, h.cfg.sigmoid, st, err)); }
        }
        ++launches;
    }
#undef RUN
#undef PROF
    return 0;
}

// Sub-batch split: frames [lo, hi) of chunk i when B frames are dealt to n chunks.
static void chunk_range(int B, int n, int i, int& lo, int& hi) {
    const int base = B / n, rem = B % n;
    lo = i * base + (i < rem ? i : rem);
    hi = lo + base + (i < rem ? 1 : 0);
}

static int n_chunks(const Handle& h, int B) {
    int n = h.n_streams < 1 ? 1 : h.n_streams;
    return n > B ? B : n;
}

static void carve_all(const Handle& h, int B, Arena& ar, std::vector<Workspace>& ws) {
    // head: network outputs (inv [B,S,S], seg [B,C,S,S]) when called through soccdpt_forward.  Always reserved, so the
    // zero-halo images occupy the same bytes whichever entry point runs (their borders must stay zero).
    ar.take<float>((size_t)B * h.img * h.img * (1 + h.cfg.num_classes));
    const int n = n_chunks(h, B);
    ws.resize(n);
    for (int i = 0; i < n; ++i) {
        int lo, hi;
        chunk_range(B, n, i, lo, hi);
        carve(h, hi - lo, ar, ws[i]);
    }
}

size_t model_workspace_bytes(Handle& h, int B) {
    Arena ar(nullptr, 0);
    std::vector<Workspace> ws;
    carve_all(h, B, ar, ws);
    return ar.off + 256;
}

int model_set_streams(Handle& h, int n, std::string& err) {
    if (n < 1 || n > 8) { err = "soccdpt_set_streams: n must be in 1..8"; return 1; }
    while ((int)h.sub_streams.size() < n - 1) {
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { err = "soccdpt_set_streams: stream creation failed"; return 1; }
        h.sub_streams.push_back(s);
    }
    while ((int)h.join_events.size() < n) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { err = "soccdpt_set_streams: event creation failed"; return 1; }
        h.join_events.push_back(e);
    }
    h.n_streams = n;
    model_drop_graph(h);
    return 0;
}

// Frames are independent through the whole network, so the batch is dealt to n_streams sub-batches that run
// CONCURRENTLY on internal streams (fork from / join into the caller's stream with events; no host sync).  The
// encoder's and the coarse decoder levels' launches are latency-bound at small M, and co-scheduling independent
// sub-batches fills the CUs they leave idle (measured: profiles/).
static int network_eager(Handle& h, const float* x, int B, float* inv256, float* seg256, void* ws, size_t ws_bytes, hipStream_t st,
                         std::string& err) {
    Arena ar(ws, ws_bytes);
    std::vector<Workspace> wsp;
    carve_all(h, B, ar, wsp);
    if (ar.off > ws_bytes) { err = "soccdpt_network: workspace too small"; return 1; }
    const int n = (int)wsp.size();
    const size_t S2 = (size_t)h.img * h.img;
    int launches = 0;
    if (n == 1) {
        if (run_chunk(h, wsp[0], x, B, inv256, seg256, st, launches, err)) return 1;
        h.launches = launches;
        return 0;
    }
    hipEvent_t fork = h.join_events[0];
    if (hipEventRecord(fork, st) != hipSuccess) { err = "soccdpt_network: event record failed"; return 1; }
    for (int i = 0; i < n; ++i) {
        int lo, hi;
        chunk_range(B, n, i, lo, hi);
        hipStream_t cs = (i == 0) ? st : h.sub_streams[i - 1];
        if (i > 0 && hipStreamWaitEvent(cs, fork, 0) != hipSuccess) { err = "soccdpt_network: stream wait failed"; return 1; }
        if (run_chunk(h, wsp[i], x + (size_t)lo * 3 * S2, hi - lo, inv256 + (size_t)lo * S2, seg256 + (size_t)lo * h.cfg.num_classes * S2, cs,
                      launches, err))
            return 1;
        if (i > 0) {
            if (hipEventRecord(h.join_events[i], cs) != hipSuccess || hipStreamWaitEvent(st, h.join_events[i], 0) != hipSuccess) {
                err = "soccdpt_network: join failed";
                return 1;
            }
        }
    }
    h.launches = launches;
    return 0;
}

void model_drop_graph(Handle& h) {
    if (h.graph_exec) (void)hipGraphExecDestroy(h.graph_exec);
    if (h.graph) (void)hipGraphDestroy(h.graph);
    h.graph_exec = nullptr;
    h.graph = nullptr;
    h.eager_calls = 0;
}

// With soccdpt_set_graph the launch sequence (all sub-batch streams, fork/join edges included) is captured into a
// hipGraph the second time the same argument tuple is seen and replayed afterwards: ~130-1000 host launches per
// forward become one graph launch, which is what lets the concurrent sub-batches actually overlap on the GPU
// instead of being serialised by the host's launch rate.
int model_network(Handle& h, const float* x, int B, float* inv256, float* seg256, void* ws, size_t ws_bytes, hipStream_t st,
                  std::string& err) {
    if (!h.is_prepared) { err = "soccdpt_network: call soccdpt_prepare after binding weights"; return 1; }
    if (B <= 0 || !x || !inv256 || !seg256 || !ws) { err = "soccdpt_network: bad argument"; return 1; }
    {   // The library owns the zero-halo invariant: the workspace layout (and so the position of every conv border) depends on
        // (B, stream count); the first call with a new (buffer, B, streams) tuple zero-fills the buffer on the caller's stream.
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (dev != h.device) { err = "soccdpt_network: the handle was created on device " + std::to_string(h.device) + " but device " + std::to_string(dev) + " is current"; return 1; }
        const size_t need = model_workspace_bytes(h, B);
        if (ws_bytes < need) { err = "soccdpt_network: workspace too small (soccdpt_workspace_bytes)"; return 1; }
        Handle::WsKey wk{ws, B, h.n_streams};
        if (!(wk == h.ws_key)) {
            model_drop_graph(h);
            if (hipMemsetAsync(ws, 0, need, st) != hipSuccess) { err = "soccdpt_network: workspace zero-fill failed"; return 1; }
            h.ws_key = wk;
            h.ws_zero_fills++;
        }
    }
    if (!h.use_graph || h.prof.on) return network_eager(h, x, B, inv256, seg256, ws, ws_bytes, st, err);
    Handle::GraphKey key;
    key.x = x; key.inv = inv256; key.seg = seg256; key.ws = ws; key.B = B; key.streams = h.n_streams;
    if (!(key == h.graph_key)) {
        model_drop_graph(h);
        h.graph_key = key;
    }
    if (h.eager_calls++ == 0 && !h.graph_exec) return network_eager(h, x, B, inv256, seg256, ws, ws_bytes, st, err);  // warms lazy state
    // run on the library's capture stream, ordered after / before the caller's stream by two events
    hipStream_t cs = h.graph_stream;
    if (hipEventRecord(h.graph_in, st) != hipSuccess || hipStreamWaitEvent(cs, h.graph_in, 0) != hipSuccess) {
        err = "soccdpt_network: graph fork failed";
        return 1;
    }
    if (!h.graph_exec) {
        if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { err = "soccdpt_network: begin capture failed"; return 1; }
        const int rc = network_eager(h, x, B, inv256, seg256, ws, ws_bytes, cs, err);
        hipGraph_t g = nullptr;
        const hipError_t ec = hipStreamEndCapture(cs, &g);
        if (rc || ec != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            if (!rc) err = std::string("soccdpt_network: end capture failed: ") + hipGetErrorString(ec);
            return 1;
        }
        h.graph = g;
        if (hipGraphInstantiate(&h.graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
            model_drop_graph(h);
            err = "soccdpt_network: graph instantiate failed";
            return 1;
        }
    }
    if (hipGraphLaunch(h.graph_exec, cs) != hipSuccess) { err = "soccdpt_network: hipGraphLaunch failed"; return 1; }
    if (hipEventRecord(h.graph_out, cs) != hipSuccess || hipStreamWaitEvent(st, h.graph_out, 0) != hipSuccess) {
        err = "soccdpt_network: graph join failed";
        return 1;
    }
    return 0;
}

}  // namespace soccdpt
