// Training step, ViT-hybrid encoder (dpt_hybrid_384): train-mode forward with saved activations and the backward of
// timm vit_base_resnet50_384 as the reference wires it (/root/reference/SOccDPT/model/backbones/vit.py:147-258: ResNetV2 stem + stages (3, 4, 9)
// with weight-standardised 'SAME' convolutions and GroupNorm, HybridEmbed projection, class token + position embedding, 12 pre-norm ViT-B
// blocks, hooks on stages[0], stages[1], blocks[8], blocks[11]; backbones/utils.py:27-133: ProjectReadout + reassemble convolutions).
// The forward is launch for launch the eval path of model.cpp in exact f32, with every intermediate kept; kernels: train_hybrid.hip, train.hip,
// hybrid.hip, igemm.hip.
#include "train_internal.h"

namespace soccdpt {

// train_hybrid.hip
int th_gn_bwd(const float* dout, const float* x, const float* stats, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, float* scratch, int B,
              int HW, int C, int cpg, int relu, hipStream_t st, std::string& err);
int th_ws_bwd(const float* dwh, const float* wh, const float* w, float* dw, int Cout, int Cin, int k, int Kpad, float eps, hipStream_t st, std::string& err);
int th_conv_w_dgrad_tap(const float* wt, float* out, int N, int C, hipStream_t st, std::string& err);
int th_im2colT_gen(const float* halo, float* out, int B, int Hi, int Ho, int C, int stride, int pad, size_t Mp, hipStream_t st, std::string& err);
int th_col2im(const float* dcol, float* dx, int B, int Hi, int Ho, int C, int stride, int pad, int accumulate, hipStream_t st, std::string& err);
int th_stride_gather(const float* in, float* out, int B, int Hi, int Ho, int C, int stride, hipStream_t st, std::string& err);
int th_stride_scatter_add(const float* dg, float* dx, int B, int Hi, int Ho, int C, int stride, hipStream_t st, std::string& err);
int th_maxpool_bwd(const float* dpool, const float* raw, const float* stats, const float* gamma, const float* beta, uint8_t* idx, float* dA, int B, int Hi, int C, int cpg,
                   hipStream_t st, std::string& err);
int th_readout_cat(const float* tok, float* cat, int B, int NT, int E, hipStream_t st, std::string& err);
int th_readout_cat_bwd(const float* dcat, float* dtok, int B, int NT, int E, int accumulate, hipStream_t st, std::string& err);
int th_tokens_to_patches(const float* dtok, float* dpatch, int B, int NT, int E, hipStream_t st, std::string& err);
size_t th_vit_attention_part_floats(int B, int N, int heads);
int th_vit_attention_fwd(const float* qkv, float* out, float* rowstat, float* part, int B, int N, int heads, hipStream_t st, std::string& err);
int th_vit_attention_bwd_mfma(const float* qkv, const float* O, const float* dO, float* rowstat, float* dqkv, int B, int N, int heads, hipStream_t st, std::string& err, int op);   // train_attn.hip (op: 0 exact f32, 1 bf16, 2 fp16 products)
int th_vit_attention_bwd(const float* qkv, const float* O, const float* dO, const float* rowstat, float* part, float* dqkv, int B, int N, int heads, hipStream_t st,
                         std::string& err);

namespace trn {
namespace {

const float kWsEps = 1e-8f;    // timm StdConv2dSame
const float kLnEps = 1e-6f;    // timm VisionTransformer norm_layer

std::string rn_key(int s, int j) { return ENC + "patch_embed.backbone.stages." + std::to_string(s) + ".blocks." + std::to_string(j) + "."; }
std::string vit_key(int i) { return ENC + "blocks." + std::to_string(i) + "."; }

// the bottleneck list of ResNetV2 (3, 4, 9): geometry only
void describe_blocks(const Arch& a, std::vector<RnBlkT>& out) {
    out.clear();
    int prev = a.stem_ch, r = a.img / 4;
    for (int s3 = 0; s3 < 3; ++s3) {
        const int cout = 256 << s3, mid = cout / 4;
        for (int j = 0; j < a.rn_layers[s3]; ++j) {
            RnBlkT b{};
            b.cin = prev; b.cout = cout; b.mid = mid; b.proj = (j == 0); b.stride = (j == 0 && s3 > 0) ? 2 : 1;
            b.rin = r; b.rout = r / b.stride;
            b.key = rn_key(s3, j);
            out.push_back(b);
            prev = cout;
            r = b.rout;
        }
    }
}

// Backward of a 3x3 convolution with tap-major weights Wtap [N][9][C] over a zero-haloed input [B][Hi+2][Hi+2][C], output Ho x Ho.
// stride 1 / pad 1: dgrad as the same conv over the rotated filter; otherwise dcol = dY Wtap, then col2im.  dWtap_out [N][9][C].
int conv_gen_bwd(Ctx& c, const float* dY, const float* Xhalo, const float* Wtap, int Hi, int Ho, int N, int C, int stride, int pad, float* dX_out, float* dWtap_out,
                 float* db) {
    Tape& T = c.T;
    const int B = c.B;
    const size_t Mo = (size_t)B * Ho * Ho;
    {
        // 16-bit amp modes, un-strided convolutions of the wide stages (N, C multiples of 128): dgrad on the 16-bit igemm over the zero-bordered dY image, weight
        // gradient from the operands as stored in halo pixel order (train_wgrad_tn.hip) -- the path the decoder convolutions take in conv3_bwd, with the
        // tap-major standardised weights of this encoder on both ends.  SOCCDPT_WGRAD_TRANSPOSE=1 keeps the f32 path below.
        static const bool tn_off = getenv("SOCCDPT_WGRAD_TRANSPOSE") != nullptr;
        const int amp = c.h.train_amp;
        const int rp = Ho + 2;
        const size_t Kh = (size_t)B * rp * rp, Kp = (Kh + 63) / 64 * 64, mrg = (size_t)rp + 1;
        if (!tn_off && (amp == 1 || amp == 2) && stride == 1 && pad == 1 && Hi == Ho && N % 128 == 0 && C % 128 == 0 && tr_wgrad_tn_ok(Kp, N, C, 9)) {
            const int F16 = amp == 2 ? 1 : 0, cvt = F16 ? 5 : 0;
            uint16_t* h16 = reinterpret_cast<uint16_t*>(T.S_halo);
            TRY(tr_to_halo_full(dY, h16, B, Ho, Ho, N, 1 + F16, c.st, c.err));
            if (dX_out) {
                uint16_t* w16 = reinterpret_cast<uint16_t*>(T.S_wt);
                TRY(th_conv_w_dgrad_tap(Wtap, T.S_dw, N, C, c.st, c.err));                         // [C][9][N] f32, rotated
                TRY(launch_cvt_bf16(T.S_dw, w16, (size_t)C * 9 * N, cvt, c.st, c.err));
                IgemmDesc d;
                d.X = h16; d.Wt = w16; d.M = (int)Mo; d.N = C; d.Cin = N; d.taps = 9; d.H = Ho; d.W = Ho; d.out_f32 = dX_out;
                TRY(gemm16(c, d));
            }
            if (dWtap_out) {
                uint16_t* xb = reinterpret_cast<uint16_t*>(T.S_T2);
                hipError_t e = Kp > Kh ? hipMemsetAsync(h16 + Kh * N, 0, (Kp - Kh) * N * 2, c.st) : hipSuccess;
                if (e == hipSuccess) e = hipMemsetAsync(xb, 0, mrg * C * 2, c.st);
                if (e == hipSuccess) e = hipMemsetAsync(xb + (mrg + Kh) * C, 0, (Kp - Kh + mrg) * C * 2, c.st);
                if (e != hipSuccess) { c.err = std::string("conv_gen_bwd memset: ") + hipGetErrorString(e); return 1; }
                TRY(launch_cvt_bf16(Xhalo, xb + mrg * C, Kh * C, cvt, c.st, c.err));
                c.T.xt_tn_src = nullptr;   // S_T2 no longer holds conv3_bwd's staged image
                TRY(tr_wgrad_tn(h16, N, xb + mrg * C, C, Kp, N, C, 9, rp, F16, T.sk_part, kTrainSkPartFloats, dWtap_out, c.st, c.err));
            }
            if (db) TRY(tr_colsum(dY, nullptr, db, T.S_col, Mo, N, 0, c.st, c.err));
            return 0;
        }
    }
    if (dX_out) {
        if (stride == 1 && pad == 1) {
            const size_t hb = (size_t)B * (Ho + 2) * (Ho + 2) * N * sizeof(float);
            hipError_t e = hipMemsetAsync(T.S_halo, 0, hb, c.st);
            if (e != hipSuccess) { c.err = std::string("conv_gen_bwd memset: ") + hipGetErrorString(e); return 1; }
            TRY(tr_to_halo(dY, T.S_halo, B, Ho, Ho, N, c.st, c.err));
            TRY(th_conv_w_dgrad_tap(Wtap, T.S_wt, N, C, c.st, c.err));
            IgemmDesc d;
            d.X = T.S_halo; d.Wt = T.S_wt; d.M = (int)Mo; d.N = C; d.Cin = N; d.taps = 9; d.H = Ho; d.W = Ho; d.out_f32 = dX_out;
            TRY(gemm(c, d));
        } else {
            TRY(tr_transpose(Wtap, T.S_wt, N, 9 * C, N, c.st, c.err));   // [9C][N]
            IgemmDesc d;
            d.X = dY; d.Wt = T.S_wt; d.M = (int)Mo; d.N = 9 * C; d.Cin = N; d.ldx = N; d.out_f32 = T.S_T2;   // dcol [Mo][9][C]
            TRY(gemm(c, d));
            TRY(th_col2im(T.S_T2, dX_out, B, Hi, Ho, C, stride, pad, 0, c.st, c.err));
        }
    }
    if (dWtap_out) {
        const int Mp = (int)((Mo + 31) / 32 * 32);
        TRY(tr_transpose(dY, T.S_T1, (int)Mo, N, Mp, c.st, c.err));
        TRY(th_im2colT_gen(Xhalo, T.S_T2, B, Hi, Ho, C, stride, pad, (size_t)Mp, c.st, c.err));
        IgemmDesc d;
        d.X = T.S_T1; d.Wt = T.S_T2; d.M = N; d.N = 9 * C; d.Cin = Mp; d.ldx = Mp; d.out_f32 = dWtap_out;
        TRY(gemm_wgrad(c, d, false));   // f32 staging (the strided / weight-standardised convolutions do not take the amp path)
    }
    if (db) TRY(tr_colsum(dY, nullptr, db, T.S_col, Mo, N, 0, c.st, c.err));
    return 0;
}

// GroupNorm statistics ride on the producing convolution as per-tile partials; its reader (gn_apply, gn_finish for the stem) adds them up and leaves
// {mean, rstd} in `stats` for the backward pass (model.cpp's forward does the same: DESIGN.md 11.5)
void with_stats(Ctx& c, IgemmDesc& d, float* stats, int cout, int hw, int slot = 0) {
    HyTape& Y = c.T.hy;
    d.gn_stats = stats; d.gn_part = Y.gn_part[slot]; d.gn_bm_out = &Y.gn_bm[slot]; d.gn_cpg = cout / 32; d.gn_hw = hw; d.gn_eps = 1e-5f;
    d.gn_part_floats = Y.gn_part_floats;
}
void from_partials(Ctx& c, GnApplyArgs& g, int slot, int slot2 = -1) {
    HyTape& Y = c.T.hy;
    g.part = Y.gn_part[slot]; g.tps = g.HW / Y.gn_bm[slot];
    if (slot2 >= 0) { g.part2 = Y.gn_part[slot2]; g.tps2 = g.HW / Y.gn_bm[slot2]; }
}

}  // namespace

void hy_carve_halo(const Handle& h, int B, TArena& ar, Tape& T) {
    HyTape& Y = T.hy;
    describe_blocks(h.arch, Y.blk);
    for (auto& b : Y.blk) b.t1 = ar.f((size_t)B * (b.rin + 2) * (b.rin + 2) * b.mid);
    const int G = h.arch.grid();
    Y.pp4_in = ar.f((size_t)B * (G + 2) * (G + 2) * h.arch.fdim(3));
}

void hy_carve(const Handle& h, int B, TArena& ar, Tape& T, size_t& maxAct) {
    const Arch& a = h.arch;
    HyTape& Y = T.hy;
    const int S = a.img, H1 = S / 2, H2 = S / 4, E = a.vit_dim, G = a.grid(), NT = G * G + 1;
    const size_t M1s = (size_t)B * H1 * H1, Mt = (size_t)B * NT, Mp = (size_t)B * G * G;
    Y.a0 = ar.f(M1s * 160);
    Y.w_stem = ar.f((size_t)a.stem_ch * 160);
    Y.stem_raw = ar.f(M1s * a.stem_ch);
    Y.stem_stats = ar.f((size_t)B * 32 * 2);
    Y.pool = ar.f((size_t)B * H2 * H2 * a.stem_ch);
    Y.pool_idx = reinterpret_cast<uint8_t*>(ar.f(((size_t)B * H2 * H2 * a.stem_ch + 3) / 4));
    maxAct = std::max(maxAct, M1s * 160);
    for (auto& b : Y.blk) {
        const size_t Min = (size_t)B * b.rin * b.rin, Mout = (size_t)B * b.rout * b.rout;
        b.w_ds = b.proj ? ar.f((size_t)b.cout * b.cin) : nullptr;
        b.w_c1 = ar.f((size_t)b.mid * b.cin);
        b.w_c2 = ar.f((size_t)b.mid * 9 * b.mid);
        b.w_c3 = ar.f((size_t)b.cout * b.mid);
        b.ds_raw = b.proj ? ar.f(Mout * b.cout) : nullptr;
        b.ds_stats = b.proj ? ar.f((size_t)B * 64) : nullptr;
        b.c1_raw = ar.f(Min * b.mid);
        b.c1_stats = ar.f((size_t)B * 64);
        b.c2_raw = ar.f(Mout * b.mid);
        b.c2_stats = ar.f((size_t)B * 64);
        b.t2 = ar.f(Mout * b.mid);
        b.c3_raw = ar.f(Mout * b.cout);
        b.c3_stats = ar.f((size_t)B * 64);
        b.out = ar.f(Mout * b.cout);
        maxAct = std::max(maxAct, std::max(Min * (size_t)std::max(b.cin, b.mid), Mout * (size_t)b.cout));
        maxAct = std::max(maxAct, Mout * 9 * (size_t)b.mid);   // dcol of the strided 3x3
    }
    Y.gn_part_floats = M1s * 2;   // (M / 32 tiles) x 32 groups x 2 at the stem's M
    for (int i = 0; i < 2; ++i) Y.gn_part[i] = ar.f(Y.gn_part_floats);
    Y.pe_y = ar.f(Mp * E);
    Y.x0 = ar.f(Mt * E);
    Y.vb.assign(a.vit_depth, VitBlkT{});
    for (auto& v : Y.vb) {
        v.ln1 = ar.f(Mt * E);
        v.qkv = ar.f(Mt * 3 * E);
        v.attn = ar.f(Mt * E);
        v.x1 = ar.f(Mt * E);
        v.ln2 = ar.f(Mt * E);
        v.hpre = ar.f(Mt * 4 * E);
        v.hact = ar.f(Mt * 4 * E);
        v.xout = ar.f(Mt * E);
        v.rowstat = ar.f((size_t)B * a.vit_heads * NT * 2);
    }
    for (int k = 0; k < 2; ++k) {
        Y.cat[k] = ar.f(Mp * 2 * E);
        Y.ro_pre[k] = ar.f(Mp * E);
        Y.ro_act[k] = ar.f(Mp * E);
    }
    Y.w_pp4 = ar.f((size_t)a.fdim(3) * 9 * a.fdim(3));
    maxAct = std::max(maxAct, std::max(Mt * 4 * E, Mp * 9 * (size_t)a.fdim(3)));
    Y.GT = ar.f(Mt * E);
    Y.GR = ar.f((size_t)B * H2 * H2 * 256);
    Y.xg = ar.f((size_t)B * (H2 / 2) * (H2 / 2) * 256);
    Y.attn_part = ar.f(th_vit_attention_part_floats(B, NT, a.vit_heads));
}

int hy_forward(Ctx& c, const float* x) {
    Handle& h = c.h;
    const Arch& a = h.arch;
    Tape& T = c.T;
    HyTape& Y = T.hy;
    const int B = c.B;
    hipStream_t st = c.st;
    std::string& err = c.err;
    const int S = a.img, H1 = S / 2, H2 = S / 4;
    const std::string bb = ENC + "patch_embed.backbone.";
    // ---- stem: Conv 7x7 / 2 'SAME' (im2col + GEMM) -> GroupNorm + ReLU -> MaxPool 3x3 / 2 'SAME' ----
    TRY(launch_stem_im2col(x, Y.a0, 2, B, S, st, err));
    TRY(launch_ws_conv_w(c.W(bb + "stem.conv.weight"), Y.w_stem, 2, a.stem_ch, 3, 7, 160, kWsEps, st, err));
    {
        IgemmDesc d;
        d.X = Y.a0; d.Wt = Y.w_stem; d.M = B * H1 * H1; d.N = a.stem_ch; d.Cin = 160; d.ldx = 160; d.out_f32 = Y.stem_raw;
        with_stats(c, d, Y.stem_stats, a.stem_ch, H1 * H1);
        TRY(gemm(c, d));
        TRY(launch_gn_finish(Y.gn_part[0], Y.stem_stats, B, H1 * H1 / Y.gn_bm[0], 32, H1 * H1, a.stem_ch / 32, 1e-5f, st, err));
        TRY(launch_gn_relu_maxpool(Y.stem_raw, Y.stem_stats, c.W(bb + "stem.norm.weight"), c.W(bb + "stem.norm.bias"), Y.pool, 2, B, H1, a.stem_ch, a.stem_ch / 32, st, err));
    }
    const float* xcur = Y.pool;
    int hook = 0, cnt = 0, stage = 0;
    for (auto& b : Y.blk) {
        const int Min = B * b.rin * b.rin, Mout = B * b.rout * b.rout;
        const std::string& k = b.key;
        b.xin = xcur;
        if (b.proj) {
            TRY(launch_ws_conv_w(c.W(k + "downsample.conv.weight"), b.w_ds, 2, b.cout, b.cin, 1, b.cin, kWsEps, st, err));
            IgemmDesc d;
            d.X = b.xin; d.Wt = b.w_ds; d.M = Mout; d.N = b.cout; d.Cin = b.cin; d.out_f32 = b.ds_raw;
            if (b.stride == 1) d.ldx = b.cin;
            else { d.gather1 = 1; d.stride = b.stride; d.pad = 0; d.in_halo = 0; d.Hi = b.rin; d.Wi = b.rin; d.H = b.rout; d.W = b.rout; }
            with_stats(c, d, b.ds_stats, b.cout, b.rout * b.rout, 1);
            TRY(gemm_fwd(c, d, (size_t)Min * b.cin, (size_t)b.cout * b.cin));   // (x3 only for the un-strided form: gemm_fwd checks)
        }
        {
            TRY(launch_ws_conv_w(c.W(k + "conv1.weight"), b.w_c1, 2, b.mid, b.cin, 1, b.cin, kWsEps, st, err));
            IgemmDesc d;
            d.X = b.xin; d.Wt = b.w_c1; d.M = Min; d.N = b.mid; d.Cin = b.cin; d.ldx = b.cin; d.out_f32 = b.c1_raw;
            with_stats(c, d, b.c1_stats, b.mid, b.rin * b.rin);
            TRY(gemm_fwd(c, d, (size_t)Min * b.cin, (size_t)b.mid * b.cin));
            GnApplyArgs g;
            g.raw = b.c1_raw; g.stats = b.c1_stats; g.gamma = c.W(k + "norm1.weight"); g.beta = c.W(k + "norm1.bias"); g.out_halo = b.t1;
            g.M = (size_t)Min; g.HW = b.rin * b.rin; g.W = b.rin; g.C = b.mid; g.cpg = b.mid / 32;
            from_partials(c, g, 0);
            TRY(launch_gn_apply(g, 2, st, err));
        }
        {
            TRY(launch_ws_conv_w(c.W(k + "conv2.weight"), b.w_c2, 2, b.mid, b.mid, 3, 9 * b.mid, kWsEps, st, err));
            IgemmDesc d;
            d.X = b.t1; d.Wt = b.w_c2; d.M = Mout; d.N = b.mid; d.Cin = b.mid; d.taps = 9; d.H = b.rout; d.W = b.rout; d.Hi = b.rin; d.Wi = b.rin;
            d.stride = b.stride; d.pad = b.stride == 1 ? 1 : 0; d.in_halo = 1; d.out_f32 = b.c2_raw;
            with_stats(c, d, b.c2_stats, b.mid, b.rout * b.rout);
            TRY(gemm_fwd(c, d, (size_t)B * (b.rin + 2) * (b.rin + 2) * b.mid, (size_t)9 * b.mid * b.mid));
            GnApplyArgs g;
            g.raw = b.c2_raw; g.stats = b.c2_stats; g.gamma = c.W(k + "norm2.weight"); g.beta = c.W(k + "norm2.bias"); g.out_op = b.t2;
            g.M = (size_t)Mout; g.HW = b.rout * b.rout; g.W = b.rout; g.C = b.mid; g.cpg = b.mid / 32;
            from_partials(c, g, 0);
            TRY(launch_gn_apply(g, 2, st, err));
        }
        {
            TRY(launch_ws_conv_w(c.W(k + "conv3.weight"), b.w_c3, 2, b.cout, b.mid, 1, b.mid, kWsEps, st, err));
            IgemmDesc d;
            d.X = b.t2; d.Wt = b.w_c3; d.M = Mout; d.N = b.cout; d.Cin = b.mid; d.ldx = b.mid; d.out_f32 = b.c3_raw;
            with_stats(c, d, b.c3_stats, b.cout, b.rout * b.rout);
            TRY(gemm_fwd(c, d, (size_t)Mout * b.mid, (size_t)b.cout * b.mid));
            GnApplyArgs g;
            g.raw = b.c3_raw; g.stats = b.c3_stats; g.gamma = c.W(k + "norm3.weight"); g.beta = c.W(k + "norm3.bias");
            if (b.proj) { g.raw2 = b.ds_raw; g.stats2 = b.ds_stats; g.gamma2 = c.W(k + "downsample.norm.weight"); g.beta2 = c.W(k + "downsample.norm.bias"); }
            else g.res = b.xin;
            g.out_f32 = b.out;
            ++cnt;
            if (stage < 2 && cnt == a.rn_layers[stage]) g.out_halo = T.feat[hook++];   // hooks on stages[0], stages[1] (vit.py:164-167)
            g.M = (size_t)Mout; g.HW = b.rout * b.rout; g.W = b.rout; g.C = b.cout; g.cpg = b.cout / 32;
            from_partials(c, g, 0, b.proj ? 1 : -1);
            TRY(launch_gn_apply(g, 2, st, err));
            if (cnt == a.rn_layers[stage]) { ++stage; cnt = 0; }
        }
        xcur = b.out;
    }
    (void)H2;
    // ---- ViT-B over g*g + 1 tokens ----
    const int E = a.vit_dim, G = a.grid(), NT = G * G + 1, Mt = B * NT, Mp = B * G * G;
    {
        IgemmDesc d;
        d.X = xcur; d.Wt = c.W(ENC + "patch_embed.proj.weight"); d.M = Mp; d.N = E; d.Cin = 1024; d.ldx = 1024; d.bias = c.W(ENC + "patch_embed.proj.bias"); d.out_f32 = Y.pe_y;
        TRY(gemm(c, d));
        TRY(launch_vit_tokens_ln(Y.pe_y, c.W(ENC + "cls_token"), c.W(ENC + "pos_embed"), Y.x0, c.W(vit_key(0) + "norm1.weight"), c.W(vit_key(0) + "norm1.bias"), Y.vb[0].ln1, 2,
                                 B, NT, E, kLnEps, st, err));
    }
    const float* tcur = Y.x0;
    for (int i = 0; i < a.vit_depth; ++i) {
        VitBlkT& v = Y.vb[i];
        const std::string k = vit_key(i);
        v.xin = tcur;
        if (i > 0) TRY(launch_ln_rows(const_cast<float*>(v.xin), c.W(k + "norm1.weight"), c.W(k + "norm1.bias"), v.ln1, 2, Mt, E, kLnEps, st, err));
        IgemmDesc d;
        d.X = v.ln1; d.Wt = c.W(k + "attn.qkv.weight"); d.M = Mt; d.N = 3 * E; d.Cin = E; d.ldx = E; d.bias = c.W(k + "attn.qkv.bias"); d.out_f32 = v.qkv;
        TRY(gemm_fwd(c, d, (size_t)Mt * E, (size_t)3 * E * E));   // x3 operands in the amp modes (train_step.cpp: gemm_fwd), exact f32 otherwise
        // forward: the exact-f32 MFMA kernel of the inference path (vit_attention.hip); the VALU pair of round 2 stays behind SOCCDPT_ATTN_BWD_VALU
        static const bool attn_valu = getenv("SOCCDPT_ATTN_BWD_VALU") != nullptr;
        if (attn_valu) TRY(th_vit_attention_fwd(v.qkv, v.attn, v.rowstat, Y.attn_part, B, NT, a.vit_heads, st, err));
        else TRY(launch_vit_attention(v.qkv, v.attn, SOCCDPT_PREC_F32, B, NT, a.vit_heads, st, err));
        d = IgemmDesc();
        d.X = v.attn; d.Wt = c.W(k + "attn.proj.weight"); d.M = Mt; d.N = E; d.Cin = E; d.ldx = E; d.bias = c.W(k + "attn.proj.bias"); d.res1 = v.xin; d.out_f32 = v.x1;
        TRY(gemm_fwd(c, d, (size_t)Mt * E, (size_t)E * E));
        TRY(launch_ln_rows(v.x1, c.W(k + "norm2.weight"), c.W(k + "norm2.bias"), v.ln2, 2, Mt, E, kLnEps, st, err));
        d = IgemmDesc();
        d.X = v.ln2; d.Wt = c.W(k + "mlp.fc1.weight"); d.M = Mt; d.N = 4 * E; d.Cin = E; d.ldx = E; d.bias = c.W(k + "mlp.fc1.bias"); d.act = ACT_GELU;
        d.out_f32 = v.hpre; d.out_op = v.hact;
        TRY(gemm_fwd(c, d, (size_t)Mt * E, (size_t)4 * E * E));
        d = IgemmDesc();
        d.X = v.hact; d.Wt = c.W(k + "mlp.fc2.weight"); d.M = Mt; d.N = E; d.Cin = 4 * E; d.ldx = 4 * E; d.bias = c.W(k + "mlp.fc2.bias"); d.res1 = v.x1; d.out_f32 = v.xout;
        TRY(gemm_fwd(c, d, (size_t)Mt * 4 * E, (size_t)4 * E * E));
        tcur = v.xout;
    }
    // ---- act_postprocess3 / 4: ProjectReadout -> Conv1x1 (-> Conv3x3 / 2) ----
    for (int k = 0; k < 2; ++k) {
        const std::string ap = HYB + "act_postprocess" + std::to_string(3 + k) + ".";
        TRY(th_readout_cat(Y.vb[a.vit_hooks[k]].xout, Y.cat[k], B, NT, E, st, err));
        IgemmDesc d;
        d.X = Y.cat[k]; d.Wt = c.W(ap + "0.project.0.weight"); d.M = Mp; d.N = E; d.Cin = 2 * E; d.ldx = 2 * E; d.bias = c.W(ap + "0.project.0.bias"); d.act = ACT_GELU;
        d.out_f32 = Y.ro_pre[k]; d.out_op = Y.ro_act[k];
        TRY(gemm(c, d));
        d = IgemmDesc();
        d.X = Y.ro_act[k]; d.Wt = c.W(ap + "3.weight"); d.M = Mp; d.N = a.fdim(2 + k); d.Cin = E; d.ldx = E; d.bias = c.W(ap + "3.bias"); d.H = G; d.W = G;
        d.out_op = k == 0 ? T.feat[2] : Y.pp4_in; d.out_halo = 1;
        TRY(gemm(c, d));
        if (k == 1) {
            TRY(launch_conv_w(c.W(ap + "4.weight"), nullptr, Y.w_pp4, 1, 0, a.fdim(3), a.fdim(3), st, err));
            d = IgemmDesc();
            d.X = Y.pp4_in; d.Wt = Y.w_pp4; d.M = B * (G / 2) * (G / 2); d.N = a.fdim(3); d.Cin = a.fdim(3); d.taps = 9; d.H = G / 2; d.W = G / 2; d.Hi = G; d.Wi = G;
            d.stride = 2; d.pad = 1; d.in_halo = 1; d.bias = c.W(ap + "4.bias"); d.out_op = T.feat[3]; d.out_halo = 1;
            TRY(gemm(c, d));
        }
    }
    return 0;
}

int hy_backward(Ctx& c) {
    Handle& h = c.h;
    const Arch& a = h.arch;
    Tape& T = c.T;
    HyTape& Y = T.hy;
    const int B = c.B;
    hipStream_t st = c.st;
    std::string& err = c.err;
    float** G = T.G;
    const int E = a.vit_dim, Gd = a.grid(), NT = Gd * Gd + 1;
    const size_t Mt = (size_t)B * NT, Mp = (size_t)B * Gd * Gd;
    // re-derive the input pointers of the forward walk
    {
        const float* xcur = Y.pool;
        for (auto& b : Y.blk) { b.xin = xcur; xcur = b.out; }
        const float* tcur = Y.x0;
        for (auto& v : Y.vb) { v.xin = tcur; tcur = v.xout; }
    }
    // weight-standardised convolution: gradient of the standardised weights (tap-major, in S_dw) -> parameter gradient
    auto ws_grad = [&](const std::string& key, const float* wh, int Cout, int Cin, int k, int Kpad) -> int {
        float* dw = c.Gd(key);
        if (!dw) return 0;
        return th_ws_bwd(T.S_dw, wh, c.W(key), dw, Cout, Cin, k, Kpad, kWsEps, st, err);
    };
    // ---- read-outs: gradient of a hooked token stream from the gradient of its reassembled map ----
    auto readout_bwd = [&](int k, int accumulate) -> int {
        const std::string ap = HYB + "act_postprocess" + std::to_string(3 + k) + ".";
        const float* dmap = T.DF[2 + k];
        if (k == 1) {   // Conv2d(768, 768, 3, stride 2, padding 1)
            TRY(conv_gen_bwd(c, dmap, Y.pp4_in, Y.w_pp4, Gd, Gd / 2, a.fdim(3), a.fdim(3), 2, 1, G[0], c.Gd(ap + "4.weight") ? T.S_dw : nullptr, c.Gd(ap + "4.bias")));
            if (float* dw = c.Gd(ap + "4.weight")) TRY(tr_wgrad_permute(T.S_dw, dw, a.fdim(3), a.fdim(3), st, err));
            dmap = G[0];
        }
        TRY(linear_bwd(c, dmap, Y.ro_act[k], c.W(ap + "3.weight"), Mp, a.fdim(2 + k), E, G[1], nullptr, c.Gd(ap + "3.weight"), c.Gd(ap + "3.bias")));
        TRY(tr_gelu_bwd(G[1], Y.ro_pre[k], G[1], Mp * E, st, err));
        TRY(linear_bwd(c, G[1], Y.cat[k], c.W(ap + "0.project.0.weight"), Mp, E, 2 * E, G[2], nullptr, c.Gd(ap + "0.project.0.weight"), c.Gd(ap + "0.project.0.bias")));
        TRY(th_readout_cat_bwd(G[2], Y.GT, B, NT, E, accumulate, st, err));
        return 0;
    };
    // ---- ViT blocks, last -> first ----
    for (int i = a.vit_depth - 1; i >= 0; --i) {
        if (i == a.vit_hooks[1]) TRY(readout_bwd(1, 0));     // blocks[11] is the last block: its output reaches the outputs through the read-out only
        if (i == a.vit_hooks[0]) TRY(readout_bwd(0, 1));
        if (i > a.vit_hooks[1]) continue;
        VitBlkT& v = Y.vb[i];
        const std::string k = vit_key(i);
        // xout = x1 + fc2(gelu(fc1(LN2(x1))))
        TRY(linear_bwd(c, Y.GT, v.hact, c.W(k + "mlp.fc2.weight"), Mt, E, 4 * E, G[0], nullptr, c.Gd(k + "mlp.fc2.weight"), c.Gd(k + "mlp.fc2.bias")));
        TRY(tr_gelu_bwd(G[0], v.hpre, G[0], Mt * 4 * E, st, err));
        TRY(linear_bwd(c, G[0], v.ln2, c.W(k + "mlp.fc1.weight"), Mt, 4 * E, E, G[1], nullptr, c.Gd(k + "mlp.fc1.weight"), c.Gd(k + "mlp.fc1.bias")));
        TRY(ln_bwd(c, v.x1, c.W(k + "norm2.weight"), G[1], G[2], G[3], Mt, E, c.Gd(k + "norm2.weight"), c.Gd(k + "norm2.bias"), kLnEps));
        TRY(tr_axpy(G[2], Y.GT, Mt * E, st, err));                                   // G2 = d x1
        // x1 = xin + proj(attn(qkv(LN1(xin))))
        TRY(linear_bwd(c, G[2], v.attn, c.W(k + "attn.proj.weight"), Mt, E, E, G[0], nullptr, c.Gd(k + "attn.proj.weight"), c.Gd(k + "attn.proj.bias")));
        static const bool attn_valu = getenv("SOCCDPT_ATTN_BWD_VALU") != nullptr;
        if (attn_valu) TRY(th_vit_attention_bwd(v.qkv, v.attn, G[0], v.rowstat, Y.attn_part, G[4], B, NT, a.vit_heads, st, err));
        else {
            static const bool attn_f32 = getenv("SOCCDPT_ATTN_BWD_F32") != nullptr;   // A/B: keep the exact products in the amp modes too
            const int attn_op = (!attn_f32 && (c.h.train_amp == 1 || c.h.train_amp == 2)) ? c.h.train_amp : 0;
            TRY(th_vit_attention_bwd_mfma(v.qkv, v.attn, G[0], v.rowstat, G[4], B, NT, a.vit_heads, st, err, attn_op));
        }
        TRY(linear_bwd(c, G[4], v.ln1, c.W(k + "attn.qkv.weight"), Mt, 3 * E, E, G[1], nullptr, c.Gd(k + "attn.qkv.weight"), c.Gd(k + "attn.qkv.bias")));
        TRY(ln_bwd(c, v.xin, c.W(k + "norm1.weight"), G[1], G[0], G[3], Mt, E, c.Gd(k + "norm1.weight"), c.Gd(k + "norm1.bias"), kLnEps));
        TRY(copy_d2d(c, Y.GT, G[2], Mt * E * 4, "hy_backward"));
        TRY(tr_axpy(Y.GT, G[0], Mt * E, st, err));                                   // GT = d xin
    }
    // ---- tokens = cat(cls, proj(features)) + pos_embed ----
    {
        float* dpos = c.Gd(ENC + "pos_embed");
        float* dcls = c.Gd(ENC + "cls_token");
        if (dpos || dcls) {
            // sum over the batch of the token-stream gradient: [B][NT*E] -> [NT*E]  (tr_colsum: rows = samples)
            float* tmp = dpos ? dpos : G[3];
            TRY(tr_colsum(Y.GT, nullptr, tmp, T.S_col, (size_t)B, NT * E, 0, st, err));
            if (dcls) TRY(copy_d2d(c, dcls, tmp, (size_t)E * 4, "hy_backward"));
        }
        TRY(th_tokens_to_patches(Y.GT, G[0], B, NT, E, st, err));
        TRY(linear_bwd(c, G[0], Y.blk.back().out, c.W(ENC + "patch_embed.proj.weight"), Mp, E, 1024, Y.GR, nullptr, c.Gd(ENC + "patch_embed.proj.weight"),
                       c.Gd(ENC + "patch_embed.proj.bias")));
    }
    // ---- ResNetV2 stages, last block -> first ----
    int stage = 2, cnt = a.rn_layers[2];
    for (int bi = (int)Y.blk.size() - 1; bi >= 0; --bi) {
        RnBlkT& b = Y.blk[bi];
        const std::string& k = b.key;
        const size_t Min = (size_t)B * b.rin * b.rin, Mout = (size_t)B * b.rout * b.rout;
        if (stage < 2 && cnt == a.rn_layers[stage]) TRY(tr_axpy(Y.GR, T.DF[stage], Mout * b.cout, st, err));   // hooked stage output
        // out = relu(GN3(conv3(t2)) + shortcut)
        TRY(tr_relu_bwd(Y.GR, b.out, nullptr, G[0], Mout * b.cout, st, err));                       // G0 = d (sum)
        TRY(th_gn_bwd(G[0], b.c3_raw, b.c3_stats, c.W(k + "norm3.weight"), c.W(k + "norm3.bias"), G[1], c.Gd(k + "norm3.weight"), c.Gd(k + "norm3.bias"), T.S_col, B,
                      b.rout * b.rout, b.cout, b.cout / 32, 0, st, err));
        TRY(linear_bwd(c, G[1], b.t2, b.w_c3, Mout, b.cout, b.mid, G[2], nullptr, c.Gd(k + "conv3.weight") ? T.S_dw : nullptr, nullptr));
        TRY(ws_grad(k + "conv3.weight", b.w_c3, b.cout, b.mid, 1, b.mid));
        TRY(th_gn_bwd(G[2], b.c2_raw, b.c2_stats, c.W(k + "norm2.weight"), c.W(k + "norm2.bias"), G[2], c.Gd(k + "norm2.weight"), c.Gd(k + "norm2.bias"), T.S_col, B,
                      b.rout * b.rout, b.mid, b.mid / 32, 1, st, err));
        TRY(conv_gen_bwd(c, G[2], b.t1, b.w_c2, b.rin, b.rout, b.mid, b.mid, b.stride, b.stride == 1 ? 1 : 0, G[1], c.Gd(k + "conv2.weight") ? T.S_dw : nullptr, nullptr));
        TRY(ws_grad(k + "conv2.weight", b.w_c2, b.mid, b.mid, 3, 9 * b.mid));
        TRY(th_gn_bwd(G[1], b.c1_raw, b.c1_stats, c.W(k + "norm1.weight"), c.W(k + "norm1.bias"), G[1], c.Gd(k + "norm1.weight"), c.Gd(k + "norm1.bias"), T.S_col, B,
                      b.rin * b.rin, b.mid, b.mid / 32, 1, st, err));
        TRY(linear_bwd(c, G[1], b.xin, b.w_c1, Min, b.mid, b.cin, G[3], nullptr, c.Gd(k + "conv1.weight") ? T.S_dw : nullptr, nullptr));   // G3 = d xin (conv path)
        TRY(ws_grad(k + "conv1.weight", b.w_c1, b.mid, b.cin, 1, b.cin));
        if (!b.proj) {
            TRY(tr_axpy(G[3], G[0], Min * b.cin, st, err));                                        // identity shortcut
        } else {
            TRY(th_gn_bwd(G[0], b.ds_raw, b.ds_stats, c.W(k + "downsample.norm.weight"), c.W(k + "downsample.norm.bias"), G[2], c.Gd(k + "downsample.norm.weight"),
                          c.Gd(k + "downsample.norm.bias"), T.S_col, B, b.rout * b.rout, b.cout, b.cout / 32, 0, st, err));
            const float* xs = b.xin;
            if (b.stride != 1) { TRY(th_stride_gather(b.xin, Y.xg, B, b.rin, b.rout, b.cin, b.stride, st, err)); xs = Y.xg; }
            TRY(linear_bwd(c, G[2], xs, b.w_ds, Mout, b.cout, b.cin, G[1], nullptr, c.Gd(k + "downsample.conv.weight") ? T.S_dw : nullptr, nullptr));
            TRY(ws_grad(k + "downsample.conv.weight", b.w_ds, b.cout, b.cin, 1, b.cin));
            if (b.stride == 1) TRY(tr_axpy(G[3], G[1], Min * b.cin, st, err));
            else TRY(th_stride_scatter_add(G[1], G[3], B, b.rin, b.rout, b.cin, b.stride, st, err));
        }
        TRY(copy_d2d(c, Y.GR, G[3], Min * b.cin * 4, "hy_backward"));
        if (--cnt == 0 && stage > 0) { --stage; cnt = a.rn_layers[stage]; }
    }
    // ---- stem ----
    {
        const std::string bb = ENC + "patch_embed.backbone.";
        const int H1 = a.img / 2;
        const size_t M1s = (size_t)B * H1 * H1;
        float* dwk = c.Gd(bb + "stem.conv.weight");
        float* dg = c.Gd(bb + "stem.norm.weight");
        float* dbt = c.Gd(bb + "stem.norm.bias");
        if (dwk || dg || dbt) {
            TRY(th_maxpool_bwd(Y.GR, Y.stem_raw, Y.stem_stats, c.W(bb + "stem.norm.weight"), c.W(bb + "stem.norm.bias"), Y.pool_idx, G[0], B, H1, a.stem_ch, a.stem_ch / 32, st,
                               err));
            TRY(th_gn_bwd(G[0], Y.stem_raw, Y.stem_stats, c.W(bb + "stem.norm.weight"), c.W(bb + "stem.norm.bias"), G[0], dg, dbt, T.S_col, B, H1 * H1, a.stem_ch,
                          a.stem_ch / 32, 1, st, err));
            if (dwk) {
                TRY(linear_bwd(c, G[0], Y.a0, nullptr, M1s, a.stem_ch, 160, nullptr, nullptr, T.S_dw, nullptr));
                TRY(th_ws_bwd(T.S_dw, Y.w_stem, c.W(bb + "stem.conv.weight"), dwk, a.stem_ch, 3, 7, 160, kWsEps, st, err));
            }
        }
    }
    return 0;
}

}  // namespace trn
}  // namespace soccdpt
