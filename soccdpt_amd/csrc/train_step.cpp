// Training step of SOccDPT_V3 (Swin-V2 encoders here, the ViT-hybrid encoder in train_hybrid_step.cpp; decoder and heads shared): train-mode
// forward that keeps every activation the backward needs (the "tape"),
// and the backward that autograd runs for the reference (scripts/train_SOccDPT.py:360-393 over model/SOccDPT.py:660-685, model/dpt.py:142-232,
// model/blocks.py:391-497 and timm's SwinTransformerV2).  Arithmetic by soccdpt_train_set_amp: 0 = exact f32 (every GEMM-shaped gradient through the f32
// MFMA igemm), 3 = x3 split-fp16 operands in every GEMM of the step (f32-grade), 1 / 2 = bf16 / fp16 operands in the gradient GEMMs with an x3 forward;
// the tape, the weights and the gradients are f32 in every mode.  Weight gradients: split-K igemm over transposed operands (gemm_wgrad), or -- 16-bit and x3
// modes, shapes permitting -- from the operands as stored (train_wgrad_tn.hip).  Attention backward: train_attn.hip.  The rest: train.hip.
// Train mode differs from eval in the seg head (model/SOccDPT.py:660-671: BatchNorm2d on batch statistics with running-buffer updates, Dropout(0.1) live)
// and in the encoder's stochastic depth (timm's drop_path_rate = 0.1, soccdpt_train_set_drop_path).
//
// Gradients are WRITTEN (not accumulated) to the buffers bound with soccdpt_bind_grad; a weight without a bound gradient is frozen and its
// weight-gradient GEMM is skipped (the reference freezes / partially unfreezes the encoder: model/loss.py:110-152).
#include <cstdio>
#include <cstdlib>

#include "train_internal.h"

#include <cstring>

namespace soccdpt {
namespace trn {

// One walk decides the layout; with base == nullptr it only measures.
void carve(const Handle& h, int B, TArena& ar, Tape& T) {
    const Arch& a = h.arch;
    const int F = h.cfg.features, G = a.grid();
    const size_t M0 = (size_t)B * G * G;
    auto halo = [&](int r, int C) { return ar.f((size_t)B * (r + 2) * (r + 2) * C); };
    // ---- halo zone ----
    ar.f(0);
    T.halo_lo = (ar.off + 255) & ~size_t(255);
    for (int l = 0; l < 4; ++l) {
        const int r = a.fres(l);
        T.feat[l] = halo(r, a.fdim(l));
        T.lrn_relu[l] = halo(r, F);
        T.t1[l] = l < 3 ? halo(r, F) : nullptr;
        T.out_relu[l] = l < 3 ? halo(r, F) : nullptr;
        T.t2[l] = halo(r, F);
    }
    const int r1 = 2 * a.fres(0), r0 = 4 * a.fres(0);
    T.path1 = halo(r1, F);
    T.d1u = halo(r0, F / 2);
    T.sk_count = reinterpret_cast<unsigned*>(ar.f(kTrainSkCountWords));   // split-K arrival counters: zero at rest (zeroed with the halos)
    if (a.hybrid) hy_carve_halo(h, B, ar, T);
    ar.f(0);
    T.halo_hi = (ar.off + 255) & ~size_t(255);
    size_t maxAct = M0 * 64;
    size_t maxDS = 0, maxStat = 0, maxTab = 0, maxPart = 0;
    if (a.hybrid) hy_carve(h, B, ar, T, maxAct);
    else {
    // ---- encoder tape ----
    T.patches = ar.f(M0 * 64);
    T.pe_wpad = ar.f((size_t)a.embed * 64);
    T.pe_pre = ar.f(M0 * a.embed);
    T.x0 = ar.f(M0 * a.embed);
    for (int s = 0; s < 4; ++s) {
        const int C = a.dim(s), res = a.res(s), H = a.heads[s], ws = a.ws(s);
        const size_t M = (size_t)B * res * res;
        T.blk[s].assign(a.depths[s], BlkT{});
        for (int j = 0; j < a.depths[s]; ++j) {
            BlkT& b = T.blk[s][j];
            b.qkv_bias = ar.f(3 * C);
            b.scale = ar.f(H);
            b.table = ar.f((size_t)(2 * ws - 1) * (2 * ws - 1) * H);
            b.bias_acc = ar.f(attn_bias_elems(ws, H));
            b.dp = ar.f(2 * (size_t)B + 2);
            b.qkv = ar.f(M * 3 * C);
            b.attn = ar.f(M * C);
            b.a_pre = ar.f(M * C);
            b.x1 = ar.f(M * C);
            b.hpre = ar.f(M * 4 * C);
            b.hact = ar.f(M * 4 * C);
            b.m_pre = ar.f(M * C);
            b.xout = ar.f(M * C);
        }
        maxAct = std::max(maxAct, M * 4 * C);
        const size_t nwin = (size_t)B * (res / ws) * (res / ws), N = (size_t)ws * ws;
        maxDS = std::max(maxDS, nwin * H * N * N);
        maxStat = std::max(maxStat, nwin * H * N * 2);
        maxPart = std::max(maxPart, nwin * H * ((N + 63) / 64));
        maxTab = std::max(maxTab, (size_t)(2 * ws - 1) * (2 * ws - 1) * H);
        if (s < 3) {
            T.mg[s] = ar.f(M * C);            // [M/4][4C]
            T.mr_pre[s] = ar.f(M / 4 * 2 * C);
            T.mx[s] = ar.f(M / 4 * 2 * C);
        }
    }
    }
    // ---- decoder tape ----
    for (int l = 0; l < 4; ++l) {
        const size_t M = (size_t)B * a.fres(l) * a.fres(l);
        T.lrn_raw[l] = ar.f(M * F);
        T.out_raw[l] = l < 3 ? ar.f(M * F) : nullptr;
        T.u[l] = ar.f(M * F);
        T.oc[l] = ar.f(M * F);
        T.w_lrn[l] = ar.f((size_t)F * 9 * a.fdim(l));
        for (int u = 0; u < 2; ++u)
            for (int c = 0; c < 2; ++c) T.w_rcu[l][u][c] = ar.f((size_t)F * 9 * F);
        maxAct = std::max(maxAct, M * F);
    }
    const size_t M1 = (size_t)B * r1 * r1, M0p = (size_t)B * r0 * r0;
    T.d1 = ar.f(M1 * (F / 2));
    T.e = ar.f(M0p * 32);
    T.inv = ar.f(M0p);
    T.seg = ar.f(M0p * 3);
    T.w_d0 = ar.f((size_t)(F / 2) * 9 * F);
    T.w_d2 = ar.f((size_t)32 * 9 * (F / 2));
    T.w_s0 = ar.f((size_t)F * 9 * F);
    T.c_raw = ar.f(M1 * F);
    T.bn_stats = ar.f(2 * F);
    T.r = ar.f(M1 * F);
    T.logits = ar.f(M1 * 4);
    T.keep = reinterpret_cast<uint8_t*>(ar.f((M1 * F + 3) / 4));
    maxAct = std::max(maxAct, std::max(M1 * F, M0p * (size_t)(F / 2)));
    maxAct = std::max(maxAct, M0p * 33);
    T.maxAct = maxAct;
    // ---- backward scratch ----
    const size_t Cmax0 = a.hybrid ? 1024 : (size_t)a.dim(3);
    for (auto& g : T.G) g = ar.f(maxAct);
    T.GX = ar.f(a.hybrid ? 64 : M0 * a.embed);   // the largest token-stream gradient (stage 0; later stages are smaller)
    T.GP = ar.f(M1 * F);
    T.DOC = ar.f((size_t)B * a.fres(0) * a.fres(0) * F);
    for (int l = 0; l < 4; ++l) T.DF[l] = ar.f((size_t)B * a.fres(l) * a.fres(l) * a.fdim(l));
    T.S_T1 = ar.f(maxAct + 128 * 4 * Cmax0);
    T.S_T2 = ar.f(std::max(std::max(M1 * 9 * F, M0p * 9 * (size_t)(F / 2)), maxAct) + 128 * 9 * Cmax0);
    T.S_halo = ar.f(std::max((size_t)B * (r1 + 2) * (r1 + 2) * F, (size_t)B * (r0 + 2) * (r0 + 2) * (size_t)(F / 2)) + 64 * (size_t)F);   // + the k-tile padding rows of train_wgrad_tn.hip
    const size_t Cmax = a.hybrid ? 1024 : a.dim(3);
    const size_t wmax = std::max(std::max((size_t)9 * F * F, 4 * Cmax * Cmax), a.hybrid ? (size_t)9 * 768 * 768 : 0);
    T.S_wt = ar.f(wmax);
    T.S_dw = ar.f(wmax);
    T.S_dw_n = wmax;
    T.S_col = ar.f((size_t)1 << 20);
    {   // slots of the staged dgrad weight operands (stage_weights): Linear [N][K] and 1x1 conv weights with N, K multiples of 32 and K > 32, 3x3 conv weights with N, C multiples of 32
        T.wt_off.assign(h.weights.size(), -1);
        size_t tot = 0;
        for (size_t i = 0; i < h.weights.size(); ++i) {
            const auto& sh = h.weights[i].shape;
            const bool lin = (sh.size() == 2 || (sh.size() == 4 && sh[2] == 1 && sh[3] == 1)) && sh[0] % 32 == 0 && sh[1] % 32 == 0 && sh[1] > 32;
            const bool c3 = sh.size() == 4 && sh[2] == 3 && sh[3] == 3 && sh[0] % 32 == 0 && sh[1] % 32 == 0;
            if (!lin && !c3) continue;
            T.wt_off[i] = (long long)tot;
            tot += (h.weights[i].numel() + 63) & ~size_t(63);
        }
        T.WT = ar.f(tot);
    }
    T.sk_part = ar.f(kTrainSkPartFloats);
    T.tn_arena = ar.f(kTrainTnArenaFloats);
    T.S_vec = ar.f(std::max((size_t)4 * Cmax, (size_t)4 * F) * 2);
    T.dS = ar.f(maxDS);
    T.rowstat = ar.f(maxStat);
    T.dscale_part = ar.f(maxPart * 4);
    T.attn_part = ar.f(a.hybrid ? 64 : 4 * (M0 * 3 * a.embed + maxStat));
    T.dtable = ar.f(maxTab);
    T.dt = ar.f(maxTab);
    T.S_cpb = ar.f(a.hybrid ? 64 : (size_t)2 * (2 * a.window - 1) * (2 * a.window - 1) * 512);
}

// x3: the operands are x3 split-fp16 tensors (train amp mode 3: three fp16 MFMAs per product, f32-grade results; same tile set and
// split-K decisions as the exact-f32 GEMMs)
int gemm(Ctx& c, IgemmDesc d, bool x3) {
    d.f32 = x3 ? 0 : 1;
    d.x3 = x3 ? 1 : 0;
    // Small grids with a long K (coarse decoder levels, stage-3 Linear layers, their dgrads): split K like the weight-gradient GEMMs do
    const long tiles = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
    const long nk = (long)d.taps * d.Cin / 32;
    if (tiles <= 96 && nk >= 48 && !d.ln_g && !d.gn_stats && !d.out_dot && d.stride == 1 && d.pad == 1 && d.in_halo == 1 && !d.Hi && !d.gather1 && !d.grp_rows &&
        (size_t)tiles <= kTrainSkCountWords) {
        long S = (256 + tiles - 1) / tiles;
        if (S > nk / 8) S = nk / 8;
        if (S > 16) S = 16;
        while (S > 1 && (size_t)S * d.M * d.N > kTrainSkPartFloats) --S;
        if (S > 1) { d.splitk = (int)S; d.sk_part = c.T.sk_part; d.sk_count = c.T.sk_count; d.sk_part_floats = kTrainSkPartFloats; d.sk_count_words = kTrainSkCountWords; }
    }
    return launch_igemm(d, c.st, c.err);
}

// Forward GEMM.  With any train amp mode (the 16-bit modes too: their forward stays f32-grade, so the ReLU masks are those of the f32 step) the operands -- f32 tape tensors and f32 (tap-major) weights -- are converted to the x3 split-fp16
// format into backward scratch (S_T2 / S_wt, idle during the forward) and the product runs as three fp16 MFMAs per k-step; everything the launch
// writes stays f32 (out_f32, and out_op as an f32 tensor: out_op_f32), so the tape and the backward are unchanged.  x_elems / w_elems: elements of
// the X buffer (a halo image counts its border) and of the weight matrix.
int gemm_fwd(Ctx& c, IgemmDesc d, size_t x_elems, size_t w_elems) {
    const bool x3 = c.h.train_amp != 0 && d.Cin % 32 == 0 && (d.taps == 9 || d.ldx % 16 == 0) && x_elems % 16 == 0 && w_elems % 16 == 0 && !d.ln_g &&
                    !d.grp_rows && !d.gather1 && d.stride == 1 && d.pad == 1 && d.in_halo == 1;
    if (!x3) return gemm(c, d);
    uint16_t* xs = reinterpret_cast<uint16_t*>(c.T.S_T2);
    uint16_t* wsx = reinterpret_cast<uint16_t*>(c.T.S_wt);
    TRY(tr_cvt_x3_pair(static_cast<const float*>(d.X), xs, x_elems, static_cast<const float*>(d.Wt), wsx, w_elems, c.st, c.err));
    d.X = xs; d.Wt = wsx; d.out_op_f32 = 1;
    return gemm(c, d, true);
}

// Weight-gradient GEMM over TRANSPOSED operands: few output tiles, K = pixels.  Split K so that about two workgroups per CU exist; the partial tiles are
// summed in split order -- by a second launch (sk_defer) for the big tiles and for >= 4 splits, by the last workgroup to arrive otherwise: deterministic
// either way.  amp: bf16 / fp16 operands (K padded to 128 by the caller).
int gemm_wgrad(Ctx& c, IgemmDesc d, bool bf16_operands, bool x3) {
    const bool amp = bf16_operands;   // the CALLER says what its staging kernels wrote (amp applies per GEMM: shapes that do not fit stay f32)
    d.f32 = (amp || x3) ? 0 : 1;
    d.x3 = x3 ? 1 : 0;
    d.f16 = c.h.train_amp == 2 ? 1 : 0;   // 16-bit format of the amp mode: bf16 (1) or fp16 with the caller's loss scaling (2)
    // Tiles.  Default for the wide layers (M, N multiples of 128, >= 8 tiles): the 8-wave 128 x 128 tile -- these launches are L2 -> LDS fill bound and it
    // carries 64 FLOP per staged byte (16-bit) against 21 for 32 x 64 -- with the DEFERRED reduction (igemm.h sk_defer).  Rounds 2 and early 3 measured
    // the big tiles slower (f32 4 waves 51.2 vs 45.2 ms, x3 37.5 vs 36.6 ms per step): that was the last-arriver reduction, one workgroup walking 14
    // partial tiles of 64 KB with L2-bypassing loads (~400 us per launch).  SOCCDPT_SK_SMALL_TILES=1 selects the round-2 forms for A/B.
    long tiles = amp ? (long)((d.M + 31) / 32) * ((d.N + 63) / 64) : (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
    long nk = (long)d.taps * d.Cin / (amp ? 128 : 32);
    static const bool small_tiles = getenv("SOCCDPT_SK_SMALL_TILES") != nullptr;
    const bool big = !small_tiles && d.M % 128 == 0 && d.N % 128 == 0 && d.Cin % 64 == 0 && (long)(d.M / 128) * (d.N / 128) >= 8 &&
                     (!d.wt_grp_rows || d.wt_grp_rows % 128 == 0);
    if (big) {
        d.tune = amp ? 46 : 3;
        tiles = (long)(d.M / 128) * (d.N / 128);
        nk = (long)d.taps * d.Cin / (amp ? 64 : 32);
    }
    long S = (512 + tiles - 1) / tiles;
    if (x3 || big) S = 512 / tiles > 0 ? 512 / tiles : 1;   // fill-bound at two workgroups per CU: at most ONE round of them (576 workgroups = 1.125 rounds cost 1 ms per step)
    if (S > nk / (amp ? 2 : 8)) S = nk / (amp ? 2 : 8);
    if (S > 64) S = 64;
    while (S > 1 && (size_t)S * d.M * d.N > kTrainSkPartFloats) --S;
    if (S > 1 && (size_t)tiles <= kTrainSkCountWords) {
        d.splitk = (int)S; d.sk_part = c.T.sk_part; d.sk_count = c.T.sk_count; d.sk_part_floats = kTrainSkPartFloats; d.sk_count_words = kTrainSkCountWords;
        static const bool no_defer = getenv("SOCCDPT_SK_NO_DEFER") != nullptr;
        if ((big || S >= 4) && !no_defer && d.N % 4 == 0) d.sk_defer = 1;   // many splits: sum them in a second chip-wide launch instead of in the last workgroup
    } else if (big) {
        d.tune = -1;
    }
    return launch_igemm(d, c.st, c.err);
}

// amp: a backward GEMM with bf16 operands (f32 accumulate, f32 outputs)
int gemm16(Ctx& c, IgemmDesc d) {
    d.f32 = 0;
    d.f16 = c.h.train_amp == 2 ? 1 : 0;
    return launch_igemm(d, c.st, c.err);
}

int copy_d2d(Ctx& c, void* dst, const void* src, size_t bytes, const char* what) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c.st);
    if (e != hipSuccess) { c.err = std::string(what) + ": " + hipGetErrorString(e); return 1; }
    return 0;
}

// A/B switch: SOCCDPT_BIAS_COLSUM=1 keeps the separate column-sum launches for the bias gradients of the layers whose weight gradient runs on wgrad_tn
static bool bias_in_wgrad() {
    static const bool off = getenv("SOCCDPT_BIAS_COLSUM") != nullptr;
    return !off;
}

// A/B switch: SOCCDPT_WGRAD_TRANSPOSE=1 keeps the transposing weight-gradient path of round 2 in the 16-bit amp modes
static bool wgrad_tn_on() {
    static const bool off = getenv("SOCCDPT_WGRAD_TRANSPOSE") != nullptr;
    return !off;
}

int stage_weights(Ctx& c) {
    static const bool off = getenv("SOCCDPT_NO_WEIGHT_BATCH") != nullptr;   // A/B switch: one staging launch per layer, as in rounds 2-3
    Tape& T = c.T;
    T.wt_by_ptr.clear();
    if (off || !T.WT) return 0;
    const int amp = c.h.train_amp;
    const int fmt = amp == 3 ? 3 : amp == 2 ? 1 : amp == 1 ? 0 : -1;
    TrBatchTable t;
    t.n = 0;
    int tiles = 0;
    auto flush = [&]() -> int {
        if (t.n) TRY(tr_weight_batch(t, tiles, fmt, c.st, c.err));
        t.n = 0;
        tiles = 0;
        return 0;
    };
    for (size_t i = 0; i < c.h.weights.size(); ++i) {
        if (T.wt_off[i] < 0 || !c.h.weights[i].ptr) continue;
        const auto& w = c.h.weights[i];
        TrBatchEntry& e = t.e[t.n++];
        e.src = w.ptr;
        e.dst = T.WT + T.wt_off[i];
        e.R = (int)w.shape[0];
        e.C = (int)w.shape[1];
        e.kind = (w.shape.size() == 4 && w.shape[2] == 3) ? 1 : 0;
        e.tile0 = tiles;
        tiles += e.kind ? (int)(((size_t)e.R * e.C * 9 + 1023) / 1024) : ((e.C + 31) / 32) * ((e.R + 31) / 32);
        T.wt_by_ptr[w.ptr] = T.wt_off[i];
        if (t.n == kTrBatchMax) TRY(flush());
    }
    return flush();
}

const void* staged_wt(const Ctx& c, const float* W) {
    const auto it = c.T.wt_by_ptr.find(W);
    return it == c.T.wt_by_ptr.end() ? nullptr : static_cast<const void*>(c.T.WT + it->second);
}

// y = x W^T + b backward.  dY [M][N], X [M][K], W [N][K].  dX_out = dY W (+ dX_res); dW = dY^T X; db = colsum(dY).
int linear_bwd(Ctx& c, const float* dY, const float* X, const float* W, size_t M, int N, int K, float* dX_out, const float* dX_res, float* dW, float* db) {
    Tape& T = c.T;
    const bool x3 = c.h.train_amp == 3 && N % 32 == 0 && K % 32 == 0 && K > 32;              // x3 split-fp16 operands (f32-grade, 4 bytes per element)
    const bool amp = (c.h.train_amp == 1 || c.h.train_amp == 2) && N % 32 == 0 && K % 4 == 0 && K > 32;   // mixed precision: 16-bit operands for the two gradient GEMMs
    const int F16 = c.h.train_amp == 2 ? 1 : 0;
    // Both operand conversions of the layer in ONE launch when the weight gradient takes its operands as stored (wgrad_tn): dY for the two gradient
    // GEMMs, X for the weight gradient (round 5; two launches of a scalar kernel per layer before)
    const bool tn16 = amp && dW && wgrad_tn_on() && tr_wgrad_tn_ok((M + 63) / 64 * 64, N, K, 1) && (M * N) % 4 == 0 && (M * K) % 4 == 0;
    const bool tn3 = x3 && dW && wgrad_tn_on() && tr_wgrad_tn_ok((M + 63) / 64 * 64, N, K, 1);
    bool staged = false;   // S_T1 holds dY and S_T2 holds X in the launch format
    // a gradient written into scratch (standardised ResNetV2 kernels, the padded patch embedding) is read by its caller's next launch: summed at once; parameter gradients wait for the batched sum
    TnDefer* const df = c.may_defer(dW, bias_in_wgrad() ? db : nullptr) ? &c.tn : nullptr;   // (the qkv bias gradient, e.g., goes through scratch into q_bias / v_bias)
    if (tn16) { TRY(tr_cvt16_pair(dY, reinterpret_cast<uint16_t*>(T.S_T1), M * N, X, reinterpret_cast<uint16_t*>(T.S_T2), M * K, F16, c.st, c.err)); staged = true; }
    else if (tn3) { TRY(tr_cvt_x3_pair(dY, T.S_T1, M * N, X, T.S_T2, M * K, c.st, c.err)); staged = true; }
    if (dX_out) {
        IgemmDesc d;
        d.M = (int)M; d.N = K; d.Cin = N; d.ldx = N; d.res1 = dX_res; d.out_f32 = dX_out;
        if (x3) {
            const void* w3 = staged_wt(c, W);
            uint16_t* a3 = reinterpret_cast<uint16_t*>(T.S_T1);
            if (!w3) { TRY(tr_transpose16(W, reinterpret_cast<uint16_t*>(T.S_wt), N, K, N, 3, c.st, c.err)); w3 = T.S_wt; }
            if (!staged) TRY(launch_cvt_bf16(dY, a3, M * N, 3, c.st, c.err));
            d.X = a3; d.Wt = w3;
            TRY(gemm(c, d, true));
        } else if (amp) {
            const void* w16 = staged_wt(c, W);
            uint16_t* a16 = reinterpret_cast<uint16_t*>(T.S_T1);
            if (!w16) { TRY(tr_transpose16(W, reinterpret_cast<uint16_t*>(T.S_wt), N, K, N, F16, c.st, c.err)); w16 = T.S_wt; }
            if (!staged) TRY(launch_cvt_bf16(dY, a16, M * N, F16 ? 5 : 0, c.st, c.err));   // fp16: IEEE conversion, an overflow of the scaled gradient becomes inf
            d.X = a16; d.Wt = w16;
            TRY(gemm16(c, d));
        } else {
            const void* wt = staged_wt(c, W);
            if (!wt) { TRY(tr_transpose(W, T.S_wt, N, K, N, c.st, c.err)); wt = T.S_wt; }   // [K][N]
            d.X = dY; d.Wt = wt;
            TRY(gemm(c, d));
        }
    }
    if (dW) {
        IgemmDesc d;
        d.M = N; d.N = K; d.out_f32 = dW;
        if (x3 && wgrad_tn_on() && tr_wgrad_tn_ok((M + 63) / 64 * 64, N, K, 1)) {
            // x3 operands as stored (train_wgrad_tn.hip: wgrad_tn_x3_kernel); zero bytes are x3 zeros
            char* a3 = reinterpret_cast<char*>(T.S_T1);
            char* x3p = reinterpret_cast<char*>(T.S_T2);
            const size_t Mp = (M + 63) / 64 * 64;
            if (!staged) {
                if (!dX_out) TRY(launch_cvt_bf16(dY, reinterpret_cast<uint16_t*>(a3), M * N, 3, c.st, c.err));
                TRY(launch_cvt_bf16(X, reinterpret_cast<uint16_t*>(x3p), M * K, 3, c.st, c.err));
            }
            if (Mp > M) {
                hipError_t e = hipMemsetAsync(a3 + M * N * 4, 0, (Mp - M) * N * 4, c.st);
                if (e == hipSuccess) e = hipMemsetAsync(x3p + M * K * 4, 0, (Mp - M) * K * 4, c.st);
                if (e != hipSuccess) { c.err = std::string("linear_bwd memset: ") + hipGetErrorString(e); return 1; }
            }
            // (the bias gradient = column sums of dY rides in the same launch: one more MFMA per fragment against ones, train_wgrad_tn.hip)
            TRY(tr_wgrad_tn(reinterpret_cast<uint16_t*>(a3), N, reinterpret_cast<uint16_t*>(x3p), K, Mp, N, K, 1, 0, 3, T.sk_part, kTrainSkPartFloats, dW, c.st, c.err, bias_in_wgrad() ? db : nullptr, df));
            if (db && !bias_in_wgrad()) TRY(tr_colsum(dY, nullptr, db, T.S_col, M, N, 0, c.st, c.err));
            return 0;
        } else if (x3) {
            const int Mp = (int)((M + 31) / 32 * 32);
            uint16_t* y3 = reinterpret_cast<uint16_t*>(T.S_T1);
            uint16_t* x3p = reinterpret_cast<uint16_t*>(T.S_T2);
            TRY(tr_transpose16(dY, y3, (int)M, N, Mp, 3, c.st, c.err));
            TRY(tr_transpose16(X, x3p, (int)M, K, Mp, 3, c.st, c.err));
            d.X = y3; d.Wt = x3p; d.Cin = Mp; d.ldx = Mp;
        } else if (amp && wgrad_tn_on() && tr_wgrad_tn_ok((M + 63) / 64 * 64, N, K, 1)) {
            // operands as stored (train_wgrad_tn.hip): dY in 16 bit is what the dgrad launch above already staged; X needs one conversion, no transposes.
            // Token counts that are not a k-tile multiple (577-token ViT sequences) get zero rows appended to both operands.
            uint16_t* a16 = reinterpret_cast<uint16_t*>(T.S_T1);
            uint16_t* x16 = reinterpret_cast<uint16_t*>(T.S_T2);
            const size_t Mp = (M + 63) / 64 * 64;
            if (!staged) {
                if (!dX_out) TRY(launch_cvt_bf16(dY, a16, M * N, F16 ? 5 : 0, c.st, c.err));
                TRY(launch_cvt_bf16(X, x16, M * K, F16 ? 5 : 0, c.st, c.err));
            }
            if (Mp > M) {
                hipError_t e = hipMemsetAsync(a16 + M * N, 0, (Mp - M) * N * 2, c.st);
                if (e == hipSuccess) e = hipMemsetAsync(x16 + M * K, 0, (Mp - M) * K * 2, c.st);
                if (e != hipSuccess) { c.err = std::string("linear_bwd memset: ") + hipGetErrorString(e); return 1; }
            }
            TRY(tr_wgrad_tn(a16, N, x16, K, Mp, N, K, 1, 0, F16, T.sk_part, kTrainSkPartFloats, dW, c.st, c.err, bias_in_wgrad() ? db : nullptr, df));
            if (db && !bias_in_wgrad()) TRY(tr_colsum(dY, nullptr, db, T.S_col, M, N, 0, c.st, c.err));
            return 0;
        } else if (amp) {
            const int Mp = (int)((M + 127) / 128 * 128);
            uint16_t* y16 = reinterpret_cast<uint16_t*>(T.S_T1);
            uint16_t* x16 = reinterpret_cast<uint16_t*>(T.S_T2);
            TRY(tr_transpose16(dY, y16, (int)M, N, Mp, F16, c.st, c.err));
            TRY(tr_transpose16(X, x16, (int)M, K, Mp, F16, c.st, c.err));
            d.X = y16; d.Wt = x16; d.Cin = Mp; d.ldx = Mp;
        } else {
            const int Mp = (int)((M + 31) / 32 * 32);                    // k-tile multiple; the padding rows are zero
            TRY(tr_transpose(dY, T.S_T1, (int)M, N, Mp, c.st, c.err));   // [N][Mp]
            TRY(tr_transpose(X, T.S_T2, (int)M, K, Mp, c.st, c.err));    // [K][Mp]
            d.X = T.S_T1; d.Wt = T.S_T2; d.Cin = Mp; d.ldx = Mp;
        }
        TRY(gemm_wgrad(c, d, amp, x3));
    }
    if (db) TRY(tr_colsum(dY, nullptr, db, T.S_col, M, N, 0, c.st, c.err));
    return 0;
}

// y = conv3x3(Xhalo, W) + b backward.  dY plain [B*r*r][N], Xhalo [B][r+2][r+2][C], W [N][C][3][3].
int conv3_bwd(Ctx& c, const float* dY, const float* Xhalo, const float* W, int r, int N, int C, float* dX_out, const float* dX_res, float* dW, float* db,
              bool reuse_xt) {
    Tape& T = c.T;
    const int B = c.B;
    const size_t M = (size_t)B * r * r;
    const bool x3 = c.h.train_amp == 3 && N % 32 == 0 && C % 32 == 0;
    const bool amp = (c.h.train_amp == 1 || c.h.train_amp == 2) && N % 32 == 0 && C % 32 == 0;
    const int F16 = c.h.train_amp == 2 ? 1 : 0;
    if (dX_out) {
        const bool full = N % 8 == 0;   // the staging kernel writes the zero border itself; otherwise clear the buffer first
        if (!full) {
            const size_t hb = (size_t)B * (r + 2) * (r + 2) * N * (amp ? 2 : 4);
            hipError_t e = hipMemsetAsync(T.S_halo, 0, hb, c.st);
            if (e != hipSuccess) { c.err = std::string("conv3_bwd memset: ") + hipGetErrorString(e); return 1; }
        }
        IgemmDesc d;
        d.M = (int)M; d.N = C; d.Cin = N; d.taps = 9; d.H = r; d.W = r; d.res1 = dX_res; d.out_f32 = dX_out;
        if (x3) {
            uint16_t* h3 = reinterpret_cast<uint16_t*>(T.S_halo);
            const void* w3 = staged_wt(c, W);
            TRY(tr_to_halo_full(dY, h3, B, r, r, N, 3, c.st, c.err));
            if (!w3) { TRY(tr_conv_w_dgrad16(W, reinterpret_cast<uint16_t*>(T.S_wt), N, C, 3, c.st, c.err)); w3 = T.S_wt; }
            d.X = h3; d.Wt = w3;
            TRY(gemm(c, d, true));
        } else if (amp) {
            uint16_t* h16 = reinterpret_cast<uint16_t*>(T.S_halo);
            const void* w16 = staged_wt(c, W);
            TRY(tr_to_halo_full(dY, h16, B, r, r, N, 1 + F16, c.st, c.err));
            if (!w16) { TRY(tr_conv_w_dgrad16(W, reinterpret_cast<uint16_t*>(T.S_wt), N, C, F16, c.st, c.err)); w16 = T.S_wt; }
            d.X = h16; d.Wt = w16;
            TRY(gemm16(c, d));
        } else {
            if (full) TRY(tr_to_halo_full(dY, T.S_halo, B, r, r, N, 0, c.st, c.err));
            else TRY(tr_to_halo(dY, T.S_halo, B, r, r, N, c.st, c.err));
            const void* wt = staged_wt(c, W);
            if (!wt) { TRY(tr_conv_w_dgrad(W, T.S_wt, N, C, c.st, c.err)); wt = T.S_wt; }   // [C][9][N], rotated
            d.X = T.S_halo; d.Wt = wt;
            TRY(gemm(c, d));
        }
    }
    if (dW && (amp || x3) && wgrad_tn_on() && tr_wgrad_tn_ok((size_t)((size_t)B * (r + 2) * (r + 2) + 63) / 64 * 64, N, C, 9)) {
        // Operands as stored, in halo pixel order (train_wgrad_tn.hip): A = dY as the zero-bordered 16-bit image the dgrad launch staged, B = the input's halo image
        // converted to 16 bit; tap (ky, kx) reads B (ky - 1)(r + 2) + (kx - 1) rows further on.  K is padded to a k-tile with zero rows of A; B gets
        // zero margins of r + 3 rows on both sides (border pixels of A are zero, but 0 * NaN is not).
        const int rp = r + 2;
        const size_t Kh = (size_t)B * rp * rp, Kp = (Kh + 63) / 64 * 64, mrg = (size_t)rp + 1;
        const size_t es = x3 ? 4 : 2;                              // bytes per operand element
        const int fmt = x3 ? 3 : 1 + F16, cvt = x3 ? 3 : (F16 ? 5 : 0);
        char* h16 = reinterpret_cast<char*>(T.S_halo);
        char* xb = reinterpret_cast<char*>(T.S_T2);
        if (!dX_out) TRY(tr_to_halo_full(dY, h16, B, r, r, N, fmt, c.st, c.err));
        hipError_t e = Kp > Kh ? hipMemsetAsync(h16 + Kh * N * es, 0, (Kp - Kh) * N * es, c.st) : hipSuccess;
        if (!reuse_xt || T.xt_tn_src != Xhalo) {
            T.xt_tn_src = Xhalo;
            if (e == hipSuccess) e = hipMemsetAsync(xb, 0, mrg * C * es, c.st);
            if (e == hipSuccess) e = hipMemsetAsync(xb + (mrg + Kh) * C * es, 0, (Kp - Kh + mrg) * C * es, c.st);
            if (e == hipSuccess) TRY(launch_cvt_bf16(Xhalo, reinterpret_cast<uint16_t*>(xb + mrg * C * es), Kh * C, cvt, c.st, c.err));
        }
        if (e != hipSuccess) { c.err = std::string("conv3_bwd memset: ") + hipGetErrorString(e); return 1; }
        if (C % 4 == 0 && c.may_defer(dW, bias_in_wgrad() ? db : nullptr)) {   // deferred: the batched sum writes the parameter layout itself
            TRY(tr_wgrad_tn(reinterpret_cast<uint16_t*>(h16), N, reinterpret_cast<uint16_t*>(xb + mrg * C * es), C, Kp, N, C, 9, rp, x3 ? 3 : F16, T.sk_part, kTrainSkPartFloats,
                            dW, c.st, c.err, bias_in_wgrad() ? db : nullptr, &c.tn, C));
        } else {
        TRY(tr_wgrad_tn(reinterpret_cast<uint16_t*>(h16), N, reinterpret_cast<uint16_t*>(xb + mrg * C * es), C, Kp, N, C, 9, rp, x3 ? 3 : F16, T.sk_part, kTrainSkPartFloats,
                        T.S_dw, c.st, c.err, bias_in_wgrad() ? db : nullptr));   // dY in halo order: its zero border adds nothing to the column sums
        TRY(tr_wgrad_permute(T.S_dw, dW, N, C, c.st, c.err));
        }
        if (bias_in_wgrad()) db = nullptr;
    } else if (dW && x3 && C % 64 == 0) {
        // x3, no im2col: like the f32 form below, but an x3 tensor is cut in 8-element units, so the views must start at multiples of 16 elements:
        // the pixel order pads every halo row to rpp = roundup(r + 2, 16) pixels (vertical taps = +- rpp) and the horizontal taps read three copies
        // of the transposed image pre-shifted by -1 / 0 / +1 pixel (x_halo_T_kernel).  9 views of 3 copies instead of a 9-fold im2col^T.
        const int rp = r + 2, rpp = (rp + 15) / 16 * 16, Mh = B * rp * rpp, margin = rpp + 16;
        const int ld = (2 * margin + Mh + 31) / 32 * 32;
        const size_t head = (size_t)rpp + 16;                        // zeroed elements in front of and behind each copy (the +- rpp views)
        const size_t copy_elems = (size_t)C * ld + 2 * head;
        char* yT = reinterpret_cast<char*>(T.S_T1);
        char* xT = reinterpret_cast<char*>(T.S_T2);
        TRY(tr_dy_halo_T(dY, yT, 3, B, r, N, margin, ld, c.st, c.err, rpp));
        if (!reuse_xt || T.xt_tn_src) {
            T.xt_tn_src = nullptr;
            for (int kx = 0; kx < 3; ++kx) {
                char* base = xT + (size_t)kx * copy_elems * 4;
                hipError_t e = hipMemsetAsync(base, 0, head * 4, c.st);
                if (e == hipSuccess) e = hipMemsetAsync(base + (head + (size_t)C * ld) * 4, 0, head * 4, c.st);
                if (e != hipSuccess) { c.err = std::string("conv3_bwd memset: ") + hipGetErrorString(e); return 1; }
                TRY(tr_x_halo_T_x3(Xhalo, base + head * 4, B, r, C, margin - (kx - 1), ld, rpp, c.st, c.err));
            }
        }
        IgemmDesc d;
        d.X = yT; d.Wt = xT; d.M = N; d.N = 9 * C; d.Cin = ld; d.ldx = ld; d.out_f32 = T.S_dw;
        d.wt_grp_rows = C; d.wt_rp = rpp; d.wt_base = (int)head; d.wt_kx = (int)copy_elems;
        TRY(gemm_wgrad(c, d, false, true));
        TRY(tr_wgrad_permute(T.S_dw, dW, N, C, c.st, c.err));
    } else if (dW && (C % 64 != 0 || x3)) {   // (layer1_rn of tiny_256, C = 96: a weight tile would straddle two taps) explicit im2col^T
        IgemmDesc d;
        d.M = N; d.N = 9 * C; d.out_f32 = T.S_dw;
        if (x3) {
            const int Mp = (int)((M + 31) / 32 * 32);
            uint16_t* y3 = reinterpret_cast<uint16_t*>(T.S_T1);
            uint16_t* x3p = reinterpret_cast<uint16_t*>(T.S_T2);
            TRY(tr_transpose16(dY, y3, (int)M, N, Mp, 3, c.st, c.err));
            TRY(tr_im2colT16(Xhalo, x3p, B, r, r, C, (size_t)Mp, 3, c.st, c.err));
            d.X = y3; d.Wt = x3p; d.Cin = Mp; d.ldx = Mp;
        } else if (amp) {
            const int Mp = (int)((M + 127) / 128 * 128);
            uint16_t* y16 = reinterpret_cast<uint16_t*>(T.S_T1);
            uint16_t* x16 = reinterpret_cast<uint16_t*>(T.S_T2);
            TRY(tr_transpose16(dY, y16, (int)M, N, Mp, F16, c.st, c.err));
            TRY(tr_im2colT16(Xhalo, x16, B, r, r, C, (size_t)Mp, F16, c.st, c.err));
            d.X = y16; d.Wt = x16; d.Cin = Mp; d.ldx = Mp;
        } else {
            const int Mp = (int)((M + 31) / 32 * 32);
            TRY(tr_transpose(dY, T.S_T1, (int)M, N, Mp, c.st, c.err));             // [N][Mp]
            TRY(tr_im2colT(Xhalo, T.S_T2, B, r, r, C, (size_t)Mp, c.st, c.err));   // [9C][Mp]
            d.X = T.S_T1; d.Wt = T.S_T2; d.Cin = Mp; d.ldx = Mp;
        }
        TRY(gemm_wgrad(c, d, amp, x3));
        TRY(tr_wgrad_permute(T.S_dw, dW, N, C, c.st, c.err));
    } else if (dW) {
        // No im2col: both operands transposed in halo pixel order, tap (ky, kx) = the plain GEMM over a shifted view of the ONE transposed halo
        // image (train.hip: dy_halo_T_kernel).  Margins of r + 3 zero columns on both sides of every row absorb the shifts.
        const int rp = r + 2, Mh = B * rp * rp, margin = r + 3;
        const int ld = (2 * margin + Mh + 127) / 128 * 128;
        const size_t esz = amp ? 2 : 4;
        char* yT = reinterpret_cast<char*>(T.S_T1);
        char* xT = reinterpret_cast<char*>(T.S_T2);            // amp: two copies, shifted by 0 / 1 element, so that every tap's base stays 4-byte aligned
        const size_t head = (size_t)(margin + 13) / 8 * 8;     // zeroed elements in FRONT of each copy: the most negative shift reads base - (r + 4)
        const size_t copy_bytes = ((size_t)(C + 1) * ld + 64 + head) * esz;
        xT += head * esz;
        TRY(tr_dy_halo_T(dY, yT, amp ? 1 + F16 : 0, B, r, N, margin, ld, c.st, c.err));
        if (!reuse_xt || T.xt_tn_src) {
            T.xt_tn_src = nullptr;
            for (int cp = 0; cp < (amp ? 2 : 1); ++cp) {
                char* base = xT + cp * copy_bytes;
                // headroom + the first row's left margin; every other gap is the zero tail of a row (tr_transpose pads rows up to ld)
                hipError_t e = hipMemsetAsync(base - head * esz, 0, (head + margin) * esz, c.st);
                if (e != hipSuccess) { c.err = std::string("conv3_bwd memset: ") + hipGetErrorString(e); return 1; }
                if (amp) TRY(tr_transpose16(Xhalo, reinterpret_cast<uint16_t*>(base) + margin - cp, Mh, C, ld, F16, c.st, c.err));
                else TRY(tr_transpose(Xhalo, reinterpret_cast<float*>(base) + margin, Mh, C, ld, c.st, c.err));
            }
        }
        {   // ONE GEMM: M = N out-channels, N = 9 groups of C rows of the weight operand (one shifted view per tap), K = ld halo-order pixels
            IgemmDesc d;
            d.X = yT; d.Wt = xT - head * esz; d.M = N; d.N = 9 * C; d.Cin = ld; d.ldx = ld; d.out_f32 = T.S_dw;
            d.wt_grp_rows = C; d.wt_rp = rp; d.wt_base = (int)head;
            d.wt_odd = amp ? (int)(copy_bytes / esz) - 1 : 0;   // amp: taps with kx != 1 read copy 1 (x[k + 1] at k) so that every base stays 4-byte aligned
            TRY(gemm_wgrad(c, d, amp));
        }
        TRY(tr_wgrad_permute(T.S_dw, dW, N, C, c.st, c.err));
    }
    if (db) TRY(tr_colsum(dY, nullptr, db, T.S_col, M, N, 0, c.st, c.err));
    return 0;
}

// out = LN(y) g + b backward: d y -> dy; gamma / beta gradients
int ln_bwd(Ctx& c, const float* y, const float* g, const float* dout, float* dy, float* xhat, size_t M, int C, float* dg, float* dbeta, float eps) {
    TRY(tr_ln_bwd(y, g, dout, dy, xhat, (int)M, C, eps, c.st, c.err));
    if (dg && dbeta) return tr_colsum2(dout, xhat, dg, dbeta, c.T.S_col, M, C, c.st, c.err);   // one pass over dout for both (same addition order as the single forms)
    if (dg) TRY(tr_colsum(dout, xhat, dg, c.T.S_col, M, C, 0, c.st, c.err));
    if (dbeta) TRY(tr_colsum(dout, nullptr, dbeta, c.T.S_col, M, C, 0, c.st, c.err));
    return 0;
}

IgemmDesc conv_desc(const void* X, int Cin, const void* Wt, int N, int r, int B) {
    IgemmDesc d;
    d.X = X; d.Wt = Wt; d.M = B * r * r; d.N = N; d.Cin = Cin; d.taps = 9; d.H = r; d.W = r;
    return d;
}

bool any_grad(const Handle& h, const std::string& prefix) {
    for (const auto& w : h.weights)
        if (w.grad && w.key.compare(0, prefix.size(), prefix) == 0) return true;
    return false;
}

int check_train(Handle& h, int B, const void* ws, size_t ws_bytes, std::string& err) {
    if (h.cfg.precision != SOCCDPT_PREC_F32) { err = "soccdpt_train_*: the training step is built for SOCCDPT_PREC_F32 handles only"; return 1; }
    if (h.arch.hybrid && h.arch.grid() != 24) { err = "soccdpt_train_*: the ViT-hybrid encoder trains at 384 x 384 (position embedding used as stored)"; return 1; }
    if (h.cfg.features != 256 || h.cfg.num_classes != 3) { err = "soccdpt_train_*: features must be 256 and num_classes 3"; return 1; }
    if (B < 1 || !ws) { err = "soccdpt_train_*: bad arguments"; return 1; }
    for (const auto& w : h.weights)
        if (!w.ptr) { err = "soccdpt_train_*: weight not bound: " + w.key; return 1; }
    if (ws_bytes < train_workspace_bytes(h, B)) { err = "soccdpt_train_*: workspace too small"; return 1; }
    return 0;
}

}  // namespace trn
using namespace trn;

// Named tape tensors for tests / debugging: f32, plain [rows][C] unless noted.
int train_workspace_tensor(Handle& h, int B, const char* name, size_t* byte_offset, size_t* elems) {
    if (B < 1 || !name) return 1;
    char base[256];
    TArena ar(base);   // a non-null base: offsets come out as pointer differences
    Tape T;
    carve(h, B, ar, T);
    const Arch& a = h.arch;
    const int F = h.cfg.features, r1 = 2 * a.fres(0), r0 = 4 * a.fres(0);
    const size_t M1 = (size_t)B * r1 * r1, M0p = (size_t)B * r0 * r0;
    const float* p = nullptr;
    size_t n = 0;
    const std::string k = name;
    if (k == "seg_conv") { p = T.c_raw; n = M1 * F; }                 // seg_head.0 output (pre-BatchNorm)
    else if (k == "seg_act") { p = T.r; n = M1 * F; }                 // after BatchNorm + ReLU + Dropout
    else if (k == "seg_logits") { p = T.logits; n = M1 * 3; }
    else if (k == "depth_conv2") { p = T.e; n = M0p * 32; }           // output_conv.2 output (pre-ReLU)
    else if (k == "depth_conv0") { p = T.d1; n = M1 * (size_t)(F / 2); }
    else if (k == "d_path1") { p = T.GP; n = M1 * F; }                // gradient w.r.t. path_1 (after the backward)
    else if (k.compare(0, 10, "drop_path.") == 0 && !a.hybrid) {      // "drop_path.<stage>.<block>": the [2][B] DropPath scales of one Swin block
        int s2 = -1, j2 = -1;
        if (sscanf(k.c_str() + 10, "%d.%d", &s2, &j2) == 2 && s2 >= 0 && s2 < 4 && j2 >= 0 && j2 < a.depths[s2]) { p = T.blk[s2][j2].dp; n = 2 * (size_t)B; }
    }
    else {
        for (int l = 0; l < 4 && !p; ++l) {
            const size_t M = (size_t)B * a.fres(l) * a.fres(l);
            const std::string sl = std::to_string(l);
            if (k == "lrn_raw" + sl) { p = T.lrn_raw[l]; n = M * F; }
            else if (k == "fusion_out" + sl) { p = T.oc[l]; n = M * F; }
            else if (k == "rcu2_out" + sl) { p = T.u[l]; n = M * F; }
            else if (k == "fused_raw" + sl && l < 3) { p = T.out_raw[l]; n = M * F; }
            else if (k == "d_feat" + sl) { p = T.DF[l]; n = M * a.fdim(l); }
        }
    }
    if (!p) return 1;
    *byte_offset = (size_t)(reinterpret_cast<const char*>(p) - base);
    *elems = n;
    return 0;
}

size_t train_workspace_bytes(Handle& h, int B) {
    TArena ar(nullptr);
    Tape T;
    carve(h, B, ar, T);
    return ar.off + 256;
}

int train_forward(Handle& h, const float* x, int B, float* inv, float* seg, void* ws, size_t ws_bytes, float dropout_p, unsigned seed, hipStream_t st, std::string& err) {
    if (check_train(h, B, ws, ws_bytes, err)) return 1;
    const Arch& a = h.arch;
    TArena ar(ws);
    Tape T;
    carve(h, B, ar, T);
    T.dropout_p = dropout_p;
    Ctx c{h, T, B, st, err};
    const int F = h.cfg.features;
    {
        hipError_t e = hipMemsetAsync(static_cast<char*>(ws) + T.halo_lo, 0, T.halo_hi - T.halo_lo, st);
        if (e != hipSuccess) { err = std::string("train_forward memset: ") + hipGetErrorString(e); return 1; }
    }
    if (a.hybrid) { TRY(hy_forward(c, x)); } else {
    // ---------------- encoder ----------------
    const int G = a.grid(), C0 = a.embed;
    const size_t M0 = (size_t)B * G * G;
    TRY(tr_patch_im2col(x, T.patches, B, a.img, st, err));
    TRY(tr_pad_cols(c.W(ENC + "patch_embed.proj.weight"), T.pe_wpad, C0, 48, 64, st, err));
    {
        IgemmDesc d;
        d.X = T.patches; d.Wt = T.pe_wpad; d.M = (int)M0; d.N = C0; d.Cin = 64; d.ldx = 64; d.bias = c.W(ENC + "patch_embed.proj.bias"); d.out_f32 = T.pe_pre;
        TRY(gemm_fwd(c, d, M0 * 64, (size_t)C0 * 64));
        TRY(launch_ln_residual(T.pe_pre, c.W(ENC + "patch_embed.norm.weight"), c.W(ENC + "patch_embed.norm.bias"), T.x0, nullptr, nullptr, nullptr, 0, (int)M0, C0, 0, G, 0,
                               st, err));
    }
    const float* xcur = T.x0;
    int blk_index = 0;
    const int nblk_total = a.depths[0] + a.depths[1] + a.depths[2] + a.depths[3];
    for (int s = 0; s < 4; ++s) {
        const int C = a.dim(s), res = a.res(s), M = B * res * res, wsz = a.ws(s), H = a.heads[s];
        for (int j = 0; j < a.depths[s]; ++j) {
            BlkT& b = T.blk[s][j];
            const std::string k = blk_key(s, j);
            b.xin = xcur;
            // stochastic depth (timm DropPath on both residual branches; rate rising linearly over the blocks): per-sample scales for this block
            const float dp_p = nblk_total > 1 ? h.train_drop_path * (float)blk_index / (float)(nblk_total - 1) : 0.f;
            const bool dp_on = h.train_drop_path > 0.f;
            if (dp_on) {
                TRY(tr_drop_path_fill(b.dp, B, dp_p, seed, 2u * (unsigned)blk_index, st, err));
                TRY(tr_drop_path_fill(b.dp + B, B, dp_p, seed, 2u * (unsigned)blk_index + 1u, st, err));
            }
            ++blk_index;
            TRY(launch_qkv_bias(c.W(k + "attn.q_bias"), c.W(k + "attn.v_bias"), b.qkv_bias, C, st, err));
            TRY(launch_logit_scale(c.W(k + "attn.logit_scale"), b.scale, H, st, err));
            TRY(launch_cpb_table(c.W(k + "attn.cpb_mlp.0.weight"), c.W(k + "attn.cpb_mlp.0.bias"), c.W(k + "attn.cpb_mlp.2.weight"), b.table, wsz, a.pretrained_window[s], H,
                                 st, err));
            TRY(launch_attn_bias(b.table, b.bias_acc, wsz, H, st, err));
            IgemmDesc d;
            d.X = b.xin; d.Wt = c.W(k + "attn.qkv.weight"); d.M = M; d.N = 3 * C; d.Cin = C; d.ldx = C; d.bias = b.qkv_bias; d.out_f32 = b.qkv;
            TRY(gemm_fwd(c, d, (size_t)M * C, (size_t)3 * C * C));
            TRY(launch_window_attention_f32(b.qkv, b.bias_acc, b.table, b.scale, b.attn, B, res, wsz, a.shift(s, j), H, st, err));
            d = IgemmDesc();
            d.X = b.attn; d.Wt = c.W(k + "attn.proj.weight"); d.M = M; d.N = C; d.Cin = C; d.ldx = C; d.bias = c.W(k + "attn.proj.bias"); d.out_f32 = b.a_pre;
            TRY(gemm_fwd(c, d, (size_t)M * C, (size_t)C * C));
            TRY(copy_d2d(c, b.x1, b.xin, (size_t)M * C * 4, "train_forward copy"));
            TRY(launch_ln_residual(b.a_pre, c.W(k + "norm1.weight"), c.W(k + "norm1.bias"), b.x1, nullptr, nullptr, nullptr, 0, M, C, 1, res, 0, st, err,
                                   dp_on ? b.dp : nullptr, res * res));
            d = IgemmDesc();
            d.X = b.x1; d.Wt = c.W(k + "mlp.fc1.weight"); d.M = M; d.N = 4 * C; d.Cin = C; d.ldx = C; d.bias = c.W(k + "mlp.fc1.bias"); d.act = ACT_GELU;
            d.out_f32 = b.hpre; d.out_op = b.hact;
            TRY(gemm_fwd(c, d, (size_t)M * C, (size_t)4 * C * C));
            d = IgemmDesc();
            d.X = b.hact; d.Wt = c.W(k + "mlp.fc2.weight"); d.M = M; d.N = C; d.Cin = 4 * C; d.ldx = 4 * C; d.bias = c.W(k + "mlp.fc2.bias"); d.out_f32 = b.m_pre;
            TRY(gemm_fwd(c, d, (size_t)M * 4 * C, (size_t)4 * C * C));
            TRY(copy_d2d(c, b.xout, b.x1, (size_t)M * C * 4, "train_forward copy"));
            TRY(launch_ln_residual(b.m_pre, c.W(k + "norm2.weight"), c.W(k + "norm2.bias"), b.xout, nullptr, nullptr, j == a.hooks[s] ? T.feat[s] : nullptr, 0, M, C, 1, res, 0,
                                   st, err, dp_on ? b.dp + B : nullptr, res * res));
            xcur = b.xout;
        }
        if (s < 3) {
            const std::string dk = ENC + "layers." + std::to_string(s) + ".downsample.";
            TRY(launch_merge_gather(xcur, T.mg[s], B, res, C, 4, st, err));
            IgemmDesc d;
            d.X = T.mg[s]; d.Wt = c.W(dk + "reduction.weight"); d.M = M / 4; d.N = 2 * C; d.Cin = 4 * C; d.ldx = 4 * C; d.out_f32 = T.mr_pre[s];
            TRY(gemm_fwd(c, d, (size_t)M * C, (size_t)8 * C * C));
            TRY(launch_ln_residual(T.mr_pre[s], c.W(dk + "norm.weight"), c.W(dk + "norm.bias"), T.mx[s], nullptr, nullptr, nullptr, 0, M / 4, 2 * C, 0, res / 2, 0, st, err));
            xcur = T.mx[s];
        }
    }
    }
    // ---------------- decoder ----------------
    for (int l = 0; l < 4; ++l) {
        TRY(launch_conv_w(c.W(SCR + "layer" + std::to_string(l + 1) + "_rn.weight"), nullptr, T.w_lrn[l], 1, 0, F, a.fdim(l), st, err));
        const std::string rb = SCR + "refinenet" + std::to_string(l + 1) + ".";
        for (int u = 0; u < 2; ++u) {
            if (l == 3 && u == 0) continue;
            const std::string ub = rb + "resConfUnit" + std::to_string(u + 1) + ".";
            TRY(launch_conv_w(c.W(ub + "conv1.weight"), nullptr, T.w_rcu[l][u][0], 1, 0, F, F, st, err));
            TRY(launch_conv_w(c.W(ub + "conv2.weight"), nullptr, T.w_rcu[l][u][1], 1, 0, F, F, st, err));
        }
    }
    for (int l = 3; l >= 0; --l) {
        const int r = a.fres(l), M = B * r * r;
        const std::string rb = SCR + "refinenet" + std::to_string(l + 1) + ".";
        {
            IgemmDesc d = conv_desc(T.feat[l], a.fdim(l), T.w_lrn[l], F, r, B);
            d.out_f32 = T.lrn_raw[l]; d.out_op = T.lrn_relu[l]; d.out_halo = 1; d.act = ACT_RELU;
            TRY(gemm_fwd(c, d, Halo{r, r, a.fdim(l)}.elems(B), (size_t)F * 9 * a.fdim(l)));
        }
        const float* fused_raw = T.lrn_raw[l];
        const float* fused_relu = T.lrn_relu[l];
        if (l < 3) {
            const std::string ub = rb + "resConfUnit1.";
            IgemmDesc d = conv_desc(T.lrn_relu[l], F, T.w_rcu[l][0][0], F, r, B);
            d.bias = c.W(ub + "conv1.bias"); d.act = ACT_RELU; d.out_op = T.t1[l]; d.out_halo = 1;
            TRY(gemm_fwd(c, d, Halo{r, r, F}.elems(B), (size_t)F * 9 * F));
            d = conv_desc(T.t1[l], F, T.w_rcu[l][0][1], F, r, B);
            d.bias = c.W(ub + "conv2.bias"); d.res1 = T.lrn_raw[l];
            d.res2 = T.oc[l + 1]; d.res2_h = a.fres(l + 1); d.res2_w = a.fres(l + 1);
            d.out_f32 = T.out_raw[l]; d.out_op = T.out_relu[l]; d.out_halo = 1; d.act = ACT_RELU;
            TRY(gemm_fwd(c, d, Halo{r, r, F}.elems(B), (size_t)F * 9 * F));
            fused_raw = T.out_raw[l];
            fused_relu = T.out_relu[l];
        }
        {
            const std::string ub = rb + "resConfUnit2.";
            IgemmDesc d = conv_desc(fused_relu, F, T.w_rcu[l][1][0], F, r, B);
            d.bias = c.W(ub + "conv1.bias"); d.act = ACT_RELU; d.out_op = T.t2[l]; d.out_halo = 1;
            TRY(gemm_fwd(c, d, Halo{r, r, F}.elems(B), (size_t)F * 9 * F));
            d = conv_desc(T.t2[l], F, T.w_rcu[l][1][1], F, r, B);
            d.bias = c.W(ub + "conv2.bias"); d.res1 = fused_raw; d.out_f32 = T.u[l];
            TRY(gemm_fwd(c, d, Halo{r, r, F}.elems(B), (size_t)F * 9 * F));
        }
        {
            IgemmDesc d;
            d.X = T.u[l]; d.Wt = c.W(rb + "out_conv.weight"); d.M = M; d.N = F; d.Cin = F; d.ldx = F; d.bias = c.W(rb + "out_conv.bias"); d.out_f32 = T.oc[l];
            TRY(gemm_fwd(c, d, (size_t)M * F, (size_t)F * F));
        }
        if (l == 0) TRY(launch_bilinear(T.oc[0], 0, nullptr, nullptr, T.path1, 1, 0, B, r, r, 2 * r, 2 * r, F, st, err));
    }
    // ---------------- heads ----------------
    const int r1 = 2 * a.fres(0), r0 = 4 * a.fres(0);
    const size_t M1 = (size_t)B * r1 * r1, M0p = (size_t)B * r0 * r0;
    TRY(launch_conv_w(c.W(SCR + "output_conv.0.weight"), nullptr, T.w_d0, 1, 0, F / 2, F, st, err));
    TRY(launch_conv_w(c.W(SCR + "output_conv.2.weight"), nullptr, T.w_d2, 1, 0, 32, F / 2, st, err));
    TRY(launch_conv_w(c.W("seg_head.0.weight"), nullptr, T.w_s0, 1, 0, F, F, st, err));
    {
        IgemmDesc d = conv_desc(T.path1, F, T.w_d0, F / 2, r1, B);
        d.bias = c.W(SCR + "output_conv.0.bias"); d.out_f32 = T.d1;
        TRY(gemm_fwd(c, d, Halo{r1, r1, F}.elems(B), (size_t)(F / 2) * 9 * F));
        TRY(launch_bilinear(T.d1, 0, nullptr, nullptr, T.d1u, 1, 0, B, r1, r1, r0, r0, F / 2, st, err));
        d = conv_desc(T.d1u, F / 2, T.w_d2, 32, r0, B);
        d.bias = c.W(SCR + "output_conv.2.bias"); d.out_f32 = T.e;
        TRY(gemm_fwd(c, d, Halo{r0, r0, F / 2}.elems(B), (size_t)32 * 9 * (F / 2)));
        TRY(tr_depth_tail_fwd(T.e, c.W(SCR + "output_conv.4.weight"), c.W(SCR + "output_conv.4.bias"), T.inv, M0p, 32, st, err));
        TRY(copy_d2d(c, inv, T.inv, M0p * 4, "train_forward inv"));
    }
    {
        IgemmDesc d = conv_desc(T.path1, F, T.w_s0, F, r1, B);
        d.out_f32 = T.c_raw;
        TRY(gemm_fwd(c, d, Halo{r1, r1, F}.elems(B), (size_t)F * 9 * F));
        TRY(tr_bn_stats(T.c_raw, T.bn_stats, const_cast<float*>(c.W("seg_head.1.running_mean")), const_cast<float*>(c.W("seg_head.1.running_var")), T.S_col, F, M1, 1e-5f,
                        0.1f, st, err));
        TRY(tr_bn_relu_dropout_fwd(T.c_raw, T.bn_stats, c.W("seg_head.1.weight"), c.W("seg_head.1.bias"), T.r, T.keep, M1, F, dropout_p, seed, st, err));
        TRY(launch_seg_tail(T.r, 1, 0, c.W("seg_head.4.weight"), c.W("seg_head.4.bias"), T.logits, T.seg, B, r1, r1, h.cfg.sigmoid, st, err));
        TRY(copy_d2d(c, seg, T.seg, M0p * 3 * 4, "train_forward seg"));
    }
    return 0;
}

// Encoder part of the backward: consumes the gradients of the four hooked feature maps (T.DF[l], [pixels][channels]) left by the decoder pass.
static int encoder_backward(Ctx& c) {
    Handle& h = c.h;
    const Arch& a = h.arch;
    Tape& T = c.T;
    const int B = c.B;
    hipStream_t st = c.st;
    std::string& err = c.err;
    float** G = T.G;
    if (a.hybrid) return hy_backward(c);
    {
        const float* xcur = T.x0;
        for (int s = 0; s < 4; ++s) {
            for (auto& b : T.blk[s]) { b.xin = xcur; xcur = b.xout; }
            if (s < 3) xcur = T.mx[s];
        }
    }
    // ---------------- encoder, last stage -> first ----------------
    bool have = false;   // GX holds a gradient
    // trainable parameters at or before (stage s, block j) in forward order?  The walk ends below the earliest one.
    auto trains_upto = [&](int s, int j) {
        if (any_grad(h, ENC + "patch_embed.")) return true;
        for (int t = 0; t <= s; ++t) {
            if (t < s && any_grad(h, ENC + "layers." + std::to_string(t) + ".")) return true;
            if (t == s)
                for (int i = 0; i <= j; ++i)
                    if (any_grad(h, blk_key(s, i))) return true;
        }
        return false;
    };
    for (int s = 3; s >= 0; --s) {
        const int C = a.dim(s), res = a.res(s), wsz = a.ws(s), H = a.heads[s];
        const size_t M = (size_t)B * res * res;
        for (int j = a.depths[s] - 1; j >= 0; --j) {
            BlkT& b = T.blk[s][j];
            const std::string k = blk_key(s, j);
            if (j == a.hooks[s]) {
                if (have) TRY(tr_axpy(T.GX, T.DF[s], M * C, st, err));
                else TRY(copy_d2d(c, T.GX, T.DF[s], M * C * 4, "train_backward"));
                have = true;
            }
            if (!have) continue;   // blocks after the last hooked one do not reach the outputs
            if (!trains_upto(s, j)) return 0;
            // xout = x1 + dp2 * LN2(m_pre)      (dp: the forward's per-sample DropPath scales; the gradient entering the branch is scaled alike)
            const bool dp_on = h.train_drop_path > 0.f;
            const float* g_mlp = T.GX;
            if (dp_on) { TRY(tr_scale_rows(T.GX, G[4], b.dp + B, M, C, res * res, st, err)); g_mlp = G[4]; }
            TRY(ln_bwd(c, b.m_pre, c.W(k + "norm2.weight"), g_mlp, G[0], G[1], M, C, c.Gd(k + "norm2.weight"), c.Gd(k + "norm2.bias")));
            TRY(linear_bwd(c, G[0], b.hact, c.W(k + "mlp.fc2.weight"), M, C, 4 * C, G[2], nullptr, c.Gd(k + "mlp.fc2.weight"), c.Gd(k + "mlp.fc2.bias")));
            TRY(tr_gelu_bwd(G[2], b.hpre, G[2], M * 4 * C, st, err));
            TRY(linear_bwd(c, G[2], b.x1, c.W(k + "mlp.fc1.weight"), M, 4 * C, C, G[3], T.GX, c.Gd(k + "mlp.fc1.weight"), c.Gd(k + "mlp.fc1.bias")));   // G3 = d x1
            // x1 = xin + dp1 * LN1(a_pre)
            const float* g_att = G[3];
            if (dp_on) { TRY(tr_scale_rows(G[3], G[4], b.dp, M, C, res * res, st, err)); g_att = G[4]; }
            TRY(ln_bwd(c, b.a_pre, c.W(k + "norm1.weight"), g_att, G[0], G[1], M, C, c.Gd(k + "norm1.weight"), c.Gd(k + "norm1.bias")));
            TRY(linear_bwd(c, G[0], b.attn, c.W(k + "attn.proj.weight"), M, C, C, G[2], nullptr, c.Gd(k + "attn.proj.weight"), c.Gd(k + "attn.proj.bias")));
            // exact-f32 MFMA form (train_attn.hip); the VALU kernels of round 2 stay selectable for A/B and as the reference form
            static const bool attn_valu = getenv("SOCCDPT_ATTN_BWD_VALU") != nullptr;
            // amp modes: the four products of the attention backward on 16-bit MFMAs too (autocast semantics); SOCCDPT_ATTN_BWD_F32=1 keeps them exact (A/B)
            static const bool attn_f32 = getenv("SOCCDPT_ATTN_BWD_F32") != nullptr;
            const int attn_op = (!attn_f32 && (c.h.train_amp == 1 || c.h.train_amp == 2)) ? c.h.train_amp : 0;
            const int dslots = attn_valu ? 0 : tr_attention_bwd_mfma_slots(wsz);
            if (dslots) TRY(tr_attention_bwd_mfma(b.qkv, b.attn, G[2], b.table, b.scale, T.dS, T.rowstat, T.dscale_part, G[4], B, res, wsz, a.shift(s, j), H, st, err, attn_op));
            else TRY(tr_attention_bwd(b.qkv, b.attn, G[2], b.table, b.scale, T.dS, T.rowstat, T.dscale_part, T.attn_part, G[4], B, res, wsz, a.shift(s, j), H, st, err));
            {
                float* dls = c.Gd(k + "attn.logit_scale");
                float* dw0 = c.Gd(k + "attn.cpb_mlp.0.weight");
                float* db0 = c.Gd(k + "attn.cpb_mlp.0.bias");
                float* dw2 = c.Gd(k + "attn.cpb_mlp.2.weight");
                if (dls || dw0 || db0 || dw2)
                    TRY(tr_attn_param_grads(T.dS, T.dscale_part, b.table, c.W(k + "attn.logit_scale"), c.W(k + "attn.cpb_mlp.0.weight"), c.W(k + "attn.cpb_mlp.0.bias"),
                                            c.W(k + "attn.cpb_mlp.2.weight"), T.dtable, T.dt, T.S_cpb, dls, dw0, db0, dw2, B * (res / wsz) * (res / wsz), wsz,
                                            a.pretrained_window[s], H, st, err, dslots));
            }
            {
                float* dq = c.Gd(k + "attn.q_bias");
                float* dv = c.Gd(k + "attn.v_bias");
                float* dbias = (dq || dv) ? T.S_vec : nullptr;
                TRY(linear_bwd(c, G[4], b.xin, c.W(k + "attn.qkv.weight"), M, 3 * C, C, T.GX, G[3], c.Gd(k + "attn.qkv.weight"), dbias));
                if (dbias) TRY(tr_qv_bias_grad(dbias, dq, dv, C, st, err));
            }
        }
        if (!have) continue;
        {   // anything trainable in patch_embed or stages < s (their blocks and PatchMerging)?
            bool below = any_grad(h, ENC + "patch_embed.");
            for (int t = 0; t < s; ++t) below = below || any_grad(h, ENC + "layers." + std::to_string(t) + ".");
            if (!below) return 0;
        }
        if (s > 0) {
            // x_s = LN(reduction(gather(x_{s-1})))   (timm PatchMerging of Swin-V2: reduction then norm)
            const int Cp = a.dim(s - 1);
            const std::string dk = ENC + "layers." + std::to_string(s - 1) + ".downsample.";
            TRY(ln_bwd(c, T.mr_pre[s - 1], c.W(dk + "norm.weight"), T.GX, G[0], G[1], M, C, c.Gd(dk + "norm.weight"), c.Gd(dk + "norm.bias")));
            TRY(linear_bwd(c, G[0], T.mg[s - 1], c.W(dk + "reduction.weight"), M, C, 4 * Cp, G[2], nullptr, c.Gd(dk + "reduction.weight"), nullptr));
            TRY(tr_merge_scatter(G[2], T.GX, B, a.res(s - 1), Cp, st, err));
        } else {
            const int C0 = a.embed;
            TRY(ln_bwd(c, T.pe_pre, c.W(ENC + "patch_embed.norm.weight"), T.GX, G[0], G[1], M, C0, c.Gd(ENC + "patch_embed.norm.weight"), c.Gd(ENC + "patch_embed.norm.bias")));
            float* dw = c.Gd(ENC + "patch_embed.proj.weight");
            TRY(linear_bwd(c, G[0], T.patches, T.pe_wpad, M, C0, 64, nullptr, nullptr, dw ? T.S_dw + 65536 : nullptr, c.Gd(ENC + "patch_embed.proj.bias")));
            if (dw) TRY(tr_pad_cols(T.S_dw + 65536, dw, C0, 64, 48, st, err));
        }
    }
    return 0;
}

// Test entry: the encoder backward alone, from caller-supplied gradients of the hooked feature maps (d_feat[l]: [B * fres(l)^2][fdim(l)] f32).
int train_backward_encoder(Handle& h, int B, const float* const* d_feat, void* ws, size_t ws_bytes, hipStream_t st, std::string& err) {
    if (check_train(h, B, ws, ws_bytes, err)) return 1;
    TArena ar(ws);
    Tape T;
    carve(h, B, ar, T);
    Ctx c{h, T, B, st, err};
    c.arm_defer(T.tn_arena, kTrainTnArenaFloats);
    for (int l = 0; l < 4; ++l) {
        if (!d_feat || !d_feat[l]) { err = "soccdpt_train_backward_encoder: null feature gradient"; return 1; }
        const size_t n = (size_t)B * h.arch.fres(l) * h.arch.fres(l) * h.arch.fdim(l);
        TRY(copy_d2d(c, T.DF[l], d_feat[l], n * 4, "train_backward_encoder"));
    }
    TRY(stage_weights(c));
    TRY(encoder_backward(c));
    return tn_flush(c.tn, st, err);   // the deferred weight-gradient sums of this pass
}

int train_backward(Handle& h, const float* x, int B, const float* d_inv, const float* d_seg, void* ws, size_t ws_bytes, hipStream_t st, std::string& err) {
    if (check_train(h, B, ws, ws_bytes, err)) return 1;
    (void)x;
    const Arch& a = h.arch;
    TArena ar(ws);
    Tape T;
    carve(h, B, ar, T);
    T.dropout_p = h.train_key.dropout_p;
    Ctx c{h, T, B, st, err};
    c.arm_defer(T.tn_arena, kTrainTnArenaFloats);   // weight-gradient partials wait for ONE batched sum at the end of the pass (train.h TnDefer)
    TRY(stage_weights(c));
    auto pass = [&]() -> int {
    const int F = h.cfg.features;
    float** G = T.G;
    const int r1 = 2 * a.fres(0), r0 = 4 * a.fres(0);
    const size_t M1 = (size_t)B * r1 * r1, M0p = (size_t)B * r0 * r0;
    const bool enc_train = any_grad(h, HYB);   // "depth_net.pretrained.": the timm model and, for the hybrid, the act_postprocess read-outs
    // Frozen prefixes (freeze helpers of model/loss.py, PatchWiseInplace): the gradient stops flowing where nothing upstream is trainable
    bool lvl_own[4], lvl_side[4];   // level l: RCU2 / out_conv / anything coarser  |  RCU1 + layer_rn + encoder hook
    for (int l = 0; l < 4; ++l) {
        const std::string rb = SCR + "refinenet" + std::to_string(l + 1) + ".";
        lvl_own[l] = any_grad(h, rb + "out_conv") || any_grad(h, rb + "resConfUnit2");
        lvl_side[l] = enc_train || any_grad(h, rb + "resConfUnit1") || any_grad(h, SCR + "layer" + std::to_string(l + 1) + "_rn");
    }
    bool need_level[5];             // the gradient has to reach level l's out_conv output
    need_level[4] = false;
    for (int l = 3; l >= 0; --l) need_level[l] = lvl_own[l] || lvl_side[l] || need_level[l + 1];
    // ---------------- depth head ----------------
    {
        TRY(tr_depth_tail_bwd(d_inv, T.inv, T.e, c.W(SCR + "output_conv.4.weight"), G[0], G[1], M0p, 32, st, err));
        float* dw4 = c.Gd(SCR + "output_conv.4.weight");
        float* db4 = c.Gd(SCR + "output_conv.4.bias");
        if (dw4 || db4) {
            TRY(tr_colsum(G[1], nullptr, T.S_vec, T.S_col, M0p, 33, 0, st, err));
            if (dw4) TRY(copy_d2d(c, dw4, T.S_vec, 32 * 4, "train_backward"));
            if (db4) TRY(copy_d2d(c, db4, T.S_vec + 32, 4, "train_backward"));
        }
        TRY(conv3_bwd(c, G[0], T.d1u, c.W(SCR + "output_conv.2.weight"), r0, 32, F / 2, G[2], nullptr, c.Gd(SCR + "output_conv.2.weight"), c.Gd(SCR + "output_conv.2.bias")));
        TRY(tr_bilinear_bwd(G[2], G[3], B, r1, r1, r0, r0, F / 2, 0, st, err));
        TRY(conv3_bwd(c, G[3], T.path1, c.W(SCR + "output_conv.0.weight"), r1, F / 2, F, need_level[0] ? T.GP : nullptr, nullptr, c.Gd(SCR + "output_conv.0.weight"),
                      c.Gd(SCR + "output_conv.0.bias")));
    }
    // ---------------- seg head ----------------
    {
        TRY(tr_seg_act_bwd(d_seg, T.seg, G[0], B, 3, r0, h.cfg.sigmoid, st, err));
        TRY(tr_bilinear_bwd(G[0], G[1], B, r1, r1, r0, r0, 3, 0, st, err));   // d logits [M1][3]
        if (float* dw = c.Gd("seg_head.4.weight")) TRY(tr_smallk_wgrad(G[1], T.r, dw, T.S_col, M1, F, 3, st, err));
        if (float* db = c.Gd("seg_head.4.bias")) {
            // colsum needs N >= 1: three columns
            TRY(tr_colsum(G[1], nullptr, db, T.S_col, M1, 3, 0, st, err));
        }
        TRY(tr_smallk_dgrad(G[1], c.W("seg_head.4.weight"), G[2], M1, F, 3, st, err));
        TRY(tr_bn_relu_dropout_bwd_pre(G[2], T.r, T.keep, G[0], M1 * F, T.dropout_p, st, err));
        TRY(tr_bn_xhat(T.c_raw, T.bn_stats, G[3], M1, F, st, err));
        float* dbeta = T.S_vec;
        float* dgamma = T.S_vec + F;
        TRY(tr_colsum(G[0], nullptr, dbeta, T.S_col, M1, F, 0, st, err));
        TRY(tr_colsum(G[0], G[3], dgamma, T.S_col, M1, F, 0, st, err));
        if (float* p = c.Gd("seg_head.1.bias")) TRY(copy_d2d(c, p, dbeta, F * 4, "train_backward"));
        if (float* p = c.Gd("seg_head.1.weight")) TRY(copy_d2d(c, p, dgamma, F * 4, "train_backward"));
        TRY(tr_bn_bwd(G[0], T.c_raw, T.bn_stats, c.W("seg_head.1.weight"), dbeta, dgamma, G[2], M1, F, st, err));
        // path_1's im2col^T is still in S_T2 from output_conv.0's weight gradient (same input image, nothing in between writes S_T2)
        const bool xt_ready = c.Gd(SCR + "output_conv.0.weight") != nullptr;
        TRY(conv3_bwd(c, G[2], T.path1, c.W("seg_head.0.weight"), r1, F, F, need_level[0] ? T.GP : nullptr, T.GP, c.Gd("seg_head.0.weight"), nullptr, xt_ready));
    }
    if (!need_level[0]) return 0;
    TRY(tr_bilinear_bwd(T.GP, T.DOC, B, a.fres(0), a.fres(0), r1, r1, F, 0, st, err));
    // ---------------- decoder, fine -> coarse ----------------
    for (int l = 0; l < 4; ++l) {
        const int r = a.fres(l);
        const size_t M = (size_t)B * r * r;
        const std::string rb = SCR + "refinenet" + std::to_string(l + 1) + ".";
        TRY(linear_bwd(c, T.DOC, T.u[l], c.W(rb + "out_conv.weight"), M, F, F, G[0], nullptr, c.Gd(rb + "out_conv.weight"), c.Gd(rb + "out_conv.bias")));
        const float* fused_raw = l < 3 ? T.out_raw[l] : T.lrn_raw[l];
        const float* fused_relu = l < 3 ? T.out_relu[l] : T.lrn_relu[l];
        {
            const std::string ub = rb + "resConfUnit2.";
            TRY(conv3_bwd(c, G[0], T.t2[l], c.W(ub + "conv2.weight"), r, F, F, G[1], nullptr, c.Gd(ub + "conv2.weight"), c.Gd(ub + "conv2.bias")));
            TRY(tr_relu_bwd_halo(G[1], T.t2[l], nullptr, G[1], B, r, r, F, st, err));
            TRY(conv3_bwd(c, G[1], fused_relu, c.W(ub + "conv1.weight"), r, F, F, G[2], nullptr, c.Gd(ub + "conv1.weight"), c.Gd(ub + "conv1.bias")));
            TRY(tr_relu_bwd(G[2], fused_raw, G[0], G[3], M * F, st, err));   // d fused_raw
        }
        const float* d_lrn = G[3];
        if (l < 3 && need_level[l + 1])
            TRY(tr_bilinear_bwd(G[3], T.DOC, B, a.fres(l + 1), a.fres(l + 1), r, r, F, 0, st, err));   // gradient of the coarser level's out_conv output
        if (!lvl_side[l]) {
            if (!need_level[l + 1]) return 0;
            continue;
        }
        if (l < 3) {
            const std::string ub = rb + "resConfUnit1.";
            TRY(conv3_bwd(c, G[3], T.t1[l], c.W(ub + "conv2.weight"), r, F, F, G[1], nullptr, c.Gd(ub + "conv2.weight"), c.Gd(ub + "conv2.bias")));
            TRY(tr_relu_bwd_halo(G[1], T.t1[l], nullptr, G[1], B, r, r, F, st, err));
            TRY(conv3_bwd(c, G[1], T.lrn_relu[l], c.W(ub + "conv1.weight"), r, F, F, G[2], nullptr, c.Gd(ub + "conv1.weight"), c.Gd(ub + "conv1.bias")));
            TRY(tr_relu_bwd(G[2], T.lrn_raw[l], G[3], G[0], M * F, st, err));
            d_lrn = G[0];
        }
        const std::string lk = SCR + "layer" + std::to_string(l + 1) + "_rn.weight";
        TRY(conv3_bwd(c, d_lrn, T.feat[l], c.W(lk), r, F, a.fdim(l), enc_train ? T.DF[l] : nullptr, nullptr, c.Gd(lk), nullptr));
    }
    if (!enc_train) return 0;
    return encoder_backward(c);
    };
    TRY(pass());
    return tn_flush(c.tn, st, err);
}

}  // namespace soccdpt
