// Swin-V2 window attention with the qkv projection inside the kernel (round 6): one workgroup per (batch, window, head) computes
// its own q, k, v from the block input and the head's slice of Wqkv on the matrix cores, then runs the same key loop as attention.hip.
//
// Replaces, for one Swin block, `qkv = x @ Wqkv^T + cat(q_bias, 0, v_bias)` + WindowAttention.forward (timm SwinTransformerV2Block._attn, call site
// /root/reference/SOccDPT/model/backbones/swin2.py:25-27; maths HF modeling_swinv2.py:389-452) = the qkv igemm launch AND the window_attention launch of
// rounds 1-5: the [M][3C] q / k / v tensor (written once, read once) never exists, and one of the block's dependent launches disappears.
//
// Decomposition (CDNA4):
//  * GEMM phase.  The window's tokens are cut into blocks of 32; a wave owns one block and ALL 96 output columns of the head (q, k, v: 32 each), so a token's
//    input row is read by exactly one wave: it comes straight from global memory (L2) into registers -- 32 contiguous bytes per lane and 32-column step,
//    the cyclic shift / window partition folded into the row index as in attention.hip -- and never touches LDS.  The 96 x C weight slice is shared by all
//    waves: it goes through LDS in K-panels of 96 / 192 / 384 columns -- the WHOLE slice as one panel where it fits (C <= 192, and C = 384 with plain
//    16-bit weights): global -> registers -> LDS once, one memory latency; otherwise two or four panels, the next one's weights and input columns in
//    flight under the current panel's MFMAs (rows padded by 16 bytes: conflict-free ds_read_b128).  q and k use the "swapped" product (A = W rows d, B = x columns = tokens): a lane then owns one
//    token column with 16 of its 32 dims, so the L2 norm of cosine attention is lane-local + one cross-half add, and the normalised rows go into the
//    swizzled Q-hat / K-hat images as 8-byte stores.  v uses the plain product (A = x rows = tokens, B = W columns d): a lane owns dim d with 16 tokens,
//    four consecutive ones per register group -> V^T rows as 8-byte stores.  The x fragment registers serve both products (the A and B operands of
//    v_mfma_f32_32x32x16 have the same per-lane layout).
//  * x2w groups (half16.h: fp16 activations, weights as x3 pairs): the panel holds the hi and the lo image of the slice, every product is two MFMAs into
//    two accumulators, result = hi + lo * 2^-11 -- what igemm's x2w tiles compute.
//  * The q, k, v values are normalised / stored from the f32 accumulators (the unfused chain rounds them to fp16 in between: one rounding less here).
//  * Attention phase: window_attention_flash_core (attention_body.h), unchanged.
#include <stdlib.h>

#include "attention_body.h"

namespace soccdpt {

namespace {

template <int WS, int PK, bool X2W>
struct QkvCfg {
    using A = AttnGenCfg<WS>;
    static constexpr int THREADS = WS == 8 ? 128 : A::THREADS;   // 8 x 8 windows: two token blocks, two waves
    static constexpr int WAVES = THREADS / 64;
    static_assert(A::NT == WAVES, "one block of 32 tokens per wave (16 x 16 and 8 x 8 windows)");
    static constexpr int ROWB = PK * 2 + 16;                       // bytes per weight row of a panel image (padded: conflict-free 16-byte reads)
    static constexpr int IMG = 96 * ROWB;                          // one image (hi or lo) of one panel
    static constexpr int PANEL = IMG * (X2W ? 2 : 1);
    static constexpr int W_OFF = (A::LDS + 15) / 16 * 16;
    static constexpr int LDS = W_OFF + PANEL;                      // Q-hat, K-hat, V^T + one weight panel
};

template <int WS, bool F16, int PK, bool X2W, bool MULTI>
__global__ __launch_bounds__((QkvCfg<WS, PK, X2W>::THREADS)) void window_attention_qkv_kernel(const bf16_t* __restrict__ x, const void* __restrict__ wqkv,
                                                                                           const float* __restrict__ qkv_bias, const float* __restrict__ bias_acc,
                                                                                           const float* __restrict__ scale, bf16_t* __restrict__ out, int res, int shift,
                                                                                           int heads, int out_x3, unsigned long long* __restrict__ stamps) {
    using Q = QkvCfg<WS, PK, X2W>;
    using A = AttnGenCfg<WS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Ks = smem + A::KS_OFF;
    char* Vt = smem + A::VT_OFF;
    char* Wl = smem + Q::W_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int C = heads * 32;
    const int nw = res / WS;
    int bid = blockIdx.x;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const float hscale = scale[head];
    auto token_row = [&](int p) -> size_t {
        const int r = p / WS, c = p % WS;
        int sy = wy * WS + r + shift, sx = wx * WS + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int npanels = MULTI ? C / PK : 1;   // MULTI == false: the whole slice is one panel (C == PK)
    constexpr int CH = PK / 8;            // 16-byte chunks (8 columns) per weight row of a panel
    constexpr int WLD = (96 * CH + Q::THREADS - 1) / Q::THREADS;   // chunks per thread and panel
    constexpr int XC = PK / 32;           // 32-column steps per panel: two 16-byte fragments per lane each

    // diagnostics (tools/wattn_qkv_stamps.py; nullptr in the forward): 100 MHz s_memrealtime at entry / first panel staged / GEMM done / q, k, v in LDS / exit
    if (stamps && tid == 0) stamps[5 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memrealtime();

    // ---- weight panel: global -> registers -> LDS.  Row n of the slice: q rows head*32.., k rows C + head*32.., v rows 2C + head*32.. of Wqkv [3C][C] ----
    uint4 wreg[WLD][X2W ? 2 : 1];
    auto wload = [&](int pn) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int idx = i * Q::THREADS + tid;
            if (96 * CH % Q::THREADS == 0 || idx < 96 * CH) {
                const int n = idx / CH, ch = idx % CH;
                const size_t row = (size_t)(n >> 5) * C + head * 32 + (n & 31);
                const size_t e = row * C + (size_t)pn * PK + ch * 8;      // first of 8 consecutive columns
                if constexpr (X2W) {   // x3 unit = 32 bytes: hi chunk first in even units, second in odd ones (half16.h)
                    const char* u = static_cast<const char*>(wqkv) + (e >> 3) * 32;
                    const int odd = (int)((e >> 3) & 1);
                    wreg[i][0] = *reinterpret_cast<const uint4*>(u + (odd ? 16 : 0));
                    wreg[i][1] = *reinterpret_cast<const uint4*>(u + (odd ? 0 : 16));
                } else {
                    wreg[i][0] = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(wqkv) + e);
                }
            }
        }
    };
    auto wstore = [&]() {
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int idx = i * Q::THREADS + tid;
            if (96 * CH % Q::THREADS == 0 || idx < 96 * CH) {
                const int n = idx / CH, ch = idx % CH;
                *reinterpret_cast<uint4*>(Wl + n * Q::ROWB + ch * 16) = wreg[i][0];
                if constexpr (X2W) *reinterpret_cast<uint4*>(Wl + Q::IMG + n * Q::ROWB + ch * 16) = wreg[i][1];
            }
        }
    };

    // ---- the wave's block of 32 tokens: row pointer (cyclic shift + window partition in the index); lane half h reads columns 32 c + 16 h .. + 15 of every step ----
    const int tb = wave;
    const int p = tb * 32 + r32;
    const bf16_t* xrow = x + token_row(p) * (size_t)C + 16 * h;
    // input columns of the current and the next panel: two register sets used alternately (a copy at the end of a panel made the compiler interleave the
    // moves with the MFMAs, each waiting for the NEXT panel's loads: no overlap at all -- tools/wattn_qkv_stamps.py showed 3.5 us per panel for 0.65 us of MFMAs)
    uint4 xa[XC][2], xb[XC][2];
    auto xload = [&](uint4 (&dst)[XC][2], int pn) {
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            dst[c][0] = *reinterpret_cast<const uint4*>(xrow + (size_t)pn * PK + 32 * c);
            dst[c][1] = *reinterpret_cast<const uint4*>(xrow + (size_t)pn * PK + 32 * c + 8);
        }
    };
    f32x16 aq, ak, av, lq, lk, lv;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) { aq[rg] = 0.f; ak[rg] = 0.f; av[rg] = 0.f; lq[rg] = 0.f; lk[rg] = 0.f; lv[rg] = 0.f; }

    // one memory latency for the first panel: the input columns are requested BEFORE the weights, so the wait for the weights (the store below) covers them
    // too, and no load is outstanding when the panel loop is entered (a static s_waitcnt inside the MFMA section would otherwise have to assume the worst)
    xload(xa, 0);
    wload(0);
    wstore();
    __syncthreads();
    if (stamps && tid == 0) stamps[5 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();

    auto panel = [&](int pn, const uint4 (&xr)[XC][2], uint4 (&xn)[XC][2]) {
        const bool more = MULTI && pn + 1 < npanels;
        if (more) { wload(pn + 1); xload(xn, pn + 1); }   // the next panel's weights and input columns are in flight under this panel's MFMAs
        const char* wb = Wl + r32 * Q::ROWB + 32 * h;     // row r32 of a 32-row group; the lane half's 32 bytes of every 64-byte step
#pragma unroll
        for (int c = 0; c < XC; ++c) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h16x8 xf = __builtin_bit_cast(h16x8, xr[c][s]);
                const int off = 64 * c + 16 * s;
                const h16x8 wq = *reinterpret_cast<const h16x8*>(wb + off);
                const h16x8 wk = *reinterpret_cast<const h16x8*>(wb + 32 * Q::ROWB + off);
                const h16x8 wv = *reinterpret_cast<const h16x8*>(wb + 64 * Q::ROWB + off);
                aq = mfma_32x32x16<F16>(wq, xf, aq);
                ak = mfma_32x32x16<F16>(wk, xf, ak);
                av = mfma_32x32x16<F16>(xf, wv, av);
                if constexpr (X2W) {
                    const h16x8 wql = *reinterpret_cast<const h16x8*>(wb + Q::IMG + off);
                    const h16x8 wkl = *reinterpret_cast<const h16x8*>(wb + Q::IMG + 32 * Q::ROWB + off);
                    const h16x8 wvl = *reinterpret_cast<const h16x8*>(wb + Q::IMG + 64 * Q::ROWB + off);
                    lq = mfma_32x32x16<F16>(wql, xf, lq);
                    lk = mfma_32x32x16<F16>(wkl, xf, lk);
                    lv = mfma_32x32x16<F16>(xf, wvl, lv);
                }
            }
        }
        if (more) {
            __syncthreads();   // everybody has read this panel
            wstore();
            __syncthreads();
        }
    };
    if constexpr (MULTI) {
#pragma unroll 1
        for (int pn = 0; pn < npanels; pn += 2) {
            panel(pn, xa, xb);
            if (pn + 1 < npanels) panel(pn + 1, xb, xa);
        }
    } else {
        panel(0, xa, xb);
    }
    if (stamps && tid == 0) stamps[5 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    if constexpr (X2W) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            aq[rg] += lq[rg] * (1.0f / 2048.f);
            ak[rg] += lk[rg] * (1.0f / 2048.f);
            av[rg] += lv[rg] * (1.0f / 2048.f);
        }
    }
    // ---- q, k: lane = token column r32, registers = dims d = (rg & 3) + 8 (rg >> 2) + 4 h.  Bias, L2 norm over the 32 dims, logit scale into q-hat ----
    {
        const float* bq = qkv_bias + head * 32 + 4 * h;
        float qs = 0.f, ks = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b0 = *reinterpret_cast<const float4*>(bq + 8 * g), b1 = *reinterpret_cast<const float4*>(bq + C + 8 * g);
            aq[4 * g] += b0.x; aq[4 * g + 1] += b0.y; aq[4 * g + 2] += b0.z; aq[4 * g + 3] += b0.w;
            ak[4 * g] += b1.x; ak[4 * g + 1] += b1.y; ak[4 * g + 2] += b1.z; ak[4 * g + 3] += b1.w;
        }
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) { qs += aq[rg] * aq[rg]; ks += ak[rg] * ak[rg]; }
        qs += __shfl_xor(qs, 32);
        ks += __shfl_xor(ks, 32);
        const float qi = hscale / fmaxf(sqrtf(qs), 1e-12f);  // F.normalize eps
        const float ki = 1.0f / fmaxf(sqrtf(ks), 1e-12f);
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // chunk g = dims 8 g .. 8 g + 7 (16 bytes, swizzled by the row); this lane's four dims sit at byte 8 h of it
            const int sw = (g ^ ((p >> 2) & 3)) * 16 + 8 * h;
            uint2 qo, ko;
            qo.x = pack_h2<F16>(aq[4 * g] * qi, aq[4 * g + 1] * qi); qo.y = pack_h2<F16>(aq[4 * g + 2] * qi, aq[4 * g + 3] * qi);
            ko.x = pack_h2<F16>(ak[4 * g] * ki, ak[4 * g + 1] * ki); ko.y = pack_h2<F16>(ak[4 * g + 2] * ki, ak[4 * g + 3] * ki);
            *reinterpret_cast<uint2*>(Qs + p * 64 + sw) = qo;
            *reinterpret_cast<uint2*>(Ks + p * 64 + sw) = ko;
        }
        // ---- v: lane = dim r32, registers = tokens tb * 32 + (rg & 3) + 8 (rg >> 2) + 4 h -> V^T row r32, four consecutive tokens per store ----
        const float bv = qkv_bias[2 * C + head * 32 + r32];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int t0 = tb * 32 + 8 * g + 4 * h;
            uint2 vo;
            vo.x = pack_h2<F16>(av[4 * g] + bv, av[4 * g + 1] + bv);
            vo.y = pack_h2<F16>(av[4 * g + 2] + bv, av[4 * g + 3] + bv);
            *reinterpret_cast<uint2*>(Vt + r32 * A::VT_STRIDE + t0 * 2) = vo;
        }
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[5 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    window_attention_flash_core<WS, F16>(Qs, Ks, Vt, bias_acc, out, res, shift, heads, out_x3, head, b, wy, wx, 0, A::NT, wave, lane, Q::WAVES);
    if (stamps && tid == 0) stamps[5 * (size_t)blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();   // wave 0's own end (the waves finish within one key tile of each other)
}

template <int WS, bool F16, int PK, bool X2W, bool MULTI>
int launch_one(const bf16_t* x, const void* wqkv, const float* qkv_bias, const float* bias_acc, const float* scale, bf16_t* out, unsigned blocks, int res, int shift,
               int heads, int out_x3, hipStream_t st, std::string& err, unsigned long long* stamps) {
    using Q = QkvCfg<WS, PK, X2W>;
    static_assert(Q::LDS <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr;
    if (attr.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attention_qkv_kernel<WS, F16, PK, X2W, MULTI>), hipFuncAttributeMaxDynamicSharedMemorySize, Q::LDS) != hipSuccess) {
            err = "window_attention_qkv: hipFuncSetAttribute failed";
            return 1;
        }
        attr.done();
    }
    SOCCDPT_LAUNCH((window_attention_qkv_kernel<WS, F16, PK, X2W, MULTI>), dim3(blocks), dim3(Q::THREADS), Q::LDS, st, x, wqkv, qkv_bias, bias_acc, scale, out, res, shift, heads, out_x3, stamps);
    return check_launch("window_attention_qkv", err);
}

// panel width: the whole slice where it fits beside the attention operands (one memory latency), otherwise halves / quarters
int panel_of(int C, int x2w) {
    if (C == 96) return 96;
    if (C == 192) return 192;
    if (C == 384) return x2w ? 192 : 384;
    if (C == 768) return x2w ? 192 : 384;
    return 0;
}

}  // namespace

bool window_attention_qkv_supported(int ws, int C, int x2w) {
    if (ws != 16 && ws != 8) return false;
    return panel_of(C, x2w) != 0;   // the widths of swinv2_tiny_window16_256: 96, 192, 384, 768
}

int launch_window_attention_qkv(const bf16_t* x, const void* wqkv, const float* qkv_bias, const float* bias_acc, const float* scale, bf16_t* out, int hf, int x2w,
                                int B, int res, int ws, int shift, int heads, hipStream_t st, std::string& err, int out_x3, unsigned long long* stamps) {
    const int C = heads * 32;
    if (res % ws != 0) { err = "window_attention_qkv: res % ws != 0"; return 1; }
    if (!window_attention_qkv_supported(ws, C, x2w)) { err = "window_attention_qkv: window size / width not instantiated"; return 1; }
    if ((out_x3 || x2w) && !hf) { err = "window_attention_qkv: the x3 output and the x2w weights belong to the fp16 kernels"; return 1; }
    if (ws == 8 && shift != 0) { err = "window_attention_qkv: shifted 8x8 windows are not instantiated"; return 1; }
    const int nw = res / ws;
    const unsigned blocks = (unsigned)(B * nw * nw * heads);
    const int pk = panel_of(C, x2w);
#define QKV_GO(WS_, F_, PK_, X_) if (C != PK_) return launch_one<WS_, F_, PK_, X_, true>(x, wqkv, qkv_bias, bias_acc, scale, out, blocks, res, shift, heads, out_x3, st, err, stamps); else return launch_one<WS_, F_, PK_, X_, false>(x, wqkv, qkv_bias, bias_acc, scale, out, blocks, res, shift, heads, out_x3, st, err, stamps)
#define QKV_PK(WS_, F_, X_) do { if (pk == 96) { QKV_GO(WS_, F_, 96, X_); } else if (pk == 192) { QKV_GO(WS_, F_, 192, X_); } else { if constexpr (!X_) { QKV_GO(WS_, F_, 384, false); } } } while (0)
    if (ws == 16) {
        if (!hf) QKV_PK(16, false, false);
        else if (x2w) QKV_PK(16, true, true);
        else QKV_PK(16, true, false);
    } else {
        if (!hf) QKV_PK(8, false, false);
        else if (x2w) QKV_PK(8, true, true);
        else QKV_PK(8, true, false);
    }
#undef QKV_PK
#undef QKV_GO
    err = "window_attention_qkv: no instantiation";
    return 1;
}

}  // namespace soccdpt
