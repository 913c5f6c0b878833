// Non-GEMM kernels of the ViT-hybrid encoder (dpt_hybrid_384, BASELINE configs[2]) for gfx950.
//
// They stand in for the timm 0.6.12 pieces that /root/reference/SOccDPT/model/backbones/vit.py:244-258 creates and
// forward_flex (vit.py:44-85) runs, restated in oracle/soccdpt_ref.py (resnetv2_backbone, vit_block, hybrid_encoder):
//   * StdConv2dSame weight standardisation (folded once per weight load: soccdpt_prepare),
//   * the 7x7 / stride-2 stem convolution as im2col + igemm (Cin = 3 is no MFMA shape),
//   * GroupNormAct(32 groups, eps 1e-5) (+ ReLU, + shortcut, + ReLU), with the statistics produced by the convolution's own
//     epilogue (igemm.hip, ST instantiation), and the stem's GroupNorm + ReLU + MaxPool2dSame(3, 2) fused,
//   * class token + position embedding + the first pre-norm LayerNorm, and the pre-norm LayerNorm (eps 1e-6) of every ViT block.
// All of them are HBM-bound elementwise / row kernels: 16-byte accesses, NHWC, no LDS staging needed except the row reductions.
// OUT selects the operand format written for the next GEMM: 0 = bf16, 1 = fp16, 2 = f32, 3 = x3 split fp16 (half16.h; SOCCDPT_PREC_F16X3).
#include <stdio.h>
#include <stdlib.h>
#include "half16.h"
#include "kernels.h"

namespace soccdpt {
namespace {

template <int OUT>
struct OutT { typedef uint16_t type; };
template <>
struct OutT<2> { typedef float type; };
template <>
struct OutT<3> { typedef float type; };   // x3: 4 bytes per element

// store 4 consecutive values as the operand type
template <int OUT>
__device__ __forceinline__ void store4(void* base, size_t idx, float a, float b, float c, float d) {
    if constexpr (OUT == 2) {
        *reinterpret_cast<float4*>(static_cast<float*>(base) + idx) = make_float4(a, b, c, d);
    } else if constexpr (OUT == 3) {
        x3_store4(base, idx, a, b, c, d);
    } else {
        uint2 p;
        p.x = pack_h2<OUT == 1>(a, b);
        p.y = pack_h2<OUT == 1>(c, d);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(base) + idx) = p;
    }
}

// ---- weight standardisation: w [Cout][Cin][k][k] f32 -> out [Cout][Kpad] tap-major ((ky*k + kx)*Cin + ci), zero padded ----
// (w - mean) / sqrt(var + eps) per output channel, biased variance over Cin*k*k: timm StdConv2dSame.forward evaluates exactly this
// through F.batch_norm(training=True) on the [1, Cout, fan_in] view.
template <int OUT>
__global__ __launch_bounds__(256) void ws_conv_w_kernel(const float* __restrict__ w, void* __restrict__ out, int Cin, int kk, int Kpad, float eps) {
    __shared__ double red[2][256];
    const int co = blockIdx.x, tid = threadIdx.x;
    const int fan = Cin * kk;
    const float* src = w + (size_t)co * fan;
    double a = 0.0, q = 0.0;
    for (int i = tid; i < fan; i += 256) { const double v = src[i]; a += v; q += v * v; }
    red[0][tid] = a; red[1][tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    const double mean = red[0][0] / fan;
    double var = red[1][0] / fan - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
    typename OutT<OUT>::type* dst = static_cast<typename OutT<OUT>::type*>(out) + (size_t)co * Kpad;
    for (int i = tid; i < Kpad; i += 256) {
        float v = 0.f;
        if (i < fan) {
            const int tap = i / Cin, ci = i - tap * Cin;
            v = (src[(size_t)ci * kk + tap] - fm) * fr;
        }
        if constexpr (OUT == 2) dst[i] = v;
        else if constexpr (OUT == 3) x3_store1(out, (size_t)co * Kpad + i, v);
        else dst[i] = f2h<OUT == 1>(v);
    }
}

// ---- stem im2col: x [B][3][S][S] f32 NCHW -> A [B*Ho*Wo][160], k = (ky*7 + kx)*3 + c, zero beyond 147 ----
// Conv2d(3, 64, 7, stride 2) with TF 'SAME' padding: total pad 5 at S = 384 -> 2 left / top, 3 right / bottom.
template <int OUT>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, void* __restrict__ A, int B, int S, int Ho, int pad_lo) {
    const size_t total = (size_t)B * Ho * Ho * 20;   // 20 chunks of 8 k per row
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ch = (int)(gid % 20);
    const size_t m = gid / 20;
    const int ox = (int)(m % Ho), oy = (int)((m / Ho) % Ho), b = (int)(m / ((size_t)Ho * Ho));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ch * 8 + j;
        const int tap = k / 3, c = k - tap * 3;
        const int ky = tap / 7, kx = tap - ky * 7;
        const int iy = oy * 2 + ky - pad_lo, ix = ox * 2 + kx - pad_lo;
        const bool ok = k < 147 && iy >= 0 && iy < S && ix >= 0 && ix < S;
        const int cy = iy < 0 ? 0 : (iy < S ? iy : S - 1), cx = ix < 0 ? 0 : (ix < S ? ix : S - 1), cc = k < 147 ? c : 0;   // clamped address, select after the load
        const float xv = x[(((size_t)b * 3 + cc) * S + cy) * S + cx];
        v[j] = ok ? xv : 0.f;
    }
    store4<OUT>(A, m * 160 + ch * 8, v[0], v[1], v[2], v[3]);
    store4<OUT>(A, m * 160 + ch * 8 + 4, v[4], v[5], v[6], v[7]);
}

// GroupNorm of 4 consecutive channels c..c+3 of sample b: ((x - mean) * rstd) * gamma + beta, the order torch.group_norm uses
struct Gn4 {
    float mean, rstd;
    float4 g, be;
    __device__ __forceinline__ void load(const float* stats, const float* gamma, const float* beta, int b, int c, int C, int cpg) {
        const int G = C / cpg;
        const float2 s = *reinterpret_cast<const float2*>(stats + ((size_t)b * G + c / cpg) * 2);
        mean = s.x; rstd = s.y;
        g = *reinterpret_cast<const float4*>(gamma + c);
        be = *reinterpret_cast<const float4*>(beta + c);
    }
    __device__ __forceinline__ float4 apply(float4 v) const {
        return make_float4((v.x - mean) * rstd * g.x + be.x, (v.y - mean) * rstd * g.y + be.y, (v.z - mean) * rstd * g.z + be.z,
                           (v.w - mean) * rstd * g.w + be.w);
    }
};
// cpg == 2: a float4 spans two groups
struct Gn4x2 {
    float2 s0, s1;
    float4 g, be;
    __device__ __forceinline__ void load(const float* stats, const float* gamma, const float* beta, int b, int c, int C) {
        const int G = C / 2;
        const float4 s = *reinterpret_cast<const float4*>(stats + ((size_t)b * G + c / 2) * 2);
        s0 = make_float2(s.x, s.y); s1 = make_float2(s.z, s.w);
        g = *reinterpret_cast<const float4*>(gamma + c);
        be = *reinterpret_cast<const float4*>(beta + c);
    }
    __device__ __forceinline__ float4 apply(float4 v) const {
        return make_float4((v.x - s0.x) * s0.y * g.x + be.x, (v.y - s0.x) * s0.y * g.y + be.y, (v.z - s1.x) * s1.y * g.z + be.z,
                           (v.w - s1.x) * s1.y * g.w + be.w);
    }
};

// {mean, 1 / sqrt(var + eps)} from the f64 sums of a group (biased variance, torch.nn.GroupNorm).  Written without an IEEE f64 division or square root:
// those are ~150 instructions per wave, and with every workgroup of gn_apply finishing its own statistics they were 2.6 us of a 10 us launch (round 5
// kernel trace).  inv_cnt = 1 / (pixels * channels per group) from the host; 1 / sqrt by the f32 hardware estimate + one Newton step in f64 (1e-14).
__device__ __forceinline__ float2 gn_mean_rstd(double a, double q, double inv_cnt, float eps) {
    const double mean = a * inv_cnt;
    double var = q * inv_cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double x = var + (double)eps;
    const double y0 = (double)rsqrtf((float)x);
    const double y1 = y0 * (1.5 - 0.5 * x * y0 * y0);
    return make_float2((float)mean, (float)y1);
}

// a += sum of x, q += sum of y over tiles t0, t0 + step, ... < tps of one group (base -> tile 0's float2, tiles G float2s apart), in that order, in f64.
// U loads are issued back to back (predicated) before the first is used: the partials were written by the previous kernel on other XCDs, every dependent
// round trip is ~2 us -- a loop of load-then-add over 9 tiles made a 13 us launch take 30 (round 5 kernel trace).  Callers size U so that one trip suffices.
template <int U>
__device__ __forceinline__ void gn_sum_partials(const float2* __restrict__ base, int t0, int step, int tps, int G, double& a, double& q) {
    for (int t = t0; t < tps; t += U * step) {
        float2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int tt = t + u * step; v[u] = base[(size_t)(tt < tps ? tt : tps - 1) * G]; }   // clamped, not predicated: a load under a branch
#pragma unroll                                                                                                          // comes with its own s_waitcnt vmcnt(0)
        for (int u = 0; u < U; ++u) { const int tt = t + u * step; a += tt < tps ? (double)v[u].x : 0.0; q += tt < tps ? (double)v[u].y : 0.0; }
    }
}

// ---- GroupNorm statistics from the per-tile partials of the producing convolution (igemm ST epilogue) ----
// part [tiles of the launch][G][2] = {sum, sum of squares} per M tile and group; a sample owns tps consecutive tiles.  Thread (sl = tid / gw,
// j = tid % gw) walks tiles sl, sl + nsl, ... of group g0 + j in f64 (eight loads in flight), the slices are added in slice order: a fixed order for
// a given (tps, gw), so the statistics are bitwise reproducible.  Result: st[j] = {mean, 1 / sqrt(var + eps)} (biased variance, torch.nn.GroupNorm)
// for j < gw; ends with a barrier.  red: 512 doubles of LDS.  gw must divide 256.
__device__ __forceinline__ void gn_finish_groups(const float* __restrict__ part, int tps, int G, int b, int g0, int gw, double inv_cnt, float eps, float2* st, double* red) {
    const int tid = threadIdx.x, nsl = 256 / gw, j = tid % gw, sl = tid / gw;
    double a = 0.0, q = 0.0;
    gn_sum_partials<8>(reinterpret_cast<const float2*>(part) + (size_t)b * tps * G + g0 + j, sl, nsl, tps, G, a, q);
    red[tid * 2] = a; red[tid * 2 + 1] = q;
    __syncthreads();
    if (tid < gw) {
        double sa = 0.0, sq = 0.0;
        for (int c2 = 0; c2 < nsl; ++c2) { sa += red[(c2 * gw + tid) * 2]; sq += red[(c2 * gw + tid) * 2 + 1]; }
        st[tid] = gn_mean_rstd(sa, sq, inv_cnt, eps);
    }
    __syncthreads();
}

// stand-alone finish (the stem, whose reader is the max-pool kernel; tests): one workgroup per (sample, gw groups)
__global__ __launch_bounds__(256) void gn_finish_kernel(const float* __restrict__ part, float* __restrict__ stats, int tps, int G, int gw, int hw, int cpg, float eps) {
    __shared__ float2 st[256];
    __shared__ double red[512];
    const int nb = G / gw, b = blockIdx.x / nb, g0 = (blockIdx.x - b * nb) * gw;
    gn_finish_groups(part, tps, G, b, g0, gw, 1.0 / ((double)hw * cpg), eps, st, red);
    if (threadIdx.x < gw) *reinterpret_cast<float2*>(stats + ((size_t)b * G + g0 + threadIdx.x) * 2) = st[threadIdx.x];
}

// ---- GroupNorm apply (+ shortcut) (+ ReLU): one thread = the same 4 channels of U pixels, a workgroup = 256 * U consecutive float4s of ONE sample ----
//   y = GN(raw)                                   (+ GN2(raw2): the projection shortcut of a stage's first block)
//                                                 (+ res:       the identity shortcut, f32 residual stream)
//   y = relu(y) when relu != 0
// out_f32 [M][C] (residual stream), out_op [M][C] and out_halo [B][H+2][W+2][C] (operand copies) are each optional.
// Statistics, by mode: 0 read from stats / stats2; 1 every thread adds the per-tile partials of its own group(s) (short lists, no barrier); 2 the workgroup
// shares the walk through LDS (two barriers) -- from the producing convolution's partials (igemm statistics epilogue), while the raw loads are in flight.  Round 5: the
// producer's own last-arriver finish cost 2-7 us on the critical path of every ResNetV2 convolution (tools/rn_stamps.py).  All U (+U) loads of a thread are
// issued before anything else and the statistics are put together ONCE per thread: these launches are one or two rounds of workgroups long, i.e. latency --
// tried and measured (kernel trace, us per launch, 2304 x 1024 channels + residual): one float4 per thread reading finished statistics 7.9, with
// the per-thread walk 10.5 (nine workgroups per CU where eight fit: two rounds of a longer chain), a wave sharing the walk by shuffles 10.5, lanes of a
// group taking different tiles 10.3 (and 30 at 4608 workgroups: four cache lines per quad of lanes), IEEE f64 division / square root in the finish +2.6.
struct GnApplyDev {
    const float *raw, *stats, *gamma, *beta, *raw2, *stats2, *gamma2, *beta2, *res, *part, *part2;
    float *out_f32, *stats_w, *stats2_w;
    void *out_op, *out_halo;
    int relu, HW, W, C, cpg, halo_mode, tps, tps2, mode;
    size_t M;
    double inv_cnt;   // 1 / (HW * cpg)
    float eps;
};
// statistics of ONE group from its tps partials, a thread for itself
__device__ __forceinline__ float2 gn_finish_direct(const float* __restrict__ part, int tps, int G, int b, int g, double inv_cnt, float eps) {
    double a = 0.0, q = 0.0;
    gn_sum_partials<12>(reinterpret_cast<const float2*>(part) + (size_t)b * tps * G + g, 0, 1, tps, G, a, q);
    return gn_mean_rstd(a, q, inv_cnt, eps);
}
template <int OUT, int U>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnApplyDev a) {
    __shared__ float2 st[2][256];
    __shared__ double red[512];
    const int C = a.C, cpg = a.cpg, G = C / cpg, c4 = C / 4;
    // host: HW * c4 is a multiple of 256 * U (whole workgroups inside one sample) and c4 divides 256 (a thread keeps its channels)
    const size_t k0 = (size_t)blockIdx.x * (256 * U) + threadIdx.x;      // float4 index of this thread's first pixel
    const int b = (int)(k0 / ((size_t)a.HW * c4));
    const int c = (int)(k0 % c4) * 4;
    float4 v[U], w[U];
    const float* second = a.raw2 ? a.raw2 : (a.res ? a.res : a.raw);   // always loaded (no branch around a load: the compiler waits for each such load by itself)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t e = (k0 + (size_t)u * 256) * 4;
        v[u] = *reinterpret_cast<const float4*>(a.raw + e);
        w[u] = *reinterpret_cast<const float4*>(second + e);
    }
    const int g0 = c / cpg, g1 = cpg == 2 ? g0 + 1 : g0;   // cpg == 2: a float4 spans two groups
    // the affine parameters too are requested before the statistics are put together (pointer selects, no load under a condition)
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c), be = *reinterpret_cast<const float4*>(a.beta + c);
    const float4 ga2 = *reinterpret_cast<const float4*>((a.raw2 ? a.gamma2 : a.gamma) + c), be2 = *reinterpret_cast<const float4*>((a.raw2 ? a.beta2 : a.beta) + c);
    float2 s0, s1, t0 = make_float2(0.f, 0.f), t1 = t0;
    if (a.mode == 2) {
        gn_finish_groups(a.part, a.tps, G, b, 0, G, a.inv_cnt, a.eps, st[0], red);
        if (a.raw2) gn_finish_groups(a.part2, a.tps2, G, b, 0, G, a.inv_cnt, a.eps, st[1], red);
        s0 = st[0][g0]; s1 = st[0][g1];
        if (a.raw2) { t0 = st[1][g0]; t1 = st[1][g1]; }
        if (k0 - (size_t)b * a.HW * c4 < 256 && (int)threadIdx.x < G) {   // first workgroup of the sample
            *reinterpret_cast<float2*>(a.stats_w + ((size_t)b * G + threadIdx.x) * 2) = st[0][threadIdx.x];
            if (a.raw2) *reinterpret_cast<float2*>(a.stats2_w + ((size_t)b * G + threadIdx.x) * 2) = st[1][threadIdx.x];
        }
    } else if (a.mode == 1) {
        s0 = gn_finish_direct(a.part, a.tps, G, b, g0, a.inv_cnt, a.eps);
        s1 = cpg == 2 ? gn_finish_direct(a.part, a.tps, G, b, g1, a.inv_cnt, a.eps) : s0;
        if (a.raw2) {
            t0 = gn_finish_direct(a.part2, a.tps2, G, b, g0, a.inv_cnt, a.eps);
            t1 = cpg == 2 ? gn_finish_direct(a.part2, a.tps2, G, b, g1, a.inv_cnt, a.eps) : t0;
        }
        if (k0 - (size_t)b * a.HW * c4 < (size_t)c4 && c % cpg == 0) {   // first pixel of the sample, first thread of every group (cpg == 2: of every pair of groups)
            *reinterpret_cast<float2*>(a.stats_w + ((size_t)b * G + g0) * 2) = s0;
            if (cpg == 2) *reinterpret_cast<float2*>(a.stats_w + ((size_t)b * G + g1) * 2) = s1;
            if (a.raw2) {
                *reinterpret_cast<float2*>(a.stats2_w + ((size_t)b * G + g0) * 2) = t0;
                if (cpg == 2) *reinterpret_cast<float2*>(a.stats2_w + ((size_t)b * G + g1) * 2) = t1;
            }
        }
#ifdef SOCCDPT_ABLATIONS
    } else if (a.mode == 3 || a.mode == 4) {   // timing-only ablations of mode 1: 3 = the partial loads + f32 sums only, 4 = the loads only (one tile)
        s0 = *reinterpret_cast<const float2*>(a.stats + ((size_t)b * G + g0) * 2);
        s1 = s0;
        const float2* base = reinterpret_cast<const float2*>(a.part) + (size_t)b * a.tps * G + g0;
        float acc = 0.f;
        const int nt = a.mode == 3 ? a.tps : 1;
        for (int t = 0; t < nt; ++t) acc += base[(size_t)t * G].x;
        s0.x += 0.f * acc;
#endif
    } else {
        const float* sp2 = a.raw2 ? a.stats2 : a.stats;
        s0 = *reinterpret_cast<const float2*>(a.stats + ((size_t)b * G + g0) * 2);
        s1 = *reinterpret_cast<const float2*>(a.stats + ((size_t)b * G + g1) * 2);
        t0 = *reinterpret_cast<const float2*>(sp2 + ((size_t)b * G + g0) * 2);
        t1 = *reinterpret_cast<const float2*>(sp2 + ((size_t)b * G + g1) * 2);
    }
    // ((x - mean) * rstd) * gamma + beta, the order torch.group_norm uses
    auto norm = [](float4 x, float2 p0, float2 p1, float4 g, float4 bb) {
        return make_float4((x.x - p0.x) * p0.y * g.x + bb.x, (x.y - p0.x) * p0.y * g.y + bb.y, (x.z - p1.x) * p1.y * g.z + bb.z, (x.w - p1.x) * p1.y * g.w + bb.w);
    };
    const int H = a.HW / a.W;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t k = k0 + (size_t)u * 256;
        const size_t m = k / c4;
        float4 y = norm(v[u], s0, s1, ga, be);
        if (a.raw2) { const float4 s = norm(w[u], t0, t1, ga2, be2); y.x += s.x; y.y += s.y; y.z += s.z; y.w += s.w; }
        else if (a.res) { y.x += w[u].x; y.y += w[u].y; y.z += w[u].z; y.w += w[u].w; }
        if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        const size_t e = k * 4;
        if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + e) = y;
        if (a.out_op) store4<OUT>(a.out_op, e, y.x, y.y, y.z, y.w);
        if (a.out_halo) {
            const int rem = (int)(m - (size_t)b * a.HW);
            const int py = rem / a.W, px = rem - py * a.W;
            const size_t hidx = (((size_t)b * (H + 2) + py + 1) * (a.W + 2) + px + 1) * C + c;
            // halo_mode: format of the halo image when it differs from out_op's (SOCCDPT_PREC_MIXED: the hooked stage output feeds the decoder's group)
            if (a.halo_mode == OUT) store4<OUT>(a.out_halo, hidx, y.x, y.y, y.z, y.w);
            else if (a.halo_mode == 3) store4<3>(a.out_halo, hidx, y.x, y.y, y.z, y.w);
            else if (a.halo_mode == 2) store4<2>(a.out_halo, hidx, y.x, y.y, y.z, y.w);
            else if (a.halo_mode == 1) store4<1>(a.out_halo, hidx, y.x, y.y, y.z, y.w);
            else store4<0>(a.out_halo, hidx, y.x, y.y, y.z, y.w);
        }
    }
}

// ---- stem: GroupNorm + ReLU + MaxPool2dSame(3, stride 2) fused.  raw [B][Hi][Hi][C] -> out [B][Ho][Ho][C] operand type.
// 'SAME' pooling pads right / bottom with -inf (timm MaxPool2dSame), i.e. out-of-range taps are simply skipped.
template <int OUT>
__global__ __launch_bounds__(256) void gn_relu_maxpool_kernel(const float* __restrict__ raw, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, void* __restrict__ out, int B, int Hi, int Ho, int C, int cpg) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    const size_t total = (size_t)B * Ho * Ho * c4;
    if (gid >= total) return;
    const size_t m = gid / c4;
    const int c = (int)(gid - m * c4) * 4;
    const int ox = (int)(m % Ho), oy = (int)((m / Ho) % Ho), b = (int)(m / ((size_t)Ho * Ho));
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);   // every candidate is >= 0 after the ReLU and the centre tap always exists
    // the nine taps are loaded first, from clamped coordinates (a tap beyond the image repeats the last row / column: the maximum does not change), and only then
    // normalised: loads under `if (ix < Hi)` came out as nine dependent round trips (round 5, tools/isa_scan.py)
    float4 tap[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 + ky < Hi ? oy * 2 + ky : Hi - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 + kx < Hi ? ox * 2 + kx : Hi - 1;
            tap[ky * 3 + kx] = *reinterpret_cast<const float4*>(raw + (((size_t)b * Hi + iy) * Hi + ix) * C + c);
        }
    }
    Gn4 n4; Gn4x2 n2;
    if (cpg == 2) n2.load(stats, gamma, beta, b, c, C); else n4.load(stats, gamma, beta, b, c, C, cpg);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 v = cpg == 2 ? n2.apply(tap[k]) : n4.apply(tap[k]);
        best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
    }
    store4<OUT>(out, m * C + c, best.x, best.y, best.z, best.w);
}

// ---- LayerNorm rows: one wave per row of C = 768 (12 values per lane as three float4) ----
// TOK != 0: the row is first assembled from the patch projection: row 0 of a sample = cls + pos[0], row t = y[b][t-1] + pos[t]
// (forward_flex, /root/reference/SOccDPT/model/backbones/vit.py:58-77), and written to xf.
template <int OUT, int TOK>
__global__ __launch_bounds__(256) void ln768_kernel(const float* __restrict__ y, const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ xf,
                                                    const float* __restrict__ g, const float* __restrict__ be, void* __restrict__ xb, int rows, int ntok, float eps) {
    constexpr int C = 768;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[3];
    if constexpr (TOK) {
        const int b = row / ntok, t = row - b * ntok;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = k * 256 + lane * 4;
            const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)t * C + c);
            const float4 s = t == 0 ? *reinterpret_cast<const float4*>(cls + c) : *reinterpret_cast<const float4*>(y + ((size_t)b * (ntok - 1) + t - 1) * C + c);
            v[k] = make_float4(s.x + p.x, s.y + p.y, s.z + p.z, s.w + p.w);
            *reinterpret_cast<float4*>(xf + (size_t)row * C + c) = v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = *reinterpret_cast<const float4*>(xf + (size_t)row * C + k * 256 + lane * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = v[k].x - mean, b2 = v[k].y - mean, c2 = v[k].z - mean, d2 = v[k].w - mean;
        q += (a * a + b2 * b2) + (c2 * c2 + d2 * d2);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / C) + eps);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = k * 256 + lane * 4;
        const float4 gg = *reinterpret_cast<const float4*>(g + c), bb = *reinterpret_cast<const float4*>(be + c);
        store4<OUT>(xb, (size_t)row * C + c, (v[k].x - mean) * rstd * gg.x + bb.x, (v[k].y - mean) * rstd * gg.y + bb.y,
                    (v[k].z - mean) * rstd * gg.z + bb.z, (v[k].w - mean) * rstd * gg.w + bb.w);
    }
}

// position-embedding resize at prepare time (_resize_pos_embed, vit.py:23-41): bilinear, align_corners=False, [1 + g0*g0][C] -> [1 + g*g][C]
__global__ void pos_embed_resize_kernel(const float* __restrict__ pos, float* __restrict__ out, int g0, int g, int C) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(1 + g * g) * C;
    if (gid >= total) return;
    const int c = (int)(gid % C), t = (int)(gid / C);
    if (t == 0) { out[gid] = pos[c]; return; }
    const int oy = (t - 1) / g, ox = (t - 1) % g;
    const float sc = (float)g0 / (float)g;
    float fy = ((float)oy + 0.5f) * sc - 0.5f, fx = ((float)ox + 0.5f) * sc - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < g0 - 1), x1 = x0 + (x0 < g0 - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto P = [&](int yy, int xx) { return pos[(size_t)(1 + yy * g0 + xx) * C + c]; };
    out[gid] = (1.f - ly) * ((1.f - lx) * P(y0, x0) + lx * P(y0, x1)) + ly * ((1.f - lx) * P(y1, x0) + lx * P(y1, x1));
}

}  // namespace

constexpr int kGnDirectTps = 12, kGnCoopTps = 64;   // see launch_gn_apply
int launch_gn_finish(const float* part, float* stats, int B, int tps, int G, int hw, int cpg, float eps, hipStream_t st, std::string& err);

#define DISPATCH_OUT(mode, ...)                                              \
    do {                                                                     \
        if ((mode) == 3) { constexpr int OUT = 3; __VA_ARGS__; }             \
        else if ((mode) == 2) { constexpr int OUT = 2; __VA_ARGS__; }        \
        else if ((mode) == 1) { constexpr int OUT = 1; __VA_ARGS__; }        \
        else { constexpr int OUT = 0; __VA_ARGS__; }                         \
    } while (0)

int launch_ws_conv_w(const float* w, void* out, int out_mode, int Cout, int Cin, int k, int Kpad, float eps, hipStream_t st, std::string& err) {
    if (Kpad < Cin * k * k) { err = "ws_conv_w: Kpad too small"; return 1; }
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(ws_conv_w_kernel<OUT>, dim3((unsigned)Cout), dim3(256), 0, st, w, out, Cin, k * k, Kpad, eps));
    return check_launch("ws_conv_w", err);
}

int launch_stem_im2col(const float* x, void* A, int out_mode, int B, int S, hipStream_t st, std::string& err) {
    if (S % 2) { err = "stem_im2col: odd image size"; return 1; }
    const int Ho = S / 2;
    const int total_pad = (Ho - 1) * 2 + 7 - S;   // TF 'SAME'
    const int pad_lo = (total_pad > 0 ? total_pad : 0) / 2;
    const size_t total = (size_t)B * Ho * Ho * 20;
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(stem_im2col_kernel<OUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, A, B, S, Ho, pad_lo));
    return check_launch("stem_im2col", err);
}

int launch_gn_apply(const GnApplyArgs& a, int out_mode, hipStream_t st, std::string& err) {
    if (a.C % 4 || a.cpg < 2 || (a.cpg != 2 && a.cpg % 4) || a.C % a.cpg || a.HW <= 0 || a.M % (size_t)a.HW) { err = "gn_apply: bad geometry"; return 1; }
    if (a.raw2 && a.res) { err = "gn_apply: one shortcut kind at a time"; return 1; }
    const int G = a.C / a.cpg, B = (int)(a.M / (size_t)a.HW), c4 = a.C / 4;
    const size_t per_sample = (size_t)a.HW * c4;
    if (c4 > 256 || 256 % c4 || per_sample % 256 || G > 256) { err = "gn_apply: C / 4 must divide 256 and a sample must be whole workgroups"; return 1; }
    if (a.part && (a.tps <= 0 || (a.raw2 && (!a.part2 || a.tps2 <= 0)) || !a.stats || (a.raw2 && !a.stats2))) { err = "gn_apply: bad deferred-statistics arguments"; return 1; }
    GnApplyDev d;
    d.raw = a.raw; d.stats = a.stats; d.gamma = a.gamma; d.beta = a.beta; d.raw2 = a.raw2; d.stats2 = a.stats2; d.gamma2 = a.gamma2; d.beta2 = a.beta2; d.res = a.res;
    d.part = a.part; d.part2 = a.part2; d.out_f32 = a.out_f32; d.stats_w = const_cast<float*>(a.stats); d.stats2_w = const_cast<float*>(a.stats2);
    d.out_op = a.out_op; d.out_halo = a.out_halo; d.relu = a.relu; d.HW = a.HW; d.W = a.W; d.C = a.C; d.cpg = a.cpg; d.halo_mode = a.halo_mode < 0 ? out_mode : a.halo_mode;
    d.tps = a.tps; d.tps2 = a.tps2; d.M = a.M; d.eps = a.eps; d.inv_cnt = 1.0 / ((double)a.HW * a.cpg);
    // Statistics from partials: up to kGnDirectTps tiles per sample every thread adds its own group's (ResNetV2 stage 2: 9 tiles), up to kGnCoopTps the
    // workgroup shares the walk through LDS (stage 1: 18 .. 36), longer lists (stage 0: 72 .. 144) would cost more than the pass itself -- a small launch
    // adds them up first (gn_finish_kernel) and this one reads {mean, rstd} like any other
    const int tmax = a.part ? (a.raw2 && a.tps2 > a.tps ? a.tps2 : a.tps) : 0;
    d.mode = !a.part ? 0 : (tmax <= kGnDirectTps ? 1 : (tmax <= kGnCoopTps && 256 % G == 0 ? 2 : 0));
#ifdef SOCCDPT_ABLATIONS
    static const int dbg_stale = getenv("SOCCDPT_DBG_GN_STALE") ? atoi(getenv("SOCCDPT_DBG_GN_STALE")) : 0;   // timing-only ablation: 1 = no finish at all (statistics of the previous forward), 2 = also one pixel per thread
    static const bool dbg_warned = dbg_stale ? (fprintf(stderr, "soccdpt: SOCCDPT_DBG_GN_STALE is set: TIMING-ONLY ablation -- GroupNorm statistics are those of the previous forward, results are wrong unless the input repeats\n"), true) : false;
    (void)dbg_warned;
    bool skip_finish = false;   // bit 0: the per-thread walks, bit 1: the workgroup walks, bit 2: the finish launches
    if (a.part && d.mode == 1 && (dbg_stale & 1)) { d.mode = 0; skip_finish = true; }
    if (a.part && d.mode == 1 && !a.raw2 && (dbg_stale == 8 || dbg_stale == 16)) { d.mode = dbg_stale == 8 ? 3 : 4; skip_finish = true; }
    if (a.part && d.mode == 2 && (dbg_stale & 2)) { d.mode = 0; skip_finish = true; }
    if (a.part && d.mode == 0 && !skip_finish && (dbg_stale & 4)) skip_finish = true;
#else
    constexpr bool skip_finish = false;
#endif
    if (a.part && d.mode == 0 && !skip_finish) {
        if (launch_gn_finish(a.part, d.stats_w, B, a.tps, G, a.HW, a.cpg, a.eps, st, err)) return 1;
        if (a.raw2 && launch_gn_finish(a.part2, d.stats2_w, B, a.tps2, G, a.HW, a.cpg, a.eps, st, err)) return 1;
    }
    // pixels per thread: as many (4, 2, 1) as still leave about a thousand workgroups -- one round of workgroups instead of two or more
    const size_t total = a.M * (size_t)c4;
    int U = 1;
    if (per_sample % 1024 == 0 && total / 1024 >= 1024) U = 4;
    else if (per_sample % 512 == 0 && total / 512 >= 1024) U = 2;
    const dim3 grid((unsigned)(total / (256 * (size_t)U)));
    if (U == 4) DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((gn_apply_kernel<OUT, 4>), grid, dim3(256), 0, st, d));
    else if (U == 2) DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((gn_apply_kernel<OUT, 2>), grid, dim3(256), 0, st, d));
    else DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((gn_apply_kernel<OUT, 1>), grid, dim3(256), 0, st, d));
    return check_launch("gn_apply", err);
}

int launch_gn_finish(const float* part, float* stats, int B, int tps, int G, int hw, int cpg, float eps, hipStream_t st, std::string& err) {
    if (!part || !stats || B <= 0 || tps <= 0 || G <= 0 || G > 256) { err = "gn_finish: bad arguments"; return 1; }
    int gw = G;                      // few tiles: one workgroup per sample; many: split the groups so that every thread still has work
    while (gw > 1 && gw % 2 == 0 && 256 / gw < tps / 4) gw /= 2;
    if (256 % gw) { err = "gn_finish: the group count must divide 256"; return 1; }
    SOCCDPT_LAUNCH(gn_finish_kernel, dim3((unsigned)(B * (G / gw))), dim3(256), 0, st, part, stats, tps, G, gw, hw, cpg, eps);
    return check_launch("gn_finish", err);
}

int launch_gn_relu_maxpool(const float* raw, const float* stats, const float* gamma, const float* beta, void* out, int out_mode, int B, int Hi, int C, int cpg,
                           hipStream_t st, std::string& err) {
    const int Ho = (Hi + 1) / 2;
    const size_t total = (size_t)B * Ho * Ho * (C / 4);
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(gn_relu_maxpool_kernel<OUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, raw, stats, gamma, beta, out, B, Hi,
                                               Ho, C, cpg));
    return check_launch("gn_relu_maxpool", err);
}

int launch_vit_tokens_ln(const float* y, const float* cls, const float* pos, float* xf, const float* g, const float* be, void* xb, int out_mode, int B, int ntok,
                         int C, float eps, hipStream_t st, std::string& err) {
    if (C != 768) { err = "vit_tokens_ln: C must be 768"; return 1; }
    const int rows = B * ntok;
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((ln768_kernel<OUT, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, y, cls, pos, xf, g, be, xb, rows, ntok, eps));
    return check_launch("vit_tokens_ln", err);
}

int launch_ln_rows(float* xf, const float* g, const float* be, void* xb, int out_mode, int rows, int C, float eps, hipStream_t st, std::string& err) {
    if (C != 768) { err = "ln_rows: C must be 768"; return 1; }
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((ln768_kernel<OUT, 0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, nullptr, nullptr, nullptr, xf, g, be, xb, rows,
                                               0, eps));
    return check_launch("ln_rows", err);
}

int launch_pos_embed_resize(const float* pos, float* out, int g0, int g, int C, hipStream_t st, std::string& err) {
    const size_t total = (size_t)(1 + g * g) * C;
    SOCCDPT_LAUNCH(pos_embed_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pos, out, g0, g, C);
    return check_launch("pos_embed_resize", err);
}

}  // namespace soccdpt
