// Training criterion at camera resolution, forward + gradient w.r.t. the network outputs (first component of
// the patch-wise training step) -- HBM-bound stencil / reduction kernels, no up-sampled prediction tensors in the autograd sense:
//
//   y_disp_pred = clamp_1e-8(bicubic(inv))           /root/reference/SOccDPT/model/SOccDPT.py:270-287
//   y_seg_pred  = nearest(seg)                        /root/reference/SOccDPT/model/SOccDPT.py:278-282
//   loss_disp   = ScaleAndShiftInvariantLoss          /root/reference/SOccDPT/loss/ssi_loss.py:5-158 (alpha 0.5, 4 scales,
//                                                      batch-based reduction, scale/shift INSIDE the autograd graph)
//   loss_seg    = BCELoss(mean) over masked pixels    /root/reference/SOccDPT/scripts/train_SOccDPT.py:323-338
//   loss        = w_d * loss_disp + w_s * loss_seg    /root/reference/SOccDPT/scripts/train_SOccDPT.py:380-386
//
// The reference builds ~40 full-resolution autograd temporaries per step (B x 1080 x 1920 each).  Here:
//   A  up-sample once (kept as the raw bicubic value: the clamp flag is raw < 1e-8), masked least-squares sums and the mask
//      counts of the 4 gradient-loss sub-grids                                            (1 read of inv/target/mask, 1 write)
//   B  per-pixel dL/d(ssi): data term + the +-sign() stencil of the 4 sub-grids, loss numerators, and the two per-image
//      sums that carry the gradient through the scale/shift solve                                                  (1 write)
//   C  dL/d(inv) by a separable GATHER through the bicubic footprint (vertical, then horizontal): deterministic, no atomics
//   D  BCE value + gradient per network-resolution cell (gather over the cell's nearest-neighbour footprint)
// Scalars are accumulated in f64 (block sums + one atomic per block); the reference sums in f32.
#include "kernels.h"
#include "resample.h"

namespace soccdpt {

namespace {

constexpr int NSC = 4;  // gradient-loss scales (ssi_loss.py:104-121)

// f64 scalar block: [b][IMG_*] per image, then GLOB_* once
enum { IMG_A00 = 0, IMG_A01, IMG_A11, IMG_B0, IMG_B1, IMG_GS, IMG_GT, IMG_N };
enum { GLOB_M0 = 0, GLOB_MK = 1 /* +k */, GLOB_E0 = GLOB_MK + NSC, GLOB_LK /* +k */, GLOB_BCE = GLOB_LK + NSC, GLOB_NSEG, GLOB_N };

struct LossParams {
    int B, H, W, h, w, C;
    int compute_ss;
    float alpha, w_d, w_s;
};

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;  // valid in thread 0
}

__device__ __forceinline__ float clamp_pred(float raw) { return raw < 1e-8f ? 1e-8f : raw; }

// ---- A: bicubic up-sampling (raw), LSQ sums, sub-grid mask counts.  One thread = one camera pixel.
__global__ __launch_bounds__(256) void loss_up_stats_kernel(LossParams P, const float* __restrict__ inv, const float* __restrict__ y,
                                                             const uint8_t* __restrict__ mask, float* __restrict__ raw_up, double* __restrict__ sc) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const size_t npix = (size_t)P.H * P.W;
    const float sy = (float)P.h / (float)P.H, sx = (float)P.w / (float)P.W;
    const float* src = inv + (size_t)b * P.h * P.w;
    double a[5] = {0, 0, 0, 0, 0}, mk[NSC] = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / P.W), c = (int)(i - (size_t)r * P.W);
        const Taps ty = cubic_taps(r, P.h, sy), tx = cubic_taps(c, P.w, sx);
        float rows[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* rp = src + (size_t)ty.idx[k] * P.w;
            rows[k] = dot4(rp[tx.idx[0]], rp[tx.idx[1]], rp[tx.idx[2]], rp[tx.idx[3]], tx.w);
        }
        const float raw = dot4(rows[0], rows[1], rows[2], rows[3], ty.w);
        raw_up[(size_t)b * npix + i] = raw;
        if (mask[(size_t)b * npix + i]) {
            const float p = clamp_pred(raw), t = y[(size_t)b * npix + i];
            a[0] += (double)(p * p);
            a[1] += (double)p;
            a[2] += 1.0;
            a[3] += (double)(p * t);
            a[4] += (double)t;
#pragma unroll
            for (int k = 1; k < NSC; ++k)
                if ((r & ((1 << k) - 1)) == 0 && (c & ((1 << k) - 1)) == 0) mk[k] += 1.0;
        }
    }
    double* img = sc + (size_t)b * IMG_N;
    double* glob = sc + (size_t)P.B * IMG_N;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double s = block_sum_d(a[k], sh);
        if (threadIdx.x == 0) {
            atomicAdd(&img[IMG_A00 + k], s);
            if (k == 2) atomicAdd(&glob[GLOB_M0], s);
        }
    }
#pragma unroll
    for (int k = 1; k < NSC; ++k) {
        const double s = block_sum_d(mk[k], sh);
        if (threadIdx.x == 0) atomicAdd(&glob[GLOB_MK + k], s);
    }
}

// scale / shift of image b in f32 like the reference (x_0, x_1; zero when det == 0) -- recomputed wherever needed
// (the 2x2 solve and its derivative are evaluated in f64: they subtract sums of ~1e6 terms of similar size)
struct Solve {
    float s, t;
    double det, a00, a01, a11, b0, b1;
    bool ok;
};
__device__ __forceinline__ Solve solve_image(const LossParams& P, const double* img) {
    Solve q;
    q.a00 = img[IMG_A00]; q.a01 = img[IMG_A01]; q.a11 = img[IMG_A11]; q.b0 = img[IMG_B0]; q.b1 = img[IMG_B1];
    q.det = q.a00 * q.a11 - q.a01 * q.a01;
    q.ok = P.compute_ss && q.det != 0.0;
    q.s = P.compute_ss ? 0.f : 1.f;
    q.t = 0.f;
    if (q.ok) {
        q.s = (float)((q.a11 * q.b0 - q.a01 * q.b1) / q.det);
        q.t = (float)((-q.a01 * q.b0 + q.a00 * q.b1) / q.det);
    }
    return q;
}

// ---- B: g = dL_disp / d(ssi) per pixel, loss numerators, per-image sum(g p) and sum(g).
__global__ __launch_bounds__(256) void loss_terms_kernel(LossParams P, const float* __restrict__ raw_up, const float* __restrict__ y,
                                                          const uint8_t* __restrict__ mask, float* __restrict__ g_out, double* __restrict__ sc) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const size_t npix = (size_t)P.H * P.W;
    double* img = sc + (size_t)b * IMG_N;
    double* glob = sc + (size_t)P.B * IMG_N;
    const Solve q = solve_image(P, img);
    const float M0 = (float)glob[GLOB_M0];
    float invMk[NSC];
    invMk[0] = M0 > 0.f ? 1.f / M0 : 0.f;
#pragma unroll
    for (int k = 1; k < NSC; ++k) {
        const float mkv = (float)glob[GLOB_MK + k];
        invMk[k] = mkv > 0.f ? 1.f / mkv : 0.f;
    }
    const float* rp = raw_up + (size_t)b * npix;
    const float* yp = y + (size_t)b * npix;
    const uint8_t* mp = mask + (size_t)b * npix;
    auto diff_at = [&](int r, int c, float& m) -> float {  // mask * (ssi - target) at (r, c)
        const size_t o = (size_t)r * P.W + c;
        m = mp[o] ? 1.f : 0.f;
        return m * (q.s * clamp_pred(rp[o]) + q.t - yp[o]);
    };
    double e0 = 0.0, lk[NSC] = {0, 0, 0, 0}, gs = 0.0, gt = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / P.W), c = (int)(i - (size_t)r * P.W);
        float m;
        const float d = diff_at(r, c, m);
        const float p = clamp_pred(rp[i]);
        float g = 0.f;
        if (m != 0.f) {
            e0 += (double)(d * d);
            g = d * invMk[0];  // d/dssi of sum(m res^2) / (2 sum M)
        }
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const int step = 1 << k;
            if ((r & (step - 1)) || (c & (step - 1))) continue;
            float acc_sign = 0.f;
            if (c + step < P.W) {  // right neighbour: D = diff_q - diff_i
                float mq;
                const float dq = diff_at(r, c + step, mq);
                const float w = m * mq, D = dq - d;
                lk[k] += (double)(fabsf(D) * w);
                acc_sign -= (D > 0.f ? 1.f : (D < 0.f ? -1.f : 0.f)) * w;
            }
            if (c - step >= 0) {   // left neighbour: D = diff_i - diff_p
                float mq;
                const float dq = diff_at(r, c - step, mq);
                const float D = d - dq;
                acc_sign += (D > 0.f ? 1.f : (D < 0.f ? -1.f : 0.f)) * (m * mq);
            }
            if (r + step < P.H) {
                float mq;
                const float dq = diff_at(r + step, c, mq);
                const float w = m * mq, D = dq - d;
                lk[k] += (double)(fabsf(D) * w);
                acc_sign -= (D > 0.f ? 1.f : (D < 0.f ? -1.f : 0.f)) * w;
            }
            if (r - step >= 0) {
                float mq;
                const float dq = diff_at(r - step, c, mq);
                const float D = d - dq;
                acc_sign += (D > 0.f ? 1.f : (D < 0.f ? -1.f : 0.f)) * (m * mq);
            }
            g += P.alpha * invMk[k] * m * acc_sign;  // d(diff)/d(ssi) = m
        }
        g_out[(size_t)b * npix + i] = g;
        gs += (double)(g * p);
        gt += (double)g;
    }
    double s = block_sum_d(e0, sh);
    if (threadIdx.x == 0) atomicAdd(&glob[GLOB_E0], s);
#pragma unroll
    for (int k = 0; k < NSC; ++k) {
        s = block_sum_d(lk[k], sh);
        if (threadIdx.x == 0) atomicAdd(&glob[GLOB_LK + k], s);
    }
    s = block_sum_d(gs, sh);
    if (threadIdx.x == 0) atomicAdd(&img[IMG_GS], s);
    s = block_sum_d(gt, sh);
    if (threadIdx.x == 0) atomicAdd(&img[IMG_GT], s);
}

// dL/dp at one camera pixel: through ssi = s p + t directly, and through s(p), t(p) of the masked least-squares solve.
// d s / d p_i and d t / d p_i are m_i times affine functions of (p_i, y_i) with per-image coefficients, so
//   dL/dp_i = [raw_i >= 1e-8] * w_d * ( s g_i + m_i (c0 + c1 p_i + c2 y_i) ),   c = Gs * coef(ds) + Gt * coef(dt)
// The coefficients subtract sums of ~1e6 terms of similar size and are formed in f64 once per image.
struct Chain {
    float s, c0, c1, c2;
};
__device__ __forceinline__ Chain chain_image(const LossParams& P, const double* img) {
    const Solve q = solve_image(P, img);
    Chain ch{q.s, 0.f, 0.f, 0.f};
    if (q.ok) {
        const double Gs = img[IMG_GS], Gt = img[IMG_GT], inv_det = 1.0 / q.det;
        const double num_s = q.a11 * q.b0 - q.a01 * q.b1, num_t = -q.a01 * q.b0 + q.a00 * q.b1;
        // d det = 2 (a11 p - a01);  d num_s = a11 y - b1;  d num_t = -b0 - a01 y + 2 b1 p     (d a00 = 2p, d a01 = 1, d b0 = y)
        const double ks = num_s * inv_det, kt = num_t * inv_det;
        const double ds0 = (-q.b1 + 2.0 * ks * q.a01) * inv_det, ds1 = (-2.0 * ks * q.a11) * inv_det, ds2 = q.a11 * inv_det;
        const double dt0 = (-q.b0 + 2.0 * kt * q.a01) * inv_det, dt1 = (2.0 * q.b1 - 2.0 * kt * q.a11) * inv_det, dt2 = -q.a01 * inv_det;
        ch.c0 = (float)(Gs * ds0 + Gt * dt0);
        ch.c1 = (float)(Gs * ds1 + Gt * dt1);
        ch.c2 = (float)(Gs * ds2 + Gt * dt2);
    }
    return ch;
}
__device__ __forceinline__ float dldp_at(const LossParams& P, const Chain& ch, float raw, float yv, bool m, float g) {
    if (raw < 1e-8f) return 0.f;  // clamped in place: no gradient (model/SOccDPT.py:284-285)
    float v = ch.s * g;
    if (m) v += ch.c0 + ch.c1 * raw + ch.c2 * yv;
    return P.w_d * v;
}

// ---- C1: vertical gather.  T[b][ys][X] = sum over camera rows Y whose bicubic footprint holds source row ys of wy * dL/dp[Y][X]
__global__ __launch_bounds__(256) void loss_bwd_vert_kernel(LossParams P, const float* __restrict__ raw_up, const float* __restrict__ y,
                                                             const uint8_t* __restrict__ mask, const float* __restrict__ g,
                                                             const double* __restrict__ sc, float* __restrict__ T) {
    const int b = blockIdx.z, ys = blockIdx.y;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    if (X >= P.W) return;
    const Chain ch = chain_image(P, sc + (size_t)b * IMG_N);
    const float sy = (float)P.h / (float)P.H;
    const size_t npix = (size_t)P.H * P.W;
    // camera rows that can touch source row ys: floor(real) in [ys-2, ys+1] (+ border clamping: scan a safe superset)
    const float inv_s = (float)P.H / (float)P.h;
    int Y0 = (int)floorf(((float)ys - 2.5f) * inv_s) - 1, Y1 = (int)ceilf(((float)ys + 2.5f) * inv_s) + 1;
    if (ys <= 1) Y0 = 0;                  // clamped taps: rows above the first source rows
    if (ys >= P.h - 2) Y1 = P.H - 1;
    Y0 = Y0 < 0 ? 0 : Y0;
    Y1 = Y1 > P.H - 1 ? P.H - 1 : Y1;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        const Taps ty = cubic_taps(Y, P.h, sy);
        float wsum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ty.idx[k] == ys) wsum += ty.w[k];
        if (wsum == 0.f) continue;
        const size_t o = (size_t)b * npix + (size_t)Y * P.W + X;
        acc += wsum * dldp_at(P, ch, raw_up[o], y[o], mask[o] != 0, g[o]);
    }
    T[((size_t)b * P.h + ys) * P.W + X] = acc;
}

// ---- C2: horizontal gather.  d_inv[b][ys][xs] = sum over camera columns X touching source column xs of wx * T[b][ys][X]
__global__ __launch_bounds__(256) void loss_bwd_horz_kernel(LossParams P, const float* __restrict__ T, float* __restrict__ d_inv) {
    const size_t total = (size_t)P.B * P.h * P.w;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xs = (int)(i % P.w);
    const size_t row = i / P.w;  // b * h + ys
    const float sx = (float)P.w / (float)P.W, inv_s = (float)P.W / (float)P.w;
    int X0 = (int)floorf(((float)xs - 2.5f) * inv_s) - 1, X1 = (int)ceilf(((float)xs + 2.5f) * inv_s) + 1;
    if (xs <= 1) X0 = 0;
    if (xs >= P.w - 2) X1 = P.W - 1;
    X0 = X0 < 0 ? 0 : X0;
    X1 = X1 > P.W - 1 ? P.W - 1 : X1;
    const float* Tr = T + row * P.W;
    float acc = 0.f;
    for (int X = X0; X <= X1; ++X) {
        const Taps tx = cubic_taps(X, P.w, sx);
        float wsum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tx.idx[k] == xs) wsum += tx.w[k];
        if (wsum != 0.f) acc += wsum * Tr[X];
    }
    d_inv[i] = acc;
}

// ---- D: BCE over the nearest-neighbour footprint of every network-resolution cell.  Unnormalised: value and gradient are
// divided by the masked-pixel count afterwards (it is only known at the end of this pass).
__global__ __launch_bounds__(256) void loss_bce_kernel(LossParams P, const float* __restrict__ seg, const float* __restrict__ y_seg,
                                                        const uint8_t* __restrict__ mask_seg, float* __restrict__ d_seg, double* __restrict__ sc) {
    __shared__ double sh[4];
    double* glob = sc + (size_t)P.B * IMG_N;
    const size_t total = (size_t)P.B * P.C * P.h * P.w;
    const float sy = (float)P.h / (float)P.H, sx = (float)P.w / (float)P.W;
    double lsum = 0.0, nsum = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xs = (int)(i % P.w), ys = (int)((i / P.w) % P.h);
        const size_t bc = i / ((size_t)P.w * P.h);
        const float qv = seg[i];
        // torch.nn.BCELoss: log terms clamped at -100; backward (q - y) / max((1 - q) q, 1e-12)
        const float lq = fmaxf(logf(qv), -100.f), l1q = fmaxf(log1pf(-qv), -100.f);
        const float inv_den = 1.f / fmaxf((1.f - qv) * qv, 1e-12f);
        // camera rows / columns whose nearest source is (ys, xs): floor(Y * sy) == ys  (clamped at the last source index)
        int Y0 = (int)ceilf((float)ys / sy) - 1, Y1 = (int)ceilf((float)(ys + 1) / sy) + 1;
        int X0 = (int)ceilf((float)xs / sx) - 1, X1 = (int)ceilf((float)(xs + 1) / sx) + 1;
        Y0 = Y0 < 0 ? 0 : Y0; X0 = X0 < 0 ? 0 : X0;
        Y1 = Y1 > P.H - 1 ? P.H - 1 : Y1; X1 = X1 > P.W - 1 ? P.W - 1 : X1;
        float gacc = 0.f;
        for (int Y = Y0; Y <= Y1; ++Y) {
            if (nearest_src(Y, P.h, sy) != ys) continue;
            const size_t ro = (bc * P.H + Y) * P.W;
            for (int X = X0; X <= X1; ++X) {
                if (nearest_src(X, P.w, sx) != xs || !mask_seg[ro + X]) continue;
                const float t = y_seg[ro + X];
                lsum += (double)(-(t * lq + (1.f - t) * l1q));
                nsum += 1.0;
                gacc += (qv - t) * inv_den;
            }
        }
        d_seg[i] = gacc;
    }
    double s = block_sum_d(lsum, sh);
    if (threadIdx.x == 0) atomicAdd(&glob[GLOB_BCE], s);
    s = block_sum_d(nsum, sh);
    if (threadIdx.x == 0) atomicAdd(&glob[GLOB_NSEG], s);
}

// ---- finalisation: scalars out[0..2] = loss, loss_disp, loss_seg (+ out[3 + 2b], out[4 + 2b] = scale, shift); d_seg *= w_s / N
__global__ void loss_finish_kernel(LossParams P, const double* __restrict__ sc, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double* glob = sc + (size_t)P.B * IMG_N;
    float ld = 0.f;
    const float M0 = (float)glob[GLOB_M0];
    if (M0 > 0.f) ld = (float)glob[GLOB_E0] / (2.f * M0);
    if (P.alpha > 0.f) {
        float reg = 0.f;
        for (int k = 0; k < NSC; ++k) {
            const float mk = (float)glob[k == 0 ? GLOB_M0 : GLOB_MK + k];
            if (mk > 0.f) reg += (float)glob[GLOB_LK + k] / mk;
        }
        ld += P.alpha * reg;
    }
    const float n = (float)glob[GLOB_NSEG];
    const float ls = n > 0.f ? (float)glob[GLOB_BCE] / n : 0.f;   // BCELoss(mean) of an empty selection is NaN in torch; 0 here
    out[0] = P.w_d * ld + P.w_s * ls;
    out[1] = ld;
    out[2] = ls;
    for (int b = 0; b < P.B; ++b) {
        const Solve q = solve_image(P, sc + (size_t)b * IMG_N);
        out[3 + 2 * b] = q.s;
        out[4 + 2 * b] = q.t;
    }
}
__global__ __launch_bounds__(256) void loss_scale_dseg_kernel(LossParams P, const double* __restrict__ sc, float* __restrict__ d_seg) {
    const size_t total = (size_t)P.B * P.C * P.h * P.w;
    const float n = (float)sc[(size_t)P.B * IMG_N + GLOB_NSEG];
    const float k = n > 0.f ? P.w_s / n : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) d_seg[i] *= k;
}

}  // namespace

size_t loss_scratch_bytes(int B, int H, int W, int h, int w) {
    const size_t npix = (size_t)B * H * W;
    return ((size_t)B * IMG_N + GLOB_N) * sizeof(double) + 256 + (2 * npix + (size_t)B * h * W) * sizeof(float);
}

int launch_training_loss(int B, int H, int W, int h, int w, int C, int compute_ss, float alpha, float w_d, float w_s, const float* inv,
                         const float* seg, const float* y_disp, const uint8_t* mask_disp, const float* y_seg, const uint8_t* mask_seg,
                         float* out, float* d_inv, float* d_seg, void* scratch, hipStream_t st, std::string& err) {
    if (B <= 0 || H <= 0 || W <= 0 || h <= 1 || w <= 1 || C <= 0) { err = "training_loss: bad sizes"; return 1; }
    if (!inv || !seg || !y_disp || !mask_disp || !y_seg || !mask_seg || !out || !d_inv || !d_seg || !scratch) { err = "training_loss: null pointer"; return 1; }
    LossParams P{B, H, W, h, w, C, compute_ss ? 1 : 0, alpha, w_d, w_s};
    const size_t nsc = (size_t)B * IMG_N + GLOB_N, npix = (size_t)H * W;
    double* sc = static_cast<double*>(scratch);
    float* raw_up = reinterpret_cast<float*>(static_cast<char*>(scratch) + ((nsc * sizeof(double) + 255) / 256) * 256);
    float* g = raw_up + (size_t)B * npix;
    float* T = g + (size_t)B * npix;
    hipError_t e = hipMemsetAsync(sc, 0, nsc * sizeof(double), st);
    if (e != hipSuccess) { err = std::string("training_loss: ") + hipGetErrorString(e); return 1; }
    // few, fat workgroups: every workgroup ends with ~8 f64 atomics on the same handful of addresses, and those serialise
    // (2048 x B workgroups: 590 us; 256 x B: see tools/loss_bench.py)
    unsigned gx = (unsigned)((npix + 255) / 256);
    if (gx > 256) gx = 256;
    SOCCDPT_LAUNCH(loss_up_stats_kernel, dim3(gx, B), dim3(256), 0, st, P, inv, y_disp, mask_disp, raw_up, sc);
    SOCCDPT_LAUNCH(loss_terms_kernel, dim3(gx, B), dim3(256), 0, st, P, raw_up, y_disp, mask_disp, g, sc);
    SOCCDPT_LAUNCH(loss_bwd_vert_kernel, dim3((W + 255) / 256, h, B), dim3(256), 0, st, P, raw_up, y_disp, mask_disp, g, sc, T);
    SOCCDPT_LAUNCH(loss_bwd_horz_kernel, dim3((unsigned)(((size_t)B * h * w + 255) / 256)), dim3(256), 0, st, P, T, d_inv);
    const size_t nseg = (size_t)B * C * h * w;
    unsigned gs = (unsigned)((nseg + 255) / 256);
    if (gs > 1024) gs = 1024;
    SOCCDPT_LAUNCH(loss_bce_kernel, dim3(gs), dim3(256), 0, st, P, seg, y_seg, mask_seg, d_seg, sc);
    SOCCDPT_LAUNCH(loss_scale_dseg_kernel, dim3(gs), dim3(256), 0, st, P, sc, d_seg);
    SOCCDPT_LAUNCH(loss_finish_kernel, dim3(1), dim3(64), 0, st, P, sc, out);
    return check_launch("training_loss", err);
}

}  // namespace soccdpt
