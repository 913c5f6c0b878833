// On-GPU evaluation metrics (the step after the hot path) — HBM-bound reductions.
//   compute_scale_and_shift  /root/reference/SOccDPT/loss/ssi_loss.py:5-32       (masked least squares per image)
//   compute_masked_errors    /root/reference/SOccDPT/utils/__init__.py:109-158   (abs_rel, sq_rel, rmse, rmse_log, a1..a3
//                                                                                 over ALL masked pixels of the batch)
//   IoU@0.5 per image        /root/reference/SOccDPT/utils/__init__.py:314-330
// The reference round-trips every prediction to the host (`.cpu().numpy()`) per batch; here predictions stay in HBM.
// Sums are accumulated per thread in f32 over short runs, per block in f64, and merged with one f64 atomic per block
// and quantity (the reference sums in f32; differences are at the 1e-6 level).
#include "kernels.h"

namespace soccdpt {

__device__ __forceinline__ double block_sum(double v, double* sh) {
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

// sums[b][5] += {sum m p p, sum m p, sum m, sum m p t, sum m t}
__global__ __launch_bounds__(256) void lsq_sums_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const uint8_t* __restrict__ mask,
                                                        double* __restrict__ sums, size_t npix) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const float* p = pred + (size_t)b * npix;
    const float* t = gt + (size_t)b * npix;
    const uint8_t* m = mask + (size_t)b * npix;
    double a[5] = {0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        if (m[i]) {
            const float pv = p[i], tv = t[i];
            a[0] += (double)(pv * pv);
            a[1] += (double)pv;
            a[2] += 1.0;
            a[3] += (double)(pv * tv);
            a[4] += (double)tv;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double s = block_sum(a[k], sh);
        if (threadIdx.x == 0) atomicAdd(&sums[b * 5 + k], s);
    }
}

// scale/shift per image from the 5 sums (x0, x1 of the reference; zero when det == 0), computed in f32 like the reference
__device__ __forceinline__ void solve_lsq(const double* s, float& scale, float& shift) {
    const float a00 = (float)s[0], a01 = (float)s[1], a11 = (float)s[2], b0 = (float)s[3], b1 = (float)s[4];
    const float det = a00 * a11 - a01 * a01;
    scale = 0.f;
    shift = 0.f;
    if (det != 0.f) {
        scale = (a11 * b0 - a01 * b1) / det;
        shift = (-a01 * b0 + a00 * b1) / det;
    }
}

// acc[8] += {sum |g-p|/g, sum (g-p)^2/g, sum (g-p)^2, sum (ln g - ln p)^2, #(thr<1.25), #(thr<1.25^2), #(thr<1.25^3), count}
__global__ __launch_bounds__(256) void masked_err_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const uint8_t* __restrict__ mask,
                                                          const double* __restrict__ sums, double* __restrict__ acc, size_t npix) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    float scale, shift;
    solve_lsq(sums + b * 5, scale, shift);
    const float* p = pred + (size_t)b * npix;
    const float* t = gt + (size_t)b * npix;
    const uint8_t* m = mask + (size_t)b * npix;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        if (m[i]) {
            const float g = t[i];
            const float q = scale * p[i] + shift;  // y_pred_ssi
            const float d = g - q;
            const float thr = fmaxf(g / q, q / g);
            const float lg = logf(g) - logf(q);
            a[0] += (double)(fabsf(d) / g);
            a[1] += (double)((d * d) / g);
            a[2] += (double)(d * d);
            a[3] += (double)(lg * lg);
            a[4] += (thr < 1.25f) ? 1.0 : 0.0;
            a[5] += (thr < 1.5625f) ? 1.0 : 0.0;
            a[6] += (thr < 1.953125f) ? 1.0 : 0.0;
            a[7] += 1.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double s = block_sum(a[k], sh);
        if (threadIdx.x == 0) atomicAdd(&acc[k], s);
    }
}

// out[0..6] = abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 (NaN/inf -> 0 like the reference); out[7 + 2b], out[8 + 2b] = scale, shift
__global__ void depth_finalize_kernel(const double* __restrict__ sums, const double* __restrict__ acc, float* __restrict__ out, int B) {
    if (threadIdx.x == 0) {
        const double n = acc[7];
        double v[7];
        v[0] = acc[0] / n;
        v[1] = acc[1] / n;
        v[2] = sqrt(acc[2] / n);
        v[3] = sqrt(acc[3] / n);
        v[4] = acc[4] / n;
        v[5] = acc[5] / n;
        v[6] = acc[6] / n;
        for (int k = 0; k < 7; ++k) out[k] = (isnan(v[k]) || isinf(v[k])) ? 0.f : (float)v[k];
    }
    if ((int)threadIdx.x < B) {
        float s, t;
        solve_lsq(sums + threadIdx.x * 5, s, t);
        out[7 + 2 * threadIdx.x] = s;
        out[8 + 2 * threadIdx.x] = t;
    }
}

// counts[b][c][2] += {|pred>0.5 & gt>0.5|, |pred>0.5 | gt>0.5|}
__global__ __launch_bounds__(256) void iou_counts_kernel(const float* __restrict__ pred, const float* __restrict__ gt, unsigned long long* __restrict__ counts,
                                                          size_t npix) {
    const int bc = blockIdx.y;
    const float* p = pred + (size_t)bc * npix;
    const float* t = gt + (size_t)bc * npix;
    unsigned int inter = 0, uni = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const bool pm = p[i] > 0.5f, tm = t[i] > 0.5f;
        inter += (pm && tm) ? 1u : 0u;
        uni += (pm || tm) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inter += __shfl_xor(inter, o);
        uni += __shfl_xor(uni, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[bc * 2 + 0], (unsigned long long)inter);
        atomicAdd(&counts[bc * 2 + 1], (unsigned long long)uni);
    }
}

__global__ void iou_finalize_kernel(const unsigned long long* __restrict__ counts, float* __restrict__ out, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float iou = 0.f;
    for (int c = 0; c < C; ++c) {
        const float inter = (float)counts[(b * C + c) * 2], uni = (float)counts[(b * C + c) * 2 + 1];
        iou += inter / (uni + 1e-7f);
    }
    out[b] = iou / (float)C;
}

size_t metrics_scratch_bytes(int B, int C) { return (size_t)(B * 5 + 8) * sizeof(double) + (size_t)B * C * 2 * sizeof(unsigned long long); }

int launch_depth_metrics(const float* pred, const float* gt, const uint8_t* mask, int B, size_t npix, float* out, void* scratch, hipStream_t st,
                         std::string& err) {
    if (B <= 0 || B > 256 || npix == 0) { err = "depth_metrics: bad shape"; return 1; }
    double* sums = static_cast<double*>(scratch);
    double* acc = sums + B * 5;
    if (hipMemsetAsync(scratch, 0, (size_t)(B * 5 + 8) * sizeof(double), st) != hipSuccess) { err = "depth_metrics: memset failed"; return 1; }
    unsigned bx = (unsigned)((npix + 256 * 8 - 1) / (256 * 8));
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    SOCCDPT_LAUNCH(lsq_sums_kernel, dim3(bx, B), dim3(256), 0, st, pred, gt, mask, sums, npix);
    SOCCDPT_LAUNCH(masked_err_kernel, dim3(bx, B), dim3(256), 0, st, pred, gt, mask, sums, acc, npix);
    SOCCDPT_LAUNCH(depth_finalize_kernel, dim3(1), dim3(256), 0, st, sums, acc, out, B);
    return check_launch("depth_metrics", err);
}

int launch_iou_metrics(const float* pred, const float* gt, int B, int C, size_t npix, float* out, void* scratch, hipStream_t st, std::string& err) {
    if (B <= 0 || C <= 0 || npix == 0) { err = "iou_metrics: bad shape"; return 1; }
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + (size_t)(B * 5 + 8) * sizeof(double));
    if (hipMemsetAsync(counts, 0, (size_t)B * C * 2 * sizeof(unsigned long long), st) != hipSuccess) { err = "iou_metrics: memset failed"; return 1; }
    unsigned bx = (unsigned)((npix + 256 * 8 - 1) / (256 * 8));
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    SOCCDPT_LAUNCH(iou_counts_kernel, dim3(bx, B * C), dim3(256), 0, st, pred, gt, counts, npix);
    SOCCDPT_LAUNCH(iou_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, st, counts, out, B, C);
    return check_launch("iou_metrics", err);
}

}  // namespace soccdpt
