// Backward of the camera-resolution tail of SOccDPT.get_semantic_occupancy (/root/reference/SOccDPT/model/SOccDPT.py:264-353) -- what torch
// autograd runs when a criterion written in torch ops is applied to the tuple SOccDPT_V3.forward returns under net.train()
// (/root/reference/SOccDPT/scripts/train_SOccDPT.py:365-391: `net_patch(x)` ... `grad_scaler.scale(loss).backward()`):
//
//   inv_up = bicubic(inv) ; inv_up[inv_up < 1e-8] = 1e-8      -> d inv  = bicubic^T( [inv_up not clamped] * (d inv_up + points term) )
//   seg_up = nearest(seg)                                      -> d seg  = sum of d seg_up over each cell's nearest-neighbour footprint
//   points = ((v - cx) d / fx, (u - cy) d / fy, d), d = 1 / inv_up (pixels 0..2 of each image scaled by pc_scale)
//                                                              -> points term = -d^2 * (gX (v - cx) / fx + gY (u - cy) / fy + gZ)
// The occupancy grid is a scatter of constants (no gradient).  Deterministic gathers (no float atomics): the separable form of loss.hip's C1 / C2
// kernels with the upstream gradient read from memory instead of being the fused criterion's.  HBM-bound: one read of the camera-resolution
// gradients.  The result feeds soccdpt_train_backward.
#include "internal.h"
#include "kernels.h"
#include "resample.h"

namespace soccdpt {
namespace {

struct UpBwdParams {
    int B, h, w, H, W, C;
    float fx, fy, cx, cy;
    float pc_scale[3];
};

// dL / d inv_up at one camera pixel, including the path through the back-projected point
__device__ __forceinline__ float up_grad_at(const UpBwdParams& P, float out, const float* d_inv_up, const float* d_points, size_t o, int Y, int X) {
    if (out == 1e-8f) return 0.f;   // clamped in the forward (model/SOccDPT.py:286-287): a constant.  (A raw value of exactly 1e-8 is treated as clamped.)
    float g = d_inv_up ? d_inv_up[o] : 0.f;
    if (d_points) {
        float d = 1.0f / out;
        if (isinf(d) || isnan(d)) return g;
        const size_t n = (size_t)Y * P.W + X;
        const float* gp = d_points + o * 3;
        float t = gp[0] * (((float)X - P.cx) / P.fx) + gp[1] * (((float)Y - P.cy) / P.fy) + gp[2];
        if (n < 3) t *= P.pc_scale[n];
        g -= t * d * d;
    }
    return g;
}

// T[b][ys][X] = sum over camera rows Y whose bicubic footprint holds source row ys of wy * g[Y][X]
__global__ __launch_bounds__(256) void up_bwd_vert_kernel(UpBwdParams P, const float* __restrict__ inv_up, const float* __restrict__ d_inv_up,
                                                           const float* __restrict__ d_points, float* __restrict__ T) {
    const int b = blockIdx.z, ys = blockIdx.y;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    if (X >= P.W) return;
    const float sy = (float)P.h / (float)P.H, inv_s = (float)P.H / (float)P.h;
    const size_t npix = (size_t)P.H * P.W;
    int Y0 = (int)floorf(((float)ys - 2.5f) * inv_s) - 1, Y1 = (int)ceilf(((float)ys + 2.5f) * inv_s) + 1;
    if (ys <= 1) Y0 = 0;                  // border-clamped taps
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
        acc += wsum * up_grad_at(P, inv_up[o], d_inv_up, d_points, o, Y, X);
    }
    T[((size_t)b * P.h + ys) * P.W + X] = acc;
}

// d_inv[b][ys][xs] = sum over camera columns X touching source column xs of wx * T[b][ys][X]
__global__ __launch_bounds__(256) void up_bwd_horz_kernel(UpBwdParams P, const float* __restrict__ T, float* __restrict__ d_inv) {
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

// d_seg[b][c][ys][xs] = sum of d_seg_up over the camera pixels whose nearest source cell is (ys, xs)
__global__ __launch_bounds__(256) void up_bwd_nearest_kernel(UpBwdParams P, const float* __restrict__ d_seg_up, float* __restrict__ d_seg) {
    const size_t total = (size_t)P.B * P.C * P.h * P.w;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xs = (int)(i % P.w);
    size_t r = i / P.w;
    const int ys = (int)(r % P.h);
    const size_t bc = r / P.h;
    const float sy = (float)P.h / (float)P.H, sx = (float)P.w / (float)P.W;
    int Y0 = (int)floorf((float)ys / sy) - 1, Y1 = (int)floorf((float)(ys + 1) / sy) + 1;
    int X0 = (int)floorf((float)xs / sx) - 1, X1 = (int)floorf((float)(xs + 1) / sx) + 1;
    Y0 = Y0 < 0 ? 0 : Y0; X0 = X0 < 0 ? 0 : X0;
    Y1 = Y1 > P.H - 1 ? P.H - 1 : Y1; X1 = X1 > P.W - 1 ? P.W - 1 : X1;
    if (ys == P.h - 1) Y1 = P.H - 1;   // the min(.., in - 1) of the nearest index
    if (xs == P.w - 1) X1 = P.W - 1;
    const float* src = d_seg_up + bc * (size_t)P.H * P.W;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        if (nearest_src(Y, P.h, sy) != ys) continue;
        for (int X = X0; X <= X1; ++X)
            if (nearest_src(X, P.w, sx) == xs) acc += src[(size_t)Y * P.W + X];
    }
    d_seg[i] = acc;
}

}  // namespace

size_t upsample_bwd_scratch_bytes(const soccdpt_config& cfg, int B, int h) { return (size_t)B * h * cfg.cam_width * sizeof(float); }

int launch_upsample_bwd(const soccdpt_config& cfg, const float* inv_up, const float* d_inv_up, const float* d_seg_up, const float* d_points, int B, int h, int w,
                        float* d_inv, float* d_seg, void* scratch, hipStream_t st, std::string& err) {
    if (B <= 0 || h <= 0 || w <= 0 || !inv_up || !d_inv || !d_seg || !scratch) { err = "soccdpt_project_backward: bad argument"; return 1; }
    UpBwdParams P;
    P.B = B; P.h = h; P.w = w; P.H = cfg.cam_height; P.W = cfg.cam_width; P.C = cfg.num_classes;
    P.fx = cfg.fx; P.fy = cfg.fy; P.cx = cfg.cx; P.cy = cfg.cy;
    for (int i = 0; i < 3; ++i) P.pc_scale[i] = cfg.pc_scale[i];
    float* T = static_cast<float*>(scratch);
    if (d_inv_up || d_points) {
        SOCCDPT_LAUNCH(up_bwd_vert_kernel, dim3((unsigned)((P.W + 255) / 256), (unsigned)h, (unsigned)B), dim3(256), 0, st, P, inv_up, d_inv_up, d_points, T);
        const size_t n = (size_t)B * h * w;
        SOCCDPT_LAUNCH(up_bwd_horz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, static_cast<const float*>(T), d_inv);
    } else if (hipMemsetAsync(d_inv, 0, (size_t)B * h * w * sizeof(float), st) != hipSuccess) { err = "soccdpt_project_backward: memset failed"; return 1; }
    if (d_seg_up) {
        const size_t n = (size_t)B * P.C * h * w;
        SOCCDPT_LAUNCH(up_bwd_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, d_seg_up, d_seg);
    } else if (hipMemsetAsync(d_seg, 0, (size_t)B * P.C * h * w * sizeof(float), st) != hipSuccess) { err = "soccdpt_project_backward: memset failed"; return 1; }
    return check_launch("upsample_bwd", err);
}

}  // namespace soccdpt
