// Input transform of the network: uint8 HWC camera frame -> float32 CHW network input, replacing
// Compose([Resize(cv2.INTER_CUBIC, ensure_multiple_of=32, "minimal"), NormalizeImage, PrepareForNet])
// (/root/reference/SOccDPT/model/loader.py:256-270, model/transforms.py:53-251) -- today a host-side cv2 call per 1920x1080 frame.
// The resampling follows OpenCV's 8-bit bicubic fixed-point path operation by operation (oracle/input_transform_ref.py; this TU
// is compiled with -ffp-contract=off): float coefficients from the float fractional offset, int16 coefficients at scale 2^11,
// int32 horizontal then vertical sums over border-replicated taps, (v + 2^21) >> 22 saturated to [0, 255].
// HBM-bound and tiny: one thread per output pixel reads a 4 x 4 x 3-byte neighbourhood (down-scaling does not pre-filter) and
// writes 3 floats; B x 3 x 256 x 256 outputs from B x 6.2 MB of frames.
#include "kernels.h"

namespace soccdpt {
namespace {

struct AxisTaps {
    int s;       // first tap index (may be < 0 or > size - 4: taps are clamped one by one)
    int c[4];    // int16-range coefficients, scale 2048
};

__device__ __forceinline__ AxisTaps cubic_axis(int d, double scale) {
    const float A = -0.75f;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    float c[4];
    c[0] = ((A * (f + 1.f) - 5.f * A) * (f + 1.f) + 8.f * A) * (f + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * f - (A + 3.f)) * f * f + 1.f;
    c[2] = ((A + 2.f) * (1.f - f) - (A + 3.f)) * (1.f - f) * (1.f - f) + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
    AxisTaps t;
    t.s = s - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = (int)rintf(c[k] * 2048.f);   // round half to even, like cvRound
        t.c[k] = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
    }
    return t;
}

__global__ __launch_bounds__(256) void input_transform_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int Hs, int Ws,
                                                                  int Hd, int Wd, double scale_x, double scale_y, double m0, double m1, double m2,
                                                                  double s0, double s1, double s2) {
    const size_t total = (size_t)B * Hd * Wd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int dx = (int)(i % Wd);
        const size_t r = i / Wd;
        const int dy = (int)(r % Hd), b = (int)(r / Hd);
        const AxisTaps tx = cubic_axis(dx, scale_x), ty = cubic_axis(dy, scale_y);
        int xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int x = tx.s + j; xs[j] = x < 0 ? 0 : (x > Ws - 1 ? Ws - 1 : x); }
        int acc[3] = {0, 0, 0};
        const uint8_t* base = img + (size_t)b * Hs * Ws * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int y = ty.s + k;
            y = y < 0 ? 0 : (y > Hs - 1 ? Hs - 1 : y);
            const uint8_t* row = base + (size_t)y * Ws * 3;
            int h[3] = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint8_t* p = row + (size_t)xs[j] * 3;
                h[0] += (int)p[0] * tx.c[j];
                h[1] += (int)p[1] * tx.c[j];
                h[2] += (int)p[2] * tx.c[j];
            }
            acc[0] += h[0] * ty.c[k];
            acc[1] += h[1] * ty.c[k];
            acc[2] += h[2] * ty.c[k];
        }
        const double mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v = (acc[c] + (1 << 21)) >> 22;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out[(((size_t)b * 3 + c) * Hd + dy) * Wd + dx] = (float)(((double)v - mean[c]) / sd[c]);   // float64 like numpy, then float32
        }
    }
}

}  // namespace

int launch_input_transform_u8(const uint8_t* img, int B, int Hs, int Ws, int Hd, int Wd, const double* mean, const double* stdv, float* out,
                              hipStream_t st, std::string& err) {
    if (!img || !out || !mean || !stdv) { err = "soccdpt_input_transform_u8: null pointer"; return 1; }
    if (B < 1 || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1) { err = "soccdpt_input_transform_u8: sizes must be positive"; return 1; }
    for (int c = 0; c < 3; ++c)
        if (stdv[c] == 0.0) { err = "soccdpt_input_transform_u8: std must be non-zero"; return 1; }
    // OpenCV: inv_scale = (double)dst / src; scale = 1. / inv_scale
    const double scale_x = 1.0 / ((double)Wd / (double)Ws), scale_y = 1.0 / ((double)Hd / (double)Hs);
    const size_t total = (size_t)B * Hd * Wd;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    SOCCDPT_LAUNCH(input_transform_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, out, B, Hs, Ws, Hd, Wd, scale_x, scale_y, mean[0],
                       mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    return check_launch("input_transform_u8", err);
}

}  // namespace soccdpt
