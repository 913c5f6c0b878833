// Kernels of the in-product precision-map calibration (calibrate.cpp, soccdpt_prec_calibrate): decode a named intermediate of the forward
// from whatever operand format it is kept in (f32 / bf16 / fp16 / x3 split fp16, plain or zero-halo NHWC) into compact f32 NHWC, and the two
// sums of a relative L2 error, sum (a - b)^2 and sum b^2, in float64 with a fixed reduction order (the same inputs give the same map).
// Also the weight fingerprint that tells the shipped map's own weights from any other checkpoint.
#include "calibrate.h"

namespace soccdpt {
namespace {

// kind codes of model_workspace_tensor: 0 f32, 1 bf16, 4 fp16, 6 x3 (plain NHWC); 3 f32, 2 bf16, 5 fp16, 7 x3 (zero-halo NHWC [B][H+2][W+2][C])
__device__ __forceinline__ float load_elem(const void* src, int kind, size_t e) {
    switch (kind) {
        case 0: case 3: return static_cast<const float*>(src)[e];
        case 1: case 2: return __uint_as_float((uint32_t)static_cast<const uint16_t*>(src)[e] << 16);
        case 4: case 5: return (float)static_cast<const _Float16*>(src)[e];
        default: {   // x3: units of 8 elements = 16 bytes of hi + 16 bytes of lo, hi first in even units (half16.h)
            const size_t u = e >> 3;
            const int i = (int)(e & 7), odd = (int)(u & 1);
            const _Float16* p = static_cast<const _Float16*>(src) + u * 16;
            const float hi = (float)p[(odd ? 8 : 0) + i], lo = (float)p[(odd ? 0 : 8) + i];
            return hi + lo * (1.0f / 2048.f);
        }
    }
}

__global__ __launch_bounds__(256) void calib_decode_kernel(const void* __restrict__ src, int kind, int B, int H, int W, int C, float* __restrict__ dst) {
    const size_t n = (size_t)B * H * W * C;
    const bool halo = kind == 2 || kind == 3 || kind == 5 || kind == 7;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t e = i;
        if (halo) {
            const size_t c = i % C, p = i / C;
            const size_t x = p % W, y = (p / W) % H, b = p / ((size_t)W * H);
            e = ((b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c;
        }
        dst[i] = load_elem(src, kind, e);
    }
}

// partial[block] = {sum (a - b)^2, sum b^2} over the block's grid-stride elements, f64; fixed tree order inside the block
__global__ __launch_bounds__(256) void calib_sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, double* __restrict__ partial) {
    __shared__ double sh[2][256];
    double d = 0.0, r = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double x = (double)a[i], y = (double)b[i];
        d += (x - y) * (x - y);
        r += y * y;
    }
    sh[0][threadIdx.x] = d; sh[1][threadIdx.x] = r;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}

__global__ void calib_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double d = 0.0, r = 0.0;
        for (int i = 0; i < nblocks; ++i) { d += partial[2 * i]; r += partial[2 * i + 1]; }
        out2[0] = d; out2[1] = r;
    }
}

// order-independent 64-bit sum of the f32 bit patterns of one tensor, scaled by `mult` (one atomic per block)
__global__ __launch_bounds__(256) void calib_fingerprint_kernel(const uint32_t* __restrict__ w, size_t n, unsigned long long mult, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sh[256];
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)w[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, sh[0] * mult);
}

}  // namespace

int launch_calib_decode(const void* src, int kind, int B, int H, int W, int C, float* dst, hipStream_t st, std::string& err) {
    if (kind < 0 || kind > 7) { err = "calibrate: unknown tensor kind"; return 1; }
    const size_t n = (size_t)B * H * W * C;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    SOCCDPT_LAUNCH(calib_decode_kernel, dim3(blocks), dim3(256), 0, st, src, kind, B, H, W, C, dst);
    return 0;
}

int launch_calib_sqdiff(const float* a, const float* b, size_t n, double* partial, double* out2, hipStream_t st, std::string& err) {
    (void)err;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)kCalibPartialBlocks);
    SOCCDPT_LAUNCH(calib_sqdiff_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, n, partial);
    SOCCDPT_LAUNCH(calib_finish_kernel, dim3(1), dim3(64), 0, st, partial, blocks, out2);
    return 0;
}

int launch_calib_fingerprint(const float* w, size_t n, unsigned long long mult, unsigned long long* out, hipStream_t st, std::string& err) {
    (void)err;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256);
    SOCCDPT_LAUNCH(calib_fingerprint_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(w), n, mult, out);
    return 0;
}

}  // namespace soccdpt
