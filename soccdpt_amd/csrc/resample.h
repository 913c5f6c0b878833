// torch-CPU resampling arithmetic shared by projection.hip (bit-exact contract, compiled with -ffp-contract=off) and loss.hip:
// upsample_bicubic2d (align_corners=False, A = -0.75) source index / coefficient formulas exactly as pinned in DESIGN.md §2,
// and upsample_nearest2d's source index.  (ATen UpSample.h / UpSampleKernel.cpp; call sites
// /root/reference/SOccDPT/model/SOccDPT.py:270-282.)
#pragma once
#include <hip/hip_runtime.h>

namespace soccdpt {

__device__ __forceinline__ float cc1(float x) {
    float t = fmaf(1.25f, x, -2.25f);
    return (t * x) * x + 1.0f;
}
__device__ __forceinline__ float cc2(float x) {
    float t = fmaf(-0.75f, x, 3.75f);
    t = fmaf(t, x, -6.0f);
    return t * x + 3.0f;
}

struct Taps {
    int idx[4];
    float w[4];
};

__device__ __forceinline__ Taps cubic_taps(int dst, int in, float scale) {
    Taps tp;
    float real = fmaf(scale, (float)dst + 0.5f, -0.5f);
    int ii = (int)floorf(real);
    if (ii > in - 1) ii = in - 1;
    float t = real - (float)ii;
    t = t < 0.0f ? 0.0f : t;
    t = t > 1.0f ? 1.0f : t;
    tp.w[0] = cc2(t + 1.0f);
    tp.w[1] = cc1(t);
    float u = 1.0f - t;
    tp.w[2] = cc1(u);
    tp.w[3] = cc2(u + 1.0f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int k = ii + j - 1;
        k = k < 0 ? 0 : k;
        k = k > in - 1 ? in - 1 : k;
        tp.idx[j] = k;
    }
    return tp;
}

__device__ __forceinline__ float dot4(float v0, float v1, float v2, float v3, const float* w) {
    return fmaf(v3, w[3], fmaf(v2, w[2], fmaf(v0, w[0], v1 * w[1])));
}

__device__ __forceinline__ int nearest_src(int dst, int in, float scale) {
    int s = (int)floorf((float)dst * scale);
    return s > in - 1 ? in - 1 : s;
}

}  // namespace soccdpt
