// GELU(erf) of timm's Mlp (act_layer=nn.GELU), shared by the igemm epilogue and the fused MLP kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace soccdpt {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute): 1 rcp + 1 exp + 7 fma.  Used on the bf16 path,
// whose outputs are rounded to 8 mantissa bits anyway; the exact-f32 path keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);  // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

}  // namespace soccdpt
