// 16-bit operand formats of the MFMA paths.  Every kernel that reads or writes 16-bit activations / weights is templated on
// `bool F16`: false = bf16 (8-bit significand, SOCCDPT_PREC_BF16), true = IEEE fp16 (11-bit significand, SOCCDPT_PREC_F16;
// what the reference's own `optimize=True` path computes in: /root/reference/SOccDPT/model/loader.py:126-139 `.half()`).
// Both feed MFMA at the same rate (v_mfma_f32_*_bf16 / v_mfma_f32_*_f16) and accumulate in f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace soccdpt {

typedef __attribute__((ext_vector_type(8))) short h16x8;  // 8 raw 16-bit operands (one 16-byte LDS chunk / MFMA fragment)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// f32 -> 16-bit, round to nearest even; fp16 saturates at +-65504 instead of producing inf
template <bool F16>
__device__ __forceinline__ uint16_t f2h(float f) {
    if constexpr (F16) {
        const _Float16 x = (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);
        return __builtin_bit_cast(uint16_t, x);
    } else {
        const __bf16 x = (__bf16)f;
        return __builtin_bit_cast(uint16_t, x);
    }
}
template <bool F16>
__device__ __forceinline__ float h2f(uint16_t b) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, b);
    else return __builtin_bit_cast(float, (uint32_t)b << 16);
}
template <bool F16>
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    return (uint32_t)f2h<F16>(a) | ((uint32_t)f2h<F16>(b) << 16);
}
// the two halves of a packed pair
template <bool F16>
__device__ __forceinline__ float h_lo(uint32_t u) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
    else return __builtin_bit_cast(float, u << 16);
}
template <bool F16>
__device__ __forceinline__ float h_hi(uint32_t u) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
    else return __builtin_bit_cast(float, u & 0xffff0000u);
}

template <bool F16>
__device__ __forceinline__ f32x4_t mfma_16x16x32(h16x8 a, h16x8 b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16_t mfma_32x32x16(h16x8 a, h16x8 b, f32x16_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

}  // namespace soccdpt
