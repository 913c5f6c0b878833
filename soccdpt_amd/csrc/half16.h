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
// f32 -> IEEE fp16 WITHOUT the clamp: overflow gives +-inf and a NaN stays a NaN.  The fp16 amp mode of the training step stages its scaled
// operands through this, so that an overflow reaches the gradients as a non-finite value and GradScaler skips the step and backs the scale off
// -- torch.cuda.amp semantics (/root/reference/SOccDPT/scripts/train_SOccDPT.py:340,390-393).  A saturating conversion would clip silently.
__device__ __forceinline__ uint16_t f2h_ieee(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
// f32 -> 16-bit operand of a value KNOWN to lie in the format's finite range (softmax probabilities in [0, 1]): f2h without the clamp -- the same bits, one
// v_med3_f32 per element less in the attention kernels' key loops, which are VALU-bound (round 5: 16 of ~180 issue slots per 32 x 32 key tile)
template <bool F16>
__device__ __forceinline__ uint16_t f2h_inrange(float f) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)f);
    else return __builtin_bit_cast(uint16_t, (__bf16)f);
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

// ---- "x3" split-fp16 operand format (SOCCDPT_PREC_F16X3) -------------------------------------------------------------------
// A value a is carried as TWO fp16 numbers, hi = RN16(a) and lo = RN16((a - hi) * 2^11): together 22+ significand bits (the pair
// represents a to 2^-24 relative, like f32 itself; the 2^11 keeps lo in fp16's normal range).  A product is then three fp16 MFMAs
//     a b  ~=  hi_a hi_b  +  2^-11 (hi_a lo_b + lo_a hi_b)            (the dropped lo_a lo_b term is <= 2^-24 |a b|)
// with the hi*hi sum and the cross sum in two f32 accumulators: near-f32 results at 1/3 of the fp16 MFMA rate instead of the f32
// MFMA's 1/16.  Memory layout of an x3 tensor: 4 bytes per element like f32 (same buffer sizes and element offsets); every aligned
// group of 8 elements ("unit" u = e >> 3) is 32 bytes = one 16-byte chunk of 8 hi values + one 16-byte chunk of 8 lo values, hi chunk
// FIRST in even units and SECOND in odd units.  (The alternation makes the MFMA fragment reads of igemm's XOR-swizzled 128-byte LDS
// rows conflict-free: lane quarter q reads unit q, and chunks {0, 3, 4, 7} / {1, 2, 5, 6} fall on distinct bank slots where
// {0, 2, 4, 6} would collide pairwise.)  Rows of an x3 tensor must start at multiples of 16 elements.
struct x3_t { uint32_t v; };   // element tag of the x3 instantiations: 4 bytes per element

// Valid magnitude range: |v| < 65520 (beyond it hi is +-inf, like an IEEE fp16 conversion -- an overflowing or diverging activation / gradient
// reaches the outputs as a non-finite value instead of being clipped to 65504, so parity checks and GradScaler's found_inf see it); the pair keeps
// >= 22 significand bits down to |v| ~ 2^-3 * 2^-11 (lo normal), degrades gradually below that and is hi alone (11 bits, then fp16 subnormals
// of 2^-24 absolute) below |v| ~ 6e-8.  There is no per-tensor exponent shift: callers whose values live below ~1e-4 (gradients under a large
// batch mean) scale by a power of two first (the training step's loss scale, soccdpt_hip.h).
__device__ __forceinline__ void x3_split(float v, _Float16& hi, _Float16& lo) {
    // No contraction across this function's boundary: callers pass products (`acc * inv`), and whether `v - hi` became fma(acc, inv, -hi) -- the
    // exact product, a different lo -- used to depend on the code around the inlined call: the same attention body gave different lo words as a
    // stand-alone kernel and inside the persistent stage kernel (round 4; 3 % of the words of one tensor).
#pragma clang fp contract(off)
    // ... and no single-rounding shortcut either: with v = a * b in the caller the compiler may emit v_fma_mixlo_f16 (the EXACT product rounded once to
    // fp16) for the conversion below, which differs from fp16(f32(a * b)) whenever the f32 product sits on an fp16 tie -- observed as hi words one ulp
    // apart (and lo = +-0.5 ulp) between two instantiations of the same body.  The empty asm pins v to its f32 value first.
    asm volatile("" : "+v"(v));
    hi = (_Float16)v;                          // round to nearest even; overflow -> +-inf, NaN stays NaN
    const float hf = (float)hi;
    float r = (v - hf) * 2048.f;               // exact in f32 for finite hi
    r = (hf - hf == 0.f) ? r : 0.f;            // hi non-finite: no correction term (inf - inf / NaN would poison lo with a second NaN source)
    lo = (_Float16)__builtin_amdgcn_fmed3f(r, -65504.f, 65504.f);
}
// byte offsets of element e's hi / lo halves from the tensor base
__device__ __forceinline__ size_t x3_hi_off(size_t e) { return (e >> 3) * 32 + (((e >> 3) & 1) ? 16 : 0) + (e & 7) * 2; }
__device__ __forceinline__ size_t x3_lo_off(size_t e) { return (e >> 3) * 32 + (((e >> 3) & 1) ? 0 : 16) + (e & 7) * 2; }
// store 4 consecutive elements e .. e+3 (e % 4 == 0): two 8-byte stores
__device__ __forceinline__ void x3_store4(void* base, size_t e, float a, float b, float c, float d) {
    _Float16 h[4], l[4];
    x3_split(a, h[0], l[0]); x3_split(b, h[1], l[1]); x3_split(c, h[2], l[2]); x3_split(d, h[3], l[3]);
    uint2 ph, pl;
    ph.x = (uint32_t)__builtin_bit_cast(uint16_t, h[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[1]) << 16);
    ph.y = (uint32_t)__builtin_bit_cast(uint16_t, h[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[3]) << 16);
    pl.x = (uint32_t)__builtin_bit_cast(uint16_t, l[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, l[1]) << 16);
    pl.y = (uint32_t)__builtin_bit_cast(uint16_t, l[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, l[3]) << 16);
    char* p = static_cast<char*>(base);
    *reinterpret_cast<uint2*>(p + x3_hi_off(e)) = ph;
    *reinterpret_cast<uint2*>(p + x3_lo_off(e)) = pl;
}
// store one whole unit, elements e .. e+7 (e % 8 == 0): two 16-byte stores
__device__ __forceinline__ void x3_store8(void* base, size_t e, const float (&v)[8]) {
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        _Float16 h0, l0, h1, l1;
        x3_split(v[2 * k], h0, l0);
        x3_split(v[2 * k + 1], h1, l1);
        ph[k] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
        pl[k] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
    }
    char* p = static_cast<char*>(base);
    *reinterpret_cast<uint4*>(p + x3_hi_off(e)) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    *reinterpret_cast<uint4*>(p + x3_lo_off(e)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}
__device__ __forceinline__ void x3_store1(void* base, size_t e, float a) {
    _Float16 h, l;
    x3_split(a, h, l);
    char* p = static_cast<char*>(base);
    *reinterpret_cast<_Float16*>(p + x3_hi_off(e)) = h;
    *reinterpret_cast<_Float16*>(p + x3_lo_off(e)) = l;
}
__device__ __forceinline__ float x3_load1(const void* base, size_t e) {
    const char* p = static_cast<const char*>(base);
    return (float)*reinterpret_cast<const _Float16*>(p + x3_hi_off(e)) + (float)*reinterpret_cast<const _Float16*>(p + x3_lo_off(e)) * (1.0f / 2048.f);
}

}  // namespace soccdpt
