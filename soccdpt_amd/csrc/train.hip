// Backward (and train-mode forward) kernels of the SOccDPT_V3 training step for gfx950: exact f32, or 16-bit operand copies for the gradient
// GEMMs under amp (soccdpt_train_set_amp).
//
// What autograd does for the reference in scripts/train_SOccDPT.py:390-393 (`grad_scaler.scale(loss).backward()`) over
// model/SOccDPT.py:660-685, model/dpt.py:142-232, model/blocks.py:391-497 and timm's SwinTransformerV2 blocks.  GEMM-shaped gradients
// (dgrad of every Linear / convolution, wgrad of every weight) run on the f32 MFMA igemm (igemm.hip) over operands these kernels lay out:
//   dX = dY W            -> igemm(X = dY, Wt = W^T)                      (transpose_kernel; conv: flipped tap-major weights + zero-haloed dY)
//   dW = dY^T X          -> igemm(X = dY^T [N][M], Wt = X^T [K][M])      (transpose_kernel; 3x3 conv: both operands in halo pixel order and nine
//                                                                          shifted views of ONE transposed halo image, dy_halo_T_kernel; im2colT_kernel
//                                                                          only where a 64-row weight tile would straddle two taps, C % 64 != 0)
// Everything else here is an elementwise / row / column-reduction kernel, deterministic (no float atomics): fixed-order tree reductions.
#include <type_traits>
#include "half16.h"
#include "kernels.h"
#include "train.h"

namespace soccdpt {
namespace {

constexpr float LN100 = 4.605170185988092f;

// staging kernels write f32 or, in the mixed-precision mode (SOCCDPT train amp: bf16 MFMA operands for the gradient GEMMs), bf16
template <typename OT> __device__ __forceinline__ OT cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t cvt_out<uint16_t>(float v) { return f2h<false>(v); }
struct f16raw { uint16_t v; };   // IEEE fp16 bits (the amp mode with loss scaling); uint16_t alone means bf16
template <> __device__ __forceinline__ f16raw cvt_out<f16raw>(float v) { return f16raw{f2h_ieee(v)}; }   // overflow -> inf: GradScaler must see it (half16.h)
struct x3raw { uint32_t v; };    // element of an x3 split-fp16 operand tensor (half16.h): 4 bytes, written as an fp16 hi / lo pair
template <typename OT> __device__ __forceinline__ void store_out(OT* out, size_t idx, float v) { out[idx] = cvt_out<OT>(v); }
template <> __device__ __forceinline__ void store_out<x3raw>(x3raw* out, size_t idx, float v) { x3_store1(out, idx, v); }

// ---------------- layout ----------------
// [R][C] -> [C][Rp] (Rp >= R: rows padded with zeros up to the k-tile multiple the igemm needs), 32 x 32 tiles through LDS
template <typename OT>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, OT* __restrict__ out, int R, int C, int Rp) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: column block of `in`, by: row block
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        t[i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, r = by + tx;
        if (c < C && r < Rp) store_out<OT>(out, (size_t)c * Rp + r, t[tx][i]);
    }
}

// zero-haloed NHWC image [B][H+2][W+2][C] -> im2col^T [(tap*C + c)][m], m = (b, y, x): the Wt operand of the 3x3 wgrad GEMM
template <typename OT>
__global__ __launch_bounds__(256) void im2colT_kernel(const float* __restrict__ halo, OT* __restrict__ out, int B, int H, int W, int C, size_t Mp) {
    __shared__ float t[32][33];
    const size_t M = (size_t)B * H * W;
    const int tap = blockIdx.z, ky = tap / 3, kx = tap % 3;
    const size_t m0 = (size_t)blockIdx.y * 32;
    const int c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const size_t m = m0 + i;
        float v = 0.f;
        if (m < M && c0 + tx < C) {
            const int b = (int)(m / ((size_t)H * W)), r = (int)(m - (size_t)b * H * W), y = r / W, x = r - y * W;
            v = halo[(((size_t)b * (H + 2) + y + ky) * (W + 2) + x + kx) * C + c0 + tx];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const size_t m = m0 + tx;
        if (c < C && m < Mp) store_out<OT>(out, ((size_t)tap * C + c) * Mp + m, t[tx][i]);
    }
}

// Weight gradient of a 3x3 convolution WITHOUT im2col: with both operands transposed in HALO pixel order (mh over B x (r+2) x (r+2), the
// gradient zero on the border pixels), tap (ky, kx) is the plain GEMM  dW_t[n][c] = sum_mh dYh^T[n][mh] * Xh^T[c][mh + off_t],
// off_t = (ky - 1)(r + 2) + (kx - 1): nine pointer offsets into ONE transposed halo image (train_step.cpp: conv3_bwd).  The igemm's 16-byte
// LDS-DMA loads accept the 4-byte-aligned bases this produces (tests/tools/unaligned_operand_probe.py: same results, same speed).
// dY [B*r*r][N] plain -> out [N][ld]: column margin + mh holds the pixel's gradient (zero on border pixels and in the margins)
// rpp: row pitch of the pixel order in the transposed image (>= r + 2).  The f32 / 16-bit paths use rpp = r + 2; the x3 path pads every halo row
// to a multiple of 16 pixels so that the vertical tap shifts (+- rpp elements) keep the 8-element units of an x3 tensor intact.
template <typename OT>
__global__ __launch_bounds__(256) void dy_halo_T_kernel(const float* __restrict__ dy, OT* __restrict__ out, int B, int r, int N, int margin, int ld, int rpp) {
    __shared__ float t[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int rp = r + 2, Mh = B * rp * rpp;
    for (int i = ty; i < 32; i += 8) {
        const int mh = k0 + i - margin, n = n0 + tx;
        float v = 0.f;
        if (mh >= 0 && mh < Mh && n < N) {
            const int b = mh / (rp * rpp), q = mh - b * rp * rpp, yh = q / rpp, xh = q - yh * rpp;
            if (yh >= 1 && yh <= r && xh >= 1 && xh <= r) v = dy[((size_t)(b * r + yh - 1) * r + xh - 1) * N + n];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, k = k0 + tx;
        if (n < N && k < ld) store_out<OT>(out, (size_t)n * ld + k, t[tx][i]);
    }
}
// The other operand in the same pitched pixel order: halo image [B][r+2][r+2][C] -> out [C][ld], column col0 + (b (r+2) + yh) rpp + xh holds
// halo[b][yh][xh][c]; pad columns (xh >= r + 2), the margins and everything beyond the last pixel are zero.  col0 = margin - (kx - 1) makes the
// copy read at an ALIGNED column deliver the horizontal tap kx (three copies; the vertical taps are +- rpp, a multiple of 16).
template <typename OT>
__global__ __launch_bounds__(256) void x_halo_T_kernel(const float* __restrict__ halo, OT* __restrict__ out, int B, int r, int C, int col0, int ld, int rpp) {
    __shared__ float t[32][33];
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int rp = r + 2, Mh = B * rp * rpp;
    for (int i = ty; i < 32; i += 8) {
        const int mh = k0 + i - col0, c = c0 + tx;
        float v = 0.f;
        if (mh >= 0 && mh < Mh && c < C) {
            const int row = mh / rpp, xh = mh - row * rpp;     // row = b (r + 2) + yh
            if (xh < rp) v = halo[((size_t)row * rp + xh) * C + c];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (c < C && k < ld) store_out<OT>(out, (size_t)c * ld + k, t[tx][i]);
    }
}
// dW slabs [9][N][C] -> parameter layout [N][C][3][3]
__global__ void wgrad_permute9_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C) {
    const size_t n = (size_t)N * C * 9;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        const size_t r = i / 9;    // n * C + c
        out[i] = in[(size_t)tap * N * C + r];
    }
}

// W [N][C][3][3] -> Wd [C][2-ky][2-kx][N] (tap-major, flipped): Wt operand of the 3x3 dgrad (a convolution of dY with the rotated filter)
template <typename OT>
__global__ void conv_w_dgrad_kernel(const float* __restrict__ w, OT* __restrict__ out, int N, int C) {
    const size_t n = (size_t)N * C * 9;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int nn = (int)(i % N);
        size_t r = i / N;
        const int tap = (int)(r % 9), c = (int)(r / 9);
        store_out<OT>(out, i, w[((size_t)nn * C + c) * 9 + (8 - tap)]);
    }
}

// many weights, one launch (TrBatchTable): block -> entry by the entries' first-block numbers, then the body of transpose_kernel (Rp = R) or
// conv_w_dgrad_kernel (1024 elements per block)
template <typename OT>
__global__ __launch_bounds__(256) void weight_batch_kernel(const TrBatchTable t) {
    __shared__ float tl[32][33];
    int i = 0;
    while (i + 1 < t.n && (int)blockIdx.x >= t.e[i + 1].tile0) ++i;
    const float* __restrict__ in = t.e[i].src;
    OT* __restrict__ out = static_cast<OT*>(t.e[i].dst);
    const int R = t.e[i].R, C = t.e[i].C, local = (int)blockIdx.x - t.e[i].tile0;
    if (t.e[i].kind == 0) {
        const int tiles_x = (C + 31) / 32;
        const int bx = (local % tiles_x) * 32, by = (local / tiles_x) * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int k = ty; k < 32; k += 8) {
            const int r = by + k, c = bx + tx;
            tl[k][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
        }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            const int c = bx + k, r = by + tx;
            if (c < C && r < R) store_out<OT>(out, (size_t)c * R + r, tl[tx][k]);
        }
    } else {
        const size_t n = (size_t)R * C * 9;
        for (size_t q = (size_t)local * 1024 + threadIdx.x; q < n && q < (size_t)(local + 1) * 1024; q += 256) {
            const int nn = (int)(q % R);
            const size_t r = q / R;
            const int tap = (int)(r % 9), c = (int)(r / 9);
            store_out<OT>(out, q, in[((size_t)nn * C + c) * 9 + (8 - tap)]);
        }
    }
}

// dW tap-major [N][9][C] -> parameter layout [N][C][3][3]
__global__ void wgrad_permute_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C) {
    const size_t n = (size_t)N * C * 9;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        size_t r = i / 9;
        const int c = (int)(r % C), nn = (int)(r / C);
        out[i] = in[((size_t)nn * 9 + tap) * C + c];
    }
}

// plain [B][H][W][C] <-> halo [B][H+2][W+2][C]
template <typename OT>
__global__ void to_halo_kernel(const float* __restrict__ in, OT* __restrict__ out, int B, int H, int W, int C) {
    const size_t n = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H), b = (int)(r / H);
        store_out<OT>(out, (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c, in[i]);
    }
}
// The same over the WHOLE halo image, 8 channels per thread (C % 8 == 0): border pixels are written as zeros, so the caller needs no memset of
// the buffer, and every load / store is 16 or 32 bytes wide.
template <typename OT>
__global__ void to_halo_full8_kernel(const float* __restrict__ in, OT* __restrict__ out, int B, int H, int W, int C) {
    const int C8 = C >> 3;
    const size_t n = (size_t)B * (H + 2) * (W + 2) * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        size_t r = i / C8;
        const int x = (int)(r % (W + 2)) - 1;
        r /= (W + 2);
        const int y = (int)(r % (H + 2)) - 1, b = (int)(r / (H + 2));
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const float* p = in + (((size_t)b * H + y) * W + x) * C + c;
            v0 = *reinterpret_cast<const float4*>(p);
            v1 = *reinterpret_cast<const float4*>(p + 4);
        }
        const size_t o = i * 8;
        if constexpr (sizeof(OT) == 4 && !std::is_same<OT, x3raw>::value) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = v0;
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o + 4) = v1;
        } else if constexpr (std::is_same<OT, x3raw>::value) {
            x3_store4(out, o, v0.x, v0.y, v0.z, v0.w);
            x3_store4(out, o + 4, v1.x, v1.y, v1.z, v1.w);
        } else {
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) out[o + k] = cvt_out<OT>(f[k]);
        }
    }
}
// out[i] (+)= halo interior
__global__ void from_halo_kernel(const float* __restrict__ halo, float* __restrict__ out, int B, int H, int W, int C, int accumulate) {
    const size_t n = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H), b = (int)(r / H);
        const float v = halo[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c];
        out[i] = accumulate ? out[i] + v : v;
    }
}

// ---------------- reductions ----------------
// column sums of [M][N] (bias gradients, LayerNorm gamma / beta gradients): out[n] = sum_m a[m][n] (* b[m][n] when b != nullptr).
// Two deterministic stages: a block = 64 columns x 4 row phases over one chunk of rows (partials per chunk), then the chunks in order.
// TWO: a second output column set, the plain sum of a (LayerNorm: gamma and beta gradients from one pass over dout); part = [chunks][2N] then.
// Four independent row streams per thread keep four loads in flight; the order of additions per output does not depend on TWO.
template <bool TWO>
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part, size_t M, int N, int chunks) {
    __shared__ float red[TWO ? 2 : 1][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    const int ch = blockIdx.y;
    const size_t per = (M + chunks - 1) / chunks, lo = (size_t)ch * per, hi = lo + per < M ? lo + per : M;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, t[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        size_t m = lo + ty;
        for (; m + 12 < hi; m += 16) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = a[(m + 4 * u) * N + n]; bv[u] = b ? b[(m + 4 * u) * N + n] : 1.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[u] += b ? av[u] * bv[u] : av[u]; if (TWO) t[u] += av[u]; }
        }
        for (; m < hi; m += 4) { const float av = a[m * N + n]; s[0] += b ? av * b[m * N + n] : av; if (TWO) t[0] += av; }
    }
    red[0][ty][tx] = (s[0] + s[1]) + (s[2] + s[3]);
    if (TWO) red[TWO ? 1 : 0][ty][tx] = (t[0] + t[1]) + (t[2] + t[3]);
    __syncthreads();
    if (ty == 0 && n < N) {
        const int NN = TWO ? 2 * N : N;
        part[(size_t)ch * NN + n] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        if (TWO) part[(size_t)ch * NN + N + n] = (red[TWO ? 1 : 0][0][tx] + red[TWO ? 1 : 0][1][tx]) + (red[TWO ? 1 : 0][2][tx] + red[TWO ? 1 : 0][3][tx]);
    }
}
// second stage: the chunk partials [chunks][NN] summed in chunk order, 64 columns x 4 phases per block (phase p takes chunks p, p+4, ...); columns
// >= N go to out2 (the TWO form above)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out2, int N, int NN, int chunks, int accumulate) {
    __shared__ float red[16][64];   // 16 phases: a launch has only NN / 64 workgroups, so the chain of dependent loads per thread is what it costs
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (n < NN)
        for (int c = ty; c < chunks; c += 16) s += part[(size_t)c * NN + n];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < NN) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) v += (red[i][tx] + red[i + 1][tx]) + (red[i + 2][tx] + red[i + 3][tx]);
        float* o = n < N ? out + n : out2 + (n - N);
        *o = accumulate ? *o + v : v;
    }
}
// sum over the leading axis of [R][n] with n large (attention logit gradients summed over windows): one thread per column
__global__ void rowsum_kernel(const float* a, float* out, int R, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += a[(size_t)r * n + i];
        out[i] = s;
    }
}

// f32 -> x3 of TWO tensors in one launch (the X and W operands of a forward GEMM, train_step.cpp gemm_fwd): the conversions are a few microseconds each,
// so the second launch cost as much as its work
__global__ void cvt_x3_pair_kernel(const float* __restrict__ in0, void* __restrict__ out0, size_t n0, const float* __restrict__ in1, void* __restrict__ out1,
                                   size_t n1) {   // n0, n1 % 4 == 0
    const size_t total = n0 + n1;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * blockDim.x * 4) {
        if (i < n0) {
            const float4 v = *reinterpret_cast<const float4*>(in0 + i);
            x3_store4(out0, i, v.x, v.y, v.z, v.w);
        } else {
            const size_t j = i - n0;
            const float4 v = *reinterpret_cast<const float4*>(in1 + j);
            x3_store4(out1, j, v.x, v.y, v.z, v.w);
        }
    }
}

// f32 -> 16 bit (bf16, or IEEE fp16 without saturation: the scaled operands of the fp16 amp mode) of TWO tensors in one launch, four elements per thread:
// the dY and X operands of a Linear layer's two gradient GEMMs (train_step.cpp linear_bwd).  Round 4 converted them with two launches of a one-element-per-
// thread kernel (129 launches, 0.81 ms per bf16-amp step).
template <bool F16>
__global__ void cvt16_pair_kernel(const float* __restrict__ in0, uint16_t* __restrict__ out0, size_t n0, const float* __restrict__ in1, uint16_t* __restrict__ out1,
                                  size_t n1) {   // n0, n1 % 4 == 0
    const size_t total = n0 + n1;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * blockDim.x * 4) {
        const bool first = i < n0;
        const size_t j = first ? i : i - n0;
        const float4 v = *reinterpret_cast<const float4*>((first ? in0 : in1) + j);
        uint2 p;
        if constexpr (F16) {
            p.x = (uint32_t)f2h_ieee(v.x) | ((uint32_t)f2h_ieee(v.y) << 16);
            p.y = (uint32_t)f2h_ieee(v.z) | ((uint32_t)f2h_ieee(v.w) << 16);
        } else {
            p.x = pack_h2<false>(v.x, v.y);
            p.y = pack_h2<false>(v.z, v.w);
        }
        *reinterpret_cast<uint2*>((first ? out0 : out1) + j) = p;
    }
}

// ---------------- elementwise ----------------
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {   // y += x
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] += x[i];
}
// dX = dY * (ref > 0)  (+ add), ref = the forward value whose ReLU was taken (plain layout)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref, const float* __restrict__ add, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = ref[i] > 0.f ? dy[i] : 0.f;
        if (add) v += add[i];
        dx[i] = v;
    }
}
// same with `ref` in zero-halo layout [B][H+2][W+2][C] (the saved post-ReLU operand images of the decoder)
__global__ void relu_bwd_halo_kernel(const float* __restrict__ dy, const float* __restrict__ ref_halo, const float* __restrict__ add, float* __restrict__ dx, int B, int H, int W,
                                     int C) {
    const size_t n = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H), b = (int)(r / H);
        float v = ref_halo[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c] > 0.f ? dy[i] : 0.f;
        if (add) v += add[i];
        dx[i] = v;
    }
}
// d/dx gelu(x) (erf form), dX = dY * gelu'(pre)
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = pre[i];
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        dx[i] = dy[i] * (cdf + x * pdf);
    }
}

// LayerNorm backward, one wave per row: out = LN(y) * g + b.  dy = rstd * (g*dout - mean(g*dout) - xhat * mean(g*dout*xhat)).
// Also writes xhat (for the gamma gradient: colsum(dout * xhat)); dy may alias dout.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g, const float* dout, float* dy, float* __restrict__ xhat_out,
                                                     int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* yr = y + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += yr[c];
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = yr[c] - mean; q += d * d; }
    for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    float a = 0.f, bsum = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (yr[c] - mean) * rstd, gd = g[c] * dout[(size_t)row * C + c];
        a += gd;
        bsum += gd * xh;
    }
    for (int o = 1; o < 64; o <<= 1) { a += __shfl_xor(a, o); bsum += __shfl_xor(bsum, o); }
    a /= (float)C;
    bsum /= (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (yr[c] - mean) * rstd, gd = g[c] * dout[(size_t)row * C + c];
        if (xhat_out) xhat_out[(size_t)row * C + c] = xh;
        dy[(size_t)row * C + c] = rstd * (gd - a - xh * bsum);
    }
}

// bilinear x(H/h) align_corners=True backward, NHWC: dlo[b][y][x][c] = sum over the high-res pixels that sample (y, x) of weight * dhi.
// Gather form (deterministic): a low-res row y is touched by high-res rows Y with floor(Y*sy) in {y-1, y}.
__global__ void bilinear_bwd_kernel(const float* __restrict__ dhi, float* __restrict__ dlo, int B, int h, int w, int H, int W, int C, int accumulate) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const size_t n = (size_t)B * h * w * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h), b = (int)(r / h);
        // candidate high-res rows: Y*sy in (y-1, y+1)
        int Y0 = sy > 0.f ? (int)ceilf(((float)y - 1.0f) / sy) : 0, Y1 = sy > 0.f ? (int)floorf(((float)y + 1.0f) / sy) : H - 1;
        int X0 = sx > 0.f ? (int)ceilf(((float)x - 1.0f) / sx) : 0, X1 = sx > 0.f ? (int)floorf(((float)x + 1.0f) / sx) : W - 1;
        Y0 = Y0 < 0 ? 0 : Y0; X0 = X0 < 0 ? 0 : X0; Y1 = Y1 > H - 1 ? H - 1 : Y1; X1 = X1 > W - 1 ? W - 1 : X1;
        float acc = 0.f;
        for (int Y = Y0; Y <= Y1; ++Y) {
            const float fy = sy * (float)Y;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1);
            const float ly = fy - (float)y0;
            float wy = 0.f;
            if (y0 == y) wy += 1.f - ly;
            if (y1 == y) wy += ly;
            if (wy == 0.f) continue;
            for (int X = X0; X <= X1; ++X) {
                const float fx = sx * (float)X;
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1);
                const float lx = fx - (float)x0;
                float wx = 0.f;
                if (x0 == x) wx += 1.f - lx;
                if (x1 == x) wx += lx;
                if (wx != 0.f) acc += wy * wx * dhi[(((size_t)b * H + Y) * W + X) * C + c];
            }
        }
        dlo[i] = accumulate ? dlo[i] + acc : acc;
    }
}
// four channels per thread (C % 4 == 0): the same taps, 16-byte loads
__global__ void bilinear_bwd4_kernel(const float* __restrict__ dhi, float* __restrict__ dlo, int B, int h, int w, int H, int W, int C, int accumulate) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int C4 = C >> 2;
    const size_t n = (size_t)B * h * w * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        size_t r = i / C4;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h), b = (int)(r / h);
        // candidate high-res rows: Y*sy in (y-1, y+1)
        int Y0 = sy > 0.f ? (int)ceilf(((float)y - 1.0f) / sy) : 0, Y1 = sy > 0.f ? (int)floorf(((float)y + 1.0f) / sy) : H - 1;
        int X0 = sx > 0.f ? (int)ceilf(((float)x - 1.0f) / sx) : 0, X1 = sx > 0.f ? (int)floorf(((float)x + 1.0f) / sx) : W - 1;
        Y0 = Y0 < 0 ? 0 : Y0; X0 = X0 < 0 ? 0 : X0; Y1 = Y1 > H - 1 ? H - 1 : Y1; X1 = X1 > W - 1 ? W - 1 : X1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int Y = Y0; Y <= Y1; ++Y) {
            const float fy = sy * (float)Y;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1);
            const float ly = fy - (float)y0;
            float wy = 0.f;
            if (y0 == y) wy += 1.f - ly;
            if (y1 == y) wy += ly;
            if (wy == 0.f) continue;
            for (int X = X0; X <= X1; ++X) {
                const float fx = sx * (float)X;
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1);
                const float lx = fx - (float)x0;
                float wx = 0.f;
                if (x0 == x) wx += 1.f - lx;
                if (x1 == x) wx += lx;
                if (wx != 0.f) {
                    const float4 v = *reinterpret_cast<const float4*>(dhi + (((size_t)b * H + Y) * W + X) * C + c);
                    const float ww = wy * wx;
                    acc.x += ww * v.x; acc.y += ww * v.y; acc.z += ww * v.z; acc.w += ww * v.w;
                }
            }
        }
        float4* o = reinterpret_cast<float4*>(dlo + i * 4);
        if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
        *o = acc;
    }
}

// ---------------- seg head (train mode) ----------------
// BatchNorm2d with batch statistics over [M][C] (NHWC rows): stats[c] = {mean, invstd}; y = relu(gamma * xhat + beta) * keep / (1 - p);
// running_mean / running_var updated like nn.BatchNorm2d(momentum 0.1, unbiased variance).  Deterministic: colsum kernels feed bn_stats.
// Two passes in float64 (mean, then the centred second moment): the normalised values feed a ReLU, and a mean that is off by 1e-6 of
// the spread flips the mask of the activations nearest zero -- each flip is an O(1) local gradient difference against autograd.
// pass: 0 -> part[ch][c] = sum x;  1 -> part[ch][c] = sum (x - mean[c])^2   (block = 64 channels x 4 row phases, `chunks` row chunks)
__global__ __launch_bounds__(256) void bn_moment_part_kernel(const float* __restrict__ x, const double* __restrict__ mean, double* __restrict__ part, size_t M, int C,
                                                             int chunks, int pass) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, ch = blockIdx.y;
    const size_t per = (M + chunks - 1) / chunks, lo = (size_t)ch * per, hi = lo + per < M ? lo + per : M;
    double s = 0.0;
    if (c < C) {
        const double mu = pass ? mean[c] : 0.0;
        for (size_t m = lo + ty; m < hi; m += 4) {
            const double v = (double)x[m * C + c] - mu;
            s += pass ? v * v : v;
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) part[(size_t)ch * C + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// pass 0: mean[c];  pass 1: stats[c] = {mean, invstd} and the running buffers (momentum, unbiased variance) like nn.BatchNorm2d
__global__ void bn_moment_final_kernel(const double* __restrict__ part, double* __restrict__ mean, float* __restrict__ stats, float* running_mean, float* running_var, int C,
                                       size_t M, int chunks, float eps, float momentum, int pass) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * C + c];
    if (!pass) { mean[c] = s / (double)M; return; }
    const double var = s / (double)M;
    stats[2 * c] = (float)mean[c];
    stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean[c];
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (double)M / (double)(M > 1 ? M - 1 : 1));
    }
}
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {   // counter-based dropout mask (not torch's generator: the mask is saved, parity tests use p = 0)
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ void bn_relu_dropout_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* __restrict__ out, uint8_t* __restrict__ keep, size_t M, int C, float p,
                                           uint32_t seed) {
    const size_t n = M * C;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = (x[i] - stats[2 * c]) * stats[2 * c + 1] * gamma[c] + beta[c];
        v = v > 0.f ? v : 0.f;
        uint8_t k = 1;
        if (p > 0.f) {
            const uint32_t r = hash_u32((uint32_t)i * 0x9E3779B9U + seed);
            k = ((float)(r >> 8) * (1.0f / 16777216.0f)) >= p;
        }
        keep[i] = k;
        out[i] = k ? v * inv_keep : 0.f;
    }
}
// d(pre-activation of the BN affine): dz = dout * keep/(1-p) * (out > 0)   (out = relu(...) * keep / (1-p), so out > 0 <=> relu passed and kept)
__global__ void bn_relu_dropout_bwd_pre_kernel(const float* __restrict__ dout, const float* __restrict__ out, const uint8_t* __restrict__ keep, float* __restrict__ dz, size_t n, float p) {
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dz[i] = (keep[i] && out[i] > 0.f) ? dout[i] * inv_keep : 0.f;
}
// dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); sums[c] = {sum dz, sum dz*xhat} from the colsum kernels (= dbeta, dgamma)
__global__ void bn_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                              const float* __restrict__ dbeta, const float* __restrict__ dgamma, float* __restrict__ dx, size_t M, int C) {
    const size_t n = M * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float xh = (x[i] - stats[2 * c]) * stats[2 * c + 1];
        dx[i] = gamma[c] * stats[2 * c + 1] * (dz[i] - dbeta[c] / (float)M - xh * dgamma[c] / (float)M);
    }
}
// xhat for the gamma gradient of BatchNorm
__global__ void bn_xhat_kernel(const float* __restrict__ x, const float* __restrict__ stats, float* __restrict__ xh, size_t M, int C) {
    const size_t n = M * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        xh[i] = (x[i] - stats[2 * c]) * stats[2 * c + 1];
    }
}

// Conv2d(C, K, 1) with tiny K (seg head: 3 classes): forward logits[m][k] = x[m][:] . w[k][:] + b[k]; backward dx[m][c] = sum_k dl[m][k] w[k][c]
__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                                                         size_t M, int C, int K) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += x[row * C + c] * w[(size_t)k * C + c];
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        if (lane == 0) out[row * K + k] = s + (bias ? bias[k] : 0.f);
    }
}
__global__ void smallk_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w, float* __restrict__ dx, size_t M, int C, int K) {
    const size_t n = M * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t m = i / C;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += dl[m * K + k] * w[(size_t)k * C + c];
        dx[i] = s;
    }
}
// dW[k][c] = sum_m dl[m][k] x[m][c] via per-chunk partials (deterministic): part[ch][k][c], K <= 4
__global__ __launch_bounds__(256) void smallk_wgrad_part_kernel(const float* __restrict__ dl, const float* __restrict__ x, float* __restrict__ part, size_t M, int C, int K,
                                                                int chunks) {
    __shared__ float red[4][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int ch = blockIdx.y;
    const size_t per = (M + chunks - 1) / chunks, lo = (size_t)ch * per, hi = lo + per < M ? lo + per : M;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (size_t m = lo + ty; m < hi; m += 4) {
            const float xv = x[m * C + c];
            for (int k = 0; k < K; ++k) s[k] += dl[m * K + k] * xv;
        }
    for (int k = 0; k < 4; ++k) red[ty][k][tx] = s[k];
    __syncthreads();
    if (ty == 0 && c < C)
        for (int k = 0; k < K; ++k) part[((size_t)ch * K + k) * C + c] = (red[0][k][tx] + red[1][k][tx]) + (red[2][k][tx] + red[3][k][tx]);
}

// seg activation + x2 upsample: d logits_up[b][y][x][k] from d seg [B][K][S][S] (NCHW) and the saved output: ScaledTanh' = 2 y (1 - y), sigmoid' = y (1 - y)
__global__ void seg_act_bwd_kernel(const float* __restrict__ dseg, const float* __restrict__ seg, float* __restrict__ dup, int B, int K, int S, int sigmoid) {
    const size_t n = (size_t)B * S * S * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        size_t r = i / K;
        const int x = (int)(r % S);
        r /= S;
        const int y = (int)(r % S), b = (int)(r / S);
        const size_t j = (((size_t)b * K + k) * S + y) * S + x;
        const float o = seg[j];
        dup[i] = dseg[j] * (sigmoid ? o * (1.f - o) : 2.f * o * (1.f - o));
    }
}

// depth head tail: inv = relu(sum_k relu(e[m][k]) * w4[k] + b4).  Forward (train) and backward: de[m][k] = dinv * (inv > 0) * w4[k] * (e > 0);
// per-row terms for dw4 (colsum of dz * relu(e)) and db4 (sum dz) are written to rowterm[m][K+1].
__global__ void depth_tail_fwd_kernel(const float* __restrict__ e, const float* __restrict__ w4, const float* __restrict__ b4, float* __restrict__ inv, size_t M, int K) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
        float s = b4[0];
        for (int k = 0; k < K; ++k) s += fmaxf(e[m * K + k], 0.f) * w4[k];
        inv[m] = fmaxf(s, 0.f);
    }
}
__global__ void depth_tail_bwd_kernel(const float* __restrict__ dinv, const float* __restrict__ inv, const float* __restrict__ e, const float* __restrict__ w4,
                                      float* __restrict__ de, float* __restrict__ rowterm, size_t M, int K) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
        const float dz = inv[m] > 0.f ? dinv[m] : 0.f;
        for (int k = 0; k < K; ++k) {
            const float ev = e[m * K + k];
            de[m * K + k] = ev > 0.f ? dz * w4[k] : 0.f;
            rowterm[m * (K + 1) + k] = dz * fmaxf(ev, 0.f);
        }
        rowterm[m * (K + 1) + K] = dz;
    }
}

// K = 32 forms: 8 threads per pixel, one float4 each (a thread per pixel walked a 128-byte row per lane: 617 us for the backward at B = 8)
__global__ void depth_tail_fwd32_kernel(const float* __restrict__ e, const float* __restrict__ w4, const float* __restrict__ b4, float* __restrict__ inv, size_t M) {
    const int c = threadIdx.x & 7;
    const float4 w = *reinterpret_cast<const float4*>(w4 + 4 * c);
    const float bias = b4[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * 8; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(e + i * 4);
        float s = fmaxf(v.x, 0.f) * w.x + fmaxf(v.y, 0.f) * w.y + fmaxf(v.z, 0.f) * w.z + fmaxf(v.w, 0.f) * w.w;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (c == 0) inv[i >> 3] = fmaxf(s + bias, 0.f);
    }
}
__global__ void depth_tail_bwd32_kernel(const float* __restrict__ dinv, const float* __restrict__ inv, const float* __restrict__ e, const float* __restrict__ w4,
                                        float* __restrict__ de, float* __restrict__ rowterm, size_t M) {
    const int c = threadIdx.x & 7;
    const float4 w = *reinterpret_cast<const float4*>(w4 + 4 * c);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * 8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i >> 3;
        const float dz = inv[m] > 0.f ? dinv[m] : 0.f;
        const float4 v = *reinterpret_cast<const float4*>(e + i * 4);
        *reinterpret_cast<float4*>(de + i * 4) = make_float4(v.x > 0.f ? dz * w.x : 0.f, v.y > 0.f ? dz * w.y : 0.f, v.z > 0.f ? dz * w.z : 0.f, v.w > 0.f ? dz * w.w : 0.f);
        float* rt = rowterm + m * 33 + 4 * c;
        rt[0] = dz * fmaxf(v.x, 0.f); rt[1] = dz * fmaxf(v.y, 0.f); rt[2] = dz * fmaxf(v.z, 0.f); rt[3] = dz * fmaxf(v.w, 0.f);
        if (c == 7) rt[4] = dz;
    }
}

// ---------------- Swin-V2 pieces ----------------
// PatchMerging gather backward: dg [B][R/2][R/2][4C] -> dx [B][R][R][C] (every token belongs to exactly one block: a permutation)
__global__ void merge_scatter_kernel(const float* __restrict__ dg, float* __restrict__ dx, int B, int R, int C) {
    const size_t n = (size_t)B * R * R * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % R);
        r /= R;
        const int y = (int)(r % R), b = (int)(r / R);
        const int blk = (y & 1) + 2 * (x & 1);   // timm order x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]
        dx[i] = dg[((((size_t)b * (R / 2) + y / 2) * (R / 2) + x / 2) * 4 + blk) * C + c];
    }
}
// patch embedding: x [B][3][S][S] -> patches [M][64] (k = c*16 + ky*4 + kx as in proj.weight [C0][3][4][4]; columns 48..63 zero)
__global__ void patch_im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int S) {
    const int G = S / 4;
    const size_t n = (size_t)B * G * G * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i & 63);
        size_t r = i >> 6;
        const int gx = (int)(r % G);
        r /= G;
        const int gy = (int)(r % G), b = (int)(r / G);
        float v = 0.f;
        if (k < 48) {
            const int c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
            v = x[(((size_t)b * 3 + c) * S + gy * 4 + ky) * S + gx * 4 + kx];
        }
        out[i] = v;
    }
}
// [N][64] -> [N][48] (drop the zero padding of the patch weight gradient) or the reverse with zero fill
__global__ void pad_cols_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int cin, int cout) {
    const size_t n = (size_t)N * cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cout), r = (int)(i / cout);
        out[i] = c < cin ? in[(size_t)r * cin + c] : 0.f;
    }
}

// Cosine window attention backward, one workgroup (64 threads) per (sample, window, head, 64-query block); one thread = one query.
//   q^ = scale * q/|q|, k^ = k/|k|, S = q^ k^T + bias(rel) + mask, P = softmax(S), O = P v
// Pass A (this kernel, per query): with the row statistics of attn_rowstat_kernel and delta = dO . O, for every key: p, dP = dO . v, dS = p (dP - delta);
//   accumulates dq^ (registers), and writes dS to a [.., N, N] scratch for pass B (per key) -- deterministic, no atomics.
// The bias-table and logit-scale gradients are reduced from that scratch by their own kernels.
__global__ __launch_bounds__(64) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ table,
                                                        const float* __restrict__ scale, const float* __restrict__ rowstat, const float* __restrict__ attn_out,
                                                        float* __restrict__ dS_out, float* __restrict__ dqkv_part, float* __restrict__ dscale_part, int res, int ws,
                                                        int shift, int heads, int nseg, size_t part_stride) {
    __shared__ float Kh[64][33];
    __shared__ float Vs[64][33];
    __shared__ float dsT[64][65];
    const int N = ws * ws, nqb = (N + 63) / 64;
    const int C = heads * 32, nw = res / ws;
    int bid = blockIdx.x;
    const int seg = bid % nseg;      // key tiles seg, seg + nseg, ...: more waves for the stages with few windows (deterministic partial sums)
    bid /= nseg;
    float* dqkv = dqkv_part + (size_t)seg * part_stride;
    const int qb = bid % nqb;
    bid /= nqb;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const int widx = (b * nw + wy) * nw + wx;
    const int tid = threadIdx.x;
    auto token_row = [&](int p) -> size_t {
        const int r = p / ws, c = p % ws;
        int sy = wy * ws + r + shift, sx = wx * ws + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int q = qb * 64 + tid;
    const bool qv = q < N;
    const int qc = qv ? q : N - 1;
    const int rq = qc / ws, cq = qc % ws;
    const float sc = scale[head];
    float qn[32], dOr[32], dqh[32];
    float qnorm;
    {
        const float* src = qkv + token_row(qc) * (size_t)(3 * C) + head * 32;
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { qn[d] = src[d]; ss += qn[d] * qn[d]; }
        qnorm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int d = 0; d < 32; ++d) { qn[d] /= qnorm; dqh[d] = 0.f; }
        const float* dsrc = dO + token_row(qc) * (size_t)C + head * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) dOr[d] = dsrc[d];
    }
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    const int half = ws / 2;
    auto stage = [&](int k0) {
        __syncthreads();
        const int k = k0 + tid;
        const int kc = k < N ? k : N - 1;
        const float* src = qkv + token_row(kc) * (size_t)(3 * C) + head * 32;
        float ss = 0.f;
        float kr[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) { kr[d] = src[C + d]; ss += kr[d] * kr[d]; }
        const float ki = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int d = 0; d < 32; ++d) { Kh[tid][d] = kr[d] * ki; Vs[tid][d] = src[2 * C + d]; }
        __syncthreads();
    };
    // row statistics {max, sum exp} come from attn_rowstat_kernel; delta = sum_j p_j (dO . v_j) = dO . O with O the saved attention output
    float m, l, delta = 0.f;
    {
        const float* rs = rowstat + (((size_t)widx * heads + head) * N + qc) * 2;
        m = rs[0];
        l = rs[1];
        const float* orow = attn_out + token_row(qc) * (size_t)C + head * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) delta = fmaf(dOr[d], orow[d], delta);
    }
    // dS, dq^.  The dS tile of 64 queries x 64 keys goes through LDS so that the global rows are written 256 bytes at a time
    // (a lane writing its own row was one 4-byte store per cache line).
    float dsc = 0.f;
    float* dSbase = dS_out + ((size_t)widx * heads + head) * N * N;
    for (int k0 = seg * 64; k0 < N; k0 += 64 * nseg) {
        stage(k0);
        const int nk = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < nk; ++kk) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;   // independent FMA chains
#pragma unroll
            for (int d = 0; d < 32; d += 4) {
                s0 = fmaf(qn[d], Kh[kk][d], s0); s1 = fmaf(qn[d + 1], Kh[kk][d + 1], s1); s2 = fmaf(qn[d + 2], Kh[kk][d + 2], s2); s3 = fmaf(qn[d + 3], Kh[kk][d + 3], s3);
                p0 = fmaf(dOr[d], Vs[kk][d], p0); p1 = fmaf(dOr[d + 1], Vs[kk][d + 1], p1); p2 = fmaf(dOr[d + 2], Vs[kk][d + 2], p2); p3 = fmaf(dOr[d + 3], Vs[kk][d + 3], p3);
            }
            const float dot = (s0 + s1) + (s2 + s3), dp = (p0 + p1) + (p2 + p3);
            const int k = k0 + kk, rk = k / ws, ck = k % ws;
            float sl = dot * sc + table[(size_t)((rq - rk + ws - 1) * (2 * ws - 1) + (cq - ck + ws - 1)) * heads + head];
            if ((lastrow && ((rk >= half) != (rq >= half))) || (lastcol && ((ck >= half) != (cq >= half)))) sl += -100.0f;
            const float ds = __expf(sl - m) / l * (dp - delta);
            dsT[tid][kk] = ds;
#pragma unroll
            for (int d = 0; d < 32; ++d) dqh[d] = fmaf(ds, Kh[kk][d], dqh[d]);
            dsc += ds * dot;      // d scale: S = scale * (qn . k^) + ...
        }
        __syncthreads();
        for (int i = 0; i < 64; ++i) {
            const int qq = qb * 64 + i;
            if (qq < N && tid < nk) dSbase[(size_t)qq * N + k0 + tid] = dsT[i][tid];
        }
    }
    if (qv) {
        // dq^ is w.r.t. q^ = scale * qn: dqn = scale * dqh; dq = (dqn - qn (qn . dqn)) / |q|
        float dotq = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) dotq = fmaf(qn[d], dqh[d] * sc, dotq);
        float* dst = dqkv + token_row(qc) * (size_t)(3 * C) + head * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) dst[d] = (dqh[d] * sc - qn[d] * dotq) / qnorm;
    }
    // per-(window, head, query block) partial of d scale (reduced in fixed order by attn_scale_reduce_kernel)
    dsc = qv ? dsc : 0.f;
    for (int o = 1; o < 64; o <<= 1) dsc += __shfl_xor(dsc, o);
    if (tid == 0) dscale_part[(((size_t)widx * heads + head) * nqb + qb) * nseg + seg] = dsc;
}

// Pass B, one thread = one key: P[q][k] is recomputed from the logits with the row statistics of attn_rowstat_kernel, dS is read back:
// dk^_k = sum_q dS[q][k] q^_q ; dv_k = sum_q P[q][k] dO_q.
__global__ __launch_bounds__(64) void attn_bwd_k_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ table,
                                                        const float* __restrict__ scale, const float* __restrict__ dS_in, const float* __restrict__ rowstat,
                                                        float* __restrict__ dqkv_part, int res, int ws, int shift, int heads, int nseg, size_t part_stride) {
    __shared__ float Qh[64][33];
    __shared__ float dOs[64][33];
    __shared__ float st_m[64], st_l[64];
    const int N = ws * ws, nkb = (N + 63) / 64;
    const int C = heads * 32, nw = res / ws;
    int bid = blockIdx.x;
    const int seg = bid % nseg;
    bid /= nseg;
    float* dqkv = dqkv_part + (size_t)seg * part_stride;
    const int kb = bid % nkb;
    bid /= nkb;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const int widx = (b * nw + wy) * nw + wx;
    const int tid = threadIdx.x;
    auto token_row = [&](int p) -> size_t {
        const int r = p / ws, c = p % ws;
        int sy = wy * ws + r + shift, sx = wx * ws + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int k = kb * 64 + tid;
    const bool kv = k < N;
    const int kc = kv ? k : N - 1;
    const int rk = kc / ws, ck = kc % ws;
    const float sc = scale[head];
    float kn[32], dkh[32], dv[32];
    float knorm;
    {
        const float* src = qkv + token_row(kc) * (size_t)(3 * C) + head * 32 + C;
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { kn[d] = src[d]; ss += kn[d] * kn[d]; }
        knorm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int d = 0; d < 32; ++d) { kn[d] /= knorm; dkh[d] = 0.f; dv[d] = 0.f; }
    }
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    const int half = ws / 2;
    for (int q0 = seg * 64; q0 < N; q0 += 64 * nseg) {
        __syncthreads();
        {
            const int q = q0 + tid;
            const int qc = q < N ? q : N - 1;
            const float* src = qkv + token_row(qc) * (size_t)(3 * C) + head * 32;
            float ss = 0.f;
            float qr[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) { qr[d] = src[d]; ss += qr[d] * qr[d]; }
            const float qi = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            const float* dsrc = dO + token_row(qc) * (size_t)C + head * 32;
#pragma unroll
            for (int d = 0; d < 32; ++d) { Qh[tid][d] = qr[d] * qi; dOs[tid][d] = dsrc[d]; }
            const float* rs = rowstat + (((size_t)widx * heads + head) * N + qc) * 2;
            st_m[tid] = rs[0];
            st_l[tid] = rs[1];
        }
        __syncthreads();
        const int nq = (N - q0) < 64 ? (N - q0) : 64;
        for (int qq = 0; qq < nq; ++qq) {
            const int q = q0 + qq, rq = q / ws, cq = q % ws;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(Qh[qq][d], kn[d], s);
            s *= sc;
            s += table[(size_t)((rq - rk + ws - 1) * (2 * ws - 1) + (cq - ck + ws - 1)) * heads + head];
            if ((lastrow && ((rk >= half) != (rq >= half))) || (lastcol && ((ck >= half) != (cq >= half)))) s += -100.0f;
            const float p = __expf(s - st_m[qq]) / st_l[qq];
            const float ds = dS_in[(((size_t)widx * heads + head) * N + q) * N + kc];
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                dkh[d] = fmaf(ds * sc, Qh[qq][d], dkh[d]);     // dS/dk^ = scale * qn
                dv[d] = fmaf(p, dOs[qq][d], dv[d]);
            }
        }
    }
    if (kv) {
        float dotk = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) dotk = fmaf(kn[d], dkh[d], dotk);
        float* dst = dqkv + token_row(kc) * (size_t)(3 * C) + head * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            dst[C + d] = (dkh[d] - kn[d] * dotk) / knorm;
            dst[2 * C + d] = dv[d];
        }
    }
}

// row statistics {max, sum exp} of every query (pass A of the forward recomputation, shared by attn_bwd_k_kernel)
__global__ __launch_bounds__(64) void attn_rowstat_kernel(const float* __restrict__ qkv, const float* __restrict__ table, const float* __restrict__ scale,
                                                          float* __restrict__ rowstat_part, int res, int ws, int shift, int heads, int nseg, size_t part_stride) {
    __shared__ float Kh[64][33];
    const int N = ws * ws, nqb = (N + 63) / 64;
    const int C = heads * 32, nw = res / ws;
    int bid = blockIdx.x;
    const int seg = bid % nseg;
    bid /= nseg;
    float* rowstat = rowstat_part + (size_t)seg * part_stride;
    const int qb = bid % nqb;
    bid /= nqb;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const int widx = (b * nw + wy) * nw + wx;
    const int tid = threadIdx.x;
    auto token_row = [&](int p) -> size_t {
        const int r = p / ws, c = p % ws;
        int sy = wy * ws + r + shift, sx = wx * ws + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int q = qb * 64 + tid;
    const int qc = q < N ? q : N - 1;
    const int rq = qc / ws, cq = qc % ws;
    const float sc = scale[head];
    float qn[32];
    {
        const float* src = qkv + token_row(qc) * (size_t)(3 * C) + head * 32;
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { qn[d] = src[d]; ss += qn[d] * qn[d]; }
        const float qi = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int d = 0; d < 32; ++d) qn[d] *= qi;
    }
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    const int half = ws / 2;
    float m = -3.0e38f, l = 0.f;
    for (int k0 = seg * 64; k0 < N; k0 += 64 * nseg) {
        __syncthreads();
        {
            const int k = k0 + tid;
            const int kc = k < N ? k : N - 1;
            const float* src = qkv + token_row(kc) * (size_t)(3 * C) + head * 32 + C;
            float ss = 0.f;
            float kr[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) { kr[d] = src[d]; ss += kr[d] * kr[d]; }
            const float ki = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int d = 0; d < 32; ++d) Kh[tid][d] = kr[d] * ki;
        }
        __syncthreads();
        const int nk = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < nk; ++kk) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(qn[d], Kh[kk][d], s);
            s *= sc;
            const int k = k0 + kk, rk = k / ws, ck = k % ws;
            s += table[(size_t)((rq - rk + ws - 1) * (2 * ws - 1) + (cq - ck + ws - 1)) * heads + head];
            if ((lastrow && ((rk >= half) != (rq >= half))) || (lastcol && ((ck >= half) != (cq >= half)))) s += -100.0f;
            const float mn = fmaxf(m, s);
            l = l * __expf(m - mn) + __expf(s - mn);
            m = mn;
        }
    }
    if (q < N) {
        float* rs = rowstat + (((size_t)widx * heads + head) * N + q) * 2;
        rs[0] = m;
        rs[1] = l;
    }
}

// merge the per-segment {max, sum exp} pairs into the row statistics
__global__ void attn_rowstat_combine_kernel(const float* __restrict__ part, float* __restrict__ rowstat, size_t rows, int nseg) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
        float M = -3.0e38f;
        for (int s = 0; s < nseg; ++s) M = fmaxf(M, part[((size_t)s * rows + i) * 2]);
        float L = 0.f;
        for (int s = 0; s < nseg; ++s) {
            const float l = part[((size_t)s * rows + i) * 2 + 1];
            if (l > 0.f) L += l * __expf(part[((size_t)s * rows + i) * 2] - M);
        }
        rowstat[i * 2] = M;
        rowstat[i * 2 + 1] = L;
    }
}
// sum of the per-segment gradient slabs, in segment order
__global__ void attn_seg_sum_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int nseg) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nseg; ++k) s += part[(size_t)k * n + i];
        out[i] = s;
    }
}

// d table[r][h] = sum over (q, k with rel(q, k) = r) of dSsum[h][q][k], dSsum = dS summed over the windows (rowsum_kernel).
// One wave per (r, h): lane = one key row rk of the window (ws <= 64), the ck loop inside; fixed-order wave reduction.
__global__ __launch_bounds__(64) void attn_table_grad_kernel(const float* __restrict__ dSsum, float* __restrict__ dtable, int ws, int heads) {
    const int T = 2 * ws - 1, N = ws * ws;
    const int i = blockIdx.x;                       // (r, h)
    const int h = i % heads, r = i / heads;
    const int dr = r / T - (ws - 1), dc = r % T - (ws - 1);   // rq - rk, cq - ck
    const float* base = dSsum + (size_t)h * N * N;
    const int rk = threadIdx.x;
    float s = 0.f;
    const int rq = rk + dr;
    if (rk < ws && rq >= 0 && rq < ws) {
        for (int ck = 0; ck < ws; ++ck) {
            const int cq = ck + dc;
            if (cq < 0 || cq >= ws) continue;
            s += base[(size_t)(rq * ws + cq) * N + rk * ws + ck];
        }
    }
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) dtable[i] = s;
}
// logit_scale gradient: scale = exp(min(ls, ln 100)); d ls = scale * d scale when ls < ln 100
__global__ void attn_scale_reduce_kernel(const float* __restrict__ part, const float* __restrict__ ls, float* __restrict__ dls, int nwin, int heads, int nqb) {
    __shared__ float red[256];
    const int h = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nwin * nqb; i += 256) s += part[((size_t)(i / nqb) * heads + h) * nqb + (i % nqb)];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l = ls[h];
        dls[h] = l < LN100 ? red[0] * __expf(l) : 0.f;
    }
}

// continuous position bias MLP backward: table = 16 sigmoid(t), t = relu(coords W0^T + b0) W2^T.
// dt = dtable * 16 s (1 - s) with s = table / 16; the hidden layer is recomputed into hid / dhid [(2ws-1)^2][512].
__global__ void cpb_dt_kernel(const float* __restrict__ dtable, const float* __restrict__ table, float* __restrict__ dt, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = table[i] * (1.0f / 16.0f);
    dt[i] = dtable[i] * 16.0f * s * (1.0f - s);
}
__device__ __forceinline__ void cpb_coords(int r, int ws, int pws, float& c0, float& c1) {
    const int T = 2 * ws - 1;
    const float den = (float)((pws > 0 ? pws : ws) - 1);
    float a = (float)(r / T - (ws - 1)) / den * 8.0f, b = (float)(r % T - (ws - 1)) / den * 8.0f;
    const float inv = 1.0f / log2f(8.0f);
    c0 = (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f)) * log2f(fabsf(a) + 1.0f) * inv;
    c1 = (b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f)) * log2f(fabsf(b) + 1.0f) * inv;
}
__global__ void cpb_hidden_kernel(const float* __restrict__ dt, const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w2,
                                  float* __restrict__ hid, float* __restrict__ dhid, int ws, int pws, int heads) {
    const int T2 = (2 * ws - 1) * (2 * ws - 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T2 * 512) return;
    const int j = i & 511, r = i >> 9;
    float c0, c1;
    cpb_coords(r, ws, pws, c0, c1);
    const float pre = c0 * w0[2 * j] + c1 * w0[2 * j + 1] + b0[j];
    hid[i] = fmaxf(pre, 0.f);
    float dh = 0.f;
    if (pre > 0.f)
        for (int h = 0; h < heads; ++h) dh += dt[(size_t)r * heads + h] * w2[(size_t)h * 512 + j];
    dhid[i] = dh;
}
// out[o][j] = sum_r coef_o[r] * mat[r][j] for j < 512: grid (8, outputs), block = 64 columns x 4 row phases.
//   mode 0 (dW2): coef_o[r] = dt[r][o], mat = hid.   mode 1 (dW0 / db0): o = 0, 1 -> coords, o = 2 -> 1; mat = dhid; out written as dw0[j][0..1], db0[j]
__global__ __launch_bounds__(1024) void cpb_reduce_kernel(const float* __restrict__ mat, const float* __restrict__ dt, float* __restrict__ out0, float* __restrict__ out1,
                                                         int ws, int pws, int heads, int mode) {
    __shared__ float red[16][64];
    const int T2 = (2 * ws - 1) * (2 * ws - 1);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx, o = blockIdx.y;
    auto coef_of = [&](int r) -> float {
        if (mode == 0) return dt[(size_t)r * heads + o];
        float c0, c1;
        cpb_coords(r, ws, pws, c0, c1);
        return o == 0 ? c0 : (o == 1 ? c1 : 1.0f);
    };
    float s4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent row streams: four loads in flight per thread
    int r = ty;
    for (; r + 48 < T2; r += 64) {
        float cf[4], mv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { cf[u] = coef_of(r + 16 * u); mv[u] = mat[(size_t)(r + 16 * u) * 512 + j]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += cf[u] * mv[u];
    }
    for (; r < T2; r += 16) s4[0] += coef_of(r) * mat[(size_t)r * 512 + j];
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0) {
        float v = 0.f;
        for (int p = 0; p < 16; ++p) v += red[p][tx];
        if (mode == 0) out0[(size_t)o * 512 + j] = v;
        else if (o < 2) { if (out0) out0[2 * j + o] = v; }
        else if (out1) out1[j] = v;
    }
}
// q_bias / v_bias gradients from the qkv bias gradient [3C] (the k part has no parameter)
__global__ void qv_bias_grad_kernel(const float* __restrict__ dqkv_bias, float* __restrict__ dq, float* __restrict__ dv, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    if (dq) dq[i] = dqkv_bias[i];
    if (dv) dv[i] = dqkv_bias[2 * C + i];
}

// Stochastic depth (timm DropPath, drop_path_rate of SwinTransformerV2 -- 0.1 by default in timm 0.6.12, not overridden by
// /root/reference/SOccDPT/model/backbones/swin2.py:15-30): out[b] = keep_b / (1 - p), keep_b ~ Bernoulli(1 - p) from a counter hash of
// (seed, stream_id, b).  The mask generator is not torch's (like the Dropout mask: parity runs use rate 0 or read the scales back).
__global__ void drop_path_fill_kernel(float* __restrict__ out, int B, float p, unsigned seed, unsigned stream_id) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    unsigned h = seed * 0x9E3779B1u ^ (stream_id + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (unsigned)b * 0xC2B2AE35u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    out[b] = (p > 0.f && u < p) ? 0.f : 1.0f / (1.0f - p);
}
// out[m][:] = in[m][:] * scale[m / rows_per_scale]
__global__ void scale_rows_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ scale, size_t n4, int C4, int rows_per_scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float s = scale[(i / C4) / rows_per_scale];
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// GradScaler.unscale_ + inf check of the fp16 amp mode, over a run of the flat gradient buffer: g *= inv_scale; *found |= any non-finite
__global__ void unscale_check_kernel(float* __restrict__ g, size_t n, float inv_scale, int* __restrict__ found) {
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i] * inv_scale;
        g[i] = v;
        bad |= !(fabsf(v) <= 3.0e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(found, 1);
}

inline unsigned gs_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

}  // namespace

#define TK(name) return check_launch(name, err)

int tr_weight_batch(const TrBatchTable& t, int total_tiles, int fmt, hipStream_t st, std::string& err) {
    if (t.n < 1 || t.n > kTrBatchMax || total_tiles < 1) { err = "weight_batch: bad table"; return 1; }
    if (fmt == 3) SOCCDPT_LAUNCH(weight_batch_kernel<x3raw>, dim3(total_tiles), dim3(256), 0, st, t);
    else if (fmt == 1 || fmt == 2) SOCCDPT_LAUNCH(weight_batch_kernel<f16raw>, dim3(total_tiles), dim3(256), 0, st, t);
    else if (fmt == 0) SOCCDPT_LAUNCH(weight_batch_kernel<uint16_t>, dim3(total_tiles), dim3(256), 0, st, t);
    else SOCCDPT_LAUNCH(weight_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, st, t);
    TK("weight_batch");
}
int tr_transpose(const float* in, float* out, int R, int C, int Rp, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(transpose_kernel<float>, dim3((C + 31) / 32, (Rp + 31) / 32), dim3(256), 0, st, in, out, R, C, Rp);
    TK("transpose");
}
int tr_transpose16(const float* in, uint16_t* out, int R, int C, int Rp, int f16, hipStream_t st, std::string& err) {
    if (f16 == 3) SOCCDPT_LAUNCH(transpose_kernel<x3raw>, dim3((C + 31) / 32, (Rp + 31) / 32), dim3(256), 0, st, in, reinterpret_cast<x3raw*>(out), R, C, Rp);   // x3: 4 bytes per element
    else if (f16) SOCCDPT_LAUNCH(transpose_kernel<f16raw>, dim3((C + 31) / 32, (Rp + 31) / 32), dim3(256), 0, st, in, reinterpret_cast<f16raw*>(out), R, C, Rp);
    else SOCCDPT_LAUNCH(transpose_kernel<uint16_t>, dim3((C + 31) / 32, (Rp + 31) / 32), dim3(256), 0, st, in, out, R, C, Rp);
    TK("transpose16");
}
int tr_im2colT(const float* halo, float* out, int B, int H, int W, int C, size_t Mp, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(im2colT_kernel<float>, dim3((C + 31) / 32, (unsigned)((Mp + 31) / 32), 9), dim3(256), 0, st, halo, out, B, H, W, C, Mp);
    TK("im2colT");
}
int tr_im2colT16(const float* halo, uint16_t* out, int B, int H, int W, int C, size_t Mp, int f16, hipStream_t st, std::string& err) {
    if (f16 == 3) SOCCDPT_LAUNCH(im2colT_kernel<x3raw>, dim3((C + 31) / 32, (unsigned)((Mp + 31) / 32), 9), dim3(256), 0, st, halo, reinterpret_cast<x3raw*>(out), B, H, W, C, Mp);
    else if (f16) SOCCDPT_LAUNCH(im2colT_kernel<f16raw>, dim3((C + 31) / 32, (unsigned)((Mp + 31) / 32), 9), dim3(256), 0, st, halo, reinterpret_cast<f16raw*>(out), B, H, W, C, Mp);
    else SOCCDPT_LAUNCH(im2colT_kernel<uint16_t>, dim3((C + 31) / 32, (unsigned)((Mp + 31) / 32), 9), dim3(256), 0, st, halo, out, B, H, W, C, Mp);
    TK("im2colT16");
}
int tr_dy_halo_T(const float* dy, void* out, int out16, int B, int r, int N, int margin, int ld, hipStream_t st, std::string& err, int rpp) {
    const dim3 grid((N + 31) / 32, (ld + 31) / 32);
    if (rpp <= 0) rpp = r + 2;
    if (out16 == 3) SOCCDPT_LAUNCH(dy_halo_T_kernel<x3raw>, grid, dim3(256), 0, st, dy, static_cast<x3raw*>(out), B, r, N, margin, ld, rpp);
    else if (out16 == 2) SOCCDPT_LAUNCH(dy_halo_T_kernel<f16raw>, grid, dim3(256), 0, st, dy, static_cast<f16raw*>(out), B, r, N, margin, ld, rpp);
    else if (out16) SOCCDPT_LAUNCH(dy_halo_T_kernel<uint16_t>, grid, dim3(256), 0, st, dy, static_cast<uint16_t*>(out), B, r, N, margin, ld, rpp);
    else SOCCDPT_LAUNCH(dy_halo_T_kernel<float>, grid, dim3(256), 0, st, dy, static_cast<float*>(out), B, r, N, margin, ld, rpp);
    TK("dy_halo_T");
}
// x3 only: the pitched transposed halo image of the activation (x_halo_T_kernel)
int tr_x_halo_T_x3(const float* halo, void* out, int B, int r, int C, int col0, int ld, int rpp, hipStream_t st, std::string& err) {
    const dim3 grid((C + 31) / 32, (ld + 31) / 32);
    SOCCDPT_LAUNCH(x_halo_T_kernel<x3raw>, grid, dim3(256), 0, st, halo, static_cast<x3raw*>(out), B, r, C, col0, ld, rpp);
    TK("x_halo_T_x3");
}
int tr_wgrad_permute9(const float* in, float* out, int N, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(wgrad_permute9_kernel, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, in, out, N, C);
    TK("wgrad_permute9");
}
int tr_conv_w_dgrad(const float* w, float* out, int N, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(conv_w_dgrad_kernel<float>, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, w, out, N, C);
    TK("conv_w_dgrad");
}
int tr_conv_w_dgrad16(const float* w, uint16_t* out, int N, int C, int f16, hipStream_t st, std::string& err) {
    if (f16 == 3) SOCCDPT_LAUNCH(conv_w_dgrad_kernel<x3raw>, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, w, reinterpret_cast<x3raw*>(out), N, C);
    else if (f16) SOCCDPT_LAUNCH(conv_w_dgrad_kernel<f16raw>, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, w, reinterpret_cast<f16raw*>(out), N, C);
    else SOCCDPT_LAUNCH(conv_w_dgrad_kernel<uint16_t>, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, w, out, N, C);
    TK("conv_w_dgrad16");
}
int tr_wgrad_permute(const float* in, float* out, int N, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(wgrad_permute_kernel, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, in, out, N, C);
    TK("wgrad_permute");
}
int tr_to_halo_full(const float* in, void* out, int B, int H, int W, int C, int fmt, hipStream_t st, std::string& err) {
    // fmt: 0 f32, 1 bf16, 2 fp16, 3 x3.  Writes the border too (zeros): no memset needed.  C % 8 == 0.
    if (C % 8) { err = "tr_to_halo_full: C must be a multiple of 8"; return 1; }
    const dim3 g(gs_blocks((size_t)B * (H + 2) * (W + 2) * (C / 8))), b(256);
    if (fmt == 0) SOCCDPT_LAUNCH(to_halo_full8_kernel<float>, g, b, 0, st, in, static_cast<float*>(out), B, H, W, C);
    else if (fmt == 1) SOCCDPT_LAUNCH(to_halo_full8_kernel<uint16_t>, g, b, 0, st, in, static_cast<uint16_t*>(out), B, H, W, C);
    else if (fmt == 2) SOCCDPT_LAUNCH(to_halo_full8_kernel<f16raw>, g, b, 0, st, in, static_cast<f16raw*>(out), B, H, W, C);
    else SOCCDPT_LAUNCH(to_halo_full8_kernel<x3raw>, g, b, 0, st, in, static_cast<x3raw*>(out), B, H, W, C);
    return check_launch("to_halo_full", err);
}
int tr_to_halo(const float* in, float* out, int B, int H, int W, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(to_halo_kernel<float>, dim3(gs_blocks((size_t)B * H * W * C)), dim3(256), 0, st, in, out, B, H, W, C);
    TK("to_halo");
}
int tr_to_halo16(const float* in, uint16_t* out, int B, int H, int W, int C, int f16, hipStream_t st, std::string& err) {
    if (f16 == 3) SOCCDPT_LAUNCH(to_halo_kernel<x3raw>, dim3(gs_blocks((size_t)B * H * W * C)), dim3(256), 0, st, in, reinterpret_cast<x3raw*>(out), B, H, W, C);
    else if (f16) SOCCDPT_LAUNCH(to_halo_kernel<f16raw>, dim3(gs_blocks((size_t)B * H * W * C)), dim3(256), 0, st, in, reinterpret_cast<f16raw*>(out), B, H, W, C);
    else SOCCDPT_LAUNCH(to_halo_kernel<uint16_t>, dim3(gs_blocks((size_t)B * H * W * C)), dim3(256), 0, st, in, out, B, H, W, C);
    TK("to_halo16");
}
int tr_from_halo(const float* halo, float* out, int B, int H, int W, int C, int accumulate, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(from_halo_kernel, dim3(gs_blocks((size_t)B * H * W * C)), dim3(256), 0, st, halo, out, B, H, W, C, accumulate);
    TK("from_halo");
}
// scratch: up to 65536 + N floats
int tr_colsum(const float* a, const float* b, float* out, float* scratch, size_t M, int N, int accumulate, hipStream_t st, std::string& err) {
    const int cb = (N + 63) / 64;
    int chunks = 512 / cb;
    if ((size_t)chunks > (M + 15) / 16) chunks = (int)((M + 15) / 16);
    if (chunks < 1) chunks = 1;
    SOCCDPT_LAUNCH(colsum_part_kernel<false>, dim3(cb, chunks), dim3(256), 0, st, a, b, scratch, M, N, chunks);
    SOCCDPT_LAUNCH(colsum_final_kernel, dim3(cb), dim3(1024), 0, st, scratch, out, static_cast<float*>(nullptr), N, N, chunks, accumulate);
    TK("colsum");
}
// out_ab[n] = sum_m a b, out_a[n] = sum_m a from ONE pass over a (LayerNorm gamma / beta gradients).  scratch: up to 131072 + 2N floats
int tr_colsum2(const float* a, const float* b, float* out_ab, float* out_a, float* scratch, size_t M, int N, hipStream_t st, std::string& err) {
    const int cb = (N + 63) / 64;
    int chunks = 512 / cb;
    if ((size_t)chunks > (M + 15) / 16) chunks = (int)((M + 15) / 16);
    if (chunks < 1) chunks = 1;
    SOCCDPT_LAUNCH(colsum_part_kernel<true>, dim3(cb, chunks), dim3(256), 0, st, a, b, scratch, M, N, chunks);
    SOCCDPT_LAUNCH(colsum_final_kernel, dim3((2 * N + 63) / 64), dim3(1024), 0, st, scratch, out_ab, out_a, N, 2 * N, chunks, 0);
    TK("colsum2");
}
int tr_cvt_x3_pair(const float* in0, void* out0, size_t n0, const float* in1, void* out1, size_t n1, hipStream_t st, std::string& err) {
    if ((n0 | n1) % 16) { err = "cvt_x3_pair: x3 tensors are multiples of 16 elements"; return 1; }
    size_t blocks = ((n0 + n1) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SOCCDPT_LAUNCH(cvt_x3_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in0, out0, n0, in1, out1, n1);
    TK("cvt_x3_pair");
}
// mode: 0 bf16, 1 fp16 (IEEE, no clamp); in1 may be null (n1 = 0)
int tr_cvt16_pair(const float* in0, uint16_t* out0, size_t n0, const float* in1, uint16_t* out1, size_t n1, int f16, hipStream_t st, std::string& err) {
    if ((n0 | n1) % 4) { err = "cvt16_pair: element counts must be multiples of 4"; return 1; }
    size_t blocks = ((n0 + n1) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (f16) SOCCDPT_LAUNCH(cvt16_pair_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, in0, out0, n0, in1, out1, n1);
    else SOCCDPT_LAUNCH(cvt16_pair_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, in0, out0, n0, in1, out1, n1);
    TK("cvt16_pair");
}
int tr_axpy(float* y, const float* x, size_t n, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(axpy_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, y, x, n);
    TK("axpy");
}
int tr_relu_bwd(const float* dy, const float* ref, const float* add, float* dx, size_t n, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(relu_bwd_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, dy, ref, add, dx, n);
    TK("relu_bwd");
}
int tr_relu_bwd_halo(const float* dy, const float* ref_halo, const float* add, float* dx, int B, int H, int W, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(relu_bwd_halo_kernel, dim3(gs_blocks((size_t)B * H * W * C)), dim3(256), 0, st, dy, ref_halo, add, dx, B, H, W, C);
    TK("relu_bwd_halo");
}
int tr_gelu_bwd(const float* dy, const float* pre, float* dx, size_t n, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(gelu_bwd_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, dy, pre, dx, n);
    TK("gelu_bwd");
}
int tr_ln_bwd(const float* y, const float* g, const float* dout, float* dy, float* xhat, int M, int C, float eps, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, st, y, g, dout, dy, xhat, M, C, eps);
    TK("ln_bwd");
}
int tr_bilinear_bwd(const float* dhi, float* dlo, int B, int h, int w, int H, int W, int C, int accumulate, hipStream_t st, std::string& err) {
    if (C % 4 == 0) SOCCDPT_LAUNCH(bilinear_bwd4_kernel, dim3(gs_blocks((size_t)B * h * w * (C / 4))), dim3(256), 0, st, dhi, dlo, B, h, w, H, W, C, accumulate);
    else SOCCDPT_LAUNCH(bilinear_bwd_kernel, dim3(gs_blocks((size_t)B * h * w * C)), dim3(256), 0, st, dhi, dlo, B, h, w, H, W, C, accumulate);
    TK("bilinear_bwd");
}
// scratch: (128 * C + C) doubles
int tr_bn_stats(const float* x, float* stats, float* rmean, float* rvar, void* scratch, int C, size_t M, float eps, float momentum, hipStream_t st, std::string& err) {
    const int chunks = 128;
    double* part = static_cast<double*>(scratch);
    double* mean = part + (size_t)chunks * C;
    for (int pass = 0; pass < 2; ++pass) {
        SOCCDPT_LAUNCH(bn_moment_part_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, mean, part, M, C, chunks, pass);
        SOCCDPT_LAUNCH(bn_moment_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, part, mean, stats, rmean, rvar, C, M, chunks, eps, momentum, pass);
    }
    TK("bn_stats");
}
int tr_bn_relu_dropout_fwd(const float* x, const float* stats, const float* gamma, const float* beta, float* out, uint8_t* keep, size_t M, int C, float p, uint32_t seed,
                           hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(bn_relu_dropout_fwd_kernel, dim3(gs_blocks(M * C)), dim3(256), 0, st, x, stats, gamma, beta, out, keep, M, C, p, seed);
    TK("bn_relu_dropout_fwd");
}
int tr_bn_relu_dropout_bwd_pre(const float* dout, const float* out, const uint8_t* keep, float* dz, size_t n, float p, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(bn_relu_dropout_bwd_pre_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, dout, out, keep, dz, n, p);
    TK("bn_relu_dropout_bwd_pre");
}
int tr_bn_bwd(const float* dz, const float* x, const float* stats, const float* gamma, const float* dbeta, const float* dgamma, float* dx, size_t M, int C, hipStream_t st,
              std::string& err) {
    SOCCDPT_LAUNCH(bn_bwd_kernel, dim3(gs_blocks(M * C)), dim3(256), 0, st, dz, x, stats, gamma, dbeta, dgamma, dx, M, C);
    TK("bn_bwd");
}
int tr_bn_xhat(const float* x, const float* stats, float* xh, size_t M, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(bn_xhat_kernel, dim3(gs_blocks(M * C)), dim3(256), 0, st, x, stats, xh, M, C);
    TK("bn_xhat");
}
int tr_smallk_fwd(const float* x, const float* w, const float* bias, float* out, size_t M, int C, int K, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(smallk_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, w, bias, out, M, C, K);
    TK("smallk_fwd");
}
int tr_smallk_dgrad(const float* dl, const float* w, float* dx, size_t M, int C, int K, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(smallk_dgrad_kernel, dim3(gs_blocks(M * C)), dim3(256), 0, st, dl, w, dx, M, C, K);
    TK("smallk_dgrad");
}
// scratch: 256 * K * C floats (K <= 4)
int tr_smallk_wgrad(const float* dl, const float* x, float* dw, float* scratch, size_t M, int C, int K, hipStream_t st, std::string& err) {
    if (K > 4) { err = "smallk_wgrad: K > 4"; return 1; }
    const int chunks = 256;
    SOCCDPT_LAUNCH(smallk_wgrad_part_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, dl, x, scratch, M, C, K, chunks);
    SOCCDPT_LAUNCH(colsum_final_kernel, dim3((K * C + 63) / 64), dim3(1024), 0, st, scratch, dw, static_cast<float*>(nullptr), K * C, K * C, chunks, 0);
    TK("smallk_wgrad");
}
int tr_seg_act_bwd(const float* dseg, const float* seg, float* dup, int B, int K, int S, int sigmoid, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(seg_act_bwd_kernel, dim3(gs_blocks((size_t)B * S * S * K)), dim3(256), 0, st, dseg, seg, dup, B, K, S, sigmoid);
    TK("seg_act_bwd");
}
int tr_depth_tail_fwd(const float* e, const float* w4, const float* b4, float* inv, size_t M, int K, hipStream_t st, std::string& err) {
    if (K == 32) SOCCDPT_LAUNCH(depth_tail_fwd32_kernel, dim3(gs_blocks(M * 8)), dim3(256), 0, st, e, w4, b4, inv, M);
    else SOCCDPT_LAUNCH(depth_tail_fwd_kernel, dim3(gs_blocks(M)), dim3(256), 0, st, e, w4, b4, inv, M, K);
    TK("depth_tail_fwd");
}
int tr_depth_tail_bwd(const float* dinv, const float* inv, const float* e, const float* w4, float* de, float* rowterm, size_t M, int K, hipStream_t st, std::string& err) {
    if (K == 32) SOCCDPT_LAUNCH(depth_tail_bwd32_kernel, dim3(gs_blocks(M * 8)), dim3(256), 0, st, dinv, inv, e, w4, de, rowterm, M);
    else SOCCDPT_LAUNCH(depth_tail_bwd_kernel, dim3(gs_blocks(M)), dim3(256), 0, st, dinv, inv, e, w4, de, rowterm, M, K);
    TK("depth_tail_bwd");
}
int tr_merge_scatter(const float* dg, float* dx, int B, int R, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(merge_scatter_kernel, dim3(gs_blocks((size_t)B * R * R * C)), dim3(256), 0, st, dg, dx, B, R, C);
    TK("merge_scatter");
}
int tr_patch_im2col(const float* x, float* out, int B, int S, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(patch_im2col_kernel, dim3(gs_blocks((size_t)B * (S / 4) * (S / 4) * 64)), dim3(256), 0, st, x, out, B, S);
    TK("patch_im2col");
}
int tr_pad_cols(const float* in, float* out, int N, int cin, int cout, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(pad_cols_kernel, dim3(gs_blocks((size_t)N * cout)), dim3(256), 0, st, in, out, N, cin, cout);
    TK("pad_cols");
}
// Segments of the walked axis per (window, head, tile): up to 4 (one per 64-row tile of a 16 x 16 window)
static int attn_nseg(int ws) {
    const int tiles = (ws * ws + 63) / 64;
    return tiles < 4 ? tiles : 4;
}
// dS scratch: nwin * heads * N * N floats; rowstat: nwin * heads * N * 2; dscale_part: nwin * heads * nqb * 4;
// part: 4 * (B * res * res * 3 * heads * 32 + nwin * heads * N * 2) floats of scratch
int tr_attention_bwd(const float* qkv, const float* attn_out, const float* dO, const float* table, const float* scale, float* dS, float* rowstat, float* dscale_part,
                     float* part, float* dqkv, int B, int res, int ws, int shift, int heads, hipStream_t st, std::string& err) {
    const int nw = res / ws, N = ws * ws, nqb = (N + 63) / 64, nseg = attn_nseg(ws);
    const unsigned blocks = (unsigned)(B * nw * nw * heads * nqb * nseg);
    const size_t rows = (size_t)B * nw * nw * heads * N, nq = (size_t)B * res * res * 3 * heads * 32;
    float* part_stat = part + (size_t)nseg * nq;
    SOCCDPT_LAUNCH(attn_rowstat_kernel, dim3(blocks), dim3(64), 0, st, qkv, table, scale, part_stat, res, ws, shift, heads, nseg, rows * 2);
    SOCCDPT_LAUNCH(attn_rowstat_combine_kernel, dim3(gs_blocks(rows)), dim3(256), 0, st, part_stat, rowstat, rows, nseg);
    SOCCDPT_LAUNCH(attn_bwd_q_kernel, dim3(blocks), dim3(64), 0, st, qkv, dO, table, scale, rowstat, attn_out, dS, part, dscale_part, res, ws, shift, heads, nseg, nq);
    SOCCDPT_LAUNCH(attn_bwd_k_kernel, dim3(blocks), dim3(64), 0, st, qkv, dO, table, scale, dS, rowstat, part, res, ws, shift, heads, nseg, nq);
    SOCCDPT_LAUNCH(attn_seg_sum_kernel, dim3(gs_blocks(nq)), dim3(256), 0, st, part, dqkv, nq, nseg);
    TK("attention_bwd");
}
// dS is overwritten by its sum over the windows (first nwin = 1 slab); hid: 2 * (2ws-1)^2 * 512 floats of scratch
int tr_attn_param_grads(float* dS, const float* dscale_part, const float* table, const float* ls, const float* w0, const float* b0, const float* w2, float* dtable,
                        float* dt, float* hid, float* dls, float* dw0, float* db0, float* dw2, int nwin, int ws, int pws, int heads, hipStream_t st, std::string& err,
                        int dscale_slots) {
    const int T2 = (2 * ws - 1) * (2 * ws - 1), N = ws * ws, nqb = (N + 63) / 64;
    if (dls) SOCCDPT_LAUNCH(attn_scale_reduce_kernel, dim3(heads), dim3(256), 0, st, dscale_part, ls, dls, nwin, heads, dscale_slots > 0 ? dscale_slots : nqb * attn_nseg(ws));
    if (dw0 || db0 || dw2) {
        const size_t n = (size_t)heads * N * N;
        // in place: column i of slab 0 is read before it is written, the other slabs are only read
        SOCCDPT_LAUNCH(rowsum_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, dS, dS, nwin, n);
        SOCCDPT_LAUNCH(attn_table_grad_kernel, dim3(T2 * heads), dim3(64), 0, st, dS, dtable, ws, heads);
        SOCCDPT_LAUNCH(cpb_dt_kernel, dim3((T2 * heads + 255) / 256), dim3(256), 0, st, dtable, table, dt, T2 * heads);
        float* dhid = hid + (size_t)T2 * 512;
        SOCCDPT_LAUNCH(cpb_hidden_kernel, dim3((T2 * 512 + 255) / 256), dim3(256), 0, st, dt, w0, b0, w2, hid, dhid, ws, pws, heads);
        if (dw2) SOCCDPT_LAUNCH(cpb_reduce_kernel, dim3(8, heads), dim3(1024), 0, st, hid, dt, dw2, (float*)nullptr, ws, pws, heads, 0);
        if (dw0 || db0) SOCCDPT_LAUNCH(cpb_reduce_kernel, dim3(8, 3), dim3(1024), 0, st, dhid, dt, dw0, db0, ws, pws, heads, 1);
    }
    TK("attn_param_grads");
}
int tr_drop_path_fill(float* out, int B, float p, unsigned seed, unsigned stream_id, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(drop_path_fill_kernel, dim3((B + 63) / 64), dim3(64), 0, st, out, B, p, seed, stream_id);
    TK("drop_path_fill");
}
int tr_scale_rows(const float* in, float* out, const float* scale, size_t M, int C, int rows_per_scale, hipStream_t st, std::string& err) {
    const size_t n4 = M * (size_t)(C / 4);
    SOCCDPT_LAUNCH(scale_rows_kernel, dim3(gs_blocks(n4)), dim3(256), 0, st, in, out, scale, n4, C / 4, rows_per_scale);
    TK("scale_rows");
}
int tr_unscale_check(float* g, size_t n, float inv_scale, int* found, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(unscale_check_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, g, n, inv_scale, found);
    TK("unscale_check");
}
int tr_qv_bias_grad(const float* dqkv_bias, float* dq, float* dv, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(qv_bias_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, st, dqkv_bias, dq, dv, C);
    TK("qv_bias_grad");
}

}  // namespace soccdpt
