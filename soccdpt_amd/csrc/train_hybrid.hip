// Backward kernels of the ViT-hybrid encoder (dpt_hybrid_384: timm vit_base_resnet50_384 = ResNetV2 stem + stages, HybridEmbed, 12 pre-norm
// ViT-B blocks; /root/reference/SOccDPT/model/backbones/vit.py:147-258, backbones/utils.py:27-133), exact f32, deterministic.
// GEMM-shaped gradients go through the igemm like the Swin path (train_step.cpp); here: GroupNorm backward (with the ReLU mask recomputed
// from the saved raw convolution output), weight-standardisation backward, strided / 'SAME' convolution plumbing (generalised im2col^T,
// col2im as a gather), max-pool backward, the ProjectReadout concatenation, global softmax attention backward.
#include "kernels.h"

namespace soccdpt {
namespace {

inline unsigned gs_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

// ---------------- GroupNorm backward ----------------
// y = ((x - mean_bg) * rstd_bg) * gamma_c + beta_c, out = relu ? max(y, 0) : y.  dy_eff = (relu && y <= 0) ? 0 : dout.
//   dx = rstd * (gamma dy_eff - m1 - xhat m2),  m1 = mean_{hw, c in g}(gamma dy_eff),  m2 = mean_{hw, c in g}(gamma dy_eff xhat)
// stage 1: per (sample, channel) partial sums of dy_eff and dy_eff * xhat over a chunk of pixels
__global__ __launch_bounds__(256) void gn_bwd_part_kernel(const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ part, int HW, int C, int cpg,
                                                          int chunks, int relu) {
    __shared__ float red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, ch = blockIdx.y, b = blockIdx.z;
    const int per = (HW + chunks - 1) / chunks, lo = ch * per, hi = lo + per < HW ? lo + per : HW;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const int G = C / cpg;
        const float mean = stats[((size_t)b * G + c / cpg) * 2], rstd = stats[((size_t)b * G + c / cpg) * 2 + 1];
        const float g = gamma[c], be = beta[c];
        for (int p = lo + ty; p < hi; p += 4) {
            const size_t i = ((size_t)b * HW + p) * C + c;
            const float xh = (x[i] - mean) * rstd;
            float dy = dout[i];
            if (relu && !(xh * g + be > 0.f)) dy = 0.f;
            s1 += dy;
            s2 += dy * xh;
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        float* o = part + (((size_t)b * chunks + ch) * 2) * C + c;
        o[0] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        o[C] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
    }
}
// stage 2 (one workgroup of 1024 threads PER SAMPLE, one thread per channel): chunk sums -> S1, S2 per (b, c); group means gm[b][g] = {m1, m2}.
// Workgroup 0 also walks the other samples' partials for dgamma[c] = sum_b S2, dbeta[c] = sum_b S1 (fixed order: deterministic, no atomics).  The first
// version was ONE workgroup of 256 threads looping over samples, channels and chunks: 55 us per launch, 2.9 ms per dpt_hybrid_384 training step.
__global__ __launch_bounds__(1024) void gn_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ gamma, float* __restrict__ gm, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int B, int HW, int C, int cpg, int chunks) {
    __shared__ float s1[1024], s2[1024];
    const int c = threadIdx.x, G = C / cpg, b0 = blockIdx.x;
    auto sums = [&](int b, float& a, float& q) {
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
        int ch = 0;
        for (; ch + 3 < chunks; ch += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* o = part + (((size_t)b * chunks + ch + u) * 2) * C + c;
                a4[u] += o[0];
                q4[u] += o[C];
            }
        }
        for (; ch < chunks; ++ch) {
            const float* o = part + (((size_t)b * chunks + ch) * 2) * C + c;
            a4[0] += o[0];
            q4[0] += o[C];
        }
        a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    };
    float a = 0.f, q = 0.f;
    if (c < C) sums(b0, a, q);
    s1[c] = a;
    s2[c] = q;
    __syncthreads();
    if (c < G) {
        float m1 = 0.f, m2 = 0.f;
        for (int j = 0; j < cpg; ++j) {
            const int cc = c * cpg + j;
            m1 += gamma[cc] * s1[cc];
            m2 += gamma[cc] * s2[cc];
        }
        const float inv = 1.0f / ((float)HW * (float)cpg);
        gm[((size_t)b0 * G + c) * 2] = m1 * inv;
        gm[((size_t)b0 * G + c) * 2 + 1] = m2 * inv;
    }
    if (b0 == 0 && c < C && (dgamma || dbeta)) {
        float db = a, dg = q;
        for (int b = 1; b < B; ++b) {
            float a2, q2;
            sums(b, a2, q2);
            db += a2;
            dg += q2;
        }
        if (dgamma) dgamma[c] = dg;
        if (dbeta) dbeta[c] = db;
    }
}
__global__ void gn_bwd_apply_kernel(const float* dout, const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ gm, float* dx, size_t M, int HW, int C, int cpg, int relu) {
    const size_t n = M * C;
    const int G = C / cpg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t m = i / C;
        const int b = (int)(m / HW), g = c / cpg;
        const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
        const float xh = (x[i] - mean) * rstd;
        float dy = dout[i];
        if (relu && !(xh * gamma[c] + beta[c] > 0.f)) dy = 0.f;
        dx[i] = rstd * (gamma[c] * dy - gm[((size_t)b * G + g) * 2] - xh * gm[((size_t)b * G + g) * 2 + 1]);
    }
}

// ---------------- weight standardisation backward ----------------
// w_hat = (w - mean) * rstd over the fan-in of one output channel (biased variance, eps); dw = rstd (dwh - mean(dwh) - w_hat mean(dwh w_hat)).
// dwh / w_hat are tap-major [Cout][Kpad] (index tap * Cin + ci), dw is written in the parameter layout [Cout][Cin][k*k].
__global__ __launch_bounds__(256) void ws_bwd_kernel(const float* __restrict__ dwh, const float* __restrict__ wh, const float* __restrict__ w, float* __restrict__ dw,
                                                     int Cin, int kk, int Kpad, float eps) {
    __shared__ double red[4][256];
    const int co = blockIdx.x, tid = threadIdx.x, fan = Cin * kk;
    const float* src = w + (size_t)co * fan;
    const float* dh = dwh + (size_t)co * Kpad;
    const float* h = wh + (size_t)co * Kpad;
    double a = 0.0, q = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < fan; i += 256) {
        const double v = src[i];
        a += v;
        q += v * v;
        s1 += (double)dh[i];
        s2 += (double)dh[i] * (double)h[i];
    }
    red[0][tid] = a; red[1][tid] = q; red[2][tid] = s1; red[3][tid] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
            for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    const double mean = red[0][0] / fan;
    double var = red[1][0] / fan - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float m1 = (float)(red[2][0] / fan), m2 = (float)(red[3][0] / fan);
    for (int i = tid; i < fan; i += 256) {
        const int tap = i / Cin, ci = i - tap * Cin;
        dw[(size_t)co * fan + (size_t)ci * kk + tap] = rstd * (dh[i] - m1 - h[i] * m2);
    }
}

// tap-major W [N][9][C] -> dgrad operand [C][8 - tap][N] (the rotated filter); cf. conv_w_dgrad_kernel (parameter-layout input)
__global__ void conv_w_dgrad_tap_kernel(const float* __restrict__ wt, float* __restrict__ out, int N, int C) {
    const size_t n = (size_t)N * C * 9;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int nn = (int)(i % N);
        const size_t r = i / N;
        const int tap = (int)(r % 9), c = (int)(r / 9);
        out[i] = wt[((size_t)nn * 9 + (8 - tap)) * C + c];
    }
}

// ---------------- strided / 'SAME' 3x3 convolution plumbing ----------------
// im2col^T of a zero-haloed NHWC image for output pixel (oy, ox) reading halo pixel (oy * stride + ky + off, ox * stride + kx + off)
// (off = 1 - pad).  out [(tap * C + c)][Mp], columns >= M zero.
__global__ __launch_bounds__(256) void im2colT_gen_kernel(const float* __restrict__ halo, float* __restrict__ out, int B, int Hi, int Ho, int C, int stride, int off,
                                                          size_t Mp) {
    __shared__ float t[32][33];
    const size_t M = (size_t)B * Ho * Ho;
    const int tap = blockIdx.z, ky = tap / 3, kx = tap % 3;
    const size_t m0 = (size_t)blockIdx.y * 32;
    const int c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const size_t m = m0 + i;
        float v = 0.f;
        if (m < M && c0 + tx < C) {
            const int b = (int)(m / ((size_t)Ho * Ho)), r = (int)(m - (size_t)b * Ho * Ho), y = r / Ho, x = r - y * Ho;
            v = halo[(((size_t)b * (Hi + 2) + y * stride + ky + off) * (Hi + 2) + x * stride + kx + off) * C + c0 + tx];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const size_t m = m0 + tx;
        if (c < C && m < Mp) out[((size_t)tap * C + c) * Mp + m] = t[tx][i];
    }
}
// col2im as a gather: dX[b][iy][ix][c] = sum over taps of dcol[(b, oy, ox)][tap][c] with oy * stride + ky - pad = iy (and likewise x)
__global__ void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int Hi, int Ho, int C, int stride, int pad, int accumulate) {
    const size_t n = (size_t)B * Hi * Hi * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ix = (int)(r % Hi);
        r /= Hi;
        const int iy = (int)(r % Hi), b = (int)(r / Hi);
        float s = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Ho) continue;
                s += dcol[((((size_t)b * Ho + oy) * Ho + ox) * 9 + ky * 3 + kx) * C + c];
            }
        }
        dx[i] = accumulate ? dx[i] + s : s;
    }
}
// plain [B][Hi][Hi][C] -> rows of the stride-s pixels [B][Ho][Ho][C] (1x1 stride-2 shortcut convolution), and its transpose (scatter, others zero / kept)
__global__ void stride_gather_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Hi, int Ho, int C, int stride) {
    const size_t n = (size_t)B * Ho * Ho * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % Ho);
        r /= Ho;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        out[i] = in[(((size_t)b * Hi + oy * stride) * Hi + ox * stride) * C + c];
    }
}
__global__ void stride_scatter_add_kernel(const float* __restrict__ dg, float* __restrict__ dx, int B, int Hi, int Ho, int C, int stride) {
    const size_t n = (size_t)B * Ho * Ho * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % Ho);
        r /= Ho;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        dx[(((size_t)b * Hi + oy * stride) * Hi + ox * stride) * C + c] += dg[i];
    }
}

// ---------------- stem max-pool (3x3 / 2, TF 'SAME': the extra pixel right / bottom, -inf padding) backward ----------------
// The pooled input is relu(GN(raw)); it is recomputed from the saved raw convolution output.  Pass 1 records the window position (0..8) of
// the first maximum in scan order (torch.max_pool2d's choice); pass 2 gathers per input pixel.
__device__ __forceinline__ float gn_relu_at(const float* x, const float* stats, const float* gamma, const float* beta, int b, int p, int HW, int C, int cpg, int c) {
    const int G = C / cpg;
    const float mean = stats[((size_t)b * G + c / cpg) * 2], rstd = stats[((size_t)b * G + c / cpg) * 2 + 1];
    const float y = ((x[((size_t)b * HW + p) * C + c] - mean) * rstd) * gamma[c] + beta[c];
    return y > 0.f ? y : 0.f;
}
__global__ void maxpool_argmax_kernel(const float* __restrict__ raw, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      uint8_t* __restrict__ idx, int B, int Hi, int Ho, int C, int cpg) {
    const size_t n = (size_t)B * Ho * Ho * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % Ho);
        r /= Ho;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        float best = -3.0e38f;
        int bi = 0;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
            if (iy >= Hi) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx;
                if (ix >= Hi) continue;
                const float v = gn_relu_at(raw, stats, gamma, beta, b, iy * Hi + ix, Hi * Hi, C, cpg, c);
                if (v > best) { best = v; bi = ky * 3 + kx; }
            }
        }
        idx[i] = (uint8_t)bi;
    }
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx, float* __restrict__ dA, int B, int Hi, int Ho, int C) {
    const size_t n = (size_t)B * Hi * Hi * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ix = (int)(r % Hi);
        r /= Hi;
        const int iy = (int)(r % Hi), b = (int)(r / Hi);
        float s = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= Ho) continue;
                const size_t o = (((size_t)b * Ho + oy) * Ho + ox) * C + c;
                if (idx[o] == ky * 3 + kx) s += dpool[o];
            }
        }
        dA[i] = s;
    }
}

// stem weights: parameter layout [64][3][7][7] <-> the GEMM's [64][160] (k = (ky*7 + kx)*3 + c) is the tap-major layout with Cin = 3: ws kernels cover it.

// ---------------- ProjectReadout concatenation ----------------
// tok [B][NT][E] -> cat [B*(NT-1)][2E] = (token row, class-token row)
__global__ void readout_cat_kernel(const float* __restrict__ tok, float* __restrict__ cat, int B, int NT, int E) {
    const size_t n = (size_t)B * (NT - 1) * 2 * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % (2 * E));
        const size_t r = i / (2 * E);
        const int t = (int)(r % (NT - 1)), b = (int)(r / (NT - 1));
        cat[i] = e < E ? tok[((size_t)b * NT + 1 + t) * E + e] : tok[((size_t)b * NT) * E + (e - E)];
    }
}
// dtok[b][1 + t][:] (+)= dcat[.][:E];  dtok[b][0][:] (+)= sum_t dcat[.][E:]   (one thread per (b, token row incl. class row, e))
__global__ void readout_cat_bwd_kernel(const float* __restrict__ dcat, float* __restrict__ dtok, int B, int NT, int E, int accumulate) {
    const size_t n = (size_t)B * NT * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const size_t r = i / E;
        const int t = (int)(r % NT), b = (int)(r / NT);
        float v;
        if (t > 0) v = dcat[((size_t)b * (NT - 1) + t - 1) * 2 * E + e];
        else {
            v = 0.f;
            for (int k = 0; k < NT - 1; ++k) v += dcat[((size_t)b * (NT - 1) + k) * 2 * E + E + e];
        }
        dtok[i] = accumulate ? dtok[i] + v : v;
    }
}
// token stream [B][NT][E] <-> patch rows [B][NT-1][E] (class row dropped / zero)
__global__ void tokens_to_patches_kernel(const float* __restrict__ dtok, float* __restrict__ dpatch, int B, int NT, int E) {
    const size_t n = (size_t)B * (NT - 1) * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const size_t r = i / E;
        const int t = (int)(r % (NT - 1)), b = (int)(r / (NT - 1));
        dpatch[i] = dtok[((size_t)b * NT + 1 + t) * E + e];
    }
}

// ---------------- global softmax attention, forward (train mode) and backward (timm vision_transformer.Attention, head dim 64) ----------------
// qkv [B*N][3*H*64]; P = softmax(q k^T / 8); O = P v.  One thread = one query (forward, dq) or one key (dk, dv); the other axis is walked in
// tiles of 64 rows staged in LDS.  A (sample, head) pair offers only ceil(N / 64) workgroups of one wave, so the walked axis is also dealt to
// NSEG workgroups (tile t goes to segment t % NSEG) whose partial results a combine kernel merges in segment order (deterministic).
// rowstat[b][h][q] = {max, sum exp} is written by the forward and reused by the backward.
constexpr int VD = 64;
constexpr int NSEG = 8;
__global__ __launch_bounds__(64) void vit_attn_fwd_seg_kernel(const float* __restrict__ qkv, float* __restrict__ part_o, float* __restrict__ part_ml, int N, int heads,
                                                              int B) {
    __shared__ __attribute__((aligned(16))) float Ks[64][VD];
    __shared__ __attribute__((aligned(16))) float Vs[64][VD];
    const int nqb = (N + 63) / 64, E = heads * VD;
    int bid = blockIdx.x;
    const int seg = bid % NSEG;
    bid /= NSEG;
    const int qb = bid % nqb;
    bid /= nqb;
    const int head = bid % heads, b = bid / heads;
    const int tid = threadIdx.x, q = qb * 64 + tid, qc = q < N ? q : N - 1;
    float qr[VD], o[VD];
    const float* src = qkv + ((size_t)b * N + qc) * 3 * E + head * VD;
#pragma unroll
    for (int d = 0; d < VD; ++d) { qr[d] = src[d] * 0.125f; o[d] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    for (int k0 = seg * 64; k0 < N; k0 += 64 * NSEG) {
        __syncthreads();
        for (int i = 0; i < 64; ++i) {   // row i of the tile, lane = channel: coalesced global reads, conflict-free LDS writes
            const int k = k0 + i, kc = k < N ? k : N - 1;
            const float* ks = qkv + ((size_t)b * N + kc) * 3 * E + E + head * VD;
            Ks[i][tid] = ks[tid];
            Vs[i][tid] = ks[E + tid];
        }
        __syncthreads();
        const int nk = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < nk; ++kk) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains: the 64-long dependent FMA chain was the bound
#pragma unroll
            for (int d = 0; d < VD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(&Ks[kk][d]);
                s0 = fmaf(qr[d], kv.x, s0); s1 = fmaf(qr[d + 1], kv.y, s1); s2 = fmaf(qr[d + 2], kv.z, s2); s3 = fmaf(qr[d + 3], kv.w, s3);
            }
            const float s = (s0 + s1) + (s2 + s3);
            const float mn = fmaxf(m, s);
            const float al = __expf(m - mn), p = __expf(s - mn);
            l = l * al + p;
#pragma unroll
            for (int d = 0; d < VD; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(&Vs[kk][d]);
                o[d] = fmaf(p, vv.x, o[d] * al); o[d + 1] = fmaf(p, vv.y, o[d + 1] * al); o[d + 2] = fmaf(p, vv.z, o[d + 2] * al); o[d + 3] = fmaf(p, vv.w, o[d + 3] * al);
            }
            m = mn;
        }
    }
    if (q < N) {
        const size_t row = (size_t)b * N + q;
        float* po = part_o + ((size_t)seg * B * N + row) * E + head * VD;
#pragma unroll
        for (int d = 0; d < VD; ++d) po[d] = o[d];
        float* pm = part_ml + (((size_t)seg * B * heads + (size_t)b * heads + head) * N + q) * 2;
        pm[0] = m; pm[1] = l;
    }
}
__global__ void vit_attn_fwd_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ out, float* __restrict__ rowstat, int N,
                                            int heads, int B) {
    const int E = heads * VD;
    const size_t n = (size_t)B * N * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E), head = e / VD;
        const size_t row = i / E;
        const int q = (int)(row % N), b = (int)(row / N);
        float ms[NSEG], ls[NSEG], M = -3.0e38f;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const float* pm = part_ml + (((size_t)s * B * heads + (size_t)b * heads + head) * N + q) * 2;
            ms[s] = pm[0]; ls[s] = pm[1];
            M = fmaxf(M, ms[s]);
        }
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const float w = ls[s] > 0.f ? __expf(ms[s] - M) : 0.f;
            L += ls[s] * w;
            acc += part_o[((size_t)s * B * N + row) * E + e] * w;
        }
        out[i] = acc / L;
        if ((e & (VD - 1)) == 0) {
            float* rs = rowstat + (((size_t)b * heads + head) * N + q) * 2;
            rs[0] = M; rs[1] = L;
        }
    }
}
// partial dq[seg][q] = sum over the segment's keys of dS[q][k] k[k],  dS = P (dO . v - delta), delta = dO . O
__global__ __launch_bounds__(64) void vit_attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ O,
                                                            const float* __restrict__ rowstat, float* __restrict__ part, int N, int heads, int B) {
    __shared__ __attribute__((aligned(16))) float Ks[64][VD];
    __shared__ __attribute__((aligned(16))) float Vs[64][VD];
    const int nqb = (N + 63) / 64, E = heads * VD;
    int bid = blockIdx.x;
    const int seg = bid % NSEG;
    bid /= NSEG;
    const int qb = bid % nqb;
    bid /= nqb;
    const int head = bid % heads, b = bid / heads;
    const int tid = threadIdx.x, q = qb * 64 + tid, qc = q < N ? q : N - 1;
    float qr[VD], dOr[VD], dq[VD];
    float delta = 0.f;
    {
        const float* src = qkv + ((size_t)b * N + qc) * 3 * E + head * VD;
        const float* dr = dO + ((size_t)b * N + qc) * E + head * VD;
        const float* orow = O + ((size_t)b * N + qc) * E + head * VD;
#pragma unroll
        for (int d = 0; d < VD; ++d) { qr[d] = src[d] * 0.125f; dOr[d] = dr[d]; dq[d] = 0.f; delta = fmaf(dOr[d], orow[d], delta); }
    }
    const float* rs = rowstat + (((size_t)b * heads + head) * N + qc) * 2;
    const float m = rs[0], il = 1.0f / rs[1];
    for (int k0 = seg * 64; k0 < N; k0 += 64 * NSEG) {
        __syncthreads();
        for (int i = 0; i < 64; ++i) {   // row i of the tile, lane = channel: coalesced global reads, conflict-free LDS writes
            const int k = k0 + i, kc = k < N ? k : N - 1;
            const float* ks = qkv + ((size_t)b * N + kc) * 3 * E + E + head * VD;
            Ks[i][tid] = ks[tid];
            Vs[i][tid] = ks[E + tid];
        }
        __syncthreads();
        const int nk = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < nk; ++kk) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int d = 0; d < VD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(&Ks[kk][d]);
                const float4 vv = *reinterpret_cast<const float4*>(&Vs[kk][d]);
                s0 = fmaf(qr[d], kv.x, s0); s1 = fmaf(qr[d + 1], kv.y, s1); s2 = fmaf(qr[d + 2], kv.z, s2); s3 = fmaf(qr[d + 3], kv.w, s3);
                p0 = fmaf(dOr[d], vv.x, p0); p1 = fmaf(dOr[d + 1], vv.y, p1); p2 = fmaf(dOr[d + 2], vv.z, p2); p3 = fmaf(dOr[d + 3], vv.w, p3);
            }
            const float s = (s0 + s1) + (s2 + s3), dp = (p0 + p1) + (p2 + p3);
            const float ds = __expf(s - m) * il * (dp - delta);
#pragma unroll
            for (int d = 0; d < VD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(&Ks[kk][d]);
                dq[d] = fmaf(ds, kv.x, dq[d]); dq[d + 1] = fmaf(ds, kv.y, dq[d + 1]); dq[d + 2] = fmaf(ds, kv.z, dq[d + 2]); dq[d + 3] = fmaf(ds, kv.w, dq[d + 3]);
            }
        }
    }
    if (q < N) {
        float* dst = part + ((size_t)seg * B * N + (size_t)b * N + q) * 3 * E + head * VD;
#pragma unroll
        for (int d = 0; d < VD; ++d) dst[d] = dq[d] * 0.125f;
    }
}
// partial dk[seg][k] = sum over the segment's queries of dS[q][k] q[q] / 8,  dv[seg][k] = sum P[q][k] dO[q]
__global__ __launch_bounds__(64) void vit_attn_bwd_k_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ O,
                                                            const float* __restrict__ rowstat, float* __restrict__ part, int N, int heads, int B) {
    __shared__ __attribute__((aligned(16))) float Qs[64][VD];
    __shared__ __attribute__((aligned(16))) float dOs[64][VD];
    __shared__ float st[64][3];
    const int nkb = (N + 63) / 64, E = heads * VD;
    int bid = blockIdx.x;
    const int seg = bid % NSEG;
    bid /= NSEG;
    const int kb = bid % nkb;
    bid /= nkb;
    const int head = bid % heads, b = bid / heads;
    const int tid = threadIdx.x, k = kb * 64 + tid, kc = k < N ? k : N - 1;
    float kr[VD], vr[VD], dk[VD], dv[VD];
    {
        const float* ks = qkv + ((size_t)b * N + kc) * 3 * E + E + head * VD;
#pragma unroll
        for (int d = 0; d < VD; ++d) { kr[d] = ks[d]; vr[d] = ks[E + d]; dk[d] = 0.f; dv[d] = 0.f; }
    }
    for (int q0 = seg * 64; q0 < N; q0 += 64 * NSEG) {
        __syncthreads();
        for (int i = 0; i < 64; ++i) {   // row i of the tile, lane = channel
            const int q = q0 + i, qc = q < N ? q : N - 1;
            Qs[i][tid] = qkv[((size_t)b * N + qc) * 3 * E + head * VD + tid] * 0.125f;
            const float dv_ = dO[((size_t)b * N + qc) * E + head * VD + tid];
            dOs[i][tid] = dv_;
            float pr = dv_ * O[((size_t)b * N + qc) * E + head * VD + tid];   // delta_i = dO_i . O_i: a wave reduction
            for (int o = 1; o < 64; o <<= 1) pr += __shfl_xor(pr, o);
            if (tid == 0) st[i][2] = pr;
        }
        {
            const int q = q0 + tid, qc = q < N ? q : N - 1;
            const float* rs = rowstat + (((size_t)b * heads + head) * N + qc) * 2;
            st[tid][0] = rs[0]; st[tid][1] = 1.0f / rs[1];
        }
        __syncthreads();
        const int nq = (N - q0) < 64 ? (N - q0) : 64;
        for (int qq = 0; qq < nq; ++qq) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int d = 0; d < VD; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(&Qs[qq][d]);
                const float4 gv = *reinterpret_cast<const float4*>(&dOs[qq][d]);
                s0 = fmaf(qv.x, kr[d], s0); s1 = fmaf(qv.y, kr[d + 1], s1); s2 = fmaf(qv.z, kr[d + 2], s2); s3 = fmaf(qv.w, kr[d + 3], s3);
                p0 = fmaf(gv.x, vr[d], p0); p1 = fmaf(gv.y, vr[d + 1], p1); p2 = fmaf(gv.z, vr[d + 2], p2); p3 = fmaf(gv.w, vr[d + 3], p3);
            }
            const float s = (s0 + s1) + (s2 + s3), dp = (p0 + p1) + (p2 + p3);
            const float p = __expf(s - st[qq][0]) * st[qq][1];
            const float ds = p * (dp - st[qq][2]);
#pragma unroll
            for (int d = 0; d < VD; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(&Qs[qq][d]);
                const float4 gv = *reinterpret_cast<const float4*>(&dOs[qq][d]);
                dk[d] = fmaf(ds, qv.x, dk[d]); dk[d + 1] = fmaf(ds, qv.y, dk[d + 1]); dk[d + 2] = fmaf(ds, qv.z, dk[d + 2]); dk[d + 3] = fmaf(ds, qv.w, dk[d + 3]);
                dv[d] = fmaf(p, gv.x, dv[d]); dv[d + 1] = fmaf(p, gv.y, dv[d + 1]); dv[d + 2] = fmaf(p, gv.z, dv[d + 2]); dv[d + 3] = fmaf(p, gv.w, dv[d + 3]);
            }
        }
    }
    if (k < N) {
        float* dst = part + ((size_t)seg * B * N + (size_t)b * N + k) * 3 * E + E + head * VD;
#pragma unroll
        for (int d = 0; d < VD; ++d) { dst[d] = dk[d]; dst[E + d] = dv[d]; }   // Qs already carries the 1/8
    }
}
// dqkv = sum over the segments, in segment order
__global__ void seg_sum_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NSEG; ++k) s += part[(size_t)k * n + i];
        out[i] = s;
    }
}

}  // namespace

#define TK(name) return check_launch(name, err)

// scratch: B * chunks * 2 * C + B * (C / cpg) * 2 floats.  dx may alias dout.
int th_gn_bwd(const float* dout, const float* x, const float* stats, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, float* scratch, int B,
              int HW, int C, int cpg, int relu, hipStream_t st, std::string& err) {
    if (C > 1024 || C % cpg) { err = "gn_bwd: bad channel count"; return 1; }
    int chunks = 512 / (((C + 63) / 64) * B);
    if (chunks > (HW + 15) / 16) chunks = (HW + 15) / 16;
    if (chunks < 1) chunks = 1;
    float* part = scratch;
    float* gm = scratch + (size_t)B * chunks * 2 * C;
    SOCCDPT_LAUNCH(gn_bwd_part_kernel, dim3((C + 63) / 64, chunks, B), dim3(256), 0, st, dout, x, stats, gamma, beta, part, HW, C, cpg, chunks, relu);
    SOCCDPT_LAUNCH(gn_bwd_reduce_kernel, dim3(B), dim3(1024), 0, st, part, gamma, gm, dgamma, dbeta, B, HW, C, cpg, chunks);
    if (dx) SOCCDPT_LAUNCH(gn_bwd_apply_kernel, dim3(gs_blocks((size_t)B * HW * C)), dim3(256), 0, st, dout, x, stats, gamma, beta, gm, dx, (size_t)B * HW, HW, C, cpg, relu);
    TK("gn_bwd");
}
int th_ws_bwd(const float* dwh, const float* wh, const float* w, float* dw, int Cout, int Cin, int k, int Kpad, float eps, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(ws_bwd_kernel, dim3((unsigned)Cout), dim3(256), 0, st, dwh, wh, w, dw, Cin, k * k, Kpad, eps);
    TK("ws_bwd");
}
int th_conv_w_dgrad_tap(const float* wt, float* out, int N, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(conv_w_dgrad_tap_kernel, dim3(gs_blocks((size_t)N * C * 9)), dim3(256), 0, st, wt, out, N, C);
    TK("conv_w_dgrad_tap");
}
int th_im2colT_gen(const float* halo, float* out, int B, int Hi, int Ho, int C, int stride, int pad, size_t Mp, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(im2colT_gen_kernel, dim3((C + 31) / 32, (unsigned)((Mp + 31) / 32), 9), dim3(256), 0, st, halo, out, B, Hi, Ho, C, stride, 1 - pad, Mp);
    TK("im2colT_gen");
}
int th_col2im(const float* dcol, float* dx, int B, int Hi, int Ho, int C, int stride, int pad, int accumulate, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(col2im_kernel, dim3(gs_blocks((size_t)B * Hi * Hi * C)), dim3(256), 0, st, dcol, dx, B, Hi, Ho, C, stride, pad, accumulate);
    TK("col2im");
}
int th_stride_gather(const float* in, float* out, int B, int Hi, int Ho, int C, int stride, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(stride_gather_kernel, dim3(gs_blocks((size_t)B * Ho * Ho * C)), dim3(256), 0, st, in, out, B, Hi, Ho, C, stride);
    TK("stride_gather");
}
int th_stride_scatter_add(const float* dg, float* dx, int B, int Hi, int Ho, int C, int stride, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(stride_scatter_add_kernel, dim3(gs_blocks((size_t)B * Ho * Ho * C)), dim3(256), 0, st, dg, dx, B, Hi, Ho, C, stride);
    TK("stride_scatter_add");
}
int th_maxpool_bwd(const float* dpool, const float* raw, const float* stats, const float* gamma, const float* beta, uint8_t* idx, float* dA, int B, int Hi, int C, int cpg,
                   hipStream_t st, std::string& err) {
    const int Ho = (Hi + 1) / 2;
    SOCCDPT_LAUNCH(maxpool_argmax_kernel, dim3(gs_blocks((size_t)B * Ho * Ho * C)), dim3(256), 0, st, raw, stats, gamma, beta, idx, B, Hi, Ho, C, cpg);
    SOCCDPT_LAUNCH(maxpool_bwd_kernel, dim3(gs_blocks((size_t)B * Hi * Hi * C)), dim3(256), 0, st, dpool, idx, dA, B, Hi, Ho, C);
    TK("maxpool_bwd");
}
int th_readout_cat(const float* tok, float* cat, int B, int NT, int E, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(readout_cat_kernel, dim3(gs_blocks((size_t)B * (NT - 1) * 2 * E)), dim3(256), 0, st, tok, cat, B, NT, E);
    TK("readout_cat");
}
int th_readout_cat_bwd(const float* dcat, float* dtok, int B, int NT, int E, int accumulate, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(readout_cat_bwd_kernel, dim3(gs_blocks((size_t)B * NT * E)), dim3(256), 0, st, dcat, dtok, B, NT, E, accumulate);
    TK("readout_cat_bwd");
}
int th_tokens_to_patches(const float* dtok, float* dpatch, int B, int NT, int E, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(tokens_to_patches_kernel, dim3(gs_blocks((size_t)B * (NT - 1) * E)), dim3(256), 0, st, dtok, dpatch, B, NT, E);
    TK("tokens_to_patches");
}
// part: NSEG * B * N * 3 * E floats of scratch; rowstat: B * heads * N * 2 floats, written by the forward and read by the backward
size_t th_vit_attention_part_floats(int B, int N, int heads) { return (size_t)NSEG * B * N * 3 * heads * VD; }
int th_vit_attention_fwd(const float* qkv, float* out, float* rowstat, float* part, int B, int N, int heads, hipStream_t st, std::string& err) {
    const unsigned blocks = (unsigned)(B * heads * ((N + 63) / 64) * NSEG);
    float* part_ml = part + (size_t)NSEG * B * N * heads * VD;
    SOCCDPT_LAUNCH(vit_attn_fwd_seg_kernel, dim3(blocks), dim3(64), 0, st, qkv, part, part_ml, N, heads, B);
    SOCCDPT_LAUNCH(vit_attn_fwd_combine_kernel, dim3(gs_blocks((size_t)B * N * heads * VD)), dim3(256), 0, st, part, part_ml, out, rowstat, N, heads, B);
    TK("vit_attention_fwd");
}
int th_vit_attention_bwd(const float* qkv, const float* O, const float* dO, const float* rowstat, float* part, float* dqkv, int B, int N, int heads, hipStream_t st,
                         std::string& err) {
    const unsigned blocks = (unsigned)(B * heads * ((N + 63) / 64) * NSEG);
    SOCCDPT_LAUNCH(vit_attn_bwd_q_kernel, dim3(blocks), dim3(64), 0, st, qkv, dO, O, rowstat, part, N, heads, B);
    SOCCDPT_LAUNCH(vit_attn_bwd_k_kernel, dim3(blocks), dim3(64), 0, st, qkv, dO, O, rowstat, part, N, heads, B);
    const size_t n = (size_t)B * N * 3 * heads * VD;
    SOCCDPT_LAUNCH(seg_sum_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, part, dqkv, n);
    TK("vit_attention_bwd");
}

}  // namespace soccdpt
