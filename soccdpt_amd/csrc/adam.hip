// Fused multi-tensor Adam step (the optimiser of the reference's training loop,
// /root/reference/SOccDPT/scripts/train_SOccDPT.py:311-318: torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8,
// weight_decay, amsgrad=False)).  torch's single-tensor path issues ~8 elementwise ATen kernels per parameter tensor and step
// (330 tensors for SOccDPT_V3); here up to 48 tensors share one launch (descriptor table in the kernel arguments) and every
// element is read and written once: p, m, v updated in place from g.
//   g' = g + wd * p;  m = m + (1 - b1) (g' - m);  v = b2 v + (1 - b2) g'^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)            (torch/optim/adam.py _single_tensor_adam)
#include "kernels.h"

namespace soccdpt {

namespace {
constexpr int kAdamChunk = 48;
struct AdamTable {
    float* p[kAdamChunk];
    const float* g[kAdamChunk];
    float* m[kAdamChunk];
    float* v[kAdamChunk];
    unsigned long long n[kAdamChunk];
};
struct AdamScalars {
    float one_minus_b1, b2, one_minus_b2, wd, step_size, inv_bc2_sqrt, eps;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamTable t, AdamScalars s) {
    const int k = blockIdx.y;
    const size_t n = t.n[k];
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        float gi = g[i];
        if (s.wd != 0.f) gi = fmaf(s.wd, pi, gi);
        const float mi = fmaf(s.one_minus_b1, gi - m[i], m[i]);
        const float vi = fmaf(s.one_minus_b2 * gi, gi, s.b2 * v[i]);
        const float denom = sqrtf(vi) * s.inv_bc2_sqrt + s.eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - s.step_size * (mi / denom);
    }
}
}  // namespace

int launch_adam(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                const size_t* sizes, double lr, double beta1, double beta2, double eps, double weight_decay, int step, hipStream_t st,
                std::string& err) {
    if (n_tensors < 0 || step < 1 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !sizes))) { err = "adam: bad arguments"; return 1; }
    // scalar bookkeeping in f64 like the Python floats of torch/optim/adam.py, cast once
    // (hyper-parameters arrive as doubles: 1 - 0.999f is off by 1.3e-5 relative)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars s{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)weight_decay, (float)(lr / bc1),
                  (float)(1.0 / sqrt(bc2)), (float)eps};
    for (int base = 0; base < n_tensors; base += kAdamChunk) {
        AdamTable t{};
        const int cnt = n_tensors - base < kAdamChunk ? n_tensors - base : kAdamChunk;
        size_t nmax = 0;
        for (int k = 0; k < cnt; ++k) {
            if (!params[base + k] || !grads[base + k] || !exp_avg[base + k] || !exp_avg_sq[base + k]) { err = "adam: null tensor pointer"; return 1; }
            t.p[k] = params[base + k]; t.g[k] = grads[base + k]; t.m[k] = exp_avg[base + k]; t.v[k] = exp_avg_sq[base + k];
            t.n[k] = sizes[base + k];
            nmax = sizes[base + k] > nmax ? sizes[base + k] : nmax;
        }
        unsigned gx = (unsigned)((nmax + 255) / 256);
        if (gx > 512) gx = 512;
        if (gx < 1) gx = 1;
        SOCCDPT_LAUNCH(adam_kernel, dim3(gx, cnt), dim3(256), 0, st, t, s);
        if (check_launch("adam", err)) return 1;
    }
    return 0;
}

}  // namespace soccdpt
