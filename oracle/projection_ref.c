/* CPU oracle for the projection stage (TEST INFRASTRUCTURE ONLY).
 *
 * Scalar, float-exact restatement of
 *   SOccDPT.get_semantic_occupancy   /root/reference/SOccDPT/model/SOccDPT.py:264-372
 *   rotate_points                    /root/reference/SOccDPT/model/SOccDPT.py:60-130
 *   SOccDPT.points_to_occupancy_grid /root/reference/SOccDPT/model/SOccDPT.py:374-463
 * as executed by PyTorch's CPU kernels (every operation rounds to f32; the few
 * fused multiply-adds below are the ones the CPU build really performs and are
 * written out with fmaf; compile with -ffp-contract=off).
 *
 * Pinned in the build container: bit-for-bit equal to the reference's own code
 * for inv_up / seg_up / points / occupancy (tests/test_oracle_projection.py,
 * fixtures tests/golden/projection_B2.npz).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path never does.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* torch upsample_bicubic2d (ATen UpSampleKernel.cpp / UpSample.h), A = -0.75.
 * Contraction pattern pinned empirically against torch 2.10 CPU:
 *   source index  real = fma(scale, dst + 0.5, -0.5)
 *   cc1(x) = (fma(1.25, x, -2.25) * x) * x + 1          (last add NOT fused)
 *   cc2(x) = fma(fma(-0.75, x, 3.75), x, -6) * x + 3     (last add NOT fused)
 *   dot4   = fma(v3,w3, fma(v2,w2, fma(v0,w0, v1*w1)))   (inner over x, outer over y)
 */
static inline float cc1(float x) { float t = fmaf(1.25f, x, -2.25f); return (t * x) * x + 1.0f; }
static inline float cc2(float x) { float t = fmaf(-0.75f, x, 3.75f); t = fmaf(t, x, -6.0f); return t * x + 3.0f; }

static void cubic_taps(int dst, int in, int out, int idx[4], float w[4]) {
    float scale = (float)in / (float)out;
    float real = fmaf(scale, (float)dst + 0.5f, -0.5f);
    int64_t ii = (int64_t)floorf(real);
    if (ii > in - 1) ii = in - 1;
    float t = real - (float)ii;
    if (t < 0.0f) t = 0.0f;
    if (t > 1.0f) t = 1.0f;
    w[0] = cc2(t + 1.0f);
    w[1] = cc1(t);
    float u = 1.0f - t;
    w[2] = cc1(u);
    w[3] = cc2(u + 1.0f);
    for (int j = 0; j < 4; ++j) {
        int64_t k = ii + j - 1;
        if (k < 0) k = 0;
        if (k > in - 1) k = in - 1;
        idx[j] = (int)k;
    }
}

static inline float dot4(const float v[4], const float w[4]) {
    return fmaf(v[3], w[3], fmaf(v[2], w[2], fmaf(v[0], w[0], v[1] * w[1])));
}

/* legacy 'nearest': src = min(floor(dst * (in/out)), in-1) */
static inline int nearest_src(int dst, int in, int out) {
    float scale = (float)in / (float)out;
    int64_t s = (int64_t)floorf((float)dst * scale);
    if (s > in - 1) s = in - 1;
    return (int)s;
}

/* einsum('bnm,mj->bnj') on the CPU = sgemm with K = 3:
 * out_j = fma(p2, R[2][j], fma(p1, R[1][j], p0 * R[0][j]))   (pinned: SURVEY.md §8a) */
static inline void rot3(const float p[3], const float* R, float o[3]) {
    for (int j = 0; j < 3; ++j) o[j] = fmaf(p[2], R[6 + j], fmaf(p[1], R[3 + j], p[0] * R[j]));
}

/* cam = {fx, fy, cx, cy}; rot = three row-major 3x3 matrices (Ra, Rb, Rc);
 * occ_bits: (gx*gy*gz*C + 31)/32 words, bit index == linear index of
 * grid[i][j][k][c]; the union over the whole batch (model/SOccDPT.py:449-455).
 * Any output pointer may be NULL. Returns 0. */
int soccdpt_ref_project(const float* inv, const float* seg, int B, int h, int w, int C, int Hc, int Wc,
                        const float* cam, const float* pc_scale, const float* pc_shift, const float* rot,
                        const float* occ_shape, const int* grid, float* inv_up, float* seg_up, float* points,
                        uint32_t* occ_bits) {
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const size_t npix = (size_t)Hc * Wc;
    const size_t nbits = (size_t)grid[0] * grid[1] * grid[2] * C;
    if (occ_bits) memset(occ_bits, 0, ((nbits + 31) / 32) * sizeof(uint32_t));
    const float gsx = (float)grid[0], gsy = (float)grid[1], gsz = (float)grid[2];
    for (int b = 0; b < B; ++b) {
        const float* src = inv + (size_t)b * h * w;
        for (int u = 0; u < Hc; ++u) {
            int iy[4];
            float wy[4];
            cubic_taps(u, h, Hc, iy, wy);
            const int su = nearest_src(u, h, Hc);
            for (int v = 0; v < Wc; ++v) {
                int ix[4];
                float wx[4];
                cubic_taps(v, w, Wc, ix, wx);
                float t[4];
                for (int i = 0; i < 4; ++i) {
                    const float* r = src + (size_t)iy[i] * w;
                    float s[4] = {r[ix[0]], r[ix[1]], r[ix[2]], r[ix[3]]};
                    t[i] = dot4(s, wx);
                }
                float iv = dot4(t, wy);
                if (iv < 1e-8f) iv = 1e-8f; /* NaN compares false and stays NaN */
                float d = 1.0f / iv;
                if (isinf(d) || isnan(d)) d = INFINITY;
                const size_t n = (size_t)u * Wc + v;
                if (inv_up) inv_up[(size_t)b * npix + n] = iv;

                const int sv = nearest_src(v, w, Wc);
                float sem[8];
                for (int c = 0; c < C; ++c) {
                    sem[c] = seg[(((size_t)b * C + c) * h + su) * w + sv];
                    if (seg_up) seg_up[((size_t)b * C + c) * npix + n] = sem[c];
                }

                float p[3];
                p[0] = (((float)v - cx) * d) / fx;
                p[1] = (((float)u - cy) * d) / fy;
                p[2] = d;
                if (n < 3) { /* model/SOccDPT.py:351-353 indexes the POINT axis */
                    for (int k = 0; k < 3; ++k) p[k] = p[k] * pc_scale[n] + pc_shift[n];
                }
                if (points) {
                    float* o = points + ((size_t)b * npix + n) * 3;
                    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
                }
                if (!occ_bits) continue;
                float q[3], r2[3], r3[3];
                rot3(p, rot, q);
                rot3(q, rot + 9, r2);
                rot3(r2, rot + 18, r3);
                if (!(isfinite(r3[0]) && isfinite(r3[1]) && isfinite(r3[2]))) continue;
                const float fi = (r3[0] / occ_shape[0]) * gsx;
                const float fj = (r3[1] / occ_shape[1]) * gsy;
                const float fk = (r3[2] / occ_shape[2]) * gsz;
                /* float -> int64 truncation; only the window (0, grid) matters */
                if (!(fi > -9.0e18f && fi < 9.0e18f && fj > -9.0e18f && fj < 9.0e18f && fk > -9.0e18f && fk < 9.0e18f))
                    continue;
                const int64_t i = (int64_t)fi, j = (int64_t)fj, k = (int64_t)fk;
                if (!(0 < i && i < grid[0] && 0 < j && j < grid[1] && 0 < k && k < grid[2])) continue;
                for (int c = 0; c < C; ++c) {
                    if (sem[c] != 0.0f) {
                        const size_t bit = (((size_t)i * grid[1] + j) * grid[2] + k) * C + c;
                        occ_bits[bit >> 5] |= 1u << (bit & 31);
                    }
                }
            }
        }
    }
    return 0;
}

/* Expand packed bits to the reference's dense fp32 grid [B, gx, gy, gz, C]
 * (every batch row receives the same union grid). */
void soccdpt_ref_expand_occ(const uint32_t* occ_bits, int B, size_t nvox_classes, float* occ) {
    for (int b = 0; b < B; ++b)
        for (size_t n = 0; n < nvox_classes; ++n)
            occ[(size_t)b * nvox_classes + n] = (float)((occ_bits[n >> 5] >> (n & 31)) & 1u);
}
