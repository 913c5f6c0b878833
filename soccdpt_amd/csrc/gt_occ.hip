// Ground-truth occupancy generator on the GPU (the step after the path, training targets of the
// occupancy variants) -- replaces the per-frame numpy pipeline of
//   /root/reference/SOccDPT/datasets/bdd_helper.py:433-530  OccupancyProcessor.process_frame
//       (disparity -> depth, upper half hidden, camera points, pc_scale / pc_shift, rotate_points(7, 0, 0))
//   /root/reference/SOccDPT/datasets/bdd_helper.py:288-352  transform_points_to_occupancy_grid_vect
//       (trunc voxel index, strict 0 < idx < size, COUNTING np.add.at, grid = counts > point_count_threshold)
// One thread per pixel in float64 where numpy computes in float64; the float contract is the one of oracle/gt_occ_ref.c
// (pinned bit for bit by the reference's own class: tests/golden/gt_occupancy.npz).  This translation unit is compiled with
// -ffp-contract=off: the only fused operations are the explicit fma() of the three K = 3 dgemm-order rotations.
#include "kernels.h"

namespace soccdpt {

namespace {
struct GtParams {
    int B, H, W, C;
    double fx, fy, cx, cy, base_focal;
    double pc_scale[3], pc_shift[3];
    double rot[27];
    float occ_shape[3];
    int grid[3];
    float threshold;
};

__device__ __forceinline__ void rot3d(const double* p, const double* M, double* o) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = fma(p[2], M[6 + j], fma(p[1], M[3 + j], p[0] * M[j]));
}

__global__ __launch_bounds__(256) void gt_points_count_kernel(GtParams P, const float* __restrict__ disparity, const int32_t* __restrict__ seg_class,
                                                               float* __restrict__ depth_out, double* __restrict__ points_out,
                                                               uint32_t* __restrict__ counts) {
    const size_t npix = (size_t)P.H * P.W, total = (size_t)P.B * npix;
    const size_t ncell = (size_t)P.grid[0] * P.grid[1] * P.grid[2] * P.C;
    // every lane runs the same number of iterations (the loop bound is rounded up to whole waves) so that the wave-wide
    // run-length aggregation below sees all 64 lanes
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (total + stride - 1) / stride;
    for (size_t it = 0; it < rounds; ++it) {
        const size_t t = it * stride + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        long long cell = -1;   // flat counter index of this pixel's (voxel, class), -1 = contributes nothing
        if (t < total) {
            const size_t b = t / npix, n = t - b * npix;
            const int u = (int)(n / P.W), v = (int)(n - (size_t)u * P.W);
            const float r = 1.0f / disparity[t];
            float depth = (float)(P.base_focal * (double)r);
            if (u < P.H / 2) depth = __builtin_inff();
            if (isinf(depth) || isnan(depth)) depth = 0.0f;
            if (depth_out) depth_out[t] = depth;
            double p[3], a[3], bq[3], q[3];
            p[0] = (((double)v - P.cx) * (double)depth) / P.fx;
            p[1] = (((double)u - P.cy) * (double)depth) / P.fy;
            p[2] = (double)depth;
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = p[k] * P.pc_scale[k] + P.pc_shift[k];
            rot3d(p, P.rot, a);
            rot3d(a, P.rot + 9, bq);
            rot3d(bq, P.rot + 18, q);
            if (points_out) {
                points_out[3 * t] = q[0];
                points_out[3 * t + 1] = q[1];
                points_out[3 * t + 2] = q[2];
            }
            if (!(isinf(q[0]) || isinf(q[1]) || isinf(q[2]) || isnan(q[0]) || isnan(q[1]) || isnan(q[2]))) {
                long long idx[3];
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double f = (q[k] / (double)P.occ_shape[k]) * (double)P.grid[k];
                    if (!(f > -9.0e18 && f < 9.0e18)) { ok = false; idx[k] = 0; continue; }
                    idx[k] = (long long)f;  // trunc toward zero, like ndarray.astype(int)
                    if (!(0 < idx[k] && idx[k] < P.grid[k])) ok = false;
                }
                const int c = seg_class[t];
                if (ok && c >= 0 && c < P.C) cell = (long long)(b * ncell + (((size_t)idx[0] * P.grid[1] + idx[1]) * P.grid[2] + idx[2]) * P.C + c);
            }
        }
        // Neighbouring pixels of a camera row mostly fall into the same voxel: one atomic per RUN of equal cells inside the wave
        // (the count of a run is the distance to the next run head), instead of one contended atomic per pixel (measured at
        // 1080 x 1920, tools/gt_occ_bench.py: 1.93 -> 0.075 ms per frame).  Integer counts: the order of additions is irrelevant.
        const int lane = threadIdx.x & 63;
        const long long left = __shfl_up(cell, 1);
        const bool head = lane == 0 || left != cell;
        const unsigned long long heads = __ballot(head);
        if (head && cell >= 0) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int run = above ? (__ffsll((long long)above)) : (64 - lane);
            atomicAdd(counts + cell, (uint32_t)run);
        }
    }
}

__global__ __launch_bounds__(256) void gt_threshold_kernel(const uint32_t* __restrict__ counts, uint8_t* __restrict__ grid, size_t n, float threshold) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        grid[i] = (float)counts[i] > threshold ? 1 : 0;
}
}  // namespace

int launch_gt_occupancy(int B, int H, int W, int C, const double* intr /*fx,fy,cx,cy,baseline*/, const double* pc_scale, const double* pc_shift,
                        const double* rot27, const float* occ_shape, const int* grid, float threshold, const float* disparity,
                        const int32_t* seg_class, float* depth, double* points, uint32_t* counts, uint8_t* occ, hipStream_t st,
                        std::string& err) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || !intr || !pc_scale || !pc_shift || !rot27 || !occ_shape || !grid || !disparity || !seg_class ||
        !counts || !occ) { err = "gt_occupancy: bad arguments"; return 1; }
    GtParams P;
    P.B = B; P.H = H; P.W = W; P.C = C;
    P.fx = intr[0]; P.fy = intr[1]; P.cx = intr[2]; P.cy = intr[3];
    P.base_focal = intr[4] * ((intr[0] + intr[1]) / 2.0);   // baseline * focal_length, evaluated in float64 like the Python scalars
    for (int k = 0; k < 3; ++k) { P.pc_scale[k] = pc_scale[k]; P.pc_shift[k] = pc_shift[k]; P.occ_shape[k] = occ_shape[k]; P.grid[k] = grid[k]; }
    for (int k = 0; k < 27; ++k) P.rot[k] = rot27[k];
    P.threshold = threshold;
    const size_t ncell = (size_t)B * grid[0] * grid[1] * grid[2] * C;
    hipError_t e = hipMemsetAsync(counts, 0, ncell * sizeof(uint32_t), st);
    if (e != hipSuccess) { err = std::string("gt_occupancy: ") + hipGetErrorString(e); return 1; }
    const size_t total = (size_t)B * H * W;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    SOCCDPT_LAUNCH(gt_points_count_kernel, dim3(blocks), dim3(256), 0, st, P, disparity, seg_class, depth, points, counts);
    unsigned tb = (unsigned)((ncell + 255) / 256);
    if (tb > 8192) tb = 8192;
    SOCCDPT_LAUNCH(gt_threshold_kernel, dim3(tb), dim3(256), 0, st, counts, occ, ncell, threshold);
    return check_launch("gt_occupancy", err);
}

}  // namespace soccdpt
