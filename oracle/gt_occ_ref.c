/* TEST INFRASTRUCTURE -- scalar C restatement of the reference's ground-truth occupancy generator
 *   /root/reference/SOccDPT/datasets/bdd_helper.py:433-530 (OccupancyProcessor.process_frame: disparity -> depth -> camera points ->
 *   pc_scale / pc_shift -> rotate_points(7,0,0)) and :288-352 (transform_points_to_occupancy_grid_vect: trunc voxel index, strict
 *   0 < idx < size, counting np.add.at, occupancy_grid = counts > point_count_threshold).
 * Float contract (numpy 2.x promotion rules, as the reference runs in this container; pinned by tests/golden/gt_occupancy.npz):
 *   r      = 1.0f / disparity                     (np.reciprocal on float32)
 *   depth  = (float)( (baseline * focal) * (double)r )      baseline*focal is a float64 scalar -> float64 array -> .astype(float32)
 *   rows [0, H/2) are hidden: depth = 0; inf / nan depth -> 0
 *   X = (((double)v - cx) * (double)depth) / fx ;  Y = (((double)u - cy) * (double)depth) / fy ;  Z = (double)depth
 *   p_k = p_k * pc_scale[k] + pc_shift[k]         (separate roundings)
 *   three rotations q_j = fma(p2, M2j, fma(p1, M1j, p0 * M0j)), M = R^T  (OpenBLAS dgemm order for K = 3, probed bit for bit)
 *   idx_k = (long) trunc( (q_k / (double)occ_shape_f32[k]) * (double)grid[k] ) ; skipped if any q is inf / nan
 *   counts[i][j][k][class] += 1 where 0 < i < g0, 0 < j < g1, 0 < k < g2 ; grid = counts > threshold
 * Compiled with -ffp-contract=off; the only fused operations are the explicit fma() calls. */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int H, W, C;
    double fx, fy, cx, cy, base_focal;     /* base_focal = baseline * (fx + fy) / 2 */
    double pc_scale[3], pc_shift[3];
    double rot[27];                        /* Ra^T, Rb^T, Rc^T row-major (the matrices np.dot multiplies by) */
    float occ_shape[3];
    int grid[3];
    float threshold;
} gt_occ_params;

static void rot3(const double* p, const double* M, double* o) {
    for (int j = 0; j < 3; ++j) o[j] = fma(p[2], M[6 + j], fma(p[1], M[3 + j], p[0] * M[j]));
}

/* disparity [H][W] f32, seg_class [H][W] i32 -> depth [H][W] f32 (or NULL), points [H*W][3] f64 (or NULL),
 * counts [g0][g1][g2][C] u32 (zeroed here), grid [g0][g1][g2][C] u8 */
void gt_occupancy_ref(const gt_occ_params* P, const float* disparity, const int32_t* seg_class, float* depth_out, double* points_out,
                      uint32_t* counts, uint8_t* grid) {
    const size_t ncell = (size_t)P->grid[0] * P->grid[1] * P->grid[2] * P->C;
    memset(counts, 0, ncell * sizeof(uint32_t));
    for (int u = 0; u < P->H; ++u) {
        for (int v = 0; v < P->W; ++v) {
            const size_t n = (size_t)u * P->W + v;
            const float r = 1.0f / disparity[n];
            float depth = (float)(P->base_focal * (double)r);
            if (u < P->H / 2) depth = INFINITY;
            if (isinf(depth) || isnan(depth)) depth = 0.0f;
            if (depth_out) depth_out[n] = depth;
            double p[3], a[3], b[3], q[3];
            p[0] = (((double)v - P->cx) * (double)depth) / P->fx;
            p[1] = (((double)u - P->cy) * (double)depth) / P->fy;
            p[2] = (double)depth;
            for (int k = 0; k < 3; ++k) p[k] = p[k] * P->pc_scale[k] + P->pc_shift[k];
            rot3(p, P->rot, a);
            rot3(a, P->rot + 9, b);
            rot3(b, P->rot + 18, q);
            if (points_out) { points_out[3 * n] = q[0]; points_out[3 * n + 1] = q[1]; points_out[3 * n + 2] = q[2]; }
            if (isinf(q[0]) || isinf(q[1]) || isinf(q[2]) || isnan(q[0]) || isnan(q[1]) || isnan(q[2])) continue;
            long idx[3];
            int ok = 1;
            for (int k = 0; k < 3; ++k) {
                const double f = (q[k] / (double)P->occ_shape[k]) * (double)P->grid[k];
                if (!(f > -9.0e18 && f < 9.0e18)) { ok = 0; break; }
                idx[k] = (long)f;  /* trunc toward zero, like ndarray.astype(int) */
                if (!(0 < idx[k] && idx[k] < P->grid[k])) ok = 0;
            }
            const int c = seg_class[n];
            if (!ok || c < 0 || c >= P->C) continue;
            counts[(((size_t)idx[0] * P->grid[1] + idx[1]) * P->grid[2] + idx[2]) * P->C + c] += 1;
        }
    }
    for (size_t i = 0; i < ncell; ++i) grid[i] = (float)counts[i] > P->threshold ? 1 : 0;
}
