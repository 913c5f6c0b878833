// Winograd F(2x2, 3x3) form of the decoder's 3x3 stride-1 convolutions for gfx950 (round 6; numerics: tools/winograd_numerics.py, profiles/r06_winograd_numerics.json).
//
// Replaces, for the launches model.cpp routes here, nn.Conv2d(C, N, 3, padding=1) of ResidualConvUnit_custom (/root/reference/SOccDPT/model/blocks.py:391-414; call sites
// FeatureFusionBlock_custom.forward :466-497, DPT.forward model/dpt.py:152-182) with the same fused epilogues igemm.hip gives them: + bias, + residual (f32), + bilinearly
// sampled coarser-level residual, ReLU, f32 output and / or 16-bit / x3 operand copy (plain or zero-halo NHWC).
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A      per 4 x 4 input tile d -> 2 x 2 output pixels: 16 products per (tile, cin, cout) instead of 36.
//
// The 16 positions (i, j) of the transformed domain are 16 independent GEMMs  M_ij[tile][n] = sum_c V_ij[tile][c] U_ij[c][n].  U = G g G^T is prepared once per weight
// load (wino_w_kernel: transformed in f32, then rounded to the 16-bit operand format, laid out [C/32][16][N][32] so that a k-chunk of a position is contiguous); V = B^T d B
// is built on the fly from the 16-bit zero-halo input image -- exactly in f32, rounded once to the operand format -- and never touches HBM.
//
// Decomposition (CDNA4):
//  * workgroup = 8 waves = 16 x 16 output pixels (8 x 8 Winograd tiles) x 64 output channels, all 16 positions.  16 positions x 64 tiles x 64 channels are 65536
//    accumulators: they fit the register file only when the positions are dealt over the waves.  Wave (ri, th) owns ROW ri of the 4 x 4 position grid (4 positions)
//    for tile half th (32 tiles = two MFMA column fragments) and all 64 channels (four row fragments): 4 x 2 x 4 accumulator fragments = 128 registers per lane.
//  * v_mfma_f32_16x16x32: A = U_ij (rows = 16 channels), B = V_ij (columns = 16 tiles): a lane then owns one tile with 4 consecutive channels per fragment, so the output
//    transform is lane-local per channel and the stores are 4 channels wide.
//  * per k-chunk of 32 input channels the 18 x 18 pixel patch (26 KB) and the chunk's U rows (16 positions x 64 channels x 64 bytes = 64 KB) sit in LDS.  The U rows arrive
//    by LDS-DMA (global_load_lds_dwordx4: no register round trip -- a first version staged them through 32 registers per lane, which the compiler put in scratch, so
//    every chunk paid the full load latency: 157 us against 53 us for the direct launch) into one of two buffers, the next chunk's under the current chunk's arithmetic;
//    the 16-byte chunks of a row are XOR-swizzled by (row >> 2) & 3 through the SOURCE address each lane reads (the DMA destination is lane-contiguous).  Patch pixels are
//    padded to 80 bytes; the pixels of one fragment are two pixels apart, which alone would leave only the even 16-byte bank groups in use: pixels of odd tile rows rotate
//    their four chunks by one slot into the pad (two-way conflicts otherwise).  The patch goes through 3 registers per lane.
//  * the input transform needs, for position row ri, only the TWO patch rows with a non-zero B^T[ri][.] coefficient: T[b] = d[a1][b] +- d[a2][b] (b = 0..3), then
//    V_0 = T0 - T2, V_1 = T1 + T2, V_2 = T2 - T1, V_3 = T1 - T3: 8 pixel reads and 64 additions per tile and 8 channels for the wave's four positions.
//  * after the k loop each wave folds its four positions along j (Z_x = sum_j A^T[x][j] M_ij: two values), the rows meet in LDS (128 KB of f32, the whole array reused) and
//    every wave finishes a quarter of the channels: Y[y][x] = sum_i A^T[y][i] Z_i,x, then the epilogue.
//  * two barriers per chunk: everybody done with the patch / the DMA landed, then the next patch written.
#include <stdlib.h>

#include "gelu.h"
#include "half16.h"
#include "igemm.h"
#include "kernels.h"

namespace soccdpt {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4w;

constexpr int W_THREADS = 512;
constexpr int W_ROWB = 80;                          // bytes per LDS row (64 of data + 16 of padding)
constexpr int W_PATCH = 18 * 18 * W_ROWB;           // 25920
constexpr int W_UOFF = (W_PATCH + 1023) / 1024 * 1024; // 26624
constexpr int W_UBYTES = 16 * 64 * 64;              // 65536 per buffer, two buffers
constexpr int W_XCH = 8 * 2 * 2 * 4 * 64 * 16;      // 131072: [wave][x][tf][nf][lane] float4
constexpr int W_LDS = (W_UOFF + 2 * W_UBYTES) > W_XCH ? (W_UOFF + 2 * W_UBYTES) : W_XCH;   // 157696

struct WinoDev {
    const uint16_t* X;      // zero-halo NHWC [B][H + 2][W + 2][C], 16-bit
    const uint16_t* U;      // [C / 32][16][N][32], 16-bit
    int B, H, W, C, N;
    const float* bias;      // [N] or null
    const float* res1;      // [M][N] f32 or null
    const float* res2;      // [B][res2_h][res2_w][N] f32, sampled bilinearly (align_corners) at the output pixel; or null
    int res2_h, res2_w;
    int act;                // ACT_NONE / ACT_RELU (on the operand copy; on the f32 output too when act_on_f32)
    int act_on_f32;
    float* out_f32;         // [M][N] or null
    void* out_op;           // 16-bit or x3 operand copy, plain [M][N] or zero-halo [B][H + 2][W + 2][N]; or null
    int out_halo;
    int out_x3;             // out_op is written in the x3 format (fp16 kernels)
    unsigned long long* stamps;   // diagnostics (tools/wino_bench.py): 8 x s_memrealtime per workgroup; null in the forward
};

template <bool F16>
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[2 * k] = h_lo<F16>(u[k]); f[2 * k + 1] = h_hi<F16>(u[k]); }
}

template <bool F16>
__global__ __launch_bounds__(W_THREADS) void wino_conv_kernel(WinoDev d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ps = smem;
    char* Us = smem + W_UOFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = wave & 3, th = wave >> 2;
    const int t16 = lane & 15, kq = lane >> 4;
    const int nblk = d.N / 64;
    int bid = blockIdx.x;
    const int n0 = (bid % nblk) * 64;
    bid /= nblk;
    const int tbw = d.W / 16, tbh = d.H / 16;
    const int tbx = bid % tbw;
    bid /= tbw;
    const int tby = bid % tbh;
    const int b = bid / tbh;
    const int Wp = d.W + 2;
    const int nchunks = d.C / 32;

    // ---- staging.  U rows: LDS-DMA, lane l of DMA instruction (it, wave) fills 16-byte chunk q = (it * 8 + wave) * 64 + l of the buffer = (row q >> 2, position q & 3),
    // reading the row's chunk (q & 3) ^ ((row >> 2) & 3).  Patch: global -> 3 registers -> LDS ----
    uint4 pr0, pr1, pr2;   // patch staging registers (scalars, not an array: an array captured by the lambdas below stayed in scratch)
    const uint16_t* xbase = d.X + ((size_t)(b * (d.H + 2) + tby * 16) * Wp + tbx * 16) * d.C;
    int usrc[8];   // element offset of this lane's source chunk inside a k-chunk's [16][N][32] slab, per DMA instruction
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int q = (it * 8 + wave) * 64 + lane, row = q >> 2, slot = (q & 3) ^ ((row >> 2) & 3);
        usrc[it] = ((row >> 6) * d.N + (row & 63)) * 32 + slot * 8;
    }
    auto udma = [&](int ch, int buf) {
        const uint16_t* ub = d.U + ((size_t)ch * 16 * d.N + n0) * 32;
#pragma unroll
        for (int it = 0; it < 8; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + usrc[it]),
                                             (__attribute__((address_space(3))) void*)(Us + buf * W_UBYTES + (it * 8 + wave) * 1024), 16, 0, 0);
    };
    int psrc[3], pdst[3];   // per staging round: element offset of the lane's source chunk in the image (chunk 0), byte offset of its LDS slot (< 0: no work in round 2)
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int idx = it * W_THREADS + tid;
        const bool ok = idx < 18 * 18 * 4;
        const int pix = ok ? idx >> 2 : 0, slot = idx & 3, py = pix / 18, px = pix - py * 18;
        psrc[it] = (py * Wp + px) * d.C + slot * 8;
        pdst[it] = ok ? pix * W_ROWB + (slot + ((py >> 1) & 1)) * 16 : -1;
    }
#define W_PLOAD(ch_) do { pr0 = *reinterpret_cast<const uint4*>(xbase + psrc[0] + (ch_) * 32); pr1 = *reinterpret_cast<const uint4*>(xbase + psrc[1] + (ch_) * 32); \
                          pr2 = *reinterpret_cast<const uint4*>(xbase + psrc[2] + (ch_) * 32); } while (0)   /* unconditional: lanes without work re-read pixel 0 */
#define W_PSTORE() do { *reinterpret_cast<uint4*>(Ps + pdst[0]) = pr0; *reinterpret_cast<uint4*>(Ps + pdst[1]) = pr1; \
                        if (pdst[2] >= 0) *reinterpret_cast<uint4*>(Ps + pdst[2]) = pr2; } while (0)

    f32x4w acc[4][2][4];   // [position j of row ri][tile fragment][channel fragment]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tf = 0; tf < 2; ++tf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[j][tf][nf] = f32x4w{0.f, 0.f, 0.f, 0.f};

    // patch rows of position row ri: B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] -> T = d[a1] + sg * d[a2]
    const int a1 = ri == 0 ? 0 : (ri == 2 ? 2 : 1);
    const int a2 = ri == 0 ? 2 : (ri == 1 ? 2 : (ri == 2 ? 1 : 3));
    const float sg = ri == 1 ? 1.f : -1.f;
    int poff[2][2];   // byte offset of pixel (2 trow + a, 2 tcol) of this lane's tile in fragment tf, rows a1 / a2, with the lane's chunk slot
#pragma unroll
    for (int tf = 0; tf < 2; ++tf) {
        const int trow = 4 * th + 2 * tf + (t16 >> 3), tcol = t16 & 7;
        const int pa[2] = {2 * trow + a1, 2 * trow + a2};
#pragma unroll
        for (int k = 0; k < 2; ++k) poff[tf][k] = (pa[k] * 18 + 2 * tcol) * W_ROWB + (kq + ((pa[k] >> 1) & 1)) * 16;
    }
    const int ufrag = ((4 * ri) * 64 + t16) * 64 + ((kq ^ ((t16 >> 2) & 3)) * 16);   // row (position 4 ri + j, channel 16 nf + t16): + (j * 64 + nf * 16) * 64; swizzle bits = the row's

    if (d.stamps && tid == 0) d.stamps[8 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    udma(0, 0);
    W_PLOAD(0);
    W_PSTORE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (d.stamps && tid == 0) d.stamps[8 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (d.stamps && tid == 0 && ch == 4) d.stamps[8 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        const char* Ub = Us + (ch & 1) * W_UBYTES + ufrag;
        if (more) { udma(ch + 1, (ch & 1) ^ 1); W_PLOAD(ch + 1); }
        // ---- input transform: V_j (j = 0..3) of both tile fragments, 8 channels per lane ----
        h16x8 vfrag[2][4];
        if constexpr (F16) {
            // packed fp16 arithmetic (v_pk_fma_f16 / v_pk_add_f16: 32 instructions per fragment; the f32 form -- 32 v_fma_mix + 64 additions + 64 conversions + packing --
            // made the chunk VALU-bound: 0.9 us of a 3 us chunk by the stamps).  T = d[a1] +- d[a2] and V = T +- T' each round once to fp16: the second rounding adds about
            // as much variance as the single rounding of the f32 form (measured by tests/test_kernels_gpu.py against float64)
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 sg2 = {(_Float16)sg, (_Float16)sg};
#pragma unroll
            for (int tf = 0; tf < 2; ++tf) {
                h2 T[4][4];
#pragma unroll
                for (int bcol = 0; bcol < 4; ++bcol) {
                    const uint4 q1 = *reinterpret_cast<const uint4*>(Ps + poff[tf][0] + bcol * W_ROWB), q2 = *reinterpret_cast<const uint4*>(Ps + poff[tf][1] + bcol * W_ROWB);
                    const uint32_t u1[4] = {q1.x, q1.y, q1.z, q1.w}, u2[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) T[bcol][k] = __builtin_elementwise_fma(__builtin_bit_cast(h2, u2[k]), sg2, __builtin_bit_cast(h2, u1[k]));
                }
                uint32_t pk[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pk[0][k] = __builtin_bit_cast(uint32_t, (h2)(T[0][k] - T[2][k]));
                    pk[1][k] = __builtin_bit_cast(uint32_t, (h2)(T[1][k] + T[2][k]));
                    pk[2][k] = __builtin_bit_cast(uint32_t, (h2)(T[2][k] - T[1][k]));
                    pk[3][k] = __builtin_bit_cast(uint32_t, (h2)(T[1][k] - T[3][k]));
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) vfrag[tf][jj] = __builtin_bit_cast(h16x8, make_uint4(pk[jj][0], pk[jj][1], pk[jj][2], pk[jj][3]));
            }
        } else {
#pragma unroll
        for (int tf = 0; tf < 2; ++tf) {
            float T[4][8];
#pragma unroll
            for (int bcol = 0; bcol < 4; ++bcol) {
                float p1[8], p2[8];
                unpack8<F16>(*reinterpret_cast<const uint4*>(Ps + poff[tf][0] + bcol * W_ROWB), p1);
                unpack8<F16>(*reinterpret_cast<const uint4*>(Ps + poff[tf][1] + bcol * W_ROWB), p2);
#pragma unroll
                for (int k = 0; k < 8; ++k) T[bcol][k] = fmaf(sg, p2[k], p1[k]);   // exact: +-1 x a 16-bit value + a 16-bit value in f32
            }
            uint32_t pk[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pk[0][k] = (uint32_t)f2h_inrange<F16>(T[0][2 * k] - T[2][2 * k]) | ((uint32_t)f2h_inrange<F16>(T[0][2 * k + 1] - T[2][2 * k + 1]) << 16);
                pk[1][k] = (uint32_t)f2h_inrange<F16>(T[1][2 * k] + T[2][2 * k]) | ((uint32_t)f2h_inrange<F16>(T[1][2 * k + 1] + T[2][2 * k + 1]) << 16);
                pk[2][k] = (uint32_t)f2h_inrange<F16>(T[2][2 * k] - T[1][2 * k]) | ((uint32_t)f2h_inrange<F16>(T[2][2 * k + 1] - T[1][2 * k + 1]) << 16);
                pk[3][k] = (uint32_t)f2h_inrange<F16>(T[1][2 * k] - T[3][2 * k]) | ((uint32_t)f2h_inrange<F16>(T[1][2 * k + 1] - T[3][2 * k + 1]) << 16);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) vfrag[tf][jj] = __builtin_bit_cast(h16x8, make_uint4(pk[jj][0], pk[jj][1], pk[jj][2], pk[jj][3]));
        }
        }
        if (d.stamps && tid == 0 && ch == 4) d.stamps[8 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
        // ---- 4 positions x 4 channel fragments x 2 tile fragments ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const h16x8 uf = *reinterpret_cast<const h16x8*>(Ub + (j * 64 + nf * 16) * 64);
                acc[j][0][nf] = mfma_16x16x32<F16>(uf, vfrag[0][j], acc[j][0][nf]);
                acc[j][1][nf] = mfma_16x16x32<F16>(uf, vfrag[1][j], acc[j][1][nf]);
            }
        }
        if (d.stamps && tid == 0 && ch == 4) d.stamps[8 * (size_t)blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next U buffer has landed (and its patch registers)
        __syncthreads();
        if (d.stamps && tid == 0 && ch == 4) d.stamps[8 * (size_t)blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();          // everybody has read this chunk's patch (also before the exchange below reuses the array); everybody's DMA has landed
        if (more) {
            W_PSTORE();
            __syncthreads();
        }
    }

    if (d.stamps && tid == 0) d.stamps[8 * (size_t)blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
    // ---- fold the wave's four positions along j: Z_x = sum_j A^T[x][j] M_ij, A^T = [[1,1,1,0],[0,1,-1,-1]]; park them for the other rows ----
    float4* Zs = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int tf = 0; tf < 2; ++tf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const f32x4w z0 = acc[0][tf][nf] + acc[1][tf][nf] + acc[2][tf][nf];
            const f32x4w z1 = acc[1][tf][nf] - acc[2][tf][nf] - acc[3][tf][nf];
            Zs[((((ri * 2 + th) * 2 + 0) * 2 + tf) * 4 + nf) * 64 + lane] = make_float4(z0[0], z0[1], z0[2], z0[3]);
            Zs[((((ri * 2 + th) * 2 + 1) * 2 + tf) * 4 + nf) * 64 + lane] = make_float4(z1[0], z1[1], z1[2], z1[3]);
        }
    __syncthreads();
    // ---- this wave finishes channel fragment nf = ri of its tile half: Y[y][x] = sum_i A^T[y][i] Z_i,x, then the epilogue of the pixel's 4 channels ----
    const int nf = ri;
    const int n = n0 + 16 * nf + 4 * kq;
    float4 bia = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias) bia = *reinterpret_cast<const float4*>(d.bias + n);
#pragma unroll
    for (int tf = 0; tf < 2; ++tf) {
        const int trow = 4 * th + 2 * tf + (t16 >> 3), tcol = t16 & 7;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float4 z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = Zs[((((i * 2 + th) * 2 + x) * 2 + tf) * 4 + nf) * 64 + lane];
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                float v[4];
                if (y == 0) { v[0] = z[0].x + z[1].x + z[2].x; v[1] = z[0].y + z[1].y + z[2].y; v[2] = z[0].z + z[1].z + z[2].z; v[3] = z[0].w + z[1].w + z[2].w; }
                else { v[0] = z[1].x - z[2].x - z[3].x; v[1] = z[1].y - z[2].y - z[3].y; v[2] = z[1].z - z[2].z - z[3].z; v[3] = z[1].w - z[2].w - z[3].w; }
                const int oy = tby * 16 + 2 * trow + y, ox = tbx * 16 + 2 * tcol + x;
                const size_t m = ((size_t)b * d.H + oy) * d.W + ox;
                const size_t orow = m * d.N;
                v[0] += bia.x; v[1] += bia.y; v[2] += bia.z; v[3] += bia.w;
                if (d.res1) {
                    const float4 r4 = *reinterpret_cast<const float4*>(d.res1 + orow + n);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (d.res2) {   // the same expression, in the same order, as igemm's sampled-residual epilogue
                    const float sy = d.H > 1 ? (float)(d.res2_h - 1) / (float)(d.H - 1) : 0.f;
                    const float sx = d.W > 1 ? (float)(d.res2_w - 1) / (float)(d.W - 1) : 0.f;
                    const float fy = sy * (float)oy, fx = sx * (float)ox;
                    const int y0 = (int)fy, x0 = (int)fx;
                    const int y1 = y0 + (y0 < d.res2_h - 1), x1 = x0 + (x0 < d.res2_w - 1);
                    const float uly = fy - (float)y0, ulx = fx - (float)x0;
                    const size_t pb = (size_t)b * d.res2_h * d.res2_w;
                    const float4 a00 = *reinterpret_cast<const float4*>(d.res2 + (pb + (size_t)y0 * d.res2_w + x0) * d.N + n);
                    const float4 a01 = *reinterpret_cast<const float4*>(d.res2 + (pb + (size_t)y0 * d.res2_w + x1) * d.N + n);
                    const float4 a10 = *reinterpret_cast<const float4*>(d.res2 + (pb + (size_t)y1 * d.res2_w + x0) * d.N + n);
                    const float4 a11 = *reinterpret_cast<const float4*>(d.res2 + (pb + (size_t)y1 * d.res2_w + x1) * d.N + n);
                    const float hy = 1.f - uly, hx = 1.f - ulx;
                    v[0] += hy * (hx * a00.x + ulx * a01.x) + uly * (hx * a10.x + ulx * a11.x);
                    v[1] += hy * (hx * a00.y + ulx * a01.y) + uly * (hx * a10.y + ulx * a11.y);
                    v[2] += hy * (hx * a00.z + ulx * a01.z) + uly * (hx * a10.z + ulx * a11.z);
                    v[3] += hy * (hx * a00.w + ulx * a01.w) + uly * (hx * a10.w + ulx * a11.w);
                }
                float a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = d.act == ACT_RELU ? fmaxf(v[r], 0.f) : v[r];
                if (d.out_f32) {
                    const bool af = d.act_on_f32 != 0;
                    *reinterpret_cast<float4*>(d.out_f32 + orow + n) = make_float4(af ? a[0] : v[0], af ? a[1] : v[1], af ? a[2] : v[2], af ? a[3] : v[3]);
                }
                if (d.out_op) {
                    const size_t e = d.out_halo ? (((size_t)(b * (d.H + 2) + oy + 1) * Wp + ox + 1) * d.N + n) : (orow + n);
                    if (F16 && d.out_x3) x3_store4(d.out_op, e, a[0], a[1], a[2], a[3]);
                    else {
                        uint2 p;
                        p.x = pack_h2<F16>(a[0], a[1]);
                        p.y = pack_h2<F16>(a[2], a[3]);
                        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.out_op) + e) = p;
                    }
                }
            }
        }
    }
    if (d.stamps && tid == 0) d.stamps[8 * (size_t)blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();
}

#undef W_PLOAD
#undef W_PSTORE

// [N][C][3][3] f32 (x scale[n]) -> U = G g G^T in f32 -> 16-bit [C / 32][16][N][32]
template <bool F16>
__global__ __launch_bounds__(256) void wino_w_kernel(const float* __restrict__ w, const float* __restrict__ scale, uint16_t* __restrict__ U, int N, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * C) return;
    const int c = (int)(i % C), n = (int)(i / C);
    float g[3][3];
    const float s = scale ? scale[n] : 1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) g[a][bb] = w[((size_t)n * C + c) * 9 + a * 3 + bb] * s;
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
        t[0][bb] = g[0][bb];
        t[1][bb] = 0.5f * (g[0][bb] + g[1][bb] + g[2][bb]);
        t[2][bb] = 0.5f * (g[0][bb] - g[1][bb] + g[2][bb]);
        t[3][bb] = g[2][bb];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) U[(((size_t)(c >> 5) * 16 + (a * 4 + j)) * N + n) * 32 + (c & 31)] = f2h<F16>(u[j]);
    }
}

}  // namespace

bool wino_supported(int H, int W, int C, int N) { return H % 16 == 0 && W % 16 == 0 && H >= 16 && W >= 16 && C % 32 == 0 && C >= 32 && N % 64 == 0; }
size_t wino_weight_elems(int N, int C) { return (size_t)16 * N * C; }

int launch_wino_weights(const float* w, const float* scale, void* U, int hf, int N, int C, hipStream_t st, std::string& err) {
    if (C % 32) { err = "wino_weights: C must be a multiple of 32"; return 1; }
    const size_t n = (size_t)N * C;
    if (hf) SOCCDPT_LAUNCH(wino_w_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, scale, static_cast<uint16_t*>(U), N, C);
    else SOCCDPT_LAUNCH(wino_w_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, scale, static_cast<uint16_t*>(U), N, C);
    return check_launch("wino_weights", err);
}

int launch_wino_conv(const WinoArgs& a, hipStream_t st, std::string& err) {
    if (!wino_supported(a.H, a.W, a.C, a.N)) { err = "wino_conv: geometry not supported (H, W multiples of 16; C of 32; N of 64)"; return 1; }
    if (!a.X || !a.U || (!a.out_f32 && !a.out_op)) { err = "wino_conv: null operand / no output"; return 1; }
    if (a.out_x3 && !a.hf) { err = "wino_conv: the x3 operand copy belongs to the fp16 kernel"; return 1; }
    if (a.act != ACT_NONE && a.act != ACT_RELU) { err = "wino_conv: ReLU or no activation"; return 1; }
    WinoDev d;
    d.X = static_cast<const uint16_t*>(a.X); d.U = static_cast<const uint16_t*>(a.U); d.B = a.B; d.H = a.H; d.W = a.W; d.C = a.C; d.N = a.N;
    d.bias = a.bias; d.res1 = a.res1; d.res2 = a.res2; d.res2_h = a.res2_h; d.res2_w = a.res2_w; d.act = a.act; d.act_on_f32 = a.act_on_f32;
    d.out_f32 = a.out_f32; d.out_op = a.out_op; d.out_halo = a.out_halo; d.out_x3 = a.out_x3; d.stamps = a.stamps;
    static PerDeviceOnce attr;
    if (attr.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        if (e != hipSuccess) { err = std::string("wino_conv: hipFuncSetAttribute: ") + hipGetErrorString(e); return 1; }
        attr.done();
    }
    const unsigned blocks = (unsigned)(a.B * (a.H / 16) * (a.W / 16) * (a.N / 64));
    if (a.hf) SOCCDPT_LAUNCH(wino_conv_kernel<true>, dim3(blocks), dim3(W_THREADS), W_LDS, st, d);
    else SOCCDPT_LAUNCH(wino_conv_kernel<false>, dim3(blocks), dim3(W_THREADS), W_LDS, st, d);
    return check_launch("wino_conv", err);
}

}  // namespace soccdpt
