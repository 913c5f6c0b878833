// Fused tail of the depth head for gfx950 (bf16 mode):
//   Interpolate(x2, bilinear, align_corners=True) -> Conv2d(128, 32, 3, pad 1) + bias -> ReLU -> Conv2d(32, 1, 1) + bias -> ReLU
//   (/root/reference/SOccDPT/model/dpt.py:207-216), input d1 = output of Conv2d(256,128,3) at half resolution.
// The unfused form writes and re-reads a 134 MB up-sampled bf16 image (B=8) and, as an implicit GEMM with N = 32,
// stages every input pixel 9 times for only 32 output channels (25 FLOP per staged byte).  Here a persistent
// workgroup keeps the 32 x 1152 weight matrix in REGISTERS (each wave its 16 channels as 36 MFMA fragments), builds the (8+2) x (16+2) up-sampled halo patch of
// each 8 x 16 output tile directly from the low-resolution map (4 x 16-byte loads + lerp per 8 channels), and runs
// the 9 taps x 4 k-steps of v_mfma_f32_16x16x32_bf16 out of LDS: no up-sampled image in HBM, ~12x fewer staged bytes.
#include "half16.h"
#include "kernels.h"

namespace soccdpt {

typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int CIN = 128, TH = 8, TW = 16, PH = TH + 2, PW = TW + 2;
constexpr int KTOT = 9 * CIN;                 // 1152
constexpr int P_ROWB = CIN * 2 + 16;          // 272 B per patch pixel, same trick

constexpr int P_BYTES = PH * PW * P_ROWB;     // 48960
constexpr int SRH = 8, SRW = 12;              // low-res source window of one patch: <= 7 x 11 pixels (+1 spare)
constexpr int S_ROWB = CIN * 2;               // 256 B per source pixel
constexpr int S_BYTES = SRH * SRW * S_ROWB;   // 24576
constexpr int T_BYTES = 512;                   // per-tile row / column interpolation tables
constexpr int R_BYTES = TH * TW * 4;            // cross-wave reduction of the two channel tiles
constexpr int LDS_BYTES = 2 * P_BYTES + 2 * S_BYTES + T_BYTES + R_BYTES;  // 148096: two patches and two source windows (round 6: the next tile's patch is built while this one is multiplied)
constexpr int NTHR = 512;                      // 8 waves: one output row of the 8 x 16 tile each
}  // namespace

// d1: 16-bit [B][h][w][128] plain NHWC; wt: 16-bit [32][9*128] tap-major; out: f32 [B][2h][2w].  F16: operands are fp16, else bf16.
//
// Round 6 (profiles/r06*_pmc_stall.json: 45 % of the wave cycles of this kernel issue VALU work -- the bilinear build of the patch, ~820
// wave instructions per wave and tile -- against 16 % for its 72 MFMAs, and until round 5 a barrier kept the two phases apart for the whole workgroup):
// the patch is double-buffered and a tile's MFMAs run in the same barrier interval as the build of the NEXT tile's patch; the two waves of a SIMD take
// the two phases in opposite order (waves 0-3 build first, waves 4-7 multiply first), so a SIMD always has one wave for its VALU and one for its
// matrix core.  Two barriers per tile instead of four.
template <bool F16>
__global__ __launch_bounds__(512) void depth_tail_kernel(const bf16_t* __restrict__ d1, const bf16_t* __restrict__ wt, const float* __restrict__ bias,
                                                          const float* __restrict__ w4, float b4, float* __restrict__ out, int B, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ps0 = smem;
    char* Ss0 = smem + 2 * P_BYTES;
    int* Ti = reinterpret_cast<int*>(smem + 2 * P_BYTES + 2 * S_BYTES);  // [0..9] row: src offset (bytes) of the upper row | valid<<30 ; [32..49] col
    float* Tf = reinterpret_cast<float*>(Ti) + 64;                          // [0..9] ly ; [32..49] lx
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = 2 * h, W = 2 * w;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int ntiles = B * tiles_y * tiles_x;
    // ---- weights: this wave's 16 output channels x 1152 k as 36 MFMA A-fragments, resident in registers for the whole
    // persistent loop (144 VGPRs; an LDS-resident copy made every wave re-read 74 KB per tile and the kernel LDS-read-bound)
    const int frow = lane & 15, fq = lane >> 4;
    const int ntile = wave & 1, rowpair = wave >> 1;  // 8 waves = 2 channel tiles x 4 row pairs of the 8 x 16 output tile
    h16x8 wf[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) wf[k] = *reinterpret_cast<const h16x8*>(wt + (size_t)(ntile * 16 + frow) * KTOT + k * 32 + fq * 8);
    const float sy = (float)(h - 1) / (float)(H - 1), sx = (float)(w - 1) / (float)(W - 1);
    const float4 bia = *reinterpret_cast<const float4*>(bias + ntile * 16 + fq * 4);
    const float4 w4v = *reinterpret_cast<const float4*>(w4 + ntile * 16 + fq * 4);
    float* red = reinterpret_cast<float*>(smem + 2 * P_BYTES + 2 * S_BYTES + T_BYTES);  // [8 rows][16 px] partial dots of channel tile 1
    // low-res source window of a tile -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, nobody waits for it until the tile's turn): the window's
    // 96 pixels x 256 B are 24 wave-sized pieces of 4 pixels, three per wave; lane = (pixel of the piece, 16-byte chunk)
    auto issue_window = [&](int tile, int buf) {
        const int tx = tile % tiles_x;
        int r = tile / tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const bf16_t* src = d1 + (size_t)b * h * w * CIN;
        const int Yc0 = y0 < 0 ? 0 : y0, Xc0 = x0 < 0 ? 0 : x0;
        const int sy0 = (int)(sy * (float)Yc0), sx0 = (int)(sx * (float)Xc0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave * 3 + i, p = piece * 4 + (lane >> 4), ch = lane & 15;
            int yy = sy0 + p / SRW, xx = sx0 + p % SRW;
            yy = yy > h - 1 ? h - 1 : yy;
            xx = xx > w - 1 ? w - 1 : xx;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)yy * w + xx) * CIN + ch * 8),
                                             (__attribute__((address_space(3))) void*)(Ss0 + buf * S_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    // interpolation tables of a tile (a few threads; read by the build after the next barrier)
    auto tables = [&](int tile) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;  // up-sampled coordinates of the patch origin
        const int Yc0 = y0 < 0 ? 0 : y0, Xc0 = x0 < 0 ? 0 : x0;
        const int sy0 = (int)(sy * (float)Yc0), sx0 = (int)(sx * (float)Xc0);  // first source row / column any patch pixel touches
        if (tid < PH) {  // rows: byte offset of source row yy0 inside the window, step to yy1, weight
            const int Y = y0 + tid;
            const bool ok = Y >= 0 && Y < H;
            const float fy = sy * (float)(ok ? Y : 0);
            const int yy0 = (int)fy;
            Ti[tid] = ((yy0 - sy0) * SRW * S_ROWB) | ((yy0 < h - 1) ? (1 << 28) : 0) | (ok ? (1 << 30) : 0);
            Tf[tid] = fy - (float)yy0;
        } else if (tid >= 32 && tid < 32 + PW) {
            const int X = x0 + tid - 32;
            const bool ok = X >= 0 && X < W;
            const float fx = sx * (float)(ok ? X : 0);
            const int xx0 = (int)fx;
            Ti[tid] = ((xx0 - sx0) * S_ROWB) | ((xx0 < w - 1) ? (1 << 28) : 0) | (ok ? (1 << 30) : 0);
            Tf[tid] = fx - (float)xx0;
        }
    };
    // the up-sampled halo patch from a staged window: thread = fixed 8-channel chunk, strided pixels
    auto build = [&](const char* Ss, char* Ps) {
        const int ch = tid & 15;
        const char* sb = Ss + ch * 16;
        for (int p = tid >> 4; p < PH * PW; p += NTHR / 16) {
            const int py = p / PW, px = p - py * PW;
            const int ty_ = Ti[py], tx_ = Ti[32 + px];
            uint4 o = make_uint4(0, 0, 0, 0);  // conv zero padding outside the image
            if ((ty_ & tx_) >> 30) {
                const float ly = Tf[py], lx = Tf[32 + px], hy = 1.f - ly, hx = 1.f - lx;
                const int o00 = (ty_ & 0xfffffff) + (tx_ & 0xfffffff);
                const int dyo = ((ty_ >> 28) & 1) * SRW * S_ROWB, dxo = ((tx_ >> 28) & 1) * S_ROWB;
                const uint4 a00 = *reinterpret_cast<const uint4*>(sb + o00);
                const uint4 a01 = *reinterpret_cast<const uint4*>(sb + o00 + dxo);
                const uint4 a10 = *reinterpret_cast<const uint4*>(sb + o00 + dyo);
                const uint4 a11 = *reinterpret_cast<const uint4*>(sb + o00 + dyo + dxo);
                const uint32_t u00[4] = {a00.x, a00.y, a00.z, a00.w}, u01[4] = {a01.x, a01.y, a01.z, a01.w};
                const uint32_t u10[4] = {a10.x, a10.y, a10.z, a10.w}, u11[4] = {a11.x, a11.y, a11.z, a11.w};
                uint32_t ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = hy * (hx * h_lo<F16>(u00[j]) + lx * h_lo<F16>(u01[j])) + ly * (hx * h_lo<F16>(u10[j]) + lx * h_lo<F16>(u11[j]));
                    const float hi = hy * (hx * h_hi<F16>(u00[j]) + lx * h_hi<F16>(u01[j])) + ly * (hx * h_hi<F16>(u10[j]) + lx * h_hi<F16>(u11[j]));
                    ov[j] = pack_h2<F16>(lo, hi);
                }
                o = make_uint4(ov[0], ov[1], ov[2], ov[3]);
            }
            *reinterpret_cast<uint4*>(Ps + p * P_ROWB + ch * 16) = o;
        }
    };
    // 9 taps x 4 k-steps: wave = 2 output rows x 16 channels, weights from registers, patch rows from LDS
    f32x4 acc[2];
    auto multiply = [&](const char* Ps) {
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // by PATCH row m = 0..3 of the wave's two output rows: row m feeds output row 0 as tap row ky = m and output row 1 as ky = m - 1, so its fragments are
        // read once for both (48 ds_read_b128 per tile and wave instead of 72; each accumulator still sees its taps in (ky, kx, ks) order: same bits)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int prow = (2 * rowpair + m) * PW + frow + kx;
                    const h16x8 xf = *reinterpret_cast<const h16x8*>(Ps + prow * P_ROWB + (ks * 32 + fq * 8) * 2);
                    if (m <= 2) acc[0] = mfma_16x16x32<F16>(wf[(m * 3 + kx) * 4 + ks], xf, acc[0]);
                    if (m >= 1) acc[1] = mfma_16x16x32<F16>(wf[((m - 1) * 3 + kx) * 4 + ks], xf, acc[1]);
                }
            }
        }
    };
    const int tile0 = blockIdx.x, step = gridDim.x;
    if (tile0 >= ntiles) return;
    // ---- prologue: window and patch of the first tile; raw barriers throughout (__syncthreads() would also drain the LDS-DMA of a later window, which is
    // the latency being hidden) ----
    issue_window(tile0, 0);
    tables(tile0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile0 + step < ntiles) issue_window(tile0 + step, 1);
    build(Ss0, Ps0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // build(tile0) has read the tables
    int iter = 0;
    for (int tile = tile0; tile < ntiles; tile += step, ++iter) {
        const int tx = tile % tiles_x;
        int r = tile / tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int cur = iter & 1;
        const bool next = tile + step < ntiles;
        if (next) tables(tile + step);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of the NEXT tile's window have landed, its table entries are written, its share of THIS tile's patch too
        __builtin_amdgcn_s_barrier();                                  // ... everybody's; and the previous tile's MFMAs have left the other patch buffer, its epilogue the `red` words
        // window buffer `cur` was last read by the build of THIS tile's patch, one barrier interval ago: the tile after next goes there
        if (tile + 2 * step < ntiles) issue_window(tile + 2 * step, cur);
        const char* Pc = Ps0 + cur * P_BYTES;
        if (wave < 4) {
            if (next) build(Ss0 + (cur ^ 1) * S_BYTES, Ps0 + (cur ^ 1) * P_BYTES);
            multiply(Pc);
        } else {
            multiply(Pc);
            if (next) build(Ss0 + (cur ^ 1) * S_BYTES, Ps0 + (cur ^ 1) * P_BYTES);
        }
        // ---- epilogue: + bias, ReLU, 1x1 (32 -> 1) + bias, ReLU; the two channel tiles meet through LDS ----
        float part[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sdot = fmaxf(acc[j][0] + bia.x, 0.f) * w4v.x + fmaxf(acc[j][1] + bia.y, 0.f) * w4v.y + fmaxf(acc[j][2] + bia.z, 0.f) * w4v.z +
                         fmaxf(acc[j][3] + bia.w, 0.f) * w4v.w;
            sdot += __shfl_xor(sdot, 16);
            sdot += __shfl_xor(sdot, 32);
            part[j] = sdot;
        }
        if (ntile == 1 && fq == 0) {
            red[(2 * rowpair + 0) * TW + frow] = part[0];
            red[(2 * rowpair + 1) * TW + frow] = part[1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ntile == 0 && fq == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int Y = ty * TH + 2 * rowpair + j, X = tx * TW + frow;
                out[((size_t)b * H + Y) * W + X] = fmaxf(part[j] + red[(2 * rowpair + j) * TW + frow] + b4, 0.f);
            }
        }
    }
}

int launch_depth_tail(const bf16_t* d1, const bf16_t* wt, const float* bias, const float* w4, float b4, float* out, int hf, int B, int h,
                      int w, hipStream_t st, std::string& err) {
    if ((2 * h) % TH || (2 * w) % TW) { err = "depth_tail: output size must be a multiple of 8 x 16"; return 1; }
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&depth_tail_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&depth_tail_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { err = std::string("depth_tail: ") + hipGetErrorString(e); return 1; }
        attr_done.done();
    }
    const int ntiles = B * (2 * h / TH) * (2 * w / TW);
    const int blocks = ntiles < 256 ? ntiles : 256;  // persistent: one workgroup per CU keeps the weights resident
    if (hf) SOCCDPT_LAUNCH(depth_tail_kernel<true>, dim3(blocks), dim3(NTHR), LDS_BYTES, st, d1, wt, bias, w4, b4, out, B, h, w);
    else SOCCDPT_LAUNCH(depth_tail_kernel<false>, dim3(blocks), dim3(NTHR), LDS_BYTES, st, d1, wt, bias, w4, b4, out, B, h, w);
    return check_launch("depth_tail", err);
}

}  // namespace soccdpt
