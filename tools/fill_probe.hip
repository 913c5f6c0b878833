// Stand-alone probe of what bounds a big implicit-GEMM tile on MI355X: the per-CU L2 -> LDS fill rate by LDS-DMA
// (global_load_lds_dwordx4), by register staging (global_load_dwordx4 + ds_write_b128) and by both at once, alone and beside the LDS fragment
// reads and MFMAs of a GEMM k-tile, with the in-kernel clock (s_memtime / s_memrealtime) stamped beside every figure.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fill_probe.hip -o tools/fill_probe.out && tools/fill_probe.out
//
// One workgroup = 512 threads (8 waves) stages one TILE (32 KB = the operand bytes of a 128 x 128 x 64 16-bit k-tile; 16 KB where noted) per
// iteration into a two-slot LDS ring, one tile in flight across the barrier (counted vmcnt), then every wave reads R 1-KB fragments
// (ds_read_b128) and issues M v_mfma_f32_16x16x32_f16 on them.  Source: SHARED = every workgroup walks the same 2 MB (L2-resident: weights,
// or activation rows re-read by the taps); STREAM = every workgroup walks its own slice of a 1 GB buffer (HBM / Infinity Cache).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: LDS-DMA for every 16-byte piece; 1: register staging for every piece; 2: pieces alternate (half DMA, half registers); 3: no staging
// PIECES: 16-byte pieces per thread per tile (4 = 32 KB per workgroup, 2 = 16 KB)
template <int MODE, int PIECES, int R, int M>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, size_t tiles_in_src, size_t tile_stride, int per_wg_stride_tiles, int iters,
                                             unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 512 * 16 * 4;   // LDS slot: 32 KB whatever PIECES is (the fragment reads walk all of it)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t wg0 = (size_t)blockIdx.x * per_wg_stride_tiles;
    constexpr int NDMA = MODE == 0 ? PIECES : (MODE == 2 ? PIECES / 2 : 0);
    constexpr int NREG = MODE == 1 ? PIECES : (MODE == 2 ? PIECES - PIECES / 2 : 0);
    uint4 regs[NREG > 0 ? NREG : 1];
    auto tile_src = [&](int it) { return src + ((wg0 + (size_t)it) % tiles_in_src) * tile_stride; };
    auto issue = [&](int it, int slot) {
        const char* s = tile_src(it);
        char* d = smem + slot * TILE;
#pragma unroll
        for (int j = 0; j < NDMA; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (size_t)(j * 512 + tid) * 16),
                                             (__attribute__((address_space(3))) void*)(d + (j * 512 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NREG; ++j) regs[j] = *reinterpret_cast<const uint4*>(s + (size_t)((NDMA + j) * 512 + tid) * 16);
    };
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long t0 = 0, r0 = 0;
    if (MODE != 3) issue(0, 0);
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        const int slot = it & 1;
        if (MODE != 3) {
            if (NREG > 0) {   // this tile's registers -> LDS, then the next tile's loads go out
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < NREG; ++j) *reinterpret_cast<uint4*>(smem + slot * TILE + ((NDMA + j) * 512 + tid) * 16) = regs[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (it + 1 < iters) issue(it + 1, slot ^ 1);
        }
        const char* sb = smem + slot * TILE;
        h16x8 f[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = *reinterpret_cast<const h16x8*>(sb + ((r * 37 + wave * 5) % 32) * 1024 + lane * 16);
        if (R == 0) f[0] = h16x8{(_Float16)1.f, (_Float16)2.f, (_Float16)0.5f, (_Float16)3.f, (_Float16)1.5f, (_Float16)0.25f, (_Float16)2.5f, (_Float16)0.75f};
#pragma unroll
        for (int m = 0; m < M; ++m)
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[m % (R > 0 ? R : 1)], f[(m * 3 + 1) % (R > 0 ? R : 1)], acc[m & 3], 0, 0, 0);
        if (R > 0 && M == 0) { asm volatile("" ::"v"(f[0]), "v"(f[R - 1])); }
        if (MODE == 3 && (R > 0)) __builtin_amdgcn_s_barrier();
    }
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678f) sink[tid] = s;
}

struct Buf { char* shared; char* stream; unsigned long long* stamps; float* sink; };

template <int MODE, int PIECES, int R, int M>
static void run(const char* name, const Buf& b, bool stream_src, int wg_per_cu, int iters) {
    const int grid = 256 * wg_per_cu;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : 64 * 1024;   // 1 per CU: ask for more than half the LDS
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, PIECES, R, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t tile_stride = 32768;
    const size_t tiles = stream_src ? (size_t)(1ull << 30) / tile_stride : (size_t)(2u << 20) / tile_stride;
    const int per_wg = stream_src ? (int)(tiles / grid) : 7;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<unsigned long long> st(2 * grid);
    double clk = 0, cyc = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        probe<MODE, PIECES, R, M><<<grid, 512, lds>>>(stream_src ? b.stream : b.shared, tiles, tile_stride, per_wg, iters, b.stamps, b.sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(st.data(), b.stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
            std::vector<double> c, k;
            for (int i = 0; i < grid; ++i) { c.push_back((double)st[2 * i]); k.push_back((double)st[2 * i] / ((double)st[2 * i + 1] * 10.0)); }   // cycles per ns = GHz (realtime ticks at 100 MHz)
            std::sort(c.begin(), c.end()); std::sort(k.begin(), k.end());
            cyc = c[grid / 2]; clk = k[grid / 2];
        }
    }
    const double bytes_per_wg = MODE == 3 ? 0.0 : (double)iters * PIECES * 512 * 16;
    const double cyc_per_tile = cyc / iters;
    const double fill_b_per_clk_cu = bytes_per_wg * wg_per_cu / cyc;
    const double mfma_busy = (double)M * 8 /*waves*/ * wg_per_cu * 16.0 / 4.0 / cyc_per_tile;   // 16 cycles per MFMA, 4 SIMDs
    const double tf = (double)grid * iters * 8 * M * 16384.0 / (best * 1e-3) / 1e12;
    printf("%-58s %s %d WG/CU: %8.1f us  clock %.2f GHz  %7.0f cyc/tile  fill %5.1f B/clk/CU = %5.1f GB/s/CU  LDS reads %5.1f B/clk/CU  MFMA busy %.3f  (%.0f TFLOP/s)\n",
           name, stream_src ? "STREAM" : "SHARED", wg_per_cu, best * 1e3, clk, cyc_per_tile, fill_b_per_clk_cu, fill_b_per_clk_cu * clk,
           (double)R * 1024 * 8 * wg_per_cu / cyc_per_tile, mfma_busy, tf);
    fflush(stdout);
}

int main() {
    Buf b;
    CK(hipMalloc(&b.shared, 2u << 20));
    CK(hipMalloc(&b.stream, 1ull << 30));
    CK(hipMalloc(&b.stamps, sizeof(unsigned long long) * 2 * 1024));
    CK(hipMalloc(&b.sink, 4096));
    {   // random fp16-looking bytes (finite, |v| < 4): operand bits set the clock the chip holds
        std::vector<uint16_t> h((2u << 20) / 2);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(((s >> 16) & 0x83ff) | 0x3c00 - ((s >> 9) & 0x0c00)); }
        CK(hipMemcpy(b.shared, h.data(), 2u << 20, hipMemcpyHostToDevice));
        for (size_t o = 0; o < (1ull << 30); o += 2u << 20) CK(hipMemcpy(b.stream + o, b.shared, 2u << 20, hipMemcpyDeviceToDevice));
    }
    const int IT = 400;
    printf("# fill path alone (no fragment reads, no MFMA)\n");
    run<0, 4, 0, 0>("LDS-DMA 32 KB/tile", b, false, 1, IT);
    run<0, 4, 0, 0>("LDS-DMA 32 KB/tile", b, false, 2, IT);
    run<1, 4, 0, 0>("register staging 32 KB/tile", b, false, 1, IT);
    run<1, 4, 0, 0>("register staging 32 KB/tile", b, false, 2, IT);
    run<2, 4, 0, 0>("half DMA + half registers 32 KB/tile", b, false, 1, IT);
    run<2, 4, 0, 0>("half DMA + half registers 32 KB/tile", b, false, 2, IT);
    run<0, 4, 0, 0>("LDS-DMA 32 KB/tile", b, true, 2, IT);
    run<1, 4, 0, 0>("register staging 32 KB/tile", b, true, 2, IT);
    printf("# matrix side alone (no staging): 12 fragment reads + 16 MFMAs per wave and tile = the 64 x 32 wave tile of the 8-wave 128 x 128 x 64 kernel\n");
    run<3, 4, 12, 16>("12 ds_read_b128 + 16 MFMA, barrier per tile", b, false, 2, IT);
    run<3, 4, 0, 16>("16 MFMA only (operands in registers)", b, false, 2, IT);
    run<3, 4, 12, 0>("12 ds_read_b128 only", b, false, 2, IT);
    run<3, 4, 16, 32>("16 reads + 32 MFMA (64 x 64 wave tile)", b, false, 1, IT);
    run<3, 4, 16, 32>("16 reads + 32 MFMA (64 x 64 wave tile)", b, false, 2, IT);
    printf("# the whole k-tile: staging + fragment reads + MFMAs\n");
    run<0, 4, 12, 16>("128x128x64 shape: DMA 32 KB + 12 reads + 16 MFMA", b, false, 2, IT);
    run<0, 4, 12, 16>("128x128x64 shape: DMA 32 KB + 12 reads + 16 MFMA", b, true, 2, IT);
    run<2, 4, 12, 16>("128x128x64 shape: half DMA half regs + 12 reads + 16 MFMA", b, false, 2, IT);
    run<1, 4, 12, 16>("128x128x64 shape: register staging + 12 reads + 16 MFMA", b, false, 2, IT);
    run<0, 2, 12, 16>("same, 16 KB staged per tile (halo reuse of the activation tile)", b, false, 2, IT);
    run<0, 4, 16, 32>("256x128x64 shape: DMA 32 KB(of 48) + 16 reads + 32 MFMA", b, false, 1, IT);
    run<0, 4, 16, 32>("256x128x64 shape, 2 WG/CU (32-deep would fit): DMA 32 KB + 16 reads + 32 MFMA", b, false, 2, IT);
    run<0, 2, 16, 32>("256x128x64 shape with halo reuse: DMA 16 KB(of 23) + 16 reads + 32 MFMA", b, false, 1, IT);
    run<0, 2, 16, 32>("256x128x64 shape with halo reuse: DMA 16 KB + 16 reads + 32 MFMA", b, false, 2, IT);
    return 0;
}
