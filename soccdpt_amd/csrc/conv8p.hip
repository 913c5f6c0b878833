// Phase-interleaved 3x3 implicit-GEMM convolution for the large decoder / head convolutions (gfx950).
//
// Same arithmetic and epilogue contract as igemm_kernel (igemm.hip) for taps == 9, Cin % 64 == 0: out[m][n] = epilogue(sum_k X[m][k] Wt[n][k])
// with the 3x3 taps gathered from a zero-haloed NHWC image by the LDS-DMA source address.  What differs is the main loop, built after
// the guide's 8-phase GEMM schedule (cdna_hip_programming.md section 5, "The 256^2 8-phase template") and MI355X_MICROARCH.md
// "Two waves per SIMD" item 9 (stagger waves 4-7):
//   * tile BM x BN = 256 x 256 (or 128 x 256 / 256 x 128), 64-deep k-tiles, 8 waves as 2 (pixel rows) x 4 (channel columns): a wave owns
//     (BM/2) x (BN/4) outputs = 32 (16) accumulator tiles, twice (1.5x) the MFMAs per LDS fragment byte of the 128 x 128 tile;
//   * a k-tile is FOUR phases, one output quadrant each: P1 reads W-sub0 + X-sub0 fragments, P2 W-sub1, P3 X-sub1, P4 nothing (W-sub0 is
//     still in registers); every phase is a READ interval (ds_read fragments, issue one LDS-DMA staging unit) and a MATRIX interval
//     (MFMAs under s_setprio 1), each closed by a raw s_barrier;
//   * waves 4-7 (the second pixel-row half; they share SIMDs with waves 0-3) run ONE barrier behind: while one wave of a SIMD issues its
//     MFMAs the other one reads / stages, so the matrix pipe of every SIMD is fed in every interval;
//   * LDS holds two k-tile buffers; the staging UNITS are cut by time of use, not by wave: U1 = W-sub0 rows, U2 = X-sub0 rows (both last read
//     in P1), U3 = W-sub1 rows (P2), U4 = X-sub1 rows (P3).  A unit of k-tile t+2 is issued into the buffer of k-tile t one phase after that
//     region's last read -- U1 at P2(t), U2 at P3(t), U3 at P4(t), U4 at P1(t+1) -- i.e. six to seven phases (1.5 k-tiles) before its
//     first use, with ONE counted s_waitcnt vmcnt per phase ("all but the five youngest units"), never vmcnt(0) in the steady state.
//     Write-after-read: the reading phase ends with lgkmcnt(0) + barrier before any wave issues into the region.  Read-after-write: the
//     counted wait of EVERY wave precedes a barrier that precedes the first read (group 1 waits in its READ interval, group 0 in its MATRIX
//     interval: the same barrier tick).
#include <type_traits>

#include "gelu.h"
#include "half16.h"
#include "igemm.h"

namespace soccdpt {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BM_, int BN_>
struct C8 {
    static constexpr int BM = BM_, BN = BN_, BK = 64, THREADS = 512;
    static constexpr int ROWB = 128;                       // bytes per LDS tile row (64 x 16 bit)
    static constexpr int TM = BM / 32, TN = BN / 64;       // 16 x 16 tiles per wave: (BM/2)/16 x (BN/4)/16
    static constexpr int HM = TM / 2, HN = TN / 2;         // tiles per sub (quadrant side)
    static constexpr int X_BYTES = BM * ROWB, W_BYTES = BN * ROWB, BUF = X_BYTES + W_BYTES;
    static constexpr int LX = (BM / 2) * 8 / THREADS;      // LDS-DMA instructions per thread per X unit (BM/2 rows x 8 chunks)
    static constexpr int LW = (BN / 2) * 8 / THREADS;
    static constexpr int VM_STEADY = 2 * LW + 2 * LX + (LW < LX ? LW : LX);   // loads of the five youngest units (worst alignment)
    static constexpr int VM_START = 3 * (LW + LX);                             // before phase 0: U3(0), U4(0), U1..U4(1) may be in flight
    static constexpr int LDS = 2 * BUF;
    static_assert(LX >= 1 && LW >= 1 && TM >= 2 && TN >= 2, "tile too small for the quadrant schedule");
};

// VAR: ablation switches for tools/conv8p_bench.py (0 in production): 1 no s_setprio, 2 fragment wait after the barrier, 4 no staging after
// the prologue (WRONG results, timing only), 8 no fragment reads after k-tile 0 (WRONG), 16 no MFMAs (WRONG)
template <class C, bool F16, int VAR = 0>
__global__ __launch_bounds__(512) void conv8p_kernel(IgemmDesc d, int nk, int kpt, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN, HM = C::HM, HN = C::HN;
    const uint16_t* const Xp = static_cast<const uint16_t*>(d.X);
    const uint16_t* const Wtp = static_cast<const uint16_t*>(d.Wt);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // wm == 1: waves 4-7, the staggered group

    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (as igemm_kernel)
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int Ktot = 9 * d.Cin;
    const int Wp = d.W + 2;
    const int Wpi = (d.Wi ? d.Wi : d.W) + 2 * d.in_halo, Hpi = (d.Hi ? d.Hi : d.H) + 2 * d.in_halo;

    // ---- staging sources.  Unit rows -> tile rows: W-sub s of wave column c: rows c*(BN/4) + s*(BN/8) + 0..BN/8-1;
    //      X-sub s of wave row r: rows r*(BM/2) + s*(BM/4) + 0..BM/4-1.  A thread moves chunk (i*512 + tid) of a unit: unit row = chunk >> 3.
    uint32_t xs_off[2][C::LX], ws_off[2][C::LW];
    uint32_t xs_lds[2][C::LX], ws_lds[2][C::LW];   // wave-uniform LDS byte offsets inside a k-tile buffer
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < C::LX; ++i) {
            const int cid = i * 512 + tid;
            const int ur = cid >> 3, c = cid & 7;                         // unit row 0..BM/2-1
            const int row = (ur / (BM / 4)) * (BM / 2) + s * (BM / 4) + (ur % (BM / 4));
            int m = m0 + row;
            m = m < d.M ? m : d.M - 1;
            const int hw = d.H * d.W;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / d.W, x = rem - y * d.W;
            const uint32_t base = (uint32_t)(((b * Hpi + y * d.stride + d.in_halo - d.pad) * Wpi + x * d.stride + d.in_halo - d.pad) * d.Cin);
            xs_off[s][i] = base + (uint32_t)((c ^ (row & 7)) * 8);
            const int ur0 = (i * 512 + wave * 64) >> 3;                  // first unit row of this wave's 1 KB piece
            const int row0 = (ur0 / (BM / 4)) * (BM / 2) + s * (BM / 4) + (ur0 % (BM / 4));
            xs_lds[s][i] = (uint32_t)(row0 * C::ROWB);
        }
#pragma unroll
        for (int i = 0; i < C::LW; ++i) {
            const int cid = i * 512 + tid;
            const int ur = cid >> 3, c = cid & 7;                         // unit row 0..BN/2-1
            const int row = (ur / (BN / 8)) * (BN / 4) + s * (BN / 8) + (ur % (BN / 8));
            int n = n0 + row;
            n = n < d.N ? n : d.N - 1;
            ws_off[s][i] = (uint32_t)n * (uint32_t)Ktot + (uint32_t)((c ^ (row & 7)) * 8);
            const int ur0 = (i * 512 + wave * 64) >> 3;
            const int row0 = (ur0 / (BN / 8)) * (BN / 4) + s * (BN / 8) + (ur0 % (BN / 8));
            ws_lds[s][i] = (uint32_t)(C::X_BYTES + row0 * C::ROWB);
        }
    }
    // unit u (0: W-sub0, 1: X-sub0, 2: W-sub1, 3: X-sub1) of k-tile kt into buffer kt & 1
    auto stage_unit = [&](int kt, int u) __attribute__((always_inline)) {
        const int tap = kt / kpt, kc = kt - tap * kpt;
        const int ky = tap / 3, kx = tap - ky * 3;
        char* sb = smem + (kt & 1) * C::BUF;
        if (u & 1) {
            const uint32_t xk = (uint32_t)((ky * Wpi + kx) * d.Cin + kc * 64);
            const int s = u >> 1;
#pragma unroll
            for (int i = 0; i < C::LX; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xp + xs_off[s][i] + xk),
                                                 (__attribute__((address_space(3))) void*)(sb + xs_lds[s][i]), 16, 0, 0);
        } else {
            const uint32_t wk = (uint32_t)kt * 64;
            const int s = u >> 1;
#pragma unroll
            for (int i = 0; i < C::LW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wtp + ws_off[s][i] + wk),
                                                 (__attribute__((address_space(3))) void*)(sb + ws_lds[s][i]), 16, 0, 0);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias of this lane's channels, requested before the first LDS-DMA group (older than every counted wait)
    float4 bias_pre[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        int n = n0 + wn * TN * 16 + i * 16 + (lane >> 4) * 4;
        n = n < d.N ? n : 0;
        bias_pre[i] = d.bias ? *reinterpret_cast<const float4*>(d.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    asm volatile("" ::: "memory");

    // fragment read offsets inside a k-tile buffer (row & 7 == lane & 7 for every fragment row: tile bases are multiples of 16)
    const int frow = lane & 15, fq = lane >> 4;
    int xr_off[2], wr_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int q = ((ks * 4 + fq) ^ (frow & 7)) * 16;
        xr_off[ks] = (wm * (BM / 2) + frow) * C::ROWB + q;
        wr_off[ks] = C::X_BYTES + (wn * (BN / 4) + frow) * C::ROWB + q;
    }

    // ---- prologue: k-tiles 0 and 1 completely (8 units), then the group stagger ----
#pragma unroll
    for (int u = 0; u < 4; ++u) stage_unit(0, u);
    if (nk > 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) stage_unit(1, u);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_START) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::LW + C::LX) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier interval behind

    h16x8 wf[2][HN][2], xf[HM][2];   // W fragments of both subs stay resident over the k-tile; X fragments of the current sub
    const int nphase = 4 * nk;
    // staging stops once k-tile nk-1 has been issued: the unit issued in phase q belongs to k-tile (q + 7) / 4, unit (q + 3) % 4  (q >= 1)
    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + (kt & 1) * C::BUF;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = kt * 4 + p;
            // ================= READ interval =================
            if ((VAR & 8) && kt > 0) {
            } else if (p == 0) {
#pragma unroll
                for (int i = 0; i < HN; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) wf[0][i][ks] = *reinterpret_cast<const h16x8*>(sb + wr_off[ks] + i * 16 * C::ROWB);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HM; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) xf[j][ks] = *reinterpret_cast<const h16x8*>(sb + xr_off[ks] + j * 16 * C::ROWB);
            } else if (p == 1) {
#pragma unroll
                for (int i = 0; i < HN; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) wf[1][i][ks] = *reinterpret_cast<const h16x8*>(sb + wr_off[ks] + (HN + i) * 16 * C::ROWB);
            } else if (p == 2) {
#pragma unroll
                for (int j = 0; j < HM; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) xf[j][ks] = *reinterpret_cast<const h16x8*>(sb + xr_off[ks] + (HM + j) * 16 * C::ROWB);
            }
            // one staging unit per phase from phase 1 on: unit (q + 3) % 4 of k-tile (q + 7) / 4 into the region read last in phase q - 1
            const int st_kt = (q + 7) >> 2;
            const bool staging = q >= 1 && st_kt < nk;
            if (staging && !(VAR & 4)) stage_unit(st_kt, (q + 3) & 3);
            if (wm == 1) {   // group 1: the counted wait of the phase sits in the READ interval (same barrier tick as group 0's MATRIX interval)
                if (q + 7 < nphase) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_STEADY) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!(VAR & 2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (VAR & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
            // ================= MATRIX interval: quadrant (X sub p >> 1, W sub (p == 1 || p == 2)) =================
            if (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
            if (!(VAR & 16)) {
                const int ws = (p == 1 || p == 2) ? 1 : 0, xs = p >> 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < HN; ++i)
#pragma unroll
                        for (int j = 0; j < HM; ++j)
                            acc[ws * HN + i][xs * HM + j] = mfma_16x16x32<F16>(wf[ws][i][ks], xf[j][ks], acc[ws * HN + i][xs * HM + j]);
            }
            else {
#pragma unroll
                for (int i = 0; i < HN; ++i) { asm volatile("" ::"v"(wf[0][i][0]), "v"(wf[1][i][1])); }
#pragma unroll
                for (int j = 0; j < HM; ++j) asm volatile("" ::"v"(xf[j][0]), "v"(xf[j][1]));
            }
            if (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
            if (wm == 0) {
                if (q + 7 < nphase) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_STEADY) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the last barrier of the staggered group

    // ---- epilogue (igemm_kernel's generic one: bias, residual(s), ReLU, f32 / operand / halo stores) ----
    const int N = d.N;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * (BM / 2) + j * 16 + (lane & 15);
        const bool mv = m < d.M;
        const size_t orow = (size_t)m * N;
        size_t hrow = 0;
        size_t up00 = 0, up01 = 0, up10 = 0, up11 = 0;
        float uly = 0.f, ulx = 0.f;
        if ((d.out_halo || d.res2_h) && mv) {
            const int hw = d.H * d.W;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / d.W, x = rem - y * d.W;
            hrow = ((size_t)(b * (d.H + 2) + y + 1) * Wp + x + 1) * N;
            if (d.res2_h) {
                const float sy = d.H > 1 ? (float)(d.res2_h - 1) / (float)(d.H - 1) : 0.f;
                const float sx = d.W > 1 ? (float)(d.res2_w - 1) / (float)(d.W - 1) : 0.f;
                const float fy = sy * (float)y, fx = sx * (float)x;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < d.res2_h - 1), x1 = x0 + (x0 < d.res2_w - 1);
                uly = fy - (float)y0;
                ulx = fx - (float)x0;
                const size_t pb = (size_t)b * d.res2_h * d.res2_w;
                up00 = (pb + (size_t)y0 * d.res2_w + x0) * N;
                up01 = (pb + (size_t)y0 * d.res2_w + x1) * N;
                up10 = (pb + (size_t)y1 * d.res2_w + x0) * N;
                up11 = (pb + (size_t)y1 * d.res2_w + x1) * N;
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / 4) + i * 16 + (lane >> 4) * 4;
            if (!mv || n >= N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const float4 b4 = bias_pre[i];
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            if (d.res1) {
                const float4 r4 = *reinterpret_cast<const float4*>(d.res1 + orow + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (d.res2 && d.res2_h) {
                const float4 a00 = *reinterpret_cast<const float4*>(d.res2 + up00 + n), a01 = *reinterpret_cast<const float4*>(d.res2 + up01 + n);
                const float4 a10 = *reinterpret_cast<const float4*>(d.res2 + up10 + n), a11 = *reinterpret_cast<const float4*>(d.res2 + up11 + n);
                const float hy = 1.f - uly, hx = 1.f - ulx;
                v[0] += hy * (hx * a00.x + ulx * a01.x) + uly * (hx * a10.x + ulx * a11.x);
                v[1] += hy * (hx * a00.y + ulx * a01.y) + uly * (hx * a10.y + ulx * a11.y);
                v[2] += hy * (hx * a00.z + ulx * a01.z) + uly * (hx * a10.z + ulx * a11.z);
                v[3] += hy * (hx * a00.w + ulx * a01.w) + uly * (hx * a10.w + ulx * a11.w);
            } else if (d.res2) {
                const float4 r4 = *reinterpret_cast<const float4*>(d.res2 + orow + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = d.act == ACT_RELU ? fmaxf(v[r], 0.f) : (d.act == ACT_GELU ? gelu_fast(v[r]) : v[r]);
            if (d.out_f32) {
                const float* s = d.act_on_f32 ? a : v;
                *reinterpret_cast<float4*>(d.out_f32 + orow + n) = make_float4(s[0], s[1], s[2], s[3]);
            }
            if (d.out_op) {
                uint2 p;
                p.x = pack_h2<F16>(a[0], a[1]);
                p.y = pack_h2<F16>(a[2], a[3]);
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.out_op) + (d.out_halo ? hrow : orow) + n) = p;
            }
        }
    }
}

template <int VAR>
int launch_c8_var(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    using C = C8<256, 256>;
    const int nk = 9 * d.Cin / 64, kpt = d.Cin / 64;
    const int mtiles = (d.M + C::BM - 1) / C::BM, ntiles = (d.N + C::BN - 1) / C::BN;
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8p_kernel<C, false, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        attr_done.done();
    }
    SOCCDPT_LAUNCH((conv8p_kernel<C, false, VAR>), dim3((unsigned)(mtiles * ntiles)), dim3(512), C::LDS, stream, d, nk, kpt, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("conv8p launch: ") + hipGetErrorString(e); return 1; }
    return 0;
}

template <class C>
int launch_c8(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    const int nk = 9 * d.Cin / 64, kpt = d.Cin / 64;
    const int mtiles = (d.M + C::BM - 1) / C::BM, ntiles = (d.N + C::BN - 1) / C::BN;
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8p_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8p_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) { err = std::string("conv8p: hipFuncSetAttribute: ") + hipGetErrorString(e); return 1; }
        attr_done.done();
    }
    if (d.f16) SOCCDPT_LAUNCH((conv8p_kernel<C, true>), dim3((unsigned)(mtiles * ntiles)), dim3(512), C::LDS, stream, d, nk, kpt, ntiles);
    else SOCCDPT_LAUNCH((conv8p_kernel<C, false>), dim3((unsigned)(mtiles * ntiles)), dim3(512), C::LDS, stream, d, nk, kpt, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("conv8p launch: ") + hipGetErrorString(e); return 1; }
    return 0;
}

}  // namespace

// variant: 0 = 256 x 256, 1 = 128 x 256, 2 = 256 x 128 (pixels x channels)
bool conv8p_supported(const IgemmDesc& d, int variant) {
    if (d.f32 || d.taps != 9 || d.Cin % 64 != 0 || d.ln_g || d.splitk > 1 || d.gn_stats || d.out_dot || d.seg2_k || d.grp_rows) return false;
    if (9 * d.Cin / 64 < 2) return false;
    const int bn = variant == 2 ? 128 : 256;
    if (variant < 0 || variant > 9) return false;
    return d.N % bn == 0;
}

int launch_conv8p(const IgemmDesc& d, int variant, hipStream_t stream, std::string& err) {
    if (!conv8p_supported(d, variant)) { err = "conv8p: unsupported descriptor for this variant"; return 1; }
    switch (variant) {
        case 3: return launch_c8_var<1>(d, stream, err);    // ablations (tools/conv8p_bench.py): configuration ids 33..
        case 4: return launch_c8_var<2>(d, stream, err);
        case 5: return launch_c8_var<4>(d, stream, err);
        case 6: return launch_c8_var<8>(d, stream, err);
        case 7: return launch_c8_var<16>(d, stream, err);
        case 8: return launch_c8_var<3>(d, stream, err);
        case 9: return launch_c8_var<12>(d, stream, err);
        case 0: return launch_c8<C8<256, 256>>(d, stream, err);
        case 1: return launch_c8<C8<128, 256>>(d, stream, err);
        default: return launch_c8<C8<256, 128>>(d, stream, err);
    }
}

}  // namespace soccdpt
