#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ __attribute__((aligned(16))) short s[16 * 16];
    for (int i = threadIdx.x; i < 16 * 16; i += 64) s[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + (4 * g + q) * 16 + 4 * p));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short h[256], o[256];
    for (int i = 0; i < 256; ++i) h[i] = (short)i;   // value = row * 16 + col
    short *d, *r;
    hipMalloc(&d, 512); hipMalloc(&r, 512);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r);
    hipMemcpy(o, r, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int e = 0; e < 4; ++e) { const int want = (4 * g + e) * 16 + i; if (o[l * 4 + e] != want) ++bad; }
    }
    printf("lane 0: %d %d %d %d  lane 5: %d %d %d %d  lane 17: %d %d %d %d  mismatches vs 'lane i gets column i of rows 4g..4g+3': %d\n", o[0], o[1], o[2], o[3], o[20], o[21], o[22], o[23],
           o[68], o[69], o[70], o[71], bad);
    return 0;
}
