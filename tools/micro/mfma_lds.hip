// What does feeding MFMA operands from LDS cost?  One wave per SIMD (and two), per k-step 4 x v_mfma_f32_32x32x2
// (256 cycles) + the LDS reads that deliver their A operands 3 k-steps ahead, in different widths.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define M32(acc, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
#define SB() __builtin_amdgcn_sched_barrier(0)
// MODE 0: no LDS   1: 4 x ds_read_b32 per k-step   2: 2 x ds_read_b64   3: 1 x ds_read_b128   4: 1 x ds_read_b32 (quarter of the data)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float ring[3][4];
    for (int r = 0; r < 3; ++r) for (int i = 0; i < 4; ++i) ring[r][i] = 1.0f + i;
    const float b = b0 + threadIdx.x * 2e-3f;
    const float *lp = lds + (threadIdx.x & 63) * 4;  // 16-byte aligned per lane
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ring[u][i];
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ring[u][i] = lp[i + 256 * u + 1024];
            } else if (MODE == 2) {
                f32x2 v0 = *reinterpret_cast<const f32x2 *>(lp + 256 * u + 1024), v1 = *reinterpret_cast<const f32x2 *>(lp + 256 * u + 1026);
                ring[u][0] = v0[0]; ring[u][1] = v0[1]; ring[u][2] = v1[0]; ring[u][3] = v1[1];
            } else if (MODE == 3) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(lp + 256 * u + 1024);
                ring[u][0] = v[0]; ring[u][1] = v[1]; ring[u][2] = v[2]; ring[u][3] = v[3];
            } else if (MODE == 4) {
                ring[u][0] = lp[256 * u + 1024];
            }
            SB();
#pragma unroll
            for (int i = 0; i < 4; ++i) M32(acc[i], a[i], b);
            SB();
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * 4096);
    (void)hipMalloc(&cyc, 8 * 4096);
    const int iters = 3000;
#define R(name, MODE)                                                                                              \
    for (int bpc = 1; bpc <= 2; ++bpc) {                                                                           \
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(256 * bpc), dim3(256), 0, 0, out, cyc, iters, 0.5f); \
        (void)hipDeviceSynchronize();                                                                              \
        unsigned long long h[512]; (void)hipMemcpy(h, cyc, 8 * 256 * bpc, hipMemcpyDeviceToHost);                 \
        double s = 0; for (int i = 0; i < 256 * bpc; ++i) s += (double)h[i];                                       \
        printf("%-40s waves/SIMD %d : %7.1f cycles per k-step per wave (4 MFMA = 256)\n", name, bpc, s / (256 * bpc) / iters); \
    }
    R("no LDS", 0) R("4 x ds_read_b32", 1) R("2 x ds_read_b64", 2) R("1 x ds_read_b128", 3) R("1 x ds_read_b32", 4)
    return 0;
}
