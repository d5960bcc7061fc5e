// What does feeding MFMA operands from LDS cost?  Per k-step 4 x v_mfma_f32_32x32x2 (256 cycles) and the LDS reads that
// deliver the 4 A operands four k-steps ahead (ring of 4, fully unrolled: no register moves), in different widths.
// One wave per SIMD and two.  Cycles from s_memtime are normalised by the MFMA-only run of the same launch shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define M32(acc, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
#define SB() __builtin_amdgcn_sched_barrier(0)
// MODE 0: no LDS   1: 4 x ds_read_b32 (scattered)   2: 2 x ds_read_b64   3: 1 x ds_read_b128   4: 1 x ds_read_b32
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.0f + (i & 7) * 0.25f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float r0[4], r1[4], r2[4], r3[4];
    for (int i = 0; i < 4; ++i) r0[i] = r1[i] = r2[i] = r3[i] = 1.0f + i;
    const float b = b0 + threadIdx.x * 2e-3f;
    const float *lp = lds + (threadIdx.x & 63) * 4;  // 16-byte aligned per lane
#define RELOAD(r, u)                                                                                                   \
    if (MODE == 1) { r[0] = lp[(u) * 1040]; r[1] = lp[(u) * 1040 + 261]; r[2] = lp[(u) * 1040 + 522]; r[3] = lp[(u) * 1040 + 783]; } \
    else if (MODE == 2) { f32x2 v0 = *reinterpret_cast<const f32x2 *>(lp + (u) * 1040), v1 = *reinterpret_cast<const f32x2 *>(lp + (u) * 1040 + 522); \
                          r[0] = v0[0]; r[1] = v0[1]; r[2] = v1[0]; r[3] = v1[1]; }                                    \
    else if (MODE == 3) { f32x4 v = *reinterpret_cast<const f32x4 *>(lp + (u) * 1040); r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; } \
    else if (MODE == 4) { r[0] = lp[(u) * 1040]; }
#define STEP(r, u)                                                    \
    {                                                                 \
        float a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3];            \
        SB();                                                         \
        M32(acc[0], a0, b); M32(acc[1], a1, b); M32(acc[2], a2, b); M32(acc[3], a3, b); \
        SB();                                                         \
        RELOAD(r, u)                                                  \
        SB();                                                         \
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 4) { STEP(r0, 0) STEP(r1, 1) STEP(r2, 2) STEP(r3, 3) }
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
    const int iters = 4000;
    double base[3] = {0, 0, 0};
#define R(name, MODE)                                                                                              \
    for (int bpc = 1; bpc <= 2; ++bpc) {                                                                           \
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(256 * bpc), dim3(256), 0, 0, out, cyc, iters, 0.5f); \
        (void)hipDeviceSynchronize();                                                                              \
        unsigned long long h[512]; (void)hipMemcpy(h, cyc, 8 * 256 * bpc, hipMemcpyDeviceToHost);                 \
        double s = 0; for (int i = 0; i < 256 * bpc; ++i) s += (double)h[i];                                       \
        s = s / (256 * bpc) / iters;                                                                               \
        if (MODE == 0) base[bpc] = s;                                                                              \
        printf("%-34s waves/SIMD %d : %7.1f ticks per k-step per wave, %.3f x the MFMA-only loop\n", name, bpc, s, s / base[bpc]); \
    }
    R("no LDS", 0) R("4 x ds_read_b32 (scattered)", 1) R("2 x ds_read_b64", 2) R("1 x ds_read_b128", 3) R("1 x ds_read_b32", 4)
    return 0;
}
