// How much of the non-MFMA work of a k-step hides in the shadow of v_mfma_f32_32x32x2_f32 (64 cycles each)?
// One wave per SIMD (grid 256 x 256 threads) and two; cycles per iteration of {4 MFMA + extras} from s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define M32(acc, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
#define SB() __builtin_amdgcn_sched_barrier(0)

// MODE 0: MFMA only          1: + 4 independent v_cndmask        2: 4 v_cndmask feeding the MFMAs right behind them
// MODE 3: 4 v_cndmask feeding the NEXT iteration's MFMAs (issued after this iteration's MFMAs)
// MODE 4: + 4 ds_read_b32 (ring, consumed 3 iterations later, no VALU)   5: ds_read + dependent cndmask (the conv2 k-step)
// MODE 6: ds_read + cndmask for the next iteration issued after the MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float a0, float b0, unsigned m) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a[4], nx[4], ring[3][4];
    for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x * 1e-3f * (i + 1); nx[i] = a[i]; for (int r = 0; r < 3; ++r) ring[r][i] = a[i]; }
    float b = b0 + threadIdx.x * 2e-3f;
    const bool pm = (m >> (threadIdx.x & 7)) & 1;
    const float *lp = lds + (threadIdx.x & 63);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float cur[4];
            if (MODE == 4 || MODE == 5 || MODE == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = ring[u][i];
#pragma unroll
                for (int i = 0; i < 4; ++i) ring[u][i] = lp[64 * i + 256 * u + ((it & 1) << 10)];
            }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nx[i] = pm ? nx[i] : b;
            }
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = pm ? nx[i] : 0.0f;
            }
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = pm ? cur[i] : 0.0f;
            }
            if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = cur[i];
            }
            SB();
#pragma unroll
            for (int i = 0; i < 4; ++i) M32(acc[i], a[i], b);
            SB();
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = pm ? nx[i] : 0.0f;
                SB();
            }
            if (MODE == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = pm ? cur[i] : 0.0f;
                SB();
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s + nx[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * 256 * 4096);
    hipMalloc(&cyc, 8 * 4096);
    const int iters = 3000;
#define R(name, MODE)                                                                                              \
    for (int bpc = 1; bpc <= 2; ++bpc) {                                                                           \
        hipLaunchKernelGGL((k<MODE>), dim3(256 * bpc), dim3(256), 0, 0, out, cyc, iters, 1.0f, 0.5f, 0xA5u);      \
        hipLaunchKernelGGL((k<MODE>), dim3(256 * bpc), dim3(256), 0, 0, out, cyc, iters, 1.0f, 0.5f, 0xA5u);      \
        hipDeviceSynchronize();                                                                                    \
        unsigned long long h[512]; hipMemcpy(h, cyc, 8 * 256 * bpc, hipMemcpyDeviceToHost);                       \
        double s = 0; for (int i = 0; i < 256 * bpc; ++i) s += (double)h[i];                                       \
        printf("%-58s waves/SIMD %d : %7.1f cycles per k-step (4 MFMA = 256)\n", name, bpc, s / (256 * bpc) / iters); \
    }
    R("0 MFMA only", 0) R("1 + 4 independent v_cndmask", 1) R("2 4 v_cndmask feeding the MFMAs behind them", 2)
    R("3 4 v_cndmask for the next k-step, after the MFMAs", 3) R("4 + 4 ds_read_b32 ring (3 ahead)", 4)
    R("5 ds_read ring + dependent cndmask (conv2 k-step)", 5) R("6 ds_read ring + cndmask for next step after the MFMAs", 6)
    return 0;
}
