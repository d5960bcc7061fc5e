// Does v_mfma_f32_16x16x4_f32 issue every 32 cycles when cycling over N in-place accumulators?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define M16V(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define M16A(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define M32V(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) { if (MODE == 0) M16V(acc[i]); else M16A(acc[i]); }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) M32V(acc[i]);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char *name, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; bpc *= 2) {
        int grid = 256 * bpc, it = 600 / bpc;
        for (int rep = 0; rep < 3; ++rep) launch(grid, it);
        hipEventRecord(e0);
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) launch(grid, it);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * it * 8 * 2048.0;
        printf("%-28s waves/SIMD %d : %7.1f us, %.1f TFLOP/s\n", name, bpc, ms / reps * 1e3, flops / (ms / reps * 1e-3) / 1e12);
    }
}
int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 4096);
#define R(name, ...) run(name, [&](int g, int it) { hipLaunchKernelGGL((__VA_ARGS__), dim3(g), dim3(256), 0, 0, out, it, 1.0f, 0.5f); })
    R("16x16x4 VGPR 8 acc", k<8, 0>); R("16x16x4 VGPR 4 acc", k<4, 0>); R("16x16x4 VGPR 2 acc", k<2, 0>); R("16x16x4 VGPR 1 acc", k<1, 0>);
    R("16x16x4 AGPR 8 acc", k<8, 1>); R("16x16x4 AGPR 2 acc", k<2, 1>);
    R("32x32x2 VGPR 4 acc", k32<4>); R("32x32x2 VGPR 2 acc", k32<2>); R("32x32x2 VGPR 1 acc", k32<1>);
    return 0;
}
