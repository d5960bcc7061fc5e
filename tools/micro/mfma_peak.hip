// Micro-benchmark: what do the f32-input MFMA shapes sustain on this chip for kernel durations like ours
// (tens of microseconds), with 1/2/4 waves per SIMD?  Pure register loop, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char *name, F launch, double flops_per_iter_per_wave) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 4; bpc *= 2)
        for (int iters : {600, 6000}) {
            int grid = 256 * bpc, it = iters / bpc;
            for (int rep = 0; rep < 3; ++rep) launch(grid, it);
            hipEventRecord(e0);
            const int reps = 20;
            for (int rep = 0; rep < reps; ++rep) launch(grid, it);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 * it * flops_per_iter_per_wave;
            printf("%s waves/SIMD %d iters %5d : %7.1f us, %.1f TFLOP/s\n", name, bpc, it, ms / reps * 1e3, flops / (ms / reps * 1e-3) / 1e12);
        }
}
int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 4096);
    run("16x16x4 x8acc", [&](int g, int it) { hipLaunchKernelGGL(k16<8>, dim3(g), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 8 * 2048.0);
    run("16x16x4 x2acc", [&](int g, int it) { hipLaunchKernelGGL(k16<2>, dim3(g), dim3(256), 0, 0, out, it * 4, 1.0f, 0.5f); }, 4 * 2 * 2048.0);
    run("32x32x2 x4acc", [&](int g, int it) { hipLaunchKernelGGL(k32<4>, dim3(g), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 4 * 4096.0);
    run("32x32x2 x1acc", [&](int g, int it) { hipLaunchKernelGGL(k32<1>, dim3(g), dim3(256), 0, 0, out, it * 4, 1.0f, 0.5f); }, 4 * 4096.0);
    return 0;
}
