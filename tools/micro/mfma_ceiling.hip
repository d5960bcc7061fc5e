// Ceiling check for the f32 matrix pipe on gfx950: independent v_mfma chains with no memory traffic at all.
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_peak tools/micro/mfma_ceiling.hip && /tmp/mfma_peak
// Prints TFLOP/s for 32x32x2 and 16x16x4 f32 at 1 and 2 waves per SIMD and NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char *name, F launch, double flop_per_wave_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = flop_per_wave_iter * iters * blocks * 4.0;
    printf("%-34s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms * 1e-9);
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    // 32x32x2: 4096 flop per instruction; 4 x NACC instructions per loop iteration
#define R32(N, BLK) run("32x32x2 f32 NACC=" #N, [&](int b, int it) { hipLaunchKernelGGL((k32<N>), dim3(b), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 4096.0 * 4 * N, BLK, iters);
#define R16(N, BLK) run("16x16x4 f32 NACC=" #N, [&](int b, int it) { hipLaunchKernelGGL((k16<N>), dim3(b), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 2048.0 * 4 * N, BLK, iters);
    printf("-- one wave per SIMD (256 blocks of 4 waves)\n");
    R32(1, 256) R32(2, 256) R32(4, 256) R32(8, 256) R32(16, 256)
    R16(4, 256) R16(8, 256) R16(16, 256)
    printf("-- two waves per SIMD (512 blocks)\n");
    R32(4, 512) R32(8, 512)
    R16(8, 512)
    printf("-- four waves per SIMD (1024 blocks)\n");
    R32(4, 1024)
    return 0;
}
