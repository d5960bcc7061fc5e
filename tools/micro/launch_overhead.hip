// Per-launch cost of back-to-back kernels in one stream (HIP events over N launches) against the span the waves
// themselves see (s_memrealtime first start -> last end): what a kernel boundary costs on this chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void k_empty(unsigned long long *st) {
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); st[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}
// every block writes `per_block` floats (streaming stores), optionally after spinning `spin` clocks
__global__ void k_write(float *out, size_t per_block, unsigned long long *st, long long spin) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    long long c0 = clock64();
    while (clock64() - c0 < spin) {}
    float *o = out + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}
template <typename F>
void run(const char *name, int blocks, unsigned long long *d_st, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), d_st, 16 * blocks, hipMemcpyDeviceToHost);
    unsigned long long s = ~0ull, e = 0;
    for (int i = 0; i < blocks; ++i) { s = std::min(s, h[2 * i]); e = std::max(e, h[2 * i + 1]); }
    printf("%-44s per launch %7.1f us   in-kernel span %7.1f us\n", name, ms / reps * 1e3, (e - s) / 100.0);
}
int main() {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, sizeof(float) * (64u << 20));
    (void)hipMalloc(&st, 16 * 65536);
    run("empty 256 x 512", 256, st, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, st); });
    run("empty 256 x 512, 141 KB LDS", 256, st, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 141 * 1024, 0, st); });
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_write), hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024);
    for (int mb : {1, 8, 32, 64, 128}) {
        char nm[96];
        size_t per = (size_t)mb * (1 << 20) / 4 / 256;
        snprintf(nm, sizeof nm, "write %3d MB, 256 x 512", mb);
        run(nm, 256, st, [&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(512), 0, 0, out, per, st, 0LL); });
        snprintf(nm, sizeof nm, "spin 100 us + write %3d MB", mb);
        run(nm, 256, st, [&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(512), 0, 0, out, per, st, 10000LL); });
    }
    return 0;
}
