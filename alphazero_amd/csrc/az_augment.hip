// az_augment.hip -- symmetry augmentation of self-play samples on the device.
//
// Replaces the Python loop of AlphaZeroTrainer.self_play (trainer.py:275-284) with its helpers
// Sample.create_reflection_twin / create_rotation_twin (trainer.py:80-118) and the networks'
// reflect_neural_output / rotate_neural_output (othello.py:414-450, connect4.py:437-445,
// tictactoe.py:343-367): every sample with move_idx >= 2 gets, in this order,
//   reflection_horizontal, rotation_90, reflection_horizontal+rotation_90, rotation_180,
//   reflection_horizontal+rotation_180, rotation_270, reflection_horizontal+rotation_270
// (Connect4: the reflection only).  Pure byte/float permutations: bit-exact, HBM-bound
// (per twin: cells B + 4 A B read and written).
#include <hipcub/hipcub.hpp>

#include "az_device.h"
#include "az_host.h"

__global__ void k_aug_flags(const int *meta, long long S, int *flags) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) flags[i] = meta[i * 4 + 1] >= 2 ? 1 : 0;  // trainer.py:278
}

// source cell (r,c) of output cell (i,j) under transform code t: 0 none, 1 reflH, 2 rot90, 3 reflH+rot90, 4 rot180, ...
// np.flip(axis=1): out[i][j] = in[i][W-1-j];  np.rot90(m,k): k=1 out[i][j] = in[j][n-1-i], k=2 in[n-1-i][n-1-j], k=3 in[n-1-j][i]
AZ_D void aug_source(int t, int n, int W, int i, int j, int *r, int *c) {
    int k = t >> 1;  // quarter turns
    int rr, cc;
    if (k == 0) { rr = i; cc = j; }
    else if (k == 1) { rr = j; cc = n - 1 - i; }
    else if (k == 2) { rr = n - 1 - i; cc = n - 1 - j; }
    else { rr = n - 1 - j; cc = i; }
    if (t & 1) cc = W - 1 - cc;  // the rotation acts on the reflected board
    *r = rr; *c = cc;
}

__global__ void k_augment(GameDesc gd, int n_twins, const int8_t *state, const float *pi, const int8_t *z, const int *meta,
                          const int *flags, const int *offs, long long S, int8_t *o_state, float *o_pi, int8_t *o_z, int *o_meta) {
    long long i = (long long)blockIdx.x;
    if (i >= S || !flags[i]) return;
    const int H = gd.H, W = gd.W, cells = gd.cells, A = gd.A;
    for (int tw = 0; tw < n_twins; ++tw) {
        const int t = tw + 1;
        const long long o = (long long)offs[i] * n_twins + tw;
        for (int x = threadIdx.x; x < cells; x += blockDim.x) {
            int r, c;
            aug_source(t, H, W, x / W, x % W, &r, &c);
            o_state[o * cells + x] = state[i * cells + r * W + c];
        }
        for (int a = threadIdx.x; a < A; a += blockDim.x) {
            float v;
            if (gd.game == AZ_CONNECT4) v = pi[i * A + (A - 1 - a)];  // np.flip of the 7 column priors
            else if (a >= cells) v = pi[i * A + a];                   // pass entry is kept in place
            else { int r, c; aug_source(t, H, W, a / W, a % W, &r, &c); v = pi[i * A + r * W + c]; }
            o_pi[o * A + a] = v;
        }
        if (threadIdx.x == 0) {
            o_z[o] = z[i];
            o_meta[o * 4 + 0] = meta[i * 4 + 0]; o_meta[o * 4 + 1] = meta[i * 4 + 1];
            o_meta[o * 4 + 2] = meta[i * 4 + 2]; o_meta[o * 4 + 3] = t;  // transformation code instead of the action
        }
    }
}

#define AZ_TRY(x) do { int _rc = (x); if (_rc != AZ_OK) return _rc; } while (0)

extern "C" int az_augment_count(int game, const int32_t *d_meta, int64_t S, int64_t *n_out, void *stream) {
    AZ_REQUIRE(n_out && S >= 0, AZ_EINVAL, "bad arguments");
    *n_out = 0;
    if (S == 0) return AZ_OK;
    AZ_REQUIRE(d_meta, AZ_EINVAL, "null meta");
    hipStream_t st = (hipStream_t)stream;
    int *flags = nullptr, *total = nullptr;
    AZ_HIP(hipMalloc((void **)&flags, sizeof(int) * S));
    AZ_HIP(hipMalloc((void **)&total, sizeof(int)));
    hipLaunchKernelGGL(k_aug_flags, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, d_meta, (long long)S, flags);
    void *tmp = nullptr; size_t tmp_bytes = 0;
    AZ_HIP(hipcub::DeviceReduce::Sum(tmp, tmp_bytes, flags, total, (int)S, st));
    AZ_HIP(hipMalloc(&tmp, tmp_bytes));
    AZ_HIP(hipcub::DeviceReduce::Sum(tmp, tmp_bytes, flags, total, (int)S, st));
    int h = 0;
    AZ_HIP(hipMemcpyAsync(&h, total, sizeof(int), hipMemcpyDeviceToHost, st));
    AZ_HIP(hipStreamSynchronize(st));
    (void)hipFree(tmp); (void)hipFree(flags); (void)hipFree(total);
    int twins = game == AZ_CONNECT4 ? 1 : 7;  // DATA_AUGMENT_STRATEGIES, games/registers.py:37-50
    *n_out = (int64_t)h * twins;
    return AZ_OK;
}

extern "C" int az_augment(int game, int H, int W, const int8_t *d_state, const float *d_pi, const int8_t *d_z, const int32_t *d_meta,
                          int64_t S, int8_t *d_out_state, float *d_out_pi, int8_t *d_out_z, int32_t *d_out_meta, int64_t out_capacity,
                          void *stream) {
    GameDesc gd;
    AZ_TRY(az_make_game_desc(game, H, W, &gd));
    if (S == 0) return AZ_OK;
    AZ_REQUIRE(d_state && d_pi && d_z && d_meta && d_out_state && d_out_pi && d_out_z && d_out_meta, AZ_EINVAL, "null argument");
    AZ_REQUIRE(game == AZ_CONNECT4 || H == W, AZ_EINVAL, "rotations need a square board");
    AZ_REQUIRE(S < (1LL << 31), AZ_EINVAL, "too many samples");
    hipStream_t st = (hipStream_t)stream;
    const int twins = game == AZ_CONNECT4 ? 1 : 7;
    int *flags = nullptr, *offs = nullptr;
    AZ_HIP(hipMalloc((void **)&flags, sizeof(int) * S));
    AZ_HIP(hipMalloc((void **)&offs, sizeof(int) * S));
    hipLaunchKernelGGL(k_aug_flags, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, d_meta, (long long)S, flags);
    void *tmp = nullptr; size_t tmp_bytes = 0;
    AZ_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, offs, (int)S, st));
    AZ_HIP(hipMalloc(&tmp, tmp_bytes));
    AZ_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, offs, (int)S, st));
    int last_off = 0, last_flag = 0;
    AZ_HIP(hipMemcpyAsync(&last_off, offs + (S - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    AZ_HIP(hipMemcpyAsync(&last_flag, flags + (S - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    AZ_HIP(hipStreamSynchronize(st));
    int rc = AZ_OK;
    if ((int64_t)(last_off + last_flag) * twins > out_capacity) {
        az_set_error("augmentation needs room for %lld samples, got %lld", (long long)(last_off + last_flag) * twins, (long long)out_capacity);
        rc = AZ_ECAPACITY;
    } else {
        hipLaunchKernelGGL(k_augment, dim3((unsigned)S), dim3(64), 0, st, gd, twins, d_state, d_pi, d_z, d_meta, flags, offs, (long long)S,
                           d_out_state, d_out_pi, d_out_z, d_out_meta);
        if (hipStreamSynchronize(st) != hipSuccess) { az_set_error("augmentation kernel failed"); rc = AZ_EHIP; }
    }
    (void)hipFree(tmp); (void)hipFree(flags); (void)hipFree(offs);
    return rc;
}
