// One 3x3 convolution layer of the trunk (32 -> 32 channels, pad 1, 8x8 planes, folded bias, ReLU) as an EXACT block-fixed-point
// integer GEMM on the int8 matrix pipe -- the construction of DESIGN section 10 applied to a convolution; a measurement for the next
// round, not product code.  A wave owns a board pair; its activations live in LDS as three int8 digit planes [plane][board][position +
// one zero slot][32 channels]; a tap of the 3x3 stencil is ONE v_mfma_i32_32x32x32_i8 step (K = 32 input channels):
//   A = the tap's weights  [32 oc rows][32 ic]   (lane: oc = lane & 31, 16 bytes of ic per half)
//   B = the shifted plane  [32 ic][32 positions] (lane: position = lane & 31, 16 bytes of ic per half; outside the plane: the zero slot)
//   C [oc][position]: a lane holds, for ITS position, four consecutive output channels per register group -> the next layer's digits
//   are packed in-lane (no cross-lane traffic) and written as dwords.
// Per layer and pair: 9 taps x 4 position tiles x 9 digit pairs = 324 MFMAs, five int32 accumulators per tile; the board's next
// exponent is a wave-local maximum.  The output digits + exponents are checked against a CPU restatement (exact integers).
// Three variants: a board pair per wave at one wave per SIMD (k_conv_i8), one board per wave at two waves per SIMD (k_conv_i8_w8<false>: one
// wave's epilogue runs under the other's MFMAs), and the latter with an all-integer re-quantisation (k_conv_i8_w8<true>, its own
// specification).  MI355X: 11.5 / 8.8 / 11.6 us per layer and board pair (the 324 MFMAs alone: 4.3 us): the epilogue bounds all three, and
// 64-bit integer arithmetic is no cheaper than the float64 combination (profiles/r04_micro_conv_i8.txt).
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/conv_i8 tools/micro/conv_i8.hip && /tmp/conv_i8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define NCH 32
#define P 64          // positions of an 8x8 plane
#define SLOTS (P + 1) // + the zero slot
#define ACT_PLANE (2 * SLOTS * NCH)                 // bytes of one digit plane of a pair
#define ACT_BYTES (3 * ACT_PLANE)
#define W_BYTES (9 * 3 * NCH * NCH)

__host__ __device__ inline int q_value(float x, int E) {
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7f800000u) == 0x7f800000u) return 0;
    return (int)rintf(ldexpf(x, 148 - E));
}
__host__ __device__ inline void q_digits(int q, int &d0, int &d1, int &d2) {
    d0 = ((q + 128) & 255) - 128;
    const int q1 = (q - d0) >> 8;
    d1 = ((q1 + 128) & 255) - 128;
    d2 = (q1 - d1) >> 8;
}

// in / out: [pair][3 planes][2 boards][SLOTS][32] int8; ein / eout: [pair][2]; w: [9][3][32 oc][32 ic]; ew, bias: [32]
__global__ __launch_bounds__(256) void k_conv_i8(const int8_t *__restrict__ in, const int *__restrict__ ein, const int8_t *__restrict__ w, const int *__restrict__ ew,
                                                 const float *__restrict__ bias, int8_t *__restrict__ out, int *__restrict__ eout, int n_pairs, int reps) {
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *ws = lds;                                   // weights, shared by the workgroup's four waves
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    int8_t *ai = lds + W_BYTES + wave * 2 * ACT_BYTES;  // this wave's input and output activations
    int8_t *ao = ai + ACT_BYTES;
    for (int i = tid; i < W_BYTES / 16; i += 256) reinterpret_cast<int4 *>(ws)[i] = reinterpret_cast<const int4 *>(w)[i];
    __syncthreads();
    const int ewv = ew[r];  // unused lanes' value is never taken: the epilogue reads ew per output channel below
    (void)ewv;
    for (int pair = blockIdx.x * 4 + wave; pair < n_pairs; pair += gridDim.x * 4) {
        const int8_t *src = in + (size_t)pair * ACT_BYTES;
        for (int i = lane; i < ACT_BYTES / 16; i += 64) reinterpret_cast<int4 *>(ai)[i] = reinterpret_cast<const int4 *>(src)[i];
        const int e0 = ein[2 * pair], e1 = ein[2 * pair + 1];
        for (int rep = 0; rep < reps; ++rep) {
            // one board after the other: its two position tiles x five accumulators = 160 registers live at a time
            int Eo[2];
#pragma unroll
            for (int bd = 0; bd < 2; ++bd) {
                v16i acc[2][5];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int s = 0; s < 5; ++s)
#pragma unroll
                        for (int t = 0; t < 16; ++t) acc[jj][s][t] = 0;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    v4i a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const v4i *>(ws + ((tap * 3 + p) * NCH + r) * NCH + 16 * h);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int pos = 32 * jj + r, y = (pos >> 3) + dy, x = (pos & 7) + dx;
                        const int sp = (y >= 0 && y < 8 && x >= 0 && x < 8) ? y * 8 + x : P;
                        const int8_t *bp = ai + (bd * SLOTS + sp) * NCH + 16 * h;
                        v4i b[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const v4i *>(bp + p * ACT_PLANE);
#define PAIR(PA, PB) acc[jj][PA + PB] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[PA], b[PB], acc[jj][PA + PB], 0, 0, 0);
                        PAIR(0, 0) PAIR(0, 1) PAIR(0, 2) PAIR(1, 2) PAIR(2, 2) PAIR(1, 0) PAIR(1, 1) PAIR(2, 1) PAIR(2, 0)
#undef PAIR
                    }
                }
                // epilogue: C[oc][position]; register t: oc = (t & 3) + 8 (t >> 2) + 4 h, position column = r of tile jj
                float v[2][16];
                unsigned mx = 0;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int oc = (t & 3) + 8 * (t >> 2) + 4 * h;
                        const double hi = fma((double)acc[jj][4][t], 65536.0, fma((double)acc[jj][3][t], 256.0, (double)acc[jj][2][t]));
                        const double lo = fma((double)acc[jj][1][t], 256.0, (double)acc[jj][0][t]);
                        float val = (float)ldexp(fma(hi, 65536.0, lo), (bd == 0 ? e0 : e1) + ew[oc] - 296) + bias[oc];
                        val = val > 0.0f ? val : 0.0f;
                        v[jj][t] = val;
                        unsigned bits;
                        memcpy(&bits, &val, 4);
                        mx = max(mx, bits);
                    }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
                int E = (int)(mx >> 23);
                E = E < 1 ? 1 : (E > 254 ? 254 : E);
                Eo[bd] = E;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int pos = 32 * jj + r;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {  // four consecutive output channels 8 g + 4 h ..
                        unsigned wd[3] = {0, 0, 0};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            int d0, d1, d2;
                            q_digits(q_value(v[jj][4 * g + k], E), d0, d1, d2);
                            wd[0] |= (unsigned)(d0 & 255) << (8 * k); wd[1] |= (unsigned)(d1 & 255) << (8 * k); wd[2] |= (unsigned)(d2 & 255) << (8 * k);
                        }
#pragma unroll
                        for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned *>(ao + p * ACT_PLANE + (bd * SLOTS + pos) * NCH + 8 * g + 4 * h) = wd[p];
                    }
                }
            }
            const int E0 = Eo[0], E1 = Eo[1];
            if (lane == 0 && rep == reps - 1) { eout[2 * pair] = E0; eout[2 * pair + 1] = E1; }
            __builtin_amdgcn_s_waitcnt(0);
        }
        // zero slots of the output, then copy out
        if (lane < 2 * 3 * 2) {  // (plane, board, 16-byte half) of the zero slots
            const int p = lane / 4, b = (lane / 2) % 2, hh = lane % 2;
            *reinterpret_cast<int4 *>(ao + p * ACT_PLANE + (b * SLOTS + P) * NCH + 16 * hh) = make_int4(0, 0, 0, 0);
        }
        int8_t *dst = out + (size_t)pair * ACT_BYTES;
        for (int i = lane; i < ACT_BYTES / 16; i += 64) reinterpret_cast<int4 *>(dst)[i] = reinterpret_cast<const int4 *>(ao)[i];
    }
}

// The same layer with ONE board per wave and eight waves per workgroup (two per SIMD): one wave's epilogue (VALU: float64 combination,
// re-quantisation) runs under the other wave's MFMAs.  in / out as above, a "pair" being two consecutive boards.
template <bool INTQ>
__global__ __launch_bounds__(512) void k_conv_i8_w8(const int8_t *__restrict__ in, const int *__restrict__ ein, const int8_t *__restrict__ w, const int *__restrict__ ew,
                                                    const float *__restrict__ bias, int8_t *__restrict__ out, int *__restrict__ eout, int n_pairs, int reps) {
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *ws = lds;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    constexpr int BP = SLOTS * NCH;              // bytes of one digit plane of ONE board
    int8_t *ai = lds + W_BYTES + wave * 6 * BP;  // [3][SLOTS][32] in, then the same out
    int8_t *ao = ai + 3 * BP;
    for (int i = tid; i < W_BYTES / 16; i += 512) reinterpret_cast<int4 *>(ws)[i] = reinterpret_cast<const int4 *>(w)[i];
    __syncthreads();
    for (int board = blockIdx.x * 8 + wave; board < 2 * n_pairs; board += gridDim.x * 8) {
        const int pair = board >> 1, bd = board & 1;
        for (int p = 0; p < 3; ++p)
            for (int i = lane; i < BP / 16; i += 64)
                reinterpret_cast<int4 *>(ai + p * BP)[i] = reinterpret_cast<const int4 *>(in + (size_t)pair * ACT_BYTES + p * ACT_PLANE + bd * BP)[i];
        const int e0 = ein[board];
        for (int rep = 0; rep < reps; ++rep) {
            v16i acc[2][5];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int s = 0; s < 5; ++s)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[jj][s][t] = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                v4i a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const v4i *>(ws + ((tap * 3 + p) * NCH + r) * NCH + 16 * h);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int pos = 32 * jj + r, y = (pos >> 3) + dy, x = (pos & 7) + dx;
                    const int sp = (y >= 0 && y < 8 && x >= 0 && x < 8) ? y * 8 + x : P;
                    v4i b[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const v4i *>(ai + p * BP + sp * NCH + 16 * h);
#define PAIR(PA, PB) acc[jj][PA + PB] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[PA], b[PB], acc[jj][PA + PB], 0, 0, 0);
                    PAIR(0, 0) PAIR(0, 1) PAIR(0, 2) PAIR(1, 2) PAIR(2, 2) PAIR(1, 0) PAIR(1, 1) PAIR(2, 1) PAIR(2, 0)
#undef PAIR
                }
            }
            int E;
            if constexpr (!INTQ) {
                float v[2][16];
                unsigned mx = 0;
    #pragma unroll
                for (int jj = 0; jj < 2; ++jj)
    #pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int oc = (t & 3) + 8 * (t >> 2) + 4 * h;
                        const double hi = fma((double)acc[jj][4][t], 65536.0, fma((double)acc[jj][3][t], 256.0, (double)acc[jj][2][t]));
                        const double lo = fma((double)acc[jj][1][t], 256.0, (double)acc[jj][0][t]);
                        float val = (float)ldexp(fma(hi, 65536.0, lo), e0 + ew[oc] - 296) + bias[oc];
                        val = val > 0.0f ? val : 0.0f;
                        v[jj][t] = val;
                        unsigned bits;
                        memcpy(&bits, &val, 4);
                        mx = max(mx, bits);
                    }
    #pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
                E = (int)(mx >> 23);
                E = E < 1 ? 1 : (E > 254 ? 254 : E);
    #pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int pos = 32 * jj + r;
    #pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned wd[3] = {0, 0, 0};
    #pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            int d0, d1, d2;
                            q_digits(q_value(v[jj][4 * g + k], E), d0, d1, d2);
                            wd[0] |= (unsigned)(d0 & 255) << (8 * k); wd[1] |= (unsigned)(d1 & 255) << (8 * k); wd[2] |= (unsigned)(d2 & 255) << (8 * k);
                        }
    #pragma unroll
                        for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned *>(ao + p * BP + pos * NCH + 8 * g + 4 * h) = wd[p];
                    }
                }

            } else {
                // integer re-quantisation: Y = max(X + B, 0) with the bias at the accumulator's scale (value = Y 2^s, s = e + ew[oc] - 296);
                // the board's next exponent from the exact leading bit; q' = Y 2^(s + 148 - E') rounded to nearest even by one shift
                long long Bq[16];
                int soc[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int oc = (t & 3) + 8 * (t >> 2) + 4 * h;
                    soc[t] = e0 + ew[oc] - 296;
                    Bq[t] = (long long)rint(ldexp((double)bias[oc], -soc[t]));
                }
                int mxe = -100000;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        long long X = (long long)acc[jj][0][t] + ((long long)acc[jj][1][t] << 8) + ((long long)acc[jj][2][t] << 16) + ((long long)acc[jj][3][t] << 24) +
                                      ((long long)acc[jj][4][t] << 32);
                        long long Y = X + Bq[t];
                        if (Y > 0) mxe = max(mxe, 63 - __clzll(Y) + soc[t]);
                    }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mxe = max(mxe, __shfl_xor(mxe, o));
                E = mxe + 127;
                E = E < 1 ? 1 : (E > 254 ? 254 : E);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int pos = 32 * jj + r;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned wd[3] = {0, 0, 0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const int t = 4 * g + kk;
                            long long X = (long long)acc[jj][0][t] + ((long long)acc[jj][1][t] << 8) + ((long long)acc[jj][2][t] << 16) + ((long long)acc[jj][3][t] << 24) +
                                          ((long long)acc[jj][4][t] << 32);
                            long long Y = X + Bq[t];
                            Y = Y > 0 ? Y : 0;
                            const int sh = E - 148 - soc[t];
                            int q;
                            if (sh >= 63) q = 0;
                            else if (sh > 0) q = (int)((Y + ((1ll << (sh - 1)) - 1 + ((Y >> sh) & 1))) >> sh);
                            else q = (int)(Y << (-sh));
                            int d0, d1, d2;
                            q_digits(q, d0, d1, d2);
                            wd[0] |= (unsigned)(d0 & 255) << (8 * kk); wd[1] |= (unsigned)(d1 & 255) << (8 * kk); wd[2] |= (unsigned)(d2 & 255) << (8 * kk);
                        }
#pragma unroll
                        for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned *>(ao + p * BP + pos * NCH + 8 * g + 4 * h) = wd[p];
                    }
                }
            }
            if (lane == 0 && rep == reps - 1) eout[board] = E;
            __builtin_amdgcn_s_waitcnt(0);
        }
        if (lane < 6) *reinterpret_cast<int4 *>(ao + (lane / 2) * BP + P * NCH + 16 * (lane % 2)) = make_int4(0, 0, 0, 0);
        for (int p = 0; p < 3; ++p)
            for (int i = lane; i < BP / 16; i += 64)
                reinterpret_cast<int4 *>(out + (size_t)pair * ACT_BYTES + p * ACT_PLANE + bd * BP)[i] = reinterpret_cast<const int4 *>(ao + p * BP)[i];
    }
}

int main() {
    const int n_pairs = 4096, reps_timed = 64;
    srand(1);
    std::vector<float> W(9 * NCH * NCH), bias(NCH);  // W[tap][oc][ic]
    for (auto &x : W) x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto &x : bias) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    std::vector<int> ew(NCH);
    std::vector<int8_t> wd(W_BYTES);
    std::vector<int> wq(9 * NCH * NCH);
    for (int oc = 0; oc < NCH; ++oc) {  // one exponent per output channel over its 9 x 32 weights
        float m = 0;
        for (int t = 0; t < 9; ++t) for (int ic = 0; ic < NCH; ++ic) m = fmaxf(m, fabsf(W[(t * NCH + oc) * NCH + ic]));
        uint32_t b; memcpy(&b, &m, 4);
        int E = (int)(b >> 23); E = E < 1 ? 1 : (E > 254 ? 254 : E);
        ew[oc] = E;
        for (int t = 0; t < 9; ++t) for (int ic = 0; ic < NCH; ++ic) {
            const int q = q_value(W[(t * NCH + oc) * NCH + ic], E);
            wq[(t * NCH + oc) * NCH + ic] = q;
            int d[3]; q_digits(q, d[0], d[1], d[2]);
            for (int p = 0; p < 3; ++p) wd[((t * 3 + p) * NCH + oc) * NCH + ic] = (int8_t)d[p];
        }
    }
    // inputs: random non-negative activations per board, quantised with the board's exponent
    std::vector<int8_t> in((size_t)n_pairs * ACT_BYTES, 0);
    std::vector<int> ein(2 * n_pairs), inq((size_t)n_pairs * 2 * P * NCH);
    for (int pr = 0; pr < n_pairs; ++pr)
        for (int b = 0; b < 2; ++b) {
            std::vector<float> a(P * NCH);
            float m = 0;
            for (auto &x : a) { x = (rand() % 3 == 0) ? 0.f : rand() / (float)RAND_MAX * (1 + pr % 7); m = fmaxf(m, x); }
            uint32_t bb; memcpy(&bb, &m, 4);
            int E = (int)(bb >> 23); E = E < 1 ? 1 : (E > 254 ? 254 : E);
            ein[2 * pr + b] = E;
            for (int pos = 0; pos < P; ++pos) for (int ic = 0; ic < NCH; ++ic) {
                const int q = q_value(a[pos * NCH + ic], E);
                inq[(((size_t)pr * 2 + b) * P + pos) * NCH + ic] = q;
                int d[3]; q_digits(q, d[0], d[1], d[2]);
                for (int p = 0; p < 3; ++p) in[(size_t)pr * ACT_BYTES + p * ACT_PLANE + (b * SLOTS + pos) * NCH + ic] = (int8_t)d[p];
            }
        }
    int8_t *d_in, *d_out, *d_w; int *d_ein, *d_eout, *d_ew; float *d_bias;
    hipMalloc(&d_in, in.size()); hipMalloc(&d_out, in.size()); hipMalloc(&d_w, W_BYTES);
    hipMalloc(&d_ein, ein.size() * 4); hipMalloc(&d_eout, ein.size() * 4); hipMalloc(&d_ew, NCH * 4); hipMalloc(&d_bias, NCH * 4);
    hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice); hipMemcpy(d_w, wd.data(), W_BYTES, hipMemcpyHostToDevice);
    hipMemcpy(d_ein, ein.data(), ein.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_ew, ew.data(), NCH * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_bias, bias.data(), NCH * 4, hipMemcpyHostToDevice);
    const int lds = W_BYTES + 4 * 2 * ACT_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_i8), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_i8_w8<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_i8_w8<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int variant = 0; variant < 3; ++variant) {
    printf("== %s\n", variant == 0 ? "one board pair per wave, four waves per workgroup" : (variant == 1 ? "one board per wave, eight waves per workgroup (two per SIMD)" : "the same with the integer re-quantisation epilogue (its own specification, restated on the CPU)"));
    hipMemset(d_out, 0x55, in.size());
#define LAUNCH(np_, reps_)                                                                                                       \
    if (variant == 0) hipLaunchKernelGGL(k_conv_i8, dim3(256), dim3(256), lds, 0, d_in, d_ein, d_w, d_ew, d_bias, d_out, d_eout, np_, reps_); \
    else if (variant == 1) hipLaunchKernelGGL(k_conv_i8_w8<false>, dim3(256), dim3(512), lds, 0, d_in, d_ein, d_w, d_ew, d_bias, d_out, d_eout, np_, reps_); \
    else hipLaunchKernelGGL(k_conv_i8_w8<true>, dim3(256), dim3(512), lds, 0, d_in, d_ein, d_w, d_ew, d_bias, d_out, d_eout, np_, reps_);
    LAUNCH(n_pairs, 1)
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<int8_t> out(in.size()); std::vector<int> eout(ein.size());
    hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost); hipMemcpy(eout.data(), d_eout, eout.size() * 4, hipMemcpyDeviceToHost);
    // CPU restatement on a sample of pairs: exact integer convolution, the same conversions (variant 2: the integer re-quantisation)
    long long bad = 0, checked = 0;
    for (int pr = 0; pr < n_pairs; pr += 97)
        for (int b = 0; b < 2; ++b) {
            std::vector<long long> Xs(P * NCH);
            for (int pos = 0; pos < P; ++pos) for (int oc = 0; oc < NCH; ++oc) {
                long long X = 0;
                for (int t = 0; t < 9; ++t) {
                    const int y = (pos >> 3) + t / 3 - 1, x = (pos & 7) + t % 3 - 1;
                    if (y < 0 || y >= 8 || x < 0 || x >= 8) continue;
                    const int *a = &inq[(((size_t)pr * 2 + b) * P + y * 8 + x) * NCH], *ww = &wq[(t * NCH + oc) * NCH];
                    for (int ic = 0; ic < NCH; ++ic) X += (long long)a[ic] * ww[ic];
                }
                Xs[pos * NCH + oc] = X;
            }
            std::vector<int> qn(P * NCH);
            int E;
            if (variant < 2) {
                std::vector<float> v(P * NCH);
                float m = 0;
                for (int pos = 0; pos < P; ++pos) for (int oc = 0; oc < NCH; ++oc) {
                    float val = (float)ldexp((double)Xs[pos * NCH + oc], ein[2 * pr + b] + ew[oc] - 296) + bias[oc];
                    val = val > 0.f ? val : 0.f;
                    v[pos * NCH + oc] = val; m = fmaxf(m, val);
                }
                uint32_t bb; memcpy(&bb, &m, 4);
                E = (int)(bb >> 23); E = E < 1 ? 1 : (E > 254 ? 254 : E);
                for (int i = 0; i < P * NCH; ++i) qn[i] = q_value(v[i], E);
            } else {
                std::vector<long long> Y(P * NCH);
                int mxe = -100000;
                for (int pos = 0; pos < P; ++pos) for (int oc = 0; oc < NCH; ++oc) {
                    const int so = ein[2 * pr + b] + ew[oc] - 296;
                    long long y = Xs[pos * NCH + oc] + (long long)rint(ldexp((double)bias[oc], -so));
                    if (y > 0) { const int fl = 63 - __builtin_clzll((unsigned long long)y) + so; mxe = fl > mxe ? fl : mxe; }
                    Y[pos * NCH + oc] = y > 0 ? y : 0;
                }
                E = mxe + 127; E = E < 1 ? 1 : (E > 254 ? 254 : E);
                for (int pos = 0; pos < P; ++pos) for (int oc = 0; oc < NCH; ++oc) {
                    const int so = ein[2 * pr + b] + ew[oc] - 296, sh = E - 148 - so;
                    const long long y = Y[pos * NCH + oc];
                    qn[pos * NCH + oc] = sh >= 63 ? 0 : (sh > 0 ? (int)((y + ((1ll << (sh - 1)) - 1 + ((y >> sh) & 1))) >> sh) : (int)(y << (-sh)));
                }
            }
            if (E != eout[2 * pr + b]) ++bad;
            for (int pos = 0; pos < P; ++pos) for (int oc = 0; oc < NCH; ++oc) {
                int d[3]; q_digits(qn[pos * NCH + oc], d[0], d[1], d[2]);
                for (int p = 0; p < 3; ++p) { ++checked; if (out[(size_t)pr * ACT_BYTES + p * ACT_PLANE + (b * SLOTS + pos) * NCH + oc] != (int8_t)d[p]) ++bad; }
            }
        }
    printf("check: %lld digits of %d boards compared with the CPU restatement, %lld differ\n", checked, 2 * ((n_pairs + 96) / 97), bad);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    for (int np : {1024, 2048, 4096}) {  // 1, 2, 4 pairs per wave on 1024 waves
        LAUNCH(np, 4)
        hipDeviceSynchronize();
        hipEventRecord(t0);
        LAUNCH(np, reps_timed)
        hipEventRecord(t1); hipEventSynchronize(t1);
        float ms; hipEventElapsedTime(&ms, t0, t1);
        const double layers = (double)np * reps_timed;  // layer-pairs executed
        const double us = 1e3 * ms / (reps_timed * (np / 1024.0));  // one layer of one board pair per wave, all 1024 waves of the chip at once
        (void)layers;
        printf("%d pairs x %d layers: %.3f ms = %.2f us per layer and board pair (= %.1f K cycles at 2.4 GHz; 324 MFMAs are 10.4 K; a direct f32 layer of k_trunk2 takes ~23 K)\n",
               np, reps_timed, ms, us, us * 2.4);
    }
    if (bad) return 1;
    }
    return 0;
}
