// az_net.hip -- policy-value network forward on MI355X (gfx950), float32 in / float32 accumulate.
//
// Replaces OthelloNet / Connect4Net / TicTacToeNet .forward in eval mode (othello.py:341-382,
// connect4.py:370-412, tictactoe.py:289-316) + PolicyValueNetwork.predict (base.py:350-355).
//
// All matrix work runs on the f32-input matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32): exact f32
// products and a k-ordered f32 accumulation chain, i.e. the reference's arithmetic type (the
// reference computes in torch float32), not a reduced-precision path.  Eval-mode BatchNorm is
// folded into the preceding layer in float64 at upload (az_net_commit).
//
// Accumulation order (restated by the CPU oracle so tests can demand bit equality):
//   conv : acc = b'[oc]; for tap = ky*3+kx ascending, for ic ascending: acc = fma(in, w', acc); relu
//   dense: acc = b'[n];  for k ascending: acc = fma(x[k], W'[n][k], acc); (relu)
//   heads: softmax as m = max, e = det_exp(l - m), S = sequential sum, p = e / S; v = det_tanh
//
// Kernels
//   k_trunk2<CH,CW>: conv1..4, two boards per wavefront on 32x32x2 MFMA, persistent 8-wave workgroups, activations
//                    never leave the wave's private LDS region; weights stream from L2 already tiled in MFMA
//                    B-fragment order (one coalesced dword per lane).  Batches from 4096 boards up.
//   k_trunk<CH,CW> : the same layers, one wavefront per board on 16x16x4 MFMA: smaller batches.
//   k_gemm<...>    : LDS-tiled f32 MFMA GEMM with bias(+ReLU) epilogue for fc1 / fc2.
//   k_heads        : policy+value GEMM (N padded to 16) fused with softmax / tanh.
//   k_mlp          : the 316-parameter TicTacToe MLP, one thread per board.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "az_device.h"
#include <hip/hip_ext.h>
#include "az_host.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define NCH 32
// LDS plane stride: the smallest ODD value > x -- the 16 / 32 output channels a wave instruction writes at one position land
// on distinct banks (oc * PS mod 32 is a permutation for odd PS), and slot x of every plane is a spare (kept zero where a
// kernel redirects out-of-plane reads to it).  8x8 planes: 65, 7x6: 43, 6x6: 37.
#define PLANE_STRIDE(x) (((x) + 1) | 1)

#ifdef AZ_PROBE  // diagnostic build only (make PROBE=1): per-phase shader-clock stamps of wave 0 of every block
__device__ unsigned long long az_probe_buf[8192 * 8];  // k_gemm: 8 words per block; k_trunk2: 8 words per (block, wave)
#define STAMP(var) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(var)
#endif

struct TrunkParams {
    const float *w1f, *b1;       // conv1 B-fragment order [3 k-steps][2][64] (taps 9..11 zero), bias [32]
    const float *cb[3];          // conv2..4 folded bias [32]
    const float *w1p;            // conv1, 32x32x2 B-fragment order [5 k-steps][64] (tap 9 zero)
    const float *wp[3];          // conv2..4, 32x32x2 B fragments, four k-steps per lane contiguous: [9 taps][4][64 lanes][4]
    const float *wq[3];          // conv2..4, 16x16x4 B fragments (k_trunk; k_trunk2's 16-row tiles): [9 taps][4][64 lanes][4] (fragment i = 2 j + nt)
    const float *wu;             // conv2 in the Winograd F(2x2,3x3) form: U = G g' G^T as 16x16x4 B fragments
                                 // [2 passes][8 k-steps][4][64 lanes][4]: fragment e = 2 f8 + nt of frequency f = 8 p + f8
    const float *wu32;           // the same U as 32x32x2 B fragments [4][16 frequencies][64 lanes][4 k-steps] (conv2_wino32)
};

#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define A_RING 3  // A-operand LDS reads are issued this many k-steps ahead of the MFMAs that use them

// implicit-GEMM 3x3 convolution of one board on MFMA: M = P_OUT positions (MT tiles of 16),
// N = 32 oc (2 tiles), K = 9 taps x 32 ic walked tap-major.
//   in_lds : [32 ic][IN_PS] planes of width IN_W.  PAD = 0: "valid" conv, output (y,x) reads in[(y+ky)*IN_W + x+kx].
//            PAD = 1: "same" conv on an un-haloed plane: out-of-plane taps are read (from in-bounds
//            LDS of the same wave) and replaced by 0 through a per-lane 9-bit validity mask.
//   Weight fragments of one tap (8 k-steps x 2 oc tiles) sit in registers; the next tap's 16 coalesced
//   dword loads are issued a whole tap ahead so L2 latency hides under the MFMAs.
template <int P_OUT, int W_OUT, int H_OUT, int IN_W, int IN_PS, int PAD, int MT>
AZ_D void conv_mfma(const float *in_lds, const float *__restrict__ wf, const float *__restrict__ bias, int lane,
                    f32x4 (&acc)[MT][2]) {
    const int m_lane = lane & 15, kq = lane >> 4;
    int abase[MT];
    unsigned vmask[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int p = 16 * mt + m_lane;
        p = p < P_OUT ? p : P_OUT - 1;
        const int y = p / W_OUT, x = p % W_OUT;
        abase[mt] = kq * IN_PS + y * IN_W + x - PAD * (IN_W + 1);
        unsigned vm = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int iy = y + t / 3 - PAD, ix = x + t % 3 - PAD;
            vm |= (unsigned)(iy >= 0 && iy < H_OUT + 2 - 2 * PAD && ix >= 0 && ix < W_OUT + 2 - 2 * PAD) << t;
        }
        vmask[mt] = vm;
    }
    const float *wl = wf + 4 * lane;  // fragment i = 2 j + nt of a tap sits at [tap][i / 4][lane][i % 4]: 16-byte loads
    float bfr[2][16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(wl + q * 256);
        bfr[0][4 * q] = v[0]; bfr[0][4 * q + 1] = v[1]; bfr[0][4 * q + 2] = v[2]; bfr[0][4 * q + 3] = v[3];
    }
    const float bv0 = bias[m_lane], bv1 = bias[16 + m_lane];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt][0] = (f32x4){bv0, bv0, bv0, bv0};
        acc[mt][1] = (f32x4){bv1, bv1, bv1, bv1};
    }
    float ar[A_RING][MT];
#define CONV_LOAD(c, mt) in_lds[abase[mt] + ((c) / 8 / 3) * IN_W + ((c) / 8 % 3) + 4 * ((c) % 8) * IN_PS]
#pragma unroll
    for (int c = 0; c < A_RING; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ar[c][mt] = CONV_LOAD(c, mt);
#pragma unroll
    for (int c = 0; c < 72; ++c) {
        const int tap = c / 8, j = c % 8;
        if (j == 0 && tap < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(wl + ((tap + 1) * 4 + q) * 256);
                bfr[(tap + 1) & 1][4 * q] = v[0]; bfr[(tap + 1) & 1][4 * q + 1] = v[1];
                bfr[(tap + 1) & 1][4 * q + 2] = v[2]; bfr[(tap + 1) & 1][4 * q + 3] = v[3];
            }
        }
        float ac[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v = ar[c % A_RING][mt];
            if (PAD) v = ((vmask[mt] >> tap) & 1u) ? v : 0.0f;
            ac[mt] = v;
        }
        if (c + A_RING < 72) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ar[c % A_RING][mt] = CONV_LOAD(c + A_RING, mt);
        }
        // keep the loads above issued BEFORE this step's MFMAs (hipcc otherwise sinks them to their
        // first use and every k-step stalls on LDS / L2 latency)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][0] = MFMA(ac[mt], bfr[tap & 1][2 * j + 0], acc[mt][0]);
            acc[mt][1] = MFMA(ac[mt], bfr[tap & 1][2 * j + 1], acc[mt][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CONV_LOAD
}

template <int P_OUT, int OUT_PS, int MT>
AZ_D void store_relu_lds(float *out, int lane, const f32x4 (&acc)[MT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = 16 * mt + (lane >> 4) * 4 + r;
                float v = acc[mt][nt][r];
                if (m < P_OUT) out[(nt * 16 + (lane & 15)) * OUT_PS + m] = v > 0.0f ? v : 0.0f;
            }
}

// ---------------------------------------------------------------------------------------------
// conv2 ("same", 32 -> 32) of ONE board in the Winograd F(2x2,3x3) form: 2.25x
// fewer multiplications than the direct form (conv2 is 54 % of the trunk's work).  The board's 2x2 output tiles are the rows of a 16x16x4 MFMA tile (8x8 plane:
// 16 tiles, 7x6: 12), and for each of the 16 "frequencies" f = 4 i + j the products over the input channels are one
// k-ordered MFMA chain:   M[f][tile][oc] = sum over ic ascending of V[f][tile][ic] * U[f][ic][oc]      (from 0)
// with V = B^T d B of the tile's zero-padded 4x4 input patch d and U = G g' G^T folded at upload.
//   * every lane transforms its own operand: lane (m, kq) reads the patch of tile m in plane 4 j + kq straight from the
//     un-haloed LDS planes (out-of-plane elements are redirected to the plane's zero slot) and forms V with 16 adds;
//   * two passes of 8 frequencies (frequency rows i = 0,1 / 2,3: patch rows 0-2 / 1-3), so that 64 accumulator registers
//     are live instead of 128; the inverse transform Y = A^T M A is taken along the frequency COLUMNS first, which lets
//     pass 0 hand over four values per output tile and channel (32 registers);
//   * U arrives from L2 in B-fragment order, one k-step ahead; the next k-step's patch is read under the MFMAs.
// Arithmetic (restated operation for operation by the CPU oracle's conv2_winograd): T = rows of B^T d, V = T B, the chains above,
// R[i][0] = (M[i][0]+M[i][1])+M[i][2], R[i][1] = (M[i][1]-M[i][2])-M[i][3], Y[0][c] = (R[0][c]+R[1][c])+R[2][c],
// Y[1][c] = (R[1][c]-R[2][c])-R[3][c], out = relu(Y + bias): one IEEE operation per step, in this order.
// The output overwrites the input planes (all reads are done before the first write).
// ---------------------------------------------------------------------------------------------
template <int CH, int CW, int PS, int RING>
AZ_D void conv2_wino(float *act, const float4 *wu4, const float *__restrict__ bias, int lane) {
    constexpr int TW = (CW + 1) / 2, NTL = ((CH + 1) / 2) * TW, P1 = CH * CW;
    static_assert(NTL <= 16 && PS > P1, "one 16-row tile of 2x2 output tiles per board; a zero slot behind every plane");
    const int m = lane & 15, kq = lane >> 4;
    const int t = m < NTL ? m : NTL - 1;  // padding rows repeat the last tile (finite operands, never stored)
    const int ty = t / TW, tx = t % TW;
    // wu4: the U fragments, in global memory (RING = 3: requested two k-steps ahead) or copied into the workgroup's LDS (RING = 2)
    const float4 *ul = wu4 + lane;
    float cu[2][4][2], cw_[2][4][2];  // pass 0 -> pass 1: R[0][c] + R[1][c] and R[1][c] per (nt, r)
    float yo[2][4][2][2];             // Y[i][c] per (nt, r)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int off[12];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int iy = 2 * ty - 1 + a + p, ix = 2 * tx - 1 + b;
                off[a * 4 + b] = kq * PS + ((iy >= 0 && iy < CH && ix >= 0 && ix < CW) ? iy * CW + ix : P1);
            }
        f32x4 acc[8][2];
#pragma unroll
        for (int f = 0; f < 8; ++f) { acc[f][0] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; acc[f][1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        // U fragments: a ring of RING k-steps, requested RING - 1 k-steps ahead of their use
        float4 ub[RING][4];
#pragma unroll
        for (int jj = 0; jj < RING - 1; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) ub[jj][q] = ul[(size_t)((p * 8 + jj) * 4 + q) * 64];
        float d[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) d[e] = act[off[e]];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float dn[12];
            if (j + RING - 1 < 8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ub[(j + RING - 1) % RING][q] = ul[(size_t)((p * 8 + j + RING - 1) * 4 + q) * 64];
            }
            if (j + 1 < 8) {
#pragma unroll
                for (int e = 0; e < 12; ++e) dn[e] = act[off[e] + 4 * (j + 1) * PS];
            }
            float T[2][4], V[8];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (p == 0) { T[0][b] = d[b] - d[8 + b]; T[1][b] = d[4 + b] + d[8 + b]; }      // T[0] = d0 - d2, T[1] = d1 + d2
                else { T[0][b] = d[4 + b] - d[b]; T[1][b] = d[b] - d[8 + b]; }                 // T[2] = d2 - d1, T[3] = d1 - d3 (rows 1..3 loaded)
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                V[4 * i + 0] = T[i][0] - T[i][2]; V[4 * i + 1] = T[i][1] + T[i][2];
                V[4 * i + 2] = T[i][2] - T[i][1]; V[4 * i + 3] = T[i][1] - T[i][3];
            }
            __builtin_amdgcn_sched_barrier(0);  // the loads above stay issued ahead of this k-step's MFMAs
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const float4 u = ub[j % RING][f >> 1];
                acc[f][0] = MFMA(V[f], (f & 1) ? u.z : u.x, acc[f][0]);
                acc[f][1] = MFMA(V[f], (f & 1) ? u.w : u.y, acc[f][1]);
            }
#pragma unroll
            for (int f = 0; f < 8; ++f) { asm volatile("" : "+a"(acc[f][0])); asm volatile("" : "+a"(acc[f][1])); }
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < 8) {
#pragma unroll
                for (int e = 0; e < 12; ++e) d[e] = dn[e];
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float R[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    R[i][0] = (acc[4 * i + 0][nt][r] + acc[4 * i + 1][nt][r]) + acc[4 * i + 2][nt][r];
                    R[i][1] = (acc[4 * i + 1][nt][r] - acc[4 * i + 2][nt][r]) - acc[4 * i + 3][nt][r];
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (p == 0) { cu[nt][r][c] = R[0][c] + R[1][c]; cw_[nt][r][c] = R[1][c]; }
                    else { yo[nt][r][0][c] = cu[nt][r][c] + R[0][c]; yo[nt][r][1][c] = (cw_[nt][r][c] - R[0][c]) - R[1][c]; }
                }
            }
    }
    LDS_FENCE();  // every read of the input planes has returned: the output may overwrite them
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float bv = bias[nt * 16 + m];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tt = 4 * kq + r, y0 = 2 * (tt / TW), x0 = 2 * (tt % TW);  // C layout of 16x16x4: row 4 (lane >> 4) + r, column lane & 15
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float v = yo[nt][r][i][c] + bv;
                    if (tt < NTL && y0 + i < CH && x0 + c < CW) act[(nt * 16 + m) * PS + (y0 + i) * CW + x0 + c] = v > 0.0f ? v : 0.0f;
                }
        }
    }
}

// The same Winograd conv2 for BOTH boards of a wave at once on v_mfma_f32_32x32x2_f32: the 2 x 16 output tiles of two 8x8
// boards are the 32 rows of one MFMA tile, so a U fragment feeds 64 cycles of matrix work (half the fragment rate of the
// 16-row form) and comes from LDS.  All 16 frequency accumulators are live at once (256 registers): this form runs in the
// one-wave-per-SIMD variant of k_trunk2 (4 waves per workgroup, 512 registers per wave).  Same chains (ic ascending per
// frequency, from 0), same transforms, same operation order as conv2_wino and the oracle: identical bits.
//   u2: [jp (8)][f (16)][lane (64)] float2 = k-steps 2 jp, 2 jp + 1 of frequency f: U[f][ic = 2 j + (lane >> 5)][oc = lane & 31]
// Software pipeline, two k-steps deep: while the 16 MFMAs of k-step j run on V(j), the VALU transforms the patch of k-step j + 1
// (read from LDS during k-step j - 1) into V(j + 1) and the LDS reads of the patch of k-step j + 2 go out.  With one wave per SIMD
// a transform that sits between its own LDS reads and its own MFMAs stops the matrix pipe for an LDS latency twice per k-step
// (measured: conv2 at 61 % MFMA occupancy in that form).
template <int CH, int CW, int PS, int OFF1>
AZ_D void conv2_wino32(float *act, const float2 *u2, const float *__restrict__ bias, int lane) {
    constexpr int TW = (CW + 1) / 2, NTL = ((CH + 1) / 2) * TW, P1 = CH * CW;
    static_assert(2 * NTL <= 32 && PS > P1, "the tiles of two boards fill one 32-row MFMA tile");
    const int m = lane & 31, kk = lane >> 5;
    const int mc = m < 2 * NTL ? m : 2 * NTL - 1;  // padding rows repeat the last tile (never stored)
    const int bd = mc / NTL, t = mc % NTL, ty = t / TW, tx = t % TW;
    int off[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int iy = 2 * ty - 1 + a, ix = 2 * tx - 1 + b;
            off[a * 4 + b] = bd * OFF1 + kk * PS + ((iy >= 0 && iy < CH && ix >= 0 && ix < CW) ? iy * CW + ix : P1);
        }
    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.0f;
    const float2 *ul = u2 + lane;
    float d[16], V[2][16];
    // the same 32 IEEE additions as conv2_wino, two per instruction where the pairing allows (v_pk_add_f32): rows over column pairs
    // (stage 1: T = B^T d, eight packed operations), then per row i (stage 2, three operations) (V[i][0], V[i][3]) = (T[i][0], T[i][1]) -
    // (T[i][2], T[i][3]) in one packed subtraction, V[i][1] = T[i][1] + T[i][2], V[i][2] = T[i][2] - T[i][1]
#define DP(e) ((f32x2){d[e], d[(e) + 1]})
#define WINO32_STAGE1(q, Tl, Th)                              \
    {                                                         \
        if ((q) == 0) Tl[0] = DP(0) - DP(8);                  \
        if ((q) == 1) Th[0] = DP(2) - DP(10);                 \
        if ((q) == 2) Tl[1] = DP(4) + DP(8);                  \
        if ((q) == 3) Th[1] = DP(6) + DP(10);                 \
        if ((q) == 4) Tl[2] = DP(8) - DP(4);                  \
        if ((q) == 5) Th[2] = DP(10) - DP(6);                 \
        if ((q) == 6) Tl[3] = DP(4) - DP(12);                 \
        if ((q) == 7) Th[3] = DP(6) - DP(14);                 \
    }
#define WINO32_STAGE2(i, c, Tl, Th, Vout)                                                          \
    {                                                                                              \
        if ((c) == 0) { const f32x2 v03 = Tl[i] - Th[i]; Vout[4 * (i) + 0] = v03.x; Vout[4 * (i) + 3] = v03.y; } \
        if ((c) == 1) Vout[4 * (i) + 1] = Tl[i].y + Th[i].x;                                       \
        if ((c) == 2) Vout[4 * (i) + 2] = Th[i].x - Tl[i].y;                                       \
    }
    // fence that instruction selection honours as well as the scheduler (a bare sched_barrier lets LDS reads gather in front of it)
#define WINO32_FENCE() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = act[off[e]];
    float2 ub[2][16];  // U fragments of k-steps (2 jp, 2 jp + 1), double-buffered
#pragma unroll
    for (int f = 0; f < 16; ++f) ub[0][f] = ul[f * 64];
    {
        f32x2 Tl[4], Th[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) WINO32_STAGE1(q, Tl, Th)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) WINO32_STAGE2(i, c, Tl, Th, V[0])
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = act[off[e] + 2 * PS];
    WINO32_FENCE()
    // one wave per SIMD: nothing else fills the matrix pipe, so everything that is not an MFMA is dealt out by hand into the 64-cycle
    // shadows of this k-step's 16 MFMAs.  Slot g, in front of MFMA g: stage-1 operation g (g < 8), one stage-2 operation (slots 2-13),
    // two LDS reads of the patch of k-step j + 2 (slots 8-15, once stage 1 has consumed d; rows in the order the next stage 1 wants
    // them) and, on even k-steps, the U fragment pair of frequency g for k-steps j + 2, j + 3.
    constexpr int DORD[16] = {0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x2 Tl[4], Th[4];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (j + 1 < 16) {
                if (g < 8) WINO32_STAGE1(g, Tl, Th)
                if (g >= 2 && g < 14) WINO32_STAGE2((g - 2) / 3, (g - 2) % 3, Tl, Th, V[(j + 1) & 1])
            }
            if (j + 2 < 16 && g >= 8) {
                const int e0 = DORD[2 * (g - 8)], e1 = DORD[2 * (g - 8) + 1];
                d[e0] = act[off[e0] + 2 * (j + 2) * PS];
                d[e1] = act[off[e1] + 2 * (j + 2) * PS];
            }
            if (j % 2 == 0 && j + 2 < 16) ub[(j / 2 + 1) & 1][g] = ul[((j / 2 + 1) * 16 + g) * 64];
            WINO32_FENCE()
            acc[g] = MFMA32(V[j & 1][g], (j % 2 == 0) ? ub[(j / 2) & 1][g].x : ub[(j / 2) & 1][g].y, acc[g]);
            asm volatile("" : "+a"(acc[g]));
            WINO32_FENCE()
        }
    }
#undef WINO32_STAGE1
#undef WINO32_STAGE2
#undef WINO32_FENCE
#undef DP
    LDS_FENCE();  // every read of the input planes has returned: the output may overwrite them
    const float bv = bias[m];
    const f32x2 bv2 = {bv, bv};
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) {  // two accumulator registers (two neighbouring tiles) per step: packed f32 additions
        f32x2 R[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 a0 = {acc[4 * q + 0][2 * ip], acc[4 * q + 0][2 * ip + 1]}, a1 = {acc[4 * q + 1][2 * ip], acc[4 * q + 1][2 * ip + 1]};
            const f32x2 a2 = {acc[4 * q + 2][2 * ip], acc[4 * q + 2][2 * ip + 1]}, a3 = {acc[4 * q + 3][2 * ip], acc[4 * q + 3][2 * ip + 1]};
            R[q][0] = (a0 + a1) + a2;
            R[q][1] = (a1 - a2) - a3;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x2 y0v = ((R[0][c] + R[1][c]) + R[2][c]) + bv2, y1v = ((R[1][c] - R[2][c]) - R[3][c]) + bv2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * ip + h;
                const int r = 8 * (i / 4) + 4 * kk + (i % 4);  // C layout of 32x32x2: this register's row (tile), column lane & 31
                const int rb = r / NTL, rt = r % NTL, y0 = 2 * (rt / TW), x0 = 2 * (rt % TW);
                const float u0 = h ? y0v.y : y0v.x, u1 = h ? y1v.y : y1v.x;
                if (r < 2 * NTL && x0 + c < CW) {
                    if (y0 < CH) act[rb * OFF1 + m * PS + y0 * CW + x0 + c] = u0 > 0.0f ? u0 : 0.0f;
                    if (y0 + 1 < CH) act[rb * OFF1 + m * PS + (y0 + 1) * CW + x0 + c] = u1 > 0.0f ? u1 : 0.0f;
                }
            }
        }
    }
}

template <int CH, int CW>
struct TrunkGeom {
    static constexpr int P1 = CH * CW, PW = CW + 2, PH = CH + 2;
    static constexpr int INP = ((PH * PW + 15) / 16) * 16;  // padded input plane (also the guard band in front of the planes)
    static constexpr int PS = PLANE_STRIDE(P1);             // one plane stride for every layer's activations
    static constexpr int H3 = CH - 2, W3 = CW - 2, P3 = H3 * W3;
    static constexpr int H4 = CH - 4, W4 = CW - 4, P4 = H4 * W4;
    static constexpr int MT2 = (P1 + 15) / 16, MT3 = (P3 + 15) / 16, MT4 = (P4 + 15) / 16;
    static constexpr int TAIL = 16;  // guard band behind the planes (masked out-of-plane reads of the last channel)
    static constexpr int WAVE_FLOATS = INP + NCH * PS + TAIL;
    static constexpr int LDS_BYTES = 4 * WAVE_FLOATS * 4;
    static_assert(INP >= CW + 1 && TAIL >= CW + 1, "guard bands for the masked out-of-plane reads");
};

// One wavefront per board; activations never leave the wave's private LDS region and every layer writes
// its output IN PLACE over its input (the outputs wait in the MFMA accumulators until the layer's last
// LDS read has been consumed):
//   inp : (CH+2)x(CW+2) zero-padded input plane        act : [32 ch][PS] planes, conv1 -> conv2 -> conv3 outputs
// 8.8 KB of LDS per wave -> four 4-wave blocks per CU, i.e. four waves per SIMD: the MFMA pipe always has
// another wave's k-steps to run while one wave is in a load / store / conv1 phase.
template <int CH, int CW, bool WINO>
__global__ __launch_bounds__(256, 2) void k_trunk(const float *__restrict__ in, int B, const int *__restrict__ dyn_count, TrunkParams tp, float *__restrict__ feat) {
    using G = TrunkGeom<CH, CW>;
    if (dyn_count) { int c = *dyn_count; B = c < B ? c : B; }  // rows actually filled this step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;  // waves are independent: no workgroup barrier below
    float *inp = smem + wave * G::WAVE_FLOATS;
    float *act = inp + G::INP;
    for (int i = lane; i < G::INP; i += 64) inp[i] = 0.0f;
    if (WINO && lane < NCH) act[lane * G::PS + G::P1] = 0.0f;  // every plane's zero slot: where the Winograd patches read outside the plane
    LDS_FENCE();
    for (int p = lane; p < G::P1; p += 64) inp[(p / CW + 1) * G::PW + (p % CW) + 1] = in[(size_t)b * G::P1 + p];
    LDS_FENCE();
    {  // conv1 1->32, pad 1 (othello.py:370) as a K = 12 MFMA product: taps 0..8, taps 9..11 carry zero weights
        f32x4 acc[G::MT2][2];
        const int m_lane = lane & 15, kq = lane >> 4;
        const float bv0 = tp.b1[m_lane], bv1 = tp.b1[16 + m_lane];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            int tap = 4 * s + kq;
            tap = tap < 9 ? tap : 8;  // finite operand for the zero-weight columns
            const int toff = (tap / 3) * G::PW + tap % 3;
            const float b0 = tp.w1f[(s * 2 + 0) * 64 + lane], b1 = tp.w1f[(s * 2 + 1) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < G::MT2; ++mt) {
                int p = 16 * mt + m_lane;
                p = p < G::P1 ? p : G::P1 - 1;
                const float a = inp[(p / CW) * G::PW + (p % CW) + toff];
                if (s == 0) { acc[mt][0] = (f32x4){bv0, bv0, bv0, bv0}; acc[mt][1] = (f32x4){bv1, bv1, bv1, bv1}; }
                acc[mt][0] = MFMA(a, b0, acc[mt][0]);
                acc[mt][1] = MFMA(a, b1, acc[mt][1]);
            }
        }
        store_relu_lds<G::P1, G::PS, G::MT2>(act, lane, acc);
    }
    LDS_FENCE();
    if constexpr (WINO) {  // conv2 32->32, pad 1 (othello.py:371), Winograd form
        conv2_wino<CH, CW, G::PS, 3>(act, reinterpret_cast<const float4 *>(tp.wu), tp.cb[0], lane);
    } else {  // conv2 32->32, pad 1 (othello.py:371)
        f32x4 acc[G::MT2][2];
        conv_mfma<G::P1, CW, CH, CW, G::PS, 1, G::MT2>(act, tp.wq[0], tp.cb[0], lane, acc);
        LDS_FENCE();
        store_relu_lds<G::P1, G::PS, G::MT2>(act, lane, acc);
    }
    LDS_FENCE();
    {  // conv3 32->32, valid (othello.py:372)
        f32x4 acc[G::MT3][2];
        conv_mfma<G::P3, G::W3, G::H3, CW, G::PS, 0, G::MT3>(act, tp.wq[1], tp.cb[1], lane, acc);
        LDS_FENCE();
        store_relu_lds<G::P3, G::PS, G::MT3>(act, lane, acc);
    }
    LDS_FENCE();
    {  // conv4 32->32, valid (othello.py:373) -> flattened NCHW features (othello.py:374)
        f32x4 acc[G::MT4][2];
        conv_mfma<G::P4, G::W4, G::H4, G::W3, G::PS, 0, G::MT4>(act, tp.wq[2], tp.cb[2], lane, acc);
        float *fo = feat + (size_t)b * (NCH * G::P4);
#pragma unroll
        for (int mt = 0; mt < G::MT4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = 16 * mt + (lane >> 4) * 4 + r;
                    float v = acc[mt][nt][r];
                    if (m < G::P4) fo[(nt * 16 + (lane & 15)) * G::P4 + m] = v > 0.0f ? v : 0.0f;
                }
    }
}

// ---------------------------------------------------------------------------------------------
// k_trunk_q: ONE board per WORKGROUP of eight (up to 256 boards) or four waves, for the batches that leave most of the chip idle (a single
// game's search, arenas, self-play waves of a few hundred games): with one wave per board the trunk is one wave's walk through ~860
// dependent-issue MFMAs (18-19 us whatever the batch, up to 1024 boards).  Here the board's 16x16 output tiles of every layer are dealt
// over the waves -- unit u = 2 mt + nt (position tile mt, channel half nt) goes to wave u % NW --, the activations live in LDS planes the
// waves share, and a layer is: compute from the planes into registers, barrier (every wave has finished reading), store in place,
// barrier.  Winograd conv2: the 16 tiles of the board are the 16 rows of the MFMA tile and cannot be split, so waves 0..3 take one
// (channel half, pass) each -- a pass is 8 of the 16 frequencies -- and pass 0 hands its half of the inverse transform to pass 1 through
// LDS across the barrier that is there anyway.  Every output element keeps k_trunk's chain (bias, tap-major / ic-minor; Winograd: the
// same transforms in the same order, ic ascending per frequency): identical bits.  10.3-10.9 us up to 256 boards, 16.0 at 512.
// ---------------------------------------------------------------------------------------------
// one 16x16 tile of conv_mfma: positions 16 mt .. 16 mt + 15, output channels 16 nt .. 16 nt + 15 (mt, nt wave-uniform)
template <int P_OUT, int W_OUT, int H_OUT, int IN_W, int IN_PS, int PAD>
AZ_D f32x4 conv_tile(const float *in_lds, const float *__restrict__ wf, const float *__restrict__ bias, int lane, int mt, int nt) {
    const int m_lane = lane & 15, kq = lane >> 4;
    int p = 16 * mt + m_lane;
    p = p < P_OUT ? p : P_OUT - 1;
    const int y = p / W_OUT, x = p % W_OUT;
    const int abase = kq * IN_PS + y * IN_W + x - PAD * (IN_W + 1);
    unsigned vmask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int iy = y + t / 3 - PAD, ix = x + t % 3 - PAD;
        vmask |= (unsigned)(iy >= 0 && iy < H_OUT + 2 - 2 * PAD && ix >= 0 && ix < W_OUT + 2 - 2 * PAD) << t;
    }
    const float *wl = wf + 4 * lane;  // fragment i = 2 j + nt of a tap sits at [tap][i / 4][lane][i % 4]
    float bfr[2][8];
#define TILE_BLOAD(tap_, buf_)                                                             \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                        \
        const f32x4 v = *reinterpret_cast<const f32x4 *>(wl + ((tap_) * 4 + q) * 256);     \
        bfr[buf_][2 * q] = nt ? v[1] : v[0];                                               \
        bfr[buf_][2 * q + 1] = nt ? v[3] : v[2];                                           \
    }
    TILE_BLOAD(0, 0)
    const float bv = bias[16 * nt + m_lane];
    f32x4 acc = (f32x4){bv, bv, bv, bv};
    float ar[A_RING];
#define TILE_LOAD(c) in_lds[abase + ((c) / 8 / 3) * IN_W + ((c) / 8 % 3) + 4 * ((c) % 8) * IN_PS]
#pragma unroll
    for (int c = 0; c < A_RING; ++c) ar[c] = TILE_LOAD(c);
#pragma unroll
    for (int c = 0; c < 72; ++c) {
        const int tap = c / 8, j = c % 8;
        if (j == 0 && tap < 8) { TILE_BLOAD(tap + 1, (tap + 1) & 1) }
        float v = ar[c % A_RING];
        if (PAD) v = ((vmask >> tap) & 1u) ? v : 0.0f;
        if (c + A_RING < 72) ar[c % A_RING] = TILE_LOAD(c + A_RING);
        __builtin_amdgcn_sched_barrier(0);
        acc = MFMA(v, bfr[tap & 1][j], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef TILE_LOAD
#undef TILE_BLOAD
    return acc;
}

template <int P_OUT, int OUT_PS>
AZ_D void store_tile_relu_lds(float *out, int lane, const f32x4 &acc, int mt, int nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 16 * mt + (lane >> 4) * 4 + r;
        const float v = acc[r];
        if (m < P_OUT) out[(nt * 16 + (lane & 15)) * OUT_PS + m] = v > 0.0f ? v : 0.0f;
    }
}

// ONE pass p (frequency rows 2 p, 2 p + 1) of conv2_wino for ONE channel half nt (the same patch reads, transforms and chains): R[r][i][c] = the column-transformed sums of frequency row 2 p + i for output
// tile 4 (lane >> 4) + r, channel 16 nt + (lane & 15) -- what conv2_wino keeps between its two passes.  p and nt are wave-uniform.
template <int CH, int CW, int PS>
AZ_D void conv2_wino_pass(const float *act, const float4 *wu4, int lane, int nt, int p, float (&R)[4][2][2]) {
    constexpr int TW = (CW + 1) / 2, NTL = ((CH + 1) / 2) * TW, P1 = CH * CW, RING = 3;
    const int m = lane & 15, kq = lane >> 4;
    const int t = m < NTL ? m : NTL - 1;
    const int ty = t / TW, tx = t % TW;
    const float4 *ul = wu4 + lane + (size_t)p * 8 * 4 * 64;
    int off[12];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int iy = 2 * ty - 1 + a + p, ix = 2 * tx - 1 + b;
            off[a * 4 + b] = kq * PS + ((iy >= 0 && iy < CH && ix >= 0 && ix < CW) ? iy * CW + ix : P1);
        }
    f32x4 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float4 ub[RING][4];
#pragma unroll
    for (int jj = 0; jj < RING - 1; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) ub[jj][q] = ul[(size_t)(jj * 4 + q) * 64];
    float d[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) d[e] = act[off[e]];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float dn[12];
        if (j + RING - 1 < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ub[(j + RING - 1) % RING][q] = ul[(size_t)((j + RING - 1) * 4 + q) * 64];
        }
        if (j + 1 < 8) {
#pragma unroll
            for (int e = 0; e < 12; ++e) dn[e] = act[off[e] + 4 * (j + 1) * PS];
        }
        float T[2][4], V[8];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // pass 0: T[0] = d0 - d2, T[1] = d1 + d2 (patch rows 0..2 loaded); pass 1: T[2] = d2 - d1, T[3] = d1 - d3 (rows 1..3 loaded)
            const float t0a = d[b] - d[8 + b], t1a = d[4 + b] + d[8 + b], t0b = d[4 + b] - d[b];
            T[0][b] = p == 0 ? t0a : t0b;
            T[1][b] = p == 0 ? t1a : t0a;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            V[4 * i + 0] = T[i][0] - T[i][2]; V[4 * i + 1] = T[i][1] + T[i][2];
            V[4 * i + 2] = T[i][2] - T[i][1]; V[4 * i + 3] = T[i][1] - T[i][3];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float4 u = ub[j % RING][f >> 1];
            acc[f] = MFMA(V[f], (f & 1) ? (nt ? u.w : u.z) : (nt ? u.y : u.x), acc[f]);
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) asm volatile("" : "+a"(acc[f]));
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < 8) {
#pragma unroll
            for (int e = 0; e < 12; ++e) d[e] = dn[e];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            R[r][i][0] = (acc[4 * i + 0][r] + acc[4 * i + 1][r]) + acc[4 * i + 2][r];
            R[r][i][1] = (acc[4 * i + 1][r] - acc[4 * i + 2][r]) - acc[4 * i + 3][r];
        }
}

template <int CH, int CW, bool WINO, int NW>
__global__ __launch_bounds__(64 * NW) void k_trunk_q(const float *__restrict__ in, int B, const int *__restrict__ dyn_count, TrunkParams tp, float *__restrict__ feat) {
    using G = TrunkGeom<CH, CW>;
    if (dyn_count) { int c = *dyn_count; B = c < B ? c : B; }
    const int b = blockIdx.x;
    if (b >= B) return;  // the whole workgroup: no barrier is left behind
    __shared__ __attribute__((aligned(16))) float smem_q[G::WAVE_FLOATS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float *inp = smem_q;
    float *act = inp + G::INP;
    static_assert(G::P1 <= 256 && G::INP <= 256 && (NW == 4 || NW == 8), "one element per thread");
    const float cell = tid < G::P1 ? in[(size_t)b * G::P1 + tid] : 0.0f;  // in flight while the planes are cleared
    if (tid < G::INP) inp[tid] = 0.0f;
    if (WINO && tid < NCH) act[tid * G::PS + G::P1] = 0.0f;  // every plane's zero slot (Winograd patches outside the plane)
    __syncthreads();
    if (tid < G::P1) inp[(tid / CW + 1) * G::PW + (tid % CW) + 1] = cell;
    __syncthreads();
    constexpr int UPW = 8 / NW;  // units per wave: a layer has at most 2 * 4 of them (MT <= 4)
    static_assert(G::MT2 <= 4 && G::MT3 <= 4 && G::MT4 <= 4, "at most eight 16x16 tiles per layer");
    {   // conv1 1->32, pad 1, K = 12 (taps 9..11 carry zero weights)
        const int m_lane = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave + NW * i;
            if (u < 2 * G::MT2) {
                const int mt = u >> 1, nt = u & 1;
                const float bv = tp.b1[16 * nt + m_lane];
                f32x4 acc = (f32x4){bv, bv, bv, bv};
                int p = 16 * mt + m_lane;
                p = p < G::P1 ? p : G::P1 - 1;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    int tap = 4 * s + kq;
                    tap = tap < 9 ? tap : 8;
                    const float a = inp[(p / CW) * G::PW + (p % CW) + (tap / 3) * G::PW + tap % 3];
                    acc = MFMA(a, tp.w1f[(s * 2 + nt) * 64 + lane], acc);
                }
                store_tile_relu_lds<G::P1, G::PS>(act, lane, acc, mt, nt);  // conv1 reads inp, writes act: no hazard
            }
        }
    }
    __syncthreads();
    if constexpr (WINO) {  // conv2, Winograd form: waves 0..3 = (channel half nt = wave & 1, pass p = wave >> 1)
        constexpr int TW = (CW + 1) / 2, NTL = ((CH + 1) / 2) * TW;
        __shared__ float xch[2][16][64];  // pass 0 -> pass 1 of the same channel half: R[0][c] + R[1][c] and R[1][c] per (r, c)
        float R[4][2][2], yo[4][2][2];
        const int nt2 = wave & 1;
        if (wave < 4) conv2_wino_pass<CH, CW, G::PS>(act, reinterpret_cast<const float4 *>(tp.wu), lane, nt2, wave >> 1, R);
        if (wave < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) { xch[nt2][(r * 2 + c) * 2][lane] = R[r][0][c] + R[r][1][c]; xch[nt2][(r * 2 + c) * 2 + 1][lane] = R[r][1][c]; }
        }
        __syncthreads();  // every read of the conv1 planes has returned; pass 0's half of the inverse transform is in LDS
        if (wave >= 2 && wave < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float cu = xch[nt2][(r * 2 + c) * 2][lane], cw_ = xch[nt2][(r * 2 + c) * 2 + 1][lane];
                    yo[r][0][c] = cu + R[r][0][c];
                    yo[r][1][c] = (cw_ - R[r][0][c]) - R[r][1][c];
                }
            const int m = lane & 15, kq = lane >> 4;
            const float bv = tp.cb[0][nt2 * 16 + m];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tt = 4 * kq + r, y0 = 2 * (tt / TW), x0 = 2 * (tt % TW);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float v = yo[r][i][c] + bv;
                        if (tt < NTL && y0 + i < CH && x0 + c < CW) act[(nt2 * 16 + m) * G::PS + (y0 + i) * CW + x0 + c] = v > 0.0f ? v : 0.0f;
                    }
            }
        }
    } else {
        f32x4 acc[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave + NW * i;
            if (u < 2 * G::MT2) acc[i] = conv_tile<G::P1, CW, CH, CW, G::PS, 1>(act, tp.wq[0], tp.cb[0], lane, u >> 1, u & 1);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave + NW * i;
            if (u < 2 * G::MT2) store_tile_relu_lds<G::P1, G::PS>(act, lane, acc[i], u >> 1, u & 1);
        }
    }
    __syncthreads();
    {   // conv3 32->32, valid
        f32x4 acc[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave + NW * i;
            if (u < 2 * G::MT3) acc[i] = conv_tile<G::P3, G::W3, G::H3, CW, G::PS, 0>(act, tp.wq[1], tp.cb[1], lane, u >> 1, u & 1);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave + NW * i;
            if (u < 2 * G::MT3) store_tile_relu_lds<G::P3, G::PS>(act, lane, acc[i], u >> 1, u & 1);
        }
    }
    __syncthreads();
    {   // conv4 32->32, valid -> flattened NCHW features
        float *fo = feat + (size_t)b * (NCH * G::P4);
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave + NW * i;
            if (u < 2 * G::MT4) {
                const int mt = u >> 1, nt = u & 1;
                const f32x4 acc = conv_tile<G::P4, G::W4, G::H4, G::W3, G::PS, 0>(act, tp.wq[2], tp.cb[2], lane, mt, nt);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mt + (lane >> 4) * 4 + r;
                    const float v = acc[r];
                    if (m < G::P4) fo[(nt * 16 + (lane & 15)) * G::P4 + m] = v > 0.0f ? v : 0.0f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_trunk2: the same four layers with TWO boards per wavefront on v_mfma_f32_32x32x2_f32.
// M = the positions of both boards (tiles of 32 rows), N = the 32 output channels (one tile), K = 9 taps x 32 ic
// walked tap-major, two input channels per MFMA (lane>>5 selects which).  Against the 16x16x4 version this
// halves the LDS operand reads and the weight-fragment loads per FLOP, gives conv4 a full tile (2 x 16 rows) and
// uses the MFMA shape that sustains the higher rate on this chip.  Accumulation order per output is unchanged
// (bias, then tap-major / ic-minor), so results stay bit-identical.
// ---------------------------------------------------------------------------------------------
#define A_RING2 3

// Tiling of a layer's 2 x P_OUT output rows: MT tiles of 32 rows on 32x32x2 and, when at most 16 rows are left
// over, one 16-row tile on 16x16x4 (two 16-wide channel tiles) instead of a fourth-empty 32-row tile -- Othello 8x8
// conv3: 72 rows = 2 x 32 + 8 -> 80 row-slots issued instead of 96.
template <int P_OUT>
struct ConvPlan {
    static constexpr int ROWS = 2 * P_OUT, REM = ROWS % 32;
    static constexpr bool R16 = REM > 0 && REM <= 16;
    static constexpr int MT = ROWS / 32 + ((REM > 16) ? 1 : 0);
    static constexpr int MTA = MT > 0 ? MT : 1;  // array extent
};

// PAD = 1 ("same" conv on un-haloed planes): a tap that falls outside the plane is redirected, per lane and per
// tap, to the plane's spare slot (offset P_IN of every plane, kept zero) -- one address select per tile per TAP, and
// the MFMA consumes the LDS data directly (a per-k-step value select sat on the LDS -> VALU -> MFMA critical path
// and cost 12-25 % of conv2).
//   wf  : 32x32x2 B fragments [9 taps][4][64 lanes][4 k-steps];  w0 = tap 0, preloaded by the caller
//   wf16: 16x16x4 B fragments [9 taps][4][64 lanes][4] (fragment 2 j + nt; only read when the plan has a 16-row tile)
template <int P_OUT, int W_OUT, int H_OUT, int IN_W, int IN_PS, int OFF1, int PAD>
AZ_D void conv32(const float *in_lds, const float *__restrict__ wf, const float *__restrict__ wf16, const float bv, const float (&bv16)[2],
                 const float (&w0)[16], const float (&w16)[16], int lane, f32x16 (&acc)[ConvPlan<P_OUT>::MTA], f32x4 (&acc16)[2]) {
    using PL = ConvPlan<P_OUT>;
    constexpr int MT = PL::MT, MTA = PL::MTA;
    constexpr bool R16 = PL::R16;
    const int m_lane = lane & 31, kk = lane >> 5;
    int abase[MTA], zbase[MTA];
    unsigned vmask[MTA];
#define ROW_SETUP(row, kplane, ab, zb, vm_out)                                                                   \
    {                                                                                                            \
        int r_ = (row);                                                                                          \
        r_ = r_ < 2 * P_OUT ? r_ : 2 * P_OUT - 1;                                                                \
        const int bd_ = r_ >= P_OUT ? 1 : 0, p_ = r_ - bd_ * P_OUT;                                              \
        const int y_ = p_ / W_OUT, x_ = p_ % W_OUT;                                                              \
        ab = bd_ * OFF1 + (kplane) * IN_PS + y_ * IN_W + x_ - PAD * (IN_W + 1);                                  \
        zb = bd_ * OFF1 + (kplane) * IN_PS + (H_OUT + 2 - 2 * PAD) * IN_W; /* the plane's spare zero slot */     \
        unsigned vm_ = 0;                                                                                        \
        _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) {                                                       \
            int iy_ = y_ + t_ / 3 - PAD, ix_ = x_ + t_ % 3 - PAD;                                                \
            vm_ |= (unsigned)(iy_ >= 0 && iy_ < H_OUT + 2 - 2 * PAD && ix_ >= 0 && ix_ < W_OUT + 2 - 2 * PAD) << t_; \
        }                                                                                                        \
        vm_out = vm_;                                                                                            \
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ROW_SETUP(32 * mt + m_lane, kk, abase[mt], zbase[mt], vmask[mt])
    int abase16 = 0, zbase16 = 0;
    unsigned vmask16 = 0;
    if (R16) ROW_SETUP(32 * MT + (lane & 15), lane >> 4, abase16, zbase16, vmask16)
#undef ROW_SETUP
    const float *wl = wf + 4 * lane, *wl16 = wf16 + 4 * lane;
    float bfr[2][16], b16[2][16];  // tap 0 arrives preloaded: its L2 latency was paid under the previous layer
#pragma unroll
    for (int i = 0; i < 16; ++i) { bfr[0][i] = w0[i]; if (R16) b16[0][i] = w16[i]; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = bv;
    if (R16) { acc16[0] = (f32x4){bv16[0], bv16[0], bv16[0], bv16[0]}; acc16[1] = (f32x4){bv16[1], bv16[1], bv16[1], bv16[1]}; }
    int tb[2][MTA], tb16[2] = {0, 0};  // per-tap operand bases (double-buffered: the rings run ahead across tap boundaries)
#define TAP_OFF(t) (((t) / 3) * IN_W + (t) % 3)
#define TAP_BASE(t, mt) (PAD ? (((vmask[mt] >> (t)) & 1u) ? abase[mt] + TAP_OFF(t) : zbase[mt]) : abase[mt] + TAP_OFF(t))
#define TAP_BASE16(t) (PAD ? (((vmask16 >> (t)) & 1u) ? abase16 + TAP_OFF(t) : zbase16) : abase16 + TAP_OFF(t))
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { tb[0][mt] = TAP_BASE(0, mt); tb[1][mt] = TAP_BASE(1, mt); }
    if (R16) { tb16[0] = TAP_BASE16(0); tb16[1] = TAP_BASE16(1); }
    float ar[A_RING2][MTA], ar16[2] = {0.0f, 0.0f};
#define CONV_LOAD(c, mt) in_lds[tb[((c) / 16) & 1][mt] + 2 * ((c) % 16) * IN_PS]
#define CONV_LOAD16(c16) in_lds[tb16[((c16) / 8) & 1] + 4 * ((c16) % 8) * IN_PS]
#pragma unroll
    for (int c = 0; c < A_RING2; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ar[c][mt] = CONV_LOAD(c, mt);
    if (R16) { ar16[0] = CONV_LOAD16(0); ar16[1] = CONV_LOAD16(1); }
#pragma unroll
    for (int c = 0; c < 144; ++c) {
        const int tap = c / 16, j = c % 16;
        if (j == 0 && tap < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // four 16-byte loads per tap instead of sixteen dwords: VMEM issue slots are not free
                const f32x4 v = *reinterpret_cast<const f32x4 *>(wl + ((tap + 1) * 4 + q) * 256);
                bfr[(tap + 1) & 1][4 * q + 0] = v[0]; bfr[(tap + 1) & 1][4 * q + 1] = v[1];
                bfr[(tap + 1) & 1][4 * q + 2] = v[2]; bfr[(tap + 1) & 1][4 * q + 3] = v[3];
                if (R16) {
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(wl16 + ((tap + 1) * 4 + q) * 256);
                    b16[(tap + 1) & 1][4 * q + 0] = u[0]; b16[(tap + 1) & 1][4 * q + 1] = u[1];
                    b16[(tap + 1) & 1][4 * q + 2] = u[2]; b16[(tap + 1) & 1][4 * q + 3] = u[3];
                }
            }
        }
        float ac[MTA], a16 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ac[mt] = ar[c % A_RING2][mt];
        if (R16 && j % 2 == 0) a16 = ar16[(c / 2) % 2];
        if (j == A_RING2 && tap >= 1 && tap < 8) {  // every load of tap-1 has been issued: its base slot takes tap+1
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) tb[(tap + 1) & 1][mt] = TAP_BASE(tap + 1, mt);
            if (R16) tb16[(tap + 1) & 1] = TAP_BASE16(tap + 1);
        }
        if (c + A_RING2 < 144) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ar[c % A_RING2][mt] = CONV_LOAD(c + A_RING2, mt);
        }
        if (R16 && j % 2 == 0 && c / 2 + 2 < 72) ar16[(c / 2) % 2] = CONV_LOAD16(c / 2 + 2);
        __builtin_amdgcn_sched_barrier(0);  // loads stay issued ahead of this step's MFMAs (see conv_mfma)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA32(ac[mt], bfr[tap & 1][j], acc[mt]);
        if (R16 && j % 2 == 0) {  // one 16x16x4 k-step (4 input channels) per two 32x32x2 k-steps (2 each)
            acc16[0] = MFMA(a16, b16[tap & 1][j + 0], acc16[0]);
            acc16[1] = MFMA(a16, b16[tap & 1][j + 1], acc16[1]);
        }
        // an (empty) ordered use of every accumulator: without it the optimizer, to which an MFMA is a pure call,
        // sinks whole tile chains below the k-loop towards the epilogue and parks their operands in scratch
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+a"(acc[mt]));
        if (R16 && j % 2 == 0) { asm volatile("" : "+a"(acc16[0])); asm volatile("" : "+a"(acc16[1])); }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CONV_LOAD
#undef CONV_LOAD16
#undef TAP_BASE
#undef TAP_BASE16
#undef TAP_OFF
}

// Epilogue: ReLU and hand every accumulator element to `sink(row, oc, value)`, row < 2 * P_OUT.
// 32x32 tile register i holds row 8*(i/4) + 4*(lane>>5) + i%4, column lane&31; 16x16 tile register r holds row
// 4*(lane>>4) + r, column lane&15.
template <int P_OUT, typename SINK>
AZ_D void conv_epilogue(int lane, const f32x16 (&acc)[ConvPlan<P_OUT>::MTA], const f32x4 (&acc16)[2], SINK sink) {
    using PL = ConvPlan<P_OUT>;
#pragma unroll
    for (int mt = 0; mt < PL::MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 32 * mt + 8 * (i / 4) + 4 * (lane >> 5) + (i % 4);
            const float v = acc[mt][i];
            if (r < 2 * P_OUT) sink(r, lane & 31, v > 0.0f ? v : 0.0f);
        }
    if (PL::R16) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = 32 * PL::MT + 4 * (lane >> 4) + r4;
                const float v = acc16[nt][r4];
                if (r < 2 * P_OUT) sink(r, nt * 16 + (lane & 15), v > 0.0f ? v : 0.0f);
            }
    }
}

// Persistent: one 8-wave workgroup per CU (two waves per SIMD) and every wave walks board pairs handed out by its
// SIMD's queue; conv1's weights and all biases stay in registers, the next pair's input and the next layer's first
// weight fragments are fetched under the current layer's MFMAs, so a wave pays global-memory latency once, at its
// start.  Measured on MI355X, 16384 boards: the MFMA pipes are busy 87 % of the 325 us (the rest: the younger wave
// of each SIMD runs ~40 % slower than the older one while they share the pipe and finishes its last pair alone;
// s_setprio can swap the roles but not level them).
template <int CH, int CW, int WPB, bool WINO>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, WPB == 4 ? 1 : 2))) void k_trunk2(const float *__restrict__ in, int B, const int *__restrict__ dyn_count, TrunkParams tp, float *__restrict__ feat) {
    using G = TrunkGeom<CH, CW>;
    using PL2 = ConvPlan<G::P1>;
    using PL3 = ConvPlan<G::P3>;
    using PL4 = ConvPlan<G::P4>;
    constexpr int OFF1 = G::WAVE_FLOATS;  // the wave's second board lives right behind the first
    constexpr int MT1 = (2 * G::P1 + 31) / 32;  // conv1 (K = 10) simply runs whole 32-row tiles
    constexpr int NIN = (2 * G::P1 + 63) / 64;
    if (dyn_count) { int c = *dyn_count; B = c < B ? c : B; }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pairs = (B + 1) >> 1;
    // Board pairs are handed out by ticket counters in LDS, one queue per SIMD (its two waves share one MFMA pipe, so
    // every SIMD must get the same work): queue q of block b owns the pairs (4 b + q) + ticket x 4 x blocks.  A
    // device-wide atomic queue was far slower than the imbalance it removed (hot address: 375 us).
    int *ticket_ctr = reinterpret_cast<int *>(smem);
    if (threadIdx.x < 4) ticket_ctr[threadIdx.x] = 0;
    // Winograd conv2 with room in LDS (7x6 planes): the 64 KB of U fragments are copied in once per (persistent) workgroup
    // and every wave reads them from there -- a 16-row tile eats one 256-byte fragment per 32-cycle MFMA, which L2 cannot feed
    constexpr bool ULDS = WINO && (64 + WPB * 2 * G::WAVE_FLOATS * 4 + 65536 <= 160 * 1024);
    float4 *u_lds = reinterpret_cast<float4 *>(smem + 16 + WPB * 2 * G::WAVE_FLOATS);
    if constexpr (ULDS) {
        static_assert((16 + WPB * 2 * G::WAVE_FLOATS) % 4 == 0, "16-byte alignment of the U region");
        const float4 *src = reinterpret_cast<const float4 *>(WPB == 4 ? tp.wu32 : tp.wu);  // one wave per SIMD: the 32-row form
        if constexpr (WPB == 4) {  // conv2_wino32 reads two k-steps at a time: [jp (8)][f (16)][lane (64)] float2, conflict-free 8-byte reads
            float2 *u2 = reinterpret_cast<float2 *>(u_lds);
            for (int i = threadIdx.x; i < 4096; i += 64 * WPB) {
                const float4 v = src[i];
                const int jq = i >> 10, fl = i & 1023;  // source: [jq (4)][f][lane] float4 = k-steps 4 jq .. 4 jq + 3
                u2[(2 * jq) * 1024 + fl] = make_float2(v.x, v.y);
                u2[(2 * jq + 1) * 1024 + fl] = make_float2(v.z, v.w);
            }
        } else {
            for (int i = threadIdx.x; i < 4096; i += 64 * WPB) u_lds[i] = src[i];
        }
    }
    __syncthreads();  // the only workgroup barrier: the waves are independent from here on
    const int simd = (__builtin_amdgcn_s_getreg((4 << 11) | (4 << 6) | 4) & 3);  // HW_REG_HW_ID bits [5:4]
    // a wave draws from its own SIMD's queue and, once that is empty, from the other three: every pair is processed
    // whatever the hardware reports as SIMD id, and a SIMD that ran ahead helps out at the end
    int qoff = 0;
#define NEXT_TICKET(var)                                                                                                   \
    {                                                                                                                      \
        var = pairs;                                                                                                       \
        while (qoff < 4) {                                                                                                 \
            const int q_ = (simd + qoff) & 3;                                                                              \
            int t_ = 0;                                                                                                    \
            if (lane == 0) t_ = atomicAdd(ticket_ctr + q_, 1);                                                             \
            var = 4 * (int)blockIdx.x + q_ + __builtin_amdgcn_readfirstlane(t_) * 4 * (int)gridDim.x;                      \
            if (var < pairs) break;                                                                                        \
            ++qoff;                                                                                                        \
        }                                                                                                                  \
    }
    int pair;
    NEXT_TICKET(pair)
    if (pair >= pairs) return;
    const int m_lane = lane & 31, kk = lane >> 5;
    float xin[NIN];
#define LOAD_INPUT(pr)                                                                            \
    _Pragma("unroll") for (int u = 0; u < NIN; ++u) {                                             \
        const int q = lane + 64 * u;                                                              \
        const size_t idx = (size_t)(pr) * (2 * G::P1) + q;                                        \
        xin[u] = (q < 2 * G::P1 && idx < (size_t)B * G::P1) ? in[idx] : 0.0f;                     \
    }
#define LOAD_W0(l) { _Pragma("unroll") for (int q = 0; q < 4; ++q) { const f32x4 v_ = *reinterpret_cast<const f32x4 *>(tp.wp[l] + (q * 64 + lane) * 4); \
        w0[4 * q] = v_[0]; w0[4 * q + 1] = v_[1]; w0[4 * q + 2] = v_[2]; w0[4 * q + 3] = v_[3]; } }
#define LOAD_W16(l, PLAN) { if (PLAN::R16) { _Pragma("unroll") for (int q = 0; q < 4; ++q) { const f32x4 v_ = *reinterpret_cast<const f32x4 *>(tp.wq[l] + (q * 64 + lane) * 4); \
        w16[4 * q] = v_[0]; w16[4 * q + 1] = v_[1]; w16[4 * q + 2] = v_[2]; w16[4 * q + 3] = v_[3]; } } }
    LOAD_INPUT(pair)
    float w1[5], w0[16], w16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w16[i] = 0.0f;
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) w1[s5] = tp.w1p[s5 * 64 + lane];
    const float bv1 = tp.b1[m_lane], bv2 = tp.cb[0][m_lane], bv3 = tp.cb[1][m_lane], bv4 = tp.cb[2][m_lane];
    float bq2[2] = {0.0f, 0.0f}, bq3[2] = {0.0f, 0.0f}, bq4[2] = {0.0f, 0.0f};  // biases in the 16x16 tiles' column order
    if (PL2::R16) { bq2[0] = tp.cb[0][lane & 15]; bq2[1] = tp.cb[0][16 + (lane & 15)]; }
    if (PL3::R16) { bq3[0] = tp.cb[1][lane & 15]; bq3[1] = tp.cb[1][16 + (lane & 15)]; }
    if (PL4::R16) { bq4[0] = tp.cb[2][lane & 15]; bq4[1] = tp.cb[2][16 + (lane & 15)]; }
    if (!WINO) {
        LOAD_W0(0)
        LOAD_W16(0, PL2)
    }
    float *inp = smem + 16 + wave * 2 * G::WAVE_FLOATS;  // 64 bytes in front hold the queues
    float *act = inp + G::INP;
    for (int i = lane; i < 2 * G::WAVE_FLOATS; i += 64) if (i % G::WAVE_FLOATS < G::INP) inp[i] = 0.0f;
    act[kk * OFF1 + m_lane * G::PS + G::P1] = 0.0f;  // every plane's spare slot: where conv2's out-of-plane taps read
    LDS_FENCE();
#ifdef AZ_PROBE
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tq5 = 0, ph_in = 0, ph_c1 = 0, ph_c2 = 0, ph_c3 = 0, ph_c4 = 0;
    int n_pairs_done = 0;
#endif
    while (true) {
        int nxt;
        NEXT_TICKET(nxt)
        // the per-lane LDS addresses below are loop invariants; hoisted out of this loop they would all stay live
        // (hundreds of registers, spills) -- an opaque copy of the lane id keeps them inside the round
        int ln = lane;
        asm volatile("" : "+v"(ln));
        ln &= 63;  // gives the value range back (folds the tile bounds checks)
        const int b0 = 2 * pair;
        const bool two = b0 + 1 < B;
        STAMP(tq0)
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const int q = ln + 64 * u;
            const int bd = q >= G::P1 ? 1 : 0, p = q - bd * G::P1;
            if (q < 2 * G::P1) inp[bd * OFF1 + (p / CW + 1) * G::PW + (p % CW) + 1] = xin[u];
        }
        if (nxt < pairs) { LOAD_INPUT(nxt) }  // consumed at the top of the next round
        LDS_FENCE();
        auto to_lds = [&](int P_OUT) {
            return [=](int r, int oc, float v) { const int bd = r >= P_OUT ? 1 : 0; act[bd * OFF1 + oc * G::PS + (r - bd * P_OUT)] = v; };
        };
        STAMP(tq1)
        {  // conv1 1->32, pad 1, as a K = 10 product: taps 0..8, tap 9 carries zero weights
            f32x16 acc[MT1];
            int pbase[MT1];
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) {
                int r = 32 * mt + (ln & 31);
                r = r < 2 * G::P1 ? r : 2 * G::P1 - 1;
                const int bd = r >= G::P1 ? 1 : 0, p = r - bd * G::P1;
                pbase[mt] = bd * OFF1 + (p / CW) * G::PW + (p % CW);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = bv1;
            }
#pragma unroll
            for (int s5 = 0; s5 < 5; ++s5) {
                int tap = 2 * s5 + (ln >> 5);
                tap = tap < 9 ? tap : 8;  // finite operand for the zero-weight column
                const int toff = (tap / 3) * G::PW + tap % 3;
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) acc[mt] = MFMA32(inp[pbase[mt] + toff], w1[s5], acc[mt]);
            }
            const auto sink = to_lds(G::P1);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = 32 * mt + 8 * (i / 4) + 4 * (ln >> 5) + (i % 4);
                    const float v = acc[mt][i];
                    if (r < 2 * G::P1) sink(r, ln & 31, v > 0.0f ? v : 0.0f);
                }
        }
        LDS_FENCE();
        STAMP(tq2)
        if constexpr (WINO) {  // conv2 32->32, pad 1, Winograd form: one board after the other (a board's tiles fill a 16-row MFMA tile)
            if constexpr (ULDS && WPB == 4) {
                conv2_wino32<CH, CW, G::PS, OFF1>(act, reinterpret_cast<const float2 *>(u_lds), tp.cb[0], ln);
            } else if constexpr (ULDS) {
                conv2_wino<CH, CW, G::PS, 2>(act, u_lds, tp.cb[0], ln);
                conv2_wino<CH, CW, G::PS, 2>(act + OFF1, u_lds, tp.cb[0], ln);
            } else {
                conv2_wino<CH, CW, G::PS, 3>(act, reinterpret_cast<const float4 *>(tp.wu), tp.cb[0], ln);
                conv2_wino<CH, CW, G::PS, 3>(act + OFF1, reinterpret_cast<const float4 *>(tp.wu), tp.cb[0], ln);
            }
            LOAD_W0(1)
            LOAD_W16(1, PL3)
        } else {  // conv2 32->32, pad 1
            f32x16 acc[PL2::MTA];
            f32x4 acc16[2];
            conv32<G::P1, CW, CH, CW, G::PS, OFF1, 1>(act, tp.wp[0], tp.wq[0], bv2, bq2, w0, w16, ln, acc, acc16);
            LOAD_W0(1)
            LOAD_W16(1, PL3)
            LDS_FENCE();
            conv_epilogue<G::P1>(ln, acc, acc16, to_lds(G::P1));
        }
        LDS_FENCE();
        STAMP(tq3)
        {  // conv3 32->32, valid
            f32x16 acc[PL3::MTA];
            f32x4 acc16[2];
            conv32<G::P3, G::W3, G::H3, CW, G::PS, OFF1, 0>(act, tp.wp[1], tp.wq[1], bv3, bq3, w0, w16, ln, acc, acc16);
            LOAD_W0(2)
            LOAD_W16(2, PL4)
            LDS_FENCE();
            conv_epilogue<G::P3>(ln, acc, acc16, to_lds(G::P3));
        }
        LDS_FENCE();
        STAMP(tq4)
        {  // conv4 32->32, valid -> flattened NCHW features
            f32x16 acc[PL4::MTA];
            f32x4 acc16[2];
            conv32<G::P4, G::W4, G::H4, G::W3, G::PS, OFF1, 0>(act, tp.wp[2], tp.wq[2], bv4, bq4, w0, w16, ln, acc, acc16);
            if (!WINO) {
                LOAD_W0(0)
                LOAD_W16(0, PL2)
            }
            if constexpr (G::P4 % 4 == 0 && !PL4::R16) {
                // registers 4g .. 4g+3 of a tile are four consecutive positions of one board: one 16-byte store each
#pragma unroll
                for (int mt = 0; mt < PL4::MT; ++mt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int r0 = 32 * mt + 8 * g4 + 4 * (ln >> 5);
                        const int bd = r0 >= G::P4 ? 1 : 0;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float x = acc[mt][4 * g4 + e]; v[e] = x > 0.0f ? x : 0.0f; }
                        if (r0 < 2 * G::P4 && (bd == 0 || two))
                            *reinterpret_cast<f32x4 *>(feat + (size_t)(b0 + bd) * (NCH * G::P4) + (ln & 31) * G::P4 + (r0 - bd * G::P4)) = v;
                    }
            } else {
                conv_epilogue<G::P4>(ln, acc, acc16, [=](int r, int oc, float v) {
                    const int bd = r >= G::P4 ? 1 : 0;
                    if (bd == 0 || two) feat[(size_t)(b0 + bd) * (NCH * G::P4) + oc * G::P4 + (r - bd * G::P4)] = v;
                });
            }
        }
#ifdef AZ_PROBE
        STAMP(tq5)
        ph_in += tq1 - tq0; ph_c1 += tq2 - tq1; ph_c2 += tq3 - tq2; ph_c3 += tq4 - tq3; ph_c4 += tq5 - tq4; ++n_pairs_done;
#endif
        pair = nxt;
        if (pair >= pairs) break;
    }
#ifdef AZ_PROBE
    if (lane == 0 && blockIdx.x < 1024) {  // az_probe_buf: 8 words per (block, wave)
        unsigned long long *o = az_probe_buf + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = ph_in; o[1] = ph_c1; o[2] = ph_c2; o[3] = ph_c3; o[4] = ph_c4; o[5] = (unsigned long long)n_pairs_done; o[6] = tq5; o[7] = 0;
    }
#endif
#undef NEXT_TICKET
#undef LOAD_INPUT
#undef LOAD_W0
#undef LOAD_W16
}

// C[M][N] = act(A[M][K] * Bw[K][N] + bias[N]);  K % 32 == 0, N % BN == 0.
// f32 MFMA 32x32x2 (sustains ~150 TFLOP/s from a single in-place accumulator chain on this chip; the
// 16x16x4 shape cycling over many accumulators measured 10-25 % lower: tools/micro/mfma_peak2.hip).
// LDS double buffer, BK = 32, one barrier per K tile; the global loads of tile t+2 are issued while
// tile t is computed (two register staging sets); fragment reads run one k-step ahead of the MFMAs.

template <int BM, int BN, int WM, int WN, bool RELU, int KT>
__global__ __launch_bounds__(256, 2) void k_gemm(const float *__restrict__ A, const float *__restrict__ Bw,
                                              const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                              const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (b and b+8 share an L2), so the
    // linear id is remapped to give every XCD a contiguous run of row bands.  Speed only.
    const int nbx = N / BN, nb = nbx * (int)gridDim.y;
    int bid = (int)blockIdx.y * nbx + (int)blockIdx.x;
    if (nb % 8 == 0) bid = (bid % 8) * (nb / 8) + bid / 8;
    const int tile_y = bid / nbx, tile_x = bid % nbx;
    if (tile_y * BM >= M) return;  // whole block beyond the rows filled this step (uniform exit)
    constexpr int BK = 32, WAVES_N = BN / WN, TM = WM / 32, TN = WN / 32;
    static_assert((BM / WM) * WAVES_N == 4 && TM >= 1 && TN >= 1, "4 waves per block, 32x32 MFMA tiles");
    constexpr int ASTR = BK + 1;  // (m + k) mod 32 : conflict-free A-fragment reads (32 rows, one k)
    constexpr int BSTR = BN;      // a B-fragment read is 32 consecutive floats of one k row
    constexpr int A_LD = BM * BK / 4 / 256, B_LD = (BK * BN / 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float *As = gsm;                    // [2][BM][ASTR]
    float *Bs = gsm + 2 * BM * ASTR;    // [2][BK][BSTR]   (2*BM*ASTR is a multiple of 4 floats: 16-byte aligned)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int bm0 = tile_y * BM, bn0 = tile_x * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        float bv = bias[bn0 + wn0 + tn * 32 + (lane & 31)];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = bv;
    }
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = sa0 = sa1 = sa2 = sa3 = sb0 = sb1 = sb2 = sb3 = make_float4(0, 0, 0, 0);
#define LD_A(i, r, k0)                                                                    \
    if constexpr ((i) < A_LD) {                                                           \
        int idx = tid + (i) * 256, row = idx / (BK / 4), q = idx % (BK / 4);              \
        int gr = bm0 + row;                                                               \
        gr = gr < M ? gr : M - 1;                                                         \
        r = *reinterpret_cast<const float4 *>(A + (size_t)gr * K + (k0) + 4 * q);         \
    }
#define LD_B(i, r, k0)                                                                    \
    if constexpr ((i) < B_LD) {                                                           \
        int idx = tid + (i) * 256;                                                        \
        if (BK * BN / 4 >= 256 * ((i) + 1) || idx < BK * BN / 4) {                        \
            int kr = idx / (BN / 4), c4 = idx % (BN / 4);                                 \
            r = *reinterpret_cast<const float4 *>(Bw + (size_t)((k0) + kr) * N + bn0 + 4 * c4); \
        }                                                                                 \
    }
#define ST_A(i, r, as)                                                                    \
    if constexpr ((i) < A_LD) {                                                           \
        int idx = tid + (i) * 256, row = idx / (BK / 4), q = idx % (BK / 4);              \
        float *d = (as) + row * ASTR + 4 * q;                                             \
        d[0] = r.x; d[1] = r.y; d[2] = r.z; d[3] = r.w;                                   \
    }
#define ST_B(i, r, bs)                                                                    \
    if constexpr ((i) < B_LD) {                                                           \
        int idx = tid + (i) * 256;                                                        \
        if (BK * BN / 4 >= 256 * ((i) + 1) || idx < BK * BN / 4) {                        \
            int kr = idx / (BN / 4), c4 = idx % (BN / 4);                                 \
            *reinterpret_cast<float4 *>((bs) + kr * BSTR + 4 * c4) = r;                   \
        }                                                                                 \
    }
#define GEMM_ISSUE_R(k0) { LD_A(0, ra0, k0) LD_A(1, ra1, k0) LD_A(2, ra2, k0) LD_A(3, ra3, k0) LD_B(0, rb0, k0) LD_B(1, rb1, k0) LD_B(2, rb2, k0) LD_B(3, rb3, k0) }
#define GEMM_ISSUE_S(k0) { LD_A(0, sa0, k0) LD_A(1, sa1, k0) LD_A(2, sa2, k0) LD_A(3, sa3, k0) LD_B(0, sb0, k0) LD_B(1, sb1, k0) LD_B(2, sb2, k0) LD_B(3, sb3, k0) }
#define GEMM_STORE_R(buf)                                                                 \
    {                                                                                     \
        float *as_ = As + (buf) * BM * ASTR, *bs_ = Bs + (buf) * BK * BSTR;               \
        ST_A(0, ra0, as_) ST_A(1, ra1, as_) ST_A(2, ra2, as_) ST_A(3, ra3, as_)           \
        ST_B(0, rb0, bs_) ST_B(1, rb1, bs_) ST_B(2, rb2, bs_) ST_B(3, rb3, bs_)           \
    }
#define GEMM_STORE_S(buf)                                                                 \
    {                                                                                     \
        float *as_ = As + (buf) * BM * ASTR, *bs_ = Bs + (buf) * BK * BSTR;               \
        ST_A(0, sa0, as_) ST_A(1, sa1, as_) ST_A(2, sa2, as_) ST_A(3, sa3, as_)           \
        ST_B(0, sb0, bs_) ST_B(1, sb1, bs_) ST_B(2, sb2, bs_) ST_B(3, sb3, bs_)           \
    }
#define FRD 4  /* fragment ring: LDS reads run FRD-1 k-steps ahead of the MFMAs (a 32x32x2 k-step is only 64-128 MFMA cycles) */
#define GEMM_COMPUTE(buf)                                                                 \
    {                                                                                     \
        const float *as = As + (buf) * BM * ASTR + (wm0 + (lane & 31)) * ASTR + (lane >> 5);   \
        const float *bs = Bs + (buf) * BK * BSTR + (lane >> 5) * BSTR + wn0 + (lane & 31);     \
        float afr[FRD][TM], bfr[FRD][TN];                                                 \
        _Pragma("unroll") for (int p = 0; p < FRD - 1; ++p) {                             \
            _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) afr[p][tm] = as[tm * 32 * ASTR + p * 2];          \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) bfr[p][tn] = bs[p * 2 * BSTR + tn * 32];          \
        }                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < BK / 2; ++ks) {                           \
            if (ks + FRD - 1 < BK / 2) {                                                  \
                _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) afr[(ks + FRD - 1) % FRD][tm] = as[tm * 32 * ASTR + (ks + FRD - 1) * 2];   \
                _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) bfr[(ks + FRD - 1) % FRD][tn] = bs[(ks + FRD - 1) * 2 * BSTR + tn * 32];   \
            }                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                            \
            _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                             \
                _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = MFMA32(afr[ks % FRD][tm], bfr[ks % FRD][tn], acc[tm][tn]); \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
    }
    static_assert(A_LD <= 4 && B_LD <= 4, "staging register budget");
    // KT > 0: the tile loop is fully unrolled.  hipcc's s_waitcnt insertion is exact only in straight-line code;
    // around a loop back-edge it drains every outstanding global load at the top of each tile, which
    // serialises the prefetch with the MFMAs (global latency here is about one tile of MFMA work).
    const int T = KT > 0 ? KT : K / BK;
    GEMM_ISSUE_R(0)
    if (T > 1) GEMM_ISSUE_S(BK)
    GEMM_STORE_R(0)
    __syncthreads();
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, ph_issue = 0, ph_comp = 0, ph_store = 0, ph_bar = 0, p_begin = 0;
    (void)p0; (void)p1; (void)p2; (void)p3; (void)p4; (void)ph_issue; (void)ph_comp; (void)ph_store; (void)ph_bar; (void)p_begin;
    STAMP(p_begin)
#ifdef AZ_PROBE
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
#define GEMM_PAIR(t)                                                                      \
    {                                                                                     \
        STAMP(p0)                                                                         \
        if ((t) + 2 < T) GEMM_ISSUE_R(((t) + 2) * BK)                                     \
        STAMP(p1)                                                                         \
        GEMM_COMPUTE(0)                                                                   \
        STAMP(p2)                                                                         \
        if ((t) + 1 < T) GEMM_STORE_S(1)                                                  \
        STAMP(p3)                                                                         \
        __syncthreads();                                                                  \
        STAMP(p4)                                                                         \
        ph_issue += p1 - p0; ph_comp += p2 - p1; ph_store += p3 - p2; ph_bar += p4 - p3;  \
        if ((t) + 1 < T) {                                                                \
            STAMP(p0)                                                                     \
            if ((t) + 3 < T) GEMM_ISSUE_S(((t) + 3) * BK)                                 \
            STAMP(p1)                                                                     \
            GEMM_COMPUTE(1)                                                               \
            STAMP(p2)                                                                     \
            if ((t) + 2 < T) GEMM_STORE_R(0)                                              \
            STAMP(p3)                                                                     \
            __syncthreads();                                                              \
            STAMP(p4)                                                                     \
            ph_issue += p1 - p0; ph_comp += p2 - p1; ph_store += p3 - p2; ph_bar += p4 - p3;  \
        }                                                                                 \
    }
    if constexpr (KT > 0) {
#pragma unroll
        for (int t = 0; t < KT; t += 2) GEMM_PAIR(t)
    } else {
        for (int t = 0; t < T; t += 2) GEMM_PAIR(t)
    }
#undef GEMM_PAIR
#undef GEMM_ISSUE_R
#undef GEMM_ISSUE_S
#undef GEMM_STORE_R
#undef GEMM_STORE_S
#undef GEMM_COMPUTE
#undef FRD
#undef LD_A
#undef LD_B
#undef ST_A
#undef ST_B
#ifdef AZ_PROBE
    if (tid == 0 && bid < 8192) {
        unsigned long long p_end = 0;
        STAMP(p_end)
        unsigned long long *o = az_probe_buf + (size_t)bid * 8;
        o[0] = ph_issue; o[1] = ph_comp; o[2] = ph_store; o[3] = ph_bar; o[4] = p_end - p_begin; o[5] = p_begin; o[6] = p_end;
        o[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;  // 100 MHz ticks
    }
#endif
    // C/D layout of 32x32x2: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = bm0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int col = bn0 + wn0 + tn * 32 + (lane & 31);
                float v = acc[tm][tn][r];
                if (RELU) v = v > 0.0f ? v : 0.0f;
                if (row < M) C[(size_t)row * N + col] = v;
            }
}

// ---------------------------------------------------------------------------------------------
// k_gemm_solo: the same product for LARGE row counts (from one 256x256 tile per CU up) with ONE wave per SIMD: every wave
// owns a 128x128 tile (16 accumulator tiles of 32x32 = 256 registers of its 512), the workgroup a 256x256 tile.  Against
// k_gemm's 64x64 per wave this halves the LDS fragment reads and the barriers per MFMA (16 MFMAs per 8 fragment reads, one
// barrier per 256 MFMAs per wave) -- the two-waves-per-SIMD kernel spends 46 % of each wave's time outside its MFMA chains
// (in-kernel stamps) and leaves the pipe idle whenever both partners are there (MFMA busy 76-80 %).  A K tile of a wave is
// 16 k-steps x 16 MFMAs = 16 K cycles of matrix work, several times the global-memory latency: one register set and a runtime
// K loop suffice.  The next tile's loads and LDS stores ride INSIDE the k-steps (see SOLO_COMPUTE): as a block between two tiles
// they cost 11 % of the loop (18.5 K cycles per tile against 17.3 K; tools/probe_gemm_solo.py).  Measured at 32768 rows: fc1 259,
// fc2 248 us against k_gemm's 280 / 268; MFMA busy 0.82 / 0.86; HBM traffic 1.16x / 1.08x the algorithmic bytes (k_gemm 2.09x / 1.54x).
// Accumulation: bias, then k ascending on v_mfma_f32_32x32x2_f32 -- the chain of k_gemm / k_dense_small / the oracle.
// ---------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_gemm_solo(const float *__restrict__ A, const float *__restrict__ Bw,
                                                                                           const float *__restrict__ bias, float *__restrict__ C, int M, int N,
                                                                                           int K, const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    constexpr int BM = 256, BN = 256, BK = 32, ASTR = BK + 1, BSTR = BN;
    const int nbx = N / BN, nb = nbx * (int)gridDim.y;
    int bid = (int)blockIdx.y * nbx + (int)blockIdx.x;
    if (nb % 8 == 0) bid = (bid % 8) * (nb / 8) + bid / 8;  // XCD-aware tile order, as k_gemm
    const int tile_y = bid / nbx, tile_x = bid % nbx;
    // The grid is sized for the launch's capacity; the rows really present (dyn_count) may be far fewer -- the last plies of a
    // self-play wave evaluate a few hundred leaves.  A 256-row tile would then leave most CUs idle while a few grind through whole
    // tiles, so the tile HEIGHT follows the rows: 64 or 128 rows per workgroup (one or two 32-row tiles per wave instead of four)
    // as soon as the grid has enough workgroups for that many row tiles.  Columns, staging and the K loop are unchanged.
    int tme = 4;
    if ((M + 63) / 64 <= (int)gridDim.y) tme = 1;
    else if ((M + 127) / 128 <= (int)gridDim.y) tme = 2;
    const int BMe = 64 * tme;
    if (tile_y * BMe >= M) return;
#ifdef AZ_PROBE
    unsigned long long pk0, pk1, pl0, pl1;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    STAMP(pk0)
#endif
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float *As = gsm;                  // [2][BM][ASTR]
    float *Bs = gsm + 2 * BM * ASTR;  // [2][BK][BSTR]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave >> 1) * 32 * tme, wn0 = (wave & 1) * 128;
    const int bm0 = tile_y * BMe, bn0 = tile_x * BN;
    // accumulators start at the bias THROUGH the matrix pipe (1 * bias[col] + 0 * 0 on a zero accumulator: exact): a plain splat of
    // four bias values over 256 registers makes the compiler hold a second copy of them and spill (measured: 100 dwords)
    f32x16 acc[4][4];
    {
        const float one = lane < 32 ? 1.0f : 0.0f;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const float bv = lane < 32 ? bias[bn0 + wn0 + tn * 32 + lane] : 0.0f;
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                acc[tm][tn] = MFMA32(one, bv, z);
            }
        }
    }
    // staging slots: A float4 #i of this thread is (row (tid >> 3) + 32 i, q = tid & 7), BK / 4 = 8 per row; B float4 #i is
    // (k row (tid >> 6) + 4 i, c4 = tid & 63), BN / 4 = 64 per row.  Named registers, not arrays: arrays behind lambdas stayed in scratch.
    // Addresses are a wave-uniform base (A + k0, B + (k0 + 4 i) N: scalar registers) plus a 32-bit per-lane byte offset fixed for the
    // whole kernel -- no vector address arithmetic per tile, so a load can sit between two MFMAs without holding the next one up.
    const int arow = tid >> 3, aq = tid & 7;
    unsigned aoff0, aoff1, aoff2, aoff3, aoff4, aoff5, aoff6, aoff7;
#define SOLO_AOFF(i)                                                       \
    {                                                                      \
        int gr = bm0 + arow + 32 * i;                                      \
        gr = gr < M ? gr : M - 1;                                          \
        aoff##i = (unsigned)(((size_t)gr * K + 4 * aq) * sizeof(float));   \
    }
    SOLO_AOFF(0) SOLO_AOFF(1) SOLO_AOFF(2) SOLO_AOFF(3) SOLO_AOFF(4) SOLO_AOFF(5) SOLO_AOFF(6) SOLO_AOFF(7)
    const unsigned boff = (unsigned)(((size_t)(tid >> 6) * N + 4 * (tid & 63)) * sizeof(float));
    const float *Bt = Bw + bn0;
    float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
#define SOLO_LDA(i, k0) ra##i = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(A + (k0)) + aoff##i);
#define SOLO_LDB(i, k0) rb##i = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Bt + (size_t)((k0) + 4 * i) * N) + boff);
#define SOLO_LD(i, k0) { SOLO_LDA(i, k0) SOLO_LDB(i, k0) }
#define SOLO_ISSUE(k0) SOLO_LD(0, k0) SOLO_LD(1, k0) SOLO_LD(2, k0) SOLO_LD(3, k0) SOLO_LD(4, k0) SOLO_LD(5, k0) SOLO_LD(6, k0) SOLO_LD(7, k0)
#define SOLO_STA(i, as_)                       \
    {                                          \
        float *d = (as_) + 32 * i * ASTR;      \
        d[0] = ra##i.x;                        \
        d[1] = ra##i.y;                        \
        d[2] = ra##i.z;                        \
        d[3] = ra##i.w;                        \
    }
#define SOLO_STB(i, bs_) *reinterpret_cast<float4 *>((bs_) + 4 * i * BSTR) = rb##i;
#define SOLO_ST(i) { SOLO_STA(i, as_) SOLO_STB(i, bs_) }
#define SOLO_STASH(buf)                                                                                                      \
    {                                                                                                                        \
        float *as_ = As + (buf) * BM * ASTR + arow * ASTR + 4 * aq, *bs_ = Bs + (buf) * BK * BSTR + (tid >> 6) * BSTR + 4 * (tid & 63); \
        SOLO_ST(0) SOLO_ST(1) SOLO_ST(2) SOLO_ST(3) SOLO_ST(4) SOLO_ST(5) SOLO_ST(6) SOLO_ST(7)                              \
    }
    // one K tile of the wave: 16 k-steps of 16 MFMAs; the fragments of step ks + 1 are read under the MFMAs of step ks.  With NEXT
    // the following tile is staged under the same MFMAs: its global loads go out two per k-step over steps 0-7, the registers go to the
    // other LDS buffer one slot per k-step over steps 8-15.  One wave per SIMD means nothing else covers an instruction that is not
    // in the shadow of an MFMA (64 cycles each), so the memory instructions are dealt out by hand: at most two LDS reads and one
    // staging instruction in front of every group of four MFMAs, fenced so that the scheduler does not gather them again.
    // scheduling fence that instruction selection honours too: a plain sched_barrier left the LDS reads free to gather ahead of it
#define SOLO_FENCE() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define SOLO_COMPUTE(buf, NEXT, k0n, TME)                                                                            \
    {                                                                                                            \
        const float *as = As + (buf) * BM * ASTR + (wm0 + (lane & 31)) * ASTR + (lane >> 5);                     \
        const float *bs = Bs + (buf) * BK * BSTR + (lane >> 5) * BSTR + wn0 + (lane & 31);                       \
        float *as_ = As + ((buf) ^ 1) * BM * ASTR + arow * ASTR + 4 * aq, *bs_ = Bs + ((buf) ^ 1) * BK * BSTR + (tid >> 6) * BSTR + 4 * (tid & 63); \
        float afr[2][4], bfr[2][4];                                                                              \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) { if (x < TME) afr[0][x] = as[x * 32 * ASTR]; bfr[0][x] = bs[x * 32]; } \
        _Pragma("unroll") for (int ks = 0; ks < BK / 2; ++ks) {                                                  \
            _Pragma("unroll") for (int tm = 0; tm < 4; ++tm) {                                                   \
                /* quarter tm of the k-step: two fragment reads of step ks + 1 and one staging instruction, then 4 MFMAs */ \
                if (ks + 1 < BK / 2) {                                                                           \
                    if (tm < TME) afr[(ks + 1) & 1][tm] = as[tm * 32 * ASTR + (ks + 1) * 2];                     \
                    bfr[(ks + 1) & 1][tm] = bs[(ks + 1) * 2 * BSTR + tm * 32];                                   \
                }                                                                                                \
                if (NEXT && tm == 0) {                                                                           \
                    if (ks == 0) SOLO_LDA(0, k0n)                                                                \
                    if (ks == 1) SOLO_LDA(1, k0n)                                                                \
                    if (ks == 2) SOLO_LDA(2, k0n)                                                                \
                    if (ks == 3) SOLO_LDA(3, k0n)                                                                \
                    if (ks == 4) SOLO_LDA(4, k0n)                                                                \
                    if (ks == 5) SOLO_LDA(5, k0n)                                                                \
                    if (ks == 6) SOLO_LDA(6, k0n)                                                                \
                    if (ks == 7) SOLO_LDA(7, k0n)                                                                \
                    if (ks == 8) SOLO_STA(0, as_)                                                                \
                    if (ks == 9) SOLO_STA(1, as_)                                                                \
                    if (ks == 10) SOLO_STA(2, as_)                                                               \
                    if (ks == 11) SOLO_STA(3, as_)                                                               \
                    if (ks == 12) SOLO_STA(4, as_)                                                               \
                    if (ks == 13) SOLO_STA(5, as_)                                                               \
                    if (ks == 14) SOLO_STA(6, as_)                                                               \
                    if (ks == 15) SOLO_STA(7, as_)                                                               \
                }                                                                                                \
                if (NEXT && tm == 2) {                                                                           \
                    if (ks == 0) SOLO_LDB(0, k0n)                                                                \
                    if (ks == 1) SOLO_LDB(1, k0n)                                                                \
                    if (ks == 2) SOLO_LDB(2, k0n)                                                                \
                    if (ks == 3) SOLO_LDB(3, k0n)                                                                \
                    if (ks == 4) SOLO_LDB(4, k0n)                                                                \
                    if (ks == 5) SOLO_LDB(5, k0n)                                                                \
                    if (ks == 6) SOLO_LDB(6, k0n)                                                                \
                    if (ks == 7) SOLO_LDB(7, k0n)                                                                \
                    if (ks == 8) SOLO_STB(0, bs_)                                                                \
                    if (ks == 9) SOLO_STB(1, bs_)                                                                \
                    if (ks == 10) SOLO_STB(2, bs_)                                                               \
                    if (ks == 11) SOLO_STB(3, bs_)                                                               \
                    if (ks == 12) SOLO_STB(4, bs_)                                                               \
                    if (ks == 13) SOLO_STB(5, bs_)                                                               \
                    if (ks == 14) SOLO_STB(6, bs_)                                                               \
                    if (ks == 15) SOLO_STB(7, bs_)                                                               \
                }                                                                                                \
                SOLO_FENCE()                                                                                     \
                if (tm < TME) {                                                                                  \
                    _Pragma("unroll") for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = MFMA32(afr[ks & 1][tm], bfr[ks & 1][tn], acc[tm][tn]); \
                }                                                                                                \
                SOLO_FENCE()                                                                                     \
            }                                                                                                    \
        }                                                                                                        \
    }
    const int T = K / BK;
    SOLO_ISSUE(0)
    SOLO_STASH(0)
    __syncthreads();
    STAMP(pl0)
    // C/D layout of 32x32x2: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Inside each copy of the loop:
    // a common epilogue behind the three copies makes the compiler shuffle the 256 accumulators at the join.
#define SOLO_EPILOGUE(TME)                                                                                       \
    _Pragma("unroll") for (int tm = 0; tm < TME; ++tm) _Pragma("unroll") for (int tn = 0; tn < 4; ++tn)          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                         \
            const int row = bm0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);                      \
            const int col = bn0 + wn0 + tn * 32 + (lane & 31);                                                   \
            float v = acc[tm][tn][r];                                                                            \
            if (RELU) v = v > 0.0f ? v : 0.0f;                                                                   \
            if (row < M) C[(size_t)row * N + col] = v;                                                           \
        }
    // the last tile is peeled: no conditional staging inside the loop; one copy of the loop per tile height (compile-time TME)
#define SOLO_MAIN(TME)                                       \
    {                                                        \
        for (int t = 0; t + 1 < T; ++t) {                    \
            SOLO_COMPUTE(t & 1, true, (t + 1) * BK, TME)     \
            __syncthreads();                                 \
        }                                                    \
        STAMP(pl1)                                           \
        SOLO_COMPUTE((T - 1) & 1, false, 0, TME)             \
        SOLO_EPILOGUE(TME)                                   \
    }
    if (tme == 4) SOLO_MAIN(4)
    else if (tme == 2) SOLO_MAIN(2)
    else SOLO_MAIN(1)
#undef SOLO_MAIN
#undef SOLO_EPILOGUE
#ifdef AZ_PROBE
    STAMP(pk1)
    if (tid == 0) {  // shader-clock cycles of the whole block and of its main loop, the 100 MHz real-time clock over the same span
        unsigned long long *pb = az_probe_buf + (size_t)((int)blockIdx.y * nbx + (int)blockIdx.x) % 8192 * 8;
        pb[0] = pk1 - pk0; pb[1] = pl1 - pl0; pb[2] = __builtin_amdgcn_s_memrealtime() - rt0; pb[3] = (unsigned long long)(T - 1);
    }
#endif
}


// k_gemm_solo's scheme for row counts that do not fill the chip with 256x256 tiles (fc1 at 8192 rows, fc2 at 16384): waves of
// 32 TM x 32 TN (2 x 2 waves per workgroup, tile 64 TM x 64 TN), the same hand-placed k-steps -- in front of the TN MFMAs of "quarter"
// tm one A and TN / TM B fragment reads of the next k-step and ONE staging instruction: the next K tile's NA + NB global loads from
// k-step 0 on, its LDS stores from k-step 8 on.  Instantiated for (TM, TN) = (2, 2): 128x128 tiles, two workgroups per CU (66 KB of
// LDS each), i.e. two waves per SIMD.  Same accumulation chain.
template <bool RELU, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm_solo_t(const float *__restrict__ A, const float *__restrict__ Bw, const float *__restrict__ bias,
                                                     float *__restrict__ C, int M, int N, int K, const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, ASTR = BK + 1, BSTR = BN;
    constexpr int NA = 2 * TM, NB = 2 * TN, NL = NA + NB, BQ = BN / 4, BROWS = 256 / BQ, BPQ = TN / TM;
    static_assert(TN % TM == 0 && NL <= 8 * TM && BQ <= 256 && 256 % BQ == 0, "staging plan: loads in k-steps 0-7, stores in 8-15");
    const int nbx = N / BN, nb = nbx * (int)gridDim.y;
    int bid = (int)blockIdx.y * nbx + (int)blockIdx.x;
    if (nb % 8 == 0) bid = (bid % 8) * (nb / 8) + bid / 8;  // XCD-aware tile order, as k_gemm
    const int tile_y = bid / nbx, tile_x = bid % nbx;
    if (tile_y * BM >= M) return;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float *As = gsm;                  // [2][BM][ASTR]
    float *Bs = gsm + 2 * BM * ASTR;  // [2][BK][BSTR]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
    const int bm0 = tile_y * BM, bn0 = tile_x * BN;
    f32x16 acc[TM][TN];
    {   // bias through the matrix pipe, as k_gemm_solo
        const float one = lane < 32 ? 1.0f : 0.0f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const float bv = lane < 32 ? bias[bn0 + wn0 + tn * 32 + lane] : 0.0f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                acc[tm][tn] = MFMA32(one, bv, z);
            }
        }
    }
    const int arow = tid >> 3, aq = tid & 7, brow = tid / BQ, bc4 = tid % BQ;
    static_assert(NA <= 4, "A staging slots");
    unsigned aoff0 = 0, aoff1 = 0, aoff2 = 0, aoff3 = 0;
#define T_SETOFF(i, var)                                                   \
    if ((i) < NA) {                                                        \
        int gr = bm0 + arow + 32 * (i);                                    \
        gr = gr < M ? gr : M - 1;                                          \
        var = (unsigned)(((size_t)gr * K + 4 * aq) * sizeof(float));       \
    }
    T_SETOFF(0, aoff0) T_SETOFF(1, aoff1) T_SETOFF(2, aoff2) T_SETOFF(3, aoff3)
#undef T_SETOFF
    const unsigned boff = (unsigned)(((size_t)brow * N + 4 * bc4) * sizeof(float));
    const float *Bt = Bw + bn0;
    // staging registers by NAME (slot p < NA: A float4 #p, else B float4 #(p - NA)): register arrays indexed through the slot
    // arithmetic were left in scratch for some (TM, TN)
    float4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7, sr8, sr9, sr10, sr11;
    static_assert(NL <= 12, "staging registers");
#define T_SR(p) ((p) == 0 ? sr0 : (p) == 1 ? sr1 : (p) == 2 ? sr2 : (p) == 3 ? sr3 : (p) == 4 ? sr4 : (p) == 5 ? sr5 : (p) == 6 ? sr6 : (p) == 7 ? sr7 : (p) == 8 ? sr8 : (p) == 9 ? sr9 : (p) == 10 ? sr10 : sr11)
#define T_AOFF(p) ((p) == 0 ? aoff0 : (p) == 1 ? aoff1 : (p) == 2 ? aoff2 : aoff3)
#define T_LD(p, k0)                                                                                                                          \
    {                                                                                                                                        \
        if ((p) < NA) T_SR(p) = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(A + (k0)) + T_AOFF(p));                     \
        else T_SR(p) = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Bt + (size_t)((k0) + BROWS * ((p) - NA)) * N) + boff); \
    }
#define T_ST(p, as_, bs_)                                                                        \
    {                                                                                            \
        const float4 v = T_SR(p);                                                                \
        if ((p) < NA) {                                                                          \
            float *d = (as_) + 32 * (p) * ASTR;                                                  \
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;                                      \
        } else {                                                                                 \
            *reinterpret_cast<float4 *>((bs_) + BROWS * ((p) - NA) * BSTR) = v;                  \
        }                                                                                        \
    }
#define T_FENCE() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define T_COMPUTE(buf, NEXT, k0n)                                                                                \
    {                                                                                                            \
        const float *as = As + (buf) * BM * ASTR + (wm0 + (lane & 31)) * ASTR + (lane >> 5);                     \
        const float *bs = Bs + (buf) * BK * BSTR + (lane >> 5) * BSTR + wn0 + (lane & 31);                       \
        float *as_ = As + ((buf) ^ 1) * BM * ASTR + arow * ASTR + 4 * aq, *bs_ = Bs + ((buf) ^ 1) * BK * BSTR + brow * BSTR + 4 * bc4; \
        float afr[2][TM], bfr[2][TN];                                                                            \
        _Pragma("unroll") for (int x = 0; x < TM; ++x) afr[0][x] = as[x * 32 * ASTR];                            \
        _Pragma("unroll") for (int x = 0; x < TN; ++x) bfr[0][x] = bs[x * 32];                                   \
        _Pragma("unroll") for (int ks = 0; ks < BK / 2; ++ks) {                                                  \
            _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                                  \
                if (ks + 1 < BK / 2) {                                                                           \
                    afr[(ks + 1) & 1][tm] = as[tm * 32 * ASTR + (ks + 1) * 2];                                   \
                    _Pragma("unroll") for (int j = 0; j < BPQ; ++j)                                              \
                        bfr[(ks + 1) & 1][tm * BPQ + j] = bs[(ks + 1) * 2 * BSTR + (tm * BPQ + j) * 32];         \
                }                                                                                                \
                if (NEXT) {                                                                                      \
                    const int pl = ks * TM + tm, ps = (ks - 8) * TM + tm;                                        \
                    if (pl < NL) T_LD(pl, k0n)                                                                   \
                    if (ks >= 8 && ps < NL) T_ST(ps, as_, bs_)                                                   \
                }                                                                                                \
                T_FENCE()                                                                                        \
                _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = MFMA32(afr[ks & 1][tm], bfr[ks & 1][tn], acc[tm][tn]); \
                T_FENCE()                                                                                        \
            }                                                                                                    \
        }                                                                                                        \
    }
    const int T = K / BK;
#pragma unroll
    for (int p = 0; p < NL; ++p) T_LD(p, 0)
    {
        float *as_ = As + arow * ASTR + 4 * aq, *bs_ = Bs + brow * BSTR + 4 * bc4;
#pragma unroll
        for (int p = 0; p < NL; ++p) T_ST(p, as_, bs_)
    }
    __syncthreads();
    for (int t = 0; t + 1 < T; ++t) {  // the last tile is peeled: no conditional staging inside the loop
        T_COMPUTE(t & 1, true, (t + 1) * BK)
        __syncthreads();
    }
    T_COMPUTE((T - 1) & 1, false, 0)
#undef T_LD
#undef T_ST
#undef T_SR
#undef T_AOFF
#undef T_FENCE
#undef T_COMPUTE
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = bn0 + wn0 + tn * 32 + (lane & 31);
                float v = acc[tm][tn][r];
                if (RELU) v = v > 0.0f ? v : 0.0f;
                if (row < M) C[(size_t)row * N + col] = v;
            }
}

// Dense layer for a handful of rows (single-game search, small arenas): one wavefront per (row, 64 output columns),
// plain fmaf in k order -- the same chain the MFMA tiles compute, so the results are bit-identical -- with 16 to 32 weight
// loads in flight per lane.  The tiled GEMM walks its K tiles serially inside one workgroup (~1 us per tile whatever the
// row count: 21 / 40 us for fc1 / fc2); this kernel needs 9 / 16 us.
template <bool RELU>
__global__ __launch_bounds__(64) void k_dense_small(const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ bias,
                                                    float *__restrict__ Y, int M, int N, int K, const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    const int m = blockIdx.y, lane = threadIdx.x;
    if (m >= M) return;
    const float *x = X + (size_t)m * K;  // wave-uniform: the row's values arrive through scalar loads
    const int n = blockIdx.x * 64 + lane;
    const int nc = n < N ? n : N - 1;
    const float *w = Wt + nc;
    float acc = bias[nc];
    float wa[16], wb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wa[i] = w[(size_t)i * N];
    for (int k0 = 0; k0 < K; k0 += 32) {  // K % 32 == 0; two 16-deep weight buffers in ping-pong (no register moves)
#pragma unroll
        for (int i = 0; i < 16; ++i) wb[i] = w[(size_t)(k0 + 16 + i) * N];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fmaf(x[k0 + i], wa[i], acc);
        if (k0 + 32 < K) {
#pragma unroll
            for (int i = 0; i < 16; ++i) wa[i] = w[(size_t)(k0 + 32 + i) * N];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fmaf(x[k0 + 16 + i], wb[i], acc);
    }
    if (RELU) acc = acc > 0.0f ? acc : 0.0f;
    if (n < N) Y[(size_t)m * N + n] = acc;
}

// ---------------------------------------------------------------------------------------------
// k_dense_frag: fc1 / fc2 for everything below the row counts that fill the chip with k_gemm's tiles (a single game's search, arenas,
// self-play waves of a few hundred to a few thousand games).  There the layer's time is the length of ONE accumulation chain, not
// the chip's FLOPs: k_gemm walks K in 32x32x2 steps (k += 2 per 64 cycles: 15 us for K = 1024 whatever the row count, 20 / 37.5 us
// measured for fc1 / fc2 from 256 to 2048 rows), k_dense_small in dependent fmas (19 us).  v_mfma_f32_16x16x4_f32 advances k by 4
// every 32 cycles -- a chain four times shorter for the same K -- and a 16x16 tile per wave gives 2048 waves already at 512 rows.
// The scheme is k_heads2's: workgroup = NT waves, wave t owns column tile blockIdx.x * NT + t of the block's 16 rows; the rows go to
// LDS once; the weights never touch LDS: a copy in B-fragment order [K/16][N/16][64 lanes][4 k-steps] (k_retile_q, made at commit
// time from the folded [K][N] matrix: a permutation, same values) is streamed from L2 with one coalesced 16-byte load per 16 k,
// PF loads ahead.  Accumulation: bias, then k ascending -- the chain of k_gemm / k_dense_small / the oracle: identical bits.
// ---------------------------------------------------------------------------------------------
template <bool RELU, int NT>
__global__ __launch_bounds__(64 * NT) void k_dense_frag(const float *__restrict__ X, const float *__restrict__ Wq, const float *__restrict__ bias,
                                                        float *__restrict__ Y, int M, int N, int K, const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    constexpr int NTHR = 64 * NT, PF = 8;
    const int brow0 = blockIdx.y * 16;
    if (brow0 >= M) return;  // uniform
    extern __shared__ __attribute__((aligned(16))) float dfs[];  // [16][K + 4]: row stride = 4 mod 64 banks, the fragment reads are conflict-free
    const int XSTR = K + 4, NKB = K / 16, NTILES = N / 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int ct = blockIdx.x * NT + wave;
    const float4 *wq = reinterpret_cast<const float4 *>(Wq) + (size_t)ct * 64 + lane;
    const size_t wstep = (size_t)NTILES * 64;
    float4 bq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) bq[p] = wq[(size_t)p * wstep];  // in flight while the rows are staged
    const float bv = bias[ct * 16 + m];
    {   // the block's rows -> LDS; K % 64 == 0.  Rows past M are not staged at all (a single game's search fills ONE of the 16): what the
        // MFMAs then read for them is whatever the LDS holds -- every output row depends on its own A row only, and those rows are never stored
        const int kv = K / 4, rows = (M - brow0) < 16 ? (M - brow0) : 16, nv = rows * kv;
        for (int q0 = 0; q0 < nv; q0 += 4 * NTHR) {
            float4 rx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = q0 + tid + NTHR * i;
                rx[i] = make_float4(0, 0, 0, 0);
                if (q < nv) rx[i] = reinterpret_cast<const float4 *>(X + (size_t)(brow0 + q / kv) * K)[q % kv];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = q0 + tid + NTHR * i;
                if (q < nv) *reinterpret_cast<float4 *>(dfs + (q / kv) * XSTR + 4 * (q % kv)) = rx[i];
            }
        }
    }
    __syncthreads();
    f32x4 acc = (f32x4){bv, bv, bv, bv};
    const float *xa = dfs + m * XSTR + kq;  // A fragment of k-step ks: xa[4 * ks]
    for (int kb0 = 0; kb0 < NKB; kb0 += PF) {  // NKB % PF == 0
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kb = kb0 + p;
            const float4 b = bq[p];
            if (kb + PF < NKB) bq[p] = wq[(size_t)(kb + PF) * wstep];
            const float a0 = xa[16 * kb], a1 = xa[16 * kb + 4], a2 = xa[16 * kb + 8], a3 = xa[16 * kb + 12];
            __builtin_amdgcn_sched_barrier(0);  // loads stay issued ahead of this block's MFMAs
            acc = MFMA(a0, b.x, acc);
            acc = MFMA(a1, b.y, acc);
            acc = MFMA(a2, b.z, acc);
            acc = MFMA(a3, b.w, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C layout of 16x16x4: column lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = brow0 + 4 * kq + r;
        float v = acc[r];
        if (RELU) v = v > 0.0f ? v : 0.0f;
        if (row < M) Y[(size_t)row * N + ct * 16 + m] = v;
    }
}

// dst[kb][nt][lane][i] = W[k = 16 kb + 4 i + (lane >> 4)][n = 16 nt + (lane & 15)] of a folded [K][N] matrix (k_dense_frag's weights)
__global__ void k_retile_q(const float *__restrict__ W, float *__restrict__ dst, int K, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * N) return;
    const int ii = i % 4, lane = (i / 4) % 64, NTL = N / 16, nt = (i / 256) % NTL, kb = i / (256 * NTL);
    dst[i] = W[(size_t)(16 * kb + 4 * ii + (lane >> 4)) * N + 16 * nt + (lane & 15)];
}

template <int BM, int BN>
constexpr int gemm_lds_bytes() { return 4 * (2 * BM * (32 + 1) + 2 * 32 * BN); }

// policy + value heads (othello.py:379-382, base.py:355): logits = h2 * Wh + bh with
// Wh = [fc_probs | fc_value | 0-pad] of width NH = 16*NT; then softmax over the first A columns
// and tanh of column A.  64 rows per block (4 waves x 16 rows).
// 32 rows per block; wave w owns row tile (w >> 1) and the column tiles [0, NS) or [NS, NT) (NS = ceil(NT/2)).
// K is walked in chunks of 128: the next chunk's X rows and Wh rows are fetched into registers (coalesced
// 16-byte loads) while the current chunk's MFMAs run from LDS, so L2 latency is paid once per chunk;
// inside a chunk the LDS fragment reads run one k-step ahead of the MFMAs.
template <int NT>
__global__ __launch_bounds__(256) void k_heads(const float *__restrict__ X, const float *__restrict__ Wh,
                                               const float *__restrict__ bh, int M, int K, int A,
                                               float *__restrict__ probs, float *__restrict__ value, const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    if ((int)blockIdx.x * 32 >= M) return;
    constexpr int NH = NT * 16, KC = 128, XSTR = KC + 2, WSTR = NH, NS = (NT + 1) / 2;  // NH = 16 (mod 32): conflict-free
    constexpr int X_LD = 32 * KC / 4 / 256, W_LD = (KC * NH / 4 + 255) / 256;            // float4 per thread per chunk
    static_assert(X_LD == 4 && W_LD <= 10, "staging registers");
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float *Xs = hsm;                       // [32][XSTR]
    float *Ws = Xs + 32 * XSTR;            // [KC][WSTR]
    float *Ls = Ws + KC * WSTR;            // [32][NH + 1]
    float *rsum = Ls + 32 * (NH + 1);      // [32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int brow0 = blockIdx.x * 32;
    const int mt = wave >> 1, nt0 = (wave & 1) ? NS : 0, ntn = (wave & 1) ? NT - NS : NS;
    f32x4 acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        float bv = i < ntn ? bh[(nt0 + i) * 16 + (lane & 15)] : 0.0f;
        acc[i] = (f32x4){bv, bv, bv, bv};
    }
    float4 rx[X_LD], rw[W_LD];
#pragma unroll
    for (int i = 0; i < X_LD; ++i) rx[i] = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < W_LD; ++i) rw[i] = make_float4(0, 0, 0, 0);
    // per-thread copy slots, computed once: X float4 #i is (row, q) with KC/4 = 32 per row; the Wh chunk
    // [kc, kc+KC) x NH is one contiguous run of KC*NH floats both in HBM and in Ws (no index math)
    const float *xsrc[X_LD];
    int xq[X_LD], xdst[X_LD];
#pragma unroll
    for (int i = 0; i < X_LD; ++i) {
        int idx = tid + i * 256, row = idx >> 5, q = idx & 31;
        int gr = brow0 + row;
        gr = gr < M ? gr : M - 1;
        xsrc[i] = X + (size_t)gr * K + 4 * q;
        xq[i] = 4 * q;
        xdst[i] = row * XSTR + 4 * q;
    }
    auto issue = [&](int kc) {
#pragma unroll
        for (int i = 0; i < X_LD; ++i)
            rx[i] = kc + xq[i] < K ? *reinterpret_cast<const float4 *>(xsrc[i] + kc) : make_float4(0, 0, 0, 0);
        const float *wsrc = Wh + (size_t)kc * NH;
        const int wlim = (K - kc) * NH;  // floats of Wh left from row kc on
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            int f = 4 * (tid + i * 256);
            if (f < KC * NH) rw[i] = f < wlim ? *reinterpret_cast<const float4 *>(wsrc + f) : make_float4(0, 0, 0, 0);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            float *d = Xs + xdst[i];
            *reinterpret_cast<float2 *>(d) = make_float2(rx[i].x, rx[i].y);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(rx[i].z, rx[i].w);
        }
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            int f = 4 * (tid + i * 256);
            if (f < KC * NH) *reinterpret_cast<float4 *>(Ws + f) = rw[i];
        }
    };
    issue(0);
    int boff[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) boff[i] = i < ntn ? i * 16 : 0;
    const float *xa = Xs + (mt * 16 + (lane & 15)) * XSTR + (lane >> 4);
    const float *wb = Ws + (lane >> 4) * WSTR + nt0 * 16 + (lane & 15);
    for (int kc = 0; kc < K; kc += KC) {
        stash();
        __syncthreads();
        if (kc + KC < K) issue(kc + KC);
        float a_n = xa[0], b_n[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) b_n[i] = wb[boff[i]];
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            const float a = a_n;
            float b[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) b[i] = b_n[i];
            if (ks + 1 < KC / 4) {
                a_n = xa[(ks + 1) * 4];
#pragma unroll
                for (int i = 0; i < NS; ++i) b_n[i] = wb[(ks + 1) * 4 * WSTR + boff[i]];
            }
            __builtin_amdgcn_sched_barrier(0);
            // uniform code for both wave kinds: a wave with fewer column tiles recomputes tile 0 into a
            // spare accumulator (a predicated MFMA makes hipcc bounce accumulators through VGPRs)
#pragma unroll
            for (int i = 0; i < NS; ++i) acc[i] = MFMA(a, b[i], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NS; ++i)
        if (i < ntn)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ls[(mt * 16 + (lane >> 4) * 4 + r) * (NH + 1) + (nt0 + i) * 16 + (lane & 15)] = acc[i][r];
    __syncthreads();
    // softmax (base.py:355 exp(log_softmax)): max and exp in parallel, the row sum sequentially in
    // ascending action order (one lane per row) so that the result is reproducible bit for bit
    const int row_l = tid >> 3, sub = tid & 7;  // 8 lanes per row
    float *lrow = Ls + row_l * (NH + 1);
    float m = -__builtin_inff();
    for (int a = sub; a < A; a += 8) m = fmaxf(m, lrow[a]);
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 8));
    const float vlogit = lrow[A];
    for (int a = sub; a < A; a += 8) lrow[a] = az_det_expf(lrow[a] - m);
    __syncthreads();
    if (tid < 32) {
        const float *l = Ls + tid * (NH + 1);
        float s = 0.0f;
        int a = 0;
        for (; a + 8 <= A; a += 8) {
            float t0 = l[a], t1 = l[a + 1], t2 = l[a + 2], t3 = l[a + 3], t4 = l[a + 4], t5 = l[a + 5], t6 = l[a + 6], t7 = l[a + 7];
            s += t0; s += t1; s += t2; s += t3; s += t4; s += t5; s += t6; s += t7;
        }
        for (; a < A; ++a) s += l[a];
        rsum[tid] = s;
    }
    __syncthreads();
    const int row = brow0 + row_l;
    if (row < M) {
        const float s = rsum[row_l];
        for (int a = sub; a < A; a += 8) probs[(size_t)row * A + a] = lrow[a] / s;
        if (sub == 0) value[row] = az_det_tanhf(vlogit);
    }
}

// ---------------------------------------------------------------------------------------------
// k_heads2: the policy + value heads for batches from a few hundred rows up (replaces k_heads, which re-staged the whole
// head matrix through LDS for every 32 rows: 0.32 of the MFMA peak at 32768 rows, 18 us at 4096).
//   * workgroup = NT wavefronts, wave t owns column tile t (16 logits) of the block's RB = 16 RH rows: NT x M/RB waves fill
//     the chip already at 4096 rows (RH = 1: 1280 waves) -- a 16-row tile with all its columns in ONE wave would leave
//     three quarters of the SIMDs idle there;
//   * the block's rows of h2 are copied into LDS ONCE (every 16-byte load of the block in flight together, one barrier);
//   * the head matrix never touches LDS: it is stored in MFMA B-fragment order [K/16][NT][64 lanes][4 k-steps], a wave
//     streams its tile's fragments from L2 with one coalesced 16-byte load per 16 k, several loads ahead of their use;
//   * softmax as everywhere: exact maximum, az_det_expf, the row sum in ascending action order, p = e / S; tanh value.
// Accumulation: bias, then k ascending on v_mfma_f32_16x16x4_f32 -- the same chain as k_heads / k_heads_small / the oracle.
// ---------------------------------------------------------------------------------------------
template <int NT, int RH, int K>
constexpr int heads2_lds_bytes() { return 4 * (16 * RH * (K + 2) + 16 * RH * (NT * 16 + 1) + 16 * RH); }

template <int NT, int RH, int K>
__global__ __launch_bounds__(64 * NT) void k_heads2(const float *__restrict__ X, const float *__restrict__ Wq, const float *__restrict__ bh,
                                                    int M, int A, float *__restrict__ probs, float *__restrict__ value,
                                                    const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    constexpr int NH = NT * 16, RB = 16 * RH, XSTR = K + 2, NTHR = 64 * NT, NKB = K / 16, PF = 4;  // PF: fragment loads in flight (8: 45.9 vs 39 us at 32768 rows; 16 at <= 1024 rows: no faster)
    static_assert(K % 16 == 0 && NKB >= PF && RB <= NTHR && NTHR % 8 == 0, "tile plan");
    const int brow0 = blockIdx.x * RB;
    if (brow0 >= M) return;  // uniform
    extern __shared__ __attribute__((aligned(16))) float h2s[];
    float *Xs = h2s;                      // [RB][XSTR]
    float *Ls = Xs + RB * XSTR;           // [RB][NH + 1]
    float *rsum = Ls + RB * (NH + 1);     // [RB]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    // this wave's fragments: Wq[kb][wave][lane] (float4 = k-steps 4 kb .. 4 kb + 3); the first PF are requested before
    // the rows are staged, so their L2 latency overlaps the staging
    const float4 *wq = reinterpret_cast<const float4 *>(Wq) + (size_t)wave * 64 + lane;
    float4 bq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) bq[p] = wq[(size_t)p * NT * 64];
    const float bv = bh[wave * 16 + m];
    {   // the block's rows of X -> LDS.  Rows past M are not staged (their logits come out of whatever the LDS holds and are never stored:
        // an output row depends on its own A row only)
        constexpr int NV = RB * K / 4, C = (NV + NTHR - 1) / NTHR;
        const int nv = ((M - brow0) < RB ? (M - brow0) : RB) * (K / 4);
        float4 rx[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int q = tid + NTHR * i;
            rx[i] = make_float4(0, 0, 0, 0);
            if (q < nv) rx[i] = reinterpret_cast<const float4 *>(X + (size_t)(brow0 + q / (K / 4)) * K)[q % (K / 4)];
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int q = tid + NTHR * i;
            if (q < nv) {
                float *d = Xs + (q / (K / 4)) * XSTR + 4 * (q % (K / 4));  // 8-byte aligned (XSTR is even)
                *reinterpret_cast<float2 *>(d) = make_float2(rx[i].x, rx[i].y);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(rx[i].z, rx[i].w);
            }
        }
    }
    __syncthreads();
    f32x4 acc[RH];
#pragma unroll
    for (int h = 0; h < RH; ++h) acc[h] = (f32x4){bv, bv, bv, bv};
    const float *xa = Xs + m * XSTR + kq;  // A fragment of k-step ks, row half h: xa[h * 16 * XSTR + 4 * ks]
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const float4 b = bq[kb % PF];
        if (kb + PF < NKB) bq[kb % PF] = wq[(size_t)(kb + PF) * NT * 64];
        float a[4][RH];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < RH; ++h) a[i][h] = xa[h * 16 * XSTR + 16 * kb + 4 * i];
        __builtin_amdgcn_sched_barrier(0);  // loads stay issued ahead of this block's MFMAs
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            acc[h] = MFMA(a[0][h], b.x, acc[h]);
            acc[h] = MFMA(a[1][h], b.y, acc[h]);
            acc[h] = MFMA(a[2][h], b.z, acc[h]);
            acc[h] = MFMA(a[3][h], b.w, acc[h]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // C layout of 16x16x4: column lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ls[(16 * h + 4 * kq + r) * (NH + 1) + wave * 16 + m] = acc[h][r];
    __syncthreads();
    // softmax (base.py:355 exp(log_softmax)), 8 lanes per row: max and exp in parallel, the row sum by one lane in
    // ascending action order
    const int sub = tid & 7;
    for (int row_l = tid >> 3; row_l < RB; row_l += NTHR / 8) {
        float *lrow = Ls + row_l * (NH + 1);
        float mx = -__builtin_inff();
        for (int a = sub; a < A; a += 8) mx = fmaxf(mx, lrow[a]);
#pragma unroll
        for (int o = 4; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
        if (sub == 0 && brow0 + row_l < M) value[brow0 + row_l] = az_det_tanhf(lrow[A]);
        for (int a = sub; a < A; a += 8) lrow[a] = az_det_expf(lrow[a] - mx);
    }
    __syncthreads();
    if (tid < RB) {
        const float *l = Ls + tid * (NH + 1);
        float s = 0.0f;
        int a = 0;
        for (; a + 8 <= A; a += 8) {
            float t0 = l[a], t1 = l[a + 1], t2 = l[a + 2], t3 = l[a + 3], t4 = l[a + 4], t5 = l[a + 5], t6 = l[a + 6], t7 = l[a + 7];
            s += t0; s += t1; s += t2; s += t3; s += t4; s += t5; s += t6; s += t7;
        }
        for (; a < A; ++a) s += l[a];
        rsum[tid] = s;
    }
    __syncthreads();
    // the block's RB x A probabilities are one contiguous run of the output: coalesced stores
    const int n_rows = (M - brow0) < RB ? (M - brow0) : RB;
    for (int idx = tid; idx < n_rows * A; idx += NTHR) {
        const int r = idx / A, a = idx - r * A;
        probs[(size_t)brow0 * A + idx] = Ls[r * (NH + 1) + a] / rsum[r];
    }
}

// Heads for a handful of rows: one workgroup per row, one lane per logit (fmaf in k order, as k_dense_small), then the
// same softmax / tanh arithmetic as k_heads (exact max, az_det_expf, sum in ascending action order).  ~7 us against 15.
__global__ __launch_bounds__(128) void k_heads_small(const float *__restrict__ X, const float *__restrict__ Wh, const float *__restrict__ bh,
                                                     int M, int K, int A, int NH, float *__restrict__ probs, float *__restrict__ value,
                                                     const int *__restrict__ dyn_count) {
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    const int m = blockIdx.x, tid = threadIdx.x;
    if (m >= M) return;
    __shared__ float lg[128];
    __shared__ float red[2];
    const float *x = X + (size_t)m * K;  // wave-uniform
    const int n = tid < NH ? tid : NH - 1;
    const float *w = Wh + n;
    float acc = bh[n];
    float wa[16], wb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wa[i] = w[(size_t)i * NH];
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wb[i] = w[(size_t)(k0 + 16 + i) * NH];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fmaf(x[k0 + i], wa[i], acc);
        if (k0 + 32 < K) {
#pragma unroll
            for (int i = 0; i < 16; ++i) wa[i] = w[(size_t)(k0 + 32 + i) * NH];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fmaf(x[k0 + 16 + i], wb[i], acc);
    }
    lg[tid] = tid < A ? acc : -__builtin_inff();
    if (tid == A) red[1] = acc;  // value logit
    __syncthreads();
    if (tid < 64) {  // the maximum is exact whatever the order
        float mx = fmaxf(lg[tid], lg[tid + 64]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (tid == 0) red[0] = mx;
    }
    __syncthreads();
    const float mx = red[0];
    const float e = tid < A ? az_det_expf(acc - mx) : 0.0f;
    lg[tid] = e;
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int a = 0; a < A; ++a) s += lg[a];  // ascending action order
        red[0] = s;
        value[m] = az_det_tanhf(red[1]);
    }
    __syncthreads();
    if (tid < A) probs[(size_t)m * A + tid] = e / red[0];
}

// ---------------------------------------------------------------------------------------------
// k_tail_small: fc1 + fc2 + heads of the SMALL conv nets (Connect4Net: 192 -> 64 -> 32 -> 7 + 1, connect4.py:382-412) in
// one launch.  At these widths the three dense stages are 29 KFLOP per board against 1.28 MFLOP of convolutions: as three
// kernels they cost three dependent launches of 4-6 us each (latency, not work).  One workgroup takes 4 x R rows through
// all three layers:
//   * ONE memory round trip: the three weight matrices (58 KB), the biases and the rows' features are copied into LDS
//     with every 16-byte load of the block in flight at once;
//   * fc1   lane n of a wave owns output n (F1 = 64) for the wave's R rows: acc[r] = fmaf(x[r][k], W1[k][n], acc[r]) for k
//           ascending -- x[r][k..k+3] one LDS broadcast read, W1[k][.] one conflict-free 256-byte LDS read shared by the rows;
//   * fc2   the k-th operand h1[r][k] sits in lane k's register: v_readlane broadcasts it (k is a compile-time constant);
//   * heads the same, lanes 0 .. NH-1; softmax over the A policy logits with the exact maximum, az_det_expf and the sum in
//           ascending action order (as k_heads / the oracle), tanh of logit A.
// Every output is the k-ascending fmaf chain the MFMA tiles compute: bit-identical to the three-kernel path.
// ---------------------------------------------------------------------------------------------
AZ_D float readlane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

template <int FIN, int F1, int F2, int NH, int R>
constexpr int tail_lds_floats() { return FIN * F1 + F1 * F2 + F2 * NH + F1 + F2 + NH + 4 * R * FIN; }

template <int FIN, int F1, int F2, int NH, int A, int R>
__global__ __launch_bounds__(256) void k_tail_small(const float *__restrict__ feat, const float *__restrict__ W1, const float *__restrict__ b1,
                                                    const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ Wh,
                                                    const float *__restrict__ bh, int M, float *__restrict__ probs, float *__restrict__ value,
                                                    const int *__restrict__ dyn_count) {
    static_assert(F1 == 64 && F2 <= 64 && NH <= 64 && A < NH && FIN % 4 == 0 && (FIN * F1) % 4 == 0 && (F1 * F2) % 4 == 0 && (F2 * NH) % 4 == 0,
                  "one lane per fc1 output; 16-byte copies");
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    constexpr int ROWS = 4 * R;
    const int brow0 = blockIdx.x * ROWS;
    if (brow0 >= M) return;  // uniform for the workgroup
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float *w1s = tsm, *w2s = w1s + FIN * F1, *whs = w2s + F1 * F2, *b1s = whs + F2 * NH, *b2s = b1s + F1, *bhs = b2s + F2, *xs = bhs + NH;
    const int tid = threadIdx.x;
    {   // every load first, then every LDS store: one round trip for the whole block
        constexpr int NW1 = FIN * F1 / 4, NW2 = F1 * F2 / 4, NWH = F2 * NH / 4, NX = ROWS * FIN / 4;
        constexpr int C1 = (NW1 + 255) / 256, C2 = (NW2 + 255) / 256, CH_ = (NWH + 255) / 256, CX = (NX + 255) / 256;
        float4 r1[C1], r2[C2], rh[CH_], rx[CX];
#define IN_(N, q) (((N) % 256 == 0) || (q) < (N))  /* compile-time true where the copy is a whole number of rounds */
#pragma unroll
        for (int i = 0; i < C1; ++i) r1[i] = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < C2; ++i) r2[i] = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CH_; ++i) rh[i] = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CX; ++i) rx[i] = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < C1; ++i) { const int q = tid + 256 * i; if (IN_(NW1, q)) r1[i] = reinterpret_cast<const float4 *>(W1)[q]; }
#pragma unroll
        for (int i = 0; i < C2; ++i) { const int q = tid + 256 * i; if (IN_(NW2, q)) r2[i] = reinterpret_cast<const float4 *>(W2)[q]; }
#pragma unroll
        for (int i = 0; i < CH_; ++i) { const int q = tid + 256 * i; if (IN_(NWH, q)) rh[i] = reinterpret_cast<const float4 *>(Wh)[q]; }
#pragma unroll
        for (int i = 0; i < CX; ++i) {
            const int q = tid + 256 * i;
            if (IN_(NX, q)) {
                int row = brow0 + q / (FIN / 4);
                row = row < M ? row : M - 1;
                rx[i] = reinterpret_cast<const float4 *>(feat + (size_t)row * FIN)[q % (FIN / 4)];
            }
        }
        float bb = 0.0f;
        if (tid < F1) bb = b1[tid]; else if (tid < F1 + F2) bb = b2[tid - F1]; else if (tid < F1 + F2 + NH) bb = bh[tid - F1 - F2];
#pragma unroll
        for (int i = 0; i < C1; ++i) { const int q = tid + 256 * i; if (IN_(NW1, q)) reinterpret_cast<float4 *>(w1s)[q] = r1[i]; }
#pragma unroll
        for (int i = 0; i < C2; ++i) { const int q = tid + 256 * i; if (IN_(NW2, q)) reinterpret_cast<float4 *>(w2s)[q] = r2[i]; }
#pragma unroll
        for (int i = 0; i < CH_; ++i) { const int q = tid + 256 * i; if (IN_(NWH, q)) reinterpret_cast<float4 *>(whs)[q] = rh[i]; }
#pragma unroll
        for (int i = 0; i < CX; ++i) { const int q = tid + 256 * i; if (IN_(NX, q)) reinterpret_cast<float4 *>(xs)[q] = rx[i]; }
        if (tid < F1 + F2 + NH) b1s[tid] = bb;  // b1s, b2s, bhs are contiguous
#undef IN_
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int row0 = brow0 + wave * R;
    const float *xw = xs + wave * R * FIN;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = b1s[lane];
#pragma unroll 4
    for (int k = 0; k < FIN; k += 4) {
        float w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = w1s[(k + i) * F1 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 x = *reinterpret_cast<const float4 *>(xw + r * FIN + k);  // same address in every lane: broadcast
            acc[r] = fmaf(x.x, w[0], acc[r]); acc[r] = fmaf(x.y, w[1], acc[r]);
            acc[r] = fmaf(x.z, w[2], acc[r]); acc[r] = fmaf(x.w, w[3], acc[r]);
        }
    }
    float h1[R], a2[R];
    {
        const int n2 = lane < F2 ? lane : F2 - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) { h1[r] = acc[r] > 0.0f ? acc[r] : 0.0f; a2[r] = b2s[n2]; }
#pragma unroll
        for (int k = 0; k < F1; ++k) {
            const float w = w2s[k * F2 + n2];
#pragma unroll
            for (int r = 0; r < R; ++r) a2[r] = fmaf(readlane_f(h1[r], k), w, a2[r]);
        }
    }
    float h2[R], lg[R];
    {
        const int j = lane < NH ? lane : NH - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) { h2[r] = a2[r] > 0.0f ? a2[r] : 0.0f; lg[r] = bhs[j]; }
#pragma unroll
        for (int k = 0; k < F2; ++k) {
            const float w = whs[k * NH + j];
#pragma unroll
            for (int r = 0; r < R; ++r) lg[r] = fmaf(readlane_f(h2[r], k), w, lg[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float m = readlane_f(lg[r], 0);
#pragma unroll
        for (int a = 1; a < A; ++a) m = fmaxf(m, readlane_f(lg[r], a));
        const float e = lane < A ? az_det_expf(lg[r] - m) : 0.0f;
        float s = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) s += readlane_f(e, a);  // ascending action order
        const float vl = readlane_f(lg[r], A);
        if (row0 + r < M) {
            if (lane < A) probs[(size_t)(row0 + r) * A + lane] = e / s;
            if (lane == 0) value[row0 + r] = az_det_tanhf(vl);
        }
    }
}

// k_tail_mfma: the same three layers on v_mfma_f32_16x16x4_f32.  k_tail_small walks every row's 192 + 64 + 32 dependent fmas and first
// copies all three weight matrices (58 KB) into LDS for its 16 rows; here a workgroup's 16 rows are ONE MFMA row tile, fc1's four column
// tiles go to the four waves (a chain of 48 MFMAs each), fc2's two to waves 0 and 1 (16 MFMAs), the heads' one to wave 0 (8), the
// activations pass through LDS between the layers and the weights come straight from L2 in B-fragment layout (lane (n, kq) reads
// W[4 ks + kq][n0 + n]: 64-byte runs of the row-major matrix), all requested before the first barrier.  Same chains (bias, then k
// ascending), same softmax arithmetic (exact maximum, az_det_expf, the sum in ascending action order, e / S; az_det_tanhf): identical bits.
template <int FIN, int F1, int F2, int NH, int A>
__global__ __launch_bounds__(256) void k_tail_mfma(const float *__restrict__ feat, const float *__restrict__ W1, const float *__restrict__ b1,
                                                   const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ Wh,
                                                   const float *__restrict__ bh, int M, float *__restrict__ probs, float *__restrict__ value,
                                                   const int *__restrict__ dyn_count) {
    static_assert(F1 == 64 && F2 == 32 && NH == 16 && A < NH && FIN % 16 == 0, "four / two / one column tiles for the four waves");
    if (dyn_count) { int c = *dyn_count; M = c < M ? c : M; }
    const int brow0 = blockIdx.x * 16;
    if (brow0 >= M) return;  // uniform for the workgroup
    constexpr int XS = FIN + 4, H1S = F1 + 4, H2S = F2 + 4, LS = NH + 1;  // row strides = 4 mod 64 banks (fragment reads conflict-free)
    __shared__ __attribute__((aligned(16))) float xs[16 * XS];
    __shared__ __attribute__((aligned(16))) float h1s[16 * H1S];
    __shared__ __attribute__((aligned(16))) float h2s[16 * H2S];
    __shared__ float lgs[16 * LS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 15, kq = lane >> 4;
    // the block's rows: 16 x FIN / 4 float4
    constexpr int NX = 16 * FIN / 4, CX = (NX + 255) / 256;
    float4 rx[CX];
#pragma unroll
    for (int i = 0; i < CX; ++i) {
        const int q = tid + 256 * i;
        rx[i] = make_float4(0, 0, 0, 0);
        if (q < NX) {
            int row = brow0 + q / (FIN / 4);
            row = row < M ? row : M - 1;
            rx[i] = reinterpret_cast<const float4 *>(feat + (size_t)row * FIN)[q % (FIN / 4)];
        }
    }
    // this wave's weight fragments of all three layers, requested now (their latency runs under the row staging)
    float w1f[FIN / 4], w2f[F1 / 4], whf[F2 / 4];
#pragma unroll
    for (int ks = 0; ks < FIN / 4; ++ks) w1f[ks] = W1[(size_t)(4 * ks + kq) * F1 + 16 * wave + m];
    if (wave < 2) {
#pragma unroll
        for (int ks = 0; ks < F1 / 4; ++ks) w2f[ks] = W2[(size_t)(4 * ks + kq) * F2 + 16 * wave + m];
    }
    if (wave == 0) {
#pragma unroll
        for (int ks = 0; ks < F2 / 4; ++ks) whf[ks] = Wh[(size_t)(4 * ks + kq) * NH + m];
    }
    const float bv1 = b1[16 * wave + m], bv2 = wave < 2 ? b2[16 * wave + m] : 0.0f, bvh = bh[m];
#pragma unroll
    for (int i = 0; i < CX; ++i) {
        const int q = tid + 256 * i;
        if (q < NX) *reinterpret_cast<float4 *>(xs + (q / (FIN / 4)) * XS + 4 * (q % (FIN / 4))) = rx[i];
    }
    __syncthreads();
    {   // fc1: column tile `wave`
        f32x4 acc = (f32x4){bv1, bv1, bv1, bv1};
        const float *xa = xs + m * XS + kq;
#pragma unroll
        for (int ks = 0; ks < FIN / 4; ++ks) acc = MFMA(xa[4 * ks], w1f[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) h1s[(4 * kq + r) * H1S + 16 * wave + m] = acc[r] > 0.0f ? acc[r] : 0.0f;  // C layout: row 4 kq + r, column m
    }
    __syncthreads();
    if (wave < 2) {  // fc2
        f32x4 acc = (f32x4){bv2, bv2, bv2, bv2};
        const float *ha = h1s + m * H1S + kq;
#pragma unroll
        for (int ks = 0; ks < F1 / 4; ++ks) acc = MFMA(ha[4 * ks], w2f[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) h2s[(4 * kq + r) * H2S + 16 * wave + m] = acc[r] > 0.0f ? acc[r] : 0.0f;
    }
    __syncthreads();
    if (wave == 0) {  // heads: logits | value | padding
        f32x4 acc = (f32x4){bvh, bvh, bvh, bvh};
        const float *ha = h2s + m * H2S + kq;
#pragma unroll
        for (int ks = 0; ks < F2 / 4; ++ks) acc = MFMA(ha[4 * ks], whf[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) lgs[(4 * kq + r) * LS + m] = acc[r];
    }
    __syncthreads();
    if (tid < 16) {  // one lane per row: the arithmetic of k_tail_small / k_heads, action for action
        const int row = brow0 + tid;
        const float *l = lgs + tid * LS;
        float mx = l[0];
#pragma unroll
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, l[a]);
        float e[A];
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) { e[a] = az_det_expf(l[a] - mx); sum += e[a]; }
        if (row < M) {
#pragma unroll
            for (int a = 0; a < A; ++a) probs[(size_t)row * A + a] = e[a] / sum;
            value[row] = az_det_tanhf(l[A]);
        }
    }
}

template <int NT>
constexpr int heads_lds_bytes() { return 4 * (32 * (128 + 2) + 128 * NT * 16 + 32 * (NT * 16 + 1) + 32); }

struct MlpParams { float f1w[81], f1b[9], f2w[81], f2b[9], hw[90], hb[10]; };

// The parameters live in device memory behind a pointer that never changes (as the conv nets' weights do): a captured
// HIP graph of the search replays this launch with the weights az_net_commit uploaded last, not the ones at capture.
__global__ void k_mlp(const float *__restrict__ in, int B, const int *__restrict__ dyn_count, const MlpParams *__restrict__ pp, float *__restrict__ probs, float *__restrict__ value) {
    const MlpParams &p = *pp;
    if (dyn_count) { int c = *dyn_count; B = c < B ? c : B; }
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float x[9], h1[9], h2[9], lg[10];
    for (int i = 0; i < 9; ++i) x[i] = in[(size_t)b * 9 + i];
    for (int j = 0; j < 9; ++j) { float a = p.f1b[j]; for (int k = 0; k < 9; ++k) a = fmaf(x[k], p.f1w[k * 9 + j], a); h1[j] = a > 0.0f ? a : 0.0f; }
    for (int j = 0; j < 9; ++j) { float a = p.f2b[j]; for (int k = 0; k < 9; ++k) a = fmaf(h1[k], p.f2w[k * 9 + j], a); h2[j] = a > 0.0f ? a : 0.0f; }
    for (int j = 0; j < 10; ++j) { float a = p.hb[j]; for (int k = 0; k < 9; ++k) a = fmaf(h2[k], p.hw[k * 10 + j], a); lg[j] = a; }
    float m = lg[0];
    for (int a = 1; a < 9; ++a) m = lg[a] > m ? lg[a] : m;
    float s = 0.0f;
    for (int a = 0; a < 9; ++a) { lg[a] = az_det_expf(lg[a] - m); s += lg[a]; }
    for (int a = 0; a < 9; ++a) probs[(size_t)b * 9 + a] = lg[a] / s;
    value[b] = az_det_tanhf(lg[9]);
}

// ---------------------------------------------------------------------------------------------
// Exact block-fixed-point dense layers on the int8 matrix pipe (AZ_DENSE_I8=1; OthelloNet's fc1 / fc2; the CPU oracle restates the same
// arithmetic under the same switch, function dense_layer_q).  v_mfma_i32_32x32x32_i8 runs at 32x the rate of the f32-input MFMA
// and its integer sums are exact, so a float32 layer can be had from it without giving up bit equality:
//   * a vector of K floats (a row of activations; the K weights of one output column) shares ONE exponent E = the biased f32 exponent of
//     its largest magnitude (clamped to [1, 254]); q[k] = rint(x[k] * 2^(148 - E)), |q| <= 2^22, is held as three balanced base-256
//     digits d0 + 256 d1 + 65536 d2 (each in [-128, 127]; non-finite elements quantise to 0);
//   * the nine digit-plane products of a (row, column) pair are int8 MFMAs accumulated exactly in int32, products of equal weight
//     256^(i + j) in one accumulator (five per output; |sum| < 2^26 at K = 1024);
//   * X = sum_s acc_s 256^s is the exact integer dot product of the two quantised vectors -- whatever the tile shape, the K order or the
//     kernel variant -- and the output is relu((float)ldexp((double)X, E_row + E_col - 296) + bias): X -> double and double -> float
//     rounded to nearest even once each.
// Against float64 the form is as close as the float32 fma chain it replaces (OthelloNet 8x8: pi 9e-9, v 1.4e-7; chain: 1.0e-8, 2.0e-7).
// ---------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

AZ_D int q_value_dev(float x, int E) {
    const unsigned b = __float_as_uint(x);
    if ((b & 0x7f800000u) == 0x7f800000u) return 0;
    return (int)rintf(ldexpf(x, 148 - E));
}
AZ_D void q_digits(int q, int &d0, int &d1, int &d2) {
    d0 = ((q + 128) & 255) - 128;
    const int q1 = (q - d0) >> 8;
    d1 = ((q1 + 128) & 255) - 128;
    d2 = (q1 - d1) >> 8;
}

// rows of X [M][K] (float32) -> digit planes D[p][row][K] (int8, p = 0: least significant) + the row exponents.  One wave per row.
__global__ __launch_bounds__(256) void k_q_rows(const float *__restrict__ X, int M, int K, const int *__restrict__ dyn, int8_t *__restrict__ D,
                                                size_t plane, int *__restrict__ E) {
    if (dyn) { int c = *dyn; M = c < M ? c : M; }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;  // wave-uniform
    const float4 *xr = reinterpret_cast<const float4 *>(X + (size_t)row * K);
    float4 v[4];
    unsigned mx = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = c * 64 + lane;
        v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * i < K) v[c] = xr[i];
        mx = max(mx, max(max(__float_as_uint(v[c].x) & 0x7fffffffu, __float_as_uint(v[c].y) & 0x7fffffffu),
                         max(__float_as_uint(v[c].z) & 0x7fffffffu, __float_as_uint(v[c].w) & 0x7fffffffu)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    int Ex = (int)(mx >> 23);
    Ex = Ex < 1 ? 1 : (Ex > 254 ? 254 : Ex);
    if (lane == 0) E[row] = Ex;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = c * 64 + lane;
        if (4 * i >= K) continue;
        int d[4][3];
        q_digits(q_value_dev(v[c].x, Ex), d[0][0], d[0][1], d[0][2]);
        q_digits(q_value_dev(v[c].y, Ex), d[1][0], d[1][1], d[1][2]);
        q_digits(q_value_dev(v[c].z, Ex), d[2][0], d[2][1], d[2][2]);
        q_digits(q_value_dev(v[c].w, Ex), d[3][0], d[3][1], d[3][2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned w = (unsigned)(d[0][p] & 255) | ((unsigned)(d[1][p] & 255) << 8) | ((unsigned)(d[2][p] & 255) << 16) | ((unsigned)(d[3][p] & 255) << 24);
            *reinterpret_cast<unsigned *>(D + p * plane + (size_t)row * K + 4 * i) = w;
        }
    }
}

// the folded weights W [K][N] (float32) -> digit planes D[p][n][K] + the column exponents: one wave per output column (commit time)
__global__ __launch_bounds__(256) void k_q_cols(const float *__restrict__ W, int K, int N, int8_t *__restrict__ D, int *__restrict__ E) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    unsigned mx = 0;
    for (int k = lane; k < K; k += 64) mx = max(mx, __float_as_uint(W[(size_t)k * N + n]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    int Ex = (int)(mx >> 23);
    Ex = Ex < 1 ? 1 : (Ex > 254 ? 254 : Ex);
    if (lane == 0) E[n] = Ex;
    const size_t plane = (size_t)N * K;
    for (int k = lane; k < K; k += 64) {
        int d0, d1, d2;
        q_digits(q_value_dev(W[(size_t)k * N + n], Ex), d0, d1, d2);
        D[(size_t)n * K + k] = (int8_t)d0; D[plane + (size_t)n * K + k] = (int8_t)d1; D[2 * plane + (size_t)n * K + k] = (int8_t)d2;
    }
}

// C[M][N] = relu(dequant(A digits x B digits) + bias).  Workgroup = NWM x NWN waves (eight: two per SIMD), wave tile = WM x WN MFMA tiles
// of 32 x 32, five int32 accumulators per tile (digit products of equal weight).  K is walked in chunks of 32 (one MFMA step).  A chunk
// of an operand is a set of 1 KB blocks, one per (32 rows, digit plane), laid out [k half h][row r][16 bytes] -- exactly the order in
// which the 64 lanes of the MFMA fragment read it (lane = 32 h + r: one linear, conflict-free ds_read_b128 per fragment) and exactly what
// ONE global_load_lds_dwordx4 writes (LDS destination = wave-uniform base + 16 lane; the per-lane SOURCE address picks row r, k half h).
// The chunks go global -> LDS without touching a register, four stages deep: the loads of chunk kc + 3 are issued while chunk kc is
// multiplied, each wave waits for ITS loads of chunk kc with a counted s_waitcnt vmcnt, the barrier behind it makes everybody's visible,
// and the stage being refilled was last read before that barrier.  The int8 A / B fragments pair byte j of lane half h in both
// operands, so whatever order the instruction walks its 32 k in, every k meets its partner.
template <int NWM, int NWN, int WM, int WN>
constexpr int qgemm_lds_bytes() { return 4 * 3 * (NWM * WM + NWN * WN) * 1024; }

template <int NWM, int NWN, int WM, int WN, bool RELU>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN == 4 ? 2 : 1)) void k_qgemm(const int8_t *__restrict__ Ad, size_t planeA, const int *__restrict__ Ea, const int8_t *__restrict__ Bd,
                                                          size_t planeB, const int *__restrict__ Eb, const float *__restrict__ bias, float *__restrict__ C, int M,
                                                          int N, int K, const int *__restrict__ dyn) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (the LDS-DMA builtin and the counted waits exist on the device side only)
    if (dyn) { int c = *dyn; M = c < M ? c : M; }
    constexpr int NW = NWM * NWN, GA = NWM * WM, GB = NWN * WN, BM = GA * 32, BN = GB * 32, NBLK = 3 * (GA + GB), PER = (NBLK + NW - 1) / NW, STAGE = NBLK * 1024;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= M) return;  // uniform
    extern __shared__ __attribute__((aligned(16))) int8_t qlds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave / NWN, wn = wave % NWN, r = lane & 31, h = lane >> 5;
    v16i acc[WM][WN][5];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[i][j][s5][t] = 0;
    // this wave's blocks of a chunk (a wave with one block fewer than the others loads the chunk's first blocks once more: every wave
    // then has PER loads per chunk in flight, which is what the counted waits assume)
    const int8_t *src[PER];
    int dst[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        int b = wave + NW * e;
        if (b >= NBLK) b -= NBLK;
        const bool isA = b < 3 * GA;
        const int g = (isA ? b : b - 3 * GA) / 3, p = (isA ? b : b - 3 * GA) % 3;
        src[e] = isA ? Ad + p * planeA + (size_t)(m0 + 32 * g + r) * K + 16 * h : Bd + p * planeB + (size_t)(n0 + 32 * g + r) * K + 16 * h;
        dst[e] = b * 1024;
    }
    const int NKC = K / 32;
#define QG_ISSUE(kc_)                                                                                                                           \
    _Pragma("unroll") for (int e = 0; e < PER; ++e)                                                                                             \
        __builtin_amdgcn_global_load_lds(src[e] + (kc_) * 32, (__attribute__((address_space(3))) void *)(qlds + ((kc_) & 3) * STAGE + dst[e]), 16, 0, 0);
    QG_ISSUE(0)
    if (NKC > 1) QG_ISSUE(1)
    if (NKC > 2) QG_ISSUE(2)
    for (int kc = 0; kc < NKC; ++kc) {
        // my loads of chunk kc have landed when at most the later chunks' (two, one or none) are still in flight
        if (kc + 2 < NKC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (kc + 1 < NKC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kc + 3 < NKC) QG_ISSUE(kc + 3)  // into the stage chunk kc - 1 was read from: everybody is past that
        const int8_t *St = qlds + (kc & 3) * STAGE + 16 * lane;
        v4i a[WM][3], b[WN][3];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const v4i *>(St + ((wm * WM + i) * 3 + p) * 1024);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const v4i *>(St + (3 * GA + (wn * WN + j) * 3 + p) * 1024);
        // the nine digit pairs in an order that keeps two MFMAs on one accumulator apart
#define QG_PAIR(PA_, PB_)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int j = 0; j < WN; ++j)                                \
        acc[i][j][PA_ + PB_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i][PA_], b[j][PB_], acc[i][j][PA_ + PB_], 0, 0, 0);
        QG_PAIR(0, 0) QG_PAIR(0, 1) QG_PAIR(0, 2) QG_PAIR(1, 2) QG_PAIR(2, 2) QG_PAIR(1, 0) QG_PAIR(1, 1) QG_PAIR(2, 1) QG_PAIR(2, 0)
#undef QG_PAIR
    }
#undef QG_ISSUE
    // C layout of 32x32: column lane & 31, row (t & 3) + 8 (t >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        int ea[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = m0 + (wm * WM + i) * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
            ea[t] = Ea[row] - 296;  // rows past M lie inside the buffer (a whole tile of rows is allocated)
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + (wn * WN + j) * 32 + r;
            const int eb = Eb[col];
            const float bv = bias[col];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int row = m0 + (wm * WM + i) * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
                const double hi = fma((double)acc[i][j][4][t], 65536.0, fma((double)acc[i][j][3][t], 256.0, (double)acc[i][j][2][t]));
                const double lo = fma((double)acc[i][j][1][t], 256.0, (double)acc[i][j][0][t]);
                const double x = fma(hi, 65536.0, lo);  // = (double)X, one rounding
                float v = (float)ldexp(x, ea[t] + eb) + bv;
                if (RELU) v = v > 0.0f ? v : 0.0f;
                if (row < M) C[(size_t)row * N + col] = v;
            }
        }
    }
#endif
}

// The same layer for a handful of rows (the latency regime: one game's search, small waves): 16 rows x 64 columns per workgroup, no
// separate quantisation launch.  Every thread fetches its share of the 16 rows into registers (one round trip), the rows' largest
// magnitudes are reduced with shuffles and one exchange through LDS, the digits are formed from the registers and written to LDS as
// three planes; wave w then owns the 16-column tile w: v_mfma_i32_16x16x64_i8, its weight fragments (16 bytes per lane, digit plane and
// 64-k step) straight from L2 and requested before anything else.  Five int32 accumulators: the dependent chain is K / 64 steps of two
// or three MFMAs instead of the K / 4 dependent f32 MFMAs of k_dense_frag.  Same X, same epilogue: the same bits as k_qgemm.
template <int K, bool RELU>
__global__ __launch_bounds__(256) void k_qdense_small(const float *__restrict__ X, const int8_t *__restrict__ Bd, size_t planeB, const int *__restrict__ Eb,
                                                      const float *__restrict__ bias, float *__restrict__ C, int M, int N, const int *__restrict__ dyn) {
    if (dyn) { int c = *dyn; M = c < M ? c : M; }
    const int row0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
    if (row0 >= M) return;  // uniform
    constexpr int N4 = K / 4, IT = 16 * N4 / 256, DS = K + 16, P = N4 >= 64 ? N4 / 64 : 1, W = N4 >= 64 ? 64 : N4, NS = K / 64, HS = NS < 8 ? NS : 8;
    static_assert(IT * 256 == 16 * N4 && (N4 >= 64 ? N4 % 64 == 0 : 64 % N4 == 0), "whole float4s per thread, whole rows per wave part");
    __shared__ __attribute__((aligned(16))) int8_t dq[3 * 16 * DS];
    __shared__ unsigned pm[16][P];
    __shared__ int es[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ncol = n0 + 16 * wave + (lane & 15), kq = lane >> 4;
    const int8_t *wb = Bd + (size_t)ncol * K + 16 * kq;
    v4i bf[HS][3];  // this wave's weight fragments of the first 512 k: they do not depend on the rows
#pragma unroll
    for (int st = 0; st < HS; ++st)
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[st][p] = *reinterpret_cast<const v4i *>(wb + p * planeB + 64 * st);
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int q = tid + 256 * i, r = q / N4, c = q % N4;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < M) v[i] = reinterpret_cast<const float4 *>(X + (size_t)(row0 + r) * K)[c];
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int q = tid + 256 * i, r = q / N4, c = q % N4;
        unsigned mx = max(max(__float_as_uint(v[i].x) & 0x7fffffffu, __float_as_uint(v[i].y) & 0x7fffffffu),
                          max(__float_as_uint(v[i].z) & 0x7fffffffu, __float_as_uint(v[i].w) & 0x7fffffffu));
#pragma unroll
        for (int o = W / 2; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if ((c & (W - 1)) == 0) pm[r][c / W % P] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int q = tid + 256 * i, r = q / N4, c = q % N4;
        unsigned mx = pm[r][0];
#pragma unroll
        for (int e = 1; e < P; ++e) mx = max(mx, pm[r][e]);
        int Ex = (int)(mx >> 23);
        Ex = Ex < 1 ? 1 : (Ex > 254 ? 254 : Ex);
        if (c == 0) es[r] = Ex;
        int d[4][3];
        q_digits(q_value_dev(v[i].x, Ex), d[0][0], d[0][1], d[0][2]);
        q_digits(q_value_dev(v[i].y, Ex), d[1][0], d[1][1], d[1][2]);
        q_digits(q_value_dev(v[i].z, Ex), d[2][0], d[2][1], d[2][2]);
        q_digits(q_value_dev(v[i].w, Ex), d[3][0], d[3][1], d[3][2]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<unsigned *>(dq + (p * 16 + r) * DS + 4 * c) =
                (unsigned)(d[0][p] & 255) | ((unsigned)(d[1][p] & 255) << 8) | ((unsigned)(d[2][p] & 255) << 16) | ((unsigned)(d[3][p] & 255) << 24);
    }
    __syncthreads();
    v4i acc[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) acc[s5] = (v4i){0, 0, 0, 0};
    const int8_t *ab = dq + (lane & 15) * DS + 16 * kq;
#pragma unroll
    for (int h0 = 0; h0 < NS; h0 += HS) {
        if (h0 > 0) {  // the second half's weight fragments (K = 1024)
#pragma unroll
            for (int st = 0; st < HS; ++st)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[st][p] = *reinterpret_cast<const v4i *>(wb + p * planeB + 64 * (h0 + st));
        }
#pragma unroll
        for (int st = 0; st < HS; ++st) {
            v4i af[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const v4i *>(ab + p * 16 * DS + 64 * (h0 + st));
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0], bf[st][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0], bf[st][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0], bf[st][2], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1], bf[st][2], acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2], bf[st][2], acc[4], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1], bf[st][0], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1], bf[st][1], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2], bf[st][1], acc[3], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2], bf[st][0], acc[2], 0, 0, 0);
        }
    }
    // C layout of 16x16: column lane & 15, rows 4 (lane >> 4) + t
    const int eb = Eb[ncol] - 296;
    const float bv = bias[ncol];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = 4 * kq + t, row = row0 + r;
        const double hi = fma((double)acc[4][t], 65536.0, fma((double)acc[3][t], 256.0, (double)acc[2][t]));
        const double lo = fma((double)acc[1][t], 256.0, (double)acc[0][t]);
        const double x = fma(hi, 65536.0, lo);
        float val = (float)ldexp(x, es[r] + eb) + bv;
        if (RELU) val = val > 0.0f ? val : 0.0f;
        if (row < M) C[(size_t)row * N + ncol] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// layers k_dense_frag can run: K in whole 128-k groups of fragments (PF = 8 loads of 16 k), the 16 rows of K + 4 floats within the LDS
static bool frag_shape(int N, int K) { return N % 64 == 0 && K % 128 == 0 && K <= 2048; }

// AZ_DENSE_I8=1: OthelloNet's fc1 / fc2 in the exact block-fixed-point form (k_q_rows + k_qgemm); the CPU oracle reads the same variable
static bool use_qdense(int game) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("AZ_DENSE_I8"); v = (e && atoi(e) == 1) ? 1 : 0; }
    return v == 1 && game == AZ_OTHELLO;
}

struct az_net {
    int game, H, W, CH, CW, A, F1, F2, FIN, NH, max_batch;
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::pair<float *, size_t>> raw_dev;  // state_dict tensors handed over in device memory (az_net_set_tensor_device)
    std::vector<void *> allocs;
    TrunkParams tp;
    float *fc1w, *fc1b, *fc2w, *fc2b, *hw, *hb;
    float *hwq = nullptr;  // head matrix in 16x16x4 B-fragment order [F2/16][NH/16][64 lanes][4] (k_heads2)
    float *fc1wq = nullptr, *fc2wq = nullptr;  // fc1 / fc2 in the same fragment order (k_dense_frag); null where the shapes do not tile
    float *feat, *h1, *h2;
    // exact block-fixed-point dense layers (AZ_DENSE_I8=1, OthelloNet): digit planes + exponents of the two weight matrices and of the
    // activation rows (one buffer, used by fc1 and then by fc2)
    bool qd_on = false;
    int8_t *qd_w1 = nullptr, *qd_w2 = nullptr, *qd_a = nullptr;
    int *qd_e1 = nullptr, *qd_e2 = nullptr, *qd_ea = nullptr;
    size_t qd_rows = 0;  // rows of a digit plane of qd_a (max_batch rounded up to a whole tile)
    MlpParams mlp;           // host staging of the TicTacToe MLP
    MlpParams *mlp_dev = nullptr;  // what k_mlp reads
    bool committed;
    // live per-stage timing (az_net_profile): HIP events around every stage launch, harvested in batches
    bool prof = false;
    int last_trunk_two_boards = 0;
    std::vector<hipEvent_t> prof_ev;  // [PROF_SLOTS][4 stages][start, stop]
    std::vector<int> prof_has;        // bit s: stage s of that forward launched a kernel (the fused Connect4 tail: stage 1 only)
    std::vector<int> prof_kind;       // trunk kernel used by the forward in that slot
    int prof_used = 0;
    double prof_empty_ms = 0.0;   // kept for az_net_profile_overhead: the start / stop events of a launch need no correction
    int last_trunk_q = 0;
    // k_trunk2, fc1 / fc2 on the tiled GEMMs (k_gemm, k_gemm_solo, k_gemm_solo_t; Connect4Net: the fused tail in the fc1 slot), heads,
    // k_trunk (one board per wave) | fc1 / fc2 on the small-batch kernels (k_dense_frag, k_dense_small), k_trunk_q: a slot per KERNEL
    // family, so that a slot's mean is the figure rocprofv3 lists for that kernel
    double prof_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
#define PROF_SLOTS 2048

#define AZ_TRY(x) do { int _rc = (x); if (_rc != AZ_OK) return _rc; } while (0)

static int net_alloc(az_net *n, float **p, size_t count) {
    void *q = nullptr;
    AZ_HIP(hipMalloc(&q, count * sizeof(float)));
    n->allocs.push_back(q);
    *p = (float *)q;
    return AZ_OK;
}

extern "C" int az_net_create(int game, int H, int W, int max_batch, az_net **out) {
    AZ_REQUIRE(out && max_batch > 0, AZ_EINVAL, "bad arguments");
    GameDesc gd;
    AZ_TRY(az_make_game_desc(game, H, W, &gd));
    az_net *n = new az_net();
    n->game = game; n->H = H; n->W = W; n->A = gd.A; n->max_batch = max_batch; n->committed = false;
    if (game == AZ_OTHELLO) { n->CH = H; n->CW = W; n->F1 = 1024; n->F2 = 512; }       // othello.py:352-358
    else if (game == AZ_CONNECT4) { n->CH = W; n->CW = H; n->F1 = 64; n->F2 = 32; }     // connect4.py:382-388, 399
    else { n->CH = 3; n->CW = 3; n->F1 = 9; n->F2 = 9; }
    n->FIN = game == AZ_TICTACTOE ? 9 : NCH * (n->CH - 4) * (n->CW - 4);
    n->NH = ((n->A + 1 + 15) / 16) * 16;
    if (game != AZ_TICTACTOE) {
        // 8x8, 6x6 and 7x6 planes run the tuned kernels (k_trunk2 from 4096 boards up, Winograd conv2); every other plane between 5x5
        // and 8x8 -- Connect4Net on the board sizes the engine plays (connect4.py:343-368) -- runs the one-board-per-wave kernel in the
        // direct form at any batch size.  Planes below 5x5 leave conv4 without an output (the reference's forward fails there too).
        bool ok = n->CH >= 5 && n->CH <= 8 && n->CW >= 5 && n->CW <= 8;
        if (!ok) { delete n; az_set_error("no conv-trunk kernel instantiated for a %dx%d plane", n->CH, n->CW); return AZ_EINVAL; }
        int rc = AZ_OK;
        float *p;
#define NA(field, cnt) if (rc == AZ_OK) { rc = net_alloc(n, &p, (cnt)); field = p; }
        NA(n->tp.w1f, 3 * 2 * 64) NA(n->tp.b1, NCH)
        for (int l = 0; l < 3; ++l) { NA(n->tp.cb[l], NCH) }
        NA(n->tp.w1p, 5 * 64)
        for (int l = 0; l < 3; ++l) { NA(n->tp.wp[l], 9 * 16 * 64) NA(n->tp.wq[l], 9 * 16 * 64) }
        NA(n->tp.wu, 2 * 8 * 4 * 64 * 4) NA(n->tp.wu32, 4 * 16 * 64 * 4)
        NA(n->fc1w, (size_t)n->FIN * n->F1) NA(n->fc1b, n->F1) NA(n->fc2w, (size_t)n->F1 * n->F2) NA(n->fc2b, n->F2)
        NA(n->hw, (size_t)n->F2 * n->NH) NA(n->hb, n->NH) NA(n->hwq, (size_t)n->F2 * n->NH)
        if (frag_shape(n->F1, n->FIN)) { NA(n->fc1wq, (size_t)n->FIN * n->F1) }
        if (frag_shape(n->F2, n->F1)) { NA(n->fc2wq, (size_t)n->F1 * n->F2) }
        NA(n->feat, (size_t)max_batch * n->FIN) NA(n->h1, (size_t)max_batch * n->F1) NA(n->h2, (size_t)max_batch * n->F2)
        n->qd_on = use_qdense(game) && n->FIN % 64 == 0 && n->F1 % 128 == 0 && n->F2 % 128 == 0 && n->F1 <= 1024 && n->FIN <= 1024;
        if (rc == AZ_OK && use_qdense(game) && !n->qd_on) {
            // the oracle switches every OthelloNet under the variable: a shape the fixed-point kernels do not tile must not silently run the f32 chains
            az_set_error("AZ_DENSE_I8=1: no fixed-point dense kernels for FIN=%d, F1=%d, F2=%d (need FIN %% 64 = 0, F1 and F2 %% 128 = 0, both K <= 1024)", n->FIN, n->F1, n->F2);
            rc = AZ_EINVAL;
        }
        if (n->qd_on) {
            n->qd_rows = ((size_t)max_batch + 127) / 128 * 128;
            const size_t kmax = (size_t)(n->FIN > n->F1 ? n->FIN : n->F1);
            NA(p, (3 * (size_t)n->F1 * n->FIN + 3) / 4) n->qd_w1 = reinterpret_cast<int8_t *>(p);
            NA(p, (3 * (size_t)n->F2 * n->F1 + 3) / 4) n->qd_w2 = reinterpret_cast<int8_t *>(p);
            NA(p, (3 * n->qd_rows * kmax + 3) / 4) n->qd_a = reinterpret_cast<int8_t *>(p);
            NA(p, n->F1) n->qd_e1 = reinterpret_cast<int *>(p);
            NA(p, n->F2) n->qd_e2 = reinterpret_cast<int *>(p);
            NA(p, n->qd_rows) n->qd_ea = reinterpret_cast<int *>(p);
            if (rc == AZ_OK && hipMemset(n->qd_a, 0, 3 * n->qd_rows * kmax) != hipSuccess) rc = AZ_EHIP;  // rows past the batch are read by the last tile, never stored
        }
#undef NA
        if (rc != AZ_OK) { az_net_destroy(n); return rc; }
    } else {
        float *p = nullptr;
        if (net_alloc(n, &p, (sizeof(MlpParams) + sizeof(float) - 1) / sizeof(float)) != AZ_OK) { az_net_destroy(n); return AZ_EHIP; }
        n->mlp_dev = reinterpret_cast<MlpParams *>(p);
    }
    *out = n;
    return AZ_OK;
}

extern "C" void az_net_destroy(az_net *n) {
    if (!n) return;
    for (void *p : n->allocs) (void)hipFree(p);
    for (auto &e : n->prof_ev) (void)hipEventDestroy(e);
    delete n;
}

extern "C" int az_net_action_size(const az_net *n) { return n ? n->A : 0; }

extern "C" int64_t az_net_flops_per_board(const az_net *n) {
    if (!n) return 0;
    if (n->game == AZ_TICTACTOE) return 2 * (81 + 81 + 90);
    int64_t p1 = n->CH * n->CW, p3 = (n->CH - 2) * (n->CW - 2), p4 = (n->CH - 4) * (n->CW - 4);
    int64_t mac = p1 * 9 * NCH + p1 * 9 * NCH * NCH + p3 * 9 * NCH * NCH + p4 * 9 * NCH * NCH;
    mac += (int64_t)n->FIN * n->F1 + (int64_t)n->F1 * n->F2 + (int64_t)n->F2 * (n->A + 1);
    return 2 * mac;
}

extern "C" int az_net_set_tensor(az_net *n, const char *name, const float *h_data, int64_t numel) {
    AZ_REQUIRE(n && name && h_data && numel > 0, AZ_EINVAL, "bad arguments");
    n->raw[name] = std::vector<float>(h_data, h_data + numel);
    n->committed = false;
    return AZ_OK;
}

static int need(az_net *n, const std::string &k, size_t numel, const std::vector<float> **out) {
    auto it = n->raw.find(k);
    AZ_REQUIRE(it != n->raw.end(), AZ_ESTATE, "missing tensor '%s'", k.c_str());
    AZ_REQUIRE(it->second.size() == numel, AZ_EINVAL, "tensor '%s' has %zu elements, expected %zu", k.c_str(),
               it->second.size(), numel);
    *out = &it->second;
    return AZ_OK;
}

#define BN_EPS 1e-5

// s = gamma / sqrt(var + eps);  w' = (float)(w * s);  b' = (float)((b - mean) * s + beta)   [float64]
static int bn_scale(az_net *n, const std::string &bn, int C, std::vector<double> &s, std::vector<double> &shift_mean,
                    std::vector<double> &beta) {
    const std::vector<float> *g, *b, *m, *v;
    AZ_TRY(need(n, bn + ".weight", C, &g)); AZ_TRY(need(n, bn + ".bias", C, &b));
    AZ_TRY(need(n, bn + ".running_mean", C, &m)); AZ_TRY(need(n, bn + ".running_var", C, &v));
    s.resize(C); shift_mean.resize(C); beta.resize(C);
    for (int i = 0; i < C; ++i) {
        s[i] = (double)(*g)[i] / sqrt((double)(*v)[i] + BN_EPS);
        shift_mean[i] = (double)(*m)[i];
        beta[i] = (double)(*b)[i];
    }
    return AZ_OK;
}

static int upload(float *dst, const std::vector<float> &src, hipStream_t st) {
    AZ_HIP(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice, st));
    AZ_HIP(hipStreamSynchronize(st));  // src is a temporary
    return AZ_OK;
}

// the folded [K][N] matrix once more in B-fragment order (k_dense_frag); a layer whose shape does not tile has no such copy
static int retile_q(const float *w, float *dst, int K, int N, hipStream_t st) {
    if (!dst) return AZ_OK;
    hipLaunchKernelGGL(k_retile_q, dim3((unsigned)(((size_t)K * N + 255) / 256)), dim3(256), 0, st, w, dst, K, N);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

// the digit planes of a folded dense matrix (AZ_DENSE_I8)
static int quant_w(az_net *n, int layer, hipStream_t st) {
    if (!n->qd_on) return AZ_OK;
    const int K = layer == 1 ? n->FIN : n->F1, N = layer == 1 ? n->F1 : n->F2;
    hipLaunchKernelGGL(k_q_cols, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, layer == 1 ? n->fc1w : n->fc2w, K, N, layer == 1 ? n->qd_w1 : n->qd_w2,
                       layer == 1 ? n->qd_e1 : n->qd_e2);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_net_commit(az_net *n, void *stream) {
    AZ_REQUIRE(n, AZ_EINVAL, "null net");
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> s, mean, beta;
    const std::vector<float> *w, *b;
    if (n->game == AZ_TICTACTOE) {
        const char *fc[2] = {"fc1", "fc2"};
        const char *bn[2] = {"bn1", "bn2"};
        for (int l = 0; l < 2; ++l) {
            AZ_TRY(bn_scale(n, bn[l], 9, s, mean, beta));
            AZ_TRY(need(n, std::string(fc[l]) + ".weight", 81, &w)); AZ_TRY(need(n, std::string(fc[l]) + ".bias", 9, &b));
            float *fw = l == 0 ? n->mlp.f1w : n->mlp.f2w, *fb = l == 0 ? n->mlp.f1b : n->mlp.f2b;
            for (int j = 0; j < 9; ++j) {
                fb[j] = (float)(((double)(*b)[j] - mean[j]) * s[j] + beta[j]);
                for (int k = 0; k < 9; ++k) fw[k * 9 + j] = (float)((double)(*w)[j * 9 + k] * s[j]);
            }
        }
        const std::vector<float> *pw, *pb, *vw, *vb;
        AZ_TRY(need(n, "fc_probs.weight", 81, &pw)); AZ_TRY(need(n, "fc_probs.bias", 9, &pb));
        AZ_TRY(need(n, "fc_value.weight", 9, &vw)); AZ_TRY(need(n, "fc_value.bias", 1, &vb));
        for (int a = 0; a < 9; ++a) { n->mlp.hb[a] = (*pb)[a]; for (int k = 0; k < 9; ++k) n->mlp.hw[k * 10 + a] = (*pw)[a * 9 + k]; }
        n->mlp.hb[9] = (*vb)[0];
        for (int k = 0; k < 9; ++k) n->mlp.hw[k * 10 + 9] = (*vw)[k];
        AZ_HIP(hipMemcpyAsync(n->mlp_dev, &n->mlp, sizeof(MlpParams), hipMemcpyHostToDevice, st));
        AZ_HIP(hipStreamSynchronize(st));
        n->committed = true;
        return AZ_OK;
    }
    for (int l = 0; l < 4; ++l) {
        int IC = l == 0 ? 1 : NCH;
        std::string cn = "conv" + std::to_string(l + 1), bnn = "bn" + std::to_string(l + 1);
        AZ_TRY(bn_scale(n, bnn, NCH, s, mean, beta));
        AZ_TRY(need(n, cn + ".weight", (size_t)NCH * IC * 9, &w)); AZ_TRY(need(n, cn + ".bias", NCH, &b));
        std::vector<float> fb(NCH);
        for (int oc = 0; oc < NCH; ++oc) fb[oc] = (float)(((double)(*b)[oc] - mean[oc]) * s[oc] + beta[oc]);
        if (l == 0) {
            // MFMA B-fragment order [k-step s][nt][lane] = W'[oc = nt*16 + (lane&15)][tap = 4s + (lane>>4)], 0 for tap >= 9
            std::vector<float> fw(3 * 2 * 64, 0.0f);
            for (int sidx = 0; sidx < 3; ++sidx)
                for (int nt = 0; nt < 2; ++nt)
                    for (int lane = 0; lane < 64; ++lane) {
                        int oc = nt * 16 + (lane & 15), t = 4 * sidx + (lane >> 4);
                        if (t < 9) fw[(sidx * 2 + nt) * 64 + lane] = (float)((double)(*w)[oc * 9 + t] * s[oc]);
                    }
            AZ_TRY(upload((float *)n->tp.w1f, fw, st)); AZ_TRY(upload((float *)n->tp.b1, fb, st));
            // 32x32x2 B-fragment order [k-step s][lane] = W'[oc = lane&31][tap = 2s + (lane>>5)], 0 for tap 9
            std::vector<float> fp(5 * 64, 0.0f);
            for (int sidx = 0; sidx < 5; ++sidx)
                for (int lane = 0; lane < 64; ++lane) {
                    int oc = lane & 31, t = 2 * sidx + (lane >> 5);
                    if (t < 9) fp[sidx * 64 + lane] = (float)((double)(*w)[oc * 9 + t] * s[oc]);
                }
            AZ_TRY(upload((float *)n->tp.w1p, fp, st));
        } else {
            AZ_TRY(upload((float *)n->tp.cb[l - 1], fb, st));
            // 32x32x2 B fragments, [tap][j / 4][lane][j % 4] = W'[oc = lane&31][ic = 2j + (lane>>5)][tap]: a lane fetches the
            // fragments of four consecutive k-steps with one 16-byte load
            std::vector<float> fp(9 * 16 * 64);
            for (int t = 0; t < 9; ++t)
                for (int j = 0; j < 16; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        int oc = lane & 31, ic = 2 * j + (lane >> 5);
                        fp[((t * 4 + j / 4) * 64 + lane) * 4 + j % 4] = (float)((double)(*w)[(oc * NCH + ic) * 9 + t] * s[oc]);
                    }
            AZ_TRY(upload((float *)n->tp.wp[l - 1], fp, st));
            // the same for the 16x16x4 fragments (k_trunk, and k_trunk2's 16-row tiles): fragment i = 2 j8 + nt at
            // [tap][i / 4][lane][i % 4] = W'[oc = nt*16 + (lane&15)][ic = 4 j8 + (lane>>4)][tap]
            for (int t = 0; t < 9; ++t)
                for (int j = 0; j < 8; ++j)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane) {
                            int oc = nt * 16 + (lane & 15), ic = 4 * j + (lane >> 4), i = 2 * j + nt;
                            fp[((t * 4 + i / 4) * 64 + lane) * 4 + i % 4] = (float)((double)(*w)[(oc * NCH + ic) * 9 + t] * s[oc]);
                        }
            AZ_TRY(upload((float *)n->tp.wq[l - 1], fp, st));
            if (l == 1) {
                // conv2 in the Winograd F(2x2,3x3) form: U[f = 4 i + jj][ic][oc] = (G g' G^T)[i][jj], g' = w * s, all in float64, rounded
                // once; stored as 16x16x4 B fragments [pass p][k-step j][e / 4][lane][e % 4], e = 2 f8 + nt, f = 8 p + f8:
                // lane (n = lane & 15, kq = lane >> 4) holds U[f][ic = 4 j + kq][oc = 16 nt + n]
                static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
                std::vector<float> fu(2 * 8 * 4 * 64 * 4);
                for (int pp = 0; pp < 2; ++pp)
                    for (int j = 0; j < 8; ++j)
                        for (int e = 0; e < 16; ++e)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int f = 8 * pp + e / 2, nt = e % 2, oc = 16 * nt + (lane & 15), ic = 4 * j + (lane >> 4);
                                const int fi = f / 4, fj = f % 4;
                                double t3[3];
                                for (int b = 0; b < 3; ++b) {
                                    const double g0 = (double)(*w)[(oc * NCH + ic) * 9 + 0 * 3 + b] * s[oc], g1 = (double)(*w)[(oc * NCH + ic) * 9 + 1 * 3 + b] * s[oc],
                                                 g2 = (double)(*w)[(oc * NCH + ic) * 9 + 2 * 3 + b] * s[oc];
                                    t3[b] = (G[fi][0] * g0 + G[fi][1] * g1) + G[fi][2] * g2;
                                }
                                fu[(((size_t)(pp * 8 + j) * 4 + e / 4) * 64 + lane) * 4 + e % 4] = (float)((t3[0] * G[fj][0] + t3[1] * G[fj][1]) + t3[2] * G[fj][2]);
                            }
                AZ_TRY(upload((float *)n->tp.wu, fu, st));
                // and as 32x32x2 B fragments [jq][f][lane][i]: k-step j = 4 jq + i, lane (oc = lane & 31, kk = lane >> 5) holds U[f][ic = 2 j + kk][oc]
                std::vector<float> fv(4 * 16 * 64 * 4);
                for (int jq = 0; jq < 4; ++jq)
                    for (int f = 0; f < 16; ++f)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 4; ++i) {
                                const int oc = lane & 31, ic = 2 * (4 * jq + i) + (lane >> 5), fi = f / 4, fj = f % 4;
                                double t3[3];
                                for (int b = 0; b < 3; ++b) {
                                    const double g0 = (double)(*w)[(oc * NCH + ic) * 9 + 0 * 3 + b] * s[oc], g1 = (double)(*w)[(oc * NCH + ic) * 9 + 1 * 3 + b] * s[oc],
                                                 g2 = (double)(*w)[(oc * NCH + ic) * 9 + 2 * 3 + b] * s[oc];
                                    t3[b] = (G[fi][0] * g0 + G[fi][1] * g1) + G[fi][2] * g2;
                                }
                                fv[(((size_t)jq * 16 + f) * 64 + lane) * 4 + i] = (float)((t3[0] * G[fj][0] + t3[1] * G[fj][1]) + t3[2] * G[fj][2]);
                            }
                AZ_TRY(upload((float *)n->tp.wu32, fv, st));
            }
        }
    }
    {
        AZ_TRY(bn_scale(n, "fc_bn1", n->F1, s, mean, beta));
        AZ_TRY(need(n, "fc1.weight", (size_t)n->F1 * n->FIN, &w)); AZ_TRY(need(n, "fc1.bias", n->F1, &b));
        std::vector<float> fw((size_t)n->FIN * n->F1), fb(n->F1);
        for (int j = 0; j < n->F1; ++j) {
            fb[j] = (float)(((double)(*b)[j] - mean[j]) * s[j] + beta[j]);
            for (int k = 0; k < n->FIN; ++k) fw[(size_t)k * n->F1 + j] = (float)((double)(*w)[(size_t)j * n->FIN + k] * s[j]);
        }
        AZ_TRY(upload(n->fc1w, fw, st)); AZ_TRY(upload(n->fc1b, fb, st));
        AZ_TRY(retile_q(n->fc1w, n->fc1wq, n->FIN, n->F1, st));
        AZ_TRY(quant_w(n, 1, st));
    }
    {
        AZ_TRY(bn_scale(n, "fc_bn2", n->F2, s, mean, beta));
        AZ_TRY(need(n, "fc2.weight", (size_t)n->F2 * n->F1, &w)); AZ_TRY(need(n, "fc2.bias", n->F2, &b));
        std::vector<float> fw((size_t)n->F1 * n->F2), fb(n->F2);
        for (int j = 0; j < n->F2; ++j) {
            fb[j] = (float)(((double)(*b)[j] - mean[j]) * s[j] + beta[j]);
            for (int k = 0; k < n->F1; ++k) fw[(size_t)k * n->F2 + j] = (float)((double)(*w)[(size_t)j * n->F1 + k] * s[j]);
        }
        AZ_TRY(upload(n->fc2w, fw, st)); AZ_TRY(upload(n->fc2b, fb, st));
        AZ_TRY(retile_q(n->fc2w, n->fc2wq, n->F1, n->F2, st));
        AZ_TRY(quant_w(n, 2, st));
    }
    {
        const std::vector<float> *pw, *pb, *vw, *vb;
        AZ_TRY(need(n, "fc_probs.weight", (size_t)n->A * n->F2, &pw)); AZ_TRY(need(n, "fc_probs.bias", n->A, &pb));
        AZ_TRY(need(n, "fc_value.weight", n->F2, &vw)); AZ_TRY(need(n, "fc_value.bias", 1, &vb));
        std::vector<float> fw((size_t)n->F2 * n->NH, 0.0f), fb(n->NH, 0.0f);
        for (int a = 0; a < n->A; ++a) { fb[a] = (*pb)[a]; for (int k = 0; k < n->F2; ++k) fw[(size_t)k * n->NH + a] = (*pw)[(size_t)a * n->F2 + k]; }
        fb[n->A] = (*vb)[0];
        for (int k = 0; k < n->F2; ++k) fw[(size_t)k * n->NH + n->A] = (*vw)[k];
        AZ_TRY(upload(n->hw, fw, st)); AZ_TRY(upload(n->hb, fb, st));
        // the same matrix in B-fragment order: [kb][nt][lane][i] = Wh[k = 16 kb + 4 i + (lane >> 4)][n = 16 nt + (lane & 15)]
        const int NT = n->NH / 16;
        std::vector<float> fq((size_t)n->F2 * n->NH, 0.0f);
        if (n->F2 % 16 == 0)
            for (int kb = 0; kb < n->F2 / 16; ++kb)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 4; ++i)
                            fq[(((size_t)kb * NT + nt) * 64 + lane) * 4 + i] = fw[(size_t)(16 * kb + 4 * i + (lane >> 4)) * n->NH + 16 * nt + (lane & 15)];
        AZ_TRY(upload(n->hwq, fq, st));
    }
    n->committed = true;
    return AZ_OK;
}

// ---------------------------------------------------------------------------------------------
// The same fold + re-tiling on the device (weight hand-off after optimize_network without a host round trip,
// trainer.py:383-387): one thread per destination element, float64 arithmetic in the host code's operation order
// (no contraction), so both paths produce identical bits.
// ---------------------------------------------------------------------------------------------
enum { FOLD_BIAS = 0, FOLD_W1F, FOLD_W1P, FOLD_WP, FOLD_WQ, FOLD_WINO, FOLD_WINO32, FOLD_DENSE_T, FOLD_HEADS_W, FOLD_HEADS_WQ, FOLD_HEADS_B };
struct FoldJob {
    int mode, n_dst, K, N, A;
    const float *w, *b, *g, *beta, *mean, *var;  // w: the layer's weight (heads: fc_probs.weight), b: bias (heads: fc_value.*)
    float *dst;
};
AZ_D double fold_scale(const FoldJob &j, int c) { return (double)j.g[c] / sqrt((double)j.var[c] + BN_EPS); }

__global__ void k_fold(FoldJob j) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.n_dst) return;
    float out = 0.0f;
    switch (j.mode) {
        case FOLD_BIAS: out = (float)(((double)j.b[i] - (double)j.mean[i]) * fold_scale(j, i) + (double)j.beta[i]); break;
        case FOLD_W1F: {  // [k-step s][nt][lane] = W'[oc = nt*16 + (lane&15)][tap = 4s + (lane>>4)], 0 for tap >= 9
            const int lane = i % 64, nt = (i / 64) % 2, sidx = i / 128;
            const int oc = nt * 16 + (lane & 15), t = 4 * sidx + (lane >> 4);
            if (t < 9) out = (float)((double)j.w[oc * 9 + t] * fold_scale(j, oc));
            break;
        }
        case FOLD_W1P: {  // [k-step s][lane] = W'[oc = lane&31][tap = 2s + (lane>>5)], 0 for tap 9
            const int lane = i % 64, sidx = i / 64;
            const int oc = lane & 31, t = 2 * sidx + (lane >> 5);
            if (t < 9) out = (float)((double)j.w[oc * 9 + t] * fold_scale(j, oc));
            break;
        }
        case FOLD_WP: {  // [tap][jj / 4][lane][jj % 4] = W'[oc = lane&31][ic = 2 jj + (lane>>5)][tap]
            const int r = i % 4, lane = (i / 4) % 64, q = (i / 256) % 4, t = i / 1024;
            const int jj = 4 * q + r, oc = lane & 31, ic = 2 * jj + (lane >> 5);
            out = (float)((double)j.w[(oc * NCH + ic) * 9 + t] * fold_scale(j, oc));
            break;
        }
        case FOLD_WQ: {  // fragment f = 2 j8 + nt at [tap][f / 4][lane][f % 4] = W'[oc = nt*16 + (lane&15)][ic = 4 j8 + (lane>>4)][tap]
            const int r = i % 4, lane = (i / 4) % 64, q = (i / 256) % 4, t = i / 1024;
            const int f = 4 * q + r, j8 = f / 2, nt = f % 2, oc = nt * 16 + (lane & 15), ic = 4 * j8 + (lane >> 4);
            out = (float)((double)j.w[(oc * NCH + ic) * 9 + t] * fold_scale(j, oc));
            break;
        }
        case FOLD_WINO: {  // conv2 Winograd fragments (see az_net_commit): [p][j][e / 4][lane][e % 4]
            const int r = i % 4, lane = (i / 4) % 64, q = (i / 256) % 4, jj8 = (i / 1024) % 8, pp = i / 8192;
            const int e = 4 * q + r, f = 8 * pp + e / 2, nt = e % 2, oc = 16 * nt + (lane & 15), ic = 4 * jj8 + (lane >> 4), fi = f / 4, fj = f % 4;
            const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const double sc = fold_scale(j, oc);
            double t3[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const double g0 = (double)j.w[(oc * NCH + ic) * 9 + b] * sc, g1 = (double)j.w[(oc * NCH + ic) * 9 + 3 + b] * sc, g2 = (double)j.w[(oc * NCH + ic) * 9 + 6 + b] * sc;
                t3[b] = (G[fi][0] * g0 + G[fi][1] * g1) + G[fi][2] * g2;
            }
            out = (float)((t3[0] * G[fj][0] + t3[1] * G[fj][1]) + t3[2] * G[fj][2]);
            break;
        }
        case FOLD_WINO32: {  // the same U as 32x32x2 fragments [jq][f][lane][i]
            const int ii = i % 4, lane = (i / 4) % 64, f = (i / 256) % 16, jq = i / 4096;
            const int oc = lane & 31, ic = 2 * (4 * jq + ii) + (lane >> 5), fi = f / 4, fj = f % 4;
            const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const double sc = fold_scale(j, oc);
            double t3[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const double g0 = (double)j.w[(oc * NCH + ic) * 9 + b] * sc, g1 = (double)j.w[(oc * NCH + ic) * 9 + 3 + b] * sc, g2 = (double)j.w[(oc * NCH + ic) * 9 + 6 + b] * sc;
                t3[b] = (G[fi][0] * g0 + G[fi][1] * g1) + G[fi][2] * g2;
            }
            out = (float)((t3[0] * G[fj][0] + t3[1] * G[fj][1]) + t3[2] * G[fj][2]);
            break;
        }
        case FOLD_DENSE_T: {  // dst[k][n] = W[n][k] * s[n]
            const int n = i % j.N, k = i / j.N;
            out = (float)((double)j.w[(size_t)n * j.K + k] * fold_scale(j, n));
            break;
        }
        case FOLD_HEADS_W: {  // dst[k][a] = fc_probs.weight[a][k] | fc_value.weight[k] | 0   (row width N)
            const int a = i % j.N, k = i / j.N;
            out = a < j.A ? j.w[(size_t)a * j.K + k] : (a == j.A ? j.b[k] : 0.0f);
            break;
        }
        case FOLD_HEADS_WQ: {  // the same in 16x16x4 B-fragment order [kb][nt][lane][i]: k = 16 kb + 4 i + (lane >> 4), a = 16 nt + (lane & 15)
            const int ii = i % 4, lane = (i / 4) % 64, NT = j.N / 16, nt = (i / 256) % NT, kb = i / (256 * NT);
            const int k = 16 * kb + 4 * ii + (lane >> 4), a = 16 * nt + (lane & 15);
            out = a < j.A ? j.w[(size_t)a * j.K + k] : (a == j.A ? j.b[k] : 0.0f);
            break;
        }
        default: {  // FOLD_HEADS_B: dst[a] = fc_probs.bias[a] | fc_value.bias | 0
            out = i < j.A ? j.w[i] : (i == j.A ? j.b[0] : 0.0f);
            break;
        }
    }
    j.dst[i] = out;
}

extern "C" int az_net_set_tensor_device(az_net *n, const char *name, const float *d_data, int64_t numel, void *stream) {
    AZ_REQUIRE(n && name && d_data && numel > 0, AZ_EINVAL, "bad arguments");
    auto it = n->raw_dev.find(name);
    if (it == n->raw_dev.end() || it->second.second != (size_t)numel) {
        float *p = nullptr;
        AZ_TRY(net_alloc(n, &p, (size_t)numel));  // a replaced buffer stays in `allocs` until az_net_destroy (shapes never change in practice)
        n->raw_dev[name] = std::make_pair(p, (size_t)numel);
        it = n->raw_dev.find(name);
    }
    AZ_HIP(hipMemcpyAsync(it->second.first, d_data, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    n->committed = false;
    return AZ_OK;
}

static int need_dev(az_net *n, const std::string &k, size_t numel, const float **out) {
    auto it = n->raw_dev.find(k);
    AZ_REQUIRE(it != n->raw_dev.end(), AZ_ESTATE, "missing device tensor '%s'", k.c_str());
    AZ_REQUIRE(it->second.second == numel, AZ_EINVAL, "tensor '%s' has %zu elements, expected %zu", k.c_str(), it->second.second, numel);
    *out = it->second.first;
    return AZ_OK;
}

static int fold_launch(az_net *n, int mode, float *dst, int n_dst, const std::string &wname, size_t wn, const std::string &bname, size_t bn,
                       const std::string &bnname, int C, int K, int N, int A, hipStream_t st) {
    FoldJob j;
    memset(&j, 0, sizeof j);
    j.mode = mode; j.n_dst = n_dst; j.K = K; j.N = N; j.A = A; j.dst = dst;
    if (!wname.empty()) AZ_TRY(need_dev(n, wname, wn, &j.w));
    if (!bname.empty()) AZ_TRY(need_dev(n, bname, bn, &j.b));
    if (!bnname.empty()) {
        AZ_TRY(need_dev(n, bnname + ".weight", C, &j.g)); AZ_TRY(need_dev(n, bnname + ".bias", C, &j.beta));
        AZ_TRY(need_dev(n, bnname + ".running_mean", C, &j.mean)); AZ_TRY(need_dev(n, bnname + ".running_var", C, &j.var));
    }
    hipLaunchKernelGGL(k_fold, dim3((unsigned)((n_dst + 255) / 256)), dim3(256), 0, st, j);
    return AZ_OK;
}

// az_net_commit for tensors set with az_net_set_tensor_device: nothing touches the host
extern "C" int az_net_commit_device(az_net *n, void *stream) {
    AZ_REQUIRE(n, AZ_EINVAL, "null net");
    hipStream_t st = (hipStream_t)stream;
    if (n->game == AZ_TICTACTOE) {
        MlpParams *m = n->mlp_dev;
        AZ_TRY(fold_launch(n, FOLD_DENSE_T, m->f1w, 81, "fc1.weight", 81, "", 0, "bn1", 9, 9, 9, 0, st));
        AZ_TRY(fold_launch(n, FOLD_BIAS, m->f1b, 9, "", 0, "fc1.bias", 9, "bn1", 9, 0, 0, 0, st));
        AZ_TRY(fold_launch(n, FOLD_DENSE_T, m->f2w, 81, "fc2.weight", 81, "", 0, "bn2", 9, 9, 9, 0, st));
        AZ_TRY(fold_launch(n, FOLD_BIAS, m->f2b, 9, "", 0, "fc2.bias", 9, "bn2", 9, 0, 0, 0, st));
        AZ_TRY(fold_launch(n, FOLD_HEADS_W, m->hw, 90, "fc_probs.weight", 81, "fc_value.weight", 9, "", 0, 9, 10, 9, st));
        AZ_TRY(fold_launch(n, FOLD_HEADS_B, m->hb, 10, "fc_probs.bias", 9, "fc_value.bias", 1, "", 0, 0, 10, 9, st));
        AZ_HIP(hipGetLastError());
        n->committed = true;
        return AZ_OK;
    }
    for (int l = 0; l < 4; ++l) {
        const int IC = l == 0 ? 1 : NCH;
        const std::string cn = "conv" + std::to_string(l + 1), bnn = "bn" + std::to_string(l + 1);
        const size_t wn = (size_t)NCH * IC * 9;
        if (l == 0) {
            AZ_TRY(fold_launch(n, FOLD_BIAS, (float *)n->tp.b1, NCH, "", 0, cn + ".bias", NCH, bnn, NCH, 0, 0, 0, st));
            AZ_TRY(fold_launch(n, FOLD_W1F, (float *)n->tp.w1f, 3 * 2 * 64, cn + ".weight", wn, "", 0, bnn, NCH, 0, 0, 0, st));
            AZ_TRY(fold_launch(n, FOLD_W1P, (float *)n->tp.w1p, 5 * 64, cn + ".weight", wn, "", 0, bnn, NCH, 0, 0, 0, st));
        } else {
            AZ_TRY(fold_launch(n, FOLD_BIAS, (float *)n->tp.cb[l - 1], NCH, "", 0, cn + ".bias", NCH, bnn, NCH, 0, 0, 0, st));
            AZ_TRY(fold_launch(n, FOLD_WP, (float *)n->tp.wp[l - 1], 9 * 16 * 64, cn + ".weight", wn, "", 0, bnn, NCH, 0, 0, 0, st));
            AZ_TRY(fold_launch(n, FOLD_WQ, (float *)n->tp.wq[l - 1], 9 * 16 * 64, cn + ".weight", wn, "", 0, bnn, NCH, 0, 0, 0, st));
            if (l == 1) AZ_TRY(fold_launch(n, FOLD_WINO, (float *)n->tp.wu, 2 * 8 * 4 * 64 * 4, cn + ".weight", wn, "", 0, bnn, NCH, 0, 0, 0, st));
            if (l == 1) AZ_TRY(fold_launch(n, FOLD_WINO32, (float *)n->tp.wu32, 4 * 16 * 64 * 4, cn + ".weight", wn, "", 0, bnn, NCH, 0, 0, 0, st));
        }
    }
    AZ_TRY(fold_launch(n, FOLD_DENSE_T, n->fc1w, n->FIN * n->F1, "fc1.weight", (size_t)n->F1 * n->FIN, "", 0, "fc_bn1", n->F1, n->FIN, n->F1, 0, st));
    AZ_TRY(fold_launch(n, FOLD_BIAS, n->fc1b, n->F1, "", 0, "fc1.bias", n->F1, "fc_bn1", n->F1, 0, 0, 0, st));
    AZ_TRY(retile_q(n->fc1w, n->fc1wq, n->FIN, n->F1, st));
        AZ_TRY(quant_w(n, 1, st));
    AZ_TRY(fold_launch(n, FOLD_DENSE_T, n->fc2w, n->F1 * n->F2, "fc2.weight", (size_t)n->F2 * n->F1, "", 0, "fc_bn2", n->F2, n->F1, n->F2, 0, st));
    AZ_TRY(fold_launch(n, FOLD_BIAS, n->fc2b, n->F2, "", 0, "fc2.bias", n->F2, "fc_bn2", n->F2, 0, 0, 0, st));
    AZ_TRY(retile_q(n->fc2w, n->fc2wq, n->F1, n->F2, st));
        AZ_TRY(quant_w(n, 2, st));
    AZ_TRY(fold_launch(n, FOLD_HEADS_W, n->hw, n->F2 * n->NH, "fc_probs.weight", (size_t)n->A * n->F2, "fc_value.weight", n->F2, "", 0, n->F2, n->NH, n->A, st));
    AZ_TRY(fold_launch(n, FOLD_HEADS_B, n->hb, n->NH, "fc_probs.bias", n->A, "fc_value.bias", 1, "", 0, 0, n->NH, n->A, st));
    if (n->F2 % 16 == 0)
        AZ_TRY(fold_launch(n, FOLD_HEADS_WQ, n->hwq, n->F2 * n->NH, "fc_probs.weight", (size_t)n->A * n->F2, "fc_value.weight", n->F2, "", 0, n->F2, n->NH, n->A, st));
    AZ_HIP(hipGetLastError());
    n->committed = true;
    return AZ_OK;
}

// conv2 in the Winograd F(2x2,3x3) form (2.25x fewer multiplications; exact to the same 5e-7 as the direct form against the
// reference's torch forward; bit-equal to its restatement in the oracle, which reads the same switch AZ_WINOGRAD):
//   * 8x8 planes: ON.  From 4096 boards up both boards of a wave form one 32-row MFMA tile (conv2_wino32) in the
//     one-wave-per-SIMD variant of k_trunk2: 512 registers hold all 16 frequency accumulators, the LDS the other four waves
//     would use holds the transformed weights.  MI355X: 473 vs 526 us at 32768 boards, 71 vs 80 us at 4096.  Below 4096
//     boards k_trunk runs the 16-row form (conv2_wino; same chains, same bits): 21 vs 23 us at 1024 boards.
//   * 7x6 planes (Connect4): ON, in the same 32-row one-wave-per-SIMD form (12 tiles per board, 24 of the 32 rows of the MFMA tile):
//     100 vs 105.5 us at 8192 boards, 52.9 vs 54.8 at 4096; equal below 4096 boards (16-row form in k_trunk).  The 16-row form in
//     the two-waves-per-SIMD kernel had not paid (103 vs 101 us: with 16-row tiles every 32-cycle MFMA needs an operand transformed
//     by the wave itself and a 256-byte weight fragment -- the instruction stream, not the matrix pipe, is the limit).
//   * 6x6 planes: never (9 tiles per board would leave the MFMA tile 44 % empty).
// AZ_WINOGRAD=0 switches everything back to the direct form.
static bool use_wino(int CH, int CW) {
    static int mode = -2;
    if (mode == -2) { const char *e = getenv("AZ_WINOGRAD"); mode = e ? atoi(e) : -1; }
    if (mode == 0) return false;
    return (CH == 8 && CW == 8) || (CH == 7 && CW == 6);  // AZ_WINOGRAD=0: direct form everywhere (the oracle follows the same variable)
}

// Every forward kernel is launched through AZ_LAUNCH.  Under az_net_profile the launch carries a start and a stop event
// (hipExtLaunchKernelGGL): they take the dispatch's own begin / end timestamps -- the two numbers rocprofv3 reports a kernel's duration
// from -- so a slot's mean IS that kernel's average duration, with nothing to calibrate away (events recorded BETWEEN launches measure
// marker to marker: kernel + 3-5 us of dispatch and marker processing).  Otherwise (and inside a graph capture) it is a plain launch.
// The state is per host thread: az_net_profile brackets run_stage on the calling thread, and two networks forwarding from two threads
// (an arena's two engines, a trainer beside a search) must not see each other's events or small-batch bookkeeping.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local int g_ev_used = 0;
#define AZ_LAUNCH_EV(ev0, ev1, kern, grid, block, lds, st, ...)                                                              \
    do {                                                                                                                     \
        if (g_ev_start) { hipExtLaunchKernelGGL(kern, grid, block, lds, st, ev0, ev1, 0, __VA_ARGS__); g_ev_used = 1; }      \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                    \
    } while (0)
#define AZ_LAUNCH(kern, grid, block, lds, st, ...) AZ_LAUNCH_EV(g_ev_start, g_ev_stop, kern, grid, block, lds, st, __VA_ARGS__)
// a stage of TWO launches (k_q_rows + k_qgemm): the first carries the stage's start event, the second its stop event -- the slot then
// holds first begin -> second end, the stage as the engine pays for it
static thread_local int g_ev_split = 0;
#define AZ_LAUNCH_FIRST(kern, grid, block, lds, st, ...)                                                                     \
    do { AZ_LAUNCH_EV(g_ev_start, nullptr, kern, grid, block, lds, st, __VA_ARGS__); g_ev_split = 1; } while (0)
#define AZ_LAUNCH_MAYBE_LAST(kern, grid, block, lds, st, ...)                                                                \
    do { AZ_LAUNCH_EV(g_ev_split ? nullptr : g_ev_start, g_ev_stop, kern, grid, block, lds, st, __VA_ARGS__); g_ev_split = 0; } while (0)

template <int CH, int CW, bool WINO>
static int launch_trunk2(az_net *n, const float *in, int B, const int *dyn, hipStream_t st) {
    using G = TrunkGeom<CH, CW>;
    // one workgroup per CU, two boards per wave: two waves per SIMD -- or, for the Winograd conv2 (8x8 and 7x6 planes), ONE wave per
    // SIMD with 512 registers (all 16 frequency accumulators live) and the transformed weights in the LDS the other four waves would use
    constexpr int WPB = WINO ? 4 : 8;
    constexpr int lds_planes = 64 + WPB * 2 * G::WAVE_FLOATS * 4;
    constexpr int lds_bytes = lds_planes + ((WINO && lds_planes + 65536 <= 160 * 1024) ? 65536 : 0);  // + the Winograd U fragments where they fit
    static_assert(lds_bytes <= 160 * 1024, "k_trunk2 workgroup does not fit the CU's LDS");
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t pr;
        int dev = 0;
        AZ_HIP(hipGetDevice(&dev));
        AZ_HIP(hipGetDeviceProperties(&pr, dev));
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trunk2<CH, CW, WPB, WINO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        n_cu = pr.multiProcessorCount;
    }
    const int pairs = (B + 1) / 2, want = (pairs + 3) / 4;
    AZ_LAUNCH((k_trunk2<CH, CW, WPB, WINO>), dim3(want < n_cu ? want : n_cu), dim3(64 * WPB), lds_bytes, st, in, B, dyn, n->tp, n->feat);
    return AZ_OK;
}

static bool trunk_v1() { static int v = -1; if (v < 0) { const char *e = getenv("AZ_TRUNK_V1"); v = (e && atoi(e)) ? 1 : 0; } return v == 1; }

// boards up to which the trunk runs four waves per board (AZ_TRUNK_Q_MAX; 0: never)
static int trunk_q_max() { static int v = -1; if (v < 0) { const char *e = getenv("AZ_TRUNK_Q_MAX"); v = e ? atoi(e) : 512; } return v; }

template <int CH, int CW, bool WINO>
static int launch_trunk(az_net *n, const float *in, int B, const int *dyn, hipStream_t st) {
    // two boards per wave on 32x32x2 pays from ~4096 boards up (16384: 325 vs 331 us); below that the one-board-
    // per-wave kernel fills the chip better (2048: 45 vs 77 us).  AZ_TRUNK_V1=1 forces the latter.
    n->last_trunk_two_boards = (!trunk_v1() && B >= 4096) ? 1 : 0;
    n->last_trunk_q = 0;
    if (n->last_trunk_two_boards) return launch_trunk2<CH, CW, WINO>(n, in, B, dyn, st);
    if (B <= trunk_q_max()) {  // few boards: four waves per board (k_trunk_q)
        n->last_trunk_q = 1;
        static int nw_env = -1;
        if (nw_env < 0) { const char *e = getenv("AZ_TRUNK_Q_WAVES"); nw_env = e ? atoi(e) : 0; }
        const int nw = nw_env ? nw_env : (B <= 256 ? 8 : 4);  // eight waves per board while that leaves a SIMD at most two (11.7 vs 13.2 us; 512 boards: 23.0 vs 18.6)
        if (nw == 8) AZ_LAUNCH((k_trunk_q<CH, CW, WINO, 8>), dim3((unsigned)B), dim3(512), 0, st, in, B, dyn, n->tp, n->feat);
        else AZ_LAUNCH((k_trunk_q<CH, CW, WINO, 4>), dim3((unsigned)B), dim3(256), 0, st, in, B, dyn, n->tp, n->feat);
        return AZ_OK;
    }
    using G = TrunkGeom<CH, CW>;
    static bool attr_set = false;
    static int lds_bytes = G::LDS_BYTES;
    if (!attr_set) {
        // resident blocks per CU = 160 KB / lds_bytes.  Measured on MI355X (4096 boards): 3 or 2 blocks per CU
        // 88 us, 4 blocks 94 us, 1 block 92 us -> pad the request to 3; AZ_TRUNK_BLOCKS_PER_CU overrides.
        const char *e = getenv("AZ_TRUNK_BLOCKS_PER_CU");
        int want = e ? atoi(e) : 3;
        if (want >= 1 && want <= 8 && 160 * 1024 / want > G::LDS_BYTES) lds_bytes = (160 * 1024 / want) & ~15;
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trunk<CH, CW, WINO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    AZ_LAUNCH((k_trunk<CH, CW, WINO>), dim3((B + 3) / 4), dim3(256), lds_bytes, st, in, B, dyn, n->tp, n->feat);
    return AZ_OK;
}

template <int BM, int BN, int WM, int WN, int KT>
static int gemm_launch(const float *A, const float *Bw, const float *bias, float *C, int M, int N, int K, bool relu, const int *dyn, hipStream_t st) {
    constexpr int lds = gemm_lds_bytes<BM, BN>();
    static bool attr_set = false;
    if (!attr_set) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm<BM, BN, WM, WN, true, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm<BM, BN, WM, WN, false, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    dim3 grid(N / BN, (M + BM - 1) / BM);
    if (relu) AZ_LAUNCH((k_gemm<BM, BN, WM, WN, true, KT>), grid, dim3(256), lds, st, A, Bw, bias, C, M, N, K, dyn);
    else AZ_LAUNCH((k_gemm<BM, BN, WM, WN, false, KT>), grid, dim3(256), lds, st, A, Bw, bias, C, M, N, K, dyn);
    return AZ_OK;
}

// K of the networks' dense layers gets an unrolled instantiation: 512 / 1024 (Othello 8x8), 128 (Othello 6x6),
// 192 / 64 (Connect4); anything else runs the runtime loop
template <int BM, int BN, int WM, int WN>
static int gemm_go(const float *A, const float *Bw, const float *bias, float *C, int M, int N, int K, bool relu, const int *dyn, hipStream_t st) {
    switch (K) {
        case 1024: return gemm_launch<BM, BN, WM, WN, 32>(A, Bw, bias, C, M, N, K, relu, dyn, st);
        case 512: return gemm_launch<BM, BN, WM, WN, 16>(A, Bw, bias, C, M, N, K, relu, dyn, st);
        case 192: return gemm_launch<BM, BN, WM, WN, 6>(A, Bw, bias, C, M, N, K, relu, dyn, st);
        case 128: return gemm_launch<BM, BN, WM, WN, 4>(A, Bw, bias, C, M, N, K, relu, dyn, st);
        case 64: return gemm_launch<BM, BN, WM, WN, 2>(A, Bw, bias, C, M, N, K, relu, dyn, st);
        default: return gemm_launch<BM, BN, WM, WN, 0>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    }
}

template <int TM, int TN>
static int solo_t_launch(const float *A, const float *Bw, const float *bias, float *C, int M, int N, int K, bool relu, const int *dyn, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN, lds = 4 * (2 * BM * 33 + 2 * 32 * BN);
    static bool attr_set = false;
    if (!attr_set) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_solo_t<true, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_solo_t<false, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    dim3 grid(N / BN, (M + BM - 1) / BM);
    if (relu) AZ_LAUNCH((k_gemm_solo_t<true, TM, TN>), grid, dim3(256), lds, st, A, Bw, bias, C, M, N, K, dyn);
    else AZ_LAUNCH((k_gemm_solo_t<false, TM, TN>), grid, dim3(256), lds, st, A, Bw, bias, C, M, N, K, dyn);
    return AZ_OK;
}

// which kernel a dense layer of M rows runs on (one place: launch_gemm dispatches on it, az_net_stage_kernel reports it)
enum GemmKind { GK_SMALL, GK_FRAG, GK_SOLO, GK_SOLO_T, GK_TILED };
static int gemm_solo_env() { static int solo = -2; if (solo == -2) { const char *e = getenv("AZ_GEMM_SOLO"); solo = e ? atoi(e) : -1; } return solo; }
// rows up to which k_dense_frag runs a layer.  Measured on OthelloNet 8x8 (us, k_dense_frag | what ran before: k_dense_small up to 128 / 256
// rows, k_gemm above): fc1 (K = 512) 6.6 | 8.7 at 1 row, 7.3 | 21.0 at 256, 9.9 | 20.9 at 512, 16.3 | 21.2 at 1024, 22.3 | 21.0 at 1536;
// fc2 (K = 1024) 11.2 | 15.9 at 1 row, 11.7 | 24.1 at 256, 11.9 | 39.8 at 512, 17.5 | 39.8 at 1024, 28.2 | 40.0 at 1536, 34.5 | 40.0 at 2048,
// 65 | 37 at 4096 (every 16 rows stream the whole weight matrix from L2: ~9 TB/s from 1024 rows up, the kernel's bound there).
// AZ_DENSE_FRAG_MAX overrides both limits; 0 switches the kernel off (the kernels of the rounds before).
static int frag_max_rows(int K) {
    static int v = -2;
    if (v == -2) { const char *e = getenv("AZ_DENSE_FRAG_MAX"); v = e ? atoi(e) : -1; }
    return v >= 0 ? v : (K >= 1024 ? 2048 : 1024);
}
static GemmKind gemm_kind(int M, int N, int K, bool have_q = true) {
    if (have_q && M <= frag_max_rows(K) && frag_shape(N, K)) return GK_FRAG;  // latency-bound row counts: the shortest accumulation chain
    // measured crossover against the tiled GEMM (MI355X): K = 512 up to 128 rows (15 vs 21 us), K = 1024 up to 256 rows (28 vs 40 us)
    if (M <= (K >= 1024 ? 256 : 128) && K % 32 == 0) return GK_SMALL;  // few rows: latency matters, not throughput
    // large row counts: the one-wave-per-SIMD kernel (256x256 workgroup tiles), from one tile per CU up.  AZ_GEMM_SOLO=0 / 1 forces.
    const int solo = gemm_solo_env();
    const long long tiles = (long long)((M + 255) / 256) * (N / 256);
    if (solo != 0 && N % 256 == 0 && (solo == 1 || tiles >= 256)) return GK_SOLO;
    // below that: the same scheme on 128x128 tiles from TWO tiles per CU up (fc1 from 8192 rows, fc2 from 16384; AZ_GEMM_SOLO=2
    // forces it).  Measured against k_gemm: 76 vs 80 us (fc1, 8192 rows), 137 vs 144 (fc2, 16384); with ONE tile per CU it loses
    // (4096 rows: 42.6 vs 41.4 us, 64x128 tiles 57 vs 42): a 4096-cycle K tile does not carry its barrier and LDS turnaround
    // without a partner wave, and k_gemm's two blocks per CU are exactly that partner.
    if (solo != 0 && solo != 1 && N % 128 == 0) {
        const long long t128 = (long long)((M + 127) / 128) * (N / 128);
        if (solo == 2 || (solo < 0 && t128 >= 512)) return GK_SOLO_T;
    }
    return GK_TILED;
}

static int frag_go(const float *A, const float *Bq, const float *bias, float *C, int M, int N, int K, bool relu, const int *dyn, hipStream_t st) {
    constexpr int NT = 4;
    const int lds = 4 * 16 * (K + 4);
    static int attr_lds = 0;
    if (lds > attr_lds) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense_frag<true, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense_frag<false, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    dim3 grid((unsigned)(N / (16 * NT)), (unsigned)((M + 15) / 16));
    if (relu) AZ_LAUNCH((k_dense_frag<true, NT>), grid, dim3(64 * NT), lds, st, A, Bq, bias, C, M, N, K, dyn);
    else AZ_LAUNCH((k_dense_frag<false, NT>), grid, dim3(64 * NT), lds, st, A, Bq, bias, C, M, N, K, dyn);
    return AZ_OK;
}

static thread_local int g_last_gemm_small = 0;  // set by launch_gemm when it served the layer with a small-batch kernel (the profiler books those apart)

static int launch_gemm(const float *A, const float *Bw, const float *Bq, const float *bias, float *C, int M, int N, int K, bool relu, const int *dyn, hipStream_t st) {
    AZ_REQUIRE(K % 32 == 0, AZ_EINVAL, "GEMM K=%d is not a multiple of 32", K);
    switch (gemm_kind(M, N, K, Bq != nullptr)) {
        case GK_FRAG: g_last_gemm_small = 1; return frag_go(A, Bq, bias, C, M, N, K, relu, dyn, st);
        case GK_SMALL: {
            g_last_gemm_small = 1;
            dim3 grid((unsigned)((N + 63) / 64), (unsigned)M);
            if (relu) AZ_LAUNCH((k_dense_small<true>), grid, dim3(64), 0, st, A, Bw, bias, C, M, N, K, dyn);
            else AZ_LAUNCH((k_dense_small<false>), grid, dim3(64), 0, st, A, Bw, bias, C, M, N, K, dyn);
            return AZ_OK;
        }
        case GK_SOLO: {
            constexpr int lds = 4 * (2 * 256 * 33 + 2 * 32 * 256);
            static bool attr_set = false;
            if (!attr_set) {
                AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_solo<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_solo<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                attr_set = true;
            }
            dim3 grid(N / 256, (M + 255) / 256);
            if (relu) AZ_LAUNCH((k_gemm_solo<true>), grid, dim3(256), lds, st, A, Bw, bias, C, M, N, K, dyn);
            else AZ_LAUNCH((k_gemm_solo<false>), grid, dim3(256), lds, st, A, Bw, bias, C, M, N, K, dyn);
            return AZ_OK;
        }
        case GK_SOLO_T: return solo_t_launch<2, 2>(A, Bw, bias, C, M, N, K, relu, dyn, st);
        default: break;
    }
    // pick the largest tile that still gives every CU two resident blocks (512 blocks on 256 CUs):
    // a block's barrier and LDS-fill phases then overlap the other block's MFMAs
    const long long mb128 = (M + 127) / 128, mb64 = (M + 63) / 64;
    static int force = -1;
    if (force < 0) { const char *e = getenv("AZ_GEMM_TILE"); force = e ? atoi(e) : 0; }
    if (force == 1 && N % 128 == 0) return gemm_go<128, 128, 64, 64>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (force == 2 && N % 64 == 0) return gemm_go<128, 64, 32, 64>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (force == 3 && N % 64 == 0) return gemm_go<64, 64, 32, 32>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (force == 4 && N % 128 == 0) return gemm_go<64, 128, 32, 64>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (N % 128 == 0 && mb128 * (N / 128) >= 512) return gemm_go<128, 128, 64, 64>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (N % 64 == 0 && mb128 * (N / 64) >= 512) return gemm_go<128, 64, 32, 64>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (N % 64 == 0 && mb64 * (N / 64) >= 512) return gemm_go<64, 64, 32, 32>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (N % 128 == 0) return gemm_go<64, 128, 32, 64>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (N % 64 == 0) return gemm_go<64, 64, 32, 32>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    if (N % 32 == 0) return gemm_go<128, 32, 32, 32>(A, Bw, bias, C, M, N, K, relu, dyn, st);
    az_set_error("GEMM N=%d is not a multiple of 32", N);
    return AZ_EINVAL;
}

#define QD_SMALL_MAX 512  // rows up to which the one-launch small-batch kernel serves a fixed-point dense layer
static int qd_small_max() {  // AZ_QD_SMALL_MAX overrides the row limit (0: never); launch_qdense and az_net_stage_kernel both ask here
    static int v = -2;
    if (v == -2) { const char *e = getenv("AZ_QD_SMALL_MAX"); v = e ? atoi(e) : QD_SMALL_MAX; }
    return v;
}
static bool qd_small_serves(int B, int N, int K) { return B <= qd_small_max() && N % 64 == 0 && (K == 128 || K == 512 || K == 1024); }

template <int NWM, int NWN, int WM, int WN>
static int qgemm_go(az_net *n, int layer, int B, const int *dyn, hipStream_t st) {
    constexpr int lds = qgemm_lds_bytes<NWM, NWN, WM, WN>(), BM = NWM * WM * 32, BN = NWN * WN * 32;
    static_assert(lds <= 160 * 1024, "k_qgemm does not fit the CU's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_qgemm<NWM, NWN, WM, WN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int K = layer == 1 ? n->FIN : n->F1, N = layer == 1 ? n->F1 : n->F2;
    AZ_LAUNCH_MAYBE_LAST((k_qgemm<NWM, NWN, WM, WN, true>), dim3((unsigned)(N / BN), (unsigned)((B + BM - 1) / BM)), dim3(64 * NWM * NWN), lds, st, n->qd_a,
              n->qd_rows * (size_t)K, n->qd_ea, layer == 1 ? n->qd_w1 : n->qd_w2, (size_t)N * K, layer == 1 ? n->qd_e1 : n->qd_e2, layer == 1 ? n->fc1b : n->fc2b,
              layer == 1 ? n->h1 : n->h2, B, N, K, dyn);
    return AZ_OK;
}

// fc1 / fc2 in the exact block-fixed-point form: quantise the rows of the layer's input, then the int8 GEMM
static int launch_qdense(az_net *n, int layer, int B, const int *dyn, hipStream_t st) {
    const int K = layer == 1 ? n->FIN : n->F1, N = layer == 1 ? n->F1 : n->F2;
    if (qd_small_serves(B, N, K)) {
        const float *x = layer == 1 ? n->feat : n->h1;
        const int8_t *w = layer == 1 ? n->qd_w1 : n->qd_w2;
        const int *e = layer == 1 ? n->qd_e1 : n->qd_e2;
        const float *bias = layer == 1 ? n->fc1b : n->fc2b;
        float *out = layer == 1 ? n->h1 : n->h2;
        const dim3 grid((unsigned)(N / 64), (unsigned)((B + 15) / 16));
        g_last_gemm_small = 1;
        if (K == 128) AZ_LAUNCH((k_qdense_small<128, true>), grid, dim3(256), 0, st, x, w, (size_t)N * K, e, bias, out, B, N, dyn);
        else if (K == 512) AZ_LAUNCH((k_qdense_small<512, true>), grid, dim3(256), 0, st, x, w, (size_t)N * K, e, bias, out, B, N, dyn);
        else AZ_LAUNCH((k_qdense_small<1024, true>), grid, dim3(256), 0, st, x, w, (size_t)N * K, e, bias, out, B, N, dyn);
        return AZ_OK;
    }
    AZ_LAUNCH_FIRST(k_q_rows, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, layer == 1 ? n->feat : n->h1, B, K, dyn, n->qd_a, n->qd_rows * (size_t)K, n->qd_ea);
    static int cfg = -1;  // AZ_QG_CFG: tile plan for A/B runs (every plan gives the same bits)
    if (cfg < 0) { const char *e = getenv("AZ_QG_CFG"); cfg = e ? atoi(e) : 0; }
    if (cfg == 1) return qgemm_go<2, 2, 1, 2>(n, layer, B, dyn, st);  // 64 x 128, four waves, two workgroups per CU
    if (cfg == 2) return qgemm_go<2, 2, 1, 1>(n, layer, B, dyn, st);  // 64 x 64, four waves
    if (cfg == 3) return qgemm_go<4, 2, 1, 1>(n, layer, B, dyn, st);  // 128 x 64, eight waves
    if (cfg == 4) return qgemm_go<4, 2, 1, 2>(n, layer, B, dyn, st);  // 128 x 128, eight waves
    if (cfg == 5) return qgemm_go<2, 4, 1, 1>(n, layer, B, dyn, st);  // 64 x 128, eight waves
    // 128 x 128 tiles from one workgroup per CU up, 64 x 128 below
    if ((long long)((B + 127) / 128) * (N / 128) >= 256) return qgemm_go<4, 2, 1, 2>(n, layer, B, dyn, st);
    return qgemm_go<2, 4, 1, 1>(n, layer, B, dyn, st);
}

template <int NT>
static int heads_go(az_net *n, int B, float *probs, float *value, const int *dyn, hipStream_t st) {
    constexpr int lds = heads_lds_bytes<NT>();
    static bool attr_set = false;
    if (!attr_set) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_heads<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    AZ_LAUNCH((k_heads<NT>), dim3((B + 31) / 32), dim3(256), lds, st, n->h2, n->hw, n->hb, B, n->F2, n->A, probs, value, dyn);
    return AZ_OK;
}

template <int NT, int RH>
static int heads2_go(az_net *n, int B, float *probs, float *value, const int *dyn, hipStream_t st) {
    constexpr int lds = heads2_lds_bytes<NT, RH, 512>();
    static_assert(lds <= 160 * 1024, "k_heads2 does not fit the CU's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_heads2<NT, RH, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    AZ_LAUNCH((k_heads2<NT, RH, 512>), dim3((unsigned)((B + 16 * RH - 1) / (16 * RH))), dim3(64 * NT), lds, st, n->h2, n->hwq, n->hb, B, n->A,
                       probs, value, dyn);
    return AZ_OK;
}

// rows up to which the row-per-workgroup heads kernel runs: 128 where k_heads2 has no instantiation; never for OthelloNet (F2 = 512), whose
// k_heads2 is faster at every size (8.6 vs 9.8 us at one row, 9.6 vs 10.3 at 128).  AZ_HEADS_SMALL_MAX forces a limit for A/B runs.
static int heads_small_max(const az_net *n) {
    static int v = -2;
    if (v == -2) { const char *e = getenv("AZ_HEADS_SMALL_MAX"); v = e ? atoi(e) : -1; }
    if (v >= 0) return v;
    return (n->F2 == 512 && (n->NH == 80 || n->NH == 48)) ? 0 : 128;
}

static int launch_heads(az_net *n, int B, float *probs, float *value, const int *dyn, hipStream_t st) {
    if (B <= heads_small_max(n) && n->NH <= 128 && n->F2 % 32 == 0) {  // few rows: latency, not throughput
        AZ_LAUNCH(k_heads_small, dim3((unsigned)B), dim3(128), 0, st, n->h2, n->hw, n->hb, B, n->F2, n->A, n->NH, probs, value, dyn);
        return AZ_OK;
    }
    static int v1 = -1;
    if (v1 < 0) { const char *e = getenv("AZ_HEADS_V1"); v1 = (e && atoi(e)) ? 1 : 0; }  // the round-1 kernel, for A/B runs
    static int rh = -1;
    if (rh < 0) { const char *e = getenv("AZ_HEADS_RH"); rh = e ? atoi(e) : 0; }  // tuning: force 16-row (1) / 32-row (2) blocks
    if (!v1 && n->F2 == 512) {  // OthelloNet.  Measured (us, 16-row / 32-row blocks; round-1 kernel): 4096 rows 9.9 / 14.5 / 15.5,
        // 8192: 13.6 / 15.0, 16384: 20.3 / 21.4, 32768: 39.0 / 55.7 / 44.4 -> 16-row blocks (more waves in flight) everywhere
        const bool two = rh == 2;
        if (n->NH == 80) return two ? heads2_go<5, 2>(n, B, probs, value, dyn, st) : heads2_go<5, 1>(n, B, probs, value, dyn, st);
        if (n->NH == 48) return two ? heads2_go<3, 2>(n, B, probs, value, dyn, st) : heads2_go<3, 1>(n, B, probs, value, dyn, st);
    }
    switch (n->NH / 16) {
        case 1: return heads_go<1>(n, B, probs, value, dyn, st);
        case 3: return heads_go<3>(n, B, probs, value, dyn, st);
        case 5: return heads_go<5>(n, B, probs, value, dyn, st);
        default: az_set_error("no heads kernel for padded width %d", n->NH); return AZ_EINVAL;
    }
}

// AZ_TAIL_V1=1: the fma kernel of the rounds before (k_tail_small) instead of k_tail_mfma, for A/B runs
static bool tail_v1() { static int v = -1; if (v < 0) { const char *e = getenv("AZ_TAIL_V1"); v = (e && atoi(e)) ? 1 : 0; } return v == 1; }

// fc1 + fc2 + heads as ONE launch where the dense layers are small (Connect4Net); returns false when no instantiation fits
static bool tail_is_fused(const az_net *n) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("AZ_NO_FUSED_TAIL"); off = (e && atoi(e)) ? 1 : 0; }
    return !off && n->F1 == 64 && n->F2 == 32 && n->NH == 16 && n->FIN == 192 && n->A == 7;  // Connect4 6x7 (8x8: 131 KB of fc1 weights, not worth restaging)
}

template <int FIN, int A, int R>
static int tail_go(az_net *n, int B, float *probs, float *value, const int *dyn, hipStream_t st) {
    constexpr int lds = 4 * tail_lds_floats<FIN, 64, 32, 16, R>();
    static_assert(lds <= 160 * 1024, "k_tail_small does not fit the CU's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail_small<FIN, 64, 32, 16, A, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    AZ_LAUNCH((k_tail_small<FIN, 64, 32, 16, A, R>), dim3((unsigned)((B + 4 * R - 1) / (4 * R))), dim3(256), lds, st, n->feat, n->fc1w,
                       n->fc1b, n->fc2w, n->fc2b, n->hw, n->hb, B, probs, value, dyn);
    return AZ_OK;
}

static int launch_tail(az_net *n, int B, float *probs, float *value, const int *dyn, hipStream_t st) {
    if (!tail_v1()) {
        AZ_LAUNCH((k_tail_mfma<192, 64, 32, 16, 7>), dim3((unsigned)((B + 15) / 16)), dim3(256), 0, st, n->feat, n->fc1w, n->fc1b, n->fc2w,
                           n->fc2b, n->hw, n->hb, B, probs, value, dyn);
        return AZ_OK;
    }
    static int r8 = -1;
    if (r8 < 0) { const char *e = getenv("AZ_TAIL_R8_FROM"); r8 = e ? atoi(e) : 0x7fffffff; }
    // 16 rows per workgroup (measured: 9.3 us up to 4096 rows, 12.7 us at 8192; 32 rows per workgroup: 19 us at every size)
    return B >= r8 ? tail_go<192, 7, 8>(n, B, probs, value, dyn, st) : tail_go<192, 7, 4>(n, B, probs, value, dyn, st);
}

// the other planes between 5x5 and 8x8: k_trunk (one board per wave, direct form) at every batch size
template <int CH, int CW>
static int launch_trunk_plain(az_net *n, const float *in, int B, const int *dyn, hipStream_t st) {
    using G = TrunkGeom<CH, CW>;
    static bool attr_set = false;
    static int lds_bytes = G::LDS_BYTES;
    if (!attr_set) {
        if (160 * 1024 / 3 > G::LDS_BYTES) lds_bytes = (160 * 1024 / 3) & ~15;  // three blocks per CU, as launch_trunk
        AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trunk<CH, CW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    n->last_trunk_two_boards = 0;
    AZ_LAUNCH((k_trunk<CH, CW, false>), dim3((B + 3) / 4), dim3(256), lds_bytes, st, in, B, dyn, n->tp, n->feat);
    return AZ_OK;
}

static int launch_trunk_other(az_net *n, const float *in, int B, const int *dyn, hipStream_t st) {
#define PLANE(ch, cw) if (n->CH == ch && n->CW == cw) return launch_trunk_plain<ch, cw>(n, in, B, dyn, st);
    PLANE(5, 5) PLANE(5, 6) PLANE(5, 7) PLANE(5, 8) PLANE(6, 5) PLANE(6, 7) PLANE(6, 8) PLANE(7, 5) PLANE(7, 7) PLANE(7, 8) PLANE(8, 5) PLANE(8, 6) PLANE(8, 7)
#undef PLANE
    az_set_error("no conv-trunk kernel instantiated for a %dx%d plane", n->CH, n->CW);
    return AZ_EINVAL;
}

static int run_stage(az_net *n, int stage, const float *d_input, int B, const int *dyn, float *d_probs, float *d_value, hipStream_t st) {
    if (stage >= 1 && tail_is_fused(n)) return stage == 1 ? launch_tail(n, B, d_probs, d_value, dyn, st) : AZ_OK;  // stages 2, 3 ran inside stage 1
    switch (stage) {
        case 0:
            if (n->CH == 8 && n->CW == 8) return use_wino(8, 8) ? launch_trunk<8, 8, true>(n, d_input, B, dyn, st) : launch_trunk<8, 8, false>(n, d_input, B, dyn, st);
            if (n->CH == 6 && n->CW == 6) return launch_trunk<6, 6, false>(n, d_input, B, dyn, st);
            if (n->CH == 7 && n->CW == 6) return use_wino(7, 6) ? launch_trunk<7, 6, true>(n, d_input, B, dyn, st) : launch_trunk<7, 6, false>(n, d_input, B, dyn, st);
            return launch_trunk_other(n, d_input, B, dyn, st);
        case 1: return n->qd_on ? launch_qdense(n, 1, B, dyn, st) : launch_gemm(n->feat, n->fc1w, n->fc1wq, n->fc1b, n->h1, B, n->F1, n->FIN, true, dyn, st);
        case 2: return n->qd_on ? launch_qdense(n, 2, B, dyn, st) : launch_gemm(n->h1, n->fc2w, n->fc2wq, n->fc2b, n->h2, B, n->F2, n->F1, true, dyn, st);
        default: return launch_heads(n, B, d_probs, d_value, dyn, st);
    }
}

static int prof_harvest(az_net *n) {
    if (n->prof_used == 0) return AZ_OK;
    for (int i = n->prof_used - 1; i >= 0; --i) {  // wait for the last launch that carries events
        int last = -1;
        for (int s = 3; s >= 0 && last < 0; --s) if (n->prof_has[i] & (1 << s)) last = s;
        if (last >= 0) { AZ_HIP(hipEventSynchronize(n->prof_ev[((size_t)i * 4 + last) * 2 + 1])); break; }
    }
    for (int i = 0; i < n->prof_used; ++i)
        for (int s = 0; s < 4; ++s) {
            if (!(n->prof_has[i] & (1 << s))) continue;
            float ms = 0.0f;
            AZ_HIP(hipEventElapsedTime(&ms, n->prof_ev[((size_t)i * 4 + s) * 2], n->prof_ev[((size_t)i * 4 + s) * 2 + 1]));
            const int kind = n->prof_kind[i];  // bit 0: k_trunk2, bit 1: k_trunk_q, bit 2 / 3: fc1 / fc2 on a small-batch kernel
            int slot = s;
            if (s == 0) slot = (kind & 1) ? 0 : ((kind & 2) ? 7 : 4);
            else if (s == 1 && (kind & 4)) slot = 5;
            else if (s == 2 && (kind & 8)) slot = 6;
            n->prof_ms[slot] += ms;
            n->prof_n[slot] += 1;
        }
    n->prof_used = 0;
    return AZ_OK;
}

// az_net_profile(net, 1): every forward from now on launches its stage kernels with a start and a stop event each
// (AZ_LAUNCH); az_net_profile_read harvests them: total ms and launch count per kernel family (include/az_amd.h).
extern "C" int az_net_profile(az_net *n, int enable) {
    AZ_REQUIRE(n, AZ_EINVAL, "null net");
    if (enable && n->prof_ev.empty()) {
        n->prof_ev.resize((size_t)PROF_SLOTS * 8);
        n->prof_kind.assign(PROF_SLOTS, 0);
        n->prof_has.assign(PROF_SLOTS, 0);
        for (auto &e : n->prof_ev) AZ_HIP(hipEventCreate(&e));
    }
    if (enable) { n->prof_used = 0; for (int i = 0; i < 8; ++i) { n->prof_ms[i] = 0; n->prof_n[i] = 0; } }
    n->prof = enable != 0;
    return AZ_OK;
}

extern "C" int az_net_profiling(const az_net *n) { return n && n->prof ? 1 : 0; }

// Round 1-3 recorded events BETWEEN the launches and subtracted the calibrated cost of an empty event-to-event interval (4.6-4.9 us);
// against rocprofv3's per-dispatch table that over-corrects a 35 us kernel by ~1.9 us (a marker behind a kernel is processed while the
// kernel runs; two markers back to back are not).  With start / stop events on the launch itself there is nothing to subtract:
// az_net_profile_overhead reports 0 and stays for callers that subtract it.
extern "C" int az_net_profile_overhead(az_net *n, double *ms_per_interval) {
    AZ_REQUIRE(n && ms_per_interval, AZ_EINVAL, "null argument");
    *ms_per_interval = n->prof_empty_ms;
    return AZ_OK;
}

extern "C" int az_net_profile_read(az_net *n, double *ms_total, int64_t *launches) {
    AZ_REQUIRE(n && ms_total && launches, AZ_EINVAL, "null argument");
    AZ_TRY(prof_harvest(n));
    for (int i = 0; i < 8; ++i) { ms_total[i] = n->prof_ms[i]; launches[i] = n->prof_n[i]; }
    return AZ_OK;
}

static int forward_impl(az_net *n, const float *d_input, int B, const int *dyn, float *d_probs, float *d_value, void *stream) {
    AZ_REQUIRE(n && d_input && d_probs && d_value, AZ_EINVAL, "null argument");
    AZ_REQUIRE(n->committed, AZ_ESTATE, "az_net_commit has not been called since the last az_net_set_tensor");
    AZ_REQUIRE(B > 0 && B <= n->max_batch, AZ_EINVAL, "batch %d outside (0, max_batch=%d]", B, n->max_batch);
    hipStream_t st = (hipStream_t)stream;
    if (n->game == AZ_TICTACTOE) {
        hipLaunchKernelGGL(k_mlp, dim3((B + 63) / 64), dim3(64), 0, st, d_input, B, dyn, n->mlp_dev, d_probs, d_value);
        return AZ_OK;
    }
    if (!n->prof) {
        for (int s = 0; s < 4; ++s) AZ_TRY(run_stage(n, s, d_input, B, dyn, d_probs, d_value, st));
        return AZ_OK;
    }
    if (n->prof_used == PROF_SLOTS) AZ_TRY(prof_harvest(n));
    hipEvent_t *ev = n->prof_ev.data() + (size_t)n->prof_used * 8;
    int kind = 0, has = 0, rc = AZ_OK;
    for (int s = 0; s < 4 && rc == AZ_OK; ++s) {
        g_last_gemm_small = 0;
        g_ev_start = ev[2 * s]; g_ev_stop = ev[2 * s + 1]; g_ev_used = 0;
        rc = run_stage(n, s, d_input, B, dyn, d_probs, d_value, st);
        g_ev_start = g_ev_stop = nullptr;
        if (g_ev_used) has |= 1 << s;
        if (s == 0) kind |= n->last_trunk_two_boards ? 1 : (n->last_trunk_q ? 2 : 0);
        if ((s == 1 || s == 2) && g_last_gemm_small) kind |= s == 1 ? 4 : 8;
    }
    AZ_TRY(rc);
    n->prof_has[n->prof_used] = has;
    n->prof_kind[n->prof_used++] = kind;
    return AZ_OK;
}

extern "C" int az_net_forward(az_net *n, const float *d_input, int B, float *d_probs, float *d_value, void *stream) {
    return forward_impl(n, d_input, B, nullptr, d_probs, d_value, stream);
}

extern "C" int az_net_forward_dyn(az_net *n, const float *d_input, const int32_t *d_count, int max_B, float *d_probs,
                                  float *d_value, void *stream) {
    AZ_REQUIRE(d_count, AZ_EINVAL, "null count pointer");
    return forward_impl(n, d_input, max_B, d_count, d_probs, d_value, stream);
}

extern "C" int az_net_time_stage(az_net *n, int stage, int B, int iters, void *stream, float *ms_per_launch) {
    AZ_REQUIRE(n && ms_per_launch && iters > 0, AZ_EINVAL, "bad arguments");
    AZ_REQUIRE(n->committed, AZ_ESTATE, "network not committed");
    AZ_REQUIRE(B > 0 && B <= n->max_batch, AZ_EINVAL, "batch %d outside (0, max_batch=%d]", B, n->max_batch);
    AZ_REQUIRE(n->game != AZ_TICTACTOE || stage < 0, AZ_EINVAL, "the TicTacToe MLP has a single stage");
    hipStream_t st = (hipStream_t)stream;
    float *in = nullptr, *pr = nullptr, *va = nullptr;
    AZ_HIP(hipMalloc((void **)&in, (size_t)B * n->H * n->W * sizeof(float)));
    AZ_HIP(hipMalloc((void **)&pr, (size_t)B * n->A * sizeof(float)));
    AZ_HIP(hipMalloc((void **)&va, (size_t)B * sizeof(float)));
    AZ_HIP(hipMemsetAsync(in, 0, (size_t)B * n->H * n->W * sizeof(float), st));
    hipEvent_t e0, e1;
    AZ_HIP(hipEventCreate(&e0)); AZ_HIP(hipEventCreate(&e1));
    int rc = AZ_OK;
    for (int it = -2; it < iters && rc == AZ_OK; ++it) {  // two untimed warm-up launches
        if (it == 0) (void)hipEventRecord(e0, st);
        rc = stage < 0 ? az_net_forward(n, in, B, pr, va, st) : run_stage(n, stage, in, B, nullptr, pr, va, st);
    }
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_per_launch = ms / (float)iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(in); (void)hipFree(pr); (void)hipFree(va);
    return rc;
}

// Name of the kernel stage `stage` (0 trunk, 1 fc1, 2 fc2, 3 heads) launches for a batch of B boards: what rocprofv3 lists the
// launch under (template arguments abbreviated).  bench.py labels its roofline object with it.
extern "C" int az_net_stage_kernel(const az_net *n, int stage, int B, char *buf, int cap) {
    AZ_REQUIRE(n && buf && cap > 0 && stage >= 0 && stage <= 3 && B > 0, AZ_EINVAL, "bad arguments");
    const char *name = "";
    if (n->game == AZ_TICTACTOE) name = "k_mlp";
    else if (stage >= 1 && tail_is_fused(n)) name = tail_v1() ? "k_tail_small" : "k_tail_mfma";
    else if (stage == 0) {
        const bool tuned = (n->CH == 8 && n->CW == 8) || (n->CH == 6 && n->CW == 6) || (n->CH == 7 && n->CW == 6);
        name = (tuned && !trunk_v1() && B >= 4096) ? (use_wino(n->CH, n->CW) ? "k_trunk2<Winograd conv2>" : "k_trunk2") : ((tuned && B <= trunk_q_max()) ? "k_trunk_q" : "k_trunk");
    }
    else if (stage == 3) name = (B <= heads_small_max(n) && n->NH <= 128 && n->F2 % 32 == 0) ? "k_heads_small" : (n->F2 == 512 ? "k_heads2" : "k_heads");
    else if (n->qd_on) name = qd_small_serves(B, stage == 1 ? n->F1 : n->F2, stage == 1 ? n->FIN : n->F1) ? "k_qdense_small" : "k_q_rows + k_qgemm";
    else {
        const int N = stage == 1 ? n->F1 : n->F2, K = stage == 1 ? n->FIN : n->F1;
        switch (gemm_kind(B, N, K, (stage == 1 ? n->fc1wq : n->fc2wq) != nullptr)) {
            case GK_SMALL: name = "k_dense_small"; break;
            case GK_FRAG: name = "k_dense_frag"; break;
            case GK_SOLO: name = "k_gemm_solo"; break;
            case GK_SOLO_T: name = "k_gemm_solo_t"; break;
            default: name = "k_gemm"; break;
        }
    }
    snprintf(buf, (size_t)cap, "%s", name);
    return AZ_OK;
}

#ifdef AZ_PROBE
extern "C" int az_debug_read_probe(unsigned long long *h_out, int n_words) {
    AZ_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(az_probe_buf), sizeof(unsigned long long) * (size_t)n_words));
    return AZ_OK;
}
#endif
