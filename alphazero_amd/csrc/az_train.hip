// az_train.hip -- the optimisation step of AlphaZeroTrainer.optimize_network (trainer.py:320-381) on MI355X (gfx950):
// forward in TRAIN mode (BatchNorm batch statistics, dropout), loss, backward and the momentum-SGD update of OthelloNet /
// Connect4Net (othello.py:341-382, connect4.py:370-412) as fourteen hand-written kernels per step (fifteen where several workgroups share
// a board; nineteen above batch 128, where the dense layers are row-split), float32 in / float32 accumulate on the f32-input matrix cores
// (v_mfma_f32_16x16x4_f32), on device-resident samples.
//
//   loss   = -sum(pi * log_softmax(logits)) / B + sum((tanh(u) - z)^2) / B                       (trainer.py:352-354)
//   update = torch.optim.SGD(lr, momentum, weight_decay): g += wd p;  m = mu m + g;  p -= lr m  (trainer.py:326), on EVERY parameter
//   BatchNorm (train): batch mean / biased variance, eps 1e-5; running stats <- 0.9 old + 0.1 (mean, unbiased variance)
//   dropout p on the two dense layers: a Philox4x32-10 mask keyed (seed, step, layer, element) -- torch's own stream cannot be matched
//
// Why a chain of launches and no persistent kernel: train-mode BatchNorm puts a batch-wide reduction behind every layer, forward and
// backward.  A dependent kernel boundary costs ~1.5 us on this chip, a grid-wide barrier inside one launch ~4-5 us
// (MI355X_MICROARCH.md, rows "boundary" / "barrier-xcd"), so the reductions are cut at launch boundaries: every kernel leaves
// per-workgroup partials (count / mean / M2 for the forward statistics, double sums for the backward ones), the consumer combines
// them in a fixed order -- deterministic, no float atomics.  The step's launch parameters never change (step counter, learning rate,
// permutation offset and loss slot live in device memory), so the host captures it once as a HIP graph and replays it.
//
// Layout: activations are NHWC ([board][position][32 channels]); the flatten in front of fc1 is the reference's NCHW order, so
// fc1.weight is held with its input index permuted (k = position * 32 + channel) and permuted back on export.
//   kernel            grid                         work
//   k_conv1_fwd       boards                       gather int8 state -> conv1 (VALU, K = 9) -> c1 + statistics partial
//   k_conv_fwd x3     boards                       BN+ReLU of the previous layer on load -> implicit GEMM on MFMA -> c_l + partial
//   k_fc_fwd x2       16 columns per workgroup     (BN4+ReLU on load) GEMM over all rows -> column statistics -> BN1d, ReLU, dropout
//   k_heads_fwd       16 rows per workgroup        logits, log_softmax, tanh, the two loss sums, d loss / d logits
//   k_heads_bwd       16 columns of fc2            d h2, fc_bn2 backward, heads weight gradient + update
//   k_fc_dgrad        16 columns of fc1            d h1 (old W2), fc_bn1 backward
//   k_mix1            tiles | 16 input columns     fc2 weight gradient + update | d a4 (old W1), ReLU mask, column sums for bn4
//   k_mix2            tiles | boards               fc1 weight gradient + update | conv4 backward (data + weight partials)
//   k_conv_bwd x2     boards                       conv3 / conv2 backward; conv2's also leaves conv1's weight gradient as per-board MOMENTS
//                                                  (d.F1B: sum x_tap dy0, sum x_tap xhat0, sum x_tap -- the gradient is linear in them, with the
//                                                  batch-wide BatchNorm sums as coefficients), which k_update combines
//   k_conv1_bwd       boards                       conv1 weight partials -- only where d.F1B is off (workgroups sharing a board; batch > 128, where
//                                                  its launch also carries fc1's weight-gradient tiles)
//   k_update          elements                     conv weights / biases / BN2d affine: reduce partials + SGD; running stats; loss log; ++step
// (rocprofv3 on a replayed step, profiles/r05_train_othello8_64_fold.txt: consecutive kernels follow each other with no gap; what a launch
// costs is inside its own duration -- 5-6 us for k_update / k_conv1_fwd, which do next to nothing -- so a launch is worth removing only when
// its work fits into a neighbour for less than that.  conv1's backward did (7.3 us gone, +0.8 per k_conv_bwd, +0.7 in k_update); conv1's
// forward inside conv2's (statistics from the batch's 54 input moments, bit masks and popcounts: 25.6 against 5.8 + 12.6 us) and the two
// heads kernels in one (every column workgroup recomputing the batch's logits: 52.8 against 13.4 + 11.3 us) did not and were taken out.)
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "az_device.h"
#include "az_host.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// MFMA 16x16x4 f32 operand layout (lane l): A[row l & 15][k l >> 4], B[k l >> 4][col l & 15], C/D[row 4 (l >> 4) + r][col l & 15]

#ifdef AZ_TPROBE  // diagnostic build (make TPROBE=1): wall-clock stamps (100 MHz) of workgroup 0 at the phase boundaries of every kernel
__device__ unsigned long long az_tprobe[32 * 16];
#define TSTAMP(k, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) az_tprobe[(k) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TBEG(id) do { if (threadIdx.x == 0) az_tprobe[(id) * 16 + 14] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TEND(id) do { __syncthreads(); if (threadIdx.x == 0) az_tprobe[(id) * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(k, i)
#define TBEG(id)
#define TEND(id)
#endif

#define NCH 32
#define LDP 36        // LDS row stride of a 32-channel row: 36 = 4 mod 32 -> the 16 x 4 (row, k) lanes of an A fragment hit every bank twice
#define TPB 256
#define BN_EPS 1e-5
#define BN_MOM 0.1
#define FPART 65      // forward statistics partial: n, mean[32], M2[32]
#define MAXB 512

struct Hyper {        // device-resident so that a captured step never goes stale
    float lr, momentum, wd, drop_p;
    unsigned seed;
    int step;         // steps done since az_trainer_begin (k_update increments it)
    int perm_off;     // permutation entries consumed so far: row of board b = perm[perm_off + b]
    int loss_off;     // slot of this step's losses in loss_pi / loss_v
    long long n_samples;  // rows of the caller's sample arrays: a permutation entry outside [0, n_samples) is an error, not an address
    int err;          // sticky: set by sample_row, reported (and cleared) by az_trainer_check / the next az_trainer_steps
    int pad_;
};

// row of the sample arrays that batch slot i trains on (trainer.py:288-318 draws the indices from len(memory)); an entry outside the
// arrays raises the error flag and reads row 0 instead of faulting the GPU
AZ_D long long sample_row(const long long *perm, int i, const Hyper &hp, Hyper *dev) {
    long long r = perm[i];
    if (r < 0 || r >= hp.n_samples) { atomicOr(&dev->err, 1); r = 0; }
    return r;
}

struct PSet {         // one set of tensors in the step's own layout (parameters; the momentum buffers mirror it)
    float *cw[4];     // conv1 [9][32 oc]; conv2..4 [9 taps][32 ic][32 oc]
    float *cb[4];     // [32]
    float *bg[4], *bb[4];   // BatchNorm2d weight / bias [32]
    float *w1, *b1, *g1, *be1;  // fc1 [F1][FIN] (input index NHWC), fc_bn1
    float *w2, *b2, *g2, *be2;  // fc2 [F2][F1], fc_bn2
    float *wh, *bh;   // heads [NHP][F2]: rows 0..A-1 fc_probs, row A fc_value, rest zero; [NHP]
};

struct TDims {
    int CH, CW, P1, H3, W3, P3, H4, W4, P4, FIN, F1, F2, A, NH, NHP;
    int B;    // batch size (multiple of 16, <= MAXB)
    int NB;   // workgroups of the per-board kernels = statistic partials per layer (<= 256)
    int S;    // workgroups that share a board (4 up to 64 boards, 2 up to 128, else 1): position tiles / taps are dealt out among them
    int NRB;  // row blocks of the dense kernels: 1 = a workgroup sees ALL rows of its 16 columns (batch <= 128); above, the rows are split
              // into NRB blocks of RB so that the whole chip works, and the batch statistics go through per-block partials (k_fc_fin, k_bn1d_bwd_fin)
    int RB;   // rows per row block (multiple of 16)
    int F1B;  // 1: conv1's weight gradient is folded into conv2's backward (per-board moment partials, combined by k_update): no k_conv1_bwd launch
};
#define C1P 640       // conv1 partial in the folded form: A[9][32] | C[9][32] | sum xhat [32] | Bx[9] (see conv_bwd_body, l = 1)
#define NRBMAX 8

struct TPtr {
    PSet p, m;
    float *rm[4], *rv[4], *rm1, *rv1, *rm2, *rv2;  // running statistics
    const int8_t *state; const float *pi; const int8_t *z; const long long *perm;
    float *loss_pi, *loss_v;
    Hyper *hp;
    float *x0, *c[4], *fpart[4];
    float *y1, *h1, *mu1, *iv1, *y2, *h2, *mu2, *iv2, *dlog, *losspart;
    float *dz2, *dz1, *dy[4];
    double *bpart[3];   // backward sums of bn1..bn3: [NB][64] (sum dy, sum dy xhat per channel)
    double *colsum;     // backward sums of bn4 per fc1 input column and row block: [NRB][FIN][2]
    double *fstat[2];   // row-split dense forward: (n, mean, M2) of y per (row block, column): [NRBMAX][N][3]
    double *bstat[2];   // row-split BatchNorm1d backward: (sum dy, sum dy xhat, sum xhat) per (row block, column): [NRBMAX][N][3]
    float *fdone[4];    // finished forward statistics of BatchNorm2d l: mean[32] inv[32] scale[32] shift[32] var[32], written once per step by
                        // workgroup 0 of the first kernel that combines the layer's partials (conv_fwd l + 1; fc1's forward for l = 3);
                        // the backward kernels and k_update read these 160 floats instead of walking NB partials again
    double *bdone[4];   // finished backward sums of BatchNorm2d l: sum dy [32], sum dy xhat [32], written by workgroup 0 of the kernel that
                        // combines them for its own use (conv_bwd l; k_conv1_bwd for l = 0); k_update's affine gradients read them
    float *hwpart;      // row-split heads backward: the row block's share of d Wh [NRBMAX][NHP][F2], then of d bh [NRBMAX][NHP]
    float *gw[4];       // weight-gradient partials [NB][9*32*32 + 32] (conv1: [NB][9*32 + 32]): weights then bias
};

// ---------------------------------------------------------------------------------------------------------------- helpers
AZ_D int conv_hin(const TDims &d, int l) { return l <= 2 ? d.CH : d.H3; }   // l = 1..3: conv2, conv3, conv4
AZ_D int conv_win(const TDims &d, int l) { return l <= 2 ? d.CW : d.W3; }
AZ_D int conv_hout(const TDims &d, int l) { return l == 1 ? d.CH : (l == 2 ? d.H3 : d.H4); }
AZ_D int conv_wout(const TDims &d, int l) { return l == 1 ? d.CW : (l == 2 ? d.W3 : d.W4); }
AZ_D int plane_of(const TDims &d, int l) { return l <= 1 ? d.P1 : (l == 2 ? d.P3 : d.P4); }  // positions of c[l], l = 0..3

// Combination of the (n, mean, M2) partials of a layer in a fixed order: all TPB threads call; s_mean / s_var (biased) valid after
// return.  One pass in double: N = sum n_p, S = sum n_p mean_p, Q = sum (M2_p + n_p mean_p^2); mean = S / N, M2 = Q - N mean^2 (the
// subtraction costs log10(mean^2 / var) of double's 16 digits: harmless; the partials themselves are shifted sums, so nothing was lost
// in float).  No division and no dependent load inside the loop.
AZ_D void bn2d_combine(const float *part, int npart, float *s_mean, float *s_var, double *scr /* [8][32][3] */) {
    const int t = threadIdx.x, ch = t & 31, grp = t >> 5;
    double N = 0.0, S = 0.0, Q = 0.0;
    if (t < 256) {  // the dense kernels run up to 1024 threads: the first 256 walk the partials
#pragma unroll 4
        for (int p = grp; p < npart; p += 8) {
            const double nb = part[p * FPART], mb = part[p * FPART + 1 + ch], qb = part[p * FPART + 33 + ch];
            N += nb; S += nb * mb; Q += qb + nb * mb * mb;
        }
        scr[(grp * 32 + ch) * 3 + 0] = N; scr[(grp * 32 + ch) * 3 + 1] = S; scr[(grp * 32 + ch) * 3 + 2] = Q;
    }
    __syncthreads();
    if (t < 32) {
        N = 0.0; S = 0.0; Q = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) { N += scr[(g * 32 + ch) * 3]; S += scr[(g * 32 + ch) * 3 + 1]; Q += scr[(g * 32 + ch) * 3 + 2]; }
        const double mean = S / N;
        double M2 = Q - N * mean * mean;
        if (M2 < 0.0) M2 = 0.0;
        s_mean[ch] = (float)mean; s_var[ch] = (float)(M2 / N);
    }
    __syncthreads();
}

// The same in two halves, for kernels that need several statistics at once: fpart_walk issues this thread's loads and sums (no barrier),
// so that the walks over two or three partial arrays are in flight together; the caller stores the sums, synchronises once and lets
// 32 threads finish every statistic.
AZ_D void fpart_walk(const float *part, int npart, double &N, double &S, double &Q) {
    const int t = threadIdx.x, ch = t & 31, grp = t >> 5;
    N = 0.0; S = 0.0; Q = 0.0;
#pragma unroll 8
    for (int p = grp; p < npart; p += 8) {
        const double nb = part[p * FPART], mb = part[p * FPART + 1 + ch], qb = part[p * FPART + 33 + ch];
        N += nb; S += nb * mb; Q += qb + nb * mb * mb;
    }
}
AZ_D void bsum_walk(const TDims &d, const TPtr &q, int l, double &S1, double &S2) {  // backward sums of BatchNorm2d l (0..3)
    const int t = threadIdx.x, ch = t & 31, grp = t >> 5;
    S1 = 0.0; S2 = 0.0;
    if (l == 3) {
        for (int rb = 0; rb < d.NRB; ++rb)  // row blocks in order, positions dealt over the groups: a fixed summation order
            for (int p = grp; p < d.P4; p += 8) { S1 += q.colsum[2 * ((size_t)rb * d.FIN + p * 32 + ch)]; S2 += q.colsum[2 * ((size_t)rb * d.FIN + p * 32 + ch) + 1]; }
    } else {
#pragma unroll 8
        for (int p = grp; p < d.NB; p += 8) { S1 += q.bpart[l][(size_t)p * 64 + ch]; S2 += q.bpart[l][(size_t)p * 64 + 32 + ch]; }
    }
}
// finish a forward statistic from the eight group sums at scr[g * 32 * W + ch * W + o .. o + 2]: mean, 1 / sqrt(var + eps), scale, shift
AZ_D void fpart_finish(const TDims &d, const TPtr &q, int l, const double *scr, int W, int o, float *s_scale, float *s_shift, float *s_mean, float *s_inv,
                       float *done = nullptr) {
    const int ch = threadIdx.x;
    double N = 0.0, S = 0.0, Q = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) { N += scr[(g * 32 + ch) * W + o]; S += scr[(g * 32 + ch) * W + o + 1]; Q += scr[(g * 32 + ch) * W + o + 2]; }
    const double mean = S / N;
    double M2 = Q - N * mean * mean;
    if (M2 < 0.0) M2 = 0.0;
    const float inv = (float)(1.0 / sqrt(M2 / N + BN_EPS));
    const float sc = q.p.bg[l][ch] * inv;
    s_mean[ch] = (float)mean; s_inv[ch] = inv; s_scale[ch] = sc; s_shift[ch] = q.p.bb[l][ch] - (float)mean * sc;
    if (done) { done[ch] = s_mean[ch]; done[32 + ch] = inv; done[64 + ch] = sc; done[96 + ch] = s_shift[ch]; done[128 + ch] = (float)(M2 / N); }
}
// the finished statistics of layer l as the first combining kernel left them (threads 0..31 call)
AZ_D void fdone_load(const float *done, float *s_scale, float *s_shift, float *s_mean, float *s_inv) {
    const int ch = threadIdx.x;
    s_mean[ch] = done[ch]; s_inv[ch] = done[32 + ch]; s_scale[ch] = done[64 + ch]; s_shift[ch] = done[96 + ch];
}

// scale / shift of train-mode BatchNorm2d l (0..3) from its forward partials; also mean / 1/sqrt(var + eps)
AZ_D void bn2d_prepare(const TDims &d, const TPtr &q, int l, float *s_scale, float *s_shift, float *s_mean, float *s_inv, double *scr, float *done = nullptr) {
    bn2d_combine(q.fpart[l], d.NB, s_mean, s_inv, scr);
    if (threadIdx.x < 32) {
        const int ch = threadIdx.x;
        const float var = s_inv[ch];
        const float inv = (float)(1.0 / sqrt((double)var + BN_EPS));
        const float sc = q.p.bg[l][ch] * inv;
        s_inv[ch] = inv; s_scale[ch] = sc; s_shift[ch] = q.p.bb[l][ch] - s_mean[ch] * sc;
        if (done) { done[ch] = s_mean[ch]; done[32 + ch] = inv; done[64 + ch] = sc; done[96 + ch] = s_shift[ch]; done[128 + ch] = var; }
    }
    __syncthreads();
}
// the same from the record the first combining kernel of the step left (no walk over the partials): all threads call
AZ_D void bn2d_from_done(const float *done, float *s_scale, float *s_shift, float *s_mean, float *s_inv) {
    if (threadIdx.x < 32) fdone_load(done, s_scale, s_shift, s_mean, s_inv);
    __syncthreads();
}

// per-channel (mean, M2) of an LDS plane pl[P][LDP] (two passes, double); every thread returns the values of channel t & 31
AZ_D void plane_stats(const float *pl, int P, double *scr /* [8][32] */, double &mean, double &M2) {
    const int t = threadIdx.x, ch = t & 31, grp = t >> 5;
    double s = 0.0;
    for (int p = grp; p < P; p += 8) s += pl[p * LDP + ch];
    scr[grp * 32 + ch] = s;
    __syncthreads();
    double tot = 0.0;
    for (int g = 0; g < 8; ++g) tot += scr[g * 32 + ch];
    mean = tot / P;
    __syncthreads();
    double qq = 0.0;
    for (int p = grp; p < P; p += 8) { const double dl = pl[p * LDP + ch] - mean; qq += dl * dl; }
    scr[grp * 32 + ch] = qq;
    __syncthreads();
    M2 = 0.0;
    for (int g = 0; g < 8; ++g) M2 += scr[g * 32 + ch];
    __syncthreads();
}

AZ_D void chan_merge(double &n, double &mean, double &M2, double nb, double mb, double qb) {
    const double nn = n + nb, dl = mb - mean;
    mean += dl * nb / nn; M2 += qb + dl * dl * n * nb / nn; n = nn;
}

// per-channel sums of a plane and of plane x xhat over its positions (double), accumulated into the caller's registers (channel t & 31,
// position group t >> 5: the cross-group reduction happens once, at the end of the kernel)
AZ_D void plane_sums(const float *pl, const float *xh, int P, double &s1, double &s2) {
    const int t = threadIdx.x, ch = t & 31, grp = t >> 5;
    for (int p = grp; p < P; p += 8) { const double v = pl[p * LDP + ch]; s1 += v; s2 += v * (double)xh[p * LDP + ch]; }
}

// momentum SGD with weight decay on one element (torch.optim.SGD, dampening 0, no Nesterov)
AZ_D void sgd(float *p, float *m, float g, const Hyper &hp) {
    const float gg = fmaf(hp.wd, *p, g);
    const float mm = fmaf(hp.momentum, *m, gg);
    *m = mm;
    *p = fmaf(-hp.lr, mm, *p);
}

AZ_D float dropout_scale(const Hyper &hp, int layer, unsigned idx) {  // 0 (dropped) or 1 / (1 - p)
    if (hp.drop_p <= 0.0f) return 1.0f;
    const Philox4 r = az_philox(hp.seed, 0x54524e00u + (u32)layer, (u32)hp.step, idx, 0x44524f50u, 0u);
    const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);
    return u < hp.drop_p ? 0.0f : 1.0f / (1.0f - hp.drop_p);
}

// ---------------------------------------------------------------------------------------------------------------- conv1 forward
__global__ __launch_bounds__(TPB) void k_conv1_fwd(TDims d, TPtr q) {
    __shared__ float xin[10 * 10];
    __shared__ float wl[9 * 32], bl[32];
    __shared__ float pl[64 * LDP];
    __shared__ double scr[8 * 32];
    const int t = threadIdx.x, WP = d.CW + 2;
    if (blockIdx.x == 0) TBEG(0);
    const Hyper hp = *q.hp;
    for (int i = t; i < 9 * 32; i += TPB) wl[i] = q.p.cw[0][i];
    if (t < 32) bl[t] = q.p.cb[0][t];
    for (int i = t; i < 100; i += TPB) xin[i] = 0.0f;
    double n = 0.0, mean = 0.0, M2 = 0.0;
    // the board's row (permutation -> sample row -> 64 int8 cells: two dependent trips) is fetched one board ahead
    float xv = 0.0f;
    if ((int)blockIdx.x < d.B && t < d.P1) xv = (float)q.state[sample_row(q.perm, hp.perm_off + blockIdx.x, hp, q.hp) * d.P1 + t];
    __syncthreads();
    for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
        if (t < d.P1) {
            xin[(t / d.CW + 1) * WP + t % d.CW + 1] = xv;
            q.x0[b * d.P1 + t] = xv;
        }
        if (b + (int)gridDim.x < d.B && t < d.P1) xv = (float)q.state[sample_row(q.perm, hp.perm_off + b + gridDim.x, hp, q.hp) * d.P1 + t];
        __syncthreads();
        for (int i = t; i < d.P1 * 32; i += TPB) {
            const int p = i >> 5, oc = i & 31, r = p / d.CW, c = p % d.CW;
            float acc = bl[oc];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc = fmaf(xin[(r + tap / 3) * WP + c + tap % 3], wl[tap * 32 + oc], acc);
            q.c[0][((size_t)b * d.P1 + p) * 32 + oc] = acc;
            pl[p * LDP + oc] = acc;
        }
        __syncthreads();
        double mb, qb;
        plane_stats(pl, d.P1, scr, mb, qb);
        chan_merge(n, mean, M2, (double)d.P1, mb, qb);
    }
    if (t < 32) {
        float *o = q.fpart[0] + (size_t)blockIdx.x * FPART;
        if (t == 0) o[0] = (float)n;
        o[1 + t] = (float)mean; o[33 + t] = (float)M2;
    }
    if (blockIdx.x == 0) TEND(0);
}

// ---------------------------------------------------------------------------------------------------------------- conv2..4 forward
// implicit GEMM per board: M = output positions (tiles of 16), N = 32 oc (2 tiles), K = 9 taps x 32 ic.
// LDS: wl[288][LDP] weights (row tap*32+ic, column oc) | ain[(Hin+2 pad)(Win+2 pad)][LDP] input activation with a zero halo | pl[64][LDP]
// | red[4 waves][16][LDP].  A board is shared by S workgroups (position tiles dealt round robin) and a tile's K by the four waves of
// its workgroup: at 64 boards all 256 CUs work, and a layer with a single tile (conv4) no longer leaves three waves idle.
#define CONV_FWD_LDS_FLOATS (288 * LDP + 100 * LDP + 64 * LDP + 64 * LDP + 4 * 32 + 64)
#define CONV_FWD_LDS_BYTES (CONV_FWD_LDS_FLOATS * 4 + 8 * 32 * 3 * 8)
__global__ __launch_bounds__(TPB) void k_conv_fwd(TDims d, TPtr q, int l /* 1..3 */) {
    extern __shared__ __align__(16) float lds[];
    float *wl = lds, *ain = wl + 288 * LDP, *pl = ain + 100 * LDP, *red = pl + 64 * LDP, *s_scale = red + 64 * LDP, *s_shift = s_scale + 32,
          *s_mean = s_shift + 32, *s_inv = s_mean + 32;
    double *scr = (double *)(s_inv + 32 + 64);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4;
    const int Hin = conv_hin(d, l), Win = conv_win(d, l), Hout = conv_hout(d, l), Wout = conv_wout(d, l), pad = l == 1 ? 1 : 0;
    const int Pin = Hin * Win, Pout = Hout * Wout, WP = Win + 2 * pad, MT = (Pout + 15) / 16;
    if (blockIdx.x == 0) TBEG(l);
    TSTAMP(1, 0);
    const float *cin = q.c[l - 1];
    float *cout = q.c[l];
    float r_c[8];  // the board's input plane travels through registers: first board before anything else, the next one under the MFMAs
    auto fetch = [&](int b) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (t + TPB * k < Pin * 32) r_c[k] = cin[(size_t)b * Pin * 32 + t + TPB * k];
    };
    const int S = d.S, part = blockIdx.x % S, bfirst = blockIdx.x / S, bstride = gridDim.x / S;
    if (bfirst < d.B) fetch(bfirst);
    float4 wreg[9];  // 9216 floats as 2304 float4: nine per thread, all in flight at once
#pragma unroll
    for (int i = 0; i < 9; ++i) wreg[i] = *(const float4 *)(q.p.cw[l] + (i * TPB + t) * 4);
    const float bias_t = q.p.cb[l][t & 31];
    {
        double a0, a1, a2;
        fpart_walk(q.fpart[l - 1], d.NB, a0, a1, a2);
        double *o = scr + (size_t)((t >> 5) * 32 + (t & 31)) * 3;
        o[0] = a0; o[1] = a1; o[2] = a2;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int e = (i * TPB + t) * 4;
        *(float4 *)(wl + (e >> 5) * LDP + (e & 31)) = wreg[i];
    }
    for (int i = t; i < 100 * LDP; i += TPB) ain[i] = 0.0f;
    int *pio = (int *)(s_inv + 32);  // input position -> offset of its cell in the haloed plane (no division in the board loop)
    if (t < Pin) pio[t] = ((t / Win + pad) * WP + t % Win + pad) * LDP;
    TSTAMP(1, 1);
    __syncthreads();
    if (t < 32) fpart_finish(d, q, l - 1, scr, 3, 0, s_scale, s_shift, s_mean, s_inv, blockIdx.x == 0 ? q.fdone[l - 1] : nullptr);
    __syncthreads();
    TSTAMP(1, 2);
    double n = 0.0, mean = 0.0, M2 = 0.0;
    for (int b = bfirst; b < d.B; b += bstride) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + TPB * k, p = i >> 5, ic = i & 31;
            if (i < Pin * 32) {
                const float v = fmaf(r_c[k], s_scale[ic], s_shift[ic]);
                ain[pio[p] + ic] = v > 0.0f ? v : 0.0f;
            }
        }
        if (b + bstride < d.B) fetch(b + bstride);
        __syncthreads();
        TSTAMP(1, 3);
        int nloc = 0;  // rows of this workgroup's tiles that lie inside the plane (its tiles are mt = part, part + S, ...)
        const int my_tiles = (MT - part + S - 1) / S;
        if (my_tiles >= 3) {
            // three or four tiles: one tile per wave, its whole K in that wave (no reduction, no extra barrier)
            const int mt = part + S * wave;
            if (mt < MT) {
                int m = 16 * mt + n16;
                if (m >= Pout) m = Pout - 1;  // rows beyond the plane compute a copy of the last position; never stored
                const int base = (m / Wout) * WP + m % Wout;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float *ap = ain + (base + (tap / 3) * WP + tap % 3) * LDP + kq;
                    const float *wp = wl + (tap * 32 + kq) * LDP + n16;
#pragma unroll
                    for (int icb = 0; icb < 8; ++icb) {
                        const float a = ap[icb * 4];
                        acc0 = MFMA(a, wp[icb * 4 * LDP], acc0);
                        acc1 = MFMA(a, wp[icb * 4 * LDP + 16], acc1);
                    }
                }
                const float b0 = __shfl(bias_t, n16, 32), b1 = __shfl(bias_t, 16 + n16, 32);  // bias_t holds channel t & 31
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = 16 * mt + 4 * kq + r;
                    if (mm < Pout) {
                        const float v0 = acc0[r] + b0, v1 = acc1[r] + b1;
                        cout[((size_t)b * Pout + mm) * 32 + n16] = v0; cout[((size_t)b * Pout + mm) * 32 + 16 + n16] = v1;
                        pl[(16 * wave + 4 * kq + r) * LDP + n16] = v0; pl[(16 * wave + 4 * kq + r) * LDP + 16 + n16] = v1;
                    }
                }
            }
            for (int w = 0; w < my_tiles; ++w) nloc += min(16, Pout - 16 * (part + S * w));
            __syncthreads();
        } else
        for (int mt = part; mt < MT; mt += S) {  // one or two tiles: a tile's 72 k-steps (9 taps x 8 blocks of 4 input channels) split over the four waves
            int m = 16 * mt + n16;
            if (m >= Pout) m = Pout - 1;
            const int base = (m / Wout) * WP + m % Wout;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 18; ++jj) {
                const int ks = 18 * wave + jj, tap = ks >> 3, icb = ks & 7, tr = (tap * 11) >> 5, tc = tap - 3 * tr;
                const float a = ain[(base + tr * WP + tc) * LDP + icb * 4 + kq];
                const float *wp = wl + (tap * 32 + icb * 4 + kq) * LDP + n16;
                acc0 = MFMA(a, wp[0], acc0);
                acc1 = MFMA(a, wp[16], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { red[(wave * 16 + 4 * kq + r) * LDP + n16] = acc0[r]; red[(wave * 16 + 4 * kq + r) * LDP + 16 + n16] = acc1[r]; }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = (t >> 5) + 8 * h, oc = t & 31, mm = 16 * mt + rr;
                if (mm < Pout) {
                    const float v = ((red[rr * LDP + oc] + red[(16 + rr) * LDP + oc]) + (red[(32 + rr) * LDP + oc] + red[(48 + rr) * LDP + oc])) + bias_t;
                    cout[((size_t)b * Pout + mm) * 32 + oc] = v;
                    pl[(nloc + rr) * LDP + oc] = v;
                }
            }
            nloc += min(16, Pout - 16 * mt);
            __syncthreads();
        }
        TSTAMP(1, 4);
        if (nloc > 0) {
            double mb, qb;
            plane_stats(pl, nloc, scr, mb, qb);  // ends with a barrier: the next board may overwrite ain / pl
            chan_merge(n, mean, M2, (double)nloc, mb, qb);
        }
        TSTAMP(1, 5);
    }
    if (t < 32) {
        float *o = q.fpart[l] + (size_t)blockIdx.x * FPART;
        if (t == 0) o[0] = (float)n;
        o[1 + t] = (float)mean; o[33 + t] = (float)M2;
    }
    if (blockIdx.x == 0) TEND(l);
}

// ---------------------------------------------------------------------------------------------------------------- dense layers: shared pieces
// The dense kernels give a workgroup 16 output columns and ALL rows of the batch, so that the batch statistics of BatchNorm1d are
// local to it; its NW waves split K, their partial sums meet in LDS.  At batch 64 every wave would otherwise walk K in a dozen
// dependent round trips to L2 (measured: 1.2 us per 16-deep step), so (a) NW grows as the batch shrinks (16 waves at <= 64 rows), and
// (b) a wave loads PF steps' worth of fragments before it issues their MFMAs.
//   batch <= 64: RTM 4, NW 16, PF 1 | <= 128: 8, 8, 1 | <= 256: 16, 4, 1 | <= 512: 32, 4, one buffer    (red[NW][B][16] <= 128 KB)
static inline int fc_lds_bytes(int NW, int B) { return (NW * B * 16 + 4 * 32) * 4 + 768 * 8; }

// acc[rt] += A[16 rt + m][k] Bt[n][k] over k in [kbeg, kend): both operands k-contiguous, a lane loads four consecutive k as one float4
// and feeds them to four MFMAs (the k index inside a step of 16 is 4 kq + i for A and B alike: the products pair up, only the
// summation order differs from k ascending).  BNIN: A = relu(scale[k & 31] a + shift[k & 31]) (bn4 + ReLU formed on load).
// Two register buffers of PF steps each: batch i + 1 is loaded before batch i is multiplied; `hook` runs right after the first
// batch's loads have been issued (the caller's own memory work -- combining BatchNorm partials -- overlaps them).
template <int RTM, int PF, bool BNIN>
struct NtBatch {
    float4 bf[PF], af[PF][RTM];
    AZ_D void load(const float *A, int lda, const float *Brow, int RT, int k0, int kend, int n16, int kq) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (k0 + 16 * p < kend) {
                bf[p] = *(const float4 *)(Brow + k0 + 16 * p + 4 * kq);
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
                    if (rt < RT) af[p][rt] = *(const float4 *)(A + (size_t)(16 * rt + n16) * lda + k0 + 16 * p + 4 * kq);
            }
    }
    AZ_D void mul(f32x4 (&acc)[RTM], int RT, int k0, int kend, int kq, const float *s_scale, const float *s_shift) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (k0 + 16 * p < kend) {
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
                if (BNIN) {
                    const int c0 = (k0 + 16 * p + 4 * kq) & 31;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sc[i] = s_scale[c0 + i]; sh[i] = s_shift[c0 + i]; }
                }
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
                    if (rt < RT) {
                        float4 a = af[p][rt];
                        if (BNIN) {
                            a.x = fmaxf(fmaf(a.x, sc[0], sh[0]), 0.f); a.y = fmaxf(fmaf(a.y, sc[1], sh[1]), 0.f);
                            a.z = fmaxf(fmaf(a.z, sc[2], sh[2]), 0.f); a.w = fmaxf(fmaf(a.w, sc[3], sh[3]), 0.f);
                        }
                        acc[rt] = MFMA(a.x, bf[p].x, acc[rt]); acc[rt] = MFMA(a.y, bf[p].y, acc[rt]);
                        acc[rt] = MFMA(a.z, bf[p].z, acc[rt]); acc[rt] = MFMA(a.w, bf[p].w, acc[rt]);
                    }
            }
    }
};

template <int RTM, int PF, bool BNIN, typename Hook>
AZ_D void nt_kloop(f32x4 (&acc)[RTM], const float *A, int lda, const float *Brow, int RT, int kbeg, int kend, int n16, int kq,
                   const float *s_scale, const float *s_shift, Hook hook) {
    if constexpr (PF == 0) {  // one step per batch, one buffer (32 row tiles: the batch alone is 132 registers)
        NtBatch<RTM, 1, BNIN> b;
        b.load(A, lda, Brow, RT, kbeg, kend, n16, kq);
        hook();
        for (int k0 = kbeg; k0 < kend; k0 += 16) {
            b.mul(acc, RT, k0, kend, kq, s_scale, s_shift);
            if (k0 + 16 < kend) b.load(A, lda, Brow, RT, k0 + 16, kend, n16, kq);
        }
        return;
    }
    constexpr int PFE = PF == 0 ? 1 : PF, ST = 16 * PFE;
    NtBatch<RTM, PFE, BNIN> b0, b1;
    b0.load(A, lda, Brow, RT, kbeg, kend, n16, kq);
    if (kbeg + ST < kend) b1.load(A, lda, Brow, RT, kbeg + ST, kend, n16, kq);
    hook();
    for (int k0 = kbeg; k0 < kend; k0 += 2 * ST) {
        b0.mul(acc, RT, k0, kend, kq, s_scale, s_shift);
        if (k0 + 2 * ST < kend) b0.load(A, lda, Brow, RT, k0 + 2 * ST, kend, n16, kq);
        if (k0 + ST < kend) {
            b1.mul(acc, RT, k0 + ST, kend, kq, s_scale, s_shift);
            if (k0 + 3 * ST < kend) b1.load(A, lda, Brow, RT, k0 + 3 * ST, kend, n16, kq);
        }
    }
}

// the same with B[k][n] stored k-major (row k of W, column k0c + n): one dword per MFMA for B, float4 per four MFMAs for A = dZ
template <int RTM, int PF>
struct NnBatch {
    float bw[PF][4];
    float4 af[PF][RTM];
    AZ_D void load(const float *dZ, int J, const float *W, int ldw, int RT, int j0, int jend, int n16, int kq) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (j0 + 16 * p < jend) {
                const float *wp = W + (size_t)(j0 + 16 * p + 4 * kq) * ldw + n16;
                bw[p][0] = wp[0]; bw[p][1] = wp[ldw]; bw[p][2] = wp[2 * (size_t)ldw]; bw[p][3] = wp[3 * (size_t)ldw];
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
                    if (rt < RT) af[p][rt] = *(const float4 *)(dZ + (size_t)(16 * rt + n16) * J + j0 + 16 * p + 4 * kq);
            }
    }
    AZ_D void mul(f32x4 (&acc)[RTM], int RT, int j0, int jend) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (j0 + 16 * p < jend)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
                    if (rt < RT) {
                        acc[rt] = MFMA(af[p][rt].x, bw[p][0], acc[rt]); acc[rt] = MFMA(af[p][rt].y, bw[p][1], acc[rt]);
                        acc[rt] = MFMA(af[p][rt].z, bw[p][2], acc[rt]); acc[rt] = MFMA(af[p][rt].w, bw[p][3], acc[rt]);
                    }
    }
};

template <int RTM, int PF, typename Hook>
AZ_D void nn_kloop(f32x4 (&acc)[RTM], const float *dZ, int J, const float *W, int ldw, int RT, int jbeg, int jend, int n16, int kq, Hook hook) {
    if constexpr (PF == 0) {
        NnBatch<RTM, 1> b;
        b.load(dZ, J, W, ldw, RT, jbeg, jend, n16, kq);
        hook();
        for (int j0 = jbeg; j0 < jend; j0 += 16) {
            b.mul(acc, RT, j0, jend);
            if (j0 + 16 < jend) b.load(dZ, J, W, ldw, RT, j0 + 16, jend, n16, kq);
        }
        return;
    }
    constexpr int PFE = PF == 0 ? 1 : PF, ST = 16 * PFE;
    NnBatch<RTM, PFE> b0, b1;
    b0.load(dZ, J, W, ldw, RT, jbeg, jend, n16, kq);
    if (jbeg + ST < jend) b1.load(dZ, J, W, ldw, RT, jbeg + ST, jend, n16, kq);
    hook();
    for (int j0 = jbeg; j0 < jend; j0 += 2 * ST) {
        b0.mul(acc, RT, j0, jend);
        if (j0 + 2 * ST < jend) b0.load(dZ, J, W, ldw, RT, j0 + 2 * ST, jend, n16, kq);
        if (j0 + ST < jend) {
            b1.mul(acc, RT, j0 + ST, jend);
            if (j0 + 3 * ST < jend) b1.load(dZ, J, W, ldw, RT, j0 + 3 * ST, jend, n16, kq);
        }
    }
}

// partial sums of the NW waves -> red[0][r][col] (element (r, col) is read and written by the one thread that owns it)
template <int RTM, int NW>
AZ_D void reduce_waves(f32x4 (&acc)[RTM], float *red, int B, int RT, int wave, int n16, int kq) {
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt)
        if (rt < RT)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * B + 16 * rt + 4 * kq + r) * 16 + n16] = acc[rt][r];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 256) {
        const int col = t & 15, rg = t >> 4;
        for (int r = rg; r < B; r += 16) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[((size_t)w * B + r) * 16 + col];
            red[r * 16 + col] = v;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------- dense forward
// y = A W^T + b for 16 columns n0.. and all rows, column statistics, BatchNorm1d (train), ReLU, dropout.
// layer 1: A = relu(bn4(c4)) formed on load (channel = k & 31), layer 2: A = h1.
// SPLIT: the workgroup (blockIdx.x, blockIdx.y) owns 16 columns and row block blockIdx.y (RTM * 16 rows): it writes y and the
// (n, mean, M2) partial of its rows per column; k_fc_fin combines the partials of a column in row-block order and forms h.
template <int RTM, int NW, int PF, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void k_fc_fwd(TDims d, TPtr q, int layer) {
    extern __shared__ __align__(16) float lds[];
    const int rb = SPLIT ? blockIdx.y : 0, r0 = rb * RTM * 16;
    const int B = SPLIT ? min(d.B - r0, RTM * 16) : d.B, RT = B / 16;
    float *red = lds, *s_scale = red + NW * B * 16, *s_shift = s_scale + 32, *s_mean = s_shift + 32, *s_inv = s_mean + 32;
    double *scr = (double *)(s_inv + 32);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4;
    const int N = layer == 1 ? d.F1 : d.F2, K = layer == 1 ? d.FIN : d.F1, n0 = blockIdx.x * 16;
    const float *A = (layer == 1 ? q.c[3] : q.h1) + (size_t)r0 * K, *W = layer == 1 ? q.p.w1 : q.p.w2;
    if (blockIdx.x == 0 && rb == 0) TBEG(3 + layer);
    const Hyper hp = *q.hp;
    TSTAMP(2, 0);
    const bool act = t < 256;
    const int col = t & 15, rg = (t >> 4) & 15, n = n0 + col;
    // what the epilogue needs from memory goes out now, under the K loop
    const float bias = (layer == 1 ? q.p.b1 : q.p.b2)[n];
    float g_ = 0.f, b_ = 0.f, rm_old = 0.f, rv_old = 0.f;
    float *rm = layer == 1 ? q.rm1 : q.rm2, *rv = layer == 1 ? q.rv1 : q.rv2;
    if (!SPLIT) { g_ = (layer == 1 ? q.p.g1 : q.p.g2)[n]; b_ = (layer == 1 ? q.p.be1 : q.p.be2)[n]; rm_old = rm[n]; rv_old = rv[n]; }
    const int chunk = ((K / 16 + NW - 1) / NW) * 16, kbeg = wave * chunk, kend = min(K, kbeg + chunk);
    f32x4 acc[RTM];
#pragma unroll
    for (int i = 0; i < RTM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (layer == 1)
        nt_kloop<RTM, PF, true>(acc, A, K, W + (size_t)(n0 + n16) * K, RT, kbeg, kend, n16, kq, s_scale, s_shift,
                                [&] { bn2d_prepare(d, q, 3, s_scale, s_shift, s_mean, s_inv, scr, (blockIdx.x == 0 && rb == 0) ? q.fdone[3] : nullptr); });
    else nt_kloop<RTM, PF, false>(acc, A, K, W + (size_t)(n0 + n16) * K, RT, kbeg, kend, n16, kq, nullptr, nullptr, [] {});
    TSTAMP(2, 1);
    reduce_waves<RTM, NW>(acc, red, B, RT, wave, n16, kq);
    TSTAMP(2, 2);
    float v[RTM];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < RTM; ++i)
        if (i < RT) { v[i] = red[(rg + 16 * i) * 16 + col] + bias; s += v[i]; }
    if (act) scr[rg * 16 + col] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) tot += scr[g * 16 + col];
    const double mean = tot / B;
    __syncthreads();
    double qq = 0.0;
#pragma unroll
    for (int i = 0; i < RTM; ++i)
        if (i < RT) { const double dl = v[i] - mean; qq += dl * dl; }
    if (act) scr[rg * 16 + col] = qq;
    __syncthreads();
    if (!act) return;
    double M2 = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) M2 += scr[g * 16 + col];
    float *y = layer == 1 ? q.y1 : q.y2, *h = layer == 1 ? q.h1 : q.h2;
    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < RTM; ++i)
            if (i < RT) y[(size_t)(r0 + rg + 16 * i) * N + n] = v[i];
        if (rg == 0) {
            double *fs = q.fstat[layer - 1] + ((size_t)rb * N + n) * 3;
            fs[0] = (double)B; fs[1] = mean; fs[2] = M2;
        }
        return;
    }
    const double var = M2 / B;
#ifdef AZ_TPROBE
    if (blockIdx.x == 0 && t == 0) az_tprobe[(3 + layer) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    const float inv = (float)(1.0 / sqrt(var + BN_EPS)), mu = (float)mean;
#pragma unroll
    for (int i = 0; i < RTM; ++i)
        if (i < RT) {
            const int r = rg + 16 * i;
            y[(size_t)r * N + n] = v[i];
            const float xh = (v[i] - mu) * inv;
            float hh = fmaxf(fmaf(g_, xh, b_), 0.f);
            hh *= dropout_scale(hp, layer, (unsigned)(r * N + n));
            h[(size_t)r * N + n] = hh;
        }
    TSTAMP(2, 3);
    if (rg == 0) {
        (layer == 1 ? q.mu1 : q.mu2)[n] = mu;
        (layer == 1 ? q.iv1 : q.iv2)[n] = inv;
        rm[n] = (float)((1.0 - BN_MOM) * rm_old + BN_MOM * mean);
        rv[n] = (float)((1.0 - BN_MOM) * rv_old + BN_MOM * (M2 / (B > 1 ? B - 1 : 1)));
    }
#ifdef AZ_TPROBE
    if (blockIdx.x == 0 && t == 0) az_tprobe[(3 + layer) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
}

// second stage of the row-split dense forward: workgroup (16 columns, row block).  Every workgroup combines the row blocks' partials of
// its columns in block order (Chan's merge: a fixed order, the same bits in every run), then y -> h = dropout(relu(bn(y))) for its
// rows; row block 0 also writes the batch statistics and the running statistics.
__global__ __launch_bounds__(TPB) void k_fc_fin(TDims d, TPtr q, int layer) {
    const int t = threadIdx.x, col = t & 15, rg = t >> 4, rb = blockIdx.y, r0 = rb * d.RB, Bl = min(d.B - r0, d.RB);
    const int N = layer == 1 ? d.F1 : d.F2, n = blockIdx.x * 16 + col;
    const Hyper hp = *q.hp;
    const float g_ = (layer == 1 ? q.p.g1 : q.p.g2)[n], b_ = (layer == 1 ? q.p.be1 : q.p.be2)[n];
    const float *y = layer == 1 ? q.y1 : q.y2;
    float *h = layer == 1 ? q.h1 : q.h2;
    float yv[NRBMAX * 4 / 4 * 2];  // <= 128 rows per block / 16 row groups = 8 values
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (rg + 16 * i < Bl) yv[i] = y[(size_t)(r0 + rg + 16 * i) * N + n];
    double cn = 0.0, mean = 0.0, M2 = 0.0;
    for (int b = 0; b < d.NRB; ++b) {
        const double *fs = q.fstat[layer - 1] + ((size_t)b * N + n) * 3;
        if (b == 0) { cn = fs[0]; mean = fs[1]; M2 = fs[2]; } else chan_merge(cn, mean, M2, fs[0], fs[1], fs[2]);
    }
    const float inv = (float)(1.0 / sqrt(M2 / cn + BN_EPS)), mu = (float)mean;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (rg + 16 * i < Bl) {
            const int r = r0 + rg + 16 * i;
            const float xh = (yv[i] - mu) * inv;
            float hh = fmaxf(fmaf(g_, xh, b_), 0.f);
            hh *= dropout_scale(hp, layer, (unsigned)(r * N + n));
            h[(size_t)r * N + n] = hh;
        }
    if (rb == 0 && rg == 0) {
        float *rm = layer == 1 ? q.rm1 : q.rm2, *rv = layer == 1 ? q.rv1 : q.rv2;
        (layer == 1 ? q.mu1 : q.mu2)[n] = mu;
        (layer == 1 ? q.iv1 : q.iv2)[n] = inv;
        rm[n] = (float)((1.0 - BN_MOM) * rm[n] + BN_MOM * mean);
        rv[n] = (float)((1.0 - BN_MOM) * rv[n] + BN_MOM * (M2 / (cn > 1.0 ? cn - 1.0 : 1.0)));
    }
}

// ---------------------------------------------------------------------------------------------------------------- heads forward + loss
// 16 rows per workgroup: logits = h2 Wh^T + bh (K split over NW waves), log_softmax over the A policy columns, tanh of the
// value column, the two loss sums of the rows and d loss / d logits (trainer.py:352-354 with B = config.batch_size)
template <int NT, int NW>
__global__ __launch_bounds__(NW * 64) void k_heads_fwd(TDims d, TPtr q) {
    __shared__ float red[NW][16][NT * 16];
    __shared__ float lg[16][NT * 16 + 1];
    __shared__ double rowl[16][2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4;
    const int K = d.F2, r0 = blockIdx.x * 16, NHP = NT * 16, A = d.A;
    if (blockIdx.x == 0) TBEG(6);
    const Hyper hp = *q.hp;
    const int chunk = ((K / 16 + NW - 1) / NW) * 16, kbeg = wave * chunk, kend = min(K, kbeg + chunk);
    // the targets of row t >> 4 (perm -> pi, z: two dependent trips) travel under the K loop
    const int row = (t >> 4) & 15, j = t & 15;
    const long long brow = sample_row(q.perm, hp.perm_off + r0 + row, hp, q.hp);
    float pit[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) pit[i] = j + 16 * i < A ? q.pi[brow * A + j + 16 * i] : 0.f;
    const float zz = (float)q.z[brow];
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 64) {  // four steps of 16 per trip, all loads first: a wave's whole chunk of OthelloNet's K = 512 is one trip
        float4 af[4], bf[4][NT];
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (k0 + 16 * p < kend) {
                af[p] = *(const float4 *)(q.h2 + (size_t)(r0 + n16) * K + k0 + 16 * p + 4 * kq);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[p][nt] = *(const float4 *)(q.p.wh + (size_t)(16 * nt + n16) * K + k0 + 16 * p + 4 * kq);
            }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (k0 + 16 * p < kend)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = MFMA(af[p].x, bf[p][nt].x, acc[nt]); acc[nt] = MFMA(af[p].y, bf[p][nt].y, acc[nt]);
                    acc[nt] = MFMA(af[p].z, bf[p][nt].z, acc[nt]); acc[nt] = MFMA(af[p].w, bf[p][nt].w, acc[nt]);
                }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][16 * nt + n16] = acc[nt][r];
    __syncthreads();
    for (int i = t; i < 16 * NHP; i += NW * 64) {
        const int row = i / NHP, a = i % NHP;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][row][a];
        lg[row][a] = v + q.p.bh[a];
    }
    __syncthreads();
    if (t < 256) {
        const float invB = 1.0f / (float)d.B;
        float mx = -3.0e38f;
        for (int a = j; a < A; a += 16) mx = fmaxf(mx, lg[row][a]);
        for (int o = 8; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float se = 0.f, spi = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (j + 16 * i < A) { se += expf(lg[row][j + 16 * i] - mx); spi += pit[i]; }
        for (int o = 8; o; o >>= 1) { se += __shfl_xor(se, o, 16); spi += __shfl_xor(spi, o, 16); }
        const float lse = mx + logf(se);
        float lpi = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int a = j + 16 * i;
            float g = 0.f;
            if (a < A) {
                const float lp = lg[row][a] - lse;
                lpi = fmaf(pit[i], lp, lpi);
                g = (expf(lp) * spi - pit[i]) * invB;
            } else if (a == A) {
                const float vv = tanhf(lg[row][A]);
                g = 2.0f * (vv - zz) * (1.0f - vv * vv) * invB;
            }
            q.dlog[(size_t)(r0 + row) * NHP + a] = g;
        }
        for (int o = 8; o; o >>= 1) lpi += __shfl_xor(lpi, o, 16);
        if (j == 0) {
            const float vv = tanhf(lg[row][A]);
            rowl[row][0] = -(double)lpi; rowl[row][1] = (double)(vv - zz) * (double)(vv - zz);
        }
    }
    __syncthreads();
    if (t == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int r = 0; r < 16; ++r) { a0 += rowl[r][0]; a1 += rowl[r][1]; }
        q.losspart[blockIdx.x * 2] = (float)a0; q.losspart[blockIdx.x * 2 + 1] = (float)a1;
    }
    if (blockIdx.x == 0) TEND(6);
}

// ---------------------------------------------------------------------------------------------------------------- BatchNorm1d backward
// dh: LDS [B][16], the gradient w.r.t. the layer's output h = dropout(relu(bn(y))) for columns n0..n0+15.  Thread (col, row group)
// of the first 256 threads owns rows rg, rg + 16, ...: mask (h > 0 <=> ReLU passed and the unit was kept), batch sums,
// dz = g inv (dy - mean(dy) - xhat mean(dy xhat)), then the SGD update of the BatchNorm affine pair and of the dense bias (its gradient,
// sum dz, is zero up to rounding).  Every thread of the workgroup must call (barriers inside).
// SPLIT (row-split path): the workgroup holds rows r0 .. r0 + Bl only.  It masks dh -> dy (stored where dz will be), and leaves the three
// sums of its rows per column -- sum dy, sum dy xhat, sum xhat -- in bstat; k_bn1d_bwd_fin combines them in row-block order and forms dz.
template <int RTM, bool SPLIT = false>
AZ_D void bn1d_bwd(const TDims &d, const TPtr &q, const Hyper &hp, int layer, int n0, const float *dh, double *scr /* [3][16][16] */, int r0 = 0, int Bl = 0) {
    const int B = SPLIT ? Bl : d.B, RT = B / 16, t = threadIdx.x, col = t & 15, rg = (t >> 4) & 15, n = n0 + col, N = layer == 1 ? d.F1 : d.F2;
    const bool act = t < 256;
    const float *y = (layer == 1 ? q.y1 : q.y2) + (size_t)r0 * N, *h = (layer == 1 ? q.h1 : q.h2) + (size_t)r0 * N;
    float *dz = (layer == 1 ? q.dz1 : q.dz2) + (size_t)r0 * N;
    const float mu = (layer == 1 ? q.mu1 : q.mu2)[n], iv = (layer == 1 ? q.iv1 : q.iv2)[n];
    float *gam = layer == 1 ? q.p.g1 : q.p.g2, *bet = layer == 1 ? q.p.be1 : q.p.be2, *bia = layer == 1 ? q.p.b1 : q.p.b2;
    float *mgam = layer == 1 ? q.m.g1 : q.m.g2, *mbet = layer == 1 ? q.m.be1 : q.m.be2, *mbia = layer == 1 ? q.m.b1 : q.m.b2;
    const float keep = hp.drop_p > 0.0f ? 1.0f / (1.0f - hp.drop_p) : 1.0f;
    float dyv[RTM], xh[RTM];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (act) {
        float hv[RTM], yv[RTM];
#pragma unroll
        for (int i = 0; i < RTM; ++i)
            if (i < RT) { hv[i] = h[(size_t)(rg + 16 * i) * N + n]; yv[i] = y[(size_t)(rg + 16 * i) * N + n]; }
#pragma unroll
        for (int i = 0; i < RTM; ++i)
            if (i < RT) {
                dyv[i] = hv[i] > 0.0f ? dh[(rg + 16 * i) * 16 + col] * keep : 0.0f;
                xh[i] = (yv[i] - mu) * iv;
                s1 += dyv[i]; s2 += (double)dyv[i] * (double)xh[i];
                if (SPLIT) { s3 += xh[i]; dz[(size_t)(rg + 16 * i) * N + n] = dyv[i]; }
            }
        scr[rg * 16 + col] = s1; scr[256 + rg * 16 + col] = s2;
        if (SPLIT) scr[512 + rg * 16 + col] = s3;
    }
    __syncthreads();
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { S1 += scr[g * 16 + col]; S2 += scr[256 + g * 16 + col]; }
    if (SPLIT) {
        if (act && rg == 0) {
            double S3 = 0.0;
#pragma unroll
            for (int g = 0; g < 16; ++g) S3 += scr[512 + g * 16 + col];
            double *bs = q.bstat[layer - 1] + ((size_t)(r0 / d.RB) * N + n) * 3;
            bs[0] = S1; bs[1] = S2; bs[2] = S3;
        }
        __syncthreads();
        return;
    }
    __syncthreads();
    if (act) {
        const float k = gam[n] * iv, m1 = (float)(S1 / B), m2 = (float)(S2 / B);
        double sd = 0.0;
#pragma unroll
        for (int i = 0; i < RTM; ++i)
            if (i < RT) {
                const float v = k * ((dyv[i] - m1) - xh[i] * m2);
                dz[(size_t)(rg + 16 * i) * N + n] = v;
                sd += v;
            }
        scr[rg * 16 + col] = sd;
    }
    __syncthreads();
    if (act && rg == 0) {
        double SD = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) SD += scr[g * 16 + col];
        sgd(gam + n, mgam + n, (float)S2, hp);
        sgd(bet + n, mbet + n, (float)S1, hp);
        sgd(bia + n, mbia + n, (float)SD, hp);
    }
    __syncthreads();
}

// second stage of the row-split BatchNorm1d backward: workgroup (16 columns, row block).  The row blocks' sums are added in block
// order; dz = g inv (dy - mean(dy) - xhat mean(dy xhat)) in place over the dy the first stage left; row block 0 updates the affine
// pair and the dense bias.  The bias gradient sum dz is zero in exact arithmetic (BatchNorm removes the batch mean); what float32
// leaves of it is the rounding of the two means, k ((S1 - B m1) - m2 Sx) with the ROUNDED m1, m2 -- computed here from the sums
// instead of a third pass over the rows.
__global__ __launch_bounds__(TPB) void k_bn1d_bwd_fin(TDims d, TPtr q, int layer) {
    const int t = threadIdx.x, col = t & 15, rg = t >> 4, rb = blockIdx.y, r0 = rb * d.RB, Bl = min(d.B - r0, d.RB);
    const int N = layer == 1 ? d.F1 : d.F2, n = blockIdx.x * 16 + col;
    const Hyper hp = *q.hp;
    const float *y = layer == 1 ? q.y1 : q.y2;
    float *dz = layer == 1 ? q.dz1 : q.dz2;
    float *gam = layer == 1 ? q.p.g1 : q.p.g2, *bet = layer == 1 ? q.p.be1 : q.p.be2, *bia = layer == 1 ? q.p.b1 : q.p.b2;
    float *mgam = layer == 1 ? q.m.g1 : q.m.g2, *mbet = layer == 1 ? q.m.be1 : q.m.be2, *mbia = layer == 1 ? q.m.b1 : q.m.b2;
    const float mu = (layer == 1 ? q.mu1 : q.mu2)[n], iv = (layer == 1 ? q.iv1 : q.iv2)[n], gv = gam[n];
    float dyv[8], yv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (rg + 16 * i < Bl) { dyv[i] = dz[(size_t)(r0 + rg + 16 * i) * N + n]; yv[i] = y[(size_t)(r0 + rg + 16 * i) * N + n]; }
    double S1 = 0.0, S2 = 0.0, S3 = 0.0;
    for (int b = 0; b < d.NRB; ++b) {
        const double *bs = q.bstat[layer - 1] + ((size_t)b * N + n) * 3;
        S1 += bs[0]; S2 += bs[1]; S3 += bs[2];
    }
    const float k = gv * iv, m1 = (float)(S1 / d.B), m2 = (float)(S2 / d.B);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (rg + 16 * i < Bl) dz[(size_t)(r0 + rg + 16 * i) * N + n] = k * ((dyv[i] - m1) - ((yv[i] - mu) * iv) * m2);
    if (rb == 0 && rg == 0) {
        const double SD = (double)k * ((S1 - (double)d.B * (double)m1) - (double)m2 * S3);
        sgd(gam + n, mgam + n, (float)S2, hp);
        sgd(bet + n, mbet + n, (float)S1, hp);
        sgd(bia + n, mbia + n, (float)SD, hp);
    }
    if (layer == 2) {  // the heads: shares of d Wh / d bh that k_heads_bwd's row blocks left, added in block order; row block rb of this
        const int F2 = d.F2, NHP = d.NHP;  // launch takes the rows a = rb (mod NRB) of its 16 columns
        for (int a = rg; a < d.NH; a += 16) {
            if (a % d.NRB != rb) continue;
            float g = 0.f;
            for (int b = 0; b < d.NRB; ++b) g += q.hwpart[((size_t)b * NHP + a) * F2 + n];
            sgd(q.p.wh + (size_t)a * F2 + n, q.m.wh + (size_t)a * F2 + n, g, hp);
        }
        if (blockIdx.x == 0 && rb == 0 && t < d.NH) {
            float g = 0.f;
            for (int b = 0; b < d.NRB; ++b) g += q.hwpart[(size_t)NRBMAX * NHP * F2 + (size_t)b * NHP + t];
            sgd(q.p.bh + t, q.m.bh + t, g, hp);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- heads backward
// 16 columns j0.. of fc2's output: d h2 = dlog Wh (K = NHP), fc_bn2 backward -> dz2, then the heads' weight gradient for the
// same columns (dWh[a][j] = sum_b dlog[b][a] h2[b][j]) and its update: this workgroup is the only reader and writer of Wh[:, j0..]
// SPLIT: workgroup (16 columns, row block): d h2 and the first stage of fc_bn2's backward for its rows; the heads' weight gradient (a
// sum over ALL rows) is taken by the workgroups of row block 0.
template <int RTM, int NT, bool SPLIT>
__global__ __launch_bounds__(TPB) void k_heads_bwd(TDims d, TPtr q) {
    extern __shared__ __align__(16) float lds[];
    const int rb = SPLIT ? blockIdx.y : 0, r0 = rb * RTM * 16;
    const int B = d.B, Bl = SPLIT ? min(B - r0, RTM * 16) : B, RT = Bl / 16, NHP = NT * 16, F2 = d.F2;
    float *dh = lds;
    double *scr = (double *)(dh + Bl * 16);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4, j0 = blockIdx.x * 16;
    if (blockIdx.x == 1 && rb == 0) TBEG(7);
    const Hyper hp = *q.hp;
    float wb[NT][4];  // Wh[a][j0 + n16] for the lane's rows a = 16 nt + 4 kq + i: loaded once, used by every row tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) wb[nt][i] = q.p.wh[(size_t)(16 * nt + 4 * kq + i) * F2 + j0 + n16];
    for (int rt = wave; rt < RT; rt += 4) {
        float4 af[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt] = *(const float4 *)(q.dlog + (size_t)(r0 + 16 * rt + n16) * NHP + 16 * nt + 4 * kq);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc = MFMA(af[nt].x, wb[nt][0], acc); acc = MFMA(af[nt].y, wb[nt][1], acc); acc = MFMA(af[nt].z, wb[nt][2], acc); acc = MFMA(af[nt].w, wb[nt][3], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[(16 * rt + 4 * kq + r) * 16 + n16] = acc[r];
    }
    __syncthreads();
    bn1d_bwd<RTM, SPLIT>(d, q, hp, 2, j0, dh, scr, r0, Bl);
    if (SPLIT) {
        // the heads' weight gradient is a sum over ALL rows: this workgroup adds up its own rows (d Wh[a][j0..] over r0 .. r0 + Bl) and
        // leaves the share in hwpart[rb]; k_bn1d_bwd_fin (layer 2, row block 0) adds the shares in block order and updates Wh, bh
        float *wp = q.hwpart + (size_t)rb * NHP * F2;
        for (int at = wave; at < NT; at += 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int b0 = 0; b0 < Bl; b0 += 32) {
                float a[8], h[8];
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    if (b0 + 4 * p < Bl) { a[p] = q.dlog[(size_t)(r0 + b0 + 4 * p + kq) * NHP + 16 * at + n16]; h[p] = q.h2[(size_t)(r0 + b0 + 4 * p + kq) * F2 + j0 + n16]; }
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    if (b0 + 4 * p < Bl) acc = MFMA(a[p], h[p], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) wp[(size_t)(16 * at + 4 * kq + r) * F2 + j0 + n16] = acc[r];
        }
        if (blockIdx.x == 0) {  // d bh: column sums of dlog over this block's rows, 16 row groups then LDS
            float *bp = q.hwpart + (size_t)NRBMAX * NHP * F2 + (size_t)rb * NHP;
            const int col = t & 15, rg = t >> 4;
            for (int at = 0; at < NT; ++at) {
                double g = 0.0;
                for (int b = rg; b < Bl; b += 16) g += q.dlog[(size_t)(r0 + b) * NHP + 16 * at + col];
                scr[rg * 16 + col] = g;
                __syncthreads();
                if (rg == 0) {
                    double G = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) G += scr[k * 16 + col];
                    bp[16 * at + col] = (float)G;
                }
                __syncthreads();
            }
        }
        return;
    }
    for (int at = wave; at < NT; at += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b0 = 0; b0 < B; b0 += 32) {  // eight steps of 4 rows per trip: all loads first
            float a[8], h[8];
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (b0 + 4 * p < B) { a[p] = q.dlog[(size_t)(b0 + 4 * p + kq) * NHP + 16 * at + n16]; h[p] = q.h2[(size_t)(b0 + 4 * p + kq) * F2 + j0 + n16]; }
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (b0 + 4 * p < B) acc = MFMA(a[p], h[p], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 16 * at + 4 * kq + r;
            if (a < d.NH) sgd(q.p.wh + (size_t)a * F2 + j0 + n16, q.m.wh + (size_t)a * F2 + j0 + n16, acc[r], hp);
        }
    }
    if (blockIdx.x == 0)
        for (int a = t; a < d.NH; a += TPB) {
            double g = 0.0;
#pragma unroll 8
            for (int b = 0; b < B; ++b) g += q.dlog[(size_t)b * NHP + a];
            sgd(q.p.bh + a, q.m.bh + a, (float)g, hp);
        }
    if (blockIdx.x == 1 && rb == 0) TEND(7);
}

// ---------------------------------------------------------------------------------------------------------------- dense data gradient
// d h1 = dz2 W2 (the not yet updated W2) for 16 columns of fc1's output and all rows, then fc_bn1 backward
template <int RTM, int NW, int PF, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void k_fc_dgrad(TDims d, TPtr q) {
    extern __shared__ __align__(16) float lds[];
    const int rb = SPLIT ? blockIdx.y : 0, r0 = rb * RTM * 16, Bl = SPLIT ? min(d.B - r0, RTM * 16) : d.B;
    float *red = lds;
    double *scr = (double *)(red + NW * Bl * 16 + 4 * 32);
    if (blockIdx.x == 0 && rb == 0) TBEG(8);
    const Hyper hp = *q.hp;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4, RT = Bl / 16, J = d.F2;
    const int chunk = ((J / 16 + NW - 1) / NW) * 16, jbeg = wave * chunk, jend = min(J, jbeg + chunk);
    f32x4 acc[RTM];
#pragma unroll
    for (int i = 0; i < RTM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    nn_kloop<RTM, PF>(acc, q.dz2 + (size_t)r0 * J, J, q.p.w2 + blockIdx.x * 16, d.F1, RT, jbeg, jend, n16, kq, [] {});
    reduce_waves<RTM, NW>(acc, red, Bl, RT, wave, n16, kq);
    bn1d_bwd<RTM, SPLIT>(d, q, hp, 1, blockIdx.x * 16, red, scr, r0, Bl);
    if (blockIdx.x == 0 && rb == 0) TEND(8);
}

// ---------------------------------------------------------------------------------------------------------------- dense weight gradient
// dW[j][k] = sum_b dZ[b][j] X[b][k] + SGD update, one 32 x 32 tile per wave (2 x 2 MFMA tiles, K = b in steps of 4, eight steps'
// operands loaded before their MFMAs).  layer 1: X = relu(bn4(c4)) formed on load.
template <int WB>  // steps of 4 rows whose operands are loaded before their MFMAs (4 where 16 waves share the register file, else 8)
AZ_D void fc_wgrad_tile(const TDims &d, const TPtr &q, const Hyper &hp, int layer, int tile, const float *s_scale, const float *s_shift) {
    const int lane = threadIdx.x & 63, n16 = lane & 15, kq = lane >> 4;
    const int N = layer == 1 ? d.F1 : d.F2, K = layer == 1 ? d.FIN : d.F1, TK = K / 32;
    if (tile >= (N / 32) * TK) return;
    const int j0 = 32 * (tile / TK), k0 = 32 * (tile % TK);
    const float *dZ = layer == 1 ? q.dz1 : q.dz2, *X = layer == 1 ? q.c[3] : q.h1;
    float *W = layer == 1 ? q.p.w1 : q.p.w2, *M = layer == 1 ? q.m.w1 : q.m.w2;
    float sc0 = 1.f, sh0 = 0.f, sc1 = 1.f, sh1 = 0.f;
    if (layer == 1) { sc0 = s_scale[n16]; sh0 = s_shift[n16]; sc1 = s_scale[16 + n16]; sh1 = s_shift[16 + n16]; }  // k0 is a multiple of 32: channel = 16 nt + n16
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wo[2][2][4], mo[2][2][4];  // the tile's weights and momenta: in flight under the K loop
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t idx = (size_t)(j0 + 16 * mt + 4 * kq + r) * K + k0 + 16 * nt + n16;
                wo[mt][nt][r] = W[idx]; mo[mt][nt][r] = M[idx];
            }
    for (int b0 = 0; b0 < d.B; b0 += 4 * WB) {
        float a0[WB], a1[WB], x0[WB], x1[WB];
#pragma unroll
        for (int p = 0; p < WB; ++p)
            if (b0 + 4 * p < d.B) {
                const float *zp = dZ + (size_t)(b0 + 4 * p + kq) * N + j0 + n16, *xp = X + (size_t)(b0 + 4 * p + kq) * K + k0 + n16;
                a0[p] = zp[0]; a1[p] = zp[16]; x0[p] = xp[0]; x1[p] = xp[16];
            }
#pragma unroll
        for (int p = 0; p < WB; ++p)
            if (b0 + 4 * p < d.B) {
                float u0 = x0[p], u1 = x1[p];
                if (layer == 1) { u0 = fmaxf(fmaf(u0, sc0, sh0), 0.f); u1 = fmaxf(fmaf(u1, sc1, sh1), 0.f); }
                acc[0][0] = MFMA(a0[p], u0, acc[0][0]); acc[0][1] = MFMA(a0[p], u1, acc[0][1]);
                acc[1][0] = MFMA(a1[p], u0, acc[1][0]); acc[1][1] = MFMA(a1[p], u1, acc[1][1]);
            }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t idx = (size_t)(j0 + 16 * mt + 4 * kq + r) * K + k0 + 16 * nt + n16;
                const float gg = fmaf(hp.wd, wo[mt][nt][r], acc[mt][nt][r]);
                const float mm = fmaf(hp.momentum, mo[mt][nt][r], gg);
                M[idx] = mm;
                W[idx] = fmaf(-hp.lr, mm, wo[mt][nt][r]);
            }
}

static inline int fc_wgrad_blocks(int N, int K, int NW) { return ((N / 32) * (K / 32) + NW - 1) / NW; }

// The same tile with its K (the batch rows) split over FOUR waves: wave (tile, ks) multiplies rows [ks B/4, (ks+1) B/4), the four
// partial tiles meet in LDS (part[wave][32 x 32]) and wave ks = 0 adds them in order and updates.  At batch 512 one wave walking all
// rows is sixteen dependent trips of eight MFMA steps; a quarter each is four.  Every wave of the workgroup must call (one barrier).
AZ_D void fc_wgrad_tile_ks(const TDims &d, const TPtr &q, const Hyper &hp, int layer, int tile, int ks, float *part, const float *s_scale = nullptr,
                           const float *s_shift = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n16 = lane & 15, kq = lane >> 4;
    const int N = layer == 1 ? d.F1 : d.F2, K = layer == 1 ? d.FIN : d.F1, TK = K / 32;
    const bool live = tile < (N / 32) * TK;
    const int j0 = 32 * (tile / TK), k0 = 32 * (tile % TK);
    const float *dZ = layer == 1 ? q.dz1 : q.dz2, *X = layer == 1 ? q.c[3] : q.h1;
    float *W = layer == 1 ? q.p.w1 : q.p.w2, *M = layer == 1 ? q.m.w1 : q.m.w2;
    float sc0 = 1.f, sh0 = 0.f, sc1 = 1.f, sh1 = 0.f;  // layer 1: X = relu(bn4(c4)) formed on load; k0 is a multiple of 32: channel = 16 nt + n16
    if (layer == 1) { sc0 = s_scale[n16]; sh0 = s_shift[n16]; sc1 = s_scale[16 + n16]; sh1 = s_shift[16 + n16]; }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wo[2][2][4], mo[2][2][4];
    if (live && ks == 0)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t idx = (size_t)(j0 + 16 * mt + 4 * kq + r) * K + k0 + 16 * nt + n16;
                    wo[mt][nt][r] = W[idx]; mo[mt][nt][r] = M[idx];
                }
    if (live) {
        const int per = ((d.B / 4 + 3) / 4) * 4, bbeg = ks * per, bend = min(d.B, bbeg + per);
        for (int b0 = bbeg; b0 < bend; b0 += 32) {
            float a0[8], a1[8], x0[8], x1[8];
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (b0 + 4 * p < bend) {
                    const float *zp = dZ + (size_t)(b0 + 4 * p + kq) * N + j0 + n16, *xp = X + (size_t)(b0 + 4 * p + kq) * K + k0 + n16;
                    a0[p] = zp[0]; a1[p] = zp[16]; x0[p] = xp[0]; x1[p] = xp[16];
                }
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (b0 + 4 * p < bend) {
                    float u0 = x0[p], u1 = x1[p];
                    if (layer == 1) { u0 = fmaxf(fmaf(u0, sc0, sh0), 0.f); u1 = fmaxf(fmaf(u1, sc1, sh1), 0.f); }
                    acc[0][0] = MFMA(a0[p], u0, acc[0][0]); acc[0][1] = MFMA(a0[p], u1, acc[0][1]);
                    acc[1][0] = MFMA(a1[p], u0, acc[1][0]); acc[1][1] = MFMA(a1[p], u1, acc[1][1]);
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(size_t)wave * 1024 + ((mt * 2 + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    }
    __syncthreads();
    if (!live || ks != 0) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((mt * 2 + nt) * 4 + r) * 64 + lane;
                const float g = ((part[(size_t)wave * 1024 + o] + part[(size_t)(wave + 1) * 1024 + o]) + part[(size_t)(wave + 2) * 1024 + o]) + part[(size_t)(wave + 3) * 1024 + o];
                const size_t idx = (size_t)(j0 + 16 * mt + 4 * kq + r) * K + k0 + 16 * nt + n16;
                const float gg = fmaf(hp.wd, wo[mt][nt][r], g);
                const float mm = fmaf(hp.momentum, mo[mt][nt][r], gg);
                M[idx] = mm;
                W[idx] = fmaf(-hp.lr, mm, wo[mt][nt][r]);
            }
}

// k_mix1: [0, nw) fc2 weight gradient + update (reads dz2, h1; writes W2 -- the data gradient through W2 ran in the launch before) |
//         [nw, nw + FIN/16) d a4 = dz1 W1 for 16 input columns (old W1), ReLU mask of a4, dy4 and the per-column sums bn4's backward needs
// SPLIT: the d a4 part has a workgroup per (16 input columns, row block): block index nw + cb * NRB + rb; its column sums go to the
// row block's slice of colsum (the consumers add the slices in order).
template <int RTM, int NW, int PF, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void k_mix1(TDims d, TPtr q, int nw) {
    extern __shared__ __align__(16) float lds[];
    const Hyper hp = *q.hp;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4;
    // SPLIT: the data-gradient workgroups come FIRST in the grid (they are the longer chain: K = F1), the weight-gradient tiles fill in behind
    const int ndg = SPLIT ? (int)gridDim.x - nw : 0, wid = SPLIT ? (int)blockIdx.x - ndg : (int)blockIdx.x;
    if (SPLIT ? wid >= 0 : wid < nw) {
        if (wid == 0) TBEG(9);
        if (SPLIT) fc_wgrad_tile_ks(d, q, hp, 2, wid * (NW / 4) + (wave >> 2), wave & 3, lds);
        else fc_wgrad_tile<(NW >= 16 ? 4 : 8)>(d, q, hp, 2, wid * NW + wave, nullptr, nullptr);
        if (wid == 0) TEND(9);
        return;
    }
    if ((int)blockIdx.x == (SPLIT ? 0 : nw)) TBEG(10);
    const int id = SPLIT ? (int)blockIdx.x : (int)blockIdx.x - nw, rb = SPLIT ? id % d.NRB : 0, r0 = rb * RTM * 16;
    const int B = SPLIT ? min(d.B - r0, RTM * 16) : d.B, RT = B / 16, FIN = d.FIN, J = d.F1, k0 = (SPLIT ? id / d.NRB : id) * 16;
    float *red = lds, *s_scale = red + NW * B * 16, *s_shift = s_scale + 32, *s_mean = s_shift + 32, *s_inv = s_mean + 32;
    double *scr = (double *)(s_inv + 32);
    const int chunk = ((J / 16 + NW - 1) / NW) * 16, jbeg = wave * chunk, jend = min(J, jbeg + chunk);
    f32x4 acc[RTM];
#pragma unroll
    for (int i = 0; i < RTM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    nn_kloop<RTM, PF>(acc, q.dz1 + (size_t)r0 * J, J, q.p.w1 + k0, FIN, RT, jbeg, jend, n16, kq,
                      [&] { bn2d_from_done(q.fdone[3], s_scale, s_shift, s_mean, s_inv); });  // conv4's finished statistics arrive under the first loads
    reduce_waves<RTM, NW>(acc, red, B, RT, wave, n16, kq);
    const int col = t & 15, rg = (t >> 4) & 15, k = k0 + col, ch = k & 31;
    double s1 = 0.0, s2 = 0.0;
    if (t < 256) {
        for (int r = rg; r < B; r += 16) {
            const float cv = q.c[3][(size_t)(r0 + r) * FIN + k];
            const float v = fmaf(cv, s_scale[ch], s_shift[ch]) > 0.0f ? red[r * 16 + col] : 0.0f;
            q.dy[3][(size_t)(r0 + r) * FIN + k] = v;
            s1 += v; s2 += (double)v * (double)((cv - s_mean[ch]) * s_inv[ch]);
        }
        scr[rg * 16 + col] = s1; scr[256 + rg * 16 + col] = s2;
    }
    __syncthreads();
    if (t < 16) {
        double S1 = 0.0, S2 = 0.0;
        for (int g = 0; g < 16; ++g) { S1 += scr[g * 16 + col]; S2 += scr[256 + g * 16 + col]; }
        q.colsum[2 * ((size_t)rb * FIN + k)] = S1; q.colsum[2 * ((size_t)rb * FIN + k) + 1] = S2;
    }
    if ((int)blockIdx.x == (SPLIT ? 0 : nw)) TEND(10);
}

// ---------------------------------------------------------------------------------------------------------------- conv backward
// conv l (1..3 = conv2..conv4) for the boards of one workgroup.  Per board, with dz = BatchNorm backward of dy_l formed on load:
//   data gradient   d a_{l-1}[q][ic] = sum_tap sum_oc dz[q - tap + pad][oc] W[tap][ic][oc]   (M = input positions, N = ic, K = tap x oc)
//                   -> masked by a_{l-1} > 0 -> dy_{l-1}, and the two sums BatchNorm l-1's backward needs
//   weight gradient dW[tap][ic][oc] += sum_p a_{l-1}[p + tap - pad][ic] dz[p][oc]                (M = ic, N = oc, K = positions), one
//                   (ic tile, oc tile) pair per wave, nine accumulators that live across the workgroup's boards
// LDS: wl[288][LDP] | dzp[<=100][LDP] dz with a zero halo of 2 - pad | ap[<=100][LDP] a_{l-1} with a zero halo of pad | xh[64][LDP] | dpl[64][LDP]
#define CONV_BWD_LDS_FLOATS (288 * LDP + 100 * LDP + 100 * LDP + 64 * LDP + 64 * LDP + 10 * 32 + 192 + 64 * LDP + 128 + 64)
#define CONV_BWD_LDS_BYTES (CONV_BWD_LDS_FLOATS * 4 + 2048 * 8)
AZ_D void conv_bwd_body(const TDims &d, const TPtr &q, int l, int bidx, int nblocks, float *lds) {
    float *wl = lds, *dzp = wl + 288 * LDP, *ap = dzp + 100 * LDP, *xh = ap + 100 * LDP, *dpl = xh + 64 * LDP;
    float *k1 = dpl + 64 * LDP, *sh_o = k1 + 32, *mean_o = sh_o + 32, *inv_o = mean_o + 32, *k2 = inv_o + 32, *k3 = k2 + 32;
    float *sc_i = k3 + 32, *sh_i = sc_i + 32, *mean_i = sh_i + 32, *inv_i = mean_i + 32;
    // l = 1 with d.F1B: the board's masked data gradient dy0 is kept as a plane (d0p) beside xhat0 (xh) and the board's input x (xin, haloed):
    // conv1's weight gradient sum_p x[p + tap] dz0[p][oc] is LINEAR in three per-board moments -- A = sum x_tap dy0, C = sum x_tap xhat0,
    // Bx = sum x_tap -- because dz0 = k1 (dy0 - k2 - k3 xhat0) with batch-wide k2, k3 that only the NEXT launch knows: the moments are
    // formed here, k_update combines them (no k_conv1_bwd launch)
    float *d0p = inv_i + 32 + 192, *xin = d0p + 64 * LDP;
    int *p1off = (int *)(xin + 128);
    double *scr = (double *)(p1off + 64);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n16 = lane & 15, kq = lane >> 4, ch = t & 31, grp = t >> 5;
    const bool fold1 = l == 1 && d.F1B;
    const int Hin = conv_hin(d, l), Win = conv_win(d, l), Hout = conv_hout(d, l), Wout = conv_wout(d, l), pad = l == 1 ? 1 : 0, hal = 2 - pad;
    const int Pin = Hin * Win, Pout = Hout * Wout, WZ = Wout + 2 * hal, WA = Win + 2 * pad, MTin = (Pin + 15) / 16;
    const int GSZ = 9 * 32 * 32 + 32;
    TSTAMP(3, 0);
    const float *dyo = q.dy[l], *co = q.c[l], *ci = q.c[l - 1];
    float *dyi = q.dy[l - 1];
    // a board's three planes travel through registers: the first board's loads go out before anything else, the next board's while
    // this one is being computed (<= 64 positions x 32 channels / 256 threads = 8 values per plane and thread)
    float r_dy[8], r_co[8], r_ci[8], r_x = 0.0f;
    auto fetch = [&](int b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + TPB * k;
            if (i < Pout * 32) { r_dy[k] = dyo[(size_t)b * Pout * 32 + i]; r_co[k] = co[(size_t)b * Pout * 32 + i]; }
            if (i < Pin * 32) r_ci[k] = ci[(size_t)b * Pin * 32 + i];
        }
        if (fold1 && t < Pin) r_x = q.x0[(size_t)b * Pin + t];
    };
    const int S = d.S, part = bidx % S, bfirst = bidx / S, bstride = nblocks / S;
    if (bfirst < d.B) fetch(bfirst);
    float4 wreg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wreg[i] = *(const float4 *)(q.p.cw[l] + (i * TPB + t) * 4);
    {   // the backward sums of BatchNorm l are fresh partials and are walked here; the two FORWARD statistics this kernel needs (layers l
        // and l - 1) were finished by the forward pass: 2 x 128 floats from q.fdone instead of two more walks over NB partials
        double b0, b1;
        bsum_walk(d, q, l, b0, b1);
        double *o = scr + (size_t)(grp * 32 + ch) * 8;
        o[3] = b0; o[4] = b1;
    }
    if (t < 32) { fdone_load(q.fdone[l], k1, sh_o, mean_o, inv_o); fdone_load(q.fdone[l - 1], sc_i, sh_i, mean_i, inv_i); }  // k1 = gamma_l / sqrt(var_l + eps)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int e = (i * TPB + t) * 4;
        *(float4 *)(wl + (e >> 5) * LDP + (e & 31)) = wreg[i];
    }
    for (int i = t; i < 200 * LDP; i += TPB) dzp[i] = 0.0f;  // dzp and ap (contiguous): the halos stay zero
    int *pzo = (int *)(inv_i + 32), *pao = pzo + 64, *pio = pao + 64;  // output position -> offset of its cell in dzp / of its top-left tap in ap; input position -> its cell in ap
    if (t < Pout) { pzo[t] = ((t / Wout + hal) * WZ + t % Wout + hal) * LDP; pao[t] = ((t / Wout) * WA + t % Wout) * LDP; }
    if (t < Pin) pio[t] = ((t / Win + pad) * WA + t % Win + pad) * LDP;
    if (fold1) {
        if (t < Pin) p1off[t] = (t / Win) * WA + t % Win;  // top-left tap of input position t in the haloed input plane
        if (t < 128) xin[t] = 0.0f;
    }
    __syncthreads();
    if (t < 32) {
        double S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) { S1 += scr[(g * 32 + ch) * 8 + 3]; S2 += scr[(g * 32 + ch) * 8 + 4]; }
        const double Nn = (double)d.B * Pout;
        k2[ch] = (float)(S1 / Nn); k3[ch] = (float)(S2 / Nn);
        if (bidx == 0) { q.bdone[l][ch] = S1; q.bdone[l][32 + ch] = S2; }
    }
    __syncthreads();
    TSTAMP(3, 1);
    f32x4 wacc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wmt = wave >> 1, wnt = wave & 1;
    double s1 = 0.0, s2 = 0.0, sb = 0.0;  // this thread's share of sum dy, sum dy xhat (channel t & 31, rows t >> 5 and + 8 of every tile) and of sum dz
    f32x4 facc = {0.f, 0.f, 0.f, 0.f}, fbx = {0.f, 0.f, 0.f, 0.f};  // fold1: this wave's 16 x 16 tile of the moment products (rows 4 kq + r = tap, columns 16 (wave & 1) + n16)
    for (int b = bfirst; b < d.B; b += bstride) {
        if (fold1 && t < Pin) xin[(t / Win + 1) * WA + t % Win + 1] = r_x;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + TPB * k, p = i >> 5, cc = i & 31;
            if (i < Pout * 32) {
                const float xo = (r_co[k] - mean_o[cc]) * inv_o[cc];
                dzp[pzo[p] + cc] = k1[cc] * ((r_dy[k] - k2[cc]) - xo * k3[cc]);
            }
            if (i < Pin * 32) {
                const float a = fmaf(r_ci[k], sc_i[cc], sh_i[cc]);
                ap[pio[p] + cc] = a > 0.0f ? a : 0.0f;
                xh[p * LDP + cc] = (r_ci[k] - mean_i[cc]) * inv_i[cc];
            }
        }
        if (b + bstride < d.B) fetch(b + bstride);
        __syncthreads();
        TSTAMP(3, 2);
        const int my_tiles = (MTin - part + S - 1) / S;
        if (my_tiles >= 3) {  // data gradient, three or four tiles: one per wave with its whole K (no reduction)
            const int mt = part + S * wave;
            if (mt < MTin) {
                int m = 16 * mt + n16;
                if (m >= Pin) m = Pin - 1;
                const int qr = m / Win, qc = m % Win;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float *zp = dzp + ((qr - tap / 3 + pad + hal) * WZ + qc - tap % 3 + pad + hal) * LDP + kq;
                    const float *wp = wl + (tap * 32 + n16) * LDP + kq;
#pragma unroll
                    for (int ocb = 0; ocb < 8; ++ocb) {
                        const float a = zp[4 * ocb];
                        acc0 = MFMA(a, wp[4 * ocb], acc0);
                        acc1 = MFMA(a, wp[16 * LDP + 4 * ocb], acc1);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = 16 * mt + 4 * kq + r;
                    if (mm < Pin) {
                        const float v0 = ap[pio[mm] + n16] > 0.0f ? acc0[r] : 0.0f, v1 = ap[pio[mm] + 16 + n16] > 0.0f ? acc1[r] : 0.0f;
                        dyi[((size_t)b * Pin + mm) * 32 + n16] = v0; dyi[((size_t)b * Pin + mm) * 32 + 16 + n16] = v1;
                        dpl[mm * LDP + n16] = v0; dpl[mm * LDP + 16 + n16] = v1;
                        if (fold1) { d0p[mm * LDP + n16] = v0; d0p[mm * LDP + 16 + n16] = v1; }
                    }
                }
            }
            __syncthreads();
            for (int mm = grp; mm < Pin; mm += 8)  // this workgroup's rows: tiles part, part + S, ...
                if (((mm >> 4) - part) % S == 0) { const double v = dpl[mm * LDP + ch]; s1 += v; s2 += v * (double)xh[mm * LDP + ch]; }
            __syncthreads();
        } else
        for (int mt = part; mt < MTin; mt += S) {  // one or two tiles: a tile's 72 k-steps split over the four waves
            int m = 16 * mt + n16;
            if (m >= Pin) m = Pin - 1;
            const int qr = m / Win, qc = m % Win;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 18; ++jj) {
                const int ks = 18 * wave + jj, tap = ks >> 3, ocb = ks & 7, tr = (tap * 11) >> 5, tc = tap - 3 * tr;
                const float a = dzp[((qr - tr + pad + hal) * WZ + qc - tc + pad + hal) * LDP + 4 * ocb + kq];
                const float *wp = wl + (tap * 32 + n16) * LDP + 4 * ocb + kq;
                acc0 = MFMA(a, wp[0], acc0);
                acc1 = MFMA(a, wp[16 * LDP], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { dpl[(wave * 16 + 4 * kq + r) * LDP + n16] = acc0[r]; dpl[(wave * 16 + 4 * kq + r) * LDP + 16 + n16] = acc1[r]; }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = (t >> 5) + 8 * h, mm = 16 * mt + rr;
                if (mm < Pin) {
                    const float g = (dpl[rr * LDP + ch] + dpl[(16 + rr) * LDP + ch]) + (dpl[(32 + rr) * LDP + ch] + dpl[(48 + rr) * LDP + ch]);
                    const float v = ap[pio[mm] + ch] > 0.0f ? g : 0.0f;
                    dyi[((size_t)b * Pin + mm) * 32 + ch] = v;
                    if (fold1) d0p[mm * LDP + ch] = v;
                    s1 += v; s2 += (double)v * (double)xh[mm * LDP + ch];
                }
            }
            __syncthreads();
        }
        if (fold1) {  // (d0p is complete: both paths above end with a barrier)
            // three small products on the matrix cores, K = the board's positions: rows = the nine shifted planes x_t (+ row 9 = ones, rows
            // 10..15 = 0), columns = 32 channels of dy0 (waves 0, 1: A), of xhat0 (waves 2, 3: C; its row 9 is sum xhat), and, on wave 0, a
            // column of ones (Bx[t] = sum x_t).  Accumulators live across the workgroup's boards.
            const float *pln = (wave >> 1) ? xh : d0p;
            const int toff = n16 < 9 ? (n16 / 3) * WA + n16 % 3 : 0, nt = wave & 1;
            for (int p0 = 0; p0 < Pin; p0 += 16) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + 4 * u + kq, ok = p < Pin;
                    av[u] = !ok ? 0.0f : (n16 < 9 ? xin[p1off[p] + toff] : (n16 == 9 ? 1.0f : 0.0f));
                    bv[u] = ok ? pln[p * LDP + 16 * nt + n16] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    facc = MFMA(av[u], bv[u], facc);
                    if (wave == 0) fbx = MFMA(av[u], 1.0f, fbx);
                }
            }
        }
        TSTAMP(3, 3);
        for (int p0 = 0; p0 < Pout; p0 += 8) {  // weight gradient of this workgroup's taps ((tap & (S - 1)) == part): K = output positions, four per MFMA,
            // two k-steps per trip with every LDS operand read before the first MFMA (offsets from the tables: no division here)
            const int pA = p0 + kq, pB = p0 + 4 + kq, okA = pA < Pout, okB = pB < Pout;
            const int ozA = okA ? pzo[pA] : 0, oaA = okA ? pao[pA] : 0, ozB = okB ? pzo[pB] : 0, oaB = okB ? pao[pB] : 0;
            const float bzA = okA ? dzp[ozA + 16 * wnt + n16] : 0.0f, bzB = okB ? dzp[ozB + 16 * wnt + n16] : 0.0f;
            float aA[9], aB[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                if ((tap & (S - 1)) == part) {
                    aA[tap] = ap[oaA + ((tap / 3) * WA + tap % 3) * LDP + 16 * wmt + n16];
                    aB[tap] = ap[oaB + ((tap / 3) * WA + tap % 3) * LDP + 16 * wmt + n16];
                }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                if ((tap & (S - 1)) == part) { wacc[tap] = MFMA(aA[tap], bzA, wacc[tap]); wacc[tap] = MFMA(aB[tap], bzB, wacc[tap]); }
        }
        TSTAMP(3, 4);
        if (part == 0)
            for (int p = grp; p < Pout; p += 8) sb += dzp[((p / Wout + hal) * WZ + p % Wout + hal) * LDP + ch];
        __syncthreads();
    }
    TSTAMP(3, 5);
    float *gp = q.gw[l] + (size_t)bfirst * GSZ;  // the S workgroups of a board group fill disjoint tap slices of one partial
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        if ((tap & (S - 1)) == part)
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[(tap * 32 + 16 * wmt + 4 * kq + r) * 32 + 16 * wnt + n16] = wacc[tap][r];
    if (fold1) {
        float *g1p = q.gw[0] + (size_t)bidx * C1P;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tap = 4 * kq + r, oc = 16 * (wave & 1) + n16;
            if (tap < 9) g1p[(wave >> 1) * 288 + tap * 32 + oc] = facc[r];
            else if (tap == 9 && (wave >> 1)) g1p[576 + oc] = facc[r];   // sum xhat
            if (wave == 0 && n16 == 0 && tap < 9) g1p[608 + tap] = fbx[r];
        }
    }
    scr[grp * 32 + ch] = s1; scr[256 + grp * 32 + ch] = s2; scr[512 + grp * 32 + ch] = sb;
    __syncthreads();
    if (t < 32) {
        double S1 = 0.0, S2 = 0.0, SB = 0.0;
        for (int g = 0; g < 8; ++g) { S1 += scr[g * 32 + ch]; S2 += scr[256 + g * 32 + ch]; SB += scr[512 + g * 32 + ch]; }
        q.bpart[l - 1][(size_t)bidx * 64 + ch] = S1; q.bpart[l - 1][(size_t)bidx * 64 + 32 + ch] = S2;
        if (part == 0) gp[9 * 32 * 32 + ch] = (float)SB;
    }
    TSTAMP(3, 6);
}

__global__ __launch_bounds__(TPB) void k_conv_bwd(TDims d, TPtr q, int l) {
    extern __shared__ __align__(16) float lds[];
    if (blockIdx.x == 0) TBEG(12 + l);
    conv_bwd_body(d, q, l, blockIdx.x, gridDim.x, lds);
    if (blockIdx.x == 0) TEND(12 + l);
}

// k_mix2: [0, nw) fc1 weight gradient + update (the data gradient through W1 ran in the launch before) | [nw, nw + NB) conv4 backward
__global__ __launch_bounds__(TPB) void k_mix2(TDims d, TPtr q, int nw) {
    extern __shared__ __align__(16) float lds[];
    if ((int)blockIdx.x < nw) {
        if (blockIdx.x == 0) TBEG(11);
        float *s_scale = lds, *s_shift = s_scale + 32, *s_mean = s_shift + 32, *s_inv = s_mean + 32;
        double *scr = (double *)(s_inv + 32);
        const Hyper hp = *q.hp;
        bn2d_from_done(q.fdone[3], s_scale, s_shift, s_mean, s_inv);
        (void)scr;
        fc_wgrad_tile<8>(d, q, hp, 1, blockIdx.x * 4 + (threadIdx.x >> 6), s_scale, s_shift);
        if (blockIdx.x == 0) TEND(11);
        return;
    }
    if ((int)blockIdx.x == nw) TBEG(12);
    conv_bwd_body(d, q, 3, (int)blockIdx.x - nw, (int)gridDim.x - nw, lds);
    if ((int)blockIdx.x == nw) TEND(12);
}

// ---------------------------------------------------------------------------------------------------------------- conv1 backward
// dW1[tap][oc] = sum over boards and positions of x0[p + tap - 1] dz1[p][oc] (VALU: K = 9 taps x 1 input channel); thread i owns
// element i (and, for i < 32, element 256 + i) of the 288 + 32 partial
// Row-split path (batch > 128): the workgroups behind the first NB are fc1's weight-gradient tiles (one 32 x 32 tile each, K = the batch
// rows split over the four waves).  In k_mix2 they share a launch with conv4's backward, whose 107 KB of LDS allow ONE workgroup per CU:
// 128 tile workgroups of sixteen dependent trips queued with 256 board workgroups for 1.5 rounds; here they need 23 KB and run beside
// the conv1 workgroups.  (W1 was last read by k_mix1's data gradient and is next read by the following step's forward.)
#define CONV1_WG_LDS_BYTES (4 * 1024 * 4 + 4 * 32 * 4 + 768 * 8)
__global__ __launch_bounds__(TPB) void k_conv1_bwd(TDims d, TPtr q) {
    if ((int)blockIdx.x >= d.NB) {
        extern __shared__ __align__(16) float wg_lds[];
        float *part = wg_lds, *s_scale = part + 4 * 1024, *s_shift = s_scale + 32, *s_mean = s_shift + 32, *s_inv = s_mean + 32;
        double *wscr = (double *)(s_inv + 32);
        const Hyper hp = *q.hp;
        bn2d_from_done(q.fdone[3], s_scale, s_shift, s_mean, s_inv);
        (void)wscr;
        fc_wgrad_tile_ks(d, q, hp, 1, (int)blockIdx.x - d.NB, threadIdx.x >> 6, part, s_scale, s_shift);
        return;
    }
    __shared__ float xin[10 * 10];
    __shared__ float dzl[64 * LDP];
    __shared__ float k1[32], sh_o[32], mean_o[32], inv_o[32], k2[32], k3[32];
    __shared__ int poff[64];  // position -> offset of its top-left tap in the haloed input plane (no division in the inner loops)
    __shared__ double scr[768];
    const int t = threadIdx.x, ch = t & 31, grp = t >> 5, WP = d.CW + 2, P1 = d.P1;
    if (blockIdx.x == 0) TBEG(15);
    // a board's planes travel through registers: the first board's loads go out before the statistics are combined, the next board's
    // while this one is being multiplied (<= 64 positions x 32 channels / 256 threads = 8 values per plane and thread)
    float r_c[8], r_dy[8], r_x = 0.0f;
    auto fetch = [&](int b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + TPB * k;
            if (i < P1 * 32) { r_c[k] = q.c[0][(size_t)b * P1 * 32 + i]; r_dy[k] = q.dy[0][(size_t)b * P1 * 32 + i]; }
        }
        if (t < P1) r_x = q.x0[(size_t)b * P1 + t];
    };
    if ((int)blockIdx.x < d.B) fetch(blockIdx.x);
    if (t < P1) poff[t] = (t / d.CW) * WP + t % d.CW;
    bn2d_from_done(q.fdone[0], k1, sh_o, mean_o, inv_o);
    {
        double S1 = 0.0, S2 = 0.0;
#pragma unroll 8
        for (int p = grp; p < d.NB; p += 8) { S1 += q.bpart[0][(size_t)p * 64 + ch]; S2 += q.bpart[0][(size_t)p * 64 + 32 + ch]; }
        scr[grp * 32 + ch] = S1; scr[256 + grp * 32 + ch] = S2;
        __syncthreads();
        if (t < 32) {
            S1 = 0.0; S2 = 0.0;
            for (int g = 0; g < 8; ++g) { S1 += scr[g * 32 + ch]; S2 += scr[256 + g * 32 + ch]; }
            const double Nn = (double)d.B * P1;
            k2[ch] = (float)(S1 / Nn); k3[ch] = (float)(S2 / Nn);
            if (blockIdx.x == 0) { q.bdone[0][ch] = S1; q.bdone[0][32 + ch] = S2; }
        }
    }
    for (int i = t; i < 100; i += TPB) xin[i] = 0.0f;
    __syncthreads();
    float g0 = 0.0f, g1 = 0.0f;  // element t (tap = t >> 5, oc = t & 31) and element 256 + t (tap 8) / bias (t < 32: 288 + t handled by g1b)
    float gb = 0.0f;
    for (int b = blockIdx.x; b < d.B; b += d.NB) {
        if (t < P1) xin[(t / d.CW + 1) * WP + t % d.CW + 1] = r_x;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = t + TPB * k, oc = i & 31;
            if (i < P1 * 32) {
                const float xo = (r_c[k] - mean_o[oc]) * inv_o[oc];
                dzl[(i >> 5) * LDP + oc] = k1[oc] * ((r_dy[k] - k2[oc]) - xo * k3[oc]);
            }
        }
        if (b + d.NB < d.B) fetch(b + d.NB);
        __syncthreads();
        {
            const int tap = t >> 5, oc = t & 31;  // taps 0..7
            const int toff = (tap / 3) * WP + tap % 3;
#pragma unroll 8
            for (int p = 0; p < P1; ++p) g0 = fmaf(xin[poff[p] + toff], dzl[p * LDP + oc], g0);
            if (t < 32) {
#pragma unroll 8
                for (int p = 0; p < P1; ++p) {
                    g1 = fmaf(xin[poff[p] + 2 * WP + 2], dzl[p * LDP + oc], g1);  // tap 8
                    gb += dzl[p * LDP + oc];
                }
            }
        }
        __syncthreads();
    }
    float *gp = q.gw[0] + (size_t)blockIdx.x * (9 * 32 + 32);
    gp[t] = g0;
    if (t < 32) { gp[256 + t] = g1; gp[288 + t] = gb; }
    if (blockIdx.x == 0) TEND(15);
}

// ---------------------------------------------------------------------------------------------------------------- update
// one thread per element of conv1..4 weights / biases and of the four BatchNorm2d affine pairs: reduce the per-workgroup partials in a
// fixed order, momentum SGD.  The last workgroup also moves the BatchNorm2d running statistics, writes the step's losses and advances
// the step counter, the permutation offset and the loss slot.
#define UPD_W1 288
#define UPD_W (288 + 3 * 9216)
#define UPD_B (UPD_W + 4 * 32)
#define UPD_G (UPD_B + 4 * 32)
#define UPD_ALL (UPD_G + 4 * 32)
__global__ __launch_bounds__(TPB) void k_update(TDims d, TPtr q) {
    __shared__ float s_mean[32], s_var[32];
    __shared__ double scr[1280];
    const int G = (int)gridDim.x - (d.F1B ? 6 : 0);  // the regular grid; behind it, with d.F1B, the six workgroups of conv1's folded gradient
    const bool fold_wg = (int)blockIdx.x >= G;
    if (blockIdx.x == 0) TBEG(16);
    if ((int)blockIdx.x == G - 1) TBEG(17);
    // The last workgroup advances step / perm_off / loss_off for the next step while the others are still running: nobody but that
    // workgroup reads those three fields here; everybody reads lr / momentum / wd, which no kernel writes.
    Hyper hp{};
    hp.lr = q.hp->lr; hp.momentum = q.hp->momentum; hp.wd = q.hp->wd;
    if ((int)blockIdx.x == G - 1) { hp.step = q.hp->step; hp.perm_off = q.hp->perm_off; hp.loss_off = q.hp->loss_off; }
    const int t = threadIdx.x;
    if (!fold_wg && (int)blockIdx.x >= G - 4) {  // the last four regular workgroups: one BatchNorm2d layer's running statistics each
        const int l = G - 1 - (int)blockIdx.x;
        if (t < 32) { s_mean[t] = q.fdone[l][t]; s_var[t] = q.fdone[l][128 + t]; }  // mean and biased variance as the forward pass finished them
        if (t < 32) {
            const double n = (double)d.B * plane_of(d, l);
            q.rm[l][t] = (float)((1.0 - BN_MOM) * q.rm[l][t] + BN_MOM * s_mean[t]);
            q.rv[l][t] = (float)((1.0 - BN_MOM) * q.rv[l][t] + BN_MOM * ((double)s_var[t] * n / (n > 1.0 ? n - 1.0 : 1.0)));
        }
        if (l == 0 && t == 64) {  // (another wave than the one finishing the statistics) the step's losses, then the counters of the next step
            double a0 = 0.0, a1 = 0.0;
            for (int i = 0; i < d.B / 16; ++i) { a0 += q.losspart[2 * i]; a1 += q.losspart[2 * i + 1]; }
            q.loss_pi[hp.loss_off] = (float)(a0 / d.B); q.loss_v[hp.loss_off] = (float)(a1 / d.B);
            q.hp->step = hp.step + 1; q.hp->perm_off = hp.perm_off + d.B; q.hp->loss_off = hp.loss_off + 1;
#ifdef AZ_TPROBE
            az_tprobe[17 * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        return;
    }
    // 64 elements per workgroup; the four waves each sum a quarter of the element's partials (wave w: partials [w NP/4, (w+1) NP/4), sixteen
    // loads per trip), the quarters meet in LDS and are added in order -- a fixed summation order, a quarter of the dependent trips
    const int lane = t & 63, part = t >> 6, e = blockIdx.x * 64 + lane;
    double *qs = scr;  // [4][64]
    TSTAMP(4, 0);
    double acc = 0.0;
    int kind = -1, l = 0, i = 0, c = 0;
    bool gam = false;
    // d.F1B: conv1's gradient arrives as per-board moments (conv_bwd_body, l = 1) and is combined by SIX WORKGROUPS OF ITS OWN behind the
    // regular grid (288 weights, 32 biases, BatchNorm 0's 64 affine values): every thread of them walks the same loads -- this wave's
    // quarter of the NB partials of A, C, Bx (weights) or sum xhat (bias) and of BatchNorm 0's backward sums S1, S2 (nobody has finished
    // bdone[0] in this form) -- sixteen partials per trip, every load before the first add.  The regular workgroups skip layer 0.
    double f1[4] = {0.0, 0.0, 0.0, 0.0};
    if (fold_wg) {
        const int fe = ((int)blockIdx.x - G) * 64 + lane, oc = fe & 31;
        const bool fw = fe < 288, fb = fe >= 288 && fe < 320;
        const int ia = fw ? fe : 0, ic = fw ? 288 + fe : (fb ? 576 + oc : 0), ix = fw ? 608 + (fe >> 5) : 608;
        const int per = (d.NB + 3) / 4, pbeg = part * per, pend = min(d.NB, pbeg + per);
        for (int p0 = pbeg; p0 < pend; p0 += 16) {
            double b1[16], b2[16];
            float ga[16], gc[16], gx[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + u < pend ? p0 + u : pend - 1;
                const float *g1p = q.gw[0] + (size_t)p * C1P;
                b1[u] = q.bpart[0][(size_t)p * 64 + oc]; b2[u] = q.bpart[0][(size_t)p * 64 + 32 + oc];
                ga[u] = g1p[ia]; gc[u] = g1p[ic]; gx[u] = g1p[ix];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (p0 + u < pend) { f1[0] += b1[u]; f1[1] += b2[u]; acc += ga[u]; f1[2] += gc[u]; f1[3] += gx[u]; }
        }
        kind = fw ? 3 : (fb ? 4 : 5);
        l = 0; i = fe; c = oc; gam = fe >= 320 && fe < 352;
    } else if (d.F1B && (e < UPD_W1 || (e >= UPD_W && e < UPD_W + 32) || (e >= UPD_B && e < UPD_ALL && ((e - UPD_B) & 127) < 32))) {
        kind = -1;  // layer 0: the fold workgroups' (UPD_W, UPD_B, UPD_G are multiples of 32)
    } else
    if (e < UPD_W) {
        kind = 0;
        l = e < UPD_W1 ? 0 : 1 + (e - UPD_W1) / 9216; i = e < UPD_W1 ? e : (e - UPD_W1) % 9216;
        const int sz = l == 0 ? 9 * 32 + 32 : 9 * 32 * 32 + 32;
        const int NP = l == 0 ? d.NB : d.NB / d.S;  // conv2..4: the S workgroups that share boards fill one partial together
        const int per = (NP + 3) / 4, pbeg = part * per, pend = min(NP, pbeg + per);
        const float *gp = q.gw[l] + i;
        float g = 0.0f;
        for (int p0 = pbeg; p0 < pend; p0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p0 + u < pend ? gp[(size_t)(p0 + u) * sz] : 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) g += v[u];
        }
        acc = g;
    } else if (e < UPD_B) {
        kind = 1;
        l = (e - UPD_W) >> 5; c = (e - UPD_W) & 31;
        const int sz = l == 0 ? 9 * 32 + 32 : 9 * 32 * 32 + 32;
        const int NP = l == 0 ? d.NB : d.NB / d.S;
        const int per = (NP + 3) / 4, pbeg = part * per, pend = min(NP, pbeg + per);
        float g = 0.0f;
#pragma unroll 8
        for (int p = pbeg; p < pend; ++p) g += q.gw[l][(size_t)p * sz + sz - 32 + c];
        acc = g;
    } else if (e < UPD_ALL) {
        kind = 2;
        gam = e < UPD_G;
        l = ((e - (gam ? UPD_B : UPD_G)) >> 5); c = e & 31;  // UPD_B and UPD_G are multiples of 32
        // the sums over all boards and positions were finished by the kernel that needed them first (conv_bwd l, k_conv1_bwd): one load
        const double S = part == 0 ? q.bdone[l][(gam ? 32 : 0) + c] : 0.0;
        acc = S;
    }
    qs[part * 64 + lane] = acc;
    if (kind >= 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) scr[256 * (k + 1) + part * 64 + lane] = f1[k];
    }
    __syncthreads();
    if (part != 0 || kind < 0) return;
    if (kind >= 3) {
        double T[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) T[k] = ((scr[256 * k + lane] + scr[256 * k + 64 + lane]) + scr[256 * k + 128 + lane]) + scr[256 * k + 192 + lane];
        const double S1 = T[1], S2 = T[2], Nn = (double)d.B * d.P1;
        const float k1 = q.fdone[0][64 + c], k2 = (float)(S1 / Nn), k3 = (float)(S2 / Nn);  // as k_conv1_bwd forms them
        if (kind == 3) {       // dW1[tap][oc] = k1 (A - k2 Bx - k3 C)
            const float g = (float)((double)k1 * ((T[0] - (double)k2 * T[4]) - (double)k3 * T[3]));
            sgd(q.p.cw[0] + i, q.m.cw[0] + i, g, hp);
        } else if (kind == 4) {  // d b1[oc] = sum dz0 = k1 ((S1 - N k2) - k3 sum xhat): zero in exact arithmetic, the rounding of the two means in float32
            const float g = (float)((double)k1 * ((S1 - Nn * (double)k2) - (double)k3 * T[3]));
            sgd(q.p.cb[0] + c, q.m.cb[0] + c, g, hp);
        } else {               // BatchNorm 0's affine pair: gamma <- sum dy xhat, beta <- sum dy
            if (gam) sgd(q.p.bg[0] + c, q.m.bg[0] + c, (float)S2, hp);
            else sgd(q.p.bb[0] + c, q.m.bb[0] + c, (float)S1, hp);
        }
        return;
    }
    if (kind == 2) {
        const double S = ((qs[lane] + qs[64 + lane]) + qs[128 + lane]) + qs[192 + lane];
        if (gam) sgd(q.p.bg[l] + c, q.m.bg[l] + c, (float)S, hp);
        else sgd(q.p.bb[l] + c, q.m.bb[l] + c, (float)S, hp);
    } else {
        const float g = (((float)qs[lane] + (float)qs[64 + lane]) + (float)qs[128 + lane]) + (float)qs[192 + lane];
        if (kind == 0) sgd(q.p.cw[l] + i, q.m.cw[l] + i, g, hp);
        else sgd(q.p.cb[l] + c, q.m.cb[l] + c, g, hp);
    }
    TSTAMP(4, 1);
#ifdef AZ_TPROBE
    if (blockIdx.x == 0 && t == 0) az_tprobe[16 * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------------------------- TicTacToeNet
// The 316-parameter MLP of tictactoe.py:289-316 (fc1 9->9, bn1, ReLU, fc2 9->9, bn2, ReLU, fc_probs 9->9 | fc_value 9->1; no dropout):
// one whole optimisation step -- forward, loss, backward, momentum SGD, running statistics -- in ONE launch of ONE workgroup; the
// stock PyTorch step is ~70 launches.  Rows live in LDS ([B][12] per activation), a thread owns rows t, t + 256; batch reductions
// (BatchNorm statistics, weight gradients) are taken in row order by the thread that owns the output element: deterministic.
// Parameter block (torch layouts): w1[81] b1[9] g1[9] be1[9] | w2[81] b2[9] g2[9] be2[9] | wp[81] bp[9] | wv[9] bv[1]  = 316 floats.
#define TT_W1 0
#define TT_B1 81
#define TT_G1 90
#define TT_BE1 99
#define TT_W2 108
#define TT_B2 189
#define TT_G2 198
#define TT_BE2 207
#define TT_WP 216
#define TT_BP 297
#define TT_WV 306
#define TT_BV 315
#define TT_N 316
#define TT_LD 12
struct TttPtr {
    float *p, *m;         // parameters / momentum [316]
    float *rs;            // running statistics: rm1[9] rv1[9] rm2[9] rv2[9]
    const int8_t *state; const float *pi; const int8_t *z; const long long *perm;
    float *loss_pi, *loss_v;
    Hyper *hp;
};

// column statistics of act[B][TT_LD] (columns 0..8) into s_mu / s_iv; thread (col = t & 15 < 9, rg = t >> 4) partial sums, two passes
AZ_D void ttt_colstats(const float *act, int B, float *s_mu, float *s_iv, double *s_m2, double *scr) {
    const int t = threadIdx.x, col = t & 15, rg = t >> 4;
    double s = 0.0;
    if (col < 9) for (int r = rg; r < B; r += 16) s += act[r * TT_LD + col];
    scr[rg * 16 + col] = s;
    __syncthreads();
    double tot = 0.0;
    for (int g = 0; g < 16; ++g) tot += scr[g * 16 + col];
    const double mean = tot / B;
    __syncthreads();
    double qq = 0.0;
    if (col < 9) for (int r = rg; r < B; r += 16) { const double dl = act[r * TT_LD + col] - mean; qq += dl * dl; }
    scr[rg * 16 + col] = qq;
    __syncthreads();
    if (t < 9) {
        double M2 = 0.0;
        for (int g = 0; g < 16; ++g) M2 += scr[g * 16 + t];
        s_mu[t] = (float)mean; s_iv[t] = (float)(1.0 / sqrt(M2 / B + BN_EPS)); s_m2[t] = M2;
    }
    __syncthreads();
}

// BatchNorm1d backward on dh[B][TT_LD] (gradient w.r.t. relu(bn(y))): dh <- dz in place; sums for the affine pair in s_s1 / s_s2
AZ_D void ttt_bn_bwd(float *dh, const float *y, int B, const float *s_mu, const float *s_iv, const float *gam, double *s_s1, double *s_s2, double *scr) {
    const int t = threadIdx.x, col = t & 15, rg = t >> 4;
    double s1 = 0.0, s2 = 0.0;
    if (col < 9)
        for (int r = rg; r < B; r += 16) {
            const float xh = (y[r * TT_LD + col] - s_mu[col]) * s_iv[col];
            const float d = dh[r * TT_LD + col];
            s1 += d; s2 += (double)d * (double)xh;
        }
    scr[rg * 16 + col] = s1; scr[256 + rg * 16 + col] = s2;
    __syncthreads();
    if (t < 9) {
        double S1 = 0.0, S2 = 0.0;
        for (int g = 0; g < 16; ++g) { S1 += scr[g * 16 + t]; S2 += scr[256 + g * 16 + t]; }
        s_s1[t] = S1; s_s2[t] = S2;
    }
    __syncthreads();
    if (col < 9) {
        const float k = gam[col] * s_iv[col], m1 = (float)(s_s1[col] / B), m2 = (float)(s_s2[col] / B);
        for (int r = rg; r < B; r += 16) {
            const float xh = (y[r * TT_LD + col] - s_mu[col]) * s_iv[col];
            dh[r * TT_LD + col] = k * ((dh[r * TT_LD + col] - m1) - xh * m2);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(TPB) void k_ttt_step(TttPtr q, int B) {
    extern __shared__ __align__(16) float lds[];
    float *w = lds;                         // parameters [316] (+ pad to 320)
    float *x = w + 320, *y1 = x + B * TT_LD, *a1 = y1 + B * TT_LD, *y2 = a1 + B * TT_LD, *a2 = y2 + B * TT_LD, *dl = a2 + B * TT_LD, *d2 = dl + B * TT_LD,
          *d1 = d2 + B * TT_LD, *gr = d1 + B * TT_LD;  // gr: gradients [320]
    float *s_mu1 = gr + 320, *s_iv1 = s_mu1 + 16, *s_mu2 = s_iv1 + 16, *s_iv2 = s_mu2 + 16;
    double *scr = (double *)(s_iv2 + 16), *s_m2a = scr + 512, *s_m2b = s_m2a + 16, *s_s1 = s_m2b + 16, *s_s2 = s_s1 + 16, *s_loss = s_s2 + 16;
    const int t = threadIdx.x;
    const Hyper hp = *q.hp;
    for (int i = t; i < TT_N; i += TPB) w[i] = q.p[i];
    for (int r = t; r < B; r += TPB) {
        const long long row = sample_row(q.perm, hp.perm_off + r, hp, q.hp);
#pragma unroll
        for (int k = 0; k < 9; ++k) x[r * TT_LD + k] = (float)q.state[row * 9 + k];
    }
    __syncthreads();
    for (int r = t; r < B; r += TPB)  // fc1
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float acc = w[TT_B1 + j];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(x[r * TT_LD + k], w[TT_W1 + j * 9 + k], acc);
            y1[r * TT_LD + j] = acc;
        }
    __syncthreads();
    ttt_colstats(y1, B, s_mu1, s_iv1, s_m2a, scr);
    for (int r = t; r < B; r += TPB) {  // bn1 + relu, fc2
        float a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { a[k] = fmaxf(fmaf(w[TT_G1 + k], (y1[r * TT_LD + k] - s_mu1[k]) * s_iv1[k], w[TT_BE1 + k]), 0.f); a1[r * TT_LD + k] = a[k]; }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float acc = w[TT_B2 + j];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(a[k], w[TT_W2 + j * 9 + k], acc);
            y2[r * TT_LD + j] = acc;
        }
    }
    __syncthreads();
    ttt_colstats(y2, B, s_mu2, s_iv2, s_m2b, scr);
    double lpi = 0.0, lv = 0.0;
    const float invB = 1.0f / (float)B;
    for (int r = t; r < B; r += TPB) {  // bn2 + relu, heads, loss, d loss / d logits
        const long long row = sample_row(q.perm, hp.perm_off + r, hp, q.hp);
        float a[9], lg[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { a[k] = fmaxf(fmaf(w[TT_G2 + k], (y2[r * TT_LD + k] - s_mu2[k]) * s_iv2[k], w[TT_BE2 + k]), 0.f); a2[r * TT_LD + k] = a[k]; }
        float mx = -3.0e38f, u = w[TT_BV];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float acc = w[TT_BP + j];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(a[k], w[TT_WP + j * 9 + k], acc);
            lg[j] = acc; mx = fmaxf(mx, acc);
            u = fmaf(a[j], w[TT_WV + j], u);
        }
        float se = 0.f, spi = 0.f, pit[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) { se += expf(lg[j] - mx); pit[j] = q.pi[row * 9 + j]; spi += pit[j]; }
        const float lse = mx + logf(se);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float lp = lg[j] - lse;
            lpi -= (double)(pit[j] * lp);
            dl[r * TT_LD + j] = (expf(lp) * spi - pit[j]) * invB;
        }
        const float vv = tanhf(u), zz = (float)q.z[row];
        lv += (double)(vv - zz) * (double)(vv - zz);
        dl[r * TT_LD + 9] = 2.0f * (vv - zz) * (1.0f - vv * vv) * invB;
    }
    scr[t] = lpi; scr[256 + t] = lv;
    __syncthreads();
    if (t == 0) {
        double a0 = 0.0, a1_ = 0.0;
        for (int i = 0; i < TPB; ++i) { a0 += scr[i]; a1_ += scr[256 + i]; }
        s_loss[0] = a0 / B; s_loss[1] = a1_ / B;
    }
    __syncthreads();
    for (int r = t; r < B; r += TPB) {  // d a2 = dl [Wp; wv], masked by the ReLU
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float acc = dl[r * TT_LD + 9] * w[TT_WV + k];
#pragma unroll
            for (int j = 0; j < 9; ++j) acc = fmaf(dl[r * TT_LD + j], w[TT_WP + j * 9 + k], acc);
            d2[r * TT_LD + k] = a2[r * TT_LD + k] > 0.0f ? acc : 0.0f;
        }
    }
    // heads gradients: thread e < 90 owns element (j, k) of [Wp; wv] (j = 9: the value row), threads 90..99 the biases
    if (t < 100) {
        double g = 0.0;
        if (t < 90) { const int j = t / 9, k = t % 9; for (int r = 0; r < B; ++r) g += (double)(dl[r * TT_LD + j] * a2[r * TT_LD + k]); }
        else { const int j = t - 90; for (int r = 0; r < B; ++r) g += dl[r * TT_LD + j]; }
        gr[t < 81 ? TT_WP + t : (t < 90 ? TT_WV + (t - 81) : (t < 99 ? TT_BP + (t - 90) : TT_BV))] = (float)g;
    }
    __syncthreads();
    ttt_bn_bwd(d2, y2, B, s_mu2, s_iv2, w + TT_G2, s_s1, s_s2, scr);  // d2 <- dz2
    if (t < 9) { gr[TT_G2 + t] = (float)s_s2[t]; gr[TT_BE2 + t] = (float)s_s1[t]; }
    for (int r = t; r < B; r += TPB) {  // d a1 = dz2 W2, masked
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) acc = fmaf(d2[r * TT_LD + j], w[TT_W2 + j * 9 + k], acc);
            d1[r * TT_LD + k] = a1[r * TT_LD + k] > 0.0f ? acc : 0.0f;
        }
    }
    if (t < 90) {  // fc2 gradients
        double g = 0.0;
        if (t < 81) { const int j = t / 9, k = t % 9; for (int r = 0; r < B; ++r) g += (double)(d2[r * TT_LD + j] * a1[r * TT_LD + k]); }
        else { const int j = t - 81; for (int r = 0; r < B; ++r) g += d2[r * TT_LD + j]; }
        gr[t < 81 ? TT_W2 + t : TT_B2 + (t - 81)] = (float)g;
    }
    __syncthreads();
    ttt_bn_bwd(d1, y1, B, s_mu1, s_iv1, w + TT_G1, s_s1, s_s2, scr);  // d1 <- dz1
    if (t < 9) { gr[TT_G1 + t] = (float)s_s2[t]; gr[TT_BE1 + t] = (float)s_s1[t]; }
    if (t < 90) {  // fc1 gradients
        double g = 0.0;
        if (t < 81) { const int j = t / 9, k = t % 9; for (int r = 0; r < B; ++r) g += (double)(d1[r * TT_LD + j] * x[r * TT_LD + k]); }
        else { const int j = t - 81; for (int r = 0; r < B; ++r) g += d1[r * TT_LD + j]; }
        gr[t < 81 ? TT_W1 + t : TT_B1 + (t - 81)] = (float)g;
    }
    __syncthreads();
    for (int i = t; i < TT_N; i += TPB) {  // momentum SGD on every parameter
        const float gg = fmaf(hp.wd, w[i], gr[i]);
        const float mm = fmaf(hp.momentum, q.m[i], gg);
        q.m[i] = mm;
        q.p[i] = fmaf(-hp.lr, mm, w[i]);
    }
    if (t < 9) {
        q.rs[t] = (float)((1.0 - BN_MOM) * q.rs[t] + BN_MOM * s_mu1[t]);
        q.rs[9 + t] = (float)((1.0 - BN_MOM) * q.rs[9 + t] + BN_MOM * (s_m2a[t] / (B > 1 ? B - 1 : 1)));
        q.rs[18 + t] = (float)((1.0 - BN_MOM) * q.rs[18 + t] + BN_MOM * s_mu2[t]);
        q.rs[27 + t] = (float)((1.0 - BN_MOM) * q.rs[27 + t] + BN_MOM * (s_m2b[t] / (B > 1 ? B - 1 : 1)));
    }
    if (t == 0) {
        q.loss_pi[hp.loss_off] = (float)s_loss[0]; q.loss_v[hp.loss_off] = (float)s_loss[1];
        q.hp->step = hp.step + 1; q.hp->perm_off = hp.perm_off + B; q.hp->loss_off = hp.loss_off + 1;
    }
}
static inline int ttt_lds_bytes(int B) { return (320 + 8 * B * TT_LD + 320 + 64) * 4 + (512 + 16 * 5) * 8; }

// ================================================================================================================ host side
// torch layout <-> the step's layout.  kind 0: copy; 1: conv weight [oc][ic][3][3] <-> [tap][ic][oc] (a = input channels);
// 2: fc1.weight [j][c * P4 + pos] <-> [j][pos * 32 + c] (a = P4).  dir 0: load (dst = ours), 1: store (dst = torch's)
__global__ void k_relayout(float *dst, const float *src, long long n, int kind, int a, int dir) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // index in torch layout
    if (i >= n) return;
    long long j = i;
    if (kind == 1) {
        const int tap = (int)(i % 9), ic = (int)((i / 9) % a), oc = (int)(i / (9LL * a));
        j = ((long long)tap * a + ic) * 32 + oc;
    } else if (kind == 2) {
        const int fin = 32 * a, k = (int)(i % fin), c = k / a, pos = k % a;
        j = (i / fin) * fin + pos * 32 + c;
    }
    if (dir == 0) dst[j] = src[i]; else dst[i] = src[j];
}

struct TensorRef { float *ptr; long long numel; int kind, a; };

struct az_trainer {
    int game = 0, H = 0, W = 0, max_batch = 0;
    TDims d;
    TPtr q;
    TttPtr tq;  // game == AZ_TICTACTOE: the whole step is one launch (k_ttt_step)
    std::vector<void *> allocs;
    std::vector<std::pair<float *, size_t>> momenta;  // zeroed by az_trainer_begin
    std::map<std::string, TensorRef> tensors;
    std::map<std::string, std::pair<void *, long long>> debug;  // name -> (device pointer, element count) of the workspace buffers
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraphExec_t graph = nullptr, graph_k = nullptr;  // one step / graph_steps consecutive steps
    int graph_steps = 8;
    bool graphs_ok = true, attrs_set = false;
    long long steps_done = 0;
    // what the captured graph was recorded with
    const void *g_state = nullptr, *g_pi = nullptr, *g_z = nullptr, *g_perm = nullptr, *g_lp = nullptr, *g_lv = nullptr;
    int g_B = 0;
};

#define AZ_TRY(x) do { int _rc = (x); if (_rc != AZ_OK) return _rc; } while (0)

template <typename T>
static int talloc(az_trainer *t, T **p, size_t n) {
    void *v = nullptr;
    AZ_HIP(hipMalloc(&v, (n ? n : 1) * sizeof(T)));
    AZ_HIP(hipMemset(v, 0, (n ? n : 1) * sizeof(T)));
    t->allocs.push_back(v);
    *p = (T *)v;
    return AZ_OK;
}

static int palloc(az_trainer *t, float *PSet::*field, size_t n) {  // a parameter and its momentum buffer
    AZ_TRY(talloc(t, &(t->q.p.*field), n));
    AZ_TRY(talloc(t, &(t->q.m.*field), n));
    t->momenta.push_back({t->q.m.*field, n});
    return AZ_OK;
}

extern "C" int az_trainer_create(int game, int H, int W, int max_batch, az_trainer **out) {
    AZ_REQUIRE(out, AZ_EINVAL, "null argument");
    AZ_REQUIRE(game == AZ_OTHELLO || game == AZ_CONNECT4 || game == AZ_TICTACTOE, AZ_EINVAL, "unknown game %d", game);
    if (game == AZ_TICTACTOE) {
        AZ_REQUIRE(H == 3 && W == 3, AZ_EINVAL, "TicTacToe board is 3x3");
        AZ_REQUIRE(max_batch >= 2 && max_batch <= 256, AZ_EINVAL, "TicTacToeNet training step: batch size in [2, 256], got %d", max_batch);
        az_trainer *t = new az_trainer();
        t->game = game; t->H = H; t->W = W; t->max_batch = max_batch;
        memset(&t->d, 0, sizeof t->d); memset(&t->q, 0, sizeof t->q); memset(&t->tq, 0, sizeof t->tq);
        t->d.B = max_batch;
        int rc = talloc(t, &t->tq.p, 320);
        if (rc == AZ_OK) rc = talloc(t, &t->tq.m, 320);
        if (rc == AZ_OK) rc = talloc(t, &t->tq.rs, 36);
        if (rc == AZ_OK) rc = talloc(t, &t->tq.hp, 1);
        if (rc == AZ_OK && (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&t->ev_out, hipEventDisableTiming) != hipSuccess)) { az_set_error("could not create the trainer's stream"); rc = AZ_EHIP; }
        if (rc != AZ_OK) { az_trainer_destroy(t); return rc; }
        t->q.hp = t->tq.hp;
        t->momenta.push_back({t->tq.m, 320});
        { const char *g = getenv("AZ_TRAIN_GRAPH"); if (g && atoi(g) == 0) t->graphs_ok = false; }
        { const char *g = getenv("AZ_TRAIN_GRAPH_STEPS"); if (g && atoi(g) >= 1 && atoi(g) <= 64) t->graph_steps = atoi(g); }
        auto reg = [&](const char *name, float *p, long long n) { t->tensors[name] = TensorRef{p, n, 0, 0}; };
        float *P = t->tq.p, *R = t->tq.rs;  // state-dict names of tictactoe.py:296-306
        reg("fc1.weight", P + TT_W1, 81); reg("fc1.bias", P + TT_B1, 9); reg("bn1.weight", P + TT_G1, 9); reg("bn1.bias", P + TT_BE1, 9);
        reg("fc2.weight", P + TT_W2, 81); reg("fc2.bias", P + TT_B2, 9); reg("bn2.weight", P + TT_G2, 9); reg("bn2.bias", P + TT_BE2, 9);
        reg("fc_probs.weight", P + TT_WP, 81); reg("fc_probs.bias", P + TT_BP, 9); reg("fc_value.weight", P + TT_WV, 9); reg("fc_value.bias", P + TT_BV, 1);
        reg("bn1.running_mean", R, 9); reg("bn1.running_var", R + 9, 9); reg("bn2.running_mean", R + 18, 9); reg("bn2.running_var", R + 27, 9);
        *out = t;
        return AZ_OK;
    }
    AZ_REQUIRE(max_batch >= 16 && max_batch <= MAXB && max_batch % 16 == 0, AZ_EINVAL, "batch size must be a multiple of 16 in [16, %d], got %d", MAXB, max_batch);
    TDims d;
    memset(&d, 0, sizeof d);
    if (game == AZ_OTHELLO) {
        AZ_REQUIRE(H == W && H >= 6 && H <= 8 && H % 2 == 0, AZ_EINVAL, "OthelloNet training step: board 6x6 or 8x8, got %dx%d", H, W);
        d.CH = H; d.CW = W; d.F1 = 1024; d.F2 = 512; d.A = H * W + 1;
    } else {
        AZ_REQUIRE(H >= 5 && H <= 8 && W >= 5 && W <= 8, AZ_EINVAL, "Connect4Net training step: board between 5x5 and 8x8, got %dx%d", W, H);
        d.CH = W; d.CW = H; d.F1 = 64; d.F2 = 32; d.A = W;  // input.view(-1, 1, width, height) (connect4.py:399)
    }
    d.P1 = d.CH * d.CW; d.H3 = d.CH - 2; d.W3 = d.CW - 2; d.P3 = d.H3 * d.W3; d.H4 = d.CH - 4; d.W4 = d.CW - 4; d.P4 = d.H4 * d.W4;
    d.FIN = 32 * d.P4; d.NH = d.A + 1; d.NHP = (d.NH + 15) / 16 * 16;
    AZ_REQUIRE(d.NHP == 16 || d.NHP == 48 || d.NHP == 80, AZ_EINVAL, "no heads kernel for %d outputs", d.NH);
    d.B = max_batch; d.S = 1; d.NB = max_batch < 256 ? max_batch : 256; d.NRB = 1; d.RB = max_batch;
    az_trainer *t = new az_trainer();
    t->game = game; t->H = H; t->W = W; t->max_batch = max_batch; t->d = d;
    memset(&t->q, 0, sizeof t->q);
    int rc = AZ_OK;
    const size_t B = (size_t)max_batch, NBmax = 256;
#define TA(p, n) if (rc == AZ_OK) rc = talloc(t, &t->q.p, (n))
#define PA(f, n) if (rc == AZ_OK) rc = palloc(t, &PSet::f, (n))
    for (int l = 0; l < 4 && rc == AZ_OK; ++l) {
        const size_t wn = l == 0 ? 9 * 32 : 9 * 32 * 32;
        rc = talloc(t, &t->q.p.cw[l], wn); if (rc == AZ_OK) rc = talloc(t, &t->q.m.cw[l], wn); t->momenta.push_back({t->q.m.cw[l], wn});
        float **pp[3] = {&t->q.p.cb[l], &t->q.p.bg[l], &t->q.p.bb[l]}, **mm[3] = {&t->q.m.cb[l], &t->q.m.bg[l], &t->q.m.bb[l]};
        for (int k = 0; k < 3 && rc == AZ_OK; ++k) { rc = talloc(t, pp[k], 32); if (rc == AZ_OK) rc = talloc(t, mm[k], 32); t->momenta.push_back({*mm[k], 32}); }
        TA(rm[l], 32); TA(rv[l], 32);
        const size_t P = l <= 1 ? d.P1 : (l == 2 ? d.P3 : d.P4);
        TA(c[l], B * P * 32); TA(dy[l], B * P * 32); TA(fpart[l], NBmax * FPART); TA(fdone[l], 160); TA(bdone[l], 64);
        TA(gw[l], NBmax * (l == 0 ? (size_t)C1P : wn + 32));
        if (l < 3) TA(bpart[l], NBmax * 64);
    }
    PA(w1, (size_t)d.F1 * d.FIN); PA(b1, d.F1); PA(g1, d.F1); PA(be1, d.F1);
    PA(w2, (size_t)d.F2 * d.F1); PA(b2, d.F2); PA(g2, d.F2); PA(be2, d.F2);
    PA(wh, (size_t)d.NHP * d.F2); PA(bh, d.NHP);
    TA(rm1, d.F1); TA(rv1, d.F1); TA(rm2, d.F2); TA(rv2, d.F2);
    TA(x0, B * d.P1); TA(y1, B * d.F1); TA(h1, B * d.F1); TA(mu1, d.F1); TA(iv1, d.F1);
    TA(y2, B * d.F2); TA(h2, B * d.F2); TA(mu2, d.F2); TA(iv2, d.F2);
    TA(dlog, B * d.NHP); TA(losspart, (B / 16) * 2); TA(dz2, B * d.F2); TA(dz1, B * d.F1);
    TA(colsum, (size_t)NRBMAX * d.FIN * 2); TA(hp, 1);
    TA(hwpart, (size_t)NRBMAX * d.NHP * (d.F2 + 1));
    TA(fstat[0], (size_t)NRBMAX * d.F1 * 3); TA(fstat[1], (size_t)NRBMAX * d.F2 * 3); TA(bstat[0], (size_t)NRBMAX * d.F1 * 3); TA(bstat[1], (size_t)NRBMAX * d.F2 * 3);
#undef TA
#undef PA
    if (rc == AZ_OK && (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&t->ev_out, hipEventDisableTiming) != hipSuccess)) {
        az_set_error("could not create the trainer's stream");
        rc = AZ_EHIP;
    }
    if (rc != AZ_OK) { az_trainer_destroy(t); return rc; }
    { const char *g = getenv("AZ_TRAIN_GRAPH"); if (g && atoi(g) == 0) t->graphs_ok = false; }
    { const char *g = getenv("AZ_TRAIN_GRAPH_STEPS"); if (g && atoi(g) >= 1 && atoi(g) <= 64) t->graph_steps = atoi(g); }
    // names as in the reference's state dict (othello.py:341-368, connect4.py:370-389)
    auto reg = [&](const std::string &name, float *p, long long n, int kind = 0, int a = 0) { t->tensors[name] = TensorRef{p, n, kind, a}; };
    for (int l = 0; l < 4; ++l) {
        const std::string c = "conv" + std::to_string(l + 1), b = "bn" + std::to_string(l + 1);
        reg(c + ".weight", t->q.p.cw[l], l == 0 ? 9 * 32 : 9 * 32 * 32, 1, l == 0 ? 1 : 32);
        reg(c + ".bias", t->q.p.cb[l], 32);
        reg(b + ".weight", t->q.p.bg[l], 32); reg(b + ".bias", t->q.p.bb[l], 32);
        reg(b + ".running_mean", t->q.rm[l], 32); reg(b + ".running_var", t->q.rv[l], 32);
    }
    reg("fc1.weight", t->q.p.w1, (long long)d.F1 * d.FIN, 2, d.P4); reg("fc1.bias", t->q.p.b1, d.F1);
    reg("fc_bn1.weight", t->q.p.g1, d.F1); reg("fc_bn1.bias", t->q.p.be1, d.F1); reg("fc_bn1.running_mean", t->q.rm1, d.F1); reg("fc_bn1.running_var", t->q.rv1, d.F1);
    reg("fc2.weight", t->q.p.w2, (long long)d.F2 * d.F1); reg("fc2.bias", t->q.p.b2, d.F2);
    reg("fc_bn2.weight", t->q.p.g2, d.F2); reg("fc_bn2.bias", t->q.p.be2, d.F2); reg("fc_bn2.running_mean", t->q.rm2, d.F2); reg("fc_bn2.running_var", t->q.rv2, d.F2);
    reg("fc_probs.weight", t->q.p.wh, (long long)d.A * d.F2); reg("fc_probs.bias", t->q.p.bh, d.A);
    reg("fc_value.weight", t->q.p.wh + (size_t)d.A * d.F2, d.F2); reg("fc_value.bias", t->q.p.bh + d.A, 1);
    auto dbg = [&](const char *name, void *p, long long n) { t->debug[name] = {p, n}; };
    for (int l = 0; l < 4; ++l) {
        const long long P = l <= 1 ? d.P1 : (l == 2 ? d.P3 : d.P4);
        dbg(("c" + std::to_string(l + 1)).c_str(), t->q.c[l], (long long)B * P * 32);
        dbg(("dy" + std::to_string(l + 1)).c_str(), t->q.dy[l], (long long)B * P * 32);
        dbg(("fpart" + std::to_string(l + 1)).c_str(), t->q.fpart[l], (long long)NBmax * FPART);
    }
    dbg("x0", t->q.x0, B * d.P1); dbg("y1", t->q.y1, B * d.F1); dbg("h1", t->q.h1, B * d.F1); dbg("y2", t->q.y2, B * d.F2); dbg("h2", t->q.h2, B * d.F2);
    dbg("dlog", t->q.dlog, B * d.NHP); dbg("dz2", t->q.dz2, B * d.F2); dbg("dz1", t->q.dz1, B * d.F1); dbg("losspart", t->q.losspart, (B / 16) * 2);
    *out = t;
    return AZ_OK;
}

extern "C" void az_trainer_destroy(az_trainer *t) {
    if (!t) return;
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->graph) (void)hipGraphExecDestroy(t->graph);
    if (t->graph_k) (void)hipGraphExecDestroy(t->graph_k);
    for (void *p : t->allocs) (void)hipFree(p);
    if (t->ev_in) (void)hipEventDestroy(t->ev_in);
    if (t->ev_out) (void)hipEventDestroy(t->ev_out);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

// the trainer works on a stream of its own (graph capture is not allowed on the legacy default stream): ordered behind the caller's
// stream on entry, and the caller's stream is ordered behind it on exit -- no host synchronisation
static int t_enter(az_trainer *t, hipStream_t user) {
    AZ_HIP(hipEventRecord(t->ev_in, user));
    AZ_HIP(hipStreamWaitEvent(t->stream, t->ev_in, 0));
    return AZ_OK;
}
static int t_leave(az_trainer *t, hipStream_t user) {
    AZ_HIP(hipEventRecord(t->ev_out, t->stream));
    AZ_HIP(hipStreamWaitEvent(user, t->ev_out, 0));
    return AZ_OK;
}

static int relayout(az_trainer *t, const char *name, float *theirs, long long numel, int dir, hipStream_t user) {
    AZ_REQUIRE(t && name && theirs, AZ_EINVAL, "null argument");
    auto it = t->tensors.find(name);
    AZ_REQUIRE(it != t->tensors.end(), AZ_EINVAL, "unknown tensor '%s'", name);
    const TensorRef &r = it->second;
    AZ_REQUIRE(numel == r.numel, AZ_EINVAL, "tensor '%s' has %lld elements, expected %lld", name, numel, r.numel);
    AZ_TRY(t_enter(t, user));
    const unsigned blocks = (unsigned)((numel + 255) / 256);
    if (dir == 0) hipLaunchKernelGGL(k_relayout, dim3(blocks), dim3(256), 0, t->stream, r.ptr, (const float *)theirs, numel, r.kind, r.a, 0);
    else hipLaunchKernelGGL(k_relayout, dim3(blocks), dim3(256), 0, t->stream, theirs, (const float *)r.ptr, numel, r.kind, r.a, 1);
    AZ_HIP(hipGetLastError());
    return t_leave(t, user);
}

extern "C" int az_trainer_load(az_trainer *t, const char *name, const float *d_src, int64_t numel, void *stream) {
    return relayout(t, name, const_cast<float *>(d_src), numel, 0, (hipStream_t)stream);
}

extern "C" int az_trainer_store(az_trainer *t, const char *name, float *d_dst, int64_t numel, void *stream) {
    return relayout(t, name, d_dst, numel, 1, (hipStream_t)stream);
}

// The stream is idle: read and clear the sticky flag of sample_row.  `rejected` names the call that finds the flag of an EARLIER
// az_trainer_steps call and therefore does nothing itself (nullptr: az_trainer_check, which reports on the steps it waited for).
static int check_rows(az_trainer *t, const char *rejected = nullptr) {
    int err = 0;
    AZ_HIP(hipMemcpy(&err, &t->q.hp->err, sizeof err, hipMemcpyDeviceToHost));
    if (err) {
        const int zero = 0;
        AZ_HIP(hipMemcpy(&t->q.hp->err, &zero, sizeof zero, hipMemcpyHostToDevice));
        if (rejected)
            AZ_REQUIRE(false, AZ_EINVAL, "%s rejected, nothing of it ran: a permutation entry of an EARLIER az_trainer_steps call was outside [0, n_samples) "
                       "(those batch slots trained on row 0) and nobody had called az_trainer_check; the flag is now cleared", rejected);
        AZ_REQUIRE(false, AZ_EINVAL, "a permutation entry of the last az_trainer_steps call was outside [0, n_samples): those batch slots trained on row 0");
    }
    return AZ_OK;
}

extern "C" int az_trainer_begin(az_trainer *t, float lr, float momentum, float weight_decay, float dropout_p, uint32_t seed, void *stream) {
    AZ_REQUIRE(t, AZ_EINVAL, "null argument");
    AZ_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, AZ_EINVAL, "dropout probability %g outside [0, 1)", (double)dropout_p);
    hipStream_t user = (hipStream_t)stream;
    AZ_HIP(hipStreamSynchronize(t->stream));
    AZ_TRY(check_rows(t, "az_trainer_begin"));  // h{} below would wipe an unreported flag: report it first, before anything is touched
    AZ_TRY(t_enter(t, user));
    for (auto &m : t->momenta) AZ_HIP(hipMemsetAsync(m.first, 0, m.second * sizeof(float), t->stream));
    Hyper h{};
    h.lr = lr; h.momentum = momentum; h.wd = weight_decay; h.drop_p = dropout_p; h.seed = seed; h.step = 0; h.perm_off = 0; h.loss_off = 0;
    AZ_HIP(hipMemcpyAsync(t->q.hp, &h, sizeof h, hipMemcpyHostToDevice, t->stream));
    AZ_HIP(hipStreamSynchronize(t->stream));  // h lives on this frame
    t->steps_done = 0;
    return t_leave(t, user);
}

extern "C" int az_trainer_set_lr(az_trainer *t, float lr, void *stream) {
    AZ_REQUIRE(t, AZ_EINVAL, "null argument");
    hipStream_t user = (hipStream_t)stream;
    AZ_TRY(t_enter(t, user));
    AZ_HIP(hipMemcpyAsync(&t->q.hp->lr, &lr, sizeof lr, hipMemcpyHostToDevice, t->stream));
    AZ_HIP(hipStreamSynchronize(t->stream));
    return t_leave(t, user);
}

static bool wgrad_beside_conv1() {  // AZ_TRAIN_WG_LATE=0: fc1's weight gradient stays in k_mix2 on the row-split path too
    static int v = -1;
    if (v < 0) { const char *e = getenv("AZ_TRAIN_WG_LATE"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v == 1;
}

static bool fc2_rows64() {  // AZ_TRAIN_FC2_RB64=0: fc2's forward keeps the step's row-block size
    static int v = -1;
    if (v < 0) { const char *e = getenv("AZ_TRAIN_FC2_RB64"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v == 1;
}

template <int RTM, int NW, int PF, int NT, bool SPLIT>
static int enqueue_step_t(az_trainer *t) {
    const TDims &d = t->d;
    const TPtr &q = t->q;
    hipStream_t st = t->stream;
    const int Bw = SPLIT ? RTM * 16 : d.B;  // rows a dense workgroup holds
    const int fcl = fc_lds_bytes(NW, Bw), hbl = Bw * 16 * 4 + 768 * 8;
    const unsigned NRB = SPLIT ? (unsigned)d.NRB : 1u;
#define SETATTR(k, bytes) AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)))
    if (!t->attrs_set) {
        SETATTR(k_conv_fwd, CONV_FWD_LDS_BYTES); SETATTR(k_conv_bwd, CONV_BWD_LDS_BYTES); SETATTR(k_mix2, CONV_BWD_LDS_BYTES);
        t->attrs_set = true;
    }
    {   // per instantiation: the largest batch this configuration serves (set every time: cheap, and correct across batch sizes)
        const int fmax = fc_lds_bytes(NW, RTM * 16);
        SETATTR((k_fc_fwd<RTM, NW, PF, SPLIT>), fmax); SETATTR((k_fc_dgrad<RTM, NW, PF, SPLIT>), fmax); SETATTR((k_mix1<RTM, NW, PF, SPLIT>), fmax);
        SETATTR((k_heads_bwd<RTM, NT, SPLIT>), RTM * 16 * 16 * 4 + 768 * 8);
    }
#undef SETATTR
#define SETATTR2(k, bytes) AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)))
    const dim3 tb(TPB), tw(NW * 64);
    hipLaunchKernelGGL(k_conv1_fwd, dim3(d.NB), tb, 0, st, d, q);
    for (int l = 1; l <= 3; ++l) hipLaunchKernelGGL(k_conv_fwd, dim3(d.NB), tb, CONV_FWD_LDS_BYTES, st, d, q, l);
    hipLaunchKernelGGL((k_fc_fwd<RTM, NW, PF, SPLIT>), dim3(d.F1 / 16, NRB), tw, fcl, st, d, q, 1);
    if (SPLIT) hipLaunchKernelGGL(k_fc_fin, dim3(d.F1 / 16, NRB), tb, 0, st, d, q, 1);
    if (SPLIT && RTM == 8 && fc2_rows64()) {
        // fc2 has half of fc1's columns: at 128 rows per block its forward would run 32 x NRB = 128 workgroups on 256 CUs.  It takes the
        // 64-row configuration (sixteen waves split K = 1024) instead; its statistics partials and its finalize kernel follow that block size
        TDims d2 = d;
        d2.RB = 64; d2.NRB = (d.B + 63) / 64;
        SETATTR2((k_fc_fwd<4, 16, 1, true>), fc_lds_bytes(16, 64));
        hipLaunchKernelGGL((k_fc_fwd<4, 16, 1, true>), dim3(d.F2 / 16, (unsigned)d2.NRB), dim3(16 * 64), fc_lds_bytes(16, 64), st, d2, q, 2);
        hipLaunchKernelGGL(k_fc_fin, dim3(d.F2 / 16, (unsigned)d2.NRB), tb, 0, st, d2, q, 2);
    } else {
        hipLaunchKernelGGL((k_fc_fwd<RTM, NW, PF, SPLIT>), dim3(d.F2 / 16, NRB), tw, fcl, st, d, q, 2);
        if (SPLIT) hipLaunchKernelGGL(k_fc_fin, dim3(d.F2 / 16, NRB), tb, 0, st, d, q, 2);
    }
    hipLaunchKernelGGL((k_heads_fwd<NT, 8>), dim3(d.B / 16), dim3(8 * 64), 0, st, d, q);
    hipLaunchKernelGGL((k_heads_bwd<RTM, NT, SPLIT>), dim3(d.F2 / 16, NRB), tb, hbl, st, d, q);
    if (SPLIT) hipLaunchKernelGGL(k_bn1d_bwd_fin, dim3(d.F2 / 16, NRB), tb, 0, st, d, q, 2);
    hipLaunchKernelGGL((k_fc_dgrad<RTM, NW, PF, SPLIT>), dim3(d.F1 / 16, NRB), tw, fcl, st, d, q);
    if (SPLIT) hipLaunchKernelGGL(k_bn1d_bwd_fin, dim3(d.F1 / 16, NRB), tb, 0, st, d, q, 1);
    const int nw2 = fc_wgrad_blocks(d.F2, d.F1, SPLIT ? NW / 4 : NW), nw1 = fc_wgrad_blocks(d.F1, d.FIN, 4);
    hipLaunchKernelGGL((k_mix1<RTM, NW, PF, SPLIT>), dim3(nw2 + (d.FIN / 16) * NRB), tw, fcl, st, d, q, nw2);
    const bool wg_late = SPLIT && wgrad_beside_conv1();  // fc1's weight gradient beside conv1's backward instead of conv4's
    if (wg_late) hipLaunchKernelGGL(k_conv_bwd, dim3(d.NB), tb, CONV_BWD_LDS_BYTES, st, d, q, 3);
    else hipLaunchKernelGGL(k_mix2, dim3(nw1 + d.NB), tb, CONV_BWD_LDS_BYTES, st, d, q, nw1);
    hipLaunchKernelGGL(k_conv_bwd, dim3(d.NB), tb, CONV_BWD_LDS_BYTES, st, d, q, 2);
    hipLaunchKernelGGL(k_conv_bwd, dim3(d.NB), tb, CONV_BWD_LDS_BYTES, st, d, q, 1);
    if (wg_late) hipLaunchKernelGGL(k_conv1_bwd, dim3(d.NB + (d.F1 / 32) * (d.FIN / 32)), tb, CONV1_WG_LDS_BYTES, st, d, q);
    else if (!d.F1B) hipLaunchKernelGGL(k_conv1_bwd, dim3(d.NB), tb, 0, st, d, q);
    hipLaunchKernelGGL(k_update, dim3((UPD_ALL + 63) / 64 + 4 + (d.F1B ? 6 : 0)), tb, 0, st, d, q);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

template <int RTM, int NW, int PF, bool SPLIT = false>
static int enqueue_step_r(az_trainer *t) {
    switch (t->d.NHP) {
        case 16: return enqueue_step_t<RTM, NW, PF, 1, SPLIT>(t);
        case 48: return enqueue_step_t<RTM, NW, PF, 3, SPLIT>(t);
        default: return enqueue_step_t<RTM, NW, PF, 5, SPLIT>(t);
    }
}

// rows per row block of the dense kernels for batch size B: 0 = no split (a workgroup sees all rows of its 16 columns: <= 128 rows,
// where sixteen or eight waves splitting K keep a 16-column workgroup short); above, blocks of 64 rows (the batch-64 configuration:
// sixteen waves split K) up to 368 rows and of 128 rows (eight waves) from 384 on, so that fc1 / fc2 run 4 x 64 / 4 x 32 workgroups at
// batch 512 instead of 64 / 32 (measured, ms per step at 256 / 512: unsplit 0.412 / 0.658, blocks of 64 0.315 / 0.442, of 128
// 0.337 / 0.408).  AZ_TRAIN_RB = 0 / 64 / 128 overrides.
static int dense_row_block(int B) {
    static int force = -2;
    if (force == -2) { const char *e = getenv("AZ_TRAIN_RB"); force = e ? atoi(e) : -1; }
    if (force == 0 || ((force == 64 || force == 128) && B > force)) return force;
    return B > 128 ? (B >= 384 ? 128 : 64) : 0;
}

static int enqueue_step(az_trainer *t) {  // (row tiles, waves that split K, prefetch depth) by batch size: see "dense layers: shared pieces"
    if (t->game == AZ_TICTACTOE) {
        const int lds = ttt_lds_bytes(t->d.B);
        if (!t->attrs_set) {
            AZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ttt_step), hipFuncAttributeMaxDynamicSharedMemorySize, ttt_lds_bytes(256)));
            t->attrs_set = true;
        }
        hipLaunchKernelGGL(k_ttt_step, dim3(1), dim3(TPB), lds, t->stream, t->tq, t->d.B);
        AZ_HIP(hipGetLastError());
        return AZ_OK;
    }
    const int RT = t->d.B / 16;
    if (t->d.NRB > 1) return t->d.RB == 64 ? enqueue_step_r<4, 16, 1, true>(t) : enqueue_step_r<8, 8, 1, true>(t);
    if (RT <= 4) return enqueue_step_r<4, 16, 1>(t);  // (both register buffers two steps deep: 0.219 vs 0.195 ms at batch 64)
    if (RT <= 8) return enqueue_step_r<8, 8, 1>(t);
    if (RT <= 16) return enqueue_step_r<16, 4, 1>(t);
    return enqueue_step_r<32, 4, 0>(t);
}

// n_steps optimisation steps on device-resident samples: step s trains on rows d_perm[s * B .. s * B + B) of (d_state int8 [S][cells],
// d_pi f32 [S][A], d_z int8 [S]) and writes its policy / value loss to d_loss_pi[s] / d_loss_v[s].  Asynchronous on `stream`.
// waits for the steps enqueued so far and reports a permutation entry that was out of range (the kernels clamp it and raise a flag)
extern "C" int az_trainer_check(az_trainer *t) {
    AZ_REQUIRE(t, AZ_EINVAL, "null trainer");
    AZ_HIP(hipStreamSynchronize(t->stream));
    return check_rows(t);
}

extern "C" int az_trainer_steps(az_trainer *t, const int8_t *d_state, const float *d_pi, const int8_t *d_z, int64_t n_samples, const int64_t *d_perm,
                                int32_t n_steps, int32_t B, float *d_loss_pi, float *d_loss_v, void *stream) {
    AZ_REQUIRE(t && d_state && d_pi && d_z && d_perm && d_loss_pi && d_loss_v, AZ_EINVAL, "null argument");
    AZ_REQUIRE(n_samples > 0, AZ_EINVAL, "n_samples = %lld: the sample arrays are empty", (long long)n_samples);
    if (t->game == AZ_TICTACTOE) AZ_REQUIRE(B >= 2 && B <= t->max_batch, AZ_EINVAL, "batch size %d outside [2, %d]", B, t->max_batch);
    else AZ_REQUIRE(B >= 16 && B <= t->max_batch && B % 16 == 0, AZ_EINVAL, "batch size %d: need a multiple of 16 in [16, %d]", B, t->max_batch);
    AZ_REQUIRE(n_steps >= 0, AZ_EINVAL, "negative step count");
    if (n_steps == 0) return AZ_OK;
    hipStream_t user = (hipStream_t)stream;
    // the flag of an earlier call nobody checked: reported FIRST, while this call has changed nothing (cached graphs, pointers, Hyper)
    AZ_HIP(hipStreamSynchronize(t->stream));
    AZ_TRY(check_rows(t, "az_trainer_steps"));
    AZ_TRY(t_enter(t, user));
    const bool same = t->g_state == d_state && t->g_pi == d_pi && t->g_z == d_z && t->g_perm == d_perm && t->g_lp == d_loss_pi && t->g_lv == d_loss_v && t->g_B == B;
    if (!same && t->graph) { (void)hipGraphExecDestroy(t->graph); t->graph = nullptr; }
    if (!same && t->graph_k) { (void)hipGraphExecDestroy(t->graph_k); t->graph_k = nullptr; }
    {   // workgroups per board in the conv kernels (AZ_TRAIN_SPLIT = 1 / 2 / 4 overrides; must divide 256)
        static int force = -1;
        if (force < 0) { const char *e = getenv("AZ_TRAIN_SPLIT"); force = e ? atoi(e) : 0; }
        int S = force == 1 || force == 2 || force == 4 ? force : 1;
        while (S > 1 && B * S > 256) S >>= 1;
        t->d.B = B; t->d.S = S; t->d.NB = B * S < 256 ? B * S : 256;
        const int rb = dense_row_block(B);
        t->d.RB = rb ? rb : B; t->d.NRB = rb ? (B + rb - 1) / rb : 1;
        // conv1's backward folded into conv2's (one launch less) where a workgroup holds whole boards and fc1's weight gradient does not
        // need k_conv1_bwd's launch (AZ_TRAIN_FOLD1=0: the separate kernel everywhere)
        static int fold = -1;
        if (fold < 0) { const char *e = getenv("AZ_TRAIN_FOLD1"); fold = e ? atoi(e) : 1; }
        t->d.F1B = (fold && S == 1 && rb == 0) ? 1 : 0;
    }
    t->q.state = d_state; t->q.pi = d_pi; t->q.z = d_z; t->q.perm = (const long long *)d_perm; t->q.loss_pi = d_loss_pi; t->q.loss_v = d_loss_v;
    t->tq.state = d_state; t->tq.pi = d_pi; t->tq.z = d_z; t->tq.perm = (const long long *)d_perm; t->tq.loss_pi = d_loss_pi; t->tq.loss_v = d_loss_v;
    t->g_state = d_state; t->g_pi = d_pi; t->g_z = d_z; t->g_perm = d_perm; t->g_lp = d_loss_pi; t->g_lv = d_loss_v; t->g_B = B;
    AZ_HIP(hipStreamSynchronize(t->stream));  // t_enter ordered the stream behind the caller's: the Hyper tail below is written by the host
    // perm_off, loss_off: this call's arrays start at 0 (the step counter keeps running: dropout streams); n_samples bounds its rows
    struct { int perm_off, loss_off; long long n_samples; int err, pad_; } tail = {0, 0, (long long)n_samples, 0, 0};
    static_assert(sizeof tail == sizeof(Hyper) - offsetof(Hyper, perm_off), "Hyper tail layout");
    AZ_HIP(hipMemcpy(&t->q.hp->perm_off, &tail, sizeof tail, hipMemcpyHostToDevice));
    int done = 0;
    if (t->steps_done == 0) {  // the very first step of a trainer runs as plain launches: kernel attributes get set outside a capture
        AZ_TRY(enqueue_step(t));
        done = 1; t->steps_done = 1;
    }
    // Replays: one graph of graph_k consecutive steps (a graph launch costs the host ~7 us per kernel node plus a fixed part, so
    // a 15-node graph per step would leave the host the bottleneck) and a one-step graph for the remainder.
    auto capture = [&](int k, hipGraphExec_t *out) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); t->graphs_ok = false; return; }
        int rc = AZ_OK;
        for (int i = 0; i < k && rc == AZ_OK; ++i) rc = enqueue_step(t);
        const hipError_t er = hipStreamEndCapture(t->stream, &g);
        if (rc != AZ_OK || er != hipSuccess || hipGraphInstantiate(out, g, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            *out = nullptr; t->graphs_ok = false;
        }
        if (g) (void)hipGraphDestroy(g);
    };
    if (t->graphs_ok && !t->graph && n_steps - done >= 1) capture(1, &t->graph);
    if (t->graphs_ok && !t->graph_k && t->graph_steps > 1 && n_steps - done >= t->graph_steps) capture(t->graph_steps, &t->graph_k);
    while (done < n_steps) {
        if (t->graph_k && n_steps - done >= t->graph_steps) { AZ_HIP(hipGraphLaunch(t->graph_k, t->stream)); done += t->graph_steps; t->steps_done += t->graph_steps; }
        else if (t->graph) { AZ_HIP(hipGraphLaunch(t->graph, t->stream)); done++; t->steps_done++; }
        else { AZ_TRY(enqueue_step(t)); done++; t->steps_done++; }
    }
    return t_leave(t, user);
}

// test access to the workspace (activations, gradients, partials) of the LAST step
extern "C" int az_trainer_debug(az_trainer *t, const char *name, void **d_ptr, int64_t *numel) {
    AZ_REQUIRE(t && name && d_ptr && numel, AZ_EINVAL, "null argument");
#ifdef AZ_TPROBE
    if (!strcmp(name, "probe")) {
        void *p = nullptr;
        AZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(az_tprobe)));
        *d_ptr = p; *numel = 32 * 16 * 2;  // as float-sized words
        return AZ_OK;
    }
#endif
    auto it = t->debug.find(name);
    AZ_REQUIRE(it != t->debug.end(), AZ_EINVAL, "unknown workspace buffer '%s'", name);
    *d_ptr = it->second.first; *numel = it->second.second;
    return AZ_OK;
}
