/*
 * az_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See az_oracle.h.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/alphazero/).  Board rules walk the int8 grid exactly like the
 * Python code does (no bitboards here: the HIP product uses bitboards, so the two
 * implementations are independent).  Tree statistics are float64 like the Python
 * objects; priors are float32 until Dirichlet noise promotes them.
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off).
 */
#include "az_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* deterministic math (same algorithms, operation by operation, in the HIP   */
/* kernels: alphazero_amd/csrc/az_detmath.h) so that CPU and GPU agree bit    */
/* for bit.  No libm calls except sqrt (IEEE-exact) on these paths.           */
/* ======================================================================== */

static inline float bits_to_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline double bits_to_d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static inline uint64_t d_to_bits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }

float orc_det_expf(float x) {
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(k, -0.693359375f, x);          /* ln2 hi (exact in 10 bits) */
    r = fmaf(k, 2.12194440e-4f, r);               /* -ln2 lo */
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int ki = (int)k;
    return p * bits_to_f((uint32_t)(ki + 127) << 23);
}

float orc_det_tanhf(float x) {
    float ax = fabsf(x);
    float t;
    if (ax > 10.0f) {
        t = 1.0f;
    } else {
        float e = orc_det_expf(-2.0f * ax);
        t = (1.0f - e) / (1.0f + e);
    }
    return x < 0.0f ? -t : t;
}

double orc_det_log(double x) {
    if (!(x > 0.0)) return -INFINITY;
    uint64_t u = d_to_bits(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = bits_to_d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    double lm = 2.0 * s * (1.0 + z * p);
    double de = (double)e;
    return de * 0.693147180369123816490 + (lm + de * 1.90821492927058770002e-10);
}

double orc_det_exp(double x) {
    if (!(x > -700.0)) return 0.0;
    if (x > 700.0) x = 700.0;
    double k = floor(x * 1.4426950408889634 + 0.5);
    double r = (x - k * 0.693147180369123816490) - k * 1.90821492927058770002e-10;
    double p = 1.0 / 87178291200.0; /* 1/14! */
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int ki = (int)k;
    return p * bits_to_d((uint64_t)(ki + 1023) << 52);
}

/* node.N ** (1. / temp) of MCT.get_action_probs (mcts.py:114-116) for temperatures other than 0 and 1, as the HIP engine computes
 * it (az_det_pow, csrc/az_device.h): exp(log(n) * inv_temp) on the fixed polynomials above -- the same bits on both sides; the
 * distance to libm's pow (what the reference calls) is pinned by tests/test_oracle_mct.py::test_det_pow_against_libm. */
double orc_det_pow(double n, double inv_temp) {
    if (!(n > 0.0)) return 0.0;
    return orc_det_exp(orc_det_log(n) * inv_temp);
}

void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { P_TIE_SELECT = 1, P_NOISE_NORMAL = 2, P_NOISE_BOOST = 3, P_MOVE_SAMPLE = 4, P_TIE_MOVE = 5,
       P_ROLLOUT_EXPAND = 6, P_PLAYOUT = 7 };

static inline double u53(uint32_t a, uint32_t b) { /* uniform in [0,1) with 53 bits */
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* ======================================================================== */
/* rules                                                                     */
/* ======================================================================== */

static const int DIRS[8][2] = {{1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}, {0, 1}}; /* othello.py:61 */

void orc_board_init(orc_board *b, int game, int H, int W) {
    memset(b, 0, sizeof(*b));
    b->game = game; b->H = H; b->W = W; b->player = 1;
    if (game == ORC_OTHELLO) { /* othello.py:102-109 */
        int n = H;
        b->grid[(n / 2 - 1) * n + (n / 2 - 1)] = 1;
        b->grid[(n / 2) * n + (n / 2)] = 1;
        b->grid[(n / 2 - 1) * n + (n / 2)] = -1;
        b->grid[(n / 2) * n + (n / 2 - 1)] = -1;
    }
}

int orc_action_size(const orc_board *b) {
    if (b->game == ORC_OTHELLO) return b->H * b->W + 1; /* othello.py:129-131 */
    if (b->game == ORC_CONNECT4) return b->W;           /* connect4.py:139-141 */
    return 9;                                           /* tictactoe.py:107-109 */
}

int orc_pass_action(const orc_board *b) { return b->game == ORC_OTHELLO ? b->H * b->W : -1; }

/* othello.py:141-153 : number of flips from (r,c) in direction d for `player`; cells in out */
static int oth_flips(const orc_board *b, int r, int c, int d, int player, int *out) {
    int n = b->H, cnt = 0;
    int rr = r + DIRS[d][0], cc = c + DIRS[d][1];
    while (rr >= 0 && rr < n && cc >= 0 && cc < n) {
        int v = b->grid[rr * n + cc];
        if (v == 0) return 0;
        if (v == player) return cnt;
        if (out) out[cnt] = rr * n + cc;
        cnt++;
        rr += DIRS[d][0]; cc += DIRS[d][1];
    }
    return 0;
}

static int oth_cell_legal(const orc_board *b, int r, int c, int player) {
    int n = b->H;
    if (r < 0 || r >= n || c < 0 || c >= n || b->grid[r * n + c] != 0) return 0;
    for (int d = 0; d < 8; ++d)
        if (oth_flips(b, r, c, d, player, NULL) > 0) return 1;
    return 0;
}

static int c4_free_rows(const orc_board *b, int col) { /* connect4.py:110-112 */
    int f = 0;
    for (int r = 0; r < b->H; ++r) f += (b->grid[r * b->W + col] == 0);
    return f;
}

int orc_is_legal(const orc_board *b, int action, int player) {
    if (player != 1 && player != -1) player = b->player; /* othello.py:143 */
    if (b->game == ORC_OTHELLO) { /* othello.py:155-174 */
        int n = b->H;
        if (action == n * n) {
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c)
                    if (b->grid[r * n + c] == 0)
                        for (int d = 0; d < 8; ++d)
                            if (oth_flips(b, r, c, d, player, NULL) > 0) return 0;
            return 1;
        }
        if (action < 0 || action > n * n) return 0;
        return oth_cell_legal(b, action / n, action % n, player);
    }
    if (b->game == ORC_CONNECT4) { /* connect4.py:146-156 */
        if (action < 0 || action >= b->W) return 0;
        return c4_free_rows(b, action) > 0;
    }
    if (action < 0 || action >= 9) return 0; /* tictactoe.py:139-143 */
    return b->grid[action] == 0;
}

int orc_legal_moves(const orc_board *b, int player, int *out) {
    if (player != 1 && player != -1) player = b->player;
    int cnt = 0;
    if (b->game == ORC_OTHELLO) { /* othello.py:176-189 */
        int n = b->H;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c)
                if (oth_cell_legal(b, r, c, player)) out[cnt++] = r * n + c;
        if (cnt == 0) out[cnt++] = n * n;
        return cnt;
    }
    if (b->game == ORC_CONNECT4) { /* connect4.py:158-163 */
        for (int c = 0; c < b->W; ++c)
            if (c4_free_rows(b, c) > 0) out[cnt++] = c;
        return cnt;
    }
    for (int i = 0; i < 9; ++i) /* tictactoe.py:145-150 */
        if (b->grid[i] == 0) out[cnt++] = i;
    return cnt;
}

int orc_play(orc_board *b, int action) {
    if (!orc_is_legal(b, action, 0)) return -1;
    if (b->game == ORC_OTHELLO) { /* othello.py:196-210 */
        int n = b->H;
        if (action == n * n) { b->player = -b->player; return 0; }
        int r = action / n, c = action % n, cells[8];
        for (int d = 0; d < 8; ++d) {
            int k = oth_flips(b, r, c, d, b->player, cells);
            for (int i = 0; i < k; ++i) b->grid[cells[i]] = (int8_t)b->player;
        }
        b->grid[r * n + c] = (int8_t)b->player;
        b->player = -b->player;
        return 0;
    }
    if (b->game == ORC_CONNECT4) { /* connect4.py:170-181 */
        int row = c4_free_rows(b, action) - 1;
        if (b->grid[row * b->W + action] != 0) return -1;
        b->grid[row * b->W + action] = (int8_t)b->player;
        b->player = -b->player;
        return 0;
    }
    b->grid[action] = (int8_t)b->player; /* tictactoe.py:157-164 */
    b->player = -b->player;
    return 0;
}

/* connect4.py:183-210 ; `flip` reads np.fliplr(grid) */
static int c4_check(const orc_board *b, int r, int c, int dr, int dc, int flip) {
    int pos = 0, neg = 0;
    while (r >= 0 && r < b->H && c >= 0 && c < b->W) {
        int v = b->grid[r * b->W + (flip ? (b->W - 1 - c) : c)];
        if (v == 1) { pos++; neg = 0; }
        else if (v == -1) { pos = 0; neg++; }
        else { pos = 0; neg = 0; }
        if (pos == 4) return 1;
        if (neg == 4) return -1;
        r += dr; c += dc;
    }
    return 0;
}

/* connect4.py:212-245 : returns 1/-1 winner, 0 draw, 2 = not over (None) */
static int c4_status(const orc_board *b) {
    int s;
    for (int flip = 0; flip < 2; ++flip) {
        for (int i = 1; i < b->H - 3; ++i) if ((s = c4_check(b, i, 0, 1, 1, flip)) != 0) return s;
        for (int i = 0; i < b->W - 3; ++i) if ((s = c4_check(b, 0, i, 1, 1, flip)) != 0) return s;
    }
    for (int i = 0; i < b->H; ++i) if ((s = c4_check(b, i, 0, 0, 1, 0)) != 0) return s;
    for (int i = 0; i < b->W; ++i) if ((s = c4_check(b, 0, i, 1, 0, 0)) != 0) return s;
    int free_cells = 0;
    for (int c = 0; c < b->W; ++c) free_cells += c4_free_rows(b, c);
    return free_cells == 0 ? 0 : 2;
}

static void ttt_sums(const orc_board *b, int s[8]) { /* tictactoe.py:111-117 */
    const int8_t *g = b->grid;
    for (int r = 0; r < 3; ++r) s[r] = g[3 * r] + g[3 * r + 1] + g[3 * r + 2];
    for (int c = 0; c < 3; ++c) s[3 + c] = g[c] + g[3 + c] + g[6 + c];
    s[6] = g[0] + g[4] + g[8];
    s[7] = g[2] + g[4] + g[6];
}

int orc_is_over(const orc_board *b) {
    if (b->game == ORC_OTHELLO) { /* othello.py:212-219 */
        int mv[ORC_MAX_ACTIONS], n = b->H;
        int k1 = orc_legal_moves(b, 1, mv); int p1 = (k1 == 1 && mv[0] == n * n);
        int k2 = orc_legal_moves(b, -1, mv); int p2 = (k2 == 1 && mv[0] == n * n);
        return p1 && p2;
    }
    if (b->game == ORC_CONNECT4) return c4_status(b) != 2; /* connect4.py:247-249 */
    int s[8]; ttt_sums(b, s); /* tictactoe.py:166-171 */
    for (int i = 0; i < 8; ++i) if (s[i] == 3 || s[i] == -3) return 1;
    for (int i = 0; i < 9; ++i) if (b->grid[i] == 0) return 0;
    return 1;
}

int orc_score(const orc_board *b) {
    if (b->game == ORC_TICTACTOE) { /* tictactoe.py:119-126 : inf (reported as 32767) if the side to move can win at once */
        int al[8], fr[8];
        ttt_sums(b, al);
        orc_board e = *b;
        for (int i = 0; i < 9; ++i) e.grid[i] = (int8_t)(b->grid[i] == 0);
        ttt_sums(&e, fr);
        for (int i = 0; i < 8; ++i) if (b->player * al[i] * (fr[i] > 0) == 2) return 32767;
        return 0;
    }
    int s = 0;
    for (int i = 0; i < b->H * b->W; ++i) s += b->player * b->grid[i];
    return s;
}

int orc_winner(const orc_board *b, int *winner) {
    if (b->game == ORC_OTHELLO) { /* othello.py:221-229 */
        if (!orc_is_over(b)) return -1;
        int s = orc_score(b);
        *winner = s == 0 ? 0 : (s > 0 ? b->player : -b->player);
        return 0;
    }
    if (b->game == ORC_CONNECT4) { /* connect4.py:251-258 */
        int s = c4_status(b);
        if (s == 2) return -1;
        *winner = s;
        return 0;
    }
    if (!orc_is_over(b)) return -1; /* tictactoe.py:173-184 */
    int s[8]; ttt_sums(b, s);
    for (int i = 0; i < 8; ++i) if (s[i] == 3) { *winner = 1; return 0; }
    for (int i = 0; i < 8; ++i) if (s[i] == -3) { *winner = -1; return 0; }
    *winner = 0;
    return 0;
}

/* ======================================================================== */
/* closed-form fake net (tests): priors are dyadic rationals so that the      */
/* float32 renormalisation of get_normalized_probs is order-independent.      */
/* ======================================================================== */

uint64_t orc_board_hash(const orc_board *b) { /* hash of the canonical board player*grid */
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < b->H * b->W; ++i) {
        uint64_t cg = (uint64_t)(b->player * b->grid[i] + 1);
        h = (h ^ cg) * 0x100000001B3ULL;
    }
    return splitmix64(h);
}

void orc_fakenet_eval(void *ctx, const orc_board *b, float *probs, float *v_net) {
    (void)ctx;
    uint64_t h = orc_board_hash(b);
    int A = orc_action_size(b);
    for (int a = 0; a < A; ++a) {
        uint64_t w = 1 + (splitmix64(h + (uint64_t)(a + 1) * 0x9E3779B97F4A7C15ULL) >> 58);
        probs[a] = (float)w / 4096.0f;
    }
    uint64_t t = splitmix64(h ^ 0xD1B54A32D192ED03ULL);
    int sel = (int)((t >> 10) & 15);
    float v = ((float)(int)(t & 1023) - 512.0f) / 512.0f;
    if (sel == 0) v = 0.0f;
    if (sel == 1) v = 6.103515625e-05f; /* 2^-14 < 1e-4 : exercises the draw threshold of mcts.py:208 */
    *v_net = v;
}

/* ======================================================================== */
/* networks                                                                   */
/* ======================================================================== */

#define NCH 32

struct orc_convnet {
    int game, H, W;      /* board shape */
    int ch, cw;          /* conv input plane: rows, cols (connect4 views (6,7) as (7,6), connect4.py:399) */
    int A, F1, F2, FIN;  /* action size, fc1 width, fc2 width, fc1 input */
    /* raw tensors (state_dict layout) */
    float *conv_w[4], *conv_b[4], *bn_g[4], *bn_b[4], *bn_m[4], *bn_v[4];
    float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *fbn_g[2], *fbn_b[2], *fbn_m[2], *fbn_v[2];
    float *fp_w, *fp_b, *fv_w, *fv_b;
    /* folded */
    float *cw_f[4]; /* conv1: [9][32] ; conv2-4: [9][32 ic][32 oc] */
    float *cb_f[4];
    float *c2u;     /* conv2 for the Winograd F(2x2,3x3) form: U[16 frequencies][32 ic][32 oc] = G g' G^T (float64, rounded once) */
    int wino;       /* conv2 runs in the Winograd form (the product's choice for this plane shape) */
    float *f1w, *f1b, *f2w, *f2b; /* [K][N] transposed for vectorisation over N */
    int qdense;                   /* fc1 / fc2 in the exact block-fixed-point form (the product's AZ_DENSE_I8 switch) */
    int32_t *f1q, *f2q;           /* quantised folded weights [N][K] */
    int *f1e, *f2e;               /* their per-output-column exponents */
    float *hw, *hb;               /* heads: [512][A+1] (last column = value) */
    int folded;
};

static float *dupf(const float *d, int64_t n) {
    float *p = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(p, d, sizeof(float) * (size_t)n);
    return p;
}

static void q_weights(const float *w, int K, int N, int32_t **q, int **e);
void orc_convnet_set_winograd(orc_convnet *n, int on) { n->wino = on; }
void orc_convnet_set_qdense(orc_convnet *n, int on) { n->qdense = on && n->game == ORC_OTHELLO; }
int orc_convnet_qdense(const orc_convnet *n) { return n->qdense; }
int orc_convnet_winograd(const orc_convnet *n) { return n->wino; }

orc_convnet *orc_convnet_create(int game, int H, int W) {
    orc_convnet *n = (orc_convnet *)calloc(1, sizeof(*n));
    n->game = game; n->H = H; n->W = W;
    if (game == ORC_OTHELLO) { n->ch = H; n->cw = W; n->A = H * W + 1; n->F1 = 1024; n->F2 = 512; }
    else { n->ch = W; n->cw = H; n->A = W; n->F1 = 64; n->F2 = 32; } /* connect4.py:360-365,399 */
    {   /* the product's policy (use_wino in az_net.hip), followed here so that the two stay bit-equal: conv2 in the Winograd form on
           8x8 and 7x6 planes by default, nowhere with AZ_WINOGRAD=0 */
        const char *e = getenv("AZ_WINOGRAD");
        const int mode = e ? atoi(e) : -1;
        n->wino = mode != 0 && ((n->ch == 8 && n->cw == 8) || (n->ch == 7 && n->cw == 6));
    }
    {   /* the product's policy (use_qdense in az_net.hip): OthelloNet's fc1 / fc2 in the exact block-fixed-point form when AZ_DENSE_I8=1 */
        const char *e = getenv("AZ_DENSE_I8");
        n->qdense = game == ORC_OTHELLO && e && atoi(e) == 1;
    }
    n->FIN = NCH * (n->ch - 4) * (n->cw - 4);
    return n;
}

void orc_convnet_destroy(orc_convnet *n) {
    if (!n) return;
    for (int i = 0; i < 4; ++i) {
        free(n->conv_w[i]); free(n->conv_b[i]); free(n->bn_g[i]); free(n->bn_b[i]); free(n->bn_m[i]); free(n->bn_v[i]);
        free(n->cw_f[i]); free(n->cb_f[i]);
    }
    for (int i = 0; i < 2; ++i) { free(n->fbn_g[i]); free(n->fbn_b[i]); free(n->fbn_m[i]); free(n->fbn_v[i]); }
    free(n->f1q); free(n->f2q); free(n->f1e); free(n->f2e);
    free(n->fc1_w); free(n->fc1_b); free(n->fc2_w); free(n->fc2_b); free(n->fp_w); free(n->fp_b); free(n->fv_w); free(n->fv_b);
    free(n->f1w); free(n->f1b); free(n->f2w); free(n->f2b); free(n->hw); free(n->hb); free(n->c2u);
    free(n);
}

static int set_slot(float **slot, const float *data, int64_t numel, int64_t expect) {
    if (numel != expect) return -1;
    free(*slot);
    *slot = dupf(data, numel);
    return 0;
}

int orc_convnet_set_tensor(orc_convnet *n, const char *name, const float *data, int64_t numel) {
    char buf[64];
    for (int i = 0; i < 4; ++i) {
        int64_t wn = (i == 0) ? NCH * 9 : NCH * NCH * 9;
        snprintf(buf, sizeof buf, "conv%d.weight", i + 1); if (!strcmp(name, buf)) return set_slot(&n->conv_w[i], data, numel, wn);
        snprintf(buf, sizeof buf, "conv%d.bias", i + 1); if (!strcmp(name, buf)) return set_slot(&n->conv_b[i], data, numel, NCH);
        snprintf(buf, sizeof buf, "bn%d.weight", i + 1); if (!strcmp(name, buf)) return set_slot(&n->bn_g[i], data, numel, NCH);
        snprintf(buf, sizeof buf, "bn%d.bias", i + 1); if (!strcmp(name, buf)) return set_slot(&n->bn_b[i], data, numel, NCH);
        snprintf(buf, sizeof buf, "bn%d.running_mean", i + 1); if (!strcmp(name, buf)) return set_slot(&n->bn_m[i], data, numel, NCH);
        snprintf(buf, sizeof buf, "bn%d.running_var", i + 1); if (!strcmp(name, buf)) return set_slot(&n->bn_v[i], data, numel, NCH);
    }
    if (!strcmp(name, "fc1.weight")) return set_slot(&n->fc1_w, data, numel, (int64_t)n->F1 * n->FIN);
    if (!strcmp(name, "fc1.bias")) return set_slot(&n->fc1_b, data, numel, n->F1);
    if (!strcmp(name, "fc2.weight")) return set_slot(&n->fc2_w, data, numel, (int64_t)n->F2 * n->F1);
    if (!strcmp(name, "fc2.bias")) return set_slot(&n->fc2_b, data, numel, n->F2);
    for (int i = 0; i < 2; ++i) {
        int64_t w = i == 0 ? n->F1 : n->F2;
        snprintf(buf, sizeof buf, "fc_bn%d.weight", i + 1); if (!strcmp(name, buf)) return set_slot(&n->fbn_g[i], data, numel, w);
        snprintf(buf, sizeof buf, "fc_bn%d.bias", i + 1); if (!strcmp(name, buf)) return set_slot(&n->fbn_b[i], data, numel, w);
        snprintf(buf, sizeof buf, "fc_bn%d.running_mean", i + 1); if (!strcmp(name, buf)) return set_slot(&n->fbn_m[i], data, numel, w);
        snprintf(buf, sizeof buf, "fc_bn%d.running_var", i + 1); if (!strcmp(name, buf)) return set_slot(&n->fbn_v[i], data, numel, w);
    }
    if (!strcmp(name, "fc_probs.weight")) return set_slot(&n->fp_w, data, numel, (int64_t)n->A * n->F2);
    if (!strcmp(name, "fc_probs.bias")) return set_slot(&n->fp_b, data, numel, n->A);
    if (!strcmp(name, "fc_value.weight")) return set_slot(&n->fv_w, data, numel, n->F2);
    if (!strcmp(name, "fc_value.bias")) return set_slot(&n->fv_b, data, numel, 1);
    return -2; /* unknown key (e.g. num_batches_tracked): caller may ignore */
}

#define BN_EPS 1e-5 /* torch.nn.BatchNorm default */

/* eval-mode BN folded into the preceding affine map, computed in float64:
 *   s = gamma / sqrt(var + eps);  w' = (float)(w * s);  b' = (float)((b - mean) * s + beta) */
int orc_convnet_fold(orc_convnet *n) {
    for (int i = 0; i < 4; ++i)
        if (!n->conv_w[i] || !n->conv_b[i] || !n->bn_g[i] || !n->bn_b[i] || !n->bn_m[i] || !n->bn_v[i]) return -1;
    if (!n->fc1_w || !n->fc1_b || !n->fc2_w || !n->fc2_b || !n->fp_w || !n->fp_b || !n->fv_w || !n->fv_b) return -1;
    for (int i = 0; i < 2; ++i) if (!n->fbn_g[i] || !n->fbn_b[i] || !n->fbn_m[i] || !n->fbn_v[i]) return -1;
    for (int l = 0; l < 4; ++l) {
        int IC = l == 0 ? 1 : NCH;
        free(n->cw_f[l]); free(n->cb_f[l]);
        n->cw_f[l] = (float *)malloc(sizeof(float) * 9 * IC * NCH);
        n->cb_f[l] = (float *)malloc(sizeof(float) * NCH);
        for (int oc = 0; oc < NCH; ++oc) {
            double s = (double)n->bn_g[l][oc] / sqrt((double)n->bn_v[l][oc] + BN_EPS);
            n->cb_f[l][oc] = (float)(((double)n->conv_b[l][oc] - (double)n->bn_m[l][oc]) * s + (double)n->bn_b[l][oc]);
            for (int ic = 0; ic < IC; ++ic)
                for (int t = 0; t < 9; ++t)
                    n->cw_f[l][(t * IC + ic) * NCH + oc] = (float)((double)n->conv_w[l][(oc * IC + ic) * 9 + t] * s);
        }
    }
    {   /* Winograd weights of conv2: U = G g' G^T with g' = w * s in float64, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] */
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        free(n->c2u);
        n->c2u = (float *)malloc(sizeof(float) * 16 * NCH * NCH);
        for (int oc = 0; oc < NCH; ++oc) {
            double s = (double)n->bn_g[1][oc] / sqrt((double)n->bn_v[1][oc] + BN_EPS);
            for (int ic = 0; ic < NCH; ++ic) {
                double g[3][3], t[4][3];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = (double)n->conv_w[1][(oc * NCH + ic) * 9 + a * 3 + b] * s;
                for (int i = 0; i < 4; ++i) for (int b = 0; b < 3; ++b) t[i][b] = (G[i][0] * g[0][b] + G[i][1] * g[1][b]) + G[i][2] * g[2][b];
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
                    n->c2u[((i * 4 + j) * NCH + ic) * NCH + oc] = (float)((t[i][0] * G[j][0] + t[i][1] * G[j][1]) + t[i][2] * G[j][2]);
            }
        }
    }
    free(n->f1w); free(n->f1b); free(n->f2w); free(n->f2b); free(n->hw); free(n->hb);
    n->f1w = (float *)malloc(sizeof(float) * (size_t)n->FIN * n->F1); n->f1b = (float *)malloc(sizeof(float) * n->F1);
    n->f2w = (float *)malloc(sizeof(float) * (size_t)n->F1 * n->F2); n->f2b = (float *)malloc(sizeof(float) * n->F2);
    for (int j = 0; j < n->F1; ++j) {
        double s = (double)n->fbn_g[0][j] / sqrt((double)n->fbn_v[0][j] + BN_EPS);
        n->f1b[j] = (float)(((double)n->fc1_b[j] - (double)n->fbn_m[0][j]) * s + (double)n->fbn_b[0][j]);
        for (int k = 0; k < n->FIN; ++k) n->f1w[(size_t)k * n->F1 + j] = (float)((double)n->fc1_w[(size_t)j * n->FIN + k] * s);
    }
    for (int j = 0; j < n->F2; ++j) {
        double s = (double)n->fbn_g[1][j] / sqrt((double)n->fbn_v[1][j] + BN_EPS);
        n->f2b[j] = (float)(((double)n->fc2_b[j] - (double)n->fbn_m[1][j]) * s + (double)n->fbn_b[1][j]);
        for (int k = 0; k < n->F1; ++k) n->f2w[(size_t)k * n->F2 + j] = (float)((double)n->fc2_w[(size_t)j * n->F1 + k] * s);
    }
    if (n->game == ORC_OTHELLO) {  /* always prepared: orc_convnet_set_qdense may switch the form on after the fold */
        q_weights(n->f1w, n->FIN, n->F1, &n->f1q, &n->f1e);
        q_weights(n->f2w, n->F1, n->F2, &n->f2q, &n->f2e);
    }
    int A1 = n->A + 1;
    n->hw = (float *)malloc(sizeof(float) * (size_t)n->F2 * A1); n->hb = (float *)malloc(sizeof(float) * A1);
    for (int a = 0; a < n->A; ++a) {
        n->hb[a] = n->fp_b[a];
        for (int k = 0; k < n->F2; ++k) n->hw[(size_t)k * A1 + a] = n->fp_w[(size_t)a * n->F2 + k];
    }
    n->hb[n->A] = n->fv_b[0];
    for (int k = 0; k < n->F2; ++k) n->hw[(size_t)k * A1 + n->A] = n->fv_w[k];
    n->folded = 1;
    return 0;
}

const float *orc_convnet_folded(const orc_convnet *n, const char *name, int64_t *numel) {
    if (!n->folded) return NULL;
    for (int l = 0; l < 4; ++l) {
        char buf[32];
        snprintf(buf, sizeof buf, "conv%d.w", l + 1);
        if (!strcmp(name, buf)) { *numel = 9 * (l == 0 ? 1 : NCH) * NCH; return n->cw_f[l]; }
        snprintf(buf, sizeof buf, "conv%d.b", l + 1);
        if (!strcmp(name, buf)) { *numel = NCH; return n->cb_f[l]; }
    }
    if (!strcmp(name, "fc1.w")) { *numel = (int64_t)n->FIN * n->F1; return n->f1w; }
    if (!strcmp(name, "fc1.b")) { *numel = n->F1; return n->f1b; }
    if (!strcmp(name, "fc2.w")) { *numel = (int64_t)n->F1 * n->F2; return n->f2w; }
    if (!strcmp(name, "fc2.b")) { *numel = n->F2; return n->f2b; }
    if (!strcmp(name, "heads.w")) { *numel = (int64_t)n->F2 * (n->A + 1); return n->hw; }
    if (!strcmp(name, "heads.b")) { *numel = n->A + 1; return n->hb; }
    return NULL;
}

/* one conv layer, 3x3, stride 1; accumulation order (also the HIP kernel's):
 *   acc = b'[oc];  for tap = ky*3+kx ascending: for ic ascending: acc = fmaf(in, w', acc);  relu.
 * Out-of-plane taps are skipped (== adding fmaf(0, w, acc)). */
static void conv_layer(const float *in, int IC, int ih, int iw, int pad, const float *w, const float *bias,
                       float *out, int oh, int ow) {
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            float acc[NCH];
            for (int oc = 0; oc < NCH; ++oc) acc[oc] = bias[oc];
            for (int t = 0; t < 9; ++t) {
                int iy = y + t / 3 - pad, ix = x + t % 3 - pad;
                if (iy < 0 || iy >= ih || ix < 0 || ix >= iw) continue;
                for (int ic = 0; ic < IC; ++ic) {
                    float a = in[(ic * ih + iy) * iw + ix];
                    const float *wr = w + (size_t)(t * IC + ic) * NCH;
                    for (int oc = 0; oc < NCH; ++oc) acc[oc] = fmaf(a, wr[oc], acc[oc]);
                }
            }
            for (int oc = 0; oc < NCH; ++oc) out[(oc * oh + y) * ow + x] = acc[oc] > 0.0f ? acc[oc] : 0.0f;
        }
}

/* conv2 ("same", 32 -> 32) in the Winograd F(2x2,3x3) form: the arithmetic of the HIP trunk kernels' Winograd conv2 path
 * (8x8 and 7x6 planes unless AZ_WINOGRAD=0; 2.25x fewer multiplications on full tiles).  Per 2x2 output tile with its zero-padded 4x4 input patch d (per input channel):
 *   V = B^T d B      T[0]=d[0]-d[2], T[1]=d[1]+d[2], T[2]=d[2]-d[1], T[3]=d[1]-d[3] (rows), then the same on the columns of T
 *   M[f][oc] = sum over ic ASCENDING of fmaf(V[f][ic], U[f][ic][oc], M) from 0, for each of the 16 frequencies f = 4 i + j
 *   Y = A^T M A      first along the columns of M: R[i][0]=(M[i][0]+M[i][1])+M[i][2], R[i][1]=(M[i][1]-M[i][2])-M[i][3],
 *                    then along the rows: Y[0][c]=(R[0][c]+R[1][c])+R[2][c], Y[1][c]=(R[1][c]-R[2][c])-R[3][c]
 *                    (the order in which the HIP kernels, which hold frequency rows 0-1 and 2-3 in two passes, can combine them)
 *   out = relu(Y + bias)
 * Every line is one IEEE float32 operation in the order written. */
static void conv2_winograd(const float *in, int ih, int iw, const float *U, const float *bias, float *out) {
    const int th = (ih + 1) / 2, tw = (iw + 1) / 2;
    for (int ty = 0; ty < th; ++ty)
        for (int tx = 0; tx < tw; ++tx) {
            float M[16][NCH];
            for (int f = 0; f < 16; ++f) for (int oc = 0; oc < NCH; ++oc) M[f][oc] = 0.0f;
            for (int ic = 0; ic < NCH; ++ic) {
                float d[4][4], T[4][4], V[4][4];
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        int iy = 2 * ty - 1 + a, ix = 2 * tx - 1 + b;
                        d[a][b] = (iy < 0 || iy >= ih || ix < 0 || ix >= iw) ? 0.0f : in[(ic * ih + iy) * iw + ix];
                    }
                for (int b = 0; b < 4; ++b) { T[0][b] = d[0][b] - d[2][b]; T[1][b] = d[1][b] + d[2][b]; T[2][b] = d[2][b] - d[1][b]; T[3][b] = d[1][b] - d[3][b]; }
                for (int a = 0; a < 4; ++a) { V[a][0] = T[a][0] - T[a][2]; V[a][1] = T[a][1] + T[a][2]; V[a][2] = T[a][2] - T[a][1]; V[a][3] = T[a][1] - T[a][3]; }
                for (int f = 0; f < 16; ++f) {
                    const float v = V[f / 4][f % 4];
                    const float *ur = U + (size_t)(f * NCH + ic) * NCH;
                    for (int oc = 0; oc < NCH; ++oc) M[f][oc] = fmaf(v, ur[oc], M[f][oc]);
                }
            }
            for (int oc = 0; oc < NCH; ++oc) {
                float R[4][2], Y[2][2];
                for (int i = 0; i < 4; ++i) {
                    R[i][0] = (M[4 * i + 0][oc] + M[4 * i + 1][oc]) + M[4 * i + 2][oc];
                    R[i][1] = (M[4 * i + 1][oc] - M[4 * i + 2][oc]) - M[4 * i + 3][oc];
                }
                for (int c = 0; c < 2; ++c) { Y[0][c] = (R[0][c] + R[1][c]) + R[2][c]; Y[1][c] = (R[1][c] - R[2][c]) - R[3][c]; }
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j) {
                        int y = 2 * ty + i, x = 2 * tx + j;
                        if (y < ih && x < iw) { float v = Y[i][j] + bias[oc]; out[(oc * ih + y) * iw + x] = v > 0.0f ? v : 0.0f; }
                    }
            }
        }
}

/* ---- exact block-fixed-point dense layers (AZ_DENSE_I8=1; az_net.hip: k_q_rows, k_q_cols, k_qgemm on v_mfma_i32_32x32x32_i8) ----
 * A vector x[0..K) (a row of activations, or the K weights of one output) shares one exponent: E = biased f32 exponent of max |x|
 * (clamped to [1, 254]), q[k] = rint(x[k] * 2^(148 - E)), so |q| <= 2^22 (three balanced base-256 digits on the GPU); non-finite
 * elements quantise to 0.  A dot product is the EXACT integer X = sum q_a q_b (any summation order gives the same X: every kernel
 * variant and tile shape agrees bit for bit by construction), and the layer's output is
 *   relu( (float)ldexp((double)X, E_a + E_b - 296) + bias )
 * with the int64 -> double and the double -> float conversions each rounded to nearest even. */
static int q_exponent(const float *x, int K, int stride) {
    uint32_t mx = 0;
    for (int k = 0; k < K; ++k) {
        uint32_t b;
        memcpy(&b, &x[(size_t)k * stride], 4);
        b &= 0x7fffffffu;
        if (b > mx) mx = b;
    }
    int E = (int)(mx >> 23);
    return E < 1 ? 1 : (E > 254 ? 254 : E);
}
static int32_t q_value(float x, int E) {
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7f800000u) == 0x7f800000u) return 0;
    return (int32_t)rintf(ldexpf(x, 148 - E));
}
/* the folded weights w[K][N] -> q[N][K], e[N] */
static void q_weights(const float *w, int K, int N, int32_t **q, int **e) {
    free(*q); free(*e);
    *q = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * N);
    *e = (int *)malloc(sizeof(int) * (size_t)N);
    for (int j = 0; j < N; ++j) {
        const int E = q_exponent(w + j, K, N);
        (*e)[j] = E;
        for (int k = 0; k < K; ++k) (*q)[(size_t)j * K + k] = q_value(w[(size_t)k * N + j], E);
    }
}
static void dense_layer_q(const float *x, int K, const int32_t *wq, const int *we, const float *b, int N, float *out, int relu) {
    int32_t qa[4096];
    const int Ea = q_exponent(x, K, 1);
    for (int k = 0; k < K; ++k) qa[k] = q_value(x[k], Ea);
    for (int j = 0; j < N; ++j) {
        const int32_t *qb = wq + (size_t)j * K;
        int64_t X = 0;
        for (int k = 0; k < K; ++k) X += (int64_t)qa[k] * (int64_t)qb[k];
        float v = (float)ldexp((double)X, Ea + we[j] - 296) + b[j];
        out[j] = (relu && !(v > 0.0f)) ? 0.0f : v;
    }
}

/* dense layer: acc[n] = b[n]; for k ascending: acc[n] = fmaf(x[k], W[k][n], acc[n]) */
static void dense_layer(const float *x, int K, const float *w, const float *b, int N, float *out, int relu) {
    for (int j = 0; j < N; ++j) out[j] = b[j];
    for (int k = 0; k < K; ++k) {
        float a = x[k];
        const float *wr = w + (size_t)k * N;
        for (int j = 0; j < N; ++j) out[j] = fmaf(a, wr[j], out[j]);
    }
    if (relu) for (int j = 0; j < N; ++j) out[j] = out[j] > 0.0f ? out[j] : 0.0f;
}

/* exp(log_softmax) of base.py:350-355 restated as a softmax:
 *   m = max; e_a = det_expf(l_a - m); S = sum ascending; p_a = e_a / S */
static void softmax_det(const float *logits, int A, float *probs) {
    float m = logits[0];
    for (int a = 1; a < A; ++a) m = logits[a] > m ? logits[a] : m;
    float s = 0.0f;
    for (int a = 0; a < A; ++a) { probs[a] = orc_det_expf(logits[a] - m); s += probs[a]; }
    for (int a = 0; a < A; ++a) probs[a] = probs[a] / s;
}

void orc_convnet_forward(const orc_convnet *n, const float *input, int B, float *probs, float *v) {
    int ch = n->ch, cw = n->cw;
    float *a1 = (float *)malloc(sizeof(float) * NCH * ch * cw);
    float *a2 = (float *)malloc(sizeof(float) * NCH * ch * cw);
    float *a3 = (float *)malloc(sizeof(float) * NCH * ch * cw);
    float *a4 = (float *)malloc(sizeof(float) * NCH * ch * cw);
    float *h1 = (float *)malloc(sizeof(float) * n->F1);
    float *h2 = (float *)malloc(sizeof(float) * n->F2);
    float *lg = (float *)malloc(sizeof(float) * (n->A + 1));
    for (int b = 0; b < B; ++b) {
        const float *x = input + (size_t)b * ch * cw;
        conv_layer(x, 1, ch, cw, 1, n->cw_f[0], n->cb_f[0], a1, ch, cw);              /* othello.py:370 */
        if (n->wino) conv2_winograd(a1, ch, cw, n->c2u, n->cb_f[1], a2);               /* :371, Winograd form */
        else conv_layer(a1, NCH, ch, cw, 1, n->cw_f[1], n->cb_f[1], a2, ch, cw);      /* :371 */
        conv_layer(a2, NCH, ch, cw, 0, n->cw_f[2], n->cb_f[2], a3, ch - 2, cw - 2);   /* :372 */
        conv_layer(a3, NCH, ch - 2, cw - 2, 0, n->cw_f[3], n->cb_f[3], a4, ch - 4, cw - 4); /* :373 */
        if (n->qdense) {
            dense_layer_q(a4, n->FIN, n->f1q, n->f1e, n->f1b, n->F1, h1, 1);            /* :376, exact fixed-point form */
            dense_layer_q(h1, n->F1, n->f2q, n->f2e, n->f2b, n->F2, h2, 1);             /* :377 */
        } else {
        dense_layer(a4, n->FIN, n->f1w, n->f1b, n->F1, h1, 1);                          /* :376 (dropout off in eval) */
        dense_layer(h1, n->F1, n->f2w, n->f2b, n->F2, h2, 1);                           /* :377 */
        }
        dense_layer(h2, n->F2, n->hw, n->hb, n->A + 1, lg, 0);                          /* :379-380 */
        softmax_det(lg, n->A, probs + (size_t)b * n->A);                                /* :382 + base.py:355 */
        v[b] = orc_det_tanhf(lg[n->A]);
    }
    free(a1); free(a2); free(a3); free(a4); free(h1); free(h2); free(lg);
}

void orc_convnet_eval(void *ctx, const orc_board *b, float *probs, float *v_net) { /* base.py:357-367 */
    const orc_convnet *n = (const orc_convnet *)ctx;
    float in[ORC_MAX_CELLS];
    for (int i = 0; i < b->H * b->W; ++i) in[i] = (float)(b->player * b->grid[i]);
    orc_convnet_forward(n, in, 1, probs, v_net);
}

struct orc_mlpnet {
    float *w[4], *b[4];                 /* fc1, fc2, fc_probs, fc_value */
    float *g[2], *be[2], *m[2], *va[2]; /* bn1, bn2 */
    float f1w[81], f1b[9], f2w[81], f2b[9], hw[90], hb[10];
    int folded;
};

orc_mlpnet *orc_mlpnet_create(void) { return (orc_mlpnet *)calloc(1, sizeof(orc_mlpnet)); }
void orc_mlpnet_destroy(orc_mlpnet *n) {
    if (!n) return;
    for (int i = 0; i < 4; ++i) { free(n->w[i]); free(n->b[i]); }
    for (int i = 0; i < 2; ++i) { free(n->g[i]); free(n->be[i]); free(n->m[i]); free(n->va[i]); }
    free(n);
}
int orc_mlpnet_set_tensor(orc_mlpnet *n, const char *name, const float *data, int64_t numel) {
    static const char *fc[4] = {"fc1", "fc2", "fc_probs", "fc_value"};
    char buf[64];
    for (int i = 0; i < 4; ++i) {
        int out = i == 3 ? 1 : 9;
        snprintf(buf, sizeof buf, "%s.weight", fc[i]); if (!strcmp(name, buf)) return set_slot(&n->w[i], data, numel, out * 9);
        snprintf(buf, sizeof buf, "%s.bias", fc[i]); if (!strcmp(name, buf)) return set_slot(&n->b[i], data, numel, out);
    }
    for (int i = 0; i < 2; ++i) {
        snprintf(buf, sizeof buf, "bn%d.weight", i + 1); if (!strcmp(name, buf)) return set_slot(&n->g[i], data, numel, 9);
        snprintf(buf, sizeof buf, "bn%d.bias", i + 1); if (!strcmp(name, buf)) return set_slot(&n->be[i], data, numel, 9);
        snprintf(buf, sizeof buf, "bn%d.running_mean", i + 1); if (!strcmp(name, buf)) return set_slot(&n->m[i], data, numel, 9);
        snprintf(buf, sizeof buf, "bn%d.running_var", i + 1); if (!strcmp(name, buf)) return set_slot(&n->va[i], data, numel, 9);
    }
    return -2;
}
int orc_mlpnet_fold(orc_mlpnet *n) {
    for (int i = 0; i < 4; ++i) if (!n->w[i] || !n->b[i]) return -1;
    for (int i = 0; i < 2; ++i) if (!n->g[i] || !n->be[i] || !n->m[i] || !n->va[i]) return -1;
    for (int l = 0; l < 2; ++l) {
        float *fw = l == 0 ? n->f1w : n->f2w, *fb = l == 0 ? n->f1b : n->f2b;
        for (int j = 0; j < 9; ++j) {
            double s = (double)n->g[l][j] / sqrt((double)n->va[l][j] + BN_EPS);
            fb[j] = (float)(((double)n->b[l][j] - (double)n->m[l][j]) * s + (double)n->be[l][j]);
            for (int k = 0; k < 9; ++k) fw[k * 9 + j] = (float)((double)n->w[l][j * 9 + k] * s);
        }
    }
    for (int a = 0; a < 9; ++a) { n->hb[a] = n->b[2][a]; for (int k = 0; k < 9; ++k) n->hw[k * 10 + a] = n->w[2][a * 9 + k]; }
    n->hb[9] = n->b[3][0];
    for (int k = 0; k < 9; ++k) n->hw[k * 10 + 9] = n->w[3][k];
    n->folded = 1;
    return 0;
}
void orc_mlpnet_forward(const orc_mlpnet *n, const float *input, int B, float *probs, float *v) {
    for (int b = 0; b < B; ++b) { /* tictactoe.py:298-316 */
        float h1[9], h2[9], lg[10];
        dense_layer(input + b * 9, 9, n->f1w, n->f1b, 9, h1, 1);
        dense_layer(h1, 9, n->f2w, n->f2b, 9, h2, 1);
        dense_layer(h2, 9, n->hw, n->hb, 10, lg, 0);
        softmax_det(lg, 9, probs + b * 9);
        v[b] = orc_det_tanhf(lg[9]);
    }
}
void orc_mlpnet_eval(void *ctx, const orc_board *b, float *probs, float *v_net) {
    float in[9];
    for (int i = 0; i < 9; ++i) in[i] = (float)(b->player * b->grid[i]);
    orc_mlpnet_forward((const orc_mlpnet *)ctx, in, 1, probs, v_net);
}

/* ======================================================================== */
/* Monte-Carlo tree (mcts.py)                                                 */
/* ======================================================================== */

typedef struct {
    int action, parent, N, first_child, n_children;
    double Q, P;
    int p_is_f32;     /* P is still the float32 of get_normalized_probs (numpy NEP-50 typing) */
    int has_noise;    /* mcts.py:26 */
    int64_t probs_off; /* offset into probs pool, -1 = None (mcts.py:25) */
} orc_node;

struct orc_mct {
    orc_mct_cfg cfg;
    orc_node *nodes; int n_nodes, cap_nodes;
    float *probs; int64_t n_probs, cap_probs;
    int root, A;
    int ply, sim;
    int64_t n_evals;
    int n_rollouts;
    int max_path; /* longest root..leaf path (nodes) any simulation walked: lets tests prove they reach deep paths */
    int rng_exhausted; /* the Gamma rejection sampler gave up (never observed; reported, not hidden) */
};

static int new_node(orc_mct *t, int action, int parent, double P, int p_is_f32) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (orc_node *)realloc(t->nodes, sizeof(orc_node) * (size_t)t->cap_nodes);
    }
    orc_node *n = &t->nodes[t->n_nodes];
    n->action = action; n->parent = parent; n->N = 0; n->first_child = -1; n->n_children = 0;
    n->Q = 0.0; n->P = P; n->p_is_f32 = p_is_f32; n->has_noise = 0; n->probs_off = -1;
    return t->n_nodes++;
}

static float *alloc_probs(orc_mct *t, int node, int A) {
    if (t->n_probs + A > t->cap_probs) {
        t->cap_probs = t->cap_probs ? t->cap_probs * 2 : 65536;
        t->probs = (float *)realloc(t->probs, sizeof(float) * (size_t)t->cap_probs);
    }
    t->nodes[node].probs_off = t->n_probs;
    t->n_probs += A;
    return t->probs + t->nodes[node].probs_off;
}

orc_mct *orc_mct_create(const orc_mct_cfg *cfg) {
    orc_mct *t = (orc_mct *)calloc(1, sizeof(*t));
    t->cfg = *cfg;
    orc_mct_reset(t, cfg->game_id);
    return t;
}
void orc_mct_destroy(orc_mct *t) { if (t) { free(t->nodes); free(t->probs); free(t); } }
void orc_mct_reset(orc_mct *t, uint32_t game_id) { /* players.py:228-234 : fresh MCT, same nn */
    t->cfg.game_id = game_id;
    t->n_nodes = 0; t->n_probs = 0; t->ply = 0; t->sim = 0; t->n_evals = 0; t->n_rollouts = 0;
    t->max_path = 0; t->rng_exhausted = 0;
    t->root = new_node(t, -1, -1, 0.0, 0);
}
void orc_mct_set_ply(orc_mct *t, int ply) { t->ply = ply; }
int orc_mct_root_n(const orc_mct *t) { return t->nodes[t->root].N; }
int orc_mct_n_nodes(const orc_mct *t) { return t->n_nodes; }
int64_t orc_mct_n_evals(const orc_mct *t) { return t->n_evals; }
int orc_mct_max_path_len(const orc_mct *t) { return t->max_path; }

static uint32_t rng_below(const orc_mct *t, int purpose, uint32_t idx, uint32_t n) {
    uint32_t r[4];
    orc_philox4x32(t->cfg.seed, t->cfg.game_id, (uint32_t)t->ply, (uint32_t)t->sim, (uint32_t)purpose, idx, r);
    return (uint32_t)(((uint64_t)r[0] * n) >> 32);
}

/* utils.py:28-34 : argmax with random tie-break over children [first, first+n) */
static int fair_max_child(const orc_mct *t, int first, int n, const double *key, int purpose, uint32_t idx) {
    double mx = key[0];
    for (int i = 1; i < n; ++i) if (key[i] > mx) mx = key[i];
    int ties[ORC_MAX_ACTIONS], nt = 0;
    for (int i = 0; i < n; ++i) if (key[i] == mx) ties[nt++] = i;
    if (t->cfg.tie_mode == ORC_TIE_LOWEST) return first + ties[0];
    return first + ties[rng_below(t, purpose, idx, (uint32_t)nt)]; /* drawn even when nt == 1 (utils.py:34) */
}

static double puct(const orc_mct *t, int c) { /* mcts.py:44-46, c_puct = 1.0 */
    const orc_node *n = &t->nodes[c];
    double pn = (double)t->nodes[n->parent].N;
    return n->Q + ((n->P * sqrt(pn)) / (double)(1 + n->N));
}
static double uct(const orc_mct *t, int c) { /* mcts.py:38-42, c = sqrt(2) */
    const orc_node *n = &t->nodes[c];
    if (n->N == 0) return INFINITY;
    /* np.log restated with the deterministic log shared with the HIP engine (differs from libm by <= 1 ulp) */
    return n->Q + sqrt(2.0) * sqrt(orc_det_log((double)t->nodes[n->parent].N) / (double)n->N);
}

static int pick_child(const orc_mct *t, int node, uint32_t depth) {
    const orc_node *n = &t->nodes[node];
    double key[ORC_MAX_ACTIONS];
    for (int i = 0; i < n->n_children; ++i)
        key[i] = t->cfg.eval_method == ORC_EVAL_NEURAL ? puct(t, n->first_child + i) : uct(t, n->first_child + i);
    return fair_max_child(t, n->first_child, n->n_children, key, P_TIE_SELECT, depth);
}

/* othello.py:384-402 / connect4.py:414-428 / tictactoe.py:318-334 */
static void expand_neural(orc_mct *t, int node, const orc_board *b) {
    int legal[ORC_MAX_ACTIONS];
    int k = orc_legal_moves(b, 0, legal);
    const float *probs = t->probs + t->nodes[node].probs_off;
    float s = 0.0f;
    for (int i = 0; i < k; ++i) s += probs[legal[i]]; /* float32 running sum, ascending action */
    int first = t->n_nodes;
    for (int i = 0; i < k; ++i) {
        if (s < 1e-6f) new_node(t, legal[i], node, 1.0 / (double)k, 0); /* uniform fallback: Python float */
        else new_node(t, legal[i], node, (double)(probs[legal[i]] / s), 1);
    }
    t->nodes[node].first_child = first;
    t->nodes[node].n_children = k;
}

static int select_node(orc_mct *t, orc_board *b) { /* mcts.py:127-171 */
    int node = t->root;
    uint32_t depth = 0;
#define PATH_SEEN(nodes_on_path) do { if ((int)(nodes_on_path) > t->max_path) t->max_path = (int)(nodes_on_path); } while (0)
    while (t->nodes[node].n_children != 0) {
        int c = pick_child(t, node, depth++);
        if (orc_play(b, t->nodes[c].action) != 0) return -1;
        node = c;
        if (t->nodes[node].N == 0) { PATH_SEEN(depth + 1); return node; }
    }
    PATH_SEEN(depth + 1);
    if (orc_is_over(b)) return node;
    if (t->cfg.eval_method == ORC_EVAL_ROLLOUT) {
        int legal[ORC_MAX_ACTIONS];
        int k = orc_legal_moves(b, 0, legal);
        int first = t->n_nodes;
        for (int i = 0; i < k; ++i) new_node(t, legal[i], node, 0.0, 0);
        t->nodes[node].first_child = first; t->nodes[node].n_children = k;
        int c = first + (int)rng_below(t, P_ROLLOUT_EXPAND, depth, (uint32_t)k); /* mcts.py:163-165 */
        if (orc_play(b, t->nodes[c].action) != 0) return -1;
        PATH_SEEN(depth + 2);
        return c;
    }
    if (t->nodes[node].probs_off < 0) return -1; /* "should not happen", mcts.py:157 */
    expand_neural(t, node, b);
    int c = pick_child(t, node, depth);
    if (orc_play(b, t->nodes[c].action) != 0) return -1;
    PATH_SEEN(depth + 2);
    return c;
#undef PATH_SEEN
}

static void back_propagate(orc_mct *t, int node, int player_id, double outcome) { /* mcts.py:197-223 */
    double reward;
    if (fabs(outcome) < 1e-4) reward = 0.0;
    else reward = ((double)player_id * outcome > 0.0) ? -fabs(outcome) : fabs(outcome);
    while (node >= 0) {
        orc_node *n = &t->nodes[node];
        n->Q = ((double)n->N * n->Q + reward) / (double)(n->N + 1);
        n->N += 1;
        node = n->parent;
        reward = (reward == 0.0) ? 0.0 : -reward;
    }
}

/* Gamma(alpha) for the Dirichlet draw, in log space: Marsaglia-Tsang on alpha+1 with the
 * U^(1/alpha) boost; normals by the polar method.  Statistically equal to np.random.dirichlet
 * (mcts.py:238); bit-equal to the HIP engine. */
static double log_gamma_draw(orc_mct *t, double alpha, uint32_t j) {
    double d = (alpha + 1.0) - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d), g = d;
    int accepted = 0;
    for (uint32_t att = 0; att < 64; ++att) {
        uint32_t r[4], q[4];
        orc_philox4x32(t->cfg.seed, t->cfg.game_id, (uint32_t)t->ply, (uint32_t)t->sim, P_NOISE_NORMAL, j | (att << 8), r);
        orc_philox4x32(t->cfg.seed, t->cfg.game_id, (uint32_t)t->ply, (uint32_t)t->sim, P_NOISE_NORMAL, j | (att << 8) | 0x80000000u, q);
        double u1 = 2.0 * u53(r[0], r[1]) - 1.0, u2 = 2.0 * u53(r[2], r[3]) - 1.0;
        double s = u1 * u1 + u2 * u2;
        if (!(s < 1.0) || s == 0.0) continue;
        double x = u1 * sqrt(-2.0 * orc_det_log(s) / s);
        double v = 1.0 + c * x;
        if (!(v > 0.0)) continue;
        v = v * v * v;
        double u = 1.0 - u53(q[0], q[1]);
        if (orc_det_log(u) < 0.5 * x * x + d - d * v + d * orc_det_log(v)) { g = d * v; accepted = 1; break; }
    }
    if (!accepted) t->rng_exhausted = 1; /* p < 1e-38 per draw; the search then fails instead of using g = d */
    uint32_t r[4];
    orc_philox4x32(t->cfg.seed, t->cfg.game_id, (uint32_t)t->ply, (uint32_t)t->sim, P_NOISE_BOOST, j, r);
    double ub = 1.0 - u53(r[0], r[1]);
    return orc_det_log(g) + orc_det_log(ub) / alpha;
}

static void root_noise(orc_mct *t, const orc_board *rootb) { /* mcts.py:235-240 */
    orc_node *root = &t->nodes[t->root];
    int k = root->n_children;
    double eta[ORC_MAX_ACTIONS];
    if (t->cfg.noise_mode == ORC_NOISE_HASH) {
        uint64_t h = orc_board_hash(rootb), tot = 0, w[ORC_MAX_ACTIONS];
        for (int i = 0; i < k; ++i) {
            int a = t->nodes[root->first_child + i].action;
            w[i] = 1 + (splitmix64(h + (uint64_t)(a + 1) * 0xBF58476D1CE4E5B9ULL) >> 54);
            tot += w[i];
        }
        for (int i = 0; i < k; ++i) eta[i] = (double)w[i] / (double)tot;
    } else {
        double lg[ORC_MAX_ACTIONS], m = -INFINITY, s = 0.0;
        for (int i = 0; i < k; ++i) { lg[i] = log_gamma_draw(t, t->cfg.dirichlet_alpha, (uint32_t)i); if (lg[i] > m) m = lg[i]; }
        for (int i = 0; i < k; ++i) { eta[i] = orc_det_exp(lg[i] - m); s += eta[i]; }
        for (int i = 0; i < k; ++i) eta[i] = eta[i] / s;
    }
    double eps = t->cfg.dirichlet_epsilon;
    for (int i = 0; i < k; ++i) {
        orc_node *c = &t->nodes[root->first_child + i];
        double keep = c->p_is_f32 ? (double)((float)(1.0 - eps) * (float)c->P) : (1.0 - eps) * c->P;
        c->P = keep + eps * eta[i];
        c->p_is_f32 = 0;
    }
}

static int rollout(orc_mct *t, orc_board *b) { /* mcts.py:173-180 */
    int legal[ORC_MAX_ACTIONS], w = 0;
    uint32_t step = 0;
    while (!orc_is_over(b)) {
        int k = orc_legal_moves(b, 0, legal);
        orc_play(b, legal[rng_below(t, P_PLAYOUT, step++, (uint32_t)k)]);
    }
    orc_winner(b, &w);
    return w;
}

static int search_iter(orc_mct *t, const orc_board *rootb) { /* mcts.py:226-252 */
    orc_board b = *rootb;
    float pr[ORC_MAX_ACTIONS], vnet;
    if (t->cfg.eval_method == ORC_EVAL_NEURAL) {
        if (t->nodes[t->root].probs_off < 0) {
            t->cfg.eval(t->cfg.eval_ctx, &b, pr, &vnet); /* value discarded, mcts.py:231-233 */
            t->n_evals++;
            memcpy(alloc_probs(t, t->root, t->A), pr, sizeof(float) * (size_t)t->A);
        }
        if (t->cfg.dirichlet_alpha >= 0.0 && t->cfg.dirichlet_epsilon >= 0.0 && t->cfg.noise_mode != ORC_NOISE_OFF) {
            orc_node *root = &t->nodes[t->root];
            if (!root->has_noise && root->n_children > 0) { root->has_noise = 1; root_noise(t, rootb); }
        }
    }
    int node = select_node(t, &b);
    if (node < 0) return -1;
    int player_to_play = b.player;
    double outcome;
    if (t->cfg.eval_method == ORC_EVAL_ROLLOUT) {
        outcome = (double)rollout(t, &b);
    } else { /* mcts.py:182-195 */
        if (orc_is_over(&b)) { int w = 0; orc_winner(&b, &w); outcome = (double)w; }
        else {
            if (t->nodes[node].probs_off >= 0) return -1;
            t->cfg.eval(t->cfg.eval_ctx, &b, pr, &vnet);
            t->n_evals++;
            memcpy(alloc_probs(t, node, t->A), pr, sizeof(float) * (size_t)t->A);
            outcome = (double)b.player * (double)vnet; /* base.py:366 */
        }
    }
    back_propagate(t, node, player_to_play, outcome);
    t->n_rollouts++;
    return 0;
}

int orc_mct_search(orc_mct *t, const orc_board *root, int n_sim) { /* mcts.py:254-269 */
    t->A = orc_action_size(root);
    t->n_rollouts = 0;
    for (int i = 0; i < n_sim; ++i) {
        t->sim = i;
        if (search_iter(t, root) != 0) return -1;
        if (t->rng_exhausted) return -1;
    }
    return 0;
}

void orc_mct_change_root(orc_mct *t, int action) { /* mcts.py:118-125 */
    const orc_node *r = &t->nodes[t->root];
    for (int i = 0; i < r->n_children; ++i)
        if (t->nodes[r->first_child + i].action == action) {
            t->root = r->first_child + i;
            t->nodes[t->root].parent = -1;
            return;
        }
    t->root = new_node(t, -1, -1, 0.0, 0);
}

int orc_mct_root_children(const orc_mct *t, int *actions, int *N, double *Q, double *P) {
    const orc_node *r = &t->nodes[t->root];
    for (int i = 0; i < r->n_children; ++i) {
        const orc_node *c = &t->nodes[r->first_child + i];
        if (actions) actions[i] = c->action;
        if (N) N[i] = c->N;
        if (Q) Q[i] = c->Q;
        if (P) P[i] = c->P;
    }
    return r->n_children;
}

int orc_mct_choose(orc_mct *t, const orc_board *rootb, double temp, double *pi, int *visits) {
    const orc_node *r = &t->nodes[t->root];
    int A = orc_action_size(rootb), k = r->n_children;
    t->sim = 0xFFFF;
    for (int a = 0; a < A; ++a) { pi[a] = 0.0; if (visits) visits[a] = 0; }
    if (k == 0) { int pa = orc_pass_action(rootb); if (pa >= 0) pi[pa] = 1.0; return pa; } /* mcts.py:101-102 */
    for (int i = 0; i < k; ++i) if (visits) visits[t->nodes[r->first_child + i].action] = t->nodes[r->first_child + i].N;
    if (temp == 0.0) { /* mcts.py:110-112 */
        double key[ORC_MAX_ACTIONS];
        for (int i = 0; i < k; ++i) key[i] = (double)t->nodes[r->first_child + i].N;
        int c = fair_max_child(t, r->first_child, k, key, P_TIE_MOVE, 0);
        pi[t->nodes[c].action] = 1.0;
        return t->nodes[c].action;
    }
    double val[ORC_MAX_ACTIONS], sum = 0.0; /* mcts.py:114-116 */
    const double inv_temp = 1.0 / temp;
    for (int i = 0; i < k; ++i) {
        double n = (double)t->nodes[r->first_child + i].N;
        val[i] = temp == 1.0 ? n : orc_det_pow(n, inv_temp);
        sum += val[i];
    }
    for (int i = 0; i < k; ++i) pi[t->nodes[r->first_child + i].action] = val[i] / sum;
    if (k == 1) return t->nodes[r->first_child].action; /* players.py:185-186 */
    uint32_t rr[4]; /* players.py:188-189 : inverse CDF over children in ascending action order */
    orc_philox4x32(t->cfg.seed, t->cfg.game_id, (uint32_t)t->ply, 0xFFFFu, P_MOVE_SAMPLE, 0, rr);
    double u = u53(rr[0], rr[1]), cum = 0.0;
    int last = -1;
    for (int i = 0; i < k; ++i) {
        double p = val[i] / sum;
        if (p > 0.0) last = i;
        cum += p;
        if (u < cum) return t->nodes[r->first_child + i].action;
    }
    return t->nodes[r->first_child + last].action;
}

/* RandomPlayer (kind 0, players.py:76-94: board.get_random_move) / GreedyPlayer (kind 1, players.py:97-123: best
 * -get_score() of the position after the move, fair_max among equals) for the side to move.  The uniform draw among the
 * candidates (ascending action order) is keyed like the HIP engine's k_baseline_moves. */
int orc_baseline_move(const orc_board *b, int kind, uint32_t seed, uint32_t game_id, int ply, int tie_mode) {
    int legal[ORC_MAX_ACTIONS], cand[ORC_MAX_ACTIONS], nc = 0;
    int k = orc_legal_moves(b, 0, legal);
    if (k <= 0) return -1;
    if (b->game == ORC_OTHELLO && legal[0] == orc_pass_action(b)) return legal[0]; /* forced pass: no draw */
    if (kind == 0) { for (int i = 0; i < k; ++i) cand[nc++] = legal[i]; }
    else {
        int best = -1000000;
        for (int i = 0; i < k; ++i) {
            orc_board c = *b;
            orc_play(&c, legal[i]);
            int sc = orc_score(&c);
            sc = (b->game == ORC_TICTACTOE) ? (sc != 0 ? -1 : 0) : -sc; /* -inf compares below 0 */
            if (sc > best) { best = sc; nc = 0; }
            if (sc == best) cand[nc++] = legal[i];
        }
    }
    if (kind == 1 && tie_mode == ORC_TIE_LOWEST) return cand[0]; /* the reference's fair_max patched to the lowest action (golden G7) */
    uint32_t r[4];
    orc_philox4x32(seed, game_id, (uint32_t)ply, 0xFFFEu, P_TIE_MOVE, (uint32_t)kind, r);
    return cand[(uint32_t)(((uint64_t)r[0] * (uint32_t)nc) >> 32)];
}

/* ======================================================================== */
/* self-play (trainer.py:215-273)                                             */
/* ======================================================================== */

static double linear_temp(int step, int tmax, int tmin) { /* schedulers.py:33-40 */
    if (step <= tmax) return 1.0;
    if (step >= tmin) return 0.0;
    return 1.0 - (double)(step - tmax) / (double)(tmin - tmax);
}

int64_t orc_selfplay(const orc_selfplay_cfg *cfg, orc_eval_fn eval, void *eval_ctx, uint32_t first_game_id,
                     int n_games, int64_t max_samples, int8_t *states, float *pis, int8_t *zs, int32_t *meta,
                     int32_t *visits, int64_t *n_evals) {
    orc_mct_cfg mc;
    mc.eval_method = cfg->eval_method; mc.eval = eval; mc.eval_ctx = eval_ctx;
    mc.dirichlet_alpha = cfg->dirichlet_alpha; mc.dirichlet_epsilon = cfg->dirichlet_epsilon;
    mc.tie_mode = cfg->tie_mode; mc.noise_mode = cfg->noise_mode; mc.seed = cfg->seed; mc.game_id = first_game_id;
    orc_mct *t = orc_mct_create(&mc);
    int64_t S = 0, evals = 0;
    orc_board b;
    orc_board_init(&b, cfg->game, cfg->H, cfg->W);
    int A = orc_action_size(&b), cells = cfg->H * cfg->W;
    for (int g = 0; g < n_games; ++g) {
        orc_board_init(&b, cfg->game, cfg->H, cfg->W); /* trainer.py:229 */
        orc_mct_reset(t, first_game_id + (uint32_t)g); /* trainer.py:230 */
        int64_t first = S;
        int move_counter = 0;
        while (!orc_is_over(&b)) { /* trainer.py:235 */
            if (S >= max_samples) { orc_mct_destroy(t); return -1; }
            double temp = linear_temp(move_counter, cfg->temp_max_step, cfg->temp_min_step);
            orc_mct_set_ply(t, move_counter);
            if (orc_mct_search(t, &b, cfg->n_sim) != 0) { orc_mct_destroy(t); return -1; }
            double pi[ORC_MAX_ACTIONS];
            int vis[ORC_MAX_ACTIONS];
            int action = orc_mct_choose(t, &b, temp, pi, vis);
            for (int i = 0; i < cells; ++i) states[S * cells + i] = b.grid[i]; /* raw; normalised below */
            for (int a = 0; a < A; ++a) pis[S * A + a] = (float)pi[a];
            if (visits) for (int a = 0; a < A; ++a) visits[S * A + a] = vis[a];
            meta[S * 4 + 0] = (int32_t)(first_game_id + (uint32_t)g);
            meta[S * 4 + 1] = move_counter;
            meta[S * 4 + 2] = b.player;
            meta[S * 4 + 3] = action;
            S++;
            if (orc_play(&b, action) != 0) { orc_mct_destroy(t); return -1; } /* trainer.py:253 */
            orc_mct_change_root(t, action);                                    /* trainer.py:256 */
            move_counter++;
        }
        int w = 0;
        orc_winner(&b, &w); /* trainer.py:262 */
        for (int64_t s = first; s < S; ++s) { /* Sample.normalize, trainer.py:74-78 */
            int p = meta[s * 4 + 2];
            for (int i = 0; i < cells; ++i) states[s * cells + i] = (int8_t)(states[s * cells + i] * p);
            zs[s] = (int8_t)(w * p);
        }
        evals += orc_mct_n_evals(t);
    }
    if (n_evals) *n_evals = evals;
    orc_mct_destroy(t);
    return S;
}

/* ======================================================================== */
/* batch helpers (tests compare whole arrays against the HIP kernels)         */
/* ======================================================================== */

static void load_board(orc_board *b, int game, int H, int W, const int8_t *grid, int player) {
    memset(b, 0, sizeof(*b));
    b->game = game; b->H = H; b->W = W; b->player = player;
    memcpy(b->grid, grid, (size_t)(H * W));
}

void orc_batch_legal(int game, int H, int W, const int8_t *grids, const int8_t *players, const int8_t *for_player,
                     int64_t n, uint8_t *legal) {
    orc_board b;
    int mv[ORC_MAX_ACTIONS];
    for (int64_t i = 0; i < n; ++i) {
        load_board(&b, game, H, W, grids + i * H * W, players[i]);
        int A = orc_action_size(&b);
        memset(legal + i * A, 0, (size_t)A);
        int k = orc_legal_moves(&b, for_player ? for_player[i] : 0, mv);
        for (int j = 0; j < k; ++j) legal[i * A + mv[j]] = 1;
    }
}

void orc_batch_play(int game, int H, int W, const int8_t *grids, const int8_t *players, const int32_t *actions, int64_t n,
                    int8_t *out_grids, int8_t *out_players, int32_t *status) {
    orc_board b;
    for (int64_t i = 0; i < n; ++i) {
        load_board(&b, game, H, W, grids + i * H * W, players[i]);
        status[i] = orc_play(&b, actions[i]) == 0 ? 0 : -5;
        memcpy(out_grids + i * H * W, b.grid, (size_t)(H * W));
        out_players[i] = (int8_t)b.player;
    }
}

void orc_batch_status(int game, int H, int W, const int8_t *grids, const int8_t *players, int64_t n, uint8_t *over,
                      int8_t *winner, int32_t *score) {
    orc_board b;
    for (int64_t i = 0; i < n; ++i) {
        load_board(&b, game, H, W, grids + i * H * W, players[i]);
        int w = 2;
        over[i] = (uint8_t)orc_is_over(&b);
        if (over[i]) orc_winner(&b, &w);
        winner[i] = (int8_t)(over[i] ? w : 2);
        int s = 0;
        for (int c = 0; c < H * W; ++c) s += b.player * b.grid[c];
        score[i] = s;
    }
}

/* random playouts: records every position reached (for large-scale GPU parity sweeps) */
int64_t orc_random_positions(int game, int H, int W, uint32_t seed, int n_games, int64_t cap, int8_t *grids, int8_t *players,
                             int32_t *actions) {
    int64_t n = 0;
    int mv[ORC_MAX_ACTIONS];
    for (int g = 0; g < n_games; ++g) {
        orc_board b;
        orc_board_init(&b, game, H, W);
        uint32_t step = 0;
        while (!orc_is_over(&b) && n < cap) {
            int k = orc_legal_moves(&b, 0, mv);
            uint32_t r[4];
            orc_philox4x32(seed, (uint32_t)g, step++, 0, 99, 0, r);
            int a = mv[(uint32_t)(((uint64_t)r[0] * (uint32_t)k) >> 32)];
            memcpy(grids + n * H * W, b.grid, (size_t)(H * W));
            players[n] = (int8_t)b.player;
            actions[n] = a;
            n++;
            orc_play(&b, a);
        }
    }
    return n;
}
