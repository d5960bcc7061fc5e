// az_engine.hip -- lock-step batched MCTS self-play on MI355X (gfx950).
//
// G concurrent games each advance ONE simulation per lock-step; the G pending leaves form one
// batch for the policy-value network.  Every tree still sees strictly sequential simulations, so
// per-game semantics equal the reference's MCT.search (mcts.py:226-269) exactly -- no virtual loss.
//
// HBM layout (struct of arrays, all indexed [slot] or [slot][node]):
//   boards   : 2 x u64 bitboards + int8 side-to-move per slot (root and current leaf)
//   tree     : per-slot bump-allocated node pool of `C` nodes; a node's children are contiguous
//              N:i32  Q:f64  P:f64  parent:i32  first_child:i32  n_children:u8  action:u8  flags:u8  winner:i8
//   net i/o  : nn_in[G][cells] f32 canonical leaf boards, probs[G][A] f32, value[G] f32
//   samples  : state i8[S][cells], pi f32[S][A], z i8[S], meta i32[S][4], visits i32[S][A]
//
// Lazy expansion (mcts.py:151-160) is kept observable-equivalent with eager allocation: when a leaf
// is evaluated its children are created at once from the renormalised priors but stay invisible
// (flag F_EXPANDED clear) until the node's next visit, which is when the reference materialises
// them.  This stores n_children priors per evaluated node instead of the raw probs[A] vector.
//
// v1 mapping: one thread per game (the net forward dominates the step).
#include <string.h>

#include <vector>

#include "az_device.h"
#include "az_host.h"

#define F_EXPANDED 1
#define F_TERMINAL 2
#define F_PF32 4
#define F_NOISED 8
#define F_EVALUATED 16

#define LS_NONE 0
#define LS_EVAL 1
#define LS_TERM 2

#define ERR_NODE_POOL 1
#define ERR_SAMPLE_CAP 2
#define ERR_PLY_CAP 4
#define ERR_INTERNAL 8

enum { CTR_SAMPLES = 0, CTR_GAMES_DONE, CTR_NET_EVALS, CTR_NEXT_GAME, CTR_TOTAL_GAMES, CTR_FIRST_ID, CTR_PLIES, CTR_COUNT };

struct EngDev {
    GameDesc gd;
    int G, C, A, max_plies;
    double alpha, eps;
    int tie_mode, noise_mode, tmax, tmin;
    u32 seed;
    long long sample_cap;
    u64 *root_p1, *root_m1; int8_t *root_player;
    int *root, *n_nodes, *ply; u32 *game_id; uint8_t *active, *root_fresh;
    int *leaf; u64 *leaf_p1, *leaf_m1; int8_t *leaf_player, *leaf_status, *leaf_winner;
    int *nN; double *nQ, *nP; int *nparent, *nfirst; uint8_t *nnch, *nact, *nflags; int8_t *nwin;
    float *nn_in, *probs, *value;
    int *samp_idx;
    int8_t *o_state; float *o_pi; int8_t *o_z; int *o_meta, *o_visits;
    unsigned long long *ctr;
    int *err, *max_nodes;
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
AZ_D void init_node(const EngDev &E, size_t i, int action, int parent, double P, int flags) {
    E.nN[i] = 0; E.nQ[i] = 0.0; E.nP[i] = P; E.nparent[i] = parent; E.nfirst[i] = -1;
    E.nnch[i] = 0; E.nact[i] = (uint8_t)action; E.nflags[i] = (uint8_t)flags; E.nwin[i] = 0;
}

AZ_D void start_position(const GameDesc &gd, BB &b) {
    b.p1 = 0; b.m1 = 0; b.player = 1;
    if (gd.game == AZ_OTHELLO) {  // othello.py:102-109
        int h = gd.H / 2;
        b.p1 = (1ULL << ((h - 1) * 8 + (h - 1))) | (1ULL << (h * 8 + h));
        b.m1 = (1ULL << ((h - 1) * 8 + h)) | (1ULL << (h * 8 + (h - 1)));
    }
}

AZ_D void write_nn_input(const EngDev &E, int g, const BB &b) {  // base.py:363 : player * grid
    float *dst = E.nn_in + (size_t)g * E.gd.cells;
    for (int r = 0; r < E.gd.H; ++r)
        for (int c = 0; c < E.gd.W; ++c) dst[r * E.gd.W + c] = (float)(b.player * az_cell_value(b, r, c));
}

// get_normalized_probs (othello.py:384-402, connect4.py:414-428, tictactoe.py:318-334) + add_child
AZ_D int create_children(const EngDev &E, int g, int node, const BB &bb, const float *pr) {
    const GameDesc &gd = E.gd;
    size_t base = (size_t)g * E.C;
    u64 bits = az_legal_bits(gd, bb, bb.player);
    bool pass = (gd.game == AZ_OTHELLO && bits == 0);
    int k = pass ? 1 : __popcll(bits);
    int fc = E.n_nodes[g];
    if (k <= 0 || fc + k > E.C) { atomicOr(E.err, k <= 0 ? ERR_INTERNAL : ERR_NODE_POOL); return -1; }
    E.n_nodes[g] = fc + k;
    atomicMax(E.max_nodes, fc + k);
    float s = 0.0f;  // float32 running sum in ascending action order
    if (pass) s += pr[gd.A - 1];
    else for (u64 m = bits; m; m &= m - 1) s += pr[az_bit_to_action(gd, __ffsll((long long)m) - 1)];
    bool uniform = s < 1e-6f;
    int i = 0;
    if (pass) {
        init_node(E, base + fc, gd.A - 1, node, uniform ? 1.0 : (double)(pr[gd.A - 1] / s), uniform ? 0 : F_PF32);
    } else {
        for (u64 m = bits; m; m &= m - 1, ++i) {
            int a = az_bit_to_action(gd, __ffsll((long long)m) - 1);
            init_node(E, base + fc + i, a, node, uniform ? 1.0 / (double)k : (double)(pr[a] / s), uniform ? 0 : F_PF32);
        }
    }
    E.nfirst[base + node] = fc;
    E.nnch[base + node] = (uint8_t)k;
    E.nflags[base + node] |= F_EVALUATED;
    return 0;
}

// fair_max over PUCT (mcts.py:44-46, 137; utils.py:28-34)
AZ_D int pick_child(const EngDev &E, int g, int node, int ply, int sim, int depth) {
    size_t base = (size_t)g * E.C;
    int fc = E.nfirst[base + node], nc = E.nnch[base + node];
    double sq = sqrt((double)E.nN[base + node]);
    double best = -__builtin_inf();
    int cnt = 0, first = 0;
    for (int i = 0; i < nc; ++i) {
        size_t c = base + fc + i;
        double key = E.nQ[c] + (E.nP[c] * sq) / (double)(1 + E.nN[c]);
        if (key > best) { best = key; cnt = 1; first = i; }
        else if (key == best) cnt++;
    }
    if (E.tie_mode == AZ_TIE_LOWEST) return fc + first;
    Philox4 r = az_philox(E.seed, E.game_id[g], (u32)ply, (u32)sim, AZ_P_TIE_SELECT, (u32)depth);
    int k = (int)(((u64)r.x * (u64)cnt) >> 32);
    if (cnt == 1) return fc + first;
    for (int i = 0; i < nc; ++i) {
        size_t c = base + fc + i;
        double key = E.nQ[c] + (E.nP[c] * sq) / (double)(1 + E.nN[c]);
        if (key == best) { if (k == 0) return fc + i; --k; }
    }
    return fc + first;
}

AZ_D double log_gamma_draw(const EngDev &E, u32 gid, int ply, int sim, double alpha, u32 j) {
    double d = (alpha + 1.0) - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d), g = d;
    for (u32 att = 0; att < 64; ++att) {
        Philox4 r = az_philox(E.seed, gid, (u32)ply, (u32)sim, AZ_P_NOISE_NORMAL, j | (att << 8));
        Philox4 q = az_philox(E.seed, gid, (u32)ply, (u32)sim, AZ_P_NOISE_NORMAL, j | (att << 8) | 0x80000000u);
        double u1 = 2.0 * az_u53(r.x, r.y) - 1.0, u2 = 2.0 * az_u53(r.z, r.w) - 1.0;
        double s = u1 * u1 + u2 * u2;
        if (!(s < 1.0) || s == 0.0) continue;
        double x = u1 * sqrt(-2.0 * az_det_log(s) / s);
        double v = 1.0 + c * x;
        if (!(v > 0.0)) continue;
        v = v * v * v;
        double u = 1.0 - az_u53(q.x, q.y);
        if (az_det_log(u) < 0.5 * x * x + d - d * v + d * az_det_log(v)) { g = d * v; break; }
    }
    Philox4 r = az_philox(E.seed, gid, (u32)ply, (u32)sim, AZ_P_NOISE_BOOST, j);
    double ub = 1.0 - az_u53(r.x, r.y);
    return az_det_log(g) + az_det_log(ub) / alpha;
}

// mcts.py:235-240 : P <- (1-eps) P + eps eta over the root's children
AZ_D void apply_root_noise(const EngDev &E, int g, int root, const BB &rb, int ply, int sim) {
    size_t base = (size_t)g * E.C;
    int fc = E.nfirst[base + root], k = E.nnch[base + root];
    double eta[AZ_MAX_ACTIONS];
    if (E.noise_mode == AZ_NOISE_HASH) {
        u64 h = az_board_hash(E.gd, rb), tot = 0;
        for (int i = 0; i < k; ++i) {
            u64 w = 1 + (az_splitmix64(h + (u64)(E.nact[base + fc + i] + 1) * 0xBF58476D1CE4E5B9ULL) >> 54);
            eta[i] = (double)w;
            tot += w;
        }
        for (int i = 0; i < k; ++i) eta[i] = eta[i] / (double)tot;
    } else {
        double m = -__builtin_inf(), s = 0.0;
        for (int i = 0; i < k; ++i) {
            eta[i] = log_gamma_draw(E, E.game_id[g], ply, sim, E.alpha, (u32)i);
            if (eta[i] > m) m = eta[i];
        }
        for (int i = 0; i < k; ++i) { eta[i] = az_det_exp(eta[i] - m); s += eta[i]; }
        for (int i = 0; i < k; ++i) eta[i] = eta[i] / s;
    }
    for (int i = 0; i < k; ++i) {
        size_t c = base + fc + i;
        double P = E.nP[c];
        double keep = (E.nflags[c] & F_PF32) ? (double)((float)(1.0 - E.eps) * (float)P) : (1.0 - E.eps) * P;
        E.nP[c] = keep + E.eps * eta[i];
        E.nflags[c] &= (uint8_t)~F_PF32;
    }
    E.nflags[base + root] |= F_NOISED;
}

AZ_D void back_propagate(const EngDev &E, int g, int node, int player_to_play, double outcome) {  // mcts.py:197-223
    size_t base = (size_t)g * E.C;
    double reward;
    if (fabs(outcome) < 1e-4) reward = 0.0;
    else reward = ((double)player_to_play * outcome > 0.0) ? -fabs(outcome) : fabs(outcome);
    while (node >= 0) {
        size_t i = base + node;
        int n = E.nN[i];
        E.nQ[i] = ((double)n * E.nQ[i] + reward) / (double)(n + 1);
        E.nN[i] = n + 1;
        node = E.nparent[i];
        reward = (reward == 0.0) ? 0.0 : -reward;
    }
}

AZ_D void reset_slot(const EngDev &E, int g, u32 game_id) {
    BB b;
    start_position(E.gd, b);
    E.root_p1[g] = b.p1; E.root_m1[g] = b.m1; E.root_player[g] = (int8_t)b.player;
    E.root[g] = 0; E.n_nodes[g] = 1; E.ply[g] = 0; E.game_id[g] = game_id; E.active[g] = 1;
    E.leaf_status[g] = LS_NONE;
    init_node(E, (size_t)g * E.C, 0, -1, 0.0, 0);
}

// ---------------------------------------------------------------------------------------------
// kernels (one thread per slot)
// ---------------------------------------------------------------------------------------------
__global__ void k_reset_all(EngDev E, u32 first_id, int n_games) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        for (int i = 0; i < CTR_COUNT; ++i) E.ctr[i] = 0;
        E.ctr[CTR_NEXT_GAME] = (unsigned long long)(n_games < E.G ? n_games : E.G);
        E.ctr[CTR_TOTAL_GAMES] = (unsigned long long)n_games;
        E.ctr[CTR_FIRST_ID] = first_id;
        *E.err = 0; *E.max_nodes = 0;
    }
    if (g >= E.G) return;
    if (g < n_games) reset_slot(E, g, first_id + (u32)g);
    else { E.active[g] = 0; E.leaf_status[g] = LS_NONE; }
}

// mcts.py:231-233 : a root without priors is evaluated first (value discarded)
__global__ void k_root_prep(EngDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    uint8_t fresh = 0;
    if (E.active[g]) {
        size_t r = (size_t)g * E.C + E.root[g];
        if (!(E.nflags[r] & (F_EVALUATED | F_TERMINAL))) {
            BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
            write_nn_input(E, g, b);
            fresh = 1;
        }
    }
    E.root_fresh[g] = fresh;
}

__global__ void k_root_init(EngDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G || !E.root_fresh[g]) return;
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    create_children(E, g, E.root[g], b, E.probs + (size_t)g * E.A);
    atomicAdd(&E.ctr[CTR_NET_EVALS], 1ULL);
}

// select_node (mcts.py:127-171) up to the point where the leaf needs its evaluation
__global__ void k_select(EngDev E, int sim) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    if (!E.active[g]) { E.leaf_status[g] = LS_NONE; return; }
    size_t base = (size_t)g * E.C;
    int ply = E.ply[g];
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    int node = E.root[g];
    if (E.noise_mode != AZ_NOISE_OFF && E.alpha >= 0.0 && E.eps >= 0.0) {  // mcts.py:235-240
        uint8_t f = E.nflags[base + node];
        if ((f & F_EXPANDED) && !(f & F_NOISED)) apply_root_noise(E, g, node, b, ply, sim);
    }
    int depth = 0;
    for (;;) {
        uint8_t f = E.nflags[base + node];
        if (f & F_EXPANDED) {
            int c = pick_child(E, g, node, ply, sim, depth++);
            az_play(E.gd, b, E.nact[base + c]);
            node = c;
            if (E.nN[base + c] == 0) break;  // mcts.py:143-144
            continue;
        }
        if (f & F_TERMINAL) break;  // mcts.py:146-147
        if (!(f & F_EVALUATED)) { atomicOr(E.err, ERR_INTERNAL); E.leaf_status[g] = LS_NONE; return; }
        E.nflags[base + node] = f | F_EXPANDED;  // mcts.py:151-160 : children become visible now
        int c = pick_child(E, g, node, ply, sim, depth);
        az_play(E.gd, b, E.nact[base + c]);
        node = c;
        break;
    }
    E.leaf[g] = node; E.leaf_p1[g] = b.p1; E.leaf_m1[g] = b.m1; E.leaf_player[g] = (int8_t)b.player;
    uint8_t lf = E.nflags[base + node];
    if (lf & F_TERMINAL) {
        E.leaf_status[g] = LS_TERM; E.leaf_winner[g] = E.nwin[base + node];
    } else {
        int w = 0;
        if (az_status(E.gd, b, &w)) {  // mcts.py:185-186
            E.nflags[base + node] = lf | F_TERMINAL; E.nwin[base + node] = (int8_t)w;
            E.leaf_status[g] = LS_TERM; E.leaf_winner[g] = (int8_t)w;
        } else {
            write_nn_input(E, g, b);
            E.leaf_status[g] = LS_EVAL;
        }
    }
}

// nn_evaluation bookkeeping (mcts.py:188-191) + back_propagate (mcts.py:197-223)
__global__ void k_backup(EngDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    int st = E.leaf_status[g];
    if (st == LS_NONE) return;
    int node = E.leaf[g];
    BB b = {E.leaf_p1[g], E.leaf_m1[g], E.leaf_player[g]};
    double outcome;
    if (st == LS_EVAL) {
        if (create_children(E, g, node, b, E.probs + (size_t)g * E.A) != 0) { E.active[g] = 0; return; }
        outcome = (double)b.player * (double)E.value[g];  // base.py:366
        atomicAdd(&E.ctr[CTR_NET_EVALS], 1ULL);
    } else {
        outcome = (double)E.leaf_winner[g];
    }
    back_propagate(E, g, node, b.player, outcome);
    E.leaf_status[g] = LS_NONE;
}

AZ_D double linear_temp(int step, int tmax, int tmin) {  // schedulers.py:33-40
    if (step <= tmax) return 1.0;
    if (step >= tmin) return 0.0;
    return 1.0 - (double)(step - tmax) / (double)(tmin - tmax);
}

// get_action_probs (mcts.py:95-116) + move choice (players.py:184-189) + Sample (trainer.py:244-250)
// + play_move / change_root (trainer.py:253-256) + end-of-game bookkeeping (trainer.py:262-268)
__global__ void k_move(EngDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G || !E.active[g]) return;
    const GameDesc &gd = E.gd;
    size_t base = (size_t)g * E.C;
    int ply = E.ply[g], root = E.root[g];
    u32 gid = E.game_id[g];
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    int fc = E.nfirst[base + root], nc = E.nnch[base + root];
    if (nc == 0 || !(E.nflags[base + root] & F_EXPANDED)) { atomicOr(E.err, ERR_INTERNAL); E.active[g] = 0; return; }
    double temp = linear_temp(ply, E.tmax, E.tmin);

    long long si = (long long)atomicAdd(&E.ctr[CTR_SAMPLES], 1ULL);
    if (si >= E.sample_cap) { atomicOr(E.err, ERR_SAMPLE_CAP); si = -1; }
    if (ply >= E.max_plies) { atomicOr(E.err, ERR_PLY_CAP); E.active[g] = 0; return; }
    E.samp_idx[(size_t)g * E.max_plies + ply] = (int)si;
    float *pi = si >= 0 ? E.o_pi + (size_t)si * E.A : nullptr;
    int *vis = si >= 0 ? E.o_visits + (size_t)si * E.A : nullptr;
    if (si >= 0) for (int a = 0; a < E.A; ++a) { pi[a] = 0.0f; vis[a] = 0; }

    int chosen = fc;
    if (temp == 0.0) {  // fair_max by N
        int best = -1, cnt = 0, first = 0;
        for (int i = 0; i < nc; ++i) {
            int n = E.nN[base + fc + i];
            if (n > best) { best = n; cnt = 1; first = i; } else if (n == best) cnt++;
        }
        int pick = first;
        if (E.tie_mode == AZ_TIE_RANDOM && cnt > 1) {
            Philox4 r = az_philox(E.seed, gid, (u32)ply, 0xFFFFu, AZ_P_TIE_MOVE, 0);
            int k = (int)(((u64)r.x * (u64)cnt) >> 32);
            for (int i = 0; i < nc; ++i)
                if (E.nN[base + fc + i] == best) { if (k == 0) { pick = i; break; } --k; }
        }
        chosen = fc + pick;
        if (si >= 0) pi[E.nact[base + chosen]] = 1.0f;
    } else {
        double sum = 0.0;
        for (int i = 0; i < nc; ++i) {
            double n = (double)E.nN[base + fc + i];
            sum += (temp == 1.0) ? n : pow(n, 1.0 / temp);
        }
        double u = 2.0, cum = 0.0;
        if (nc > 1) {
            Philox4 r = az_philox(E.seed, gid, (u32)ply, 0xFFFFu, AZ_P_MOVE_SAMPLE, 0);
            u = az_u53(r.x, r.y);
        }
        int last = 0; bool found = false;
        for (int i = 0; i < nc; ++i) {
            double n = (double)E.nN[base + fc + i];
            double p = ((temp == 1.0) ? n : pow(n, 1.0 / temp)) / sum;
            if (si >= 0) pi[E.nact[base + fc + i]] = (float)p;
            if (p > 0.0) last = i;
            cum += p;
            if (!found && u < cum) { chosen = fc + i; found = true; }
        }
        if (!found) chosen = fc + (nc == 1 ? 0 : last);
    }
    int action = E.nact[base + chosen];
    if (si >= 0) {
        for (int i = 0; i < nc; ++i) vis[E.nact[base + fc + i]] = E.nN[base + fc + i];
        int8_t *st = E.o_state + (size_t)si * gd.cells;
        for (int r = 0; r < gd.H; ++r)
            for (int c = 0; c < gd.W; ++c) st[r * gd.W + c] = (int8_t)(b.player * az_cell_value(b, r, c));
        int *m = E.o_meta + (size_t)si * 4;
        m[0] = (int)gid; m[1] = ply; m[2] = b.player; m[3] = action;
        E.o_z[si] = 0;
    }
    az_play(gd, b, action);
    E.root_p1[g] = b.p1; E.root_m1[g] = b.m1; E.root_player[g] = (int8_t)b.player;
    E.root[g] = chosen;
    E.nparent[base + chosen] = -1;  // mcts.py:121-123
    E.ply[g] = ply + 1;
    atomicAdd(&E.ctr[CTR_PLIES], 1ULL);
    int w = 0;
    if (az_status(gd, b, &w)) {  // trainer.py:235, 262-265
        for (int p = 0; p <= ply; ++p) {
            int s2 = E.samp_idx[(size_t)g * E.max_plies + p];
            if (s2 >= 0) E.o_z[s2] = (int8_t)(w * E.o_meta[(size_t)s2 * 4 + 2]);
        }
        atomicAdd(&E.ctr[CTR_GAMES_DONE], 1ULL);
        unsigned long long nx = atomicAdd(&E.ctr[CTR_NEXT_GAME], 1ULL);
        if (nx < E.ctr[CTR_TOTAL_GAMES]) reset_slot(E, g, (u32)E.ctr[CTR_FIRST_ID] + (u32)nx);
        else E.active[g] = 0;
    }
}

// closed-form fake network (tests): reads the canonical board back from nn_in
__global__ void k_fakenet(EngDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    const float *in = E.nn_in + (size_t)g * E.gd.cells;
    u64 h = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < E.gd.cells; ++i) h = (h ^ (u64)((int)in[i] + 1)) * 0x100000001B3ULL;
    h = az_splitmix64(h);
    float *pr = E.probs + (size_t)g * E.A;
    for (int a = 0; a < E.A; ++a) {
        u64 w = 1 + (az_splitmix64(h + (u64)(a + 1) * 0x9E3779B97F4A7C15ULL) >> 58);
        pr[a] = (float)w / 4096.0f;
    }
    u64 t = az_splitmix64(h ^ 0xD1B54A32D192ED03ULL);
    int sel = (int)((t >> 10) & 15);
    float v = ((float)(int)(t & 1023) - 512.0f) / 512.0f;
    if (sel == 0) v = 0.0f;
    if (sel == 1) v = 6.103515625e-05f;
    E.value[g] = v;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct az_engine {
    az_engine_cfg cfg;
    EngDev d;
    az_net *net;
    hipStream_t stream;
    std::vector<void *> allocs;
    unsigned long long *h_ctr;  // pinned
    int *h_err;                 // pinned [2] : err, max_nodes
    long long lockstep_iters;
};

int az_make_game_desc(int game, int H, int W, GameDesc *gd) {
    AZ_REQUIRE(game >= 0 && game <= 2, AZ_EINVAL, "unknown game id %d", game);
    if (game == AZ_OTHELLO) {
        AZ_REQUIRE(H == W && H >= 4 && H <= 8, AZ_EINVAL, "Othello board must be n x n with 4 <= n <= 8, got %dx%d", H, W);
        AZ_REQUIRE(H % 2 == 0, AZ_EINVAL, "Board size must be even but got n=%d", H);  // othello.py:87-88
    } else if (game == AZ_CONNECT4) {
        AZ_REQUIRE(H >= 4 && W >= 4, AZ_EINVAL, "Borad size must be at least 4x4, got %dx%d", W, H);  // connect4.py:90-91
        AZ_REQUIRE(H <= 8 && W <= 8, AZ_EINVAL, "Connect4 board larger than 8x8 is not supported (%dx%d)", W, H);
    } else {
        AZ_REQUIRE(H == 3 && W == 3, AZ_EINVAL, "TicTacToe board is 3x3");
    }
    gd->game = game; gd->H = H; gd->W = W; gd->cells = H * W;
    gd->A = game == AZ_OTHELLO ? H * W + 1 : (game == AZ_CONNECT4 ? W : 9);
    u64 v = 0;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) v |= 1ULL << (r * 8 + c);
    gd->valid = v;
    return AZ_OK;
}

template <typename T>
static int dev_alloc(az_engine *e, T **p, size_t n) {
    void *q = nullptr;
    AZ_HIP(hipMalloc(&q, n * sizeof(T)));
    AZ_HIP(hipMemsetAsync(q, 0, n * sizeof(T), e->stream));
    e->allocs.push_back(q);
    *p = (T *)q;
    return AZ_OK;
}

#define AZ_TRY(x) do { int _rc = (x); if (_rc != AZ_OK) return _rc; } while (0)

static inline dim3 grid_for(int n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }
#define TB 64

extern "C" int az_engine_create(const az_engine_cfg *cfg, az_net *net, void *stream, az_engine **out) {
    AZ_REQUIRE(cfg && out, AZ_EINVAL, "null argument");
    GameDesc gd;
    AZ_TRY(az_make_game_desc(cfg->game, cfg->H, cfg->W, &gd));
    AZ_REQUIRE(cfg->n_slots > 0 && cfg->n_sim > 0, AZ_EINVAL, "n_slots and n_sim must be positive");
    AZ_REQUIRE(cfg->node_capacity >= 2 * AZ_MAX_ACTIONS, AZ_EINVAL, "node_capacity too small");
    AZ_REQUIRE(cfg->max_plies > 0 && cfg->sample_capacity > 0, AZ_EINVAL, "max_plies / sample_capacity must be positive");
    AZ_REQUIRE(cfg->temp_min_step >= cfg->temp_max_step, AZ_EINVAL,
               "temp_min_step should be greater than temp_max_step for linear scheduler.");  // schedulers.py:29-30
    AZ_REQUIRE(cfg->evaluator == AZ_EVAL_FAKE || net != nullptr, AZ_ESTATE, "a network is required for AZ_EVAL_NET");
    if (cfg->evaluator == AZ_EVAL_NET)
        AZ_REQUIRE(az_net_action_size(net) == gd.A, AZ_EINVAL, "network action size %d != game action size %d",
                   az_net_action_size(net), gd.A);
    az_engine *e = new az_engine();
    e->cfg = *cfg; e->net = net; e->stream = (hipStream_t)stream; e->lockstep_iters = 0;
    EngDev &d = e->d;
    d.gd = gd; d.G = cfg->n_slots; d.C = cfg->node_capacity; d.A = gd.A; d.max_plies = cfg->max_plies;
    d.alpha = cfg->dirichlet_alpha; d.eps = cfg->dirichlet_epsilon; d.tie_mode = cfg->tie_mode;
    d.noise_mode = cfg->noise_mode; d.tmax = cfg->temp_max_step; d.tmin = cfg->temp_min_step; d.seed = cfg->seed;
    d.sample_cap = cfg->sample_capacity;
    size_t G = d.G, NC = G * (size_t)d.C, S = (size_t)cfg->sample_capacity;
    int rc = AZ_OK;
#define A_(p, n) if (rc == AZ_OK) rc = dev_alloc(e, &d.p, (n))
    A_(root_p1, G); A_(root_m1, G); A_(root_player, G); A_(root, G); A_(n_nodes, G); A_(ply, G); A_(game_id, G);
    A_(active, G); A_(root_fresh, G); A_(leaf, G); A_(leaf_p1, G); A_(leaf_m1, G); A_(leaf_player, G);
    A_(leaf_status, G); A_(leaf_winner, G);
    A_(nN, NC); A_(nQ, NC); A_(nP, NC); A_(nparent, NC); A_(nfirst, NC); A_(nnch, NC); A_(nact, NC); A_(nflags, NC); A_(nwin, NC);
    A_(nn_in, G * gd.cells); A_(probs, G * gd.A); A_(value, G);
    A_(samp_idx, G * (size_t)d.max_plies);
    A_(o_state, S * gd.cells); A_(o_pi, S * gd.A); A_(o_z, S); A_(o_meta, S * 4); A_(o_visits, S * gd.A);
    A_(ctr, CTR_COUNT); A_(err, 1); A_(max_nodes, 1);
#undef A_
    if (rc == AZ_OK && hipHostMalloc((void **)&e->h_ctr, sizeof(unsigned long long) * CTR_COUNT) != hipSuccess) rc = AZ_EHIP;
    if (rc == AZ_OK && hipHostMalloc((void **)&e->h_err, sizeof(int) * 2) != hipSuccess) rc = AZ_EHIP;
    if (rc != AZ_OK) { az_engine_destroy(e); return rc; }
    if (hipStreamSynchronize(e->stream) != hipSuccess) { az_engine_destroy(e); az_set_error("stream sync failed"); return AZ_EHIP; }
    *out = e;
    return AZ_OK;
}

extern "C" void az_engine_destroy(az_engine *e) {
    if (!e) return;
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->h_ctr) (void)hipHostFree(e->h_ctr);
    if (e->h_err) (void)hipHostFree(e->h_err);
    delete e;
}

static int forward(az_engine *e) {
    EngDev &d = e->d;
    if (e->cfg.evaluator == AZ_EVAL_FAKE) {
        hipLaunchKernelGGL(k_fakenet, grid_for(d.G, TB), dim3(TB), 0, e->stream, d);
        return AZ_OK;
    }
    return az_net_forward(e->net, d.nn_in, d.G, d.probs, d.value, e->stream);
}

static int do_search(az_engine *e, int n_sim) {
    EngDev &d = e->d;
    dim3 gr = grid_for(d.G, TB), bl(TB);
    hipLaunchKernelGGL(k_root_prep, gr, bl, 0, e->stream, d);
    AZ_TRY(forward(e));
    hipLaunchKernelGGL(k_root_init, gr, bl, 0, e->stream, d);
    for (int s = 0; s < n_sim; ++s) {
        hipLaunchKernelGGL(k_select, gr, bl, 0, e->stream, d, s);
        AZ_TRY(forward(e));
        hipLaunchKernelGGL(k_backup, gr, bl, 0, e->stream, d);
    }
    e->lockstep_iters += n_sim + 1;
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

static int fetch_counters(az_engine *e) {
    AZ_HIP(hipMemcpyAsync(e->h_ctr, e->d.ctr, sizeof(unsigned long long) * CTR_COUNT, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(&e->h_err[0], e->d.err, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(&e->h_err[1], e->d.max_nodes, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

static int check_err(az_engine *e) {
    int f = e->h_err[0];
    if (f & ERR_NODE_POOL) { az_set_error("tree node pool exhausted (node_capacity=%d)", e->cfg.node_capacity); return AZ_ECAPACITY; }
    if (f & ERR_SAMPLE_CAP) { az_set_error("sample buffer exhausted (sample_capacity=%lld)", (long long)e->cfg.sample_capacity); return AZ_ECAPACITY; }
    if (f & ERR_PLY_CAP) { az_set_error("game longer than max_plies=%d", e->cfg.max_plies); return AZ_ECAPACITY; }
    if (f & ERR_INTERNAL) { az_set_error("internal tree invariant violated"); return AZ_ESTATE; }
    return AZ_OK;
}

extern "C" int az_engine_run(az_engine *e, uint32_t first_game_id, int32_t n_games) {
    AZ_REQUIRE(e && n_games > 0, AZ_EINVAL, "bad arguments");
    EngDev &d = e->d;
    e->lockstep_iters = 0;
    hipLaunchKernelGGL(k_reset_all, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, (u32)first_game_id, (int)n_games);
    long long max_iters = ((long long)n_games / d.G + 2) * (long long)d.max_plies + 8;
    for (long long it = 0; it < max_iters; ++it) {
        AZ_TRY(do_search(e, e->cfg.n_sim));
        hipLaunchKernelGGL(k_move, grid_for(d.G, TB), dim3(TB), 0, e->stream, d);
        AZ_TRY(fetch_counters(e));
        AZ_TRY(check_err(e));
        if (e->h_ctr[CTR_GAMES_DONE] >= (unsigned long long)n_games) return AZ_OK;
    }
    az_set_error("self-play did not finish within %lld plies", max_iters);
    return AZ_ESTATE;
}

extern "C" int az_engine_get_stats(az_engine *e, az_engine_stats *out) {
    AZ_REQUIRE(e && out, AZ_EINVAL, "null argument");
    AZ_TRY(fetch_counters(e));
    out->games_done = (int64_t)e->h_ctr[CTR_GAMES_DONE];
    long long s = (long long)e->h_ctr[CTR_SAMPLES];
    out->samples = s < e->cfg.sample_capacity ? s : e->cfg.sample_capacity;
    out->net_evals = (int64_t)e->h_ctr[CTR_NET_EVALS];
    out->plies = (int64_t)e->h_ctr[CTR_PLIES];
    out->lockstep_iters = e->lockstep_iters;
    out->max_nodes_used = e->h_err[1];
    out->error_flags = e->h_err[0];
    return AZ_OK;
}

extern "C" int az_engine_samples(az_engine *e, int64_t *n_samples, const int8_t **d_states, const float **d_pis,
                                 const int8_t **d_zs, const int32_t **d_meta, const int32_t **d_visits) {
    AZ_REQUIRE(e && n_samples, AZ_EINVAL, "null argument");
    AZ_TRY(fetch_counters(e));
    long long s = (long long)e->h_ctr[CTR_SAMPLES];
    *n_samples = s < e->cfg.sample_capacity ? s : e->cfg.sample_capacity;
    if (d_states) *d_states = e->d.o_state;
    if (d_pis) *d_pis = e->d.o_pi;
    if (d_zs) *d_zs = e->d.o_z;
    if (d_meta) *d_meta = e->d.o_meta;
    if (d_visits) *d_visits = e->d.o_visits;
    return AZ_OK;
}

extern "C" int az_engine_set_roots(az_engine *e, const int8_t *h_grids, const int8_t *h_players, const uint32_t *h_game_ids,
                                   const int32_t *h_plies, int32_t n_roots) {
    AZ_REQUIRE(e && h_grids && h_players, AZ_EINVAL, "null argument");
    EngDev &d = e->d;
    AZ_REQUIRE(n_roots > 0 && n_roots <= d.G, AZ_EINVAL, "n_roots must be in [1, n_slots]");
    hipLaunchKernelGGL(k_reset_all, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, 0u, (int)n_roots);
    std::vector<u64> p1(n_roots), m1(n_roots);
    std::vector<u32> gid(n_roots);
    std::vector<int> ply(n_roots);
    for (int i = 0; i < n_roots; ++i) {
        u64 a = 0, b = 0;
        for (int r = 0; r < d.gd.H; ++r)
            for (int c = 0; c < d.gd.W; ++c) {
                int v = h_grids[(size_t)i * d.gd.cells + r * d.gd.W + c];
                AZ_REQUIRE(v >= -1 && v <= 1, AZ_EINVAL, "grid values must be -1, 0 or 1");
                if (v > 0) a |= 1ULL << (r * 8 + c);
                if (v < 0) b |= 1ULL << (r * 8 + c);
            }
        AZ_REQUIRE(h_players[i] == 1 || h_players[i] == -1, AZ_EINVAL, "player must be +1 or -1");
        p1[i] = a; m1[i] = b;
        gid[i] = h_game_ids ? h_game_ids[i] : (u32)i;
        ply[i] = h_plies ? h_plies[i] : 0;
    }
    AZ_HIP(hipMemcpyAsync(d.root_p1, p1.data(), sizeof(u64) * n_roots, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipMemcpyAsync(d.root_m1, m1.data(), sizeof(u64) * n_roots, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipMemcpyAsync(d.root_player, h_players, n_roots, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipMemcpyAsync(d.game_id, gid.data(), sizeof(u32) * n_roots, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipMemcpyAsync(d.ply, ply.data(), sizeof(int) * n_roots, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

extern "C" int az_engine_search(az_engine *e, int32_t n_sim) {
    AZ_REQUIRE(e && n_sim > 0, AZ_EINVAL, "bad arguments");
    AZ_TRY(do_search(e, n_sim));
    AZ_TRY(fetch_counters(e));
    return check_err(e);
}

extern "C" int az_engine_advance(az_engine *e) {
    AZ_REQUIRE(e, AZ_EINVAL, "null argument");
    EngDev &d = e->d;
    // games that end here must not be refilled: cap the queue at what has been started
    hipLaunchKernelGGL(k_move, grid_for(d.G, TB), dim3(TB), 0, e->stream, d);
    AZ_TRY(fetch_counters(e));
    return check_err(e);
}

extern "C" int az_engine_root_children(az_engine *e, int32_t slot, int32_t *h_actions, int32_t *h_N, double *h_Q,
                                       double *h_P, int32_t *count, int32_t *root_N) {
    AZ_REQUIRE(e && count, AZ_EINVAL, "null argument");
    EngDev &d = e->d;
    AZ_REQUIRE(slot >= 0 && slot < d.G, AZ_EINVAL, "slot out of range");
    AZ_HIP(hipStreamSynchronize(e->stream));
    int root = 0, fc = 0, rn = 0;
    uint8_t nc = 0, fl = 0;
    size_t base = (size_t)slot * d.C;
    AZ_HIP(hipMemcpy(&root, d.root + slot, sizeof(int), hipMemcpyDeviceToHost));
    AZ_HIP(hipMemcpy(&fc, d.nfirst + base + root, sizeof(int), hipMemcpyDeviceToHost));
    AZ_HIP(hipMemcpy(&nc, d.nnch + base + root, 1, hipMemcpyDeviceToHost));
    AZ_HIP(hipMemcpy(&fl, d.nflags + base + root, 1, hipMemcpyDeviceToHost));
    AZ_HIP(hipMemcpy(&rn, d.nN + base + root, sizeof(int), hipMemcpyDeviceToHost));
    if (root_N) *root_N = rn;
    if (!(fl & F_EXPANDED)) nc = 0;  // children not materialised yet in the reference's tree
    *count = nc;
    if (nc == 0) return AZ_OK;
    std::vector<uint8_t> act(nc);
    AZ_HIP(hipMemcpy(act.data(), d.nact + base + fc, nc, hipMemcpyDeviceToHost));
    if (h_actions) for (int i = 0; i < nc; ++i) h_actions[i] = act[i];
    if (h_N) AZ_HIP(hipMemcpy(h_N, d.nN + base + fc, sizeof(int) * nc, hipMemcpyDeviceToHost));
    if (h_Q) AZ_HIP(hipMemcpy(h_Q, d.nQ + base + fc, sizeof(double) * nc, hipMemcpyDeviceToHost));
    if (h_P) AZ_HIP(hipMemcpy(h_P, d.nP + base + fc, sizeof(double) * nc, hipMemcpyDeviceToHost));
    return AZ_OK;
}
