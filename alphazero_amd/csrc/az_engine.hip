// az_engine.hip -- lock-step batched MCTS self-play on MI355X (gfx950).
//
// G concurrent games each advance ONE simulation per lock-step; the G pending leaves form one
// batch for the policy-value network.  Every tree still sees strictly sequential simulations, so
// per-game semantics equal the reference's MCT.search (mcts.py:226-269) exactly -- no virtual loss.
//
// HBM layout
//   boards   : 2 x u64 bitboards + int8 side-to-move per slot (root and current leaf), SoA over slots
//   tree     : per slot two pools of `C` 32-byte nodes (one node = two 16-byte accesses; a node's children
//              are contiguous, so 16 lanes read 16 children as one 512-byte run).  Nodes are bump-allocated
//              during a search; at every move the kept subtree is copied breadth-first into the other pool
//              (change_root, mcts.py:118-125), so a game's live tree stays a few hundred KB, contiguous:
//              { f64 Q, f64 P, i32 N, i32 parent, i32 first_child, u8 n_children, u8 action, u8 flags, i8 winner }
//   net i/o  : nn_in[G][cells] f32 canonical leaf boards, probs[G][A] f32, value[G] f32
//   samples  : state i8[S][cells], pi f32[S][A], z i8[S], meta i32[S][4], visits i32[S][A]
//
// Mapping: 16 lanes per game (4 games per wavefront, 16 per 256-thread block).  PUCT scoring, prior
// renormalisation, child creation and Dirichlet draws are spread over the 16 lanes (one child
// each); arg-max / tie counting use width-16 shuffles and wave ballots; the board of the walk is
// replicated in the group's registers and advanced with bitboard shifts.
//
// Lazy expansion (mcts.py:151-160) is kept observable-equivalent with eager allocation: when a leaf
// is evaluated its children are created at once from the renormalised priors but stay invisible
// (flag F_EXPANDED clear) until the node's next visit, which is when the reference materialises
// them.  This stores n_children priors per evaluated node instead of the raw probs[A] vector.
#include <string.h>

#include <map>
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
#define ERR_RNG 16  // the Gamma rejection sampler of the root noise gave up (p < 1e-38 per draw): reported, never papered over

#define LPG 16          // lanes per game
#define GPB (256 / LPG) // games per 256-thread block

enum { CTR_SAMPLES = 0, CTR_GAMES_DONE, CTR_NET_EVALS, CTR_NEXT_GAME, CTR_TOTAL_GAMES, CTR_FIRST_ID, CTR_PLIES, CTR_COUNT };

struct __attribute__((aligned(16))) Node {
    double Q, P;
    int N, parent, first;
    uint8_t nch, act, flags;
    int8_t win;
};
static_assert(sizeof(Node) == 32, "node is two 16-byte accesses");

struct EngDev {
    GameDesc gd;
    int G, C, A, max_plies;
    double alpha, eps;
    int tie_mode, noise_mode, tmax, tmin;
    u32 sim_base;  // simulations already run on the current roots: keeps Philox counters distinct across repeated az_engine_search calls
    int rollout;  // 1: TreeEval.ROLLOUT (plain UCT + random playouts, mcts.py:38-42, 152-154, 173-180), no network
    u32 seed;
    long long sample_cap;
    u64 *root_p1, *root_m1; int8_t *root_player;
    int *root, *n_nodes, *ply; u32 *game_id; uint8_t *active, *root_fresh;
    int8_t *side;  // arena: the colour this engine searches for in the slot (0 = both, self-play)
    int *leaf; u64 *leaf_p1, *leaf_m1; int8_t *leaf_player, *leaf_status, *leaf_winner;
    int *path, *path_len;  // root..leaf node indices of the pending simulation ([G][LPG]; longer paths chase parents)
    Node *nodes;        // [G][2][C]
    uint8_t *pool_sel;  // which of the slot's two pools holds the live tree
    float *nn_in, *probs, *value;
    int *row_of_slot;  // network batch row holding the slot's pending leaf (leaves are compacted)
    int *evals;        // per-slot network evaluations since the last move (folded into ctr[] once per ply)
    int *batch_cnt;    // [3] rows filled: two alternating lock-step counters + the root-prior pass
    int *samp_idx;
    int8_t *o_state; float *o_pi; int8_t *o_z; int *o_meta, *o_visits;
    unsigned long long *ctr;
    int *err, *max_nodes;
    int *max_path;  // longest root..leaf path (nodes) of any simulation, recorded only beyond LPG (the parent-chasing backup)
};

// ---------------------------------------------------------------------------------------------
// node access (two 16-byte transactions) and 16-lane group primitives
// ---------------------------------------------------------------------------------------------
AZ_D Node load_node(const Node *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Node n;
    n.Q = __longlong_as_double((long long)(((u64)a.y << 32) | a.x));
    n.P = __longlong_as_double((long long)(((u64)a.w << 32) | a.z));
    n.N = (int)b.x; n.parent = (int)b.y; n.first = (int)b.z;
    n.nch = (uint8_t)(b.w & 0xff); n.act = (uint8_t)((b.w >> 8) & 0xff); n.flags = (uint8_t)((b.w >> 16) & 0xff);
    n.win = (int8_t)(b.w >> 24);
    return n;
}

AZ_D void store_node(Node *p, const Node &n) {
    u64 q = (u64)__double_as_longlong(n.Q), pp = (u64)__double_as_longlong(n.P);
    uint4 a = make_uint4((u32)q, (u32)(q >> 32), (u32)pp, (u32)(pp >> 32));
    uint4 b = make_uint4((u32)n.N, (u32)n.parent, (u32)n.first,
                         (u32)n.nch | ((u32)n.act << 8) | ((u32)n.flags << 16) | ((u32)(uint8_t)n.win << 24));
    uint4 *d = reinterpret_cast<uint4 *>(p);
    d[0] = a; d[1] = b;
}

AZ_D Node fresh_node(int action, int parent, double P, int flags) {
    Node n;
    n.Q = 0.0; n.P = P; n.N = 0; n.parent = parent; n.first = -1; n.nch = 0; n.act = (uint8_t)action;
    n.flags = (uint8_t)flags; n.win = 0;
    return n;
}

AZ_D Node *pool_of(const EngDev &E, int g) { return E.nodes + ((size_t)g * 2 + E.pool_sel[g]) * E.C; }

// does this engine search slot g now? (active, and in arena mode only when its colour is to move)
AZ_D bool searches(const EngDev &E, int g) { return E.active[g] && (E.side[g] == 0 || E.side[g] == E.root_player[g]); }

AZ_D u32 grp_ballot(bool p) { return (u32)(__ballot(p) >> (threadIdx.x & 48)) & 0xFFFFu; }
AZ_D double grp_max(double v) {  // over the game's 16 lanes, by DPP (az_device.h: the same maximum in every lane as the shuffle butterfly gave)
    v = fmax(v, __longlong_as_double((long long)az_dpp64<AZ_DPP_SWAP1>((u64)__double_as_longlong(v))));
    v = fmax(v, __longlong_as_double((long long)az_dpp64<AZ_DPP_SWAP2>((u64)__double_as_longlong(v))));
    v = fmax(v, __longlong_as_double((long long)az_dpp64<AZ_DPP_HMIRROR>((u64)__double_as_longlong(v))));
    v = fmax(v, __longlong_as_double((long long)az_dpp64<AZ_DPP_MIRROR>((u64)__double_as_longlong(v))));
    return v;
}
AZ_D u64 grp_sum_u64(u64 v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += (u64)__shfl_xor((long long)v, m, LPG);
    return v;
}
AZ_D int kth_set_bit(u32 mask, int k) {
    for (int i = 0; i < k; ++i) mask &= mask - 1;
    return __ffs((int)mask) - 1;
}

// next free row of the network batch for this group (lane 0 draws it, the group gets it by shuffle)
AZ_D int alloc_row_grp(int *counter, int sub) {
    int row = 0;
    if (sub == 0) row = atomicAdd(counter, 1);
    return __shfl(row, 0, LPG);
}

AZ_D void start_position(const GameDesc &gd, BB &b) {
    b.p1 = 0; b.m1 = 0; b.player = 1;
    if (gd.game == AZ_OTHELLO) {  // othello.py:102-109
        int h = gd.H / 2;
        b.p1 = (1ULL << ((h - 1) * 8 + (h - 1))) | (1ULL << (h * 8 + h));
        b.m1 = (1ULL << ((h - 1) * 8 + h)) | (1ULL << (h * 8 + (h - 1)));
    }
}

// base.py:363 : player * grid, spread over the group's lanes (coalesced row of `cells` floats)
AZ_D void write_nn_input_grp(const EngDev &E, int row, const BB &b, int sub) {
    float *dst = E.nn_in + (size_t)row * E.gd.cells;
    for (int i = sub; i < E.gd.cells; i += LPG) dst[i] = (float)(b.player * az_cell_value(b, i / E.gd.W, i % E.gd.W));
}

// get_normalized_probs (othello.py:384-402, connect4.py:414-428, tictactoe.py:318-334) + add_child,
// one lane per legal action bit.  `fc` = the slot's bump pointer (first free node); returns the number of
// children created, or -1 on pool exhaustion.
AZ_D int create_children_grp(const EngDev &E, int g, Node *pool, int fc, int node, const BB &bb, const float *pr, int sub) {
    const GameDesc &gd = E.gd;
    u64 bits = az_legal_bits_grp(gd, bb, bb.player, sub);
    bool pass = (gd.game == AZ_OTHELLO && bits == 0);
    int k = pass ? 1 : __popcll(bits);
    if (k <= 0 || fc + k > E.C) { if (sub == 0) atomicOr(E.err, k <= 0 ? ERR_INTERNAL : ERR_NODE_POOL); return -1; }
    // priors of this lane's (up to 4) legal actions: independent loads, one round trip for the whole row
    float myp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bit = r * LPG + sub;
        myp[r] = (!pass && ((bits >> bit) & 1ULL)) ? pr[az_bit_to_action(gd, bit)] : 0.0f;
    }
    const float ppass = pass ? pr[gd.A - 1] : 0.0f;
    // float32 running sum in ASCENDING action order, as the reference accumulates it: the operands come from
    // their lanes by shuffle, so the order is kept without a dependent global load per action
    float s = 0.0f;
    if (pass) s += ppass;
    else
        for (u64 m = bits; m; m &= m - 1) {
            const int bit = __ffsll((long long)m) - 1, r = bit >> 4;
            const float mine = r == 0 ? myp[0] : (r == 1 ? myp[1] : (r == 2 ? myp[2] : myp[3]));
            s += __shfl(mine, bit & 15, LPG);
        }
    bool uniform = s < 1e-6f;
    if (pass) {
        if (sub == 0) store_node(pool + fc, fresh_node(gd.A - 1, node, uniform ? 1.0 : (double)(ppass / s), uniform ? 0 : F_PF32));
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int bit = r * LPG + sub;
            if ((bits >> bit) & 1ULL) {
                int idx = __popcll(bits & ((1ULL << bit) - 1ULL));
                int a = az_bit_to_action(gd, bit);
                store_node(pool + fc + idx, fresh_node(a, node, uniform ? 1.0 / (double)k : (double)(myp[r] / s), uniform ? 0 : F_PF32));
            }
        }
    }
    if (sub == 0) {
        E.n_nodes[g] = fc + k;
        pool[node].first = fc;
        pool[node].nch = (uint8_t)k;
        pool[node].flags |= F_EVALUATED;
    }
    return k;
}

// fair_max over PUCT (mcts.py:44-46, 137; utils.py:28-34): one lane per child, 16 children per round.
// The chosen child's node is handed back by shuffle from the lane that scored it (no second memory trip).
AZ_D int pick_child_grp(const EngDev &E, int g, const Node *pool, const Node &parent, int ply, int sim, int depth, int sub,
                        Node &chosen) {
    const int fc = parent.first, nc = parent.nch;
    const double sq = sqrt((double)parent.N);
    double key[4];
    int cN[4], cfirst[4];
    u32 cpack[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        key[r] = -__builtin_inf();
        cN[r] = 0; cfirst[r] = -1; cpack[r] = 0;
        int i = r * LPG + sub;
        if (i < nc) {
            Node c = load_node(pool + fc + i);
            key[r] = c.Q + (c.P * sq) / (double)(1 + c.N);
            cN[r] = c.N; cfirst[r] = c.first;
            cpack[r] = (u32)c.nch | ((u32)c.act << 8) | ((u32)c.flags << 16) | ((u32)(uint8_t)c.win << 24);
        }
    }
    double best = grp_max(fmax(fmax(key[0], key[1]), fmax(key[2], key[3])));
    u32 mask[4];
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { mask[r] = grp_ballot(key[r] == best); cnt += __popc(mask[r]); }
    if (cnt == 0 && sub == 0) atomicOr(E.err, ERR_INTERNAL);  // NaN scores (a diverged network): no key equals the maximum
    int k = 0;
    if (E.tie_mode == AZ_TIE_RANDOM && cnt > 1) {  // with a single maximum the draw cannot change the result
        Philox4 rr = az_philox(E.seed, E.game_id[g], (u32)ply, (u32)sim + E.sim_base, AZ_P_TIE_SELECT, (u32)depth);
        k = (int)(((u64)rr.x * (u64)cnt) >> 32);
    }
    int rsel = 0, lsel = 0;
    bool found = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int pc = __popc(mask[r]);
        if (!found) {
            if (k < pc) { rsel = r; lsel = kth_set_bit(mask[r], k); found = true; }
            else k -= pc;
        }
    }
    const int vN = rsel == 0 ? cN[0] : (rsel == 1 ? cN[1] : (rsel == 2 ? cN[2] : cN[3]));
    const int vF = rsel == 0 ? cfirst[0] : (rsel == 1 ? cfirst[1] : (rsel == 2 ? cfirst[2] : cfirst[3]));
    const u32 vP = rsel == 0 ? cpack[0] : (rsel == 1 ? cpack[1] : (rsel == 2 ? cpack[2] : cpack[3]));
    chosen.N = __shfl(vN, lsel, LPG);
    chosen.first = __shfl(vF, lsel, LPG);
    const u32 pk = (u32)__shfl((int)vP, lsel, LPG);
    chosen.nch = (uint8_t)(pk & 0xff); chosen.act = (uint8_t)((pk >> 8) & 0xff); chosen.flags = (uint8_t)((pk >> 16) & 0xff);
    chosen.win = (int8_t)(pk >> 24);
    chosen.Q = 0.0; chosen.P = 0.0; chosen.parent = 0;  // not needed by the walk
    return fc + rsel * LPG + lsel;
}

AZ_D double log_gamma_draw(const EngDev &E, u32 gid, int ply, int sim, double alpha, u32 j) {
    double d = (alpha + 1.0) - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d), g = d;
    bool accepted = false;
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
        if (az_det_log(u) < 0.5 * x * x + d - d * v + d * az_det_log(v)) { g = d * v; accepted = true; break; }
    }
    if (!accepted) atomicOr(E.err, ERR_RNG);
    Philox4 r = az_philox(E.seed, gid, (u32)ply, (u32)sim, AZ_P_NOISE_BOOST, j);
    double ub = 1.0 - az_u53(r.x, r.y);
    return az_det_log(g) + az_det_log(ub) / alpha;
}

AZ_D double sel4(const double (&v)[4], int r) { return r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3])); }

// mcts.py:235-240 : P <- (1-eps) P + eps eta over the root's children, one lane per child
AZ_D void apply_root_noise_grp(const EngDev &E, int g, Node *pool, int root, const Node &rn, const BB &rb, int ply, int sim, int sub) {
    const int fc = rn.first, k = rn.nch;
    double eta[4];
    Node ch[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { int i = r * LPG + sub; ch[r] = load_node(pool + fc + (i < k ? i : 0)); }
    if (E.noise_mode == AZ_NOISE_HASH) {
        u64 h = az_board_hash(E.gd, rb), w[4], tot = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = r * LPG + sub;
            w[r] = i < k ? 1 + (az_splitmix64(h + (u64)(ch[r].act + 1) * 0xBF58476D1CE4E5B9ULL) >> 54) : 0;
            tot += w[r];
        }
        tot = grp_sum_u64(tot);
#pragma unroll
        for (int r = 0; r < 4; ++r) eta[r] = (double)w[r] / (double)tot;
    } else {
        double lg[4], m = -__builtin_inf();
        const u32 gid = E.game_id[g];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = r * LPG + sub;
            lg[r] = i < k ? log_gamma_draw(E, gid, ply, sim, E.alpha, (u32)i) : -__builtin_inf();
            m = fmax(m, lg[r]);
        }
        m = grp_max(m);
#pragma unroll
        for (int r = 0; r < 4; ++r) eta[r] = (r * LPG + sub) < k ? az_det_exp(lg[r] - m) : 0.0;
        double s = 0.0;  // sequential sum in ascending child order, as in the CPU restatement
        for (int i = 0; i < k; ++i) s += __shfl(sel4(eta, i >> 4), i & 15, LPG);
#pragma unroll
        for (int r = 0; r < 4; ++r) eta[r] = eta[r] / s;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = r * LPG + sub;
        if (i < k) {
            double P = ch[r].P;
            double keep = (ch[r].flags & F_PF32) ? (double)((float)(1.0 - E.eps) * (float)P) : (1.0 - E.eps) * P;
            pool[fc + i].P = keep + E.eps * eta[r];
            pool[fc + i].flags = ch[r].flags & (uint8_t)~F_PF32;
        }
    }
    if (sub == 0) pool[root].flags = rn.flags | F_NOISED;
}

// back_propagate (mcts.py:197-223).  The select step recorded the root..leaf path, so every node on it
// is updated by its own lane in one memory round trip (reward sign alternates from the leaf up);
// paths longer than 16 nodes fall back to chasing parent pointers.  `mine` = this lane's path node (already
// loaded), returns true when the fast path ran (lane 0 then holds the root with its N already incremented).
AZ_D bool back_propagate_grp(Node *pool, int len, int my_node, Node &mine, int leaf, int player_to_play, double outcome, int sub) {
    double reward;
    if (fabs(outcome) < 1e-4) reward = 0.0;
    else reward = ((double)player_to_play * outcome > 0.0) ? -fabs(outcome) : fabs(outcome);
    if (len <= LPG) {
        if (sub < len) {
            const int up = len - 1 - sub;  // edges above the leaf
            const double r = (reward == 0.0) ? 0.0 : ((up & 1) ? -reward : reward);
            pool[my_node].Q = ((double)mine.N * mine.Q + r) / (double)(mine.N + 1);
            pool[my_node].N = mine.N + 1;
            mine.N += 1;
        }
        return true;
    }
    int node = leaf;
    while (node >= 0) {
        Node n = load_node(pool + node);  // same address in all 16 lanes: one broadcast transaction
        if (sub == 0) {
            pool[node].Q = ((double)n.N * n.Q + reward) / (double)(n.N + 1);
            pool[node].N = n.N + 1;
        }
        node = n.parent;
        reward = (reward == 0.0) ? 0.0 : -reward;
    }
    return false;
}

AZ_D void reset_slot(const EngDev &E, int g, u32 game_id) {
    BB b;
    start_position(E.gd, b);
    E.root_p1[g] = b.p1; E.root_m1[g] = b.m1; E.root_player[g] = (int8_t)b.player;
    E.root[g] = 0; E.n_nodes[g] = 1; E.ply[g] = 0; E.game_id[g] = game_id; E.active[g] = 1;
    E.leaf_status[g] = LS_NONE; E.evals[g] = 0; E.side[g] = 0;
    store_node(pool_of(E, g), fresh_node(0, -1, 0.0, 0));
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_reset_all(EngDev E, u32 first_id, int n_games) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        for (int i = 0; i < CTR_COUNT; ++i) E.ctr[i] = 0;
        E.ctr[CTR_NEXT_GAME] = (unsigned long long)(n_games < E.G ? n_games : E.G);
        E.ctr[CTR_TOTAL_GAMES] = (unsigned long long)n_games;
        E.ctr[CTR_FIRST_ID] = first_id;
        *E.err = 0; *E.max_nodes = 0; *E.max_path = 0;
        E.batch_cnt[0] = E.batch_cnt[1] = E.batch_cnt[2] = 0;
    }
    if (g >= E.G) return;
    E.pool_sel[g] = 0;
    if (g < n_games) reset_slot(E, g, first_id + (u32)g);
    else { E.active[g] = 0; E.leaf_status[g] = LS_NONE; }
}

// mcts.py:231-233 : a root without priors is evaluated first (value discarded)
__global__ __launch_bounds__(256) void k_root_prep(EngDev E, int g0, int g1) {
    const int g = g0 + blockIdx.x * GPB + (threadIdx.x >> 4), sub = threadIdx.x & (LPG - 1);
    if (g >= g1) return;
    uint8_t fresh = 0;
    if (searches(E, g)) {
        uint8_t f = pool_of(E, g)[E.root[g]].flags;
        if (!(f & (F_EVALUATED | F_TERMINAL))) {
            fresh = 1;
        }
    }
    if (fresh) {
        BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
        int row = alloc_row_grp(E.batch_cnt + 2, sub);
        write_nn_input_grp(E, row, b, sub);
        if (sub == 0) E.row_of_slot[g] = row;
    }
    if (sub == 0) E.root_fresh[g] = fresh;
}

__global__ __launch_bounds__(256) void k_root_init(EngDev E, int g0, int g1) {
    const int g = g0 + blockIdx.x * GPB + (threadIdx.x >> 4), sub = threadIdx.x & (LPG - 1);
    if (g >= g1 || !E.root_fresh[g]) return;
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    create_children_grp(E, g, pool_of(E, g), E.n_nodes[g], E.root[g], b, E.probs + (size_t)E.row_of_slot[g] * E.A, sub);
    if (sub == 0) E.evals[g] += 1;
}

#ifdef AZ_PROBE  // diagnostic build only (make PROBE=1)
__device__ unsigned long long az_step_probe[4096 * 8];
#define PSTAMP(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); pt[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define PSTAMP(i)
#endif

// One lock-step of the search for every slot:
//   BACKUP : nn_evaluation bookkeeping (mcts.py:188-191) + back_propagate (mcts.py:197-223) of the
//            leaf selected in the previous step, whose policy/value the network has just produced
//   SELECT : root noise (mcts.py:235-240) + select_node (mcts.py:127-171) of simulation `sim`,
//            writing the next leaf's canonical board into the network's input batch
template <bool BACKUP, bool SELECT>
__global__ __launch_bounds__(256) void k_step(EngDev E, int sim, int g0, int g1) {
    const int g = g0 + blockIdx.x * GPB + (threadIdx.x >> 4), sub = threadIdx.x & (LPG - 1);
    // this step's leaves are compacted into rows [0, batch_cnt[sim & 1]); the other counter (read by the
    // previous step's network kernels, which have completed) is cleared for the next step
    if (blockIdx.x == 0 && threadIdx.x == 0 && g0 == 0) { E.batch_cnt[(sim + 1) & 1] = 0; if (!SELECT) E.batch_cnt[2] = 0; }
#ifdef AZ_PROBE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PSTAMP(0)
    // Groups beyond the last slot (the slot count need not fill the last wavefront / block) stay in the kernel as
    // inert groups: every __syncthreads() below must be reached by all waves of the block the same number of times, and
    // an early return -- or barriers in a divergent branch -- of SOME lanes of a wavefront breaks exactly that (the
    // wave would arrive twice per barrier: leaf rows were then handed out before all groups had asked for one).
    const bool in_range = g < g1;
    const int gs = in_range ? g : g1 - 1;  // a valid index for the (unused) loads of an inert group
    // ---- every per-slot word is fetched here, in one batch of independent loads: the kernel is bound by
    // ---- the number of DEPENDENT memory round trips (~1 us each), not by bytes
    Node *pool = pool_of(E, gs);
    const int st = (BACKUP && in_range) ? E.leaf_status[gs] : LS_NONE;
    const int leaf = BACKUP ? E.leaf[gs] : 0;
    const BB lb = {BACKUP ? E.leaf_p1[gs] : 0, BACKUP ? E.leaf_m1[gs] : 0, BACKUP ? E.leaf_player[gs] : 1};
    const int row_prev = BACKUP ? E.row_of_slot[gs] : 0;
    const int plen_prev = BACKUP ? E.path_len[gs] : 0;
    const int path_prev = BACKUP ? E.path[(size_t)gs * LPG + sub] : 0;
    const int lwin = BACKUP ? E.leaf_winner[gs] : 0;
    int n_nodes = E.n_nodes[gs];
    int evals = E.evals[gs];
    bool active = in_range && searches(E, gs);
    const int ply = E.ply[gs];
    BB b = {E.root_p1[gs], E.root_m1[gs], E.root_player[gs]};
    int node = E.root[gs];
    Node fwd;
    bool have_root = false;
    PSTAMP(1)
    if (BACKUP && st != LS_NONE) {
        // path nodes and the network row are fetched together (second round trip)
        Node mine;
        const bool on_path = plen_prev <= LPG && sub < plen_prev;
        if (on_path) mine = load_node(pool + path_prev);
        double outcome;
        bool ok = true;
        if (st == LS_EVAL) {
            const float v = E.value[row_prev];
            int k = create_children_grp(E, g, pool, n_nodes, leaf, lb, E.probs + (size_t)row_prev * E.A, sub);
            ok = k > 0;
            n_nodes += ok ? k : 0;
            outcome = (double)lb.player * (double)v;  // base.py:366
            if (ok) evals += 1;
            else { active = false; if (sub == 0) E.active[g] = 0; }
        } else {
            outcome = (double)lwin;
        }
        PSTAMP(2)
        if (ok) {
            bool fast = back_propagate_grp(pool, plen_prev, path_prev, mine, leaf, lb.player, outcome, sub);
            if (fast && SELECT) {  // lane 0 holds the root (path[0]) with its new visit count: forward it
                fwd.N = __shfl(mine.N, 0, LPG);
                fwd.first = __shfl(mine.first, 0, LPG);
                const u32 pk = (u32)__shfl((int)((u32)mine.nch | ((u32)mine.act << 8) | ((u32)mine.flags << 16) | ((u32)(uint8_t)mine.win << 24)), 0, LPG);
                fwd.nch = (uint8_t)(pk & 0xff); fwd.act = (uint8_t)((pk >> 8) & 0xff); fwd.flags = (uint8_t)((pk >> 16) & 0xff);
                fwd.win = (int8_t)(pk >> 24);
                fwd.Q = 0.0; fwd.P = 0.0; fwd.parent = -1;
                // the root itself may be the leaf that just got its children (first visit of an unexpanded root)
                have_root = (leaf != node) || st != LS_EVAL;
            }
        }
        if (sub == 0) E.leaf_status[g] = LS_NONE;
        // the group's own stores (other lanes) must be visible to the loads below
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    PSTAMP(3)
    if (BACKUP && sub == 0 && in_range) E.evals[g] = evals;
    if (!SELECT) return;
    // From here on no group may leave early: the leaf rows are handed out per BLOCK (one global atomic per
    // 16 games instead of one per game on a single hot address) behind two workgroup barriers.
    __shared__ int s_need, s_base;
    if (threadIdx.x == 0) s_need = 0;
    int status = LS_NONE, w = 0, depth = 0, plen = 1;
    int my_path = -1;
    bool bad = false;
    Node cur;
    if (active) {
        if (have_root) cur = fwd; else cur = load_node(pool + node);
        if (E.noise_mode != AZ_NOISE_OFF && E.alpha >= 0.0 && E.eps >= 0.0 && (cur.flags & F_EXPANDED) && !(cur.flags & F_NOISED)) {
            apply_root_noise_grp(E, g, pool, node, cur, b, ply, sim, sub);
            cur.flags |= F_NOISED;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        PSTAMP(4)
        my_path = sub == 0 ? node : -1;  // lane i keeps the i-th node of the root..leaf path
        for (;;) {
            if (cur.flags & F_EXPANDED) {
                Node ch;
                int c = pick_child_grp(E, g, pool, cur, ply, sim, depth++, sub, ch);
                node = c; cur = ch;
                if (sub == plen) my_path = c;
                ++plen;
                az_play_grp(E.gd, b, cur.act, sub);
                if (cur.N == 0) break;  // mcts.py:143-144
                continue;
            }
            if (cur.flags & F_TERMINAL) break;  // mcts.py:146-147
            if (!(cur.flags & F_EVALUATED)) { bad = true; break; }
            cur.flags |= F_EXPANDED;  // mcts.py:151-160 : children become visible now
            if (sub == 0) pool[node].flags = cur.flags;
            Node ch;
            int c = pick_child_grp(E, g, pool, cur, ply, sim, depth, sub, ch);
            node = c; cur = ch;
            if (sub == plen) my_path = c;
            ++plen;
            az_play_grp(E.gd, b, cur.act, sub);
            break;
        }
        PSTAMP(5)
        E.path[(size_t)g * LPG + sub] = my_path;
        if (sub == 0) { E.path_len[g] = plen; if (plen > LPG) atomicMax(E.max_path, plen); }
        if (bad) { if (sub == 0) atomicOr(E.err, ERR_INTERNAL); }
        else if (cur.flags & F_TERMINAL) { status = LS_TERM; w = cur.win; }
        else if (az_status_grp(E.gd, b, &w, sub)) {  // mcts.py:185-186
            status = LS_TERM;
            if (sub == 0) { pool[node].flags = cur.flags | F_TERMINAL; pool[node].win = (int8_t)w; }
        } else {
            status = LS_EVAL;
        }
    }
    __syncthreads();
    int rank = 0;
    if (status == LS_EVAL && sub == 0) rank = atomicAdd(&s_need, 1);  // LDS atomic
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_need > 0 ? atomicAdd(E.batch_cnt + (sim & 1), s_need) : 0;
    __syncthreads();
    if (status == LS_EVAL) {
        const int row = s_base + __shfl(rank, 0, LPG);
        write_nn_input_grp(E, row, b, sub);
        if (sub == 0) E.row_of_slot[g] = row;
    }
    if (sub == 0 && in_range) {
        if (active) { E.leaf[g] = node; E.leaf_p1[g] = b.p1; E.leaf_m1[g] = b.m1; E.leaf_player[g] = (int8_t)b.player; E.leaf_winner[g] = (int8_t)w; }
        E.leaf_status[g] = (int8_t)status;
    }
    PSTAMP(6)
#ifdef AZ_PROBE
    if (BACKUP && SELECT && sim == 50 && (threadIdx.x & 63) == 0 && blockIdx.x < 1024) {  // az_step_probe holds 1024 blocks x 4 waves
        unsigned long long *o = az_step_probe + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 7; ++i) o[i] = pt[i];
        o[7] = (unsigned long long)depth;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// TreeEval.ROLLOUT (BASELINE config 1, the reference's default evaluation opponent): one whole simulation per
// launch -- UCT selection (mcts.py:38-42, 134-135), expansion with a uniformly random child (mcts.py:152-154,
// 163-165), random playout to the end of the game (mcts.py:173-180), back-propagation (mcts.py:197-223).
// np.log is restated by az_det_log (as in the CPU oracle).
// ---------------------------------------------------------------------------------------------
#define AZ_P_ROLLOUT_EXPAND 6
#define AZ_P_PLAYOUT 7

AZ_D int pick_child_uct_grp(const EngDev &E, u32 gid, const Node *pool, const Node &parent, int ply, int sim, int depth, int sub, Node &chosen) {
    const int fc = parent.first, nc = parent.nch;
    const double lg = az_det_log((double)parent.N);
    double key[4];
    int cN[4], cfirst[4];
    u32 cpack[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        key[r] = -__builtin_inf();
        cN[r] = 0; cfirst[r] = -1; cpack[r] = 0;
        int i = r * LPG + sub;
        if (i < nc) {
            Node c = load_node(pool + fc + i);
            key[r] = c.N == 0 ? __builtin_inf() : c.Q + 1.4142135623730951 * sqrt(lg / (double)c.N);
            cN[r] = c.N; cfirst[r] = c.first;
            cpack[r] = (u32)c.nch | ((u32)c.act << 8) | ((u32)c.flags << 16) | ((u32)(uint8_t)c.win << 24);
        }
    }
    double best = grp_max(fmax(fmax(key[0], key[1]), fmax(key[2], key[3])));
    u32 mask[4];
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { mask[r] = grp_ballot((r * LPG + sub) < nc && key[r] == best); cnt += __popc(mask[r]); }
    int k = 0;
    if (E.tie_mode == AZ_TIE_RANDOM && cnt > 1) {
        Philox4 rr = az_philox(E.seed, gid, (u32)ply, (u32)sim + E.sim_base, AZ_P_TIE_SELECT, (u32)depth);
        k = (int)(((u64)rr.x * (u64)cnt) >> 32);
    }
    int rsel = 0, lsel = 0;
    bool found = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int pc = __popc(mask[r]);
        if (!found) { if (k < pc) { rsel = r; lsel = kth_set_bit(mask[r], k); found = true; } else k -= pc; }
    }
    const int vN = rsel == 0 ? cN[0] : (rsel == 1 ? cN[1] : (rsel == 2 ? cN[2] : cN[3]));
    const int vF = rsel == 0 ? cfirst[0] : (rsel == 1 ? cfirst[1] : (rsel == 2 ? cfirst[2] : cfirst[3]));
    const u32 vP = rsel == 0 ? cpack[0] : (rsel == 1 ? cpack[1] : (rsel == 2 ? cpack[2] : cpack[3]));
    chosen.N = __shfl(vN, lsel, LPG);
    chosen.first = __shfl(vF, lsel, LPG);
    const u32 pk = (u32)__shfl((int)vP, lsel, LPG);
    chosen.nch = (uint8_t)(pk & 0xff); chosen.act = (uint8_t)((pk >> 8) & 0xff); chosen.flags = (uint8_t)((pk >> 16) & 0xff);
    chosen.win = (int8_t)(pk >> 24);
    chosen.Q = 0.0; chosen.P = 0.0; chosen.parent = 0;
    return fc + rsel * LPG + lsel;
}

__global__ __launch_bounds__(256) void k_rollout_step(EngDev E, int sim) {
    const int g = blockIdx.x * GPB + (threadIdx.x >> 4), sub = threadIdx.x & (LPG - 1);
    if (g >= E.G || !searches(E, g)) return;
    const GameDesc &gd = E.gd;
    Node *pool = pool_of(E, g);
    const int ply = E.ply[g];
    const u32 gid = E.game_id[g];
    int n_nodes = E.n_nodes[g];
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    int node = E.root[g];
    Node cur = load_node(pool + node);
    int depth = 0, plen = 1;
    int my_path = sub == 0 ? node : -1;
    bool reached_new = false;
    while (cur.flags & F_EXPANDED) {  // mcts.py:132-144
        Node ch;
        int c = pick_child_uct_grp(E, gid, pool, cur, ply, sim, depth++, sub, ch);
        node = c; cur = ch;
        if (sub == plen) my_path = c;
        ++plen;
        az_play_grp(gd, b, cur.act, sub);
        if (cur.N == 0) { reached_new = true; break; }
    }
    int w = 0;
    bool over = az_status_grp(gd, b, &w, sub);
    if (!reached_new && !over) {  // mcts.py:146-171 : expand with every legal move, step into a uniformly random child
        u64 bits = az_legal_bits_grp(gd, b, b.player, sub);
        const bool pass = (gd.game == AZ_OTHELLO && bits == 0);
        const int k = pass ? 1 : __popcll(bits);
        if (n_nodes + k > E.C) { if (sub == 0) { atomicOr(E.err, ERR_NODE_POOL); E.active[g] = 0; } return; }
        const int fc = n_nodes;
        if (pass) { if (sub == 0) store_node(pool + fc, fresh_node(gd.A - 1, node, 0.0, 0)); }
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int bit = r * LPG + sub;
                if ((bits >> bit) & 1ULL) store_node(pool + fc + __popcll(bits & ((1ULL << bit) - 1ULL)), fresh_node(az_bit_to_action(gd, bit), node, 0.0, 0));
            }
        }
        n_nodes += k;
        if (sub == 0) { pool[node].first = fc; pool[node].nch = (uint8_t)k; pool[node].flags = cur.flags | F_EXPANDED; E.n_nodes[g] = n_nodes; }
        Philox4 rr = az_philox(E.seed, gid, (u32)ply, (u32)sim + E.sim_base, AZ_P_ROLLOUT_EXPAND, (u32)depth);
        const int pick = (int)(((u64)rr.x * (u64)k) >> 32);
        int act;
        if (pass) act = gd.A - 1;
        else { u64 m = bits; for (int i = 0; i < pick; ++i) m &= m - 1; act = az_bit_to_action(gd, __ffsll((long long)m) - 1); }
        node = fc + pick;
        if (sub == plen) my_path = node;
        ++plen;
        az_play_grp(gd, b, act, sub);
        over = az_status_grp(gd, b, &w, sub);
    }
    const int player_to_play = b.player;  // mcts.py:243 : side to move at the selected node
    for (u32 step = 0; !over; ++step) {    // mcts.py:173-180 : uniformly random playout
        u64 bits = az_legal_bits_grp(gd, b, b.player, sub);
        int act;
        if (gd.game == AZ_OTHELLO && bits == 0) act = gd.A - 1;
        else {
            const int k = __popcll(bits);
            Philox4 rr = az_philox(E.seed, gid, (u32)ply, (u32)sim + E.sim_base, AZ_P_PLAYOUT, step);
            const int pick = (int)(((u64)rr.x * (u64)k) >> 32);
            u64 m = bits;
            for (int i = 0; i < pick; ++i) m &= m - 1;
            act = az_bit_to_action(gd, __ffsll((long long)m) - 1);
        }
        az_play_grp(gd, b, act, sub);
        over = az_status_grp(gd, b, &w, sub);
    }
    // back_propagate along the recorded path
    const double outcome = (double)w;
    double reward;
    if (fabs(outcome) < 1e-4) reward = 0.0;
    else reward = ((double)player_to_play * outcome > 0.0) ? -fabs(outcome) : fabs(outcome);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (plen > LPG && sub == 0) atomicMax(E.max_path, plen);
    if (plen <= LPG) {
        if (sub < plen) {
            Node n = load_node(pool + my_path);
            const int up = plen - 1 - sub;
            const double r = (reward == 0.0) ? 0.0 : ((up & 1) ? -reward : reward);
            pool[my_path].Q = ((double)n.N * n.Q + r) / (double)(n.N + 1);
            pool[my_path].N = n.N + 1;
        }
    } else {
        int nd = node;
        while (nd >= 0) {
            Node n = load_node(pool + nd);
            if (sub == 0) { pool[nd].Q = ((double)n.N * n.Q + reward) / (double)(n.N + 1); pool[nd].N = n.N + 1; }
            nd = n.parent;
            reward = (reward == 0.0) ? 0.0 : -reward;
        }
    }
}

AZ_D double linear_temp(int step, int tmax, int tmin) {  // schedulers.py:33-40
    if (step <= tmax) return 1.0;
    if (step >= tmin) return 0.0;
    return 1.0 - (double)(step - tmax) / (double)(tmin - tmax);
}

// get_action_probs (mcts.py:95-116) + move choice (players.py:184-189) + Sample (trainer.py:244-250)
// + play_move / change_root (trainer.py:253-256) + end-of-game bookkeeping (trainer.py:262-268).
// Once per ply: one thread per slot.
__global__ void k_move(EngDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G || !E.active[g]) return;
    const GameDesc &gd = E.gd;
    Node *pool = pool_of(E, g);
    int ply = E.ply[g], root = E.root[g];
    u32 gid = E.game_id[g];
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    int fc = pool[root].first, nc = pool[root].nch;
    if (nc == 0 || !(pool[root].flags & F_EXPANDED)) { atomicOr(E.err, ERR_INTERNAL); E.active[g] = 0; return; }
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
            int n = pool[fc + i].N;
            if (n > best) { best = n; cnt = 1; first = i; } else if (n == best) cnt++;
        }
        int pick = first;
        if (E.tie_mode == AZ_TIE_RANDOM && cnt > 1) {
            Philox4 r = az_philox(E.seed, gid, (u32)ply, 0xFFFFu, AZ_P_TIE_MOVE, 0);
            int k = (int)(((u64)r.x * (u64)cnt) >> 32);
            for (int i = 0; i < nc; ++i)
                if (pool[fc + i].N == best) { if (k == 0) { pick = i; break; } --k; }
        }
        chosen = fc + pick;
        if (si >= 0) pi[pool[chosen].act] = 1.0f;
    } else {
        const double inv_temp = 1.0 / temp;
        double sum = 0.0;
        for (int i = 0; i < nc; ++i) {
            double n = (double)pool[fc + i].N;
            sum += (temp == 1.0) ? n : az_det_pow(n, inv_temp);
        }
        double u = 2.0, cum = 0.0;
        if (nc > 1) {
            Philox4 r = az_philox(E.seed, gid, (u32)ply, 0xFFFFu, AZ_P_MOVE_SAMPLE, 0);
            u = az_u53(r.x, r.y);
        }
        int last = 0; bool found = false;
        for (int i = 0; i < nc; ++i) {
            double n = (double)pool[fc + i].N;
            double p = ((temp == 1.0) ? n : az_det_pow(n, inv_temp)) / sum;
            if (si >= 0) pi[pool[fc + i].act] = (float)p;
            if (p > 0.0) last = i;
            cum += p;
            if (!found && u < cum) { chosen = fc + i; found = true; }
        }
        if (!found) chosen = fc + (nc == 1 ? 0 : last);
    }
    int action = pool[chosen].act;
    if (si >= 0) {
        for (int i = 0; i < nc; ++i) vis[pool[fc + i].act] = pool[fc + i].N;
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
    pool[chosen].parent = -1;  // mcts.py:121-123
    E.ply[g] = ply + 1;
    atomicAdd(&E.ctr[CTR_PLIES], 1ULL);
    atomicAdd(&E.ctr[CTR_NET_EVALS], (unsigned long long)E.evals[g]);
    E.evals[g] = 0;
    atomicMax(E.max_nodes, E.n_nodes[g]);
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

// change_root (mcts.py:118-125) as a compaction: the subtree under the new root (E.root[g], set by k_move) is
// copied breadth-first into the slot's other pool and becomes node 0.  While a node waits in the queue its
// `first` field still holds the OLD index of its children; when its level is processed the children are copied
// to a freshly bump-allocated block and `first` is rewritten.  16 lanes take 16 queue nodes per round.
__global__ __launch_bounds__(256) void k_reroot(EngDev E) {
    const int g = blockIdx.x * GPB + (threadIdx.x >> 4), sub = threadIdx.x & (LPG - 1);
    if (g >= E.G || !E.active[g]) return;
    const int sel = E.pool_sel[g];
    const Node *src = E.nodes + ((size_t)g * 2 + sel) * E.C;
    Node *dst = E.nodes + ((size_t)g * 2 + (sel ^ 1)) * E.C;
    if (sub == 0) {
        Node r = load_node(src + E.root[g]);
        r.parent = -1;
        store_node(dst, r);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    int lo = 0, hi = 1, next = 1;
    bool overflow = false;
    while (lo < hi && !overflow) {
        for (int base = lo; base < hi; base += LPG) {
            const int i = base + sub;
            int cnt = 0, oldfc = 0;
            if (i < hi) { Node nd = load_node(dst + i); cnt = nd.nch; oldfc = nd.first; }
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < LPG; d <<= 1) { int y = __shfl_up(incl, d, LPG); if (sub >= d) incl += y; }
            const int total = __shfl(incl, LPG - 1, LPG);
            const int newfc = next + incl - cnt;
            if (next + total > E.C) { overflow = true; break; }
            if (cnt > 0) {
                // four children per trip: their eight 16-byte loads are in flight together (a load -> store -> load chain
                // paid the memory latency once per child: the kernel took ~0.4 ms per ply)
                for (int j = 0; j < cnt; j += 4) {
                    Node c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[u] = load_node(src + oldfc + (j + u < cnt ? j + u : cnt - 1));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j + u < cnt) { c[u].parent = i; store_node(dst + newfc + j + u, c[u]); }
                }
                dst[i].first = newfc;
            }
            next += total;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        lo = hi; hi = next;
    }
    if (sub == 0) {
        if (overflow) { atomicOr(E.err, ERR_NODE_POOL); E.active[g] = 0; }
        E.root[g] = 0; E.n_nodes[g] = next; E.pool_sel[g] = (uint8_t)(sel ^ 1);
    }
}

// Board.play_move + MCT.change_root (mcts.py:118-125) for a move chosen OUTSIDE the engine (arena opponent,
// human): re-root at the child if the root's children are materialised and hold the action, otherwise start
// a fresh root.  status[g] = 0 or AZ_EILLEGAL (reference: ValueError, board untouched).  One thread per slot.
__global__ void k_apply_moves(EngDev E, const int *actions, int n, int *status) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n || g >= E.G) return;
    const int a = actions[g];
    if (a < 0) { status[g] = AZ_OK; return; }  // no move for this slot
    if (!E.active[g]) { status[g] = AZ_ESTATE; return; }
    const GameDesc &gd = E.gd;
    Node *pool = pool_of(E, g);
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    u64 bits = az_legal_bits(gd, b, b.player);
    bool ok = false;
    if (a >= 0 && a < gd.A) {
        if (gd.game == AZ_OTHELLO && a == gd.A - 1) ok = (bits == 0);
        else ok = (bits >> az_action_to_bit(gd, a)) & 1ULL;
    }
    if (!ok) { status[g] = AZ_EILLEGAL; return; }
    const int root = E.root[g];
    int chosen = -1;
    if (pool[root].flags & F_EXPANDED)
        for (int i = 0; i < pool[root].nch; ++i)
            if (pool[pool[root].first + i].act == a) chosen = pool[root].first + i;
    if (chosen < 0) { chosen = 0; store_node(pool, fresh_node(0, -1, 0.0, 0)); }  // mcts.py:124-125
    pool[chosen].parent = -1;
    az_play(gd, b, a);
    E.root_p1[g] = b.p1; E.root_m1[g] = b.m1; E.root_player[g] = (int8_t)b.player;
    E.root[g] = chosen;
    E.ply[g] = E.ply[g] + 1;
    E.leaf_status[g] = LS_NONE;
    int w = 0;
    if (az_status(gd, b, &w)) E.active[g] = 0;  // finished: the board stays readable, the slot is no longer searched
    status[g] = AZ_OK;
}

// MCT.get_action_probs(temp=0) + fair_max (mcts.py:110-112): the most visited root child of every slot this
// engine is to move in; -1 elsewhere.  One thread per slot.
__global__ void k_best_moves(EngDev E, int *actions) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    actions[g] = -1;
    if (!searches(E, g)) return;
    Node *pool = pool_of(E, g);
    const int root = E.root[g], fc = pool[root].first, nc = pool[root].nch;
    if (nc == 0 || !(pool[root].flags & F_EXPANDED)) return;
    int best = -1, cnt = 0, first = 0;
    for (int i = 0; i < nc; ++i) {
        int n = pool[fc + i].N;
        if (n > best) { best = n; cnt = 1; first = i; } else if (n == best) cnt++;
    }
    int pick = first;
    if (E.tie_mode == AZ_TIE_RANDOM && cnt > 1) {
        Philox4 r = az_philox(E.seed, E.game_id[g], (u32)E.ply[g], 0xFFFFu, AZ_P_TIE_MOVE, 0);
        int k = (int)(((u64)r.x * (u64)cnt) >> 32);
        for (int i = 0; i < nc; ++i)
            if (pool[fc + i].N == best) { if (k == 0) { pick = i; break; } --k; }
    }
    actions[g] = pool[fc + pick].act;
}

// TicTacToeBoard.get_score (tictactoe.py:119-126) is +inf when the side to move holds two cells of an alignment whose
// third cell is free, else 0
AZ_D bool ttt_can_win_at_once(const BB &b) {
    const u64 own = b.player > 0 ? b.p1 : b.m1, occ = b.p1 | b.m1;
    const u64 L[8] = {0x7ULL, 0x7ULL << 8, 0x7ULL << 16, 0x010101ULL, 0x010101ULL << 1, 0x010101ULL << 2,
                      (1ULL | (1ULL << 9) | (1ULL << 18)), ((1ULL << 2) | (1ULL << 9) | (1ULL << 16))};
    bool threat = false;
    for (int l = 0; l < 8; ++l) threat |= (__popcll(own & L[l]) == 2 && __popcll(occ & L[l]) == 2);
    return threat;
}

// RandomPlayer / GreedyPlayer (players.py:76-123) for the slots where the OTHER colour is to move:
// kind 0 = uniform legal move, kind 1 = best immediate -get_score() of the position after the move, ties uniform.
__global__ void k_baseline_moves(EngDev E, int kind, u32 seed, int *actions) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    actions[g] = -1;
    if (!E.active[g] || E.side[g] == 0 || E.side[g] == E.root_player[g]) return;
    const GameDesc &gd = E.gd;
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    u64 bits = az_legal_bits(gd, b, b.player);
    if (gd.game == AZ_OTHELLO && bits == 0) { actions[g] = gd.A - 1; return; }  // forced pass
    int n = __popcll(bits);
    if (n == 0) return;
    u64 cand = bits;
    if (kind == 1) {  // greedy: keep the moves with the best score
        int best = -1000000;
        cand = 0;
        for (u64 m = bits; m; m &= m - 1) {
            int bit = __ffsll((long long)m) - 1;
            BB c = b;
            az_play(gd, c, az_bit_to_action(gd, bit));
            int sc;
            if (gd.game == AZ_TICTACTOE) {  // -get_score(): -inf when the opponent can complete a line at once (tictactoe.py:119-126)
                sc = ttt_can_win_at_once(c) ? -1 : 0;
            } else {
                sc = -(c.player * (__popcll(c.p1) - __popcll(c.m1)));  // -sum(player*grid) after the move
            }
            if (sc > best) { best = sc; cand = 1ULL << bit; } else if (sc == best) cand |= 1ULL << bit;
        }
        n = __popcll(cand);
    }
    int k = 0;  // greedy under the deterministic tie-break (tests against the reference's patched fair_max): lowest action
    if (kind == 0 || E.tie_mode == AZ_TIE_RANDOM) {
        Philox4 r = az_philox(seed, E.game_id[g], (u32)E.ply[g], 0xFFFEu, AZ_P_TIE_MOVE, (u32)kind);
        k = (int)(((u64)r.x * (u64)n) >> 32);
    }
    for (int i = 0; i < k; ++i) cand &= cand - 1;
    actions[g] = az_bit_to_action(gd, __ffsll((long long)cand) - 1);
}

__global__ void k_root_status(EngDev E, int8_t *players, uint8_t *over, int8_t *winner, int *score) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    BB b = {E.root_p1[g], E.root_m1[g], E.root_player[g]};
    int w = 2;
    bool o = az_status(E.gd, b, &w);
    players[g] = (int8_t)b.player; over[g] = o ? 1 : 0; winner[g] = (int8_t)(o ? w : 2);
    // Board.get_score from the side to move's viewpoint; TicTacToe's +inf is reported as 32767
    score[g] = E.gd.game == AZ_TICTACTOE ? (ttt_can_win_at_once(b) ? 32767 : 0) : b.player * (__popcll(b.p1) - __popcll(b.m1));
}

// closed-form fake network (tests): reads the canonical board back from nn_in
__global__ void k_fakenet(EngDev E, const int *cnt) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G || g >= *cnt) return;
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
    int *h_err;                 // pinned [3] : err, max_nodes, max_path
    long long lockstep_iters;
    u32 sim_base = 0;
    // The engine runs on a stream of its own (graph capture is not allowed on the legacy default stream); every entry
    // point first orders it behind the caller's stream and returns only after its own stream has drained.
    hipStream_t user_stream = nullptr;
    bool search_open = false;  // between az_engine_search_begin and _end
    hipEvent_t ev_in = nullptr;
    // One search = 1 + 5 n_sim kernel launches: captured once per (n_sim, batch cap) as a HIP graph and replayed
    std::map<unsigned long long, hipGraphExec_t> graphs;
    std::map<unsigned long long, int> graph_seen;
    bool graphs_ok = true;
    long long graph_replays = 0;
    int *scr_a = nullptr, *scr_b = nullptr;  // [G] int scratch of the arena entry points (moves in/out, status, scores)
    char *scr_c = nullptr;                   // [3 G] bytes
    int active_bound = 0;  // upper bound on the slots still searching (known per ply): caps the network batch, which picks the kernels
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

// between az_engine_search_begin and _end the search's launches are in flight and its error flags unread: every other entry point of
// the engine (they reorder host-side state -- sim_base, active_bound, the graph cache -- or read device state the search is writing)
// refuses until the search has been ended
#define AZ_NO_OPEN_SEARCH(e, who) AZ_REQUIRE(!(e)->search_open, AZ_ESTATE, who ": a search begun with az_engine_search_begin has not been ended (az_engine_search_end)")

extern "C" int az_engine_create(const az_engine_cfg *cfg, az_net *net, void *stream, az_engine **out) {
    AZ_REQUIRE(cfg && out, AZ_EINVAL, "null argument");
    GameDesc gd;
    AZ_TRY(az_make_game_desc(cfg->game, cfg->H, cfg->W, &gd));
    AZ_REQUIRE(cfg->n_slots > 0 && cfg->n_sim > 0, AZ_EINVAL, "n_slots and n_sim must be positive");
    AZ_REQUIRE(cfg->node_capacity >= 2 * AZ_MAX_ACTIONS, AZ_EINVAL, "node_capacity too small");
    AZ_REQUIRE(cfg->max_plies > 0 && cfg->sample_capacity > 0, AZ_EINVAL, "max_plies / sample_capacity must be positive");
    AZ_REQUIRE(cfg->temp_min_step >= cfg->temp_max_step, AZ_EINVAL,
               "temp_min_step should be greater than temp_max_step for linear scheduler.");  // schedulers.py:29-30
    AZ_REQUIRE(cfg->evaluator != AZ_EVAL_NET || net != nullptr, AZ_ESTATE, "a network is required for AZ_EVAL_NET");
    if (cfg->evaluator == AZ_EVAL_NET)
        AZ_REQUIRE(az_net_action_size(net) == gd.A, AZ_EINVAL, "network action size %d != game action size %d",
                   az_net_action_size(net), gd.A);
    az_engine *e = new az_engine();
    e->h_ctr = nullptr; e->h_err = nullptr;
    e->cfg = *cfg; e->net = net; e->user_stream = (hipStream_t)stream; e->stream = nullptr; e->lockstep_iters = 0;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming) != hipSuccess) {
        az_engine_destroy(e);
        az_set_error("could not create the engine's stream");
        return AZ_EHIP;
    }
    { const char *g = getenv("AZ_ENGINE_GRAPHS"); if (g && atoi(g) == 0) e->graphs_ok = false; }
    EngDev &d = e->d;
    d.gd = gd; d.G = cfg->n_slots; d.C = cfg->node_capacity; d.A = gd.A; d.max_plies = cfg->max_plies;
    d.alpha = cfg->dirichlet_alpha; d.eps = cfg->dirichlet_epsilon; d.tie_mode = cfg->tie_mode;
    d.rollout = cfg->evaluator == AZ_EVAL_ROLLOUT ? 1 : 0;
    d.noise_mode = cfg->noise_mode; d.tmax = cfg->temp_max_step; d.tmin = cfg->temp_min_step; d.seed = cfg->seed;
    d.sample_cap = cfg->sample_capacity;
    size_t G = d.G, NC = G * (size_t)d.C, S = (size_t)cfg->sample_capacity;
    int rc = AZ_OK;
#define A_(p, n) if (rc == AZ_OK) rc = dev_alloc(e, &d.p, (n))
    A_(root_p1, G); A_(root_m1, G); A_(root_player, G); A_(root, G); A_(n_nodes, G); A_(ply, G); A_(game_id, G);
    A_(active, G); A_(root_fresh, G); A_(side, G); A_(leaf, G); A_(leaf_p1, G); A_(leaf_m1, G); A_(leaf_player, G);
    A_(leaf_status, G); A_(leaf_winner, G); A_(path, G * LPG); A_(path_len, G);
    A_(nodes, 2 * NC); A_(pool_sel, G);
    A_(nn_in, G * gd.cells); A_(probs, G * gd.A); A_(value, G); A_(row_of_slot, G); A_(evals, G); A_(batch_cnt, 4);
    A_(samp_idx, G * (size_t)d.max_plies);
    A_(o_state, S * gd.cells); A_(o_pi, S * gd.A); A_(o_z, S); A_(o_meta, S * 4); A_(o_visits, S * gd.A);
    A_(ctr, CTR_COUNT); A_(err, 1); A_(max_nodes, 1); A_(max_path, 1);
#undef A_
    if (rc == AZ_OK) rc = dev_alloc(e, &e->scr_a, G);
    if (rc == AZ_OK) rc = dev_alloc(e, &e->scr_b, G);
    if (rc == AZ_OK) rc = dev_alloc(e, &e->scr_c, 3 * G);
    if (rc == AZ_OK && hipHostMalloc((void **)&e->h_ctr, sizeof(unsigned long long) * CTR_COUNT) != hipSuccess) rc = AZ_EHIP;
    if (rc == AZ_OK && hipHostMalloc((void **)&e->h_err, sizeof(int) * 3) != hipSuccess) rc = AZ_EHIP;
    if (rc != AZ_OK) { az_engine_destroy(e); return rc; }
    if (hipStreamSynchronize(e->stream) != hipSuccess) { az_engine_destroy(e); az_set_error("stream sync failed"); return AZ_EHIP; }
    *out = e;
    return AZ_OK;
}

static int fetch_counters(az_engine *e);
static int check_err(az_engine *e);

extern "C" void az_engine_destroy(az_engine *e) {
    if (!e) return;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->search_open) {  // destroyed with a search still open: its errors would vanish with the engine -- say so
        e->search_open = false;
        if (fetch_counters(e) == AZ_OK && check_err(e) != AZ_OK)
            fprintf(stderr, "az_engine_destroy: the search that was never ended had failed: %s\n", az_last_error());
    }
    for (auto &kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->h_ctr) (void)hipHostFree(e->h_ctr);
    if (e->h_err) (void)hipHostFree(e->h_err);
    delete e;
}

// network over the compacted leaf rows [0, *cnt)
static int forward(az_engine *e, const int *cnt, int cap) {
    EngDev &d = e->d;
    if (e->cfg.evaluator == AZ_EVAL_FAKE) {
        hipLaunchKernelGGL(k_fakenet, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, cnt);
        return AZ_OK;
    }
    return az_net_forward_dyn(e->net, d.nn_in, cnt, cap, d.probs, d.value, e->stream);
}

// MCT.search for every active slot: one root-prior pass (mcts.py:231-233; empty unless a slot holds a
// fresh root), then n_sim lock-steps of [backup+select -> network].
// the raw launch sequence of one search; `cap` bounds the network batch (leaf rows are compact: count <= searching slots)
static int enqueue_search(az_engine *e, int n_sim, int cap) {
    EngDev &d = e->d;
    dim3 gg((unsigned)((d.G + GPB - 1) / GPB)), gb(256);
    if (d.rollout) {  // no network: one launch per simulation
        for (int s = 0; s < n_sim; ++s) hipLaunchKernelGGL(k_rollout_step, gg, gb, 0, e->stream, d, s);
        AZ_HIP(hipGetLastError());
        return AZ_OK;
    }
    hipLaunchKernelGGL(k_root_prep, gg, gb, 0, e->stream, d, 0, d.G);
    AZ_TRY(forward(e, d.batch_cnt + 2, cap));
    hipLaunchKernelGGL(k_root_init, gg, gb, 0, e->stream, d, 0, d.G);
    for (int s = 0; s < n_sim; ++s) {
        if (s == 0) hipLaunchKernelGGL((k_step<false, true>), gg, gb, 0, e->stream, d, s, 0, d.G);
        else hipLaunchKernelGGL((k_step<true, true>), gg, gb, 0, e->stream, d, s, 0, d.G);
        AZ_TRY(forward(e, d.batch_cnt + (s & 1), cap));
    }
    hipLaunchKernelGGL((k_step<true, false>), gg, gb, 0, e->stream, d, n_sim, 0, d.G);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

static int do_search(az_engine *e, int n_sim) {
    EngDev &d = e->d;
    d.sim_base = e->sim_base;
    e->sim_base += (u32)n_sim;
    e->lockstep_iters += d.rollout ? n_sim : n_sim + 1;
    int cap = e->active_bound > 0 && e->active_bound < d.G ? e->active_bound : d.G;
    // graph replay needs launch parameters that do not change from search to search: the Philox counter base must be 0
    // (one search per root, as in self-play and the arena), no per-launch event recording, and a quantised batch cap
    const bool graphable = e->graphs_ok && d.sim_base == 0 && !(e->net && az_net_profiling(e->net));
    // the quantised cap of the graph path also when the search runs as plain launches under az_net_profile: the profiled step then
    // launches the kernels the timed (graph-replayed) steps launch (an exact cap of 4095 -- one game of 4096 over, as happens from
    // ply ~11 on: Othello has early wipe-outs -- would hand the trunk to the one-board-per-wave kernel for the rest of the wave)
    const int cap_q = (d.G >= 4096 && cap < 4096) ? (cap + 511) / 512 * 512 : d.G;  // below 4096 rows the network picks other kernels
    if (!graphable) return enqueue_search(e, n_sim, (e->net && az_net_profiling(e->net) && d.sim_base == 0) ? cap_q : cap);
    cap = cap_q;
    const unsigned long long key = ((unsigned long long)n_sim << 32) | (unsigned)cap;
    auto it = e->graphs.find(key);
    if (it != e->graphs.end()) {
        AZ_HIP(hipGraphLaunch(it->second, e->stream));
        e->graph_replays++;
        return AZ_OK;
    }
    if (e->graph_seen[key]++ == 0) return enqueue_search(e, n_sim, cap);  // first time: plain launches (kernel attributes get set)
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        e->graphs_ok = false;
        return enqueue_search(e, n_sim, cap);
    }
    const int rc = enqueue_search(e, n_sim, cap);  // recorded, not executed
    const hipError_t er = hipStreamEndCapture(e->stream, &g);
    if (rc != AZ_OK || er != hipSuccess || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        e->graphs_ok = false;
        return enqueue_search(e, n_sim, cap);
    }
    (void)hipGraphDestroy(g);
    e->graphs[key] = ex;
    AZ_HIP(hipGraphLaunch(ex, e->stream));
    e->graph_replays++;
    return AZ_OK;
}

// orders the engine's stream behind whatever the caller has queued on the stream it handed to az_engine_create
static int enter(az_engine *e) {
    AZ_HIP(hipEventRecord(e->ev_in, e->user_stream));
    AZ_HIP(hipStreamWaitEvent(e->stream, e->ev_in, 0));
    return AZ_OK;
}

static int fetch_counters(az_engine *e) {
    AZ_HIP(hipMemcpyAsync(e->h_ctr, e->d.ctr, sizeof(unsigned long long) * CTR_COUNT, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(&e->h_err[0], e->d.err, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(&e->h_err[1], e->d.max_nodes, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(&e->h_err[2], e->d.max_path, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

static int check_err(az_engine *e) {
    int f = e->h_err[0];
    if (f & ERR_NODE_POOL) { az_set_error("tree node pool exhausted (node_capacity=%d)", e->cfg.node_capacity); return AZ_ECAPACITY; }
    if (f & ERR_SAMPLE_CAP) { az_set_error("sample buffer exhausted (sample_capacity=%lld)", (long long)e->cfg.sample_capacity); return AZ_ECAPACITY; }
    if (f & ERR_PLY_CAP) { az_set_error("game longer than max_plies=%d", e->cfg.max_plies); return AZ_ECAPACITY; }
    if (f & ERR_RNG) { az_set_error("Dirichlet noise: the Gamma rejection sampler did not accept within 64 attempts"); return AZ_ESTATE; }
    if (f & ERR_INTERNAL) { az_set_error("internal tree invariant violated (no selectable child: NaN priors or values?)"); return AZ_ESTATE; }
    return AZ_OK;
}

extern "C" int az_engine_run(az_engine *e, uint32_t first_game_id, int32_t n_games) {
    AZ_REQUIRE(e && n_games > 0, AZ_EINVAL, "bad arguments");
    AZ_NO_OPEN_SEARCH(e, "az_engine_run");
    AZ_TRY(enter(e));
    EngDev &d = e->d;
    e->lockstep_iters = 0;
    hipLaunchKernelGGL(k_reset_all, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, (u32)first_game_id, (int)n_games);
    long long max_iters = ((long long)n_games / d.G + 2) * (long long)d.max_plies + 8;
    e->active_bound = n_games < d.G ? n_games : d.G;
    for (long long it = 0; it < max_iters; ++it) {
        e->sim_base = 0;
        AZ_TRY(do_search(e, e->cfg.n_sim));
        hipLaunchKernelGGL(k_move, grid_for(d.G, TB), dim3(TB), 0, e->stream, d);
        hipLaunchKernelGGL(k_reroot, dim3((unsigned)((d.G + GPB - 1) / GPB)), dim3(256), 0, e->stream, d);
        AZ_TRY(fetch_counters(e));
        AZ_TRY(check_err(e));
        if (e->h_ctr[CTR_GAMES_DONE] >= (unsigned long long)n_games) return AZ_OK;
        {
            unsigned long long started = e->h_ctr[CTR_NEXT_GAME] < e->h_ctr[CTR_TOTAL_GAMES] ? e->h_ctr[CTR_NEXT_GAME] : e->h_ctr[CTR_TOTAL_GAMES];
            e->active_bound = (int)(started - e->h_ctr[CTR_GAMES_DONE]);
        }
    }
    az_set_error("self-play did not finish within %lld plies", max_iters);
    return AZ_ESTATE;
}

extern "C" int az_engine_get_stats(az_engine *e, az_engine_stats *out) {
    AZ_REQUIRE(e && out, AZ_EINVAL, "null argument");
    AZ_TRY(enter(e));
    AZ_TRY(fetch_counters(e));
    out->games_done = (int64_t)e->h_ctr[CTR_GAMES_DONE];
    long long s = (long long)e->h_ctr[CTR_SAMPLES];
    out->samples = s < e->cfg.sample_capacity ? s : e->cfg.sample_capacity;
    out->net_evals = (int64_t)e->h_ctr[CTR_NET_EVALS];
    out->plies = (int64_t)e->h_ctr[CTR_PLIES];
    out->lockstep_iters = e->lockstep_iters;
    out->max_nodes_used = e->h_err[1];
    out->error_flags = e->h_err[0];
    out->graph_replays = e->graph_replays;
    out->max_path_len = e->h_err[2];
    out->reserved = 0;
    return AZ_OK;
}

extern "C" int az_engine_samples(az_engine *e, int64_t *n_samples, const int8_t **d_states, const float **d_pis,
                                 const int8_t **d_zs, const int32_t **d_meta, const int32_t **d_visits) {
    AZ_REQUIRE(e && n_samples, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_samples");
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
    AZ_NO_OPEN_SEARCH(e, "az_engine_set_roots");
    e->sim_base = 0;
    EngDev &d = e->d;
    AZ_REQUIRE(n_roots > 0 && n_roots <= d.G, AZ_EINVAL, "n_roots must be in [1, n_slots]");
    AZ_TRY(enter(e));
    e->active_bound = n_roots;
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
    AZ_NO_OPEN_SEARCH(e, "az_engine_search");
    AZ_TRY(enter(e));
    AZ_TRY(do_search(e, n_sim));
    AZ_TRY(fetch_counters(e));
    return check_err(e);
}

// az_engine_search in two halves: _begin queues the search on the engine's stream and returns, _end waits for it and reports its
// errors.  Between the two the caller may start the search of ANOTHER engine (the arena's two players think at the same time on
// disjoint slots: two latency-bound launch chains side by side); no other entry point of THIS engine may be called in between.
extern "C" int az_engine_search_begin(az_engine *e, int32_t n_sim) {
    AZ_REQUIRE(e && n_sim > 0, AZ_EINVAL, "bad arguments");
    AZ_REQUIRE(!e->search_open, AZ_ESTATE, "az_engine_search_begin: the previous search has not been ended");
    AZ_TRY(enter(e));
    AZ_TRY(do_search(e, n_sim));
    e->search_open = true;
    return AZ_OK;
}

extern "C" int az_engine_search_end(az_engine *e) {
    AZ_REQUIRE(e, AZ_EINVAL, "null argument");
    AZ_REQUIRE(e->search_open, AZ_ESTATE, "az_engine_search_end without az_engine_search_begin");
    e->search_open = false;
    AZ_TRY(fetch_counters(e));
    return check_err(e);
}

// Two engines whose searches are to overlap (az_engine_search_begin on both) need streams on DIFFERENT hardware queues.  The HIP
// runtime deals its few hardware queues (four by default) out to streams of one priority in turn, so two streams of a process that
// has made others (torch's, a trainer's) may share a queue and then run strictly one after the other (measured: an arena inside
// the trainer's process gained nothing from the overlap).  Queues of different priorities come from different pools: `b` gets a
// new stream of the highest priority.  Call before b's first search.
extern "C" int az_engine_pair(az_engine *a, az_engine *b) {
    AZ_REQUIRE(a && b && a != b, AZ_EINVAL, "two different engines are needed");
    AZ_REQUIRE(!a->search_open && !b->search_open, AZ_ESTATE, "az_engine_pair: a search is open");
    AZ_REQUIRE(b->graphs.empty(), AZ_ESTATE, "az_engine_pair must come before the second engine's searches");
    int least = 0, greatest = 0;  // numerically lower = higher priority
    AZ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (least == greatest) return AZ_OK;  // no priorities on this device: nothing to do
    hipStream_t s = nullptr;
    AZ_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    AZ_HIP(hipStreamSynchronize(b->stream));
    (void)hipStreamDestroy(b->stream);
    b->stream = s;
    return AZ_OK;
}

extern "C" int az_engine_advance(az_engine *e) {
    AZ_REQUIRE(e, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_advance");
    e->sim_base = 0;
    EngDev &d = e->d;
    // games that end here must not be refilled: cap the queue at what has been started
    hipLaunchKernelGGL(k_move, grid_for(d.G, TB), dim3(TB), 0, e->stream, d);
    hipLaunchKernelGGL(k_reroot, dim3((unsigned)((d.G + GPB - 1) / GPB)), dim3(256), 0, e->stream, d);
    AZ_TRY(fetch_counters(e));
    return check_err(e);
}

extern "C" int az_engine_root_children(az_engine *e, int32_t slot, int32_t *h_actions, int32_t *h_N, double *h_Q,
                                       double *h_P, int32_t *count, int32_t *root_N) {
    AZ_REQUIRE(e && count, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_root_children");
    EngDev &d = e->d;
    AZ_REQUIRE(slot >= 0 && slot < d.G, AZ_EINVAL, "slot out of range");
    AZ_HIP(hipStreamSynchronize(e->stream));
    int root = 0;
    uint8_t sel = 0;
    AZ_HIP(hipMemcpy(&sel, d.pool_sel + slot, 1, hipMemcpyDeviceToHost));
    size_t base = ((size_t)slot * 2 + sel) * d.C;
    AZ_HIP(hipMemcpy(&root, d.root + slot, sizeof(int), hipMemcpyDeviceToHost));
    Node rn;
    AZ_HIP(hipMemcpy(&rn, d.nodes + base + root, sizeof(Node), hipMemcpyDeviceToHost));
    if (root_N) *root_N = rn.N;
    int nc = (rn.flags & F_EXPANDED) ? rn.nch : 0;  // children not materialised yet in the reference's tree
    *count = nc;
    if (nc == 0) return AZ_OK;
    std::vector<Node> ch(nc);
    AZ_HIP(hipMemcpy(ch.data(), d.nodes + base + rn.first, sizeof(Node) * nc, hipMemcpyDeviceToHost));
    for (int i = 0; i < nc; ++i) {
        if (h_actions) h_actions[i] = ch[i].act;
        if (h_N) h_N[i] = ch[i].N;
        if (h_Q) h_Q[i] = ch[i].Q;
        if (h_P) h_P[i] = ch[i].P;
    }
    return AZ_OK;
}

// Nodes the slot's live pool holds (bump allocator top): what a caller that keeps searching one root (MCT.search called again
// and again without a move, mcts.py:226-269) checks before the next search.
extern "C" int az_engine_nodes_used(az_engine *e, int32_t slot, int32_t *n_nodes) {
    AZ_REQUIRE(e && n_nodes, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_nodes_used");
    AZ_REQUIRE(slot >= 0 && slot < e->d.G, AZ_EINVAL, "slot out of range");
    AZ_HIP(hipStreamSynchronize(e->stream));
    AZ_HIP(hipMemcpy(n_nodes, e->d.n_nodes + slot, sizeof(int), hipMemcpyDeviceToHost));
    return AZ_OK;
}

// Re-allocates the tree pools with `node_capacity` nodes each and moves every slot's trees over (node links are pool-relative).
// The reference's tree grows without bound; here the pools are sized per search and grown on demand by the single-game MCT.
extern "C" int az_engine_grow_pools(az_engine *e, int32_t node_capacity) {
    AZ_REQUIRE(e, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_grow_pools");
    EngDev &d = e->d;
    AZ_REQUIRE(node_capacity >= d.C, AZ_EINVAL, "pools can only grow (%d < %d)", node_capacity, d.C);
    if (node_capacity == d.C) return AZ_OK;
    AZ_TRY(enter(e));
    const size_t rows = 2 * (size_t)d.G;
    Node *fresh = nullptr;
    AZ_HIP(hipMalloc((void **)&fresh, rows * (size_t)node_capacity * sizeof(Node)));
    hipError_t er = hipMemsetAsync(fresh, 0, rows * (size_t)node_capacity * sizeof(Node), e->stream);
    if (er == hipSuccess)
        er = hipMemcpy2DAsync(fresh, (size_t)node_capacity * sizeof(Node), d.nodes, (size_t)d.C * sizeof(Node), (size_t)d.C * sizeof(Node), rows,
                              hipMemcpyDeviceToDevice, e->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    if (er != hipSuccess) { (void)hipFree(fresh); az_set_error("growing the node pools failed: %s", hipGetErrorString(er)); return AZ_EHIP; }
    for (auto &p : e->allocs) if (p == (void *)d.nodes) p = (void *)fresh;
    (void)hipFree(d.nodes);
    d.nodes = fresh;
    d.C = node_capacity;
    e->cfg.node_capacity = node_capacity;
    // captured searches hold the old pool pointer and capacity by value
    for (auto &kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    e->graphs.clear();
    e->graph_seen.clear();
    return AZ_OK;
}

extern "C" int az_engine_play(az_engine *e, const int32_t *h_actions, int32_t n, int32_t *h_status) {
    AZ_REQUIRE(e && h_actions && h_status, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_play");
    e->sim_base = 0;
    EngDev &d = e->d;
    AZ_REQUIRE(n > 0 && n <= d.G, AZ_EINVAL, "n must be in [1, n_slots]");
    int *d_act = e->scr_a, *d_st = e->scr_b;
    AZ_HIP(hipMemcpyAsync(d_act, h_actions, sizeof(int) * n, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_apply_moves, grid_for(n, TB), dim3(TB), 0, e->stream, d, d_act, (int)n, d_st);
    hipLaunchKernelGGL(k_reroot, dim3((unsigned)((d.G + GPB - 1) / GPB)), dim3(256), 0, e->stream, d);
    AZ_HIP(hipMemcpyAsync(h_status, d_st, sizeof(int) * n, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n; ++i)
        if (h_status[i] == AZ_EILLEGAL) { az_set_error("Illegal move %d for slot %d", h_actions[i], i); return AZ_EILLEGAL; }
    AZ_TRY(fetch_counters(e));
    return check_err(e);
}

#ifdef AZ_PROBE
extern "C" int az_debug_read_step_probe(unsigned long long *h_out, int n_words) {
    AZ_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(az_step_probe), sizeof(unsigned long long) * (size_t)n_words));
    return AZ_OK;
}
#endif

// ---- arena support ---------------------------------------------------------------
extern "C" int az_engine_set_sides(az_engine *e, const int8_t *h_sides, int32_t n) {
    AZ_REQUIRE(e && h_sides && n > 0 && n <= e->d.G, AZ_EINVAL, "bad arguments");
    AZ_NO_OPEN_SEARCH(e, "az_engine_set_sides");
    AZ_HIP(hipMemcpyAsync(e->d.side, h_sides, (size_t)n, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

static int moves_out(az_engine *e, int32_t *h_actions, int which, int kind, uint32_t seed) {
    EngDev &d = e->d;
    int *d_act = e->scr_a;
    if (which == 0) hipLaunchKernelGGL(k_best_moves, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, d_act);
    else hipLaunchKernelGGL(k_baseline_moves, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, kind, (u32)seed, d_act);
    AZ_HIP(hipMemcpyAsync(h_actions, d_act, sizeof(int) * d.G, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

extern "C" int az_engine_best_moves(az_engine *e, int32_t *h_actions) {
    AZ_REQUIRE(e && h_actions, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_best_moves");
    return moves_out(e, h_actions, 0, 0, 0);
}

extern "C" int az_engine_baseline_moves(az_engine *e, int32_t kind, uint32_t seed, int32_t *h_actions) {
    AZ_REQUIRE(e && h_actions && (kind == 0 || kind == 1), AZ_EINVAL, "bad arguments (kind: 0 random, 1 greedy)");
    AZ_NO_OPEN_SEARCH(e, "az_engine_baseline_moves");
    return moves_out(e, h_actions, 1, kind, seed);
}

extern "C" int az_engine_root_status(az_engine *e, int8_t *h_players, uint8_t *h_over, int8_t *h_winner, int32_t *h_score) {
    AZ_REQUIRE(e && h_players && h_over && h_winner && h_score, AZ_EINVAL, "null argument");
    AZ_NO_OPEN_SEARCH(e, "az_engine_root_status");
    EngDev &d = e->d;
    char *buf = e->scr_c;
    size_t G = d.G;
    int8_t *pl = (int8_t *)buf; uint8_t *ov = (uint8_t *)(buf + G); int8_t *wi = (int8_t *)(buf + 2 * G);
    int *sc = e->scr_a;
    hipLaunchKernelGGL(k_root_status, grid_for(d.G, TB), dim3(TB), 0, e->stream, d, pl, ov, wi, sc);
    AZ_HIP(hipMemcpyAsync(h_players, pl, G, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(h_over, ov, G, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(h_winner, wi, G, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipMemcpyAsync(h_score, sc, G * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}
