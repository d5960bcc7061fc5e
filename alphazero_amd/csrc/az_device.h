// az_device.h -- device-side building blocks of the MI355X self-play engine (gfx950 only).
//
//  * deterministic math + Philox4x32-10: defined operation by operation so that the engine's
//    random draws and the softmax/tanh heads are bit-reproducible (no fast-math, no contraction:
//    the library is built with -ffp-contract=off and every fused multiply-add is an explicit fmaf).
//  * bitboard rules: boards live as two u64 per game (cell (r,c) -> bit r*8+c, stride 8 for every
//    game so one shift table serves Othello 4/6/8, Connect4 up to 8x8 and TicTacToe).
//
// Reference behaviour restated here: alphazero/games/othello.py:141-229, connect4.py:143-258,
// tictactoe.py:111-184 (rules); utils.py:28-34 (fair_max); mcts.py:235-240 (Dirichlet noise).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define AZ_OTHELLO 0
#define AZ_CONNECT4 1
#define AZ_TICTACTOE 2
#define AZ_MAX_ACTIONS 65

#define AZ_TIE_LOWEST 0
#define AZ_TIE_RANDOM 1
#define AZ_NOISE_OFF 0
#define AZ_NOISE_PHILOX 1
#define AZ_NOISE_HASH 2

// Philox counter word 2 ("purpose")
#define AZ_P_TIE_SELECT 1
#define AZ_P_NOISE_NORMAL 2
#define AZ_P_NOISE_BOOST 3
#define AZ_P_MOVE_SAMPLE 4
#define AZ_P_TIE_MOVE 5

#define AZ_HD __host__ __device__ __forceinline__
#define AZ_D __device__ __forceinline__

struct GameDesc {
    int game, H, W, A, cells;
    u64 valid;  // bits of the H x W cells
};

// ---------------------------------------------------------------------------------------------
// deterministic math
// ---------------------------------------------------------------------------------------------
AZ_D float az_det_expf(float x) {
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(k, -0.693359375f, x);
    r = fmaf(k, 2.12194440e-4f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int ki = (int)k;
    return p * __uint_as_float((u32)(ki + 127) << 23);
}

AZ_D float az_det_tanhf(float x) {
    float ax = fabsf(x);
    float t;
    if (ax > 10.0f) {
        t = 1.0f;
    } else {
        float e = az_det_expf(-2.0f * ax);
        t = (1.0f - e) / (1.0f + e);
    }
    return x < 0.0f ? -t : t;
}

AZ_D double az_det_log(double x) {
    if (!(x > 0.0)) return -__builtin_inf();
    u64 u = (u64)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
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

AZ_D double az_det_exp(double x) {
    if (!(x > -700.0)) return 0.0;
    if (x > 700.0) x = 700.0;
    double k = floor(x * 1.4426950408889634 + 0.5);
    double r = (x - k * 0.693147180369123816490) - k * 1.90821492927058770002e-10;
    double p = 1.0 / 87178291200.0;
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
    return p * __longlong_as_double((long long)((u64)(ki + 1023) << 52));
}

// N ** (1 / temp) of get_action_probs (mcts.py:114-116) for temperatures other than 0 and 1: exp(log(n) * inv_temp) on the fixed
// polynomials above, so that the GPU and the CPU oracle (orc_det_pow) produce the same bits; within 1e-13 relative of libm's pow
// for the visit counts and temperatures a schedule can produce (tests/test_oracle_mct.py pins it against the reference's pi).
AZ_D double az_det_pow(double n, double inv_temp) {
    if (!(n > 0.0)) return 0.0;
    return az_det_exp(az_det_log(n) * inv_temp);
}

struct Philox4 { u32 x, y, z, w; };

AZ_D Philox4 az_philox(u32 k0, u32 k1, u32 c0, u32 c1, u32 c2, u32 c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        u64 p0 = (u64)0xD2511F53u * c0;
        u64 p1 = (u64)0xCD9E8D57u * c2;
        u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0;
        u32 n1 = (u32)p1;
        u32 n2 = (u32)(p0 >> 32) ^ c3 ^ k1;
        u32 n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o = {c0, c1, c2, c3};
    return o;
}

AZ_D double az_u53(u32 a, u32 b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

AZ_D u64 az_splitmix64(u64 z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------------
// bitboards
// ---------------------------------------------------------------------------------------------
#define AZ_COL0 0x0101010101010101ULL
#define AZ_COL7 0x8080808080808080ULL

struct BB {
    u64 p1, m1;  // discs of player +1 / player -1
    int player;  // side to move
};

AZ_D u64 sh_e(u64 x) { return (x << 1) & ~AZ_COL0; }
AZ_D u64 sh_w(u64 x) { return (x >> 1) & ~AZ_COL7; }
AZ_D u64 sh_s(u64 x) { return x << 8; }
AZ_D u64 sh_n(u64 x) { return x >> 8; }
AZ_D u64 sh_se(u64 x) { return (x << 9) & ~AZ_COL0; }
AZ_D u64 sh_sw(u64 x) { return (x << 7) & ~AZ_COL7; }
AZ_D u64 sh_ne(u64 x) { return (x >> 7) & ~AZ_COL0; }
AZ_D u64 sh_nw(u64 x) { return (x >> 9) & ~AZ_COL7; }

#define AZ_FOR_DIRS(M) M(sh_e) M(sh_w) M(sh_s) M(sh_n) M(sh_se) M(sh_sw) M(sh_ne) M(sh_nw)

// othello.py:141-189 as a flood fill: empty cells from which `own` brackets a run of `opp`
AZ_D u64 oth_legal(u64 own, u64 opp, u64 valid) {
    u64 empty = ~(own | opp) & valid;
    u64 legal = 0;
#define AZ_M(SH)                                  \
    {                                             \
        u64 x = SH(own) & opp;                    \
        x |= SH(x) & opp; x |= SH(x) & opp;       \
        x |= SH(x) & opp; x |= SH(x) & opp;       \
        x |= SH(x) & opp;                         \
        legal |= SH(x) & empty;                   \
    }
    AZ_FOR_DIRS(AZ_M)
#undef AZ_M
    return legal;
}

// othello.py:141-153, 206-208: discs flipped when `own` plays the single-bit move `mv`
AZ_D u64 oth_flips(u64 own, u64 opp, u64 mv) {
    u64 flips = 0;
#define AZ_M(SH)                                  \
    {                                             \
        u64 x = SH(mv) & opp;                     \
        x |= SH(x) & opp; x |= SH(x) & opp;       \
        x |= SH(x) & opp; x |= SH(x) & opp;       \
        x |= SH(x) & opp;                         \
        if (SH(x) & own) flips |= x;              \
    }
    AZ_FOR_DIRS(AZ_M)
#undef AZ_M
    return flips;
}

// connect4.py:183-245: does bitboard x hold four in a row (any direction)?
AZ_D bool c4_four(u64 x) {
    const u64 C0 = AZ_COL0, C1 = AZ_COL0 << 1, C6 = AZ_COL0 << 6, C7 = AZ_COL7;
    u64 a = x & (x >> 1) & ~C7;           // (r,c),(r,c+1)
    u64 b = a & (a >> 2) & ~C6;
    a = x & (x >> 8);                     // vertical
    b |= a & (a >> 16);
    a = x & (x >> 9) & ~C7;               // (r,c),(r+1,c+1)
    b |= a & (a >> 18) & ~C6;
    a = x & (x >> 7) & ~C0;               // (r,c),(r+1,c-1)
    b |= a & (a >> 14) & ~C1;
    return b != 0;
}

AZ_D u64 c4_colmask(const GameDesc& gd, int c) { return (AZ_COL0 << c) & gd.valid; }

AZ_D bool ttt_line(u64 x) {
    const u64 L[8] = {0x7ULL, 0x7ULL << 8, 0x7ULL << 16, 0x010101ULL, 0x010101ULL << 1, 0x010101ULL << 2,
                      (1ULL | (1ULL << 9) | (1ULL << 18)), ((1ULL << 2) | (1ULL << 9) | (1ULL << 16))};
    bool w = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) w |= ((x & L[i]) == L[i]);
    return w;
}

// Legal actions of the side to move as a bit set in *action-bit* space: Othello / TicTacToe use the
// cell bit (r*8+c); Connect4 uses bit c for column c.  Ascending bit order == ascending action
// index.  Othello: an empty set means the only legal action is the pass (othello.py:187-188).
AZ_D u64 az_legal_bits(const GameDesc& gd, const BB& b, int player) {
    u64 own = player > 0 ? b.p1 : b.m1, opp = player > 0 ? b.m1 : b.p1;
    if (gd.game == AZ_OTHELLO) return oth_legal(own, opp, gd.valid);
    u64 occ = b.p1 | b.m1;
    if (gd.game == AZ_TICTACTOE) return ~occ & gd.valid;
    u64 m = 0;
    for (int c = 0; c < gd.W; ++c)
        if ((occ & c4_colmask(gd, c)) != c4_colmask(gd, c)) m |= 1ULL << c;
    return m;
}

AZ_D int az_bit_to_action(const GameDesc& gd, int bit) {
    return (gd.game == AZ_CONNECT4 || gd.W == 8) ? bit : (bit >> 3) * gd.W + (bit & 7);
}
AZ_D int az_action_to_bit(const GameDesc& gd, int a) {
    // (an 8-wide board: cell index = bit index, and no integer division on the tree walk's path)
    return (gd.game == AZ_CONNECT4 || gd.W == 8) ? a : (a / gd.W) * 8 + (a % gd.W);
}

// plays a LEGAL action (legality is the caller's business); flips the side to move
AZ_D void az_play(const GameDesc& gd, BB& b, int action) {
    u64 own = b.player > 0 ? b.p1 : b.m1, opp = b.player > 0 ? b.m1 : b.p1;
    if (gd.game == AZ_OTHELLO) {
        if (action != gd.cells) {  // not a pass
            u64 mv = 1ULL << az_action_to_bit(gd, action);
            u64 f = oth_flips(own, opp, mv);
            own |= mv | f;
            opp &= ~f;
        }
    } else if (gd.game == AZ_CONNECT4) {  // connect4.py:176-180: lowest free row of the column
        u64 cm = c4_colmask(gd, action);
        int filled = __popcll((own | opp) & cm);
        own |= 1ULL << ((gd.H - 1 - filled) * 8 + action);
    } else {
        own |= 1ULL << az_action_to_bit(gd, action);
    }
    if (b.player > 0) { b.p1 = own; b.m1 = opp; } else { b.m1 = own; b.p1 = opp; }
    b.player = -b.player;
}

// is_game_over + get_winner.  Returns true when over; *winner in {-1,0,+1} (absolute player id).
// Connect4/TicTacToe: positions reachable by legal play have at most one aligned side; if both
// are aligned (unreachable) +1 is reported.
AZ_D bool az_status(const GameDesc& gd, const BB& b, int* winner) {
    if (gd.game == AZ_OTHELLO) {  // othello.py:212-229
        if (oth_legal(b.p1, b.m1, gd.valid) | oth_legal(b.m1, b.p1, gd.valid)) return false;
        int d = __popcll(b.p1) - __popcll(b.m1);
        *winner = d > 0 ? 1 : (d < 0 ? -1 : 0);
        return true;
    }
    bool w1, w2;
    if (gd.game == AZ_CONNECT4) { w1 = c4_four(b.p1); w2 = c4_four(b.m1); }
    else { w1 = ttt_line(b.p1); w2 = ttt_line(b.m1); }
    if (w1) { *winner = 1; return true; }
    if (w2) { *winner = -1; return true; }
    if (((b.p1 | b.m1) & gd.valid) == gd.valid) { *winner = 0; return true; }
    return false;
}

// ---------------------------------------------------------------------------------------------
// 16-lane cooperative variants (engine tree walk: 16 lanes per game).  The 8 flood directions of the
// Othello rules go to 8 lanes (lanes 8..15 mirror them for the other colour where that helps), the
// partial bitboards are OR-reduced with three width-8 shuffles.  All 16 lanes must call these together
// and all get the full result.
// ---------------------------------------------------------------------------------------------
struct DirLane {  // this lane's direction: 0 E, 1 S, 2 SE, 3 SW, 4 W, 5 N, 6 NW, 7 NE
    int amt;
    bool left;
    u64 mask;
};
AZ_D DirLane az_dir_lane(int sub) {
    const int d = sub & 7, k = d & 3;
    DirLane L;
    L.amt = k == 0 ? 1 : (k == 1 ? 8 : (k == 2 ? 9 : 7));
    L.left = d < 4;
    // E, SE, NE must not wrap into column 0; W, NW, SW must not wrap into column 7
    const bool to_col0 = (d == 0 || d == 2 || d == 7), to_col7 = (d == 4 || d == 6 || d == 3);
    L.mask = to_col0 ? ~AZ_COL0 : (to_col7 ? ~AZ_COL7 : ~0ULL);
    return L;
}
AZ_D u64 dsh(const DirLane& L, u64 x) { return (L.left ? (x << L.amt) : (x >> L.amt)) & L.mask; }
// Group reductions by DPP: lane permutations inside the VALU (a few cycles each) instead of ds_bpermute shuffles through the LDS crossbar
// (~100 cycles each, three to four dependent ones per reduction, several reductions per tree level on a SIMD that holds one wave).  A game's
// 16 lanes are one DPP row.  quad_perm swaps neighbours / pairs, row_half_mirror maps lane i of an 8-lane half to 7 - i, row_mirror lane i
// of the row to 15 - i: for a commutative, idempotent combine (OR, max) three steps cover 8 lanes, four cover 16, and every lane ends with
// the full result -- the same value the butterfly of shuffles gave.
#define AZ_DPP_SWAP1 0xB1     // quad_perm [1, 0, 3, 2]
#define AZ_DPP_SWAP2 0x4E     // quad_perm [2, 3, 0, 1]
#define AZ_DPP_HMIRROR 0x141  // row_half_mirror
#define AZ_DPP_MIRROR 0x140   // row_mirror
template <int CTRL>
AZ_D u32 az_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
AZ_D u64 az_dpp64(u64 v) { return ((u64)az_dpp<CTRL>((u32)(v >> 32)) << 32) | (u64)az_dpp<CTRL>((u32)v); }
AZ_D u64 or8(u64 v) {
    v |= az_dpp64<AZ_DPP_SWAP1>(v);
    v |= az_dpp64<AZ_DPP_SWAP2>(v);
    v |= az_dpp64<AZ_DPP_HMIRROR>(v);
    return v;
}
AZ_D u64 oth_legal_dir(const DirLane& L, u64 own, u64 opp, u64 empty) {
    u64 x = dsh(L, own) & opp;
    x |= dsh(L, x) & opp; x |= dsh(L, x) & opp; x |= dsh(L, x) & opp; x |= dsh(L, x) & opp; x |= dsh(L, x) & opp;
    return dsh(L, x) & empty;
}
// legal action bits of `player` (see az_legal_bits)
AZ_D u64 az_legal_bits_grp(const GameDesc& gd, const BB& b, int player, int sub) {
    if (gd.game != AZ_OTHELLO) return az_legal_bits(gd, b, player);
    u64 own = player > 0 ? b.p1 : b.m1, opp = player > 0 ? b.m1 : b.p1;
    return or8(oth_legal_dir(az_dir_lane(sub), own, opp, ~(own | opp) & gd.valid));
}
AZ_D void az_play_grp(const GameDesc& gd, BB& b, int action, int sub) {
    if (gd.game != AZ_OTHELLO) { az_play(gd, b, action); return; }
    u64 own = b.player > 0 ? b.p1 : b.m1, opp = b.player > 0 ? b.m1 : b.p1;
    if (action != gd.cells) {
        const DirLane L = az_dir_lane(sub);
        const u64 mv = 1ULL << az_action_to_bit(gd, action);
        u64 x = dsh(L, mv) & opp;
        x |= dsh(L, x) & opp; x |= dsh(L, x) & opp; x |= dsh(L, x) & opp; x |= dsh(L, x) & opp; x |= dsh(L, x) & opp;
        const u64 f = or8((dsh(L, x) & own) ? x : 0ULL);
        own |= mv | f;
        opp &= ~f;
    }
    if (b.player > 0) { b.p1 = own; b.m1 = opp; } else { b.m1 = own; b.p1 = opp; }
    b.player = -b.player;
}
AZ_D bool az_status_grp(const GameDesc& gd, const BB& b, int* winner, int sub) {
    if (gd.game != AZ_OTHELLO) return az_status(gd, b, winner);
    // lanes 0..7 flood for player +1, lanes 8..15 for player -1
    const bool second = (sub & 8) != 0;
    const u64 own = second ? b.m1 : b.p1, opp = second ? b.p1 : b.m1;
    u64 any = or8(oth_legal_dir(az_dir_lane(sub), own, opp, ~(own | opp) & gd.valid));
    any |= az_dpp64<AZ_DPP_MIRROR>(any);  // the other colour's half of the row
    if (any) return false;
    int d = __popcll(b.p1) - __popcll(b.m1);
    *winner = d > 0 ? 1 : (d < 0 ? -1 : 0);
    return true;
}

AZ_D int az_cell_value(const BB& b, int r, int c) {
    u64 bit = 1ULL << (r * 8 + c);
    return (b.p1 & bit) ? 1 : ((b.m1 & bit) ? -1 : 0);
}

// hash of the canonical board player*grid (closed-form fake net / closed-form noise, test modes)
AZ_D u64 az_board_hash(const GameDesc& gd, const BB& b) {
    u64 h = 0x9E3779B97F4A7C15ULL;
    for (int r = 0; r < gd.H; ++r)
        for (int c = 0; c < gd.W; ++c) {
            u64 cg = (u64)(b.player * az_cell_value(b, r, c) + 1);
            h = (h ^ cg) * 0x100000001B3ULL;
        }
    return az_splitmix64(h);
}
