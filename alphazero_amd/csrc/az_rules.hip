// az_rules.hip -- batched board rules on bitboards (K1/K2), one thread per position.
//
// Replaces Board.get_moves / is_legal_move / play_move / is_game_over / get_winner / get_score of
// alphazero/games/{othello,connect4,tictactoe}.py for n positions held as int8 grids in HBM.
// Each thread packs its grid into two u64 bitboards, runs the shift-and-mask rules of
// az_device.h and writes the result rows.  Byte/integer work, HBM-bound: DESIGN.md gives the
// algorithmic bytes per position.
#include "az_device.h"
#include "az_host.h"

AZ_D BB pack_grid(const GameDesc &gd, const int8_t *g, int player) {
    BB b = {0, 0, player};
    for (int r = 0; r < gd.H; ++r)
        for (int c = 0; c < gd.W; ++c) {
            int v = g[r * gd.W + c];
            if (v > 0) b.p1 |= 1ULL << (r * 8 + c);
            if (v < 0) b.m1 |= 1ULL << (r * 8 + c);
        }
    return b;
}

AZ_D void unpack_grid(const GameDesc &gd, const BB &b, int8_t *g) {
    for (int r = 0; r < gd.H; ++r)
        for (int c = 0; c < gd.W; ++c) g[r * gd.W + c] = (int8_t)az_cell_value(b, r, c);
}

__global__ void k_legal(GameDesc gd, const int8_t *grids, const int8_t *players, const int8_t *for_player, long long n,
                        uint8_t *legal) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BB b = pack_grid(gd, grids + i * gd.cells, players[i]);
    int pl = for_player ? for_player[i] : 0;
    if (pl != 1 && pl != -1) pl = b.player;  // othello.py:143
    u64 bits = az_legal_bits(gd, b, pl);
    uint8_t *out = legal + i * gd.A;
    for (int a = 0; a < gd.A; ++a) out[a] = 0;
    if (gd.game == AZ_OTHELLO && bits == 0) out[gd.A - 1] = 1;  // forced pass, othello.py:187-188
    for (u64 m = bits; m; m &= m - 1) out[az_bit_to_action(gd, __ffsll((long long)m) - 1)] = 1;
}

__global__ void k_play(GameDesc gd, const int8_t *grids, const int8_t *players, const int *actions, long long n,
                       int8_t *out_grids, int8_t *out_players, int *status) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BB b = pack_grid(gd, grids + i * gd.cells, players[i]);
    int a = actions[i];
    u64 bits = az_legal_bits(gd, b, b.player);
    bool ok = false;
    if (a >= 0 && a < gd.A) {
        if (gd.game == AZ_OTHELLO && a == gd.A - 1) ok = (bits == 0);
        else ok = (bits >> az_action_to_bit(gd, a)) & 1ULL;
    }
    if (ok) az_play(gd, b, a);
    unpack_grid(gd, b, out_grids + i * gd.cells);
    out_players[i] = (int8_t)b.player;
    status[i] = ok ? AZ_OK : AZ_EILLEGAL;
}

__global__ void k_status(GameDesc gd, const int8_t *grids, const int8_t *players, long long n, uint8_t *over, int8_t *winner,
                         int *score) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BB b = pack_grid(gd, grids + i * gd.cells, players[i]);
    int w = 2;
    bool o = az_status(gd, b, &w);
    if (over) over[i] = o ? 1 : 0;
    if (winner) winner[i] = (int8_t)(o ? w : 2);
    if (score) score[i] = b.player * (__popcll(b.p1) - __popcll(b.m1));
}

#define AZ_TRY(x) do { int _rc = (x); if (_rc != AZ_OK) return _rc; } while (0)

extern "C" int az_board_legal_batch(int game, int H, int W, const int8_t *d_grids, const int8_t *d_players,
                                    const int8_t *d_for_player, int64_t n, uint8_t *d_legal, void *stream) {
    GameDesc gd;
    AZ_TRY(az_make_game_desc(game, H, W, &gd));
    if (n == 0) return AZ_OK;  // empty batch: nothing to do (pointers may be null)
    AZ_REQUIRE(d_grids && d_players && d_legal && n > 0, AZ_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_legal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gd, d_grids, d_players,
                       d_for_player, (long long)n, d_legal);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_board_play_batch(int game, int H, int W, const int8_t *d_grids, const int8_t *d_players,
                                   const int32_t *d_actions, int64_t n, int8_t *d_out_grids, int8_t *d_out_players,
                                   int32_t *d_status, void *stream) {
    GameDesc gd;
    AZ_TRY(az_make_game_desc(game, H, W, &gd));
    if (n == 0) return AZ_OK;
    AZ_REQUIRE(d_grids && d_players && d_actions && d_out_grids && d_out_players && d_status && n > 0, AZ_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_play, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gd, d_grids, d_players,
                       d_actions, (long long)n, d_out_grids, d_out_players, d_status);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_board_status_batch(int game, int H, int W, const int8_t *d_grids, const int8_t *d_players, int64_t n,
                                     uint8_t *d_over, int8_t *d_winner, int32_t *d_score, void *stream) {
    GameDesc gd;
    AZ_TRY(az_make_game_desc(game, H, W, &gd));
    if (n == 0) return AZ_OK;
    AZ_REQUIRE(d_grids && d_players && n > 0, AZ_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_status, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gd, d_grids, d_players,
                       (long long)n, d_over, d_winner, d_score);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}
