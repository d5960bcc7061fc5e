/*
 * az_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the self-play hot path of t0m1ab/alphazero
 * (alphazero/mcts.py, alphazero/games/{othello,connect4,tictactoe}.py,
 * alphazero/base.py:350-367, alphazero/trainer.py:215-273).  It exists only to
 * check the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (alphazero_amd/) never does.
 *
 * Parity status: PINNED by golden vectors generated from the imported Python
 * reference (tools/gen_golden.py -> tests/golden/ npz files); the reference's own
 * tests pin nothing on this path.
 *
 * Encodings (shared with the fixtures and with include/az_amd.h):
 *   board   : int8 grid[H*W] row-major with {-1,0,+1}, player in {+1,-1}
 *   action  : othello  r*n+c, pass = n*n      (othello.py:384-412)
 *             tictactoe 3*r+c                 (tictactoe.py:318-341)
 *             connect4 column                 (connect4.py:414-435)
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OTHELLO 0
#define ORC_CONNECT4 1
#define ORC_TICTACTOE 2
#define ORC_MAX_CELLS 64
#define ORC_MAX_ACTIONS 65

/* tie-break of utils.fair_max (utils.py:28-34) */
#define ORC_TIE_LOWEST 0 /* deterministic: lowest action index among the maxima (test mode) */
#define ORC_TIE_RANDOM 1 /* uniform among the maxima, Philox4x32-10 stream */
/* root Dirichlet noise (mcts.py:235-240) */
#define ORC_NOISE_OFF 0
#define ORC_NOISE_PHILOX 1 /* eta ~ Dir(alpha) from the Philox stream */
#define ORC_NOISE_HASH 2   /* closed-form eta from the root board hash (test mode) */
/* move choice of MCTSPlayer.get_move (players.py:184-189) */
#define ORC_EVAL_ROLLOUT 0
#define ORC_EVAL_NEURAL 1

typedef struct {
    int32_t game, H, W, player;
    int8_t grid[ORC_MAX_CELLS];
} orc_board;

/* ---- rules -------------------------------------------------------------- */
void orc_board_init(orc_board *b, int game, int H, int W);
int orc_action_size(const orc_board *b);
int orc_pass_action(const orc_board *b); /* -1 if the game has no pass */
/* legal actions of `player` (0 = side to move) in ascending action index; returns count */
int orc_legal_moves(const orc_board *b, int player, int *out);
int orc_is_legal(const orc_board *b, int action, int player);
int orc_play(orc_board *b, int action); /* 0 ok, -1 illegal (reference raises ValueError) */
int orc_is_over(const orc_board *b);
int orc_winner(const orc_board *b, int *winner); /* 0 ok, -1 game not over */
int orc_score(const orc_board *b);               /* sum(player*grid), othello.py:133-135 */

/* ---- evaluators ---------------------------------------------------------- */
/* probs[A] float32 raw network policy, *v_net = network value in the canonical
 * (side to move = +1) frame; the tree multiplies by board.player (base.py:366). */
typedef void (*orc_eval_fn)(void *ctx, const orc_board *b, float *probs, float *v_net);

/* closed-form fake net (also implemented in tools/gen_golden.py and in the HIP engine) */
uint64_t orc_board_hash(const orc_board *b);
void orc_fakenet_eval(void *ctx, const orc_board *b, float *probs, float *v_net);

/* conv policy-value net (othello.py:306-382, connect4.py:333-412), BN folded */
typedef struct orc_convnet orc_convnet;
orc_convnet *orc_convnet_create(int game, int H, int W);
void orc_convnet_destroy(orc_convnet *n);
/* conv2 in the Winograd F(2x2,3x3) form (what the HIP trunk kernels compute for 8x8 and 7x6 planes) or as a direct conv */
void orc_convnet_set_winograd(orc_convnet *n, int on);
int orc_convnet_winograd(const orc_convnet *n);
void orc_convnet_set_qdense(orc_convnet *n, int on);  /* fc1 / fc2 of OthelloNet in the exact block-fixed-point form (AZ_DENSE_I8) */
int orc_convnet_qdense(const orc_convnet *n);
/* name = state_dict key ("conv1.weight", "bn1.running_var", ...); returns 0 ok */
int orc_convnet_set_tensor(orc_convnet *n, const char *name, const float *data, int64_t numel);
int orc_convnet_fold(orc_convnet *n); /* fold eval-mode BN into weights; 0 ok */
/* input[B][H*W] canonical boards (player*grid) as float; out probs[B][A], v[B] */
void orc_convnet_forward(const orc_convnet *n, const float *input, int B, float *probs, float *v);
void orc_convnet_eval(void *ctx, const orc_board *b, float *probs, float *v_net);
/* folded parameters, for cross-checking the product's host-side fold */
const float *orc_convnet_folded(const orc_convnet *n, const char *name, int64_t *numel);

/* 9-9-9 MLP (tictactoe.py:262-316) */
typedef struct orc_mlpnet orc_mlpnet;
orc_mlpnet *orc_mlpnet_create(void);
void orc_mlpnet_destroy(orc_mlpnet *n);
int orc_mlpnet_set_tensor(orc_mlpnet *n, const char *name, const float *data, int64_t numel);
int orc_mlpnet_fold(orc_mlpnet *n);
void orc_mlpnet_forward(const orc_mlpnet *n, const float *input, int B, float *probs, float *v);
void orc_mlpnet_eval(void *ctx, const orc_board *b, float *probs, float *v_net);

/* deterministic math shared by definition with the HIP kernels */
float orc_det_expf(float x);
float orc_det_tanhf(float x);
double orc_det_log(double x);
double orc_det_exp(double x);
double orc_det_pow(double n, double inv_temp);
void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]);

/* ---- Monte-Carlo tree (mcts.py:49-269) ----------------------------------- */
typedef struct orc_mct orc_mct;
typedef struct {
    int eval_method; /* ORC_EVAL_* */
    orc_eval_fn eval;
    void *eval_ctx;
    double dirichlet_alpha, dirichlet_epsilon; /* < 0 : None */
    int tie_mode, noise_mode;
    uint32_t seed, game_id; /* Philox key */
} orc_mct_cfg;

orc_mct *orc_mct_create(const orc_mct_cfg *cfg);
void orc_mct_destroy(orc_mct *t);
void orc_mct_reset(orc_mct *t, uint32_t game_id);
void orc_mct_set_ply(orc_mct *t, int ply); /* Philox counter word 0 */
int orc_mct_search(orc_mct *t, const orc_board *root, int n_sim); /* 0 ok */
void orc_mct_change_root(orc_mct *t, int action);
int orc_mct_root_n(const orc_mct *t);
/* root children in ascending action order; returns count */
int orc_mct_root_children(const orc_mct *t, int *actions, int *N, double *Q, double *P);
/* mcts.py:95-116 + players.py:184-189: pi[A] (float64), visits[A]; returns the chosen action */
int orc_mct_choose(orc_mct *t, const orc_board *root, double temp, double *pi, int *visits);
int orc_mct_n_nodes(const orc_mct *t);
int64_t orc_mct_n_evals(const orc_mct *t);
int orc_mct_max_path_len(const orc_mct *t); /* longest root..leaf path (in nodes) of any simulation since the last reset */
/* RandomPlayer (kind 0) / GreedyPlayer (kind 1) move for the side to move (players.py:76-123); -1 if no move */
int orc_baseline_move(const orc_board *b, int kind, uint32_t seed, uint32_t game_id, int ply, int tie_mode);

/* ---- self-play (trainer.py:215-273) -------------------------------------- */
typedef struct {
    int game, H, W;
    int n_sim;
    double dirichlet_alpha, dirichlet_epsilon;
    int temp_max_step, temp_min_step; /* LinearTemperatureScheduler, schedulers.py:20-40 */
    int tie_mode, noise_mode;
    uint32_t seed;
    int eval_method;
} orc_selfplay_cfg;

/* plays games first_game_id .. first_game_id+n_games-1 one after the other.
 * outputs (normalised samples, trainer.py:262-265):
 *   states[S][H*W] int8 = grid*player, pis[S][A] float32, zs[S] int8 = winner*player,
 *   meta[S][4] int32 = {game_id, move_idx, player, action_played}, visits[S][A] int32 (may be NULL)
 * returns the number of samples, or -1 on error / overflow of max_samples */
int64_t orc_selfplay(const orc_selfplay_cfg *cfg, orc_eval_fn eval, void *eval_ctx,
                     uint32_t first_game_id, int n_games, int64_t max_samples, int8_t *states,
                     float *pis, int8_t *zs, int32_t *meta, int32_t *visits, int64_t *n_evals);

/* ---- batch helpers for array-wise comparison with the HIP kernels ---------- */
void orc_batch_legal(int game, int H, int W, const int8_t *grids, const int8_t *players, const int8_t *for_player,
                     int64_t n, uint8_t *legal);
void orc_batch_play(int game, int H, int W, const int8_t *grids, const int8_t *players, const int32_t *actions, int64_t n,
                    int8_t *out_grids, int8_t *out_players, int32_t *status);
void orc_batch_status(int game, int H, int W, const int8_t *grids, const int8_t *players, int64_t n, uint8_t *over,
                      int8_t *winner, int32_t *score);
int64_t orc_random_positions(int game, int H, int W, uint32_t seed, int n_games, int64_t cap, int8_t *grids, int8_t *players,
                             int32_t *actions);

#ifdef __cplusplus
}
#endif
#endif
