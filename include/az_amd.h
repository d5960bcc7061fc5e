/*
 * az_amd.h -- C ABI of the MI355X-native AlphaZero self-play engine (libaz_amd.so).
 *
 * The reference (t0m1ab/alphazero) is pure Python and has no FFI; its extension surface is the
 * Board / PolicyValueNetwork / MCT / AlphaZeroTrainer.self_play classes.  Each entry point below
 * names the reference interface it stands in for (paths relative to /root/reference/alphazero/).
 * The Python host layer (alphazero_amd/) binds these with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  `stream` is a hipStream_t passed as
 *     void* (NULL = default stream).  Pointers prefixed d_ are DEVICE pointers, h_ are HOST pointers.
 *   - every function returns 0 on success or a negative AZ_E* code and never throws;
 *     az_last_error() returns the message of the calling thread's last failure.
 *   - encodings (identical to the reference's numpy objects):
 *       grid    int8 [H*W] row-major, values {-1,0,+1}      (Board.grid, base.py:112)
 *       player  int8 in {+1,-1}                              (Board.player)
 *       action  othello r*n+c, pass = n*n; tictactoe 3r+c; connect4 column
 *               (PolicyValueNetwork.to_neural_output, othello.py:404-412, connect4.py:430-435)
 *   - an engine handle is not re-entrant; different handles may be driven from different threads.
 */
#ifndef AZ_AMD_H
#define AZ_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_GAME_OTHELLO 0
#define AZ_GAME_CONNECT4 1
#define AZ_GAME_TICTACTOE 2

#define AZ_OK 0
#define AZ_EINVAL (-1)   /* bad argument (reference: ValueError in constructors) */
#define AZ_EHIP (-2)     /* HIP runtime failure */
#define AZ_ESTATE (-3)   /* call order / missing weights */
#define AZ_ECAPACITY (-4)/* node pool or sample buffer exhausted */
#define AZ_EILLEGAL (-5) /* illegal move (reference: ValueError, othello.py:199-200) */

#define AZ_TIE_MODE_LOWEST 0  /* fair_max ties -> lowest action index (deterministic, tests) */
#define AZ_TIE_MODE_RANDOM 1  /* fair_max ties -> uniform (utils.py:28-34), Philox stream */
#define AZ_NOISE_MODE_OFF 0
#define AZ_NOISE_MODE_PHILOX 1 /* root Dirichlet noise (mcts.py:235-240) from the Philox stream */
#define AZ_NOISE_MODE_HASH 2   /* closed-form noise from the root board hash (tests) */
#define AZ_EVAL_NET 0          /* PolicyValueNetwork.evaluate (base.py:357-367) on the HIP network */
#define AZ_EVAL_FAKE 1         /* closed-form fake network (tests; tools/closed_form.py) */
#define AZ_EVAL_ROLLOUT 2      /* TreeEval.ROLLOUT: plain UCT with random playouts, no network (mcts.py:38-42, 173-180) */

const char *az_last_error(void);
int az_version(void);

/* ---- batched board rules (K1/K2) -------------------------------------------------------------
 * replaces Board.get_moves / is_legal_move / play_move / is_game_over / get_winner / get_score
 * (othello.py:133-229, connect4.py:143-258, tictactoe.py:139-184) for n positions at once. */

/* d_legal[n][A] uint8: 1 where the action is legal for `d_for_player[i]` (0 = side to move,
 * base.py:163-171 `player` argument).  d_for_player may be NULL. */
int az_board_legal_batch(int game, int H, int W, const int8_t *d_grids, const int8_t *d_players,
                         const int8_t *d_for_player, int64_t n, uint8_t *d_legal, void *stream);
/* plays d_actions[i]; d_status[i] = 0 ok / AZ_EILLEGAL (board then copied unchanged). */
int az_board_play_batch(int game, int H, int W, const int8_t *d_grids, const int8_t *d_players,
                        const int32_t *d_actions, int64_t n, int8_t *d_out_grids, int8_t *d_out_players,
                        int32_t *d_status, void *stream);
/* d_over[i] uint8, d_winner[i] int8 (valid where over; 2 otherwise), d_score[i] int32 =
 * sum(player*grid) (Board.get_score of othello/connect4). */
int az_board_status_batch(int game, int H, int W, const int8_t *d_grids, const int8_t *d_players, int64_t n,
                          uint8_t *d_over, int8_t *d_winner, int32_t *d_score, void *stream);

/* ---- policy-value network (K5/K6) ------------------------------------------------------------
 * replaces OthelloNet / Connect4Net / TicTacToeNet .forward + PolicyValueNetwork.predict
 * (othello.py:341-382, connect4.py:370-412, tictactoe.py:289-316, base.py:350-355), eval mode. */
typedef struct az_net az_net;
int az_net_create(int game, int H, int W, int max_batch, az_net **out);
void az_net_destroy(az_net *net);
/* h_data: HOST float32 tensor of the torch state_dict entry `name` ("conv1.weight",
 * "bn1.running_var", "fc_probs.bias", ...).  Unknown keys (num_batches_tracked) return AZ_OK. */
int az_net_set_tensor(az_net *net, const char *name, const float *h_data, int64_t numel);
/* folds eval-mode BatchNorm into the preceding layer (float64), re-tiles the weights into MFMA
 * fragment order and uploads them.  Must be called after all tensors are set / updated. */
int az_net_commit(az_net *net, void *stream);
/* The same hand-off without a host round trip (update_network, trainer.py:383-387, when the trained module lives on the
 * GPU): d_data is a DEVICE float32 tensor, copied on `stream`; az_net_commit_device folds BatchNorm and re-tiles with
 * device kernels on `stream` (same float64 operation order as az_net_commit: identical bits).  A commit uses the
 * tensors of ITS kind only: set every tensor through the same path. */
int az_net_set_tensor_device(az_net *net, const char *name, const float *d_data, int64_t numel, void *stream);
int az_net_commit_device(az_net *net, void *stream);
/* d_input[B][H*W] float32 canonical boards (player*grid, base.py:363);
 * d_probs[B][A] = exp(log_softmax) policy, d_value[B] = tanh value. */
int az_net_forward(az_net *net, const float *d_input, int B, float *d_probs, float *d_value, void *stream);
/* same, but only the first min(*d_count, max_B) rows are evaluated: d_count is a DEVICE int written by an
 * earlier kernel on `stream` (the engine compacts the leaves that need an evaluation into the front rows). */
int az_net_forward_dyn(az_net *net, const float *d_input, const int32_t *d_count, int max_B, float *d_probs,
                       float *d_value, void *stream);
int az_net_action_size(const az_net *net);
/* algorithmic FLOPs of one forward per board (2*MAC) */
int64_t az_net_flops_per_board(const az_net *net);
/* times `iters` back-to-back launches of one forward stage with HIP events on `stream`;
 * stage: 0 conv trunk, 1 fc1, 2 fc2, 3 heads, -1 whole forward.  *ms_per_launch out. */
int az_net_time_stage(az_net *net, int stage, int B, int iters, void *stream, float *ms_per_launch);
/* name of the kernel that stage (0..3) launches for a batch of B boards, as rocprofv3 lists it (template arguments abbreviated) */
int az_net_stage_kernel(const az_net *net, int stage, int B, char *buf, int cap);
/* Live measurement (the idiom of timers.py:53-76 applied per kernel): while enabled, every forward brackets its stage
 * launches with a start and a stop event each (hipExtLaunchKernelGGL).  ms_total[8] / launches[8], one slot per kernel family so that a slot's mean is the
 * figure rocprofv3 lists for that kernel: k_trunk2, fc1 and fc2 on the tiled GEMMs (k_gemm / k_gemm_solo; Connect4Net's fused tail
 * in the fc1 slot), k_heads, k_trunk (one board per wave) | fc1, fc2 on the small-batch kernels (k_dense_frag / k_dense_small),
 * k_trunk_q. */
int az_net_profile(az_net *net, int enable);
int az_net_profiling(const az_net *net); /* 1 while enabled (the engine then launches kernel by kernel instead of replaying graphs) */
int az_net_profile_read(az_net *net, double *ms_total, int64_t *launches);
/* correction to subtract per launch from ms_total before comparing with rocprofv3's kernel durations: 0 -- every profiled launch
 * carries its own start / stop events (the dispatch's begin / end timestamps); kept for callers written against rounds 1-3, which
 * recorded events BETWEEN the launches and calibrated the cost of an empty interval */
int az_net_profile_overhead(az_net *net, double *ms_per_interval);

/* ---- self-play engine (K3/K4/K7/K8/K9) -------------------------------------------------------
 * replaces AlphaZeroTrainer.self_play (trainer.py:215-273) driving AlphaZeroPlayer.get_move
 * (players.py:158-191) / MCT.search (mcts.py:226-269) for n_slots concurrent games in lock-step. */
typedef struct az_engine az_engine;
typedef struct {
    int32_t game, H, W;
    int32_t n_slots;           /* concurrent games resident in HBM */
    int32_t n_sim;             /* Config.simulations */
    double dirichlet_alpha;    /* < 0 : None */
    double dirichlet_epsilon;  /* < 0 : None */
    int32_t temp_max_step, temp_min_step; /* LinearTemperatureScheduler (schedulers.py:20-40) */
    int32_t tie_mode, noise_mode, evaluator;
    uint32_t seed;             /* Philox key word 0; word 1 is the game id */
    int32_t node_capacity;     /* nodes per tree pool (two pools per slot; the kept subtree is compacted at every move) */
    int32_t max_plies;         /* per game, sample staging */
    int64_t sample_capacity;   /* samples the output buffers can hold */
} az_engine_cfg;

typedef struct {
    int64_t games_done, samples, net_evals, lockstep_iters, plies;
    int32_t max_nodes_used, error_flags;
    int64_t graph_replays;  /* searches issued as one HIP graph launch since the engine was created */
    int32_t max_path_len;   /* longest root..leaf path (nodes) of a simulation since the last run/set_roots; only paths
                               longer than 16 nodes are recorded (they take the parent-chasing back-propagation), else 0 */
    int32_t reserved;
} az_engine_stats;

/* `stream` (hipStream_t, may be the default stream): the stream the caller's own work is queued on.  The engine runs on a
 * stream of its own: every call first orders that stream behind `stream` and returns only when the engine's work is done.
 * One search (1 + 5 n_sim launches) is captured as a HIP graph the second time it is issued with the same shape and
 * replayed from then on (AZ_ENGINE_GRAPHS=0 in the environment switches that off). */
int az_engine_create(const az_engine_cfg *cfg, az_net *net, void *stream, az_engine **out);
void az_engine_destroy(az_engine *e);
/* plays games first_game_id .. first_game_id+n_games-1 to completion (slots are refilled as games
 * end) and blocks until done.  Samples accumulate in the output buffers from index 0. */
int az_engine_run(az_engine *e, uint32_t first_game_id, int32_t n_games);
int az_engine_get_stats(az_engine *e, az_engine_stats *out);
/* device views of the normalised samples (trainer.py:262-265, Sample.normalize):
 *   states int8 [S][H*W] = grid*player, pis float32 [S][A], zs int8 [S] = winner*player,
 *   meta int32 [S][4] = {game_id, move_idx, player, action}, visits int32 [S][A]. */
int az_engine_samples(az_engine *e, int64_t *n_samples, const int8_t **d_states, const float **d_pis,
                      const int8_t **d_zs, const int32_t **d_meta, const int32_t **d_visits);

/* finer-grained control (tests, arena-style use): */
/* puts n_roots positions into slots 0..n-1 (fresh trees); h_* are HOST arrays. */
int az_engine_set_roots(az_engine *e, const int8_t *h_grids, const int8_t *h_players, const uint32_t *h_game_ids,
                        const int32_t *h_plies, int32_t n_roots);
int az_engine_search(az_engine *e, int32_t n_sim);     /* MCT.search on every active slot */
/* MCT.get_action_probs + sampled move + Board.play_move + MCT.change_root (+ sample record) */
int az_engine_advance(az_engine *e);
/* Board.play_move + MCT.change_root (mcts.py:118-125) with externally chosen moves for slots 0..n-1 (arena
 * opponent / human): re-roots at the child when the tree holds it, else starts a fresh root.  AZ_EILLEGAL and
 * h_status[i] = AZ_EILLEGAL for an illegal move (reference: ValueError), boards untouched for those slots. */
int az_engine_play(az_engine *e, const int32_t *h_actions, int32_t n, int32_t *h_status);
/* root statistics of one slot to HOST arrays (capacity AZ_MAX 65): actions, N, Q, P */
int az_engine_root_children(az_engine *e, int32_t slot, int32_t *h_actions, int32_t *h_N, double *h_Q,
                            double *h_P, int32_t *count, int32_t *root_N);
/* The reference's tree (mcts.py:8-47 Node objects) grows without bound while MCT.search is called again and again on one root;
 * the engine's pools have a fixed size: nodes the slot's live pool holds, and re-allocation of all pools with a larger capacity
 * (trees kept).  The single-game MCT mirror grows its pools before a search could exhaust them. */
int az_engine_nodes_used(az_engine *e, int32_t slot, int32_t *n_nodes);
int az_engine_grow_pools(az_engine *e, int32_t node_capacity);

/* ---- arena support: Arena.play_game(s) (arena.py:36-185) for all slots at once -----------
 * h_sides[g] = +1/-1: the colour this engine searches for in slot g (0 = both colours, self-play).  az_engine_search
 * then only touches slots where that colour is to move.  az_engine_best_moves = MCT.get_action_probs(temp = 0) for
 * those slots (-1 elsewhere); az_engine_baseline_moves = RandomPlayer (kind 0) / GreedyPlayer (kind 1)
 * (players.py:76-123) for the slots where the OTHER colour is to move; az_engine_play applies a move vector
 * (-1 = none) to the boards and trees; az_engine_root_status reads every slot's position back (h_score = Board.get_score of
 * the side to move: sum(player*grid) for Othello / Connect4; TicTacToe 32767 for the reference's +inf, tictactoe.py:119-126, else 0). */
int az_engine_set_sides(az_engine *e, const int8_t *h_sides, int32_t n);
int az_engine_best_moves(az_engine *e, int32_t *h_actions);
int az_engine_baseline_moves(az_engine *e, int32_t kind, uint32_t seed, int32_t *h_actions);
/* az_engine_search in two halves, so that the arena's two players (arena.py:135-140: player1.get_move / player2.get_move, each on
 * the games where it is to move) think at the same time: _begin queues the search on the engine's own stream and returns, _end
 * waits for it and reports what az_engine_search would have.  Only calls on OTHER engines may come between the two. */
int az_engine_search_begin(az_engine *e, int32_t n_sim);
/* puts b's stream on a hardware queue that a's does not share (a stream of another priority), so that the two searches really
 * run side by side; before b's first search */
int az_engine_pair(az_engine *a, az_engine *b);
int az_engine_search_end(az_engine *e);
int az_engine_root_status(az_engine *e, int8_t *h_players, uint8_t *h_over, int8_t *h_winner, int32_t *h_score);

/* ---- symmetry augmentation on the device ------------------------------------
 * replaces the loop of AlphaZeroTrainer.self_play (trainer.py:275-284) over Sample.create_reflection_twin /
 * create_rotation_twin: every sample with move_idx >= 2 gets its 7 twins (Connect4: 1) in the reference's
 * order; out meta[.][3] holds the transformation code 1..7 (1 reflection_horizontal, 2 rotation_90,
 * 3 reflection_horizontal+rotation_90, 4 rotation_180, ...).  az_augment_count returns the number of twins. */
int az_augment_count(int game, const int32_t *d_meta, int64_t S, int64_t *n_out, void *stream);
int az_augment(int game, int H, int W, const int8_t *d_state, const float *d_pi, const int8_t *d_z, const int32_t *d_meta,
               int64_t S, int8_t *d_out_state, float *d_out_pi, int8_t *d_out_z, int32_t *d_out_meta, int64_t out_capacity,
               void *stream);

/* ---- training step --------------------------------------------------------------------------
 * replaces the body of AlphaZeroTrainer.optimize_network's batch loop (trainer.py:346-366: zero_grad, forward in train
 * mode, loss_pi + loss_v, backward, optimizer.step) and torch.optim.SGD(lr, momentum, weight_decay) (trainer.py:326) for
 * OthelloNet / Connect4Net on device-resident samples.  Tensors travel under the reference's state-dict names and in
 * torch's layouts (othello.py:341-368, connect4.py:370-389); DEVICE pointers throughout. */
typedef struct az_trainer az_trainer;
/* game AZ othello (0, H = W in {6, 8}) or connect4 (1, H x W board); max_batch: multiple of 16, <= 512 */
int az_trainer_create(int game, int H, int W, int max_batch, az_trainer **out);
void az_trainer_destroy(az_trainer *t);
/* parameters and BatchNorm running statistics in (load) and out (store): name = state-dict key, numel must match */
int az_trainer_load(az_trainer *t, const char *name, const float *d_src, int64_t numel, void *stream);
int az_trainer_store(az_trainer *t, const char *name, float *d_dst, int64_t numel, void *stream);
/* a fresh optimizer (momentum buffers zeroed, step counter 0): what optimize_network creates per iteration (trainer.py:326) */
int az_trainer_begin(az_trainer *t, float lr, float momentum, float weight_decay, float dropout_p, uint32_t seed, void *stream);
int az_trainer_set_lr(az_trainer *t, float lr, void *stream); /* ExponentialLR between epochs (trainer.py:327, 381) */
/* n_steps steps: step s trains on rows d_perm[s*B .. s*B+B) of the sample arrays (d_state int8 [S][H*W] = grid * player,
 * d_pi float [S][A], d_z int8 [S], S = n_samples = len(memory) of trainer.py:288-318, whose indices d_perm holds) and writes its
 * losses to d_loss_pi[s], d_loss_v[s] (trainer.py:352-353).  Asynchronous.  A permutation entry outside [0, n_samples) is never
 * used as an address: the kernels train that batch slot on row 0 and raise a sticky flag, reported as AZ_EINVAL by az_trainer_check
 * (which waits for the enqueued steps) or by the next az_trainer_steps / az_trainer_begin call -- which checks the flag FIRST and, when
 * it reports it, has changed nothing and enqueued nothing of its own (the flag is cleared by the report: repeat the call). */
int az_trainer_steps(az_trainer *t, const int8_t *d_state, const float *d_pi, const int8_t *d_z, int64_t n_samples, const int64_t *d_perm,
                     int32_t n_steps, int32_t B, float *d_loss_pi, float *d_loss_v, void *stream);
int az_trainer_check(az_trainer *t);
/* test access: device pointer and element count of a workspace buffer of the last step ("c1".."c4", "y1", "h1", "dz1", ...) */
int az_trainer_debug(az_trainer *t, const char *name, void **d_ptr, int64_t *numel);

#ifdef __cplusplus
}
#endif
#endif
