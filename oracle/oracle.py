"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under alphazero_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OTHELLO, CONNECT4, TICTACTOE = 0, 1, 2
TIE_LOWEST, TIE_RANDOM = 0, 1
NOISE_OFF, NOISE_PHILOX, NOISE_HASH = 0, 1, 2
EVAL_ROLLOUT, EVAL_NEURAL = 0, 1
GAME_IDS = {"othello": OTHELLO, "connect4": CONNECT4, "tictactoe": TICTACTOE}


class Board(C.Structure):
    _fields_ = [("game", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("player", C.c_int32),
                ("grid", C.c_int8 * 64)]

    def grid_np(self):
        return np.array(self.grid[: self.H * self.W], dtype=np.int8).reshape(self.H, self.W)

    def set_grid(self, grid, player):
        g = np.asarray(grid).astype(np.int8).reshape(-1)
        for i, v in enumerate(g):
            self.grid[i] = int(v)
        self.player = int(player)


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Board), C.POINTER(C.c_float), C.POINTER(C.c_float))


class MctCfg(C.Structure):
    _fields_ = [("eval_method", C.c_int), ("eval", C.c_void_p), ("eval_ctx", C.c_void_p),
                ("dirichlet_alpha", C.c_double), ("dirichlet_epsilon", C.c_double),
                ("tie_mode", C.c_int), ("noise_mode", C.c_int), ("seed", C.c_uint32), ("game_id", C.c_uint32)]


class SelfplayCfg(C.Structure):
    _fields_ = [("game", C.c_int), ("H", C.c_int), ("W", C.c_int), ("n_sim", C.c_int),
                ("dirichlet_alpha", C.c_double), ("dirichlet_epsilon", C.c_double),
                ("temp_max_step", C.c_int), ("temp_min_step", C.c_int),
                ("tie_mode", C.c_int), ("noise_mode", C.c_int), ("seed", C.c_uint32), ("eval_method", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "az_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("AZ_ORACLE_LIB")  # e.g. an AddressSanitizer build of az_oracle.c (CPU only)
    if not so:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
    L = C.CDLL(so)
    bp = C.POINTER(Board)
    L.orc_board_init.argtypes = [bp, C.c_int, C.c_int, C.c_int]
    L.orc_action_size.argtypes = [bp]
    L.orc_pass_action.argtypes = [bp]
    L.orc_legal_moves.argtypes = [bp, C.c_int, C.POINTER(C.c_int)]
    L.orc_is_legal.argtypes = [bp, C.c_int, C.c_int]
    L.orc_play.argtypes = [bp, C.c_int]
    L.orc_is_over.argtypes = [bp]
    L.orc_winner.argtypes = [bp, C.POINTER(C.c_int)]
    L.orc_score.argtypes = [bp]
    L.orc_board_hash.argtypes = [bp]
    L.orc_board_hash.restype = C.c_uint64
    L.orc_fakenet_eval.argtypes = [C.c_void_p, bp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_convnet_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_convnet_create.restype = C.c_void_p
    L.orc_convnet_destroy.argtypes = [C.c_void_p]
    L.orc_convnet_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    L.orc_convnet_fold.argtypes = [C.c_void_p]
    L.orc_convnet_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_convnet_folded.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    L.orc_convnet_folded.restype = C.POINTER(C.c_float)
    L.orc_convnet_set_winograd.argtypes = [C.c_void_p, C.c_int]
    L.orc_convnet_winograd.argtypes = [C.c_void_p]
    L.orc_convnet_set_qdense.argtypes = [C.c_void_p, C.c_int]
    L.orc_convnet_qdense.argtypes = [C.c_void_p]
    L.orc_mlpnet_create.restype = C.c_void_p
    L.orc_mlpnet_destroy.argtypes = [C.c_void_p]
    L.orc_mlpnet_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    L.orc_mlpnet_fold.argtypes = [C.c_void_p]
    L.orc_mlpnet_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_det_expf.argtypes = [C.c_float]
    L.orc_det_expf.restype = C.c_float
    L.orc_det_tanhf.argtypes = [C.c_float]
    L.orc_det_tanhf.restype = C.c_float
    L.orc_det_log.argtypes = [C.c_double]
    L.orc_det_log.restype = C.c_double
    L.orc_det_exp.argtypes = [C.c_double]
    L.orc_det_exp.restype = C.c_double
    L.orc_det_pow.argtypes = [C.c_double, C.c_double]
    L.orc_det_pow.restype = C.c_double
    L.orc_philox4x32.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    L.orc_mct_create.argtypes = [C.POINTER(MctCfg)]
    L.orc_mct_create.restype = C.c_void_p
    L.orc_mct_destroy.argtypes = [C.c_void_p]
    L.orc_mct_reset.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_mct_set_ply.argtypes = [C.c_void_p, C.c_int]
    L.orc_mct_search.argtypes = [C.c_void_p, bp, C.c_int]
    L.orc_mct_change_root.argtypes = [C.c_void_p, C.c_int]
    L.orc_mct_root_n.argtypes = [C.c_void_p]
    L.orc_mct_root_children.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_mct_choose.argtypes = [C.c_void_p, bp, C.c_double, C.c_void_p, C.c_void_p]
    L.orc_mct_n_nodes.argtypes = [C.c_void_p]
    L.orc_mct_n_evals.argtypes = [C.c_void_p]
    L.orc_mct_n_evals.restype = C.c_int64
    L.orc_mct_max_path_len.argtypes = [C.c_void_p]
    L.orc_baseline_move.argtypes = [bp, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.orc_selfplay.argtypes = [C.POINTER(SelfplayCfg), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.orc_selfplay.restype = C.c_int64
    vp = C.c_void_p
    L.orc_batch_legal.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int64, vp]
    L.orc_batch_play.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int64, vp, vp, vp]
    L.orc_batch_status.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_int64, vp, vp, vp]
    L.orc_random_positions.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int64, vp, vp, vp]
    L.orc_random_positions.restype = C.c_int64
    _LIB = L
    return L


def game_dims(game, n=8, width=7, height=6):
    """(game_id, H, W) of the grid for a reference game name."""
    if game == "othello":
        return OTHELLO, n, n
    if game == "connect4":
        return CONNECT4, height, width
    return TICTACTOE, 3, 3


def new_board(game_id, H, W):
    b = Board()
    lib().orc_board_init(C.byref(b), game_id, H, W)
    return b


def legal_moves(b, player=0):
    out = (C.c_int * 65)()
    k = lib().orc_legal_moves(C.byref(b), player, out)
    return list(out[:k])


def baseline_move(b, kind, seed, game_id, ply, tie_mode=TIE_RANDOM):
    """RandomPlayer ('random') / GreedyPlayer ('greedy') move for the side to move (players.py:76-123)"""
    return lib().orc_baseline_move(C.byref(b), {"random": 0, "greedy": 1}[kind], seed, game_id, ply, tie_mode)


def fakenet(b):
    A = lib().orc_action_size(C.byref(b))
    probs = np.zeros(A, dtype=np.float32)
    v = C.c_float()
    lib().orc_fakenet_eval(None, C.byref(b), probs.ctypes.data_as(C.POINTER(C.c_float)), C.byref(v))
    return probs, float(v.value)


def fn_ptr(name):
    """address of a built-in evaluator (orc_fakenet_eval / orc_convnet_eval / orc_mlpnet_eval)"""
    return C.cast(getattr(lib(), name), C.c_void_p)


class ConvNet:
    """oracle conv policy-value net fed from a torch state_dict (numpy arrays)."""

    def __init__(self, game_id, H, W, state_dict):
        L = lib()
        self.h = L.orc_convnet_create(game_id, H, W)
        self.A = H * W + 1 if game_id == OTHELLO else W
        self.cells = H * W
        for k, v in state_dict.items():
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            rc = L.orc_convnet_set_tensor(self.h, k.encode(), a.ctypes.data, a.size)
            if rc == -1:
                raise ValueError(f"bad size for {k}: {a.shape}")
        if L.orc_convnet_fold(self.h) != 0:
            raise ValueError("missing tensors")

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.cells)
        B = x.shape[0]
        probs = np.zeros((B, self.A), dtype=np.float32)
        v = np.zeros(B, dtype=np.float32)
        lib().orc_convnet_forward(self.h, x.ctypes.data, B, probs.ctypes.data, v.ctypes.data)
        return probs, v

    def set_winograd(self, on):
        """conv2 in the Winograd F(2x2,3x3) form (the product's default on 8x8 and 7x6 planes) or as a direct convolution (AZ_WINOGRAD=0)"""
        lib().orc_convnet_set_winograd(self.h, 1 if on else 0)

    def winograd(self):
        return bool(lib().orc_convnet_winograd(self.h))

    def set_qdense(self, on):
        """fc1 / fc2 of OthelloNet in the exact block-fixed-point form (the product's AZ_DENSE_I8=1) or as float32 fma chains"""
        lib().orc_convnet_set_qdense(self.h, 1 if on else 0)

    def qdense(self):
        return bool(lib().orc_convnet_qdense(self.h))

    def folded(self, name):
        n = C.c_int64()
        p = lib().orc_convnet_folded(self.h, name.encode(), C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_convnet_destroy(self.h)
                self.h = None
        except Exception:  # interpreter shutdown
            pass


class MlpNet:
    def __init__(self, state_dict):
        L = lib()
        self.h = L.orc_mlpnet_create()
        self.A, self.cells = 9, 9
        for k, v in state_dict.items():
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            if L.orc_mlpnet_set_tensor(self.h, k.encode(), a.ctypes.data, a.size) == -1:
                raise ValueError(f"bad size for {k}")
        if L.orc_mlpnet_fold(self.h) != 0:
            raise ValueError("missing tensors")

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 9)
        B = x.shape[0]
        probs = np.zeros((B, 9), dtype=np.float32)
        v = np.zeros(B, dtype=np.float32)
        lib().orc_mlpnet_forward(self.h, x.ctypes.data, B, probs.ctypes.data, v.ctypes.data)
        return probs, v

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_mlpnet_destroy(self.h)
                self.h = None
        except Exception:  # interpreter shutdown
            pass


class MCT:
    """mcts.MCT restated (oracle).  evaluator: ('fake', None) | ('conv', ConvNet) | ('mlp', MlpNet)"""

    def __init__(self, evaluator=("fake", None), eval_method=EVAL_NEURAL, alpha=-1.0, eps=-1.0,
                 tie_mode=TIE_LOWEST, noise_mode=NOISE_OFF, seed=0, game_id=0):
        kind, net = evaluator
        self._net = net
        cfg = MctCfg()
        cfg.eval_method = eval_method
        fn = {"fake": "orc_fakenet_eval", "conv": "orc_convnet_eval", "mlp": "orc_mlpnet_eval"}[kind]
        cfg.eval = fn_ptr(fn)
        cfg.eval_ctx = net.h if net is not None else None
        cfg.dirichlet_alpha, cfg.dirichlet_epsilon = alpha, eps
        cfg.tie_mode, cfg.noise_mode, cfg.seed, cfg.game_id = tie_mode, noise_mode, seed, game_id
        self.h = lib().orc_mct_create(C.byref(cfg))

    def set_ply(self, ply):
        lib().orc_mct_set_ply(self.h, ply)

    def search(self, board, n_sim):
        if lib().orc_mct_search(self.h, C.byref(board), n_sim) != 0:
            raise RuntimeError("oracle search failed")

    def change_root(self, action):
        lib().orc_mct_change_root(self.h, action)

    def root_children(self):
        a = np.zeros(65, np.int32); n = np.zeros(65, np.int32)
        q = np.zeros(65, np.float64); p = np.zeros(65, np.float64)
        k = lib().orc_mct_root_children(self.h, a.ctypes.data, n.ctypes.data, q.ctypes.data, p.ctypes.data)
        return a[:k].copy(), n[:k].copy(), q[:k].copy(), p[:k].copy()

    def root_n(self):
        return lib().orc_mct_root_n(self.h)

    def max_path_len(self):
        """longest root..leaf path (in nodes) any simulation of this tree has walked"""
        return lib().orc_mct_max_path_len(self.h)

    def reset(self, game_id):
        lib().orc_mct_reset(self.h, game_id)

    def choose(self, board, temp):
        A = lib().orc_action_size(C.byref(board))
        pi = np.zeros(65, np.float64); vis = np.zeros(65, np.int32)
        act = lib().orc_mct_choose(self.h, C.byref(board), float(temp), pi.ctypes.data, vis.ctypes.data)
        return act, pi[:A].copy(), vis[:A].copy()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_mct_destroy(self.h)
                self.h = None
        except Exception:  # interpreter shutdown
            pass


def selfplay(game_id, H, W, n_games, n_sim, evaluator=("fake", None), alpha=0.03, eps=0.25,
             temp_max_step=4, temp_min_step=4, tie_mode=TIE_RANDOM, noise_mode=NOISE_PHILOX, seed=0,
             first_game_id=0, eval_method=EVAL_NEURAL, max_plies=None):
    """trainer.self_play restated: returns dict of normalised sample arrays."""
    kind, net = evaluator
    cfg = SelfplayCfg(game_id, H, W, n_sim, alpha, eps, temp_max_step, temp_min_step, tie_mode, noise_mode, seed,
                      eval_method)
    cells = H * W
    A = H * W + 1 if game_id == OTHELLO else (W if game_id == CONNECT4 else 9)
    cap = n_games * (max_plies or (2 * cells + 8))
    states = np.zeros((cap, cells), np.int8); pis = np.zeros((cap, A), np.float32)
    zs = np.zeros(cap, np.int8); meta = np.zeros((cap, 4), np.int32); visits = np.zeros((cap, A), np.int32)
    n_evals = C.c_int64()
    fn = {"fake": "orc_fakenet_eval", "conv": "orc_convnet_eval", "mlp": "orc_mlpnet_eval"}[kind]
    S = lib().orc_selfplay(C.byref(cfg), fn_ptr(fn), net.h if net is not None else None, first_game_id, n_games,
                           cap, states.ctypes.data, pis.ctypes.data, zs.ctypes.data, meta.ctypes.data,
                           visits.ctypes.data, C.byref(n_evals))
    if S < 0:
        raise RuntimeError("oracle selfplay failed")
    return {"state": states[:S].reshape(S, H, W), "pi": pis[:S], "z": zs[:S], "meta": meta[:S],
            "visits": visits[:S], "n_evals": n_evals.value}


def action_size(game_id, H, W):
    return H * W + 1 if game_id == OTHELLO else (W if game_id == CONNECT4 else 9)


def batch_legal(game_id, H, W, grids, players, for_player=None):
    grids = np.ascontiguousarray(grids, np.int8).reshape(-1, H * W); players = np.ascontiguousarray(players, np.int8)
    n, A = len(players), action_size(game_id, H, W)
    out = np.zeros((n, A), np.uint8)
    fp = np.ascontiguousarray(for_player, np.int8) if for_player is not None else None
    lib().orc_batch_legal(game_id, H, W, grids.ctypes.data, players.ctypes.data, fp.ctypes.data if fp is not None else None,
                          n, out.ctypes.data)
    return out


def batch_play(game_id, H, W, grids, players, actions):
    grids = np.ascontiguousarray(grids, np.int8).reshape(-1, H * W); players = np.ascontiguousarray(players, np.int8)
    actions = np.ascontiguousarray(actions, np.int32)
    n = len(players)
    og = np.zeros_like(grids); op = np.zeros_like(players); st = np.zeros(n, np.int32)
    lib().orc_batch_play(game_id, H, W, grids.ctypes.data, players.ctypes.data, actions.ctypes.data, n, og.ctypes.data,
                         op.ctypes.data, st.ctypes.data)
    return og, op, st


def batch_status(game_id, H, W, grids, players):
    grids = np.ascontiguousarray(grids, np.int8).reshape(-1, H * W); players = np.ascontiguousarray(players, np.int8)
    n = len(players)
    over = np.zeros(n, np.uint8); win = np.zeros(n, np.int8); score = np.zeros(n, np.int32)
    lib().orc_batch_status(game_id, H, W, grids.ctypes.data, players.ctypes.data, n, over.ctypes.data, win.ctypes.data,
                           score.ctypes.data)
    return over, win, score


def random_positions(game_id, H, W, seed, n_games, cap):
    grids = np.zeros((cap, H * W), np.int8); players = np.zeros(cap, np.int8); actions = np.zeros(cap, np.int32)
    n = lib().orc_random_positions(game_id, H, W, seed, n_games, cap, grids.ctypes.data, players.ctypes.data,
                                   actions.ctypes.data)
    return grids[:n], players[:n], actions[:n]


def arena_games(dims, ev1, n_sim, opponent, opp_sim, seed, n_rounds, start_player=None, rounds=None, tie_mode=TIE_RANDOM):
    """Arena.play_games (arena.py:36-185) restated on the oracle: player 1 is an AlphaZero tree (evaluator ev1, no noise,
    temperature 0: what AlphaZeroTrainer.evaluate builds, trainer.py:421-425); `opponent` is "random" / "greedy"
    (players.py:76-123), "mcts" (rollout MCTSPlayer) or another evaluator tuple.  Each side owns a tree and BOTH trees
    receive every move (arena.py:98-99).  Returns (moves per game, winners, scores, stats dict of arena.py:141-147).
    `rounds`: play only these round indices of the n_rounds (rounds are independent given (seed, round)): the lists and the stats
    then cover those rounds, in that order -- how a test follows a few games of an arena too large to replay whole.
    tie_mode=TIE_LOWEST: fair_max replaced by "lowest action among the maxima" in the trees and in the greedy player: the mode
    golden G7 (the reference's Arena.play_games under that patch) is recorded in."""
    from collections import defaultdict
    gid, H, W = dims
    p2_starts = [{1: False, 2: True}.get(start_player, bool(r % 2)) for r in range(n_rounds)]
    all_moves, winners, scores = [], [], []
    played = list(range(n_rounds)) if rounds is None else [int(r) for r in rounds]
    for r in played:
        side1 = -1 if p2_starts[r] else 1
        game_id = (r + seed * 100003) & 0xFFFFFFFF
        t1 = MCT(ev1, eval_method=EVAL_NEURAL, tie_mode=tie_mode, noise_mode=NOISE_OFF, seed=seed, game_id=game_id)
        t2 = None
        if opponent == "mcts":
            t2 = MCT(("fake", None), eval_method=EVAL_ROLLOUT, tie_mode=tie_mode, noise_mode=NOISE_OFF, seed=seed + 1, game_id=game_id)
        elif not isinstance(opponent, str):
            t2 = MCT(opponent, eval_method=EVAL_NEURAL, tie_mode=tie_mode, noise_mode=NOISE_OFF, seed=seed + 1, game_id=game_id)
        b = new_board(gid, H, W)
        ply, moves = 0, []
        while not lib().orc_is_over(C.byref(b)):
            mine = b.player == side1
            if mine or t2 is not None:
                t, ns = (t1, n_sim) if mine else (t2, opp_sim)
                t.set_ply(ply)
                t.search(b, ns)
                a = t.choose(b, 0.0)[0]
            else:
                a = baseline_move(b, opponent, seed + 7, game_id, ply, tie_mode)
            if lib().orc_play(C.byref(b), a) != 0:
                raise RuntimeError("oracle arena: illegal move")
            t1.change_root(a)
            if t2 is not None:
                t2.change_root(a)
            moves.append(a)
            ply += 1
        w = C.c_int()
        lib().orc_winner(C.byref(b), C.byref(w))
        sc = abs(lib().orc_score(C.byref(b)))
        all_moves.append(moves); winners.append(w.value)
        scores.append(float("inf") if gid == TICTACTOE and sc == 32767 else sc)
    stats = {"player1": [], "player2": [], "draw": 0, "player1_starts": defaultdict(int), "player2_starts": defaultdict(int)}
    for i, r in enumerate(played):
        starter = f"player{2 if p2_starts[r] else 1}_starts"
        if winners[i] == 0:
            stats["draw"] += 1
            stats[starter]["draw"] += 1
        else:
            who = 1 if winners[i] == (-1 if p2_starts[r] else 1) else 2
            stats[f"player{who}"].append(scores[i])
            stats[starter]["win" if who == (2 if p2_starts[r] else 1) else "loss"] += 1
    return all_moves, winners, scores, stats
