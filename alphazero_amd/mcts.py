"""MCT with the reference's interface (mcts.py:49-269).

The tree lives on the GPU -- one slot of the HIP self-play engine (csrc/az_engine.hip): `search` runs the
simulations there, `change_root` re-roots / compacts the device tree.

* eval_method "neural": PUCT with the policy/value network (lock-step k_step + network forward).
* eval_method "rollout": plain UCT with random playouts (BASELINE config 1), k_rollout_step -- one launch per
  simulation, no network.

For many games at once use alphazero_amd.engine.SelfPlayEngine / alphazero_amd.arena.BatchedArena directly;
this class is the single-game plugin surface (Arena, interactive play).
"""
from time import time

import numpy as np

from .base import TreeEval
from .utils import fair_max

_GAME_IDS = {"othello": 0, "connect4": 1, "tictactoe": 2}


def _action_of(board, move):
    if board.game == "connect4":
        return int(move)
    w = board.grid.shape[1]
    if board.game == "othello" and tuple(move) == tuple(board.pass_move):
        return w * w
    return int(move[0]) * w + int(move[1])


def _move_of(board, action):
    if board.game == "connect4":
        return np.int64(action)
    w = board.grid.shape[1]
    if board.game == "othello" and action == w * w:
        return board.pass_move
    return (action // w, action % w)


class MCT:
    def __init__(self, eval_method=None, nn=None, dirichlet_alpha=None, dirichlet_epsilon=None, seed=None):
        self.n_rollouts = 0
        self.simulation_time = 0
        self.eval_method = TreeEval.to_dict()["rollout" if eval_method is None else eval_method]
        self._nn = nn
        self.dirichlet_alpha = dirichlet_alpha
        self.dirichlet_epsilon = dirichlet_epsilon
        if self._nn is not None and self.eval_method != TreeEval.NEURAL:
            raise ValueError(f"A neural network has been set for the MCT but the evaluation method is {self.eval_method}")
        self._seed = int(np.random.randint(0, 2**31 - 1)) if seed is None else int(seed)
        self._engine = None           # device tree (one engine slot)
        self._engine_board = None     # (game, H, W) the engine was built for
        self._hipnet = None
        self._root_key = None         # (grid bytes, player) of the position the device root stands for
        self._last_board = None

    # ------------------------------------------------------------------ reference surface
    @property
    def nn(self):
        return self._nn

    @nn.setter
    def nn(self, nn):
        if self.eval_method != TreeEval.NEURAL:
            raise ValueError(f"Trying to set a neural network for the MCT but the evaluation method is {self.eval_method}")
        self._nn = nn
        self._hipnet = None
        self._engine = None
        self._root_key = None

    def get_stats(self):
        return self.n_rollouts, self.simulation_time

    def search(self, board, n_sim=None, compute_time=None):
        self.n_rollouts = 0
        start = time()
        if n_sim is None and compute_time is None:
            raise ValueError("MCT.search needs to have either n_sim or compute_time specified.")
        self._sync_device_root(board, n_sim)
        if n_sim is not None:
            self._ensure_room(n_sim)
            self._engine.search(n_sim)
            self.n_rollouts = n_sim
        else:
            chunk = 1 if self.eval_method == TreeEval.NEURAL else 8
            while time() - start < compute_time:
                self._ensure_room(chunk)
                self._engine.search(chunk)
                self.n_rollouts += chunk
        self.simulation_time = time() - start

    def get_prior_probs(self):
        a, _, _, p, _ = self._children()
        if self.eval_method != TreeEval.NEURAL:
            return {_move_of(self._last_board, int(ai)): None for ai in a}
        return {_move_of(self._last_board, int(ai)): float(pi) for ai, pi in zip(a, p)}

    def get_action_probs(self, board, temp=0):
        a, n, _, _, _ = self._children()
        counts = {_move_of(board, int(ai)): int(ni) for ai, ni in zip(a, n)}
        if len(counts) == 0:
            return {board.pass_move: 1.}
        if temp == 0:
            best, _ = fair_max(counts.items(), key=lambda kv: kv[1])
            return {best: 1}, counts
        powered = {m: c ** (1. / temp) for m, c in counts.items()}
        total = sum(powered.values())
        return {m: v / total for m, v in powered.items()}, counts

    def change_root(self, move):
        if self._engine is None or self._root_key is None:
            return  # no device tree yet: the next search starts from the board it is given
        b = self._last_board.clone()
        b.play_move(move)  # raises ValueError for an illegal move, like the engine would
        self._engine.play([_action_of(self._last_board, move)])
        self._last_board = b
        self._root_key = (b.grid.astype(np.int8).tobytes(), int(b.player))

    # ------------------------------------------------------------------ device tree
    def _children(self):
        if self._engine is None:
            return [], [], [], [], 0
        return self._engine.root_children(0)

    def _sync_device_root(self, board, n_sim=None):
        from .engine import EVAL_NET, EVAL_ROLLOUT, NOISE_OFF, NOISE_PHILOX, TIE_RANDOM, SelfPlayEngine
        neural = self.eval_method == TreeEval.NEURAL
        if neural and self._nn is None:
            raise ValueError("The MCT has no neural network to evaluate positions with.")
        H, W = board.grid.shape
        if self._engine is not None and self._engine_board != (board.game, H, W):
            # the device storage was carried over a reset() from a game on another board (players.py keeps the engine); the
            # reference's reset() yields a tree usable on any board: rebuild rather than search with the wrong rules
            self._engine.close()
            self._engine, self._root_key = None, None
            if self._hipnet is not None and (self._hipnet.H, self._hipnet.W) != (H, W):
                self._hipnet = None
        if self._engine is None:
            if neural and self._hipnet is None:
                self._hipnet = self._nn.to_hip(max_batch=16)
            noisy = self.dirichlet_alpha is not None and self.dirichlet_epsilon is not None
            self._engine = SelfPlayEngine(_GAME_IDS[board.game], H, W, n_slots=1, n_sim=1, net=self._hipnet,
                                          dirichlet_alpha=self.dirichlet_alpha, dirichlet_epsilon=self.dirichlet_epsilon,
                                          temp_max_step=-1, temp_min_step=0, tie_mode=TIE_RANDOM,
                                          noise_mode=NOISE_PHILOX if noisy else NOISE_OFF,
                                          evaluator=EVAL_NET if neural else EVAL_ROLLOUT,
                                          seed=self._seed, sample_capacity=4 * H * W + 16, max_plies=4 * H * W + 16,
                                          # one slot: the pools start small and grow on demand (_ensure_room)
                                          node_capacity=1 << 14)
            self._engine_board = (board.game, H, W)
            self._plies = 0
        key = (board.grid.astype(np.int8).tobytes(), int(board.player))
        if key != self._root_key:  # tree restarted from an unexplored state (mcts.py:124-125, 231-233)
            self._engine.set_roots(board.grid.astype(np.int8)[None], np.array([board.player], np.int8),
                                   game_ids=np.array([np.random.randint(0, 2**31 - 1)], np.uint32))
            self._root_key = key
            self._used_bound = 1  # a fresh tree: the root
        self._last_board = board.clone()

    _MAX_POOL = 1 << 24  # nodes per pool (512 MiB): beyond it the engine reports AZ_ECAPACITY

    def _ensure_room(self, n_sim):
        """the reference's tree grows without bound while search() is called again and again on one root (mcts.py:226-269);
        the device pools have a size: before a search of n_sim simulations (each allocates at most one node's children, <= A <= 65
        nodes) the pools are re-allocated when what is left could run out.  The tree is kept (az_engine_grow_pools).

        The device is only asked for the node count (a stream synchronisation + a blocking copy) when a HOST-side upper bound --
        the last count read plus 66 nodes per simulation since -- nears the capacity: a compute_time search of one simulation per
        chunk would otherwise pay that round trip per simulation.  A tree that cannot fit the largest pool is refused up front."""
        eng = self._engine
        cap = eng.cfg.node_capacity
        self._used_bound = getattr(self, "_used_bound", cap) + 66 * n_sim
        if self._used_bound + 66 <= cap:
            return
        need = eng.nodes_used(0) + 66 * n_sim + 66
        self._used_bound = need - 66
        if need > cap:
            if need > self._MAX_POOL:
                raise MemoryError(f"MCT.search: the tree would need {need} nodes, beyond the device pool limit of {self._MAX_POOL} "
                                  f"(the search has run {need // 66} simulations' worth of expansions on one root without a move)")
            eng.grow_pools(min(self._MAX_POOL, max(2 * cap, need + 66 * 1024)))  # one growth covers many chunks
