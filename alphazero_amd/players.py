"""Players (players.py:76-247).  HumanPlayer (interactive stdin) is out of scope."""
from time import sleep

import numpy as np

from .base import Player
from .mcts import MCT
from .utils import fair_max


class RandomPlayer(Player):
    def __init__(self, lock_time=None, verbose=False):
        super().__init__(verbose=verbose)
        self.lock_time = lock_time

    def clone(self):
        return RandomPlayer(lock_time=self.lock_time, verbose=self.verbose)

    def get_move(self, board, temp=None):
        if self.lock_time is not None:
            sleep(self.lock_time)
        return board.get_random_move(), None, None, None


class GreedyPlayer(Player):
    """picks the move with the best immediate score (players.py:97-123)"""

    def clone(self):
        return GreedyPlayer(verbose=self.verbose)

    def get_move(self, board, temp=None):
        scored = {}
        for move in board.get_moves():
            after = board.clone()
            after.play_move(move)
            scored[move] = -after.get_score()
        return fair_max(scored.items(), key=lambda kv: kv[1])[0], None, None, None


class MCTSPlayer(Player):
    def __init__(self, n_sim=None, compute_time=None, verbose=False):
        super().__init__(verbose=verbose)
        self.n_sim, self.compute_time = n_sim, compute_time
        if self.n_sim is None and self.compute_time is None:
            raise ValueError("MCTSPlayer needs to have either n_sim or compute_time specified.")
        if self.n_sim is not None and self.compute_time is not None:
            raise ValueError("MCTSPlayer can't have both n_sim and compute_time specified.")
        self.mct = MCT()

    def clone(self):
        return MCTSPlayer(n_sim=self.n_sim, compute_time=self.compute_time, verbose=self.verbose)

    def reset(self):
        old = self.mct
        self.mct = MCT()
        # a reset drops the tree, not the device storage behind it (one engine slot: node pools, sample buffers)
        self.mct._engine, self.mct._engine_board = old._engine, old._engine_board

    def apply_move(self, move, player=None):
        self.mct.change_root(move)

    def get_move(self, board, temp=0):
        if board.is_game_over():
            raise ValueError(f"{self}.get_move was called with a board in game over state...")
        self.mct.search(board=board, n_sim=self.n_sim, compute_time=self.compute_time)
        action_probs, visit_counts = self.mct.get_action_probs(board, temp)
        items = list(action_probs.items())
        if len(items) == 1:
            best = items[0][0]
        else:
            best = items[np.random.choice(len(items), p=[p for _, p in items])][0]
        return best, action_probs, visit_counts, self.mct.get_prior_probs()

    def get_stats_after_move(self):
        n_rollouts, simulation_time = self.mct.get_stats()
        return {"n_rollouts": n_rollouts, "time": simulation_time}


class AlphaZeroPlayer(MCTSPlayer):
    def __init__(self, n_sim=None, compute_time=None, nn=None, dirichlet_alpha=None, dirichlet_epsilon=None, verbose=False):
        super().__init__(n_sim=n_sim, compute_time=compute_time, verbose=verbose)
        self.mct = MCT(eval_method="neural", nn=nn, dirichlet_alpha=dirichlet_alpha, dirichlet_epsilon=dirichlet_epsilon)

    def clone(self):
        return AlphaZeroPlayer(n_sim=self.n_sim, compute_time=self.compute_time,
                               nn=self.mct.nn.clone() if self.mct.nn is not None else None,
                               dirichlet_alpha=self.mct.dirichlet_alpha, dirichlet_epsilon=self.mct.dirichlet_epsilon,
                               verbose=self.verbose)

    def reset(self):
        old = self.mct
        self.mct = MCT(eval_method="neural", nn=old.nn, dirichlet_alpha=old.dirichlet_alpha,
                       dirichlet_epsilon=old.dirichlet_epsilon)
        # keep the uploaded weights and the device tree storage: a reset only drops the tree
        self.mct._hipnet, self.mct._engine, self.mct._engine_board = old._hipnet, old._engine, old._engine_board
        if self.mct._engine is not None:
            self.mct._plies = 0


PLAYERS_SET = {"human", "random", "greedy", "mcts", "alphazero"}
PLAYERS_REGISTER = {"random": RandomPlayer, "greedy": GreedyPlayer, "mcts": MCTSPlayer, "alphazero": AlphaZeroPlayer}
