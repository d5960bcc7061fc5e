"""Othello plugin: OthelloConfig, OthelloBoard, OthelloNet (othello.py:17-450)."""
from dataclasses import dataclass

import numpy as np

from ..base import Board, Config
from . import _bitrules as R
from ._convnet import ConvPolicyValueNet, uniform_or_normalised


@dataclass
class OthelloConfig(Config):
    game: str = "othello"
    board_size: int = 6
    simulations: int = 100
    compute_time: float = None
    dirichlet_alpha: float = 0.03
    dirichlet_epsilon: float = 0.25
    temp_scheduler_type: str = "linear"
    temp_max_step: int = 4
    temp_min_step: int = 4
    iterations: int = 30
    episodes: int = 200
    epochs: int = 10
    batch_size: int = 64
    learning_rate: float = 0.01
    data_augmentation: bool = True
    device: str = "cpu"
    eval_opponent: str = "mcts"
    eval_episodes: int = 40
    do_eval: bool = True
    save: bool = True
    push: bool = False
    save_checkpoints: bool = True
    push_checkpoints: bool = False


class OthelloNet(ConvPolicyValueNet):
    CONFIG = OthelloConfig

    def __init__(self, n=None, device=None, config=None):
        super().__init__()
        if config is not None:
            n, device = config.board_size, config.device
        self.n = n
        self.device = self.get_torch_device(device)
        if self.n is None:
            raise ValueError("The board size must be a positive and even integer like 4, 6 or 8.")
        self._build(n, n, 1024, 512, n * n + 1)

    def hip_shape(self):
        return 0, self.n, self.n

    def _index(self, move):
        return self.n * self.n if tuple(move) == (self.n, self.n) else move[0] * self.n + move[1]

    def get_normalized_probs(self, probs, legal_moves):
        return uniform_or_normalised({m: probs[self._index(m)] for m in legal_moves}, len(legal_moves))

    def to_neural_output(self, move_probs):
        pi = np.zeros(self.action_size)
        for move, p in move_probs.items():
            pi[self._index(move)] = p
        return pi

    def _board_part(self, neural_output):
        if neural_output.size != self.action_size:
            raise ValueError(f"Neural output should have size {self.action_size}, but has size {neural_output.size}")
        return neural_output[:-1].reshape(self.n, self.n), neural_output[-1:]

    def reflect_neural_output(self, neural_output, axis):
        board, tail = self._board_part(neural_output)
        return np.concatenate([np.flip(board, axis=axis).reshape(-1), tail]).astype(neural_output.dtype)

    def rotate_neural_output(self, neural_output, angle):
        board, tail = self._board_part(neural_output)
        return np.concatenate([np.rot90(board, k=angle // 90).reshape(-1), tail]).astype(neural_output.dtype)


class OthelloBoard(Board):
    """n x n Othello (othello.py:50-229).  `grid` holds 1 (black, moves first), -1 (white), 0; every query is
    answered from the current `grid`/`player`, so callers may edit them like they do with the reference."""
    CONFIG = OthelloConfig

    def __init__(self, n=None, grid=None, player=1, display_dir=None, display_mode=None, config=None):
        super().__init__(display_dir=display_dir, display_mode=display_mode)
        self.game = "othello"
        if config is not None:
            self.n = config.board_size
            self.reset()
        else:
            self.n = n
            self.grid = grid if grid is not None else self._start_grid()
            self.player = player
            self.pass_move = self.get_board_shape()
            self.max_moves = self.n * self.n - 4
        if self.n % 2 != 0:
            raise ValueError(f"Board size must be even but got n={self.n}")

    def _start_grid(self):
        g = np.zeros((self.n, self.n))
        h = self.n // 2
        g[h - 1][h - 1] = g[h][h] = 1
        g[h - 1][h] = g[h][h - 1] = -1
        return g

    def reset(self):
        self.grid = self._start_grid()
        self.player = 1
        self.pass_move = self.get_board_shape()
        self.max_moves = self.n * self.n - 4

    def __str__(self):
        return f"{type(self).__name__}{self.n}"

    def clone(self):
        return OthelloBoard(n=self.n, grid=self.grid.copy(), player=self.player, display_dir=self.display_dir)

    def get_board_shape(self):
        return self.grid.shape

    def get_n_cells(self):
        return np.prod(self.get_board_shape())

    def get_action_size(self):
        return self.n * self.n + 1

    def get_score(self):
        return np.sum(self.player * self.grid).astype(int)

    def _sides(self, player):
        player = player if player in (-1, 1) else self.player
        p1, m1 = R.pack(self.grid)
        return (p1, m1) if player > 0 else (m1, p1)

    def _legal_bits(self, player=None):
        own, opp = self._sides(player)
        return R.othello_legal(own, opp, R.valid_mask(self.n, self.n))

    def is_legal_move(self, move, player=None):
        legal = self._legal_bits(player)
        if tuple(move) == tuple(self.pass_move):
            return legal == 0
        r, c = move
        if not (0 <= r < self.n and 0 <= c < self.n):
            return False
        return bool((legal >> (r * 8 + c)) & 1)

    def get_moves(self, player=None):
        """legal cells, or [pass_move], in the reference's order: it collects them row-major into a set and returns list(set)
        (othello.py:176-189), so a caller that seeds np.random sees the same get_random_move picks here as there"""
        moves = set()
        for b in R.bits(self._legal_bits(player)):
            moves.add((b >> 3, b & 7))
        if len(moves) == 0:
            moves.add(self.pass_move)
        return list(moves)

    def get_random_move(self, player=None):
        moves = self.get_moves(player)
        return moves[np.random.choice(len(moves))]

    def play_move(self, move):
        if not self.is_legal_move(move):
            raise ValueError(f"Illegal move {move} for player {self.player}")
        if tuple(move) == (self.n, self.n):
            self.player = -self.player
            return
        own, opp = self._sides(None)
        mv = 1 << (move[0] * 8 + move[1])
        for b in R.bits(R.othello_flips(own, opp, mv) | mv):
            self.grid[b >> 3][b & 7] = self.player
        self.player = -self.player

    def is_game_over(self):
        return self._legal_bits(1) == 0 and self._legal_bits(-1) == 0

    def get_winner(self):
        if not self.is_game_over():
            raise ValueError("Game is not over yet...")
        score = self.get_score()
        if score == 0:
            return 0
        return self.player if score > 0 else -self.player
