"""TicTacToe plugin: TicTacToeConfig, TicTacToeBoard, TicTacToeNet (tictactoe.py:17-367)."""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..base import Board, Config, PolicyValueNetwork
from ._convnet import uniform_or_normalised


@dataclass
class TicTacToeConfig(Config):
    game: str = "tictactoe"
    board_size: int = 3
    simulations: int = 100
    compute_time: float = None
    dirichlet_alpha: float = 0.03
    dirichlet_epsilon: float = 0.25
    temp_scheduler_type: str = "linear"
    temp_max_step: int = 2
    temp_min_step: int = 2
    iterations: int = 30
    episodes: int = 100
    epochs: int = 10
    batch_size: int = 64
    learning_rate: float = 0.01
    data_augmentation: bool = True
    device: str = "cpu"
    eval_opponent: str = "mcts"
    eval_episodes: int = 100
    do_eval: bool = True
    save: bool = True
    push: bool = False
    save_checkpoints: bool = True
    push_checkpoints: bool = False


class TicTacToeNet(PolicyValueNetwork):
    CONFIG = TicTacToeConfig

    def __init__(self, device=None, config=None):
        super().__init__()
        self.device = self.get_torch_device(config.device if config is not None else device)
        self.action_size = 9
        self.fc1 = nn.Linear(9, 9, device=self.device)
        self.fc2 = nn.Linear(9, 9, device=self.device)
        self.fc_probs = nn.Linear(9, self.action_size, device=self.device)
        self.fc_value = nn.Linear(9, 1, device=self.device)
        self.flatten = nn.Flatten()
        self.bn1 = nn.BatchNorm1d(9, device=self.device)
        self.bn2 = nn.BatchNorm1d(9, device=self.device)

    def hip_shape(self):
        return 2, 3, 3

    def forward(self, input):
        if input.ndim == 2:
            input = input.unsqueeze(0)
        x = self.flatten(input)
        x = F.relu(self.bn1(self.fc1(x)))
        x = F.relu(self.bn2(self.fc2(x)))
        return F.log_softmax(self.fc_probs(x), dim=1), torch.tanh(self.fc_value(x))

    def get_normalized_probs(self, probs, legal_moves):
        return uniform_or_normalised({m: probs[3 * m[0] + m[1]] for m in legal_moves}, len(legal_moves))

    def to_neural_output(self, move_probs):
        pi = np.zeros(9)
        for move, p in move_probs.items():
            pi[3 * move[0] + move[1]] = p
        return pi

    def _check(self, neural_output):
        if neural_output.size != self.action_size:
            raise ValueError(f"Neural output should have size {self.action_size}, but has size {neural_output.size}")
        return neural_output.reshape(3, 3)

    def reflect_neural_output(self, neural_output, axis):
        return np.flip(self._check(neural_output), axis=axis).flatten()

    def rotate_neural_output(self, neural_output, angle):
        return np.rot90(self._check(neural_output), k=angle // 90).flatten()


class TicTacToeBoard(Board):
    """3 x 3 TicTacToe (tictactoe.py:50-184)."""
    CONFIG = TicTacToeConfig

    def __init__(self, grid=None, player=1, display_dir=None, display_mode=None, config=None):
        super().__init__(display_dir=display_dir, display_mode=display_mode)
        self.game = "tictactoe"
        if config is not None:
            self.reset()
        else:
            self.grid = grid if grid is not None else np.zeros((3, 3))
            self.player = player
            self.max_moves = 9

    def reset(self):
        self.grid = np.zeros((3, 3))
        self.player = 1
        self.max_moves = 9

    def clone(self):
        return TicTacToeBoard(grid=self.grid.copy(), player=self.player, display_dir=self.display_dir)

    def get_board_shape(self):
        return self.grid.shape

    def get_n_cells(self):
        return np.prod(self.grid.shape)

    def get_action_size(self):
        return self.get_n_cells()

    @staticmethod
    def _alignments(a):
        return np.concatenate([a.sum(axis=1), a.sum(axis=0), [np.trace(a), np.trace(np.fliplr(a))]])

    def get_alignments_sums(self):
        return self._alignments(self.grid)

    def get_nb_free_cells_on_alignments(self, player=None):
        return self._alignments((self.grid == 0).astype(int))

    def get_score(self):
        """inf when the side to move can complete a line at once, else 0 (tictactoe.py:119-126)"""
        mine = self.player * self.get_alignments_sums()
        has_room = (self.get_nb_free_cells_on_alignments() > 0).astype(int)
        return float("inf") if 2 in mine * has_room else 0

    def is_legal_move(self, move, player=None):
        return bool(0 <= move[0] < 3 and 0 <= move[1] < 3 and self.grid[move[0]][move[1]] == 0)

    def get_moves(self, player=None):
        return [(cell[0], cell[1]) for cell in np.vstack(np.where(self.grid == 0)).T]

    def get_random_move(self, player=None):
        moves = self.get_moves(player)
        return moves[np.random.choice(len(moves))]

    def play_move(self, move):
        if not self.is_legal_move(move):
            raise ValueError(f"Illegal move {move} for player {self.player}")
        self.grid[move[0]][move[1]] = self.player
        self.player = -self.player

    def is_game_over(self):
        return bool(3 in np.abs(self.get_alignments_sums()) or np.all(self.grid != 0))

    def get_winner(self):
        if not self.is_game_over():
            raise ValueError("Game is not over yet...")
        sums = self.get_alignments_sums()
        return 1 if 3 in sums else (-1 if -3 in sums else 0)
