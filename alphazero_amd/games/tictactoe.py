"""TicTacToe plugin: TicTacToeConfig, TicTacToeNet (tictactoe.py:17-47, 262-367).  Board: see boards.py."""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..base import Config, PolicyValueNetwork
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
