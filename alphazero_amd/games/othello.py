"""Othello plugin: OthelloConfig, OthelloNet (othello.py:17-47, 306-450).  OthelloBoard: see boards.py."""
from dataclasses import dataclass

import numpy as np

from ..base import Config
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
