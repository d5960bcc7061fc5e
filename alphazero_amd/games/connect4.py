"""Connect4 plugin: Connect4Config, Connect4Net (connect4.py:17-48, 333-445).  Board: see boards.py."""
from dataclasses import dataclass

import numpy as np

from ..base import Config
from ._convnet import ConvPolicyValueNet


@dataclass
class Connect4Config(Config):
    game: str = "connect4"
    board_width: int = 7
    board_height: int = 6
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


class Connect4Net(ConvPolicyValueNet):
    CONFIG = Connect4Config

    def __init__(self, board_width=None, board_height=None, device=None, config=None):
        super().__init__()
        if config is not None:
            board_width, board_height, device = config.board_width, config.board_height, config.device
        self.width, self.height = board_width, board_height
        self.device = self.get_torch_device(device)
        if self.width < 4 or self.height < 4:
            raise ValueError(f"Borad size must be at least 4x4, got {self.width}x{self.height}")
        self._build(self.width, self.height, 64, 32, self.width)

    def hip_shape(self):
        return 1, self.height, self.width

    def get_normalized_probs(self, probs, legal_moves):
        mask = np.zeros(self.action_size, dtype=bool)
        mask[legal_moves] = True
        total = np.sum(probs[mask])
        if total < 1e-6:
            print(f"The sum of the probabilities of the {len(legal_moves)} legal moves is {total}")
            return {m: 1 / len(legal_moves) for m in legal_moves}
        return {m: probs[m] / total for m in legal_moves}

    def to_neural_output(self, move_probs):
        pi = np.zeros(self.action_size)
        for move, p in move_probs.items():
            pi[move] = p
        return pi

    def reflect_neural_output(self, neural_output, axis):
        # the board only has a left-right symmetry: `axis` is ignored (connect4.py:437-445)
        if neural_output.size != self.action_size:
            raise ValueError(f"Neural output should have size {self.action_size}, but has size {neural_output.size}")
        return np.flip(neural_output)
