"""Plugin contracts of the reference (alphazero/base.py:17-421), kept call-compatible:
Config, Board, Player, PolicyValueNetwork, TemperatureScheduler and the small enums.

Only what the self-play path and its callers touch is carried over; display/plotting helpers are
out of scope.
"""
import copy
import dataclasses
import enum
import json
import os

import numpy as np
import torch
from torch import nn

DEFAULT_MODELS_PATH = os.path.join(os.getcwd(), "models/")
DEFAULT_CONFIGS_PATH = os.path.join(os.getcwd(), "configs/")  # utils.py:12 (there: beside the package sources)


class dotdict(dict):
    """dict with attribute access (utils.py:21-25)"""
    __getattr__ = dict.__getitem__


class Action:
    """type marker for moves: (row, col) tuples or a column int"""


class _ValueEnum(enum.Enum):
    @classmethod
    def to_dict(cls):
        return {m.value: m for m in cls}


class MoveFormat(_ValueEnum):
    ROW_COL = "row_col"
    ROW = "row"
    COL = "col"


class TreeEval(_ValueEnum):
    ROLLOUT = "rollout"
    NEURAL = "neural"


class DisplayMode(_ValueEnum):
    """base.py:45-48; the modes exist so that configs and constructor calls written for the reference load; rendering itself is out of scope"""
    HUMAN = "human"
    PIXEL = "pixel"


class DataTransf(_ValueEnum):
    REFLECT_H = "reflection_horizontal"
    REFLECT_V = "reflection_vertical"
    ROTATE_90 = "rotation_90"
    ROTATE_180 = "rotation_180"
    ROTATE_270 = "rotation_270"


@dataclasses.dataclass
class Config:
    """base.py:60-92 -- same fields and defaults so the reference's JSON configs load unchanged"""
    game: str = None
    simulations: int = None
    compute_time: float = None
    dirichlet_alpha: float = 0.03
    dirichlet_epsilon: float = 0.25
    temp_scheduler_type: str = "linear"
    temp_max_step: int = 15
    temp_min_step: int = 20
    iterations: int = None
    episodes: int = None
    epochs: int = None
    batch_size: int = None
    learning_rate: float = None
    data_augmentation: bool = False
    device: str = None
    eval_opponent: str = "mcts"
    eval_episodes: int = 10
    do_eval: bool = False
    save: bool = True
    push: bool = False
    save_checkpoints: bool = True
    push_checkpoints: bool = False

    def to_dict(self):
        return dotdict(copy.deepcopy(dataclasses.asdict(self)))


class Board:
    """base.py:95-225.  Attributes: game, grid (np.ndarray of {-1,0,1}), player, pass_move, max_moves."""
    CONFIG = Config

    def __init__(self, display_dir=None, display_mode=None):
        self.display_dir, self.display_mode = display_dir, display_mode
        self.game = None
        self.grid = None
        self.player = None
        self.pass_move = None
        self.max_moves = None

    def __str__(self):
        return type(self).__name__

    # the abstract contract (base.py:121-198): same names, same parameters; concrete boards live in games/*
    def reset(self):
        raise NotImplementedError

    def clone(self):
        raise NotImplementedError

    def get_board_shape(self):
        raise NotImplementedError

    def get_n_cells(self):
        raise NotImplementedError

    def get_action_size(self):
        raise NotImplementedError

    def get_score(self):
        raise NotImplementedError

    def is_legal_move(self, move, player):
        raise NotImplementedError

    def get_moves(self, player):
        raise NotImplementedError

    def get_random_move(self, player):
        raise NotImplementedError

    def play_move(self, move):
        raise NotImplementedError

    def is_game_over(self):
        raise NotImplementedError

    def get_winner(self):
        raise NotImplementedError

    def human_display(self, *args, **kwargs):
        raise NotImplementedError("board rendering is out of scope of the self-play engine")

    def pixel_display(self, *args, **kwargs):
        raise NotImplementedError("board rendering is out of scope of the self-play engine")

    def display(self, show_indexes=True, infos=None, filename=None, mode=None):
        """base.py:199-225 dispatches to human_display / pixel_display: rendering is out of scope here"""
        raise NotImplementedError("board rendering is out of scope of the self-play engine")


class Player:
    """base.py:228-264"""

    def __init__(self, verbose=False):
        self.verbose = verbose

    def __str__(self):
        return type(self).__name__

    def clone(self):
        raise NotImplementedError

    def reset(self):
        pass

    def apply_move(self, move, player=None):
        pass

    def get_move(self, board, temp=None):
        raise NotImplementedError

    def get_stats_after_move(self):
        return {}


class PolicyValueNetwork(nn.Module):
    """base.py:267-397.  forward() -> (log-probabilities, tanh value); evaluate() flips the board to the
    side to move and the value back to the absolute frame."""
    CONFIG = Config

    def __str__(self):
        return type(self).__name__

    def clone(self):
        return copy.deepcopy(self)

    def get_parameters_count(self):
        return sum(p.numel() for p in self.parameters())

    def save_model(self, model_name, model_path=None, verbose=False):
        stem = model_name.split(".")[0]
        folder = os.path.join(DEFAULT_MODELS_PATH, stem) if model_path is None else model_path
        os.makedirs(folder, exist_ok=True)
        torch.save(self.state_dict(), os.path.join(folder, f"{stem}.pt"))
        if verbose:
            print(f"{self} saved in: {folder}'")

    @classmethod
    def from_pretrained(cls, model_name, models_path=None, verbose=False):
        stem = model_name.split(".")[0]
        folder = os.path.join(DEFAULT_MODELS_PATH if models_path is None else models_path, stem)
        cfg_file, pt_file = os.path.join(folder, "config.json"), os.path.join(folder, f"{stem}.pt")
        if not os.path.isfile(cfg_file):
            raise ValueError(f"Config file not found: {cfg_file}")
        if not os.path.isfile(pt_file):
            raise ValueError(f"Model file not found: {pt_file}")
        with open(cfg_file) as f:
            model = cls(config=cls.CONFIG(**json.load(f)))
        model.load_state_dict(torch.load(pt_file))
        if verbose:
            print(f"{model} loaded from: {folder}'")
        return model

    @staticmethod
    def get_torch_device(device):
        if device is None:
            device = "cpu"
        elif device == "mps" and not torch.backends.mps.is_available():
            raise ValueError("MPS not available...")
        elif device == "cuda" and not torch.cuda.is_available():
            raise ValueError("CUDA not available...")
        return torch.device(device)

    def predict(self, input):
        self.eval()
        with torch.no_grad():
            log_p, v = self.forward(input)
        return torch.exp(log_p), v

    def evaluate(self, board):
        x = torch.tensor(board.player * board.grid, dtype=torch.float, device=self.device)
        p, v = self.predict(x)
        return p.cpu().numpy().reshape(-1), board.player * v.cpu().item()

    # the four hooks every concrete network supplies (base.py:370-397)
    def get_normalized_probs(self, probs, legal_moves):
        raise NotImplementedError

    def to_neural_output(self, move_probs):
        raise NotImplementedError

    def reflect_neural_output(self, neural_output, axis):
        raise NotImplementedError

    def rotate_neural_output(self, neural_output, angle):
        raise NotImplementedError

    # the device side: weights of this module on the HIP engine --------------------------------
    def hip_shape(self):
        """(game id, H, W) of the boards this network evaluates"""
        raise NotImplementedError

    def to_hip(self, max_batch=4096):
        """uploads the (eval-mode, BN-folded) weights to an alphazero_amd.engine.HipNet"""
        from .engine import HipNet
        gid, H, W = self.hip_shape()
        return HipNet(gid, H, W, self.state_dict(), max_batch=max_batch)


class TemperatureScheduler:
    """base.py:400-421"""

    def __init__(self, temp_max_step, temp_min_step, max_steps):
        self.temp_max_step, self.temp_min_step, self.max_steps = temp_max_step, temp_min_step, max_steps

    def compute_temperature(self, step):
        raise NotImplementedError

    def __getitem__(self, step):
        return self.compute_temperature(step)

    def __iter__(self):
        return (self.compute_temperature(s) for s in range(self.max_steps + 1))
