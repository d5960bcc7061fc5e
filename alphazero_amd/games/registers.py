"""Name-keyed registries (games/registers.py:11-50)."""
from ..base import DataTransf, MoveFormat, dotdict
from .connect4 import Connect4Board, Connect4Config, Connect4Net
from .othello import OthelloBoard, OthelloConfig, OthelloNet
from .tictactoe import TicTacToeBoard, TicTacToeConfig, TicTacToeNet

GAMES_SET = {"othello", "tictactoe", "connect4"}
CONFIGS_REGISTER = {"othello": OthelloConfig, "tictactoe": TicTacToeConfig, "connect4": Connect4Config}
BOARDS_REGISTER = {"othello": OthelloBoard, "tictactoe": TicTacToeBoard, "connect4": Connect4Board}
NETWORKS_REGISTER = {"othello": OthelloNet, "tictactoe": TicTacToeNet, "connect4": Connect4Net}
MOVE_FORMATS_REGISTER = {"othello": MoveFormat.ROW_COL, "tictactoe": MoveFormat.ROW_COL, "connect4": MoveFormat.COL}
_ROTATIONS = [DataTransf.ROTATE_90, DataTransf.ROTATE_180, DataTransf.ROTATE_270]
DATA_AUGMENT_STRATEGIES = {
    "othello": dotdict({"reflection": DataTransf.REFLECT_H, "rotations": list(_ROTATIONS)}),
    "connect4": dotdict({"reflection": DataTransf.REFLECT_H, "rotations": []}),
    "tictactoe": dotdict({"reflection": DataTransf.REFLECT_H, "rotations": list(_ROTATIONS)}),
}
