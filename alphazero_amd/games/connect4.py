"""Connect4 plugin: Connect4Config, Connect4Board, Connect4Net (connect4.py:17-445)."""
from dataclasses import dataclass

import numpy as np

from ..base import Board, Config
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


class Connect4Board(Board):
    """height x width Connect4 (connect4.py:51-258): a move is a column, discs fall to the lowest free row
    (row index height-1 is the bottom), four aligned discs win, a full board is a draw, no pass."""
    CONFIG = Connect4Config

    def __init__(self, width=None, height=None, grid=None, player=1, display_dir=None, display_mode=None, config=None):
        super().__init__(display_dir=display_dir, display_mode=display_mode)
        self.game = "connect4"
        if config is not None:
            self.width, self.height = config.board_width, config.board_height
            self.reset()
        else:
            self.width, self.height = width, height
            self.grid = grid if grid is not None else np.zeros((self.height, self.width))
            self.free_rows = self._count_free_rows()
            self.player = player
            self.max_moves = self.width * self.height
        if self.width < 4 or self.height < 4:
            raise ValueError(f"Borad size must be at least 4x4, got {self.width}x{self.height}")

    def reset(self):
        self.grid = np.zeros((self.height, self.width))
        self.free_rows = self._count_free_rows()
        self.player = 1
        self.max_moves = self.width * self.height

    def _count_free_rows(self):
        return np.sum(self.grid == 0, axis=0)

    def __str__(self):
        return f"{type(self).__name__}{self.width}x{self.height}"

    def clone(self):
        return Connect4Board(width=self.width, height=self.height, grid=self.grid.copy(), player=self.player,
                             display_dir=self.display_dir)

    def get_board_shape(self):
        return self.grid.shape

    def get_n_cells(self):
        return np.prod(self.get_board_shape())

    def get_action_size(self):
        return self.width

    def get_score(self):
        return np.sum(self.player * self.grid).astype(int)

    def is_legal_move(self, move, player=None):
        if not 0 <= move < self.width:
            raise ValueError(f"Column index must be in [0, {self.width-1}], got {move}")
        return bool(self.free_rows[move] > 0)

    def get_moves(self, player=None):
        return list(np.where(self.free_rows > 0)[0])

    def get_random_move(self, player=None):
        moves = self.get_moves(player)
        return moves[np.random.choice(len(moves))]

    def play_move(self, move):
        if not self.is_legal_move(move):
            raise ValueError(f"Illegal move {move} for player {self.player}")
        row = self.free_rows[move] - 1
        if self.grid[row][move] != 0:
            raise ValueError(f"Cell ({row},{move}) is not empty...")
        self.grid[row][move] = self.player
        self.free_rows[move] -= 1
        self.player = -self.player

    def _run_of_four(self, line):
        """first side to complete four in a row along `line` (scanned in order), else 0"""
        run, last = 0, 0
        for v in line:
            v = int(v)
            run = run + 1 if (v != 0 and v == last) else (1 if v != 0 else 0)
            last = v
            if run == 4:
                return v
        return 0

    def _status(self):
        """1 / -1 winner, 0 draw, None while the game goes on; lines are scanned in the reference's order:
        down-right diagonals, down-left diagonals, rows, columns (connect4.py:212-245)"""
        g, H, W = self.grid, self.height, self.width
        starts = [(i, 0) for i in range(1, H - 3)] + [(0, i) for i in range(W - 3)]
        for grid in (g, np.fliplr(g)):
            for (r, c) in starts:
                k = min(H - r, W - c)
                s = self._run_of_four(grid[r + i][c + i] for i in range(k))
                if s:
                    return s
        for r in range(H):
            s = self._run_of_four(g[r])
            if s:
                return s
        for c in range(W):
            s = self._run_of_four(g[:, c])
            if s:
                return s
        return 0 if np.sum(self.free_rows) == 0 else None

    def is_game_over(self):
        return self._status() is not None

    def get_winner(self):
        s = self._status()
        if s is None:
            raise ValueError("Game is not over yet...")
        return s
