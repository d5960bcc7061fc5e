"""The reference's benchmarking idiom (timers.py:11-151): wall time per self-play game and per optimisation step.

`SelfPlayTimer` and `NeuralTimer` keep the reference's constructors, methods and return values, so a script written against them
runs unchanged; what they time is this package's path -- the single-game `AlphaZeroPlayer` (tree and network on the GPU) and the
module's PyTorch forward / backward.  Two additions time the shapes the MI355X engine is built for:
`SelfPlayTimer.timeit_batched` (all games at once on the lock-step engine) and `NeuralTimer.timeit_hip` (the hand-written
training step, csrc/az_train.hip).
"""
from time import time

import numpy as np
import torch

from .games.registers import BOARDS_REGISTER, CONFIGS_REGISTER, NETWORKS_REGISTER
from .players import AlphaZeroPlayer


class SelfPlayTimer:
    """average duration of a self-play game (timers.py:11-76); data augmentation is not part of it, as in the reference"""

    def __init__(self, game, config=None):
        self.game = game
        self.config = CONFIGS_REGISTER[game]() if config is None else config
        self.board = BOARDS_REGISTER[game](config=self.config)
        self.nn = NETWORKS_REGISTER[game](config=self.config, device=self.config.device)
        self.az_player = AlphaZeroPlayer(n_sim=self.config.simulations, compute_time=self.config.compute_time, nn=self.nn, verbose=False)

    def self_play(self):
        """one game of the player against itself -> (seconds, number of moves) (timers.py:32-51)"""
        self.board.reset()
        self.az_player.reset()
        n_moves = 0
        start = time()
        while not self.board.is_game_over():
            move = self.az_player.get_move(self.board)[0]
            self.board.play_move(move)
            self.az_player.apply_move(move, player=-self.board.player)
            n_moves += 1
        return time() - start, n_moves

    def timeit(self, n_episodes=None):
        """-> (mean seconds per game, mean moves per game) over n_episodes games (default: config.episodes), printed as the reference does"""
        n = n_episodes if n_episodes is not None else self.config.episodes
        runs = [self.self_play() for _ in range(n)]
        t, s = np.array([r[0] for r in runs]), np.array([r[1] for r in runs])
        print(f"Average time to complete a self-play game: {t.mean():.2f} (+-{t.std():.2f}) seconds | {s.mean():.2f} (+-{s.std():.2f}) steps")
        return t.mean(), s.mean()

    def timeit_batched(self, n_games, seed=0):
        """the same n_games played concurrently on the lock-step engine (what AlphaZeroTrainer.self_play runs) ->
        (seconds per game = wall time / n_games, mean moves per game).  The first call also builds the engine; call twice to time
        the steady state."""
        from .engine import SelfPlayEngine
        if getattr(self, "_engine", None) is None or self._engine.cfg.n_slots != n_games:
            gid, H, W = self.nn.hip_shape()
            self._hipnet = self.nn.to_hip(max_batch=n_games)
            c = self.config
            self._engine = SelfPlayEngine(gid, H, W, n_slots=n_games, n_sim=c.simulations, net=self._hipnet, dirichlet_alpha=c.dirichlet_alpha,
                                          dirichlet_epsilon=c.dirichlet_epsilon, temp_max_step=c.temp_max_step, temp_min_step=c.temp_min_step,
                                          seed=seed)
            self._waves = 0
        torch.cuda.synchronize()
        start = time()
        smp = self._engine.run(n_games, first_game_id=self._waves * n_games)
        torch.cuda.synchronize()
        dt = time() - start
        self._waves += 1
        return dt / n_games, smp["z"].shape[0] / n_games


class NeuralTimer:
    """average duration of one optimisation step on a fake batch (timers.py:79-151)"""

    def __init__(self, game, config=None):
        self.game = game
        self.config = CONFIGS_REGISTER[game]() if config is None else config
        self.board = BOARDS_REGISTER[game](config=self.config)
        self.nn = NETWORKS_REGISTER[game](config=self.config, device=self.config.device)

    def get_fake_batch(self):
        """(input [B, h, w], pi [B, A], v [B]) of standard-normal noise on config.device (timers.py:93-118)"""
        c = self.config
        if hasattr(c, "board_size"):
            shape = (c.batch_size, c.board_size, c.board_size)
        elif hasattr(c, "board_width") and hasattr(c, "board_height"):
            shape = (c.batch_size, c.board_height, c.board_width)
        else:
            raise AttributeError("Board size/width/height not found in the config...")
        dev = c.device
        return torch.randn(*shape).to(dev), torch.randn(c.batch_size, self.board.get_action_size()).to(dev), torch.randn(c.batch_size).to(dev)

    def timeit(self, n_batches=None):
        """n_batches steps of the reference's timing loop (plain SGD(lr), loss = sum (v - z)^2 - sum pi log p, timers.py:120-151) on the
        module's PyTorch forward / backward -> mean seconds per step"""
        n = n_batches if n_batches is not None else 10
        opt = torch.optim.SGD(self.nn.parameters(), lr=self.config.learning_rate)
        self.nn.train()
        durations = []
        for _ in range(n):
            x, pi, z = self.get_fake_batch()
            start = time()
            opt.zero_grad()
            log_probs, v = self.nn(x)
            loss = torch.sum((v - z) ** 2) - torch.sum(pi * log_probs)
            loss.backward()
            opt.step()
            if x.is_cuda:
                torch.cuda.synchronize()
            durations.append(time() - start)
        d = np.array(durations)
        print(f"Average time to optimize with a batch: {d.mean():.2f} (+-{d.std():.2f}) seconds")
        return d.mean()

    def timeit_hip(self, n_batches=1000, n_samples=4096):
        """n_batches steps of the trainer's optimisation (trainer.py:346-366: momentum 0.9, weight decay 1e-4, the module's dropout) on the
        hand-written HIP step over n_samples device-resident fake samples -> mean seconds per step"""
        from .train_step import HipTrainStep, supports
        c = self.config
        if not supports(self.nn, c.batch_size):
            raise ValueError(f"the hand-written step does not cover {type(self.nn).__name__} at batch size {c.batch_size}")
        gid, H, W = self.nn.hip_shape()
        A = self.board.get_action_size()
        state = torch.randint(-1, 2, (n_samples, H, W), dtype=torch.int8, device="cuda")
        pi = torch.softmax(torch.randn(n_samples, A, device="cuda"), dim=1)
        z = torch.randint(-1, 2, (n_samples,), dtype=torch.int8, device="cuda")
        perm = torch.randint(0, n_samples, (n_batches * c.batch_size,), dtype=torch.int64, device="cuda")
        step = HipTrainStep(self.nn, max_batch=c.batch_size)
        step.load(self.nn)
        step.begin(c.learning_rate, 0.9, 1e-4, float(getattr(self.nn, "dropout", 0.0)), seed=0)
        lp, lv = torch.zeros(n_batches, device="cuda"), torch.zeros(n_batches, device="cuda")
        step.steps(state, pi, z, perm, min(20, n_batches), c.batch_size, lp, lv)  # warm-up: first step eager, graphs captured
        step.check()
        start = time()
        step.steps(state, pi, z, perm, n_batches, c.batch_size, lp, lv)
        step.check()
        dt = (time() - start) / n_batches
        step.close()
        print(f"Average time to optimize with a batch: {1e3 * dt:.3f} ms (hand-written HIP step, batch {c.batch_size})")
        return dt
