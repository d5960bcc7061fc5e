"""AlphaZeroTrainer with the reference's interface (trainer.py:30-572).  self_play runs on the HIP engine:
all episodes of an iteration are played concurrently in lock-step on the GPU instead of one after the other.
Timers, plotting and HF-hub pushes are out of scope."""
import json
import os

import numpy as np
import torch

from . import base
from .arena import Arena
from .base import DataTransf, dotdict
from .games.registers import BOARDS_REGISTER, CONFIGS_REGISTER, DATA_AUGMENT_STRATEGIES, GAMES_SET, NETWORKS_REGISTER
from .players import PLAYERS_REGISTER, PLAYERS_SET, AlphaZeroPlayer
from .schedulers import TEMP_SCHEDULERS


class Sample:
    """one training example (trainer.py:30-118)"""

    def __init__(self, state, pi, player, outcome=None, episode_idx=None, move_idx=None, transformation=None):
        self.state, self.pi, self.player, self.outcome = state, pi, player, outcome
        self.episode_idx, self.move_idx, self.transformation = episode_idx, move_idx, transformation

    def normalize(self):
        self.state = self.state * self.player
        self.outcome = self.outcome * self.player
        self.player = 1

    def _twin(self, state, pi, mode):
        tag = mode.value if self.transformation is None else f"{self.transformation}+{mode.value}"
        return Sample(state=state, pi=pi, player=self.player, outcome=self.outcome, episode_idx=self.episode_idx,
                      move_idx=self.move_idx, transformation=tag)

    def create_reflection_twin(self, reflection, mode):
        if mode not in (DataTransf.REFLECT_H, DataTransf.REFLECT_V):
            raise ValueError(f"Reflection mode must be either {DataTransf.REFLECT_H} or {DataTransf.REFLECT_V}.")
        axis = 1 if mode == DataTransf.REFLECT_H else 0
        return self._twin(np.flip(self.state.copy(), axis=axis), reflection(self.pi.copy(), axis=axis), mode)

    def create_rotation_twin(self, rotation, mode):
        angles = {DataTransf.ROTATE_90: 90, DataTransf.ROTATE_180: 180, DataTransf.ROTATE_270: 270}
        if mode not in angles:
            raise ValueError(f"Rotation mode must be either {DataTransf.ROTATE_90}, {DataTransf.ROTATE_180} or {DataTransf.ROTATE_270}.")
        return self._twin(np.rot90(self.state.copy(), k=angles[mode] // 90), rotation(self.pi.copy(), angle=angles[mode]), mode)


def augment(memory, nn, strategy):
    """symmetry augmentation of samples with move_idx >= 2 (trainer.py:275-284)"""
    extra = []
    for s in memory:
        if s.move_idx >= 2:
            mirrored = s.create_reflection_twin(nn.reflect_neural_output, mode=strategy.reflection)
            extra.append(mirrored)
            for rot in strategy.rotations:
                extra.append(s.create_rotation_twin(nn.rotate_neural_output, mode=rot))
                extra.append(mirrored.create_rotation_twin(nn.rotate_neural_output, mode=rot))
    return extra


class AlphaZeroTrainer:
    DEFAULT_EXP_NAME = "alphazero-undefined"

    def __init__(self, verbose=False, engine_slots=4096, seed=0, materialize_memory=True):
        self.game = self.config = self.board = self.nn = self.nn_twin = None
        self.az_player = self.temp_scheduler = self.data_augment_strategy = None
        self.memory = self.loss_values = self.eval_results = None
        self.verbose = verbose
        self.engine_slots, self.seed = engine_slots, seed
        self._engine = self._hipnet = None
        self.prev_nn = None
        self.device_samples = None  # the last self-play wave as CUDA tensors (state, pi, z, meta, visits)
        self.device_memory = None   # originals + symmetry twins as CUDA tensors: what optimize_network trains on
        # list[Sample] like the reference's trainer.memory; switch off for large runs (millions of Python objects)
        self.materialize_memory = materialize_memory
        # replay the SGD step as a captured HIP graph when the samples are device-resident (launch-bound otherwise)
        self.graph_sgd = True
        # "hip": the hand-written training step (csrc/az_train.hip) where it applies -- conv nets, device-resident samples, config.device
        # "cuda", batch size a multiple of 16 up to 512; "torch": the stock PyTorch loop everywhere (the checker).  sgd_backend_used
        # records what the last optimize_network ran on.
        self.sgd_backend = "hip"
        self.sgd_backend_used = None
        self._hip_step = None

    def __str__(self):
        return f"{type(self).__name__}{self.game.capitalize()}" if self.game is not None else type(self).__name__

    def print(self, log):
        if self.verbose:
            print(log)

    @staticmethod
    def print_config(config, verbose=True):
        """trainer.py:149-154"""
        if verbose:
            for pname, value in config.to_dict().items():
                print(f"- {pname}: {value}")

    @staticmethod
    def load_config_from_json(game, json_config_file):
        if json_config_file is None:
            return CONFIGS_REGISTER[game]()
        with open(json_config_file) as f:
            cfg = dotdict(json.load(f))
        return CONFIGS_REGISTER[cfg.game](**cfg)

    @staticmethod
    def estimate_training_duration(game, json_config_file=None):
        """trainer.py:166-213: the duration of a training run estimated from SelfPlayTimer (10 games) and NeuralTimer (100 batches), printed
        in the reference's five lines.  The estimate is the reference's arithmetic on this package's single-game path (timers.py); the
        batched engine plays config.episodes games at once and is far below it (SelfPlayTimer.timeit_batched times that)."""
        from datetime import timedelta

        from .timers import NeuralTimer, SelfPlayTimer
        config = AlphaZeroTrainer.load_config_from_json(game, json_config_file)
        game = game if game is not None else config.game
        if game is not None and game != config.game:
            raise ValueError(f"Game '{game}' and game '{config.game}' in the configuration file do not match.")
        AlphaZeroTrainer.print_config(config)
        episode_sec, episode_steps = SelfPlayTimer(game, config).timeit(n_episodes=10)
        batch_sec = NeuralTimer(game, config).timeit(n_batches=100)
        self_play = config.episodes * episode_sec
        n_samples = config.episodes * episode_steps
        n_batches = n_samples // config.batch_size if config.batch_size < n_samples else 1
        optim = config.epochs * batch_sec * n_batches
        evaluation = config.eval_episodes * episode_sec if config.do_eval else 0  # the opponent approximated by an AlphaZero player
        iteration = self_play + optim + evaluation
        for label, sec in (("Self-play", self_play), ("Optimization", optim), ("Iteration", iteration), ("Evaluation", evaluation)):
            print(f"{label} duration: {timedelta(seconds=round(sec))} (h:m:s)")
        print(f"TOTAL training duration: {timedelta(seconds=round(config.iterations * iteration))} (h:m:s)")

    # ------------------------------------------------------------------ self-play on the engine
    def _shape(self):
        from .engine import game_shape
        c = self.config
        return game_shape(self.game, getattr(c, "board_size", None), getattr(c, "board_width", 7), getattr(c, "board_height", 6))

    def _ensure_engine(self):
        from .engine import SelfPlayEngine
        gid, H, W, A = self._shape()
        c = self.config
        if c.simulations is None:
            raise ValueError("the batched engine needs config.simulations (compute_time-bounded search is host-only)")
        slots = max(1, min(self.engine_slots, c.episodes))
        if self._hipnet is None:
            self._hipnet = self.nn.to_hip(max_batch=slots)
        else:
            self._hipnet.load_state_dict(self.nn.state_dict())
        if self._engine is None:
            if c.temp_scheduler_type == "constant":
                tmax, tmin = -1, 0  # always greedy (schedulers.py:15-17)
            else:
                tmax, tmin = c.temp_max_step, c.temp_min_step
            plies = 2 * H * W if gid == 0 else H * W + 1
            self._engine = SelfPlayEngine(gid, H, W, n_slots=slots, n_sim=c.simulations, net=self._hipnet,
                                          dirichlet_alpha=c.dirichlet_alpha, dirichlet_epsilon=c.dirichlet_epsilon,
                                          temp_max_step=tmax, temp_min_step=tmin, seed=self.seed, max_plies=plies,
                                          sample_capacity=c.episodes * plies)
        return self._engine

    @staticmethod
    def _dist():
        """(rank, world) of the torch.distributed job this trainer runs in (one process per GPU), (0, 1) outside one"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def self_play(self, iter_idx):
        """trainer.py:215-286: all episodes are played concurrently on the engine; samples (normalised) and their
        symmetry twins stay on the device in self.device_memory; self.memory mirrors them as Sample objects"""
        from .engine import TRANSFORM_NAMES, augment_samples
        eng = self._ensure_engine()
        gid, H, W, A = self._shape()
        n = self.config.episodes
        rank, world = self._dist()
        if world == 1:
            smp = eng.run(n, first_game_id=iter_idx * n)
        else:
            # the episodes are sharded by game id (disjoint Philox streams, no collective while playing), then every
            # rank receives all samples: the same memory a single process would have built
            from .dist import all_gather_samples, shard_range
            lo, cnt, _ = shard_range(n, rank, world)
            if cnt > 0:
                smp = eng.run(cnt, first_game_id=iter_idx * n + lo)
            else:
                smp = {"state": torch.zeros((0, H, W), dtype=torch.int8, device="cuda"), "pi": torch.zeros((0, A), device="cuda"),
                       "z": torch.zeros(0, dtype=torch.int8, device="cuda"), "meta": torch.zeros((0, 4), dtype=torch.int32, device="cuda"),
                       "visits": torch.zeros((0, A), dtype=torch.int32, device="cuda")}
            smp = {k: v.cuda() for k, v in all_gather_samples(smp).items()}
        meta = smp["meta"]
        order = torch.argsort(meta[:, 0].long() * 4096 + meta[:, 1].long())  # (episode, move) order of the reference's loop
        smp = {k: v[order].contiguous() for k, v in smp.items()}
        smp["meta"][:, 0] -= iter_idx * n  # episode_idx counts from 0 in every iteration
        self.device_samples = smp
        parts = [smp]
        if self.config.data_augmentation:
            parts.append(augment_samples(gid, H, W, smp))
        self.device_memory = {k: torch.cat([p[k] for p in parts]) for k in ("state", "pi", "z", "meta")}
        self.memory = None
        if self.materialize_memory:
            st = self.device_memory["state"].cpu().numpy().astype(np.float64)
            pi = self.device_memory["pi"].cpu().numpy().astype(np.float64)
            z = self.device_memory["z"].cpu().numpy()
            mt = self.device_memory["meta"].cpu().numpy()
            S = smp["z"].shape[0]
            self.memory = [Sample(state=st[i], pi=pi[i], player=1, outcome=int(z[i]), episode_idx=int(mt[i, 0]), move_idx=int(mt[i, 1]),
                                  transformation=None if i < S else TRANSFORM_NAMES[mt[i, 3]]) for i in range(len(z))]
        self.print(f"Total number of samples: {self.device_memory['z'].shape[0]}")

    # ------------------------------------------------------------------ optimisation (trainer.py:288-387)
    def _n_samples(self):
        return self.device_memory["z"].shape[0] if self.device_memory is not None else len(self.memory)

    def _permutation(self, n, device):
        """the epoch's sample order (trainer.py:292-294 shuffles indexes with np.random); drawn on the device when the
        memory lives there.  One place, so tests can pin the order to the reference's."""
        return torch.randperm(n, device=device)

    def _batch_generator(self, n_batches):
        """random batches without replacement (trainer.py:288-318); device-resident when self_play ran on the engine"""
        bs, dev = self.config.batch_size, self.config.device
        if self.device_memory is not None:
            m = self.device_memory
            idx = self._permutation(m["z"].shape[0], m["z"].device)[: n_batches * bs].view(n_batches, bs)
            for rows in idx:
                yield (m["state"][rows].to(dev, torch.float32), m["pi"][rows].to(dev, torch.float32),
                       m["z"][rows].to(dev, torch.float32).unsqueeze(1))
            return
        idx = np.arange(len(self.memory))
        np.random.shuffle(idx)
        for rows in idx[: n_batches * bs].reshape(n_batches, bs):
            x = np.stack([self.memory[i].state for i in rows])
            pi = np.stack([self.memory[i].pi for i in rows])
            z = np.array([[self.memory[i].outcome] for i in rows], dtype=np.float64)
            yield (torch.tensor(x, dtype=torch.float32, device=dev), torch.tensor(pi, dtype=torch.float32, device=dev),
                   torch.tensor(z, dtype=torch.float32, device=dev))

    def optimize_network(self, iter_idx):
        rank, world = self._dist()
        if world > 1:
            # the reference's optimisation is one sequential SGD run (trainer.py:320-381): rank 0 does it, the others
            # wait for the weights (4.46 MB for OthelloNet 8x8)
            from .dist import broadcast_state_dict
            if rank == 0:
                self._optimize_local(iter_idx)
            else:
                self.nn_twin = self.nn.clone()
                self.loss_values[iter_idx] = {}
            broadcast_state_dict(self.nn_twin, src=0)
            return
        self._optimize_local(iter_idx)

    def _optimize_hip(self, iter_idx):
        """the reference's loop (trainer.py:320-381) on the hand-written step: per epoch one permutation, n_batches steps replayed as a
        HIP graph, the losses read back once; ExponentialLR(0.9) between epochs; a fresh optimizer (zero momentum) per iteration"""
        from .train_step import HipTrainStep
        c, m, bs = self.config, self.device_memory, self.config.batch_size
        key = (type(self.nn_twin).__name__, self.nn_twin.hip_shape(), bs)
        if self._hip_step is None or self._hip_step[0] != key:
            if self._hip_step is not None:
                self._hip_step[1].close()
            self._hip_step = (key, HipTrainStep(self.nn_twin, max_batch=bs))
        ts = self._hip_step[1]
        ts.load(self.nn_twin)
        ts.begin(c.learning_rate, 0.9, 0.0001, float(getattr(self.nn_twin, "dropout", 0.0)), seed=(self.seed * 7919 + iter_idx * 104729 + 1) & 0xFFFFFFFF)
        self.loss_values[iter_idx] = {}
        lr = c.learning_rate
        state, pi, z = m["state"].contiguous(), m["pi"].contiguous(), m["z"].contiguous()
        for epoch in range(c.epochs):
            n_batches = self._n_samples() // bs
            if n_batches == 0:
                raise ValueError(f"Too few samples in the memory ({self._n_samples()}) to create a batch with batch_size = {bs}")
            perm = self._permutation(m["z"].shape[0], m["z"].device)[: n_batches * bs].to(torch.int64).contiguous()
            lp = torch.empty(n_batches, dtype=torch.float32, device="cuda")
            lv = torch.empty(n_batches, dtype=torch.float32, device="cuda")
            ts.set_lr(lr)
            ts.steps(state, pi, z, perm, n_batches, bs, lp, lv)
            ts.check()  # the epoch is done; a permutation entry outside the memory is a ValueError
            self.loss_values[iter_idx][epoch] = {"pi": lp.cpu().tolist(), "v": lv.cpu().tolist()}
            lr = lr * 0.9  # torch.optim.lr_scheduler.ExponentialLR(gamma=0.9) (trainer.py:327)
        ts.store(self.nn_twin)

    def _optimize_local(self, iter_idx):
        self.nn_twin = self.nn.clone()
        self.nn_twin.train()
        from . import train_step
        if (self.sgd_backend == "hip" and self.device_memory is not None and torch.cuda.is_available()
                and torch.device(self.config.device).type == "cuda" and train_step.supports(self.nn_twin, self.config.batch_size)):
            self.sgd_backend_used = "hip"
            return self._optimize_hip(iter_idx)
        if (self.sgd_backend == "hip" and self.device_memory is not None and torch.device(self.config.device).type == "cuda"
                and not getattr(self, "_warned_sgd_fallback", False)):
            # never silent, but once per trainer: "hip" is the default, a user who asked for nothing is not warned every iteration
            self._warned_sgd_fallback = True
            import warnings
            warnings.warn(f"sgd_backend 'hip' does not cover {type(self.nn_twin).__name__} at batch size {self.config.batch_size} "
                          f"(csrc/az_train.hip: OthelloNet / Connect4Net at multiples of 16 up to 512, TicTacToeNet at 2..256): "
                          f"this optimize_network runs the stock PyTorch step", RuntimeWarning, stacklevel=2)
        self.sgd_backend_used = "torch"
        opt = torch.optim.SGD(self.nn_twin.parameters(), lr=self.config.learning_rate, momentum=0.9, weight_decay=0.0001)
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
        self.loss_values[iter_idx] = {}
        bs = self.config.batch_size
        graphed = (self.graph_sgd and self.device_memory is not None and torch.cuda.is_available()
                   and torch.device(self.config.device).type == "cuda")
        for epoch in range(self.config.epochs):
            n_batches = self._n_samples() // bs
            if n_batches == 0:
                raise ValueError(f"Too few samples in the memory ({self._n_samples()}) to create a batch with batch_size = {bs}")
            if graphed:
                pi_losses, v_losses = self._epoch_graphed(opt, n_batches, first=(epoch == 0))
            else:
                pi_losses, v_losses = [], []
                for x, pi, z in self._batch_generator(n_batches):
                    loss_pi, loss_v = self._sgd_step(opt, x, pi, z)
                    pi_losses.append(loss_pi.cpu().item())
                    v_losses.append(loss_v.cpu().item())
            self.loss_values[iter_idx][epoch] = {"pi": pi_losses, "v": v_losses}
            sched.step()

    def _sgd_step(self, opt, x, pi, z):
        """one optimisation step of the reference's loop (trainer.py:346-366)"""
        bs = self.config.batch_size
        opt.zero_grad(set_to_none=True)
        log_p, v = self.nn_twin(x)
        loss_pi = -torch.sum(pi * log_p) / bs
        loss_v = torch.sum((v - z) ** 2) / bs
        (loss_pi + loss_v).backward()
        opt.step()
        return loss_pi.detach(), loss_v.detach()

    def _epoch_graphed(self, opt, n_batches, first):
        """One epoch with the step (batch gather from the device memory, forward, backward, SGD update) captured once
        as a HIP graph and replayed: the eager loop is bound by ~100 kernel launches and two host syncs per step.
        The learning rate is a host scalar inside the captured optimizer kernels, so the graph is re-captured every
        epoch (ExponentialLR, trainer.py:334).  Same arithmetic as _sgd_step; losses are read back once per epoch."""
        m, bs, dev = self.device_memory, self.config.batch_size, torch.device(self.config.device)
        perm = self._permutation(m["z"].shape[0], m["z"].device)[: n_batches * bs].view(n_batches, bs)
        step_t = torch.zeros(1, dtype=torch.long, device=dev)
        pi_log = torch.zeros(n_batches, device=dev)
        v_log = torch.zeros(n_batches, device=dev)

        def batch_of(rows):
            return (m["state"].index_select(0, rows).to(dev, torch.float32), m["pi"].index_select(0, rows).to(dev, torch.float32),
                    m["z"].index_select(0, rows).to(dev, torch.float32).unsqueeze(1))

        done = 0
        if first:  # eager steps first: momentum buffers and MIOpen's algorithm choices must exist before the capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(min(3, n_batches)):
                    lp, lv = self._sgd_step(opt, *batch_of(perm[done]))
                    pi_log[done], v_log[done] = lp, lv
                    done += 1
            torch.cuda.current_stream().wait_stream(side)
        if done < n_batches:
            step_t.fill_(done)
            opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                rows = perm.index_select(0, step_t).view(-1)
                lp, lv = self._sgd_step(opt, *batch_of(rows))
                pi_log.index_copy_(0, step_t, lp.view(1))
                v_log.index_copy_(0, step_t, lv.view(1))
                step_t += 1
            # the capture itself did not execute anything: replay once per remaining batch
            for _ in range(n_batches - done):
                graph.replay()
        return pi_log.cpu().tolist(), v_log.cpu().tolist()

    def update_network(self, iter_idx):
        if self.config.do_eval and self.config.eval_opponent == "previous":
            self.prev_nn = self.nn  # the network this iteration's self-play used: the evaluation opponent
        self.nn = self.nn_twin.clone()
        self.nn_twin = None
        self.az_player.mct.nn = self.nn

    # ------------------------------------------------------------------ evaluation (trainer.py:389-446)
    def _init_evaluator(self):
        if not self.config.do_eval:
            return
        opp = self.config.eval_opponent
        if opp == "previous":
            # extension (BASELINE config 5, "arena eval vs prev net"): the freshly trained network against the one it
            # replaces, all games at once on the GPU; "alphazero" keeps the reference's refusal (trainer.py:399-400)
            if self.config.simulations is None:
                raise ValueError("Evaluation against the previous network needs config.simulations.")
            self.eval_results = {"eval_opponent": opp, "eval_episodes": self.config.eval_episodes, "results": {}}
            return
        if opp not in PLAYERS_SET:
            raise ValueError(f"Opponent player '{opp}' not found in the players register.")
        if opp == "human":
            raise ValueError("Evaluation against a HumanPlayer during training is not allowed.")
        if opp == "alphazero":
            raise ValueError("Evaluation against another AlphaZeroPlayer during training is not yet implemented.")
        self.eval_results = {"eval_opponent": opp, "eval_episodes": self.config.eval_episodes, "results": {}}

    def evaluate(self, iter_idx):
        if not self.config.do_eval:
            return
        c = self.config
        rank, world = self._dist()
        batched = c.eval_opponent in ("random", "greedy", "mcts", "previous") and c.simulations is not None
        if not batched and rank != 0:  # host Arena: rank 0 plays; the batched arena below shards its rounds over all ranks
            return
        if batched:
            # all evaluation games at once on the GPU(s) (same stats dict as Arena.play_games)
            from .arena import BatchedArena
            if c.eval_opponent == "previous":
                if self.prev_nn is None:
                    raise ValueError("eval_opponent = 'previous' needs update_network() to have run: there is no previous network yet")
                opp_arg, p2_name = self.prev_nn, "AlphaZeroPlayer(previous network)"
            else:
                opp_arg = c.eval_opponent
                p2_name = f"{PLAYERS_REGISTER[c.eval_opponent](**({'n_sim': c.simulations} if c.eval_opponent == 'mcts' else {}))}"
            arena = BatchedArena(self.game, self.nn, opponent=opp_arg, n_sim=c.simulations, seed=self.seed + iter_idx,
                                 board_size=getattr(c, "board_size", None), board_width=getattr(c, "board_width", 7),
                                 board_height=getattr(c, "board_height", 6))
            stats = arena.play_games(n_rounds=c.eval_episodes, return_stats=True)
            p1_name = f"{type(self.az_player).__name__}"
        else:
            eval_player = AlphaZeroPlayer(n_sim=c.simulations, compute_time=c.compute_time, nn=self.nn)
            kwargs = {"n_sim": c.simulations} if c.eval_opponent == "mcts" else {}
            opponent = PLAYERS_REGISTER[c.eval_opponent](**kwargs)
            arena = Arena(player1=eval_player, player2=opponent, board=BOARDS_REGISTER[self.game](config=c))
            stats = arena.play_games(n_rounds=c.eval_episodes, return_stats=True)
            p1_name, p2_name = f"{eval_player}", f"{opponent}"
        if rank != 0:
            return
        for key in ("player1", "player2", "draw"):
            stats.pop(key, None)
        self.eval_results["results"][iter_idx] = {k: dict(v) for k, v in stats.items()}
        self.eval_results["player1"], self.eval_results["player2"] = p1_name, p2_name

    # ------------------------------------------------------------------ persistence (trainer.py:448-473)
    def _model_dir(self, model_name, path):
        path = os.path.join(base.DEFAULT_MODELS_PATH, model_name) if path is None else path
        os.makedirs(path, exist_ok=True)
        return path

    def save_player_pt(self, model_name, path=None):
        self.nn.save_model(model_name, self._model_dir(model_name, path), verbose=False)

    def save_player_config(self, model_name, path=None):
        with open(os.path.join(self._model_dir(model_name, path), "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=4)

    def save_training_stats(self, model_name, path=None):
        path = self._model_dir(model_name, path)
        with open(os.path.join(path, "loss.json"), "w") as f:
            json.dump(self.loss_values, f, indent=4)
        if self.config.do_eval:
            with open(os.path.join(path, "eval.json"), "w") as f:
                json.dump(self.eval_results, f, indent=4)

    # ------------------------------------------------------------------ policy iteration (trainer.py:475-572)
    def train(self, game=None, experiment_name=None, json_config_file=None, plot=False, verbose=None):
        if game is None and json_config_file is None:
            raise ValueError("The name of the game or a JSON configuration file must be provided to launch a training.")
        self.config = self.load_config_from_json(game, json_config_file)
        self.game = game if game is not None else self.config.game
        if self.game != self.config.game:
            raise ValueError(f"Game '{game}' and game '{self.config.game}' in the configuration file do not match.")
        experiment_name = experiment_name or self.DEFAULT_EXP_NAME
        self.verbose = verbose if verbose is not None else self.verbose
        c = self.config
        c.save_checkpoints = c.save_checkpoints or c.push_checkpoints
        c.push = c.push or c.push_checkpoints
        c.save = c.save or c.push or c.save_checkpoints
        self.setup()
        writer = self._dist()[0] == 0  # multi-GPU job: rank 0 owns the files
        if writer:
            self.save_player_config(experiment_name)
        for it in range(c.iterations):
            self.print(f"\n----- Iteration {it+1}/{c.iterations} -----")
            self.self_play(it)
            self.optimize_network(it)
            self.update_network(it)
            self.evaluate(it)
            if writer:
                self.save_training_stats(experiment_name)
            if c.save_checkpoints and writer:
                self.save_player_pt(f"{experiment_name}-chkpt-{it+1}",
                                    path=os.path.join(base.DEFAULT_MODELS_PATH, experiment_name, "checkpoints"))
        if c.save and writer:
            self.save_player_pt(experiment_name)

    def setup(self):
        """objects train() creates before the loop (trainer.py:500-518); needs self.game and self.config"""
        c = self.config
        rank, world = self._dist()
        if world > 1 and torch.distributed.get_backend() == "nccl" and torch.device(c.device).type != "cuda":
            # fail before the first self-play wave, not after rank 0 has trained (the weight broadcast runs over RCCL)
            raise ValueError("distributed training over RCCL needs config.device = 'cuda'")
        self.board = BOARDS_REGISTER[self.game](config=c)
        self.nn = NETWORKS_REGISTER[self.game](config=c)
        if world > 1:
            # every rank built its own random initialisation: the job plays and evaluates with rank 0's (self-play shards,
            # the sharded arena and eval_opponent = "previous" at iteration 0 must not depend on the number of ranks)
            from .dist import broadcast_state_dict
            self.nn.to(c.device)
            broadcast_state_dict(self.nn, src=0)
        self.prev_nn = None
        self.az_player = AlphaZeroPlayer(n_sim=c.simulations, compute_time=c.compute_time, nn=self.nn,
                                         dirichlet_alpha=c.dirichlet_alpha, dirichlet_epsilon=c.dirichlet_epsilon)
        self.temp_scheduler = TEMP_SCHEDULERS[c.temp_scheduler_type](temp_max_step=c.temp_max_step, temp_min_step=c.temp_min_step,
                                                                     max_steps=self.board.max_moves)
        self.data_augment_strategy = DATA_AUGMENT_STRATEGIES[self.game] if c.data_augmentation else None
        self._init_evaluator()
        self.loss_values = {}


def freeze_config(game=None):
    """trainer.py:580-587: the default configuration of `game` (of every game if None) written as <game>.json under DEFAULT_CONFIGS_PATH"""
    os.makedirs(base.DEFAULT_CONFIGS_PATH, exist_ok=True)
    for g in ([game] if game is not None else list(GAMES_SET)):
        with open(os.path.join(base.DEFAULT_CONFIGS_PATH, f"{g}.json"), "w") as f:
            json.dump(AlphaZeroTrainer.load_config_from_json(g, json_config_file=None).to_dict(), f, indent=4)
