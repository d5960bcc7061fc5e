"""Arena (arena.py:12-185): games between two Players on one Board, win/loss/draw statistics."""
from collections import defaultdict


def _check_rounds(n_rounds, start_player):
    """arena.py:24-34"""
    if not isinstance(n_rounds, int) or n_rounds < 1:
        raise ValueError("n_rounds must be a positive integer")
    if start_player is None and n_rounds % 2 != 0:
        raise ValueError("n_rounds must be an even number or the evaluation will be biased!")
    if start_player is not None and start_player not in (1, 2):
        raise ValueError("start_player must be either 1 or 2 or None (to alternate starts)")


class Arena:
    def __init__(self, player1, player2, board):
        self.player1, self.player2, self.board = player1, player2, board
        self.game = board.game

    def play_game(self, player2_starts=False, display=False, save_frames=False, return_results=False, show_indexes=True, show_probs=False, verbose=False):
        """arena.py:36-117, positional order included; `display`, `save_frames`, `show_indexes`, `show_probs` steer the board rendering
        of the reference (out of scope here): accepted and ignored"""
        idx = 1 if player2_starts else 0
        self.board.reset()
        self.player1.reset()
        self.player2.reset()
        players = (self.player1, self.player2)
        while not self.board.is_game_over():
            move, _, _, _ = players[idx].get_move(self.board)
            self.board.play_move(move)
            if verbose:
                print(f"{players[idx]} played {move} | score = {-self.board.get_score()}")
            players[idx].apply_move(move, player=-self.board.player)      # every player keeps its own tree
            players[1 - idx].apply_move(move, player=self.board.player)   # (arena.py:98-99)
            idx = 1 - idx
        score = abs(self.board.get_score())
        winner = self.board.get_winner()
        if winner == 0:
            return {"winner": 0, "score": score} if return_results else None
        first_won = (winner == 1 and not player2_starts) or (winner == -1 and player2_starts)
        return {"winner": 1 if first_won else 2, "score": score} if return_results else None

    def play_games(self, n_rounds, start_player=None, return_stats=False, verbose=False, call_id=None):
        _check_rounds(n_rounds, start_player)
        stats = {"player1": [], "player2": [], "draw": 0,
                 "player1_starts": defaultdict(int), "player2_starts": defaultdict(int)}
        for round_idx in range(n_rounds):
            p2s = {1: False, 2: True}.get(start_player, bool(round_idx % 2))
            res = self.play_game(player2_starts=p2s, return_results=True, verbose=verbose)
            starter = f"player{2 if p2s else 1}_starts"
            if res["winner"] == 0:
                stats["draw"] += 1
                stats[starter]["draw"] += 1
            else:
                stats[f"player{res['winner']}"].append(res["score"])
                stats[starter]["win" if res["winner"] == (2 if p2s else 1) else "loss"] += 1
        return stats if return_stats else None

    def play_games_in_parallel(self, n_rounds, n_process=None, verbose=False, return_stats=False):
        """the reference forks CPU processes here (arena.py:187-257); device trees are not forked -- plays serially"""
        return self.play_games(n_rounds, return_stats=return_stats, verbose=verbose)

    @staticmethod
    def print_stats_results(player1, player2, stats):
        n1, n2, d = len(stats["player1"]), len(stats["player2"]), stats["draw"]
        print(f"{player1} wins: {n1} | {player2} wins: {n2} | draws: {d}")


class BatchedArena:
    """Arena.play_games (arena.py:119-185) with all rounds played at once on the GPU.

    player1 is an AlphaZero player (network `nn`, `n_sim` simulations, no noise, temperature 0: what
    AlphaZeroTrainer.evaluate builds, trainer.py:421-425) or "mcts" (MCTSPlayer: UCT + random playouts on the device);
    `opponent` is "random", "greedy", "mcts" or another network (evaluation against a previous network).
    Returns the reference's stats dict.
    """

    def __init__(self, game, nn, opponent="random", n_sim=100, opponent_n_sim=None, seed=0, board_size=None,
                 board_width=7, board_height=6):
        from .engine import game_shape
        self.game = game
        self.gid, self.H, self.W, self.A = game_shape(game, board_size if board_size is not None else getattr(nn, "n", None),
                                                      board_width, board_height)
        self.nn, self.opponent, self.n_sim, self.seed = nn, opponent, n_sim, seed
        self.opponent_n_sim = opponent_n_sim if opponent_n_sim is not None else n_sim
        self.overlap = True  # two tree players search at the same time (each on its own stream); False: one after the other
        self.tie_mode = None  # None: fair_max draws among equals (utils.py:28-34); tests pin it to engine.TIE_LOWEST (golden G7)

    def _engine(self, net, G, n_sim, seed):
        from .engine import EVAL_NET, EVAL_ROLLOUT, NOISE_OFF, TIE_RANDOM, SelfPlayEngine
        plies = 4 * self.H * self.W + 16
        from .engine import EVAL_FAKE
        rollout, fake = net == "mcts", net == "fake"  # "fake": the closed-form test network of tools/closed_form.py (golden G3 / G7)
        if isinstance(net, str) and not (rollout or fake):
            raise ValueError(f"player '{net}' has no search tree")
        return SelfPlayEngine(self.gid, self.H, self.W, n_slots=G, n_sim=n_sim, net=None if isinstance(net, str) else net.to_hip(max_batch=G),
                              evaluator=EVAL_ROLLOUT if rollout else (EVAL_FAKE if fake else EVAL_NET), dirichlet_alpha=None, dirichlet_epsilon=None, temp_max_step=-1, temp_min_step=0,
                              tie_mode=TIE_RANDOM if self.tie_mode is None else self.tie_mode, noise_mode=NOISE_OFF, seed=seed, max_plies=plies, sample_capacity=16)

    def play_games(self, n_rounds, start_player=None, return_stats=True, shard=True, record_moves=False):
        """all rounds at once.  Inside a torch.distributed job (one process per GPU) the rounds are sharded over the
        ranks in contiguous blocks and the per-game results (winner, score: a few bytes per game) are all-gathered, so
        every rank returns the stats of ALL rounds (optional collective 3).  Game ids are the global round
        numbers: the games do not depend on the number of ranks.  record_moves keeps the move vector of every ply in
        self.moves (tests)."""
        import numpy as np
        import torch
        from collections import defaultdict
        from .dist import gather_sharded_rows, initialized, rank_world, shard_range
        _check_rounds(n_rounds, start_player)
        rank, world = rank_world() if shard else (0, 1)
        lo, G, per = shard_range(n_rounds, rank, world)
        p2_all = np.array([{1: False, 2: True}.get(start_player, bool(r % 2)) for r in range(n_rounds)])
        side_all = np.where(p2_all, -1, 1).astype(np.int8)  # colour +1 moves first
        res = np.full((per, 2), -2, np.int32)  # (winner, score) of this rank's rounds, padded to `per` rows
        self.moves = []
        if G > 0:
            winner, score = self._play(G, side_all[lo:lo + G], np.arange(lo, lo + G, dtype=np.uint32), record_moves)
            res[:G, 0], res[:G, 1] = winner, score
        if shard and initialized():  # also with one rank: the collective is part of the path
            dev = "cuda" if torch.distributed.get_backend() != "gloo" else "cpu"
            res = gather_sharded_rows(torch.from_numpy(res).to(dev), n_rounds, force=True).cpu().numpy()
        winner, score = res[:n_rounds, 0], res[:n_rounds, 1]
        stats = {"player1": [], "player2": [], "draw": 0, "player1_starts": defaultdict(int), "player2_starts": defaultdict(int)}
        for g in range(n_rounds):
            starter = f"player{2 if p2_all[g] else 1}_starts"
            if winner[g] == 0:
                stats["draw"] += 1
                stats[starter]["draw"] += 1
            else:
                who = 1 if winner[g] == side_all[g] else 2
                sc = int(abs(score[g]))
                stats[f"player{who}"].append(float("inf") if self.gid == 2 and sc == 32767 else sc)  # tictactoe.py:119-126
                stats[starter]["win" if who == (2 if p2_all[g] else 1) else "loss"] += 1
        return stats if return_stats else None

    def _play(self, G, side1, round_ids, record_moves=False):
        """G games in lock-step on this GPU; returns (winner, score) per game"""
        import numpy as np
        board = {0: lambda: __import__("alphazero_amd.games.othello", fromlist=["OthelloBoard"]).OthelloBoard(n=self.H),
                 1: lambda: __import__("alphazero_amd.games.connect4", fromlist=["Connect4Board"]).Connect4Board(width=self.W, height=self.H),
                 2: lambda: __import__("alphazero_amd.games.tictactoe", fromlist=["TicTacToeBoard"]).TicTacToeBoard()}[self.gid]()
        grids = np.tile(board.grid.astype(np.int8)[None], (G, 1, 1))
        ones = np.ones(G, np.int8)
        # game id = (round + seed * 100003) mod 2^32: any seed is fine (np.uint32(big) raises, and uint32 + uint32 warns on wrap-around)
        ids = ((round_ids.astype(np.uint64) + np.uint64((self.seed * 100003) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        e1 = self._engine(self.nn, G, self.n_sim, self.seed)
        e1.set_roots(grids, ones, game_ids=ids)
        e1.set_sides(side1)
        e2 = None
        if self.opponent in ("mcts", "fake") or not isinstance(self.opponent, str):
            e2 = self._engine(self.opponent, G, self.opponent_n_sim, self.seed + 1)
            e2.set_roots(grids, ones, game_ids=ids)
            e2.set_sides(-side1)
            if self.overlap:
                e1.pair_with(e2)
        for _ in range(4 * self.H * self.W + 8):
            _, over, winner, score = e1.root_status()
            if over.all():
                break
            if e2 is not None and self.overlap:  # both players think at once: each engine searches the slots where its colour is to move, on its own stream
                e1.search_begin(self.n_sim)
                try:
                    e2.search_begin(self.opponent_n_sim)
                    e2.search_end()
                finally:
                    e1.search_end()
                a, b = e1.best_moves(), e2.best_moves()
            else:
                e1.search(self.n_sim)
                a = e1.best_moves()
                if e2 is not None:
                    e2.search(self.opponent_n_sim)
                    b = e2.best_moves()
                else:
                    b = e1.baseline_moves(self.opponent, seed=self.seed + 7)
            moves = np.where(a >= 0, a, b).astype(np.int32)
            if record_moves:
                self.moves.append(moves.copy())
            e1.play(moves)
            if e2 is not None:
                e2.play(moves)
        else:
            raise RuntimeError("arena games did not finish")
        self.engine_stats = [e.stats() for e in (e1, e2) if e is not None]  # graph replays, lock-steps, evaluated rows
        e1.close()
        if e2 is not None:
            e2.close()
        return winner, score
