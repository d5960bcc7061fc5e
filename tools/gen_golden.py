#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE; runs only where /root/reference is mounted).

Imports the Python reference t0m1ab/alphazero *in place* (read-only, no bytecode written) through
the two-line shim of SURVEY.md Appendix B and records small fixtures under tests/golden/:

  G1 rules_<game>.npz    random playouts + random positions: legal sets, flips, winners (bit-exact)
  G2 net_<game>.npz      network known answers under closed-form weights (tolerance 1e-5)
  G3 mct_<game>.npz      MCT root statistics under the closed-form fake net, deterministic
                         fair_max, with and without closed-form Dirichlet noise (exact N, Q 1e-12)
  G4 selfplay_<game>.npz AlphaZeroTrainer.self_play memory (samples before/after normalise and
                         after the symmetry augmentation) under the same patches
     selfplay_frac_<game>.npz  the same with temp_max_step = 2, temp_min_step = 6: plies 3-5 at tau = 0.75 / 0.5 / 0.25, where
                         get_action_probs is N ** (1 / tau) / sum (mcts.py:114-116)
  G5 stats.npz           outcome statistics of reference rollout-MCTS TicTacToe self-play
  G7 arena_<game>.npz    Arena.play_games (arena.py:119-185) between AlphaZeroPlayer (closed-form fake net, no noise) and GreedyPlayer /
                         another AlphaZeroPlayer under the deterministic fair_max: every move of every round, who started, winners,
                         scores and the stats dict
  G6 sgd_<game>.npz      AlphaZeroTrainer.optimize_network (trainer.py:320-381): per-batch policy / value losses of two
                         epochs on the G4 memory and the trained fc1 / value-head weights (closed-form initial weights,
                         dropout 0, np.random.seed pinned for the batch shuffle)

The fixtures hold data only (inputs and expected outputs).  Usage: python tools/gen_golden.py [names]
"""
import os
import sys
import types
import enum

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, ROOT)

from tools import closed_form as cf  # noqa: E402


def load_reference():
    if not os.path.isdir(os.path.join(REF, "alphazero")):
        raise SystemExit("reference not mounted: nothing to generate")
    sys.dont_write_bytecode = True
    m = types.ModuleType("aenum"); m.Enum = enum.Enum; m.NoAlias = object(); sys.modules["aenum"] = m
    p = types.ModuleType("alphazero"); p.__path__ = [os.path.join(REF, "alphazero")]; p.__version__ = "0.1.0"
    sys.modules["alphazero"] = p
    import alphazero.games.othello as oth
    import alphazero.games.connect4 as c4
    import alphazero.games.tictactoe as ttt
    import alphazero.mcts as mcts
    import alphazero.players as players
    import alphazero.trainer as trainer
    import alphazero.schedulers as schedulers
    import alphazero.games.registers as registers
    import alphazero.arena as arena
    return types.SimpleNamespace(oth=oth, c4=c4, ttt=ttt, mcts=mcts, players=players, trainer=trainer,
                                 schedulers=schedulers, registers=registers, arena=arena)


GAMES = {
    # tag: (game, board kwargs, action size, n for encoding)
    "othello8": ("othello", dict(n=8), 65, 8),
    "othello6": ("othello", dict(n=6), 37, 6),
    "othello4": ("othello", dict(n=4), 17, 4),
    "connect4": ("connect4", dict(width=7, height=6), 7, None),
    "tictactoe": ("tictactoe", dict(), 9, None),
}


def make_board(R, tag, grid=None, player=1):
    game, kw, _, _ = GAMES[tag]
    cls = {"othello": R.oth.OthelloBoard, "connect4": R.c4.Connect4Board, "tictactoe": R.ttt.TicTacToeBoard}[game]
    if grid is not None:
        return cls(grid=np.array(grid, dtype=np.float64), player=player, **kw)
    return cls(**kw)


def legal_mask(board, tag, player=None):
    game, _, A, n = GAMES[tag]
    m = np.zeros(A, dtype=bool)
    for mv in board.get_moves(player):
        m[cf.move_to_action(game, mv, n)] = True
    return m


# ------------------------------------------------------------------------------------------- G1
def gen_rules(R, tag, n_games, seed, n_positions=0):
    game, kw, A, n = GAMES[tag]
    rng = np.random.RandomState(seed)
    actions, legal, legal_other, players, offsets = [], [], [], [], [0]
    winners, scores, final_grids, final_players = [], [], [], []
    sample_positions = []  # (grid, player) of non-terminal positions, reused by G2/G3
    for g in range(n_games):
        b = make_board(R, tag)
        while not b.is_game_over():
            lm = legal_mask(b, tag)
            acts = np.flatnonzero(lm)
            a = int(acts[rng.randint(len(acts))])
            legal.append(lm); legal_other.append(legal_mask(b, tag, -b.player)); players.append(b.player)
            actions.append(a)
            if rng.rand() < 0.08:
                sample_positions.append((b.grid.copy().astype(np.int8), int(b.player)))
            # every action outside the legal set must be refused (ValueError)
            bad = [x for x in range(A) if not lm[x]]
            if bad:
                x = bad[rng.randint(len(bad))]
                c = b.clone()
                try:
                    c.play_move(cf.action_to_move(game, x, n))
                    raise AssertionError(f"{tag}: illegal action {x} accepted by the reference")
                except ValueError:
                    pass
            b.play_move(cf.action_to_move(game, a, n))
        offsets.append(len(actions))
        sc = b.get_score()
        winners.append(b.get_winner()); scores.append(32767 if sc == float('inf') else int(sc))
        final_grids.append(b.grid.astype(np.int8)); final_players.append(b.player)
    out = dict(actions=np.array(actions, np.int16), legal=np.packbits(np.array(legal), axis=1),
               legal_other=np.packbits(np.array(legal_other), axis=1), players=np.array(players, np.int8),
               offsets=np.array(offsets, np.int32), winners=np.array(winners, np.int8),
               scores=np.array(scores, np.int16), final_grids=np.array(final_grids),
               final_players=np.array(final_players, np.int8), action_size=np.int32(A))
    # random (possibly unreachable) positions: legality for both sides, game over, winner, one played move
    if n_positions:
        H, W = make_board(R, tag).grid.shape
        pg, pp, pl, plo, pover, pwin, pact, pres = [], [], [], [], [], [], [], []
        for i in range(n_positions):
            dens = rng.rand()
            grid = np.zeros((H, W), np.int8)
            if game == "connect4":  # gravity-valid: random column heights, filled from the bottom row up
                for c in range(W):
                    hgt = rng.randint(0, H + 1) if rng.rand() < 0.7 else H
                    for r in range(H - 1, H - 1 - hgt, -1):
                        grid[r, c] = rng.choice([-1, 1])
            else:
                fill = rng.rand(H, W) < dens
                grid[fill] = rng.choice([-1, 1], size=int(fill.sum()), p=[0.5, 0.5]) if rng.rand() < 0.8 else \
                    rng.choice([-1, 1], size=int(fill.sum()), p=[0.05, 0.95])
            player = int(rng.choice([-1, 1]))
            b = make_board(R, tag, grid=grid, player=player)
            lm = legal_mask(b, tag); lo = legal_mask(b, tag, -player)
            over = b.is_game_over()
            acts = np.flatnonzero(lm)
            c = b.clone()
            if len(acts):
                a = int(acts[rng.randint(len(acts))])
                c.play_move(cf.action_to_move(game, a, n))
            else:
                a = -1
            pg.append(grid); pp.append(player); pl.append(lm); plo.append(lo); pover.append(over)
            pwin.append(b.get_winner() if over else 2); pact.append(a); pres.append(c.grid.astype(np.int8))
        out.update(pos_grids=np.array(pg), pos_players=np.array(pp, np.int8), pos_legal=np.packbits(np.array(pl), axis=1),
                   pos_legal_other=np.packbits(np.array(plo), axis=1), pos_over=np.array(pover), pos_winner=np.array(pwin, np.int8),
                   pos_action=np.array(pact, np.int16), pos_result=np.array(pres))
    np.savez_compressed(os.path.join(GOLD, f"rules_{tag}.npz"), **out)
    print(f"rules_{tag}: {n_games} games, {len(actions)} plies, passes="
          f"{sum(1 for a in actions if game == 'othello' and a == A - 1)}, positions={n_positions}")
    return sample_positions


# ------------------------------------------------------------------------------------------- G2
def make_net(R, tag):
    game, kw, A, n = GAMES[tag]
    if game == "othello":
        return R.oth.OthelloNet(n=n)
    if game == "connect4":
        return R.c4.Connect4Net(board_width=7, board_height=6)
    return R.ttt.TicTacToeNet()


def gen_net(R, tag, positions):
    import torch
    torch.set_num_threads(1)
    game, kw, A, n = GAMES[tag]
    net = make_net(R, tag)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = cf.closed_form_state_dict(shapes)
    net.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    net.eval()
    pos = positions[:64]
    assert len(pos) == 64, len(pos)
    canon = np.array([g.astype(np.float32) * p for g, p in pos], dtype=np.float32)
    probs, v = net.predict(torch.tensor(canon))
    probs = probs.numpy().astype(np.float32); v = v.numpy().reshape(-1).astype(np.float32)
    # single-board evaluate() path incl. the player flip of base.py:363-366
    ev_p, ev_v = [], []
    for g, p in pos[:8]:
        b = make_board(R, tag, grid=g, player=p)
        pr, vv = net.evaluate(b)
        ev_p.append(pr); ev_v.append(vv)
    np.savez_compressed(os.path.join(GOLD, f"net_{tag}.npz"), grids=np.array([g for g, _ in pos]),
                        players=np.array([p for _, p in pos], np.int8), probs=probs, v=v,
                        eval_probs=np.array(ev_p, np.float32), eval_v=np.array(ev_v, np.float64),
                        n_params=np.int64(net.get_parameters_count()),
                        shape_keys=np.array(list(shapes.keys())), shape_vals=np.array([str(s) for s in shapes.values()]))
    print(f"net_{tag}: params={net.get_parameters_count()} probs[0,:3]={probs[0, :3]} v[:3]={v[:3]}")


# ------------------------------------------------------------------------------------------- patches
class Patches:
    """deterministic replacements for the reference's global-numpy randomness"""

    def __init__(self, R, tag):
        self.R, self.tag = R, tag
        self.game, _, self.A, self.n = GAMES[tag]
        self.mct = None          # current tree (for the Dirichlet patch)
        self.root_board = None   # board handed to MCT.search
        self.seed = 0
        self.episode = 0
        self.ply = 0
        self._orig = {}

    def act(self, move):
        return cf.move_to_action(self.game, move, self.n)

    def fair_max_lowest(self, elements, key=lambda x: x):
        elements = list(elements)
        mx = key(max(elements, key=key))
        ties = [x for x in elements if key(x) == mx]
        return min(ties, key=lambda x: self.act(x[0]))

    def dirichlet(self, alphas):
        moves = list(self.mct.root.children.keys())
        assert len(moves) == len(alphas)
        eta = cf.hash_noise(self.root_board.grid, self.root_board.player, [self.act(m) for m in moves])
        return np.array([eta[self.act(m)] for m in moves], dtype=np.float64)

    def choice(self, n, p=None):
        assert p is not None, "unpatched uniform np.random.choice reached"
        moves = list(self.mct.root.children.keys())
        assert len(moves) == n
        order = sorted(range(n), key=lambda i: self.act(moves[i]))
        u = cf.move_sample_u(self.seed, self.episode, self.ply)
        cum, last = 0.0, None
        for i in order:
            if p[i] > 0:
                last = i
            cum += float(p[i])
            if u < cum:
                return i
        return last

    def __enter__(self):
        self._orig = dict(fm=self.R.mcts.fair_max, fmp=self.R.players.fair_max, dr=np.random.dirichlet, ch=np.random.choice)
        self.R.mcts.fair_max = self.fair_max_lowest
        self.R.players.fair_max = self.fair_max_lowest  # GreedyPlayer (players.py:97-123)
        np.random.dirichlet = self.dirichlet
        np.random.choice = self.choice
        return self

    def __exit__(self, *a):
        self.R.mcts.fair_max = self._orig["fm"]
        self.R.players.fair_max = self._orig["fmp"]
        np.random.dirichlet = self._orig["dr"]
        np.random.choice = self._orig["ch"]


def fake_net_class(R, tag):
    game, kw, A, n = GAMES[tag]
    base = type(make_net(R, tag))

    class FakeNet(base):
        def evaluate(self, board):  # base.py:357-367 with the closed-form net in place of predict()
            probs, v_net = cf.fakenet(board.grid, board.player, A)
            return probs, board.player * v_net
    if game == "othello":
        return FakeNet(n=n)
    if game == "connect4":
        return FakeNet(board_width=7, board_height=6)
    return FakeNet()


def root_stats(P, mct):
    rows = []
    for mv, node in mct.root.children.items():
        rows.append((P.act(mv), int(node.N), float(node.Q), float(node.P)))
    rows.sort()
    return rows


# ------------------------------------------------------------------------------------------- G3
def gen_mct(R, tag, positions, n_pos=32):
    game, kw, A, n = GAMES[tag]
    net = fake_net_class(R, tag)
    stages = [1, 1, 8, 90]  # cumulative 1, 2, 10, 100
    recs = {"grids": [], "players": [], "noise": [], "stage": [], "rootN": [], "action": [], "N": [], "Q": [], "P": [],
            "row_off": [0], "moved": []}
    pos = positions[64:64 + n_pos] if len(positions) >= 64 + n_pos else positions[:n_pos]
    with Patches(R, tag) as P:
        for (g, p) in pos:
            for noise in (0, 1):
                board = make_board(R, tag, grid=g, player=p)
                if board.is_game_over():
                    continue
                mct = R.mcts.MCT(eval_method="neural", nn=net,
                                 dirichlet_alpha=0.03 if noise else None, dirichlet_epsilon=0.25 if noise else None)
                P.mct, P.root_board = mct, board
                stage_id = 0
                for s in stages:
                    mct.search(board, n_sim=s)
                    rows = root_stats(P, mct)
                    recs["grids"].append(np.array(g, np.int8)); recs["players"].append(p); recs["noise"].append(noise)
                    recs["stage"].append(stage_id); recs["rootN"].append(mct.root.N); recs["moved"].append(-1)
                    for r in rows:
                        recs["action"].append(r[0]); recs["N"].append(r[1]); recs["Q"].append(r[2]); recs["P"].append(r[3])
                    recs["row_off"].append(len(recs["action"]))
                    stage_id += 1
                # tau = 0 move (lowest-index tie-break), tree reuse, 100 more simulations
                probs, visits = mct.get_action_probs(board, 0)
                move = list(probs.keys())[0]
                board.play_move(move)
                mct.change_root(move)
                if not board.is_game_over():
                    P.root_board = board
                    mct.search(board, n_sim=100)
                    rows = root_stats(P, mct)
                    recs["grids"].append(np.array(g, np.int8)); recs["players"].append(p); recs["noise"].append(noise)
                    recs["stage"].append(stage_id); recs["rootN"].append(mct.root.N); recs["moved"].append(P.act(move))
                    for r in rows:
                        recs["action"].append(r[0]); recs["N"].append(r[1]); recs["Q"].append(r[2]); recs["P"].append(r[3])
                    recs["row_off"].append(len(recs["action"]))
    np.savez_compressed(os.path.join(GOLD, f"mct_{tag}.npz"), grids=np.array(recs["grids"]),
                        players=np.array(recs["players"], np.int8), noise=np.array(recs["noise"], np.int8),
                        stage=np.array(recs["stage"], np.int8), rootN=np.array(recs["rootN"], np.int32),
                        moved=np.array(recs["moved"], np.int16), action=np.array(recs["action"], np.int16),
                        N=np.array(recs["N"], np.int32), Q=np.array(recs["Q"], np.float64),
                        P=np.array(recs["P"], np.float64), row_off=np.array(recs["row_off"], np.int32))
    print(f"mct_{tag}: {len(recs['stage'])} records, {len(recs['action'])} child rows")


# ------------------------------------------------------------------------------------------- G4
def gen_selfplay(R, tag, episodes, sims, seed=7, temp_steps=None, name="selfplay"):
    """temp_steps = (temp_max_step, temp_min_step) overrides the game's config (equal steps there: tau is 1, then 0); with
    temp_max_step < temp_min_step - 1 the linear schedule (schedulers.py:33-40) passes through fractional temperatures, where
    get_action_probs computes N ** (1 / tau) (mcts.py:114-116) -- the base defaults 15 / 20 (base.py:70-72) do"""
    game, kw, A, n = GAMES[tag]
    T = R.trainer
    cfg_cls = R.registers.CONFIGS_REGISTER[game]
    extra = {}
    if game == "othello":
        extra["board_size"] = n
    if temp_steps is not None:
        extra["temp_max_step"], extra["temp_min_step"] = temp_steps
    cfg = cfg_cls(simulations=sims, episodes=episodes, data_augmentation=True, **extra)
    tr = T.AlphaZeroTrainer(verbose=False)
    tr.config, tr.game = cfg, game
    tr.board = R.registers.BOARDS_REGISTER[game](config=cfg)
    tr.nn = fake_net_class(R, tag)
    tr.az_player = R.players.AlphaZeroPlayer(n_sim=sims, nn=tr.nn, dirichlet_alpha=cfg.dirichlet_alpha,
                                             dirichlet_epsilon=cfg.dirichlet_epsilon)
    tr.temp_scheduler = R.schedulers.TEMP_SCHEDULERS[cfg.temp_scheduler_type](
        temp_max_step=cfg.temp_max_step, temp_min_step=cfg.temp_min_step, max_steps=tr.board.max_moves)
    tr.data_augment_strategy = R.registers.DATA_AUGMENT_STRATEGIES[game]
    with Patches(R, tag) as P:
        P.seed, P.episode, P.ply = seed, -1, 0
        player = tr.az_player
        orig_reset, orig_get_move = player.reset, player.get_move

        def reset():
            orig_reset()
            P.episode += 1
            P.ply = 0

        def get_move(board, temp=0):
            P.mct, P.root_board = player.mct, board
            out = orig_get_move(board, temp=temp)
            P.ply += 1
            return out
        player.reset, player.get_move = reset, get_move
        tr.self_play(0)
    mem = tr.memory
    transf = sorted(set(str(s.transformation) for s in mem))
    np.savez_compressed(
        os.path.join(GOLD, f"{name}_{tag}.npz"),
        state=np.array([s.state for s in mem]).astype(np.int8), pi=np.array([s.pi for s in mem], np.float64),
        outcome=np.array([s.outcome for s in mem], np.int8), player=np.array([s.player for s in mem], np.int8),
        episode_idx=np.array([s.episode_idx for s in mem], np.int32), move_idx=np.array([s.move_idx for s in mem], np.int32),
        transformation=np.array([transf.index(str(s.transformation)) for s in mem], np.int8), transf_names=np.array(transf),
        sims=np.int32(sims), episodes=np.int32(episodes), seed=np.int32(seed), alpha=cfg.dirichlet_alpha, eps=cfg.dirichlet_epsilon,
        temp_max_step=np.int32(cfg.temp_max_step), temp_min_step=np.int32(cfg.temp_min_step))
    n_orig = sum(1 for s in mem if s.transformation is None)
    print(f"{name}_{tag}: {n_orig} samples (+{len(mem) - n_orig} augmented), transformations={transf}")


# ------------------------------------------------------------------------------------------- G7
ARENA_PLAN = {  # tag: [(opponent, sims of player 1, sims of the opponent, rounds, start_player)]
    "tictactoe": [("greedy", 20, 0, 6, None), ("alphazero", 20, 8, 4, None), ("greedy", 10, 0, 2, 2)],
    "connect4": [("greedy", 20, 0, 4, None), ("alphazero", 20, 8, 4, None), ("alphazero", 10, 10, 2, 1)],
    "othello6": [("greedy", 16, 0, 4, None), ("alphazero", 16, 6, 4, None), ("greedy", 8, 0, 2, 2)],
    "othello8": [("greedy", 10, 0, 2, None), ("alphazero", 10, 5, 2, None)],
}


def gen_arena(R, tag):
    """the reference's Arena.play_games, every move logged (Board.play_move of the arena's own board)"""
    game, kw, A, n = GAMES[tag]
    rec = {k: [] for k in ("opponent", "sims1", "sims2", "n_rounds", "start_player", "round_off", "moves", "move_off", "p2_starts", "winner_colour",
                           "score", "p1_scores", "p1_off", "p2_scores", "p2_off", "draws", "starts")}
    rec["round_off"].append(0); rec["move_off"].append(0); rec["p1_off"].append(0); rec["p2_off"].append(0)
    enc = lambda sc: 32767 if sc == float("inf") else int(sc)  # noqa: E731
    with Patches(R, tag) as P:
        for opp, s1, s2, rounds, start in ARENA_PLAN[tag]:
            p1 = R.players.AlphaZeroPlayer(n_sim=s1, nn=fake_net_class(R, tag))
            p2 = R.players.GreedyPlayer() if opp == "greedy" else R.players.AlphaZeroPlayer(n_sim=s2, nn=fake_net_class(R, tag))
            board = make_board(R, tag)
            arena = R.arena.Arena(p1, p2, board)
            log, finals = [], []
            orig_play, orig_game = board.play_move, arena.play_game

            def play_move(move, _o=orig_play, _l=log):
                _l.append(P.act(move))
                _o(move)

            def play_game(player2_starts=False, **kwargs):
                log.clear()
                res = orig_game(player2_starts=player2_starts, **kwargs)
                finals.append((list(log), bool(player2_starts), int(board.get_winner()), enc(abs(board.get_score()))))
                return res
            board.play_move, arena.play_game = play_move, play_game
            stats = arena.play_games(n_rounds=rounds, start_player=start, return_stats=True)
            assert len(finals) == rounds
            rec["opponent"].append(0 if opp == "greedy" else 1); rec["sims1"].append(s1); rec["sims2"].append(s2)
            rec["n_rounds"].append(rounds); rec["start_player"].append(0 if start is None else start)
            for moves, p2s, win, sc in finals:
                rec["moves"].extend(moves); rec["move_off"].append(len(rec["moves"]))
                rec["p2_starts"].append(p2s); rec["winner_colour"].append(win); rec["score"].append(sc)
            rec["round_off"].append(len(rec["p2_starts"]))
            rec["p1_scores"].extend(enc(x) for x in stats["player1"]); rec["p1_off"].append(len(rec["p1_scores"]))
            rec["p2_scores"].extend(enc(x) for x in stats["player2"]); rec["p2_off"].append(len(rec["p2_scores"]))
            rec["draws"].append(stats["draw"])
            rec["starts"].append([stats[f"player{p}_starts"].get(k, 0) for p in (1, 2) for k in ("win", "loss", "draw")])
    np.savez_compressed(os.path.join(GOLD, f"arena_{tag}.npz"),
                        **{k: np.array(v, np.int8 if k in ("opponent", "p2_starts", "winner_colour", "start_player") else np.int32) for k, v in rec.items()})
    print(f"arena_{tag}: {len(rec['n_rounds'])} pairings, {len(rec['p2_starts'])} rounds, {len(rec['moves'])} moves, winners {rec['winner_colour']}")


# ------------------------------------------------------------------------------------------- G6
SGD_PLAN = {"tictactoe": dict(batch_size=16, epochs=2, shuffle_seed=4242), "connect4": dict(batch_size=32, epochs=2, shuffle_seed=4243),
            "othello6": dict(batch_size=32, epochs=2, shuffle_seed=4244),
            # the BASELINE network at the reference's batch size (othello.py:35): one epoch = 14 steps over the 940 G4 samples
            "othello8": dict(batch_size=64, epochs=1, shuffle_seed=4245)}


def gen_sgd(R, tag):
    """the reference's optimisation loop on the committed G4 memory; everything random is pinned"""
    import torch
    torch.set_num_threads(1)
    game, kw, A, n = GAMES[tag]
    plan = SGD_PLAN[tag]
    fx = np.load(os.path.join(GOLD, f"selfplay_{tag}.npz"), allow_pickle=False)
    T = R.trainer
    extra = {"board_size": n} if game == "othello" else {}
    cfg = R.registers.CONFIGS_REGISTER[game](epochs=plan["epochs"], batch_size=plan["batch_size"], **extra)
    tr = T.AlphaZeroTrainer(verbose=False)
    tr.config, tr.game = cfg, game
    tr.board = R.registers.BOARDS_REGISTER[game](config=cfg)
    net = make_net(R, tag)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.tensor(v) for k, v in cf.closed_form_state_dict(shapes).items()})
    if hasattr(net, "dropout"):
        net.dropout = 0.0  # the only unpinnable randomness of the step (torch's dropout mask)
    tr.nn = net
    tr.memory = [T.Sample(state=fx["state"][i].astype(np.float64), pi=fx["pi"][i].copy(), player=1, outcome=int(fx["outcome"][i]),
                          episode_idx=int(fx["episode_idx"][i]), move_idx=int(fx["move_idx"][i])) for i in range(len(fx["outcome"]))]
    tr.loss_values = {}
    np.random.seed(plan["shuffle_seed"])
    tr.optimize_network(0)
    sd = tr.nn_twin.state_dict()
    out = {"batch_size": np.int32(plan["batch_size"]), "epochs": np.int32(plan["epochs"]), "shuffle_seed": np.int32(plan["shuffle_seed"]),
           "n_samples": np.int32(len(tr.memory)), "lr": np.float64(cfg.learning_rate),
           "fc1_weight": sd["fc1.weight"].numpy().astype(np.float32)[:64], "fc_value_weight": sd["fc_value.weight"].numpy().astype(np.float32),
           "bn_running_mean": sd[("fc_bn1" if game != "tictactoe" else "bn1") + ".running_mean"].numpy().astype(np.float32)}
    for e in range(plan["epochs"]):
        out[f"pi_loss_{e}"] = np.array(tr.loss_values[0][e]["pi"], np.float64)
        out[f"v_loss_{e}"] = np.array(tr.loss_values[0][e]["v"], np.float64)
    np.savez_compressed(os.path.join(GOLD, f"sgd_{tag}.npz"), **out)
    print(f"sgd_{tag}: {len(tr.memory)} samples, {len(out['pi_loss_0'])} steps/epoch, first losses pi={out['pi_loss_0'][0]:.6f} v={out['v_loss_0'][0]:.6f}, "
          f"last pi={out['pi_loss_%d' % (plan['epochs'] - 1)][-1]:.6f}")


# ------------------------------------------------------------------------------------------- G5
def gen_stats(R):
    """BASELINE config 1: TicTacToe, MCTSPlayer(n_sim=100) rollout mode, temp 0, self-play outcome mix."""
    np.random.seed(12345)
    n_games = 200
    res = {1: 0, -1: 0, 0: 0}
    plies = []
    for g in range(n_games):
        b = R.ttt.TicTacToeBoard()
        pl = R.players.MCTSPlayer(n_sim=100)
        k = 0
        while not b.is_game_over():
            mv, _, _, _ = pl.get_move(b, temp=0)
            b.play_move(mv); pl.apply_move(mv)
            k += 1
        res[b.get_winner()] += 1
        plies.append(k)
    np.savez_compressed(os.path.join(GOLD, "stats.npz"), ttt_rollout_games=np.int32(n_games),
                        ttt_rollout_p1=np.int32(res[1]), ttt_rollout_m1=np.int32(res[-1]), ttt_rollout_draw=np.int32(res[0]),
                        ttt_rollout_mean_plies=np.float64(np.mean(plies)))
    print("stats: tictactoe rollout self-play", res, "mean plies", np.mean(plies))


def main():
    want = set(sys.argv[1:])
    os.makedirs(GOLD, exist_ok=True)
    R = load_reference()
    plan = {"othello8": (256, 512), "othello6": (256, 512), "othello4": (128, 512), "connect4": (512, 256),
            "tictactoe": (512, 256)}
    sp = {"othello8": (2, 30), "othello6": (3, 25), "connect4": (3, 40), "tictactoe": (4, 25)}
    for tag, (ng, npos) in plan.items():
        if want and tag not in want:
            continue
        positions = gen_rules(R, tag, ng, seed=1000 + len(tag) + ng, n_positions=npos)
        if tag != "othello4":
            gen_net(R, tag, positions)
            gen_mct(R, tag, positions)
            gen_selfplay(R, tag, *sp[tag])
            gen_selfplay(R, tag, *sp[tag], seed=11, temp_steps=(2, 6), name="selfplay_frac")  # tau = 1, 1, 1, .75, .5, .25, 0 ...
    if not want or "stats" in want:
        gen_stats(R)
    for tag in SGD_PLAN:
        if not want or "sgd" in want or f"sgd_{tag}" in want:
            gen_sgd(R, tag)
    for tag in ARENA_PLAN:
        if not want or "arena" in want or f"arena_{tag}" in want:
            gen_arena(R, tag)


if __name__ == "__main__":
    main()
