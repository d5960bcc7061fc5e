"""CPU, only where the Python reference is mounted (/root/reference: this container, never the GPU box): the host-side
mirror classes played side by side with the reference's own classes, and the committed golden fixtures regenerated
from the reference and compared with the files in tests/golden/ (they are the reference's outputs, not ours)."""
import os

import numpy as np
import pytest
import torch

REF = "/root/reference/alphazero"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted")


@pytest.fixture(scope="module")
def R():
    from tools import gen_golden
    return gen_golden.load_reference()


def _boards(R, tag):
    from alphazero_amd.games.connect4 import Connect4Board
    from alphazero_amd.games.othello import OthelloBoard
    from alphazero_amd.games.tictactoe import TicTacToeBoard
    if tag.startswith("othello"):
        n = int(tag[-1])
        return R.oth.OthelloBoard(n=n), OthelloBoard(n=n)
    if tag.startswith("connect4"):
        w, h = (7, 6) if tag == "connect4" else map(int, tag.split("_")[1].split("x"))
        return R.c4.Connect4Board(width=w, height=h), Connect4Board(width=w, height=h)
    return R.ttt.TicTacToeBoard(), TicTacToeBoard()


@pytest.mark.parametrize("tag", ["othello8", "othello6", "othello4", "connect4", "connect4_5x6", "connect4_8x8", "connect4_4x4", "tictactoe"])
def test_boards_side_by_side(R, tag):
    rng = np.random.RandomState(7)
    for game in range(12):
        ref, mine = _boards(R, tag)
        while True:
            assert ref.is_game_over() == mine.is_game_over()
            assert np.array_equal(np.asarray(ref.grid), np.asarray(mine.grid)) and ref.player == mine.player
            assert ref.get_score() == mine.get_score()
            if ref.is_game_over():
                assert ref.get_winner() == mine.get_winner()
                break
            a, b = ref.get_moves(), mine.get_moves()
            key = (lambda m: int(m)) if tag.startswith("connect4") else (lambda m: (int(m[0]), int(m[1])))
            assert sorted(map(key, a)) == sorted(map(key, b))
            assert list(map(key, a)) == list(map(key, b))  # the same ORDER too: a seeded np.random picks the same get_random_move on both
            other_a, other_b = ref.get_moves(player=-ref.player), mine.get_moves(player=-mine.player)
            assert sorted(map(key, other_a)) == sorted(map(key, other_b))
            mv = a[rng.randint(len(a))]
            assert ref.is_legal_move(mv) and mine.is_legal_move(mv)
            ref.play_move(mv)
            mine.play_move(mv)


@pytest.mark.parametrize("tag", ["othello8", "connect4", "tictactoe"])
def test_networks_side_by_side(R, tag):
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    from alphazero_amd.games.tictactoe import TicTacToeNet
    torch.manual_seed(4)
    ref = {"othello8": lambda: R.oth.OthelloNet(n=8), "connect4": lambda: R.c4.Connect4Net(board_width=7, board_height=6),
           "tictactoe": lambda: R.ttt.TicTacToeNet()}[tag]()
    torch.manual_seed(4)
    mine = {"othello8": lambda: OthelloNet(n=8), "connect4": lambda: Connect4Net(7, 6), "tictactoe": lambda: TicTacToeNet()}[tag]()
    sr, sm = ref.state_dict(), mine.state_dict()
    assert list(sr.keys()) == list(sm.keys())
    assert all(torch.equal(sr[k], sm[k]) for k in sr)  # same layer order -> same default initialisation under one seed
    ref.eval(); mine.eval()
    rb, mb = _boards(R, tag)
    x = torch.tensor(np.stack([np.asarray(rb.grid, dtype=np.float32)] * 3))
    lp_r, v_r = ref(x)
    lp_m, v_m = mine(x)
    assert torch.allclose(lp_r, lp_m, atol=1e-6) and torch.allclose(v_r, v_m, atol=1e-6)
    pr, vr = ref.evaluate(rb)  # the mirror's evaluate() runs on the HIP engine: compared in the GPU tests through G2
    assert pr.shape[0] == mine.action_size and np.allclose(pr, np.exp(lp_m[0].detach().numpy()), atol=1e-6)


@pytest.mark.parametrize("tag,n_games,n_pos,sp", [("tictactoe", 512, 256, (4, 25)), ("othello6", 256, 512, (3, 25))])
def test_committed_fixtures_are_what_the_reference_produces(R, tag, n_games, n_pos, sp, tmp_path, monkeypatch):
    """tools/gen_golden.py re-run against the mounted reference (same plan as its main()) reproduces tests/golden/"""
    from tools import gen_golden
    monkeypatch.setattr(gen_golden, "GOLD", str(tmp_path))
    positions = gen_golden.gen_rules(R, tag, n_games, seed=1000 + len(tag) + n_games, n_positions=n_pos)
    gen_golden.gen_net(R, tag, positions)
    gen_golden.gen_mct(R, tag, positions)
    gen_golden.gen_selfplay(R, tag, *sp)
    gen_golden.gen_selfplay(R, tag, *sp, seed=11, temp_steps=(2, 6), name="selfplay_frac")  # fractional temperatures
    gen_golden.gen_sgd(R, tag)  # G6: the reference's optimize_network on the memory just regenerated
    for kind in ("rules", "net", "mct", "selfplay", "selfplay_frac", "sgd"):
        name = f"{kind}_{tag}.npz"
        new = np.load(os.path.join(tmp_path, name), allow_pickle=False)
        old = np.load(os.path.join(os.path.dirname(__file__), "golden", name), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            if old[k].dtype.kind == "f":
                assert np.allclose(new[k], old[k], rtol=0, atol=1e-6), (name, k)
            else:
                assert np.array_equal(new[k], old[k]), (name, k)


@pytest.mark.parametrize("tag", ["tictactoe", "connect4", "othello6", "othello8"])
def test_committed_arena_fixtures_are_what_the_reference_produces(R, tag, tmp_path, monkeypatch):
    """golden G7 (Arena.play_games of the reference) re-generated and compared"""
    from tools import gen_golden
    monkeypatch.setattr(gen_golden, "GOLD", str(tmp_path))
    gen_golden.gen_arena(R, tag)
    new = np.load(os.path.join(tmp_path, f"arena_{tag}.npz"), allow_pickle=False)
    old = np.load(os.path.join(os.path.dirname(__file__), "golden", f"arena_{tag}.npz"), allow_pickle=False)
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        assert np.array_equal(new[k], old[k]), (tag, k)


def test_committed_sgd_fixture_of_the_baseline_network_is_what_the_reference_produces(R, tmp_path, monkeypatch):
    """golden G6 for OthelloNet 8x8: the reference's optimize_network on the committed G4 memory"""
    import shutil
    from tools import gen_golden
    gold = os.path.join(os.path.dirname(__file__), "golden")
    shutil.copy(os.path.join(gold, "selfplay_othello8.npz"), tmp_path)
    monkeypatch.setattr(gen_golden, "GOLD", str(tmp_path))
    gen_golden.gen_sgd(R, "othello8")
    new = np.load(os.path.join(tmp_path, "sgd_othello8.npz"), allow_pickle=False)
    old = np.load(os.path.join(gold, "sgd_othello8.npz"), allow_pickle=False)
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        assert np.allclose(new[k], old[k], rtol=0, atol=1e-6), k


def test_checkpoints_cross_load(R, tmp_path):
    """checkpoint wire format (base.py:288-325, trainer.py:448-473): models/<name>/{config.json,<name>.pt} written by either
    side load into the other and give the same network"""
    import json
    from alphazero_amd.games.othello import OthelloConfig, OthelloNet
    torch.manual_seed(2)
    mine = OthelloNet(config=OthelloConfig(board_size=6))
    d = os.path.join(tmp_path, "m1")
    mine.save_model("m1", model_path=d)
    json.dump(OthelloConfig(board_size=6).to_dict(), open(os.path.join(d, "config.json"), "w"))
    ref = R.oth.OthelloNet.from_pretrained("m1", models_path=str(tmp_path))
    assert all(torch.equal(v, ref.state_dict()[k]) for k, v in mine.state_dict().items())
    # the other way round: the reference writes, the mirror reads
    torch.manual_seed(3)
    ref2 = R.oth.OthelloNet(config=R.oth.OthelloConfig(board_size=6))
    d2 = os.path.join(tmp_path, "m2")
    ref2.save_model("m2", model_path=d2)
    json.dump(R.oth.OthelloConfig(board_size=6).to_dict(), open(os.path.join(d2, "config.json"), "w"))
    mine2 = OthelloNet.from_pretrained("m2", models_path=str(tmp_path))
    assert all(torch.equal(v, mine2.state_dict()[k]) for k, v in ref2.state_dict().items())
    assert mine2.n == 6 and sorted(OthelloConfig(board_size=6).to_dict()) == sorted(R.oth.OthelloConfig(board_size=6).to_dict())


# ---------------------------------------------------------------------------------------------- the public surface
# What the mirror deliberately does not carry, by name -- everything else public in the in-scope modules must exist here with the
# reference's parameter names in the reference's order (extra trailing parameters need defaults):
SURFACE_EXCEPTIONS = {
    # rendering, the interactive player, HF hub and command lines: out of scope
    "players.HumanPlayer", "utils.get_hf_token", "utils.list_models_from_hf_hub", "utils.download_model_from_hf_hub",
    "utils.download_all_models_from_hf_hub", "utils.push_model_to_hf_hub", "utils.get_rgb_code", "timers.demo", "trainer.tests",
    # the host-side tree of the reference: here the tree lives in HBM and these four steps are the device kernels k_step / k_rollout_step
    # (csrc/az_engine.hip); MCT keeps search / get_action_probs / get_prior_probs / change_root / reset, which is what Players call
    "mcts.Node", "mcts.MCT.select_node", "mcts.MCT.rollout", "mcts.MCT.nn_evaluation", "mcts.MCT.back_propagate",
}
SURFACE_MODULES = ["base", "arena", "mcts", "players", "trainer", "schedulers", "timers", "utils", "games.othello", "games.connect4",
                   "games.tictactoe", "games.registers"]


def _functions(cls):
    import inspect
    out = {}
    for name, attr in vars(cls).items():
        if name.startswith("_") and name != "__init__":  # private helpers (name-mangled or not) are not surface
            continue
        f = attr.__func__ if isinstance(attr, (staticmethod, classmethod)) else attr
        if inspect.isfunction(f):
            out[name] = (f, type(attr).__name__ if isinstance(attr, (staticmethod, classmethod)) else "function")
    return out


def test_public_surface_matches(R):
    import importlib
    import inspect
    problems = []

    def compare(label, rf, mf):
        rp, mp = inspect.signature(rf).parameters, inspect.signature(mf).parameters
        rn, mn = list(rp), list(mp)
        if mn[:len(rn)] != rn:
            problems.append(f"{label}: reference {rn}, mirror {mn}")
            return
        for extra in mn[len(rn):]:
            if mp[extra].default is inspect.Parameter.empty and mp[extra].kind not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL):
                problems.append(f"{label}: extra mirror parameter {extra} has no default")
        for name in rn:
            if (rp[name].default is inspect.Parameter.empty) != (mp[name].default is inspect.Parameter.empty) and name != "self":
                problems.append(f"{label}: parameter {name} is {'required' if rp[name].default is inspect.Parameter.empty else 'optional'} in the reference")
            elif rp[name].default is not inspect.Parameter.empty:
                rd, md = rp[name].default, mp[name].default
                if isinstance(rd, (bool, int, float, str, type(None))) and (type(rd) is not type(md) or rd != md):  # literal defaults must be the reference's
                    problems.append(f"{label}: parameter {name} defaults to {rd!r} in the reference, {md!r} here")

    for mod in SURFACE_MODULES:
        ref, mine = importlib.import_module("alphazero." + mod), importlib.import_module("alphazero_amd." + mod)
        for name, obj in vars(ref).items():
            if getattr(obj, "__module__", None) != ref.__name__ or name.startswith("_") or name == "main":
                continue
            label = f"{mod}.{name}"
            if label in SURFACE_EXCEPTIONS:
                continue
            if inspect.isfunction(obj):
                if not hasattr(mine, name):
                    problems.append(f"{label}: missing")
                else:
                    compare(label, obj, getattr(mine, name))
            elif inspect.isclass(obj):
                if not hasattr(mine, name):
                    problems.append(f"{label}: missing")
                    continue
                theirs, ours = _functions(obj), getattr(mine, name)
                for meth, (rf, kind) in theirs.items():
                    ml = f"{label}.{meth}"
                    if ml in SURFACE_EXCEPTIONS or meth in ("human_display", "pixel_display"):  # rendering: the base class refuses both
                        continue
                    attr = inspect.getattr_static(ours, meth, None)
                    if attr is None:
                        problems.append(f"{ml}: missing")
                        continue
                    mkind = type(attr).__name__ if isinstance(attr, (staticmethod, classmethod)) else "function"
                    mf = attr.__func__ if isinstance(attr, (staticmethod, classmethod)) else attr
                    if not inspect.isfunction(mf):
                        problems.append(f"{ml}: not a function here")
                    elif mkind != kind:
                        problems.append(f"{ml}: {kind} in the reference, {mkind} here")
                    else:
                        compare(ml, rf, mf)
    assert not problems, "\n".join(problems)


def test_abstract_hooks_raise_like_the_reference(R):
    """base.py:370-397: a hook a concrete network does not define is a NotImplementedError, not an AttributeError"""
    from alphazero_amd.arena import Arena
    from alphazero_amd.games.connect4 import Connect4Net
    import inspect
    with pytest.raises(NotImplementedError):
        Connect4Net(7, 6).rotate_neural_output(np.zeros(7), 1)
    with pytest.raises(NotImplementedError):
        R.c4.Connect4Net(7, 6).rotate_neural_output(np.zeros(7), 1)
    # a positional call binds as in the reference: play_game(False, False, False, True) asks for the results
    assert list(inspect.signature(Arena.play_game).parameters)[:5] == ["self", "player2_starts", "display", "save_frames", "return_results"]


def test_config_dataclasses_have_the_references_fields_and_defaults(R):
    """base.py:60-92 and games/*.py: the JSON configs of the reference load unchanged only if every field exists here under the same
    name with the same default, in the same order (positional construction)"""
    import dataclasses
    from alphazero_amd import base as mbase
    from alphazero_amd.games import connect4 as mc4, othello as moth, tictactoe as mttt
    import alphazero.base as rbase
    pairs = [(rbase.Config, mbase.Config), (R.oth.OthelloConfig, moth.OthelloConfig), (R.c4.Connect4Config, mc4.Connect4Config),
             (R.ttt.TicTacToeConfig, mttt.TicTacToeConfig)]
    for ref, mine in pairs:
        rf = [(f.name, f.default) for f in dataclasses.fields(ref)]
        mf = [(f.name, f.default) for f in dataclasses.fields(mine)]
        assert rf == mf, (ref.__name__, [x for x in rf if x not in mf], [x for x in mf if x not in rf])
        assert dict(ref().to_dict()) == dict(mine().to_dict())


def test_registers_have_the_references_keys(R):
    """games/registers.py: the same games, the same per-game entries (class names) and the same augmentation strategies"""
    import alphazero_amd.games.registers as mreg
    rreg = R.registers
    for name in ("GAMES_SET", "CONFIGS_REGISTER", "BOARDS_REGISTER", "NETWORKS_REGISTER", "DATA_AUGMENT_STRATEGIES"):
        a, b = getattr(rreg, name), getattr(mreg, name)
        assert set(a) == set(b), name
        if name.endswith("_REGISTER"):
            assert {k: v.__name__ for k, v in a.items()} == {k: v.__name__ for k, v in b.items()}, name
    for game, strat in rreg.DATA_AUGMENT_STRATEGIES.items():
        mine = mreg.DATA_AUGMENT_STRATEGIES[game]
        flat = lambda d: {k: [t.value for t in (v if isinstance(v, list) else [v])] for k, v in d.items()}
        assert flat(strat) == flat(mine), game
    assert {k: v.value for k, v in rreg.MOVE_FORMATS_REGISTER.items()} == {k: v.value for k, v in mreg.MOVE_FORMATS_REGISTER.items()}
