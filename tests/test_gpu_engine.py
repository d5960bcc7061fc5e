"""GPU parity, tree search + self-play (K3/K4/K7/K8/K9) through the C ABI.

  * G3 golden trees of the reference's MCT (fake net, deterministic ties, +/- closed-form noise):
    visit counts exact, Q and P within 1e-12.
  * G4 golden AlphaZeroTrainer.self_play sample stream.
  * engine == CPU oracle on whole self-play runs in PRODUCTION mode (random ties, Philox Dirichlet
    noise, temperature sampling): every sample (state, pi, z, visits) bit-equal.
"""
import numpy as np
import pytest

from conftest import TAGS, golden
from oracle import oracle as O
from alphazero_amd import engine as E

pytestmark = pytest.mark.gpu
MCT_TAGS = ["othello8", "othello6", "connect4", "tictactoe"]


def sort_samples(d):
    """order by (game id, move idx) -- the engine emits samples in lock-step order"""
    meta = d["meta"] if isinstance(d["meta"], np.ndarray) else d["meta"].cpu().numpy()
    order = np.lexsort((meta[:, 1], meta[:, 0]))
    return {k: (v if isinstance(v, np.ndarray) else v.cpu().numpy())[order] for k, v in d.items() if k != "n_evals"}


@pytest.mark.parametrize("tag", MCT_TAGS)
def test_mct_fixture(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"mct_{tag}.npz")
    ro, stage = fx["row_off"], fx["stage"]
    starts = np.flatnonzero(stage == 0)
    for noise in (0, 1):
        cases = [s for s in starts if fx["noise"][s] == noise]
        eng = E.SelfPlayEngine(gid, H, W, n_slots=len(cases), n_sim=100, dirichlet_alpha=0.03 if noise else None,
                               dirichlet_epsilon=0.25 if noise else None, temp_max_step=-1, temp_min_step=0,
                               tie_mode=E.TIE_LOWEST, noise_mode=E.NOISE_HASH if noise else E.NOISE_OFF,
                               evaluator=E.EVAL_FAKE, node_capacity=8192)
        eng.set_roots(np.array([fx["grids"][s] for s in cases]), np.array([fx["players"][s] for s in cases]))

        def check(rec_of_case):
            for slot, rec in enumerate(rec_of_case):
                if rec is None:
                    continue
                a, N, Q, P, rootn = eng.root_children(slot)
                sl = slice(ro[rec], ro[rec + 1])
                assert np.array_equal(a, fx["action"][sl]), (tag, rec)
                assert np.array_equal(N, fx["N"][sl]), (tag, rec, N, fx["N"][sl])
                assert rootn == fx["rootN"][rec]
                assert np.abs(Q - fx["Q"][sl]).max() <= 1e-12
                assert np.abs(P - fx["P"][sl]).max() <= 1e-12
        for k, sims in enumerate((1, 1, 8, 90)):
            eng.search(sims)
            check([s + k for s in cases])
        eng.advance()  # tau = 0 move (lowest-index tie-break), tree reuse
        eng.search(100)
        last = []
        for s in cases:
            r = s + 4
            last.append(r if r < len(stage) and stage[r] == 4 else None)
        check(last)
        smp = sort_samples(eng.samples())
        moved = [fx["moved"][r] for r in last if r is not None]
        got = [smp["meta"][i, 3] for i, r in enumerate(last) if r is not None]
        assert moved == got
        eng.close()


@pytest.mark.parametrize("tag", MCT_TAGS)
def test_selfplay_fixture(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"selfplay_{tag}.npz")
    names = [str(x) for x in fx["transf_names"]]
    orig = fx["transformation"] == names.index("None")
    eng = E.SelfPlayEngine(gid, H, W, n_slots=2, n_sim=int(fx["sims"]), dirichlet_alpha=float(fx["alpha"]),
                           dirichlet_epsilon=float(fx["eps"]), temp_max_step=int(fx["temp_max_step"]),
                           temp_min_step=int(fx["temp_min_step"]), tie_mode=E.TIE_LOWEST, noise_mode=E.NOISE_HASH,
                           evaluator=E.EVAL_FAKE, seed=int(fx["seed"]), node_capacity=16384, sample_capacity=4096)
    r = sort_samples(eng.run(int(fx["episodes"])))  # 2 slots, >2 episodes: exercises slot refill
    assert np.array_equal(r["state"], fx["state"][orig])
    assert np.array_equal(r["z"], fx["outcome"][orig])
    assert np.array_equal(r["meta"][:, 0], fx["episode_idx"][orig])
    assert np.array_equal(r["meta"][:, 1], fx["move_idx"][orig])
    assert np.abs(r["pi"].astype(np.float64) - fx["pi"][orig]).max() < 1e-7
    assert eng.stats()["games_done"] == int(fx["episodes"])


@pytest.mark.parametrize("tag,n_games,n_sim,slots", [("othello8", 96, 40, 64), ("othello6", 128, 30, 50),
                                                     ("connect4", 160, 50, 64), ("tictactoe", 256, 30, 100)])
def test_production_mode_equals_oracle_fakenet(tag, n_games, n_sim, slots):
    game, gid, H, W, A, n = TAGS[tag]
    eng = E.SelfPlayEngine(gid, H, W, n_slots=slots, n_sim=n_sim, evaluator=E.EVAL_FAKE, seed=9, node_capacity=32768,
                           sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games, first_game_id=1000))
    ref = O.selfplay(gid, H, W, n_games, n_sim, ("fake", None), seed=9, first_game_id=1000)
    st = eng.stats()
    assert st["games_done"] == n_games and st["samples"] == len(ref["z"]) and st["net_evals"] == ref["n_evals"]
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


def test_capacity_errors_are_loud():
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=8, n_sim=50, evaluator=E.EVAL_FAKE, node_capacity=256)
    with pytest.raises(E._lib.AzError, match="node pool"):
        eng.run(8)
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=8, n_sim=10, evaluator=E.EVAL_FAKE, sample_capacity=20)
    with pytest.raises(E._lib.AzError, match="sample buffer"):
        eng.run(8)


@pytest.mark.parametrize("tag,n_games,n_sim", [("othello8", 12, 25), ("connect4", 24, 40), ("tictactoe", 64, 25)])
def test_production_mode_equals_oracle_real_net(tag, n_games, n_sim):
    """end to end with the HIP network: engine samples == oracle samples (same weights, same accumulation order)"""
    from test_gpu_net import nets
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, hnet = nets(tag)
    eng = E.SelfPlayEngine(gid, H, W, n_slots=8, n_sim=n_sim, net=hnet, seed=21, node_capacity=32768,
                           sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games))
    kind = "mlp" if game == "tictactoe" else "conv"
    ref = O.selfplay(gid, H, W, n_games, n_sim, (kind, onet), seed=21)
    assert len(got["z"]) == len(ref["z"])
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("tag,n_games,n_sim,slots", [("othello8", 24, 60, 16), ("othello6", 40, 50, 40),
                                                     ("connect4", 48, 80, 32), ("tictactoe", 128, 100, 64)])
def test_rollout_mode_equals_oracle(tag, n_games, n_sim, slots):
    """TreeEval.ROLLOUT (UCT + random playouts, no network) on the device == the oracle, sample for sample"""
    game, gid, H, W, A, n = TAGS[tag]
    eng = E.SelfPlayEngine(gid, H, W, n_slots=slots, n_sim=n_sim, evaluator=E.EVAL_ROLLOUT, seed=5, node_capacity=16384,
                           sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games, first_game_id=77))
    ref = O.selfplay(gid, H, W, n_games, n_sim, ("fake", None), seed=5, first_game_id=77, eval_method=O.EVAL_ROLLOUT)
    assert eng.stats()["games_done"] == n_games and len(got["z"]) == len(ref["z"])
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k
