"""GPU parity of the code paths no fixture reaches by accident ("What's weak" 1-2), all against the pinned
CPU oracle through the C ABI:

  * self-play at the BASELINE simulation counts (Othello 8x8 @100, Connect4 @200) with the real network, sample for sample;
  * root..leaf paths LONGER than 16 nodes (the parent-chasing back-propagation and the path-overflow lanes);
  * the uniform-prior fallback of get_normalized_probs (othello.py:395-397) on every expansion;
  * a root with more than 32 children (third round of the 16-lane PUCT scan);
  * weight re-upload into a live engine whose search runs as a captured HIP graph (TicTacToe MLP and conv nets);
  * BatchNorm fold + re-tiling on the device == on the host;
  * the refusal of plane shapes without a trunk kernel.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import TAGS
from oracle import oracle as O
from alphazero_amd import engine as E
from test_gpu_engine import sort_samples

pytestmark = pytest.mark.gpu


def _np_sd(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")}


# ------------------------------------------------------------------------------------------------ BASELINE sim counts
@pytest.mark.parametrize("name,gid,H,W,games,sims", [("othello8", 0, 8, 8, 8, 100), ("connect4", 1, 6, 7, 16, 200)])
def test_selfplay_at_baseline_simulation_counts_equals_oracle(name, gid, H, W, games, sims):
    """the trimmed soak that always runs (tests/test_gpu_soak.py is the long opt-in version): random-init reference
    architecture, production mode (random ties, Philox Dirichlet noise, temperature sampling, tree reuse)"""
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(1)
    net = (OthelloNet(n=8) if gid == 0 else Connect4Net(7, 6)).eval()
    onet = O.ConvNet(gid, H, W, _np_sd(net))
    eng = E.SelfPlayEngine(gid, H, W, n_slots=games, n_sim=sims, net=net.to_hip(max_batch=games), seed=77)
    got = sort_samples(eng.run(games, first_game_id=123))
    ref = O.selfplay(gid, H, W, games, sims, ("conv", onet), seed=77, first_game_id=123)
    assert eng.stats()["net_evals"] == ref["n_evals"]
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("name,gid,H,W,slots,sims,picks", [("othello8", 0, 8, 8, 4096, 25, (9000, 11047, 13095)),
                                                           ("connect4", 1, 6, 7, 4096, 40, (9000, 10021, 13095)),
                                                           # the headline's slot count: k_gemm_solo, its tile height following the rows
                                                           ("othello8", 0, 8, 8, 32768, 20, (9000, 25383, 41767))])
def test_selfplay_on_the_large_batch_kernels_sampled_games_equal_oracle(name, gid, H, W, slots, sims, picks):
    """the kernels the benchmark runs on (two-boards-per-wave trunk with the Winograd conv2, tiled GEMMs, k_heads2 / fused tail, all
    behind the device row counter) only start at 4096 rows, where the oracle cannot follow every game.  Games are independent given
    (seed, game id): a whole wave of 4096 (32768) games is played on the engine and three of them are replayed alone on the oracle --
    samples, visit counts and policies of those games must be equal bit for bit."""
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(1)
    net = (OthelloNet(n=8) if gid == 0 else Connect4Net(7, 6)).eval()
    onet = O.ConvNet(gid, H, W, _np_sd(net))
    eng = E.SelfPlayEngine(gid, H, W, n_slots=slots, n_sim=sims, net=net.to_hip(max_batch=slots), seed=41)
    got = sort_samples(eng.run(slots, first_game_id=9000))
    assert eng.stats()["games_done"] == slots
    for g in picks:
        ref = O.selfplay(gid, H, W, 1, sims, ("conv", onet), seed=41, first_game_id=g)
        rows = got["meta"][:, 0] == g
        assert rows.sum() == len(ref["meta"]) > 0, g
        for k in ("state", "z", "meta", "visits", "pi"):
            assert np.array_equal(got[k][rows], ref[k]), (g, k)


# ------------------------------------------------------------------------------------------------ ragged slot counts
@pytest.mark.parametrize("slots", [1, 2, 3, 5, 6, 7, 13, 21, 30])
def test_slot_counts_that_do_not_fill_a_wavefront(slots):
    """16 lanes per game, 4 games per wavefront, 16 per workgroup: a slot count that leaves the last wavefront partly
    empty once handed the leaf rows out before every game had asked for one (barriers in a divergent branch; found by
    the uniform-prior test below running 6 slots).  Real network: the values of the affected slots went stale."""
    from test_gpu_net import nets
    fx, sd, onet, hnet = nets("othello6")
    games = slots + 3
    eng = E.SelfPlayEngine(0, 6, 6, n_slots=slots, n_sim=20, net=hnet, seed=31, sample_capacity=games * 72)
    got = sort_samples(eng.run(games, first_game_id=7))
    ref = O.selfplay(O.OTHELLO, 6, 6, games, 20, ("conv", onet), seed=31, first_game_id=7)
    assert eng.stats()["net_evals"] == ref["n_evals"]
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


# ------------------------------------------------------------------------------------------------ deep paths
def _late_positions(gid, H, W, seed, plies_lo, plies_hi, want, n_sim, tie, noise, alpha, eps):
    """random late positions whose oracle search walks paths of more than 16 nodes"""
    rng = np.random.default_rng(seed)
    found = []
    for _ in range(400):
        b = O.new_board(gid, H, W)
        ok = True
        for _ in range(int(rng.integers(plies_lo, plies_hi))):
            O.lib().orc_play(C.byref(b), int(rng.choice(O.legal_moves(b))))
            if O.lib().orc_is_over(C.byref(b)):
                ok = False
                break
        if not ok:
            continue
        t = O.MCT(("fake", None), alpha=alpha, eps=eps, tie_mode=tie, noise_mode=noise, seed=13, game_id=len(found))
        t.search(b, n_sim)
        if t.max_path_len() > 16:
            found.append((b.grid_np().copy(), int(b.player), t))
            if len(found) == want:
                break
    return found


@pytest.mark.parametrize("tag,plies_lo,plies_hi,n_sim", [("connect4", 10, 24, 3000), ("othello8", 30, 46, 4000)])
@pytest.mark.parametrize("production", [False, True])
def test_paths_longer_than_16_nodes_equal_oracle(tag, plies_lo, plies_hi, n_sim, production):
    game, gid, H, W, A, n = TAGS[tag]
    tie, noise = (E.TIE_RANDOM, E.NOISE_PHILOX) if production else (E.TIE_LOWEST, E.NOISE_OFF)
    alpha, eps = (0.03, 0.25) if production else (-1.0, -1.0)
    cases = _late_positions(gid, H, W, 1, plies_lo, plies_hi, 4, n_sim, tie, noise, alpha, eps)
    assert len(cases) >= 2, "no deep positions found: the test would prove nothing"
    eng = E.SelfPlayEngine(gid, H, W, n_slots=len(cases), n_sim=n_sim, dirichlet_alpha=alpha if production else None,
                           dirichlet_epsilon=eps if production else None, temp_max_step=-1, temp_min_step=0, tie_mode=tie,
                           noise_mode=noise, evaluator=E.EVAL_FAKE, seed=13, node_capacity=1 << 18)
    eng.set_roots(np.array([c[0] for c in cases]), np.array([c[1] for c in cases]))
    eng.search(n_sim)
    deepest = max(c[2].max_path_len() for c in cases)
    assert deepest > 16 and eng.stats()["max_path_len"] == deepest
    for slot, (_, _, t) in enumerate(cases):
        a, N, Q, P, rootn = eng.root_children(slot)
        oa, oN, oQ, oP = t.root_children()
        assert np.array_equal(a, oa) and np.array_equal(N, oN) and rootn == t.root_n() == n_sim
        assert np.array_equal(Q, oQ) and np.array_equal(P, oP)  # float64 statistics, same operation order: same bits
    # the move after a deep search: the kept subtree (its parent links feed the fallback) must survive the compaction
    eng.advance()
    eng.search(200)
    moved = sort_samples(eng.samples())["meta"][:, 3]
    for slot, (grid, player, t) in enumerate(cases):
        b = O.new_board(gid, H, W)
        b.set_grid(grid, player)
        act, _, _ = t.choose(b, 0.0)
        assert act == moved[slot]
        O.lib().orc_play(C.byref(b), act)
        if O.lib().orc_is_over(C.byref(b)):
            continue
        t.change_root(act)
        t.set_ply(1)
        t.search(b, 200)
        a, N, Q, P, rootn = eng.root_children(slot)
        oa, oN, oQ, oP = t.root_children()
        assert np.array_equal(N, oN) and np.array_equal(Q, oQ) and rootn == t.root_n()


# ------------------------------------------------------------------------------------------------ > 32 children
MANY_MOVES = np.array([[0, 0, 0, 0, 0, 0, 0, 0],
                       [0, -1, -1, -1, 1, -1, -1, 0],
                       [0, 1, 1, 0, -1, 1, 0, 1],
                       [0, -1, -1, 1, 0, 0, -1, 0],
                       [0, 0, 1, 0, 1, 0, -1, 0],
                       [0, -1, 0, 0, -1, 1, 1, 1],
                       [0, -1, -1, 0, -1, 1, -1, 0],
                       [0, -1, 1, 0, 0, 0, 0, 0]], np.int8)  # 34 legal moves for player +1 (found by hill climbing)


@pytest.mark.parametrize("production", [False, True])
def test_root_with_more_than_32_children(production):
    b = O.new_board(O.OTHELLO, 8, 8)
    b.set_grid(MANY_MOVES, 1)
    assert len(O.legal_moves(b)) == 34
    tie, noise = (E.TIE_RANDOM, E.NOISE_PHILOX) if production else (E.TIE_LOWEST, E.NOISE_HASH)
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=2, n_sim=400, dirichlet_alpha=0.03, dirichlet_epsilon=0.25, temp_max_step=-1,
                           temp_min_step=0, tie_mode=tie, noise_mode=noise, evaluator=E.EVAL_FAKE, seed=3, node_capacity=1 << 16)
    eng.set_roots(np.stack([MANY_MOVES, MANY_MOVES]), np.array([1, 1], np.int8), game_ids=np.array([5, 6], np.uint32))
    trees = [O.MCT(("fake", None), alpha=0.03, eps=0.25, tie_mode=tie, noise_mode=noise, seed=3, game_id=g) for g in (5, 6)]
    # one search of 400 simulations (the engine continues its Philox simulation counter across searches on one root,
    # the oracle restarts it: only single searches compare in production mode); 400 visit every one of the 34 children
    eng.search(400)
    for slot, t in enumerate(trees):
        t.search(b, 400)
        a, N, Q, P, rootn = eng.root_children(slot)
        oa, oN, oQ, oP = t.root_children()
        assert len(a) == 34 and np.array_equal(a, oa) and np.array_equal(N, oN), (N, oN)
        assert (N[32:] > 0).all(), "children 33 and 34 were never selected"
        assert np.array_equal(Q, oQ) and np.array_equal(P, oP)


# ------------------------------------------------------------------------------------------------ uniform-prior fallback
def test_uniform_prior_fallback_everywhere():
    """a policy head that puts all its mass on a cell that is never legal (a starting disc): the priors of the legal
    moves sum to less than 1e-6 at EVERY node, so every expansion takes the uniform branch (othello.py:395-397)"""
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(4)
    net = OthelloNet(n=8).eval()
    with torch.no_grad():
        net.fc_probs.bias[3 * 8 + 3] += 60.0
    hnet = net.to_hip(max_batch=16)
    x = torch.zeros((1, 64), device="cuda")
    x[0, 27] = x[0, 36] = 1.0
    x[0, 28] = x[0, 35] = -1.0
    p, _ = hnet.forward(x)
    assert float(p[0].sum() - p[0, 27]) < 1e-6  # everything but the occupied cell: below the threshold
    onet = O.ConvNet(O.OTHELLO, 8, 8, _np_sd(net))
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=6, n_sim=30, net=hnet, seed=8)
    got = sort_samples(eng.run(6))
    ref = O.selfplay(O.OTHELLO, 8, 8, 6, 30, ("conv", onet), seed=8)
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k
    # uniform priors: after 2 simulations from the start position the 4 root children hold P = (1-eps)/4 + eps*eta
    eng2 = E.SelfPlayEngine(0, 8, 8, n_slots=1, n_sim=2, net=hnet, dirichlet_alpha=None, dirichlet_epsilon=None, noise_mode=E.NOISE_OFF)
    b = O.new_board(O.OTHELLO, 8, 8)
    eng2.set_roots(b.grid_np()[None], np.array([1], np.int8))
    eng2.search(2)
    assert np.array_equal(eng2.root_children(0)[3], np.full(4, 0.25))


# ------------------------------------------------------------------------------------------------ weights into a live engine
@pytest.mark.parametrize("tag", ["tictactoe", "othello6"])
def test_reloaded_weights_reach_a_graph_replaying_engine(tag):
    """the TicTacToe MLP used to travel as a by-value kernel argument, frozen into the captured HIP
    graph of the search -- a re-upload changed nothing.  Three runs on one engine (the third search of a shape is the
    first graph replay), new weights before the last: must equal the oracle with the NEW weights."""
    from alphazero_amd.games.othello import OthelloNet
    from alphazero_amd.games.tictactoe import TicTacToeNet
    game, gid, H, W, A, n = TAGS[tag]
    mk = (lambda: TicTacToeNet()) if game == "tictactoe" else (lambda: OthelloNet(n=6))
    torch.manual_seed(10)
    net_a = mk().eval()
    torch.manual_seed(11)
    net_b = mk().eval()
    onet = (lambda m: O.MlpNet(_np_sd(m)) if game == "tictactoe" else O.ConvNet(gid, H, W, _np_sd(m)))
    kind = "mlp" if game == "tictactoe" else "conv"
    hnet = net_a.to_hip(max_batch=8)
    eng = E.SelfPlayEngine(gid, H, W, n_slots=8, n_sim=20, net=hnet, seed=2)
    for wave in range(2):
        got = sort_samples(eng.run(8, first_game_id=8 * wave))
        ref = O.selfplay(gid, H, W, 8, 20, (kind, onet(net_a)), seed=2, first_game_id=8 * wave)
        assert np.array_equal(got["pi"], ref["pi"]) and np.array_equal(got["meta"], ref["meta"])
    assert eng.stats()["graph_replays"] > 0
    hnet.load_state_dict(net_b.state_dict())             # host tensors: host fold
    got = sort_samples(eng.run(8, first_game_id=16))
    ref = O.selfplay(gid, H, W, 8, 20, (kind, onet(net_b)), seed=2, first_game_id=16)
    stale = O.selfplay(gid, H, W, 8, 20, (kind, onet(net_a)), seed=2, first_game_id=16)
    assert not np.array_equal(ref["visits"], stale["visits"]), "the two weight sets play the same games: test proves nothing"
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k
    hnet.load_state_dict(net_a.cuda().state_dict())      # device tensors: device fold, no host copy
    got = sort_samples(eng.run(8, first_game_id=16))
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], stale[k]), k


@pytest.mark.parametrize("tag", ["othello8", "othello6", "connect4", "tictactoe"])
def test_device_fold_equals_host_fold(tag):
    """az_net_set_tensor_device + az_net_commit_device (BatchNorm fold and MFMA re-tiling by device kernels) against
    az_net_set_tensor + az_net_commit (host, float64): identical network outputs, non-trivial BN statistics"""
    from test_gpu_net import nets
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, hnet_host = nets(tag)
    hnet_dev = E.HipNet(gid, H, W, {k: torch.as_tensor(v).cuda() for k, v in sd.items()}, max_batch=4096)
    grids, players, _ = O.random_positions(gid, H, W, 3, 40, 1500)
    canon = torch.as_tensor((grids * players[:, None]).astype(np.float32), device="cuda")
    for B in (canon.shape[0], 7):
        (p0, v0), (p1, v1) = hnet_host.forward(canon[:B].contiguous()), hnet_dev.forward(canon[:B].contiguous())
        assert torch.equal(p0, p1) and torch.equal(v0, v1)
    op, ov = onet.forward(canon.cpu().numpy()[:64])
    assert np.array_equal(p1.cpu().numpy()[:7], op[:7])
    with pytest.raises(E._lib.AzError, match="missing device tensor"):
        bad = E.HipNet(gid, H, W, sd, max_batch=8)
        E.check(E.lib().az_net_set_tensor_device(bad.h, b"fc_value.bias", canon.data_ptr(), 1, None))
        E.check(E.lib().az_net_commit_device(bad.h, None))


def test_plane_shapes_without_a_trunk_kernel_are_refused():
    """the reference builds OthelloNet for any even n and Connect4Net for any width x height >= 4x4 (othello.py:316-339,
    connect4.py:343-368); the HIP trunk covers every plane between 5x5 and 8x8 (below 5x5 conv4 has no output and the reference's own
    forward fails): everything else is AZ_EINVAL at az_net_create, never a silent mis-tiled launch"""
    for gid, H, W in ((1, 4, 4), (1, 4, 7), (0, 4, 4)):
        h = C.c_void_p()
        rc = E.lib().az_net_create(gid, H, W, 64, C.byref(h))
        assert rc == E._lib.AZ_EINVAL and b"no conv-trunk kernel" in E.lib().az_last_error()
    h = C.c_void_p()
    assert E.lib().az_net_create(0, 5, 5, 64, C.byref(h)) == E._lib.AZ_EINVAL  # odd Othello size: the reference's ValueError
    assert E.lib().az_net_create(1, 6, 7, 64, C.byref(h)) == 0
    E.lib().az_net_destroy(h)


@pytest.mark.parametrize("H,W", [(5, 5), (5, 6), (6, 5), (6, 8), (8, 7), (7, 7), (7, 8), (5, 8)])
def test_connect4_net_and_selfplay_on_other_board_sizes(H, W):
    """Connect4Net on the other board sizes the engine plays (height H x width W; the plane is W x H, connect4.py:399): the network is
    bit-equal to the oracle on random positions at three batch sizes (one-board-per-wave trunk at any size, the generic dense
    kernels instead of the fused 6x7 tail), and a production-mode self-play run equals the oracle's sample for sample"""
    from alphazero_amd.games.connect4 import Connect4Net
    torch.manual_seed(10 * H + W)
    net = Connect4Net(W, H).eval()
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    onet = O.ConvNet(O.CONNECT4, H, W, sd)
    hnet = E.HipNet(1, H, W, sd, max_batch=5000)
    grids, players, _ = O.random_positions(O.CONNECT4, H, W, 21 + H + W, 400, 5000)
    canon = (grids * players[:, None]).astype(np.float32)
    for B in (1, 77, min(len(players), 4500)):
        p, v = hnet.forward(torch.as_tensor(canon[:B], device="cuda"))
        op, ov = onet.forward(canon[:B])
        assert np.array_equal(p.cpu().numpy(), op) and np.array_equal(v.cpu().numpy(), ov), (H, W, B)
    eng = E.SelfPlayEngine(1, H, W, n_slots=12, n_sim=24, net=hnet, seed=4, sample_capacity=20 * (H * W + 1))
    got = eng.run(20)
    meta = got["meta"].cpu().numpy()
    order = np.lexsort((meta[:, 1], meta[:, 0]))
    ref = O.selfplay(O.CONNECT4, H, W, 20, 24, ("conv", onet), seed=4)
    for k in ("state", "z", "visits", "pi"):
        assert np.array_equal(got[k].cpu().numpy()[order], ref[k]), (H, W, k)
    eng.close()
    hnet.close()


@pytest.mark.parametrize("mode,tags", [("1", "othello8 connect4"), ("0", "othello8 connect4")])
def test_both_forms_of_conv2_are_bit_equal_to_their_oracle_forms(mode, tags):
    """conv2 runs in the Winograd F(2x2,3x3) form on 8x8 and 7x6 planes by default (the rest of the suite); AZ_WINOGRAD (read once
    per process by the library AND by the oracle) switches: "1" is the default spelled out, "0" runs the direct form everywhere.  In a child process each: known answers within 1e-5 of the reference, the network bit-equal
    to the oracle below 4096 boards (one board per wave) and above (two boards per wave), one self-play run sample for sample."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from conftest import TAGS
from oracle import oracle as O
from alphazero_amd import engine as E
from test_gpu_net import nets
from test_gpu_engine import sort_samples
for tag in %r.split():
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, hnet = nets(tag)
    assert onet.winograd() == (%r == '1')
    canon = fx['grids'].astype(np.float32) * fx['players'].astype(np.float32)[:, None, None]
    p, v = hnet.forward(torch.as_tensor(canon, device='cuda'))
    assert np.abs(p.cpu().numpy() - fx['probs']).max() < 1e-5 and np.abs(v.cpu().numpy() - fx['v']).max() < 1e-5
    grids, players, _ = O.random_positions(gid, H, W, 77, 40, 1500)
    x = (grids * players[:, None]).astype(np.float32)
    op, ov = onet.forward(x)
    p, v = hnet.forward(torch.as_tensor(x, device='cuda'))           # < 4096 rows: one board per wave
    assert np.array_equal(p.cpu().numpy(), op) and np.array_equal(v.cpu().numpy(), ov)
    big = E.HipNet(gid, H, W, sd, max_batch=4100)
    idx = (np.arange(4100) * 7 + 3) %% len(x)
    p, v = big.forward(torch.as_tensor(x[idx], device='cuda'))      # two boards per wave
    assert np.array_equal(p.cpu().numpy(), op[idx]) and np.array_equal(v.cpu().numpy(), ov[idx])
    eng = E.SelfPlayEngine(gid, H, W, n_slots=8, n_sim=20, net=hnet, seed=4)
    got = sort_samples(eng.run(8))
    ref = O.selfplay(gid, H, W, 8, 20, ('conv', onet), seed=4)
    assert all(np.array_equal(got[k], ref[k]) for k in ('state', 'z', 'meta', 'visits', 'pi'))
print('conv2 forms ok')
""" % (ROOT, ROOT, tags, mode)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AZ_WINOGRAD=mode), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "conv2 forms ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
