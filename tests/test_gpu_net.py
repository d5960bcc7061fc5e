"""GPU parity, policy-value network (K5/K6): HIP forward through the C ABI against
 (a) the golden known answers of the reference's torch forward, tolerance 1e-5, and
 (b) the CPU oracle, which restates the same accumulation order -> demanded bit-exact."""
import ast

import numpy as np
import pytest
import torch

from conftest import TAGS, golden
from oracle import oracle as O
from tools import closed_form as cf
from alphazero_amd import engine as E

pytestmark = pytest.mark.gpu
TOL = 1e-5
NET_TAGS = ["othello8", "othello6", "connect4", "tictactoe"]


def nets(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"net_{tag}.npz")
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["shape_keys"], fx["shape_vals"])}
    sd = {k: v for k, v in cf.closed_form_state_dict(shapes).items() if not k.endswith("num_batches_tracked")}
    onet = O.MlpNet(sd) if game == "tictactoe" else O.ConvNet(gid, H, W, sd)
    return fx, sd, onet, E.HipNet(gid, H, W, sd, max_batch=4096)


@pytest.mark.parametrize("tag", NET_TAGS)
def test_known_answers(tag):
    fx, sd, onet, hnet = nets(tag)
    canon = fx["grids"].astype(np.float32) * fx["players"].astype(np.float32)[:, None, None]
    probs, v = hnet.forward(torch.as_tensor(canon, device="cuda"))
    probs, v = probs.cpu().numpy(), v.cpu().numpy()
    assert np.abs(probs - fx["probs"]).max() < TOL
    assert np.abs(v - fx["v"]).max() < TOL


@pytest.mark.parametrize("tag", NET_TAGS)
def test_bit_exact_vs_oracle(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, hnet = nets(tag)
    grids, players, _ = O.random_positions(gid, H, W, 77, 40, 1500)
    canon = (grids * players[:, None]).astype(np.float32)
    if len(players) % 64 == 0:  # a ragged row count: the last workgroup of every kernel is partly filled
        canon, players = canon[:-1], players[:-1]
    B = len(players)
    assert B > 300 and B % 64 != 0
    probs, v = hnet.forward(torch.as_tensor(canon, device="cuda"))
    oprobs, ov = onet.forward(canon)
    assert np.array_equal(probs.cpu().numpy(), oprobs), np.abs(probs.cpu().numpy() - oprobs).max()
    assert np.array_equal(v.cpu().numpy(), ov)


def test_full_batch_and_ragged():
    """BASELINE config 2 batch (4096 boards) + ragged tail sizes: each row depends only on its board"""
    game, gid, H, W, A, n = TAGS["othello8"]
    fx, sd, onet, hnet = nets("othello8")
    grids, players, _ = O.random_positions(gid, H, W, 5, 80, 4096)
    canon = torch.as_tensor((grids * players[:, None]).astype(np.float32), device="cuda")
    assert canon.shape[0] == 4096
    pf, vf = hnet.forward(canon)
    for B in (1, 3, 63, 65, 130):
        p, v = hnet.forward(canon[:B].contiguous())
        assert torch.equal(p, pf[:B]) and torch.equal(v, vf[:B])
    assert float((pf.sum(1) - 1).abs().max()) < 1e-5
    idx = np.random.RandomState(0).choice(4096, 64, replace=False)
    op, ov = onet.forward(canon.cpu().numpy()[idx])
    assert np.array_equal(pf.cpu().numpy()[idx], op) and np.array_equal(vf.cpu().numpy()[idx], ov)
    with pytest.raises(ValueError):
        hnet.forward(torch.zeros((5000, 64), device="cuda"))


@pytest.mark.parametrize("tag", ["othello8", "othello6", "connect4"])
def test_large_batch_kernel_equals_small_batch_kernel(tag):
    """from 4096 boards up the trunk runs two boards per wave on 32x32x2 MFMA (k_trunk2), below that one board per
    wave on 16x16x4: same accumulation order, so the same boards must give the same bits in either -- odd and ragged
    batch sizes included (the small-batch results are the ones checked against the oracle above).  The dense layers
    switch tile shapes with the batch as well (128x128 from 8192 / 16384 rows; the one-wave-per-SIMD k_gemm_solo with 256x256
    workgroup tiles from one tile per CU up: 16384 rows for fc1, 32768 for fc2; its 128x128 sibling k_gemm_solo_t from two
    tiles per CU up: fc1 at 9000 rows, fc2 at 16500)."""
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, _ = nets(tag)
    sizes = (4096, 4099, 6001, 9000) + ((16500, 33000) if tag == "othello8" else ())  # 16500: fc1 on k_gemm_solo, fc2 on the 128x128 tile; 33000: both solo
    hnet = E.HipNet(gid, H, W, sd, max_batch=max(sizes))
    grids, players, _ = O.random_positions(gid, H, W, 11, 40, 1200)
    canon = torch.as_tensor((grids * players[:, None]).astype(np.float32), device="cuda")
    n0 = canon.shape[0]
    p_ref, v_ref = hnet.forward(canon)  # < 4096 rows: one board per wave
    op, ov = onet.forward(canon.cpu().numpy()[:200])
    assert np.array_equal(p_ref.cpu().numpy()[:200], op) and np.array_equal(v_ref.cpu().numpy()[:200], ov)
    for B in sizes:
        idx = torch.arange(B, device="cuda") % n0
        idx = (idx * 7 + 3) % n0  # not the same neighbour pairs in every pass
        p, v = hnet.forward(canon[idx].contiguous())
        assert torch.equal(p, p_ref[idx]) and torch.equal(v, v_ref[idx]), (tag, B)


def test_dynamic_row_counts_on_a_launch_sized_for_the_full_batch():
    """the hot path's form of the call (az_net_forward_dyn): the launch is sized for the engine's capacity, the rows really present
    are a device counter.  Towards the end of a self-play wave the counter falls to a few hundred rows on a 32768-row launch: the
    dense kernel then lowers its tile height (k_gemm_solo: 64 / 128 / 256 rows per workgroup), the persistent trunk and the heads
    stop at the counter.  Every count must give the bits of the plain forward on the same rows and leave the rows behind it alone."""
    game, gid, H, W, A, n = TAGS["othello8"]
    fx, sd, onet, _ = nets("othello8")
    cap = 33000
    hnet = E.HipNet(gid, H, W, sd, max_batch=cap)
    grids, players, _ = O.random_positions(gid, H, W, 12, 40, 1200)
    canon = torch.as_tensor((grids * players[:, None]).astype(np.float32), device="cuda")
    n0 = canon.shape[0]
    p_ref, v_ref = hnet.forward(canon)  # small-batch kernels: the results checked against the oracle elsewhere in this file
    idx = (torch.arange(cap, device="cuda") * 5 + 1) % n0
    x = canon[idx].contiguous()
    # 8256 / 8257 and 16512 / 16513: the last counts that still fit 64- / 128-row tiles into the 129 tile rows of this launch
    for count in (0, 1, 63, 64, 65, 500, 4097, 8192, 8256, 8257, 12000, 16384, 16512, 16513, 20000, 32768, 33000, 40000):
        c = torch.tensor([count], dtype=torch.int32, device="cuda")
        probs = torch.full((cap, A), -7.0, device="cuda")
        v = torch.full((cap,), -7.0, device="cuda")
        hnet.forward_dyn(x, c, probs, v)
        m = min(count, cap)
        assert torch.equal(probs[:m], p_ref[idx[:m]]) and torch.equal(v[:m], v_ref[idx[:m]]), count
        assert bool((probs[m:] == -7.0).all()) and bool((v[m:] == -7.0).all()), count


def test_live_stage_profile():
    """az_net_profile: every stage launch carries its own start / stop events and is booked under the kernel FAMILY that served it
    (a slot's mean is then what rocprofv3 lists for that kernel): 5000 rows run k_trunk2 / the tiled GEMMs, 100 rows k_trunk_q /
    k_dense_frag, 600 rows k_trunk (one board per wave) / k_dense_frag"""
    game, gid, H, W, A, n = TAGS["othello8"]
    fx, sd, onet, _ = nets("othello8")
    hnet = E.HipNet(gid, H, W, sd, max_batch=5000)
    x = torch.zeros((5000, 64), device="cuda")
    p0, v0 = hnet.forward(x)
    hnet.profile(True)
    for _ in range(3):
        p1, v1 = hnet.forward(x)
    for _ in range(2):
        hnet.forward(x[:100].contiguous())
    hnet.forward(x[:600].contiguous())
    prof = hnet.profile_read()
    assert set(prof) == set(E.PROFILE_SLOTS)
    counts = {k: prof[k][1] for k in prof}
    assert counts == {"k_trunk2": 3, "k_gemm fc1": 3, "k_gemm fc2": 3, "k_heads": 6, "k_trunk": 1, "small fc1": 3, "small fc2": 3, "k_trunk_q": 2}, counts
    assert all(prof[k][0] > 0 for k in prof) and prof["k_trunk2"][0] / 3 > prof["k_trunk"][0] > prof["k_trunk_q"][0] / 2
    assert prof["k_gemm fc1"][0] / 3 > prof["small fc1"][0] / 3
    assert hnet.profile_overhead_ms() == 0.0
    hnet.profile(False)
    hnet.forward(x)
    assert hnet.profile_read()["k_heads"][1] == 6  # nothing recorded while switched off
    assert torch.equal(p0, p1) and torch.equal(v0, v1)


@pytest.mark.parametrize("tag", ["othello8", "othello6"])
def test_fragment_order_dense_kernel_across_its_row_limits(tag):
    """up to 1024 (fc1, K = 512) / 2048 rows (fc2, K = 1024) the dense layers run k_dense_frag (16 rows x 16 columns per wave on 16x16x4,
    weights streamed in B-fragment order), above that the tiled GEMMs: the same boards must give the same bits on either side of
    both limits, for row counts that do not fill the last 16-row block, and for device-side row counts below the launch's size"""
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, _ = nets(tag)
    hnet = E.HipNet(gid, H, W, sd, max_batch=2304)
    assert hnet.stage_kernel(1, 1) == "k_dense_frag" and hnet.stage_kernel(2, 2048) == "k_dense_frag" and hnet.stage_kernel(2, 2049) == "k_gemm"
    grids, players, _ = O.random_positions(gid, H, W, 21, 40, 900)
    canon = torch.as_tensor((grids * players[:, None]).astype(np.float32), device="cuda")
    n0 = canon.shape[0]
    idx = (torch.arange(2304, device="cuda") * 11 + 5) % n0
    x = canon[idx].contiguous()
    p_ref, v_ref = hnet.forward(x)  # 2304 rows: both layers on the tiled GEMM
    op, ov = onet.forward(x[:96].cpu().numpy())
    assert np.array_equal(p_ref[:96].cpu().numpy(), op) and np.array_equal(v_ref[:96].cpu().numpy(), ov)
    for B in (1, 2, 15, 16, 17, 31, 100, 129, 257, 1023, 1024, 1025, 1040, 2047, 2048, 2049):
        p, v = hnet.forward(x[:B].contiguous())
        assert torch.equal(p, p_ref[:B]) and torch.equal(v, v_ref[:B]), (tag, B)
    small = E.HipNet(gid, H, W, sd, max_batch=1000)  # a launch sized for 1000 rows, the rows present a device counter
    for count in (0, 1, 16, 17, 500, 999, 1000, 5000):
        c = torch.tensor([count], dtype=torch.int32, device="cuda")
        probs = torch.full((1000, A), -7.0, device="cuda")
        v = torch.full((1000,), -7.0, device="cuda")
        small.forward_dyn(x[:1000].contiguous(), c, probs, v)
        m = min(count, 1000)
        assert torch.equal(probs[:m], p_ref[:m]) and torch.equal(v[:m], v_ref[:m]), count
        assert bool((probs[m:] == -7.0).all()) and bool((v[m:] == -7.0).all()), count


@pytest.mark.parametrize("tag", ["othello8", "othello6", "connect4"])
def test_four_waves_per_board_trunk_equals_one_wave_per_board(tag):
    """up to 512 boards the trunk deals a board's 16x16 output tiles over the four waves of a workgroup (k_trunk_q), above that one wave
    walks the whole board (k_trunk): same chains per output element, so the same boards give the same bits on both sides of the limit"""
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, _ = nets(tag)
    hnet = E.HipNet(gid, H, W, sd, max_batch=1200)
    assert hnet.stage_kernel(0, 512) == "k_trunk_q" and hnet.stage_kernel(0, 513) == "k_trunk"
    grids, players, _ = O.random_positions(gid, H, W, 31, 40, 1200)
    canon = torch.as_tensor((grids * players[:, None]).astype(np.float32), device="cuda")
    assert canon.shape[0] > 600
    p_ref, v_ref = hnet.forward(canon)  # one wave per board
    for B in (1, 2, 5, 64, 257, 300, 512, 513):
        p, v = hnet.forward(canon[:B].contiguous())
        assert torch.equal(p, p_ref[:B]) and torch.equal(v, v_ref[:B]), (tag, B)
    op, ov = onet.forward(canon[:64].cpu().numpy())
    p, v = hnet.forward(canon[:64].contiguous())
    assert np.array_equal(p.cpu().numpy(), op) and np.array_equal(v.cpu().numpy(), ov)


def test_randomised_batch_sizes_across_the_kernel_variants():
    """tools/fuzz_net.py: 200 random launches per run -- batch sizes at and around every row count where the dispatch switches kernels
    (128 ... 32768, +-1), random board order, plain and device-counted launches -- must reproduce the bits a board gets in a small
    batch, which are the oracle's"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_net
    assert fuzz_net.run(200, seed=12, verbose=False) == []
