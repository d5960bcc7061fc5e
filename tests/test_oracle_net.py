"""G2: the oracle's network forward (BN folded, fmaf chains in the HIP kernels' accumulation order)
against the reference's torch forward under closed-form weights.  Tolerance 1e-5 abs."""
import ast
import os

import numpy as np
import pytest

from conftest import TAGS, golden
from oracle import oracle as O
from tools import closed_form as cf

TOL = 1e-5


def oracle_net(tag, fx):
    game, gid, H, W, A, n = TAGS[tag]
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["shape_keys"], fx["shape_vals"])}
    sd = cf.closed_form_state_dict(shapes)
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    return O.MlpNet(sd) if game == "tictactoe" else O.ConvNet(gid, H, W, sd)


@pytest.mark.parametrize("tag", ["othello8", "othello6", "connect4", "tictactoe"])
def test_known_answers(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"net_{tag}.npz")
    net = oracle_net(tag, fx)
    canon = fx["grids"].astype(np.float32) * fx["players"].astype(np.float32)[:, None, None]
    probs, v = net.forward(canon)
    assert np.abs(probs - fx["probs"]).max() < TOL
    assert np.abs(v - fx["v"]).max() < TOL
    assert np.abs(probs.sum(1) - 1).max() < 1e-5
    # evaluate(): value is flipped back to the absolute frame by board.player (base.py:366)
    k = len(fx["eval_v"])
    assert np.abs(probs[:k] - fx["eval_probs"]).max() < TOL
    assert np.abs(v[:k].astype(np.float64) * fx["players"][:k] - fx["eval_v"]).max() < TOL


@pytest.mark.parametrize("tag", ["othello8", "othello6"])
def test_known_answers_in_the_fixed_point_dense_form(tag):
    """the same G2 fixtures (the reference's torch forward under closed-form weights) with fc1 / fc2 as exact block-fixed-point integer
    dot products (AZ_DENSE_I8, oracle/az_oracle.c dense_layer_q): the form is pinned to the reference by the same 1e-5"""
    fx = golden(f"net_{tag}.npz")
    net = oracle_net(tag, fx)
    net.set_qdense(True)
    assert net.qdense()
    canon = fx["grids"].astype(np.float32) * fx["players"].astype(np.float32)[:, None, None]
    probs, v = net.forward(canon)
    assert np.abs(probs - fx["probs"]).max() < TOL and np.abs(v - fx["v"]).max() < TOL
    net.set_qdense(False)
    p0, v0 = net.forward(canon)
    assert np.abs(probs - p0).max() < 2e-6 and np.abs(v - v0).max() < 2e-6  # and within rounding of the fma-chain form


def test_param_counts():
    # report p.7 Table 2 / SURVEY 6.1
    assert int(golden("net_othello8.npz")["n_params"]) == 1115362
    assert int(golden("net_othello6.npz")["n_params"]) == 707782
    assert int(golden("net_connect4.npz")["n_params"]) == 43208
    assert int(golden("net_tictactoe.npz")["n_params"]) == 316


def test_det_math():
    L = O.lib()
    xs = np.linspace(-80, 0, 4001).astype(np.float32)
    got = np.array([L.orc_det_expf(float(x)) for x in xs])
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 4e-7
    xs = np.linspace(-6, 6, 2001).astype(np.float32)
    got = np.array([L.orc_det_tanhf(float(x)) for x in xs])
    assert np.max(np.abs(got - np.tanh(xs.astype(np.float64)))) < 3e-7
    xs = np.exp(np.linspace(-40, 40, 3001))
    got = np.array([L.orc_det_log(float(x)) for x in xs])
    assert np.max(np.abs(got - np.log(xs))) < 1e-14 * 45
    xs = np.linspace(-600, 5, 3001)
    got = np.array([L.orc_det_exp(float(x)) for x in xs])
    assert np.max(np.abs(got - np.exp(xs)) / np.exp(xs)) < 1e-14


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    import ctypes as C
    L = O.lib()
    out = (C.c_uint32 * 4)()
    L.orc_philox4x32(0, 0, 0, 0, 0, 0, out)
    assert list(out) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    L.orc_philox4x32(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, out)
    assert list(out) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    L.orc_philox4x32(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, out)
    assert list(out) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert cf.philox4x32(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344) == tuple(out)


@pytest.mark.parametrize("tag", ["othello8", "connect4"])
def test_winograd_form_of_conv2_meets_the_reference_tolerance(tag):
    """the oracle's restatement of the product's Winograd conv2 (F(2x2,3x3), 2.25x fewer multiplications; the default on 8x8
    and 7x6 planes): within the same 1e-5 of the reference's torch forward (golden G2) as the direct form, and not the
    same bits as the direct form (the two are different arithmetic; product and oracle switch together)"""
    import ast
    from tools import closed_form as cf
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"net_{tag}.npz")
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["shape_keys"], fx["shape_vals"])}
    sd = {k: v for k, v in cf.closed_form_state_dict(shapes).items() if not k.endswith("num_batches_tracked")}
    net = O.ConvNet(gid, H, W, sd)
    assert net.winograd()  # the product's default on 8x8 and 7x6 planes
    canon = fx["grids"].astype(np.float32) * fx["players"].astype(np.float32)[:, None, None]
    net.set_winograd(False)
    p0, v0 = net.forward(canon)
    net.set_winograd(True)
    p1, v1 = net.forward(canon)
    net.set_winograd(False)
    p0, v0 = net.forward(canon)
    assert np.abs(p0 - fx["probs"]).max() < 1e-5 and np.abs(v0 - fx["v"]).max() < 1e-5
    assert np.abs(p1 - fx["probs"]).max() < 1e-5 and np.abs(v1 - fx["v"]).max() < 1e-5
    assert np.abs(p1 - p0).max() < 1e-6 and not (np.array_equal(p1, p0) and np.array_equal(v1, v0))


def test_fixed_point_dense_form_of_the_oracle():
    """AZ_DENSE_I8 (oracle/az_oracle.c dense_layer_q): fc1 / fc2 of OthelloNet as exact integer dot products of block-fixed-point operands.
    Against the float64 forward it must be as close as the float32 fma chain it replaces; the switch is per network and off by default."""
    import torch
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(5)
    net = OthelloNet(n=8).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    sd = {k: v.numpy() for k, v in net.state_dict().items() if v.dtype == torch.float32}
    orc = O.ConvNet(O.OTHELLO, 8, 8, sd)
    assert not orc.qdense() or os.environ.get("AZ_DENSE_I8") == "1"
    rng = np.random.default_rng(2)
    x = rng.integers(-1, 2, size=(48, 64)).astype(np.float32)
    x[5] = 0.0
    orc.set_qdense(False)
    p0, v0 = orc.forward(x)
    orc.set_qdense(True)
    assert orc.qdense()
    p1, v1 = orc.forward(x)
    net64 = OthelloNet(n=8).double().eval()
    net64.load_state_dict({k: (v.double() if v.dtype == torch.float32 else v) for k, v in net.state_dict().items()})
    with torch.no_grad():
        lp, v = net64(torch.tensor(x.reshape(-1, 8, 8), dtype=torch.float64))
    p, v = lp.exp().numpy(), v.numpy().ravel()
    assert not np.array_equal(p0, p1)  # a different arithmetic ...
    assert np.abs(p1 - p).max() < 2e-7 and np.abs(v1 - v).max() < 1e-6  # ... as exact as the chain
    assert np.abs(p1 - p).max() < 3 * np.abs(p0 - p).max() + 1e-8 and np.abs(v1 - v).max() < 3 * np.abs(v0 - v).max() + 1e-7
    c4 = O.ConvNet(O.CONNECT4, 6, 7, {k: v.numpy() for k, v in __import__("alphazero_amd.games.connect4", fromlist=["Connect4Net"]).Connect4Net(7, 6).eval().state_dict().items() if v.dtype == torch.float32})
    c4.set_qdense(True)
    assert not c4.qdense()  # OthelloNet only
