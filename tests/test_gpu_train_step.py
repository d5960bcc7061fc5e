"""The hand-written training step (csrc/az_train.hip; trainer.py:320-381 of the reference) on the GPU:
  * against torch autograd in float64 (tools/check_train_step.py): every activation and gradient the step keeps, the losses, and all
    parameters / BatchNorm statistics after k steps -- OthelloNet 8x8 / 6x6, Connect4Net, batch sizes from 32 to 512, dropout off and on
    (the Philox mask of the step is read back and applied in the torch model);
  * against golden G6 (the REFERENCE's optimize_network, tools/gen_golden.py::gen_sgd) through AlphaZeroTrainer with sgd_backend "hip";
  * dropout law, run-to-run determinism, and that the stock PyTorch loop stays selectable.
Tolerance: float32 against float64: 2e-4 of the buffer's largest magnitude (measured: 1e-6).  A ReLU input that is zero to rounding
may take the other branch in float64; the seeds used here have no such tie (tools/check_train_step.py prints where an error sits)."""
import ast
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, TAGS, golden

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,B,steps,dropout", [("tictactoe", 16, 3, 0.0), ("tictactoe", 64, 3, 0.0), ("tictactoe", 250, 2, 0.0), ("othello8", 64, 3, 0.0), ("othello8", 64, 2, 0.3), ("othello6", 32, 2, 0.0), ("connect4", 32, 3, 0.0),
                                                 ("connect4", 128, 2, 0.3), ("othello8", 256, 1, 0.0), ("othello8", 512, 1, 0.0), ("connect4", 512, 1, 0.0),
                                                 ("othello8", 16, 2, 0.0), ("othello6", 48, 2, 0.3), ("connect4_8x5", 32, 2, 0.0), ("connect4_5x8", 64, 2, 0.3),
                                                 # batch sizes that leave the row-tile templates partly filled (48: 3 of 4 tiles, 80: 5 of 8, 144: 9 of 16) and 320
                                                 # boards on 256 workgroups (one or two boards each: unequal counts in the statistics partials)
                                                 ("othello8", 48, 2, 0.3), ("othello6", 80, 2, 0.0), ("connect4", 144, 2, 0.3), ("othello8", 320, 1, 0.3)])
def test_training_step_equals_torch_autograd(tag, B, steps, dropout):
    """ALL three data seeds must be clean.  The one exception a float32-against-float64 comparison has by nature: a ReLU input that
    is zero to float32 rounding in the float64 run may take the other branch in the float32 step, and that one unit's difference
    spreads downstream.  tools/check_train_step.py::relu_ties finds such units in the float64 run; a seed may only be off when the
    float64 run itself has one, and the (case, seed) pairs where that happens are listed by hand in KNOWN_RELU_TIES and counted."""
    import check_train_step as C
    for seed in (0, 1, 2):
        rows = C.report(tag, B, steps, dropout, verbose=False, seed=seed)
        bad = [(n, e, s) for n, e, s in rows if e > 2e-4 * max(s, 1e-3) + 1e-6]
        if (tag, B, steps, dropout, seed) in KNOWN_RELU_TIES:
            assert C.report.ties, ("listed as a ReLU tie, but the float64 run has no unit on a kink", tag, B, steps, dropout, seed)
            continue
        assert not bad, (seed, C.report.ties, bad[:8])
    names = {n for n, _, _ in rows}
    if tag != "tictactoe":
        assert {"step0.c4", "step0.dy1", "step0.dz1", "step0.dlog.policy", "final.conv1.weight", "final.fc_bn2.running_var", "final.bn3.num_batches_tracked"} <= names
    else:
        assert {"step0.loss_pi", "final.fc1.weight", "final.bn2.running_var", "final.bn1.num_batches_tracked", "final.fc_value.bias"} <= names


# (tag, batch, steps, dropout, seed) whose float64 run has a ReLU input on the kink AND whose float32 step resolves it the other way
# (found with tools/list_relu_ties.py on an MI355X; 19 cases x 3 seeds = 57 runs)
KNOWN_RELU_TIES = {("othello8", 64, 3, 0.0, 1), ("othello8", 64, 3, 0.0, 2)}
assert len(KNOWN_RELU_TIES) <= 3


def _fixture_trainer(tag, backend):
    from tools import closed_form as cf
    from alphazero_amd.games.registers import CONFIGS_REGISTER, NETWORKS_REGISTER
    from alphazero_amd.trainer import AlphaZeroTrainer
    game, gid, H, W, A, n = TAGS[tag]
    fx, mem, net_fx = golden(f"sgd_{tag}.npz"), golden(f"selfplay_{tag}.npz"), golden(f"net_{tag}.npz")
    extra = {"board_size": n} if game == "othello" else {}
    cfg = CONFIGS_REGISTER[game](epochs=int(fx["epochs"]), batch_size=int(fx["batch_size"]), device="cuda", **extra)
    tr = AlphaZeroTrainer(verbose=False)
    tr.config, tr.game, tr.sgd_backend = cfg, game, backend
    net = NETWORKS_REGISTER[game](config=cfg)
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(net_fx["shape_keys"], net_fx["shape_vals"])}
    net.load_state_dict({k: torch.tensor(v) for k, v in cf.closed_form_state_dict(shapes).items()})
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    tr.nn = net.to("cuda")
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")  # noqa: E731
    tr.device_memory = {"state": dev(mem["state"], torch.int8), "pi": dev(mem["pi"].astype(np.float32), torch.float32),
                        "z": dev(mem["outcome"], torch.int8), "meta": torch.zeros((len(mem["outcome"]), 4), dtype=torch.int32, device="cuda")}
    rs = np.random.RandomState(int(fx["shuffle_seed"]))  # np.random.seed + np.random.shuffle of the reference's generator

    def reference_order(n_samples, device):
        idx = np.arange(n_samples)
        rs.shuffle(idx)
        return torch.as_tensor(idx, device=device)
    tr._permutation = reference_order
    tr.loss_values = {}
    return tr, fx


def _float64_trajectory(tag, fx):
    """the SAME step sequence (initial weights, batches, order, learning-rate schedule) in float64 on the CPU with torch autograd:
    the trajectory both float32 runs -- the reference's (golden G6) and the hand-written step's -- are roundings of"""
    tr, _ = _fixture_trainer(tag, "torch")
    net = tr.nn.cpu().double().train()
    m = {k: v.cpu() for k, v in tr.device_memory.items()}
    bs, epochs = int(fx["batch_size"]), int(fx["epochs"])
    opt = torch.optim.SGD(net.parameters(), lr=tr.config.learning_rate, momentum=0.9, weight_decay=0.0001)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
    out = {}
    for e in range(epochs):
        nb = m["z"].shape[0] // bs
        perm = tr._permutation(m["z"].shape[0], "cpu")[: nb * bs].view(nb, bs)
        out[e] = {"pi": [], "v": []}
        for rows in perm:
            x, pi, z = m["state"][rows].double(), m["pi"][rows].double(), m["z"][rows].double().unsqueeze(1)
            opt.zero_grad()
            log_p, v = net(x)
            loss_pi, loss_v = -torch.sum(pi * log_p) / bs, torch.sum((v - z) ** 2) / bs
            (loss_pi + loss_v).backward()
            opt.step()
            out[e]["pi"].append(float(loss_pi)); out[e]["v"].append(float(loss_v))
        sched.step()
    return out


@pytest.mark.parametrize("tag", ["tictactoe", "connect4", "othello6", "othello8"])
def test_hand_written_step_matches_the_reference_fixture(tag):
    """golden G6: the per-batch losses the REFERENCE's optimize_network logged (same initial weights, same batches in the same order,
    dropout 0).  First six steps of epoch 0 within 5e-5 of the reference (a wrong momentum / learning rate / weight decay shows from
    step 3 on at 1e-2); Connect4Net / TicTacToeNet within 2e-4 over the whole trajectory and the same final weights.

    The OthelloNets are held against the float64 run of the SAME step sequence (tools/g6_trajectories.py prints the three trajectories
    side by side).  What that shows: every float32 run follows the float64 trajectory to a few 1e-7 until one discrete event -- a ReLU
    input that is zero to rounding falls on the other side -- after which momentum SGD at lr 0.1 amplifies the one-unit difference
    3-4x per step (the reference's own float32 run on 6x6: 5e-7 for twelve steps, then 2.7e-5, 2.5e-4, ... 8.6e-3; the hand-written
    step takes the same turn at the same step; the stock MIOpen path happens to stay with float64).  So: (a) a CLEAN PREFIX of at
    least twelve steps within 5e-5 of float64 (it pins the formula, the constants and the schedule over many steps); (b) after the
    first departure at most four times as far from float64 as the reference's own float32 run has been so far, or as an envelope
    growing 4x per step from 5e-5 (the hand-written step may take such a turn where the reference does not: 8x8, step 13); (c) beyond
    epoch 0, where both float32 runs have left float64 for good (the reference's by 1e-2 .. 3e-1 per step on 6x6), EVERY step at most
    four times as far from the float64 trajectory as the reference's own float32 run has been up to that step (round 4 compared epoch
    means within 25 % + 0.03; profiles/r05_g6_trajectories.txt holds the three trajectories)."""
    tr, fx = _fixture_trainer(tag, "hip")
    tr.optimize_network(0)
    assert tr.sgd_backend_used == "hip"
    f64 = _float64_trajectory(tag, fx) if tag.startswith("othello") else None
    for k in ("pi", "v"):
        got0, ref0 = np.array(tr.loss_values[0][0][k]), fx[f"{k}_loss_0"]
        assert got0.shape == ref0.shape and np.abs(got0 - ref0)[:6].max() < 5e-5, (tag, k, np.abs(got0 - ref0)[:6])
        if f64 is None:
            for e in range(int(fx["epochs"])):
                err = np.abs(np.array(tr.loss_values[0][e][k]) - fx[f"{k}_loss_{e}"])
                assert err.max() < 2e-4, (tag, e, k, err.max())
            continue
        err = np.abs(got0 - np.array(f64[0][k]))
        dirty = np.flatnonzero(err >= 5e-5)
        clean = int(dirty[0]) if len(dirty) else len(err)
        assert clean >= 12, (tag, k, clean, err)
        ref_err = np.maximum.accumulate(np.abs(ref0 - np.array(f64[0][k])))  # how far the reference's own float32 run has strayed so far
        for i in range(clean, len(err)):
            assert err[i] <= 4 * max(ref_err[i], 5e-5 * 4.0 ** (i - clean + 1)), (tag, k, i, clean, err, ref_err)
        far = float(ref_err[-1])  # the reference's own largest distance from float64 so far, carried over the epochs
        for e in range(1, int(fx["epochs"])):
            got, ref, exact = np.array(tr.loss_values[0][e][k]), fx[f"{k}_loss_{e}"], np.array(f64[e][k])
            assert got.shape == ref.shape == exact.shape
            for i in range(len(got)):  # every step of every later epoch: at most 4x as far from float64 as the reference's float32 run has been
                far = max(far, abs(float(ref[i] - exact[i])))
                assert abs(got[i] - exact[i]) <= 4 * far, (tag, e, k, i, abs(got[i] - exact[i]), far)
    sd = {k: v.cpu().numpy() for k, v in tr.nn_twin.state_dict().items()}
    if tag in ("connect4", "tictactoe"):
        assert np.abs(sd["fc1.weight"][:64] - fx["fc1_weight"]).max() < 1e-4
        assert np.abs(sd["fc_value.weight"] - fx["fc_value_weight"]).max() < 1e-4
        assert np.abs(sd["fc_bn1.running_mean" if tag == "connect4" else "bn1.running_mean"] - fx["bn_running_mean"]).max() < 1e-4
    else:
        a_, b_ = sd["fc1.weight"][:64].ravel(), fx["fc1_weight"].ravel()
        assert float(np.dot(a_, b_) / (np.linalg.norm(a_) * np.linalg.norm(b_))) > 0.999
    n_steps = sum(len(fx[f"pi_loss_{e}"]) for e in range(int(fx["epochs"])))
    assert int(tr.nn_twin.bn1.num_batches_tracked) == n_steps and int((tr.nn_twin.fc_bn2 if tag != "tictactoe" else tr.nn_twin.bn2).num_batches_tracked) == n_steps


def test_backends_are_selectable_and_unsupported_shapes_use_the_stock_step():
    tr, fx = _fixture_trainer("connect4", "torch")
    tr.optimize_network(0)
    assert tr.sgd_backend_used == "torch"
    tr2, _ = _fixture_trainer("connect4", "hip")
    tr2.config.batch_size = 24  # not a multiple of 16: outside the hand-written step's range -> the stock step, and never silently
    with pytest.warns(RuntimeWarning, match="stock PyTorch step"):
        tr2.optimize_network(0)
    assert tr2.sgd_backend_used == "torch"
    from alphazero_amd import train_step
    from alphazero_amd.games.tictactoe import TicTacToeNet
    assert train_step.supports(TicTacToeNet(), 16) and not train_step.supports(TicTacToeNet(), 300)
    with pytest.raises(ValueError):
        train_step.HipTrainStep(tr.nn, max_batch=520)
    ts = train_step.HipTrainStep(tr.nn, max_batch=32)
    m = tr.device_memory
    # a permutation entry outside the sample arrays (one past the last row, a negative one): the kernels clamp it and raise the flag
    # that check() -- or the next steps() -- turns into the ValueError; the GPU does not fault and the trainer stays usable
    ts.load(tr.nn)
    ts.begin(0.01, 0.9, 1e-4, 0.0, seed=1)
    good = torch.arange(32, dtype=torch.int64, device="cuda")
    lp, lv = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    for wrong in (m["z"].shape[0], -1, 1 << 40):
        bad = good.clone()
        bad[7] = wrong
        ts.steps(m["state"], m["pi"], m["z"], bad, 1, 32, lp, lv)
        with pytest.raises(ValueError, match="permutation entry"):
            ts.check()
        ts.check()  # reported once, then clear
    ts.steps(m["state"], m["pi"], m["z"], bad, 1, 32, lp, lv)
    lp2, lv2 = torch.full((1,), -7.0, device="cuda"), torch.full((1,), -7.0, device="cuda")
    with pytest.raises(ValueError, match="az_trainer_steps rejected, nothing of it ran.*permutation entry"):  # nobody called check(): the next call reports it
        ts.steps(m["state"], m["pi"], m["z"], good, 1, 32, lp2, lv2)
    torch.cuda.synchronize()
    assert lp2.item() == -7.0 and lv2.item() == -7.0  # ... FIRST: the rejected call changed nothing
    ts.steps(m["state"], m["pi"], m["z"], good, 1, 32, lp, lv)
    ts.check()
    assert torch.isfinite(lp).all() and torch.isfinite(lv).all()
    ts.steps(m["state"], m["pi"], m["z"], bad, 1, 32, lp, lv)
    with pytest.raises(ValueError, match="az_trainer_begin rejected"):  # begin() would wipe the flag: it reports it instead
        ts.begin(0.01, 0.9, 1e-4, 0.0, seed=1)
    ts.begin(0.01, 0.9, 1e-4, 0.0, seed=1)
    ts.steps(m["state"], m["pi"], m["z"], good, 1, 32, lp, lv)
    ts.check()
    ts.close()
    # a module that is not the reference's architecture: not for the hand-written step
    from alphazero_amd.games.othello import OthelloNet
    odd = OthelloNet(n=8)
    assert train_step.supports(odd, 64)
    odd.fc2 = torch.nn.Linear(1024, 256)
    assert not train_step.supports(odd, 64)


@pytest.mark.parametrize("B", [128, 272, 512])
def test_dropout_law_and_determinism(B):
    """the step's dropout: the kept fraction of the units that pass the ReLU is 1 - p, kept units are scaled by 1 / (1 - p), masks
    differ from step to step and between the two layers, and a second run with the same seed reproduces the losses and weights bit for
    bit -- at 128 rows (a dense workgroup sees all rows) and on the row-split path (272 = four blocks of 64 and one of 16; 512 = four of
    128), whose statistics, heads gradient and weight-gradient tiles are combined from partials in a fixed order"""
    import check_train_step as C
    from alphazero_amd.train_step import HipTrainStep
    net = C.make_net("othello8", 3).cuda()
    state, pi, z = C.make_samples(net, 3 * B + 16)
    p = 0.3
    perm = torch.randperm(3 * B + 16, generator=torch.Generator().manual_seed(1))[: 3 * B].cuda().contiguous()

    def run():
        hip = HipTrainStep(net, max_batch=B)
        hip.load(net)
        hip.begin(0.1, 0.9, 1e-4, p, seed=77)
        lp, lv = torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda")
        masks = []
        for s in range(3):
            hip.steps(state.cuda(), pi.cuda(), z.cuda(), perm[s * B:(s + 1) * B].contiguous(), 1, B, lp[s:s + 1], lv[s:s + 1])
            h1, y1 = hip.debug("h1", (B, 1024)), hip.debug("y1", (B, 1024))
            masks.append((h1 != 0).cpu())
            if s == 0:
                mu, var = y1.mean(0), y1.var(0, unbiased=False)
                relu = torch.relu((y1 - mu) / torch.sqrt(var + 1e-5) * net.fc_bn1.weight + net.fc_bn1.bias)
                alive = relu > 1e-4
                kept = (h1 != 0) & alive
                frac = kept.sum().item() / alive.sum().item()
                assert abs(frac - (1 - p)) < 0.01, frac
                assert torch.allclose(h1[kept], relu[kept] / (1 - p), rtol=1e-4, atol=1e-5)
        out = C.copy.deepcopy(net)
        hip.store(out)
        hip.close()
        return lp.cpu(), lv.cpu(), masks, {k: v.cpu().clone() for k, v in out.state_dict().items()}
    a, b = run(), run()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[3][k], b[3][k]) for k in a[3])
    assert not torch.equal(a[2][0], a[2][1])  # a fresh mask every step
