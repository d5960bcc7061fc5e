#!/usr/bin/env python3
"""The hand-written training step (csrc/az_train.hip) against torch autograd in float64 on the CPU: every activation, every
gradient the step keeps in its workspace, the losses, and the parameters / BatchNorm statistics after k steps.
    python tools/check_train_step.py [tag] [batch] [steps] [dropout]      tag: othello8 | othello6 | connect4
Prints one line per buffer (max abs error, scale); tests/test_gpu_train_step.py asserts on the same report."""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_net(tag, seed=0):
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(seed)
    from alphazero_amd.games.tictactoe import TicTacToeNet
    net = {"othello8": lambda: OthelloNet(n=8), "othello6": lambda: OthelloNet(n=6), "connect4": lambda: Connect4Net(7, 6), "connect4_8x5": lambda: Connect4Net(8, 5),
           "connect4_5x8": lambda: Connect4Net(5, 8), "tictactoe": lambda: TicTacToeNet()}[tag]()
    with torch.no_grad():  # BatchNorm affine / running statistics away from their defaults, so that a mix-up shows
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3); m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
    return net


def make_samples(net, S, seed=1):
    g = torch.Generator().manual_seed(seed)
    gid, H, W = net.hip_shape()
    state = torch.randint(-1, 2, (S, H, W), generator=g, dtype=torch.int8)
    pi = torch.rand(S, net.action_size, generator=g) ** 3
    pi = (pi / pi.sum(1, keepdim=True)).float()
    z = torch.randint(-1, 2, (S,), generator=g, dtype=torch.int8)
    return state, pi, z


def torch_step(net, x, pi, z, bs, masks=None, p=0.0):
    """forward + backward of the reference's loss with every intermediate kept (float64 module)"""
    keep = {}
    if not hasattr(net, "conv1"):  # TicTacToeNet: the module's own forward (no workspace to compare: the step is one launch)
        logp, v = net(x)
        loss_pi = -torch.sum(pi * logp) / bs
        loss_v = torch.sum((v - z) ** 2) / bs
        (loss_pi + loss_v).backward()
        return float(loss_pi.detach()), float(loss_v.detach()), keep

    def k(name, t):
        t.retain_grad()
        keep[name] = t
        return t
    h = x.view(-1, 1, *net.plane)
    for i, (conv, bn) in enumerate(((net.conv1, net.bn1), (net.conv2, net.bn2), (net.conv3, net.bn3), (net.conv4, net.bn4))):
        c = k(f"c{i + 1}", conv(h))
        b = k(f"b{i + 1}", bn(c))
        h = F.relu(b)
    h = h.reshape(-1, net.fc1_input_size)
    y1 = k("y1", net.fc1(h))
    a1 = net.fc_bn1(y1)
    keep["a1"] = a1
    h1 = F.relu(a1)
    if masks is not None:
        h1 = h1 * masks[0] / (1.0 - p)
    h1 = k("h1", h1)
    y2 = k("y2", net.fc2(h1))
    a2 = net.fc_bn2(y2)
    keep["a2"] = a2
    h2 = F.relu(a2)
    if masks is not None:
        h2 = h2 * masks[1] / (1.0 - p)
    h2 = k("h2", h2)
    lp = k("lp", net.fc_probs(h2))
    u = k("u", net.fc_value(h2))
    logp, v = F.log_softmax(lp, dim=1), torch.tanh(u)
    loss_pi = -torch.sum(pi * logp) / bs
    loss_v = torch.sum((v - z) ** 2) / bs
    (loss_pi + loss_v).backward()
    return float(loss_pi.detach()), float(loss_v.detach()), keep


def nhwc(t):  # torch [B, C, H, W] -> the step's [B, H*W, C]
    return t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, t.shape[1])


def relu_ties(keep, rel=2e-6):
    """ReLU inputs of the float64 run that are zero to float32 rounding (|x| < rel * max|x| of their layer): such a unit may take the
    other branch in the float32 step, and the one-unit difference then spreads downstream -- a property of the data, not of the
    kernels.  -> [(layer, count, smallest |x| / max|x|)] of the layers that have any"""
    out = []
    for name in ("b1", "b2", "b3", "b4", "a1", "a2"):
        if name in keep:
            x = keep[name].detach().abs()
            m = float(x.max())
            n = int((x < rel * m).sum())
            if n:
                out.append((name, n, float(x.min()) / m))
    return out


def report(tag="othello8", B=64, steps=3, dropout=0.0, verbose=True, seed=0):
    """-> rows (buffer, max abs error, scale); report.ties = relu_ties() of every step (empty: no unit of the run sits on a ReLU kink)"""
    from alphazero_amd.train_step import HipTrainStep
    report.ties = []
    net = make_net(tag, seed)
    S = B * steps + 7
    state, pi, z = make_samples(net, S, seed=1 + 17 * seed)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(5 + seed))[: B * steps].contiguous()
    lr, mom, wd = 0.05, 0.9, 1e-4
    ref = copy.deepcopy(net).double().train()
    opt = torch.optim.SGD(ref.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    hip = HipTrainStep(net, max_batch=B)
    hip.load(net.cuda())
    hip.begin(lr, mom, wd, dropout, seed=3)
    d = {"state": state.cuda(), "pi": pi.cuda(), "z": z.cuda(), "perm": perm.cuda()}
    lpi, lv = torch.zeros(steps, device="cuda"), torch.zeros(steps, device="cuda")
    rows = []

    def cmp(name, got, want):
        shape = tuple(want.shape)
        got, want = got.detach().double().cpu().reshape(-1), want.detach().double().cpu().reshape(-1)
        diff = (got - want).abs()
        err = float(diff.max())
        scale = float(want.abs().max()) + 1e-30
        rows.append((name, err, scale))
        if verbose:
            where = ""
            if err > 1e-4 * scale + 1e-7:  # where the error sits: index of the maximum, how many elements are off, which rows / last-dim columns
                bad = (diff > 1e-4 * scale + 1e-7).reshape(shape)
                idx = np.unravel_index(int(diff.argmax()), shape)
                r = bad.reshape(shape[0], -1).any(1).nonzero().flatten().tolist()
                c = bad.reshape(-1, shape[-1]).any(0).nonzero().flatten().tolist()
                where = f"  at {idx}, {int(bad.sum())}/{bad.numel()} off; rows {r[:6]}..{r[-3:]} ({len(r)}); cols {c[:6]}..{c[-3:]} ({len(c)})"
            print(f"{name:28s} max|err| {err:10.3e}   max|ref| {scale:10.3e}   rel {err / scale:9.2e}{where}")

    gid, H, W = net.hip_shape()
    A = net.action_size
    NHP = (A + 1 + 15) // 16 * 16
    for s in range(steps):
        rws = perm[s * B:(s + 1) * B]
        # one step at a time so that the workspace of step s can be inspected: a separate perm / loss slice per call
        pcall = d["perm"][s * B:(s + 1) * B].contiguous()
        hip.steps(d["state"], d["pi"], d["z"], pcall, 1, B, lpi[s:s + 1], lv[s:s + 1])
        torch.cuda.synchronize()
        x = state[rws].double()
        masks = None
        if dropout > 0:
            F1, F2 = net.fc1.out_features, net.fc2.out_features
            masks = ((hip.debug("h1", (hip.max_batch, F1))[:B] != 0).double().cpu(), (hip.debug("h2", (hip.max_batch, F2))[:B] != 0).double().cpu())
        opt.zero_grad()
        t_pi, t_v, keep = torch_step(ref, x, pi[rws].double(), z[rws].double().unsqueeze(1), B, masks, dropout)
        report.ties += [(s,) + t for t in relu_ties(keep)]
        if s in (0, steps - 1) and hasattr(net, "conv1"):
            pre = f"step{s}."
            for i in range(4):
                c = keep[f"c{i + 1}"]
                cmp(pre + f"c{i + 1}", hip.debug(f"c{i + 1}")[: c.numel()].view(B, -1, 32), nhwc(c))
                cmp(pre + f"dy{i + 1}", hip.debug(f"dy{i + 1}")[: c.numel()].view(B, -1, 32), nhwc(keep[f"b{i + 1}"].grad))
            F1, F2 = keep["y1"].shape[1], keep["y2"].shape[1]
            cmp(pre + "y1", hip.debug("y1")[: B * F1].view(B, F1), keep["y1"]); cmp(pre + "h1", hip.debug("h1")[: B * F1].view(B, F1), keep["h1"])
            cmp(pre + "y2", hip.debug("y2")[: B * F2].view(B, F2), keep["y2"]); cmp(pre + "h2", hip.debug("h2")[: B * F2].view(B, F2), keep["h2"])
            dl = hip.debug("dlog")[: B * NHP].view(B, NHP)
            cmp(pre + "dlog.policy", dl[:, :A], keep["lp"].grad); cmp(pre + "dlog.value", dl[:, A:A + 1], keep["u"].grad)
            if NHP > A + 1:
                cmp(pre + "dlog.padding", dl[:, A + 1:], torch.zeros(B, NHP - A - 1))
            cmp(pre + "dz2", hip.debug("dz2")[: B * F2].view(B, F2), keep["y2"].grad); cmp(pre + "dz1", hip.debug("dz1")[: B * F1].view(B, F1), keep["y1"].grad)
        cmp(f"step{s}.loss_pi", lpi[s:s + 1], torch.tensor([t_pi])); cmp(f"step{s}.loss_v", lv[s:s + 1], torch.tensor([t_v]))
        opt.step()
    out = copy.deepcopy(net)
    hip.store(out)
    rsd = ref.state_dict()
    for kname, v in out.state_dict().items():
        if v.dtype == torch.float32:
            cmp("final." + kname, v, rsd[kname])
        else:
            rows.append(("final." + kname, float(abs(int(v) - int(rsd[kname]))), 1.0))
    hip.close()
    return rows


if __name__ == "__main__":
    a = sys.argv[1:]
    rows = report(a[0] if a else "othello8", int(a[1]) if len(a) > 1 else 64, int(a[2]) if len(a) > 2 else 3, float(a[3]) if len(a) > 3 else 0.0)
    bad = [(n, e, s) for n, e, s in rows if e > 2e-4 * max(s, 1e-3) + 1e-6]
    print("WORST", sorted(rows, key=lambda r: -r[1] / max(r[2], 1e-3))[:5])
    print("BAD", bad[:20], len(bad))
    print("RELU TIES (step, layer, units, smallest |x| / max|x|)", report.ties)
