#!/usr/bin/env python3
"""Experiment: time of one OthelloNet SGD step (batch 512) on the GPU: MIOpen convolutions vs im2col + GEMM, eager vs graph."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from alphazero_amd.games.othello import OthelloNet
from alphazero_amd.games import _convnet

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def conv_unfold(conv, x):
    B, C, H, W = x.shape
    p = conv.padding[0]
    cols = F.unfold(x, 3, padding=p)  # [B, C*9, L]
    out = torch.matmul(conv.weight.view(conv.out_channels, -1), cols) + conv.bias.view(1, -1, 1)
    return out.view(B, conv.out_channels, H + 2 * p - 2, W + 2 * p - 2)


def forward_unfold(self, input):
    x = input.view(-1, 1, *self.plane)
    for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3), (self.conv4, self.bn4)):
        x = F.relu(bn(conv_unfold(conv, x)))
    x = x.reshape(-1, self.fc1_input_size)
    x = F.dropout(F.relu(self.fc_bn1(self.fc1(x))), p=self.dropout, training=self.training)
    x = F.dropout(F.relu(self.fc_bn2(self.fc2(x))), p=self.dropout, training=self.training)
    return F.log_softmax(self.fc_probs(x), dim=1), torch.tanh(self.fc_value(x))


if os.environ.get("AZ_CUDNN_BENCHMARK") == "1":
    torch.backends.cudnn.benchmark = True
for name, fwd in (("MIOpen conv", None),):
    torch.manual_seed(0)
    net = OthelloNet(n=8, device="cuda").train()
    if fwd is not None:
        net.forward = fwd.__get__(net)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    x = torch.randint(-1, 2, (bs, 8, 8), device="cuda").float()
    pi = torch.softmax(torch.randn(bs, 65, device="cuda"), 1)
    z = torch.randint(-1, 2, (bs, 1), device="cuda").float()

    def step():
        opt.zero_grad(set_to_none=True)
        lp, v = net(x)
        loss = -torch.sum(pi * lp) / bs + torch.sum((v - z) ** 2) / bs
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 200
    print(f"{name:16s} batch {bs}: eager {eager * 1e3:.2f} ms/step, graph replay {graphed * 1e3:.2f} ms/step")
