"""Shared 4-conv / 2-dense policy-value trunk of OthelloNet and Connect4Net
(othello.py:306-382, connect4.py:333-412).  Layer names and creation order follow the reference so that
state_dict keys match and torch's default initialisation under a given seed is the same."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..base import PolicyValueNetwork


class ConvPolicyValueNet(PolicyValueNetwork):
    n_channels = 32
    dropout = 0.3

    def _build(self, plane_rows, plane_cols, fc1_width, fc2_width, action_size):
        dev, ch = self.device, self.n_channels
        self.plane = (plane_rows, plane_cols)
        self.action_size = action_size
        self.conv1 = nn.Conv2d(1, ch, 3, stride=1, padding=1, device=dev)
        self.conv2 = nn.Conv2d(ch, ch, 3, stride=1, padding=1, device=dev)
        self.conv3 = nn.Conv2d(ch, ch, 3, stride=1, device=dev)
        self.conv4 = nn.Conv2d(ch, ch, 3, stride=1, device=dev)
        self.bn1, self.bn2, self.bn3, self.bn4 = (nn.BatchNorm2d(ch, device=dev) for _ in range(4))
        self.fc1_input_size = ch * (plane_rows - 4) * (plane_cols - 4)
        self.fc1 = nn.Linear(self.fc1_input_size, fc1_width, device=dev)
        self.fc_bn1 = nn.BatchNorm1d(fc1_width, device=dev)
        self.fc2 = nn.Linear(fc1_width, fc2_width, device=dev)
        self.fc_bn2 = nn.BatchNorm1d(fc2_width, device=dev)
        self.fc_probs = nn.Linear(fc2_width, action_size, device=dev)
        self.fc_value = nn.Linear(fc2_width, 1, device=dev)

    def forward(self, input):
        x = input.view(-1, 1, *self.plane)  # connect4: a (6,7) grid is re-read as a (7,6) plane (connect4.py:399)
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3), (self.conv4, self.bn4)):
            x = F.relu(bn(conv(x)))
        x = x.view(-1, self.fc1_input_size)
        x = F.dropout(F.relu(self.fc_bn1(self.fc1(x))), p=self.dropout, training=self.training)
        x = F.dropout(F.relu(self.fc_bn2(self.fc2(x))), p=self.dropout, training=self.training)
        return F.log_softmax(self.fc_probs(x), dim=1), torch.tanh(self.fc_value(x))


def uniform_or_normalised(picked, n_legal):
    """shared tail of get_normalized_probs: renormalise, or fall back to uniform below 1e-6"""
    total = 0
    for p in picked.values():
        total += p
    if total < 1e-6:
        print(f"The sum of the probabilities of the {n_legal} legal moves is {total}")
        return {m: 1 / n_legal for m in picked}
    return {m: p / total for m, p in picked.items()}
