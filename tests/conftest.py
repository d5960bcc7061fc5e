import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

TAGS = {
    # tag: (game name, oracle game id, H, W, action size, n)
    "othello8": ("othello", 0, 8, 8, 65, 8),
    "othello6": ("othello", 0, 6, 6, 37, 6),
    "othello4": ("othello", 0, 4, 4, 17, 4),
    "connect4": ("connect4", 1, 6, 7, 7, None),
    "tictactoe": ("tictactoe", 2, 3, 3, 9, None),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip(f"fixture {name} missing")
    return np.load(path, allow_pickle=False)


def unpack_mask(packed, A):
    return np.unpackbits(packed, axis=-1)[..., :A].astype(bool)


def free_port():
    """a TCP port free on 127.0.0.1 right now (rendezvous of a multi-process test: never a fixed number)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
