"""GPU: the exact block-fixed-point dense layers (AZ_DENSE_I8=1: OthelloNet's fc1 / fc2 on the int8 matrix pipe, az_net.hip k_q_rows /
k_qgemm) against the CPU oracle under the same switch.  Product and oracle read the variable once per process, so every case runs in a
child process with the variable set; the checkers are the tools the default path is checked with."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, timeout=900):
    env = dict(os.environ, AZ_DENSE_I8="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    return p.stdout


def test_network_equals_oracle_bit_for_bit_at_every_tile_plan():
    """tools/check_qdense.py: OthelloNet 8x8 / 6x6, batch sizes 1 ... 8200 (both tile plans, partial tiles, an empty board), 60 repeated
    forwards as a race screen of the hand-synchronised LDS pipeline"""
    out = _run("check_qdense.py")
    assert "mismatches 0" in out and out.count("bit-equal") >= 30 and "k_qgemm" in out


def test_forward_fuzz_under_the_switch():
    """tools/fuzz_net.py: random networks and batch sizes across the kernel variants (the Connect4 / TicTacToe nets keep their float32 chains)"""
    assert "0 mismatches" in _run("fuzz_net.py", "80", "11")


def test_single_game_searches_and_trainer_loops_under_the_switch():
    """tools/fuzz_mct.py / fuzz_trainer.py: device trees and whole trainer loops (self-play, augmentation, SGD hand-off, arena) sample for
    sample equal to the oracle with the fixed-point layers on both sides"""
    assert "0 mismatches" in _run("fuzz_mct.py", "40", "12")
    assert "0 mismatches" in _run("fuzz_trainer.py", "8", "13", timeout=1500)
