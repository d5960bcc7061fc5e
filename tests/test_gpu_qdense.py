"""GPU: the exact block-fixed-point dense layers (AZ_DENSE_I8=1: OthelloNet's fc1 / fc2 on the int8 matrix pipe, az_net.hip k_q_rows /
k_qgemm) against the CPU oracle under the same switch.  Product and oracle read the variable once per process, so every case runs in a
child process with the variable set; the checkers are the tools the default path is checked with."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


soak = pytest.mark.skipif(os.environ.get("AZ_SOAK") != "1", reason="opt-in (AZ_SOAK=1): the switch is an opt-in extra, its longer slices are recorded under profiles/")


def _run(tool, *args, timeout=900, on=True):
    env = dict(os.environ)
    env.pop("AZ_DENSE_I8", None)
    if on:
        env["AZ_DENSE_I8"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    return p.stdout


def test_network_equals_oracle_bit_for_bit_at_every_tile_plan():
    """tools/check_qdense.py: OthelloNet 8x8 / 6x6, batch sizes 1 ... 8200 (both tile plans, partial tiles, an empty board), 60 repeated
    forwards as a race screen of the hand-synchronised LDS pipeline"""
    out = _run("check_qdense.py")
    assert "mismatches 0" in out and out.count("bit-equal") >= 30 and "k_qgemm" in out


@soak
def test_forward_fuzz_under_the_switch():
    """tools/fuzz_net.py: random networks and batch sizes across the kernel variants (the Connect4 / TicTacToe nets keep their float32 chains)"""
    assert "0 mismatches" in _run("fuzz_net.py", "80", "11")


@soak
def test_single_game_searches_and_trainer_loops_under_the_switch():
    """tools/fuzz_mct.py / fuzz_trainer.py: device trees and whole trainer loops (self-play, augmentation, SGD hand-off, arena) sample for
    sample equal to the oracle with the fixed-point layers on both sides"""
    assert "0 mismatches" in _run("fuzz_mct.py", "40", "12")
    assert "0 mismatches" in _run("fuzz_trainer.py", "8", "13", timeout=1500)


def test_distance_from_float64_on_self_play_positions(tmp_path):
    """>= 1e5 positions from real self-play, random-init and post-SGD weights, torch float64 on the box as the
    yardstick.  Measured (profiles/r05_qdense_accuracy.txt): the fixed-point layers are CLOSER to float64 than the fma chains in every mean
    and in the value's maximum, and 1.04x / 1.35x the chains' distance in the policy's maximum (9.4e-9 against 9.0e-9; 6.8e-7 against
    5.1e-7 -- stock torch float32 on the same boards: 1.4e-8 / 4.5e-7).  The verdict's condition is `<=` on all eight figures, so it is NOT
    met in full; with the latency condition 2(d) also open the switch stays an opt-in extra.  This test keeps the
    measurement honest: means no worse than the chains', maxima within 1.5x of theirs, everything far inside golden G2's 1e-5."""
    import json
    path = str(tmp_path / "acc.npz")
    chain = json.loads(_run("qdense_accuracy.py", "make", path, on=False).strip().splitlines()[-1])  # make ends with the eval of its own (default) arithmetic
    fixed = json.loads(_run("qdense_accuracy.py", "eval", path, on=True).strip().splitlines()[-1])
    assert chain["positions"] >= 100000 and "fma chains" in chain["arithmetic"] and "int8" in fixed["arithmetic"]
    assert "k_qgemm" in fixed["sgd"]["dense_kernels"] and "k_qgemm" not in chain["sgd"]["dense_kernels"]
    for w in ("init", "sgd"):
        for k in ("pi_mean", "v_mean"):
            assert fixed[w][k] <= 1.10 * chain[w][k], (w, k, fixed[w][k], chain[w][k])
        for k in ("pi_max", "v_max"):
            assert fixed[w][k] <= 1.5 * chain[w][k] and fixed[w][k] < 1e-5 and chain[w][k] < 1e-5, (w, k, fixed[w][k], chain[w][k])
