"""bench.py's record plumbing on the CPU: the driver's record keeps the scalar fields of `config` / `roofline` / `cpu_baseline` (nested
objects dropped, strings cut), so the fields that have to survive are flat scalars placed in front, and the
process count of the CPU baseline's multi-process leg names its source (item 8)."""
import importlib
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def test_ordered_puts_the_named_scalars_first_and_nested_objects_last():
    d = {"kernels": {"a": 1}, "note": "x" * 200, "frac": 0.7, "bound": "mfma", "traffic": None, "extra": 3, "lst": [1, 2], "kernel": "k_gemm"}
    o = bench.ordered(d, bench.ROOF_FIRST)
    keys = list(o)
    assert keys[:4] == ["bound", "kernel", "frac", "traffic"]  # ROOF_FIRST's order, whatever the insertion order was
    assert keys.index("extra") < keys.index("note") < keys.index("kernels") and keys.index("note") < keys.index("lst")
    assert o == d  # nothing lost


def test_compact_is_flat():
    r = {"value": 5000.0, "examples_per_sec": 3e5, "ms_per_step": 6000.0,
         "roofline": {"end_to_end_frac": 0.82, "kernel": "k_gemm_solo (fc1 + fc2: two launches per forward)", "frac": 0.83, "forward_frac": 0.88, "kernels": {}}}
    c = bench.compact(r)
    assert all(not isinstance(v, (dict, list)) for v in c.values())
    assert c["games_per_sec"] == 5000.0 and c["dominant_frac"] == 0.83 and len(c["dominant_kernel"]) <= 40


def test_cpu_share_names_its_source():
    cores, source, affinity, quota = bench.cpu_share()
    assert cores >= 1 and affinity >= cores and source in ("cgroup cpu quota", "sched_getaffinity")
    assert (quota is not None and quota < affinity) == (source == "cgroup cpu quota")


def test_stage_flops_are_the_surveys():
    assert sum(bench.stage_flops(8, 8, 1024, 512, 65)) == 4339712  # OthelloNet 8x8
    assert sum(bench.stage_flops(7, 6, 64, 32, 7)) == 1306752      # Connect4Net


def test_a_failing_prototype_child_is_recorded_and_does_not_fail_the_bench():
    """run_dense_i8_prototype starts bench.py once more per size in a child process with AZ_DENSE_I8=1; without a GPU the child cannot
    run -- the extra must come back with the error in it instead of raising (it must never take the line of record down)"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    out = bench.run_dense_i8_prototype(sims=4, sizes=(8,))
    assert set(out["sizes"]) == {"8"} and "error" in out["sizes"]["8"] and "games_per_sec" not in out["sizes"]["8"]
