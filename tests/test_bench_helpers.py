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


# ---------------------------------------------------------------------------------------------- the line of record
def _strict(line):
    import json

    def refuse(name):
        raise AssertionError(f"non-finite constant {name} in the line of record")
    return json.loads(line, parse_constant=refuse)


def _full_result():
    """a full result as main() assembles it: last round's own (profiles/r04_bench.json carries every nested object the run produced)"""
    import json
    import os
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))


def test_line_of_record_is_small_strict_and_flat():
    full = _full_result()
    assert len(__import__("json").dumps(full)) > 20000  # the input is the 23 KB object that broke the driver's parser in round 4
    line = bench.record_line(full)
    assert len(line) < 8192 and len(line) <= bench.LINE_BUDGET and "\n" not in line
    d = _strict(line)
    assert set(bench.TOP_KEYS) <= set(d) and d["detail"] == bench.DETAIL_NAME
    for k in ("config", "roofline", "cpu_baseline"):
        assert isinstance(d[k], dict) and d[k], k
        assert all(v is None or isinstance(v, (bool, int, float, str)) for v in d[k].values()), k  # scalars only
        assert all(len(v) <= 110 for v in d[k].values() if isinstance(v, str)), k
    assert set(d["config"]) <= set(bench.CONFIG_KEYS) and set(d["roofline"]) <= set(bench.ROOF_KEYS) and set(d["cpu_baseline"]) <= set(bench.CPU_KEYS)
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["config"]["workload"].startswith("BASELINE configs[1] literal") and d["value"] == float(f"{full['value']:.6g}")
    assert all(not isinstance(v, (dict, list)) for k, v in d.items() if k not in ("config", "roofline", "cpu_baseline"))


def test_line_of_record_survives_hostile_values():
    full = _full_result()
    full["roofline"]["frac"] = float("nan")
    full["roofline"]["achieved"] = float("inf")
    full["config"]["workload"] = "w" * 5000
    full["config"]["us_per_lockstep"] = {"nested": 1}  # a nested object where a scalar is expected: dropped, not serialised
    full["cpu_baseline"]["sample"] = "line one\nline two " * 50
    full["errors"] = {"config4": "RuntimeError('x' * 1000)" * 40}
    for i in range(200):  # keys nobody listed never reach the line
        full["config"][f"extra_{i}"] = "y" * 100
    line = bench.record_line(full)
    d = _strict(line)
    assert len(line) < 8192 and "\n" not in line
    assert d["roofline"]["frac"] is None and d["roofline"]["achieved"] is None and "us_per_lockstep" not in d["config"]
    assert len(d["config"]["workload"]) <= 110 and len(d["errors"]) <= 300 and not any(k.startswith("extra_") for k in d["config"])


def test_line_of_record_refuses_an_incomplete_result():
    import pytest
    full = _full_result()
    del full["value"]
    with pytest.raises(ValueError):
        bench.record_line(full)


def test_detail_file_is_strict_json(tmp_path, monkeypatch):
    import json
    full = _full_result()
    full["latency"]["nan_here"] = float("nan")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.write_detail(full)
    for p in (tmp_path / bench.DETAIL_NAME, tmp_path / "gpurun_out" / bench.DETAIL_NAME):
        d = _strict(p.read_text())
        assert d["latency"]["nan_here"] is None and "config5" in d and "saturated" in d


def test_dry_run_prints_one_strict_line_last():
    """`bench.py` under AZ_BENCH_DRYRUN (no GPU): the last non-empty line of stdout is strict JSON and small"""
    import os
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, AZ_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines[-1]) < 8192
    d = _strict(lines[-1])
    assert d["dryrun"] is True and d["n_gpus"] == 1
