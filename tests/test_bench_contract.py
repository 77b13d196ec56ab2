"""bench.py's host-side contract, checked without a GPU: the defaults the driver runs with, the extra blocks' child command (it must not
recurse into extras or time the CPU leg again), and the per-depth constants of the progressive sweep (reference config.py:40-41)."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_are_the_headline_workload(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.config, a.dtype, a.batch_per_gpu) == (1, "ffhq1024", "bf16", 4)
    assert a.steps >= 10 and a.warmup >= 1
    assert not a.no_extras and not a.no_b32 and not a.no_cpu_baseline and not a.sweep


def test_sweep_constants_follow_the_reference_schedule():
    assert bench.REF_BATCH_SIZES == [128, 128, 128, 64, 32, 16, 8, 4, 2]          # reference config.py:41 (one GPU)
    assert len(bench.SWEEP_GFLOP_PER_IMG) == len(bench.REF_BATCH_SIZES) == bench.CONFIGS["ffhq1024"]["depth"] + 1
    assert bench.SWEEP_GFLOP_PER_IMG[-1] * 1e9 == bench.CONFIGS["ffhq1024"]["flops_per_img"]
    assert bench.SWEEP_GFLOP_PER_IMG[5] * 1e9 == bench.CONFIGS["ffhq128"]["flops_per_img"]


def test_the_ffhq128_block_is_a_bounded_child_that_cannot_recurse(monkeypatch):
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        line = {"metric": "m", "value": 1.0, "unit": "img/s", "steps": 5, "warmup": 2, "ms_per_step": 2.0, "dtype": "fp32",
                "config": {"workload": "w"}, "evidence": "long text that is not embedded", "roofline": {"bound": "mfma"}}
        return types.SimpleNamespace(returncode=0, stdout="noise\n" + json.dumps(line) + "\n", stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    out = bench.extra_ffhq128_fp32_b64()
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and os.path.basename(cmd[1]) == "bench.py"
    for flag in ("--no-extras", "--no-cpu-baseline"):
        assert flag in cmd
    assert cmd[cmd.index("--config") + 1] == "ffhq128" and cmd[cmd.index("--dtype") + 1] == "fp32" and cmd[cmd.index("--batch-per-gpu") + 1] == "64"
    assert seen["kw"].get("timeout", 0) > 0
    assert out["value"] == 1.0 and out["roofline"] == {"bound": "mfma"} and "evidence" not in out and "wall_s" in out


def test_a_failing_child_is_recorded_not_raised(monkeypatch):
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: types.SimpleNamespace(returncode=3, stdout="", stderr="boom"))
    out = bench.extra_ffhq128_fp32_b64()
    assert "error" in out and "rc 3" in out["error"]

    def timeout(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, kw["timeout"])
    monkeypatch.setattr(subprocess, "run", timeout)
    assert "exceeded" in bench.extra_ffhq128_fp32_b64(timeout_s=1.0)["error"]


def _full_record():
    """The complete record of a default invocation (round 5's 21 KB line, kept as the first line of profiles/r05_bench_default_head.json):
    the input `compact_line` must shrink."""
    with open(os.path.join(ROOT, "profiles", "r05_bench_default_head.json")) as fh:
        return json.loads(fh.readline())


def test_the_last_stdout_line_stays_driver_readable():
    """Round 5's record went unparsed: the line had grown to 21 KB and the driver keeps ~8 KB of stdout.  The line is now built by
    `compact_line`: < 4 KB, one JSON object, with the contract's keys, `roofline` and `cpu_baseline`."""
    full = _full_record()
    assert len(json.dumps(full)) > 3 * bench.LINE_LIMIT                     # the stub really is the oversized record
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line) < 4096 == bench.LINE_LIMIT
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in j, k
    assert j["value"] == round(full["value"], 4) and j["config"]["workload"].startswith("ffhq1024")
    roof = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert "layers" not in roof and "top_layers" not in roof
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] in ("port", "reference") and j["cpu_baseline"]["cores"] >= 1
    assert j["b32"]["value"] > 0 and j["b32"]["roofline"]["frac"] > 0 and "layers" not in j["b32"]["roofline"]
    assert j["detail"] == "gpurun_out/bench_detail.json"


def test_optional_blocks_are_dropped_before_the_line_grows_past_the_limit():
    full = _full_record()
    full["sweep_top_depths"]["rows"] = full["sweep_top_depths"]["rows"] * 40          # an absurdly long optional block
    j = json.loads(bench.compact_line(full, "d.json"))
    assert "sweep_top_depths" not in j and "roofline" in j and "cpu_baseline" in j and "b32" in j
    full = _full_record()
    full["cpu_baseline"] = {"value": None, "error": "x" * 5000}
    line = bench.compact_line(full, None)
    assert len(line) < bench.LINE_LIMIT and json.loads(line)["cpu_baseline"]["value"] is None


def test_the_detail_file_holds_the_full_record(tmp_path):
    full = _full_record()
    rel = bench.write_detail(full, str(tmp_path / "sub" / "detail.json"))
    assert rel is not None
    with open(tmp_path / "sub" / "detail.json") as fh:
        assert json.load(fh) == full
