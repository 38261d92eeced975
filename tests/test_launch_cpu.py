"""CPU: `python bench.py --gpus N` / `run_segments.py --ranks N` as BARE commands launch their own ranks.

No GPU is needed: GSR_BENCH_LAUNCH_CHECK=1 / --launch-check stop after the process group's self-diagnosis (nothing is rendered).
What must hold: a bare `--gpus 2` comes back as a TWO-rank job (never world 1 under the name of 2), rank 0 alone prints, and when
the devices are not there the command says so in one JSON line and exits non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    out = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    return out, [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


def test_bare_bench_command_launches_its_own_ranks():
    out, lines = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                      GSR_BENCH_LAUNCH_CHECK="1", GSR_BENCH_BACKEND="gloo")
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout                      # rank 0 alone prints
    d = lines[0]
    assert d["n_gpus"] == 2 and d["launch_check"] is True and d["self_launched"] is True and d["value"] is None
    r = d["rccl"]
    assert r["world"] == 2 and r["ranks_seen"] == [0, 1] and r["all_ranks_present"] and r["payload_ok"] and list(r["link_GBps"]) == ["0<->1"]


def test_bench_refuses_a_world_that_is_not_what_gpus_says():
    # a launcher that started ONE rank for --gpus 2: an error line, not a 1-GPU number
    out, lines = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert out.returncode == 2 and len(lines) == 1 and lines[0]["value"] is None and "WORLD_SIZE=1" in lines[0]["error"], (out.stdout, out.stderr[-2000:])
    assert lines[0]["n_gpus"] == 2


def test_bare_bench_command_without_the_devices_fails_with_a_json_line():
    # this container has no GPU at all: --gpus 2 bare must not start anything
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices present")
    out, lines = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2"])
    assert out.returncode == 3 and len(lines) == 1, (out.stdout, out.stderr[-2000:])
    assert lines[0]["value"] is None and lines[0]["n_gpus"] == 2 and "visible" in lines[0]["error"]


def test_bare_run_segments_command_launches_its_own_ranks():
    out, lines = _run([os.path.join(ROOT, "3dgs_hierarchical_training_amd", "run_segments.py"), "--ranks", "2", "--backend", "gloo", "--launch-check"])
    assert out.returncode == 0, out.stderr[-3000:]
    done = [l for l in lines if l.get("phase") == "launch_check"]
    assert len(done) == 1 and done[0]["world"] == 2 and done[0]["ranks_seen"] == [0, 1] and done[0]["selftest_ok"] is True, lines
    out, lines = _run([os.path.join(ROOT, "3dgs_hierarchical_training_amd", "run_segments.py"), "--ranks", "2"])
    assert out.returncode == 3 and lines and lines[-1]["phase"] == "error" and "visible" in lines[-1]["error"], (out.stdout, out.stderr[-2000:])
