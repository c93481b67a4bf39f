"""CPU: `python bench.py --gpus N` invoked the way the driver invokes things (no launcher, no WORLD_SIZE).

The parent must start its N ranks as child processes before touching the GPU, relay rank 0's single JSON line and the
ranks' exit code.  ``--stub`` swaps the HIP workload for a CPU tensor going through the same
bring-up: rank environment, process group, GradientBucket's flat all-reduce, the barrier-bracketed timed region and
the MAX-over-ranks reduction -- the launcher is what is under test.  The torchrun form must keep working as well.
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ISD_DIST_BACKEND"] = "gloo"
    return env


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_plain_command_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1
    assert line["config"]["parallelism"] == "dp2" and line["config"]["self_launched"] is True
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_plain_command_reports_a_failing_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--stub",
                        "--stub-fail-rank", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
    assert "fails on purpose" in r.stderr


def test_torchrun_form_still_works():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--stub"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["config"]["self_launched"] is False


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "0", "--stub"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1


def test_rank_count_mismatch_is_an_error():
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0 and "disagree" in r.stderr
