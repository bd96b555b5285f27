"""`python bench.py --gpus N` starts its own N ranks (bench.launch_ranks): one process per GPU with the environment torch.distributed.run
would set, rank 0's JSON line relayed, failures propagated, and NO fall-back to fewer devices.  Checked here without a GPU: the ranks
run the launch-only leg (communicator census over gloo, no device work; BENCH_ASSUME_DEVICES stands in for the device count)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, **env):
    full = dict(os.environ, BENCH_LAUNCH_ONLY="1", BENCH_DIST_BACKEND="gloo", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env:
            full.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=full, capture_output=True, text=True, timeout=300)


def test_gpus_2_starts_two_ranks_and_relays_one_line():
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], BENCH_ASSUME_DEVICES="2")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["launched_by"] == "bench.py"
    assert doc["communicator"]["ranks_seen"] == 2 and doc["communicator"]["torch_world_size"] == 2
    assert doc["communicator"]["world_size_env"] == 2


def test_more_ranks_than_devices_is_an_error_not_a_smaller_run():
    res = _run(["--gpus", "8"], BENCH_ASSUME_DEVICES="1")
    assert res.returncode != 0
    assert "needs 8 HIP devices, 1 visible" in res.stderr
    assert res.stdout.strip() == ""                       # no JSON line: nothing that could be read as a measurement


def test_launcher_and_flag_must_agree():
    res = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    assert res.returncode != 0
    assert "--gpus 4 but the launcher started WORLD_SIZE=2" in res.stderr


def test_a_failing_rank_fails_the_job():
    # rank 1 is given a rendezvous nobody listens on: it must take the whole job down with a non-zero exit code
    res = _run(["--gpus", "2"], BENCH_ASSUME_DEVICES="2", BENCH_TEST_BREAK_RANK="1")
    assert res.returncode != 0
    assert "rank 1 exited with code" in res.stderr
