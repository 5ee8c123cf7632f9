"""bench.py starts its own ranks: `--gpus 2` outside a launcher's environment must run two processes that find each
other, and the line must say so.  The ranks share GPU 0 here and talk over gloo (JEA_DIST_BACKEND); on a node with
a GPU per rank the same code runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + "\n".join(ln for ln in r.stderr.splitlines() if ln.startswith("[rank"))[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_runs_two_ranks_of_the_stripe_protocol():
    line = _bench("--gpus", "2", "--config", "c4", "--c4-size", "1024", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                  JEA_DIST_BACKEND="gloo")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["comm_backend"] == "gloo"
    assert line["config"]["file_checked_against"] == "oracle, byte for byte"
    assert line["value"] > 0 and line["config"]["stripe_rows"] == 512


def test_bench_gpus_2_shards_a_batch_by_image():
    line = _bench("--gpus", "2", "--config", "c3", "--frames", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                  JEA_DIST_BACKEND="gloo")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["config"]["workload"].startswith("batch of 6 1920x1080 frames") and "3 per rank" in line["config"]["workload"]


def test_bench_c4_over_rccl_with_one_rank():
    """The RCCL calls of the stripe protocol (all-gathers of bytes from device memory to device memory, issued under the
    context's stream) on the one GPU there is: a process group of a single rank over the `nccl` backend."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    line = _bench("--gpus", "1", "--config", "c4", "--c4-size", "2048", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                  JEA_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["comm_backend"] == "nccl (RCCL)"
    assert line["config"]["file_checked_against"] == "oracle, byte for byte"


def test_default_run_on_two_ranks_carries_the_stripe_measurement():
    line = _bench("--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--c4-size", "2048", "--pool", "2",
                  JEA_DIST_BACKEND="gloo")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value"] > 0
    c4 = line["other_measurements"]["c4_16384x16384_as_stripes"]
    assert "error" not in c4, c4
    assert c4["config"]["file_checked_against"] == "oracle, byte for byte" and c4["value"] > 0


def test_default_run_prints_its_line_when_the_stripe_measurement_does_not_finish():
    """The measurement with collectives runs under a time limit: with a limit of zero the headline line must still
    appear, say what happened, and the ranks must leave with exit code 0."""
    line = _bench("--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--c4-size", "2048", "--pool", "2",
                  JEA_DIST_BACKEND="gloo", JEA_C4_LIMIT_SECONDS="0")
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "did not finish" in line["other_measurements"]["c4_16384x16384_as_stripes"]["error"]
