"""bench.py's bookkeeping that needs no GPU: the line of a multi-rank default run must be printed whatever becomes of
the measurement with collectives in it (Job.finish_after), and an embedded line keeps only what other_measurements
wants."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_embedded_line_keeps_the_measurement_and_drops_the_rest():
    bench = _bench_module()
    line = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 2, "steps": 3, "warmup": 1, "ms_per_step": 0.5, "scaling": "strong",
            "data": "d", "config": {"workload": "w"}, "roofline": {"frac": 0.1}, "cpu_baseline": {"value": 2}, "vs_baseline": None,
            "higher_is_better": True}
    kept = bench.embedded(line)
    assert kept["value"] == 1.0 and kept["config"] == {"workload": "w"} and kept["roofline"] == {"frac": 0.1}
    assert "cpu_baseline" not in kept and "vs_baseline" not in kept


_STUB = textwrap.dedent('''
    import importlib.util, json, os, sys, time
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(%(root)r, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

    class Cuda:
        def set_device(self, i): pass
    class Torch:
        cuda = Cuda()
    job = bench.Job.__new__(bench.Job)
    job.torch, job.device_index, job.dist, job.rank = Torch(), 0, None, 0
    job.ranks_seen, job.comm_name = 2, "stub"
    mode = sys.argv[1]
    def measure():
        if mode == "fine":
            return {"metric": "m", "value": 5.0, "unit": "u", "config": {"workload": "w"}}
        if mode == "raises":
            raise RuntimeError("refused")
        time.sleep(60)
    job.finish_after({"metric": "headline", "value": 1.0}, "c4", measure, 1)
    print("returned normally")
''')


def _run(mode):
    r = subprocess.run([sys.executable, "-c", _STUB % {"root": ROOT}, mode], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr
    return json.loads(lines[0]), r.stdout


def test_line_carries_the_side_measurement_when_it_finishes():
    line, out = _run("fine")
    assert line["value"] == 1.0 and line["other_measurements"]["c4"]["value"] == 5.0 and "returned normally" in out


def test_line_says_what_failed_when_the_side_measurement_raises():
    line, out = _run("raises")
    assert line["value"] == 1.0 and "RuntimeError: refused" in line["other_measurements"]["c4"]["error"] and "returned normally" in out


def test_line_is_printed_and_the_process_leaves_when_the_side_measurement_hangs():
    line, out = _run("hangs")
    assert line["value"] == 1.0 and "did not finish within 1 s" in line["other_measurements"]["c4"]["error"]
    assert "returned normally" not in out      # left through os._exit(0), without the barrier
