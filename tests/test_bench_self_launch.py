"""bench.py --gpus N without a launcher: the parent must start its ranks as children, pass their failure on
and never hang.  No GPU here: the ranks fail at their first device call, which is exactly what these tests need."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_parent_relays_a_failing_run_as_non_zero():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--frame", "256x288"], env=_env(),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "launching 2 ranks" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout[-500:])


def test_parent_kills_a_run_that_does_not_finish():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(VKR_BENCH_LAUNCH_TIMEOUT="0.2"),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 124 and "killing it" in r.stderr and r.stdout.strip() == ""
    assert time.time() - t0 < 60


def test_mismatched_launcher_is_still_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
