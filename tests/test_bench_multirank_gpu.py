"""bench.py's N > 1 branch end to end on a one-GPU box: launched exactly as the driver launches it (torch.distributed.run,
one process per rank), with the rehearsal switch that puts every rank on cuda:0 over gloo.  Checks the contract of the
one JSON line (only rank 0 prints, whole-job value, strips grid, max-over-ranks timing)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_bench_line_of_a_multi_rank_run(world):
    env = dict(os.environ, VKR_BENCH_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
           "--frame", f"512x{288 * world}"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"exactly one line on stdout, got {len(lines)}: {lines[:3]}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["config"]["grid"] == [1, world] and d["config"]["frame"] == [512, 288 * world] and d["config"]["tile_per_gpu"] == [512, 288]
    assert d["config"]["halo_px"] == 48 and d["config"]["gathered_hiz_mips"] == 4
    assert d["config"]["baseline_config"] is None  # a rehearsal frame, not the 15360x8640 frame of config 4
    # whole-job value: all ranks' pixels over the slowest rank's time
    assert abs(d["value"] - 512 * 288 * world / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] in d["per_pass_ms"] and d["roofline"]["frac"] > 0
    assert "cpu_baseline" not in d  # rank 0, N = 1 only


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts the ranks as child processes before
    it touches the GPU, relays rank 0's one line and exits with their status."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VKR_BENCH_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--frame", "512x576"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["config"]["grid"] == [1, 2] and d["config"]["launcher"] == "self (bench.py started its ranks)"
