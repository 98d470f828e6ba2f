"""The multi-GPU code path (RCCL all-gathers issued asynchronously, whole-frame Hi-Z / normals / albedo, staged
frame with TAA overlapping the first gather) rehearsed on ONE GPU with a one-rank RCCL group: results must be
bit-identical to the plain single-GPU frame.  (Real multi-rank runs are covered by the gloo tests on the oracle
and by the driver's N = 2, 4, 8 scaling runs.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

OUTPUTS = ("rays", "raw", "reflections", "blurred", "filtered", "acc_ao", "taa_target", "depth", "dn", "dv")


def test_tiled_path_on_one_rank_matches_plain_frame():
    import torch
    import torch.distributed as dist

    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        W, H = 512, 288
        results = []
        for force in (False, True):
            t = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device, force_tiled=force)
            assert t.tiled == force
            t.prepare()
            for _ in range(2):  # second frame consumes the histories written by the first
                t.step()
            t.flush()
            t.backend.sync()
            results.append({n: t.frame.download(n).to_host().copy() for n in OUTPUTS})
            if force:
                assert t.frame.last_tasks()[-1] == "SSSR_blur", "staged frame: TAA and GTAO run early, filter + blur last"
            t.frame.close()
        for n in OUTPUTS:
            assert np.array_equal(results[0][n], results[1][n]), f"{n}: tiled code path differs from the plain frame"
    finally:
        dist.destroy_process_group()


def test_bench_rebuilds_the_frame_on_balanced_strips():
    """bench.py at N > 1 measures every rank's passes, shares the times and rebuilds the C++ tiled frame on strips cut by
    cost (vkrh_balance_rows) before its timed region.  One rank cannot move its strip, but VKR_BALANCE_REBUILD=1 takes the
    rehearsal through the same sequence: measure, all_gather_object, close, rebuild with row_bounds, prepare, warm up."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29548", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", VKR_BALANCE_REBUILD="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-tiled", "--frame", "1024x576", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, out.stdout
    res = json.loads(line[0])
    assert res["config"]["strip_rows"] == [576]
    assert len(res["config"]["strip_balance"]) == 2 and all(p["rows"] == [576] and p["compute_ms"][0] > 0 for p in res["config"]["strip_balance"])
    assert res["value"] > 0
    # the calibration run reports how long the compute stream stood still for each exchange, per rank
    assert set(res["exchange_wait_ms"]) == {"hiz_gather", "albedo_gather", "taa_halo", "ao_halo", "ssr_halo"}
    assert all(len(v) == 1 and 0 <= v[0] < 1.0 for v in res["exchange_wait_ms"].values())
