"""abi.Comm never leaves a rank behind: when RCCL cannot be loaded — on rank 0, where the id is made, or on
any rank — every rank raises instead of some of them blocking in a collective.  CPU only: two processes over gloo, the
C-ABI library is asked to dlopen a file that does not exist."""
import multiprocessing as mp
import os
import socket

import pytest

from vk_renderer_amd import abi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, with_agree, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VKR_RCCL_LIBRARY="/nonexistent/librccl.so")
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def share(ident):
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    try:
        abi.Comm(rank, world, share, agree if with_agree else None)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, f"raised: {e}"))
    dist.barrier()  # both ranks are still in step on the control plane
    dist.destroy_process_group()


@pytest.mark.parametrize("with_agree", [True, False])
def test_comm_without_rccl_raises_on_every_rank(with_agree):
    if not os.path.exists(abi.PRODUCT_LIB):
        pytest.skip("HIP library not built yet")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, with_agree, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "a rank was left blocked in a collective"
    got = dict(q.get(timeout=5) for _ in range(2))
    assert all(v.startswith("raised") for v in got.values()), got
    assert all(p.exitcode == 0 for p in procs)
