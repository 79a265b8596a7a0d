"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (neuspeech1_amd.dp.GradReducer) averages the flat
gradient buffer chunk by chunk, in any chunk order, and every rank ends with identical values."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neuspeech1_amd.dp import GradReducer
    n = 10_000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(g)
    # chunks in backward-completion order: upper layers, lower layers, conv stem
    for lo, hi in ((0, 4000), (4000, 7000), (7000, n)):
        red.on_ready(lo, hi)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok = torch.allclose(g, expect)
    # inf on one rank must surface on every rank (GradScaler skip decisions stay identical)
    h = torch.ones(8)
    if rank == 1:
        h[3] = float("inf")
    r2 = GradReducer(h)
    r2.on_ready(0, 8)
    r2.finish()
    q.put((rank, ok, bool(torch.isinf(h[3]))))
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(30)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res
