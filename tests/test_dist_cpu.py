"""CPU, world_size 2 over gloo: the multi-process plumbing bench.py uses on N GPUs (weights broadcast
from rank 0 as one flat buffer, max-over-ranks timing, clip -> rank sharding).  The data path itself has
no collective: clips are independent."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    sd = {"a": torch.full((3, 4), float(rank + 1)), "b": torch.arange(5.0) * (rank + 1)}
    extra = {"c": torch.full((2,), 10.0 * (rank + 1))}
    bench.broadcast_weights(sd, extra, torch.device("cpu"), rank)
    ok = torch.equal(sd["a"], torch.full((3, 4), 1.0)) and torch.equal(sd["b"], torch.arange(5.0)) and \
        torch.equal(extra["c"], torch.full((2,), 10.0))
    t = bench.max_over_ranks(1.0 + rank, torch.device("cpu"))
    mine = bench.clips_for_rank(7, world, rank)
    out.put((rank, ok, t, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_plumbing():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert [t for _, _, t, _ in res] == [2.0, 2.0]              # MAX over ranks
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]  # clip i -> rank i mod world, disjoint cover
