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


def test_bench_launcher_starts_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` without torchrun: the launcher path starts the two ranks itself (child
    torch.distributed.run), shards a fixed clip set clip i -> rank i mod 2 (strong scaling), broadcasts rank 0's
    weights and prints ONE JSON line.  --dry-run/--backend gloo keeps it on the CPU with a stub step."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                          "--clips", "3", "--total-clips", "7", "--dry-run", "--backend", "gloo"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    assert line["clips_per_rank"] == [4, 3] and line["weights_from_rank0"] is True


def test_batches_for_rank_cover_the_clip_set():
    import bench
    for total, world, resident in [(2048, 8, 256), (2048, 1, 256), (7, 2, 3), (5, 8, 256)]:
        seen = []
        for r in range(world):
            for b in bench.batches_for_rank(total, world, r, resident):
                assert 0 < len(b) <= resident and all(c % world == r for c in b)
                seen += b
        assert sorted(seen) == list(range(total))


def _dry_run(gpus, *extra):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0",
                          "--dry-run", "--backend", "gloo"] + list(extra), capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_eight_rank_dry_run_is_the_strong_scaling_split_of_the_one_rank_run():
    """The driver's scaling curve compares N = 1, 2, 4, 8 like for like: the SAME clip set (config.clips_per_step) at every
    N.  World size 8 over gloo with the bench's default sizes (2048 clips, 256 resident): a rank's share is ONE resident batch, cut
    in two so that both of its HIP streams have work (bench.batches_for_rank, --overlap 2), a disjoint cover of the set, weights generated on rank 0 only and received by the others, ONE JSON line."""
    one = _dry_run(1, "--clips", "256", "--total-clips", "2048")
    eight = _dry_run(8, "--clips", "256", "--total-clips", "2048")
    assert one["config"]["clips_per_step"] == eight["config"]["clips_per_step"] == 2048
    assert one["scaling"] == eight["scaling"] == "strong"
    assert one["batches_per_rank"] == [[256] * 8] and eight["batches_per_rank"] == [[128, 128]] * 8
    assert one["disjoint_cover"] and eight["disjoint_cover"]
    assert eight["n_gpus"] == 8 and eight["clips_per_rank"] == [256] * 8 and eight["weights_from_rank0"] is True
    # no data-path collective.  bench.timed_region counts every torch.distributed call issued between its two
    # barriers (max over ranks); the collectives that do exist (weight broadcast, the barriers, the max-reduce) are outside.
    assert eight["collectives_in_timed_region"] == 0 and eight["collectives_total"] >= 4
    assert one["collectives_in_timed_region"] == 0 and one["collectives_total"] == 0
    # one eager Python launch loop per GPU: every rank pins itself to its own slice of the host cores (bench.pin_rank_cores); with
    # at least as many usable cores as ranks the slices are disjoint (this container: 8 cores for 8 ranks)
    import bench
    if bench.usable_cores() >= 8:
        assert eight["cores_disjoint"] is True and all(len(c) >= 1 for c in eight["cores_per_rank"]), eight["cores_per_rank"]
    assert one["cores_disjoint"] is None


def test_timed_region_counts_collectives():
    """The counter behind `collectives_in_timed_region` really sees torch.distributed calls made inside the bracket
    (a one-rank gloo group, a step that -- wrongly -- synchronises the ranks)."""
    import socket

    import torch.distributed as dist

    import bench

    bench.install_collective_counter()
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        _, inside = bench.timed_region(lambda: dist.barrier(), 3, 1, lambda: None)
        _, clean = bench.timed_region(lambda: None, 3, 1, lambda: None)
    finally:
        dist.destroy_process_group()
    assert inside == 3 and clean == 0


def test_only_rank0_generates_weights():
    """build_workload: ranks other than 0 allocate the state_dict's shapes without generating values (they are overwritten
    by the flat broadcast); the shape tables match the generators key for key."""
    import bench

    for gen, shapes in ((bench.seeded_state_dict, bench.seeded_state_dict_shapes), (bench.vitdet_state_dict, bench.vitdet_state_dict_shapes)):
        a, b = gen(), shapes()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
