"""N > 1 path on CPU: two processes, gloo backend, rendezvous on 127.0.0.1. Checks the frame sharding used by
bench.py --gpus N (disjoint cover of the batch, no data-path collective) and the timing reduction."""
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


def _worker(rank, world, port, total_frames, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts, shard_frames
    mine = shard_frames(total_frames, rank, world)
    # every rank "codes" its own frames: stand-in work = a checksum of the frame seeds (12345 + i as in bench.py)
    local = torch.tensor([sum(12345 + i for i in mine)], dtype=torch.int64)
    everyone = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(everyone, local)
    lists = [None] * world
    dist.all_gather_object(lists, mine)
    elapsed = barrier_and_max(0.5 + rank)
    total = gather_counts(len(mine))
    out.put((rank, mine, int(sum(int(t) for t in everyone)), lists, elapsed, total))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_world2():
    world, total_frames = 2, 257
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mine, checksum, lists, elapsed, total in results:
        assert sorted(x for l in lists for x in l) == list(range(total_frames)), "shards must cover the batch exactly once"
        assert set(lists[0]).isdisjoint(lists[1])
        assert abs(len(lists[0]) - len(lists[1])) <= 1
        assert checksum == sum(12345 + i for i in range(total_frames))
        assert elapsed == 1.5, "timing is the max over ranks"
        assert total == total_frames


def test_single_process_defaults():
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts, shard_frames
    assert shard_frames(10, 0, 1) == list(range(10))
    assert shard_frames(10, 3, 4) == [3, 7]
    assert barrier_and_max(0.25) == 0.25
    assert gather_counts(7) == 7
