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


def _worker8(rank, world, port, total_frames, out):
    """one of the eight ranks of `bench.py --gpus 8` as far as the CPU can play it: the sharding, the two reductions of the timed region and the
    placement of its four launch threads on a two-socket node (ranks 0-3 next to socket 0, ranks 4-7 next to socket 1; SMT siblings at +128)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts, parse_cpulist, plan_affinity, shard_frames
    mine = shard_frames(total_frames, rank, world)
    sockets = [parse_cpulist("0-63,128-191"), parse_cpulist("64-127,192-255")]
    near = [sockets[r // 4] for r in range(world)]
    cpus = plan_affinity(rank, world, 4, range(256), near, avoid={0, 1, 64})  # (the launcher and a stranger keep three cores busy)
    everything = [None] * world
    dist.all_gather_object(everything, (mine, cpus))
    elapsed = barrier_and_max(1.0 + 0.125 * rank)
    total = gather_counts(len(mine))
    out.put((rank, everything, elapsed, total))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_world8():
    """ the shape of the driver's --gpus 8 run: 8 ranks x 4 launch threads, BASELINE config 5's 256 frames"""
    world, total_frames = 8, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, total_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, everything, elapsed, total in results:
        shards = [e[0] for e in everything]
        plans = [e[1] for e in everything]
        assert sorted(x for s in shards for x in s) == list(range(total_frames)), "shards must cover the batch exactly once"
        assert all(len(s) == total_frames // world for s in shards)
        assert all(p is not None and len(p) == 4 and len(set(p)) == 4 for p in plans)
        assert len(set().union(*map(set, plans))) == 32, "launch threads of different ranks never share a core"
        for r, p in enumerate(plans):
            assert set(p) <= set(range(0, 64)) | set(range(128, 192)) if r < 4 else set(p) <= set(range(64, 128)) | set(range(192, 256))
            assert not set(p) & {0, 1, 64}
        assert elapsed == 1.0 + 0.125 * 7 and total == total_frames


def test_single_process_defaults():
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts, shard_frames
    assert shard_frames(10, 0, 1) == list(range(10))
    assert shard_frames(10, 3, 4) == [3, 7]
    assert barrier_and_max(0.25) == 0.25
    assert gather_counts(7) == 7


def test_launch_thread_affinity_plan(tmp_path):
    """gpujpeg_amd.sharding.plan_affinity: the launch threads of a rank get cores of the NUMA node of its GPU, the ranks of one node
    disjoint cores; without platform information the allowed cores are split by rank; a thread can really be bound."""
    import os
    from gpujpeg_amd.sharding import busy_cpus, gpu_local_cpus, parse_cpulist, pin_current_thread, plan_affinity
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(128))
    near = [list(range(0, 64))] * 4 + [list(range(64, 128))] * 4
    plans = [plan_affinity(r, 8, 5, allowed, near) for r in range(8)]
    assert all(len(p) == 5 and len(set(p)) == 5 for p in plans)
    assert all(set(plans[r]) <= set(near[r]) for r in range(8))
    assert len(set().union(*map(set, plans))) == 40  # disjoint
    # unknown topology: the allowed set split by rank; fewer cores than threads: cores repeat, ranks stay apart
    a, b = plan_affinity(0, 2, 3, range(4), [None, None]), plan_affinity(1, 2, 3, range(4), [None, None])
    assert set(a) == {0, 1} and set(b) == {2, 3} and len(a) == len(b) == 3
    # near cores outside what the process may use fall back to the allowed set
    assert set(plan_affinity(0, 1, 2, [5, 6], [[0, 1]])) <= {5, 6}
    assert plan_affinity(0, 1, 0, [0], [None]) is None
    # cores that strangers keep busy are left out while the share has enough others
    assert plan_affinity(0, 2, 3, range(16), [list(range(8))] * 2, avoid={0, 1}) == [2, 3, 2]
    assert set(plan_affinity(0, 1, 4, range(16), [list(range(8))], avoid={0, 1, 2})) == {3, 4, 5, 6}
    assert plan_affinity(0, 1, 4, range(16), [list(range(4))], avoid={0, 1, 2, 3}) == [0, 1, 2, 3]  # all busy: no better choice
    stat = tmp_path / "stat"  # (identical snapshots: nothing is busy; the parser takes per-core lines only)
    stat.write_text("cpu  10 0 10 100 0 0 0 0 0 0\ncpu0 5 0 5 50 0 0 0 0 0 0\ncpu1 5 0 5 50 0 0 0 0 0 0\nintr 1\n")
    assert busy_cpus(0.0, stat=str(stat)) == set()
    assert isinstance(busy_cpus(0.01), set)
    # sysfs lookup (a fake tree) and a real bind of this thread, undone afterwards
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "local_cpulist").write_text("64-127\n")
    assert gpu_local_cpus("0000:C1:00.0", sysfs=str(tmp_path)) == list(range(64, 128))
    assert gpu_local_cpus("0000:00:00.0", sysfs=str(tmp_path)) is None
    before = os.sched_getaffinity(0)
    try:
        one = sorted(before)[-1]
        assert pin_current_thread(one) and os.sched_getaffinity(0) == {one}
    finally:
        os.sched_setaffinity(0, before)
