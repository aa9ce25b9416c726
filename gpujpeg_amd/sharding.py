"""Frame sharding across the GPUs of one node.

Frames are independent and a coder is bound to one device, so a batch shards by frame with no data-path
collective: rank r of W takes frames r, r+W, r+2W, ... The only communication is the barrier around the timed
region and the max-over-ranks reduction of the elapsed time, both through torch.distributed ("nccl" = RCCL on
the GPUs, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_frames(total_frames, rank, world):
    """Indices of the frames rank `rank` processes (static round-robin, the reference's CLI order preserved per rank)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return list(range(rank, total_frames, world))


def barrier_and_max(elapsed_seconds, device=None):
    """Max over ranks of a local duration; no-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_seconds)
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_count, device=None):
    """Sum over ranks of the number of frames processed (used to report whole-job throughput)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(local_count)
    t = torch.tensor([local_count], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


# ---------------------------------------------------------------- host threads next to their GPU
# A rank drives its GPU from a handful of launch threads (one per pipeline). On a two-socket node the launch latency of a thread that
# runs on the far socket is visibly higher and the pinned staging buffers land on the wrong memory, so each rank takes cores of the NUMA
# node its GPU hangs off, disjoint from the other ranks of the node. (The reference leaves placement to the caller; its README times one
# process per GPU, README.md:94-100.)

def parse_cpulist(text):
    """'0-3,8,10-11' (the kernel's list format) -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(pci_bus_id, sysfs="/sys"):
    """Cores of the NUMA node a PCI device is attached to (None when the platform does not say)."""
    import os
    base = os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower())
    try:
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except OSError:
        return None
    return cpus or None


def busy_cpus(interval=0.05, threshold=0.2, stat="/proc/stat"):
    """Cores that other work keeps busy right now (share of non-idle time over `interval` seconds above `threshold`): a launch thread bound to
    one of them queues behind a stranger. Empty when the platform does not say."""
    import time

    def snap():
        out = {}
        try:
            for ln in open(stat):
                if ln.startswith("cpu") and ln[3:4].isdigit():
                    f = ln.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (sum(v), v[3] + v[4])  # total, idle + iowait
        except OSError:
            pass
        return out

    a = snap()
    time.sleep(interval)
    b = snap()
    busy = set()
    for c, (tot, idle) in b.items():
        if c in a and tot > a[c][0]:
            if 1.0 - (idle - a[c][1]) / (tot - a[c][0]) > threshold:
                busy.add(c)
    return busy


def plan_affinity(local_rank, local_world, threads, allowed, local_cpus_by_rank, avoid=()):
    """Cores for the `threads` launch threads of rank `local_rank`: a slice of the cores next to its GPU that no other rank of the node
    is given. `allowed` is what this process may run on at all; `local_cpus_by_rank[r]` the cores next to rank r's GPU (None = unknown).
    Ranks whose GPUs share a NUMA node split its cores evenly. Returns a list of `threads` cores (repeating when there are fewer cores
    than threads) or None when nothing can be said -- then nobody is pinned. Cores in `avoid` (busy_cpus()) are used only when the
    share has no others."""
    allowed = sorted(set(allowed))
    if not allowed or threads <= 0:
        return None

    def pool_of(r):
        near = local_cpus_by_rank[r] if r < len(local_cpus_by_rank) else None
        return [c for c in allowed if c in set(near)] if near else allowed

    pool = pool_of(local_rank) or allowed
    sharers = sorted({r for r in range(local_world) if (pool_of(r) or allowed) == pool} | {local_rank})
    per = max(1, len(pool) // len(sharers))
    k = sharers.index(local_rank)
    share = pool[k * per:(k + 1) * per] or pool
    idle = [c for c in share if c not in set(avoid)]
    if len(idle) >= min(threads, len(share)) or (idle and len(idle) * 2 >= len(share)):
        share = idle
    # spread over the share (SMT siblings are usually numbered far apart, neighbours are distinct cores)
    return [share[i % len(share)] for i in range(threads)]


def pin_current_thread(cpu):
    """Bind the CALLING thread (Linux: sched_setaffinity(0) is per thread). Returns False when the platform refuses."""
    import os
    try:
        os.sched_setaffinity(0, {cpu})
        return True
    except (AttributeError, OSError):
        return False
