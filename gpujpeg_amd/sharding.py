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
