"""Multi-GPU sharding of independent video sequences: one process per GPU, static
round-robin shard, no collective on the data path.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only for the closing barrier and the reduction of (frames, max wall time)."""
import torch


def shard_sequences(sequences, rank, world_size):
    """Static shard: rank r takes sequences[r::world_size] (longest-first order is the caller's business)."""
    return list(sequences)[rank::world_size]


def aggregate_throughput(frames, seconds, device='cpu'):
    """Whole-job frames/s = sum of frames over ranks / max wall time over ranks.  Works without an
    initialised process group (single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return frames / seconds, frames, seconds
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item() / t.item()), float(f.item()), float(t.item())
