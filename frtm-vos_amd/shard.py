"""Multi-GPU sharding of independent video sequences: one process per GPU, static shard, no collective on
the data path.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the closing barrier and the
reduction of (frames, max wall time); every rank also leaves a ``rank_<r>.json`` report, so a crashed rank can be re-run on
its own and the aggregate can be formed from the files alone."""
import json
import os

import torch


def shard_indices(n, rank, world_size, costs=None):
    """Indices of the sequences rank ``rank`` processes.  Without ``costs``: round-robin ``range(rank, n, world_size)``.
    With ``costs`` (one number per sequence, e.g. frames x objects): longest-first greedy bin packing, deterministic on every
    rank (ties -> lower index, lower rank), each rank's indices returned in ascending order."""
    if costs is None:
        return list(range(rank, n, world_size))
    assert len(costs) == n
    load = [0.0] * world_size
    mine = []
    for i in sorted(range(n), key=lambda j: (-float(costs[j]), j)):
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += float(costs[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def shard_sequences(sequences, rank, world_size, costs=None):
    """Static shard of an in-memory list (tests, synthetic data).  Datasets are sharded lazily through shard_indices."""
    seqs = list(sequences)
    return [seqs[i] for i in shard_indices(len(seqs), rank, world_size, costs)]


def write_rank_report(out_dir, rank, world_size, report):
    """``<out_dir>/rank_<r>.json``: frames, seconds, fps (+ whatever per-stage numbers the caller adds)."""
    os.makedirs(str(out_dir), exist_ok=True)
    path = os.path.join(str(out_dir), 'rank_%d.json' % rank)
    with open(path, 'w') as f:
        json.dump(dict(report, rank=rank, world_size=world_size), f, indent=1, sort_keys=True)
    return path


def aggregate_reports(out_dir, world_size):
    """Whole-job numbers from the per-rank files alone (no process group needed): sum of frames / max seconds."""
    reps = [json.load(open(os.path.join(str(out_dir), 'rank_%d.json' % r))) for r in range(world_size)]
    frames, seconds = sum(r['frames'] for r in reps), max(r['seconds'] for r in reps)
    return frames / seconds, frames, seconds


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
