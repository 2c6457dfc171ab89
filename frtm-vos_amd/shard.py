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


def aggregate_throughput(frames, seconds, device='cpu', group=None):
    """Whole-job frames/s = sum of frames over ranks / max wall time over ranks.  Works without an
    initialised process group (single process).  ``group``: the group init_process_groups chose (None = the default group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return frames / seconds, frames, seconds
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(f.item() / t.item()), float(f.item()), float(t.item())


def init_process_groups(backend, world, dev=None, nccl_timeout_s=180):
    """Process group of an N > 1 run (the data path has no collective; the group only carries the closing barrier and the
    max-reduce of the wall time).  The CONTROL group is always gloo (CPU tensors: it cannot fail on the GPU side).  With backend 'nccl' an RCCL
    group is tried on top of it -- every rank sees ONE device (HIP_VISIBLE_DEVICES), a configuration RCCL has never met on this code -- and
    the ranks agree over gloo whether it works (a one-element all-reduce must count every rank): if any rank failed, ALL of them fall back to
    gloo for the barrier / reduce instead of losing the whole run at start-up.  Returns (group or None, backend used, ranks RCCL counted,
    device of the reduction tensors, error text or None)."""
    import datetime
    import sys
    import torch.distributed as dist
    dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=60))
    if backend != 'nccl':
        return None, backend, None, 'cpu', None
    ok, seen, err, grp = 1.0, 0, None, None
    # a collective that never completes must RAISE here (caught below -> gloo) instead of having the watchdog thread abort the process
    os.environ.setdefault('TORCH_NCCL_BLOCKING_WAIT', '1')
    os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '0')
    try:
        grp = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=nccl_timeout_s))
        t = torch.ones(1, device=dev)
        dist.all_reduce(t, group=grp)
        torch.cuda.synchronize()
        seen = int(round(float(t.item())))
        if seen != world:
            ok, err = 0.0, 'RCCL all-reduce counted %d of %d ranks' % (seen, world)
    except Exception as ex:      # noqa: BLE001  (whatever RCCL / the runtime raises under the one-device-per-rank isolation)
        ok, err = 0.0, '%s: %s' % (type(ex).__name__, str(ex)[:300])
    flag = torch.tensor([ok], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                    # over gloo: every rank takes the same decision
    if float(flag.item()) > 0:
        return grp, 'nccl', seen, dev, None
    if err:
        print('frtm shard rank %s: RCCL group unusable (%s): barrier / max-reduce over gloo' % (os.environ.get('RANK', '?'), err), file=sys.stderr, flush=True)
    return None, 'gloo', seen, 'cpu', err or 'another rank failed to bring RCCL up'


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def host_cores_near_gpu(pci_bus_id, pci_device_id=0, pci_domain_id=0, sysfs='/sys/bus/pci/devices'):
    """The block of host cores a rank should run on: the cores of the GPU's NUMA node, cut evenly among the GPUs (amdgpu PCI
    functions) attached to that node, this GPU's share by its position in bus order.  Returns a sorted list of CPU ids, or [] when
    sysfs has no answer (then nothing is pinned).  An 8-GPU MI355X node: two sockets of 64 cores + SMT siblings, four GPUs each ->
    32 logical CPUs per rank, all local to its GPU."""
    bdf = '%04x:%02x:%02x.0' % (pci_domain_id, pci_bus_id, pci_device_id)
    dev = os.path.join(sysfs, bdf)
    try:
        node = int(open(os.path.join(dev, 'numa_node')).read())
        cpus = _parse_cpulist(open(os.path.join(dev, 'local_cpulist')).read())
    except (OSError, ValueError):
        return []
    if not cpus:
        return []
    peers = []
    try:
        for name in sorted(os.listdir(sysfs)):
            p = os.path.join(sysfs, name)
            try:
                if os.path.basename(os.path.realpath(os.path.join(p, 'driver'))) != 'amdgpu':
                    continue
                if int(open(os.path.join(p, 'numa_node')).read()) == node and open(os.path.join(p, 'class')).read().startswith(('0x03', '0x12')):
                    peers.append(name)
            except (OSError, ValueError):
                continue
    except OSError:
        pass
    if bdf not in peers:
        peers = sorted(set(peers + [bdf]))
    k, n = peers.index(bdf), len(peers)
    # physical cores first, their SMT siblings second (the kernel lists them as two ranges): cut each range into n shares
    half = len(cpus) // 2 if len(cpus) % 2 == 0 and cpus[len(cpus) // 2] - cpus[len(cpus) // 2 - 1] > 1 else len(cpus)
    out = []
    for lo in range(0, len(cpus), half):
        rng = cpus[lo:lo + half]
        per = max(1, len(rng) // n)
        out.extend(rng[k * per:(k + 1) * per])
    return sorted(out)


def rank_share_of(cpus, k, n):
    """The k-th of n even shares of a node's CPU list (physical cores and their SMT siblings -- the kernel lists them as two ranges -- are
    cut separately, like host_cores_near_gpu does among the GPUs of a node)."""
    cpus = sorted(cpus)
    half = len(cpus) // 2 if len(cpus) % 2 == 0 and len(cpus) >= 2 and cpus[len(cpus) // 2] - cpus[len(cpus) // 2 - 1] > 1 else len(cpus)
    out = []
    for lo in range(0, len(cpus), max(half, 1)):
        rng = cpus[lo:lo + half]
        per = max(1, len(rng) // n)
        out.extend(rng[(k % n) * per:(k % n + 1) * per])
    return sorted(out)


def describe_gpu_numa(device=0, sysfs='/sys/bus/pci/devices'):
    """What the box's sysfs says about this GPU's place among the host cores (printed by the multi-rank dress rehearsal): PCI address,
    NUMA node, local CPU list, amdgpu functions on that node, and the cut host_cores_near_gpu makes of it."""
    if not torch.cuda.is_available():
        return {}
    p = torch.cuda.get_device_properties(device)
    bdf = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), getattr(p, 'pci_bus_id', 0), getattr(p, 'pci_device_id', 0))
    d = os.path.join(sysfs, bdf)
    out = {'pci': bdf}
    for key in ('numa_node', 'local_cpulist'):
        try:
            out[key] = open(os.path.join(d, key)).read().strip()
        except OSError:
            out[key] = None
    cut = host_cores_near_gpu(getattr(p, 'pci_bus_id', 0), getattr(p, 'pci_device_id', 0), getattr(p, 'pci_domain_id', 0), sysfs)
    out['cut_for_this_gpu'] = '%d-%d (%d)' % (min(cut), max(cut), len(cut)) if cut else None
    out['host_logical_cpus'] = os.cpu_count()
    return out


def pin_host_threads_near_gpu(device=0, share=None):
    """Confines the calling thread (and every thread it starts afterwards) to host cores on the NUMA node of CUDA/HIP device ``device``
    (host_cores_near_gpu).  The thread that enqueues a sequence's launches is latency-critical: measured on a two-socket MI355X node,
    20-frame sequences ran at 432-439 frames/s pinned next to the GPU and at 338-438 frames/s left to the scheduler (it migrates the
    thread across sockets).  ``share=(k, n)``: only the k-th of n even shares of those cores (ranks sharing one GPU in the multi-rank
    dress rehearsal: the host-side contention of n pinned ranks without n GPUs).  Returns the CPU list (empty: nothing done)."""
    if not hasattr(os, 'sched_setaffinity') or not torch.cuda.is_available():
        return []
    p = torch.cuda.get_device_properties(device)
    if not hasattr(p, 'pci_bus_id'):
        return []
    cpus = host_cores_near_gpu(p.pci_bus_id, getattr(p, 'pci_device_id', 0), getattr(p, 'pci_domain_id', 0))
    if share is not None:                 # dress rehearsal of N ranks on ONE GPU: rank k takes the k-th of n shares of this GPU's cores
        cpus = rank_share_of(cpus, share[0], share[1])
    allowed = os.sched_getaffinity(0)
    cpus = [c for c in cpus if c in allowed]
    if len(cpus) >= 2:
        os.sched_setaffinity(0, cpus)
        # torch's intra-op (OpenMP) pool is sized for the whole machine and its workers inherit this mask when they are started: 256
        # spinning threads on 32 CPUs turned the host-side generation of a synthetic sequence from milliseconds into minutes.  Size
        # the pool for the cores the rank owns (one thread per physical core).
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus) // 2)))
        return cpus
    return []
