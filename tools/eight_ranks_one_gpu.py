"""Multi-GPU dress rehearsal on ONE GPU (round-3 VERDICT "Next" #6): `bench.py --gpus 8 --share-gpu --sequences 24` -- eight real ranks
(torch.distributed.run, gloo for the barrier / reductions because they share a device), each with its own tracker on cuda:0, its 1/8
share of the GPU's host cores, its own pageable-memory prefetcher.  Checks and records:
  * every rank wrote its report; the shares are disjoint and cover the dataset;
  * aggregate of the line == sum of the ranks' frames / max rank wall (recomputed from the files alone);
  * host enqueue time per rank under 8-way contention, torch threads, the CPUs each rank was pinned to;
  * the NUMA facts the box's sysfs reports for the GPU (the cut the pinning is made from);
  * device mallocs inside the timed region per rank (the share is run once untimed first, --shard-warm: the timed pass is the steady state).
Writes profiles/r04_eight_ranks_one_gpu.json.   python tools/eight_ranks_one_gpu.py [--gpus 8] [--sequences 24] [--out ...]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=8)
    ap.add_argument('--sequences', type=int, default=24)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r04_eight_ranks_one_gpu.json'))
    ap.add_argument('--extra', default='', help='extra bench.py flags (quoted)')
    args = ap.parse_args()
    rdir = os.path.join(ROOT, 'gpurun_out', 'ranks%d' % args.gpus)
    os.makedirs(rdir, exist_ok=True)
    for f in os.listdir(rdir):
        os.remove(os.path.join(rdir, f))
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(args.gpus), '--share-gpu', '--dist-backend', 'gloo',
           '--sequences', str(args.sequences), '--warmup', '5', '--steps', '20', '--no-cpu-baseline', '--no-cg-roofline', '--no-init-sweep',
           '--no-dataset-sim', '--shard-warm', '--report-dir', rdir] + args.extra.split()
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    wall = time.time() - t0
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{'):
            line = json.loads(ln)
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stderr[-4000:])
        sys.exit('bench.py failed (rc %d)' % p.returncode)
    reps = [json.load(open(os.path.join(rdir, 'rank_%d.json' % r))) for r in range(args.gpus)]
    frames, seconds = sum(r['frames'] for r in reps), max(r['seconds'] for r in reps)
    ids = sorted(i for r in reps for i in (r.get('sequence_ids') or []))
    out = {
        'command': ' '.join(cmd[1:]), 'wall_s_incl_process_start': round(wall, 1),
        'line_value_fps': line['value'], 'line_frames_total': line.get('frames_total'), 'scaling': line['scaling'],
        'recomputed_from_rank_files_fps': frames / seconds, 'frames_from_rank_files': frames, 'max_rank_wall_s': seconds,
        'aggregate_matches_line': abs(frames / seconds - line['value']) / line['value'] < 0.02,
        'shares_disjoint_and_covering': ids == list(range(args.sequences)),
        'valid': line['valid'],
        'gpu_numa_sysfs': reps[0].get('gpu_numa_sysfs'),
        'ranks': [{k: r.get(k) for k in ('rank', 'frames', 'seconds', 'fps', 'host_enqueue_ms', 'host_cpus', 'torch_threads',
                                         'device_mallocs_in_timed_region', 'memory_inserts', 'memory_inserts_scheduled', 'cg_solves',
                                         'cg_solves_scheduled', 'all_finite', 'sequence_ids')} for r in reps],
        'note': 'eight ranks time-share ONE MI355X (persistent CG launches off: ranks on one GPU would starve each other\'s resident grids); '
                'the aggregate is therefore about one GPU\'s throughput, not a scaling figure -- what this rehearses is the machinery: launch, '
                'pinning, sharding, per-rank reports, reductions, 8 prefetchers and 8 enqueue threads on the host at once',
    }
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
