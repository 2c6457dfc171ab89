"""HBM-side bytes per launch of the CG kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/cg_trace.py.
    python tools/cg_traffic_summary.py gpurun_out/pmc_cg_fetch gpurun_out/pmc_cg_write profiles/r02_cg_traffic.json
Units as in tools/pmc_summary.py (KiB; FETCH_SIZE doubled on gfx950)."""
import collections, csv, glob, json, sys


def load(d, counter):
    per = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(glob.glob(d + '/*counter_collection.csv')[0])):
        if r['Counter_Name'] == counter:
            per[r['Kernel_Name']][0] += float(r['Counter_Value'])
            per[r['Kernel_Name']][1].add(r['Dispatch_Id'])
    return {k: (v[0], len(v[1])) for k, v in per.items()}


f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- python tools/cg_trace.py   (GaussNewtonCG.run((10,)), N = 80, 480p, c = 96; '
                  'algorithmic bytes per run: 1.15 GB this formulation, 2.67 GB reference formulation; the sample features are 49.8 MB)',
       'units': 'bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024', 'kernels': {}}
keys = ('k_cg_run_persistent', 'k_filter_scores_rows', 'k_filter_wgrad', 'k_stencil', 'k_cg_step_small', 'k_vec_reduce_slabs')
for k in sorted(set(f) | set(w)):
    if any(t in k for t in keys):
        fv, nf = f.get(k, (0.0, 0)); wv, nw = w.get(k, (0.0, 0))
        out['kernels'][k[:60]] = {'launches': max(nf, nw), 'fetch_bytes_per_launch': round(2 * fv * 1024 / max(nf, 1)), 'write_bytes_per_launch': round(wv * 1024 / max(nw, 1))}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out['kernels'], indent=0))
