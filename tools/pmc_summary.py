"""Condense two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately as MI355X_MICROARCH.md prescribes)
into profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch for the conv kernel family.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json "<command>"

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is used as reported (uncalibrated)."""
import collections
import csv
import glob
import json
import sys

FAMILY = ('k_conv_igemm', 'k_conv3x3_halo', 'k_conv3x3_wino', 'k_splitk_epilogue', 'k_wino4_', 'k_wino6_')


def load(d, counter):
    f = glob.glob(d + '/*counter_collection.csv')[0]
    per = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        per[k][0] += float(r['Counter_Value'])
        per[k][1].add(r['Dispatch_Id'])
    return {k: (v[0], len(v[1])) for k, v in per.items()}


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {'command': sys.argv[4] if len(sys.argv) > 4 else '', 'units': 'bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024',
           'kernels': {}}
    fam_f = fam_w = 0.0
    fam_n = 0
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in FAMILY):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        n = max(nf, nw)
        out['kernels'][k[:80]] = {'launches': n, 'fetch_bytes_per_launch': 2 * f * 1024 / max(nf, 1),
                                  'write_bytes_per_launch': w * 1024 / max(nw, 1)}
        if 'k_splitk_epilogue' not in k and 'k_wino4_' not in k and 'k_wino6_' not in k:      # (a launch = one conv: split-K epilogues and the two transform
            fam_n += n                                                # kernels of a Winograd F(4x4,3x3) conv belong to their GEMM launch)
        fam_f += 2 * f * 1024
        fam_w += w * 1024
    out['conv_family'] = {'launches': fam_n, 'bytes_per_launch': (fam_f + fam_w) / max(fam_n, 1),
                          'fetch_bytes_per_launch': fam_f / max(fam_n, 1), 'write_bytes_per_launch': fam_w / max(fam_n, 1)}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out['conv_family']))


if __name__ == '__main__':
    main()
