"""Condense the per-kernel FETCH_SIZE / WRITE_SIZE summaries of tools/profile_configs_pmc.sh into profiles/<tag>_config{4,5}_pmc_traffic.json.
    python tools/config_pmc_summary.py r03"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
cmds = {4: '--size 720x1280 --objects 3 --late-object 10 --steps 32', 5: '--size 1080x1920 --objects 8 --memory 32 --steps 24'}
for k in (4, 5):
    per = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = os.path.join(ROOT, 'gpurun_out', 'r3', 'config%d_pmc_%s' % (k, c), 'summary.csv')
        for r in csv.DictReader(open(f)):
            e = per.setdefault(r['Kernel_Name'], {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
            e['launches'] = max(e['launches'], int(r['Launches']))
            e[c] = float(r[c + '_KiB_total'])
    out = {'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- python bench.py --no-cpu-baseline --no-cg-roofline '
                      '--no-init-sweep --no-dataset-sim --warmup 2 ' + cmds[k],
           'units': 'whole process (warm-up sequences included); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per the gfx950 note '
                    'of MI355X_MICROARCH.md, WRITE_SIZE as reported', 'kernels': {}}
    tot = 0.0
    for name, e in sorted(per.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE']))[:25]:
        b = (2 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024
        tot += b
        n = max(e['launches'], 1)
        out['kernels'][name] = {'launches': e['launches'], 'fetch_MB_per_launch': round(2 * e['FETCH_SIZE'] * 1024 / n / 1e6, 2),
                                'write_MB_per_launch': round(e['WRITE_SIZE'] * 1024 / n / 1e6, 2), 'total_GB': round(b / 1e9, 2)}
    out['total_GB'] = round(tot / 1e9, 1)
    json.dump(out, open(os.path.join(ROOT, 'profiles', '%s_config%d_pmc_traffic.json' % (tag, k)), 'w'), indent=1)
    print('config %d: %.1f GB over the listed kernels' % (k, out['total_GB']))
