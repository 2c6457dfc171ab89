mkdir -p gpurun_out/r5final
python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r5final/suite.log 2>&1; echo "suite rc=$? $(grep -v amdgpu gpurun_out/r5final/suite.log | tail -n 1)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/r5final/bench20.json; python -c "
import json; d=json.loads(open('gpurun_out/r5final/bench20.json').read()); print(d['metric'], round(d['value'],1), d['unit'], d['valid'], d['roofline']['frac'], d['roofline'].get('frac_executed'), d['cpu_baseline']['value'])"
