cd $GRAFT_REPO_ROOT; O=gpurun_out/r4final3; mkdir -p $O
( time timeout 600 python -m pytest tests -q -m gpu -x ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
( timeout 200 python bench.py --steps 20 --warmup 5 ) > $O/bench20.json 2> $O/bench20.err
tail -c 300 $O/bench20.json
