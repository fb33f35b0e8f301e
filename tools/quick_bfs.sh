# quick BFS check on the GPU box: parity tests + bench line + per-level profile
python -m pytest tests/test_bfs_gpu.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_q.log 2>&1
tail -1 gpurun_out/bench_q.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('GTEPS', d['value']/1e3, 'ms/step', d['ms_per_step'], 'enact', d['config']['enact_ms_last'])
r=d['roofline']; print('BU frac', r['frac'], [[round(x*1e3,1) if isinstance(x,float) else x for x in l] for l in r['all_levels']])
r=d['roofline_topdown_advance']; print('TD frac', r['frac'], [[round(x*1e3,1) if isinstance(x,float) else x for x in l] for l in r['levels']])
"
