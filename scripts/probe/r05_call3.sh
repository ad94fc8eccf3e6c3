set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_new.json 2> gpurun_out/bench_r05_new.err; echo rc $?; tail -c 1500 gpurun_out/bench_r05_new.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r05_new.json').read().strip().splitlines()[-1])
c=d['config']
print('value',d['value'],'ms',d['ms_per_step'],'spread',d['value_spread'])
print('spans',c.get('stream_spans_ms'),'cells',c.get('cells_segmented'),'iters',c.get('prgls_iterations'))
print('roofline',{k:d['roofline'][k] for k in ('achieved','frac','hbm_contract_frac','conv_stack_ms_per_volume','avg_launch_ms')})
print('independent',json.dumps(c.get('independent_matches'))[:600])
print('steady',c.get('steady_state'))
print('chained',json.dumps(c.get('chained'))[:500])
print('other',json.dumps(c.get('other_configs'),indent=0)[:6000])
print('err',c.get('informative_passes_error'))
print('cpu',json.dumps(d['cpu_baseline'])[:1500])
PY
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -5
