set -u
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/probe/slow_prior.py 2>&1 | grep -v amdgpu | tail -10
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_new2.json 2> gpurun_out/bench_r05_new2.err; echo rc $?; tail -c 600 gpurun_out/bench_r05_new2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r05_new2.json').read().strip().splitlines()[-1])
c=d['config']
print('value',d['value'],'ms',d['ms_per_step'],'spread',{k:d['value_spread'][k] for k in ('min','median','max')})
print('spans',c.get('stream_spans_ms'),'cells',c.get('cells_segmented'),'iters',c.get('prgls_iterations'))
print('roofline',{k:d['roofline'][k] for k in ('achieved','frac','hbm_contract_frac','conv_stack_ms_per_volume','avg_launch_ms')})
print('steady',c.get('steady_state'))
print('slow',c.get('slow_prior'))
print('indep',c['independent_matches']['volumes_per_s'], c['independent_matches'].get('with_discriminating_ffn',{}).get('volumes_per_s'))
print('chained',c['chained']['ms_per_frame'],c['chained']['stage_ms'])
for k,v in c['other_configs'].items(): print(k,{a:b for a,b in v.items() if a in ('volumes_per_s','ms_per_frame','one_frame_at_a_time_ms','prgls_iterations','cells_segmented','match_ms','ms_per_iteration','predictions_per_s','error')})
print('err',c.get('informative_passes_error'))
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['stage_s'])
PY
