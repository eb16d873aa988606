cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_final_driver.json 2> gpurun_out/r4_final_driver.err; echo "driver cmd exit $?"
python -c "import json;j=json.load(open('gpurun_out/r4_final_driver.json'));print('driver:', round(j['ms_per_step']*1e3,2),'us/step; repeats median', j['ms_per_step_median_of_repeats'], 'frac', j['roofline']['frac'], 'frac_rocprof', j['roofline'].get('frac_rocprof'), 'traffic_ratio', j['roofline'].get('traffic_ratio'), 'iwae', j['parity'].get('iwae_bound'), 'cpu', j['cpu_baseline']['value'])"
for c in configs0 configs1 configs2 run_train configs1_h512 configs2_h512; do
  python bench.py --config $c --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r4_final_$c.json 2> gpurun_out/r4_final_$c.err
  python -c "import json;j=json.load(open('gpurun_out/r4_final_$c.json'));print('$c:', round(j['ms_per_step']*1e3,2),'us/step', j['roofline']['schedule'], 'frac', round(j['roofline']['frac'],4), 't/ideal', round(j['roofline']['t_step_over_ideal'],1), 'iwae', (j['parity'].get('iwae_bound') or {}).get('rel_err'))"
done
python bench.py --config configs4_shard --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r4_final_c5.json 2> gpurun_out/r4_final_c5.err
python -c "import json;j=json.load(open('gpurun_out/r4_final_c5.json'));print('c5:', round(j['ms_per_step'],4),'ms/step', j['roofline']['schedule'], 'frac', round(j['roofline']['frac'],4), 'iwae', (j['parity'].get('iwae_bound') or {}))"
