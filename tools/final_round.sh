mkdir -p gpurun_out/r02_fin
if [ "$1" != notests ]; then python -m pytest tests -m gpu -q > gpurun_out/r02_fin/gputests.log 2>&1; tail -3 gpurun_out/r02_fin/gputests.log; fi
bash tools/profile_bench.sh r02_knee > gpurun_out/r02_fin/profile_c3.log 2>&1; tail -1 gpurun_out/r02_fin/profile_c3.log
bash tools/profile_bench.sh r02_knee_c2 --config c2 > gpurun_out/r02_fin/profile_c2.log 2>&1; tail -1 gpurun_out/r02_fin/profile_c2.log
timeout -k 10 300 python bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/r02_fin/bench_c5.json 2> gpurun_out/r02_fin/bench_c5.err
python - <<'PY'
import json
for f in ('gpurun_out/r02_knee/bench.json', 'gpurun_out/r02_knee_c2/bench.json', 'gpurun_out/r02_fin/bench_c5.json'):
    d = json.load(open(f))
    print(f, 'ms/step %.1f value %.4g dev-resident %s kernel %.1f simt %.3f rhs/cd %.1f d2h %s' % (d['ms_per_step'], d['value'], d['value_device_resident'], d['roofline']['kernel_ms'], d['fp64_valu']['simt_efficiency'], d['fp64_valu']['rhs_evals_per_catchment_day'], d['transfer']['d2h_gbs_over_run']), d.get('parity', {}).get('timed_run_sample', {}).get('max_rel_err_vs_oracle'))
PY
