set -o pipefail
mkdir -p gpurun_out/r02_cfg
# (C2 is profiled by tools/final_round.sh)
timeout -k 10 300 python bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/r02_cfg/bench_c5.json 2> gpurun_out/r02_cfg/bench_c5.err
for n in 12500 25000 50000; do timeout -k 10 300 python bench.py --members $n --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_cfg/bench_c3_shard_$n.json 2> gpurun_out/r02_cfg/shard_$n.err; done
timeout -k 10 200 python tools/time_dropin.py > gpurun_out/r02_cfg/dropin.log 2>&1; cat gpurun_out/r02_cfg/dropin.log
timeout -k 10 600 python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02_cfg/bench_c4.json 2> gpurun_out/r02_cfg/bench_c4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02_cfg/bench_*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.1f value %.4g kernel %.1f simt %.3f rhs/cd %.1f lanes/member %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['fp64_valu']['simt_efficiency'], d['fp64_valu']['rhs_evals_per_catchment_day'], d['occupancy']['lanes_per_member']), d.get('parity', {}).get('timed_run_sample', {}).get('max_rel_err_vs_oracle'))
    except Exception as ex:
        print(f, 'failed', ex)
PY
