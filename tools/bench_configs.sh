set -o pipefail
# The bench / timing runs behind profiles/<tag>_configs/ (default tag r03): C5, the strong-scaling shards of C3 measured one rank's
# share at a time on one GPU, the drop-in call, C4 at full size with one warm-up pass (the one-time allocation of its 42 GB of
# routing ring buffers stays outside the timed pass).   Usage: bash tools/bench_configs.sh [tag]   (on the GPU box, via gpurun)
TAG=${1:-r03}
O=gpurun_out/${TAG}_cfg
mkdir -p $O
timeout -k 10 300 python bench.py --config c5 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
for n in 12500 25000 50000; do timeout -k 10 300 python bench.py --members $n --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3_shard_$n.json 2> $O/shard_$n.err; done
timeout -k 10 200 python tools/time_dropin.py > $O/dropin.log 2>&1; cat $O/dropin.log
[ -n "$SKIP_C4" ] || timeout -k 10 900 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.1f value %.4g kernel %.1f simt %.3f rhs/cd %.1f lanes/member %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['fp64_valu']['simt_efficiency'], d['fp64_valu']['rhs_evals_per_catchment_day'], d['occupancy']['lanes_per_member']), d.get('parity', {}).get('timed_run_sample', {}).get('max_rel_err_vs_oracle'))
    except Exception as ex:
        print(f, 'failed', ex)
PY
