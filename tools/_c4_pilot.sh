for env in "" "SIMPLYP_PILOT_ALL_REACHES=1"; do
  echo "== $env"
  env $env python bench.py --config c4 --members 2560 --days 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stream 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ms', j['ms_per_step'], 'kernel', j['roofline']['kernel_ms'], 'pilot', j['roofline']['pilot_ms'], 'rhs/cd', j['fp64_valu']['rhs_evals_per_catchment_day'], 'simt', j['fp64_valu']['simt_efficiency'])"
done
