#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats + HBM counters (separate --pmc passes).
# Usage: tools/profile_bench.sh <tag>     outputs under gpurun_out/<tag>/
set -e -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2> $OUT/pmc_sq.err || true
find $OUT -name "*.csv" | head -50
# keep only the small summaries (traces of torch's own kernels can be large)
for f in $(find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
  python3 - "$f" <<'PY'
import sys, csv
p = sys.argv[1]
rows = list(csv.reader(open(p)))
keep = [rows[0]] + [r for r in rows[1:] if any('simplyp' in c for c in r)]
if 'kernel_stats' in p:
    keep = rows[:12]
open(p + '.summary.csv', 'w', newline='').write('\n'.join(','.join('"%s"' % c for c in r) for r in keep) + '\n')
PY
done
find $OUT -name "*.csv" ! -name "*.summary.csv" -size +200k -delete
ls -la $OUT $OUT/*/* | head -60
