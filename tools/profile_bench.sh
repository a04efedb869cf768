#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats + HBM and SQ counters (separate --pmc passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage: tools/profile_bench.sh <tag> [extra bench.py flags, e.g. --config c2]     outputs under gpurun_out/<tag>/
set -e -o pipefail
# Environment: STEPS / WARMUP of the plain bench run (default 5 / 1), TRACE_STEPS / TRACE_WARMUP of the traced run (default 4 / 1) --
# a configuration whose pass takes a minute (c4) wants STEPS=1 WARMUP=0 TRACE_STEPS=1 TRACE_WARMUP=0.
TAG=${1:-r02}
shift || true
EXTRA="$@"
STEPS=${STEPS:-5}; WARMUP=${WARMUP:-1}; TRACE_STEPS=${TRACE_STEPS:-4}; TRACE_WARMUP=${TRACE_WARMUP:-1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--no-cpu-baseline --no-parity"
if [ -z "$ONLY_TRACE" ]; then
python3 bench.py --steps $STEPS --warmup $WARMUP $EXTRA > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
fi
# the traced passes leave the table in HBM: under the profiler the streamed chunk copies sometimes run as shader blits that share
# the CUs with the persistent kernel (seen: kernel 932 ms instead of 538, copies keeping pace) -- not what an unprofiled run does
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps $TRACE_STEPS --warmup $TRACE_WARMUP --no-stream $Q $EXTRA > $OUT/bench_trace.json 2> $OUT/trace.err
if [ -z "$ONLY_TRACE" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-stream $Q $EXTRA > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-stream $Q $EXTRA > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-stream $Q $EXTRA > /dev/null 2> $OUT/pmc_sq.err || true
fi
# keep only the small summaries (traces of torch's own kernels can be large)
for f in $(find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
  python3 - "$f" <<'PY'
import sys, csv
p = sys.argv[1]
rows = list(csv.reader(open(p)))
keep = [rows[0]] + [r for r in rows[1:] if any('simplyp' in c for c in r)]
if 'kernel_stats' in p:
    keep = rows[:14]
open(p + '.summary.csv', 'w', newline='').write('\n'.join(','.join('"%s"' % c for c in r) for r in keep) + '\n')
PY
done
find $OUT -name "*.csv" ! -name "*.summary.csv" -size +200k -delete
find $OUT -name "*.summary.csv" | head -20
