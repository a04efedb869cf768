#!/bin/bash
# Streamed pass of the C3 bench against the copy granularity (opts.time_chunk_days), on whatever box this lands on.
# (Round 2 also varied the number of chunks per copy, SIMPLYP_COPY_GROUP: it changed nothing and the switch was removed in round 3.)
# Now and then a pass streams at 45-50 instead of 56 GB/s (profiles/r02_experiments.md): `quick` stops after
# the first measurement when that one is fast.     Usage: bash tools/probe_chunks.sh [quick]
mkdir -p gpurun_out/r02s
run() {
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --chunk-days $1 > gpurun_out/r02s/bench_$1.json 2> gpurun_out/r02s/bench_$1.err
  python -c "
import json
d=json.load(open('gpurun_out/r02s/bench_$1.json'))
print('chunk_days', $1, 'ms/step %.1f kernel %.1f tail %.1f chunks %d d2h %.1f GB/s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['transfer']['d2h_tail_ms'], d['transfer']['streamed_chunks'], d['transfer']['d2h_gbs_over_run']), 'device-clock GB/s', d['transfer']['stream_gbs_device_clock'])
open('gpurun_out/r02s/last_ms','w').write(str(int(d['ms_per_step'])))"
}
run 64
if [ "$1" = quick ] && [ "$(cat gpurun_out/r02s/last_ms)" -lt 880 ]; then echo "fast box: nothing to learn here"; exit 0; fi
echo "slow spell: plain copies right now"; python tools/probe_d2h_numa.py 8
run 64
for cd in 128 256 512 1024; do run $cd; done
python tools/probe_d2h_numa.py 8
