#!/bin/bash
# The per-pair screen's kernel with 1 (default), 2 or 4 row tiles multiplied at a time on as many accumulators
# (MCX_SCREEN_NACC): job time, time in the sweeps and in the screens -- C3-murray and C5's per-GPU shape, two rounds on one box
# (the jobs run with MCX_OPT_PROFILE: slower than the bench line's).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for C in c3-murray c5; do
  for M in 1 2 4 1 2 4; do
    MCX_SCREEN_NACC=$M python3 bench.py --config $C --no-pmc --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=o['murray_roofline']
print('$C accumulators: $M  job %.3f ms  sweeps %.3f ms  screens %.3f ms in %d launches  pairs left %.4f' % (o['ms_per_step'], m['total_ms'], m['screen']['total_ms'], m['screen']['launches'], m['pairs_evaluated_frac']))"
  done
done
