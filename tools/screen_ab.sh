#!/bin/bash
# The per-pair screen's kernel with two LDS buffers and one barrier per block (default) against one buffer and two barriers
# (MCX_SCREEN_ONE_BUFFER=1, rounds 3-4): job time, time in the sweeps and in the screens, pairs left -- C3-murray and C5's
# per-GPU shape, two rounds each on one box (the jobs run with MCX_OPT_PROFILE: slower than the bench line's).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for C in c3-murray c5; do
  for M in 0 1 0 1; do
    MCX_SCREEN_ONE_BUFFER=$M python3 bench.py --config $C --no-pmc --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=o['murray_roofline']
print('$C one LDS buffer: $M  job %.3f ms  sweeps %.3f ms  screens %.3f ms in %d launches  pairs left %.4f  genRemote %.3f ms' % (o['ms_per_step'], m['total_ms'], m['screen']['total_ms'], m['screen']['launches'], m['pairs_evaluated_frac'], m['whole_genremote_ms']))"
  done
done
