#!/bin/bash
# The sum screens on the K = 2 np kernel (k_screen_gemm_sums, default) against the K + 16 kernel of round 4
# (MCX_SCREEN_SUMS_OLD=1): job time, time in the sweeps and in the screens, pairs left -- C3-murray and C5's per-GPU shape.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for C in c3-murray c5; do
  for M in 0 1 0 1; do
    MCX_SCREEN_SUMS_OLD=$M python3 bench.py --config $C --no-pmc --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=o['murray_roofline']
print('$C K+16 kernel for the sums: $M  job %.3f ms  sweeps %.3f ms  screens %.3f ms in %d launches  pairs left %.4f  genRemote %.3f ms' % (o['ms_per_step'], m['total_ms'], m['screen']['total_ms'], m['screen']['launches'], m['pairs_evaluated_frac'], m['whole_genremote_ms']))"
  done
done
