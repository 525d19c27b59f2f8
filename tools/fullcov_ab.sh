#!/bin/bash
# Full-covariance kernels A/B on one box (tools/fullcov_probe.py, 65 536 chains, 500 + 1000 steps):
#   blocks per lane (MCX_OPT_BLOCKS_PER_LANE): 1 = k_fused_fast<LPC, .., FULL> (one block per lane; up to 16-D the lane's rows
#   of the factor in registers), 2 = k_fused_fastb<LPC/2, 2, .., FULL> (two mirrored blocks per lane: mcx_fastb.hpp), 0 = the
#   engine's choice; and the library rebuilt with -DMCX_FULL_T_REGS=0 (the factor in LDS at every size, as in rounds 2-4).
# usage: tools/fullcov_ab.sh   (from the repository root; needs hipcc)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/mcpar_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function -I$ROOT/include"
mkdir -p /tmp/fullcov_ab
hipcc $FLAGS -DMCX_FULL_T_REGS=0 -c -o /tmp/fullcov_ab/mcx_k_fast_full.o $CS/mcx_k_fast_full.hip
OBJS=$(ls $CS/*.o | grep -v mcx_k_fast_full.o)
hipcc --offload-arch=gfx950 -fPIC -shared -o /tmp/fullcov_ab/libmcx_tlds.so $OBJS /tmp/fullcov_ab/mcx_k_fast_full.o
for bpl in 0 1 2; do
  echo "== blocks per lane $bpl (0 = engine's choice)"
  MCX_PROBE_BPL=$bpl python3 $ROOT/tools/fullcov_probe.py 2>&1 | grep -E "again" 
done
echo "== one block per lane, factor in LDS at every size (-DMCX_FULL_T_REGS=0)"
MCX_PROBE_BPL=1 MCX_LIBMCX=/tmp/fullcov_ab/libmcx_tlds.so python3 $ROOT/tools/fullcov_probe.py 2>&1 | grep -E "again"
