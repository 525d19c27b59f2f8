#!/bin/bash
# Builds libmcx.so three times, differing only in how k_run_small treats an abandoned tuner meeting
# (MCX_MEET_VARIANT in mcpar_amd/csrc/mcx_persist.hpp: 0 = unbounded wait as in round 2, 1 = round 3's early loop exit,
# 2 = what ships), into build/ab/ -- which travels to the GPU box -- for tools/persist_ab.py to time on ONE box.
set -e
cd "$(dirname "$0")/.."
make lib >/dev/null
mkdir -p build/ab
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -Iinclude"
OBJS=$(ls mcpar_amd/csrc/*.o | grep -v mcx_k_persist.o)
for v in 0 1 2; do
  hipcc $FLAGS -DMCX_MEET_VARIANT=$v -c -o build/ab/persist_v$v.o mcpar_amd/csrc/mcx_k_persist.hip &
done
wait
for v in 0 1 2; do
  hipcc --offload-arch=gfx950 -fPIC -shared -o build/ab/libmcx_v$v.so $OBJS build/ab/persist_v$v.o
  rm -f build/ab/persist_v$v.o
done
ls -la build/ab
