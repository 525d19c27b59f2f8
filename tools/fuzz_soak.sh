#!/bin/bash
# Soak of tests/test_gpu_fuzz.py with other seeds than the fixed regression set (400 single-shard + 24 multi-shard cases per
# seed, every option of the engine drawn at random, bit-exact against the oracle).  usage: tools/fuzz_soak.sh seed [seed ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for S in "$@"; do
  MCX_FUZZ_SEED=$S timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2 | sed "s/^/seed $S: /"
done
