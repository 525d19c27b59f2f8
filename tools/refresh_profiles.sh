# On the GPU box: the bench lines, live PMC passes and rocprofv3 kernel stats that tools/collect_profiles.py turns into
# profiles/<round>_*.  usage: bash tools/refresh_profiles.sh r02
set -o pipefail
RND=${1:-r05}
CONFIGS=${CONFIGS:-"c3 c2 c5 c3-murray c3-rosen2fixed"}  # (a gpurun call is at most 20 minutes: CONFIGS="c3 c2" bash tools/refresh_profiles.sh r05, then the rest)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
G=$R/gpurun_out
mkdir -p $G
for C in $CONFIGS; do
  case $C in c3-murray) K=c3m;; c3-rosen2fixed) K=c3r2f;; *) K=$C;; esac
  rm -rf $G/${RND}_pmc_$C $G/${RND}_kt_$K
  ( cd $R && timeout -k 10 400 python3 bench.py --config $C --keep-pmc $G/${RND}_pmc_$C > $G/${RND}_bench_$C.json 2> $G/${RND}_bench_$C.err ) || { echo "bench $C failed"; tail -3 $G/${RND}_bench_$C.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d $G/${RND}_kt_$K -- python3 $R/bench.py --config $C --no-pmc --no-extras --no-cpu-baseline > $G/${RND}_kt_$K.log 2>&1 || { echo "kernel trace $C failed"; tail -3 $G/${RND}_kt_$K.log; exit 1; }
  echo "$C done"
done
case " $CONFIGS " in *" c3 "*) ;; *) exit 0;; esac
( cd $R && timeout -k 10 300 python3 bench.py --config c3 --chains 8192 --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $G/${RND}_bench_c3_8192chains.json 2> $G/${RND}_bench_c3_8192chains.err ) && echo "8192 done"
