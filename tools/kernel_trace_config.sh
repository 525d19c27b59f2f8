# On the GPU box: rocprofv3 kernel statistics of one bench configuration, top rows printed.
# usage: bash tools/kernel_trace_config.sh c3-murray [name]
set -o pipefail
C=${1:-c3-murray}; K=${2:-kt_$C}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
rm -rf $G/$K
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d $G/$K -- python3 $R/bench.py --config $C --no-pmc --no-extras --no-cpu-baseline > $G/$K.log 2>&1 || { tail -3 $G/$K.log; exit 1; }
F=$(find $G/$K -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print("%7.3f ms %5.1f %%  %6s calls  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
