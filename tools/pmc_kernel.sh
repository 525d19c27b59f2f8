# On the GPU box: one rocprofv3 --pmc pass over the measured child job of a bench configuration; prints, for kernels
# matching PATTERN, the counters' means over the dispatches with the largest grid.
# usage: bash tools/pmc_kernel.sh c3-murray srow "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" [name]
set -o pipefail
C=${1:-c3-murray}; PAT=${2:-srow}; CNT=${3:-"SQ_WAVE_CYCLES SQ_WAIT_ANY"}; K=${4:-pmck}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
N=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.CONFIGS['$C']['n'])")
rm -rf $G/$K
timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace -f csv -d $G/$K -- python3 $R/bench.py --pmc-child --config $C --chains $N > $G/$K.log 2>&1 || { tail -3 $G/$K.log; exit 1; }
F=$(find $G/$K -name '*counter_collection.csv' | head -1)
python3 - "$F" "$PAT" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size"]))
    by[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    by[key]["_dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for key in sorted(by, key=lambda k: -k[1])[:6]:
    print(key, {c: round(sum(v) / len(v), 1) for c, v in by[key].items()}, "n", len(by[key]["_dur_us"]))
PY
