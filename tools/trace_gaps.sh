# kernel timeline of the second job of `bench.py --pmc-child --config $1`: where the stream sits idle
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-c5}
rm -rf $R/gpurun_out/trace_$CFG && mkdir -p $R/gpurun_out/trace_$CFG
timeout -k 10 200 rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/trace_$CFG -- python3 $R/bench.py --pmc-child --config $CFG > $R/gpurun_out/trace_$CFG/log.txt 2>&1
python3 - "$R/gpurun_out/trace_$CFG" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]
t0=int(rows[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in rows)
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
print('second half: %d kernels, span %.2f ms, busy %.2f ms'%(len(rows),(t1-t0)/1e6,busy/1e6))
gaps=collections.defaultdict(lambda:[0,0])
prev=None
for r in rows:
    if prev is not None:
        g=int(r['Start_Timestamp'])-int(prev['End_Timestamp'])
        k=(prev['Kernel_Name'][:38],r['Kernel_Name'][:38])
        gaps[k][0]+=g; gaps[k][1]+=1
    prev=r
for k,(g,n) in sorted(gaps.items(),key=lambda kv:-kv[1][0])[:14]:
    print('%8.1f us total %4d x %7.1f us  %s -> %s'%(g/1e3,n,g/1e3/n,k[0],k[1]))
PY
