# LDS and VALU counters of the Murray sweep kernels: two rocprofv3 --pmc passes over `bench.py --pmc-child --config c5`
# (usage on the GPU box: bash tools/pmc_sweep.sh; prints per-kernel sums, raw output under gpurun_out/pmc_sw/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sw && mkdir -p $R/gpurun_out/pmc_sw
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -f csv -d $R/gpurun_out/pmc_sw/a -- python3 $R/bench.py --pmc-child --config c5 --chains 32768 > $R/gpurun_out/pmc_sw/a.log 2>&1 && timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/gpurun_out/pmc_sw/b -- python3 $R/bench.py --pmc-child --config c5 --chains 32768 > $R/gpurun_out/pmc_sw/b.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/pmc_sw/*/**/*counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'sweep' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:45],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        print(k, 'n=%d sum=%.4g mean=%.4g'%(len(v),sum(v),sum(v)/len(v)))
PY
