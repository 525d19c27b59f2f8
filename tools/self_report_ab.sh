mkdir -p gpurun_out/r5s
python -m pytest tests/test_gpu_async_run.py tests/test_gpu_fuzz.py tests/test_gpu_small_n_safety.py tests/test_gpu_user_source.py -x -q -m gpu 2>&1 | tail -4 &&
for f in "" "--by-copy" "--sync" "--sync --by-copy"; do python tools/queued_jobs_probe.py $f >> gpurun_out/r5s/q.txt 2>&1 || exit 1; done &&
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r5s/kt -- python3 $GRAFT_REPO_ROOT/tools/queued_jobs_probe.py > /dev/null 2>&1 &&
cd $GRAFT_REPO_ROOT && python tools/queued_jobs_probe.py --read gpurun_out/r5s/kt >> gpurun_out/r5s/q.txt && cat gpurun_out/r5s/q.txt
