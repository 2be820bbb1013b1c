cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p2
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/p2 -o t -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py > $GRAFT_REPO_ROOT/gpurun_out/train_under_rocprof.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/p2/t_results.db $GRAFT_REPO_ROOT/gpurun_out/train_kernel_stats.csv > /dev/null
head -12 $GRAFT_REPO_ROOT/gpurun_out/train_kernel_stats.csv | cut -c1-70,200-330
