# rocprofv3 kernel stats of the timed region only (no BA / PCIe / exclusive legs): averages comparable with bench.py's HIP events
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_l
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_l -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --cpu-seconds 0 --no-ba --no-pcie --no-exclusive > $GRAFT_REPO_ROOT/gpurun_out/prof_l.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_l.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof_l -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-120
