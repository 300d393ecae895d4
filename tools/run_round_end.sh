set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py 2>&1 | tail -1 > gpurun_out/bench_k.json
cat gpurun_out/bench_k.json | cut -c1-600
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_k.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_k -name "*kernel_stats.csv" | head
