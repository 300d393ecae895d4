# tools/run_profiles_r06.sh -- round-6 rocprofv3 evidence (run on the GPU box through gpurun; output under gpurun_out/prof_r06/)
#   1. --kernel-trace --stats of the timed region of bench.py (extract + match only)        -> r06_a_extract_kernel_stats.csv
#   2. --kernel-trace --stats of the BA leg alone (tools/ba_only.py)                         -> r06_b_ba_kernel_stats.csv
#   2b. the same of the loop-closed workload (tools/ba_loop_only.py), default and with DVM_BA_BORDER=0 -> r06_b2_* / r06_b3_*
#   2c. the densely coupled 2 000-keyframe map (tools/ba_dense_regime.py)                    -> r06_b4_*
#   3. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters incl. SQ_ACTIVE_INST_VALU) of a short extract run -> r06_pmc_*.csv
#   4. separate --pmc passes (FETCH_SIZE | WRITE_SIZE) of the BA leg (tools/ba_short.py prints its trial count)       -> r06_pmc_ba_*.csv
#   5. the same two counters on known-byte-count kernels of every access width (tools/pmc_calib.hip)                  -> r06_pmc_calib_*.csv
#   6. folded (tools/pmc_traffic.py): calibration per load width, dispatch counts and per-trial totals of the BA kernels, the
#      counter-based VALU busy fraction, and the hash of the kernel sources the counters were taken on (tools/kernels_sha.py) -> r06_pmc_traffic.json
# PMC passes never combine with the hip/hsa/memory trace domains (gpurun refuses that).
set -u
R=$GRAFT_REPO_ROOT
[ -x $R/tools/bin/pmc_calib ] || hipcc -O3 --offload-arch=gfx950 $R/tools/pmc_calib.hip -o $R/tools/bin/pmc_calib
O=$R/gpurun_out/prof_r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FOLD='import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("dvm::" in r["Kernel_Name"] or "calib_" in r["Kernel_Name"])]
keep = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Start_Timestamp", "End_Timestamp"]
keep = [k for k in keep if rows and k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})'
BENCH="python $R/bench.py --steps 3 --warmup 1 --chunks-per-step 8 --no-ba --no-pcie --no-exclusive --no-legs --no-config-legs --cpu-seconds 0"
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- $BENCH > $O/a_bench.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r06_a_extract_kernel_stats.csv
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python $R/tools/ba_only.py > $O/b_ba.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r06_b_ba_kernel_stats.csv
# 2b. the loop-closed second BA workload (kept landmarks + the flow form of the solve: k_chol_flow), and the same with both switched off
rm -rf /tmp/p2b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2b -- python $R/tools/ba_loop_only.py > $O/b2_ba_loop.log 2>&1
cp $(find /tmp/p2b -name "*kernel_stats.csv" | head -1) $O/r06_b2_ba_loop_closed_kernel_stats.csv
rm -rf /tmp/p2c && DVM_BA_BORDER=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2c -- python $R/tools/ba_loop_only.py > $O/b3_ba_loop.log 2>&1
cp $(find /tmp/p2c -name "*kernel_stats.csv" | head -1) $O/r06_b3_ba_loop_closed_noborder_flow_kernel_stats.csv
rm -rf /tmp/p2d && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2d -- python $R/tools/ba_dense_regime.py > $O/b4_ba_dense.log 2>&1
cp $(find /tmp/p2d -name "*kernel_stats.csv" | head -1) $O/r06_b4_ba_dense_2000kf_kernel_stats.csv
grep '^{"keyframes"' $O/b4_ba_dense.log > $O/r06_b4_ba_dense_2000kf.json
SHORT="python $R/bench.py --steps 1 --warmup 1 --chunks-per-step 2 --no-ba --no-pcie --no-exclusive --no-legs --no-config-legs --cpu-seconds 0"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p3 && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p3 -- $SHORT > $O/pmc_$C.log 2>&1
  python -c "$FOLD" "$(find /tmp/p3 -name '*counter_collection.csv' | head -1)" $O/r06_pmc_$C.csv
  rm -rf /tmp/p5 && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p5 -- python $R/tools/ba_short.py > $O/pmc_ba_$C.log 2>&1
  python -c "$FOLD" "$(find /tmp/p5 -name '*counter_collection.csv' | head -1)" $O/r06_pmc_ba_$C.csv
  rm -rf /tmp/p6 && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p6 -- $R/tools/bin/pmc_calib > $O/calib_$C.log 2>&1
  python -c "$FOLD" "$(find /tmp/p6 -name '*counter_collection.csv' | head -1)" $O/r06_pmc_calib_$C.csv
done
rm -rf /tmp/p4 && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p4 -- $SHORT > $O/pmc_SQ.log 2>&1
python -c "$FOLD" "$(find /tmp/p4 -name '*counter_collection.csv' | head -1)" $O/r06_pmc_sq_counters.csv
cp "$(find /tmp/p4 -name '*kernel_trace.csv' | head -1)" $O/r06_pmc_sq_kernel_trace.csv 2>/dev/null
TR=$(grep "^trials" $O/pmc_ba_FETCH_SIZE.log | tail -1 | awk "{print \$2}")
python $R/tools/valu_mix.py $O/r06_valu_mix.json > $O/valu_mix.log 2>&1 || cp $R/profiles/r03_valu_mix.json $O/r06_valu_mix.json
python $R/tools/pmc_traffic.py --fetch $O/r06_pmc_FETCH_SIZE.csv $O/r06_pmc_ba_FETCH_SIZE.csv --write $O/r06_pmc_WRITE_SIZE.csv $O/r06_pmc_ba_WRITE_SIZE.csv \
   --sq $O/r06_pmc_sq_counters.csv --sq-trace $O/r06_pmc_sq_kernel_trace.csv --calib-fetch $O/r06_pmc_calib_FETCH_SIZE.csv --calib-write $O/r06_pmc_calib_WRITE_SIZE.csv \
   --calib-bytes "$(tail -1 $O/calib_FETCH_SIZE.log)" --mix $O/r06_valu_mix.json --batch 256 --kernels-sha "$(python $R/tools/kernels_sha.py)" \
   --ba-fetch $O/r06_pmc_ba_FETCH_SIZE.csv --ba-trials ${TR:-0} > $O/r06_pmc_traffic.json
ls -la $O; head -6 $O/r06_a_extract_kernel_stats.csv | cut -c1-160; head -14 $O/r06_b_ba_kernel_stats.csv | cut -c1-160; head -c 1500 $O/r06_pmc_traffic.json
