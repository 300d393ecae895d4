cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pp -- python $R/tools/ba_only.py > /tmp/pp.log 2>&1
  python - "$(find /tmp/pp -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
try:
    rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_schur' in r['Kernel_Name']]
except Exception as e:
    print('fail', e); sys.exit(0)
acc=collections.defaultdict(list)
for r in rows: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
done
