cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in "X=1" "DVM_FAST_NW1=1"; do
  rm -rf /tmp/pf; env $e rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pf -- python $R/bench.py --steps 1 --warmup 1 --chunks-per-step 2 --no-ba --no-pcie --no-exclusive --cpu-seconds 0 > /tmp/pf.log 2>&1
  python - "$(find /tmp/pf -name '*counter_collection.csv' | head -1)" "$e" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_fast_cells' in r['Kernel_Name']]
acc=collections.defaultdict(list)
for r in rows: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v)/len(v)/1e6,1) for k,v in acc.items()})
PY
done
