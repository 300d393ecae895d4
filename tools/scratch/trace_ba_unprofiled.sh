cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ba_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ba_trace -- python $GRAFT_REPO_ROOT/tools/ba_only.py > /tmp/ba_only.log 2>&1
tail -2 /tmp/ba_only.log | cut -c1-400
F=$(find /tmp/ba_trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find last k_edge_eval<true> and print the timeline of the following trial
idx = [i for i, r in enumerate(rows) if "k_edge_eval<true>" in r["Kernel_Name"]]
s = idx[-70]; e = idx[-69]
t0 = int(rows[s]["Start_Timestamp"])
prev_end = t0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("dvm::", "")[:28]
    print(f"{(st - t0) / 1e3:9.1f} us  gap {(st - prev_end) / 1e3:6.1f}  dur {(en - st) / 1e3:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))}  {name}")
    prev_end = en
print("iteration span us", (int(rows[e]["Start_Timestamp"]) - t0) / 1e3)
PY
