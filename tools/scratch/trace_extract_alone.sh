# which kernels of the extractor pipeline run ALONE on the chip (concurrency 1), and the idle gaps, over the timed region
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ex_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ex_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --chunks-per-step 8 --no-ba --no-pcie --no-exclusive --cpu-seconds 0 > /tmp/ex.log 2>&1
F=$(find /tmp/ex_trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dvm::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_all0, t_all1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
t0 = t_all0 + int(0.35 * (t_all1 - t_all0)); t1 = t_all0 + int(0.75 * (t_all1 - t_all0))
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
ev = []
for i, r in enumerate(sel): ev += [(int(r["Start_Timestamp"]), 1, i), (int(r["End_Timestamp"]), -1, i)]
ev.sort()
active = set(); last = ev[0][0]; alone = collections.Counter(); idle_after = collections.Counter(); span = ev[-1][0] - ev[0][0]; prev_name = None
for t, d, i in ev:
    if len(active) == 1: alone[sel[next(iter(active))]["Kernel_Name"].split("(")[0].replace("dvm::", "")[:22]] += t - last
    if len(active) == 0 and prev_name: idle_after[prev_name] += t - last
    last = t
    if d == 1: active.add(i)
    else:
        active.discard(i); prev_name = sel[i]["Kernel_Name"].split("(")[0].replace("dvm::", "")[:22]
print("span ms", span / 1e6)
print("alone:", {k: round(v / span, 3) for k, v in alone.most_common()})
print("idle after:", {k: round(v / span, 3) for k, v in idle_after.most_common()})
PY
