cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ex_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ex_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --chunks-per-step 8 --no-ba --no-pcie --no-exclusive --cpu-seconds 0 > /tmp/ex.log 2>&1
F=$(find /tmp/ex_trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dvm::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 60% of the trace (timed region)
t_all0, t_all1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
t0 = t_all0 + int(0.5 * (t_all1 - t_all0))
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
span = iv[-1][1] - iv[0][0]
# union coverage and concurrency histogram
ev = []
for s, e in iv: ev += [(s, 1), (e, -1)]
ev.sort()
cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("span ms", span / 1e6, "kernels", len(sel))
for k in sorted(hist): print("concurrency", k, "frac", round(hist[k] / span, 3))
dur = collections.defaultdict(float)
for r in sel: dur[r["Kernel_Name"].split("(")[0].replace("dvm::", "")[:24]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda x: -x[1]): print(k.ljust(26), round(v / span, 3))
PY
