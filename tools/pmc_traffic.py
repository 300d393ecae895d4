#!/usr/bin/env python3
"""tools/pmc_traffic.py FETCH.csv WRITE.csv BATCH [SQ.csv] > profiles/rNN_pmc_traffic.json

Folds two rocprofv3 counter_collection CSVs (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same
bench.py command) into per-kernel HBM bytes per launch.  Counter unit: KiB per dispatch (MI355X_MICROARCH.md,
HBM / rocprofv3 section).  Template arguments are stripped from the kernel names."""
import collections
import csv
import json
import re
import sys


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"[<(].*$", "", name)
        acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = mean_by_kernel(sys.argv[1], "FETCH_SIZE")
    write = mean_by_kernel(sys.argv[2], "WRITE_SIZE")
    batch = int(sys.argv[3])
    valu = mean_by_kernel(sys.argv[4], "SQ_INSTS_VALU") if len(sys.argv) > 4 else {}
    waves = mean_by_kernel(sys.argv[4], "SQ_WAVES") if len(sys.argv) > 4 else {}
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (+ an SQ pass), separate runs of a short "
                   "bench.py extract+match loop (tools/run_profiles_rNN.sh); counters are KiB per dispatch (mean over dispatches).  Loads in these "
                   "kernels are 1-16 B per lane; the gfx950 x2 FETCH_SIZE correction for 16 B/lane streams is NOT applied "
                   "(WRITE_SIZE of k_blur7 vs its algorithmic 950532 B/frame calibrates the write side).",
           "batch": batch, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("dvm::"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"fetch_kib": f, "write_kib": w, "hbm_bytes_per_launch": (f + w) * 1024.0}
        if k in valu:
            out["kernels"][k]["valu_wave_instr_per_launch"] = valu[k]
            out["kernels"][k]["waves_per_launch"] = waves.get(k)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
