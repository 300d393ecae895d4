#!/usr/bin/env python3
"""tools/pmc_traffic.py --fetch F.csv [F2.csv] --write W.csv [W2.csv] [--sq SQ.csv] [--calib-fetch CF.csv --calib-write CW.csv
                        --calib-bytes JSON] [--mix valu_mix.json] --batch B   > profiles/rNN_pmc_traffic.json

Folds rocprofv3 counter_collection CSVs (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same commands) into per-kernel
HBM bytes per launch.  Counter unit: KiB per dispatch (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Template arguments are
stripped from the kernel names.

Calibration (round 3): tools/pmc_calib.hip streams a known byte count once per access width (1, 4, 8, 16 B per lane; 16 B at 4-byte
alignment) under the same two counters; factor[w] = known bytes / counted bytes.  A kernel's raw counters are corrected with the
factor of the width that carries most of its static global-load (store) bytes (tools/valu_mix.py lists them per kernel); both the
raw and the corrected figure are reported."""
import argparse
import collections
import csv
import json
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[<(].*$", "", name)


def count_by_kernel(paths, counter):
    """dispatches per kernel in the given counter CSVs (rows of `counter`)"""
    n = collections.Counter()
    for path in paths or []:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                n[short(re.sub(r"^void ", "", r["Kernel_Name"]))] += 1
    return n


def mean_by_kernel(paths, counter, keep_templates=False):
    acc = collections.defaultdict(list)
    for path in paths or []:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            n = re.sub(r"^void ", "", r["Kernel_Name"])
            n = re.sub(r"\(.*$", "", n).replace("> >", ">>") if keep_templates else short(n)
            acc[n].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def dominant_width(hist):
    if not hist:
        return None
    by_bytes = {int(w): int(w) * n for w, n in hist.items()}
    return max(by_bytes, key=by_bytes.get)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", nargs="+"); ap.add_argument("--write", nargs="+"); ap.add_argument("--sq", nargs="*")
    ap.add_argument("--calib-fetch"); ap.add_argument("--calib-write"); ap.add_argument("--calib-bytes"); ap.add_argument("--mix")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--kernels-sha", default=None, help="tools/kernels_sha.py of the build the counters were taken on")
    ap.add_argument("--ba-fetch", default=None, help="the BA run's FETCH_SIZE csv alone (dispatch counts per kernel)")
    ap.add_argument("--ba-trials", type=int, default=0, help="LM trials of the BA run behind --ba-fetch (tools/ba_short.py prints it)")
    ap.add_argument("--sq-trace", default=None, help="kernel_trace csv of the SQ pass (durations for the counter-based VALU busy fraction)")
    a = ap.parse_args()
    fetch, write = mean_by_kernel(a.fetch, "FETCH_SIZE"), mean_by_kernel(a.write, "WRITE_SIZE")
    valu, waves = mean_by_kernel(a.sq, "SQ_INSTS_VALU"), mean_by_kernel(a.sq, "SQ_WAVES")
    active, busy, gui = mean_by_kernel(a.sq, "SQ_ACTIVE_INST_VALU"), mean_by_kernel(a.sq, "SQ_BUSY_CYCLES"), mean_by_kernel(a.sq, "GRBM_GUI_ACTIVE")
    ba_disp = count_by_kernel([a.ba_fetch] if a.ba_fetch else [], "FETCH_SIZE")
    dur_ns = {}
    if a.sq_trace:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(a.sq_trace)):
            acc[short(re.sub(r"^void ", "", r["Kernel_Name"]))].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        dur_ns = {k: sum(v) / len(v) for k, v in acc.items()}
    cal = None
    if a.calib_fetch and a.calib_write and a.calib_bytes:
        known = json.loads(a.calib_bytes)["bytes"]
        cf, cw = mean_by_kernel([a.calib_fetch], "FETCH_SIZE", True), mean_by_kernel([a.calib_write], "WRITE_SIZE", True)
        width = {"calib_read<uint4>": 16, "calib_read<uint2>": 8, "calib_read<unsigned int>": 4, "calib_read<unsigned char>": 1, "calib_read_x4_unaligned": "16@4",
                 "calib_write<uint4>": 16, "calib_write<double>": 8, "calib_write<unsigned int>": 4, "calib_write<unsigned char>": 1}
        cal = {"read": {}, "write": {}, "raw": {}}
        # the profiler prints HIP's vector types by their template name
        alias = {"calib_read<uint4>": "calib_read<HIP_vector_type<unsigned int, 4u>>", "calib_read<uint2>": "calib_read<HIP_vector_type<unsigned int, 2u>>",
                 "calib_write<uint4>": "calib_write<HIP_vector_type<unsigned int, 4u>>"}
        for k0, w in width.items():
            src = cf if "read" in k0 else cw
            k = k0 if k0 in src else alias.get(k0, k0)
            if k in src and src[k] > 0:
                known[k] = known[k0]
                cal["read" if "read" in k else "write"][str(w)] = known[k] / (src[k] * 1024.0)
                cal["raw"][k] = {"known_bytes": known[k], "counter_kib": src[k]}
    mix = json.load(open(a.mix))["kernels"] if a.mix else {}
    mix_short = {}
    for k, v in mix.items():
        mix_short.setdefault(short(k), v)
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (+ an SQ pass), separate runs of a short bench.py extract+match loop and of "
                   "a short BA run (tools/run_profiles_r03.sh); counters are KiB per dispatch (mean over dispatches).  hbm_bytes_per_launch applies the per-width "
                   "calibration of tools/pmc_calib.hip (factor = known bytes / counted bytes) chosen by the kernel's dominant static global access width "
                   "(profiles/r03_valu_mix.json); hbm_bytes_per_launch_raw is (FETCH_SIZE + WRITE_SIZE) x 1024 uncorrected.",
           "batch": a.batch, "calibration": cal, "kernels_sha": a.kernels_sha, "ba_run": {"trials": a.ba_trials} if a.ba_trials else None, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("dvm::"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        rec = {"fetch_kib": f, "write_kib": w, "hbm_bytes_per_launch_raw": (f + w) * 1024.0}
        m = mix_short.get(k)
        fr = fw = 1.0
        if cal and m:
            lw, sw = dominant_width(m.get("global_load_bytes_per_lane")), dominant_width(m.get("global_store_bytes_per_lane"))
            fr = cal["read"].get(str(lw), 1.0); fw = cal["write"].get(str(sw), 1.0)
            rec.update(load_width=lw, store_width=sw, fetch_factor=fr, write_factor=fw)
        rec["hbm_bytes_per_launch"] = (f * fr + w * fw) * 1024.0
        if k in valu:
            rec["valu_wave_instr_per_launch"] = valu[k]
            rec["waves_per_launch"] = waves.get(k)
            if m:
                rec["mean_issue_cycles_per_valu_instr"] = m["mean_issue_cycles"]
        if k in active:
            # SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD spends executing VALU instructions, summed over the chip (MI355X_MICROARCH.md): x 4 /
            # (1024 SIMDs x the launch's cycles) = the COUNTER-based share of VALU issue slots that were busy.  The launch's cycles: its duration
            # in the kernel trace of the same pass x the shader clock the profiled pass ran at (GRBM_GUI_ACTIVE per XCD / duration when present)
            rec["sq_active_inst_valu_quadcycles"] = active[k]
            rec["sq_busy_cycles"] = busy.get(k)
            if k in gui:
                rec["grbm_gui_active"] = gui[k]
            if k in dur_ns and dur_ns[k] > 0:
                rec["profiled_duration_us"] = dur_ns[k] / 1e3
                ghz = gui[k] / 8.0 / dur_ns[k] if k in gui and gui[k] > 0 else 2.4
                if not 1.0 < ghz < 2.6:
                    ghz = 2.4
                rec["valu_busy_frac_counter"] = active[k] * 4.0 / (1024.0 * dur_ns[k] * ghz)
                rec["valu_busy_clock_ghz_assumed"] = ghz
        if ba_disp.get(k) and a.ba_trials:
            rec["dispatches_in_ba_run"] = ba_disp[k]
            rec["hbm_bytes_per_ba_trial"] = rec["hbm_bytes_per_launch"] * ba_disp[k] / a.ba_trials
        out["kernels"][k] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
