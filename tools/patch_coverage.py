#!/usr/bin/env python3
"""tools/patch_coverage.py -- what k_orient_desc HAS to fetch: the union of the 128-byte lines under the keypoints' patches (37 x 40-byte rows
of the blurred level for the descriptor, 31 x 32-byte rows of the raw level for IC_Angle), per frame of the bench stream, from the CPU
oracle's keypoints.  Compare with the kernel's counted fetch (profiles/r05_pmc_traffic.json: fetch_kib x factor / 256 frames).
Runs on the CPU (oracle only); prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dvm_slam_amd import synth
    from oracle import pyoracle as po
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    frames = synth.frame_stream(3 * nfr)[::3]
    orc = po.OrbOracle()
    sc = orc.tables()["scale"]
    line = 128
    tb = tr = buf = px = 0
    for f in frames:
        _, k, _, _ = orc.extract(f)
        for lvl in range(8):
            r, c = orc.level_dims(lvl)
            pb = (c + 63) // 64 * 64            # pitch of the blurred level
            pr = (c + 38 + 63) // 64 * 64       # pitch of the bordered raw level
            mb = np.zeros((r, pb // line + 1), bool)
            mr = np.zeros((r + 38, pr // line + 1), bool)
            kk = k[k["octave"] == lvl]
            xs = np.rint(kk["x"] / sc[lvl]).astype(int)
            ys = np.rint(kk["y"] / sc[lvl]).astype(int)
            for x, y in zip(xs, ys):
                x0, x1 = max(x - 18, 0), min(x - 18 + 39, c - 1)
                mb[max(y - 18, 0):min(y + 18, r - 1) + 1, x0 // line:x1 // line + 1] = True
                xx, yy = x + 19, y + 19
                mr[yy - 15:yy + 16, (xx - 15) // line:(xx - 15 + 31) // line + 1] = True
            tb += int(mb.sum()) * line
            tr += int(mr.sum()) * line
            buf += r * pb + (r + 38) * pr
            px += r * c
    n = len(frames)
    out = {"frames": n, "line_bytes": line, "touched_blurred_bytes_per_frame": tb / n, "touched_raw_bytes_per_frame": tr / n,
           "touched_total_bytes_per_frame": (tb + tr) / n, "both_buffers_bytes_per_frame": buf / n, "pyramid_pixels_per_frame": px / n}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")))["kernels"]["dvm::k_orient_desc"]
        out["counted_fetch_bytes_per_frame"] = d["fetch_kib"] * 1024 * d["fetch_factor"] / 256
        out["counted_over_touched"] = out["counted_fetch_bytes_per_frame"] / out["touched_total_bytes_per_frame"]
    except Exception as ex:  # noqa: BLE001
        out["counted_fetch_bytes_per_frame"] = repr(ex)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
