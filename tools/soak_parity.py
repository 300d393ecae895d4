#!/usr/bin/env python3
"""tools/soak_parity.py [seconds] [seed] -- randomized GPU-vs-oracle soak of the ORB extractor and the window matcher.

Random image sizes, contents (texture, noise, low contrast, flat regions, gradients, checkerboards), extractor
parameters (nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST), lapping areas and batch sizes; every keypoint field
and descriptor byte must be identical to the CPU oracle.  Not part of the pytest suite (runs for minutes); prints one
line per mismatch and a summary, exit code 1 on any mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvm_slam_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po     # noqa: E402


def make_image(rng, h, w):
    kind = rng.integers(0, 7)
    if kind == 0:
        return synth.small_image(int(rng.integers(1 << 30)), h, w)
    if kind == 1:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:
        return (100 + rng.integers(0, int(rng.integers(4, 40)), (h, w))).astype(np.uint8)
    if kind == 3:   # texture with flat holes (cells fall through to minTh / stay empty)
        img = synth.small_image(int(rng.integers(1 << 30)), h, w)
        for _ in range(6):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + h // 3, x:x + w // 3] = rng.integers(0, 256)
        return img
    if kind == 4:   # checkerboard of random pitch: masses of identical scores (ties everywhere)
        p = int(rng.integers(3, 17))
        yy, xx = np.mgrid[0:h, 0:w]
        return (((yy // p + xx // p) & 1) * int(rng.integers(40, 255))).astype(np.uint8)
    if kind == 5:   # smooth gradient + sparse salt
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((xx * 255 // max(w - 1, 1) + yy * 255 // max(h - 1, 1)) // 2).astype(np.uint8)
        m = rng.random((h, w)) < 0.003
        img[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
        return img
    return np.full((h, w), int(rng.integers(0, 256)), np.uint8)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    cases = bad = refused = 0
    while time.time() - t0 < budget:
        h, w = int(rng.integers(64, 1000)), int(rng.integers(80, 1300))
        if rng.random() < 0.3:
            h, w = 480, 640
        if w < h:          # portrait levels give nIni = round(W / H) = 0 root nodes: the reference divides by zero there
            h, w = w, h
        nf = int(rng.choice([100, 500, 1000, 1500, 3000]))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nl = int(rng.integers(1, 9))
        ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
        # keep the smallest level usable (the reference itself misbehaves below ~40 px)
        while nl > 1 and min(h, w) / sf ** (nl - 1) < 48:
            nl -= 1
        lap = (0, 1000) if rng.random() < 0.6 else (int(rng.integers(0, w)), int(rng.integers(0, w)))
        B = int(rng.choice([1, 1, 2, 5]))
        imgs = np.stack([make_image(rng, h, w) for _ in range(B)])
        try:
            e = capi.OrbExtractor(nf, sf, nl, ini, mn, max_batch=B)
            orc = po.OrbOracle(nf, sf, nl, ini, mn)
            if B == 1:
                res = [e.extract(imgs[0], lap=lap)]
            else:
                e.extract_batch_host(imgs, lap=lap)
                res = [e.download(f) for f in range(B)]
            for f in range(B):
                n_o, k_o, d_o, m_o = orc.extract(imgs[f], lap=lap, cap=4 * nf + 64)
                n_g, k_g, d_g, m_g = res[f]
                ok = (n_g, m_g) == (n_o, m_o) and np.array_equal(d_g, d_o) and all(
                    np.array_equal(k_g[c], k_o[c]) for c in ("x", "y", "size", "angle", "response", "octave"))
                if not ok:
                    bad += 1
                    print(f"MISMATCH case {cases} frame {f}: {h}x{w} nf={nf} sf={sf} nl={nl} th={ini}/{mn} lap={lap} B={B} "
                          f"n {n_g} vs {n_o} mono {m_g} vs {m_o}", flush=True)
            # matcher on the last frame's keypoints against themselves shifted
            if n_g > 10:
                g = capi.FrameGrid(capacity=max(2048, n_g + 64))
                g.build(k_g, d_g, bounds=(0.0, float(w), 0.0, float(h)))
                go = po.Grid(k_o, 0.0, float(w), 0.0, float(h))
                qx = (k_g["x"] + rng.normal(0, 3, n_g)).astype(np.float32); qy = (k_g["y"] + rng.normal(0, 3, n_g)).astype(np.float32)
                qr = rng.choice([2.0, 7.0, 15.0, 40.0], n_g).astype(np.float32)
                lo = (k_g["octave"] - rng.integers(0, 2, n_g)).astype(np.int32); hi = (k_g["octave"] + rng.integers(0, 2, n_g)).astype(np.int32)
                mg = g.match_window(d_g, qx, qy, qr, lo, hi)
                mo = go.match_window(d_o, d_o, qx, qy, qr, lo, hi)
                for c in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
                    if not np.array_equal(mg[c].astype(np.int64), mo[c].astype(np.int64)):
                        bad += 1
                        print(f"MATCH MISMATCH case {cases}: field {c} {h}x{w}", flush=True)
                        break
                g.close()
            e.close()
        except capi.DvmError as ex:
            if "octree capacity" in str(ex):   # a level quota above the device octree's 2 680 nodes is refused by design
                refused += 1
            else:
                bad += 1
                print(f"EXCEPTION case {cases}: {h}x{w} nf={nf} sf={sf} nl={nl} th={ini}/{mn} B={B}: {ex!r}", flush=True)
        except Exception as ex:   # a failure of either side is a finding too
            bad += 1
            print(f"EXCEPTION case {cases}: {h}x{w} nf={nf} sf={sf} nl={nl} th={ini}/{mn} B={B}: {ex!r}", flush=True)
        cases += 1
    print(f"soak: {cases} cases ({refused} refused: quota beyond the device octree), {bad} mismatches, {time.time() - t0:.0f} s, seed {seed}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
