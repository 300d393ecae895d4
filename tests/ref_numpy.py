"""Independent numpy / pure-Python restatements used to PIN THE ORACLE (tests only).

Each function restates the published definition of a primitive in the most literal way (brute
force, no shared code with oracle/ or dvm_slam_amd/csrc), so a transcription slip in the C++ oracle
shows up as a mismatch here.  Sizes are kept tiny: these are definitions, not implementations.
"""
from __future__ import annotations

import math

import numpy as np

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_is_corner(img: np.ndarray, t: int) -> np.ndarray:
    """FAST-9/16 segment test at threshold t for every pixel with a full circle (else False)."""
    h, w = img.shape
    I = img.astype(np.int32)
    out = np.zeros((h, w), bool)
    c = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE])  # [16, h-6, w-6]
    dark = ring < (c - t)[None]
    bright = ring > (c + t)[None]
    for arr in (dark, bright):
        ext = np.concatenate([arr, arr[:8]], axis=0)  # wrap
        for s in range(16):
            out[3:h - 3, 3:w - 3] |= ext[s:s + 9].all(axis=0)
    return out


def fast_score_bruteforce(img: np.ndarray) -> np.ndarray:
    """score(p) = largest threshold t for which p is still a FAST-9/16 corner (-1 if not even at t=0)."""
    h, w = img.shape
    score = np.full((h, w), -1, np.int32)
    for t in range(0, 256):
        c = fast_is_corner(img, t)
        if not c.any():
            break
        score[c] = t
    return score


def fast_detect(roi: np.ndarray, threshold: int):
    """cv::FAST(roi, threshold, nonmaxSuppression=true) by definition; returns (xs, ys, scores) row-major."""
    score = fast_score_bruteforce(roi)
    h, w = roi.shape
    s = np.where(score >= threshold, score, 0)  # corner at `threshold` <=> score >= threshold
    # a corner exists only if max strength > threshold, i.e. score >= threshold (score = strength-1)
    xs, ys, sc = [], [], []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = s[y, x]
            if score[y, x] < threshold or v <= 0:
                continue
            nb = s[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (v > nb).all():
                xs.append(x); ys.append(y); sc.append(int(v))
    return np.array(xs, np.int32), np.array(ys, np.int32), np.array(sc, np.int32)


def gaussian_blur7_fixed(img: np.ndarray, k=(18, 34, 48, 56, 48, 34, 18)) -> np.ndarray:
    """(sum_j sum_i k[j]k[i] src + 32768) >> 16 with BORDER_REFLECT_101 (numpy 'reflect' padding)."""
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    h, w = img.shape
    acc = np.zeros((h, w), np.int64)
    for j in range(7):
        for i in range(7):
            acc += k[j] * k[i] * p[j:j + h, i:i + w]
    return ((acc + 32768) >> 16).astype(np.uint8)


def gaussian_kernel7_ed(sigma=2.0):
    x = np.arange(7) - 3.0
    k = np.exp(-0.5 * x * x / sigma ** 2)
    k = k / k.sum()
    out = [0] * 7
    err = 0.0
    for i in range(3):
        adj = k[i] * 256 + err
        v = int(np.rint(adj))
        err = adj - v
        out[i] = out[6 - i] = v
    out[3] = 256 - 2 * sum(out[:3])
    return out


def resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize INTER_LINEAR 8-bit: 11-bit weights (round-half-even), H pass int32, V pass with >>4 / >>16 / +2>>2."""
    sh, sw = src.shape
    S = src.astype(np.int64)

    def axis(ssize, dsize, clamp):
        scale = 1.0 / (dsize / ssize)
        f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp:
            lo = s < 0
            f[lo] = 0; s[lo] = 0
            hi = s >= ssize - 1
            f[hi] = 0; s[hi] = ssize - 1
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, w0, w1

    sx, a0, a1 = axis(sw, dw, True)
    sy, b0, b1 = axis(sh, dh, False)
    sx1 = np.minimum(sx + 1, sw - 1)  # weight is 0 where this clamps
    H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    v = (((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def umax_table():
    hp = 15
    umax = [0] * 16
    vmax = int(math.floor(hp * math.sqrt(2.0) / 2 + 1))
    vmin = int(math.ceil(hp * math.sqrt(2.0) / 2))
    for v in range(vmax + 1):
        umax[v] = int(np.rint(math.sqrt(hp * hp - v * v)))
    v0 = 0
    for v in range(hp, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    return umax


def ic_moments(img: np.ndarray, cx: int, cy: int):
    um = umax_table()
    m10 = m01 = 0
    for v in range(-15, 16):
        d = um[abs(v)]
        for u in range(-d, d + 1):
            val = int(img[cy + v, cx + u])
            m10 += u * val
            m01 += v * val
    return m01, m10


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


# ----------------------------------------------------------------------------------------- octree
def distribute_octree(xs, ys, scores, minX, maxX, minY, maxY, N):
    """DistributeOctTree with Python lists mimicking std::list.  Returns (indices, had_sort_ties):
    when std::sort would see equal keys its (unstable) order is implementation-defined, so callers
    only compare tie-free cases."""
    n = len(xs)
    nIni = int(np.rint(np.float32(maxX - minX) / np.float32(maxY - minY)))
    hX = np.float32(maxX - minX) / np.float32(nIni)
    nodes = []  # each node: dict(x0,y0,x1,y1,keys)
    roots = []
    for i in range(nIni):
        roots.append(dict(x0=int(hX * np.float32(i)), y0=0, x1=int(hX * np.float32(i + 1)), y1=maxY - minY, keys=[]))
    for i in range(n):
        roots[int(np.float32(xs[i]) / hX)]["keys"].append(i)
    lst = [r for r in roots if r["keys"]]
    ties = False

    def split(nd):
        hx = int(math.ceil(np.float32(nd["x1"] - nd["x0"]) / 2))
        hy = int(math.ceil(np.float32(nd["y1"] - nd["y0"]) / 2))
        xm, ym = nd["x0"] + hx, nd["y0"] + hy
        ch = [dict(x0=nd["x0"], y0=nd["y0"], x1=xm, y1=ym, keys=[]), dict(x0=xm, y0=nd["y0"], x1=nd["x1"], y1=ym, keys=[]),
              dict(x0=nd["x0"], y0=ym, x1=xm, y1=nd["y1"], keys=[]), dict(x0=xm, y0=ym, x1=nd["x1"], y1=nd["y1"], keys=[])]
        for k in nd["keys"]:
            q = (0 if xs[k] < xm else 1) + (0 if ys[k] < ym else 2)
            ch[q]["keys"].append(k)
        return [c for c in ch if c["keys"]]

    finish = False
    while not finish:
        prev = len(lst)
        expandable = []
        # emulate: iterate list, children pushed to the front, parent erased
        new_front = []
        keep = []
        for nd in lst:
            if len(nd["keys"]) == 1:
                keep.append(nd)
                continue
            for c in split(nd):
                new_front.insert(0, c)
                if len(c["keys"]) > 1:
                    expandable.append(c)
        lst = new_front + keep
        nToExpand = len(expandable)
        if len(lst) >= N or len(lst) == prev:
            finish = True
        elif len(lst) + nToExpand * 3 > N:
            while not finish:
                prev = len(lst)
                keys = [(len(c["keys"]), c["x0"]) for c in expandable]
                if len(set(keys)) != len(keys):
                    ties = True
                order = sorted(range(len(expandable)), key=lambda j: keys[j])  # stable; exact only when tie-free
                cur = [expandable[j] for j in order]
                expandable = []
                for nd in reversed(cur):
                    for c in split(nd):
                        lst.insert(0, c)
                        if len(c["keys"]) > 1:
                            expandable.append(c)
                    lst.pop(next(i for i, m in enumerate(lst) if m is nd))
                    if len(lst) >= N:
                        break
                if len(lst) >= N or len(lst) == prev:
                    finish = True
    out = []
    for nd in lst:
        best = nd["keys"][0]
        for k in nd["keys"][1:]:
            if scores[k] > scores[best]:
                best = k
        out.append(best)
    return np.array(out, np.int32), ties


# ------------------------------------------------------------------------------------------ grid
def grid_features_in_area(kps, x, y, r, minLevel, maxLevel, bounds=(0.0, 640.0, 0.0, 480.0)):
    minX, maxX, minY, maxY = [np.float32(v) for v in bounds]
    wInv = np.float32(64) / np.float32(maxX - minX)
    hInv = np.float32(48) / np.float32(maxY - minY)
    x, y, r = np.float32(x), np.float32(y), np.float32(r)
    cells = {}
    for i, kp in enumerate(kps):
        px = int(_round_half_away((np.float32(kp["x"]) - minX) * wInv))
        py = int(_round_half_away((np.float32(kp["y"]) - minY) * hInv))
        if 0 <= px < 64 and 0 <= py < 48:
            cells.setdefault((px, py), []).append(i)
    c0 = max(0, int(np.floor((x - minX - r) * wInv)))
    if c0 >= 64:
        return []
    c1 = min(63, int(np.ceil((x - minX + r) * wInv)))
    if c1 < 0:
        return []
    r0 = max(0, int(np.floor((y - minY - r) * hInv)))
    if r0 >= 48:
        return []
    r1 = min(47, int(np.ceil((y - minY + r) * hInv)))
    if r1 < 0:
        return []
    check = (minLevel > 0) or (maxLevel >= 0)
    out = []
    for ix in range(c0, c1 + 1):
        for iy in range(r0, r1 + 1):
            for i in cells.get((ix, iy), []):
                kp = kps[i]
                if check:
                    if kp["octave"] < minLevel:
                        continue
                    if maxLevel >= 0 and kp["octave"] > maxLevel:
                        continue
                if abs(np.float32(kp["x"]) - x) < r and abs(np.float32(kp["y"]) - y) < r:
                    out.append(i)
    return out


def _round_half_away(v):
    v = float(v)
    return math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)


# -------------------------------------------------------------------------------------------- BA
def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def se3_exp(u):
    om, up = u[:3], u[3:]
    th = np.linalg.norm(om)
    O = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-5:
        R = np.eye(3) + O + O @ O
        V = R
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th ** 2 * (O @ O)
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * O + (th - np.sin(th)) / th ** 3 * (O @ O)
    return R, V @ up


def ba_lm_dense(Rs, ts, fixed, pts, edge_pose, edge_point, obs, info, K, delta, iterations):
    """Plain LM on the FULL normal equations (no Schur trick), g2o's control flow; rotation matrices as state."""
    fx, fy, cx, cy = K
    Rs = [R.copy() for R in Rs]; ts = [t.copy() for t in ts]; pts = pts.copy()
    free = [p for p in range(len(Rs)) if not fixed[p]]
    pi = {p: i for i, p in enumerate(free)}
    n, m = 6 * len(free), 3 * len(pts)

    def rho(e):
        if delta <= 0 or e <= delta * delta:
            return e, 1.0
        s = math.sqrt(e)
        return 2 * s * delta - delta * delta, delta / s

    def chi(Rs_, ts_, pts_):
        tot = 0.0
        for k in range(len(edge_pose)):
            Xc = Rs_[edge_pose[k]] @ pts_[edge_point[k]] + ts_[edge_pose[k]]
            e = obs[k] - np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy])
            tot += rho(info[k] * (e @ e))[0]
        return tot

    lam, ni, nbad = -1.0, 2.0, 0
    hist = []
    dx = np.zeros(n + m)
    n_fail = 0
    for it in range(iterations):
        cur = chi(Rs, ts, pts)
        ini = cur
        H = np.zeros((n + m, n + m)); b = np.zeros(n + m)
        for k in range(len(edge_pose)):
            p, l = edge_pose[k], edge_point[k]
            R = Rs[p]
            Xc = R @ pts[l] + ts[p]
            x, y, z = Xc
            e = obs[k] - np.array([fx * x / z + cx, fy * y / z + cy])
            r0, r1 = rho(info[k] * (e @ e))
            Jp = -np.array([[fx / z, 0, -fx * x / z ** 2], [0, fy / z, -fy * y / z ** 2]])
            A = Jp @ R
            S = np.array([[0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1.0]])
            B = Jp @ S
            w = r1 * info[k]
            J = np.zeros((2, n + m))
            J[:, n + 3 * l:n + 3 * l + 3] = A
            if p in pi:
                J[:, 6 * pi[p]:6 * pi[p] + 6] = B
            H += w * (J.T @ J)
            b += J.T @ (-info[k] * e * r1)
        if it == 0:
            lam = 1e-5 * np.abs(np.diag(H)).max(); ni = 2.0; nbad = 0
        rho_ = 0.0; q = 0
        while True:
            bak = ([R.copy() for R in Rs], [t.copy() for t in ts], pts.copy())
            Hd = H + lam * np.eye(n + m)
            # g2o's linear solver works on the reduced camera system and FAILS when that is not positive definite; the LM loop then
            # applies whatever x still holds -- the last successful solve, zeros before the first -- and overrides tempChi with max()
            # (optimization_algorithm_levenberg.cpp:107-127)
            ok = True
            try:
                np.linalg.cholesky(Hd[:n, :n] - Hd[:n, n:] @ np.linalg.solve(Hd[n:, n:], Hd[n:, :n]))
            except np.linalg.LinAlgError:
                ok = False
            if ok:
                dx = np.linalg.solve(Hd, b)
            n_fail += 0 if ok else 1
            for p in free:
                dR, dt = se3_exp(dx[6 * pi[p]:6 * pi[p] + 6])
                Rs[p] = dR @ Rs[p]; ts[p] = dR @ ts[p] + dt
            pts = pts + dx[n:].reshape(-1, 3)
            tmp = chi(Rs, ts, pts) if ok else np.finfo(np.float64).max
            scale = dx @ (lam * dx + b) + 1e-3
            rho_ = (cur - tmp) / scale
            if rho_ > 0 and np.isfinite(tmp):
                alpha = min(1 - (2 * rho_ - 1) ** 3, 2 / 3)
                lam *= max(1 / 3, alpha); ni = 2.0; cur = tmp
            else:
                lam *= ni; ni *= 2; Rs, ts, pts = bak
            q += 1
            if not (rho_ < 0 and q < 10):
                break
        hist.append((q, cur, lam, n_fail))
        if q == 10 or rho_ == 0:
            break
        nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
        if nbad >= 3:
            break
    return Rs, ts, pts, hist
