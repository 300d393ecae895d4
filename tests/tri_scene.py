"""TEST INFRASTRUCTURE: two keyframes looking at a cloud of points, the matches SearchForTriangulation would hand to
LocalMapping::CreateNewMapPoints (LocalMapping.cc:598-741) -- correct ones, wrong ones (reprojection / depth failures), distant
points (low parallax), octave pairs that contradict the distance ratio -- and a float64 numpy statement of the same tests
(np.linalg.svd for the triangulation) that reports how far every decision is from its threshold."""
import numpy as np

from dvm_slam_amd import synth

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def scene(seed=0, n=600, baseline=0.6, depth=(2.0, 12.0), noise_px=0.5, wrong_frac=0.15, far_frac=0.1, K=(458.0, 457.0, 367.0, 248.0)):
    rng = np.random.default_rng(seed)
    K = np.asarray(K, np.float32)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    sigma2 = (sf * sf).astype(np.float32)

    def pose(rotvec, t):
        R = synth._rot_from_axis_angle(np.asarray(rotvec, float))
        return R, np.asarray(t, float)
    R1, t1 = pose(rng.normal(0, 0.05, 3), rng.normal(0, 0.1, 3))
    R2, t2 = pose(rng.normal(0, 0.05, 3), np.array([-baseline, 0.02, 0.05]) + rng.normal(0, 0.02, 3))
    T1 = np.hstack([R1, t1[:, None]]).astype(np.float32)
    T2 = np.hstack([R2, t2[:, None]]).astype(np.float32)
    Ow1 = (-(T1[:, :3].astype(np.float64).T @ T1[:, 3].astype(np.float64))).astype(np.float32)
    Ow2 = (-(T2[:, :3].astype(np.float64).T @ T2[:, 3].astype(np.float64))).astype(np.float32)
    # points in front of camera 1
    z = rng.uniform(*depth, n)
    far = rng.random(n) < far_frac
    z[far] *= rng.uniform(80, 400, far.sum())                       # cos parallax -> 1
    u = rng.uniform(40, 700, n); v = rng.uniform(40, 440, n)
    Pc1 = np.stack([(u - K[2]) / K[0] * z, (v - K[3]) / K[1] * z, z], 1)
    Pw = (Pc1 - t1) @ R1                                             # R1^T (Pc - t)
    Pc2 = Pw @ R2.T + t2
    uv1 = np.stack([K[0] * Pc1[:, 0] / Pc1[:, 2] + K[2], K[1] * Pc1[:, 1] / Pc1[:, 2] + K[3]], 1)
    uv2 = np.stack([K[0] * Pc2[:, 0] / Pc2[:, 2] + K[2], K[1] * Pc2[:, 1] / Pc2[:, 2] + K[3]], 1)
    oct1 = rng.integers(0, 8, n)
    d1 = np.linalg.norm(Pw - Ow1, axis=1); d2 = np.linalg.norm(Pw - Ow2, axis=1)
    # the octave the other view would see the point at, +- 1, some inconsistent
    oct2 = np.clip(oct1 + np.rint(np.log(d1 / d2) / np.log(1.2)).astype(int) + rng.integers(-1, 2, n), 0, 7)
    bad_oct = rng.random(n) < 0.08
    oct2[bad_oct] = np.clip(oct1[bad_oct] + rng.choice([-4, 4], bad_oct.sum()), 0, 7)
    kps1 = np.zeros(n, KP_DTYPE); kps2 = np.zeros(n, KP_DTYPE)
    kps1["x"] = uv1[:, 0] + rng.normal(0, noise_px, n) * sf[oct1]; kps1["y"] = uv1[:, 1] + rng.normal(0, noise_px, n) * sf[oct1]
    kps2["x"] = uv2[:, 0] + rng.normal(0, noise_px, n) * sf[oct2]; kps2["y"] = uv2[:, 1] + rng.normal(0, noise_px, n) * sf[oct2]
    kps1["octave"] = oct1; kps2["octave"] = oct2
    idx2 = np.arange(n)
    wrong = rng.random(n) < wrong_frac
    idx2[wrong] = rng.integers(0, n, wrong.sum())                    # mismatches: reprojection / depth failures
    perm = rng.permutation(n)
    pairs = np.stack([np.arange(n)[perm], idx2[perm]], 1).astype(np.int32)
    return dict(K1=K, K2=K, T1w=T1, T2w=T2, Ow1=Ow1, Ow2=Ow2, kps1=kps1, kps2=kps2, pairs=pairs, sigma2_1=sigma2, sigma2_2=sigma2,
                sf1=sf, sf2=sf, ratio_factor=np.float32(1.5) * sf[1])


def numpy_reference(S, cos_parallax_max=0.9998, far_points=False, th_far=0.0):
    """float64 statement.  Returns (x3D, status, margin): margin = relative distance of the DECIDING comparison from its threshold
    (small = the float paths may legitimately decide differently)."""
    K1, K2 = S["K1"].astype(float), S["K2"].astype(float)
    T1, T2 = S["T1w"].astype(float), S["T2w"].astype(float)
    n = len(S["pairs"])
    X = np.zeros((n, 3)); st = np.zeros(n, np.int32); margin = np.full(n, np.inf)
    for m, (i1, i2) in enumerate(S["pairs"]):
        k1, k2 = S["kps1"][i1], S["kps2"][i2]
        xn1 = np.array([(k1["x"] - K1[2]) / K1[0], (k1["y"] - K1[3]) / K1[1], 1.0])
        xn2 = np.array([(k2["x"] - K2[2]) / K2[0], (k2["y"] - K2[3]) / K2[1], 1.0])
        r1, r2 = T1[:, :3].T @ xn1, T2[:, :3].T @ xn2
        c = r1 @ r2 / (np.linalg.norm(r1) * np.linalg.norm(r2))
        ms = [abs(c), abs(c - cos_parallax_max) / (1 - cos_parallax_max)]
        if not (c > 0 and c < cos_parallax_max):
            st[m] = 1; margin[m] = min(ms); continue
        A = np.stack([xn1[0] * T1[2] - T1[0], xn1[1] * T1[2] - T1[1], xn2[0] * T2[2] - T2[0], xn2[1] * T2[2] - T2[1]])
        vh = np.linalg.svd(A)[2][3]
        x = vh[:3] / vh[3]
        X[m] = x
        z1 = T1[2, :3] @ x + T1[2, 3]; z2 = T2[2, :3] @ x + T2[2, 3]
        ms.append(abs(z1) / (abs(x[2]) + 1))
        if z1 <= 0:
            st[m] = 3; margin[m] = min(ms); continue
        ms.append(abs(z2) / (abs(x[2]) + 1))
        if z2 <= 0:
            st[m] = 4; margin[m] = min(ms); continue
        done = False
        for code, T, K, kp, sig in ((5, T1, K1, k1, S["sigma2_1"]), (6, T2, K2, k2, S["sigma2_2"])):
            pc = T[:, :3] @ x + T[:, 3]
            e = (K[0] * pc[0] / pc[2] + K[2] - kp["x"]) ** 2 + (K[1] * pc[1] / pc[2] + K[3] - kp["y"]) ** 2
            lim = 5.991 * float(sig[kp["octave"]])
            ms.append(abs(e - lim) / lim)
            if e > lim:
                st[m] = code; margin[m] = min(ms); done = True; break
        if done:
            continue
        d1 = np.linalg.norm(x - S["Ow1"]); d2 = np.linalg.norm(x - S["Ow2"])
        if far_points:
            ms += [abs(d1 - th_far) / th_far, abs(d2 - th_far) / th_far]
            if d1 >= th_far or d2 >= th_far:
                st[m] = 8; margin[m] = min(ms); continue
        rd = d2 / d1; ro = float(S["sf1"][k1["octave"]]) / float(S["sf2"][k2["octave"]]); rf = float(S["ratio_factor"])
        ms += [abs(rd * rf - ro) / ro, abs(rd - ro * rf) / (ro * rf)]
        if rd * rf < ro or rd > ro * rf:
            st[m] = 9
        margin[m] = min(ms)
    return X, st, margin
