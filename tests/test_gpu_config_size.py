"""Oracle parity at the sizes BASELINE.json quotes (VERDICT r01 "config-size parity holes"):
* global BA 500 keyframes / 20 000 landmarks / 160 000 observations, Huber on and bRobust = false (LoopClosing.cc:2282),
* the DBoW2 tree at the reference's shape k = 10, L = 6 (TemplatedVocabulary.h:1107-1144; ORBvoc.txt is ~1.08 M nodes),
* the essential graph at 500 keyframes.
Tolerances are the north star's (BA poses / landmarks 1e-6, identical LM trial sequence) or stated with the test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("delta,iters", [(float(np.sqrt(5.991)), 10), (0.0, 10)])
def test_global_ba_500kf_matches_oracle(capi, oracle, delta, iters):
    from dvm_slam_amd import synth
    pr = synth.ba_problem()          # 500 KF / 20 000 landmarks / 160 000 edges, seed 0xBA5E
    assert len(pr["poses"]) == 500 and len(pr["points"]) == 20000 and len(pr["obs"]) == 160000
    e = oracle.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    po_, pto, so, chio = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, iters)
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    sg = ba.optimize(iters)
    pg, ptg = ba.result()
    chig, depth = ba.edge_chi2()
    ba.close()
    assert sg["iterations"] == so["iterations"] and sg["trials"] == so["trials"], "LM accept / reject sequence differs"
    assert sg["stop_reason"] == so["stop_reason"]
    assert abs(sg["chi2_initial"] - so["chi2_initial"]) <= 1e-9 * so["chi2_initial"]
    assert np.allclose(sg["chi2"], so["chi2"], rtol=1e-9)
    assert np.allclose(sg["lam"], so["lam"], rtol=1e-6)
    assert np.abs(pg - po_).max() < 1e-6, np.abs(pg - po_).max()
    assert np.abs(ptg - pto).max() < 1e-6, np.abs(ptg - pto).max()
    assert np.allclose(chig, chio, rtol=1e-6, atol=1e-9)
    assert np.array_equal(pg[0], po_[0]) and depth.all()


def test_global_ba_four_times_the_baseline_size(capi, oracle):
    """2 000 keyframes / 80 000 landmarks / 640 000 observations (4x BASELINE config 5; S is a 12 864^2 tile matrix, 798 non-zero
    tiles, ten elimination-tree levels): same LM trial sequence as the oracle, poses / landmarks far inside 1e-6."""
    from dvm_slam_amd import synth
    pr = synth.ba_problem(n_kf=2000, n_pts=80000, seed=7)
    assert len(pr["obs"]) == 640000
    e = oracle.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    delta = float(np.sqrt(5.991))
    po_, pto, so, _ = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, 6)
    ba = capi.BundleAdjuster()
    ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
    sg = ba.optimize(6)
    pg, ptg = ba.result()
    info = ba.schedule_info()
    ba.close()
    assert sg["trials"] == so["trials"] and sg["iterations"] == so["iterations"]
    assert np.allclose(sg["chi2"], so["chi2"], rtol=1e-9)
    assert np.abs(pg - po_).max() < 1e-6 and np.abs(ptg - pto).max() < 1e-6
    assert info["free_cameras"] == 1999 and info["nz_tiles"] > 500


@pytest.mark.parametrize("ragged", [False, True])
def test_vocabulary_reference_shape_k10_L6(capi, oracle, ragged):
    """transform() through a 6-level, fan-out-10 tree (1 111 111 nodes when full) and the keyframe-database query on the
    resulting BowVectors: word / node ids, weights, BoW doubles and the per-keyframe (common words, first word, score) identical."""
    from dvm_slam_amd import synth
    voc = synth.vocabulary(k=10, L=6, seed=106, ragged=ragged)
    assert voc["L"] == 6 and (voc["n_nodes"] == 1111111 or ragged)
    rng = np.random.default_rng(6)
    leaves = np.flatnonzero(voc["word_id"] >= 0)
    feats = voc["desc"][rng.choice(leaves, 6000)].copy()
    feats[rng.random(feats.shape) < 0.03] ^= 0x42
    feats[:100] = rng.integers(0, 256, (100, 32), dtype=np.uint8)
    v = capi.Vocabulary(voc)
    for levelsup in (0, 4, 6):      # ORB-SLAM3 uses levelsup = 4 (Frame::ComputeBoW)
        wg, ng, wtg = v.transform(feats, levelsup)
        r = oracle.vocab_transform(voc, feats, levelsup)
        assert np.array_equal(wg, r["word"]) and np.array_equal(ng, r["node"]) and np.array_equal(wtg, r["weight"])
    v.close()
    # six "keyframes" of 1000 features each -> BowVectors -> database query of the first against all
    bows = []
    for k in range(6):
        h = capi.vocab_transform_host(voc, feats[k * 1000:(k + 1) * 1000], 4)
        ro = oracle.vocab_transform(voc, feats[k * 1000:(k + 1) * 1000], 4)
        for key in ("bow_ids", "bow_vals", "fv_nodes", "fv_off", "fv_feat"):
            assert np.array_equal(h[key], ro[key]), key
        bows.append((h["bow_ids"], h["bow_vals"]))
    q = capi.vocab_transform_host(voc, np.concatenate([feats[:500], feats[1000:1500]]), 4)
    common, first, score = capi.bowdb_query_raw(bows, q["bow_ids"], q["bow_vals"])
    for k, (ids, vals) in enumerate(bows):
        shared = np.intersect1d(ids, q["bow_ids"])
        assert common[k] == len(shared) and (len(shared) == 0 or first[k] == shared[0])
        assert score[k] == np.float32(oracle.bow_score(q["bow_ids"], q["bow_vals"], ids, vals))
    assert common[0] > 100 and common[1] > 100


def test_essential_graph_500_keyframes_first_step(capi, oracle):
    """dvm_pose_graph_optimize at 500 keyframes against the oracle.  g2o differentiates EdgeSim3 numerically (delta = 1e-9): J
    carries ~5e-8 of noise, H = J'J is conditioned ~1e9 on a 500-vertex chain gauged at one vertex, lambda starts at 1e-16 --
    from the second LM iteration on the trajectory is decided by that noise (measured: |dt| 0.15 after iteration 2, both sides
    equally far from the ground truth; the reference is in the same regime).  What IS comparable, and compared here with stated
    tolerances: the first LM iteration -- a 4.8-unit correction of the drifted estimates -- agrees to chi2 rel. 5e-6,
    translations 1e-4 (2e-5 of the step), quaternions 5e-6, scales 1e-5; and the 20-iteration run brings chi2 down by two orders of
    magnitude and, on the noisy graph, ends several times closer to the ground truth than it started.  How far down is decided by rounding, on both sides
    (tools/pg_noise.py, chi2_final / chi2_initial after 20 iterations): the ORACLE ends at 1.2e-3 on the noise-free graph and
    at 1.4e-4 .. 2.1e-4 on the noisy ones; the device anywhere in 7e-5 .. 9e-3, depending on which (equally accurate, 1.24 ulp)
    reciprocal square root and which summation order the tile Cholesky's inverse uses."""
    from dvm_slam_amd import synth
    for noise, seed in ((0.002, 500), (0.0, 500)):
        pg = synth.pose_graph(n=500, noise=noise, seed=seed)
        So, sto = oracle.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=1)
        Sg, stg = capi.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=1)
        assert stg["iterations"] == int(sto[0]) == 1 and stg["trials_per_iter"][0] == int(sto[38]) == 1
        assert abs(stg["chi2_initial"] - sto[2]) <= 1e-9 * sto[2]
        assert abs(stg["chi2_final"] - sto[3]) <= 5e-6 * sto[3]
        step = np.abs(So[:, 4:7] - pg["S0"][:, 4:7]).max()
        assert step > 3.0
        assert np.abs(Sg[:, 4:7] - So[:, 4:7]).max() < 1e-4
        assert np.abs(Sg[:, :4] - So[:, :4]).max() < 5e-6
        assert np.abs(Sg[:, 7] - So[:, 7]).max() < 1e-5
        assert np.array_equal(Sg[0], pg["S0"][0])
        Sg, stg = capi.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=20)
        e0 = np.abs(pg["S0"][:, 4:7] - pg["S_gt"][:, 4:7]).max()
        if noise > 0:      # (the noise-free graph's optimum is a zero-residual valley: steps along its near-null directions do not show
            assert np.abs(Sg[:, 4:7] - pg["S_gt"][:, 4:7]).max() < 0.15 * e0      # in chi2 -- the ORACLE ends 0.75 e0 from the ground truth there)
        assert stg["chi2_final"] < 2e-2 * stg["chi2_initial"]               # docstring: the oracle itself is at 1.2e-3 here
