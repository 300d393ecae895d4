"""The flow form of the reduced solve (k_chol_flow: tile factorisation + back substitution as ONE persistent launch of tile tasks,
csrc/ba_kernels.hip) against the level launches it replaces and against the CPU oracle.

The flow form gathers a tile's updates level by level in the order the level launches apply them and runs the same tile factorisation,
panel solve and back substitution arithmetic: with the top pair kernel switched off (DVM_BA_NO_PAIR: its in-LDS solve of the last two
columns sums in another order) the two forms must agree BIT FOR BIT -- poses, landmarks, chi2, lambda, the LM trial sequence.  Against the default level
launches (pair on) and the oracle the bound is the north star's 1e-6 with identical trial sequences.  Reference recipe:
G2O/core/block_solver.hpp:354-486, G2O/solvers/linear_solver_eigen.h:89-112."""
import os

import numpy as np
import pytest

from dvm_slam_amd import capi, synth

pytestmark = pytest.mark.gpu
DELTA = float(np.sqrt(5.991))


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.int64)


def _run(pr, delta, iters, env, rounds=1):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
        ba = capi.BundleAdjuster()
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        out = []
        for _ in range(rounds):
            st = ba.optimize(iters)
            P, X = ba.result()
            chi, _ = ba.edge_chi2()
            out.append((P, X, st, chi))
        info = ba.schedule_info()
        ba.close()
        return out, info
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same_bits(a, b, tag):
    for (Pa, Xa, sa, ca), (Pb, Xb, sb, cb) in zip(a, b):
        assert sa["trials"] == sb["trials"] and sa["iterations"] == sb["iterations"] and sa["stop_reason"] == sb["stop_reason"], tag
        assert np.array_equal(_bits(sa["chi2"]), _bits(sb["chi2"])), (tag, sa["chi2"], sb["chi2"])
        assert np.array_equal(_bits(sa["lam"]), _bits(sb["lam"])), tag
        assert np.array_equal(_bits(Pa), _bits(Pb)), (tag, np.abs(Pa - Pb).max())
        assert np.array_equal(_bits(Xa), _bits(Xb)), (tag, np.abs(Xa - Xb).max())
        assert np.array_equal(_bits(ca), _bits(cb)), tag


PROBLEMS = {
    "ring500": dict(),                                                       # BASELINE config 5: 7 levels of tile columns
    "loop500": dict(laps=2, long_range_frac=0.002),                          # bench.py's second BA workload: 37 levels, 69 % tile fill
    "ring87": dict(n_kf=87, n_pts=2500, seed=11),                            # a short last tile (87 = 8 x 10 + 7)
    "ring230_k4": dict(n_kf=230, n_pts=6000, k_obs=4, seed=12),
    "web120": dict(n_kf=120, n_pts=4000, seed=13, long_range_frac=0.02),     # scattered long-range couplings: nearly dense tile structure
    "small24": dict(n_kf=24, n_pts=900, seed=14),                            # 3 tile columns
}


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_flow_equals_level_launches_bit_for_bit(name):
    pr = synth.ba_problem(**PROBLEMS[name])
    iters = 6 if len(pr["poses"]) >= 500 else 8
    # (DVM_BA_BORDER=0: the deep trees are what this file is about)
    lvl, info_l = _run(pr, DELTA, iters, {"DVM_BA_FLOW": "0", "DVM_BA_NO_PAIR": "1", "DVM_BA_NO_WINDOW": "1", "DVM_BA_BORDER": "0"}, rounds=2)
    flo, info_f = _run(pr, DELTA, iters, {"DVM_BA_FLOW": "1", "DVM_BA_NO_WINDOW": "1", "DVM_BA_BORDER": "0"}, rounds=2)
    assert info_l["levels"] == info_f["levels"] and info_l["nz_tiles"] == info_f["nz_tiles"]
    _same_bits(lvl, flo, name)
    # the default level launches (top pair in one workgroup: another summation order in its 2-column solve): same trial sequence, 1e-9
    dfl, _ = _run(pr, DELTA, iters, {"DVM_BA_FLOW": "0", "DVM_BA_NO_WINDOW": "1", "DVM_BA_BORDER": "0"}, rounds=2)
    for (Pa, Xa, sa, _), (Pb, Xb, sb, _) in zip(dfl, flo):
        assert sa["trials"] == sb["trials"]
        assert np.abs(Pa - Pb).max() < 1e-9 and np.abs(Xa - Xb).max() < 1e-9


@pytest.mark.parametrize("name,delta", [("loop500", DELTA), ("web120", DELTA), ("web120", 0.0), ("ring87", DELTA)])
def test_flow_matches_oracle(oracle, name, delta):
    pr = synth.ba_problem(**PROBLEMS[name])
    iters = 3 if len(pr["poses"]) >= 500 else 8
    e = oracle.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    Po, Xo, so, chio = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, iters)
    (Pg, Xg, sg, chig), = _run(pr, delta, iters, {"DVM_BA_FLOW": "1", "DVM_BA_NO_WINDOW": "1", "DVM_BA_BORDER": "0"})[0]
    assert sg["iterations"] == so["iterations"] and sg["trials"] == so["trials"] and sg["stop_reason"] == so["stop_reason"]
    assert np.allclose(sg["chi2"], so["chi2"], rtol=1e-9)
    assert np.abs(Pg - Po).max() < 1e-6 and np.abs(Xg - Xo).max() < 1e-6, (np.abs(Pg - Po).max(), np.abs(Xg - Xo).max())
    assert np.allclose(chig, chio, rtol=1e-6, atol=1e-9)


def test_flow_failed_factorisation_follows_g2o(oracle):
    """A reduced system that is not positive definite at the first dampings: the trial is rejected with the stale x applied (g2o's
    failed-solve branch, linear_solver_eigen.h:89-112), in the flow form as in the level launches."""
    pr = synth.ba_problem(n_kf=40, n_pts=1200, seed=15)
    pr["inv_sigma2"] = pr["inv_sigma2"].copy()
    pr["inv_sigma2"][::7] = 1e300
    lvl, _ = _run(pr, 0.0, 3, {"DVM_BA_FLOW": "0", "DVM_BA_NO_PAIR": "1", "DVM_BA_NO_WINDOW": "1"})
    flo, _ = _run(pr, 0.0, 3, {"DVM_BA_FLOW": "1", "DVM_BA_NO_WINDOW": "1"})
    assert lvl[0][2]["trials"] == flo[0][2]["trials"]
    assert np.array_equal(_bits(lvl[0][0]), _bits(flo[0][0])) and np.array_equal(_bits(lvl[0][1]), _bits(flo[0][1]))


def test_flow_concurrent_handles_make_progress():
    """Two handles solving at once from two threads: the flow kernels share the compute units (neither is resident as a whole) and
    both finish -- a task only waits for tasks before it in its own list -- with the results of a solo run."""
    import threading
    pr = synth.ba_problem(n_kf=230, n_pts=6000, k_obs=4, seed=12)
    solo, _ = _run(pr, DELTA, 6, {"DVM_BA_FLOW": "1", "DVM_BA_NO_WINDOW": "1"})
    os.environ["DVM_BA_FLOW"] = "1"
    os.environ["DVM_BA_NO_WINDOW"] = "1"
    try:
        res = [None, None]

        def work(k):
            e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
            ba = capi.BundleAdjuster()
            ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], DELTA)
            for _ in range(3):
                ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], DELTA)
                st = ba.optimize(6)
            res[k] = (ba.result(), st)
            ba.close()

        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        for r in res:
            assert r is not None
            (P, X), st = r
            assert st["trials"] == solo[0][2]["trials"]
            assert np.array_equal(_bits(P), _bits(solo[0][0])) and np.array_equal(_bits(X), _bits(solo[0][1]))
    finally:
        os.environ.pop("DVM_BA_FLOW", None)
        os.environ.pop("DVM_BA_NO_WINDOW", None)


# ---------------------------------------------------------------------------------------------- kept landmarks ("border")
@pytest.mark.parametrize("name,delta", [("loop500", DELTA), ("loop500", 0.0), ("web120", DELTA)])
def test_kept_landmarks_same_solution_shorter_tree(oracle, name, delta):
    """A few landmarks seen from far apart make the elimination tree a chain (loop500: 40 of 20 000 landmarks, 37 levels).  Left out of
    the Schur complement and kept as unknowns of the reduced system (BaView::kept_*), the tree is a bush again -- and the solution is the
    same: against the run with DVM_BA_BORDER=0 and against the oracle within the accuracy contract, identical LM trial sequences."""
    pr = synth.ba_problem(**PROBLEMS[name])
    iters = 4 if len(pr["poses"]) >= 500 else 8
    (Pk, Xk, sk, chik), = _run(pr, delta, iters, {"DVM_BA_NO_WINDOW": "1"})[0]
    info_k = _info(pr, delta, {})
    (Pn, Xn, sn, chin), = _run(pr, delta, iters, {"DVM_BA_BORDER": "0", "DVM_BA_NO_WINDOW": "1"})[0]
    info_n = _info(pr, delta, {"DVM_BA_BORDER": "0"})
    assert info_k[1]["kept_landmarks"] > 0 and info_n[1]["kept_landmarks"] == 0
    assert info_k[0]["levels"] <= 0.75 * info_n[0]["levels"], (info_k, info_n)
    assert sk["trials"] == sn["trials"] and sk["iterations"] == sn["iterations"]
    assert np.abs(Pk - Pn).max() < 1e-7 and np.abs(Xk - Xn).max() < 1e-7, (np.abs(Pk - Pn).max(), np.abs(Xk - Xn).max())
    assert np.allclose(sk["chi2"], sn["chi2"], rtol=1e-10) and np.allclose(chik, chin, rtol=1e-6, atol=1e-9)
    e = oracle.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
    Po, Xo, so, _ = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta, iters)
    assert sk["trials"] == so["trials"] and np.abs(Pk - Po).max() < 1e-6 and np.abs(Xk - Xo).max() < 1e-6


def _info(pr, delta, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
        ba = capi.BundleAdjuster()
        ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], delta)
        out = (ba.schedule_info(), ba.solve_info())
        ba.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_ring_keeps_no_landmark():
    pr = synth.ba_problem()
    assert _info(pr, DELTA, {})[1]["kept_landmarks"] == 0


_BREAK_WORKER = """
import sys, numpy as np
sys.path.insert(0, %r)
from dvm_slam_amd import capi, synth
pr = synth.ba_problem(n_kf=230, n_pts=6000, k_obs=4, seed=12)
e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
ba = capi.BundleAdjuster()
ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], float(np.sqrt(5.991)))
form0 = ba.solve_info()["form"]
st = ba.optimize(4)
p, pts = ba.result()
np.savez(sys.argv[1], p=p, pts=pts, trials=np.array(st["trials"]), chi2=np.array(st["chi2"]), form=np.array([form0 == "flow", ba.solve_info()["form"] == "flow"]))
"""


def test_flow_wait_timeout_falls_back_to_the_level_launches(tmp_path):
    """Every wait inside k_chol_flow is bounded.  With the strips publishing a sequence number nobody waits for
    (DVM_BA_DEBUG_BREAK_FLOW) the waits give up, the trial is marked, and dvm_ba_optimize repeats it with one launch per phase -- for
    good: the run ends on exactly the bits of the per-phase form (the level launches with THEIR hand-offs broken)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (("levels", {"DVM_BA_FLOW": "0", "DVM_BA_DEBUG_BREAK_HANDOFF": "1"}), ("flow_broken", {"DVM_BA_FLOW": "1", "DVM_BA_DEBUG_BREAK_FLOW": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", _BREAK_WORKER % root, out], env={**os.environ, **extra}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert list(b["form"]) == [True, False] and list(a["form"]) == [False, False]       # the flow form was chosen, and given up
    assert np.array_equal(a["trials"], b["trials"]) and np.array_equal(a["chi2"], b["chi2"])
    assert np.array_equal(a["p"], b["p"]) and np.array_equal(a["pts"], b["pts"])


def test_kept_landmarks_with_edge_flags(oracle):
    """dvm_ba_set_edge_flags (level-1 edges leave the active set, others lose their robust kernel: Optimizer.cc:1317-1354) on a problem
    that keeps landmarks: an inactive edge's Hpl block is exact zeros in the border tiles too; kept and not kept agree."""
    pr = synth.ba_problem(**PROBLEMS["loop500"])
    rng = np.random.default_rng(3)
    E = len(pr["edge_pose"])
    flags = np.full(E, 3, np.uint8)
    flags[rng.random(E) < 0.05] = 0            # inactive
    flags[rng.random(E) < 0.10] &= 1           # active, not robust
    res = []
    for env in ({}, {"DVM_BA_BORDER": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            e = capi.make_edges(pr["edge_pose"], pr["edge_point"], pr["obs"], pr["inv_sigma2"])
            ba = capi.BundleAdjuster()
            ba.set_problem(pr["poses"], pr["fixed"], pr["points"], e, pr["intrinsics"], DELTA)
            kept = ba.solve_info()["kept_landmarks"]
            ba.optimize(2)
            ba.set_edge_flags(flags)
            st = ba.optimize(3)
            P, X = ba.result()
            chi, _ = ba.edge_chi2()
            ba.close()
            res.append((kept, st, P, X, chi))
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    (ka, sa, Pa, Xa, ca), (kb, sb, Pb, Xb, cb) = res
    assert ka > 0 and kb == 0 and sa["trials"] == sb["trials"]
    assert np.abs(Pa - Pb).max() < 1e-7 and np.abs(Xa - Xb).max() < 1e-7 and np.allclose(ca, cb, rtol=1e-6, atol=1e-9)
