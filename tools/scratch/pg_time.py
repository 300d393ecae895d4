import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvm_slam_amd import capi, synth
pg = synth.pose_graph(n=500, noise=0.002, seed=500)
capi.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=1)
for it in (1, 20):
    t0 = time.perf_counter()
    S, st = capi.pose_graph_optimize(pg["S0"], pg["fixed"], pg["edges_v"], pg["edges_meas"], iterations=it)
    print("iterations", it, "ms", (time.perf_counter() - t0) * 1e3, st.get("iterations"), {k: st[k] for k in st if k.startswith("ms")})
