"""Host-side logic of the product that can be checked without a GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_introsort_emulation_matches_libstdcxx(tmp_path):
    """dvm_slam_amd/csrc/introsort_emul.h (used by the device octree) == the real std::sort, including the
    order of equal keys, for the serial form and for the split form (introsort loop + stable rank)."""
    exe = tmp_path / "check_introsort"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "check_introsort.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical to std::sort" in out.stdout


def test_ba_ordering_and_level_schedule(tmp_path):
    """dvm_slam_amd/csrc/ba_ordering.cpp: the camera order is a permutation, the level schedule of the tile Cholesky
    respects every dependency, and nested dissection shortens the chain of a loop trajectory (tools/check_ba_ordering.cpp)."""
    exe = tmp_path / "check_ba_ordering"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "check_ba_ordering.cpp"),
                           os.path.join(ROOT, "dvm_slam_amd", "csrc", "ba_ordering.cpp")])
    for args in (["499", "7", "1"], ["499", "7", "0"], ["37", "3", "1"], ["2000", "12", "1"], ["5", "2", "0"], ["1", "1", "0"],
                 ["10", "9", "0"], ["11", "2", "1"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True)
        assert out.returncode == 0, (args, out.stdout + out.stderr)
    # a deep tree (the loop-closed map of bench.py's second BA workload: 0.2 % of the landmarks seen from far apart): the flow form's
    # task list and chains are executed symbolically by the tool -- no circular wait, every tile produced once, contributor lists = the
    # level schedule's
    import itertools
    import numpy as np
    from dvm_slam_amd import synth
    pr = synth.ba_problem(laps=2, long_range_frac=0.002)
    free = {int(p): i for i, p in enumerate(np.nonzero(pr["fixed"] == 0)[0])}
    cams = [[] for _ in range(len(pr["points"]))]
    for p_, l_ in zip(pr["edge_pose"], pr["edge_point"]):
        if int(p_) in free:
            cams[int(l_)].append(free[int(p_)])
    pairs = set()
    for c in cams:
        pairs.update(itertools.combinations(sorted(set(c)), 2))
    adj = tmp_path / "adj.txt"
    adj.write_text(f"{len(free)}\n" + "".join(f"{a} {b}\n" for a, b in sorted(pairs)))
    out = subprocess.run([str(exe), "0", "0", "0", str(adj)], capture_output=True, text=True)
    assert out.returncode == 0 and re.search(r"levels=3[0-9]\b", out.stdout), out.stdout + out.stderr          # 37 or 38 levels
    def levels(n, loop):
        out = subprocess.run([str(exe), str(n), "7", loop], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        return int(out.stdout.split("levels=")[1].split()[0])
    # a ring of 50 tiles: 7 levels of tile columns (two paths of 24 -> 5, the two-tile root separator -> 2) + the rhs tile;
    # 51 tile columns one after the other before the reordering
    assert levels(500, "1") == 8
    # a partially filled tile may only be the LAST of the elimination order: the tiling is chosen so that the short tile lies in
    # the dissection's root separator -- it must not cost a level (it did: 9 at 499 cameras, the bench problem).  A short tile
    # of fewer cameras than the co-visibility span cannot separate its neighbours, there the extra level stays.
    for n in (497, 499, 507):
        assert levels(n, "1") == 8, n
        assert levels(n, "0") == 7, n
    for n in (491, 495, 501):
        assert levels(n, "1") <= 9, n


def test_generated_cholesky_panel_is_in_sync(tmp_path):
    """csrc/chol_panel.inc (the straight-line 16-column panel of k_chol_diag) is generated: the committed file must be what
    tools/gen_chol_panel.py writes, and the stream must hold the whole factorisation -- 16 reciprocal square roots, the 15
    immediate and 105 deferred rank-1 updates (a[c] -= L_.p L_cp for every p < c < 16), every column published once."""
    import re
    out = tmp_path / "panel.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_chol_panel.py"), "--out", str(out)], check=True, capture_output=True)
    text = out.read_text()
    assert text == open(os.path.join(ROOT, "dvm_slam_amd", "csrc", "chol_panel.inc")).read()
    assert text.count("__builtin_amdgcn_rsq(") == 16
    upd = re.findall(r"a\[(\d+)\] = __builtin_fma\(-a\[(\d+)\], (n\d+|p\d+_\d+(?:\.[xy])?), a\[\1\]\);", text)
    assert len(upd) == 120 and sorted((int(c), int(p)) for c, p, _ in upd) == sorted((c, p) for c in range(16) for p in range(c))
    for c, p, src in upd:                                   # the broadcast an update reads is the one of ITS column pair
        if src.startswith("p"):
            m = re.match(r"p(\d+)_(\d+)(?:\.([xy]))?", src)
            assert int(m.group(1)) == int(p) and int(m.group(2)) + (1 if m.group(3) == "y" else 0) == int(c)
        else:
            assert src == f"n{p}" and int(c) == int(p) + 1
    # per target column the updates appear in source-column order (the rolled loop's summation order)
    for c in range(16):
        ps = [int(p) for cc, p, _ in upd if int(cc) == c]
        assert ps == sorted(ps)
    assert len(re.findall(r"Pcol\[\d+\]\[lane\] = a\[\d+\];", text)) == 16
