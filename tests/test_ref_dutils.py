"""Pins the one piece of the path whose REFERENCE SOURCE compiles in this image: DUtils::Random (Thirdparty/DBoW2/DUtils/
Random.cpp:19-43), the generator Sim3Solver::iterate draws its minimal sets with (Sim3Solver.cc:166-181).

oracle/_ref/libdutils_ref.so is the reference's Random.cpp + Timestamp.cpp compiled where they lie (`make -C oracle _ref`,
__graft_entry__.build()); it is git-ignored and travels to the GPU box as a built file.  Checked against it here:
  * the inline restatement in tests/stubs/Thirdparty/DBoW2/DUtils/Random.h (what test_shims_compile.py compiles the shims with),
  * the Python restatement tests/test_gpu_shims_run.py::test_sim3_solver_class uses when the library is absent,
  * the draw-without-replacement of a minimal set built on top of it.
Everything else of the oracle stays unpinned."""
import ctypes
import os
import subprocess

import pytest

from ref_dutils import REF_SO, draw_minimal_sets, load_reference, python_random_int

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libdutils_ref.so not built (needs /root/reference)")

RANGES = [(0, 0), (0, 1), (0, 2), (0, 49), (0, 499), (5, 5000), (-3, 3), (0, 99999), (0, 2 ** 30)]
SEEDS = [0, 1, 1234, 2 ** 31 - 1]


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("dutils_stub")
    src = d / "stub.cpp"
    src.write_text('#include "Thirdparty/DBoW2/DUtils/Random.h"\n'
                   'extern "C" int stub_random_int(int lo, int hi) { return DUtils::Random::RandomInt(lo, hi); }\n')
    so = d / "libstub.so"
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-I", os.path.join(ROOT, "tests", "stubs"), "-o", str(so), str(src)])
    return ctypes.CDLL(str(so))


def test_random_int_restatements_match_the_reference_object_code(stub_lib):
    ref = load_reference()
    libc = ctypes.CDLL("libc.so.6")
    for seed in SEEDS:
        for lo, hi in RANGES:
            ref.seed(seed)
            want = [ref.random_int(lo, hi) for _ in range(500)]
            assert min(want) >= lo and max(want) <= hi
            libc.srand(seed)
            assert [stub_lib.stub_random_int(lo, hi) for _ in range(500)] == want
            libc.srand(seed)
            assert [python_random_int(libc, lo, hi) for _ in range(500)] == want


def test_seed_rand_once_seeds_once():
    ref = load_reference()
    ref.lib._ZN6DUtils6Random12SeedRandOnceEi(77)         # first call seeds (unless an earlier test of this process did) ...
    ref.seed(5)
    a = [ref.random_int(0, 1000) for _ in range(20)]
    ref.seed(5)
    ref.lib._ZN6DUtils6Random12SeedRandOnceEi(99)         # ... a later one must not touch the stream
    assert [ref.random_int(0, 1000) for _ in range(20)] == a


def test_minimal_sets_with_reference_generator():
    ref = load_reference()
    libc = ctypes.CDLL("libc.so.6")
    for n in (3, 4, 20, 137):
        ref.seed(42)
        want = draw_minimal_sets(ref.random_int, n, 200)
        libc.srand(42)
        got = draw_minimal_sets(lambda lo, hi: python_random_int(libc, lo, hi), n, 200)
        assert got == want
        for t in want:
            assert len(set(t)) == 3 and all(0 <= i < n for i in t)
