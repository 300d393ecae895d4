"""TEST INFRASTRUCTURE: DUtils::Random for the tests -- the reference's own object code (oracle/_ref/libdutils_ref.so, built
from Thirdparty/DBoW2/DUtils/Random.cpp by `make -C oracle _ref`) when it is there, a Python restatement otherwise; both draw
from libc's rand(), like the shim under test."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libdutils_ref.so")


class _Reference:
    def __init__(self, lib):
        self.lib = lib
        self._int = lib._ZN6DUtils6Random9RandomIntEii           # DUtils::Random::RandomInt(int, int)
        self._int.restype = ctypes.c_int
        self._int.argtypes = [ctypes.c_int, ctypes.c_int]
        self._seed = lib._ZN6DUtils6Random8SeedRandEi             # DUtils::Random::SeedRand(int)
        self._seed.argtypes = [ctypes.c_int]
        self._seed.restype = None

    def seed(self, s):
        self._seed(ctypes.c_int(s if s < 2 ** 31 else s - 2 ** 32))

    def random_int(self, lo, hi):
        return self._int(lo, hi)


def load_reference():
    """The reference's generator, or None when oracle/_ref was not built."""
    return _Reference(ctypes.CDLL(REF_SO)) if os.path.exists(REF_SO) else None


def python_random_int(libc, lo, hi):
    """Random.cpp:40-43 on libc's rand()."""
    return int((libc.rand() / (2147483647 + 1.0)) * (hi - lo + 1)) + lo


def draw_minimal_sets(random_int, n, count):
    """Sim3Solver.cc:166-181: three indices without replacement out of n, `count` times."""
    sets = []
    for _ in range(count):
        avail = list(range(n))
        trip = []
        for _i in range(3):
            r = random_int(0, len(avail) - 1)
            trip.append(avail[r])
            avail[r] = avail[-1]
            avail.pop()
        sets.append(trip)
    return sets
