"""Host-side logic of the product that can be checked without a GPU."""
import os
import subprocess

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
    out = subprocess.run([str(exe), "499", "7", "1"], capture_output=True, text=True)
    levels = int(out.stdout.split("levels=")[1].split()[0])
    assert levels <= 16, out.stdout   # 51 tile columns, one after the other, before the reordering
