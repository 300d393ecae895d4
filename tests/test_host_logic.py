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
