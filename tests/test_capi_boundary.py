"""The C-ABI library loads and exports every symbol include/dvmslam_hip.h declares; without a GPU
every compute entry point fails loudly (DVM_ERR_NO_DEVICE) -- there is no CPU fallback in the product."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    import glob
    names = set()
    for path in glob.glob(os.path.join(ROOT, "include", "*.h")):     # dvmslam_hip.h, dvmslam_wire.h
        txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        names |= set(re.findall(r"\b(dvm_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    names = _declared()
    assert len(names) >= 55 and "dvm_wire_validate" in names and "dvm_bowdb_query" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (dvm_[a-z0-9_]+)", out))
    assert set(names) <= exported
    assert lib.dvm_version().decode().startswith("dvmslam-hip")


def test_struct_layouts(capi):
    assert capi.KP_DTYPE.itemsize == 28      # cv::KeyPoint
    assert capi.MATCH_DTYPE.itemsize == 16
    assert capi.BA_EDGE_DTYPE.itemsize == 32
    assert ctypes.sizeof(capi.OrbParams) == 20


def test_product_does_not_reference_oracle():
    """Nothing under dvm_slam_amd/ may import, link or load anything under oracle/ (the cpu_baseline legs live beside
    bench.py at the repo root: bench.py, ba_bench.py)."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "dvm_slam_amd")):
        for f in files:
            if not f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                continue
            txt = open(os.path.join(dp, f), errors="replace").read()
            if re.search(r"(from|import)\s+oracle|liboracle|oracle/", txt) and f not in ("__init__.py",):
                bad.append(os.path.join(dp, f))
    assert not bad, bad
    deps = subprocess.run(["readelf", "-d", os.path.join(ROOT, "dvm_slam_amd", "lib", "libdvmslam_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in deps


def test_fails_loudly_without_gpu(capi):
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.DvmError) as ei:
        capi.OrbExtractor()
    assert ei.value.code == -5
    with pytest.raises(capi.DvmError):
        capi.hamming_matrix(np.zeros((2, 32), np.uint8), np.zeros((2, 32), np.uint8))
    with pytest.raises(capi.DvmError):
        capi.BundleAdjuster()
    with pytest.raises(capi.DvmError):
        capi.FrameGrid()
    # the newer entry points behave the same: no device -> an error, never a computed result
    z32 = np.zeros((2, 32), np.uint8)
    with pytest.raises(capi.DvmError):
        capi.match_lists(z32, z32, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32))
    with pytest.raises(capi.DvmError):
        capi.distinctive_descriptors(z32, np.array([0, 2], np.int32))
    with pytest.raises(capi.DvmError):
        capi.bowdb_query_raw([(np.array([1, 2], np.int32), np.array([0.5, 0.5]))], np.array([1], np.int32), np.array([1.0]))
    with pytest.raises(capi.DvmError):
        capi.sim3_hypotheses(np.ones((4, 3), np.float32), np.ones((4, 3), np.float32), np.ones(4, np.float32), np.ones(4, np.float32),
                             np.ones(4, np.float32), np.ones(4, np.float32), np.array([[0, 1, 2]], np.int32))
    with pytest.raises(capi.DvmError):
        capi.pose_optimize(np.array([[0, 0, 0, 0, 0, 0, 1.0]]), np.ones((1, 4, 3)), np.ones((1, 4, 2)), np.ones((1, 4)), [4], [500, 500, 320, 240])
    from dvm_slam_amd import wire
    with pytest.raises(capi.DvmError):
        wire.gather_keypoints_device(64, 0, 1, 64, 1, 64, 32)     # never dereferenced: refused before any launch
    # the host mirrors sit on top of the same library: they fail with it
    with pytest.raises((capi.DvmError, AssertionError)):
        capi.HostKeyFrameDatabase()


def test_graft_entry_symbols():
    import __graft_entry__ as g
    assert g.exported_symbols() == _declared()


def test_headers_are_plain_c_and_link(tmp_path):
    """include/*.h are the boundary: they must compile as C99 and as C++11, and a C program must link against the library
    (the reference-side binding is C++ shim classes over exactly these declarations)."""
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "dvmslam_hip.h"\n#include "dvmslam_wire.h"\n'
                   'int main(void) {\n  dvm_wire_header h = {0};\n  dvm_wire_layout_t L;\n  h.n_keyframes = 2; h.n_keypoints = 5;\n'
                   '  if (dvm_wire_layout(&h, &L) != DVM_OK) return 2;\n  printf("%s %llu\\n", dvm_version(), (unsigned long long)L.total_bytes);\n'
                   '  return sizeof(dvm_keypoint) == 28 && sizeof(dvm_wire_keyframe) == 192 ? 0 : 3;\n}\n')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "dvm_slam_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{inc}", "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", f"-I{inc}", "-fsyntax-only", "-x", "c++", str(src)])
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", f"-I{inc}", str(src), "-o", str(exe), f"-L{libdir}", "-ldvmslam_hip", f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("dvmslam-hip") and out.stdout.split()[-1] == "960", (out.stdout, out.stderr)
    # the host C++ mirrors' headers stand on their own too
    host = os.path.join(ROOT, "dvm_slam_amd", "host")
    hdr = tmp_path / "host.cpp"
    hdr.write_text('#include "orb_matcher.h"\n#include "orb_vocabulary.h"\n#include "keyframe_database.h"\nint main() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", f"-I{inc}", f"-I{host}", "-fsyntax-only", str(hdr)])
