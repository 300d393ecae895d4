"""The C-ABI libraries load and export every symbol include/dvmslam_hip.h / dvmslam_wire.h / dvmslam_host.h declare; without a GPU
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
    names = [n for n in _declared() if not n.startswith("dvm_exchange_")]      # (include/dvmslam_rccl.h: its own library, below)
    assert len(names) >= 55 and "dvm_wire_validate" in names and "dvm_bowdb_query" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (dvm_[a-z0-9_]+)", out))
    assert set(names) <= exported
    assert lib.dvm_version().decode().startswith("dvmslam-hip")


def test_rccl_library_exports_every_declared_symbol():
    """include/dvmslam_rccl.h is the C interface of libdvmslam_rccl.so (the inter-agent exchange for a C++ agent node): every declared
    dvm_exchange_* is exported and nothing else; the header is plain C; and libdvmslam_hip.so does NOT depend on librccl."""
    names = sorted(n for n in _declared() if n.startswith("dvm_exchange_"))
    assert len(names) >= 14 and "dvm_exchange_allreduce" in names and "dvm_exchange_allgather_varlen" in names
    so = os.path.join(ROOT, "dvm_slam_amd", "lib", "libdvmslam_rccl.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    assert set(names) == set(re.findall(r" T (dvm_[a-z0-9_]+)", out))
    need = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "librccl" in need
    need = subprocess.run(["readelf", "-d", os.path.join(ROOT, "dvm_slam_amd", "lib", "libdvmslam_hip.so")], capture_output=True, text=True).stdout
    assert "librccl" not in need
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "dvmslam_rccl.h")])


def _declared_host():
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dvmslam_host.h")).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(dvmh_[a-z0-9_]+)\s*\(", txt)))


def test_host_library_exports_every_declared_symbol(capi):
    """include/dvmslam_host.h is the C interface of libdvmslam_host.so: every declared dvmh_* entry point is exported, and nothing
    dvmh_* is exported that the header does not declare."""
    names = _declared_host()
    assert len(names) >= 30 and "dvmh_search_by_sim3" in names and "dvmh_kfdb_detect_merge_possibility" in names
    out = subprocess.run(["nm", "-D", "--defined-only", capi.HOST_LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (dvmh_[a-z0-9_]+)", out))
    assert set(names) == exported, (set(names) ^ exported)
    # the ctypes mirrors of the view structs in capi.py have the header's sizes (checked by compiling a C probe)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "dvmslam_host.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(dvmh_frame_view), '
                             'sizeof(dvmh_keyframe_view), sizeof(dvmh_map_points_view), sizeof(dvmh_feature_vector_view), sizeof(dvmh_map_point), '
                             'sizeof(dvmh_tracked_point)); return 0; }\n')
        exe = os.path.join(td, "sz")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(capi._FrameView), ctypes.sizeof(capi._KeyFrameView), ctypes.sizeof(capi._MapPointsView),
                     ctypes.sizeof(capi._FeatureVectorView), capi.MAP_POINT_DTYPE.itemsize, capi.TRACKED_POINT_DTYPE.itemsize], sizes


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
