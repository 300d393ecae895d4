"""dvm_slam_amd -- MI355X-native (gfx950) implementation of the DVM-SLAM per-agent hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI -> lib/libdvmslam_hip.so), capi.py (ctypes
binding of include/dvmslam_hip.h, mirroring the reference's ORBextractor / Frame grid / ORBmatcher /
Optimizer interfaces), synth.py (synthetic inputs).  No CPU implementation lives here.
"""
__all__ = ["capi", "synth"]
