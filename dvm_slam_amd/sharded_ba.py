"""Global bundle adjustment sharded over the ranks of a node (BASELINE.json config 5: "8 agents, global BA 500 keyframes /
20k landmarks, Schur-complement ... sharded over 8xMI355X"; reference Optimizer::GlobalBundleAdjustemnt,
src/Optimizer.cc:44-53, solves it on one CPU thread).

Every rank holds the whole problem and evaluates the observations of the landmarks it owns (landmark % world == rank);
per LM trial the partial reduced camera systems are summed with ONE all-reduce of the structurally non-zero 64x64 tiles
(~6 MB at 500 keyframes, RCCL over xGMI), every rank factors the sum redundantly (include/dvmslam_hip.h,
dvm_ba_set_problem_sharded).  This module is the torch.distributed side of that: the collective callback.

native=True (or DVM_SHARDED_NATIVE=1): the callback is the C function dvm_exchange_allreduce of libdvmslam_rccl.so (include/dvmslam_rccl.h)
on a communicator of the library's own -- rank 0 draws the RCCL unique id, torch.distributed only carries it to the other ranks once --, so
that no Python and no torch.distributed call sits on the solver's per-trial path (what a C++ agent node does; needs one GPU per rank)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi


class ShardedBundleAdjuster:
    def __init__(self, device=0, native=None):
        self.native = (os.environ.get("DVM_SHARDED_NATIVE", "0") == "1") if native is None else bool(native)
        self._R = self._comm = self._ex = None
        self.ba = capi.BundleAdjuster(device)
        self.device = device
        self.buf = None
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on_gpu = dist.is_initialized() and dist.get_backend() == "nccl"
        self.bytes_reduced = 0
        self.calls = 0

    def set_problem(self, poses, fixed, points, edges, intrinsics, huber_delta):
        self.ba.set_problem_sharded(poses, fixed, points, edges, intrinsics, huber_delta, self.rank, self.world)
        n = self.ba.allreduce_doubles()
        self.buf = torch.zeros(n, dtype=torch.float64, device=f"cuda:{self.device}")
        if self.native:
            self._native_exchange()
            f = self.ba.L.dvm_ba_set_allreduce
            f.restype = C.c_int32; f.argtypes = None
            capi.check(f(self.ba.h, C.cast(self._R.dvm_exchange_allreduce, C.c_void_p), self._ex, C.c_void_p(self.buf.data_ptr()), C.c_int64(n)))
            return
        self.ba.set_allreduce(self._allreduce, self.buf.data_ptr(), n)

    def _native_exchange(self):
        """One RCCL communicator of libdvmslam_rccl.so over the ranks of the process group (made once per adjuster)."""
        if self._ex is not None:
            return
        R = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdvmslam_rccl.so"))
        R.dvm_exchange_last_error.restype = C.c_char_p

        def ok(rc):
            if rc != 0:
                raise RuntimeError("libdvmslam_rccl: " + R.dvm_exchange_last_error().decode())
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            ok(R.dvm_exchange_unique_id(uid))
        if self.world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
        comm, ex = C.c_void_p(), C.c_void_p()
        ok(R.dvm_exchange_comm_init(uid, self.rank, self.world, self.device, C.byref(comm)))
        self._ex_stream = torch.cuda.Stream(device=self.device)     # the exchange's own stream (the solver passes its stream per call)
        ok(R.dvm_exchange_create(comm, C.c_void_p(self._ex_stream.cuda_stream), C.byref(ex)))
        self._R, self._comm, self._ex = R, comm, ex

    def _allreduce(self, buf, n, on_host, op, stream):
        try:
            rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
            self.calls += 1
            self.bytes_reduced += 8 * n
            if not dist.is_initialized():   # no process group: a single rank, the sum over ranks is the buffer itself
                return 0
            if on_host:
                a = np.ctypeslib.as_array((C.c_double * n).from_address(buf))
                t = torch.from_numpy(a)
                if self.on_gpu:
                    g = t.cuda(self.device)
                    dist.all_reduce(g, op=rop)
                    t.copy_(g.cpu())
                else:
                    dist.all_reduce(t, op=rop)
                return 0
            assert buf == self.buf.data_ptr()
            t = self.buf[:n]
            if self.on_gpu:   # RCCL, ordered on the solver's stream: after its queued kernels, before the ones queued next
                with torch.cuda.device(self.device), torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                    dist.all_reduce(t, op=rop)
            else:             # gloo (tests: several ranks sharing one GPU): through the host
                torch.cuda.synchronize()
                c = t.cpu()
                dist.all_reduce(c, op=rop)
                t.copy_(c)
                torch.cuda.synchronize()
            return 0
        except Exception as ex:   # noqa: BLE001 -- a Python exception must not unwind through the C caller
            print("sharded BA all-reduce failed:", repr(ex), flush=True)
            return 1

    def optimize(self, iterations):
        return self.ba.optimize(iterations)

    def result(self):
        return self.ba.result()

    def close(self):
        self.ba.close()
        if self._ex is not None:
            self._R.dvm_exchange_destroy(self._ex)
            self._R.dvm_exchange_comm_destroy(self._comm)
            self._ex = self._comm = None
