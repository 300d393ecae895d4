"""Global bundle adjustment sharded over the ranks of a node (BASELINE.json config 5: "8 agents, global BA 500 keyframes /
20k landmarks, Schur-complement ... sharded over 8xMI355X"; reference Optimizer::GlobalBundleAdjustemnt,
src/Optimizer.cc:44-53, solves it on one CPU thread).

Every rank holds the whole problem and evaluates the observations of the landmarks it owns (landmark % world == rank);
per LM trial the partial reduced camera systems are summed with ONE all-reduce of the structurally non-zero 64x64 tiles
(~6 MB at 500 keyframes, RCCL over xGMI), every rank factors the sum redundantly (include/dvmslam_hip.h,
dvm_ba_set_problem_sharded).  This module is the torch.distributed side of that: the collective callback."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi


class ShardedBundleAdjuster:
    def __init__(self, device=0):
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
        self.ba.set_allreduce(self._allreduce, self.buf.data_ptr(), n)

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
