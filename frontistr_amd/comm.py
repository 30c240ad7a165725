"""Host-side neighbour exchange over torch.distributed (gloo on CPU, or any initialised
backend): the transport behind fx_comm_set_host_callbacks and behind the comm hooks of the CPU test harness
in the multi-process tests.  Production multi-GPU runs use RCCL inside the library
(fx_comm_init); this path exists so that the decomposition logic (pack order, import/export
tables, reduction placement) is exercised on machines with a single GPU or none.

Mirrors hecmw_solve_send_recv_33 (hecmw_solver_SR_33.F90:42-124): one message per neighbour,
3 doubles per node, export_item order out, import_item order in.
"""
import ctypes as C

import numpy as np


class NeighborExchange:
    def __init__(self, neighbor_pe, import_index, export_index, ndof=3):
        import torch.distributed as dist
        self.dist = dist
        self.neighbor_pe = [int(p) for p in neighbor_pe]
        self.import_index = np.asarray(import_index, dtype=np.int64)
        self.export_index = np.asarray(export_index, dtype=np.int64)
        self.ndof = ndof
        self.n_import = int(self.import_index[-1])
        self.n_export = int(self.export_index[-1])

    def exchange(self, send, recv):
        """send: ndof*n_export doubles, recv: ndof*n_import doubles (numpy, contiguous)."""
        import torch
        d = self.ndof
        ops, rbufs = [], []
        for k, pe in enumerate(self.neighbor_pe):
            s0, s1 = d * self.export_index[k], d * self.export_index[k + 1]
            r0, r1 = d * self.import_index[k], d * self.import_index[k + 1]
            if s1 > s0:
                ops.append(self.dist.P2POp(self.dist.isend, torch.from_numpy(send[s0:s1].copy()), pe))
            if r1 > r0:
                t = torch.empty(r1 - r0, dtype=torch.float64)
                rbufs.append((r0, r1, t))
                ops.append(self.dist.P2POp(self.dist.irecv, t, pe))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for r0, r1, t in rbufs:
            recv[r0:r1] = t.numpy()

    def allreduce_sum(self, v):
        import torch
        t = torch.from_numpy(v)
        self.dist.all_reduce(t)          # in place on the numpy memory


def attach_host_callbacks(ctx, hecMESH, lib):
    """Install gloo-backed callbacks on a hecmw.SolverContext (keeps them alive on ctx)."""
    ex = NeighborExchange(hecMESH.neighbor_pe, hecMESH.import_index, hecMESH.export_index)
    HALO = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
    ARED = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_void_p)

    def _halo(send, recv, _u):
        s = np.ctypeslib.as_array(send, shape=(max(3 * ex.n_export, 1),))
        r = np.ctypeslib.as_array(recv, shape=(max(3 * ex.n_import, 1),))
        ex.exchange(s, r)

    def _ared(v, n, _u):
        ex.allreduce_sum(np.ctypeslib.as_array(v, shape=(n,)))

    ctx._cb = (HALO(_halo), ARED(_ared), ex)
    code = lib.fx_comm_set_host_callbacks(ctx.h, hecMESH.my_rank, hecMESH.PETOT, ctx._cb[0], ctx._cb[1], None)
    if code:
        raise RuntimeError("fx_comm_set_host_callbacks failed")
    return ex
