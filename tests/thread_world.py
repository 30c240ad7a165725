"""Several HEC-MW subdomains in ONE process: one thread and one libfistr_hip context per subdomain, halo exchange and
reductions through the library's host-callback transport (fx_comm_set_host_callbacks) served by in-process mailboxes.
Used where a test needs more ranks than the GPU box allows processes on its card (8 subdomains = configs[3]); the
multi-process transports (gloo, RCCL) are covered by tests/test_gpu_distributed.py.  ctypes releases the GIL during the
library calls and re-acquires it for the callbacks, so the ranks really run concurrently."""
import ctypes as C
import threading

import numpy as np

HALO = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
ARED = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_void_p)


class ThreadWorld:
    def __init__(self, n, timeout=180.0):
        self.n = n
        self.barrier = threading.Barrier(n, timeout=timeout)
        self.mail = {}
        self.red = [None] * n
        self.errors = [None] * n

    def callbacks(self, rank, sub, ndof=3):
        nb = [int(p) for p in sub.neighbor_pe]
        ei = np.asarray(sub.export_index, dtype=np.int64) * ndof
        ii = np.asarray(sub.import_index, dtype=np.int64) * ndof

        def halo(send, recv, _u):
            s = np.ctypeslib.as_array(send, shape=(max(int(ei[-1]), 1),))
            r = np.ctypeslib.as_array(recv, shape=(max(int(ii[-1]), 1),))
            for q, pe in enumerate(nb):
                self.mail[(rank, pe)] = s[ei[q]:ei[q + 1]].copy()
            self.barrier.wait()
            for q, pe in enumerate(nb):
                r[ii[q]:ii[q + 1]] = self.mail[(pe, rank)]
            self.barrier.wait()

        def ared(v, n, _u):
            a = np.ctypeslib.as_array(v, shape=(n,))
            self.red[rank] = a.copy()
            self.barrier.wait()
            tot = np.zeros(n)
            for k in range(self.n):          # fixed order: every rank gets the same bits
                tot += self.red[k]
            self.barrier.wait()
            a[:] = tot

        return HALO(halo), ARED(ared)

    def run(self, fn):
        """fn(rank, world) in n threads; returns the list of results, re-raises the first failure."""
        out = [None] * self.n

        def body(r):
            try:
                out[r] = fn(r, self)
            except BaseException as e:  # noqa: BLE001
                self.errors[r] = e
                self.barrier.abort()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        real = [e for e in self.errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real or any(self.errors):
            raise (real or [e for e in self.errors if e is not None])[0]
        return out
