"""TEST INFRASTRUCTURE: a CPU compute backend for pyamg_amd.distributed.DistributedSolver built
on the oracle's sequential kernels, so that partitioning, halo plans and exchange sequencing can
be exercised with gloo on machines without a GPU.  Never used by the product."""
import numpy as np
import torch

import oracle_lib
from oracle_lib import dp, ip
from pyamg_amd.distributed import (JACOBI, JACOBI_BSR1, MATVEC, MATVEC_ACC, POLY_FIRST, POLY_LAST, POLY_STEP,
                                   RESIDUAL)


class OracleBackend(object):
    def __init__(self):
        self.lib = oracle_lib.load()

    def vec(self, n):
        return torch.zeros(max(int(n), 1), dtype=torch.float64)

    def ivec(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))

    def from_host(self, t, a):
        t[:len(a)] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def to_host(self, t, n):
        return t[:n].numpy().copy()

    def mat(self, nrows, ncols, Ap, Aj, Ax):
        return (int(nrows), int(ncols), np.ascontiguousarray(Ap, dtype=np.intc),
                np.ascontiguousarray(Aj, dtype=np.intc), np.ascontiguousarray(Ax, dtype=np.float64))

    def apply(self, m, mode, xg, b, v2, out, out2, c0, gscale=1.0):
        n, nc, Ap, Aj, Ax = m
        xg_np = np.ascontiguousarray(gscale * xg.numpy()[:max(nc, 1)])
        if mode in (JACOBI, JACOBI_BSR1):
            x = xg_np.copy()
            temp = xg_np.copy()
            om = np.array([c0])
            bb = np.ascontiguousarray(b.numpy()[:max(n, 1)])
            if mode == JACOBI:
                self.lib.oracle_jacobi(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(bb), dp(temp), 0, n, 1, dp(om))
            else:
                self.lib.oracle_bsr_jacobi(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(bb), dp(temp), 0, n, 1, 1, dp(om))
            out[:n] = torch.from_numpy(x[:n])
            return
        s = np.zeros(max(n, 1))
        self.lib.oracle_csr_matvec(n, ip(Ap), ip(Aj), dp(Ax), dp(xg_np), dp(s))
        s = s[:n]
        if mode == MATVEC:
            out[:n] = torch.from_numpy(s)
        elif mode == MATVEC_ACC:
            out[:n] = torch.from_numpy(out.numpy()[:n] + s)
        elif mode == RESIDUAL:
            out[:n] = torch.from_numpy(b.numpy()[:n] - s)
        elif mode == POLY_FIRST:
            r = b.numpy()[:n] - s
            out[:n] = torch.from_numpy(r)
            out2[:n] = torch.from_numpy(c0 * r)
        elif mode == POLY_STEP:
            out[:n] = torch.from_numpy(c0 * b.numpy()[:n] + s)
        elif mode == POLY_LAST:
            h = c0 * b.numpy()[:n] + s
            out[:n] = torch.from_numpy(v2.numpy()[:n] + h)
        else:
            raise ValueError(mode)

    def build_gs(self, m, order):
        self._gs_order = getattr(self, "_gs_order", {})
        self._gs_order[id(m)] = None if order is None else np.ascontiguousarray(order, dtype=np.intc)

    def gs_sweep(self, m, x, b, reverse, bsr1):
        n, nc, Ap, Aj, Ax = m
        order = self._gs_order[id(m)]
        xe = np.ascontiguousarray(x.numpy()[:max(nc, 1)]).copy()
        bb = np.ascontiguousarray(b.numpy()[:max(n, 1)])
        if order is None:
            rs, re, rt = (n - 1, -1, -1) if reverse else (0, n, 1)
            if bsr1:
                self.lib.oracle_bsr_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(xe), dp(bb), rs, re, rt, 1)
            else:
                self.lib.oracle_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(xe), dp(bb), rs, re, rt)
        else:
            m_ = len(order)
            rs, re, rt = (m_ - 1, -1, -1) if reverse else (0, m_, 1)
            self.lib.oracle_gauss_seidel_indexed(ip(Ap), ip(Aj), dp(Ax), dp(xe), dp(bb), ip(order), rs, re, rt)
        x[:n] = torch.from_numpy(xe[:n])

    def scale(self, out, inp, c, n):
        out[:n] = torch.from_numpy(c * inp.numpy()[:n])

    def apply_rows(self, m, mode, lo, hi, xg, b, v2, out, out2, c0, gscale=1.0):
        if hi <= lo:
            return
        n = m[0]
        full = torch.zeros(max(n, 1), dtype=torch.float64)
        full2 = torch.zeros(max(n, 1), dtype=torch.float64)
        if mode in (MATVEC_ACC,):
            full[:n] = out[:n]
        v2c = None if v2 is None else v2.clone()      # in-place modes: read the old values
        self.apply(m, mode, xg, b, v2c, full, full2, c0, gscale)
        out[lo:hi] = full[lo:hi]
        if out2 is not None:
            out2[lo:hi] = full2[lo:hi]

    def axpy_scaled(self, x, r, c, n):
        x[:n] = torch.from_numpy(x.numpy()[:n] + c * r.numpy()[:n])

    def axpy(self, x, h, n):
        x[:n] = torch.from_numpy(x.numpy()[:n] + h.numpy()[:n])

    def gather(self, out, inp, idx, n):
        out[:n] = inp[idx[:n]]

    def sumsq(self, x, n, out):
        v = np.ascontiguousarray(x.numpy()[:n])
        nn = self.lib.oracle_norm2(dp(v), n) if n else 0.0
        out[0] = nn * nn

    def dense(self, Mt, b, x, n):
        M = Mt.numpy()[:n * n].reshape(n, n).T
        bb = b.numpy()[:n]
        for i in range(n):
            s = 0.0
            for k in range(n):
                s += M[i, k] * bb[k]
            x[i] = s

    def zero(self, t, n):
        t[:n] = 0.0

    def synchronize(self):
        pass
