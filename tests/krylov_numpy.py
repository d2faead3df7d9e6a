"""A NumPy stand-in for pyamg_amd.krylov's device-vector class (same method names): lets the Krylov drivers' host logic
(restart bookkeeping, breakdown branches, 1 x 1 systems) run without a GPU.  Handles are indices into a list of arrays."""
import numpy as np


class NumpyVectors(object):
    def __init__(self, A, M=None):
        self.Amat = A
        self.Mfun = M
        self.n = A.shape[0]
        self.store = []

    def new(self, count=None):
        self.store.append(np.zeros(self.n if count is None else count))
        return len(self.store) - 1

    def upload(self, host, dst=None):
        dst = self.new() if dst is None else dst
        self.store[dst][:] = np.ravel(host)
        return dst

    def download(self, src, count=None, offset=0):
        count = self.n if count is None else count
        return self.store[src][offset:offset + count].copy()

    def poke(self, dst, offset, values):
        v = np.atleast_1d(values)
        self.store[dst][offset:offset + len(v)] = v

    def peek(self, src, offset):
        return float(self.store[src][offset])

    def copy(self, dst, src, off=0):
        self.store[dst][off:] = self.store[src][off:]

    def fill(self, x, value, off=0):
        self.store[x][off:] = value

    def scale(self, out, x, c):
        self.store[out][:] = c * self.store[x]

    def axpy(self, y, a, x):
        self.store[y] += a * self.store[x]

    def xpby(self, p, beta, z):
        self.store[p][:] = beta * self.store[p] + self.store[z]

    def sub(self, out, a, b):
        self.store[out][:] = self.store[a] - self.store[b]

    def dot(self, x, y):
        return float(np.dot(self.store[x], self.store[y]))

    def norm(self, x, off=0):
        return float(np.linalg.norm(self.store[x][off:]))

    def A(self, x, out):
        self.store[out][:] = self.Amat @ self.store[x]

    def M(self, r, out):
        self.store[out][:] = self.store[r] if self.Mfun is None else self.Mfun(self.store[r])

    def residual(self, out, b, x, tmp):
        self.A(x, tmp)
        self.sub(out, b, tmp)
