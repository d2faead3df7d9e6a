import sys, os, numpy as np, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
n = int(sys.argv[1])
def build():
    A = poisson((n, n, n)); np.random.seed(0)
    spec = ("chebyshev", {"degree": 2})
    ml = smoothed_aggregation_solver(A, presmoother=spec, postsmoother=spec)
    out = []
    for l in ml.levels:
        h = hashlib.md5(np.ascontiguousarray(l.A.data).tobytes()).hexdigest()[:8]
        hj = hashlib.md5(np.ascontiguousarray(l.A.indices).tobytes()).hexdigest()[:8]
        out.append((l.A.shape[0], l.A.nnz, h, hj, getattr(l.A, "rho", None), getattr(l.A, "rho_D_inv", None),
                    l.presmoother.desc["coefficients"] if hasattr(l, "presmoother") else None))
    return out
a = build(); b = build()
for x, y in zip(a, b):
    print(x); print(y); print("same:", x == y)
