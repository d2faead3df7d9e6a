"""single-GPU resident solve vs DistributedSolver(world=1) on an own-setup hierarchy (GPU box)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
from pyamg_amd.distributed import DistributedSolver, HipBackend, levels_from_ml
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sm = sys.argv[2] if len(sys.argv) > 2 else "chebyshev"
A = poisson((n, n, n))
np.random.seed(0)
spec = ("chebyshev", {"degree": 2}) if sm == "chebyshev" else ("jacobi", {"omega": 4.0 / 3.0})
ml = smoothed_aggregation_solver(A, presmoother=spec, postsmoother=spec)
print(ml)
np.random.seed(0); b = np.random.rand(A.shape[0])
res = []
x = ml.solve(b, tol=0.0, maxiter=5, residuals=res)
levels, coarse = levels_from_ml(ml)
S = DistributedSolver(levels, coarse, HipBackend(0), 0, 1)
x2, res2 = S.solve(b, None, tol=0.0, maxiter=5)
print("resident   ", ["%.10e" % r for r in res])
print("distributed", ["%.10e" % r for r in res2])
print("x equal:", np.array_equal(x, x2), np.abs(x - x2).max())
# per-level operator check
dev = ml.device_hierarchy()
for l, L in enumerate(levels[:-1]):
    v = np.random.rand(L["A"].shape[0])
    y1 = dev.matvec(l, 0, v)
    y2 = L["A"] * v
    print("level", l, "A matvec vs scipy equal:", np.array_equal(y1, y2), type(L["A"]).__name__, L["A"].has_sorted_indices)
