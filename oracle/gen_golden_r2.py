#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (development container only): round-2 additions to tests/golden/.

Like gen_golden.py this runs the REFERENCE itself (its Python staged by ref_env.py on its own native
module) and records numbers only:

  hier_accel_*.npz   multilevel_solver.solve(b, accel=...) histories: the reference's own Krylov
                     methods (pyamg/krylov/_cg.py, _fgmres.py, _gmres*.py, _bicgstab.py) preconditioned
                     with its own cycle (multilevel.py:381-422)
  hier_c5_*.npz      configuration C5's kind of operator on a REAL unstructured tetrahedral mesh: the
                     reference's gallery/example_data/unit_cube.mat (125 vertices, 384 tetrahedra; read
                     with scipy.io.loadmat -- data, not code), BSR 3x3, block Gauss-Seidel / block Jacobi
  mesh_unit_cube.npz the vertices / tetrahedra of that mesh (a data file of the reference's gallery)

Usage:  make -C oracle ref && python oracle/gen_golden_r2.py
"""
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_env  # noqa: E402
import gen_golden as gg  # noqa: E402


def main():
    pyamg = ref_env.stage()
    only_new = "--only-new" in sys.argv
    real_gen = gg.gen_hier

    def gen_if_missing(pyamg_, name, *a, **k):
        if only_new and os.path.exists(os.path.join(gg.OUT, "hier_%s.npz" % name)):
            return
        real_gen(pyamg_, name, *a, **k)
    gg.gen_hier = gen_if_missing
    P = ref_env.poisson
    sa = lambda mc: (lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=mc, **kw))
    jac = ("jacobi", {"omega": 4.0 / 3.0})
    sgs = ("gauss_seidel", {"sweep": "symmetric"})

    # ---- Krylov acceleration (multilevel.py:381-422)
    gg.gen_hier(pyamg, "accel_cg_jacobi_2d", P((48, 48)), sa(30), jac, jac, dict(tol=1e-10, maxiter=30, accel="cg"))
    gg.gen_hier(pyamg, "accel_cg_gs_3d", P((12, 12, 12)), sa(30), sgs, sgs, dict(tol=1e-9, maxiter=30, accel="cg"),
                x0_random=True)
    gg.gen_hier(pyamg, "accel_cg_W_2d", P((30, 30)), sa(10), jac, jac, dict(tol=1e-9, maxiter=30, accel="cg", cycle="W"))
    gg.gen_hier(pyamg, "accel_fgmres_amli_2d", P((24, 24)), sa(10), jac, jac,
                dict(tol=1e-9, maxiter=20, accel="fgmres", cycle="AMLI"))
    gg.gen_hier(pyamg, "accel_fgmres_V_2d", P((40, 40)), sa(30), ("gauss_seidel", {"sweep": "forward"}),
                ("gauss_seidel", {"sweep": "forward"}), dict(tol=1e-9, maxiter=25, accel="fgmres"), x0_random=True)
    gg.gen_hier(pyamg, "accel_gmres_V_2d", P((40, 40)), sa(30), ("gauss_seidel", {"sweep": "forward"}),
                ("gauss_seidel", {"sweep": "forward"}), dict(tol=1e-9, maxiter=25, accel="gmres"))
    gg.gen_hier(pyamg, "accel_bicgstab_F_2d", P((36, 36)), sa(20), jac, jac,
                dict(tol=1e-9, maxiter=25, accel="bicgstab", cycle="F"))

    # round 3: the remaining methods of pyamg.krylov as accelerators (krylov/_cr.py, _steepest_descent.py,
    # _minimal_residual.py)
    gg.gen_hier(pyamg, "accel_cr_V_2d", P((36, 36)), sa(20), jac, jac, dict(tol=1e-9, maxiter=25, accel="cr"))
    gg.gen_hier(pyamg, "accel_cr_gs_3d", P((11, 12, 13)), sa(30), sgs, sgs, dict(tol=1e-9, maxiter=30, accel="cr"),
                x0_random=True)
    gg.gen_hier(pyamg, "accel_steepest_descent_V_2d", P((32, 32)), sa(20), sgs, sgs,
                dict(tol=1e-9, maxiter=60, accel="steepest_descent"))
    gg.gen_hier(pyamg, "accel_minimal_residual_W_2d", P((30, 30)), sa(10), jac, jac,
                dict(tol=1e-9, maxiter=60, accel="minimal_residual", cycle="W"), x0_random=True)

    # ---- Krylov iterations as smoothers (smoothing.py:481-509; relaxation/tests/test_smoothing.py:25-30) and as
    #      coarse solvers (multilevel.py:642-660).  The C oracle has no Krylov methods: these are pinned by the
    #      reference's histories alone (file prefix krylov_).
    gg.gen_hier(pyamg, "krylov_gmres3_2d", P((30, 30)), sa(10), ("gmres", {"maxiter": 3}), ("gmres", {"maxiter": 3}),
                dict(tol=1e-9, maxiter=25))
    gg.gen_hier(pyamg, "krylov_cgnr_cgne_2d", P((30, 30)), sa(10), ("cgnr", {"maxiter": 2}), ("cgne", {"maxiter": 2}),
                dict(tol=1e-9, maxiter=25))
    gg.gen_hier(pyamg, "krylov_cg_2d", P((30, 30)), sa(10), None, ("cg", {"maxiter": 2}), dict(tol=1e-9, maxiter=25))

    # ---- C5 on the reference's own unstructured tetrahedral mesh
    d = scipy.io.loadmat("/root/reference/pyamg/gallery/example_data/unit_cube.mat")
    V = np.asarray(d["vertices"], dtype=np.float64)
    E = np.asarray(d["elements"], dtype=np.int64)
    np.savez_compressed(os.path.join(gg.OUT, "mesh_unit_cube.npz"), vertices=V, elements=E.astype(np.int32))

    # (i) the reference's own P1 elasticity assembly on that mesh: natural 3x3 blocks; the face x = 0 is
    #     clamped (whole vertices removed, so the blocks stay intact) to make the operator definite
    from pyamg.gallery import linear_elasticity_p1
    A, B = linear_elasticity_p1(V, E)
    keepv = np.nonzero(V[:, 0] > 1e-12)[0]
    keep = (3 * keepv[:, None] + np.arange(3)[None, :]).ravel()
    A = sps.csr_matrix(A)[keep][:, keep].tobsr((3, 3))
    A.sort_indices()
    B = np.asarray(B)[keep]
    bgs = ("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3})
    gg.gen_hier(pyamg, "c5_elas_p1_cube_bgs", A, sa(8), bgs, bgs, dict(tol=1e-10, maxiter=40), B=B)
    bjac = ("block_jacobi", {"omega": 1.0, "blocksize": 3})
    gg.gen_hier(pyamg, "c5_elas_p1_cube_bjac", A, sa(8), bjac, bjac, dict(tol=1e-8, maxiter=60), B=B)

    # (ii) anisotropic diffusion, P1 on the same tetrahedra (assembled by pyamg_amd.gallery.p1_diffusion:
    #      the operator is an INPUT and is stored in the fixture), two vertices pinned so that n = 123 is a
    #      multiple of 3, then A.tobsr((3, 3)) and block Gauss-Seidel with blocksize 3 as relaxation.py:562-563
    from pyamg_amd.gallery import anisotropy_tensor, p1_diffusion
    K = anisotropy_tensor((1.0, 0.1, 0.01), np.pi / 6, np.pi / 5)
    Ad = p1_diffusion(V, E, K)
    keep = np.setdiff1d(np.arange(V.shape[0]), [0, 124])
    Ad = sps.csr_matrix(Ad)[keep][:, keep].tobsr((3, 3))
    Ad.sort_indices()
    gg.gen_hier(pyamg, "c5_diff_p1_cube_bgs", Ad, sa(6), bgs, bgs, dict(tol=1e-10, maxiter=40))
    gg.gen_hier(pyamg, "c5_diff_p1_cube_bjac", Ad, sa(6), bjac, bjac, dict(tol=1e-8, maxiter=60))


if __name__ == "__main__":
    main()
