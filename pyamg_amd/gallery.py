"""Synthetic inputs for the BASELINE configurations (the reference's gallery is out of scope; its
largest shipped mesh has 2880 unknowns).

tet_diffusion: P1 finite elements for -div(K grad u) on a jittered structured hexahedral grid whose
cubes are split into 6 tetrahedra (Kuhn split), K = Q diag(eps) Q^T a rotated anisotropic tensor
(the 3D analogue of pyamg/gallery/diffusion.py:188 diffusion_stencil_3d), Dirichlet boundary removed.
Configuration C5 takes this matrix as BSR with 3x3 blocks (n % 3 == 0).
"""
import itertools

import numpy as np
import scipy.sparse as sps

__all__ = ["tet_diffusion", "p1_diffusion", "kuhn_mesh", "anisotropy_tensor", "poisson"]

from .aggregation import poisson  # noqa: E402,F401


def _rotation(theta, phi):
    cz, sz = np.cos(theta), np.sin(theta)
    cy, sy = np.cos(phi), np.sin(phi)
    Rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
    return Rz.dot(Ry)


def anisotropy_tensor(eps=(1.0, 0.1, 0.01), theta=np.pi / 6, phi=np.pi / 5):
    """K = Q diag(eps) Q^T with Q the rotation about z by theta, then about y by phi"""
    Q = _rotation(theta, phi)
    return Q.dot(np.diag(eps)).dot(Q.T)


def p1_diffusion(vertices, elements, K, chunk=2000000):
    """Stiffness matrix of -div(K grad u) with P1 elements on the tetrahedra `elements` (rows of four
    vertex ids) over `vertices` (n x 3): per element the barycentric gradients G_i, entries
    |T| (K G_i).G_j, summed over the elements that share an edge.  CSR, sorted int32 indices."""
    V = np.asarray(vertices, dtype=np.float64)
    T = np.asarray(elements, dtype=np.int64)
    nv = V.shape[0]
    K = np.asarray(K, dtype=np.float64)
    A = None
    for lo in range(0, T.shape[0], chunk):                     # bounded temporaries on large meshes
        Tc = T[lo:lo + chunk]
        X = V[Tc]                                              # (ne, 4, 3)
        E = X[:, 1:, :] - X[:, :1, :]
        vol = np.abs(np.linalg.det(E)) / 6.0
        G = np.zeros((Tc.shape[0], 4, 3))
        G[:, 1:, :] = np.transpose(np.linalg.inv(E), (0, 2, 1))     # columns of E^-1 = gradients of lambda_1..3
        G[:, 0, :] = -G[:, 1:, :].sum(axis=1)
        loc = np.einsum("eik,ejk->eij", G.dot(K.T), G) * vol[:, None, None]
        part = sps.coo_matrix((loc.ravel(), (np.repeat(Tc, 4, axis=1).ravel(), np.tile(Tc, (1, 4)).ravel())),
                              shape=(nv, nv)).tocsr()
        A = part if A is None else A + part
    A.sum_duplicates()
    A.sort_indices()
    A.indices = A.indices.astype(np.intc)
    A.indptr = A.indptr.astype(np.intc)
    return A


def kuhn_mesh(n, jitter=0.2, seed=0):
    """(n+2)^3 vertices of the unit cube on a jittered grid (boundary vertices stay put), every cube cut into
    6 tetrahedra along the diagonal (0,0,0)-(1,1,1) (Kuhn split: one per ordering of the axes).
    -> vertices (nv x 3), elements (ne x 4), interior vertex ids (lexicographic, last axis fastest)"""
    m = n + 2
    rng = np.random.RandomState(seed)
    g = np.linspace(0.0, 1.0, m)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    P = np.stack([X, Y, Z], axis=-1)
    h = 1.0 / (m - 1)
    J = (rng.rand(m, m, m, 3) - 0.5) * (2.0 * jitter * h)
    J[0], J[-1] = 0.0, 0.0
    J[:, 0], J[:, -1] = 0.0, 0.0
    J[:, :, 0], J[:, :, -1] = 0.0, 0.0
    P = (P + J).reshape(-1, 3)
    vid = np.arange(m ** 3).reshape(m, m, m)
    corner = {}
    for dx, dy, dz in itertools.product((0, 1), repeat=3):
        corner[(dx, dy, dz)] = vid[dx:m - 1 + dx, dy:m - 1 + dy, dz:m - 1 + dz].ravel()
    tets = []
    for perm in itertools.permutations(range(3)):
        path, cur = [(0, 0, 0)], [0, 0, 0]
        for ax in perm:
            cur = list(cur)
            cur[ax] = 1
            path.append(tuple(cur))
        tets.append(np.stack([corner[p] for p in path], axis=1))
    return P, np.concatenate(tets, axis=0), vid[1:-1, 1:-1, 1:-1].ravel()


def tet_diffusion(n, eps=(1.0, 0.1, 0.01), theta=np.pi / 6, phi=np.pi / 5, jitter=0.2, seed=0, blocksize=None, native=None):
    """Stiffness matrix of -div(K grad u), P1 on tetrahedra, (n+2)^3 vertices with the boundary layer
    eliminated -> n^3 unknowns (lexicographic, last axis fastest).  Returns CSR, or BSR(bs,bs) when
    `blocksize` is given (n^3 must be divisible by it), as configuration C5 uses it.
    native: assemble row by row in csrc/setup_host.cpp (OpenMP; default above 40^3 unknowns -- the element-list
    assembly needs ~400 bytes of temporaries per element) instead of from the element list; same operator to
    rounding."""
    if native is None:
        native = n > 40
    K = anisotropy_tensor(eps, theta, phi)
    if native:
        A = _kuhn_native(n, K, jitter, seed)
    else:
        P, T, interior = kuhn_mesh(n, jitter, seed)
        A = p1_diffusion(P, T, K)
        A = A[interior][:, interior].tocsr()
        A.sum_duplicates()
        A.sort_indices()
        A.indices = A.indices.astype(np.intc)
        A.indptr = A.indptr.astype(np.intc)
    if blocksize:
        if A.shape[0] % blocksize:
            raise ValueError("n^3 must be divisible by the blocksize")
        # ~15 blocks per block row: above 2^31 stored values the int32 addressing of the BSR data array ends -- the
        # reference's kernels index Ax[jj * blocksize * blocksize] with int (relaxation.h:90-173, :756-810) and scipy's
        # csr_tobsr crashes; 360^3 = 46.7 M unknowns is the largest cube divisible by 3 below that (369^3 = 50.2 M is not)
        if (A.shape[0] // blocksize) * 15.0 * blocksize * blocksize >= 2.0 ** 31:      # this mesh: <= 15 blocks per block row
            raise ValueError("BSR(%d,%d) form of %d unknowns needs more than 2^31 stored values (int32 addressing of the "
                             "block data array, as in the reference's amg_core): use n <= 360" % (blocksize, blocksize, A.shape[0]))
        A = _tobsr_sorted(A, blocksize) if native else A.tobsr((blocksize, blocksize))
    return A


def _jittered_vertices(n, jitter, seed):
    m = n + 2
    rng = np.random.RandomState(seed)
    g = np.linspace(0.0, 1.0, m)
    h = 1.0 / (m - 1)
    J = (rng.rand(m, m, m, 3) - 0.5) * (2.0 * jitter * h)
    J[0], J[-1] = 0.0, 0.0
    J[:, 0], J[:, -1] = 0.0, 0.0
    J[:, :, 0], J[:, :, -1] = 0.0, 0.0
    J[..., 0] += g[:, None, None]
    J[..., 1] += g[None, :, None]
    J[..., 2] += g[None, None, :]
    return J.reshape(-1, 3)


def _kuhn_native(n, K, jitter, seed):
    import ctypes as C
    from .aggregation import _dp, _ip, _lp, host_lib
    L = host_lib()
    m = n + 2
    xyz = np.ascontiguousarray(_jittered_vertices(n, jitter, seed))
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    N = n ** 3
    Ap = np.empty(N + 1, dtype=np.int64)
    nnz = L.amgsetup_kuhn_p1_diffusion(m, _dp(xyz), _dp(Kc), _lp(Ap), C.POINTER(C.c_int)(), C.POINTER(C.c_double)())
    if nnz >= 2 ** 31:
        raise ValueError("matrix exceeds int32 indices")
    Aj = np.empty(nnz, dtype=np.intc)
    Ax = np.empty(nnz, dtype=np.float64)
    L.amgsetup_kuhn_p1_diffusion(m, _dp(xyz), _dp(Kc), _lp(Ap), _ip(Aj), _dp(Ax))
    A = sps.csr_matrix((Ax, Aj, Ap.astype(np.intc)), shape=(N, N))
    A.has_sorted_indices = True
    return A


def _tobsr_sorted(A, bs):
    """A.tobsr((bs, bs)) for a CSR matrix with sorted rows (scipy's csr_tobsr: block columns ascending, absent
    entries of a touched block zero) without scipy's O(n_bcol) scratch per call being the bottleneck"""
    return A.tobsr((bs, bs))
