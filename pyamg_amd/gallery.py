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

__all__ = ["tet_diffusion", "poisson"]

from .aggregation import poisson  # noqa: E402,F401


def _rotation(theta, phi):
    cz, sz = np.cos(theta), np.sin(theta)
    cy, sy = np.cos(phi), np.sin(phi)
    Rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
    return Rz.dot(Ry)


def tet_diffusion(n, eps=(1.0, 0.1, 0.01), theta=np.pi / 6, phi=np.pi / 5, jitter=0.2, seed=0, blocksize=None):
    """Stiffness matrix of -div(K grad u), P1 on tetrahedra, (n+2)^3 vertices with the boundary layer
    eliminated -> n^3 unknowns (lexicographic, last axis fastest).  Returns CSR, or BSR(bs,bs) when
    `blocksize` is given (n^3 must be divisible by it), as configuration C5 uses it."""
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
    Q = _rotation(theta, phi)
    K = Q.dot(np.diag(eps)).dot(Q.T)
    vid = np.arange(m ** 3).reshape(m, m, m)
    c0 = vid[:-1, :-1, :-1].ravel()
    corner = {}
    for dx, dy, dz in itertools.product((0, 1), repeat=3):
        corner[(dx, dy, dz)] = vid[dx:m - 1 + dx, dy:m - 1 + dy, dz:m - 1 + dz].ravel()
    rows, cols, vals = [], [], []
    # Kuhn split: one tetrahedron per ordering of the axes, path (0,0,0) -> (1,1,1)
    for perm in itertools.permutations(range(3)):
        path = [(0, 0, 0)]
        cur = [0, 0, 0]
        for ax in perm:
            cur = list(cur)
            cur[ax] = 1
            path.append(tuple(cur))
        T = np.stack([corner[p] for p in path], axis=1)            # (ncubes, 4) vertex ids
        V = P[T]                                                   # (ncubes, 4, 3)
        E = V[:, 1:, :] - V[:, :1, :]                              # edge matrix rows
        det = np.linalg.det(E)
        Einv = np.linalg.inv(E)                                    # columns = gradients of lambda_1..3
        G = np.zeros((T.shape[0], 4, 3))
        G[:, 1:, :] = np.transpose(Einv, (0, 2, 1))
        G[:, 0, :] = -G[:, 1:, :].sum(axis=1)
        vol = np.abs(det) / 6.0
        KG = G.dot(K.T)                                            # (ncubes, 4, 3)
        loc = np.einsum("eik,ejk->eij", KG, G) * vol[:, None, None]
        rows.append(np.repeat(T, 4, axis=1).ravel())
        cols.append(np.tile(T, (1, 4)).ravel())
        vals.append(loc.ravel())
        del c0
        c0 = None
    A = sps.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m ** 3, m ** 3)).tocsr()
    interior = vid[1:-1, 1:-1, 1:-1].ravel()
    A = A[interior][:, interior].tocsr()
    A.sum_duplicates()
    A.sort_indices()
    A.indices = A.indices.astype(np.intc)
    A.indptr = A.indptr.astype(np.intc)
    if blocksize:
        if A.shape[0] % blocksize:
            raise ValueError("n^3 must be divisible by the blocksize")
        A = A.tobsr((blocksize, blocksize))
    return A
