"""Synthetic structured hex meshes (SURVEY.md §8d generator).

Unit-spacing cube of n^3 C3D8 (HEC-MW TYPE=361) elements, (n+1)^3 nodes,
node id = 1 + i + (n+1)*(j + (n+1)*k), connectivity bottom face CCW then top
face.  Boundary conditions of the benchmark decks: all three dofs fixed on the
z=0 face, unit x-load on every z=n node.  Deterministic, no RNG.

All index arrays are int32 and 1-based, as in hecmwST_local_mesh
(hecmw_util_f.F90:232-381): ``elem_node_item`` is the flattened connectivity.
"""
import numpy as np


class CubeMesh:
    def __init__(self, n, spacing=1.0, skew=0.0):
        """n elements per edge.  ``skew`` > 0 perturbs interior nodes
        deterministically (to exercise non-trivial Jacobians in tests)."""
        self.n = int(n)
        m = self.n + 1
        self.n_node = m ** 3
        self.n_elem = self.n ** 3
        k, j, i = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
        xyz = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(np.float64) * spacing
        if skew:
            nid = np.arange(self.n_node, dtype=np.float64)
            interior = ((i > 0) & (i < self.n) & (j > 0) & (j < self.n) & (k > 0) & (k < self.n)).ravel()
            d = np.stack([np.sin(1.3 * nid + 0.1), np.sin(2.1 * nid + 0.7), np.sin(0.7 * nid + 1.9)], axis=1)
            xyz[interior] += skew * spacing * d[interior]
        self.coord = np.ascontiguousarray(xyz)                 # (n_node, 3)
        ek, ej, ei = np.meshgrid(np.arange(self.n), np.arange(self.n), np.arange(self.n), indexing="ij")
        n0 = (1 + ei + m * (ej + m * ek)).ravel()
        conn = np.stack([n0, n0 + 1, n0 + 1 + m, n0 + m,
                         n0 + m * m, n0 + 1 + m * m, n0 + 1 + m + m * m, n0 + m + m * m], axis=1)
        self.conn = np.ascontiguousarray(conn.astype(np.int32))  # (n_elem, 8), 1-based
        self.bottom_nodes = (1 + np.arange(m * m)).astype(np.int32)
        self.top_nodes = (1 + self.n * m * m + np.arange(m * m)).astype(np.int32)

    @property
    def ndof(self):
        return 3 * self.n_node

    def dirichlet(self):
        """(node, dof, value) triplets: z=0 face clamped."""
        node = np.repeat(self.bottom_nodes, 3).astype(np.int32)
        dof = np.tile(np.array([1, 2, 3], dtype=np.int32), self.bottom_nodes.size)
        val = np.zeros(node.size, dtype=np.float64)
        return node, dof, val

    def load(self):
        """!CLOAD 1.0 in x on every z=n node."""
        b = np.zeros(3 * self.n_node, dtype=np.float64)
        b[3 * (self.top_nodes - 1)] = 1.0
        return b


def cube_blocks(n):
    """Closed-form block counts of the n-element cube: (N, NPL, NPU, nb)."""
    m = n + 1
    N = m ** 3
    nb = (3 * m - 2) ** 3
    return N, (nb - N) // 2, (nb - N) // 2, nb
