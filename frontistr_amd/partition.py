"""Node-based overlapping domain decomposition of the synthetic cube (host logic).

Produces, for one rank, exactly what the reference's partitioner (hecmw_part1, depth 1,
node-based) writes into a HECMW-DIST file and what hecmwST_local_mesh carries
(hecmw_util_f.F90:298-310, file layout hecmw_io_dist.c:2192-2278):

  * internal nodes numbered 1..nn_internal first, halo nodes nn_internal+1..n_node after,
    grouped by owning neighbour rank (hecmw_mat_con.f90:59-60 relies on this);
  * every element that touches an internal node (so each internal matrix row is complete
    locally: no assembly communication);
  * neighbor_pe / import_index,item / export_index,item with matching order on both sides
    (ascending global node id inside each neighbour's group).

Weak scaling: rank (rx,ry,rz) of a px*py*pz grid owns m^3 nodes; the global cube has
(px*m, py*m, pz*m) nodes.  Boundary conditions of the benchmark deck are global: z = 0
clamped, unit x-load on the global top face.
"""
import numpy as np


class CubeSubdomain:
    def __init__(self, m, dims, rank):
        px, py, pz = dims
        self.m, self.dims, self.rank = m, dims, rank
        self.G = (px * m, py * m, pz * m)                      # global node counts
        rx, ry, rz = rank % px, (rank // px) % py, rank // (px * py)
        self.lo = np.array([rx * m, ry * m, rz * m])
        self.hi = self.lo + m                                   # owned box [lo, hi)
        G = np.array(self.G)
        elo = np.maximum(self.lo - 1, 0)                        # element range touching an owned node
        ehi = np.minimum(self.hi, G - 1)                        # exclusive
        nlo, nhi = elo, ehi + 1                                 # node box covered by those elements
        ii, jj, kk = [np.arange(nlo[d], nhi[d]) for d in range(3)]
        K, J, I = np.meshgrid(kk, jj, ii, indexing="ij")
        gi, gj, gk = I.ravel(), J.ravel(), K.ravel()
        gid = gi + self.G[0] * (gj + self.G[1] * gk)            # global node id (0-based)
        owned = ((gi >= self.lo[0]) & (gi < self.hi[0]) & (gj >= self.lo[1]) & (gj < self.hi[1]) &
                 (gk >= self.lo[2]) & (gk < self.hi[2]))
        owner = (gi // m) + px * ((gj // m) + py * (gk // m))
        # local numbering: internal in natural order, halo grouped by owner then global id
        int_idx = np.nonzero(owned)[0]
        halo_idx = np.nonzero(~owned)[0]
        order = np.lexsort((gid[halo_idx], owner[halo_idx]))
        halo_idx = halo_idx[order]
        perm = np.concatenate([int_idx, halo_idx])              # box index of local node l
        self.nn_internal = int(int_idx.size)
        self.n_node = int(perm.size)
        self.global_id = gid[perm].astype(np.int64)
        self.gijk = np.stack([gi[perm], gj[perm], gk[perm]], axis=1)
        self.coord = self.gijk.astype(np.float64)
        local_of_box = np.empty(gid.size, dtype=np.int64)
        local_of_box[perm] = np.arange(perm.size)
        # elements
        bx, by = nhi[0] - nlo[0], nhi[1] - nlo[1]
        ei, ej, ek = [np.arange(elo[d], ehi[d]) for d in range(3)]
        EK, EJ, EI = np.meshgrid(ek, ej, ei, indexing="ij")
        b0 = (EI.ravel() - nlo[0]) + bx * ((EJ.ravel() - nlo[1]) + by * (EK.ravel() - nlo[2]))
        conn_box = np.stack([b0, b0 + 1, b0 + 1 + bx, b0 + bx,
                             b0 + bx * by, b0 + 1 + bx * by, b0 + 1 + bx + bx * by, b0 + bx + bx * by], axis=1)
        self.conn = np.ascontiguousarray((local_of_box[conn_box] + 1).astype(np.int32))
        self.n_elem = int(self.conn.shape[0])
        # communication tables
        h_owner = owner[halo_idx]
        self.neighbor_pe = np.unique(h_owner).astype(np.int32)
        self.import_index = np.zeros(self.neighbor_pe.size + 1, dtype=np.int32)
        for q, pe in enumerate(self.neighbor_pe):
            self.import_index[q + 1] = self.import_index[q] + int((h_owner == pe).sum())
        self.import_item = (self.nn_internal + 1 + np.arange(halo_idx.size)).astype(np.int32)
        # export: my owned nodes that lie in neighbour q's extended box, ascending global id
        exp_items, self.export_index = [], np.zeros(self.neighbor_pe.size + 1, dtype=np.int32)
        ogi, ogj, ogk = self.gijk[:self.nn_internal].T
        ogid = self.global_id[:self.nn_internal]
        for q, pe in enumerate(self.neighbor_pe):
            qx, qy, qz = pe % px, (pe // px) % py, pe // (px * py)
            qlo = np.array([qx * m, qy * m, qz * m])
            qhi = qlo + m
            qelo = np.maximum(qlo - 1, 0)
            qehi = np.minimum(qhi, G - 1)
            sel = ((ogi >= qelo[0]) & (ogi <= qehi[0]) & (ogj >= qelo[1]) & (ogj <= qehi[1]) &
                   (ogk >= qelo[2]) & (ogk <= qehi[2]))
            idx = np.nonzero(sel)[0]
            idx = idx[np.argsort(ogid[idx], kind="stable")]
            exp_items.append(idx + 1)
            self.export_index[q + 1] = self.export_index[q] + idx.size
        self.export_item = (np.concatenate(exp_items) if exp_items else np.zeros(0)).astype(np.int32)

    # ---- benchmark deck -------------------------------------------------
    def dirichlet(self):
        nodes = (np.nonzero(self.gijk[:, 2] == 0)[0] + 1).astype(np.int32)     # internal AND halo copies
        node = np.repeat(nodes, 3).astype(np.int32)
        dof = np.tile(np.array([1, 2, 3], dtype=np.int32), nodes.size)
        return node, dof, np.zeros(node.size)

    def load(self):
        b = np.zeros(3 * self.n_node)
        top = np.nonzero(self.gijk[:, 2] == self.G[2] - 1)[0]
        b[3 * top] = 1.0
        return b

    def hecmesh(self, hip):
        hm = hip.hecmwST_local_mesh(n_node=self.n_node, nn_internal=self.nn_internal)
        px, py, pz = self.dims
        hm.my_rank, hm.PETOT = self.rank, px * py * pz
        hm.n_neighbor_pe = int(self.neighbor_pe.size)
        hm.neighbor_pe = self.neighbor_pe
        hm.import_index, hm.import_item = self.import_index, self.import_item
        hm.export_index, hm.export_item = self.export_index, self.export_item
        return hm


def cube_subdomain(m, dims, rank):
    return CubeSubdomain(m, dims, rank)
