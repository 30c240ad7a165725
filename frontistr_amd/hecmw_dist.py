"""Reader for HECMW-DIST mesh files (`<name>.<rank>`, "!HECMW-DMD-ASCII version=4"), the files
the reference partitioner hecmw_part1 writes and `hecmw_get_mesh` reads on every rank
(hecmw1/src/common/hecmw_io_dist.c: get_global_info :245, get_node_info :331, get_elem_info :479,
get_comm_info :648, get_section_info :919, get_material_info :1033, get_mpc_info :1135,
get_amp_info :1215, get_node_group_info :1315).  Field order and the fscanf/fgets token rules
(:34-103) are the reference's; only what the linear-solve hot path needs is kept: node
coordinates, TYPE=361 connectivity, global ids, the communication tables and the node groups.

Host-side logic of the multi-GPU path: the result feeds hecmw.hecmwST_local_mesh unchanged.
"""
import re

import numpy as np

_INT = re.compile(r"\s*([-+]?\d+)")
_TOK = re.compile(r"\s*(\S+)")


class _Stream:
    def __init__(self, text):
        self.t, self.p = text, 0

    def int(self):
        m = _INT.match(self.t, self.p)
        if not m:
            raise ValueError("HECMW-DIST: integer expected at offset %d" % self.p)
        self.p = m.end()
        return int(m.group(1))

    def ints(self, n):
        out = np.empty(n, dtype=np.int64)
        for i in range(n):
            out[i] = self.int()
        return out

    def double(self):
        m = _TOK.match(self.t, self.p)
        if not m:
            raise ValueError("HECMW-DIST: real expected at offset %d" % self.p)
        self.p = m.end()
        return float(m.group(1))

    def doubles(self, n):
        return np.array([self.double() for _ in range(n)], dtype=np.float64)

    def string(self):                      # skip white space, then the rest of the line (get_string :70-98)
        while self.p < len(self.t) and self.t[self.p].isspace():
            self.p += 1
        e = self.t.find("\n", self.p)
        e = len(self.t) if e < 0 else e
        s = self.t[self.p:e].rstrip()
        self.p = e
        return s

    def strings(self, n):
        return [self.string() for _ in range(n)]


class DistMesh:
    """The members of hecmwST_local_mesh the hot path reads (hecmw_util_f.F90:232-381)."""

    def hecmesh(self, hip):
        hm = hip.hecmwST_local_mesh(n_node=self.n_node, nn_internal=self.nn_internal)
        hm.my_rank, hm.PETOT = self.my_rank, self.PETOT
        hm.n_neighbor_pe = int(self.neighbor_pe.size)
        hm.neighbor_pe = self.neighbor_pe
        hm.import_index, hm.import_item = self.import_index, self.import_item
        hm.export_index, hm.export_item = self.export_index, self.export_item
        return hm

    def group(self, name):
        return self.node_groups[name]


def read_dist(path):
    s = _Stream(open(path).read())
    head = s.string()
    if not head.startswith("!HECMW-DMD-ASCII"):
        raise ValueError("%s is not a HECMW-DIST file" % path)
    m = DistMesh()
    # ---- global info
    m.flag_adapt, m.flag_initcon, m.flag_parttype, m.flag_partdepth, m.flag_version = (s.int() for _ in range(5))
    m.flag_partcontact = s.int() if m.flag_version >= 4 else 0
    m.gridfile = s.string()
    n_file = s.int()
    m.files = s.strings(n_file)
    m.header = s.string() if s.int() == 1 else ""
    m.zero_temp = s.double()
    # ---- nodes
    m.n_node = s.int()
    n_gross = s.int() if m.flag_version >= 2 else m.n_node
    m.nn_middle = s.int() if m.flag_version >= 4 else m.n_node
    m.nn_internal = s.int()
    if m.flag_parttype in (0, 2) and m.nn_internal > 0:      # element based / unknown: explicit internal list
        m.node_internal_list = s.ints(m.nn_internal)
    m.node_ID = s.ints(2 * n_gross).reshape(-1, 2)           # (local id in owner, owner rank)
    m.global_node_ID = s.ints(n_gross)
    m.node = s.doubles(3 * n_gross).reshape(-1, 3)
    m.n_dof, n_dof_grp = s.int(), s.int()
    if n_dof_grp > 0:
        s.ints(n_dof_grp + 1)
        s.ints(n_dof_grp)
    if m.flag_initcon and n_gross > 0:
        idx = s.ints(n_gross + 1)
        if idx[-1] > 0:
            s.doubles(int(idx[-1]))
    # ---- elements
    m.n_elem = s.int()
    ne_gross = s.int() if m.flag_version >= 2 else m.n_elem
    m.ne_internal = s.int()
    if m.flag_parttype in (0, 1) and m.ne_internal > 0:      # node based / unknown: explicit internal list
        m.elem_internal_list = s.ints(m.ne_internal)
    m.elem_ID = s.ints(2 * ne_gross).reshape(-1, 2)
    m.global_elem_ID = s.ints(ne_gross)
    m.elem_type = s.ints(ne_gross)
    n_type = s.int()
    m.elem_type_index = s.ints(n_type + 1)
    m.elem_type_item = s.ints(n_type)
    m.elem_node_index = s.ints(ne_gross + 1)
    m.elem_node_item = s.ints(int(m.elem_node_index[-1])).astype(np.int32)
    m.section_ID = s.ints(ne_gross)
    mi = s.ints(ne_gross + 1)
    s.ints(int(mi[-1]))
    s.int()                                                   # n_elem_mat_ID
    # ---- communication tables
    m.zero, _comm, m.PETOT, _smp, m.my_rank, _errnof, m.n_subdomain, nnb = (s.int() for _ in range(8))
    if nnb > 0:
        m.neighbor_pe = s.ints(nnb).astype(np.int32)
        m.import_index = s.ints(nnb + 1).astype(np.int32)
        m.import_item = s.ints(int(m.import_index[-1])).astype(np.int32)
        m.export_index = s.ints(nnb + 1).astype(np.int32)
        m.export_item = s.ints(int(m.export_index[-1])).astype(np.int32)
        sh = s.ints(nnb + 1)
        s.ints(int(sh[-1]))
    else:
        m.neighbor_pe = np.zeros(0, dtype=np.int32)
        m.import_index = m.export_index = np.zeros(1, dtype=np.int32)
        m.import_item = m.export_item = np.zeros(0, dtype=np.int32)
    if m.flag_adapt:
        raise NotImplementedError("adaptation data in HECMW-DIST files is outside the hot path")
    # ---- section / material / mpc / amplitude (skipped field by field to reach the groups)
    n_sect = s.int()
    if n_sect > 0:
        s.ints(n_sect); s.ints(n_sect)
        for real in (False, False, True):
            idx = s.ints(n_sect + 1)
            if idx[-1] > 0:
                (s.doubles if real else s.ints)(int(idx[-1]))
    n_mat = s.int()
    m.materials = {}
    if n_mat > 0:
        n_item, n_sub, n_tab = s.int(), s.int(), s.int()
        names = s.strings(n_mat)
        item_index = s.ints(n_mat + 1)
        sub_index = s.ints(n_item + 1)
        tab_index = s.ints(n_sub + 1)
        val = s.doubles(n_tab)
        s.doubles(n_tab)
        for k, nm in enumerate(names):                      # first item = elastic constants (E, nu)
            it0 = int(item_index[k])
            sub0 = int(sub_index[it0])
            m.materials[nm] = val[int(tab_index[sub0]):int(tab_index[sub0 + 2])] if n_sub >= sub0 + 2 else val
    n_mpc = s.int()
    if n_mpc > 0:
        idx = s.ints(n_mpc + 1)
        s.ints(int(idx[-1])); s.ints(int(idx[-1])); s.doubles(int(idx[-1])); s.doubles(n_mpc)
    m.n_mpc = n_mpc
    n_amp = s.int()
    if n_amp > 0:
        s.strings(n_amp); s.ints(n_amp); s.ints(n_amp); s.ints(n_amp)
        idx = s.ints(n_amp + 1)
        s.doubles(int(idx[-1])); s.doubles(int(idx[-1]))
    # ---- node groups
    n_grp = s.int()
    m.node_groups = {}
    if n_grp > 0:
        names = s.strings(n_grp)
        idx = s.ints(n_grp + 1)
        items = s.ints(int(idx[-1])) if idx[-1] > 0 else np.zeros(0, dtype=np.int64)
        for k, nm in enumerate(names):
            m.node_groups[nm] = items[idx[k]:idx[k + 1]].astype(np.int32)
    # ---- what the hot path consumes
    if not np.all(m.elem_type == 361):
        raise NotImplementedError("only TYPE=361 (C3D8) element groups are on the hot path")
    m.coord = np.ascontiguousarray(m.node)
    m.conn = np.ascontiguousarray(m.elem_node_item.reshape(-1, 8))
    m.global_id = (m.global_node_ID - 1).astype(np.int64)
    return m
