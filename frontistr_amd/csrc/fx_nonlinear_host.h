// Host entry points of the nonlinear path (included at the end of fistr_hip.hip): the steps of
// fstr_Newton (fstr_solve_NonLinear.f90:29-167) either side of the linear solve, on resident state.
#pragma once

static void nl_free(fx_context *c) {
  NlDev &n = c->nl;
  dev_free(n.tab); dev_free(n.coord); dev_free(n.conn);
  dev_free(n.stress); dev_free(n.strain); dev_free(n.stress_bak); dev_free(n.strain_bak);
  dev_free(n.plstrain); dev_free(n.fstat); dev_free(n.istat);
  dev_free(n.unode); dev_free(n.dunode); dev_free(n.qforce); dev_free(n.GL);
  dev_free(n.bc_flag); dev_free(n.bc_val); dev_free(n.bc_node); dev_free(n.bc_dof); dev_free(n.bc_v); dev_free(n.err);
  dev_free(n.colors.order); dev_free(n.colors.pos);
  dev_free(n.mats); dev_free(n.emat);
  dev_free(n.bk_stress); dev_free(n.bk_strain); dev_free(n.bk_stress_bak); dev_free(n.bk_strain_bak);
  dev_free(n.bk_plstrain); dev_free(n.bk_fstat); dev_free(n.bk_istat);
  for (double *t : n.tabs) { double *q = t; dev_free(q); }
  n = NlDev();
}

// fstr_solid / tGaussStatus set-up for one TYPE=361 B-bar group with one material (fstr_setup.f90:325-400,
// fstr_init_gauss mechgauss.f90:37-71): zero state, zero displacement.
static int nl_check_material(const fx_material_view *mat) {
  if (mat->harden < 0 || mat->harden > 3 || mat->nlgeom < 0 || mat->nlgeom > 2) {
    g_fx_error = "fx_nl_init: only Mises yield with BILINEAR/MULTILINEAR/SWIFT/RAMBERG-OSGOOD hardening is on the hot path";
    return FX_ERROR_UNSUPPORTED;
  }
  if (mat->plastic && mat->harden == 1 && (mat->ntab < 1 || !mat->tab)) { g_fx_error = "fx_nl_init: MULTILINEAR hardening needs a table"; return FX_ERROR_RUNTIME; }
  return 0;
}

static int nl_init_common(fx_context *c, const fx_mesh_view *mesh, int32_t n_mat, const fx_material_view *mats,
                          const int32_t *elem_mat) {
  HIP_TRY(hipSetDevice(c->device));
  if (!c->have_profile) { g_fx_error = "fx_nl_init: upload the profile first (fx_upload FX_UP_PROFILE)"; return FX_ERROR_RUNTIME; }
  if (mesh->n_node != c->A.NP) { g_fx_error = "fx_nl_init: mesh/profile size mismatch"; return FX_ERROR_RUNTIME; }
  if (n_mat < 1 || !mats || (n_mat > 1 && !elem_mat)) { g_fx_error = "fx_nl_init: materials missing"; return FX_ERROR_RUNTIME; }
  for (int32_t k = 0; k < n_mat; k++)
    if (int e = nl_check_material(&mats[k])) return e;
  if (mesh->n_elem < 1 || mesh->n_node < 1) { g_fx_error = "fx_nl_init: empty mesh"; return FX_ERROR_RUNTIME; }
  for (int64_t k = 0; k < (int64_t)8 * mesh->n_elem; k++)
    if (mesh->conn[k] < 1 || mesh->conn[k] > mesh->n_node) { g_fx_error = "fx_nl_init: node id out of range"; return FX_ERROR_RUNTIME; }
  if (n_mat > 1)
    for (int32_t e = 0; e < mesh->n_elem; e++)
      if (elem_mat[e] < 1 || elem_mat[e] > n_mat) { g_fx_error = "fx_nl_init_sections: material id out of range"; return FX_ERROR_RUNTIME; }
  nl_free(c);
  NlDev &n = c->nl;
  n.n_elem = mesh->n_elem;
  n.n_mat = n_mat;
  const size_t np3 = (size_t)3 * c->A.NP, npt = (size_t)8 * mesh->n_elem;
  if (dev_alloc(&n.coord, np3) || dev_alloc(&n.conn, npt) || dev_alloc(&n.stress, 6 * npt) || dev_alloc(&n.strain, 6 * npt) ||
      dev_alloc(&n.stress_bak, 6 * npt) || dev_alloc(&n.strain_bak, 6 * npt) || dev_alloc(&n.plstrain, npt) ||
      dev_alloc(&n.fstat, npt) || dev_alloc(&n.istat, npt) || dev_alloc(&n.unode, np3) || dev_alloc(&n.dunode, np3) ||
      dev_alloc(&n.qforce, np3) || dev_alloc(&n.GL, np3) || dev_alloc(&n.bc_flag, np3) || dev_alloc(&n.bc_val, np3) ||
      dev_alloc(&n.err, 1))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpyAsync(n.coord, mesh->coord, np3 * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(n.conn, mesh->conn, npt * 4, hipMemcpyHostToDevice, c->stream));
  // materials: one NlMat per section, hardening tables on the device
  n.h_mats.resize((size_t)n_mat);
  for (int32_t k = 0; k < n_mat; k++) {
    const fx_material_view &mv = mats[k];
    double *tab = nullptr;
    if (dev_alloc(&tab, (size_t)2 * std::max(mv.ntab, 1))) return FX_ERROR_RUNTIME;
    n.tabs.push_back(tab);
    if (mv.ntab > 0) HIP_TRY(hipMemcpyAsync(tab, mv.tab, (size_t)2 * mv.ntab * 8, hipMemcpyHostToDevice, c->stream));
    NlMat &m = n.h_mats[k];
    m.E = mv.E; m.nu = mv.nu;
    for (int i = 0; i < 3; i++) m.pl[i] = mv.plconst[i];
    m.plastic = mv.plastic ? 1 : 0; m.harden = mv.harden; m.nlgeom = mv.nlgeom; m.ntab = mv.ntab;
    m.tab = tab;
  }
  n.mat = n.h_mats[0];
  n.tab = nullptr;
  if (n_mat > 1) {
    if (dev_alloc(&n.mats, (size_t)n_mat) || dev_alloc(&n.emat, (size_t)mesh->n_elem)) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpyAsync(n.mats, n.h_mats.data(), (size_t)n_mat * sizeof(NlMat), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(n.emat, elem_mat, (size_t)mesh->n_elem * 4, hipMemcpyHostToDevice, c->stream));
  }
  // element lists: grouped by the NLGEOM flag of the element's material (one kernel instantiation per flag), inside a group
  // colour by colour (fx_order.cpp: color_elements) for the atomic-free scatter
  {
    static const bool force_atomic = getenv("FX_ASM_ATOMIC") && atoi(getenv("FX_ASM_ATOMIC")) != 0;
    std::vector<int32_t> order, off;
    const bool coloured = !force_atomic && fxo::color_elements(mesh->n_elem, 8, mesh->conn, mesh->n_node, order, off);
    if (!coloured) {
      order.resize((size_t)mesh->n_elem);
      for (int32_t e = 0; e < mesh->n_elem; e++) order[e] = e;
      off = {0, mesh->n_elem};
    }
    n.scatter_atomic = !coloured;
    std::vector<int32_t> grouped;
    grouped.reserve((size_t)mesh->n_elem);
    for (int g = 0; g < 3; g++) {
      n.grp_off[g].clear();
      bool any = false;
      for (size_t k = 0; k + 1 < off.size(); k++) {
        const size_t before = grouped.size();
        for (int32_t q = off[k]; q < off[k + 1]; q++) {
          const int32_t e = order[q];
          const int flag = n.h_mats[n_mat > 1 ? elem_mat[e] - 1 : 0].nlgeom;
          if (flag == g) grouped.push_back(e);
        }
        if (grouped.size() > before || any) {
          if (!any) n.grp_off[g].push_back((int32_t)before);
          any = true;
          n.grp_off[g].push_back((int32_t)grouped.size());
        }
      }
    }
    if (dev_alloc(&n.colors.order, (size_t)mesh->n_elem)) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpyAsync(n.colors.order, grouped.data(), (size_t)mesh->n_elem * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // grouped is a host temporary
    n.colors.n_elem = mesh->n_elem;
    n.colors.offsets = {0, mesh->n_elem};       // marks the lists as built (ensure_scatter_map)
    if (ensure_scatter_map(c, n.colors, mesh->n_elem, n.conn)) return FX_ERROR_RUNTIME;
  }
  for (double *p : {n.stress, n.strain, n.stress_bak, n.strain_bak}) HIP_TRY(hipMemsetAsync(p, 0, 6 * npt * 8, c->stream));
  for (double *p : {n.plstrain, n.fstat}) HIP_TRY(hipMemsetAsync(p, 0, npt * 8, c->stream));
  HIP_TRY(hipMemsetAsync(n.istat, 0, npt * 4, c->stream));
  for (double *p : {n.unode, n.dunode, n.qforce, n.GL, n.bc_val}) HIP_TRY(hipMemsetAsync(p, 0, np3 * 8, c->stream));
  HIP_TRY(hipMemsetAsync(n.bc_flag, 0, np3, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  n.latch = 0;
  n.ready = true;
  return 0;
}

extern "C" int fx_nl_init(fx_context *c, const fx_mesh_view *mesh, const fx_material_view *mat) {
  return nl_init_common(c, mesh, 1, mat, nullptr);
}

// Several sections (hecMESH%section_ID -> fstrSOLID%materials, fstr_setup.f90:325-400): elem_mat[e] in 1..n_mat.  The materials may
// carry different NLGEOM flags (an elastic TOTALLAG part next to an elastoplastic UPDATELAG part).
extern "C" int fx_nl_init_sections(fx_context *c, const fx_mesh_view *mesh, int32_t n_mat, const fx_material_view *mats,
                                   const int32_t *elem_mat) {
  return nl_init_common(c, mesh, n_mat, mats, elem_mat);
}

#define NL_READY(name)                                                                              \
  HIP_TRY(hipSetDevice(c->device));                                                                 \
  if (!c->nl.ready) { g_fx_error = name ": call fx_nl_init first"; return FX_ERROR_RUNTIME; }

template <int G>
static void nl_launch_stiffness_group(fx_context *c, double *Kout) {
  NlDev &n = c->nl;
  const DevCSR &A = c->A;
  const std::vector<int32_t> &off = n.grp_off[G];
  if (off.empty()) return;
  const bool one_range = Kout || n.scatter_atomic;  // element matrices out, or atomics: the group's colours in one launch
  for (size_t k = 0; k + 1 < off.size(); k++) {
    const int32_t e0 = one_range ? off.front() : off[k], e1 = one_range ? off.back() : off[k + 1];
    if (e1 > e0)
      hipLaunchKernelGGL((k_nl_stiffness<G>), dim3((e1 - e0 + FXN_EPB - 1) / FXN_EPB), dim3(FXN_BLOCK), 0, c->stream, e1, n.coord,
                         n.conn, n.unode, n.dunode, n.mat, n.latch, n.stress, n.fstat, n.istat, A.indexL, A.itemL, A.indexU, A.itemU,
                         A.D, A.AL, A.AU, Kout, n.err, (const int32_t *)n.colors.order, e0, (const int32_t *)n.colors.pos,
                         n.scatter_atomic ? 1 : 0, (const NlMat *)n.mats, (const int32_t *)n.emat);
    if (one_range) break;
  }
}
static void nl_launch_stiffness(fx_context *c, double *Kout) {  // one kernel instantiation per NLGEOM flag present
  nl_launch_stiffness_group<0>(c, Kout);
  nl_launch_stiffness_group<1>(c, Kout);
  nl_launch_stiffness_group<2>(c, Kout);
}
template <int G>
static void nl_launch_update_group(fx_context *c, double *qf_out) {
  NlDev &n = c->nl;
  const std::vector<int32_t> &off = n.grp_off[G];
  if (off.empty() || off.back() <= off.front()) return;
  const int32_t e0 = off.front(), e1 = off.back();
  // a group that holds every element is walked in the elements' own order (contiguous history arrays); the internal force is
  // scattered with atomics either way
  const int32_t *list = (e0 == 0 && e1 == n.n_elem) ? nullptr : n.colors.order;
  hipLaunchKernelGGL((k_nl_update<G>), dim3((e1 - e0 + FXN_EPB - 1) / FXN_EPB), dim3(FXN_BLOCK), 0, c->stream, e1, n.coord, n.conn,
                     n.unode, n.dunode, n.mat, n.stress, n.strain, n.stress_bak, n.strain_bak, n.plstrain, n.fstat, n.istat, n.qforce,
                     qf_out, list, e0, (const NlMat *)n.mats, (const int32_t *)n.emat);
}
static void nl_launch_update(fx_context *c, double *qf_out) {
  nl_launch_update_group<0>(c, qf_out);
  nl_launch_update_group<1>(c, qf_out);
  nl_launch_update_group<2>(c, qf_out);
}

// fstr_StiffMatrix + fstr_AddBC (fstr_StiffMatrix.f90:18-212, fstr_AddBC.f90:17-190): tangent of the current
// state (u = unode + dunode) into the resident D/AL/AU, then Dirichlet elimination with the given increments
// against the resident right-hand side.  The prescribed dofs are remembered for fx_nl_update.
extern "C" int fx_nl_stiffness(fx_context *c, int32_t n_bc, const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val,
                               float *ms_assemble) {
  NL_READY("fx_nl_stiffness");
  NlDev &n = c->nl;
  DevCSR &A = c->A;
  for (int32_t k = 0; k < n_bc; k++)
    if (bc_node[k] < 1 || bc_node[k] > A.NP) { g_fx_error = "fx_nl_stiffness: BC node id out of range"; return FX_ERROR_RUNTIME; }
  HIP_TRY(hipMemsetAsync(n.err, 0, 4, c->stream));
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(hipMemsetAsync(A.D, 0, (size_t)9 * A.NP * 8, c->stream));  // hecmw_mat_clear
  HIP_TRY(hipMemsetAsync(A.AL, 0, (size_t)9 * A.NPL * 8, c->stream));
  HIP_TRY(hipMemsetAsync(A.AU, 0, (size_t)9 * A.NPU * 8, c->stream));
  nl_launch_stiffness(c, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(n.bc_flag, 0, (size_t)3 * A.NP, c->stream));
  HIP_TRY(hipMemsetAsync(n.bc_val, 0, (size_t)3 * A.NP * 8, c->stream));
  n.n_bc = n_bc;
  if (n_bc > 0) {
    if (n_bc > n.bc_cap) {
      dev_free(n.bc_node); dev_free(n.bc_dof); dev_free(n.bc_v);
      if (dev_alloc(&n.bc_node, (size_t)n_bc) || dev_alloc(&n.bc_dof, (size_t)n_bc) || dev_alloc(&n.bc_v, (size_t)n_bc)) return FX_ERROR_RUNTIME;
      n.bc_cap = n_bc;
    }
    HIP_TRY(hipMemcpyAsync(n.bc_node, bc_node, (size_t)n_bc * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(n.bc_dof, bc_dof, (size_t)n_bc * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(n.bc_v, bc_val, (size_t)n_bc * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_bc_mark, dim3((n_bc + 255) / 256), dim3(256), 0, c->stream, n_bc, n.bc_node, n.bc_dof, n.bc_v, n.bc_flag, n.bc_val);
    const dim3 g((A.NP + 255) / 256);
    hipLaunchKernelGGL((k_bc_apply<1>), g, dim3(256), 0, c->stream, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B,
                       n.bc_flag, n.bc_val);
    hipLaunchKernelGGL((k_bc_apply<2>), g, dim3(256), 0, c->stream, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B,
                       n.bc_flag, n.bc_val);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  int32_t herr = 0;
  HIP_TRY(hipMemcpyAsync(&herr, n.err, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (ms_assemble) HIP_TRY(hipEventElapsedTime(ms_assemble, c->ev0, c->ev1));
  if (herr == 2) { g_fx_error = "###ERROR### : cannot find connectivity (element not covered by the profile)"; return FX_ERROR_RUNTIME; }
  c->have_values = true;
  c->bell_valid = false;   // the preconditioner is refreshed by the flags / recycle policy of the next solve, not here
  return 0;
}

static int nl_dot(fx_context *c, const double *x, const double *y, double *out) {  // hecmw_InnerProduct_R over the internal dofs
  if (c->max_partials < 4096 + 8) {
    dev_free(c->partials);
    if (dev_alloc(&c->partials, (size_t)(4096 + 8) * 3)) return FX_ERROR_RUNTIME;
    c->max_partials = 4096 + 8;
  }
  const int64_t len = (int64_t)3 * c->A.N;
  const int np = grid_for(len, FX_BLOCK, 2048);
  hipLaunchKernelGGL(k_dot, dim3(np), dim3(FX_BLOCK), 0, c->stream, len, x, y, c->partials, (const int32_t *)nullptr, 0);
  HIP_TRY(hipGetLastError());
  double tmp;
  return host_sum(c, np, 0, out, &tmp);
}

static int nl_halo_natural(fx_context *c, double *v) {  // hecmw_update_3_R on a vector in the reference numbering
  if (!multi_rank(c)) return 0;
  if (ensure_solver(c)) return FX_ERROR_RUNTIME;
  if (to_slots(c, v, c->W[6]) || halo_update(c, c->W[6]) || from_slots(c, c->W[6], v)) return FX_ERROR_RUNTIME;
  return 0;
}

// Start of a substep (fstr_Newton :63-68, fstr_ass_load.f90:64-91,:273): dunode = 0, GL = load at the end of the
// increment (host, 3*NP, may be NULL = no nodal load), B = GL - QFORCE.
extern "C" int fx_nl_begin_substep(fx_context *c, const double *GL) {
  NL_READY("fx_nl_begin_substep");
  NlDev &n = c->nl;
  const size_t np3 = (size_t)3 * c->A.NP;
  HIP_TRY(hipMemsetAsync(n.dunode, 0, np3 * 8, c->stream));
  if (GL) HIP_TRY(hipMemcpyAsync(n.GL, GL, np3 * 8, hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(hipMemsetAsync(n.GL, 0, np3 * 8, c->stream));
  hipLaunchKernelGGL(k_nl_residual, dim3(grid_for((int64_t)np3)), dim3(256), 0, c->stream, (int64_t)np3, n.GL, n.qforce,
                     (const uint8_t *)nullptr, c->A.B);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// After the solve (fstr_Newton :92-122): dunode += X, fstr_UpdateNewton, fstr_Update_NDForce, and the four
// norms of the convergence test: out = {|B|^2, |X|^2, |QFORCE|^2, |dunode|^2} over the internal dofs.
extern "C" int fx_nl_update(fx_context *c, double out[4], float *ms_update) {
  NL_READY("fx_nl_update");
  NlDev &n = c->nl;
  const int64_t np3 = (int64_t)3 * c->A.NP;
  hipLaunchKernelGGL(k_axpy_plain, dim3(grid_for(np3)), dim3(256), 0, c->stream, np3, 1.0, c->A.X, n.dunode);
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(hipMemsetAsync(n.qforce, 0, (size_t)np3 * 8, c->stream));
  nl_launch_update(c, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  if (n.mat.plastic) n.latch = 1;  // MatlMatrix(..., isEp=1) has now been called (calMatMatrix.f90:43-45)
  if (nl_halo_natural(c, n.qforce)) return FX_ERROR_RUNTIME;
  hipLaunchKernelGGL(k_nl_residual, dim3(grid_for(np3)), dim3(256), 0, c->stream, np3, n.GL, n.qforce, n.bc_flag, c->A.B);
  HIP_TRY(hipGetLastError());
  if (nl_halo_natural(c, c->A.B)) return FX_ERROR_RUNTIME;
  if (out) {
    if (nl_dot(c, c->A.B, c->A.B, out + 0) || nl_dot(c, c->A.X, c->A.X, out + 1) || nl_dot(c, n.qforce, n.qforce, out + 2) ||
        nl_dot(c, n.dunode, n.dunode, out + 3))
      return FX_ERROR_RUNTIME;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (ms_update) HIP_TRY(hipEventElapsedTime(ms_update, c->ev0, c->ev1));
  return 0;
}

// ---- the same two steps for a caller that keeps fstr_Newton's loop on the host (the Fortran binding of INTEGRATION.md section 5:
// fistr1's own fstr_Newton drives them).  unode / dunode are the host's (fstrSOLID%unode, %dunode: 3*NP doubles each travel per
// call, the 6.5 GB matrix does not); no boundary conditions, no right-hand side: fstr_AddBC and fstr_Update_NDForce stay the
// reference's, their hecmw_mat_ass_bc calls reach the resident matrix through fx_mat_ass_bc.
extern "C" int fx_nl_stiffness_at(fx_context *c, const double *unode, const double *dunode, float *ms_assemble) {
  NL_READY("fx_nl_stiffness_at");
  NlDev &n = c->nl;
  const size_t np3 = (size_t)3 * c->A.NP * 8;
  if (unode) HIP_TRY(hipMemcpyAsync(n.unode, unode, np3, hipMemcpyHostToDevice, c->stream));
  if (dunode) HIP_TRY(hipMemcpyAsync(n.dunode, dunode, np3, hipMemcpyHostToDevice, c->stream));
  return fx_nl_stiffness(c, 0, nullptr, nullptr, nullptr, ms_assemble);
}

// fstr_UpdateNewton (fstr_Update.f90:25-293) for the host's dunode (the host has already added the solver's X, fstr_solve_NonLinear.f90:95-97):
// stresses / strains / plastic state of every quadrature point on the device, internal force QFORCE back to the host (before the
// caller's hecmw_update_3_R, :284).
extern "C" int fx_nl_update_at(fx_context *c, const double *dunode, double *qforce, float *ms_update) {
  NL_READY("fx_nl_update_at");
  NlDev &n = c->nl;
  const size_t np3 = (size_t)3 * c->A.NP * 8;
  if (dunode) HIP_TRY(hipMemcpyAsync(n.dunode, dunode, np3, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(hipMemsetAsync(n.qforce, 0, np3, c->stream));
  nl_launch_update(c, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  if (n.mat.plastic) n.latch = 1;  // MatlMatrix(..., isEp=1) has now been called (calMatMatrix.f90:43-45)
  for (const NlMat &m : n.h_mats) if (m.plastic) n.latch = 1;
  if (qforce) HIP_TRY(hipMemcpyAsync(qforce, n.qforce, np3, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (ms_update) HIP_TRY(hipEventElapsedTime(ms_update, c->ev0, c->ev1));
  return 0;
}

// hecmw_mat_ass_bc (hecmw_mat_ass.f90:292-429) for a list of prescribed dofs on the RESIDENT matrix and right-hand side: column
// times value moved to B, row and column zeroed, unit diagonal, B = value.  The list is what fstr_AddBC would have passed call by call.
extern "C" int fx_mat_ass_bc(fx_context *c, int32_t n_bc, const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val) {
  HIP_TRY(hipSetDevice(c->device));
  if (!c->have_values) { g_fx_error = "fx_mat_ass_bc: no matrix values resident"; return FX_ERROR_RUNTIME; }
  if (n_bc <= 0) return 0;
  DevCSR &A = c->A;
  for (int32_t k = 0; k < n_bc; k++)
    if (bc_node[k] < 1 || bc_node[k] > A.NP || bc_dof[k] < 1 || bc_dof[k] > 3) { g_fx_error = "fx_mat_ass_bc: node id / dof out of range"; return FX_ERROR_RUNTIME; }
  DevScratch tmp;
  uint8_t *d_flag = nullptr;
  double *d_bcv = nullptr, *d_val = nullptr;
  int32_t *d_node = nullptr, *d_dof = nullptr;
  if (tmp.alloc(&d_flag, (size_t)3 * A.NP) || tmp.alloc(&d_bcv, (size_t)3 * A.NP) || tmp.alloc(&d_node, (size_t)n_bc) ||
      tmp.alloc(&d_dof, (size_t)n_bc) || tmp.alloc(&d_val, (size_t)n_bc))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemsetAsync(d_flag, 0, (size_t)3 * A.NP, c->stream));
  HIP_TRY(hipMemsetAsync(d_bcv, 0, (size_t)3 * A.NP * 8, c->stream));
  HIP_TRY(hipMemcpyAsync(d_node, bc_node, (size_t)n_bc * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_dof, bc_dof, (size_t)n_bc * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_val, bc_val, (size_t)n_bc * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_bc_mark, dim3((n_bc + 255) / 256), dim3(256), 0, c->stream, n_bc, d_node, d_dof, d_val, d_flag, d_bcv);
  const dim3 g((A.NP + 255) / 256);
  hipLaunchKernelGGL((k_bc_apply<1>), g, dim3(256), 0, c->stream, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B, d_flag, d_bcv);
  hipLaunchKernelGGL((k_bc_apply<2>), g, dim3(256), 0, c->stream, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B, d_flag, d_bcv);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->bell_valid = false;  // the streaming layouts re-gather the values on next use
  return 0;
}

// End of a converged substep (fstr_Newton :156-162): unode += dunode, fstr_UpdateState.
extern "C" int fx_nl_commit(fx_context *c) {
  NL_READY("fx_nl_commit");
  NlDev &n = c->nl;
  const int64_t np3 = (int64_t)3 * c->A.NP, npt = (int64_t)8 * n.n_elem;
  hipLaunchKernelGGL(k_axpy_plain, dim3(grid_for(np3)), dim3(256), 0, c->stream, np3, 1.0, n.dunode, n.unode);
  hipLaunchKernelGGL(k_nl_commit, dim3(grid_for(6 * npt)), dim3(256), 0, c->stream, npt, n.mat.plastic, n.fstat, n.plstrain, n.stress,
                     n.strain, n.stress_bak, n.strain_bak, (const NlMat *)n.mats, (const int32_t *)n.emat);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// fstr_cutback_save (load = 0) / fstr_cutback_load (load = 1), fstr_Cutback.f90:108-198, for the state that lives on the device:
// the quadrature-point history (fstr_copy_gauss of every point: strain, stress, their _bak copies, plstrain, istatus, fstatus).
// unode and QFORCE are the host's (the *_at entry points take them from fstrSOLID at every call) and are restored there by the
// reference's own routine; MatlMatrix's saved flag (`latch`) is process state the reference does not roll back either.
extern "C" int fx_nl_snapshot(fx_context *c, int load) {
  NL_READY("fx_nl_snapshot");
  NlDev &n = c->nl;
  const size_t npt = (size_t)8 * n.n_elem;
  if (!n.bk_stress) {
    if (load) { g_fx_error = "fx_nl_snapshot: nothing was saved"; return FX_ERROR_RUNTIME; }
    if (dev_alloc(&n.bk_stress, 6 * npt) || dev_alloc(&n.bk_strain, 6 * npt) || dev_alloc(&n.bk_stress_bak, 6 * npt) ||
        dev_alloc(&n.bk_strain_bak, 6 * npt) || dev_alloc(&n.bk_plstrain, npt) || dev_alloc(&n.bk_fstat, npt) || dev_alloc(&n.bk_istat, npt))
      return FX_ERROR_RUNTIME;
  }
  if (load && !n.bk_valid) { g_fx_error = "fx_nl_snapshot: nothing was saved"; return FX_ERROR_RUNTIME; }
  struct { void *live; void *bk; size_t bytes; } f[] = {
      {n.stress, n.bk_stress, 6 * npt * 8}, {n.strain, n.bk_strain, 6 * npt * 8}, {n.stress_bak, n.bk_stress_bak, 6 * npt * 8},
      {n.strain_bak, n.bk_strain_bak, 6 * npt * 8}, {n.plstrain, n.bk_plstrain, npt * 8}, {n.fstat, n.bk_fstat, npt * 8},
      {n.istat, n.bk_istat, npt * 4}};
  for (auto &e : f) {
    if (load) HIP_TRY(hipMemcpyAsync(e.live, e.bk, e.bytes, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(hipMemcpyAsync(e.bk, e.live, e.bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  n.bk_valid = true;
  return 0;
}

static int nl_copy_state(fx_context *c, const fx_nl_state_view *s, bool to_device) {
  NlDev &n = c->nl;
  const size_t np3 = (size_t)3 * c->A.NP * 8, npt = (size_t)8 * n.n_elem;
  struct { void *h; void *d; size_t bytes; } f[] = {
      {s->stress, n.stress, 6 * npt * 8}, {s->strain, n.strain, 6 * npt * 8}, {s->stress_bak, n.stress_bak, 6 * npt * 8},
      {s->strain_bak, n.strain_bak, 6 * npt * 8}, {s->plstrain, n.plstrain, npt * 8}, {s->fstat, n.fstat, npt * 8},
      {s->istat, n.istat, npt * 4}, {s->unode, n.unode, np3}, {s->dunode, n.dunode, np3}, {s->qforce, n.qforce, np3}};
  for (auto &e : f) {
    if (!e.h) continue;
    if (to_device) HIP_TRY(hipMemcpyAsync(e.d, e.h, e.bytes, hipMemcpyHostToDevice, c->stream));
    else HIP_TRY(hipMemcpyAsync(e.h, e.d, e.bytes, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int fx_nl_get_state(fx_context *c, fx_nl_state_view *s) {
  NL_READY("fx_nl_get_state");
  s->latch = c->nl.latch;
  return nl_copy_state(c, s, false);
}
extern "C" int fx_nl_set_state(fx_context *c, const fx_nl_state_view *s) {
  NL_READY("fx_nl_set_state");
  if (s->latch >= 0) c->nl.latch = s->latch ? 1 : 0;
  return nl_copy_state(c, s, true);
}

// Element-level outputs of the two kernels (tests): tangents ke (n_elem x 24 x 24 row-major) of the current state
// with u = unode + dunode, or the stress update with per-element internal forces qf (n_elem x 24), no scatter.
extern "C" int fx_nl_element_tangents(fx_context *c, double *ke) {
  NL_READY("fx_nl_element_tangents");
  NlDev &n = c->nl;
  DevScratch tmp;
  double *d = nullptr;
  if (tmp.alloc(&d, (size_t)576 * n.n_elem)) return FX_ERROR_RUNTIME;
  nl_launch_stiffness(c, d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ke, d, (size_t)576 * n.n_elem * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int fx_nl_element_update(fx_context *c, double *qf) {
  NL_READY("fx_nl_element_update");
  NlDev &n = c->nl;
  DevScratch tmp;
  double *d = nullptr;
  if (tmp.alloc(&d, (size_t)24 * n.n_elem)) return FX_ERROR_RUNTIME;
  nl_launch_update(c, d);
  HIP_TRY(hipGetLastError());
  if (n.mat.plastic) n.latch = 1;
  HIP_TRY(hipMemcpyAsync(qf, d, (size_t)24 * n.n_elem * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// One substep of fstr_Newton (fstr_solve_NonLinear.f90:29-167) around fx_solve_resident.
//   factor0/factor1: load factors at the start / end of the increment (fstr_solve_NLGEOM.f90:112-115);
//   bc_val, cload: values at load factor 1.  log: 7 doubles per Newton iteration
//   (iter, linear-solver iterations, solver return code, |B|, |X|, |QFORCE|, |dunode|).
// Returns 0 when the substep converged, FX_ERROR_NOCONV_MAXIT when max_iter ran out (the reference cuts back;
// the state is committed as fstr_Newton's caller would after its last iteration only on convergence).
extern "C" int fx_newton_substep(fx_context *c, double factor0, double factor1, int32_t n_bc, const int32_t *bc_node,
                                 const int32_t *bc_dof, const double *bc_val, const double *cload, int32_t max_iter, double converg,
                                 int32_t *Iarray, double *Rarray, double *log, int32_t *n_iter, int commit_unconverged) {
  NL_READY("fx_newton_substep");
  if (max_iter < 1) { g_fx_error = "fx_newton_substep: max_iter must be >= 1"; return FX_ERROR_RUNTIME; }
  const size_t np3 = (size_t)3 * c->A.NP;
  std::vector<double> gl, inc((size_t)std::max(n_bc, 1)), zero((size_t)std::max(n_bc, 1), 0.0);
  if (cload) {
    gl.resize(np3);
    for (size_t i = 0; i < np3; i++) gl[i] = cload[i] * factor1;
  }
  for (int32_t k = 0; k < n_bc; k++) inc[k] = bc_val[k] * (factor1 - factor0);
  int e = fx_nl_begin_substep(c, cload ? gl.data() : nullptr);
  if (e) return e;
  bool done = false, blown = false;
  int32_t it = 0;
  for (it = 1; it <= max_iter; it++) {
    e = fx_nl_stiffness(c, n_bc, bc_node, bc_dof, it == 1 ? inc.data() : zero.data(), nullptr);
    if (e) return e;
    Iarray[96] = (it == 1) ? 2 : 1;  // Iarray(97): force / need numerical factorisation (:82-86)
    HIP_TRY(hipMemsetAsync(c->A.X, 0, np3 * 8, c->stream));
    fx_solve_info info;
    memset(&info, 0, sizeof info);
    const int code = fx_solve_resident(c, Iarray, Rarray, &info, nullptr, 0);
    if (code < 0 || code == FX_ERROR_ZERO_DIAG || code == FX_ERROR_INCONS_PC) return code;
    double nrm[4];
    e = fx_nl_update(c, nrm, nullptr);
    if (e) return e;
    const double res = sqrt(nrm[0]), xnrm = sqrt(nrm[1]);
    double qnrm = sqrt(nrm[2]);
    if (qnrm < 1.0e-8) qnrm = 1.0;
    const double dunrm = (it == 1) ? xnrm : sqrt(nrm[3]);
    if (log) {
      double *l = log + (size_t)7 * (it - 1);
      l[0] = it; l[1] = info.iterations; l[2] = code; l[3] = res; l[4] = xnrm; l[5] = qnrm; l[6] = dunrm;
    }
    if (c->nl.is_linear) { done = true; break; }  // `if( isLinear ) exit` (:107): a linear analysis takes one pass, no convergence test
    if (Iarray[80] == 1) {  // hecmw_mat_get_flag_converged (:132-135)
      if (res / qnrm < converg) done = true;
      if (xnrm / dunrm < converg) done = true;
    }
    if (done) break;
    if (res / qnrm > c->nl.maxres) { blown = true; break; }  // `rres > maxres` (:140): give up at once, the caller cuts back (knstDRESN = 2)
  }
  if (n_iter) *n_iter = std::min(it, max_iter);
  if (blown) return FX_NEWTON_MAXRES;  // like the reference's `return`: nothing is committed
  if (done || commit_unconverged) {
    e = fx_nl_commit(c);
    if (e) return e;
  }
  return done ? 0 : FX_ERROR_NOCONV_MAXIT;
}

// step_ctrl(cstep)%maxres (m_step.f90:31, default 1.d+10 :78) and fstr_Newton's isLinear (.not. fstrPR%nlgeom, :50-51).
extern "C" int fx_nl_set_step_control(fx_context *c, double maxres, int is_linear) {
  c->nl.maxres = maxres > 0.0 ? maxres : 1.0e10;
  c->nl.is_linear = is_linear != 0;
  return 0;
}
