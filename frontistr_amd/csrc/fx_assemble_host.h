// Host entry points of the assembly side (included at the end of fistr_hip.hip).
#pragma once

// hecmw_mat_con (hecmw_mat_con.f90:23-268): CRS block profile from element connectivity.
// Node -> element incidence by counting sort (host threads, relaxed atomic counters: the order of a node's elements does
// not matter, its neighbour set is sorted afterwards), then per node the sorted unique set of the nodes of its elements,
// split into lower / upper; rows are independent => one contiguous chunk of nodes per host thread, each keeping the lists
// of its rows back to back.  The two-call protocol (count, then fill) would build all of that twice; the first call keeps
// the chunks (~ (NPL + NPU + NP) ints) for the second, which then only splits them.
struct MatConCache {
  const int32_t *conn = nullptr;
  int32_t NP = 0, n_elem = 0, nn = 0;
  std::vector<int64_t> cstart;             // chunk c holds the rows of nodes cstart[c] + 1 .. cstart[c + 1] (1-based)
  std::vector<std::vector<int32_t>> rows;  // per chunk: sorted unique neighbour ids (1-based, itself included) of its nodes, back to back
  std::vector<int32_t> rlen;               // per node (index 1..NP): length of its list
  void clear() {
    conn = nullptr;
    std::vector<int64_t>().swap(cstart);
    std::vector<std::vector<int32_t>>().swap(rows);
    std::vector<int32_t>().swap(rlen);
  }
  template <class F>
  void for_chunks(F f) const {  // f(chunk) on one host thread per chunk
    std::vector<std::thread> th;
    for (size_t c = 0; c + 1 < cstart.size(); c++) th.emplace_back([=] { f((int)c); });
    for (auto &t : th) t.join();
  }
};
static thread_local MatConCache g_matcon;  // per calling thread: the count and the fill call of one profile come from the same thread; concurrent callers (one thread per subdomain) do not share it

extern "C" int fx_mat_con(int32_t NP, int32_t n_elem, int32_t nn, const int32_t *conn, int32_t *indexL, int32_t *indexU,
                          int32_t *itemL, int32_t *itemU) {
  MatConCache &mc = g_matcon;
  const bool fill = itemL && itemU;
  if (!(fill && mc.conn == conn && mc.NP == NP && mc.n_elem == n_elem && mc.nn == nn && !mc.rlen.empty())) {
    mc.clear();
    PhaseTimer pt("mat_con");
    const int64_t tot = (int64_t)n_elem * nn;
    std::vector<int64_t> ptr((size_t)NP + 2, 0);
    {
      std::vector<int32_t> deg((size_t)NP + 2, 0);
      int bad = 0;
      parallel_for(tot, [&](int64_t a, int64_t b) {
        for (int64_t k = a; k < b; k++) {
          const int32_t v = conn[k];
          if (v < 1 || v > NP) { __atomic_store_n(&bad, 1, __ATOMIC_RELAXED); return; }
          __atomic_fetch_add(&deg[v + 1], 1, __ATOMIC_RELAXED);
        }
      });
      if (bad) { g_fx_error = "fx_mat_con: node id out of range"; return FX_ERROR_RUNTIME; }
      for (int32_t i = 1; i <= NP + 1; i++) ptr[i] = ptr[i - 1] + deg[i];
    }
    pt.lap("count");
    std::vector<int32_t> inc((size_t)tot);
    {
      std::vector<int64_t> pos(ptr.begin(), ptr.end());
      parallel_for(n_elem, [&](int64_t a, int64_t b) {
        for (int64_t e = a; e < b; e++)
          for (int j = 0; j < nn; j++) inc[__atomic_fetch_add(&pos[conn[(size_t)e * nn + j]], 1, __ATOMIC_RELAXED)] = (int32_t)e;
      });
    }
    pt.lap("incidence");
    const int nt = (int)std::min<int64_t>(nthreads_host(), std::max<int64_t>(1, NP / 4096));
    mc.cstart.resize((size_t)nt + 1);
    for (int c = 0; c <= nt; c++) mc.cstart[c] = (int64_t)NP * c / nt;
    mc.rows.assign((size_t)nt, std::vector<int32_t>());
    mc.rlen.assign((size_t)NP + 1, 0);
    mc.for_chunks([&](int c) {
      std::vector<int32_t> &out = mc.rows[c];
      const int64_t a = mc.cstart[c], b = mc.cstart[c + 1];
      out.reserve((size_t)((ptr[b + 1] - ptr[a + 1]) * 7 / 2 + 64));  // a hex8 node: 8 elements x 8 nodes -> 27 neighbours
      std::vector<int32_t> buf;
      for (int64_t i = a + 1; i <= b; i++) {
        buf.clear();
        for (int64_t q = ptr[i]; q < ptr[i + 1]; q++) {
          const int32_t *en = conn + (size_t)inc[q] * nn;
          buf.insert(buf.end(), en, en + nn);
        }
        std::sort(buf.begin(), buf.end());
        buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
        mc.rlen[i] = (int32_t)buf.size();
        out.insert(out.end(), buf.begin(), buf.end());
      }
    });
    mc.conn = conn; mc.NP = NP; mc.n_elem = n_elem; mc.nn = nn;
    pt.lap("row lists");
  }
  PhaseTimer pt2(fill ? "mat_con fill" : "mat_con index");
  if (!fill) {
    std::vector<int32_t> nl((size_t)NP + 1, 0), nu((size_t)NP + 1, 0);
    mc.for_chunks([&](int c) {
      const int32_t *r = mc.rows[c].data();
      for (int64_t i = mc.cstart[c] + 1; i <= mc.cstart[c + 1]; i++) {
        int32_t l = 0, u = 0;
        for (int32_t k = 0; k < mc.rlen[i]; k++) { l += (r[k] < i); u += (r[k] > i); }
        nl[i] = l; nu[i] = u;
        r += mc.rlen[i];
      }
    });
    indexL[0] = 0; indexU[0] = 0;
    int64_t cl = 0, cu = 0;
    for (int32_t i = 1; i <= NP; i++) {
      cl += nl[i]; cu += nu[i];
      if (cl > INT32_MAX || cu > INT32_MAX) { g_fx_error = "fx_mat_con: profile exceeds int32 (kint=4)"; mc.clear(); return FX_ERROR_RUNTIME; }
      indexL[i] = (int32_t)cl; indexU[i] = (int32_t)cu;
    }
    pt2.lap("indexL / indexU");
    return 0;  // the lists stay for the fill call
  }
  mc.for_chunks([&](int c) {
    const int32_t *r = mc.rows[c].data();
    for (int64_t i = mc.cstart[c] + 1; i <= mc.cstart[c + 1]; i++) {
      int32_t *pl = itemL + indexL[i - 1], *pu = itemU + indexU[i - 1];
      for (int32_t k = 0; k < mc.rlen[i]; k++) {
        const int32_t v = r[k];
        if (v < i) *pl++ = v;
        else if (v > i) *pu++ = v;
      }
      r += mc.rlen[i];
    }
  });
  pt2.lap("itemL / itemU");
  mc.clear();
  return 0;
}

// Host only: the ordering of the multicolour SSOR (hecmw_precond_SSOR_33.f90:102-111 -> hecmw_matrix_ordering_CM.f90:16-178,
// hecmw_matrix_ordering_MC.f90:15-72) as the library computes it: perm (new -> old, 1-based, N entries) and COLORindex(0:ncolor).
extern "C" int fx_ssor_ordering(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU, const int32_t *itemU,
                                int32_t ncolor_in, int32_t *perm, int32_t *colorindex, int32_t colorindex_cap, int32_t *ncolor) {
  if (N < 1 || ncolor_in < 1) { g_fx_error = "fx_ssor_ordering: N and ncolor_in must be >= 1"; return FX_ERROR_RUNTIME; }
  fxo::Graph g = fxo::build_graph(N, indexL, itemL, indexU, itemU);
  std::vector<int32_t> seq = fxo::rcm_sequence(g), p0, cidx;
  fxo::multicolor(g, seq, ncolor_in, p0, cidx);
  *ncolor = (int32_t)cidx.size() - 1;
  if ((int32_t)cidx.size() > colorindex_cap) { g_fx_error = "fx_ssor_ordering: colorindex too small"; return FX_ERROR_RUNTIME; }
  for (int32_t i = 0; i < N; i++) perm[i] = p0[i] + 1;
  std::copy(cidx.begin(), cidx.end(), colorindex);
  return 0;
}

extern "C" int fx_color_elements(int32_t NP, int32_t n_elem, int32_t nn, const int32_t *conn, int32_t *order, int32_t *offsets,
                                 int32_t *ncolor) {
  for (int64_t k = 0; k < (int64_t)n_elem * nn; k++)
    if (conn[k] < 1 || conn[k] > NP) { g_fx_error = "fx_color_elements: node id out of range"; return FX_ERROR_RUNTIME; }
  std::vector<int32_t> ord, off;
  *ncolor = 0;
  for (int k = 0; k < 65; k++) offsets[k] = 0;
  if (!fxo::color_elements(n_elem, nn, conn, NP, ord, off)) return 0;
  *ncolor = (int32_t)off.size() - 1;
  std::copy(ord.begin(), ord.end(), order);
  for (size_t k = 0; k < off.size(); k++) offsets[k] = off[k];
  for (size_t k = off.size(); k < 65; k++) offsets[k] = off.back();
  return 0;
}

static void elastic_constants(double E, double nu, double &D11, double &D12, double &D44) {
  // calElasticMatrix, 3-D case (ElasticLinear.f90:43-55)
  D11 = E * (1.0 - nu) / (1.0 - 2.0 * nu) / (1.0 + nu);
  D12 = E * nu / (1.0 - 2.0 * nu) / (1.0 + nu);
  D44 = E / (1.0 + nu) * 0.5;
}

// Colour the elements of a mesh once (fx_order.cpp: color_elements) and keep the grouped element list on the device; the
// stiffness kernels then scatter colour by colour without atomics.  FX_ASM_ATOMIC=1 keeps the single-launch atomic scatter.
static void elem_colors_free(ElemColors &ec) {
  dev_free(ec.order);
  dev_free(ec.pos);
  ec = ElemColors();
}
// the position map of k_scatter_map for the resident profile and the device connectivity d_conn (FX_ASM_MAP=0: search every time)
static int ensure_scatter_map(fx_context *c, ElemColors &ec, int32_t n_elem, const int32_t *d_conn, bool with_first = false) {
  static const bool off = getenv("FX_ASM_MAP") && atoi(getenv("FX_ASM_MAP")) == 0;
  if (off || ec.pos || ec.offsets.empty()) return 0;
  if (dev_alloc(&ec.pos, (size_t)64 * n_elem)) { (void)hipGetLastError(); ec.pos = nullptr; return 0; }  // no memory: keep searching
  const DevCSR &A = c->A;
  hipLaunchKernelGGL(k_scatter_map, dim3((unsigned)(((int64_t)64 * n_elem + 255) / 256)), dim3(256), 0, c->stream, n_elem, d_conn,
                     A.indexL, A.itemL, A.indexU, A.itemU, ec.pos);
  HIP_TRY(hipGetLastError());
  // first-write flags (FX_ASM_FIRST=0: off): which contribution to a block comes first in the order of the colour launches
  static const bool no_first = getenv("FX_ASM_FIRST") && atoi(getenv("FX_ASM_FIRST")) == 0;
  ec.first_write = false;
  if (!with_first || no_first || ec.dup_nodes) return 0;  // (the nonlinear kernels keep their own map, without flags: fx_nonlinear_host.h)
  {
    DevScratch tmp;
    int32_t *ecol = nullptr, *minD = nullptr, *minL = nullptr, *minU = nullptr;
    unsigned long long *cnt = nullptr;
    if (tmp.alloc(&ecol, (size_t)n_elem) || tmp.alloc(&minD, (size_t)A.NP) || tmp.alloc(&minL, (size_t)std::max(A.NPL, 1)) ||
        tmp.alloc(&minU, (size_t)std::max(A.NPU, 1)) || tmp.alloc(&cnt, 1)) {
      (void)hipGetLastError();
      return 0;  // no memory for the temporaries: the scatter stays read-modify-write everywhere
    }
    std::vector<int32_t> order((size_t)n_elem), color((size_t)n_elem, 0);
    HIP_TRY(hipMemcpy(order.data(), ec.order, (size_t)n_elem * 4, hipMemcpyDeviceToHost));
    for (size_t k = 0; k + 1 < ec.offsets.size(); k++)
      for (int32_t e = ec.offsets[k]; e < ec.offsets[k + 1]; e++) color[order[e]] = (int32_t)k;
    HIP_TRY(hipMemcpyAsync(ecol, color.data(), (size_t)n_elem * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(minD, 0x7F, (size_t)A.NP * 4, c->stream));  // FXA_NO_COLOR = 0x7F7F7F7F: above every colour
    HIP_TRY(hipMemsetAsync(minL, 0x7F, (size_t)std::max(A.NPL, 1) * 4, c->stream));
    HIP_TRY(hipMemsetAsync(minU, 0x7F, (size_t)std::max(A.NPU, 1) * 4, c->stream));
    HIP_TRY(hipMemsetAsync(cnt, 0, 8, c->stream));
    const dim3 g((unsigned)(((int64_t)64 * n_elem + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_scatter_first_min, g, b, 0, c->stream, n_elem, d_conn, (const int32_t *)ec.pos, (const int32_t *)ecol, minD, minL, minU);
    hipLaunchKernelGGL(k_scatter_first_flag, g, b, 0, c->stream, n_elem, d_conn, ec.pos, (const int32_t *)ecol, (const int32_t *)minD,
                       (const int32_t *)minL, (const int32_t *)minU);
    hipLaunchKernelGGL(k_count_uncovered, dim3(1024), b, 0, c->stream, (int64_t)A.NP, (const int32_t *)minD, cnt);
    if (A.NPL > 0) hipLaunchKernelGGL(k_count_uncovered, dim3(1024), b, 0, c->stream, (int64_t)A.NPL, (const int32_t *)minL, cnt);
    if (A.NPU > 0) hipLaunchKernelGGL(k_count_uncovered, dim3(1024), b, 0, c->stream, (int64_t)A.NPU, (const int32_t *)minU, cnt);
    unsigned long long uncovered = 1;
    HIP_TRY(hipMemcpyAsync(&uncovered, cnt, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // also: `color` is a host temporary
    HIP_TRY(hipGetLastError());
    ec.first_write = (uncovered == 0);  // a block nobody writes would keep what it held: then the matrix is cleared as before (the flags are harmless)
  }
  return 0;
}
static int ensure_elem_colors(fx_context *c, ElemColors &ec, int32_t n_elem, const int32_t *conn, int32_t NP) {
  static const bool force_atomic = getenv("FX_ASM_ATOMIC") && atoi(getenv("FX_ASM_ATOMIC")) != 0;
  if (force_atomic || n_elem < 1) { elem_colors_free(ec); return 0; }
  // checksum of the connectivity: per-chunk FNV-1a, chunks combined in order
  const int64_t nw = (int64_t)8 * n_elem;
  const int nchunk = 64;
  uint64_t part[nchunk];
  bool bad[nchunk], dup[nchunk];
  parallel_for(nchunk, [&](int64_t a, int64_t b) {
    for (int64_t q = a; q < b; q++) {
      uint64_t h = 1469598103934665603ull;
      bool oob = false;
      for (int64_t i = nw * q / nchunk; i < nw * (q + 1) / nchunk; i++) {
        h = (h ^ (uint32_t)conn[i]) * 1099511628211ull;
        oob |= (conn[i] < 1 || conn[i] > NP);
      }
      bool dp = false;  // an element that names a node twice (a collapsed hexahedron): two of its 64 blocks coincide
      for (int64_t e = (int64_t)n_elem * q / nchunk; e < (int64_t)n_elem * (q + 1) / nchunk; e++)
        for (int x = 0; x < 8; x++)
          for (int y = x + 1; y < 8; y++) dp |= (conn[8 * e + x] == conn[8 * e + y]);
      part[q] = h; bad[q] = oob; dup[q] = dp;
    }
  });
  uint64_t key = 1469598103934665603ull;
  for (int q = 0; q < nchunk; q++) {
    if (bad[q]) { g_fx_error = "element connectivity: node id out of range"; return FX_ERROR_RUNTIME; }
    key = (key ^ part[q]) * 1099511628211ull;
  }
  if (ec.order && ec.n_elem == n_elem && ec.key == key && !ec.offsets.empty()) return 0;
  elem_colors_free(ec);
  std::vector<int32_t> order, off;
  if (!fxo::color_elements(n_elem, 8, conn, NP, order, off)) return 0;  // a node in more than 64 elements: atomics
  if (dev_alloc(&ec.order, (size_t)n_elem)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpyAsync(ec.order, order.data(), (size_t)n_elem * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  ec.n_elem = n_elem; ec.key = key; ec.offsets = off;
  ec.dup_nodes = false;
  for (int q = 0; q < nchunk; q++) ec.dup_nodes |= dup[q];
  return 0;
}

template <int EO>
static void launch_assemble(fx_context *c, int32_t n_elem, const double *coord, const int32_t *conn, double D11, double D12,
                            double D44, double *Kout, int32_t *err, const int32_t *elem_mat = nullptr,
                            const double *mat_tab = nullptr, const ElemColors *ec = nullptr) {
  const DevCSR &A = c->A;
  if (ec && !ec->offsets.empty() && !Kout) {
    for (size_t k = 0; k + 1 < ec->offsets.size(); k++) {
      const int32_t e0 = ec->offsets[k], e1 = ec->offsets[k + 1];
      if (e1 <= e0) continue;
      hipLaunchKernelGGL((k_assemble_c3d8<EO>), dim3((e1 - e0 + FXA_EPB(EO) - 1) / FXA_EPB(EO)), dim3(FXA_BS(EO)), 0, c->stream, e1, coord,
                         conn, D11, D12, D44, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, Kout, err, elem_mat, mat_tab,
                         (const int32_t *)ec->order, e0, (const int32_t *)ec->pos);
    }
    return;
  }
  hipLaunchKernelGGL((k_assemble_c3d8<EO>), dim3((n_elem + FXA_EPB(EO) - 1) / FXA_EPB(EO)), dim3(FXA_BS(EO)), 0, c->stream, n_elem,
                     coord, conn, D11, D12, D44, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, Kout, err, elem_mat,
                     mat_tab, (const int32_t *)nullptr, 0, (const int32_t *)nullptr);
}

static int assemble_c3d8_common(fx_context *c, const fx_mesh_view *mesh, double E, double nu, int32_t n_mat, const double *Es,
                                const double *nus, const int32_t *elem_mat, int elemopt, const double *load, int32_t n_bc,
                                const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val, float *ms_assemble) {
  HIP_TRY(hipSetDevice(c->device));
  if (!c->have_profile) { g_fx_error = "fx_assemble_c3d8: upload the profile first (fx_upload FX_UP_PROFILE)"; return FX_ERROR_RUNTIME; }
  if (mesh->n_node != c->A.NP) { g_fx_error = "fx_assemble_c3d8: mesh/profile size mismatch"; return FX_ERROR_RUNTIME; }
  if (elemopt < 1 || elemopt > 3) { g_fx_error = "fx_assemble_c3d8: elemopt must be 1 (IC), 2 (B-bar) or 3 (FI)"; return FX_ERROR_UNSUPPORTED; }
  DevCSR &A = c->A;
  DevScratch tmp;
  double *d_coord = nullptr, *d_bcv = nullptr, *d_val = nullptr;
  int32_t *d_conn = nullptr, *d_err = nullptr, *d_node = nullptr, *d_dof = nullptr;
  uint8_t *d_flag = nullptr;
  if (tmp.alloc(&d_coord, (size_t)3 * mesh->n_node) || tmp.alloc(&d_conn, (size_t)8 * mesh->n_elem) || tmp.alloc(&d_err, 1))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpyAsync(d_coord, mesh->coord, (size_t)3 * mesh->n_node * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_conn, mesh->conn, (size_t)8 * mesh->n_elem * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(d_err, 0, 4, c->stream));
  if (ensure_elem_colors(c, c->asm_colors, mesh->n_elem, mesh->conn, mesh->n_node) ||
      ensure_scatter_map(c, c->asm_colors, mesh->n_elem, d_conn, true))
    return FX_ERROR_RUNTIME;  // both cached per (profile, mesh)
  double D11 = 0.0, D12 = 0.0, D44 = 0.0;
  int32_t *d_emat = nullptr;
  double *d_mtab = nullptr;
  if (n_mat > 0) {  // several sections: per-element material id + a (D11, D12, D44) table
    std::vector<double> tab((size_t)3 * n_mat);
    for (int32_t k = 0; k < n_mat; k++) elastic_constants(Es[k], nus[k], tab[3 * k], tab[3 * k + 1], tab[3 * k + 2]);
    for (int32_t e = 0; e < mesh->n_elem; e++)
      if (elem_mat[e] < 1 || elem_mat[e] > n_mat) { g_fx_error = "fx_assemble_c3d8_sections: material id out of range"; return FX_ERROR_RUNTIME; }
    if (tmp.alloc(&d_emat, (size_t)mesh->n_elem) || tmp.alloc(&d_mtab, tab.size())) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpyAsync(d_emat, elem_mat, (size_t)mesh->n_elem * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_mtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // tab is a host temporary
  } else {
    elastic_constants(E, nu, D11, D12, D44);
  }
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  // hecmw_mat_clear (fstr_StiffMatrix.f90:40)
  if (!(c->asm_colors.first_write && c->asm_colors.pos && !c->asm_colors.offsets.empty())) {  // first-write scatter: every block is stored before it is added to
    HIP_TRY(hipMemsetAsync(A.D, 0, (size_t)9 * A.NP * 8, c->stream));
    HIP_TRY(hipMemsetAsync(A.AL, 0, (size_t)9 * A.NPL * 8, c->stream));
    HIP_TRY(hipMemsetAsync(A.AU, 0, (size_t)9 * A.NPU * 8, c->stream));
  }
  if (elemopt == 1) launch_assemble<1>(c, mesh->n_elem, d_coord, d_conn, D11, D12, D44, nullptr, d_err, d_emat, d_mtab, &c->asm_colors);
  else if (elemopt == 2) launch_assemble<2>(c, mesh->n_elem, d_coord, d_conn, D11, D12, D44, nullptr, d_err, d_emat, d_mtab, &c->asm_colors);
  else launch_assemble<3>(c, mesh->n_elem, d_coord, d_conn, D11, D12, D44, nullptr, d_err, d_emat, d_mtab, &c->asm_colors);
  HIP_TRY(hipGetLastError());
  if (load) HIP_TRY(hipMemcpyAsync(A.B, load, (size_t)3 * A.NP * 8, hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(hipMemsetAsync(A.B, 0, (size_t)3 * A.NP * 8, c->stream));
  if (n_bc > 0) {
    if (tmp.alloc(&d_flag, (size_t)3 * A.NP) || tmp.alloc(&d_bcv, (size_t)3 * A.NP) || tmp.alloc(&d_node, (size_t)n_bc) ||
        tmp.alloc(&d_dof, (size_t)n_bc) || tmp.alloc(&d_val, (size_t)n_bc))
      return FX_ERROR_RUNTIME;
    for (int32_t k = 0; k < n_bc; k++)
      if (bc_node[k] < 1 || bc_node[k] > A.NP) { g_fx_error = "fx_assemble_c3d8: BC node id out of range"; return FX_ERROR_RUNTIME; }
    HIP_TRY(hipMemsetAsync(d_flag, 0, (size_t)3 * A.NP, c->stream));
    HIP_TRY(hipMemsetAsync(d_bcv, 0, (size_t)3 * A.NP * 8, c->stream));
    HIP_TRY(hipMemcpyAsync(d_node, bc_node, (size_t)n_bc * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_dof, bc_dof, (size_t)n_bc * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_val, bc_val, (size_t)n_bc * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_bc_mark, dim3((n_bc + 255) / 256), dim3(256), 0, c->stream, n_bc, d_node, d_dof, d_val, d_flag, d_bcv);
    const dim3 g((A.NP + 255) / 256);
    hipLaunchKernelGGL((k_bc_apply<1>), g, dim3(256), 0, c->stream, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU,
                       A.B, d_flag, d_bcv);
    hipLaunchKernelGGL((k_bc_apply<2>), g, dim3(256), 0, c->stream, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU,
                       A.B, d_flag, d_bcv);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  int32_t herr = 0;  // the streaming (BELL) layouts re-gather these values on next use (ensure_solver)
  HIP_TRY(hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (ms_assemble) *ms_assemble = ms;
  if (herr == 1) { g_fx_error = "PIVOT ERROR in the incompatible-mode condensation (calInverse)"; return FX_ERROR_RUNTIME; }
  if (herr == 2) { g_fx_error = "###ERROR### : cannot find connectivity (element not covered by the profile)"; return FX_ERROR_RUNTIME; }
  c->have_values = true;
  c->bell_valid = false;   // the preconditioner is refreshed by the flags / recycle policy of the next solve, not here
  return 0;
}

extern "C" int fx_assemble_c3d8(fx_context *c, const fx_mesh_view *mesh, double E, double nu, int elemopt, const double *load,
                                int32_t n_bc, const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val,
                                float *ms_assemble) {
  return assemble_c3d8_common(c, mesh, E, nu, 0, nullptr, nullptr, nullptr, elemopt, load, n_bc, bc_node, bc_dof, bc_val,
                              ms_assemble);
}

extern "C" int fx_assemble_c3d8_sections(fx_context *c, const fx_mesh_view *mesh, int32_t n_mat, const double *E, const double *nu,
                                         const int32_t *elem_mat, int elemopt, const double *load, int32_t n_bc,
                                         const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val,
                                         float *ms_assemble) {
  if (n_mat < 1 || !E || !nu || !elem_mat) { g_fx_error = "fx_assemble_c3d8_sections: materials missing"; return FX_ERROR_RUNTIME; }
  return assemble_c3d8_common(c, mesh, 0.0, 0.0, n_mat, E, nu, elem_mat, elemopt, load, n_bc, bc_node, bc_dof, bc_val,
                              ms_assemble);
}

extern "C" int fx_element_stiffness_c3d8(fx_context *c, int elemopt, const double *ecoord, double E, double nu, double *stiff) {
  HIP_TRY(hipSetDevice(c->device));
  DevScratch tmp;
  double *d_coord = nullptr, *d_k = nullptr;
  int32_t *d_conn = nullptr, *d_err = nullptr;
  if (tmp.alloc(&d_coord, 24) || tmp.alloc(&d_conn, 8) || tmp.alloc(&d_k, 576) || tmp.alloc(&d_err, 1)) return FX_ERROR_RUNTIME;
  const int32_t conn[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  HIP_TRY(hipMemcpy(d_coord, ecoord, 24 * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_conn, conn, 32, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_err, 0, 4));
  double D11, D12, D44;
  elastic_constants(E, nu, D11, D12, D44);
  const int32_t *nul = nullptr;
  double *nud = nullptr;
#define ONE(EO)                                                                                                         \
  hipLaunchKernelGGL((k_assemble_c3d8<EO>), dim3(1), dim3(FXA_BS(EO)), 0, c->stream, 1, d_coord, d_conn, D11, D12, D44, nul, \
                     nul, nul, nul, nud, nud, nud, d_k, d_err, nul, (const double *)nullptr, nul, 0, nul)
  if (elemopt == 1) ONE(1);
  else if (elemopt == 2) ONE(2);
  else if (elemopt == 3) ONE(3);
  else { g_fx_error = "elemopt must be 1, 2 or 3"; return FX_ERROR_UNSUPPORTED; }
#undef ONE
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(stiff, d_k, 576 * 8, hipMemcpyDeviceToHost));
  return 0;
}
