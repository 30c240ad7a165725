/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the stress update of a LINEAR static analysis (`!SOLUTION, TYPE=STATIC`) for
 * TYPE=361 elements with an isotropic ELASTIC material, no thermal load, no material coordinate system:
 *   fstr_UpdateNewton                    fistr1/src/analysis/static/fstr_Update.f90:25-293 (element loop :73-276, QFORCE :258-264)
 *   UpdateST_C3D8IC   (ELEMOPT361 IC)    fistr1/src/lib/static_LIB_3dIC.f90:220-455
 *   Update_C3D8Bbar   (BBAR, INFINITE)   fistr1/src/lib/static_LIB_C3D8.f90:203-547
 *   UPDATE_C3         (FI, INFINITE)     fistr1/src/lib/static_LIB_3d.f90:516-837
 * Included by hecmw_oracle.c (same translation unit: shares the element helpers).  Pinned against the reference routines
 * themselves through oracle/ref_fem_driver.f90 (mode 3), fixture tests/golden/update_linear.npz. */

/* one element: edisp[24] = total displacement (u + du), out: strain / stress [8][6], qf[24] */
void orc_update_c3d8_linear(int elemopt, const double *ecoord, const double *edisp, double E, double nu, double *strain,
                            double *stress, double *qf) {
  double D[6][6], lc[3], det;
  elastic_matrix(E, nu, D);
  if (elemopt == 1) { /* UpdateST_C3D8IC */
    double stiff[33 * 33], gd[11][3], B[6 * 33];
    double XJ[3][3], inv0[3][3], deriv[8][3], det0;
    memset(stiff, 0, sizeof stiff);
    lc[0] = lc[1] = lc[2] = 0.0;
    jacobian_hex8(lc, ecoord, &det0, XJ, inv0, deriv); /* :268-270 */
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) inv0[i][j] *= det0;
    for (int LX = 0; LX < 8; LX++) { /* :277-326: [Kdd Kda; Kad Kaa] */
      quad_point(LX, lc);
      global_deriv_hex8(lc, ecoord, &det, gd);
      for (int d = 0; d < 3; d++) {
        gd[8][d] = -2.0 * lc[0] * inv0[0][d] / det;
        gd[9][d] = -2.0 * lc[1] * inv0[1][d] / det;
        gd[10][d] = -2.0 * lc[2] * inv0[2][d] / det;
      }
      fill_B(11, gd, B, 33);
      add_BtDB(33, B, D, 1.0 * det, stiff);
    }
    double xj[81], tmpforce[9], cdisp[33];
    for (int i = 0; i < 9; i++)
      for (int j = 0; j < 9; j++) xj[j * 9 + i] = stiff[(24 + i) * 33 + (24 + j)]; /* :327, column-major for cal_inverse */
    cal_inverse(9, xj);                                                            /* :328 */
    for (int i = 0; i < 9; i++) { /* :330 [Kda]*edisp */
      double s = 0.0;
      for (int j = 0; j < 24; j++) s += stiff[(24 + i) * 33 + j] * edisp[j];
      tmpforce[i] = s;
    }
    for (int j = 0; j < 24; j++) cdisp[j] = edisp[j];
    for (int i = 0; i < 9; i++) { /* :333 -[Kaa]^-1 [Kda] edisp */
      double s = 0.0;
      for (int j = 0; j < 9; j++) s += xj[j * 9 + i] * tmpforce[j];
      cdisp[24 + i] = -s;
    }
    for (int i = 0; i < 24; i++) { /* :336 */
      double s = 0.0;
      for (int j = 0; j < 33; j++) s += stiff[i * 33 + j] * cdisp[j];
      qf[i] = s;
    }
    for (int LX = 0; LX < 8; LX++) { /* :344-451 */
      quad_point(LX, lc);
      global_deriv_hex8(lc, ecoord, &det, gd);
      for (int d = 0; d < 3; d++) {
        gd[8][d] = -2.0 * lc[0] * inv0[0][d] / det;
        gd[9][d] = -2.0 * lc[1] * inv0[1][d] / det;
        gd[10][d] = -2.0 * lc[2] * inv0[2][d] / det;
      }
      fill_B(11, gd, B, 33);
      for (int r = 0; r < 6; r++) { /* :433 EPSA = B cdisp */
        double s = 0.0;
        for (int j = 0; j < 33; j++) s += B[r * 33 + j] * cdisp[j];
        strain[6 * LX + r] = s;
      }
      for (int r = 0; r < 6; r++) { /* :437-442 */
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += D[r][k] * strain[6 * LX + k];
        stress[6 * LX + r] = s;
      }
    }
    return;
  }
  /* FI (UPDATE_C3) and B-bar (Update_C3D8Bbar), nlgeom_flag = INFINITE */
  double gd[8][3], Bbar[8][3], vol0 = 0.0;
  memset(qf, 0, 24 * sizeof(double));
  if (elemopt == 2) { /* dilatation at centroid, C3D8.f90:271-275 */
    lc[0] = lc[1] = lc[2] = 0.0;
    global_deriv_hex8(lc, ecoord, &det, Bbar);
    double tr = 0.0;
    for (int d = 0; d < 3; d++) {
      double s = 0.0;
      for (int a = 0; a < 8; a++) s += edisp[3 * a + d] * Bbar[a][d];
      tr += s;
    }
    vol0 = tr / 3.0;
  }
  for (int LX = 0; LX < 8; LX++) {
    quad_point(LX, lc);
    global_deriv_hex8(lc, ecoord, &det, gd);
    double g[3][3]; /* gdispderiv = matmul(totaldisp, gderiv) */
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0.0;
        for (int a = 0; a < 8; a++) s += edisp[3 * a + i] * gd[a][j];
        g[i][j] = s;
      }
    double dvol = (elemopt == 2) ? vol0 - (g[0][0] + g[1][1] + g[2][2]) / 3.0 : 0.0;
    double *e = &strain[6 * LX], *sg = &stress[6 * LX];
    e[0] = g[0][0] + dvol; e[1] = g[1][1] + dvol; e[2] = g[2][2] + dvol;
    e[3] = g[0][1] + g[1][0]; e[4] = g[1][2] + g[2][1]; e[5] = g[2][0] + g[0][2];
    for (int r = 0; r < 6; r++) {
      double s = 0.0;
      for (int k = 0; k < 6; k++) s += D[r][k] * e[k];
      sg[r] = s;
    }
    double B[6 * 24];
    if (elemopt == 2) {
      memset(B, 0, sizeof B);
      for (int j = 0; j < 8; j++) { /* C3D8.f90:458-481 */
        double B4 = (Bbar[j][0] - gd[j][0]) / 3.0, B6 = (Bbar[j][1] - gd[j][1]) / 3.0, B8 = (Bbar[j][2] - gd[j][2]) / 3.0;
        B[0 * 24 + 3 * j] = gd[j][0] + B4; B[0 * 24 + 3 * j + 1] = B6; B[0 * 24 + 3 * j + 2] = B8;
        B[1 * 24 + 3 * j] = B4; B[1 * 24 + 3 * j + 1] = gd[j][1] + B6; B[1 * 24 + 3 * j + 2] = B8;
        B[2 * 24 + 3 * j] = B4; B[2 * 24 + 3 * j + 1] = B6; B[2 * 24 + 3 * j + 2] = gd[j][2] + B8;
        B[3 * 24 + 3 * j] = gd[j][1]; B[3 * 24 + 3 * j + 1] = gd[j][0];
        B[4 * 24 + 3 * j + 1] = gd[j][2]; B[4 * 24 + 3 * j + 2] = gd[j][1];
        B[5 * 24 + 3 * j] = gd[j][2]; B[5 * 24 + 3 * j + 2] = gd[j][0];
      }
    } else {
      fill_B(8, gd, B, 24);
    }
    const double wg = 1.0 * det;
    for (int j = 0; j < 24; j++) { /* qf += matmul(stress, B) * wg */
      double s = 0.0;
      for (int r = 0; r < 6; r++) s += sg[r] * B[r * 24 + j];
      qf[j] += s * wg;
    }
  }
}

/* fstr_UpdateNewton's element loop for one TYPE=361 group (several materials: elem_mat 1-based, NULL = material 1):
 * strain / stress [n_elem][8][6], qforce[3*n_node] (zeroed here as :52 does). */
void orc_update_linear(int elemopt, int32_t n_node, int32_t n_elem, const double *coord, const int32_t *conn, const double *E,
                       const double *nu, const int32_t *elem_mat, const double *disp, double *strain, double *stress,
                       double *qforce) {
  memset(qforce, 0, (size_t)3 * n_node * sizeof(double));
  for (int32_t e = 0; e < n_elem; e++) {
    double ec[24], ed[24], qf[24];
    const int32_t *nod = &conn[(size_t)8 * e];
    for (int j = 0; j < 8; j++)
      for (int d = 0; d < 3; d++) {
        ec[3 * j + d] = coord[3 * (size_t)(nod[j] - 1) + d];
        ed[3 * j + d] = disp[3 * (size_t)(nod[j] - 1) + d];
      }
    const int m = elem_mat ? elem_mat[e] - 1 : 0;
    orc_update_c3d8_linear(elemopt, ec, ed, E[m], nu[m], &strain[(size_t)48 * e], &stress[(size_t)48 * e], qf);
    for (int j = 0; j < 8; j++)
      for (int d = 0; d < 3; d++) qforce[3 * (size_t)(nod[j] - 1) + d] += qf[3 * j + d];
  }
}
