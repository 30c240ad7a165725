/* TEST INFRASTRUCTURE ONLY -- never linked into the product (included at the end of hecmw_oracle.c).
 *
 * CPU restatement of the reference's nonlinear C3D8 B-bar path for one isotropic Mises material:
 *   tables            fetch_TableData / fetch_TableGrad   fistr1/src/lib/utilities/ttable.f90:183-363
 *   hardening         calHardenCoeff :175, calCurrYield :254, calYieldFunc :295   physics/Elastoplastic.f90
 *   tangent           calElastoPlasticMatrix :16-117, MatlMatrix physics/calMatMatrix.f90:28-113
 *   return mapping    BackwardEuler :351-558 (Mises branch :423-459)
 *   element tangent   STF_C3D8Bbar     static_LIB_C3D8.f90:23-200  (+ GEOMAT_C3 static_LIB_3d.f90:15-37)
 *   stress update     Update_C3D8Bbar  static_LIB_C3D8.f90:203-547
 *   state commit      fstr_UpdateState analysis/static/fstr_Update.f90:296-345
 * Pinned against the reference routines themselves (oracle/ref_nl_driver.f90 -> oracle/_ref/ref_nl):
 * tests/test_oracle_nl.py.
 *
 * One reference peculiarity is restated on purpose: MatlMatrix declares `integer :: flag = 0`
 * (calMatMatrix.f90:39), which in Fortran is an implicitly SAVEd variable.  The first call with
 * isEp=1 -- every Update_C3D8Bbar of an elastoplastic material -- latches flag=1 for the rest of
 * the process, and from then on MatlMatrix returns the *elastic* matrix for elastoplastic materials
 * too (the `isElastic(..) .or. flag==1` branch :56).  The reference's Newton loop therefore runs
 * with the elastic tangent (minus the geometric terms) after its first stress update.  The latch is
 * state of the oracle (orc_nl_latch) and of the fx_context on the GPU side.
 */

static int g_matl_flag = 0;
void orc_nl_reset_latch(void) { g_matl_flag = 0; }
int orc_nl_latch(void) { return g_matl_flag; }

/* GetTableData, 1-D table (ndepends=1, tbcol=2): ttable.f90:320-335 */
static double table_value(const orc_material *m, double a) {
  int n = m->ntab;
  const double *t = m->tab; /* (yield, pstrain) rows */
  if (a < t[1]) return t[0];
  if (a >= t[2 * (n - 1) + 1]) return t[2 * (n - 1)];
  for (int i = 0; i < n - 1; i++)
    if (a >= t[2 * i + 1] && a < t[2 * (i + 1) + 1]) {
      double lambda = (a - t[2 * i + 1]) / (t[2 * (i + 1) + 1] - t[2 * i + 1]);
      return (1.0 - lambda) * t[2 * i] + lambda * t[2 * (i + 1)];
    }
  return t[2 * (n - 1)];
}

/* GetTableGrad, ttable.f90:221-235 */
static double table_grad(const orc_material *m, double a) {
  int n = m->ntab;
  const double *t = m->tab;
  if (a < t[1]) return 0.0;
  if (a >= t[2 * (n - 1) + 1]) return 0.0;
  for (int i = 0; i < n - 1; i++)
    if (a >= t[2 * i + 1] && a < t[2 * (i + 1) + 1])
      return (t[2 * (i + 1)] - t[2 * i]) / (t[2 * (i + 1) + 1] - t[2 * i + 1]);
  return 0.0;
}

/* calCurrYield, Elastoplastic.f90:254-292 */
double orc_curr_yield(const orc_material *m, double pstrain) {
  double s0 = m->plconst[0], s1 = m->plconst[1], s2 = m->plconst[2];
  switch (m->harden) {
    case 0: return s0 + s1 * pstrain;
    case 1: return table_value(m, pstrain);
    case 2: return s1 * pow(s0 + pstrain, s2);
    case 3: return (pstrain <= s0) ? s1 : s1 * pow(pstrain / s0, 1.0 / s2);
  }
  return -1.0;
}

/* calHardenCoeff, Elastoplastic.f90:175-220 */
double orc_harden_coeff(const orc_material *m, double pstrain) {
  double s0 = m->plconst[0], s1 = m->plconst[1], s2 = m->plconst[2];
  switch (m->harden) {
    case 0: return s1;
    case 1: return table_grad(m, pstrain);
    case 2: return s1 * s2 * pow(s0 + pstrain, s2 - 1.0);
    case 3: {
      double ef = orc_curr_yield(m, pstrain);
      return s1 * pow(ef / s1, 1.0 - s2) / (s0 * s2);
    }
  }
  return -1.0;
}

/* calElastoPlasticMatrix (Mises, isotropic hardening), Elastoplastic.f90:16-117.  D row-major 6x6. */
void orc_elastoplastic_matrix(const orc_material *m, const double *stress, int istat, double extval1,
                              double *D) {
  double De[6][6], devia[6], dj2[6], a[6], da[6];
  elastic_matrix(m->E, m->nu, De);
  double J1 = stress[0] + stress[1] + stress[2];
  for (int i = 0; i < 3; i++) devia[i] = stress[i] - J1 / 3.0;
  for (int i = 3; i < 6; i++) devia[i] = stress[i];
  double J2 = 0.5 * (devia[0] * devia[0] + devia[1] * devia[1] + devia[2] * devia[2]) +
              (devia[3] * devia[3] + devia[4] * devia[4] + devia[5] * devia[5]);
  memcpy(D, De, sizeof De);
  if (istat == 0) return;
  for (int i = 0; i < 3; i++) dj2[i] = devia[i];
  for (int i = 3; i < 6; i++) dj2[i] = 2.0 * devia[i];
  for (int i = 0; i < 6; i++) dj2[i] = dj2[i] / (2.0 * sqrt(J2));
  double harden = orc_harden_coeff(m, extval1);
  for (int i = 0; i < 6; i++) a[i] = sqrt(3.0) * dj2[i];
  for (int i = 0; i < 6; i++) {
    double s = 0.0;
    for (int j = 0; j < 6; j++) s += De[i][j] * a[j];
    da[i] = s;
  }
  double dum = 0.0;
  for (int i = 0; i < 6; i++) dum += da[i] * a[i];
  dum = harden + 0.0 + dum;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) D[6 * i + j] = De[i][j] - da[i] * da[j] / dum;
}

/* MatlMatrix for the two material kinds on the path (calMatMatrix.f90:28-113) incl. the latch */
static void matl_matrix(const orc_material *m, const double *stress, int istat, double fstat1, int isEp,
                        double D[6][6]) {
  if (isEp == 1) g_matl_flag = 1;
  if (!m->plastic || g_matl_flag == 1)
    elastic_matrix(m->E, m->nu, D);
  else
    orc_elastoplastic_matrix(m, stress, istat, fstat1, &D[0][0]);
}

/* BackwardEuler, Mises branch: Elastoplastic.f90:351-459, :557 */
void orc_backward_euler(const orc_material *m, double *stress, double plstrain, int32_t *istat, double *fstat1) {
  const double tol = 1.0e-3;
  double devia[6];
  double pstrain = plstrain;
  /* calYieldFunc(matl, stress, fstat_bak): fstat_bak(:) = plstrain  (:295-348) */
  double J2;
  {
    double J1 = stress[0] + stress[1] + stress[2];
    for (int i = 0; i < 3; i++) devia[i] = stress[i] - J1 / 3.0;
    for (int i = 3; i < 6; i++) devia[i] = stress[i];
  }
  J2 = 0.5 * (devia[0] * devia[0] + devia[1] * devia[1] + devia[2] * devia[2]) +
       (devia[3] * devia[3] + devia[4] * devia[4] + devia[5] * devia[5]);
  double f = sqrt(3.0 * J2) - orc_curr_yield(m, pstrain);
  if (fabs(f) < tol) { *istat = 1; return; }
  if (f < 0.0) { *istat = 0; return; }
  *istat = 1;
  double J1 = (stress[0] + stress[1] + stress[2]) / 3.0;
  for (int i = 0; i < 3; i++) devia[i] = stress[i] - J1;
  for (int i = 3; i < 6; i++) devia[i] = stress[i];
  /* cal_equivalent_stress :120-163 (its own deviator with J1/3) */
  double yd;
  {
    double dv[6];
    double s1 = stress[0] + stress[1] + stress[2];
    for (int i = 0; i < 3; i++) dv[i] = stress[i] - s1 / 3.0;
    for (int i = 3; i < 6; i++) dv[i] = stress[i];
    double j2 = 0.5 * (dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]) + (dv[3] * dv[3] + dv[4] * dv[4] + dv[5] * dv[5]);
    yd = sqrt(3.0 * j2);
  }
  double G = m->E / (2.0 * (1.0 + m->nu));
  double dlambda = 0.0;
  for (int i = 1; i <= 5; i++) {
    double H = orc_harden_coeff(m, pstrain + dlambda);
    double dd = 3.0 * G + H + 0.0;
    dlambda = dlambda + f / dd;
    if (dlambda < 0.0) { dlambda = 0.0; *istat = 0; break; }
    double dum = orc_curr_yield(m, pstrain + dlambda);
    f = yd - 3.0 * G * dlambda - dum - (0.0 - 0.0);
    if (fabs(f) < tol * tol) break;
  }
  pstrain = pstrain + dlambda;
  double fac = 1.0 - 3.0 * dlambda * G / yd;
  for (int i = 0; i < 6; i++) devia[i] = fac * devia[i];
  for (int i = 0; i < 3; i++) stress[i] = devia[i] + J1;
  for (int i = 3; i < 6; i++) stress[i] = devia[i];
  for (int i = 0; i < 6; i++) stress[i] = stress[i] + 0.0;
  *fstat1 = pstrain;
}

/* GEOMAT_C3, static_LIB_3d.f90:15-37 */
static void geomat_c3(const double *s, double mat[6][6]) {
  memset(mat, 0, 36 * sizeof(double));
  mat[0][0] = 2.0 * s[0]; mat[0][3] = s[3]; mat[0][5] = s[5];
  mat[1][1] = 2.0 * s[1]; mat[1][3] = s[3]; mat[1][4] = s[4];
  mat[2][2] = 2.0 * s[2]; mat[2][4] = s[4]; mat[2][5] = s[5];
  mat[3][0] = mat[0][3]; mat[3][1] = mat[1][3]; mat[3][2] = mat[2][3];
  mat[3][3] = 0.5 * (s[0] + s[1]); mat[3][4] = 0.5 * s[5]; mat[3][5] = 0.5 * s[4];
  mat[4][0] = mat[0][4]; mat[4][1] = mat[1][4]; mat[4][2] = mat[2][4];
  mat[4][3] = mat[3][4]; mat[4][4] = 0.5 * (s[2] + s[1]); mat[4][5] = 0.5 * s[3];
  mat[5][0] = mat[0][5]; mat[5][1] = mat[1][5]; mat[5][2] = mat[2][5];
  mat[5][3] = mat[3][5]; mat[5][4] = mat[4][5]; mat[5][5] = 0.5 * (s[0] + s[2]);
}

static void fill_Bbar(double Bbar[8][3], double gd[8][3], double *B) {
  memset(B, 0, 6 * 24 * sizeof(double));
  for (int j = 0; j < 8; j++) {
    double B4 = (Bbar[j][0] - gd[j][0]) / 3.0, B6 = (Bbar[j][1] - gd[j][1]) / 3.0,
           B8 = (Bbar[j][2] - gd[j][2]) / 3.0;
    B[0 * 24 + 3 * j] = gd[j][0] + B4; B[0 * 24 + 3 * j + 1] = B6; B[0 * 24 + 3 * j + 2] = B8;
    B[1 * 24 + 3 * j] = B4; B[1 * 24 + 3 * j + 1] = gd[j][1] + B6; B[1 * 24 + 3 * j + 2] = B8;
    B[2 * 24 + 3 * j] = B4; B[2 * 24 + 3 * j + 1] = B6; B[2 * 24 + 3 * j + 2] = gd[j][2] + B8;
    B[3 * 24 + 3 * j] = gd[j][1]; B[3 * 24 + 3 * j + 1] = gd[j][0];
    B[4 * 24 + 3 * j + 1] = gd[j][2]; B[4 * 24 + 3 * j + 2] = gd[j][1];
    B[5 * 24 + 3 * j] = gd[j][2]; B[5 * 24 + 3 * j + 2] = gd[j][0];
  }
}

/* BL1 of the total-Lagrange branch (STF :131-158 == Update :481-505), g = gdispderiv(i,j) */
static void add_B1(double g[3][3], double gd[8][3], double *B) {
  for (int j = 0; j < 8; j++) {
    for (int c = 0; c < 3; c++) {
      B[0 * 24 + 3 * j + c] += g[c][0] * gd[j][0];
      B[1 * 24 + 3 * j + c] += g[c][1] * gd[j][1];
      B[2 * 24 + 3 * j + c] += g[c][2] * gd[j][2];
      B[3 * 24 + 3 * j + c] += g[c][1] * gd[j][0] + g[c][0] * gd[j][1];
      B[4 * 24 + 3 * j + c] += g[c][1] * gd[j][2] + g[c][2] * gd[j][1];
      B[5 * 24 + 3 * j + c] += g[c][2] * gd[j][0] + g[c][0] * gd[j][2];
    }
  }
}

/* STF_C3D8Bbar with displacement: static_LIB_C3D8.f90:23-200.  ecoord, u: 8x3 (node-major);
 * stress 8x6, istat/fstat 8 (per quadrature point); stiff row-major 24x24. */
void orc_stf_c3d8bbar_nl(const orc_material *m, const double *ecoord, const double *u, const double *stress,
                         const int32_t *istat, const double *fstat, double *stiff) {
  int flag = u ? m->nlgeom : 0;
  double elem[24], lc[3], det, Bbar[8][3], gd[8][3], B[6 * 24], D[6][6], mat[6][6];
  memset(stiff, 0, 576 * sizeof(double));
  for (int i = 0; i < 24; i++) elem[i] = ecoord[i];
  if (flag == 2) for (int i = 0; i < 24; i++) elem[i] = ecoord[i] + u[i];
  lc[0] = lc[1] = lc[2] = 0.0;
  global_deriv_hex8(lc, elem, &det, Bbar);
  for (int LX = 0; LX < 8; LX++) {
    quad_point(LX, lc);
    global_deriv_hex8(lc, elem, &det, gd);
    matl_matrix(m, stress + 6 * LX, istat ? istat[LX] : 0, fstat ? fstat[LX] : 0.0, 0, D);
    if (flag == 2) {
      geomat_c3(stress + 6 * LX, mat);
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) D[i][j] = D[i][j] - mat[i][j];
    }
    double wg = 1.0 * det;
    fill_Bbar(Bbar, gd, B);
    if (flag == 1) {
      double g[3][3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
          for (int a = 0; a < 8; a++) s += u[3 * a + i] * gd[a][j];
          g[i][j] = s;
        }
      add_B1(g, gd, B);
    }
    add_BtDB(24, B, D, wg, stiff);
    if (flag == 1 || flag == 2) { /* initial stress matrix :164-195 */
      const double *s = stress + 6 * LX;
      double S[3][3] = {{s[0], s[3], s[5]}, {s[3], s[1], s[4]}, {s[5], s[4], s[2]}};
      /* BN(3*(d-1)+c, 3*j+c) = gderiv(j,d);  Smat(c+3(d-1), c+3(e-1)) = S(d,e) */
      for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) {
          int ni = i / 3, ci = i % 3, nj = j / 3, cj = j % 3;
          if (ci != cj) continue;
          double acc = 0.0;
          for (int d = 0; d < 3; d++) {
            double sbn = 0.0; /* SBN(3*d+c, j) = sum_e S(d,e) gderiv(nj,e) */
            for (int e = 0; e < 3; e++) sbn += S[d][e] * gd[nj][e];
            acc += gd[ni][d] * sbn;
          }
          stiff[24 * i + j] += acc * wg;
        }
    }
  }
}

/* Update_C3D8Bbar: static_LIB_C3D8.f90:203-547 (no temperature).  State of the element's 8 points:
 * stress/strain (out), stress_bak/strain_bak/plstrain (in), istat/fstat (inout); qf[24]. */
void orc_update_c3d8bbar(const orc_material *m, const double *ecoord, const double *u, const double *du,
                         double *stress, double *strain, const double *stress_bak, const double *strain_bak,
                         const double *plstrain, int32_t *istat, double *fstat, double *qf) {
  int flag = m->nlgeom;
  double elem[24], elem1[24], totaldisp[24], lc[3], det, Bbar[8][3], Bbar2[8][3], gd[8][3], B[6 * 24], D[6][6];
  memset(qf, 0, 24 * sizeof(double));
  for (int i = 0; i < 24; i++) { elem[i] = ecoord[i]; totaldisp[i] = u[i] + du[i]; }
  if (flag == 2)
    for (int i = 0; i < 24; i++) {
      elem[i] = (0.5 * du[i] + u[i]) + ecoord[i];
      elem1[i] = (du[i] + u[i]) + ecoord[i];
      totaldisp[i] = du[i];
    }
  lc[0] = lc[1] = lc[2] = 0.0;
  global_deriv_hex8(lc, elem, &det, Bbar);
  double dd[3];
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
    for (int a = 0; a < 8; a++) s += totaldisp[3 * a + i] * Bbar[a][i];
    dd[i] = s;
  }
  double vol0 = (dd[0] + dd[1] + dd[2]) / 3.0;
  if (flag == 2) global_deriv_hex8(lc, elem1, &det, Bbar2);
  for (int LX = 0; LX < 8; LX++) {
    quad_point(LX, lc);
    global_deriv_hex8(lc, elem, &det, gd);
    double g[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0.0;
        for (int a = 0; a < 8; a++) s += totaldisp[3 * a + i] * gd[a][j];
        g[i][j] = s;
      }
    double dvol = vol0 - (g[0][0] + g[1][1] + g[2][2]) / 3.0;
    int isEp = m->plastic ? 1 : 0;
    double *sg = stress + 6 * LX, *eg = strain + 6 * LX;
    matl_matrix(m, sg, istat[LX], fstat[LX], isEp, D);
    double dstrain[6], dstress[6];
    dstrain[0] = g[0][0] + dvol; dstrain[1] = g[1][1] + dvol; dstrain[2] = g[2][2] + dvol;
    dstrain[3] = g[0][1] + g[1][0]; dstrain[4] = g[1][2] + g[2][1]; dstrain[5] = g[2][0] + g[0][2];
    for (int i = 0; i < 6; i++) dstrain[i] = dstrain[i] - 0.0;
    if (flag == 0) {
      for (int i = 0; i < 6; i++) eg[i] = dstrain[i] + 0.0;
      for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int j = 0; j < 6; j++) s += D[i][j] * dstrain[j];
        sg[i] = s;
      }
    } else if (flag == 1) {
      for (int c = 0; c < 3; c++)
        dstrain[c] = dstrain[c] + 0.5 * (g[0][c] * g[0][c] + g[1][c] * g[1][c] + g[2][c] * g[2][c]);
      dstrain[3] = dstrain[3] + (g[0][0] * g[0][1] + g[1][0] * g[1][1] + g[2][0] * g[2][1]);
      dstrain[4] = dstrain[4] + (g[0][1] * g[0][2] + g[1][1] * g[1][2] + g[2][1] * g[2][2]);
      dstrain[5] = dstrain[5] + (g[0][0] * g[0][2] + g[1][0] * g[1][2] + g[2][0] * g[2][2]);
      for (int i = 0; i < 6; i++) eg[i] = dstrain[i] + 0.0;
      for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int j = 0; j < 6; j++) s += D[i][j] * dstrain[j];
        sg[i] = s;
      }
    } else {
      const double *sb = stress_bak + 6 * LX, *eb = strain_bak + 6 * LX;
      double rot[3][3] = {{0}}, S[3][3], dum[3][3];
      rot[0][1] = 0.5 * (g[0][1] - g[1][0]); rot[1][0] = -rot[0][1];
      rot[1][2] = 0.5 * (g[1][2] - g[2][1]); rot[2][1] = -rot[1][2];
      rot[0][2] = 0.5 * (g[0][2] - g[2][0]); rot[2][0] = -rot[0][2];
      for (int i = 0; i < 6; i++) eg[i] = eb[i] + dstrain[i] + 0.0;
      for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int j = 0; j < 6; j++) s += D[i][j] * dstrain[j];
        dstress[i] = s;
      }
      S[0][0] = sb[0]; S[1][1] = sb[1]; S[2][2] = sb[2];
      S[0][1] = S[1][0] = sb[3]; S[1][2] = S[2][1] = sb[4]; S[2][0] = S[0][2] = sb[5];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double a = 0.0, b = 0.0;
          for (int k = 0; k < 3; k++) { a += rot[i][k] * S[k][j]; b += S[i][k] * rot[k][j]; }
          dum[i][j] = a - b;
        }
      sg[0] = sb[0] + dstress[0] + dum[0][0] - sb[0] * 3.0 * vol0;
      sg[1] = sb[1] + dstress[1] + dum[1][1] - sb[1] * 3.0 * vol0;
      sg[2] = sb[2] + dstress[2] + dum[2][2] - sb[2] * 3.0 * vol0;
      sg[3] = sb[3] + dstress[3] + dum[0][1] - sb[3] * 3.0 * vol0;
      sg[4] = sb[4] + dstress[4] + dum[1][2] - sb[4] * 3.0 * vol0;
      sg[5] = sb[5] + dstress[5] + dum[2][0] - sb[5] * 3.0 * vol0;
    }
    if (m->plastic) orc_backward_euler(m, sg, plstrain[LX], &istat[LX], &fstat[LX]);
    /* internal force */
    fill_Bbar(Bbar, gd, B);
    if (flag == 1) {
      add_B1(g, gd, B);
    } else if (flag == 2) {
      global_deriv_hex8(lc, elem1, &det, gd);
      fill_Bbar(Bbar2, gd, B);
    }
    double wg = 1.0 * det;
    for (int j = 0; j < 24; j++) {
      double s = 0.0;
      for (int i = 0; i < 6; i++) s += sg[i] * B[i * 24 + j];
      qf[j] = qf[j] + s * wg;
    }
  }
}

/* fstr_StiffMatrix.f90:38-207 for one TYPE=361 B-bar group: clear, element tangents, scatter. */
/* Several sections (hecMESH%section_ID -> fstrSOLID%materials, fstr_setup.f90:325-400): when set, element e uses
 * mats[elem_mat[e] - 1] and the `m` argument of the three loops below is ignored.  NULL switches back to one material. */
static const orc_material *g_mats = NULL;
static const int32_t *g_emat = NULL;
void orc_nl_set_sections(const orc_material *mats, const int32_t *elem_mat) { g_mats = mats; g_emat = elem_mat; }
#define MAT_OF(e) (g_mats ? &g_mats[g_emat[e] - 1] : m)

void orc_nl_stiffness(const orc_material *m, int32_t NP, int32_t n_elem, const double *coord, const int32_t *conn,
                      const double *unode, const double *dunode, const orc_gauss_state *st,
                      const int32_t *indexL, const int32_t *itemL, const int32_t *indexU, const int32_t *itemU,
                      double *D, double *AL, double *AU) {
  memset(D, 0, (size_t)9 * NP * sizeof(double));
  memset(AL, 0, (size_t)9 * indexL[NP] * sizeof(double));
  memset(AU, 0, (size_t)9 * indexU[NP] * sizeof(double));
  double ec[24], u[24], stiff[576];
  for (int32_t e = 0; e < n_elem; e++) {
    const int32_t *nd = conn + 8 * e;
    for (int j = 0; j < 8; j++)
      for (int i = 0; i < 3; i++) {
        ec[3 * j + i] = coord[3 * (nd[j] - 1) + i];
        u[3 * j + i] = unode[3 * (nd[j] - 1) + i] + dunode[3 * (nd[j] - 1) + i];
      }
    orc_stf_c3d8bbar_nl(MAT_OF(e), ec, u, st->stress + 48 * e, st->istat + 8 * e, st->fstat + 8 * e, stiff);
    orc_mat_ass_elem(NP, indexL, itemL, indexU, itemU, D, AL, AU, 8, nd, stiff);
  }
}

/* fstr_UpdateNewton, fstr_Update.f90:25-293 (TYPE=361 B-bar branch :155-163, scatter :262-268) */
void orc_nl_update(const orc_material *m, int32_t n_node, int32_t n_elem, const double *coord, const int32_t *conn,
                   const double *unode, const double *dunode, orc_gauss_state *st, double *qforce) {
  memset(qforce, 0, (size_t)3 * n_node * sizeof(double));
  double ec[24], u[24], du[24], qf[24];
  for (int32_t e = 0; e < n_elem; e++) {
    const int32_t *nd = conn + 8 * e;
    for (int j = 0; j < 8; j++)
      for (int i = 0; i < 3; i++) {
        ec[3 * j + i] = coord[3 * (nd[j] - 1) + i];
        u[3 * j + i] = unode[3 * (nd[j] - 1) + i];
        du[3 * j + i] = dunode[3 * (nd[j] - 1) + i];
      }
    orc_update_c3d8bbar(MAT_OF(e), ec, u, du, st->stress + 48 * e, st->strain + 48 * e, st->stress_bak + 48 * e,
                        st->strain_bak + 48 * e, st->plstrain + 8 * e, st->istat + 8 * e, st->fstat + 8 * e, qf);
    for (int j = 0; j < 8; j++)
      for (int i = 0; i < 3; i++) qforce[3 * (nd[j] - 1) + i] += qf[3 * j + i];
  }
}

/* fstr_UpdateState, fstr_Update.f90:296-345 + updateEPState Elastoplastic.f90:563-567 */
void orc_nl_commit(const orc_material *m, int32_t n_elem, orc_gauss_state *st) {
  for (int64_t k = 0; k < (int64_t)8 * n_elem; k++) {
    if (MAT_OF(k / 8)->plastic) st->plstrain[k] = st->fstat[k];
    for (int i = 0; i < 6; i++) {
      st->strain_bak[6 * k + i] = st->strain[6 * k + i];
      st->stress_bak[6 * k + i] = st->stress[6 * k + i];
    }
  }
}
