/* TEST INFRASTRUCTURE ONLY -- included by hecmw_oracle.c.  CPU restatement of the reference's other two
 * Krylov methods: hecmw_solve_GMRES (hecmw_solver_GMRES.f90:17-458, restarted GMRES(m) with modified
 * Gram-Schmidt and Givens rotations, right preconditioning) and hecmw_solve_GPBiCG
 * (hecmw_solver_GPBiCG.f90:17-505, pol_coef_vanilla2 :457-503).  Pinned through oracle/_ref/ref_solve
 * (the reference's own hecmw_solve with METHOD=3/4): tests/test_oracle_golden.py. */

/* [H]{y} = {s}, {x} += Minv (V y): GMRES.f90:264-292 (= :301-329, :357-424) */
static void gmres_update_x(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax, int32_t NNDOF,
                           int IROW, int NRK, const double *H, const double *S, double **V, double *AV, double *ZQ,
                           double *ZP, double *X) {
  double *SS = (double *)calloc((size_t)NRK + 1, sizeof(double)), *Y = (double *)calloc((size_t)NRK + 1, sizeof(double));
#define HH(i, j) H[((size_t)(i)-1) * NRK + ((j)-1)]
  for (int ik = 1; ik <= IROW; ik++) SS[ik] = S[ik];
  Y[IROW] = SS[IROW] / HH(IROW, IROW);
  for (int kk = IROW - 1; kk >= 1; kk--) {
    for (int jj = IROW; jj >= kk + 1; jj--) SS[kk] = SS[kk] - HH(kk, jj) * Y[jj];
    Y[kk] = SS[kk] / HH(kk, kk);
  }
  for (int32_t kk = 0; kk < NNDOF; kk++) AV[kk] = 0.0;
  for (int jj = 1; jj <= IROW; jj++)
    for (int32_t kk = 0; kk < NNDOF; kk++) AV[kk] = AV[kk] + Y[jj] * V[jj][kk];
  orc_precond_apply(A, c, P, iterPREmax, AV, ZQ, ZP);
  for (int32_t kk = 0; kk < NNDOF; kk++) X[kk] = X[kk] + ZQ[kk];
#undef HH
  free(SS); free(Y);
}

int orc_solve_gmres(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax, const double *B,
                    double *X, int MAXIT, double TOL, int NREST, int *iter_out, double *resid_out, double *hist,
                    int *nhist_out) {
  int32_t N = A->N, NP = A->NP, NNDOF = ORC_ND(A) * N;
  size_t len = (size_t)ORC_ND(A) * NP;
  if (NREST >= ORC_ND(A) * NP - 1) NREST = ORC_ND(A) * NP - 2;                              /* :88 */
  const int NRK = NREST + 7;
  double *H = (double *)calloc((size_t)NRK * NRK, sizeof(double));
  double *S = (double *)calloc((size_t)NRK + 2, sizeof(double));            /* the leading entries of WW(:,S) */
  double *WW = (double *)calloc((size_t)(NREST + 7) * len, sizeof(double));
  double *R = WW, *ZP = WW + len, *ZQ = WW + 2 * len, *W = WW + 3 * len, *AV = WW + 4 * len;
  double **V = (double **)calloc((size_t)NREST + 3, sizeof(double *));
  for (int k = 1; k <= NREST + 1; k++) V[k] = WW + (size_t)(4 + k) * len;
#define HH(i, j) H[((size_t)(i)-1) * NRK + ((j)-1)]
  const int CS = NREST + 1, SN = CS + 1;
  int error = 0, ITER = 0, I = 0, nh = 0;
  double RESID = 0.0;
  orc_matresid_33(A, c, X, B, R);                                           /* :127 */
  double BNRM2 = dotn(NNDOF, B, B, c);
  if (BNRM2 == 0.0) { MAXIT = 0; RESID = 0.0; for (size_t i = 0; i < len; i++) X[i] = 0.0; }
  for (;;) { /* OUTER :158 */
    I = 0;
    double DNRM2 = dotn(NNDOF, R, R, c);                           /* :167 */
    if (DNRM2 == 0.0) break;
    double RNORM = sqrt(DNRM2), coef = 1.0 / RNORM;
    for (int32_t ik = 0; ik < NNDOF; ik++) V[1][ik] = R[ik] * coef;
    S[1] = RNORM;
    for (int k = 2; k <= NRK; k++) S[k] = 0.0;
    int converged = 0, failed = 0;
    for (I = 1; I <= NREST; I++) {                                          /* :187 */
      ITER = ITER + 1;
      orc_precond_apply(A, c, P, iterPREmax, V[I], ZQ, ZP);                 /* :195 */
      orc_matvec_33(A, c, ZQ, W);
      for (int K = 1; K <= I; K++) {                                        /* :207-214 modified Gram-Schmidt */
        double val = dotn(NNDOF, W, V[K], c);
        for (int32_t ik = 0; ik < NNDOF; ik++) W[ik] = W[ik] - val * V[K][ik];
        HH(K, I) = val;
      }
      double val = dotn(NNDOF, W, W, c);
      if (val == 0.0) break;                                                /* :217 */
      HH(I + 1, I) = sqrt(val);
      coef = 1.0 / HH(I + 1, I);
      for (int32_t ik = 0; ik < NNDOF; ik++) V[I + 1][ik] = W[ik] * coef;
      for (int k = 1; k <= I - 1; k++) {                                    /* :232-238 */
        double VCS = HH(k, CS), VSN = HH(k, SN);
        double DTEMP = VCS * HH(k, I) + VSN * HH(k + 1, I);
        HH(k + 1, I) = VCS * HH(k + 1, I) - VSN * HH(k, I);
        HH(k, I) = DTEMP;
      }
      double AA = HH(I, I), BB = HH(I + 1, I), R0 = BB, RR;                 /* :241-257 */
      if (fabs(AA) > fabs(BB)) R0 = AA;
      double scale = fabs(AA) + fabs(BB);
      if (scale != 0.0) {
        RR = scale * sqrt((AA / scale) * (AA / scale) + (BB / scale) * (BB / scale));
        RR = copysign(1.0, R0) * RR;
        HH(I, CS) = AA / RR;
        HH(I, SN) = BB / RR;
      } else {
        HH(I, CS) = 1.0; HH(I, SN) = 0.0; RR = 0.0;
      }
      double VCS = HH(I, CS), VSN = HH(I, SN);                              /* :260-268 */
      double DTEMP = VCS * HH(I, I) + VSN * HH(I + 1, I);
      HH(I + 1, I) = VCS * HH(I + 1, I) - VSN * HH(I, I);
      HH(I, I) = DTEMP;
      DTEMP = VCS * S[I] + VSN * S[I + 1];
      S[I + 1] = VCS * S[I + 1] - VSN * S[I];
      S[I] = DTEMP;
      RESID = fabs(S[I + 1]) / sqrt(BNRM2);
      if (hist) hist[nh] = RESID;
      nh++;
      if (RESID <= TOL) {                                                   /* :278-296 */
        gmres_update_x(A, c, P, iterPREmax, NNDOF, I, NRK, H, S, V, AV, ZQ, ZP, X);
        converged = 1;
        break;
      }
      if (ITER > MAXIT) { error = ERR_NOCONV_MAXIT; failed = 1; break; }    /* :298-301 */
    }
    if (converged || failed) break;
    /* restart :311-351 (after an early `exit` of the inner loop the reference lands here as well) */
    gmres_update_x(A, c, P, iterPREmax, NNDOF, NREST, NRK, H, S, V, AV, ZQ, ZP, X);
    orc_matresid_33(A, c, X, B, R);
    DNRM2 = dotn(NNDOF, R, R, c);
    if (I + 1 <= NRK) S[I + 1] = sqrt(DNRM2 / BNRM2);
    RESID = sqrt(DNRM2 / BNRM2);
    if (RESID <= TOL) break;
    if (ITER > MAXIT) { error = ERR_NOCONV_MAXIT; break; }
  }
  if (error == ERR_NOCONV_MAXIT)                                            /* :356-425 */
    gmres_update_x(A, c, P, iterPREmax, NNDOF, I, NRK, H, S, V, AV, ZQ, ZP, X);
  if (c && c->halo) c->halo(X, c->ctx);
#undef HH
  free(H); free(S); free(WW); free(V);
  *iter_out = ITER; *resid_out = RESID;
  if (nhist_out) *nhist_out = nh;
  return error;
}

int orc_solve_gpbicg(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax, const double *B,
                     double *X, int MAXIT, double TOL, int *iter_out, double *resid_out, double *hist) {
  const int N_ITER_RECOMPUTE_R = 20;
  int32_t N = A->N, NP = A->NP, NNDOF = ORC_ND(A) * N;
  size_t len = (size_t)ORC_ND(A) * NP;
  double *WW = (double *)calloc(14 * len, sizeof(double));
  /* R=1 RT=2 T=3 TT=4 T0=5 P=6 PT=7 U=8 W1=9 Y=10 Z=11 WK=12 W2=13 ZQ=14 (:56-69) */
  double *R = WW, *RT = WW + len, *T = WW + 2 * len, *TT = WW + 3 * len, *T0 = WW + 4 * len, *Pv = WW + 5 * len,
         *PT = WW + 6 * len, *U = WW + 7 * len, *W1 = WW + 8 * len, *Y = WW + 9 * len, *Z = WW + 10 * len,
         *WK = WW + 11 * len, *W2 = WW + 12 * len, *ZQ = WW + 13 * len;
  int error = 0, iter = 0;
  double RESID = 0.0, BETA = 0.0, ALPHA, QSI, ETA, RHO, RHO1, DNRM2, COEF1;
  orc_matresid_33(A, c, X, B, R);                                           /* :113 */
  for (int32_t i = 0; i < NNDOF; i++) RT[i] = R[i];
  double BNRM2 = dotn(NNDOF, B, B, c);
  if (BNRM2 == 0.0) { iter = 0; MAXIT = 0; RESID = 0.0; for (size_t i = 0; i < len; i++) X[i] = 0.0; }
  RHO = dotn(NNDOF, RT, R, c);                                     /* :127 */
  for (iter = 1; iter <= MAXIT; iter++) {
    for (int32_t j = 0; j < NNDOF; j++) WK[j] = R[j];                       /* :155-159 */
    orc_precond_apply(A, c, P, iterPREmax, WK, R, ZQ);
    if (iter > 1) for (int32_t j = 0; j < NNDOF; j++) Pv[j] = R[j] + BETA * (Pv[j] - U[j]); /* :166-174 */
    else for (int32_t j = 0; j < NNDOF; j++) Pv[j] = R[j];
    orc_matvec_33(A, c, Pv, PT);                                            /* :184 */
    RHO1 = dotn(NNDOF, RT, PT, c);
    ALPHA = RHO / RHO1;
    for (int32_t j = 0; j < NNDOF; j++) {                                   /* :197-200 */
      Y[j] = T[j] - WK[j] + ALPHA * (-W1[j] + PT[j]);
      T[j] = WK[j] - ALPHA * PT[j];
    }
    orc_precond_apply(A, c, P, iterPREmax, T, TT, ZQ);                      /* :211-216 */
    orc_precond_apply(A, c, P, iterPREmax, T0, W2, ZQ);
    for (int32_t i = 0; i < NNDOF; i++) T0[i] = W2[i];
    orc_precond_apply(A, c, P, iterPREmax, PT, W2, ZQ);
    orc_matvec_33(A, c, TT, WK);                                            /* :221-225 */
    for (int32_t i = 0; i < NNDOF; i++) TT[i] = WK[i];
    { /* pol_coef_vanilla2 :457-503 */
      const double OMEGA = 0.707106781;
      double CG[6] = {0, 0, 0, 0, 0, 0}, gamma1 = 0.0, gamma2 = 0.0;
      for (int32_t i = 0; i < NNDOF; i++) CG[0] += T[i] * T[i];
      for (int32_t i = 0; i < NNDOF; i++) CG[1] += TT[i] * TT[i];
      for (int32_t i = 0; i < NNDOF; i++) CG[2] += T[i] * TT[i];
      if (iter > 1) {
        for (int32_t i = 0; i < NNDOF; i++) CG[3] += Y[i] * Y[i];
        for (int32_t i = 0; i < NNDOF; i++) CG[4] += Y[i] * TT[i];
        for (int32_t i = 0; i < NNDOF; i++) CG[5] += Y[i] * T[i];
        if (c && c->allreduce) c->allreduce(CG, 6, c->ctx);
        gamma1 = CG[5] / CG[3];
        gamma2 = CG[4] / CG[3];
      } else if (c && c->allreduce) c->allreduce(CG, 3, c->ctx);
      double cc = CG[2] / sqrt(CG[0] * CG[1]);
      if (fabs(cc) > OMEGA) QSI = cc * sqrt(CG[0] / CG[1]);
      else if (cc >= 0.0) QSI = OMEGA * sqrt(CG[0] / CG[1]);
      else QSI = -OMEGA * sqrt(CG[0] / CG[1]);
      ETA = gamma1 - QSI * gamma2;
    }
    if (iter > 1) {                                                         /* :244-254 */
      for (int32_t j = 0; j < NNDOF; j++) {
        U[j] = QSI * W2[j] + ETA * (T0[j] - R[j] + BETA * U[j]);
        Z[j] = QSI * R[j] + ETA * Z[j] - ALPHA * U[j];
      }
    } else {
      for (int32_t j = 0; j < NNDOF; j++) {
        U[j] = QSI * W2[j] + ETA * (T0[j] - R[j]);
        Z[j] = QSI * R[j] + ETA * Z[j] - ALPHA * U[j];
      }
    }
    for (int32_t j = 0; j < NNDOF; j++) {                                   /* :262-266 */
      X[j] = X[j] + ALPHA * Pv[j] + Z[j];
      T0[j] = T[j];
    }
    if (iter % N_ITER_RECOMPUTE_R == 0) orc_matresid_33(A, c, X, B, R);     /* :268-274 */
    else for (int32_t j = 0; j < NNDOF; j++) R[j] = T[j] - ETA * Y[j] - QSI * TT[j];
    {
      double RR[2] = {0.0, 0.0};
      for (int32_t i = 0; i < NNDOF; i++) RR[0] += R[i] * R[i];
      for (int32_t i = 0; i < NNDOF; i++) RR[1] += R[i] * RT[i];
      if (c && c->allreduce) c->allreduce(RR, 2, c->ctx);
      DNRM2 = RR[0]; COEF1 = RR[1];
    }
    BETA = ALPHA * COEF1 / (QSI * RHO);
    for (int32_t j = 0; j < NNDOF; j++) W1[j] = TT[j] + BETA * PT[j];
    RESID = sqrt(DNRM2 / BNRM2);
    RHO = COEF1;
    if (hist) hist[iter - 1] = RESID;
    if (RESID <= TOL) {                                                     /* :300-307 */
      if (iter % N_ITER_RECOMPUTE_R == 0) break;
      orc_matresid_33(A, c, X, B, R);
      DNRM2 = dotn(NNDOF, R, R, c);
      RESID = sqrt(DNRM2 / BNRM2);
      if (RESID <= TOL) break;
    }
    if (iter == MAXIT) error = ERR_NOCONV_MAXIT;
  }
  if (c && c->halo) c->halo(X, c->ctx);
  free(WW);
  *iter_out = iter; *resid_out = RESID;
  return error;
}
