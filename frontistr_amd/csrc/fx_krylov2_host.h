// GMRES(m) and GPBiCG of the reference on the resident system:
//   hecmw_solve_GMRES   hecmw1/src/solver/iterative/hecmw_solver_GMRES.f90:17-458
//   hecmw_solve_GPBiCG  hecmw1/src/solver/iterative/hecmw_solver_GPBiCG.f90:17-505 (pol_coef_vanilla2 :457-503)
// Vectors, SpMV and the preconditioner sweeps are the same device kernels as for CG/BiCGSTAB; the small scalar
// recurrences (Hessenberg matrix, Givens rotations, the GPBiCG coefficients) run on the host from
// device-reduced dot products -- a host round trip per dot (I+1 per GMRES step, 8 per GPBiCG step), i.e.
// ~0.1-0.2 ms against 3-9 ms of sweeps per iteration at 10M DOF.  Results are deterministic (fixed-order
// reductions).  Included by fistr_hip.hip after the Krylov building blocks.
#pragma once

__global__ void k_scale_copy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] * a;
}
// GPBiCG vector updates, one kernel per reference loop
__global__ void k_gp_p(int64_t n, double beta, int first, const double *__restrict__ r, const double *__restrict__ u,
                       double *__restrict__ p) {  // :166-174
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = first ? r[i] : r[i] + beta * (p[i] - u[i]);
}
__global__ void k_gp_yt(int64_t n, double alpha, const double *__restrict__ wk, const double *__restrict__ w1,
                        const double *__restrict__ pt, double *__restrict__ y, double *__restrict__ t) {  // :197-200
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    y[i] = t[i] - wk[i] + alpha * (-w1[i] + pt[i]);
    t[i] = wk[i] - alpha * pt[i];
  }
}
__global__ void k_gp_uz(int64_t n, double qsi, double eta, double alpha, double beta, int first, const double *__restrict__ w2,
                        const double *__restrict__ t0, const double *__restrict__ r, double *__restrict__ u,
                        double *__restrict__ z) {  // :244-254
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double un = first ? qsi * w2[i] + eta * (t0[i] - r[i]) : qsi * w2[i] + eta * (t0[i] - r[i] + beta * u[i]);
    u[i] = un;
    z[i] = qsi * r[i] + eta * z[i] - alpha * un;
  }
}
__global__ void k_gp_x(int64_t n, double alpha, const double *__restrict__ p, const double *__restrict__ z,
                       const double *__restrict__ t, double *__restrict__ x, double *__restrict__ t0) {  // :262-266
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] = x[i] + alpha * p[i] + z[i];
    t0[i] = t[i];
  }
}
__global__ void k_gp_r(int64_t n, double eta, double qsi, const double *__restrict__ t, const double *__restrict__ y,
                       const double *__restrict__ tt, double *__restrict__ r) {  // :272
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    r[i] = t[i] - eta * y[i] - qsi * tt[i];
}
__global__ void k_gp_w1(int64_t n, double beta, const double *__restrict__ tt, const double *__restrict__ pt,
                        double *__restrict__ w1) {  // :288-290
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    w1[i] = tt[i] + beta * pt[i];
}

struct HostKrylov {
  int iter = 0, status = 1, error = 0;
  double resid = 0.0;
  std::vector<double> hist;
  // scalars of the last iteration in fx_debug_state's order (rho rho1 beta c1 alpha omega c2 cg0 cg1 dnrm2 bnrm2 ...):
  // GPBiCG: rho = r~.r, c2 = r~.A p (ALPHA's denominator), omega = QSI, cg0 = COEF1, cg1 = ETA
  double dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

static int hk_dot(fx_context *c, const double *x, const double *y, double *out) {
  int np;
  if (dot_into_partials(c, x, y, nullptr, 0, &np)) return FX_ERROR_RUNTIME;
  double tmp;
  return host_sum(c, np, 0, out, &tmp);
}

static int ensure_extra(fx_context *c, int count) {
  if (c->extra_n >= count && c->extra_len == c->wlen) return 0;
  dev_free(c->extra);
  if (dev_alloc(&c->extra, (size_t)count * c->wlen)) return FX_ERROR_RUNTIME;
  c->extra_n = count;
  c->extra_len = c->wlen;
  return 0;
}

static int hk_prepare(fx_context *c, int maxit, double tol, int extra) {
  if (ensure_solver(c)) return FX_ERROR_RUNTIME;
  if (to_slots(c, c->A.B, c->Bs) || to_slots(c, c->A.X, c->Xs)) return FX_ERROR_RUNTIME;
  if (krylov_init_state(c, std::max(maxit, 1), tol)) return FX_ERROR_RUNTIME;  // status = RUNNING: the kernels' gates stay open
  for (int k = 0; k < 8; k++) HIP_TRY(hipMemsetAsync(c->W[k], 0, (size_t)c->wlen * 8, c->stream));
  if (ensure_extra(c, extra)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemsetAsync(c->extra, 0, (size_t)extra * c->wlen * 8, c->stream));
  if (c->precond_kind == 1 || level_sched(c))
    HIP_TRY(hipMemsetAsync(c->ssor.zs, 0, (size_t)3 * c->ssor.nslots * 8, c->stream));
  return 0;
}

// The two solvers are written against a small set of operations so that the 3x3 path (Ops33: slot-ordered vectors, the
// device-resident preconditioners) and the generic-block path (OpsNN in fx_nn_host.h) share them.
struct Ops33 {
  fx_context *c;
  double *X() const { return c->Xs; }
  double *B() const { return c->Bs; }
  double *W(int k) const { return c->W[k]; }
  double *extra(int k) const { return c->extra + (size_t)k * c->wlen; }
  int64_t veclen() const { return (int64_t)3 * c->ord.nslots; }
  size_t wbytes() const { return (size_t)c->wlen * 8; }
  int64_t ndof_np() const { return (int64_t)3 * c->A.NP; }
  int spmv(int mode, double *x, const double *b, double *y) const { return ::spmv(c, mode, 0, x, b, y, nullptr, 0); }
  int precond(const double *r, double *z) const { int np; return precond_apply(c, r, z, false, &np); }
  int dot(const double *x, const double *y, double *out) const { return hk_dot(c, x, y, out); }
  int prepare(int maxit, double tol, int extra) const { return hk_prepare(c, maxit, tol, extra); }
};

#define VLAUNCH(kern, ...) hipLaunchKernelGGL(kern, dim3(vgrid), dim3(256), 0, c->stream, n3, __VA_ARGS__)

template <class Ops>
static int gmres_solve_t(const Ops &o, int MAXIT, double TOL, int NREST, HostKrylov *out) {
  fx_context *c = o.c;
  if (NREST >= o.ndof_np() - 1) NREST = (int)(o.ndof_np() - 2);  // :88
  if (NREST < 1) NREST = 1;
  if (o.prepare(MAXIT, TOL, NREST + 1)) return FX_ERROR_RUNTIME;
  const int64_t n3 = o.veclen();
  const int vgrid = grid_for(n3, 256, 2048);
  double *X = o.X(), *B = o.B(), *R = o.W(0), *ZQ = o.W(1), *W = o.W(2), *AV = o.W(3);
  auto V = [&](int k) { return o.extra(k - 1); };  // V(1..NREST+1)
  const int NRK = NREST + 7, CS = NREST + 1, SN = CS + 1;
  std::vector<double> Hm((size_t)NRK * NRK, 0.0), S((size_t)NRK + 2, 0.0), SS((size_t)NRK + 2), Y((size_t)NRK + 2);
  auto H = [&](int i, int j) -> double & { return Hm[(size_t)(i - 1) * NRK + (j - 1)]; };
  int ITER = 0, I = 0, error = 0;
  double RESID = 0.0, BNRM2, DNRM2, val;
  auto update_x = [&](int IROW) -> int {  // [H]{y} = {s}; {x} += Minv (V y)  (:264-292)
    for (int ik = 1; ik <= IROW; ik++) SS[ik] = S[ik];
    Y[IROW] = SS[IROW] / H(IROW, IROW);
    for (int kk = IROW - 1; kk >= 1; kk--) {
      for (int jj = IROW; jj >= kk + 1; jj--) SS[kk] = SS[kk] - H(kk, jj) * Y[jj];
      Y[kk] = SS[kk] / H(kk, kk);
    }
    HIP_TRY(hipMemsetAsync(AV, 0, o.wbytes(), c->stream));
    for (int jj = 1; jj <= IROW; jj++) VLAUNCH(k_axpy_plain, Y[jj], V(jj), AV);
    if (o.precond(AV, ZQ)) return FX_ERROR_RUNTIME;
    VLAUNCH(k_axpy_plain, 1.0, ZQ, X);
    HIP_TRY(hipGetLastError());
    return 0;
  };
  if (o.spmv(1, X, B, R)) return FX_ERROR_RUNTIME;  // :127
  if (o.dot(B, B, &BNRM2)) return FX_ERROR_RUNTIME;
  if (BNRM2 == 0.0) { MAXIT = 0; RESID = 0.0; HIP_TRY(hipMemsetAsync(X, 0, o.wbytes(), c->stream)); }
  for (;;) {  // OUTER
    I = 0;
    if (o.dot(R, R, &DNRM2)) return FX_ERROR_RUNTIME;
    if (DNRM2 == 0.0) break;
    const double RNORM = sqrt(DNRM2);
    VLAUNCH(k_scale_copy, 1.0 / RNORM, R, V(1));
    S[1] = RNORM;
    for (int k = 2; k <= NRK; k++) S[k] = 0.0;
    bool converged = false, failed = false;
    for (I = 1; I <= NREST; I++) {
      ITER++;
      if (o.precond(V(I), ZQ)) return FX_ERROR_RUNTIME;  // :195
      if (o.spmv(0, ZQ, nullptr, W)) return FX_ERROR_RUNTIME;
      for (int K = 1; K <= I; K++) {  // modified Gram-Schmidt :207-214
        if (o.dot(W, V(K), &val)) return FX_ERROR_RUNTIME;
        VLAUNCH(k_axpy_plain, -val, V(K), W);
        H(K, I) = val;
      }
      if (o.dot(W, W, &val)) return FX_ERROR_RUNTIME;
      if (val == 0.0) break;
      H(I + 1, I) = sqrt(val);
      VLAUNCH(k_scale_copy, 1.0 / H(I + 1, I), W, V(I + 1));
      for (int k = 1; k <= I - 1; k++) {  // :232-238
        const double VCS = H(k, CS), VSN = H(k, SN);
        const double DTEMP = VCS * H(k, I) + VSN * H(k + 1, I);
        H(k + 1, I) = VCS * H(k + 1, I) - VSN * H(k, I);
        H(k, I) = DTEMP;
      }
      const double AA = H(I, I), BB = H(I + 1, I);  // :241-257
      double R0 = BB, RR;
      if (fabs(AA) > fabs(BB)) R0 = AA;
      const double scale = fabs(AA) + fabs(BB);
      if (scale != 0.0) {
        RR = scale * sqrt((AA / scale) * (AA / scale) + (BB / scale) * (BB / scale));
        RR = copysign(1.0, R0) * RR;
        H(I, CS) = AA / RR;
        H(I, SN) = BB / RR;
      } else {
        H(I, CS) = 1.0; H(I, SN) = 0.0;
      }
      const double VCS = H(I, CS), VSN = H(I, SN);
      double DTEMP = VCS * H(I, I) + VSN * H(I + 1, I);
      H(I + 1, I) = VCS * H(I + 1, I) - VSN * H(I, I);
      H(I, I) = DTEMP;
      DTEMP = VCS * S[I] + VSN * S[I + 1];
      S[I + 1] = VCS * S[I + 1] - VSN * S[I];
      S[I] = DTEMP;
      RESID = fabs(S[I + 1]) / sqrt(BNRM2);
      out->hist.push_back(RESID);
      if (RESID <= TOL) {
        if (update_x(I)) return FX_ERROR_RUNTIME;
        converged = true;
        break;
      }
      if (ITER > MAXIT) { error = FX_ERROR_NOCONV_MAXIT; failed = true; break; }
    }
    if (converged || failed) break;
    if (update_x(NREST)) return FX_ERROR_RUNTIME;  // restart :311-351
    if (o.spmv(1, X, B, R)) return FX_ERROR_RUNTIME;
    if (o.dot(R, R, &DNRM2)) return FX_ERROR_RUNTIME;
    if (I + 1 <= NRK) S[I + 1] = sqrt(DNRM2 / BNRM2);
    RESID = sqrt(DNRM2 / BNRM2);
    if (RESID <= TOL) break;
    if (ITER > MAXIT) { error = FX_ERROR_NOCONV_MAXIT; break; }
  }
  if (error == FX_ERROR_NOCONV_MAXIT && update_x(I)) return FX_ERROR_RUNTIME;  // :356-425
  out->iter = ITER; out->resid = RESID; out->error = error;
  out->status = error ? error : 1;
  return 0;
}

template <class Ops>
static int gpbicg_solve_t(const Ops &o, int MAXIT, double TOL, HostKrylov *out) {
  fx_context *c = o.c;
  const int RECOMPUTE = 20;
  if (o.prepare(MAXIT, TOL, 5)) return FX_ERROR_RUNTIME;
  const int64_t n3 = o.veclen();
  const int vgrid = grid_for(n3, 256, 2048);
  double *X = o.X(), *B = o.B();
  double *R = o.W(0), *RT = o.W(1), *T = o.W(2), *TT = o.W(3), *T0 = o.W(4), *P = o.W(5), *PT = o.W(6), *U = o.W(7);
  double *W1 = o.extra(0), *Y = o.extra(1), *Z = o.extra(2), *WK = o.extra(3), *W2 = o.extra(4);
  int iter = 0, error = 0;
  double RESID = 0.0, BETA = 0.0, ALPHA, QSI, ETA, RHO, RHO1, BNRM2, DNRM2, COEF1;
  if (o.spmv(1, X, B, R)) return FX_ERROR_RUNTIME;  // :113
  VLAUNCH(k_copy, R, RT);
  if (o.dot(B, B, &BNRM2)) return FX_ERROR_RUNTIME;
  if (BNRM2 == 0.0) { MAXIT = 0; RESID = 0.0; HIP_TRY(hipMemsetAsync(X, 0, o.wbytes(), c->stream)); }
  if (o.dot(RT, R, &RHO)) return FX_ERROR_RUNTIME;
  for (iter = 1; iter <= MAXIT; iter++) {
    VLAUNCH(k_copy, R, WK);  // :155-159
    if (o.precond(WK, R)) return FX_ERROR_RUNTIME;
    VLAUNCH(k_gp_p, BETA, (int)(iter == 1), R, U, P);
    if (o.spmv(0, P, nullptr, PT)) return FX_ERROR_RUNTIME;  // :184
    if (o.dot(RT, PT, &RHO1)) return FX_ERROR_RUNTIME;
    ALPHA = RHO / RHO1;
    VLAUNCH(k_gp_yt, ALPHA, WK, W1, PT, Y, T);
    if (o.precond(T, TT)) return FX_ERROR_RUNTIME;  // :211-216
    if (o.precond(T0, W2)) return FX_ERROR_RUNTIME;
    VLAUNCH(k_copy, W2, T0);
    if (o.precond(PT, W2)) return FX_ERROR_RUNTIME;
    if (o.spmv(0, TT, nullptr, WK)) return FX_ERROR_RUNTIME;  // :221-225
    VLAUNCH(k_copy, WK, TT);
    {  // pol_coef_vanilla2 :457-503
      const double OMEGA = 0.707106781;
      double CG[6] = {0, 0, 0, 0, 0, 0}, gamma1 = 0.0, gamma2 = 0.0;
      if (o.dot(T, T, &CG[0]) || o.dot(TT, TT, &CG[1]) || o.dot(T, TT, &CG[2])) return FX_ERROR_RUNTIME;
      if (iter > 1) {
        if (o.dot(Y, Y, &CG[3]) || o.dot(Y, TT, &CG[4]) || o.dot(Y, T, &CG[5])) return FX_ERROR_RUNTIME;
        gamma1 = CG[5] / CG[3];
        gamma2 = CG[4] / CG[3];
      }
      const double cc = CG[2] / sqrt(CG[0] * CG[1]);
      if (fabs(cc) > OMEGA) QSI = cc * sqrt(CG[0] / CG[1]);
      else if (cc >= 0.0) QSI = OMEGA * sqrt(CG[0] / CG[1]);
      else QSI = -OMEGA * sqrt(CG[0] / CG[1]);
      ETA = gamma1 - QSI * gamma2;
    }
    VLAUNCH(k_gp_uz, QSI, ETA, ALPHA, BETA, (int)(iter == 1), W2, T0, R, U, Z);
    VLAUNCH(k_gp_x, ALPHA, P, Z, T, X, T0);
    if (iter % RECOMPUTE == 0) {  // :268-274
      if (o.spmv(1, X, B, R)) return FX_ERROR_RUNTIME;
    } else {
      VLAUNCH(k_gp_r, ETA, QSI, T, Y, TT, R);
    }
    if (o.dot(R, R, &DNRM2) || o.dot(R, RT, &COEF1)) return FX_ERROR_RUNTIME;
    BETA = ALPHA * COEF1 / (QSI * RHO);
    VLAUNCH(k_gp_w1, BETA, TT, PT, W1);
    RESID = sqrt(DNRM2 / BNRM2);
    {
      const double d[16] = {RHO, RHO1, BETA, 0.0, ALPHA, QSI, RHO1, COEF1, ETA, DNRM2, BNRM2, RESID, TOL, (double)iter, 0.0, 0.0};
      memcpy(out->dbg, d, sizeof d);
    }
    RHO = COEF1;
    out->hist.push_back(RESID);
    if (!std::isfinite(RESID)) { error = FX_ERROR_NOCONV_MAXIT; break; }  // breakdown: same guard as the BiCGSTAB path
    if (RESID <= TOL) {  // :300-307
      if (iter % RECOMPUTE == 0) break;
      if (o.spmv(1, X, B, R)) return FX_ERROR_RUNTIME;
      if (o.dot(R, R, &DNRM2)) return FX_ERROR_RUNTIME;
      RESID = sqrt(DNRM2 / BNRM2);
      if (RESID <= TOL) break;
    }
    if (iter == MAXIT) error = FX_ERROR_NOCONV_MAXIT;
  }
  HIP_TRY(hipGetLastError());
  out->iter = iter; out->resid = RESID; out->error = error;
  out->status = error ? error : 1;
  return 0;
}
#undef VLAUNCH

static int gmres_solve(fx_context *c, int MAXIT, double TOL, int NREST, HostKrylov *out) {
  return gmres_solve_t(Ops33{c}, MAXIT, TOL, NREST, out);
}
static int gpbicg_solve(fx_context *c, int MAXIT, double TOL, HostKrylov *out) { return gpbicg_solve_t(Ops33{c}, MAXIT, TOL, out); }
