/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the HEC-MW hot path.
 *
 * Plain-C restatement of the reference algorithms the HIP library replaces.
 * Every function cites the reference file:line it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (frontistr_amd/csrc) never links or calls it.
 *
 * Pinning: checked against the real reference compiled from /root/reference
 * (oracle/_ref, built by oracle/build_ref.py) and against the committed golden
 * vectors those binaries produced (tests/golden/).
 *
 * Conventions are the reference's: int32 indices, 1-based `item` arrays,
 * index arrays dimensioned (0:NP), 3x3 blocks row-major (A(9j-8..9j)).
 */
#ifndef HECMW_ORACLE_H
#define HECMW_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t N, NP;
  const int32_t *indexL, *itemL, *indexU, *itemU; /* index: NP+1 entries; item: 1-based */
  const double *D, *AL, *AU;
  int32_t ndof; /* hecMAT%NDOF; 0 = 3 */
} orc_matrix;

/* Optional communication hooks (multi-subdomain runs driven from Python over
 * gloo).  NULL => serial no-ops, as the HECMW_SERIAL build. */
typedef void (*orc_halo_fn)(double *x, void *ctx);           /* hecmw_update_3_R */
typedef void (*orc_allreduce_fn)(double *v, int n, void *ctx); /* hecmw_allreduce_R (SUM) */
typedef struct {
  orc_halo_fn halo;
  orc_allreduce_fn allreduce;
  void *ctx;
} orc_comm;

/* las: hecmw_solver_las_33.f90:135-351, 358-380 */
void orc_matvec_33(const orc_matrix *A, const orc_comm *c, double *X, double *Y);
void orc_matresid_33(const orc_matrix *A, const orc_comm *c, double *X, const double *B, double *R);
/* hecmw_solver_misc.f90:46-70 */
double orc_inner_product(int32_t nn_internal, const double *X, const double *Y, const orc_comm *c);

/* Preconditioner state (module-level `save` data of the reference). */
typedef struct orc_precond orc_precond;
/* precond: 1,2 SSOR | 3 DIAG | 10 ILU(0).  nthreads==1 -> natural order SSOR,
 * >=2 -> RCM + multicolour (hecmw_precond_SSOR_33.f90:93-114). */
orc_precond *orc_precond_setup(const orc_matrix *A, int precond, double sigma_diag, int ncolor_in,
                               int nthreads);
void orc_precond_free(orc_precond *P);
/* hecmw_precond.f90:75-123 + 33/hecmw_precond_33.f90:74-115 */
void orc_precond_apply(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax,
                       double *R, double *Z, double *ZP);
/* Introspection for tests: SSOR ordering results. */
int orc_precond_ncolor(const orc_precond *P);
const int32_t *orc_precond_perm(const orc_precond *P);       /* new -> old, 1-based, N entries */
const int32_t *orc_precond_colorindex(const orc_precond *P); /* 0..ncolor */

/* hecmw_matrix_ordering_CM.f90:57-66, hecmw_matrix_ordering_MC.f90:15-72 */
void orc_ordering_rcm(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU,
                      const int32_t *itemU, int32_t *perm, int32_t *iperm);
void orc_ordering_mc(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU,
                     const int32_t *itemU, const int32_t *perm_cur, int ncolor_in, int32_t *ncolor_out,
                     int32_t *COLORindex, int32_t *perm, int32_t *iperm);

/* Solvers.  hist (may be NULL) receives RESID per iteration (max `maxit`).
 * Returns the reference's error code (0, 3001 MAXIT, 3002 DIVERGE_MAT, 3003 DIVERGE_PC). */
int orc_solve_cg(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax,
                 const double *B, double *X, int maxit, double tol, int *iter_out, double *resid_out,
                 double *hist);
int orc_solve_bicgstab(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax,
                       const double *B, double *X, int maxit, double tol, int *iter_out,
                       double *resid_out, double *hist);
/* hecmw_solver_GMRES.f90:17-458 (NREST = Iarray(6)); hist gets one RESID per inner iteration (nhist_out) */
int orc_solve_gmres(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax, const double *B,
                    double *X, int maxit, double tol, int nrest, int *iter_out, double *resid_out, double *hist,
                    int *nhist_out);
/* hecmw_solver_GPBiCG.f90:17-505 */
int orc_solve_gpbicg(const orc_matrix *A, const orc_comm *c, orc_precond *P, int iterPREmax, const double *B,
                     double *X, int maxit, double tol, int *iter_out, double *resid_out, double *hist);
/* hecmw_solver_Iterative.f90:13-210: Iarray/Rarray protocol, zero-RHS / zero-diag checks,
 * final ||b-Ax||/||b|| -> Iarray(81).  nthreads selects the SSOR ordering path. */
/* keep the preconditioner across orc_solve_iterative calls and rebuild it only when Iarray(97)/(98) ask (recycle policy) */
void orc_persist_precond(int on);
int orc_solve_iterative(const orc_matrix *A, const orc_comm *c, const double *B, double *X,
                        int32_t *Iarray, double *Rarray, int nthreads, int *iter_out,
                        double *resid_out, double *hist);

/* Assembly side. */
/* hecmw_mat_con.f90:23-268.  conn: 1-based, nn nodes per element.  Two-call
 * protocol: first with itemL==NULL to get NPL/NPU (indexL/indexU filled), then
 * with item arrays allocated. */
void orc_mat_con(int32_t NP, int32_t n_elem, int nn, const int32_t *conn, int32_t *indexL,
                 int32_t *indexU, int32_t *itemL, int32_t *itemU);
/* Element stiffness of a linear-elastic C3D8 (INFINITE flag), stiff row-major 24x24.
 * elemopt 1: STF_C3D8IC static_LIB_3dIC.f90:21-215; 2: STF_C3D8Bbar
 * static_LIB_C3D8.f90:23-200; 3: STF_C3 static_LIB_3d.f90:47-205. */
void orc_stf_c3d8(int elemopt, const double *ecoord /*8x3 node-major*/, double E, double nu,
                  double *stiff);
/* hecmw_mat_ass.f90:31-134 */
void orc_mat_ass_elem(int32_t NP, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU,
                      const int32_t *itemU, double *D, double *AL, double *AU, int nn,
                      const int32_t *nodLOCAL, const double *stiff);
/* hecmw_mat_ass.f90:292-429 */
void orc_mat_ass_bc(int32_t NP, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU,
                    const int32_t *itemU, double *D, double *AL, double *AU, double *B, int32_t inode,
                    int32_t idof, double RHS);
/* fstr_StiffMatrix.f90:18-212 element loop for one TYPE=361 mesh, one material. */
void orc_assemble_c3d8(int elemopt, int32_t NP, int32_t n_elem, const double *coord,
                       const int32_t *conn, double E, double nu, const int32_t *indexL,
                       const int32_t *itemL, const int32_t *indexU, const int32_t *itemU, double *D,
                       double *AL, double *AU);

void orc_assemble_c3d8_sections(int elemopt, int32_t NP, int32_t n_elem, const double *coord, const int32_t *conn,
                                const double *E, const double *nu, const int32_t *elem_mat, const int32_t *indexL,
                                const int32_t *itemL, const int32_t *indexU, const int32_t *itemU, double *D,
                                double *AL, double *AU);

/* ---- stress update of a linear static analysis (fstr_UpdateNewton with UpdateST_C3D8IC / Update_C3D8Bbar / UPDATE_C3, ELASTIC,
 *      INFINITE): restated in fstr_update_linear_oracle.c ---- */
void orc_update_c3d8_linear(int elemopt, const double *ecoord, const double *edisp, double E, double nu, double *strain,
                            double *stress, double *qf);
void orc_update_linear(int elemopt, int32_t n_node, int32_t n_elem, const double *coord, const int32_t *conn, const double *E,
                       const double *nu, const int32_t *elem_mat, const double *disp, double *strain, double *stress,
                       double *qforce);

/* ---- Nonlinear (elastoplastic) C3D8 B-bar path: restated in fstr_nl_oracle.c ---- */
typedef struct {
  double E, nu;
  int32_t plastic; /* 0: ELASTIC, 1: Mises elastoplastic */
  int32_t harden;  /* 0 BILINEAR 1 MULTILINEAR 2 SWIFT 3 RAMBERG-OSGOOD (fifth digit of mtype) */
  int32_t nlgeom;  /* 0 INFINITE 1 TOTALLAG 2 UPDATELAG */
  int32_t ntab;
  double plconst[3]; /* M_PLCONST1..3 */
  const double *tab; /* ntab rows (yield stress, plastic strain) */
} orc_material;
typedef struct { /* tGaussStatus members, flat over (element, quadrature point) */
  double *stress, *strain, *stress_bak, *strain_bak; /* [n_elem*8*6] */
  double *plstrain, *fstat;                           /* [n_elem*8]  fstat = fstatus(1) */
  int32_t *istat;                                     /* [n_elem*8]  istatus(1) */
} orc_gauss_state;
void orc_nl_reset_latch(void);
int orc_nl_latch(void);
double orc_curr_yield(const orc_material *m, double pstrain);
double orc_harden_coeff(const orc_material *m, double pstrain);
void orc_elastoplastic_matrix(const orc_material *m, const double *stress, int istat, double extval1, double *D);
void orc_backward_euler(const orc_material *m, double *stress, double plstrain, int32_t *istat, double *fstat1);
void orc_stf_c3d8bbar_nl(const orc_material *m, const double *ecoord, const double *u, const double *stress,
                         const int32_t *istat, const double *fstat, double *stiff);
void orc_update_c3d8bbar(const orc_material *m, const double *ecoord, const double *u, const double *du,
                         double *stress, double *strain, const double *stress_bak, const double *strain_bak,
                         const double *plstrain, int32_t *istat, double *fstat, double *qf);
void orc_nl_stiffness(const orc_material *m, int32_t NP, int32_t n_elem, const double *coord, const int32_t *conn,
                      const double *unode, const double *dunode, const orc_gauss_state *st,
                      const int32_t *indexL, const int32_t *itemL, const int32_t *indexU, const int32_t *itemU,
                      double *D, double *AL, double *AU);
void orc_nl_update(const orc_material *m, int32_t n_node, int32_t n_elem, const double *coord, const int32_t *conn,
                   const double *unode, const double *dunode, orc_gauss_state *st, double *qforce);
void orc_nl_commit(const orc_material *m, int32_t n_elem, orc_gauss_state *st);
/* several sections: element e uses mats[elem_mat[e] - 1] in the three loops above (NULL: the single material argument) */
void orc_nl_set_sections(const orc_material *mats, const int32_t *elem_mat);

#ifdef __cplusplus
}
#endif
#endif
