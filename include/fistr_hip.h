/* fistr_hip.h -- C ABI of the MI355X-native HEC-MW linear-solve hot path.
 *
 * Drop-in boundary.  The reference exposes Fortran module
 * procedures only; the entry points below are what a thin Fortran shim
 * (frontistr_amd/shim/hecmw_solver_hip.f90) binds with
 * bind(C) after taking c_loc() of the members of hecmwST_matrix /
 * hecmwST_local_mesh.  Plain pointers and sizes only.
 *
 * Conventions are the reference's (hecmw1/src/common/hecmw_util_f.F90:15-16,
 * :433-468): int32 indices, 1-based `item` arrays, `index` arrays with NP+1
 * entries starting at 0, fp64 values, 3x3 blocks row-major (A(9j-8..9j) = row 1,
 * row 2, row 3 of block j).
 *
 * All functions return 0 on success, a HEC-MW solver code (fx_status) for the
 * conditions the reference reports through hecmw_solve_error
 * (hecmw1/src/solver/init/hecmw_solve_error.f90:9-15), or a negative value for
 * a HIP/RCCL runtime failure (fx_last_error() then holds the message).
 */
#ifndef FISTR_HIP_H
#define FISTR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* hecmw_solve_error.f90:9-15 */
enum fx_status {
  FX_OK = 0,
  FX_ERROR_INCONS_PC = 1001,    /* E: inconsistent solver/preconditioner        */
  FX_ERROR_ZERO_DIAG = 2001,    /* E: zero component in diagonal block          */
  FX_ERROR_ZERO_RHS = 2002,     /* W: zero RHS norm (X = 0 is returned)         */
  FX_ERROR_NOCONV_MAXIT = 3001, /* W: not converged within ceratin iterations   */
  FX_ERROR_DIVERGE_MAT = 3002,  /* W: diverged due to indefinite/neg-def matrix */
  FX_ERROR_DIVERGE_PC = 3003,   /* W: diverged due to indefinite preconditioner */
  FX_NEWTON_MAXRES = 4002,      /* fx_newton_substep: NR residual above step_ctrl%maxres (knstDRESN = 2): cut back */
  FX_ERROR_RUNTIME = -1,        /* HIP / RCCL failure, see fx_last_error()      */
  FX_ERROR_UNSUPPORTED = -2     /* NDOF outside 1..6, or an option outside the hot path */
};

/* Borrowed view of hecmwST_matrix (hecmw_util_f.F90:433-468).  The caller owns
 * every array (m_fstr.f90:807-857 allocates them once); the library writes X
 * (all 3*NP entries, halo included: hecmw_solver_CG.f90:280) and never frees or
 * reallocates anything. */
typedef struct fx_matrix_view {
  int32_t N, NP, NPL, NPU, NDOF;
  const int32_t *indexL, *itemL; /* (0:NP), (NPL) */
  const int32_t *indexU, *itemU; /* (0:NP), (NPU) */
  const double *D, *AL, *AU;     /* NDOF^2 * NP, NPL, NPU (row-major NDOF x NDOF blocks) */
  const double *B;               /* NDOF*NP */
  double *X;                     /* NDOF*NP, in: initial guess, out: solution */
} fx_matrix_view;

/* Borrowed view of the communication part of hecmwST_local_mesh
 * (hecmw_util_f.F90:298-310).  Serial runs pass n_neighbor_pe = 0 and PETOT = 1
 * (or a NULL pointer). */
typedef struct fx_comm_view {
  int32_t my_rank, PETOT, nn_internal, n_node;
  int32_t n_neighbor_pe;
  const int32_t *neighbor_pe;  /* ranks, 0-based as in HEC-MW */
  const int32_t *import_index; /* (0:n_neighbor_pe) */
  const int32_t *import_item;  /* 1-based local node ids (> nn_internal) */
  const int32_t *export_index; /* (0:n_neighbor_pe) */
  const int32_t *export_item;  /* 1-based local node ids (<= nn_internal) */
} fx_comm_view;

/* What the reference prints / keeps per solve (hecmw_solver_Iterative.f90:192-208). */
typedef struct fx_solve_info {
  int32_t iterations;     /* ITER as in the '### summary of linear solver' line        */
  int32_t method, precond;
  int32_t ncolor;         /* SSOR: number of colours actually used                     */
  int32_t n_hist;         /* residual-history entries written                          */
  double resid;           /* RESID of the last iteration (sqrt(DNRM2/BNRM2))           */
  double rel_resid;       /* final ||b-Ax||/||b|| (hecmw_rel_resid_L2)                 */
  double time_setup, time_sol, time_comm, time_matvec, time_precond; /* seconds       */
} fx_solve_info;

typedef struct fx_context fx_context; /* device state the reference keeps in module `save` data */

/* ---- life cycle ------------------------------------------------------- */
/* device < 0: use LOCAL_RANK (or 0).  One context per MPI rank / subdomain. */
int fx_create(int device, fx_context **out);
void fx_destroy(fx_context *ctx);
const char *fx_last_error(void);
const char *fx_version(void);
int fx_device_synchronize(fx_context *ctx);

/* ---- the drop-in pair --------------------------------------------------- */
/* hecmw_solve(hecMESH, hecMAT), hecmw1/src/solver/hecmw_solver.f90:9 ->
 * hecmw_solve_iterative, iterative/hecmw_solver_Iterative.f90:13.
 * Iarray/Rarray are hecMAT%Iarray/Rarray (slot map hecmw_matrix_misc.f90:89-121);
 * on return Iarray(81) converged, (82) diverged, (96) nrecycle, (97),(98) are
 * updated exactly as the reference does.  hist (may be NULL) receives RESID per
 * iteration = the ITERLOG channel (hecmw_solver_CG.f90:245), at most hist_len.
 * METHOD Iarray(2): 1 CG (hecmw_solver_CG.f90:19), 2 BiCGSTAB (hecmw_solver_BiCGSTAB.f90:16),
 * 3 GMRES(NREST=Iarray(6)) (hecmw_solver_GMRES.f90:17), 4 GPBiCG (hecmw_solver_GPBiCG.f90:17);
 * PRECOND Iarray(3): 1,2 SSOR, 3 DIAG (block Jacobi), 10 ILU(0); anything else E-1001.
 * SCALING Iarray(7) /= 0: symmetric diagonal scaling around every attempt (las/hecmw_solver_scaling_33.f90);
 * SIGMA_DIAG Rarray(2) < 0: the reference's automatic retry for the ILU family; METHOD2 Iarray(8) take-over.
 * NDOF = 3 is the tuned path.  NDOF = 1, 2, 4, 5, 6 (hecmw_matvec_nn las_nn.f90:135, precond/nn + 11/22/44/66) take the
 * generic-block path: METHOD 1-4; PRECOND 1, 2, 3; SCALING; anything else E-1001 / FX_ERROR_UNSUPPORTED. */
int fx_solve(fx_context *ctx, const fx_matrix_view *mat, const fx_comm_view *comm, int32_t *Iarray,
             double *Rarray, fx_solve_info *info, double *hist, int32_t hist_len);

/* hecmw_matvec(hecMESH, hecMAT, X, Y, COMMtime), las/hecmw_solver_las.f90:57:
 * halo update of X (X's halo part is written, as hecmw_update_3_R does), then
 * Y(1:NDOF*N) = (D + AL + AU) X.  x, y are host arrays of NDOF*NP doubles.  The reference's hecmw_matvec has no
 * "matrix changed" flag and its external callers modify hecMAT between calls, so the values in `mat` are uploaded on every
 * call; mat->D == NULL means "the resident values" (repeated products with one matrix: no PCIe traffic but x and y). */
int fx_matvec(fx_context *ctx, const fx_matrix_view *mat, const fx_comm_view *comm, double *x,
              double *y, double *commtime);

/* ---- staged, device-resident form (what fx_solve is made of) ------------ */
/* Upload / refresh the system.  what: bit 0 profile (Iarray(98) semantics),
 * bit 1 values (Iarray(97)), bit 2 B, bit 3 X. */
enum { FX_UP_PROFILE = 1, FX_UP_VALUES = 2, FX_UP_RHS = 4, FX_UP_X = 8, FX_UP_ALL = 15 };
int fx_upload(fx_context *ctx, const fx_matrix_view *mat, const fx_comm_view *comm, int what);
/* hecmw_precond_setup, precond/hecmw_precond.f90:28-50 (DIAG_33 / SSOR_33 / BILU_33). */
int fx_precond_setup(fx_context *ctx, const int32_t *Iarray, const double *Rarray);
/* hecmw_solve_CG / hecmw_solve_BiCGSTAB on the resident system; no host<->device
 * traffic except the status word.  This is the region bench.py times. */
int fx_solve_resident(fx_context *ctx, int32_t *Iarray, double *Rarray, fx_solve_info *info,
                      double *hist, int32_t hist_len);
/* The same loop in pieces, for callers that time an exact number of iterations:
 * begin = r0 = b - A x0 and ||b|| (hecmw_solver_CG.f90:120-129); steps = enqueue up to
 * nsteps iterations (never past Iarray(1)) and return the device state after them
 * (iter = Fortran ITER, status 0 = still running, 1 = converged, else an fx_status). */
int fx_krylov_begin(fx_context *ctx, const int32_t *Iarray, const double *Rarray);
int fx_krylov_steps(fx_context *ctx, int32_t nsteps, int32_t *iter, int32_t *status, double *resid);
/* The ITERLOG lines of the staged loop (hecmw_solver_CG.f90:245: RESID per iteration, line i = iteration i) executed since
 * fx_krylov_begin: *n_lines = lines written so far, at most cap of them copied to hist. */
int fx_krylov_history(fx_context *ctx, double *hist, int32_t cap, int32_t *n_lines);
int fx_download_x(fx_context *ctx, double *X, int32_t n);          /* 3*NP doubles */
int fx_download_matrix(fx_context *ctx, double *D, double *AL, double *AU, double *B);
/* Resident single operations (tests, roofline timing). */
int fx_matvec_resident(fx_context *ctx, int nrepeat, float *ms_per_call); /* y = A x on work vectors */
/* The same with the launch variant chosen: 0 plain (hecmw_solver_BiCGSTAB.f90:184, :210), 1 fused partial of x.y (the
 * product of every CG iteration, hecmw_solver_CG.f90:204-211), 2 r = b - A x with the partial of r.r (hecmw_matresid,
 * las/hecmw_solver_las.f90:105-122).  HIP events on the solver stream; one untimed call first.  The product reads and
 * writes the work vectors the CG loop multiplies (p and q) and overwrites them: not to be called while a Krylov loop is
 * between fx_krylov_begin and its last step. */
int fx_spmv_resident(fx_context *ctx, int variant, int nrepeat, float *ms_per_call);
/* The same for the resident NDOF != 3 system of the last fx_solve / fx_matvec (hecmw_matvec_nn, las_nn.f90:135-310).
 * stats: NDOF, N, padded blocks of the layout, blocks of the matrix (N + NPL + NPU). */
int fx_nn_matvec_resident(fx_context *ctx, int nrepeat, float *ms_per_call, int64_t stats[4]);
int fx_precond_apply_resident(fx_context *ctx, int nrepeat, float *ms_per_call); /* z = M^-1 b, timed */
int fx_precond_apply_host(fx_context *ctx, const double *r, double *z);   /* z = M^-1 r, 3*NP doubles */
/* out[0..14]: N NP NPL NPU | M pairs, blocks, slices | ncolor | L pairs, blocks | U pairs, blocks | slices |
 * SpMV workgroups overlapped with the halo exchange (interior), ordered after it (boundary) | [15] bit 0: the last Krylov
 * loop ran in Eisenstat's form (the default for CG + multicolour SSOR, FX_EISENSTAT=0 opts out); bit 1: the ILU(0) sweeps are chain sweeps (FX_DATAFLOW=3) */
int fx_get_stats(fx_context *ctx, int64_t out[16]);
/* Where the value arrays of the sliced layouts live: out[0] bytes of the context's value arena (one large allocation taken when the
 * size of a large system first becomes known, before anything else of it: 0 = none), [1] bytes in use, [2] arrays placed in it,
 * [3..5] 1 if the value array of the SpMV layout / the lower / the upper sweep layout lies in it, [6] bytes of the SpMV layout's
 * value array, [7] arenas the one-off verification timed (the loop's SpMV on the arena, 5 ms; a slow one is replaced by another arena,
 * at most FX_ARENA_TRIES = 4, the fastest kept), [8] the SpMV's ms on the arena kept.  FX_ARENA_GB (0 = off, default 32 = at least 32 GiB). */
int fx_placement_report(fx_context *ctx, double out[9]);
/* The plane march of the level-scheduled sweeps (ILU(0) hecmw_precond_BILU_33.f90:90-157, natural-order SSOR
 * hecmw_precond_SSOR_33.f90:300-410; csrc/fx_march.h): out[0] 1 if this context's preconditioner has the march programs, [1] rows per
 * chunk, [2] chunks, [3] pair waves per workgroup, [4] / [5] rounds of the forward / backward program, [6] blocks gathered from the LDS
 * ring, [7] from memory, [8] of those in the row's own chunk, [9] / [10] the cost model's microseconds per half sweep as a march / as
 * dependency levels, [11] seconds the build took, [12] applies that took the march, [13] workgroups of the last launch, [14] rows of the
 * largest round, [15] dependency levels.  FX_MARCH = 0 off, 1 (default) when the cost model prefers it, 2 whenever the structure admits it. */
int fx_march_report(fx_context *ctx, double out[16]);
/* Host-only (no device): plan the two march programs for a block profile (indexL / itemL / indexU / itemU as hecMAT's, 1-based items, N
 * internal rows) with `chunk` rows per chunk and `waves` pair waves (1, 2, 3), replay them on the host and report: out[0] 1 = both programs
 * pass their replay (every in-chunk dependency found in its ring slot, every other one produced by an earlier chunk or round), 0 = the
 * structure is not admitted (a row with more than 14 lower or upper blocks), [1] chunks, [2] / [3] rounds forward / backward, [4] / [5]
 * blocks gathered from the ring / from memory (forward), [6] rows of the largest round, [7] dependency levels of the matrix. */
int fx_march_plan(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU, const int32_t *itemU, int32_t chunk,
                  int32_t waves, double out[8]);
/* The passes of the auto-SIGMA_DIAG / METHOD2 loop of the last solve on this context (hecmw_solver_Iterative.f90:117-157: banner
 * :125 before every pass, 'Increasing SIGMA_DIAG to' :149 before a retry): METHOD, the SIGMA_DIAG in effect and the number of
 * residual-history lines of every pass, and the lines themselves.  fx_solve's own `hist` holds the LAST pass. */
int fx_solve_attempts(fx_context *ctx, int32_t cap, int32_t *n_attempts, int32_t *method, double *sigma_diag, int32_t *n_hist);
int fx_solve_attempt_history(fx_context *ctx, int32_t attempt, double *hist, int32_t cap);
/* Tuning knob of a live context (kernel variants, workgroup sizes, sweep modes: the FX_* names documented in
 * frontistr_amd/csrc/fx_internal.h; the same names are read from the environment at fx_create).  No counterpart in the
 * reference (its equivalents are build-time choices such as the OpenMP thread count).  Unknown name: FX_ERROR_UNSUPPORTED. */
int fx_set_option(fx_context *ctx, const char *name, double value);
/* measured read-streaming rate (GB/s) of this device over the resident matrix values: the on-box
 * ceiling reported beside the 8 TB/s vendor peak */
int fx_stream_ceiling(fx_context *ctx, int nrepeat, double *gbs);
/* diagnostics: rho rho1 beta c1 alpha omega c2 cg0 cg1 dnrm2 bnrm2 resid tol iter status need_verify */
int fx_debug_state(fx_context *ctx, double out[16]);
int fx_dot_host(fx_context *ctx, const double *x, const double *y, double *result);

/* ---- assembly side ------------------------------------------------------ */
/* hecmw_mat_con, matrix/hecmw_mat_con.f90:23: CRS block profile from element
 * connectivity (conn 1-based, nn nodes per element).  Two-call protocol: with
 * itemL == NULL only indexL/indexU (NP+1 each) are filled. */
int fx_mat_con(int32_t NP, int32_t n_elem, int32_t nn, const int32_t *conn, int32_t *indexL,
               int32_t *indexU, int32_t *itemL, int32_t *itemU);

/* Host only: the ordering of the multicolour SSOR as the library computes it (reference: hecmw_precond_SSOR_33.f90:102-111 ->
 * hecmw_matrix_ordering_CM.f90:16-178 "RCM" + hecmw_matrix_ordering_MC.f90:15-72 greedy capped multicolouring; the path the
 * reference takes with more than one OpenMP thread).  perm: N entries, new -> old, 1-based; colorindex: COLORindex(0:ncolor). */
int fx_ssor_ordering(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU, const int32_t *itemU,
                     int32_t ncolor_in, int32_t *perm, int32_t *colorindex, int32_t colorindex_cap, int32_t *ncolor);

/* The same two arrays as the resident preconditioner was built with (large systems run the level ordering on the device). */
int fx_get_ssor_ordering(fx_context *ctx, int32_t *perm, int32_t *colorindex, int32_t colorindex_cap, int32_t *ncolor);

/* Host only: the element colouring behind the atomic-free stiffness scatter (no two elements of a colour share a node;
 * the reference serialises the same conflicts with `!$omp atomic`, hecmw_mat_ass.f90:72-134).  order: n_elem element ids
 * (0-based) grouped by colour; offsets: 65 entries, offsets[k]..offsets[k+1] = colour k; *ncolor = 0 when a node belongs
 * to more than 64 elements (the device then falls back to atomics). */
int fx_color_elements(int32_t NP, int32_t n_elem, int32_t nn, const int32_t *conn, int32_t *order, int32_t *offsets,
                      int32_t *ncolor);

/* fstr_StiffMatrix (fistr1/src/analysis/static/fstr_StiffMatrix.f90:18-212) for one
 * TYPE=361 element group with one isotropic linear-elastic material, then
 * hecmw_mat_ass_bc (matrix/hecmw_mat_ass.f90:292) for the listed dofs:
 * element stiffness (elemopt 1 = C3D8 incompatible modes STF_C3D8IC, 2 = B-bar
 * STF_C3D8Bbar, 3 = full integration STF_C3) and scatter-add run on the device
 * into the resident D/AL/AU; `load` (3*NP, may be NULL) becomes B.
 * Requires a profile uploaded with fx_upload(..., FX_UP_PROFILE). */
typedef struct fx_mesh_view {
  int32_t n_node, n_elem;
  const double *coord;  /* 3*n_node, hecMESH%node */
  const int32_t *conn;  /* 8*n_elem, 1-based, hecMESH%elem_node_item */
} fx_mesh_view;
int fx_assemble_c3d8(fx_context *ctx, const fx_mesh_view *mesh, double E, double nu, int elemopt,
                     const double *load, int32_t n_bc, const int32_t *bc_node,
                     const int32_t *bc_dof, const double *bc_val, float *ms_assemble);
/* The same for a group whose elements belong to several sections / materials (hecMESH%section_ID ->
 * fstrSOLID%materials, fstr_setup.f90:325-400): elem_mat[e] in 1..n_mat selects (E[m-1], nu[m-1]). */
int fx_assemble_c3d8_sections(fx_context *ctx, const fx_mesh_view *mesh, int32_t n_mat, const double *E,
                              const double *nu, const int32_t *elem_mat, int elemopt, const double *load,
                              int32_t n_bc, const int32_t *bc_node, const int32_t *bc_dof,
                              const double *bc_val, float *ms_assemble);
/* fstr_UpdateNewton of a LINEAR static analysis (`!SOLUTION, TYPE=STATIC`; fistr1/src/analysis/static/fstr_Update.f90:25-293) for
 * one TYPE=361 group of isotropic ELASTIC materials: UpdateST_C3D8IC (static_LIB_3dIC.f90:220-455, elemopt 1), Update_C3D8Bbar
 * (static_LIB_C3D8.f90:203-547, elemopt 2) or UPDATE_C3 (static_LIB_3d.f90:516-837, elemopt 3) per element, small strain.
 * disp = total displacement unode + dunode (3*n_node).  *strain, *stress: pinned host arrays OWNED BY THE LIBRARY, valid until the
 * next call, [n_elem][8][6] = gausses(1:8)%strain(1:6) / %stress(1:6) of every element; qforce (3*n_node, caller's, may be NULL) =
 * fstrSOLID%QFORCE before its halo update.  n_mat materials, elem_mat 1-based per element (NULL with one material). */
int fx_update_c3d8_linear(fx_context *ctx, const fx_mesh_view *mesh, int32_t n_mat, const double *E, const double *nu,
                          const int32_t *elem_mat, int elemopt, const double *disp, const double **strain,
                          const double **stress, double *qforce, float *ms_kernel);
/* Optional: start pinning the host staging of fx_update_c3d8_linear for n_elem elements on a helper thread and return at once. */
int fx_update_c3d8_linear_prepare(fx_context *ctx, int32_t n_elem);
/* One element stiffness through the device kernel (tests): ecoord 8x3, stiff 24x24 row-major. */
int fx_element_stiffness_c3d8(fx_context *ctx, int elemopt, const double *ecoord, double E, double nu,
                              double *stiff);

/* ---- nonlinear static loop: the steps of fstr_Newton either side of the solve -------------
 * (fistr1/src/analysis/static/fstr_solve_NonLinear.f90:29-167).  One TYPE=361 group with the
 * B-bar formulation (the reference's default for NLSTATIC, fstr_setup.f90:366-368) and one
 * isotropic material: ELASTIC or Mises elastoplastic (!PLASTIC, YIELD=MISES) with BILINEAR /
 * MULTILINEAR / SWIFT / RAMBERG-OSGOOD isotropic hardening, INFINITE / TOTALLAG (KIRCHHOFF) /
 * UPDATELAG kinematics.  Everything stays resident: gauss-point history, unode/dunode, QFORCE. */
typedef struct fx_material_view { /* tMaterial after fstr_ctrl_get_ELASTICITY/_PLASTICITY (fstr_ctrl_material.f90:60-106, :341-480) */
  double E, nu;         /* M_YOUNGS, M_POISSON */
  int32_t plastic;      /* 0: mtype ELASTIC; 1: elastoplastic, Mises */
  int32_t harden;       /* fifth digit of mtype: 0 BILINEAR 1 MULTILINEAR 2 SWIFT 3 RAMBERG-OSGOOD */
  int32_t nlgeom;       /* nlgeom_flag: 0 INFINITE 1 TOTALLAG 2 UPDATELAG */
  int32_t ntab;         /* MULTILINEAR: rows of the MC_YIELD table */
  double plconst[3];    /* M_PLCONST1..3 */
  const double *tab;    /* ntab rows (yield stress, plastic strain) */
} fx_material_view;
typedef struct fx_nl_state_view { /* host arrays, any may be NULL (skipped).  tGaussStatus members (mechgauss.f90:13-22) */
  double *stress, *strain, *stress_bak, *strain_bak; /* n_elem*8*6 */
  double *plstrain, *fstat;                           /* n_elem*8: plstrain, fstatus(1) */
  int32_t *istat;                                     /* n_elem*8: istatus(1) */
  double *unode, *dunode, *qforce;                    /* 3*NP: fstrSOLID%unode, %dunode, %QFORCE */
  int32_t latch; /* MatlMatrix's saved flag (calMatMatrix.f90:39); set_state: <0 leaves it */
} fx_nl_state_view;
/* fstr_solid / gauss-point set-up (zero state).  Needs a profile (fx_upload FX_UP_PROFILE). */
int fx_nl_init(fx_context *ctx, const fx_mesh_view *mesh, const fx_material_view *mat);
/* The same for a group whose elements belong to several sections / materials: elem_mat[e] in 1..n_mat selects mats[m-1]
 * (hecMESH%section_ID -> fstrSOLID%materials, fstr_setup.f90:325-400; every tGaussStatus%pMaterial points at its section's
 * tMaterial).  The materials may carry different NLGEOM flags (an ELASTIC TOTALLAG part next to a PLASTIC UPDATELAG part). */
int fx_nl_init_sections(fx_context *ctx, const fx_mesh_view *mesh, int32_t n_mat, const fx_material_view *mats,
                        const int32_t *elem_mat);
/* fstr_Newton :63-68 + fstr_ass_load: dunode = 0, GL (3*NP, may be NULL), B = GL - QFORCE. */
int fx_nl_begin_substep(fx_context *ctx, const double *GL);
/* fstr_StiffMatrix (fstr_StiffMatrix.f90:18-212) + fstr_AddBC (fstr_AddBC.f90:17-190) with the
 * given increments of the prescribed dofs; result in the resident D/AL/AU/B. */
int fx_nl_stiffness(fx_context *ctx, int32_t n_bc, const int32_t *bc_node, const int32_t *bc_dof,
                    const double *bc_val, float *ms_assemble);
/* dunode += X, fstr_UpdateNewton (fstr_Update.f90:25-293), fstr_Update_NDForce
 * (fstr_Residual.f90:23-71); out = {|B|^2,|X|^2,|QFORCE|^2,|dunode|^2} (may be NULL). */
int fx_nl_update(fx_context *ctx, double out[4], float *ms_update);
/* hecmw_solve (hecmw_solver.f90:9) for a matrix that was assembled ON the device: B and X of `mat` go up, the prescribed dofs
 * fstr_AddBC passed to hecmw_mat_ass_bc (hecmw_mat_ass.f90:292-429) are applied to the resident matrix and B, the resident system
 * is solved as fx_solve does, X comes back.  mat->D / AL / AU are ignored. */
int fx_solve_device_matrix(fx_context *ctx, const fx_matrix_view *mat, const fx_comm_view *comm, int32_t n_bc, const int32_t *bc_node,
                           const int32_t *bc_dof, const double *bc_val, int32_t *Iarray, double *Rarray, fx_solve_info *info,
                           double *hist, int32_t hist_len);
/* The two element loops for a caller that keeps fstr_Newton's own loop on the host (the Fortran binding of INTEGRATION.md section 5).
 * fx_nl_stiffness_at: fstr_StiffMatrix (fstr_StiffMatrix.f90:18-212) for the host's unode / dunode (3*NP each, may be NULL = keep the
 *   resident ones), tangent into the resident D / AL / AU, no boundary conditions (fstr_AddBC follows on the host side).
 * fx_nl_update_at: fstr_UpdateNewton (fstr_Update.f90:25-293) for the host's dunode; QFORCE (3*NP) is written back.
 * fx_mat_ass_bc: hecmw_mat_ass_bc (hecmw_mat_ass.f90:292-429) for a list of prescribed dofs on the resident matrix and B. */
int fx_nl_stiffness_at(fx_context *ctx, const double *unode, const double *dunode, float *ms_assemble);
int fx_nl_update_at(fx_context *ctx, const double *dunode, double *qforce, float *ms_update);
int fx_mat_ass_bc(fx_context *ctx, int32_t n_bc, const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val);
/* unode += dunode, fstr_UpdateState (fstr_Update.f90:296-345). */
int fx_nl_commit(fx_context *ctx);
/* fstr_cutback_save (load = 0) / fstr_cutback_load (load = 1) for the device-resident quadrature-point history
 * (fistr1/src/analysis/static/fstr_Cutback.f90:108-198: fstr_copy_gauss of every point): an automatic incrementation rolls the
 * state back to the last converged sub-step when Newton fails (fstr_solve_NLGEOM.f90:156-197).  unode / QFORCE stay with the host. */
int fx_nl_snapshot(fx_context *ctx, int load);
int fx_nl_get_state(fx_context *ctx, fx_nl_state_view *s);
int fx_nl_set_state(fx_context *ctx, const fx_nl_state_view *s);
/* element-level outputs of the two kernels, no scatter (tests): ke n_elem*24*24, qf n_elem*24 */
int fx_nl_element_tangents(fx_context *ctx, double *ke);
int fx_nl_element_update(fx_context *ctx, double *qf);
/* One substep of fstr_Newton around fx_solve_resident.  log: 7 doubles per Newton iteration
 * (iter, solver iterations, solver code, |B|, |X|, |QFORCE|, |dunode|).  Returns 0 (converged),
 * FX_ERROR_NOCONV_MAXIT (max_iter ran out; committed only if commit_unconverged), FX_NEWTON_MAXRES (residual above maxres,
 * fx_nl_set_step_control; never committed) or an error. */
int fx_newton_substep(fx_context *ctx, double factor0, double factor1, int32_t n_bc,
                      const int32_t *bc_node, const int32_t *bc_dof, const double *bc_val,
                      const double *cload, int32_t max_iter, double converg, int32_t *Iarray,
                      double *Rarray, double *log, int32_t *n_iter, int commit_unconverged);

/* step_ctrl(cstep)%maxres (fistr1/src/lib/m_step.f90:31, default 1.d+10): fx_newton_substep returns FX_NEWTON_MAXRES as soon as
 * |B|/|QFORCE| exceeds it (fstr_solve_NonLinear.f90:140-152, nothing committed); is_linear = fstr_Newton's isLinear
 * (.not. fstrPR%nlgeom, :50-51, :107): one pass without a convergence test. */
int fx_nl_set_step_control(fx_context *ctx, double maxres, int is_linear);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) -------------------- */
/* 128-byte ncclUniqueId made by rank 0 and broadcast by the host side
 * (torch.distributed / MPI); then every rank calls fx_comm_init. */
int fx_comm_unique_id(unsigned char id[128]);
int fx_comm_init(fx_context *ctx, const unsigned char id[128], int rank, int nranks);
/* Alternative transport (tests on one GPU, MPI builds of fistr1 without RCCL): the library
 * stages through pinned host buffers and calls back.  halo: send holds 3*n_export doubles in
 * export_item order, recv must be filled with 3*n_import doubles in import_item order
 * (hecmw_solve_send_recv_33, hecmw_solver_SR_33.F90:42-124).  allreduce: in-place SUM over
 * ranks of n doubles (hecmw_allreduce_R, hecmw_comm_f.F90:346-379). */
/* Ranks the transport itself reports (ncclCommCount of the RCCL communicator, or the count given with the host
 * callbacks; 1 without either) and the device the communicator / context is bound to. */
int fx_comm_size(fx_context *ctx, int32_t *nranks, int32_t *device);
/* What this rank asked of the transport since it was set up, for a cross-rank consistency check (bench.py --gpus N compares the
 * ledgers of all ranks over its control plane; hecmw_solver_SR_33.F90:42-124 and hecmw_comm_f.F90:346-379 are the calls counted):
 * out[0] operations, [1] hash of their sequence (kind and size of every all-reduce, position of every halo exchange: equal on all
 * ranks), [2] all-reduces, [3] their bytes, [4] halo exchanges, [5] neighbours, [6] 1 = the halo exchange has a communicator of its
 * own, then 5 per neighbour: rank, messages sent, bytes sent, messages received, bytes received.  *n_out = entries available. */
int fx_comm_ledger(fx_context *ctx, int64_t *out, int32_t cap, int32_t *n_out);
typedef void (*fx_halo_fn)(const double *send, double *recv, void *user);
typedef void (*fx_allreduce_fn)(double *v, int n, void *user);
int fx_comm_set_host_callbacks(fx_context *ctx, int rank, int nranks, fx_halo_fn halo, fx_allreduce_fn allreduce,
                               void *user);

#ifdef __cplusplus
}
#endif
#endif /* FISTR_HIP_H */
