// Internal declarations of libfistr_hip (gfx950 only).  See include/fistr_hip.h for the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/fistr_hip.h"

extern thread_local std::string g_fx_error;
int fx_fail(const char *what, const char *file, int line);

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      g_fx_error = std::string(#expr) + ": " + hipGetErrorString(_e);        \
      return fx_fail(g_fx_error.c_str(), __FILE__, __LINE__);                \
    }                                                                        \
  } while (0)

// ---------------------------------------------------------------------------
// Device copy of the reference layout (D / AL / AU with 1-based items), kept so
// that assembly and BC elimination work on the arrays fistr1 owns and so the
// matrix can be handed back bit-for-bit.
// ---------------------------------------------------------------------------
struct DevCSR {
  int32_t N = 0, NP = 0, NPL = 0, NPU = 0;
  int32_t *indexL = nullptr, *itemL = nullptr, *indexU = nullptr, *itemU = nullptr;
  double *D = nullptr, *AL = nullptr, *AU = nullptr, *B = nullptr, *X = nullptr;
};

// ---------------------------------------------------------------------------
// BELL-64: sliced block-ELL, the layout every sweep kernel streams.
//   slot  = 64*slice + lane; one thread owns one block row.
//   slice s holds block positions [pair_ptr[s], pair_ptr[s+1]) ("halves": one 3x3 block per lane
//   each), consumed two at a time: for the pair starting at position h and block element e
//   (row-major, 0..8) the two values of lane l sit in one 16-byte word
//       ((double2*)(val + h*576))[e*64 + l] = (block h, block h+1)[e]
//   and the two column ids (0-based vector slots) in one 8-byte word ((int2*)(col + h*64))[l].
//   An odd last position of a slice is stored alone (val[h*576 + e*64 + l], col[h*64 + l]):
//   hex-mesh rows have 27 blocks, pair padding would waste 1/27 of the stream.
//   => every wave-level load is 1 KiB (values) / 512 B (columns), fully coalesced,
//   and a lane never needs a cross-lane reduction.
//   Padding blocks carry value 0 and the row's own column (always a valid address).
// ---------------------------------------------------------------------------
struct Bell {
  int32_t nslots = 0, nslices = 0;
  int64_t npairs = 0;           // total block positions (halves) over all slices
  int32_t *pair_ptr = nullptr;  // nslices+1, in halves
  double *val2 = nullptr;       // npairs*9*64 doubles
  void *val2_base = nullptr;    // allocation holding val2 (placement experiments: FX_VAL2_PAD)
  int *col2 = nullptr;          // npairs*64
  int *src2 = nullptr;          // npairs*64: source block codes 3*idx+{0 D,1 AL,2 AU}, -1 padding (kept for numeric refresh)
  int32_t *slot_row = nullptr;  // nslots: 0-based node id of the slot, -1 = padding slot (may be null = identity)
  int32_t *slice_order = nullptr;  // nslices: order in which the SpMV walks the slices (null = ascending)
  // Domain-decomposed systems: the SpMV's virtual workgroups (4 consecutive entries of the slice walk each) split into those
  // none of whose rows has a halo column (interior: run while the halo exchange is in flight) and the rest (boundary).
  int32_t *wg_interior = nullptr, *wg_boundary = nullptr;
  int32_t n_wg_interior = 0, n_wg_boundary = 0;
  int64_t nblocks = 0;          // real (non padding) blocks
  size_t val2_bytes = 0;        // bytes val2_base was allocated with (its share of the arena, the power-of-two request, or the exact size)
  struct fx_context *arena_owner = nullptr;  // val2_base lies in this context's value arena (DevArena): released there, not with hipFree
  size_t bytes() const { return (size_t)npairs * 64 * (9 * 8 + 4) + (size_t)(nslices + 1) * 4; }
};

// Solver numbering.  All Krylov vectors and the BELL structures live in "slot" space:
//   kind 0 (natural):      slot = node, padded to a multiple of 64;
//   kind 1 (colour-major): the multicolour SSOR ordering of the reference (same colours), each
//                          colour starting a new 64-slot slice, natural node order inside a
//                          colour.  Neighbouring lanes then gather neighbouring vector entries
//                          both in the colour sweeps and in the SpMV (full cache lines), which
//                          the reference's old-numbering ZP indexing does not give a GPU.
// Halo nodes N+h keep slot nslots+h.  Padding slots carry no blocks, an identity diagonal
// factor and zeros in every vector, so kernels treat every slot as a row.
struct Ordering {
  int kind = -1;
  int32_t nslots = 0, nhalo = 0;
  std::vector<int32_t> slot_node;  // nslots+nhalo: 0-based node of a slot, -1 = padding
  std::vector<int32_t> slot_of;    // NP: node -> slot
  int32_t *d_slot_node = nullptr, *d_slot_of = nullptr;
  int32_t vslots() const { return nslots + nhalo; }
};

// Multicolour SSOR state (hecmw_precond_SSOR_33.f90 module variables).
// Plane march of the level-scheduled sweeps (fx_march.h): one direction's program.  The rows of a chunk (a contiguous range of the
// natural numbering, one workgroup) are executed in ROUNDS -- the chunk's own dependency levels, split to at most R rows -- and the
// matrix is stored in exactly that order: 8 lanes per row, lanes 0..6 one block pair each, lane 7 the LU of the diagonal block.
struct MarchProg {
  int64_t nrounds = 0;
  int32_t *round_ptr = nullptr;  // device [nchunks + 1]: rounds of chunk c = [round_ptr[c], round_ptr[c + 1])
  int32_t *rstart = nullptr;     // device [nrounds + 1]: march position of each round's first row
  double *val = nullptr;         // device: per round [9][8 n] double2 (n = rows of the round)
  int32_t *col = nullptr;        // device: int2 per lane (>= 0 far: row whose entry is gathered from memory; -1 padding;
                                 //   <= -2 near: -(ring slot) - 2; lane 7: (row, 0))
  int32_t *src = nullptr;        // device: int2 per lane, source codes of the two blocks (k_bell_fill's; lane 7: 3 * row)
  std::vector<int32_t> h_round_ptr, h_rstart;
};
struct MarchDev {
  bool ok = false;
  int32_t S = 0, NW = 0, nchunks = 0;  // rows per chunk, pair waves per workgroup (8 rows each), chunks
  MarchProg F, B;                      // forward (L) and backward (U) programs
  double *zf = nullptr;                // forward sweep's vector, 3 N, natural numbering
  double est_us = 0.0, est_level_us = 0.0;  // cost model: one half sweep as a march / as dependency levels
  double build_s = 0.0;
  int32_t max_round_rows = 0, far_same_chunk = 0;
  int64_t near_blocks = 0, far_blocks = 0;
};

struct SsorDev {
  int32_t ncolor = 0;
  std::vector<int32_t> color_slice;  // slice range per colour: [color_slice[c], color_slice[c+1])
  Bell L, U;                         // strictly-lower / strictly-upper parts in colour-slot order
  Bell H;                            // Eisenstat form on a subdomain: the halo-column blocks (what the localized L / U drop)
  double *alu = nullptr;             // LU of the diagonal blocks, [slice][e][lane] layout
  double sigma_diag = 1.0;           // SIGMA_DIAG the factors in alu were built with
  int64_t values_epoch = -1;         // fx_context::values_epoch the sweep layouts were filled at
  int32_t nslots = 0;                // colour-major slots (each colour padded to a 64 multiple)
  int32_t *slot_node = nullptr;      // device: slot -> 0-based node, -1 = padding
  double *zs = nullptr;              // private sweep vector, 3*nslots, colour-major
  double *zb = nullptr;              // dataflow sweeps: the backward sweep's vector (zs holds the forward one), 3*nslots
  double *lu_D = nullptr, *lu_AL = nullptr, *lu_AU = nullptr;  // ILU(0): factor values in the reference CSR layout
  std::vector<int32_t> slot_start;   // ILU(0): first slot of each level (ncolor+1)
  int32_t max_row_blocks = 0;        // ILU(0): largest number of off-diagonal blocks in a row (<= 32: lane-per-block factorisation)
  std::vector<int32_t> perm;         // new -> old (1-based), as the reference's perm(:)
  std::vector<int32_t> colorindex;   // COLORindex(0:ncolor)
  MarchDev march;                    // level-scheduled sweeps as a plane march (fx_march.h)
  int32_t *slice_level = nullptr;    // device: dependency level of every slice of the level-major layouts (ILU(0), natural SSOR)
};

struct DiagDev {
  double *alu = nullptr;  // [slice][e][lane]
  int32_t nslices = 0;
};

// Scalars of the Krylov loops, resident on the device; the host only polls `status`.
#define FX_DF_RETRY (-77)  // internal: a dataflow sweep timed out, the context switched to df_mode 0, redo the work
#define FX_ST_PAUSED 2  // KrylovState::status: waiting for the host to enqueue the true-residual check (hecmw_solver_CG.f90:259-266)
struct KrylovState {
  double rho, rho1, beta, c1, alpha, omega, c2, cg0, cg1, dnrm2, bnrm2, resid, tol;
  int32_t iter;         // Fortran ITER of the iteration being executed
  int32_t status;       // 0 running, 1 converged/exit, else fx_status code
  int32_t need_verify;  // RESID<=TOL seen, true residual check pending
  int32_t n_indef;
  int32_t maxit;
  int32_t error;        // 3001 set while running (reference sets error and falls out of the DO)
  int32_t n_hist;       // residual-history lines written (the ITERLOG lines the reference would have printed)
  int32_t pause_verify; // 1: RESID <= TOL by the recurrence parks the loop (status = FX_ST_PAUSED) and the HOST enqueues the true-residual
                        // check at its next poll -- the rare path costs no launches per iteration; 0: the check rides in every iteration, gated
  int32_t t_current;    // Eisenstat form: 1 = t = (D~+L)^-1 r is current; 0 = r was replaced by the true residual, refresh t (set by OP_VERIFY, cleared by OP_CG_RHO)
};

struct HaloDev {
  int32_t n_neighbor = 0;
  std::vector<int32_t> neighbor, import_index, export_index;
  std::vector<int32_t> h_export, h_import;                 // host, 0-based node ids
  int32_t *export_item = nullptr, *import_item = nullptr;  // device, slot ids
  double *sendbuf = nullptr, *recvbuf = nullptr;
  int32_t n_export = 0, n_import = 0;
};

// Material of the nonlinear path as the kernels take it (one isotropic Mises material per context).
struct NlMat {
  double E, nu, pl[3];
  int32_t plastic, harden, nlgeom, ntab;
  const double *tab;  // device, ntab rows (yield stress, plastic strain)
};

// fstr_solid members of the nonlinear static loop (m_fstr.f90:560-700) for one TYPE=361 B-bar group, resident.
// Elements grouped by colour (fxo::color_elements): the atomic-free scatter of the stiffness kernels.
struct ElemColors {
  int32_t n_elem = 0;
  uint64_t key = 0;                 // checksum of the connectivity the colouring belongs to
  int32_t *order = nullptr;         // device: element ids, colour by colour
  std::vector<int32_t> offsets;     // host: first position of each colour (+ end); empty = not coloured (atomics)
  int32_t *pos = nullptr;           // device: 64 per element, position of block (a, b) in AL / AU (k_scatter_map), or null
  bool dup_nodes = false;           // an element names a node twice: two of its blocks coincide, no first-write flags
  bool first_write = false;         // pos carries the first-write flags (k_scatter_first_flag) and every block of the profile is covered:
                                    // the coloured scatter stores the first contribution to a block and the matrix is not cleared first
};

struct NlDev {
  bool ready = false;
  ElemColors colors;                  // order: elements grouped by NLGEOM flag, then by colour; pos: scatter position map
  std::vector<int32_t> grp_off[3];    // per NLGEOM flag (0 INFINITE, 1 TOTALLAG, 2 UPDATELAG): positions of its colours in order (+ end)
  bool scatter_atomic = false;        // colouring failed (a node in more than 64 elements): one range per group, fp64 atomics
  int32_t n_mat = 1;
  NlMat *mats = nullptr;              // device, n_mat entries (several sections); null with one material
  int32_t *emat = nullptr;            // device, 1-based material id per element; null with one material
  std::vector<NlMat> h_mats;
  std::vector<double *> tabs;         // device tables of the materials
  int32_t n_elem = 0, n_bc = 0;
  NlMat mat = {};
  double *tab = nullptr;
  double *coord = nullptr;
  int32_t *conn = nullptr;
  // tGaussStatus (mechgauss.f90:13-22), flat over (element, quadrature point)
  double *stress = nullptr, *strain = nullptr, *stress_bak = nullptr, *strain_bak = nullptr;  // 6 per point
  double *plstrain = nullptr, *fstat = nullptr;                                               // 1 per point
  int32_t *istat = nullptr;
  double *unode = nullptr, *dunode = nullptr, *qforce = nullptr, *GL = nullptr;  // 3*NP
  uint8_t *bc_flag = nullptr;   // prescribed dofs of the current step (3*NP)
  double *bc_val = nullptr;     // their values (3*NP)
  int32_t *bc_node = nullptr, *bc_dof = nullptr;
  double *bc_v = nullptr;
  int32_t bc_cap = 0;
  int32_t *err = nullptr;
  double maxres = 1.0e10;  // step_ctrl%maxres (m_step.f90:78): Newton gives up when rres exceeds it
  bool is_linear = false;   // fstr_Newton's isLinear: one pass, no convergence test
  int latch = 0;  // MatlMatrix's saved `flag` (calMatMatrix.f90:39): set by the first elastoplastic stress update
  // fstr_cutback_save / _load (fstr_Cutback.f90:108-198): a device-side copy of the quadrature-point history to roll back to when a
  // sub-step of an automatic incrementation does not converge (fx_nl_snapshot); allocated at the first save
  double *bk_stress = nullptr, *bk_strain = nullptr, *bk_stress_bak = nullptr, *bk_strain_bak = nullptr, *bk_plstrain = nullptr, *bk_fstat = nullptr;
  int32_t *bk_istat = nullptr;
  bool bk_valid = false;
};

// TIMELOG (hecmw_solver_Iterative.f90:192-208): device time of the three phases the reference reports -- solver/matvec
// (hecmw_matvec_get_timer, comm excluded as las_33.f90:345-349 does), solver/precond (hecmw_precond_get_timer), solver/comm --
// from HIP event pairs on the solver stream, collected whenever the host polls the Krylov state.  Off unless Iarray(22) >= 1.
struct PhaseClock {
  bool on = false;
  std::vector<hipEvent_t> ev;   // pairs: start, stop
  std::vector<int> kind;        // per pair: 0 matvec, 1 precond, 2 comm
  int used = 0;                 // pairs recorded since the last collect
  int open_kind[4] = {-1, -1, -1, -1};
  int depth = 0;
  double acc[3] = {0.0, 0.0, 0.0};  // seconds
};

// The value arena (round 4).  Where hipMalloc puts a BELL value array PHYSICALLY decides the speed class of the sweep that streams it:
// the identical SpMV on identical data runs at 1.03-1.05 ms or at 1.18 ms (10.1M DOF), and a fresh power-of-two request is a
// lottery between the two (scripts/r4/region_probe.py: 5 of 14 consecutive 8 GiB blocks slow on one box, the first allocation of
// the process among them).  What was ruled out with measurements of ONE context whose arrays were moved between allocations
// (scripts/r4/*.py, profiles/r04_placement_evidence.txt): the vectors x / y (twelve choices within 0.3 %), the column ids, the
// row pointers, the workgroup -> XCD map (eight maps within 1 %), the slice order, address translation (TCP_UTCL1 misses equal
// in both classes: 47.6k of 171M requests).  What holds on every box measured: inside ONE large allocation taken before anything
// else every offset runs in the same fast class (1.044-1.048 ms at sixteen offsets of a 64 GiB block on a box whose first
// power-of-two block ran at 1.180).  So the context takes one arena when it first learns the size of a large system -- before the
// CSR arrays, the layouts and the vectors -- and the value arrays of M, L and U live in it at 2 MiB-aligned offsets (first fit; the holes a rebuilt layout leaves are re-used).
// FX_ARENA_GB: 0 = off (every array its own hipMalloc), n = arena of at least n GiB (default 32), always a power of two.
struct DevArena {
  char *base = nullptr;
  size_t bytes = 0;
  std::vector<std::pair<size_t, size_t>> blocks;  // (offset, bytes) of the arrays placed in it, by offset: first fit at 2 MiB-aligned offsets
  size_t used() const { size_t u = 0; for (auto &b : blocks) u += b.second; return u; }
  int live() const { return (int)blocks.size(); }
};

// One pass of the auto-SIGMA_DIAG / METHOD2 loop of hecmw_solve_iterative (hecmw_solver_Iterative.f90:117-157): what the reference
// prints per pass (banner, ITERLOG lines, 'Increasing SIGMA_DIAG to') is replayed by the binding from this log.
struct AttemptLog {
  int method = 0;
  double sigma_diag = 1.0;       // SIGMA_DIAG in effect during the pass
  std::vector<double> hist;      // its residual-history lines
};

// What this rank asked of the transport, in program order: every all-reduce and every halo exchange (per neighbour: messages and
// bytes each way) since the communicator was set up.  All ranks of a correct run hold the same op count and the same sequence hash
// (the kinds and sizes of the collectives and the positions of the halo exchanges between them), and rank r's sends to p equal p's
// receives from r: bench.py --gpus N compares the ledgers over its gloo control plane (fx_comm_ledger), and its watchdog prints the
// local one when a step does not come back -- a mismatch shows as a message with the op index, not only as a hang.
struct CommLedger {
  int64_t n_ops = 0, n_allreduce = 0, allreduce_bytes = 0, n_halo = 0;
  uint64_t seq_hash = 1469598103934665603ull;  // FNV-1a over (kind, size) of the ops in order
  std::vector<int64_t> peer;                   // 5 per neighbour: rank, sends, send bytes, receives, receive bytes
  void mix(uint64_t v) { for (int k = 0; k < 8; k++) { seq_hash ^= (v >> (8 * k)) & 0xff; seq_hash *= 1099511628211ull; } }
};

struct fx_context {
  CommLedger ledger;
  void *nccl_halo = nullptr;  // the halo exchange's own communicator (ncclCommSplit of nccl; null: the exchanges share nccl)
  std::vector<AttemptLog> attempts;  // of the last fx_solve / fx_solve_resident (fx_solve_attempts)
  int device = 0;
  int n_cu = 256;  // compute units of the device
  hipStream_t stream = nullptr;
  DevCSR A;
  Ordering ord;
  Bell M;  // full matrix (D + AL + AU) in slot order, for SpMV
  bool have_profile = false, have_values = false, bell_valid = false;
  bool m_symbolic = false;   // M's source map matches the current ordering
  double *Bs = nullptr, *Xs = nullptr;  // B, X in slot space
  // host copies of the profile (ordering, conversion maps)
  std::vector<int32_t> h_indexL, h_itemL, h_indexU, h_itemU;
  // preconditioner
  int precond_kind = 0;  // 0 none, 1 SSOR (multicolour), 3 DIAG, 10 ILU0, 11 SSOR in the natural order (level-scheduled like ILU0)
  // PRECOND = 1 / 2 in the reference is TWO preconditioners: with one OpenMP thread the natural-order block Gauss-Seidel, with more
  // the RCM + multicolour ordering (hecmw_precond_SSOR_33.f90:93-114; 143 vs 204 iterations on SURVEY section 0's cube).  The GPU
  // default is the multicolour one (bandwidth-bound colour sweeps); FX_SSOR_NATURAL=1 / fx_set_option selects the other, whose
  // sweeps run level by level through the dataflow launch of ILU(0) (latency-bound: 1,044 levels at 150^3 nodes).
  bool ssor_natural = false;
  DiagDev diag;
  SsorDev ssor;
  bool precond_valid = false;
  int ssor_ncolor_in = 0;
  // SSOR numbering mode: 0 = Krylov vectors natural, sweep vector colour-major (hybrid);
  //                      1 = the whole Krylov loop in colour-major numbering (default).  FX_SSOR_MODE overrides.
  // Same-process A/B at 10.1M DOF with the tuned kernels: mode 0 336 it/s (SpMV 1.11 ms, SSOR 1.61 ms),
  // mode 1 351 it/s (SpMV 1.15 ms, SSOR 1.46 ms).
  int ssor_mode = 1;
  int ssor_spw = 1;            // consecutive slices per wave in the big-colour sweeps (FX_SSOR_SPW)
  int ssor_bs = 64;            // workgroup size of the colour sweeps: 64 (default) or 256. Measured 10M DOF: 1.78 -> 1.61 ms per apply
  int pipe_max_slices = 1 << 30;  // colours with more slices use the plain row loop (with 64-thread groups: pipelined everywhere wins, 1.61 vs 1.64/1.69 ms)
  int spmv_bs = 256;              // workgroup size of the SpMV (FX_SPMV_BS)
  // Colour-/level-major vectors: the SpMV walks the slices in SPATIAL order (slices of all colours that cover the same
  // part of the mesh next to each other), so the x entries a region gathers stay in L2 across its colours.  The data
  // layout is untouched; only the block -> slice map changes.  FX_SPMV_SPATIAL=0: ascending slices.
  bool spmv_spatial = true;
  // The wave-split sweep (k_ssor_color_split, split_wps waves per slice: each wave issues all the loads of its block pairs at
  // once, partial sums meet in LDS in a fixed order) hides latency by occupancy (64 VGPRs, 8 waves per SIMD) instead of the
  // software pipeline of k_ssor_color (174 VGPRs, 2 waves per SIMD).  Round 1-2: used for latency-bound launches only (colours /
  // ILU levels of <= 2048 slices): SSOR apply 1.46-1.49 -> 1.415 ms, ILU(0) level sweeps 19.9 -> 10.7 ms.  Round 3, same context
  // (scripts/r3/ab_opts.py): EVERY colour wave-split 1.502 -> 1.478 ms per apply, Eisenstat iteration 1.918 -> 1.871 ms; forcing the
  // pipelined kernel to 3 waves per SIMD (__launch_bounds__(BS, 3): 168 VGPRs + 6-8 spills) 1.47 -> 1.58 ms.  So the default is
  // "all colours"; FX_SPLIT_MAX_SLICES brings the pipelined kernel back for larger colours, FX_SPLIT_WPS (2, 4, 8; 0 = auto:
  // 8 for ILU(0) levels, 4 for SSOR colours) sets the waves per slice.
  int split_max_slices = 1 << 30;
  int split_wps = 0;
  // Dataflow triangular sweeps (k_tri_dataflow): one persistent launch per apply, rows synchronised through
  // sentinel-tagged data instead of one launch per colour / level.  FX_DATAFLOW=0 off, 1 (default) ILU(0) levels,
  // 2 also the multicolour SSOR (measured slower than the colour launches: 1.59-1.71 against 1.50 ms per apply -- the big
  // colours are bandwidth-bound and a few hundred workgroups with one slice each in flight do not saturate HBM).
  // FX_DF_GRID workgroups (default: one per two CUs), FX_DF_WPS waves per slice (2, 4, 8), FX_DF_POLL 0 = every poll
  // re-reads all entries, 1 = only the unpublished ones, FX_DF_SLEEP s_sleep(1)s between polls.
  // Measured at 10.1M DOF, ILU(0), 1,044 levels (same process, scripts/experiments/ab_dataflow3.py): launch per level
  // 10.7 ms per apply; dataflow 4.80-4.87 ms with 128 workgroups x 8 waves (96: 5.3, 160: 4.89, 192: 4.92, 256: 5.0-5.1,
  // 512: 8.0; 4 waves per slice 5.2; re-reading all entries per poll 5.6; two polls in flight 5.9; 0.4-1.5 us of sleep
  // between polls 4.9-5.3): 2,088 dependent hand-offs of ~2.3 us each -- the polls of the waiting workgroups compete
  // with the frontier's hand-offs, so fewer pollers and fewer re-read entries are faster.
  // Round 3, same context (scripts/r3/ab_opts.py, ILU(0) at 10.1M DOF): sweep vectors in the [slice][k][lane] layout (df_soa) 4.73 -> 4.44 ms
  // per apply; with that layout one workgroup per CU instead of one per two 4.44 -> 4.27; no sleep between polls 4.27 -> 4.20; the
  // hand-off vectors in uncached device memory 4.20 -> 4.15 (not kept); re-reading all entries per poll 5.11; 4 waves per slice 5.04.
  // Round 4: the re-reads of a polling pass are issued together (no branch, hence no wait, between them: df_gather) -- one round trip per
  // pass instead of one per missing entry: 4.19 -> 3.43 ms per apply; with that, re-reading every entry per pass (FX_DF_POLL=0) is the
  // faster form, 3.33 ms (scripts/r3/ab_opts.py, same context), and the default.
  int df_mode = 1, df_grid = 0, df_wps = 8, df_poll = 0, df_sleep = 0;
  int df_presleep = 2;        // FX_DF_PRESLEEP: x 0.1 us of sleep per level of lead before a workgroup starts to poll for its next slice (round 4, same context: 0 -> 3.27, 1 -> 3.14, 2 -> 3.13, 3 -> 3.13, 5 -> 3.16, 8 -> 3.24, 12 -> 3.57 ms per apply)
  bool df_soa = true;         // private sweep vectors of the dataflow sweeps in the [slice][k][lane] layout (FX_DF_SOA=0: 3 s + k)
  int df_grid_max[3] = {128, 128, 128};  // co-resident workgroups of k_tri_dataflow<2 / 4 / 8 waves> (occupancy query at fx_create)
  int df_grid_last = 0;       // workgroups of the last dataflow launch (after the co-residency clamp)
  bool dbg_df_fail = false;   // test hook (FX_DEBUG_DF_FAIL): the dataflow launches report a timeout at once
  int df_fallbacks = 0;       // times a timed-out dataflow sweep made this context fall back to the launch-per-level sweeps (fx_get_stats)
  int32_t *df_err = nullptr;  // device: raised by a sweep whose bounded spin ran out
  // Plane march (fx_march.h, k_tri_march): the level-scheduled sweeps (ILU(0), natural-order SSOR) with whole chunks of rows per
  // workgroup, dependencies inside a chunk through an LDS ring, between chunks through the sentinel-tagged vectors.  FX_MARCH=0 (default)
  // off: bit-identical to k_tri_dataflow but, as measured at 10.1 M DOF, slower (6.3 against 3.3 ms per apply; DESIGN.md section 4 has the
  // per-round timeline); 1 when the cost model prefers it, 2 whenever the structure admits it.  FX_MARCH_CHUNK rows per chunk (0 = from
  // the matrix profile), FX_MARCH_WAVES pair waves per workgroup (1, 2, 3; 0 = from the level sizes).
  int march_mode = 0, march_chunk = 0, march_waves = 0, march_grid = 0, march_xcd = 1;
  int march_grid_max[4] = {0, 0, 0, 0};  // co-resident workgroups of k_tri_march<1 / 2 / 3 pair waves>
  int march_launches = 0;                // applies that took the march (fx_march_report)
  // software-pipelined row loop (2-deep: values + gathers of pair i+1 and ids of pair i+2 in flight while pair i
  // is multiplied; 116 VGPRs, 4 waves/SIMD).  Measured on MI355X at 10.1M DOF with the final layout (odd-tail BELL,
  // non-temporal stream loads): SpMV 1.14-1.17 -> 1.105-1.11 ms; colour sweeps 1.685 -> 1.61 ms per apply.
  // (With the earlier padded layout it cost the SpMV 1.19 -> 1.30 ms: re-measure when the layout changes.)
  bool pipe_spmv = true, pipe_ssor = true;  // FX_PIPE_SPMV / FX_PIPE_SSOR override
  // Eisenstat's form of CG + SSOR (the default since round 4; FX_EISENSTAT=0 runs hecmw_solve_CG's loop as written): with
  // M = (D~+L) D~^-1 (D~+U) and A = (D~+L) + (D~+U) + (D - 2D~), one backward and one forward triangular sweep per iteration deliver
  // p, q = A p and w = (D~+L)^-1 q -- the matrix is streamed ONCE instead of twice (SpMV + the two half sweeps).  Same x_k, r_k,
  // rho_k, alpha_k in exact arithmetic; the summation order of q and of the dot products differs, so histories agree to rounding
  // (the parity bounds of tests/test_gpu_parity.py hold in both modes), not bit for bit.  It is taken only where it is exact:
  // multicolour SSOR with iterPREmax = 1, colour-major numbering, preconditioner built from the resident values (subdomains: plus
  // the halo term H p); anything else -- a recycled preconditioner, iterPREmax != 1, BiCGSTAB -- runs the standard loop on its own.
  bool eisenstat = true, eis_active = false;
  // Eisenstat's form: rho of the NEXT iteration was already reduced together with ||r||^2 of the last one (OP_RESID_RHO: one scalar
  // stage -- on a decomposed system one 2-double all-reduce -- instead of two).  False after a begin, after an iteration that
  // recomputed the true residual and after a true-residual check: those refresh t and leave new rho partials (FX_EIS_MERGE=0: never).
  bool eis_rho_done = false, eis_merge = true;
  int eis_grid = 0;  // FX_EIS_GRID: most workgroups of one colour launch of the wave-split Eisenstat sweeps (a workgroup then walks slices b, b + grid, ...); 0 = one workgroup per slice
  bool dbg_onecolor = false; // measurement only (FX_DEBUG_ONECOLOR): the half sweeps as one launch each, dependencies ignored
  bool eis_fuse = true;      // direction update fused into the backward sweep (FX_EIS_FUSE=0: k_cg_update_p + the plain sweep)
  int64_t values_epoch = 0;  // counts the refreshes of the SpMV layout's values
  bool layout_device = true;        // BELL source maps built by k_bell_count / k_bell_map (FX_LAYOUT_DEVICE=0: host threads)
  int32_t mc_batch = 32;            // rounds of the device multicolouring between two looks at the queue length by the host (FX_MC_BATCH)
  int32_t mc_device_min = 100000;   // block rows from which the multicolouring of the SSOR set-up runs on the device (FX_MC_DEVICE_MIN)
  bool val2_pow2 = true;            // BELL value arrays of a gigabyte or more that do not fit the arena: ask hipMalloc for the next power of two (FX_VAL2_POW2=0: the exact size)
  DevArena arena;                   // the value arena of this context (see DevArena)
  // The arena makes the class the same for every array and every offset, it does not make it the fast one: a box was met whose
  // first 32 GiB were a slow region (SpMV 1.17 ms in 8 of 8 processes).  So the arena is VERIFIED once, when M's values are first
  // filled: the loop's own SpMV (p = W[2] -> q = W[1], fused p.q partial) is timed on it (1 untimed + 3 launches, 5 ms); below
  // arena_good_gbs the context takes ANOTHER arena (the earlier ones held meanwhile, so that it is different memory), moves the
  // value arrays and times again, at most arena_tries arenas; the fastest is kept, the others are freed before the set-up returns.
  // Measured consistency of this timing with the loop's: within 0.3 % (profiles/r04_placement_evidence.txt, "loop" vs "probe").
  int arena_tries = 4;              // FX_ARENA_TRIES (1 = no verification)
  double arena_good_gbs = 6450.0;   // the fast classes stream 6.57-6.8 TB/s algorithmic, the slow ones 5.9-6.4
  std::vector<DevArena> arena_tried; // earlier arenas of a verification in progress (held so that the next request is different memory)
  std::vector<float> arena_ms;      // what the verification measured, in order (fx_placement_report)
  bool arena_verified = false;
  size_t arena_min_bytes = (size_t)32 << 30;  // FX_ARENA_GB (GiB, fractions allowed; 0 = no arena)
  size_t arena_threshold = (size_t)1 << 30;   // estimated value-array bytes from which a system gets an arena (FX_ARENA_THRESHOLD_MB)
  size_t arena_max_bytes = 0;                 // FX_ARENA_MAX_MB: never take more than this (0 = no cap); arrays that do not fit get their own allocations
  int32_t bfs_batch = 16;           // levels of the device level ordering between two looks at the level state by the host (FX_BFS_BATCH)
  int32_t bfs_device_min = 100000;  // block rows from which the level ordering of the SSOR set-up runs on the device (FX_BFS_DEVICE_MIN)
  // work vectors (3*NP each)
  double *W[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool precond_valid_sweeps = false;  // multicolour SSOR layouts filled
  int iterpremax = 1;  // additive-Schwarz sweeps of hecmw_precond_33_apply
  int32_t wlen = 0;
  double *scale_vec = nullptr;  // SCALING=YES: 1/sqrt|diag|, reference numbering, 3*NP (+ slack)
  double *extra = nullptr;  // further work vectors (GMRES basis, GPBiCG), extra_n x extra_len
  int extra_n = 0;
  int32_t extra_len = 0;
  // reductions
  double *partials = nullptr;  // 3 * max_partials
  int32_t max_partials = 0;
  double *red_out = nullptr;  // small device scratch for reduced sums (8 doubles)
  KrylovState *st = nullptr;   // device
  KrylovState *st_host = nullptr;  // pinned
  double *hist = nullptr;      // device residual history
  int32_t hist_cap = 0;
  int k_method = 1, k_maxit = 0, k_it = 1;  // host mirror of the running Krylov loop
  int k_method_last = 1;       // METHOD of the last fx_solve_resident attempt
  double host_dbg[16] = {0};   // scalars of the host-driven recurrences (fx_debug_state)
  // Launch-bound sizes (<= graph_max_rows block rows, single rank): one CG / BiCGSTAB iteration is captured once per
  // solve into hipGraphs (ordinary iteration; the one that recomputes r = b - A x; in Eisenstat's form each with / without its own rho stage) and replayed -- every kernel
  // argument is constant over a solve, what changes lives in the device-resident KrylovState.
  // FX_GRAPH=0 off, 1 auto (default: SSOR / ILU(0) only, where an iteration is 40+ launches), 2 always.
  int graph_mode = 1;
  int32_t graph_max_rows = 1 << 19;
  bool k_graph = false;
  hipGraphExec_t g_iter[4] = {nullptr, nullptr, nullptr, nullptr};  // by kind of iteration: bit 0 = recomputes the true residual, bit 1 = starts with its own rho stage
  // communication
  int rank = 0, nranks = 1;
  int view_petot = 1;        // PETOT of the comm view the profile came with (require_transport)
  int32_t nn_internal = 0;
  void *nccl = nullptr;
  // host-staged communication hooks (testing / non-RCCL transports): see fx_comm_set_host_callbacks
  void (*cb_halo)(const double *, double *, void *) = nullptr;
  void (*cb_allreduce)(double *, int, void *) = nullptr;
  void *cb_user = nullptr;
  double *h_send = nullptr, *h_recv = nullptr;  // pinned staging
  HaloDev halo;
  // Interior / boundary overlap of the SpMV (the design the reference sketched and left commented out,
  // hecmw_solver_las_33.f90:242-246, :312-343 with hecmw_solver_SR_33.F90:129-273): pack on the solver stream, the
  // exchange + unpack on comm_stream while the interior rows are multiplied, the boundary rows after it.  FX_OVERLAP=0: serial.
  bool overlap = true;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_halo = nullptr;
  NlDev nl;
  ElemColors asm_colors;  // fx_assemble_c3d8
  void *nn = nullptr;  // NnDev (fx_nn_host.h): systems with NDOF != 3
  const void *host_D = nullptr, *host_AL = nullptr, *host_AU = nullptr;  // the caller's arrays of the last value upload (fx_solve)
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  PhaseClock clock;
};

