// Plane march of the level-scheduled triangular sweeps (ILU(0) hecmw_precond_BILU_33.f90:90-157, natural-order SSOR
// hecmw_precond_SSOR_33.f90:300-410).  Included by fistr_hip.hip after fx_kernels.h and the host helpers.
//
// STATUS (round 4): correct -- bit-identical to the level sweeps on every mesh tried, 10.1 M DOF included -- and SLOWER than k_tri_dataflow
// (6.3 against 3.1 ms per apply), hence opt-in (FX_MARCH=2).  DESIGN.md section 4 "Round 4: the plane march" has the measurements: a free
// round costs 0.84 us for 16 rows, a round that waits for another chunk 2.4-3.6 us, and the plane-to-plane lag alone (149 x 5-10 us) is
// what k_tri_dataflow needs for its 1,044 hand-offs.  Kept as an independent second implementation of the sweeps and as a measuring tool
// (fx_debug_march_trace / _rounds).
//
// k_tri_dataflow hands every dependency level from workgroup to workgroup through memory: 2 x 1,044 hand-offs of 1.5-2 us at 10.1 M DOF.
// Here a workgroup keeps a CHUNK -- a contiguous range of the natural numbering, on a structured mesh one plane of nodes (or a
// fraction of it) -- for the whole sweep.  Inside a chunk the rows run in ROUNDS: the chunk's own dependency levels (counting only
// the lower neighbours that lie in the chunk), a level of more than R rows split.  What a row needs from its own chunk was produced at
// most FX_MARCH_NEAR rounds earlier and is still in an LDS ring ("near"); what it needs from another chunk -- or from further back in
// its own -- is gathered from the sentinel-tagged sweep vector in memory ("far"), one round ahead of its use, and re-read until it is
// there.  So a dependency level costs one workgroup barrier plus an in-LDS chain instead of a publish -> poll round trip, and the
// round trips that remain (plane to plane) have several rounds of slack.
//
// Layout (MarchProg): the matrix in execution order.  A row is 8 consecutive lanes: lane g < 7 holds block pair g of the row's list
// (the SAME pairs, in the same order, as the W = 8 waves of k_ssor_color_split / k_tri_dataflow take), lane 7 the LU of the diagonal
// block in the .x halves.  Per round [9][8 n] double2 + [8 n] int2 column codes: every load of a wave is one contiguous kilobyte.
// Arithmetic: partial sum of lane g = that of wave g; the finishing lane adds them in lane order; so z is bit-identical to the other
// two kernels with 8 waves per slice (tests/test_gpu_parity.py::test_march_sweeps_equal_level_sweeps_bitwise).
//
// What the round-3 probe (scripts/experiments/tile_march/) taught, and this kernel obeys:
//  * every wave issues the same vector-memory instructions in every round (clamped addresses, no branch around a load), so the
//    compiler can count the loads in flight and the prefetches survive a round (`s_waitcnt vmcnt(N)`, N > 0);
//  * the pair waves never store to memory: a separate wave publishes the finished round from the ring (a wave's write-through
//    stores and its loads share one in-order counter);
//  * the workgroup barrier waits for LDS only (`s_waitcnt lgkmcnt(0); s_barrier`), not for the loads in flight.
#pragma once

#define FX_MARCH_RING 8                     // rounds of results kept in LDS
#define FX_MARCH_NEAR (FX_MARCH_RING - 1)   // an in-chunk dependency at most this many rounds back is gathered from the ring
#define FX_MARCH_MAXBLOCKS 14               // blocks of a row's lower (upper) part: 7 pair lanes

struct MarchSide { const int32_t *round_ptr, *rstart; const double2 *val; const int2 *col; };
struct MarchArgs {
  MarchSide F, B;
  int32_t nchunks;
  const double *r, *dummy;  // dummy: three zeros, what a lane that gathers nothing reads
  double *zf, *z;
  int32_t *err;
  int nsleep, xcd;
  unsigned long long *rtrace;  // diagnostics: per round of the forward sweep of chunk rtrace_chunk: ticks at the barrier, bit 63 = the round waited
  int rtrace_chunk;
  unsigned long long *trace;  // diagnostics (fx_debug_march_trace): per chunk 8 words -- forward start / end, backward start / end (100 MHz ticks), rounds that had to wait and polls, forward / backward
};

__device__ __forceinline__ void march_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ bool march_tag(double v) { return __double_as_longlong(v) == FX_DF_SENTINEL; }

// tag fill of the two sweep vectors (3 n doubles each; write-through stores, see k_df_fill)
__global__ __launch_bounds__(256) void k_march_tags(int64_t n8, double *__restrict__ a, double *__restrict__ b) {
  const double tag = __longlong_as_double(FX_DF_SENTINEL);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    __hip_atomic_store(a + i, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(b + i, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// values of one program from the reference arrays (run at every numeric refresh).  One workgroup per round.
__global__ __launch_bounds__(256) void k_march_fill(const int32_t *__restrict__ rstart, const int2 *__restrict__ src,
                                                    const double *__restrict__ D, const double *__restrict__ AL,
                                                    const double *__restrict__ AU, double sigma_diag, double2 *__restrict__ val) {
  const int rs = rstart[blockIdx.x], nl = 8 * (rstart[blockIdx.x + 1] - rs);
  for (int t = threadIdx.x; t < nl; t += 256) {
    const int2 sc = src[(size_t)8 * rs + t];
    double ax[9], ay[9];
#pragma unroll
    for (int e = 0; e < 9; e++) { ax[e] = 0.0; ay[e] = 0.0; }
    if ((t & 7) == 7) {  // the diagonal block's LU (k_alu_setup)
      const double *d = D + (size_t)9 * (sc.x / 3);
#pragma unroll
      for (int e = 0; e < 9; e++) ax[e] = d[e];
      ax[0] *= sigma_diag; ax[4] *= sigma_diag; ax[8] *= sigma_diag;
      lu33_dev(ax);
    } else {
      if (sc.x >= 0) {
        const double *p = ((sc.x % 3) == 1 ? AL : AU) + (size_t)9 * (sc.x / 3);
#pragma unroll
        for (int e = 0; e < 9; e++) ax[e] = p[e];
      }
      if (sc.y >= 0) {
        const double *p = ((sc.y % 3) == 1 ? AL : AU) + (size_t)9 * (sc.y / 3);
#pragma unroll
        for (int e = 0; e < 9; e++) ay[e] = p[e];
      }
    }
    double2 *v = val + (size_t)72 * rs + t;
#pragma unroll
    for (int e = 0; e < 9; e++) v[(size_t)e * nl] = make_double2(ax[e], ay[e]);
  }
}

// The pair waves of one chunk, one direction.  r0 <= rounds < r1 (r1 > r0).  Barriers executed: r1 - r0 + 1.
// Software pipeline, all in registers.  The matrix stream (values + column codes) does not depend on the sweep: it is loaded THREE rounds
// ahead into four buffers used in turn; the far entries and the right-hand side of a round are loaded ONE round ahead (their addresses
// are the column codes of the next round, long there) into two buffers.  The loop is unrolled by four, so a loaded value is never
// copied and no wait for a prefetch sits at the end of a round.  Loads return in order: within a round the far gathers are issued
// before the stream loads, so waiting for them does not wait for the stream.
// Round descriptors are scalars ([s, e) = march positions of the rounds rho .. rho + 3); the row count of round rho + 4 rides in the spare
// half of the finishing lanes' column codes (0 = past the chunk's end: the last round again).
template <bool FWD, int NW>
struct MarchPairs {
  static constexpr int R = 8 * NW;
  struct Buf { double2 a[9]; int2 k; };
  const MarchSide &P;
  const double *__restrict__ rvec;
  const double *zsrc, *zown, *dummy;
  double *ring;
  int32_t *ringrow, *ringn;
  double (*scr)[4][64];
  int32_t *__restrict__ err;
  bool &dead;
  const int nsleep;
  const int t, lane, w, g, q, gk, lead;
  int sA, eA, sB, eB, sC, eC, sD, eD;
  unsigned waited = 0, polls = 0;  // diagnostics: rounds of this wave that found an entry missing, re-reads
  unsigned long long *rtr = nullptr;

  __device__ __forceinline__ MarchPairs(const MarchSide &P_, const double *rvec_, const double *zsrc_, const double *zown_,
                                        const double *dummy_, double *ring_, int32_t *ringrow_, int32_t *ringn_, double (*scr_)[4][64],
                                        int32_t *err_, bool &dead_, int nsleep_)
      : P(P_), rvec(rvec_), zsrc(zsrc_), zown(zown_), dummy(dummy_), ring(ring_), ringrow(ringrow_), ringn(ringn_), scr(scr_), err(err_),
        dead(dead_), nsleep(nsleep_), t(threadIdx.x), lane(threadIdx.x & 63), w(threadIdx.x >> 6), g(threadIdx.x & 7), q(threadIdx.x >> 3),
        gk((threadIdx.x & 7) < 2 ? (threadIdx.x & 7) : 2), lead((threadIdx.x & 63) | 7) {}

  // values and column codes of the round [s, e): one contiguous kilobyte per wave and load (uniform base + 32-bit lane offset)
  __device__ __forceinline__ void load_stream(int s, int e, Buf &b) const {
    const unsigned nl = 8u * (unsigned)(e - s), tt = (unsigned)t < nl ? (unsigned)t : nl - 1u;
    const char *vb = (const char *)(P.val + (size_t)72 * s), *cb = (const char *)(P.col + (size_t)8 * s);
    b.k = ld_stream((const int2 *)(cb + 8u * tt));  // first: the codes are wanted a round before the values (as addresses of the far gathers)
#ifdef FX_MARCH_EXP_NOVALS  // timing experiment: one value word instead of nine
    b.a[0] = ld_stream((const double2 *)(vb + 16u * tt));
#pragma unroll
    for (int i = 1; i < 9; i++) b.a[i] = b.a[0];
#else
#pragma unroll
    for (int i = 0; i < 9; i++) b.a[i] = ld_stream((const double2 *)(vb + (16u * tt + 16u * (unsigned)i * nl)));
#endif
  }
  // the far entries of the round whose column codes are c and which has nrows rows: lanes that gather nothing read the dummy entry
  __device__ __forceinline__ void load_far(const int2 &c, int nrows, double (&v)[6]) const {
    const bool on = g < 7 && q < nrows;
    const double *pa = (on && c.x >= 0) ? zsrc + (size_t)3 * c.x : dummy, *pb = (on && c.y >= 0) ? zsrc + (size_t)3 * c.y : dummy;
#if defined(FX_MARCH_EXP_NOFAR)  // timing experiment: no far gathers
    for (int i = 0; i < 6; i++) v[i] = 1e-3 * (double)(c.x + i);
#elif defined(FX_MARCH_EXP_FARPLAIN)  // timing experiment: far gathers as ordinary cached loads
    for (int i = 0; i < 3; i++) { v[i] = pa[i]; v[3 + i] = pb[i]; }
#else
#pragma unroll
    for (int i = 0; i < 3; i++) { v[i] = df_load(pa + i); v[3 + i] = df_load(pb + i); }
#endif
  }
  // lanes 0..2 of a row: component g of its right-hand side (forward) / of its own forward value (backward)
  __device__ __forceinline__ double load_f(const int2 &c) const {
    const int row = __shfl(c.x, lead, 64);
    const double *p = (FWD ? rvec : zown) + (size_t)3 * row + gk;
    return FWD ? ld_stream(p) : df_load(p);
  }

  __device__ __forceinline__ void round(const int rl_local, const Buf &cur, const Buf &nxt, Buf &ld, double (&x)[6], double &f, double (&xn)[6],
                                        double &fn) {
    load_far(nxt.k, eB - sB, xn);
    fn = load_f(nxt.k);
    march_lds_barrier();  // the ring holds the previous round
    const int n = eA - sA;
    const bool act = q < n;
    const int cx = cur.k.x, cy = cur.k.y;
    bool wflag = false;
    {
      // cheap test first: the tag is all ones, and lanes that need nothing have read the dummy entry (0.0)
      unsigned h = (unsigned)__double2hiint(x[0]);
#pragma unroll
      for (int i = 1; i < 6; i++) h = max(h, (unsigned)__double2hiint(x[i]));
      if (!FWD) h = max(h, (unsigned)__double2hiint(f));
      bool any = h == 0xFFFFFFFFu;
#ifdef FX_MARCH_EXP_NOPOLL  // timing experiment: never wait
      any = false;
#endif
      if (__any(any) && !dead) {  // a far entry (or this row's own forward value) may not be published yet: look exactly, re-read what is missing
        const bool farx = act && g < 7 && cx >= 0, fary = act && g < 7 && cy >= 0, ownf = !FWD && act && g < 3;
        bool m[7];
        any = false;
#pragma unroll
        for (int i = 0; i < 3; i++) { m[i] = farx && march_tag(x[i]); m[3 + i] = fary && march_tag(x[3 + i]); any |= m[i] | m[3 + i]; }
        m[6] = ownf && march_tag(f);
        any |= m[6];
        if (__any(any)) {
          waited++; wflag = true;
          // While this round waits, the NEXT round's far entries -- requested at the top of this round, i.e. before the wait -- go
          // stale: they are re-read in the same polls, so that a wave which had to wait once does not wait in every round after it.
          const int nx = nxt.k.x, ny = nxt.k.y;
          const bool actn = q < eB - sB, nfarx = actn && g < 7 && nx >= 0, nfary = actn && g < 7 && ny >= 0, nownf = !FWD && actn && g < 3;
          const int row = __shfl(cx, lead, 64), rown = __shfl(nx, lead, 64);
          unsigned long long t0 = 0;
          int back = 0;  // back-off: a wave that waits long (a chunk far ahead of the frontier) must not flood the fabric with re-reads
          for (unsigned spins = 1;; spins++) {
            polls++;
            for (int s = 0; s < back + nsleep; s++) __builtin_amdgcn_s_sleep(4);  // 256 clocks each
            back = back < 4 ? 2 * back + 1 : back;  // 0, 1, 3, 7 x 0.1 us
            any = false;
            // all re-reads of a poll in flight together (a lane that misses nothing re-reads the dummy entry): one round trip per poll
            double v[7], vn[7];
            const double *pxa = zsrc + (size_t)3 * cx, *pya = zsrc + (size_t)3 * cy, *pfa = zown + (size_t)3 * row + gk;
            const double *pxn = zsrc + (size_t)3 * nx, *pyn = zsrc + (size_t)3 * ny, *pfn = zown + (size_t)3 * rown + gk;
            bool mn[7];
#pragma unroll
            for (int i = 0; i < 3; i++) { mn[i] = nfarx && march_tag(xn[i]); mn[3 + i] = nfary && march_tag(xn[3 + i]); }
            mn[6] = nownf && march_tag(fn);
#pragma unroll
            for (int i = 0; i < 3; i++) {
              v[i] = df_load(m[i] ? pxa + i : dummy); v[3 + i] = df_load(m[3 + i] ? pya + i : dummy);
              vn[i] = df_load(mn[i] ? pxn + i : dummy); vn[3 + i] = df_load(mn[3 + i] ? pyn + i : dummy);
            }
            v[6] = df_load(m[6] ? pfa : dummy); vn[6] = df_load(mn[6] ? pfn : dummy);
#pragma unroll
            for (int i = 0; i < 6; i++) {
              if (m[i]) { x[i] = v[i]; m[i] = march_tag(v[i]); any |= m[i]; }
              if (mn[i]) xn[i] = vn[i];
            }
            if (m[6]) { f = v[6]; m[6] = march_tag(v[6]); any |= m[6]; }
            if (mn[6]) fn = vn[6];
            if (!__any(any)) break;
            if ((spins & 31u) == 0u) {
              const unsigned long long now = __builtin_amdgcn_s_memrealtime();
              if (t0 == 0) t0 = now;
              const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (e != 0 || now - t0 > FX_DF_TIMEOUT_TICKS) {
                if (e == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
              }
            }
          }
          // nothing in flight when the slow path rejoins: the refreshed entries are the youngest loads, and a join with them pending would
          // make every round's wait for its far entries as strict as this path needs it (the stream loads would lose a round of flight)
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
      }
    }
    // the matrix stream of round + 3: after the check, so that a wave which has to poll (its re-reads are the youngest loads, waiting for
    // them waits for everything) does not wait for stream loads it has only just issued
    load_stream(sD, eD, ld);
    if (rtr && t == 0) rtr[rl_local] = __builtin_amdgcn_s_memrealtime() | (wflag ? 1ull << 63 : 0ull);
    // near gathers from the ring (lanes that gather nothing read slot 0)
    const int sx = cx <= -2 ? -cx - 2 : 0, sy = cy <= -2 ? -cy - 2 : 0;
    double xv[6];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double nx = ring[3 * sx + i], ny = ring[3 * sy + i];
      xv[i] = cx >= 0 ? x[i] : (cx <= -2 ? nx : 0.0);
      xv[3 + i] = cy >= 0 ? x[3 + i] : (cy <= -2 ? ny : 0.0);
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    bell_pair_fma(cur.a, xv, s0, s1, s2);
    scr[w][0][lane] = s0; scr[w][1][lane] = s1; scr[w][2][lane] = s2; scr[w][3][lane] = f;
    __builtin_amdgcn_wave_barrier();
    if (g == 7 && act) {  // the finishing lane: partial sums in lane order (= wave order of the split kernels), 3x3 substitution
      const double *p0 = &scr[w][0][lane - 7], *p1 = &scr[w][1][lane - 7], *p2 = &scr[w][2][lane - 7], *pf = &scr[w][3][lane - 7];
      s0 = p0[0]; s1 = p1[0]; s2 = p2[0];
#pragma unroll
      for (int i = 1; i < 7; i++) { s0 += p0[i]; s1 += p1[i]; s2 += p2[i]; }
      const double f0 = pf[0], f1 = pf[1], f2 = pf[2];
      double u[9];
#pragma unroll
      for (int e = 0; e < 9; e++) u[e] = cur.a[e].x;
      double x1, x2, x3;
      if (FWD) {
        x1 = f0 - s0; x2 = f1 - s1; x3 = f2 - s2;
        lusolve33_dev(u, x1, x2, x3);
      } else {
        lusolve33_dev(u, s0, s1, s2);
        x1 = f0 - s0; x2 = f1 - s1; x3 = f2 - s2;
      }
      const int slot = (rl_local & (FX_MARCH_RING - 1)) * R + q;
      ring[3 * slot] = x1; ring[3 * slot + 1] = x2; ring[3 * slot + 2] = x3;
      ringrow[slot] = cx;
      if (t == 7) ringn[rl_local & (FX_MARCH_RING - 1)] = n;
    }
    __builtin_amdgcn_wave_barrier();
    // descriptors: this round's finishing lanes carry the row count of round + 4
    const int nn = __builtin_amdgcn_readlane(cy, 7);
    sA = sB; eA = eB; sB = sC; eB = eC; sC = sD; eC = eD;
    if (nn > 0) { sD = eD; eD = eD + nn; }
  }

  __device__ __forceinline__ void run(const int r0, const int r1) {
    const int rl = r1 - 1;
    {
      const int rb = r0 + 1 < rl ? r0 + 1 : rl, rc = r0 + 2 < rl ? r0 + 2 : rl, rd = r0 + 3 < rl ? r0 + 3 : rl;
      sA = __builtin_amdgcn_readfirstlane(P.rstart[r0]); eA = __builtin_amdgcn_readfirstlane(P.rstart[r0 + 1]);
      sB = __builtin_amdgcn_readfirstlane(P.rstart[rb]); eB = __builtin_amdgcn_readfirstlane(P.rstart[rb + 1]);
      sC = __builtin_amdgcn_readfirstlane(P.rstart[rc]); eC = __builtin_amdgcn_readfirstlane(P.rstart[rc + 1]);
      sD = __builtin_amdgcn_readfirstlane(P.rstart[rd]); eD = __builtin_amdgcn_readfirstlane(P.rstart[rd + 1]);
    }
    Buf b0, b1, b2, b3;
    double xa[6], xb[6], fa, fb;
    load_stream(sA, eA, b0);
    load_stream(sB, eB, b1);
    load_stream(sC, eC, b2);
    load_far(b0.k, eA - sA, xa);
    fa = load_f(b0.k);
    // nothing of the prologue in flight at the loop header: the wait insertion merges the header's predecessors, and a load pending on
    // the entry edge only would put its wait into every round (once per chunk, this costs one latency)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    for (int rho = r0;; rho += 4) {
      round(rho - r0, b0, b1, b3, xa, fa, xb, fb);
      if (rho + 1 >= r1) break;
      round(rho + 1 - r0, b1, b2, b0, xb, fb, xa, fa);
      if (rho + 2 >= r1) break;
      round(rho + 2 - r0, b2, b3, b1, xa, fa, xb, fb);
      if (rho + 3 >= r1) break;
      round(rho + 3 - r0, b3, b0, b2, xb, fb, xa, fa);
      if (rho + 4 >= r1) break;
    }
    march_lds_barrier();  // the last round is in the ring (the publisher takes it from there)
  }
};

// The publishing wave: after barrier k it stores round r0 + k - 1 from the ring with write-through stores.
template <int NW>
__device__ __forceinline__ void march_publisher(const int r0, const int r1, double *dst, const double *ring, const int32_t *ringrow,
                                                const int32_t *ringn) {
  constexpr int R = 8 * NW;
  const int lane = threadIdx.x & 63;
  for (int rho = r0; rho <= r1; rho++) {
    march_lds_barrier();
    if (rho > r0) {
      const int sl = (rho - 1 - r0) & (FX_MARCH_RING - 1), n3 = 3 * ringn[sl], base = sl * R;
      for (int i = lane; i < n3; i += 64) {
        const int row = ringrow[base + i / 3];
        df_store(dst + (size_t)3 * row + i % 3, ring[3 * base + i]);
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(256) void k_tri_march(MarchArgs A, const int32_t *__restrict__ gate) {
  if (gate && *gate != 0) return;
  if (A.nsleep < 0) {  // test hook (FX_DEBUG_DF_FAIL), as k_tri_dataflow
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(A.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  constexpr int R = 8 * NW;
  __shared__ double ring[FX_MARCH_RING * R * 3];
  __shared__ int32_t ringrow[FX_MARCH_RING * R];
  __shared__ int32_t ringn[FX_MARCH_RING];
  __shared__ double scr[NW][4][64];
  const int G = gridDim.x;
  // workgroups go round-robin over the XCDs: keep consecutive chunks (plane k, plane k + 1) on one XCD (grid = a multiple of 8)
  const int first = A.xcd ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const bool pub = ((int)threadIdx.x >> 6) == NW;
  bool dead = false;
  int last = -1;
  const unsigned long long t_entry = A.rtrace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  for (int ch = first; ch < A.nchunks; ch += G) {
    const int r0 = A.F.round_ptr[ch], r1 = A.F.round_ptr[ch + 1];
    last = ch;
    if (r1 <= r0) continue;
    if (pub) march_publisher<NW>(r0, r1, A.zf, ring, ringrow, ringn);
    else {
      MarchPairs<true, NW> mp(A.F, A.r, A.zf, A.zf, A.dummy, ring, ringrow, ringn, scr, A.err, dead, A.nsleep);
      if (A.rtrace && ch == A.rtrace_chunk) { mp.rtr = A.rtrace; if (threadIdx.x == 0) A.rtrace[r1 - r0] = t_entry; }
      const unsigned long long t0 = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
      mp.run(r0, r1);
      if (A.trace && threadIdx.x == 0) {
        unsigned long long *tr = A.trace + (size_t)8 * ch;
        tr[0] = t0; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[4] = mp.waited; tr[5] = mp.polls;
      }
    }
  }
  for (int ch = last; ch >= 0; ch -= G) {  // backward: the same chunks, last first
    const int r0 = A.B.round_ptr[ch], r1 = A.B.round_ptr[ch + 1];
    if (r1 <= r0) continue;
    if (pub) march_publisher<NW>(r0, r1, A.z, ring, ringrow, ringn);
    else {
      MarchPairs<false, NW> mp(A.B, A.r, A.z, A.zf, A.dummy, ring, ringrow, ringn, scr, A.err, dead, A.nsleep);
      const unsigned long long t0 = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
      mp.run(r0, r1);
      if (A.trace && threadIdx.x == 0) {
        unsigned long long *tr = A.trace + (size_t)8 * ch;
        tr[2] = t0; tr[3] = __builtin_amdgcn_s_memrealtime(); tr[6] = mp.waited; tr[7] = mp.polls;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host: the programs
// ---------------------------------------------------------------------------------------------------------------------------------
static void march_prog_free(MarchProg &p) {
  dev_free(p.round_ptr); dev_free(p.rstart); dev_free(p.val); dev_free(p.col); dev_free(p.src);
  p.nrounds = 0;
  p.h_round_ptr.clear(); p.h_rstart.clear();
}
static void march_free(MarchDev &m) {
  march_prog_free(m.F); march_prog_free(m.B);
  dev_free(m.zf);
  m.ok = false;
}

// Rounds of one direction: per row its round (local to the chunk) and its position in the round; per chunk the round sizes.
struct MarchRounds {
  std::vector<int32_t> rho, pos;        // per row
  std::vector<int32_t> round_ptr;       // per chunk (nchunks + 1), global round index
  std::vector<int32_t> rstart;          // per round (+1), march position
  int32_t max_rows = 0;
};

// ents(i, out): the ordered neighbour list of row i for this direction (0-based rows).
template <class Ents>
static void march_rounds(int32_t N, int32_t S, int32_t R, bool fwd, Ents ents, MarchRounds &out) {
  const int32_t nch = (N + S - 1) / S;
  out.rho.assign((size_t)N, 0); out.pos.assign((size_t)N, 0);
  std::vector<std::vector<int32_t>> sizes((size_t)nch);
  parallel_for(nch, [&](int64_t c0, int64_t c1) {
    std::vector<int32_t> lev, cnt, first, nbr;
    for (int64_t c = c0; c < c1; c++) {
      const int32_t a = (int32_t)(c * S), b = (int32_t)std::min<int64_t>(N, (c + 1) * (int64_t)S), n = b - a;
      lev.assign((size_t)n, 0);
      int32_t nlev = 0;
      for (int32_t k = 0; k < n; k++) {
        const int32_t i = fwd ? a + k : b - 1 - k;
        nbr.clear();
        ents(i, nbr);
        int32_t l = 0;
        for (int32_t j : nbr)
          if (j >= a && j < b) l = std::max(l, lev[j - a]);
        lev[i - a] = l + 1;
        nlev = std::max(nlev, l + 1);
      }
      cnt.assign((size_t)nlev + 1, 0);
      for (int32_t k = 0; k < n; k++) cnt[lev[k]]++;
      // rounds: level by level, a level of more than R rows in equal parts
      first.assign((size_t)nlev + 2, 0);  // first round of each level
      std::vector<int32_t> &sz = sizes[c];
      sz.clear();
      for (int32_t l = 1; l <= nlev; l++) {
        first[l] = (int32_t)sz.size();
        const int32_t parts = (cnt[l] + R - 1) / R, per = (cnt[l] + parts - 1) / parts;
        for (int32_t p = 0, left = cnt[l]; p < parts; p++, left -= per) sz.push_back(std::min(per, left));
      }
      std::vector<int32_t> fillr((size_t)nlev + 1, 0);  // rows of the level placed so far
      for (int32_t k = 0; k < n; k++) {  // natural order inside a level
        const int32_t l = lev[k], parts = (cnt[l] + R - 1) / R, per = (cnt[l] + parts - 1) / parts, at = fillr[l]++;
        out.rho[a + k] = first[l] + at / per;
        out.pos[a + k] = at % per;
      }
    }
  });
  out.round_ptr.assign((size_t)nch + 1, 0);
  for (int32_t c = 0; c < nch; c++) out.round_ptr[c + 1] = out.round_ptr[c] + (int32_t)sizes[c].size();
  out.rstart.assign((size_t)out.round_ptr[nch] + 1, 0);
  out.max_rows = 0;
  int32_t at = 0;
  for (int32_t c = 0; c < nch; c++)
    for (size_t k = 0; k < sizes[c].size(); k++) {
      out.rstart[(size_t)out.round_ptr[c] + k] = at;
      at += sizes[c][k];
      out.max_rows = std::max(out.max_rows, sizes[c][k]);
    }
  out.rstart[(size_t)out.round_ptr[nch]] = at;
}

// Cost model of one half sweep (forward program): chunk c's round waits for its own previous round and for the rounds that publish
// its far entries (+ hop); G workgroups take the chunks round-robin.  Microseconds.
template <class Ents>
static double march_estimate(int32_t N, int32_t S, int32_t NW, int32_t G, Ents ents, const MarchRounds &mr) {
  const int32_t nch = (N + S - 1) / S;
  // measured at 10.1 M DOF (profiles/r04_march_10m_ilu0.txt): a free-running round 0.84 us with 2 pair waves, 0.96 with 3; a hand-over between
  // chunks 5 us when nothing else waits, ~9 us deep in the mesh (a round that waits costs a memory round trip, and its consumers wait too)
  const double t_round = 0.60 + 0.12 * NW, hop = 7.0;
  std::vector<double> fin((size_t)mr.round_ptr[nch], 0.0), chunk_end((size_t)nch, 0.0), need;
  std::vector<int32_t> nbr;
  double total = 0.0;
  for (int32_t c = 0; c < nch; c++) {
    const int32_t a = c * S, b = (int32_t)std::min<int64_t>(N, (int64_t)(c + 1) * S), r0 = mr.round_ptr[c], nr = mr.round_ptr[c + 1] - r0;
    need.assign((size_t)nr, 0.0);
    for (int32_t i = a; i < b; i++) {
      nbr.clear();
      ents(i, nbr);
      double w = 0.0;
      for (int32_t j : nbr)
        if (j < a) w = std::max(w, fin[(size_t)mr.round_ptr[j / S] + mr.rho[j]] + hop);
      need[mr.rho[i]] = std::max(need[mr.rho[i]], w);
    }
    double tcur = c >= G ? chunk_end[c - G] : 0.0;
    for (int32_t k = 0; k < nr; k++) {
      tcur = std::max(tcur, need[k]) + t_round;
      fin[(size_t)r0 + k] = tcur;
    }
    chunk_end[c] = tcur;
    total = std::max(total, tcur);
  }
  return total;
}

// Lanes of one direction: column codes and source codes in march order.
template <class Ents>
static void march_lanes(int32_t N, int32_t S, int32_t R, Ents ents /* (i, cols, srcs) */, const MarchRounds &mr, std::vector<int32_t> &col,
                        std::vector<int32_t> &src, int64_t &near_blocks, int64_t &far_blocks, int32_t &far_same) {
  const int32_t nch = (N + S - 1) / S;
  col.assign((size_t)16 * N, -1);
  src.assign((size_t)16 * N, -1);
  std::vector<int64_t> nearc((size_t)nch, 0), farc((size_t)nch, 0), samec((size_t)nch, 0);
  parallel_for(nch, [&](int64_t c0, int64_t c1) {
    std::vector<int32_t> nbr, sc;
    for (int64_t c = c0; c < c1; c++) {
      const int32_t a = (int32_t)(c * S), b = (int32_t)std::min<int64_t>(N, (c + 1) * (int64_t)S);
      for (int32_t i = a; i < b; i++) {
        const size_t m = (size_t)mr.rstart[(size_t)mr.round_ptr[c] + mr.rho[i]] + mr.pos[i];
        int32_t *cl = &col[16 * m], *sl = &src[16 * m];
        nbr.clear(); sc.clear();
        ents(i, nbr, sc);
        for (size_t k = 0; k < nbr.size(); k++) {
          const int32_t j = nbr[k];
          int32_t code = j;
          if (j >= a && j < b && mr.rho[i] - mr.rho[j] >= 1 && mr.rho[i] - mr.rho[j] <= FX_MARCH_NEAR) {
            code = -(((mr.rho[j] & (FX_MARCH_RING - 1)) * R + mr.pos[j]) + 2);
            nearc[c]++;
          } else {
            farc[c]++;
            if (j >= a && j < b) samec[c]++;
          }
          cl[k] = code;  // lane k / 2, half k % 2: int2 per lane = two consecutive ints
          sl[k] = sc[k];
        }
        const size_t gr = (size_t)mr.round_ptr[c] + mr.rho[i];  // the finishing lane: its row, and the row count of round + 4 (0: past the chunk's end)
        cl[14] = i; cl[15] = gr + 4 < (size_t)mr.round_ptr[c + 1] ? mr.rstart[gr + 5] - mr.rstart[gr + 4] : 0;
        sl[14] = 3 * i; sl[15] = -1;
      }
    }
  });
  near_blocks = far_blocks = 0;
  int64_t same = 0;
  for (int32_t c = 0; c < nch; c++) { near_blocks += nearc[c]; far_blocks += farc[c]; same += samec[c]; }
  far_same = (int32_t)std::min<int64_t>(same, INT32_MAX);
}

// Self-check of a program (small systems, or FX_MARCH_CHECK=1): replay the rounds of every chunk on the host with the ring holding row
// numbers instead of values, and hold every column code against the neighbour list it was built from -- a near code must find exactly
// that neighbour in its ring slot, a far code must name a row that an earlier chunk (of the sweep's order), or an earlier round of
// this chunk, produces; the row counts chained through the finishing lanes must reproduce the round table.
template <class Ents>
static bool march_check(int32_t N, int32_t S, int32_t R, bool fwd, Ents ents, const MarchRounds &mr, const std::vector<int32_t> &col,
                        std::string &why) {
  const int32_t nch = (N + S - 1) / S;
  std::vector<int32_t> ringrow((size_t)FX_MARCH_RING * R), nbr, sc;
  std::vector<int32_t> mrow((size_t)N, -1);
  for (int32_t i = 0; i < N; i++) {
    const size_t m = (size_t)mr.rstart[(size_t)mr.round_ptr[i / S] + mr.rho[i]] + mr.pos[i];
    if (m >= (size_t)N || mrow[m] != -1) { why = "march position of a row out of range or taken twice"; return false; }
    mrow[m] = i;
  }
  for (int32_t c = 0; c < nch; c++) {
    const int32_t r0 = mr.round_ptr[c], r1 = mr.round_ptr[c + 1];
    std::fill(ringrow.begin(), ringrow.end(), -1);
    int32_t rsM = mr.rstart[std::min(r0 + 3, r1 - 1)], rsMe = mr.rstart[std::min(r0 + 3, r1 - 1) + 1];
    for (int32_t r = r0; r < r1; r++) {
      const int32_t rs = mr.rstart[r], n = mr.rstart[r + 1] - rs;
      if (n < 1 || n > R) { why = "round with no rows or more than R"; return false; }
      for (int32_t qq = 0; qq < n; qq++) {
        const int32_t i = mrow[(size_t)rs + qq];
        const int32_t *cl = &col[(size_t)16 * (rs + qq)];
        if (i < 0 || i / S != c || mr.rho[i] != r - r0 || mr.pos[i] != qq || cl[14] != i) { why = "row / round table mismatch"; return false; }
        nbr.clear(); sc.clear();
        ents(i, nbr, sc);
        if (nbr.size() > FX_MARCH_MAXBLOCKS) { why = "row with more than 14 blocks"; return false; }
        for (size_t k = 0; k < 14; k++) {
          const int32_t code = cl[k];
          if (k >= nbr.size()) { if (code != -1) { why = "padding lane with a column"; return false; } continue; }
          const int32_t j = nbr[k];
          if (code <= -2) {
            if (ringrow[(size_t)(-code - 2)] != j) { why = "near code does not find its neighbour in the ring"; return false; }
          } else if (code == j) {
            const int32_t cj = j / S;
            const bool earlier = fwd ? (cj < c) : (cj > c);
            if (!(earlier || (cj == c && mr.rho[j] < mr.rho[i]))) { why = "far code names a row that is not produced before"; return false; }
          } else { why = "column code is neither its neighbour nor a ring slot"; return false; }
        }
        const int32_t want = r + 4 < r1 ? mr.rstart[r + 5] - mr.rstart[r + 4] : 0;
        if (cl[15] != want) { why = "row count of round + 4 in the finishing lane"; return false; }
      }
      for (int32_t qq = 0; qq < n; qq++) ringrow[(size_t)((r - r0) & (FX_MARCH_RING - 1)) * R + qq] = mrow[(size_t)rs + qq];
      // the descriptor chain of the kernel: after round r, [rsM, rsMe) must be round min(r + 4, last)
      const int32_t nn = col[(size_t)16 * rs + 15];
      if (nn > 0) { rsM = rsMe; rsMe += nn; }
      const int32_t tgt = std::min(r + 4, r1 - 1);
      if (rsM != mr.rstart[tgt] || rsMe != mr.rstart[tgt + 1]) { why = "descriptor chain leaves the round table"; return false; }
    }
  }
  return true;
}

static int march_upload(fx_context *c, MarchProg &p, const MarchRounds &mr, const std::vector<int32_t> &col, const std::vector<int32_t> &src) {
  const size_t nrows = (size_t)mr.rstart.back();
  p.nrounds = (int64_t)mr.rstart.size() - 1;
  p.h_round_ptr = mr.round_ptr; p.h_rstart = mr.rstart;
  if (dev_alloc(&p.round_ptr, mr.round_ptr.size()) || dev_alloc(&p.rstart, mr.rstart.size()) || dev_alloc(&p.col, (size_t)16 * nrows) ||
      dev_alloc(&p.src, (size_t)16 * nrows) || dev_alloc(&p.val, (size_t)144 * nrows))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpyAsync(p.round_ptr, mr.round_ptr.data(), mr.round_ptr.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(p.rstart, mr.rstart.data(), mr.rstart.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(p.col, col.data(), col.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(p.src, src.data(), src.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

static inline int march_nw_index(int nw) { return nw <= 1 ? 0 : nw == 2 ? 1 : 2; }

// Build the two programs for the level-scheduled preconditioner of this context (after ilu_setup_symbolic).  Leaves march.ok false
// when the structure does not qualify (a row with more than 14 lower or upper blocks) or the cost model prefers k_tri_dataflow.
static int march_build(fx_context *c) {
  SsorDev &Sd = c->ssor;
  MarchDev &M = Sd.march;
  march_free(M);
  if (c->march_mode <= 0 || c->df_mode < 1) return 0;
  const double t_begin = now_s();
  const int32_t N = c->A.N;
  const int32_t *iL = c->h_indexL.data(), *jL = c->h_itemL.data(), *iU = c->h_indexU.data(), *jU = c->h_itemU.data();
  if (N < 64) return 0;
  int32_t maxl = 0, maxu = 0;
  for (int32_t i = 0; i < N; i++) {
    maxl = std::max(maxl, iL[i + 1] - iL[i]);
    int32_t k = 0;
    for (int32_t j = iU[i]; j < iU[i + 1]; j++) k += (jU[j] <= N);
    maxu = std::max(maxu, k);
  }
  if (maxl > FX_MARCH_MAXBLOCKS || maxu > FX_MARCH_MAXBLOCKS) return 0;
  auto entsL = [=](int32_t i, std::vector<int32_t> &out) { for (int32_t j = iL[i]; j < iL[i + 1]; j++) out.push_back(jL[j] - 1); };
  auto entsU = [=](int32_t i, std::vector<int32_t> &out) { for (int32_t j = iU[i + 1] - 1; j >= iU[i]; j--) if (jU[j] <= N) out.push_back(jU[j] - 1); };
  auto entsL2 = [=](int32_t i, std::vector<int32_t> &out, std::vector<int32_t> &sc) {  // ascending columns (BILU_33 :104-111, SSOR_33 :312)
    for (int32_t j = iL[i]; j < iL[i + 1]; j++) { out.push_back(jL[j] - 1); sc.push_back(3 * j + 1); }
  };
  auto entsU2 = [=](int32_t i, std::vector<int32_t> &out, std::vector<int32_t> &sc) {  // descending columns (BILU_33 :133, SSOR_33 :369), halo columns dropped
    for (int32_t j = iU[i + 1] - 1; j >= iU[i]; j--) if (jU[j] <= N) { out.push_back(jU[j] - 1); sc.push_back(3 * j + 2); }
  };
  // chunk size: the offset of the far lower neighbours ("one plane back" on a structured mesh) -- the weighted median of the offsets
  // larger than half the largest one, over a sample of the rows
  int32_t plane = 0;
  {
    int32_t bw = 0;
    for (int32_t i = 0; i < N; i++) if (iL[i + 1] > iL[i]) bw = std::max(bw, i - (jL[iL[i]] - 1));
    std::vector<int32_t> offs;
    const int32_t step = std::max(1, N / 200000);
    for (int32_t i = 0; i < N; i += step)
      for (int32_t j = iL[i]; j < iL[i + 1]; j++) { const int32_t o = i - (jL[j] - 1); if (2 * o > bw) offs.push_back(o); }
    if (!offs.empty()) { std::nth_element(offs.begin(), offs.begin() + offs.size() / 2, offs.end()); plane = offs[offs.size() / 2]; }
  }
  const int idx_of[3] = {1, 2, 3};
  struct Cand { int32_t S, NW; double est; };
  std::vector<Cand> cands;
  if (c->march_chunk > 0) cands.push_back({std::min(N, c->march_chunk), c->march_waves, 0.0});
  else if (plane >= 64)
    for (int div : {1, 2, 3, 4, 6, 8}) cands.push_back({std::max(64, (plane + div - 1) / div), c->march_waves, 0.0});
  else cands.push_back({std::max(64, N / 128), c->march_waves, 0.0});
  MarchRounds best;
  Cand pick{0, 0, 1e300};
  for (Cand cd : cands) {
    // pair waves: enough for the chunk's typical level (levels are then rarely split); fixed by FX_MARCH_WAVES
    MarchRounds probe;
    int32_t nw = cd.NW;
    if (nw != 1 && nw != 2 && nw != 3) {
      march_rounds(N, cd.S, 1 << 20, true, entsL, probe);  // unsplit levels
      std::vector<int32_t> sz;
      for (size_t k = 0; k + 1 < probe.rstart.size(); k++) sz.push_back(probe.rstart[k + 1] - probe.rstart[k]);
      std::sort(sz.begin(), sz.end());
      const int32_t p90 = sz.empty() ? 8 : sz[(size_t)(0.9 * (sz.size() - 1))];
      nw = 3;
      for (int k = 0; k < 3; k++) if (8 * idx_of[k] >= p90) { nw = idx_of[k]; break; }
    }
    const int32_t nch = (N + cd.S - 1) / cd.S;
    const int gmax = std::max(8, c->march_grid_max[march_nw_index(nw)] / 8 * 8);
    MarchRounds mr;
    march_rounds(N, cd.S, 8 * nw, true, entsL, mr);
    const double est = march_estimate(N, cd.S, nw, std::min(gmax, (nch + 7) / 8 * 8), entsL, mr);
    if (est < pick.est) { pick = {cd.S, nw, est}; best = std::move(mr); }
  }
  M.S = pick.S; M.NW = pick.NW; M.nchunks = (N + pick.S - 1) / pick.S;
  M.est_us = pick.est;
  M.est_level_us = 1.6 * Sd.ncolor;  // k_tri_dataflow: 1.6 us per dependency level and half sweep (3.33 ms per apply at 1,044 levels)
  if (c->march_mode == 1 && !(M.est_us < 0.8 * M.est_level_us)) { M.build_s = now_s() - t_begin; return 0; }
  const int32_t R = 8 * M.NW;
  const bool check = N <= 200000 || getenv("FX_MARCH_CHECK") != nullptr;
  std::string why;
  {
    std::vector<int32_t> col, src;
    march_lanes(N, M.S, R, entsL2, best, col, src, M.near_blocks, M.far_blocks, M.far_same_chunk);
    if (check && !march_check(N, M.S, R, true, entsL2, best, col, why)) { g_fx_error = "plane march, forward program: " + why; return FX_ERROR_RUNTIME; }
    if (march_upload(c, M.F, best, col, src)) return FX_ERROR_RUNTIME;
    M.max_round_rows = best.max_rows;
  }
  {
    MarchRounds mb;
    march_rounds(N, M.S, R, false, entsU, mb);
    std::vector<int32_t> col, src;
    int64_t nb = 0, fb = 0;
    int32_t fs = 0;
    march_lanes(N, M.S, R, entsU2, mb, col, src, nb, fb, fs);
    if (check && !march_check(N, M.S, R, false, entsU2, mb, col, why)) { g_fx_error = "plane march, backward program: " + why; return FX_ERROR_RUNTIME; }
    if (march_upload(c, M.B, mb, col, src)) return FX_ERROR_RUNTIME;
    M.max_round_rows = std::max(M.max_round_rows, mb.max_rows);
  }
  if (dev_alloc(&M.zf, (size_t)3 * N + 4)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemset(M.zf, 0, ((size_t)3 * N + 4) * 8));  // the tail is the dummy entry
  M.ok = true;
  M.build_s = now_s() - t_begin;
  return 0;
}

// values: AL / AU = the factor arrays for ILU(0), the matrix itself for the natural-order SSOR
static int march_fill_values(fx_context *c, const double *AL, const double *AU, double sigma_diag) {
  MarchDev &M = c->ssor.march;
  if (!M.ok) return 0;
  for (MarchProg *p : {&M.F, &M.B}) {
    if (p->nrounds == 0) continue;
    hipLaunchKernelGGL(k_march_fill, dim3((unsigned)p->nrounds), dim3(256), 0, c->stream, p->rstart, (const int2 *)p->src, c->A.D, AL, AU,
                       sigma_diag, (double2 *)p->val);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int march_apply(fx_context *c, const double *r, double *z, const int32_t *gate, unsigned long long *trace = nullptr,
                       unsigned long long *rtrace = nullptr, int rtrace_chunk = -1) {
  MarchDev &M = c->ssor.march;
  const int32_t N = c->A.N;
  hipLaunchKernelGGL(k_march_tags, dim3(grid_for((int64_t)3 * N, 256, 2048)), dim3(256), 0, c->stream, (int64_t)3 * N, M.zf, z);
  MarchArgs a;
  a.F = {M.F.round_ptr, M.F.rstart, (const double2 *)M.F.val, (const int2 *)M.F.col};
  a.B = {M.B.round_ptr, M.B.rstart, (const double2 *)M.B.val, (const int2 *)M.B.col};
  a.nchunks = M.nchunks;
  a.r = r; a.zf = M.zf; a.z = z; a.dummy = M.zf + (size_t)3 * N;
  a.err = c->df_err;
  a.nsleep = c->dbg_df_fail ? -1 : c->df_sleep;
  a.xcd = c->march_xcd ? 1 : 0;
  a.trace = trace; a.rtrace = rtrace; a.rtrace_chunk = rtrace_chunk;
  const int gmax = std::max(8, c->march_grid_max[march_nw_index(M.NW)] / 8 * 8);
  int grid = std::min(gmax, (M.nchunks + 7) / 8 * 8);
  if (c->march_grid > 0) grid = std::max(8, std::min(grid, c->march_grid / 8 * 8));
  c->df_grid_last = grid;
  switch (M.NW) {
    case 1: hipLaunchKernelGGL((k_tri_march<1>), dim3(grid), dim3(64 * 2), 0, c->stream, a, gate); break;
    case 2: hipLaunchKernelGGL((k_tri_march<2>), dim3(grid), dim3(64 * 3), 0, c->stream, a, gate); break;
    default: hipLaunchKernelGGL((k_tri_march<3>), dim3(grid), dim3(64 * 4), 0, c->stream, a, gate); break;
  }
  HIP_TRY(hipGetLastError());
  c->march_launches++;
  return 0;
}

// Host-only: plan the two march programs for a block profile (1-based items as hecMAT's, N internal rows) with `chunk` rows per chunk and
// `waves` pair waves (1, 2, 3), replay them with march_check and report.  No device is touched: the CPU test suite runs the schedule builder
// through this on structured and unstructured profiles.  out[0] 1 = both programs pass their replay, 0 = the structure is not admitted (a row
// with more than 14 lower or upper blocks), [1] chunks, [2] / [3] rounds forward / backward, [4] / [5] near / far blocks of the forward
// program, [6] rows of the largest round, [7] dependency levels of the whole matrix (what the level sweeps need).
extern "C" int fx_march_plan(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU, const int32_t *itemU,
                             int32_t chunk, int32_t waves, double out[8]) {
  if (!indexL || !itemL || !indexU || !itemU || !out || N < 1 || chunk < 1 || waves < 1 || waves > 3) {
    g_fx_error = "fx_march_plan: bad argument";
    return FX_ERROR_RUNTIME;
  }
  for (int k = 0; k < 8; k++) out[k] = 0.0;
  const int32_t *iL = indexL, *jL = itemL, *iU = indexU, *jU = itemU;
  int32_t maxl = 0, maxu = 0, nlev = 0;
  {
    std::vector<int32_t> level((size_t)N, 0);
    for (int32_t i = 0; i < N; i++) {
      maxl = std::max(maxl, iL[i + 1] - iL[i]);
      int32_t k = 0, l = 0;
      for (int32_t j = iU[i]; j < iU[i + 1]; j++) k += (jU[j] <= N);
      maxu = std::max(maxu, k);
      for (int32_t j = iL[i]; j < iL[i + 1]; j++) l = std::max(l, level[jL[j] - 1]);
      level[i] = l + 1;
      nlev = std::max(nlev, l + 1);
    }
  }
  out[7] = nlev;
  if (maxl > FX_MARCH_MAXBLOCKS || maxu > FX_MARCH_MAXBLOCKS) return 0;
  auto entsL = [=](int32_t i, std::vector<int32_t> &o) { for (int32_t j = iL[i]; j < iL[i + 1]; j++) o.push_back(jL[j] - 1); };
  auto entsU = [=](int32_t i, std::vector<int32_t> &o) { for (int32_t j = iU[i + 1] - 1; j >= iU[i]; j--) if (jU[j] <= N) o.push_back(jU[j] - 1); };
  auto entsL2 = [=](int32_t i, std::vector<int32_t> &o, std::vector<int32_t> &sc) {
    for (int32_t j = iL[i]; j < iL[i + 1]; j++) { o.push_back(jL[j] - 1); sc.push_back(3 * j + 1); }
  };
  auto entsU2 = [=](int32_t i, std::vector<int32_t> &o, std::vector<int32_t> &sc) {
    for (int32_t j = iU[i + 1] - 1; j >= iU[i]; j--) if (jU[j] <= N) { o.push_back(jU[j] - 1); sc.push_back(3 * j + 2); }
  };
  const int32_t S = std::min(N, chunk), R = 8 * waves;
  std::string why;
  std::vector<int32_t> col, src;
  int64_t nb = 0, fb = 0;
  int32_t fs = 0;
  MarchRounds mf, mb;
  march_rounds(N, S, R, true, entsL, mf);
  march_lanes(N, S, R, entsL2, mf, col, src, nb, fb, fs);
  if (!march_check(N, S, R, true, entsL2, mf, col, why)) { g_fx_error = "fx_march_plan, forward program: " + why; return FX_ERROR_RUNTIME; }
  out[4] = (double)nb; out[5] = (double)fb;
  march_rounds(N, S, R, false, entsU, mb);
  march_lanes(N, S, R, entsU2, mb, col, src, nb, fb, fs);
  if (!march_check(N, S, R, false, entsU2, mb, col, why)) { g_fx_error = "fx_march_plan, backward program: " + why; return FX_ERROR_RUNTIME; }
  out[0] = 1.0; out[1] = (N + S - 1) / S; out[2] = (double)mf.rstart.size() - 1; out[3] = (double)mb.rstart.size() - 1;
  out[6] = std::max(mf.max_rows, mb.max_rows);
  return 0;
}
