// Round-2 negative result, removed from the product in round 3 ('Chain sweeps'): kept for reference only, not built.
// ------------------------------------------------------------------------
// Chain sweeps of ILU(0) (FX_DATAFLOW=3): a wave walks a SEGMENT of consecutive rows in the reference's own sequential
// order, one row per step, and the whole wave works on that row -- a lane holds row q of the row's b-th off-diagonal
// block (lane 16 q + b), so the 13 blocks of a hex-mesh row are one fused-multiply-add deep and a four-step butterfly wide.  The level
// kernels above hand every dependency through memory (2,088 levels x ~2.3 us at 150^3 nodes).  In natural order most of
// a row's critical dependencies are the rows just before it (i - 1 on a structured mesh): inside a segment they come
// from LDS, 0.2 us per row; only dependencies on OTHER segments cross memory, with the same self-tagged hand-off as
// k_tri_dataflow (sentinel-filled vectors, 8-byte sc1 stores and loads).  Those loads are issued W rows ahead
// (software pipeline in registers) together with the row's blocks -- read straight from the factor arrays in CRS order, no
// BELL copy -- so a consumer trails its producers by about one memory round trip and never stops to wait once the
// pipeline is primed: on a structured hex mesh the critical path is the ~450 pencil-to-pencil hand-offs instead of 1,044
// level-to-level ones.  Segments are dealt to the waves in the order of a start-time estimate made at set-up
// (fistr_hip.hip: chain_schedule), which only references earlier entries of that order: with all workgroups resident the
// first unfinished segment can always run.  Every spin is bounded by wall-clock time like k_tri_dataflow's.
//   forward : zf_i = D~_i^-1 (r_i - sum_{j in L(i)} L_ij zf_j)           (hecmw_precond_BILU_33.f90:100-124)
//   backward: zb_i = zf_i - D~_i^-1 sum_{j in U(i)} U_ij zb_j ; z_i = zb_i (:127-153; halo columns dropped)
// The sum over a row's blocks is a butterfly instead of the sequential loop: rounding differs in the last bits from the
// level kernels (tests compare within the tolerance of the ILU parity tests, not bitwise).
// ------------------------------------------------------------------------
#define FX_CH_SEG 64   // rows per segment
#define FX_CH_MAXB 16  // off-diagonal blocks per row and sweep direction the lane mapping holds

struct ChainStage {  // what lane 16 q + b holds for one row in flight
  double a0, a1, a2;  // row q of the row's b-th block
  double x0, x1, x2;  // the vector entries of that block's column
  double f;           // backward sweep: component min(q, 2) of the row's forward value
  int32_t col;        // 0-based column, -1: no block
};

__device__ __forceinline__ bool df_is_tag(double v) { return __double_as_longlong(v) == FX_DF_SENTINEL; }

// v + (v rotated by N lanes inside each row of 16 lanes): VALU data-parallel primitives, no trip through the LDS crossbar
template <int N>
__device__ __forceinline__ double row_ror_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xF, 0xF, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {  // wave-uniform copy of one lane's value
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Issue the loads of one row (a pipeline stage) from its table entry e = {index of the block in the factor array, column or
// -1}.  Branch-free on purpose: every lane always issues the same loads (idle lanes read an address a busy lane reads anyway),
// so the compiler can count the loads in flight and the step that consumes a stage waits for exactly that stage -- with
// conditional loads it drains the whole pipeline at every step.
template <bool FWD>
__device__ __forceinline__ void chain_issue(ChainStage &st, fx_i2 e, int row, const double *__restrict__ blocks,
                                            const double *__restrict__ zsrc, const double *__restrict__ zf, int qq) {
  const double *blk = blocks + ((size_t)9 * (uint32_t)e.x + 3 * qq);
  const double *xp = zsrc + (size_t)3 * (uint32_t)max(e.y, 0);
  st.a0 = ld_stream(blk); st.a1 = ld_stream(blk + 1); st.a2 = ld_stream(blk + 2);
  st.x0 = df_load(xp); st.x1 = df_load(xp + 1); st.x2 = df_load(xp + 2);
  st.f = FWD ? 0.0 : df_load(zf + (size_t)3 * row + qq);
  st.col = e.y;
}

// bounded wait until the outside entries of a stage (and, backward, the row's forward value) are published
template <bool FWD>
__device__ __forceinline__ bool chain_wait(ChainStage &S, int i, int a, int len, const double *__restrict__ zsrc,
                                           const double *__restrict__ zf, int q, int32_t *__restrict__ err, bool &dead, int nsleep,
                                           bool outside) {
  bool miss = (outside && (df_is_tag(S.x0) || df_is_tag(S.x1) || df_is_tag(S.x2))) || (!FWD && i >= 0 && df_is_tag(S.f));
  if (!__any(miss) || dead) return false;
  unsigned long long t0 = 0;
  for (unsigned spins = 1;; spins++) {
    for (int k = 0; k < nsleep; k++) __builtin_amdgcn_s_sleep(1);
    if (miss) {
      if (outside) {
        const double *xp = zsrc + (size_t)3 * S.col;
        S.x0 = df_load(xp); S.x1 = df_load(xp + 1); S.x2 = df_load(xp + 2);
      }
      if (!FWD && i >= 0) S.f = df_load(zf + (size_t)3 * i + (q < 3 ? q : 2));
      miss = (outside && (df_is_tag(S.x0) || df_is_tag(S.x1) || df_is_tag(S.x2))) || (!FWD && i >= 0 && df_is_tag(S.f));
    }
    if (!__any(miss)) return true;
    if ((spins & 255u) == 0u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || now - t0 > FX_DF_TIMEOUT_TICKS) {
        if (e == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        return true;
      }
    }
  }
}

// One chunk (<= FX_CH_SEG rows) of a chain.  Four waves, four roles -- a single wave issues one instruction every four clocks,
// and a row done by ONE wave (operand loads, tag checks, products, reduction, substitution: ~320 instructions) took 0.55 us:
//   wave 0      the CHAIN: x_i = D~_i^-1 (r_i - y'_i - A_link x_prev), nothing else; x_prev stays in registers
//   waves 1, 2  OPERANDS, alternate rows, running ahead of the chain: loads W of their rows ahead, waits for outside
//               operands, multiplies and reduces everything except the block on the row just before (the link), leaves
//               y'_i and the link block in LDS, raises ready[i]
//   wave 3      PUBLISHER: writes finished rows out with the write-through stores of the hand-off (loads and stores of one
//               wave retire in issue order; behind a write-through store every later load would count as outstanding)
struct ChainLds {
  int32_t ip[FX_CH_SEG + 1];
  fx_i2 tab[FX_CH_SEG * FX_CH_MAXB];   // [row][b] = {index of the block in the factor array (always a valid one), column or -1}
  double us[9 * FX_CH_SEG], rs[3 * FX_CH_SEG];
  double xs[3 * FX_CH_SEG];            // finished rows of the chunk
  double yp[3 * FX_CH_SEG];            // y' per step
  double al[9 * FX_CH_SEG];            // link block per step (zeros: none)
  double fs[3 * FX_CH_SEG];            // backward sweep: the row's forward value per step
  int32_t ready[FX_CH_SEG];
  int32_t ndone;
};

template <bool FWD, int W>
__device__ __forceinline__ void chain_segment(int a, int len, int ulo, int uhi, int32_t N, const int32_t *__restrict__ index,
                                              const int32_t *__restrict__ item,
                                              const double *__restrict__ blocks, const double *__restrict__ luD,
                                              const double *__restrict__ r, double *__restrict__ zf, double *__restrict__ zb,
                                              double *__restrict__ z, double *__restrict__ partials, int32_t *__restrict__ err,
                                              bool &dead, int ahead, ChainLds &L, double &p0, double &p1, double &p2, double &dsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, b = lane & 15, qq = q < 3 ? q : 2;
  double *zdst = FWD ? zf : zb;
  const double *zsrc = zdst;
  volatile int32_t *ready = L.ready, *ndone = &L.ndone;
  // data arrays are plain (the compiler may batch their accesses); the flags are volatile and fenced from the data by compiler
  // barriers -- the LDS operations of a wave execute in issue order, so that is all the ordering the hardware needs
  double *xs = L.xs, *yp = L.yp, *al = L.al, *fs = L.fs;
#define CH_ORDER() asm volatile("" ::: "memory")
  // the chunk's row pointers, column ids, diagonal factors and right-hand side: coalesced sweeps into LDS
  for (int t = threadIdx.x; t <= len; t += 256) L.ip[t] = index[a + t];
  if (threadIdx.x < FX_CH_SEG) L.ready[threadIdx.x] = 0;
  if (threadIdx.x == 0) L.ndone = 0;
  __syncthreads();
  const int32_t jbase = L.ip[0], nj = min(L.ip[len] - jbase, FX_CH_SEG * FX_CH_MAXB);
  int32_t *jc = (int32_t *)L.tab + FX_CH_SEG * FX_CH_MAXB;  // the column ids land in the upper half of the table's storage first
  for (int k = threadIdx.x; k < nj; k += 256) jc[k] = item[jbase + k];  // independent coalesced loads: one latency, not one per row
  for (int k = threadIdx.x; k < 9 * len; k += 256) L.us[k] = luD[(size_t)9 * a + k];
  if (FWD || partials)
    for (int k = threadIdx.x; k < 3 * len; k += 256) L.rs[k] = r[(size_t)3 * a + k];
  __syncthreads();
  {  // the table, built from LDS; the entries stay in registers until every thread has read the ids they overwrite
    fx_i2 e[FX_CH_SEG * FX_CH_MAXB / 256];
#pragma unroll
    for (int m = 0; m < FX_CH_SEG * FX_CH_MAXB / 256; m++) {
      const int k = threadIdx.x + 256 * m;
      const int t = min(k >> 4, len - 1), bb = k & 15;
      const int32_t j0 = L.ip[t], n = L.ip[t + 1] - j0;
      const bool has = (k >> 4) < len && bb < n;
      e[m].x = has ? j0 + bb : max(min(jbase, L.ip[len] - 1), 0);  // idle lanes re-read a block of the chunk (or block 0 when it has none)
      int32_t c = has ? jc[min(j0 + bb - jbase, FX_CH_SEG * FX_CH_MAXB - 1)] - 1 : -1;
      if (c >= N) c = -1;  // upper part: halo columns (ids > N, last) are not part of the localized preconditioner
      e[m].y = c;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < FX_CH_SEG * FX_CH_MAXB / 256; m++) L.tab[threadIdx.x + 256 * m] = e[m];
  }
  __syncthreads();
  // rows are visited a .. a+len-1 (forward) or a+len-1 .. a (backward); `step` counts visits
  auto row_of = [&](int step) { return FWD ? a + step : a + len - 1 - step; };
  if (wave == 0) {
    // ---- the chain ----
    for (int step = 0; step < len; step++) {
      const int t = row_of(step) - a;
      if (!dead) {
        unsigned long long t0 = 0;
        for (unsigned spins = 1; ready[step] == 0; spins++) {  // the operand waves run ahead: normally set long ago
          if ((spins & 1023u) == 0u) {
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
      }
      CH_ORDER();
      double y0 = yp[3 * step], y1 = yp[3 * step + 1], y2 = yp[3 * step + 2];
      double A[9], u[9];
#pragma unroll
      for (int k = 0; k < 9; k++) { A[k] = al[9 * step + k]; u[k] = L.us[9 * t + k]; }
      y0 = fma(A[2], p2, fma(A[1], p1, fma(A[0], p0, y0)));
      y1 = fma(A[5], p2, fma(A[4], p1, fma(A[3], p0, y1)));
      y2 = fma(A[8], p2, fma(A[7], p1, fma(A[6], p0, y2)));
      double x0, x1, x2;
      if (FWD) {
        x0 = L.rs[3 * t] - y0; x1 = L.rs[3 * t + 1] - y1; x2 = L.rs[3 * t + 2] - y2;
        lusolve33_dev(u, x0, x1, x2);
      } else {
        lusolve33_dev(u, y0, y1, y2);
        x0 = fs[3 * step] - y0; x1 = fs[3 * step + 1] - y1; x2 = fs[3 * step + 2] - y2;
      }
      p0 = x0; p1 = x1; p2 = x2;
      if (lane < 3) {
        const double v = lane == 0 ? x0 : (lane == 1 ? x1 : x2);
        xs[3 * t + lane] = v;
        if (!FWD && partials) dsum += L.rs[3 * t + lane] * v;
      }
      CH_ORDER();
      if (lane == 0) *ndone = step + 1;  // after the row's entries: LDS operations of a wave execute in order
    }
    __syncthreads();
    return;
  }
  if (wave == 3) {
    // ---- the publisher ----
    int pub = 0;
    while (pub < len) {
      const int nd = *ndone;
      CH_ORDER();
      if (nd > pub) {
        for (int k = lane; k < 3 * (nd - pub); k += 64) {
          const int i = row_of(pub + k / 3), cmp = k % 3;
          const double v = xs[3 * (i - a) + cmp];
          df_store(zdst + (size_t)3 * i + cmp, v);
          if (!FWD) z[(size_t)3 * i + cmp] = v;
        }
        pub = nd;
      } else __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    return;
  }
  // ---- operand waves: wave 1 takes the even steps, wave 2 the odd ones ----
  const int first = wave - 1;
  auto entry = [&](int step) {  // lanes of the idle quarter (q = 3) and of blocks a row does not have carry column -1
    fx_i2 e = L.tab[(row_of(min(step, len - 1)) - a) * FX_CH_MAXB + b];
    if (q == 3 || step >= len) e.y = -1;
    return e;
  };
  // the row the chain wave has in registers: the one before in the walk, if it belongs to this unit (a chain cut at the length
  // cap leaves the first row of the next unit depending on the last row of this one -- through memory, like any other unit's)
  const auto link_of = [&](int step) {
    const int l = FWD ? row_of(step) - 1 : row_of(step) + 1;
    return (l >= ulo && l < uhi) ? l : -2;
  };
  const auto from_memory = [&](const ChainStage &S, int step) {
    return S.col >= 0 && !(FWD ? S.col >= a : S.col < a + len) && S.col != link_of(step);
  };
  ChainStage st[W];
  // Start only when the row `ahead` steps in has its outside operands: the producers run at this chain's pace, so once the
  // chain trails them by the pipeline depth plus a round trip the loads issued ahead find published values and no step
  // waits.  (Starting at once makes EVERY step wait a round trip until the same lag has built up the slow way.)
  ChainStage probe;
  const int ka = min(len - 1, ahead);
  if (ahead > 0) chain_issue<FWD>(probe, entry(ka), row_of(ka), blocks, zsrc, zf, qq);
#pragma unroll
  for (int s = 0; s < W; s++) chain_issue<FWD>(st[s], entry(first + 2 * s), row_of(min(first + 2 * s, len - 1)), blocks, zsrc, zf, qq);
  if (ahead > 0 && chain_wait<FWD>(probe, -1, a, len, zsrc, zf, q, err, dead, 4, from_memory(probe, ka))) {
    // the probe travelled with the first rows' loads (no round trip of its own when it is satisfied at once); it had to wait,
    // so what the first rows loaded is older than what it finally saw: load them again rather than poll row by row
#pragma unroll
    for (int s = 0; s < W; s++) chain_issue<FWD>(st[s], entry(first + 2 * s), row_of(min(first + 2 * s, len - 1)), blocks, zsrc, zf, qq);
  }
  for (int base = first; base < len; base += 2 * W) {
#pragma unroll
    for (int s = 0; s < W; s++) {
      const int step = base + 2 * s;
      const bool live = step < len;  // wave-uniform; a dead step (tail of the last round) computes on the last row and stores nothing
      const int i = row_of(min(step, len - 1));
      ChainStage &S = st[s];
      const bool valid = S.col >= 0;
      const bool link = valid && S.col == link_of(min(step, len - 1));
      const bool inside = valid && !link && (FWD ? S.col >= a : S.col < a + len);
      const fx_i2 enext = entry(step + 2 * W);
      if (live && __any(inside) && !dead) {  // an older row of this very chunk (never on a structured hex mesh): the chain wave must have passed it
        const int need = FWD ? S.col - a + 1 : a + len - S.col;
        unsigned long long t0 = 0;
        for (unsigned spins = 1; __any(inside && *ndone < need); spins++)
          if ((spins & 1023u) == 0u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            if (now - t0 > FX_DF_TIMEOUT_TICKS) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break; }
          }
      }
      CH_ORDER();
      if (inside) {
        const int o = 3 * (S.col - a);
        S.x0 = xs[o]; S.x1 = xs[o + 1]; S.x2 = xs[o + 2];
      }
      if (live) chain_wait<FWD>(S, i, a, len, zsrc, zf, q, err, dead, 0, valid && !inside && !link);
      double acc = (valid && !link) ? fma(S.a2, S.x2, fma(S.a1, S.x1, S.a0 * S.x0)) : 0.0;
      acc = row_ror_add<8>(acc);
      acc = row_ror_add<4>(acc);
      acc = row_ror_add<2>(acc);
      acc = row_ror_add<1>(acc);
      if (live) {
        if (b == 0 && q < 3) {
          yp[3 * step + q] = acc;
          al[9 * step + 3 * q] = 0.0; al[9 * step + 3 * q + 1] = 0.0; al[9 * step + 3 * q + 2] = 0.0;
        }
        if (link) { al[9 * step + 3 * q] = S.a0; al[9 * step + 3 * q + 1] = S.a1; al[9 * step + 3 * q + 2] = S.a2; }
        if (!FWD && b == 0 && q < 3) fs[3 * step + q] = S.f;
        CH_ORDER();
        if (lane == 0) ready[step] = 1;  // after the operands: LDS operations of a wave execute in order
      }
      __builtin_amdgcn_wave_barrier();
      chain_issue<FWD>(st[s], enext, row_of(min(step + 2 * W, len - 1)), blocks, zsrc, zf, qq);
    }
  }
  __syncthreads();  // LDS is reused by the next chunk
#undef CH_ORDER
}

template <int W>
__global__ __launch_bounds__(256) void k_tri_chain(int32_t N, int32_t nF, const int32_t *__restrict__ startF, const int32_t *__restrict__ ordF,
                                                   int32_t nB, const int32_t *__restrict__ startB, const int32_t *__restrict__ ordB,
                                                   const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                                   const double *__restrict__ luAL, const int32_t *__restrict__ indexU,
                                                   const int32_t *__restrict__ itemU, const double *__restrict__ luAU,
                                                   const double *__restrict__ luD, const double *__restrict__ r, double *__restrict__ zf,
                                                   double *__restrict__ zb, double *__restrict__ z, double *__restrict__ partials,
                                                   const int32_t *__restrict__ gate, int32_t *__restrict__ err, int ahead) {
  if (gate && *gate != 0) return;
  __shared__ ChainLds L;
  bool dead = false;
  for (int k = blockIdx.x; k < nF; k += gridDim.x) {  // a unit = a chain of consecutive rows, walked in chunks of FX_CH_SEG
    const int u = ordF[k], lo = startF[u], hi = startF[u + 1];
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, dsum = 0.0;  // the row just done (chain wave): the dependency on it never goes through memory
    for (int a = lo; a < hi; a += FX_CH_SEG)
      chain_segment<true, W>(a, min(FX_CH_SEG, hi - a), lo, hi, N, indexL, itemL, luAL, luD, r, zf, zb, z, partials, err, dead, ahead, L, p0, p1, p2,
                             dsum);
  }
  for (int k = blockIdx.x; k < nB; k += gridDim.x) {
    const int u = ordB[k], lo = startB[u], hi = startB[u + 1];
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, dsum = 0.0;
    for (int e = hi; e > lo; e -= FX_CH_SEG) {
      const int a = max(lo, e - FX_CH_SEG);
      chain_segment<false, W>(a, e - a, lo, hi, N, indexU, itemU, luAU, luD, r, zf, zb, z, partials, err, dead, ahead, L, p0, p1, p2, dsum);
    }
    if (partials && threadIdx.x < 64) {  // the chain wave holds the unit's share of r.z
      dsum = wave_sum(dsum);
      if (threadIdx.x == 0) partials[u] = dsum;
    }
  }
}

// ------------------------------------------------------------------------
// K9: block ILU(0) (hecmw_precond_BILU_33.f90).  The reference factorises and substitutes
// strictly sequentially (FORM_ILU0_33 :185-362, apply :90-157).  Here rows are grouped into
// dependency levels (level(i) = 1 + max level of the rows in L(i)); rows of one level are
// independent, so each level is one launch and every row sees exactly the operands, in exactly
// the order, the sequential loop gives it -- same factors, same iteration counts.
