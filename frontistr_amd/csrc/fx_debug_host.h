// Diagnostics of the placement effect on the SpMV (scripts/r4/placement_probe.py).  Not part of the solver
// interface: these entry points move the arrays the SpMV touches between allocations of one live context -- same data, same
// kernel, different memory -- and time the product on chosen vectors, so that the speed classes can be tied to ONE array and to
// ONE allocation strategy from evidence.  Included once by fistr_hip.hip.
#pragma once

struct DbgArena {
  char *base = nullptr;
  size_t bytes = 0;
};
static DbgArena g_dbg_arena;
static std::vector<void *> g_dbg_leak;  // earlier homes of re-placed arrays: kept so that the next request gets different memory

// hipMemCreate / hipMemMap: `bytes` of device memory as physical chunks of `chunk` bytes behind one contiguous virtual range.
static int dbg_vmm_alloc(fx_context *c, size_t bytes, size_t chunk, char **out) {
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c->device;
  size_t gran = 0;
  HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (gran == 0) gran = (size_t)2 << 20;
  chunk = std::max(gran, (chunk + gran - 1) / gran * gran);
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void *va = nullptr;
  HIP_TRY(hipMemAddressReserve(&va, total, chunk, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    HIP_TRY(hipMemCreate(&h, chunk, &prop, 0));
    HIP_TRY(hipMemMap((char *)va + off, chunk, 0, h, 0));
    HIP_TRY(hipMemRelease(h));  // the mapping keeps the memory
  }
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof acc);
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = c->device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  HIP_TRY(hipMemSetAccess(va, total, &acc, 1));
  *out = (char *)va;
  return 0;
}

// how: 0 hipMalloc(bytes), 1 hipMalloc(next power of two), 2 VMM with chunks of `arg` MiB, 3 offset `arg` MiB inside ONE arena
// (hipMalloc of FX_DEBUG_ARENA_GB, default 32, taken at the first use), 4 hipExtMallocWithFlags(uncached)
static int dbg_alloc(fx_context *c, size_t bytes, int how, int64_t arg, char **out) {
  *out = nullptr;
  switch (how) {
    case 0: HIP_TRY(hipMalloc((void **)out, bytes)); return 0;
    case 1: {
      size_t p = (size_t)1 << 20;
      while (p < bytes) p <<= 1;
      HIP_TRY(hipMalloc((void **)out, p));
      return 0;
    }
    case 2: return dbg_vmm_alloc(c, bytes, (size_t)std::max<int64_t>(arg, 2) << 20, out);
    case 3: {
      if (!g_dbg_arena.base) {
        const char *e = getenv("FX_DEBUG_ARENA_GB");
        g_dbg_arena.bytes = (size_t)(e ? atoi(e) : 32) << 30;
        HIP_TRY(hipMalloc((void **)&g_dbg_arena.base, g_dbg_arena.bytes));
      }
      const size_t off = (size_t)arg << 20;
      if (off + bytes > g_dbg_arena.bytes) { g_fx_error = "fx_debug_replace: offset past the arena"; return FX_ERROR_RUNTIME; }
      *out = g_dbg_arena.base + off;
      return 0;
    }
    case 4: HIP_TRY(hipExtMallocWithFlags((void **)out, bytes, hipDeviceMallocUncached)); return 0;
    default: g_fx_error = "fx_debug_replace: unknown strategy"; return FX_ERROR_RUNTIME;
  }
}

// Take the arena NOW (before the library has allocated anything else, when called right after fx_create).
extern "C" int fx_debug_arena(fx_context *c, int gib) {
  HIP_TRY(hipSetDevice(c->device));
  if (g_dbg_arena.base) return 0;
  g_dbg_arena.bytes = (size_t)gib << 30;
  HIP_TRY(hipMalloc((void **)&g_dbg_arena.base, g_dbg_arena.bytes));
  return 0;
}

static double **dbg_vector(fx_context *c, int sel) {
  if (sel == -1) return &c->Bs;
  if (sel == -2) return &c->Xs;
  if (sel >= 0 && sel < 10) return &c->W[sel];
  return nullptr;
}

// Move one array of the SpMV to new memory.  what: 0 M.val2 (refilled from the resident CSR values), 1 M.col2, 2 M.pair_ptr +
// M.slice_order, 10 + k vector (k = 0..9 W[k], 10 Bs, 11 Xs).  The old memory is NOT released (g_dbg_leak) unless it lies in the arena.
extern "C" int fx_debug_replace(fx_context *c, int what, int how, int64_t arg, uint64_t *addr_out) {
  HIP_TRY(hipSetDevice(c->device));
  if (ensure_solver(c)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipStreamSynchronize(c->stream));
  Bell &M = c->M;
  char *p = nullptr;
  if (what == 0) {
    const size_t bytes = (size_t)M.npairs * 576 * 8;
    if (dbg_alloc(c, bytes, how, arg, &p)) return FX_ERROR_RUNTIME;
    if (!M.arena_owner) g_dbg_leak.push_back(M.val2_base);
    else arena_release(M.arena_owner, M.val2_base);
    M.arena_owner = nullptr;
    M.val2_base = p; M.val2 = (double *)p; M.val2_bytes = bytes;
    if (bell_fill_values(c, M)) return FX_ERROR_RUNTIME;
  } else if (what == 1) {
    const size_t bytes = (size_t)M.npairs * 64 * 4;
    if (dbg_alloc(c, bytes, how, arg, &p)) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpy(p, M.col2, bytes, hipMemcpyDeviceToDevice));
    g_dbg_leak.push_back(M.col2);
    M.col2 = (int *)p;
  } else if (what == 2) {
    const size_t bytes = ((size_t)M.nslices + 1) * 4;
    if (dbg_alloc(c, 2 * bytes + 256, how, arg, &p)) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpy(p, M.pair_ptr, bytes, hipMemcpyDeviceToDevice));
    g_dbg_leak.push_back(M.pair_ptr);
    M.pair_ptr = (int32_t *)p;
    if (M.slice_order) {
      char *q = p + (bytes + 255) / 256 * 256;
      HIP_TRY(hipMemcpy(q, M.slice_order, (size_t)M.nslices * 4, hipMemcpyDeviceToDevice));
      g_dbg_leak.push_back(M.slice_order);
      M.slice_order = (int32_t *)q;
    }
  } else if (what >= 10 && what < 22) {
    double **v = dbg_vector(c, what == 20 ? -1 : what == 21 ? -2 : what - 10);
    const size_t bytes = (size_t)c->wlen * 8;
    if (dbg_alloc(c, bytes, how, arg, &p)) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpy(p, *v, bytes, hipMemcpyDeviceToDevice));
    g_dbg_leak.push_back(*v);
    *v = (double *)p;
  } else {
    g_fx_error = "fx_debug_replace: unknown array";
    return FX_ERROR_RUNTIME;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (addr_out) *addr_out = (uint64_t)(uintptr_t)p;
  return 0;
}

// Addresses of the arrays the SpMV touches: out[0] val2, [1] col2, [2] pair_ptr, [3] slice_order, [4] Bs, [5] Xs, [6..15] W[0..9], [16] partials
extern "C" int fx_debug_addresses(fx_context *c, uint64_t out[20]) {
  const Bell &M = c->M;
  out[0] = (uint64_t)(uintptr_t)M.val2; out[1] = (uint64_t)(uintptr_t)M.col2; out[2] = (uint64_t)(uintptr_t)M.pair_ptr;
  out[3] = (uint64_t)(uintptr_t)M.slice_order; out[4] = (uint64_t)(uintptr_t)c->Bs; out[5] = (uint64_t)(uintptr_t)c->Xs;
  for (int k = 0; k < 10; k++) out[6 + k] = (uint64_t)(uintptr_t)c->W[k];
  out[16] = (uint64_t)(uintptr_t)c->partials;
  out[17] = (uint64_t)M.npairs * 576 * 8; out[18] = (uint64_t)c->wlen * 8; out[19] = 0;
  return 0;
}

// The blockIdx -> virtual workgroup map of the SpMV as a wg_list (the kernel applies it after its own xcd_block):
// map 0 = the kernel's own (each XCD one contiguous eighth of the walk), 1 = identity (consecutive workgroups round-robin over the
// XCDs: all eight walk the same region together), C >= 2 = each XCD takes chunks of C consecutive workgroups, every eighth chunk.
static int32_t *g_dbg_wgmap = nullptr;
static int g_dbg_wgmap_kind = -1, g_dbg_wgmap_n = 0;
static int dbg_xcd_block(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7, k = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
static int dbg_wgmap(fx_context *c, int map, int nwg, const int32_t **out) {
  *out = nullptr;
  if (map <= 0) return 0;
  if (g_dbg_wgmap && g_dbg_wgmap_kind == map && g_dbg_wgmap_n == nwg) { *out = g_dbg_wgmap; return 0; }
  std::vector<int32_t> f((size_t)nwg, -1);  // blockIdx -> virtual workgroup
  if (map == 1) {
    for (int b = 0; b < nwg; b++) f[b] = b;
  } else {
    std::vector<std::vector<int32_t>> lst(8);
    for (int j = 0, w = 0; w < nwg; j++)
      for (int k = 0; k < map && w < nwg; k++, w++) lst[j & 7].push_back(w);
    std::vector<int32_t> left;
    for (int x = 0; x < 8; x++) {
      size_t k = 0;
      for (int b = x; b < nwg; b += 8, k++)
        if (k < lst[x].size()) f[b] = lst[x][k];
      for (; k < lst[x].size(); k++) left.push_back(lst[x][k]);
    }
    size_t li = 0;
    for (int b = 0; b < nwg; b++)
      if (f[b] < 0) f[b] = left[li++];
  }
  std::vector<int32_t> wl((size_t)nwg);
  for (int b = 0; b < nwg; b++) wl[dbg_xcd_block(b, nwg)] = f[b];
  if (g_dbg_wgmap) (void)hipFree(g_dbg_wgmap);
  HIP_TRY(hipMalloc((void **)&g_dbg_wgmap, (size_t)nwg * 4));
  HIP_TRY(hipMemcpy(g_dbg_wgmap, wl.data(), (size_t)nwg * 4, hipMemcpyHostToDevice));
  g_dbg_wgmap_kind = map; g_dbg_wgmap_n = nwg;
  *out = g_dbg_wgmap;
  return 0;
}

// y = A x with x, y chosen (-1 Bs, -2 Xs, 0..9 W[k]); dot as k_spmv's DOT; the mean of `nrepeat` launches after one untimed.
// kind 1: k_stream_read over val2 instead of the product.  map: the workgroup map (dbg_wgmap).
extern "C" int fx_debug_spmv_ms(fx_context *c, int kind, int xsel, int ysel, int dot, int nrepeat, float *ms_out, int map) {
  HIP_TRY(hipSetDevice(c->device));
  if (ensure_solver(c)) return FX_ERROR_RUNTIME;
  double **xv = dbg_vector(c, xsel), **yv = dbg_vector(c, ysel);
  if (!xv || !yv) { g_fx_error = "fx_debug_spmv_ms: vector selector"; return FX_ERROR_RUNTIME; }
  const int nwg = spmv_nparts(c);
  const int64_t n2 = (int64_t)c->M.npairs * 576 / 2;
  const int g = std::min(256 * 32, (int)c->max_partials);
  const int32_t *wl = nullptr;
  if (dbg_wgmap(c, map, nwg, &wl)) return FX_ERROR_RUNTIME;
  struct OrderGuard {  // map -1: the slices in ASCENDING (= storage) order instead of the spatial walk, for this call only
    Bell &M; int32_t *keep;
    OrderGuard(Bell &m, bool off) : M(m), keep(m.slice_order) { if (off) M.slice_order = nullptr; }
    ~OrderGuard() { M.slice_order = keep; }
  } og(c->M, map == -1);
  auto one = [&]() -> int {
    if (kind == 1) {
      hipLaunchKernelGGL(k_stream_read, dim3(g), dim3(FX_BLOCK), 0, c->stream, n2, (const double2 *)c->M.val2, c->partials);
      return 0;
    }
    return spmv_launch(c, 0, dot, *xv, nullptr, *yv, nullptr, 0, wl, nwg);
  };
  if (one()) return FX_ERROR_RUNTIME;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < nrepeat; i++)
    if (one()) return FX_ERROR_RUNTIME;
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipGetLastError());
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_out = ms / std::max(nrepeat, 1);
  return 0;
}

// One preconditioner apply as a traced plane march (fx_march.h): per chunk 8 words -- forward start / end, backward start / end in
// ticks of the 100 MHz constant clock (relative to the earliest start), rounds of wave 0 that found an entry missing and its re-reads,
// forward / backward.  out: 8 * chunks doubles.
extern "C" int fx_debug_march_trace(fx_context *c, double *out, int32_t cap, int32_t *nchunks) {
  HIP_TRY(hipSetDevice(c->device));
  MarchDev &M = c->ssor.march;
  if (!M.ok) { g_fx_error = "fx_debug_march_trace: no march programs on this context"; return FX_ERROR_RUNTIME; }
  if (ensure_work(c)) return FX_ERROR_RUNTIME;
  *nchunks = M.nchunks;
  if (cap < 8 * M.nchunks) { g_fx_error = "fx_debug_march_trace: buffer too small"; return FX_ERROR_RUNTIME; }
  unsigned long long *tr = nullptr;
  if (dev_alloc(&tr, (size_t)8 * M.nchunks)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemsetAsync(tr, 0, (size_t)64 * M.nchunks, c->stream));
  HIP_TRY(hipMemsetAsync(c->W[6], 0, (size_t)24 * c->A.NP, c->stream));
  int rc = march_apply(c, c->W[6], c->W[7], nullptr, tr);
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = FX_ERROR_RUNTIME;
  std::vector<unsigned long long> h((size_t)8 * M.nchunks);
  if (!rc && hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FX_ERROR_RUNTIME;
  dev_free(tr);
  if (rc) return rc;
  unsigned long long t0 = ~0ull;
  for (int32_t k = 0; k < M.nchunks; k++) if (h[8 * k]) t0 = std::min(t0, h[8 * k]);
  for (int32_t k = 0; k < M.nchunks; k++) {
    for (int j = 0; j < 4; j++) out[8 * k + j] = h[8 * k + j] ? (double)(h[8 * k + j] - t0) : -1.0;
    for (int j = 4; j < 8; j++) out[8 * k + j] = (double)h[8 * k + j];
  }
  return 0;
}

// Per-round timeline of one chunk's forward sweep in a traced march apply: out[k] = microseconds (from the workgroup's entry into the kernel) at which
// round k passed its barrier, negative if that round had to wait for a far entry.  Returns the number of rounds in *nrounds.
extern "C" int fx_debug_march_rounds(fx_context *c, int32_t chunk, double *out, int32_t cap, int32_t *nrounds) {
  HIP_TRY(hipSetDevice(c->device));
  MarchDev &M = c->ssor.march;
  if (!M.ok || chunk < 0 || chunk >= M.nchunks) { g_fx_error = "fx_debug_march_rounds: no march programs / no such chunk"; return FX_ERROR_RUNTIME; }
  if (ensure_work(c)) return FX_ERROR_RUNTIME;
  const int32_t nr = M.F.h_round_ptr[chunk + 1] - M.F.h_round_ptr[chunk];
  *nrounds = nr;
  if (cap < nr) { g_fx_error = "fx_debug_march_rounds: buffer too small"; return FX_ERROR_RUNTIME; }
  unsigned long long *tr = nullptr;
  if (dev_alloc(&tr, (size_t)nr + 1)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemsetAsync(tr, 0, ((size_t)nr + 1) * 8, c->stream));
  HIP_TRY(hipMemsetAsync(c->W[6], 0, (size_t)24 * c->A.NP, c->stream));
  int rc = march_apply(c, c->W[6], c->W[7], nullptr, nullptr, tr, chunk);
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = FX_ERROR_RUNTIME;
  std::vector<unsigned long long> h((size_t)nr + 1);
  if (!rc && hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FX_ERROR_RUNTIME;
  dev_free(tr);
  if (rc) return rc;
  const unsigned long long mask = ~(1ull << 63), t0 = h[nr];  // the workgroup's entry into the kernel
  for (int32_t k = 0; k < nr; k++) {
    const double us = 0.01 * (double)((h[k] & mask) - t0);
    out[k] = (h[k] >> 63) ? -us - 1e-9 : us;
  }
  return 0;
}
