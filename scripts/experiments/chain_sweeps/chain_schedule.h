// Round-2 negative result, removed from the product in round 3: kept for reference only, not built.
// Units and visiting orders of the chain sweeps (k_tri_chain).  A unit is a CHAIN: a maximal run of consecutive rows in which
// every row depends on the one before it in the sweep's direction (a pencil of a structured mesh), cut at `maxlen` rows.
// A unit that started in the middle of such a run could not begin before its predecessor had finished -- with fixed-length
// units the whole sweep degenerates into one serial chain.  Units are sorted by an estimate of when they can start: a row
// costs one step, a value from another unit arrives `hop` steps after its row is done; a unit can start when, for every row,
// its outside dependencies will have arrived by the time the walk reaches that row.  The key is also kept above the keys of
// all units it depends on, so the order is topological: a wave that takes the entries k, k + G, ... in order only ever waits
// for entries before its own.  Returns the estimated makespans (steps) of the two sweeps.
struct ChainPlan {
  std::vector<int32_t> startF, ordF, startB, ordB;
  int64_t spanF = 0, spanB = 0;
};
static void chain_schedule(int32_t N, int hop, int maxlen, const int32_t *iL, const int32_t *jL, const int32_t *iU, const int32_t *jU,
                           ChainPlan &P) {
  std::vector<int64_t> done((size_t)N);
  std::vector<int32_t> unit_of((size_t)N);
  auto sorted = [&](const std::vector<int64_t> &key, std::vector<int32_t> &ord) {
    ord.resize(key.size());
    for (size_t k = 0; k < key.size(); k++) ord[k] = (int32_t)k;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return key[x] < key[y]; });
  };
  {  // forward: row i continues the chain of row i - 1 when i - 1 is among its lower columns (ascending: the last one)
    P.startF.assign(1, 0);
    for (int32_t i = 1; i < N; i++) {
      const bool cont = iL[i + 1] > iL[i] && jL[iL[i + 1] - 1] == i;  // 1-based id of row i - 1
      if (!cont || i - P.startF.back() >= maxlen) P.startF.push_back(i);
    }
    P.startF.push_back(N);
    const int32_t nu = (int32_t)P.startF.size() - 1;
    std::vector<int64_t> key((size_t)nu);
    for (int32_t u = 0; u < nu; u++) {
      const int32_t a = P.startF[u], b = P.startF[u + 1];
      int64_t start = 0, kdep = -1;
      for (int32_t i = a; i < b; i++) {
        unit_of[i] = u;
        for (int32_t j = iL[i]; j < iL[i + 1]; j++) {
          const int32_t cidx = jL[j] - 1;
          if (cidx >= a) continue;
          start = std::max(start, done[cidx] + hop - (i - a));
          kdep = std::max(kdep, key[unit_of[cidx]]);
        }
      }
      start = std::max(start, kdep + 1);
      key[u] = start;
      for (int32_t i = a; i < b; i++) done[i] = start + (i - a) + 1;
      P.spanF = std::max(P.spanF, start + (b - a));
    }
    sorted(key, P.ordF);
  }
  {  // backward: rows descending; row i continues the chain of row i + 1 when i + 1 is its first upper column
    std::vector<int32_t> cuts(1, N);  // descending boundaries
    for (int32_t i = N - 2; i >= 0; i--) {
      const bool cont = iU[i + 1] > iU[i] && jU[iU[i]] == i + 2;  // 1-based id of row i + 1
      if (!cont || cuts.back() - (i + 1) >= maxlen) cuts.push_back(i + 1);
    }
    cuts.push_back(0);
    P.startB.assign(cuts.rbegin(), cuts.rend());  // ascending starts, unit u = rows [startB[u], startB[u + 1])
    const int32_t nu = (int32_t)P.startB.size() - 1;
    std::vector<int64_t> key((size_t)nu);
    for (int32_t u = nu - 1; u >= 0; u--) {
      const int32_t a = P.startB[u], b = P.startB[u + 1];
      int64_t start = 0, kdep = -1;
      for (int32_t i = b - 1; i >= a; i--) {
        unit_of[i] = u;
        for (int32_t j = iU[i]; j < iU[i + 1]; j++) {
          const int32_t cidx = jU[j] - 1;
          if (cidx >= N || cidx < b) continue;
          start = std::max(start, done[cidx] + hop - (b - 1 - i));
          kdep = std::max(kdep, key[unit_of[cidx]]);
        }
      }
      start = std::max(start, kdep + 1);
      key[u] = start;
      for (int32_t i = b - 1; i >= a; i--) done[i] = start + (b - 1 - i) + 1;
      P.spanB = std::max(P.spanB, start + (b - a));
    }
    sorted(key, P.ordB);
  }
}

static int ilu_setup_chain(fx_context *c) {
  SsorDev &S = c->ssor;
  const int32_t N = c->A.N;
  const int32_t *iL = c->h_indexL.data(), *jL = c->h_itemL.data(), *iU = c->h_indexU.data(), *jU = c->h_itemU.data();
  S.chain = false;
  if (c->df_mode != 3 || N < 1) return 0;
  for (int32_t i = 0; i < N; i++)
    if (iL[i + 1] - iL[i] > FX_CH_MAXB || iU[i + 1] - iU[i] > FX_CH_MAXB) return 0;  // the lane mapping holds 16 blocks per row: level sweeps instead
  ChainPlan P;
  chain_schedule(N, c->ch_hop, c->ch_maxlen, iL, jL, iU, jU, P);
  if (getenv("FX_TIMING") && atoi(getenv("FX_TIMING")))
    fprintf(stderr, "[fx timing] chain sweeps: %zu forward / %zu backward chains, estimated makespan %lld / %lld steps (%d levels)\n",
            P.ordF.size(), P.ordB.size(), (long long)P.spanF, (long long)P.spanB, S.ncolor);
  S.ch_nF = (int32_t)P.ordF.size();
  S.ch_nB = (int32_t)P.ordB.size();
  const size_t nz = ((size_t)3 * N + 1) / 2 * 2;  // the tag fill writes 16-byte words
  dev_free(S.ch_ordF); dev_free(S.ch_ordB); dev_free(S.ch_startF); dev_free(S.ch_startB); dev_free(S.ch_zf); dev_free(S.ch_zb);
  if (dev_alloc(&S.ch_ordF, P.ordF.size()) || dev_alloc(&S.ch_ordB, P.ordB.size()) || dev_alloc(&S.ch_startF, P.startF.size()) ||
      dev_alloc(&S.ch_startB, P.startB.size()) || dev_alloc(&S.ch_zf, nz) || dev_alloc(&S.ch_zb, nz))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpy(S.ch_ordF, P.ordF.data(), P.ordF.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.ch_ordB, P.ordB.data(), P.ordB.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.ch_startF, P.startF.data(), P.startF.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.ch_startB, P.startB.data(), P.startB.size() * 4, hipMemcpyHostToDevice));
  S.chain = true;
  return 0;
}

