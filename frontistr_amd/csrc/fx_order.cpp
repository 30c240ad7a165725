// Host-side orderings of libfistr_hip: the level ordering + greedy multicolouring the
// reference uses for its threaded SSOR, so that the GPU sweep applies the SAME
// preconditioner (same colours, same iteration counts) as the reference's OpenMP path.
//
//   reference: hecmw1/src/solver/matrix/hecmw_matrix_ordering_CM.f90:16-178
//              hecmw1/src/solver/matrix/hecmw_matrix_ordering_MC.f90:15-72
//              used by precond/33/hecmw_precond_SSOR_33.f90:102-111
//
// Work is on a 0-based adjacency (lower neighbours first, then upper, halo columns
// dropped) built once; the up-to-5 candidate start nodes are explored concurrently.
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace fxo {

static int order_threads() {
  const char *e = getenv("FX_HOST_THREADS");
  int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(n, 32));
}
template <class F>
static void par_for(int64_t n, F f) {  // f(begin, end) on contiguous chunks
  const int nt = (int)std::min<int64_t>(order_threads(), std::max<int64_t>(1, n / 8192));
  if (nt <= 1) { f((int64_t)0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
  for (auto &t : th) t.join();
}

struct Graph {
  int32_t n = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> adj;  // lower neighbours (ascending) then upper (ascending), internal only
};

Graph build_graph(int32_t N, const int32_t *indexL, const int32_t *itemL, const int32_t *indexU,
                  const int32_t *itemU) {
  Graph g;
  g.n = N;
  g.ptr.assign((size_t)N + 1, 0);
  par_for(N, [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; i++) {
      int64_t c = indexL[i + 1] - indexL[i];
      for (int32_t j = indexU[i]; j < indexU[i + 1]; j++) c += (itemU[j] <= N);
      g.ptr[i + 1] = c;
    }
  });
  for (int32_t i = 0; i < N; i++) g.ptr[i + 1] += g.ptr[i];
  g.adj.resize((size_t)g.ptr[N]);
  par_for(N, [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; i++) {
      int64_t w = g.ptr[i];
      for (int32_t j = indexL[i]; j < indexL[i + 1]; j++) g.adj[w++] = itemL[j] - 1;
      for (int32_t j = indexU[i]; j < indexU[i + 1]; j++)
        if (itemU[j] <= N) g.adj[w++] = itemU[j] - 1;
    }
  });
  return g;
}

// Breadth-first level structure from `start`; nodes are appended in discovery order
// (no degree sort inside a level, exactly like ordering_CM_inner).  When a component is
// exhausted the lowest-numbered unvisited node opens the next level.
static int32_t level_order(const Graph &g, int32_t start, std::vector<int32_t> &seq) {
  const int32_t n = g.n;
  std::vector<uint8_t> seen((size_t)n, 0);
  seq.clear();
  seq.reserve(n);
  seq.push_back(start);
  seen[start] = 1;
  int32_t nlevel = 1, lo = 0, hi = 1, next_unseen = 0;
  while ((int32_t)seq.size() < n) {
    for (int32_t q = lo; q < hi && (int32_t)seq.size() < n; q++) {
      const int32_t u = seq[q];
      for (int64_t e = g.ptr[u]; e < g.ptr[u + 1]; e++) {
        const int32_t v = g.adj[e];
        if (!seen[v]) {
          seen[v] = 1;
          seq.push_back(v);
          if ((int32_t)seq.size() == n) break;
        }
      }
    }
    if ((int32_t)seq.size() == hi) {  // nothing discovered: jump to any unvisited node
      while (seen[next_unseen]) next_unseen++;
      seen[next_unseen] = 1;
      seq.push_back(next_unseen);
    }
    lo = hi;
    hi = (int32_t)seq.size();
    nlevel++;
  }
  return nlevel;
}

// "RCM" of the reference = level ordering from the best of <= 5 minimum-degree starts,
// followed by reverse_ordering, which maps node id k -> N-1-k (0-based) instead of
// reversing the sequence.  Returns the visiting sequence (new -> old, 0-based).
// the up-to-5 minimum-degree start candidates (find_minimum_degrees, hecmw_matrix_ordering_CM.f90:138-167)
std::vector<int32_t> rcm_starts_deg(int32_t n, const int32_t *deg_of) {
  int64_t degmin = n;
  std::vector<int32_t> starts;
  int32_t nties = 0;
  for (int32_t i = 0; i < n; i++) {
    const int64_t deg = deg_of[i];
    if (deg == 0) continue;
    if (deg < degmin) { degmin = deg; starts.assign(1, i); nties = 1; }
    else if (deg == degmin) { if (++nties <= 5) starts.push_back(i); }
  }
  if (starts.empty()) starts.push_back(0);
  return starts;
}
std::vector<int32_t> rcm_starts(const Graph &g) {
  std::vector<int32_t> deg((size_t)g.n);
  for (int32_t i = 0; i < g.n; i++) deg[i] = (int32_t)(g.ptr[i + 1] - g.ptr[i]);
  return rcm_starts_deg(g.n, deg.data());
}
int32_t level_order_host(const Graph &g, int32_t start, std::vector<int32_t> &seq) { return level_order(g, start, seq); }

std::vector<int32_t> rcm_sequence(const Graph &g) {
  const int32_t n = g.n;
  const std::vector<int32_t> starts = rcm_starts(g);
  std::vector<std::vector<int32_t>> seqs(starts.size());
  std::vector<int32_t> nlev(starts.size(), 0);
  std::vector<std::thread> th;
  for (size_t s = 0; s < starts.size(); s++)
    th.emplace_back([&, s] { nlev[s] = level_order(g, starts[s], seqs[s]); });
  for (auto &t : th) t.join();
  size_t best = 0;
  for (size_t s = 1; s < starts.size(); s++)
    if (nlev[s] > nlev[best]) best = s;
  std::vector<int32_t> seq = std::move(seqs[best]);
  for (auto &v : seq) v = n - 1 - v;  // reverse_ordering: id mirror
  return seq;
}

// Greedy multicolouring walking `seq`; each colour is capped at n / ncolor_in nodes and
// neighbours of a chosen node are blocked for the current colour only.
// Out: perm (new -> old, 0-based) colour by colour, colorindex (ncolor+1 entries).
void multicolor(const Graph &g, const std::vector<int32_t> &seq, int ncolor_in, std::vector<int32_t> &perm,
                std::vector<int32_t> &colorindex) {
  const int32_t n = g.n;
  const int32_t cap = n / ncolor_in;
  // The walk is sequential by nature (a node is picked iff no neighbour earlier in the order was picked this round: the
  // lexicographically first independent set, cut off at `cap`), so what can be bought is locality: the graph is relabelled by
  // position in `seq` first (in parallel), after which the marks a pick touches lie within one level width of the walker instead
  // of all over the mesh.  Same picks, same order: only the names of the nodes differ during the walk.
  std::vector<int32_t> pos((size_t)n);
  par_for(n, [&](int64_t a, int64_t b) { for (int64_t k = a; k < b; k++) pos[seq[k]] = (int32_t)k; });
  std::vector<int64_t> rptr((size_t)n + 1, 0);
  par_for(n, [&](int64_t a, int64_t b) { for (int64_t k = a; k < b; k++) rptr[k + 1] = g.ptr[seq[k] + 1] - g.ptr[seq[k]]; });
  for (int32_t k = 0; k < n; k++) rptr[k + 1] += rptr[k];
  std::vector<int32_t> radj((size_t)rptr[n]);
  par_for(n, [&](int64_t a, int64_t b) {
    for (int64_t k = a; k < b; k++) {
      int64_t w = rptr[k];
      const int32_t u = seq[k];
      for (int64_t e = g.ptr[u]; e < g.ptr[u + 1]; e++) radj[w++] = pos[g.adj[e]];
    }
  });
  std::vector<int32_t> mark((size_t)n, 0);  // by position: >0 coloured, -1 blocked this round
  std::vector<int32_t> blocked;
  // the reference rescans the whole sequence for every colour (:29-60); visiting only the nodes still uncoloured,
  // in the same order, picks the same nodes
  std::vector<int32_t> rem((size_t)n), next;
  for (int32_t k = 0; k < n; k++) rem[k] = k;
  next.reserve(rem.size());
  perm.clear();
  perm.reserve(n);
  colorindex.assign(1, 0);
  for (int32_t color = 1; (int32_t)perm.size() < n; color++) {
    int32_t cnt = 0;
    blocked.clear();
    next.clear();
    size_t q = 0;
    for (; q < rem.size(); q++) {
      const int32_t k = rem[q];
      if (mark[k] != 0) { next.push_back(k); continue; }  // blocked in this round: stays for the next one
      mark[k] = color;
      perm.push_back(seq[k]);
      cnt++;
      if (cnt == cap || (int32_t)perm.size() == n) { q++; break; }
      for (int64_t e = rptr[k]; e < rptr[k + 1]; e++) {
        const int32_t v = radj[e];
        if (mark[v] == 0) { mark[v] = -1; blocked.push_back(v); }
      }
    }
    next.insert(next.end(), rem.begin() + q, rem.end());  // not visited because the colour was full
    rem.swap(next);
    colorindex.push_back((int32_t)perm.size());
    for (int32_t v : blocked)
      if (mark[v] == -1) mark[v] = 0;
  }
}

// Greedy colouring of the elements so that no two elements of a colour share a node: the scatter of one colour into the
// block CRS arrays then needs no atomics (every matrix block is touched by at most one element per launch) and the assembled
// matrix becomes bitwise reproducible.  Natural element order on a structured hex mesh gives the 8 parity colours.
// conn: nn 1-based node ids per element.  Returns false when a node is shared by more than 64 elements.
bool color_elements(int32_t n_elem, int nn, const int32_t *conn, int32_t NP, std::vector<int32_t> &order,
                    std::vector<int32_t> &offsets) {
  std::vector<uint64_t> used((size_t)NP, 0);
  std::vector<uint8_t> col((size_t)n_elem);
  std::vector<int32_t> count(64, 0);
  int ncol = 0;
  for (int32_t e = 0; e < n_elem; e++) {
    uint64_t m = 0;
    for (int k = 0; k < nn; k++) m |= used[conn[(size_t)nn * e + k] - 1];
    if (~m == 0) return false;
    const int cidx = __builtin_ctzll(~m);
    for (int k = 0; k < nn; k++) used[conn[(size_t)nn * e + k] - 1] |= (uint64_t)1 << cidx;
    col[e] = (uint8_t)cidx;
    count[cidx]++;
    ncol = std::max(ncol, cidx + 1);
  }
  offsets.assign((size_t)ncol + 1, 0);
  for (int k = 0; k < ncol; k++) offsets[k + 1] = offsets[k] + count[k];
  std::vector<int32_t> pos(offsets.begin(), offsets.end() - 1);
  order.resize((size_t)n_elem);
  for (int32_t e = 0; e < n_elem; e++) order[pos[col[e]]++] = e;
  return true;
}

}  // namespace fxo
