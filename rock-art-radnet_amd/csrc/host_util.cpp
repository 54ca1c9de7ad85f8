// Host-side helper (no GPU work): one round of NumPy's legacy RandomState.choice(n, size, replace=False, p=p).
//
// utils.calc_region_props disables surplus anchors with np.random.choice(..., replace=False, p=...) on ~20 000
// negatives (utils.py:797,812); the draws come from NumPy's global MT19937 stream, which is part of the
// reference's contract (train.py:41,134), so the sampling must stay on the host and consume that stream exactly
// as NumPy does.  NumPy's algorithm loops "draw k uniforms -> cumsum(p) -> searchsorted -> keep first
// occurrences -> zero the found probabilities" about ln(n) times; in NumPy that costs 4-8 ms per image, which
// made the whole training step host-bound.  This restates ONE round of that loop in C (the uniforms still come
// from np.random.random_sample on the Python side, so the RNG stream is consumed identically); tests compare the
// result with np.random.choice itself.
//
// Exactness: cumsum is the same strictly sequential chain of fp64 adds, the normalisation the same IEEE
// division, and searchsorted(side='right') is order-defined (first idx with cdf[idx] > x) whatever the search
// strategy -- here a 4096-bucket jump table + short scan instead of a cold binary search.
#include <stdint.h>
#include <string.h>

namespace {
constexpr int kBuckets = 4096;
}

extern "C" int64_t radnet_host_choice_round(double* live_p, int64_t* live_idx, int64_t* n_live_io, int64_t* found, int64_t n_found,
                                            const double* x, int64_t k, double* cdf, uint8_t* sel) {
  // live_p / live_idx: the entries of p that are still non-zero, in increasing index order.  Adding the zeroed
  // entries would add 0.0 to the running sum, so the cumsum over the live entries alone yields bit-identical
  // cdf values, and searchsorted(side='right') can only ever land on a live entry.
  const int64_t n = *n_live_io;
  if (n <= 0 || k <= 0) return 0;
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {                                 // np.cumsum: strictly sequential fp64 adds
    acc += live_p[i];
    cdf[i] = acc;
  }
  const double total = cdf[n - 1];
  for (int64_t i = 0; i < n; ++i) cdf[i] /= total;                  // cdf /= cdf[-1]  (vectorises: vdivpd)
  // jump table: start[b] = first live position with cdf > b / kBuckets
  static thread_local int64_t start[kBuckets + 1];
  {
    int64_t idx = 0;
    for (int b = 0; b <= kBuckets; ++b) {
      const double edge = (double)b / (double)kBuckets;
      while (idx < n && !(cdf[idx] > edge)) ++idx;
      start[b] = idx;
    }
  }
  memset(sel, 0, (size_t)n);
  int64_t added = 0;
  for (int64_t j = 0; j < k; ++j) {
    const double v = x[j];                                          // in [0, 1)
    int b = (int)(v * (double)kBuckets);                            // exact: power-of-two scaling
    if (b < 0) b = 0;
    if (b >= kBuckets) b = kBuckets - 1;
    int64_t idx = start[b];                                         // first position with cdf > b/B  (<= answer)
    while (idx < n && !(cdf[idx] > v)) ++idx;                       // searchsorted(side='right'): first cdf > x
    if (idx >= n) idx = n - 1;                                      // unreachable for x < 1
    if (!sel[idx]) {                                                // np.unique(return_index) + sort = first occurrences, in order
      sel[idx] = 1;
      found[n_found + added] = live_idx[idx];
      ++added;
    }
  }
  int64_t w = 0;                                                    // p[found] = 0  ==  drop them from the live lists
  for (int64_t i = 0; i < n; ++i) {
    if (!sel[i]) {
      live_p[w] = live_p[i];
      live_idx[w] = live_idx[i];
      ++w;
    }
  }
  *n_live_io = w;
  return added;
}
