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
// strategy -- here an interpolation guess + short walk instead of a cold binary search.
#include <stdint.h>
#include <string.h>

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
  // searchsorted(side='right') = first position with cdf > x, whatever the search strategy.  The probabilities here are
  // (near-)uniform over the live entries (utils.py:789-795: p = (count/n)/count per channel), so cdf[i] ~ (i+1)/n and
  // position floor(x*n) is the answer or its neighbour: start there and walk to the exact boundary; a walk that gets long
  // (a strongly non-uniform p) falls back to bisection of the remaining range.  One predictable compare per draw instead of
  // a cold binary search (NumPy) or a bucket scan.
  memset(sel, 0, (size_t)n);
  int64_t added = 0;
  const double dn = (double)n;
  for (int64_t j = 0; j < k; ++j) {
    const double v = x[j];                                          // in [0, 1)
    int64_t idx = (int64_t)(v * dn);
    if (idx >= n) idx = n - 1;
    if (cdf[idx] > v) {                                             // answer <= idx: walk down while the left neighbour also exceeds v
      int steps = 0;
      while (idx > 0 && cdf[idx - 1] > v) {
        --idx;
        if (++steps == 8) {                                         // bisect [0, idx]: first position with cdf > v
          int64_t lo = 0, hi = idx;
          while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cdf[mid] > v) hi = mid; else lo = mid + 1; }
          idx = lo;
          break;
        }
      }
    } else {                                                        // answer > idx
      int steps = 0;
      ++idx;
      while (idx < n && !(cdf[idx] > v)) {
        ++idx;
        if (++steps == 8) {
          int64_t lo = idx, hi = n;                                 // first position in [idx, n) with cdf > v (n if none)
          while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cdf[mid] > v) hi = mid; else lo = mid + 1; }
          idx = lo;
          break;
        }
      }
      if (idx >= n) idx = n - 1;                                    // unreachable for x < 1 (cdf[n-1] == 1.0)
    }
    // np.unique(return_index) + sort = first occurrences, in draw order.  Branch-free (a repeat overwrites the next free
    // slot and does not advance): `found` holds one element more than can ever be found.
    found[n_found + added] = live_idx[idx];
    added += sel[idx] ^ 1;
    sel[idx] = 1;
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
