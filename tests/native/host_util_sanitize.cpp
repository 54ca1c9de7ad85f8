// AddressSanitizer / UndefinedBehaviorSanitizer driver for the host-side C of libradnet_hip (csrc/host_util.cpp:
// radnet_host_choice_round, one round of NumPy's RandomState.choice(replace=False, p) -- utils.py:797,812).  Built and run by
// tests/test_host_sanitizers.py on the CPU (GPU sanitizers are not available on the pool).  Drives the function the way
// radnet_hip.engine.choice_without_replacement does -- rounds until `size` distinct items are found -- on uniform and strongly
// non-uniform probabilities, with buffers of EXACTLY the documented sizes, so an out-of-range access is an ASan report; the
// results are checked against a plain restatement of one NumPy round (cumsum, normalise, searchsorted right, first occurrences).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

extern "C" int64_t radnet_host_choice_round(double* live_p, int64_t* live_idx, int64_t* n_live_io, int64_t* found, int64_t n_found,
                                            const double* x, int64_t k, double* cdf, uint8_t* sel);

static uint64_t s_rng = 0x9E3779B97F4A7C15ull;
static double uniform01() {
  s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17;
  return (double)(s_rng >> 11) * (1.0 / 9007199254740992.0);
}

static int run_case(int64_t n, int64_t size, int shape) {
  std::vector<double> p((size_t)n);
  double tot = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    p[(size_t)i] = shape == 0 ? 1.0 : (shape == 1 ? (i % 7 == 0 ? 50.0 : 0.01) : (double)(i + 1) * (double)(i + 1));
    tot += p[(size_t)i];
  }
  for (auto& v : p) v /= tot;
  // exactly-sized buffers, each its own allocation (heap redzones on both sides)
  std::vector<double> live_p(p), cdf((size_t)n), ref_p(p);
  std::vector<int64_t> live_idx((size_t)n), found((size_t)size + 1);      // `found` holds one element more than can be found
  std::vector<uint8_t> sel((size_t)n);
  for (int64_t i = 0; i < n; ++i) live_idx[(size_t)i] = i;
  int64_t n_live = n, n_found = 0;
  std::vector<int64_t> ref_found;
  int rounds = 0;
  while (n_found < size) {
    const int64_t k = size - n_found;
    std::vector<double> x((size_t)k);
    for (auto& v : x) v = uniform01();
    // reference round on the dense p (zeros where already found)
    std::vector<double> c((size_t)n);
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) { acc += ref_p[(size_t)i]; c[(size_t)i] = acc; }
    for (auto& v : c) v /= acc;
    std::vector<int64_t> round_new;
    for (int64_t j = 0; j < k; ++j) {
      const int64_t idx = std::upper_bound(c.begin(), c.end(), x[(size_t)j]) - c.begin();
      if (idx < n && std::find(round_new.begin(), round_new.end(), idx) == round_new.end()) round_new.push_back(idx);
    }
    for (int64_t idx : round_new) { ref_found.push_back(idx); ref_p[(size_t)idx] = 0.0; }
    const int64_t added = radnet_host_choice_round(live_p.data(), live_idx.data(), &n_live, found.data(), n_found, x.data(), k, cdf.data(), sel.data());
    if (added != (int64_t)round_new.size()) { fprintf(stderr, "case n=%lld: round %d found %lld, reference %zu\n", (long long)n, rounds, (long long)added, round_new.size()); return 1; }
    n_found += added;
    if (++rounds > 200) { fprintf(stderr, "no progress\n"); return 1; }
  }
  for (int64_t i = 0; i < size; ++i)
    if (found[(size_t)i] != ref_found[(size_t)i]) { fprintf(stderr, "case n=%lld: item %lld differs\n", (long long)n, (long long)i); return 1; }
  if (n_live != n - size) { fprintf(stderr, "live count\n"); return 1; }
  return 0;
}

int main() {
  const int64_t cases[][2] = {{1, 1}, {2, 1}, {5, 5}, {64, 10}, {1000, 256}, {20000, 128}, {20000, 19999}, {333, 332}};
  int bad = 0, n_cases = 0;
  for (const auto& c : cases)
    for (int shape = 0; shape < 3; ++shape) { bad += run_case(c[0], c[1], shape); ++n_cases; }
  // degenerate calls return 0 without touching anything
  int64_t zero = 0;
  if (radnet_host_choice_round(nullptr, nullptr, &zero, nullptr, 0, nullptr, 4, nullptr, nullptr) != 0) ++bad;
  printf("%d cases, %d failed\n", n_cases, bad);
  return bad ? 1 : 0;
}
