// Proposal decode + greedy NMS in fp64 with the reference's exact arithmetic (rpn.py:68-172,
// 299-344, 380-455).  Compiled with -ffp-contract=off: NumPy rounds every product and sum
// separately, so no FMA contraction is allowed here.
//
//   decode   one thread per anchor (a,row,col): delta decode, round-half-even, clamp, clip, validity;
//            emits a 64-bit sort key (order-preserving score bits << 32 | flat index), 0 for dropped boxes.
//   sort     descending device radix sort of the keys (rocPRIM, header-only) -> "stable ascending,
//            walk from the end": among equal scores the higher flat index comes first.
//   (alternative for rpn_to_roi, RADNET_PROPOSALS_SELECT=1: select_nms_kernel -- one workgroup: radix SELECT of the best
//    <= 4096 unexamined keys (11-bit digits, histogram in LDS), LDS bitonic sort, integer-arithmetic NMS, band by band;
//    bit-identical, measured slower than the full sort: see radnet_rpn_to_roi)
//   nms      ONE workgroup of 1024 threads walks the sorted candidates 64 at a time:
//            (1) 16 waves test the 64 candidates against all picks so far (picks live in LDS),
//            (2) each wave builds rows of the 64x64 intra-chunk suppression matrix with __ballot,
//            (3) one lane resolves the chunk sequentially with bit operations and appends picks.
//            Work is (#candidates examined) x (#picks), not N^2, and stops at max_boxes.
#include "radnet_internal.h"

#include <rocprim/rocprim.hpp>

namespace {

__device__ __forceinline__ unsigned int sortable_bits(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct DecodeArgs {
  const float* pred;
  int ld, rows, cols, a;
  double std_scaling;
  int use_regr;
  double aw[32], ah[32];
};

struct __attribute__((aligned(8))) IBox {      // integer-valued box (after np.round and the clip to the feature map), 8 bytes
  short x1, y1, x2, y2;
};

__global__ void __launch_bounds__(256) decode_kernel(DecodeArgs g, double4* __restrict__ boxes, IBox* __restrict__ iboxes,
                                                     unsigned long long* __restrict__ keys, int* __restrict__ n_valid) {
  const int hw = g.rows * g.cols;
  const int total = hw * g.a;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int a = idx / hw, pix = idx - a * hw;
  const int row = pix / g.cols, col = pix - row * g.cols;
  const float* p = g.pred + (size_t)pix * g.ld;
  const double aw = g.aw[a], ah = g.ah[a];
  double x = (double)col - aw / 2, y = (double)row - ah / 2, w = aw, h = ah;
  if (g.use_regr) {
    const float ss = (float)g.std_scaling;
    const float tx = p[g.a + 4 * a + 0] / ss, ty = p[g.a + 4 * a + 1] / ss;
    const float tw = p[g.a + 4 * a + 2] / ss, th = p[g.a + 4 * a + 3] / ss;
    const double cx = x + w / 2., cy = y + h / 2.;
    const double cx1 = (double)tx * w + cx, cy1 = (double)ty * h + cy;
    const double w1 = exp((double)tw) * w, h1 = exp((double)th) * h;
    x = rint(cx1 - w1 / 2.);
    y = rint(cy1 - h1 / 2.);
    w = rint(w1);
    h = rint(h1);
  }
  w = fmax(1.0, w);
  h = fmax(1.0, h);
  double x2 = w + x, y2 = h + y;
  const double x1 = fmax(0.0, x), y1 = fmax(0.0, y);
  x2 = fmin((double)(g.cols - 1), x2);
  y2 = fmin((double)(g.rows - 1), y2);
  // rpn.py:163: drop where (x1 - x2 >= 0) | (y1 - y2 >= 0); NaN boxes (the reference would assert) are dropped too
  const bool ok = (x1 < x2) && (y1 < y2);
  unsigned long long key = 0ull;
  if (ok) key = ((unsigned long long)sortable_bits(p[a]) << 32) | (unsigned int)idx;
  keys[idx] = key;
  if (iboxes != nullptr) {          // integer path (use_regr): every coordinate is an integer in [0, 32767]
    IBox b;
    b.x1 = (short)x1; b.y1 = (short)y1; b.x2 = (short)x2; b.y2 = (short)y2;
    iboxes[idx] = ok ? b : IBox{0, 0, 0, 0};
  } else {
    boxes[idx] = make_double4(x1, y1, x2, y2);
    if (ok) atomicAdd(n_valid, 1);
  }
}

__global__ void __launch_bounds__(256) nms_keys_kernel(const double* __restrict__ boxes, const float* __restrict__ probs, int n,
                                                       unsigned long long* __restrict__ keys, int* __restrict__ n_valid,
                                                       int* __restrict__ malformed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const double x1 = boxes[4 * idx], y1 = boxes[4 * idx + 1], x2 = boxes[4 * idx + 2], y2 = boxes[4 * idx + 3];
  if (!(x1 < x2) || !(y1 < y2)) atomicAdd(malformed, 1);     // rpn.py:400-401 asserts
  keys[idx] = ((unsigned long long)sortable_bits(probs[idx]) << 32) | (unsigned int)idx;
  if (idx == 0) *n_valid = n;
}

__device__ __forceinline__ bool suppresses(const double4& pk, double pk_area, const double4& c, double c_area, double thr) {
  // rpn.py:429-447 with i = the picked box
  const double ww = fmax(0.0, fmin(pk.z, c.z) - fmax(pk.x, c.x));
  const double hh = fmax(0.0, fmin(pk.w, c.w) - fmax(pk.y, c.y));
  const double inter = ww * hh;
  const double uni = pk_area + c_area - inter;
  const double d = uni + 1e-6;
  // The reference compares the ROUNDED quotient with thr.  An fp64 division is a ~40-instruction sequence here and
  // this test runs picks x candidates times, so decide without it whenever the outcome cannot depend on rounding:
  // with d > 0, thr > 0 and |inter - thr*d| beyond a 2^-40 relative margin (the products and the quotient each
  // carry < 2^-52), fl(inter/d) > thr  <=>  inter > thr*d.  Everything inside the margin takes the division.
  if (d > 0.0 && thr > 0.0) {
    const double t = thr * d;
    if (inter > t * (1.0 + 0x1p-40)) return true;
    if (inter < t * (1.0 - 0x1p-40)) return false;
  }
  return inter / d > thr;
}

constexpr int kMaxPicks = 1024;

__global__ void __launch_bounds__(1024) nms_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_valid_p,
                                                   const int* __restrict__ malformed, const double4* __restrict__ boxes, double thr,
                                                   int max_boxes, int* __restrict__ out_idx, int* __restrict__ out_count,
                                                   long long* __restrict__ out_boxes, float* __restrict__ out_probs,
                                                   const float* __restrict__ probs_src, int probs_stride, int probs_div) {
  __shared__ double4 pick_box[kMaxPicks];
  __shared__ double pick_area[kMaxPicks];
  __shared__ int pick_idx[kMaxPicks];
  __shared__ double4 cand_box[64];
  __shared__ double cand_area[64];
  __shared__ int cand_idx[64];
  __shared__ unsigned int sup[2];
  __shared__ unsigned long long mat[64];
  __shared__ int s_npicks;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_valid = *n_valid_p;
  if (malformed != nullptr && *malformed != 0) {
    if (tid == 0) *out_count = -1;
    return;
  }
  if (tid == 0) s_npicks = 0;
  __syncthreads();

  // The chunk's candidates come through two dependent global reads (sorted key -> box): the reads of chunk c+1 are
  // issued before chunk c is processed and land while its three phases run, instead of ~1.5 us of exposed latency
  // per chunk on a single workgroup.
  int pre_id = 0;
  double4 pre_box = make_double4(0.0, 0.0, 0.0, 0.0);
  auto fetch = [&](int base) {
    if (tid < 64 && base + tid < n_valid) {
      pre_id = (int)(keys[base + tid] & 0xFFFFFFFFull);
      pre_box = boxes[pre_id];
    }
  };
  fetch(0);
  for (int base = 0; base < n_valid; base += 64) {
    const int nc = min(64, n_valid - base);
    if (tid < 64) {
      if (tid < nc) {
        cand_idx[tid] = pre_id;
        cand_box[tid] = pre_box;
        cand_area[tid] = (pre_box.z - pre_box.x) * (pre_box.w - pre_box.y);
      }
      if (tid < 2) sup[tid] = 0u;
    }
    fetch(base + 64);
    __syncthreads();
    const int npicks = s_npicks;
    // (1) candidates vs existing picks: lane = candidate, wave = slice of the pick list
    {
      bool dead = false;
      if (lane < nc) {
        const double4 c = cand_box[lane];
        const double ca = cand_area[lane];
        for (int p = wave; p < npicks && !dead; p += 16) dead = suppresses(pick_box[p], pick_area[p], c, ca, thr);
      }
      const unsigned long long m = __ballot(dead);
      if (lane == 0 && m) {
        atomicOr(&sup[0], (unsigned int)(m & 0xFFFFFFFFull));
        atomicOr(&sup[1], (unsigned int)(m >> 32));
      }
    }
    // (2) intra-chunk matrix: wave handles rows j = wave, wave+16, ...; lane = victim i (> j)
    for (int j = wave; j < 64; j += 16) {
      bool hit = false;
      if (j < nc && lane < nc && lane > j) hit = suppresses(cand_box[j], cand_area[j], cand_box[lane], cand_area[lane], thr);
      const unsigned long long m = __ballot(hit);
      if (lane == 0) mat[j] = m;
    }
    __syncthreads();
    // (3) sequential resolve of this chunk
    if (tid == 0) {
      unsigned long long alive = ~(((unsigned long long)sup[1] << 32) | sup[0]);
      if (nc < 64) alive &= (1ull << nc) - 1ull;
      int np = npicks;
      while (alive != 0ull && np < max_boxes) {          // visits the survivors only, in candidate order
        const int j = __ffsll((long long)alive) - 1;
        pick_box[np] = cand_box[j];
        pick_area[np] = cand_area[j];
        pick_idx[np] = cand_idx[j];
        ++np;
        alive &= ~mat[j];                                 // mat[j] only has bits above j
        alive &= ~(1ull << j);
      }
      s_npicks = np;
    }
    __syncthreads();
    if (s_npicks >= max_boxes) break;
  }
  const int np = s_npicks;
  if (tid == 0) *out_count = np;
  for (int i = tid; i < np; i += blockDim.x) {
    const int id = pick_idx[i];
    if (out_idx) out_idx[i] = id;
    if (out_boxes) {
      const double4 b = pick_box[i];
      out_boxes[4 * i + 0] = (long long)b.x;     // astype('int'): truncation
      out_boxes[4 * i + 1] = (long long)b.y;
      out_boxes[4 * i + 2] = (long long)b.z;
      out_boxes[4 * i + 3] = (long long)b.w;
    }
    if (out_probs) {
      // flat index = a*hw + pix  ->  pred[pix*ld + a]
      const int a = id / probs_div, pix = id - a * probs_div;
      out_probs[i] = probs_src[(size_t)pix * probs_stride + a];
    }
  }
}

// ---- rpn_to_roi: select + sort + NMS in one workgroup ---------------------------------------------------------------
constexpr int kCap = 4096;          // candidates per band (LDS: 32 KB of keys)
constexpr int kDigitBits = 11, kDigits = 1 << kDigitBits;

// rpn.py:429-447 on integer boxes: inter and union are exact integers, the comparison is the reference's
// fl(inter / (union + 1e-6)) > thr (same margin test as `suppresses`, the division only inside the margin)
__device__ __forceinline__ bool suppresses_int(const IBox& pk, int pk_area, const IBox& c, int c_area, double thr) {
  const int iw = min((int)pk.x2, (int)c.x2) - max((int)pk.x1, (int)c.x1);
  const int ih = min((int)pk.y2, (int)c.y2) - max((int)pk.y1, (int)c.y1);
  if (iw <= 0 || ih <= 0) return false;                   // inter = 0: 0 / d > thr is false for thr > 0 (checked by the launcher)
  const int inter_i = iw * ih;
  const double inter = (double)inter_i, d = (double)(pk_area + c_area - inter_i) + 1e-6;
  const double t = thr * d;
  if (inter > t * (1.0 + 0x1p-40)) return true;
  if (inter < t * (1.0 - 0x1p-40)) return false;
  return inter / d > thr;
}

__global__ void __launch_bounds__(1024) select_nms_kernel(const unsigned long long* __restrict__ keys, int n, const IBox* __restrict__ iboxes,
                                                          double thr, int max_boxes, int* __restrict__ out_count,
                                                          long long* __restrict__ out_boxes, float* __restrict__ out_probs,
                                                          const float* __restrict__ probs_src, int probs_stride, int probs_div) {
  __shared__ unsigned long long skeys[kCap];
  __shared__ unsigned int hist[kDigits];
  __shared__ IBox pick_box[kMaxPicks];
  __shared__ int pick_area[kMaxPicks];
  __shared__ int pick_idx[kMaxPicks];
  __shared__ IBox cand_box[64];
  __shared__ int cand_area[64];
  __shared__ int cand_idx[64];
  __shared__ unsigned int sup[2];
  __shared__ unsigned long long mat[64];
  __shared__ int s_npicks, s_cnt;
  __shared__ unsigned long long s_lo, s_prefix;
  __shared__ int s_remaining, s_refine;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_npicks = 0;
  unsigned long long upper = ~0ull;                 // keys >= upper have been examined (exclusive bound of the next band)
  bool first = true;
  __syncthreads();

  while (true) {
    // ---------------- radix select: lower bound `lo` of the best <= kCap keys below `upper`
    if (tid == 0) { s_lo = 0ull; s_prefix = 0ull; s_remaining = kCap; s_refine = 1; }
    __syncthreads();
    for (int level = 0; level < 6; ++level) {
      if (!s_refine) break;                                   // uniform: read after a barrier
      const int hi_bit = 64 - kDigitBits * level;             // bits [hi_bit-1 : shift] are this level's digit
      const int shift = hi_bit - kDigitBits > 0 ? hi_bit - kDigitBits : 0;
      const int width = hi_bit - shift;                       // 11, the last level 9
      const unsigned long long prefix = s_prefix;
      for (int i = tid; i < kDigits; i += 1024) hist[i] = 0u;
      __syncthreads();
      // 8 independent loads in flight per thread: one memory round trip per 8192 keys instead of one per 1024
      for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
        unsigned long long kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = i0 + u * 1024 < n ? keys[i0 + u * 1024] : 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned long long k = kk[u];
          bool todo = k != 0ull && (first || k < upper) && (level == 0 || (k >> hi_bit) == prefix);
          const unsigned digit = (unsigned)(k >> shift) & ((1u << width) - 1u);
          // Scores cluster (the coarse digits of most keys are equal): 64 lanes adding to ONE LDS word serialise.  Up to four
          // rounds of "the first pending lane's digit: one add of the number of lanes that share it", the rest lane by lane.
#pragma unroll 1
          for (int round = 0; round < 4; ++round) {
            const unsigned long long pend = __ballot(todo);
            if (pend == 0ull) break;
            const unsigned d0 = (unsigned)__shfl((int)digit, __ffsll((long long)pend) - 1, 64);
            const unsigned long long same = __ballot(todo && digit == d0);
            if (todo && digit == d0) {
              if ((unsigned)lane == (unsigned)(__ffsll((long long)same) - 1)) atomicAdd(&hist[d0], (unsigned)__popcll(same));
              todo = false;
            }
          }
          if (todo) atomicAdd(&hist[digit], 1u);
        }
      }
      __syncthreads();
      if (wave == 0) {
        // digits from the top: take whole digits while they fit into what is left of the band
        const int nd = 1 << width, per = nd / 64;             // 32 (or 8) consecutive digits per lane, lane 63 holds the top ones
        unsigned int mine = 0;
        for (int q = 0; q < per; ++q) mine += hist[lane * per + q];
        // suffix sum over lanes (lane L gets the count of all digits in lanes > L)
        unsigned int above = 0;
        for (int l = 63; l > 0; --l) {
          const unsigned int v = __shfl(mine, l, 64);
          if (lane < l) above += v;
        }
        const int remaining = s_remaining;
        // the lane whose block contains the cut: above <= remaining < above + mine  (or the lowest lane if everything fits)
        const bool fits_all = above + mine <= (unsigned)remaining;
        const unsigned long long m_cut = __ballot(above <= (unsigned)remaining && !fits_all);
        if (m_cut == 0ull) {                                  // every key of this prefix fits
          if (lane == 0) {
            const unsigned int total = above + mine;
            s_remaining = remaining - (int)total;
            s_lo = level == 0 ? 0ull : (prefix << hi_bit);     // the whole prefix
            s_refine = 0;
          }
        } else {
          const int cl = 63 - __clzll((long long)m_cut);     // highest lane that still satisfies above <= remaining: the cut lane
          if (lane == cl) {
            unsigned int acc = above;
            int d = (lane + 1) * per;                          // first digit above this lane's block
            for (int q = per - 1; q >= 0; --q) {
              const unsigned int h = hist[lane * per + q];
              if (acc + h > (unsigned)remaining) break;
              acc += h;
              d = lane * per + q;
            }
            // digits >= d are taken whole; digit d-1 (if any keys) is refined at the next level
            s_remaining = remaining - (int)acc;
            s_lo = ((level == 0 ? 0ull : (prefix << width)) | (unsigned long long)d) << shift;
            const bool can_refine = d > 0 && shift > 0 && (remaining - (int)acc) > 0;
            s_refine = can_refine ? 1 : 0;
            s_prefix = (level == 0 ? 0ull : (prefix << width)) | (unsigned long long)(d - 1);
          }
        }
      }
      __syncthreads();
    }
    const unsigned long long lo = s_lo;
    // ---------------- gather the band [lo, upper) into LDS
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
      unsigned long long kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kk[u] = i0 + u * 1024 < n ? keys[i0 + u * 1024] : 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned long long k = kk[u];
        const bool take = k != 0ull && k >= lo && (first || k < upper);
        const unsigned long long m = __ballot(take);            // one counter update per wave, slots by rank inside the wave
        if (m != 0ull) {
          int base = 0;
          if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&s_cnt, __popcll(m));
          base = __shfl(base, __ffsll((long long)m) - 1, 64);
          const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
          if (take && slot < kCap) skeys[slot] = k;
        }
      }
    }
    __syncthreads();
    const int cnt = min(s_cnt, kCap);
    if (cnt == 0) break;                                       // nothing left below `upper`
    int P = 64;
    while (P < cnt) P <<= 1;
    for (int i = cnt + tid; i < P; i += 1024) skeys[i] = 0ull;
    __syncthreads();
    // ---------------- bitonic sort, descending
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P; i += 1024) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = skeys[i], b = skeys[ixj];
            const bool desc = (i & k) == 0;
            if (desc ? a < b : a > b) { skeys[i] = b; skeys[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
    // ---------------- greedy NMS over the band, 64 candidates at a time (all data in LDS except the box gather)
    int pre_id = 0;
    IBox pre_box = {0, 0, 0, 0};
    auto fetch = [&](int base) {
      if (tid < 64 && base + tid < cnt) {
        pre_id = (int)(skeys[base + tid] & 0xFFFFFFFFull);
        pre_box = iboxes[pre_id];
      }
    };
    fetch(0);
    bool full = false;
    for (int base = 0; base < cnt; base += 64) {
      const int nc = min(64, cnt - base);
      if (tid < 64) {
        if (tid < nc) {
          cand_idx[tid] = pre_id;
          cand_box[tid] = pre_box;
          cand_area[tid] = ((int)pre_box.x2 - (int)pre_box.x1) * ((int)pre_box.y2 - (int)pre_box.y1);
        }
        if (tid < 2) sup[tid] = 0u;
      }
      fetch(base + 64);
      __syncthreads();
      const int npicks = s_npicks;
      {
        bool dead = false;
        if (lane < nc) {
          const IBox c = cand_box[lane];
          const int ca = cand_area[lane];
          for (int p = wave; p < npicks && !dead; p += 16) dead = suppresses_int(pick_box[p], pick_area[p], c, ca, thr);
        }
        const unsigned long long m = __ballot(dead);
        if (lane == 0 && m) {
          atomicOr(&sup[0], (unsigned int)(m & 0xFFFFFFFFull));
          atomicOr(&sup[1], (unsigned int)(m >> 32));
        }
      }
      for (int j = wave; j < 64; j += 16) {
        bool hit = false;
        if (j < nc && lane < nc && lane > j) hit = suppresses_int(cand_box[j], cand_area[j], cand_box[lane], cand_area[lane], thr);
        const unsigned long long m = __ballot(hit);
        if (lane == 0) mat[j] = m;
      }
      __syncthreads();
      if (tid == 0) {
        unsigned long long alive = ~(((unsigned long long)sup[1] << 32) | sup[0]);
        if (nc < 64) alive &= (1ull << nc) - 1ull;
        int np = npicks;
        while (alive != 0ull && np < max_boxes) {
          const int j = __ffsll((long long)alive) - 1;
          pick_box[np] = cand_box[j];
          pick_area[np] = cand_area[j];
          pick_idx[np] = cand_idx[j];
          ++np;
          alive &= ~mat[j];
          alive &= ~(1ull << j);
        }
        s_npicks = np;
      }
      __syncthreads();
      if (s_npicks >= max_boxes) { full = true; break; }
    }
    if (full || lo == 0ull) break;                             // lo == 0: the band reached down to the smallest key
    upper = lo;
    first = false;
    __syncthreads();
  }
  __syncthreads();
  const int np = s_npicks;
  if (tid == 0) *out_count = np;
  for (int i = tid; i < np; i += blockDim.x) {
    const IBox b = pick_box[i];
    out_boxes[4 * i + 0] = (long long)b.x1;
    out_boxes[4 * i + 1] = (long long)b.y1;
    out_boxes[4 * i + 2] = (long long)b.x2;
    out_boxes[4 * i + 3] = (long long)b.y2;
    if (out_probs) {
      const int id = pick_idx[i];
      const int a = id / probs_div, pix = id - a * probs_div;
      out_probs[i] = probs_src[(size_t)pix * probs_stride + a];
    }
  }
}

struct WsLayout {
  unsigned long long* keys_in;
  unsigned long long* keys_out;
  double4* boxes;
  int* counters;      // [0] n_valid, [1] malformed
  void* temp;
  size_t temp_bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

WsLayout carve(void* ws, int64_t n) {
  char* p = (char*)ws;
  WsLayout L;
  L.keys_in = (unsigned long long*)p;  p += align_up((size_t)n * 8, 256);
  L.keys_out = (unsigned long long*)p; p += align_up((size_t)n * 8, 256);
  L.boxes = (double4*)p;               p += align_up((size_t)n * 32, 256);
  L.counters = (int*)p;                p += 256;
  L.temp = p;
  L.temp_bytes = (size_t)n * 16 + (4u << 20);
  return L;
}

int sort_desc(radnet_ctx* ctx, const WsLayout& L, int64_t n) {
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_keys_desc(nullptr, need, L.keys_in, L.keys_out, (size_t)n, 0, 64, ctx->stream);
  if (e != hipSuccess) RADNET_FAIL(ctx, RADNET_ERR_HIP, "radix sort size query: %s", hipGetErrorString(e));
  if (need > L.temp_bytes) RADNET_FAIL(ctx, RADNET_ERR_ARG, "radix sort needs %zu bytes of scratch, workspace has %zu", need, L.temp_bytes);
  e = rocprim::radix_sort_keys_desc(L.temp, need, L.keys_in, L.keys_out, (size_t)n, 0, 64, ctx->stream);
  if (e != hipSuccess) RADNET_FAIL(ctx, RADNET_ERR_HIP, "radix sort: %s", hipGetErrorString(e));
  return RADNET_OK;
}

}  // namespace

extern "C" uint64_t radnet_proposals_ws_bytes(int64_t n) {
  return (uint64_t)(align_up((size_t)n * 8, 256) * 2 + align_up((size_t)n * 32, 256) + 256 + (size_t)n * 16 + (4u << 20));
}

extern "C" int radnet_rpn_to_roi(radnet_ctx* ctx, const float* pred, int32_t ld_pred, int32_t rows, int32_t cols, int32_t a,
                                 const double* anchor_wh_host, double std_scaling, int32_t use_regr, double overlap_thresh,
                                 int32_t max_boxes, int64_t* out_boxes, float* out_probs, int32_t* out_count, void* ws) {
  if (!ctx || !pred || !anchor_wh_host || !out_boxes || !out_count || !ws) return RADNET_ERR_ARG;
  if (a < 1 || a > 32) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "rpn_to_roi: %d anchors per location (max 32)", a);
  if (max_boxes < 1 || max_boxes > kMaxPicks) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "rpn_to_roi: max_boxes %d (max %d)", max_boxes, kMaxPicks);
  if (ld_pred < 5 * a) RADNET_FAIL(ctx, RADNET_ERR_ARG, "rpn_to_roi: ld_pred %d < 5*A", ld_pred);
  const int64_t n = (int64_t)rows * cols * a;
  WsLayout L = carve(ws, n);
  DecodeArgs g{};
  g.pred = pred; g.ld = ld_pred; g.rows = rows; g.cols = cols; g.a = a; g.std_scaling = std_scaling; g.use_regr = use_regr;
  for (int i = 0; i < a; ++i) { g.aw[i] = anchor_wh_host[2 * i]; g.ah[i] = anchor_wh_host[2 * i + 1]; }
  // Alternative path, RADNET_PROPOSALS_SELECT=1 (use_regr: integer boxes; rows, cols < 32768; thr > 0): decode + ONE workgroup
  // that selects, sorts and suppresses band by band -- 2 launches instead of 10 and no vendor sort, bit-identical results
  // (tests/test_gpu_kernels.py), but measured SLOWER alone on the chip at 1000x600 (28 728 candidates, 300 picks after ~2 400
  // examined): radix select 78 us + LDS bitonic sort of 4 096 keys 67 us against rocPRIM's 45 us for the full sort, and the
  // greedy scan itself -- 38 chunks of 64 candidates at ~5 us, three barriers and a serial resolve each -- costs the same
  // ~200 us with integer arithmetic as with fp64 (it is latency-, not arithmetic-bound).  So the full sort stays the default.
  const char* sel_env = getenv("RADNET_PROPOSALS_SELECT");      // read per call: the tests and tools/proposals_timing.py switch it in one process
  const bool select_path = sel_env != nullptr && strcmp(sel_env, "1") == 0;
  if (use_regr && rows < 32768 && cols < 32768 && overlap_thresh > 0.0 && select_path) {
    IBox* ib = reinterpret_cast<IBox*>(L.boxes);          // the fp64 box area of the workspace holds the packed boxes instead
    hipLaunchKernelGGL(decode_kernel, dim3(radnet_cdiv(n, 256)), dim3(256), 0, ctx->stream, g, (double4*)nullptr, ib, L.keys_in, (int*)nullptr);
    RADNET_CHECK_LAUNCH(ctx, "decode");
    hipLaunchKernelGGL(select_nms_kernel, dim3(1), dim3(1024), 0, ctx->stream, L.keys_in, (int)n, ib, overlap_thresh, max_boxes, out_count,
                       (long long*)out_boxes, out_probs, pred, ld_pred, rows * cols);
    RADNET_CHECK_LAUNCH(ctx, "select_nms");
    return RADNET_OK;
  }
  RADNET_CHECK_HIP(ctx, hipMemsetAsync(L.counters, 0, 256, ctx->stream));
  hipLaunchKernelGGL(decode_kernel, dim3(radnet_cdiv(n, 256)), dim3(256), 0, ctx->stream, g, L.boxes, (IBox*)nullptr, L.keys_in, L.counters);
  RADNET_CHECK_LAUNCH(ctx, "decode");
  int rc = sort_desc(ctx, L, n);
  if (rc != RADNET_OK) return rc;
  hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), 0, ctx->stream, L.keys_out, L.counters, (const int*)nullptr, L.boxes, overlap_thresh,
                     max_boxes, (int*)nullptr, out_count, (long long*)out_boxes, out_probs, pred, ld_pred, rows * cols);
  RADNET_CHECK_LAUNCH(ctx, "nms");
  return RADNET_OK;
}

extern "C" int radnet_nms(radnet_ctx* ctx, const double* boxes, const float* probs, int32_t n, double overlap_thresh, int32_t max_boxes,
                          int32_t* out_idx, int32_t* out_count, void* ws) {
  if (!ctx || !out_idx || !out_count || !ws) return RADNET_ERR_ARG;
  if (max_boxes < 1 || max_boxes > kMaxPicks) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "nms: max_boxes %d (max %d)", max_boxes, kMaxPicks);
  if (n <= 0) {      // rpn.py:391-392: no boxes -> empty result
    RADNET_CHECK_HIP(ctx, hipMemsetAsync(out_count, 0, sizeof(int32_t), ctx->stream));
    return RADNET_OK;
  }
  if (!boxes || !probs) return RADNET_ERR_ARG;
  if (((uintptr_t)boxes & 31)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "nms: boxes must be 32-byte aligned");
  WsLayout L = carve(ws, n);
  RADNET_CHECK_HIP(ctx, hipMemsetAsync(L.counters, 0, 256, ctx->stream));
  hipLaunchKernelGGL(nms_keys_kernel, dim3(radnet_cdiv(n, 256)), dim3(256), 0, ctx->stream, boxes, probs, n, L.keys_in, L.counters, L.counters + 1);
  RADNET_CHECK_LAUNCH(ctx, "nms_keys");
  int rc = sort_desc(ctx, L, n);
  if (rc != RADNET_OK) return rc;
  hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), 0, ctx->stream, L.keys_out, L.counters, L.counters + 1, (const double4*)boxes,
                     overlap_thresh, max_boxes, out_idx, out_count, (long long*)nullptr, (float*)nullptr, (const float*)nullptr, 0, 1);
  RADNET_CHECK_LAUNCH(ctx, "nms");
  return RADNET_OK;
}
