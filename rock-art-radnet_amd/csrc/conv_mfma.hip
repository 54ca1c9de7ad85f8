// Convolution as implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Stands in for keras Conv2D / TimeDistributed(Conv2D) + FixedBatchNormalization + Add + Activation
// of the reference graph (base_models/resnet50.py:41-147,183-186; rpn.py:41-64;
// FixedBatchNormalization.py:59-85) and for the TF autodiff gradients of those layers.
//
// Design (MI355X-first):
//   * NHWC activations, weights [K=(kh,kw,c)][N]: the im2col matrix is never materialised; each workgroup gathers
//     its A tile (BM output pixels x 32 k) straight from the activation tensor with 16-byte buffer loads (4
//     consecutive channels); padding taps, ragged rows / columns and tiles past the end are an out-of-range offset
//     that the hardware answers with zeros -- no branch anywhere in the K loop.
//   * 4 wavefronts in a 2x2 arrangement (optionally 8: two grids halving every K tile); each wave owns a
//     (BM/2)x(BN/2) block of the output as 32x32 MFMA tiles in accumulator registers for the whole K loop.  32-row (32-column)
//     tiles have one wave row (column): the waves left over split every K tile between them and are summed through LDS
//     (round 4: M = 980 / 2 394 / 160-row problems fill the chip without K slices).
//   * LDS double buffer.  The gathered operand (A; both operands in dgrad) is ROW-major [row][36]: written with one
//     ds_write_b128 per 4-k chunk, read 4 k at a time with ds_read_b128 -- the k order inside a tile is free as long
//     as both operands agree (mfma_tile_rows).  Forward weights stay k-major [k][BN+4] (conflict-free ds_read_b32).
//     wgrad keeps both operands reduction-major (mfma_tile).
//   * two register stages of global loads (tile t+2 in flight while t is multiplied); loads, address arithmetic and
//     LDS stores are single operations dealt out BETWEEN the MFMA steps: a wave cannot overlap its own VALU / memory
//     instructions with its own MFMAs (tools/mfma_loop_probe.hip), so what counts is the non-MFMA instruction count
//     per tile and having other waves on the SIMD.
//   * epilogue fused through buffer descriptors: frozen-BN scale/shift (+bias), residual add, ReLU / sigmoid; for
//     dgrad the residual-path gradient add and the producer's ReLU mask.
//   * split-K inside the launch: slices write sc1 (write-through) slabs, take a ticket from a per-tile arrival
//     counter, the last arriver reduces in slice order and applies the epilogue.
//   * fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s peak), 16x less than bf16, so LDS and L2 bandwidth are far
//     from limiting; what matters is filling 256 CUs at batch 1 -- tile shape, K slices, workgroup order and waves per
//     workgroup are measured per problem shape (run_igemm).
#include "radnet_internal.h"
#include "radnet_wino4.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <set>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One launch, timed from its own dispatch when an event pair is armed (bench.py's roofline leg), plain otherwise.
#define RADNET_LAUNCH(kernel, grid, block, shmem, st, e0, e1, ...)                                                  \
  do {                                                                                                              \
    if (e0) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, e0, e1, 0, __VA_ARGS__);                          \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                           \
  } while (0)

namespace {

constexpr int BK = 32;          // reduction depth per LDS tile
constexpr int NTHREADS = 256;
#ifndef RADNET_CHAINS
#define RADNET_CHAINS 2
#endif
constexpr int kChainsSmallTile = RADNET_CHAINS;   // K-interleaved accumulator sets of the 64x64 / 128x64 / 64x128 tiles

struct GemmArgs {
  const float* x;        // gathered activation tensor (NHWC)
  const float* w;        // B operand base
  float* y;              // output [M][ldy]
  const float* scale;    // epilogue per-column scale
  const float* shift;    // epilogue per-column shift
  const float* addend;   // epilogue addend [M][ld_add]
  const float* mask;     // epilogue mask   [M][ld_mask] (zero where <= 0)
  const float* in_scale; // per-gathered-channel factor (C entries) or null
  float* partial;        // split-K partial sums [split][M][N] (null = direct epilogue)
  int H, W, C;           // gathered tensor geometry
  int OH, OW;            // output spatial geometry
  int KW, npos;          // kernel width, kh*kw
  int stride, pad_t, pad_l;
  int M, N, K;           // GEMM sizes, K = npos*C
  int ldw, ldy, ld_add, ld_mask;
  int act, act_cols;
  int flip;              // dgrad: kernel position flipped (npos-1-pos)
  int cin_fwd;           // dgrad B addressing: forward input channels (= N here)
  const int* units;      // work-unit table (8 ints per unit: tile_m, tile_n, kt_begin, kt_end, slot, pad..) or null
  unsigned long long magic_ohow, magic_ow;
  int OHOW;
  unsigned x_bytes, w_bytes;   // extents for the buffer descriptors
  unsigned y_bytes, add_bytes, mask_bytes;
  int batch;                   // > 1: blockIdx.z selects one of `batch` independent GEMMs (plain launches only)
  long long x_bstride, w_bstride, y_bstride;   // floats between consecutive problems of a batch
  unsigned* counters;          // K-split launches: arrival counter per output tile (zero outside a launch)
  unsigned long long* stamps;  // diagnostic build only (RADNET_DIAG_STAMPS): 8 words per workgroup
  int xcd_batch;               // batched launch: workgroups renumbered so that each XCD runs a contiguous run of (problem, tile)s
  int zper;                    // persistent batched launch (PERSIST kernels): consecutive problems one workgroup runs, blockIdx.z = group
};

#ifdef RADNET_DIAG_STAMPS
#define RADNET_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RADNET_STAMP(var)
#endif

__device__ __forceinline__ int div_magic(int m, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)m * magic) >> 40);
}

// Buffer loads: the 128-bit resource descriptor carries the tensor's byte size, and the hardware returns 0 for
// any offset beyond it.  Padding taps, rows past M and columns past N are therefore expressed as the offset
// kOOB instead of a branch: all of a tile's loads issue back to back and are waited for once, at the LDS store.
// 2^31, not 2^32-1: every descriptor here covers < 2 GiB (checked by the launchers), so offset + 16 can neither
// wrap around in 32-bit range arithmetic nor fall inside the buffer.
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
// voffset (per lane) + soffset (wave-uniform, an SGPR): the hardware adds them and range-checks the SUM without 32-bit
// wrap-around (tools/soffset_probe.hip: kOOB in either operand reads 0), so the uniform part of an address costs no VALU
__device__ __forceinline__ float4 buf_load4s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float buf_load1s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0));
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int2 buf_load2i(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
  return make_int2((int)v.x, (int)v.y);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
// sc1 (aux 16): write-through store / L1-bypassing agent-coherent load, for data handed to another workgroup in-launch
__device__ __forceinline__ float buf_load1_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 16));
}
__device__ __forceinline__ float4 buf_load4_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store4_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 f = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), r, (int)off, 0, 16);
}
__device__ __forceinline__ void buf_store1_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 16);
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}

__device__ __forceinline__ float f4_comp(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// One 32-deep K tile: 16 MFMA steps of depth 2, operand fragments prefetched from LDS into a register ring (one
// wave per SIMD has nobody else to hide the LDS latency behind).
// `staging(s)` is the caller's slice of operand staging for step s -- a global load of tile t+2 with its address
// arithmetic, or an LDS store of tile t+1 -- written HERE, between the MFMA steps, because that is where it has to
// execute: an MFMA occupies the matrix pipe for 64 cycles after it issues and the wave can issue independent VALU /
// memory instructions meanwhile.  With all loads in front of the first MFMA and all stores behind the last one
// (which is also where hipcc's scheduler moves them when it is free to), a lone wave per SIMD ran a 64x64 tile in
// 2070 cycles instead of 1024 (tools/stamp_probe.py).  sched_barrier(0) after every step keeps the slices in place.
struct NoStaging {
  __device__ __forceinline__ void operator()(int) const {}
};
//
// Dependent MFMAs: with one 32x32 accumulator per wave (64x64 tile) every MFMA waits for the previous one to
// retire, and a lone wave per SIMD ran at ~120 cycles per MFMA instead of 64.  CH > 1 keeps CH accumulator sets,
// step s adding into set s % CH (the caller sums the sets after the K loop), so CH*TM*TN MFMAs are independent.
// Fragments are fetched TWO steps ahead (3-slot register ring): an LDS read takes about as long as one MFMA.
template <int TM, int TN, int CH, typename Staging>
__device__ __forceinline__ void mfma_tile(const float* sA, const float* sB, int pitchA, int pitchB, int a_off, int b_off,
                                          f32x16 (&acc)[CH][TM][TN], Staging staging) {
  constexpr int kSteps = BK / 2;
  float a[3][TM], b[3][TN];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[p][i] = sA[2 * p * pitchA + a_off + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[p][j] = sB[2 * p * pitchB + b_off + j * 32];
  }
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const int cur = s % 3, nxt = (s + 2) % 3;
    if (s + 2 < kSteps) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[nxt][i] = sA[(2 * s + 4) * pitchA + a_off + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[nxt][j] = sB[(2 * s + 4) * pitchB + b_off + j * 32];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[s % CH][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[s % CH][i][j], 0, 0, 0);
    staging(s);
    // order inside the step: next step's LDS reads, this step's MFMAs, then the staging slice in their shadow
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same tile for the forward / dgrad kernel, whose gathered operand arrives as 4 consecutive k of one row: it is
// kept ROW-major in LDS ([row][kRowPitch], one ds_write_b128 per chunk -- the transposed ds_write_b32 stores it
// replaces cost ~85 cycles of MFMA time each, tools/stamp_probe.py) and its fragments are read 4 k at a time with
// ds_read_b128.  That works because the order of k inside a tile is free as long as A and B agree: MFMA step
// s = 4q + j multiplies k = 8q + j in lanes 0-31 and k = 8q + 4 + j in lanes 32-63, so lane (row, h) reads the 16
// bytes at k = 8q + 4h once per q and uses component j in step 4q + j; the k-major operand (forward weights, [k][n])
// reads row 8q + 4h + j.  kRowPitch = 36 words: a ds_read_b128 lane group (16 lanes, rows {0-3,12-15,20-27} + 4g)
// lands on 16 distinct 4-bank sets, and the 8-lane groups of the ds_write_b128 cover 32 consecutive words.
constexpr int kRowPitch = BK + 4;
// KSTEPS = 16: the wave multiplies the whole 32-deep tile; KSTEPS = 8: half of it (8-wave workgroups: waves 4-7 take
// k = 16..31, the caller shifts a_off / b_off accordingly and sums the two halves after the K loop).
template <int TM, int TN, int CH, bool B_ROWMAJOR, int KSTEPS, typename Staging>
__device__ __forceinline__ void mfma_tile_rows(const float* sA, const float* sB, int pitchB, int a_off, int b_off,
                                               f32x16 (&acc)[CH][TM][TN], Staging staging) {
  constexpr int kSteps = KSTEPS;
  float4 af[2][TM], bq[2][TN];
  float bf[3][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const float4*>(sA + a_off + i * 32 * kRowPitch);
  if (B_ROWMAJOR) {
#pragma unroll
    for (int j = 0; j < TN; ++j) bq[0][j] = *reinterpret_cast<const float4*>(sB + b_off + j * 32 * kRowPitch);
  } else {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[p][j] = sB[p * pitchB + b_off + j * 32];
  }
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const int q = s >> 2, c = s & 3;
    const bool group_reads = c == 1 && q + 1 < kSteps / 4;   // next group's 16-byte fragments, three steps ahead of their first use
    const bool step_reads = !B_ROWMAJOR && s + 2 < kSteps;
    if (group_reads) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[(q + 1) & 1][i] = *reinterpret_cast<const float4*>(sA + a_off + i * 32 * kRowPitch + 8 * (q + 1));
      if (B_ROWMAJOR) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[(q + 1) & 1][j] = *reinterpret_cast<const float4*>(sB + b_off + j * 32 * kRowPitch + 8 * (q + 1));
      }
    }
    if (step_reads) {
      const int s2 = s + 2;
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[s2 % 3][j] = sB[(8 * (s2 >> 2) + (s2 & 3)) * pitchB + b_off + j * 32];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[s % CH][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_comp(af[q & 1][i], c), B_ROWMAJOR ? f4_comp(bq[q & 1][j], c) : bf[s % 3][j],
                                                                 acc[s % CH][i][j], 0, 0, 0);
    staging(s);
    // order inside the step: the LDS reads issued here, this step's MFMAs, then the staging slice in their shadow
    if (group_reads && (B_ROWMAJOR || step_reads)) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    else if (group_reads) __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
    else if (step_reads) __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- forward / dgrad kernel -------------------------------------------------------------------------
// BMODE 0: B is [K][ldw] row-major (forward).  BMODE 1: B element (k=(pos,co), n=ci) lives at
//          w[((flip(pos)*cin_fwd + ci) * ldw) + co]  (dgrad: same weight buffer, read transposed).
// SMALLC : C == 4 (stem with the image padded to 4 channels): one 4-float chunk per kernel position.
// WAVES  : 4 = the 2x2 wave grid multiplies whole K tiles; 8 = two such grids share the tile, waves 4-7 taking the
//          second half of every 32-deep K tile (split-K INSIDE the workgroup: same LDS tile, half the staging work per
//          thread, twice the waves per SIMD for the same number of workgroups -- a wave cannot hide its own staging
//          instructions under its own MFMAs, another wave's can).  The halves are summed through LDS after the loop.
// LDS floats of one workgroup of conv_igemm_body (two buffers of an A and a B tile)
template <int BM, int BN, int BMODE>
constexpr int igemm_lds_floats() { return 2 * (BM * kRowPitch + ((BMODE == 0) ? BK * (BN + 4) : BN * kRowPitch)); }
// PERSIST kernels sum their K parts while the staging buffers already hold the next problem's first tile: own scratch behind them
template <int BM, int BN, int WAVES>
constexpr int igemm_persist_scratch_floats() {
  constexpr int WG = (BM >= 64 ? 2 : 1) * (BN >= 64 ? 2 : 1), KH = WAVES / WG;
  return (KH - 1) * BM * BN;
}

// ---- fused bottleneck tail (round 4) ----------------------------------------------------------------------------------
// A ResNet identity / conv block is 1x1 reduce -> 3x3 -> 1x1 expand (+ shortcut, ReLU) (resnet50.py:41-71, 74-128).  In stage 2
// (C = 64 / 256 on the 150x250 map) the two pointwise convs are memory-shaped launches: 1.2 GF each for 48 MB moved, 50-66 TFLOP/s,
// and three launches per block.  Cut the chain in front of the 3x3 instead of behind it and nothing needs a halo: a workgroup that
// holds a [BM rows x 64] tile of the 3x3 output holds ALL of that layer's channels for its rows, so it can go on, for the same rows,
//   y[rows][N2]  = relu(t2 . W2 * sc2 + sh2 + shortcut[rows][N2])          (branch2c, Add, Activation)
//   t'[rows][64] = relu(y . W3 * sc3 + sh3)                                 (the NEXT block's branch2a), optional
// with t2 and y passed between the three GEMMs through LDS; t2 is never written to memory, y once, and the next block's 1x1 input
// is not read back.  The tail GEMMs run on the same 2x2 (2x1) wave grid as the 3x3: A fragments from LDS in the K loop's own row-major
// layout (mfma_tile_rows), B fragments straight from the 64 KB weight matrices in L2 (every workgroup reads the same ones; one dword per
// lane and MFMA step, no staging, no barrier).  Frozen layers only: the block's intermediate activations do not exist afterwards.
struct TailArgs {
  const float* w2; const float* sc2; const float* sh2; const float* add; float* y;     // expand: [64][ldw2], columns N2 (multiple of 64)
  const float* w3; const float* sc3; const float* sh3; float* t;                        // next reduce: [N2][ldw3] -> 64 columns, or null
  int N2, ldw2, ldy2, ld_add2, ldw3, ldt;
  unsigned w2_bytes, w3_bytes, y_bytes, add_bytes, t_bytes;
};
template <int BM>
constexpr int bneck_lds_floats() { return 4 * BM * (BK + 4); }     // the 3x3 tile and one 64-column chunk of y, each as two 32-deep A tiles

// one 64-deep GEMM step of the tail: acc += A[rows][64] (LDS, two row-major 32-deep tiles) . B (fragments in registers)
// (ONE accumulator: the tail's registers decide how many workgroups share a CU, and those other waves fill the MFMA pipe between two
// dependent steps of this one)
__device__ __forceinline__ void bneck_mfma64(const float* sA, int buf_floats, int a_off, const float (&bw)[32], f32x16& acc) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 af = *reinterpret_cast<const float4*>(sA + half * buf_floats + a_off + 8 * q);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int s = 16 * half + 4 * q + c;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_comp(af, c), bw[s], acc, 0, 0, 0);
      }
    }
}
// B fragments of such a step: lane (hi, l31) multiplies k = 32 half + 8 q + 4 hi + c in step (half, q, c) -- the order mfma_tile_rows uses
__device__ __forceinline__ void bneck_load_b(__amdgpu_buffer_rsrc_t rw, unsigned voff, unsigned row0, unsigned ld4, bool live, float (&bw)[32]) {
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const unsigned k = 32u * (s >> 4) + 8u * ((s & 15) >> 2) + (s & 3);
    bw[s] = buf_load1s(rw, voff, live ? (row0 + k) * ld4 : kOOB);
  }
}

template <int BM, int WAVES, bool HAS3>
__device__ __forceinline__ void bneck_tail(const GemmArgs& g, const TailArgs& tz, float* __restrict__ lds, const f32x16& acc, const int m0) {
  constexpr int WM = BM >= 64 ? 2 : 1, WN = 2;
  static_assert(WAVES == WM * WN, "the tail runs on the 3x3's wave grid, no K parts");
  constexpr int kBufT = BM * kRowPitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  float* sT = lds;
  float* sY = lds + 2 * kBufT;
  const int col = wn * 32 + l31;                                   // this lane's column inside a 64-column chunk
  const int own_off = (wm * 32 + 4 * hi) * kRowPitch + l31;       // accumulator register 0 of this lane in a 32-deep A tile (k = its column)
  {
    const float sc = g.scale ? g.scale[col] : 1.f, sh = g.shift ? g.shift[col] : 0.f;
    float* dst = sT + wn * kBufT + own_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * kRowPitch] = fmaxf(acc[r] * sc + sh, 0.f);
  }
  __syncthreads();
  const int a_off = (wm * 32 + l31) * kRowPitch + 4 * hi;
  const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(tz.w2, tz.w2_bytes), rw3 = make_rsrc(tz.w3, HAS3 ? tz.w3_bytes : 0u);
  const __amdgpu_buffer_rsrc_t radd = make_rsrc(tz.add, tz.add ? tz.add_bytes : 0u), ry = make_rsrc(tz.y, tz.y_bytes);
  const __amdgpu_buffer_rsrc_t rsc2 = make_rsrc(tz.sc2, tz.sc2 ? (unsigned)tz.N2 * 4u : 0u), rsh2 = make_rsrc(tz.sh2, tz.sh2 ? (unsigned)tz.N2 * 4u : 0u);
  const unsigned ldw2_4 = (unsigned)tz.ldw2 * 4u, ldw3_4 = (unsigned)tz.ldw3 * 4u, ldy4 = (unsigned)tz.ldy2 * 4u, lda4 = (unsigned)tz.ld_add2 * 4u;
  const unsigned bv2 = (unsigned)(4 * hi) * ldw2_4 + (unsigned)col * 4u;           // + 64 c columns, + k rows (SGPR part)
  const unsigned bv3 = (unsigned)(4 * hi) * ldw3_4 + (unsigned)col * 4u;           // + (64 c + k) rows
  const unsigned row0 = (unsigned)(m0 + wm * 32 + 4 * hi);
  const bool rows_ok = row0 < (unsigned)g.M;                       // rows past M further down fall off the descriptors' ends
  const bool has_sc2 = tz.sc2 != nullptr;
  f32x16 acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
  float bw2[32];
  bneck_load_b(rw2, bv2, 0u, ldw2_4, true, bw2);
  const int nchunks = tz.N2 >> 6;
  for (int c = 0; c < nchunks; ++c) {
    const unsigned ncol = (unsigned)(64 * c + col);
    const unsigned vy = rows_ok ? (row0 * (unsigned)tz.ldy2 + ncol) * 4u : kOOB;
    const unsigned va = rows_ok ? (row0 * (unsigned)tz.ld_add2 + ncol) * 4u : kOOB;
    float ad[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ad[r] = buf_load1s(radd, va, (unsigned)((r & 3) + 8 * (r >> 2)) * lda4);
    const float sc2r = buf_load1(rsc2, ncol * 4u), sh2 = buf_load1(rsh2, ncol * 4u);
    const float sc2 = has_sc2 ? sc2r : 1.f;
    float bw3[32];
    if (HAS3) bneck_load_b(rw3, bv3, 64u * (unsigned)c, ldw3_4, true, bw3);
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    bneck_mfma64(sT, kBufT, a_off, bw2, acc2);
    // the next chunk's expand fragments travel under this chunk's epilogue and reduce step (columns move by 256 bytes per chunk)
    bneck_load_b(rw2, bv2 + 256u * (unsigned)(c + 1), 0u, ldw2_4, c + 1 < nchunks, bw2);
    if (HAS3 && c > 0) __syncthreads();                            // every wave is done reading the previous chunk from sY
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned rr = (unsigned)((r & 3) + 8 * (r >> 2));
      const float v = fmaxf(acc2[r] * sc2 + sh2 + ad[r], 0.f);
      buf_store1(ry, vy + rr * ldy4, v);
      if (HAS3) sY[wn * kBufT + own_off + rr * kRowPitch] = v;
    }
    if (HAS3) {
      __syncthreads();
      bneck_mfma64(sY, kBufT, a_off, bw3, acc3);
    }
  }
  if (HAS3) {
    const __amdgpu_buffer_rsrc_t rt = make_rsrc(tz.t, tz.t_bytes);
    const float sc3 = tz.sc3 ? tz.sc3[col] : 1.f, sh3 = tz.sh3 ? tz.sh3[col] : 0.f;
    const unsigned vt = rows_ok ? (row0 * (unsigned)tz.ldt + (unsigned)col) * 4u : kOOB;
    const unsigned ldt4 = (unsigned)tz.ldt * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      buf_store1(rt, vt + (unsigned)((r & 3) + 8 * (r >> 2)) * ldt4, fmaxf(acc3[r] * sc3 + sh3, 0.f));
  }
}

// COH: the output is handed to other workgroups of the SAME launch (chain kernel): stores are write-through (sc1), as the
// split-K slabs are, so that a consumer on another XCD finds them in memory.
// PERSIST (batched launches, forward form, plain epilogue): the workgroup runs g.zper CONSECUTIVE problems of the batch on its output
// tile as one long K loop -- the loads of the next problem's first K tiles are issued under the last MFMA steps of the current one,
// the finished accumulators leave with fire-and-forget stores, and the workgroup pays ONE prologue and ONE drain instead of one per
// problem.  The 36 GEMMs of a Winograd layer have 4-8 K tiles each: as 432-720 one-tile workgroups they were all prologue and
// epilogue.
// FUSE (1 / 2): the accumulators do not leave through the epilogue but feed bneck_tail (2: with the next block's 1x1 reduce)
template <int BM, int BN, int BMODE, bool SMALLC, int WAVES, bool COH = false, bool PERSIST = false, int FUSE = 0>
__device__ __forceinline__ void conv_igemm_body(const GemmArgs& g, float* __restrict__ lds, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z,
                                                const unsigned grid_x, [[maybe_unused]] const TailArgs* tz = nullptr) {
  constexpr int NT = 64 * WAVES;
  // Wave grid over the output tile: 2x2 for tiles of 64 rows / columns and more, a single wave row (column) for the 32-row
  // (32-column) tiles; the waves left over split every 32-deep K tile between them (KH parts: the 8-wave form of the 64x64
  // tile has KH = 2, the 4-wave 32x64 tile too, the 4-wave 32x32 tile KH = 4) and are summed through LDS after the loop.
  constexpr int WM = BM >= 64 ? 2 : 1, WN = BN >= 64 ? 2 : 1, WG = WM * WN;
  constexpr int KH = WAVES / WG;
  static_assert(WAVES % WG == 0 && (KH == 1 || KH == 2 || KH == 4), "wave count does not cover the tile's wave grid");
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);     // 32x32 tiles per wave in each direction
  constexpr int PB = BN + 4;                    // forward weights: k-major [BK][PB]
  constexpr int A_ITERS = BM * 8 / NT;          // float4 chunks per thread (A)
  constexpr int B_ITERS = BN * 8 / NT;
  static_assert(A_ITERS >= 1 && B_ITERS >= 1 && A_ITERS * NT == BM * 8 && B_ITERS * NT == BN * 8, "tile too small for this many threads");
  constexpr int kRowStep = NT / 8;              // rows between a thread's consecutive chunks
  constexpr int kStepsW = (BK / 2) / KH;        // MFMA steps of one wave per K tile
  constexpr int kKPart = BK / KH;               // depth of a wave's part of the K tile
  // one LDS buffer: A row-major [BM][kRowPitch]; B k-major [BK][PB] (forward) or row-major [BN][kRowPitch] (dgrad)
  constexpr int kBufA = BM * kRowPitch;
  constexpr int kBufB = (BMODE == 0) ? BK * PB : BN * kRowPitch;
  float* sA0 = lds;
  float* sB0 = lds + 2 * kBufA;

  RADNET_STAMP(t_start);
#ifdef RADNET_DIAG_STAMPS
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_first = t_start, t_loop = t_start;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wgi = wave % WG;                    // place in the wave grid
  const int wm = wgi / WN, wn = wgi % WN;
  const int khalf = wave / WG;                  // which part of every K tile (0 when the wave grid takes all waves)
  // Work assignment.  Plain launch: one workgroup per output tile.  Unit-table launch (g.units != null): the host
  // cut the linearised (tile, k-tile) iteration space into near-equal chunks so every CU gets the same amount of
  // MFMA work whatever the tile count (stream-K style); a unit is (tile, k range, partial slot or -1).
  int m0, n0, unit_kb = 0, unit_ke = 0, slot = -1, slot0 = 0, n_slices = 1, tile_id = 0;
  if (g.units != nullptr) {
    const int4 u0 = reinterpret_cast<const int4*>(g.units)[2 * bid_x];
    const int4 u1 = reinterpret_cast<const int4*>(g.units)[2 * bid_x + 1];
    m0 = u0.x * BM; n0 = u0.y * BN; unit_kb = u0.z; unit_ke = u0.w;
    slot = u1.x; slot0 = u1.y; n_slices = u1.z; tile_id = u1.w;
  } else {
    m0 = bid_x * BM; n0 = bid_y * BN;
  }

  // ---- per-thread A rows: decode m -> (image, oh, ow) once
  const int a_kc = tid & 7;
  // a_base = byte offset of (image, ih0, iw0, channel 0), possibly negative (halo); a tap (kh, kw, ci) then adds one
  // per-tile offset, and only the two range compares remain per load (no multiplies in the K loop: v_mul_lo_u32
  // is a 16-cycle instruction).  Rows past M get ih0 far below zero, which fails the range compare of every tap.
  int a_base[A_ITERS], a_ih0[A_ITERS], a_iw0[A_ITERS];
  const int a_cbytes = (SMALLC ? 4 : g.C) * 4;
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int m = m0 + (tid >> 3) + kRowStep * i;
    const int mc = m < g.M ? m : 0;
    const int img = div_magic(mc, g.magic_ohow);
    const int rem = mc - img * g.OHOW;
    const int oh = div_magic(rem, g.magic_ow);
    const int ow = rem - oh * g.OW;
    a_ih0[i] = m < g.M ? oh * g.stride - g.pad_t : -(1 << 24);
    a_iw0[i] = ow * g.stride - g.pad_l;
    a_base[i] = ((img * g.H + (oh * g.stride - g.pad_t)) * g.W + a_iw0[i]) * a_cbytes;
  }

  // Channel-tiled layers (!SMALLC): everything that changes from K tile to K tile is wave-uniform -- the tap (kh, kw), the
  // first channel, the weight row -- and travels in the load's SGPR offset; what is left per lane is a constant offset and
  // ONE bit per tap ("this tap of this output row falls into the padding"), folded into bit 31 of the offset.  The
  // descriptor of x starts a_bias bytes early so that the per-lane part of a halo row is never negative.
  const unsigned a_bias = SMALLC ? 0u : (unsigned)((g.pad_t * g.W + g.pad_l) * g.C * 4);
  unsigned a_voff[A_ITERS], a_inv[A_ITERS];
  if (!SMALLC) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const bool row_ok = a_ih0[i] > -(1 << 23);
      unsigned inv = 0u;
      for (int p = 0, kh = 0, kw = 0; p < g.npos; ++p) {
        const bool ok = ((unsigned)(a_ih0[i] + kh) < (unsigned)g.H) & ((unsigned)(a_iw0[i] + kw) < (unsigned)g.W);
        inv |= (ok ? 0u : 1u) << p;
        if (++kw == g.KW) { kw = 0; ++kh; }
      }
      a_voff[i] = row_ok ? (unsigned)(a_base[i] + (int)a_bias + a_kc * 16) : 0u;
      a_inv[i] = row_ok ? inv : ~0u;
    }
  }

  // forward weights: byte offset of this thread's chunk (row kr, column n) in K tile 0; a tile adds BK rows
  int b_base[B_ITERS];
  bool b_nvalid[B_ITERS];
  unsigned b_voff[B_ITERS];
  const int b_tile_bytes = BK * g.ldw * 4;
  if (BMODE == 0) {
    constexpr int CPR = BN / 4;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int c = tid + NT * i;
      const int kr = c / CPR, n = n0 + (c - kr * CPR) * 4;
      b_base[i] = (kr * g.ldw + n) * 4;
      b_nvalid[i] = n < g.N;
      b_voff[i] = b_nvalid[i] ? (unsigned)b_base[i] : kOOB;
    }
  } else {
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {       // dgrad: row n of w^T (forward input channel), this thread's 4 channels
      const int n = n0 + (tid >> 3) + kRowStep * i;
      b_voff[i] = n < g.N ? ((unsigned)n * (unsigned)g.ldw + (unsigned)a_kc * 4u) * 4u : kOOB;
    }
  }

  const int nk_total = (g.K + BK - 1) / BK;
  // batched launch (radnet_gemm_batched): problem blockIdx.z of a strided batch, same geometry; PERSIST: problems bz .. bz + nz - 1
  const long long bz = g.batch > 1 ? (long long)bid_z * (PERSIST ? g.zper : 1) : 0;
  const int nz = PERSIST ? ((int)bz + g.zper <= g.batch ? g.zper : g.batch - (int)bz) : 1;
  const int kt_begin = g.units != nullptr ? unit_kb : 0;
  const int kt_end = g.units != nullptr ? unit_ke : nk_total * nz;

  // running position of the current K tile: kernel position pos = (kh, kw) and first channel ci0, k0 = pos*C + ci0
  int pos = 0, ci0 = 0, kh_run = 0, kw_run = 0;
  if (!SMALLC) {
    int k0 = kt_begin * BK;
    pos = k0 / g.C;
    ci0 = k0 - pos * g.C;
    kh_run = pos / g.KW;
    kw_run = pos - kh_run * g.KW;
  }

  // (PERSIST: the descriptors span the nz problems; rows past M and columns past N are out of range through their per-lane kOOB bit,
  // not through the extent, and K has no ragged tile -- the launcher checks C % BK == 0)
  const unsigned span_x = PERSIST ? (unsigned)((nz - 1) * g.x_bstride * 4) : 0u, span_w = PERSIST ? (unsigned)((nz - 1) * g.w_bstride * 4) : 0u;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(g.x + bz * g.x_bstride) - a_bias, g.x_bytes ? g.x_bytes + a_bias + span_x : 0u);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.w + bz * g.w_bstride, g.w_bytes + span_w);
  const bool has_in_scale = g.in_scale != nullptr;
  const __amdgpu_buffer_rsrc_t rscale = make_rsrc(g.in_scale, has_in_scale ? (unsigned)g.C * 4u : 0u);
  // Two register stages: the loads of tile t+2 are issued while tile t is being multiplied and tile t+1 waits in the
  // other stage, so a memory round trip (1-2 us when the line comes from the Infinity Cache or HBM) has TWO tile
  // times to complete.
  struct Stage {
    float4 a[A_ITERS], b[B_ITERS], s;
  };
  Stage st0, st1;
  st0.s = make_float4(1, 1, 1, 1);
  st1.s = make_float4(1, 1, 1, 1);

  // Operand staging, cut into single operations so that mfma_tile can deal them out between the MFMA steps.
  // Entirely branch-free (a tile past the end of this workgroup's K range, live == false, loads from kOOB -> 0, and
  // its LDS store writes zeros into the buffer nobody reads again): the K loop body is ONE basic block.
  constexpr int kLoadOps = A_ITERS + B_ITERS;                                   // one 16-byte buffer load each
  constexpr int kStoreOps = A_ITERS + B_ITERS;                                  // one ds_write_b128 each
  // state of the tile being loaded (tile_begin -> load_op)
  int t_kt = 0, t_kh = 0, t_kw = 0, t_fpos = 0, t_aoff = 0;
  bool t_live = false, t_kv = false;
  unsigned s_a = kOOB, s_b = kOOB, s_sh = 0;       // wave-uniform: SGPR offsets of the tile's A / B loads, tap -> bit-31 shift
  unsigned p_item_a = 0u, p_item_b = 0u;           // PERSIST: byte offset of the running problem in the spanning descriptors
  int p_kin = 0;                                   //          K tile inside the running problem

  auto tile_begin = [&](int kt, bool live, Stage& st) {
    t_kt = kt;
    t_live = live;
    if (SMALLC) {
      const int p = kt * 8 + a_kc;           // kernel position of this thread's chunk
      t_kh = p / g.KW;
      t_kw = p - t_kh * g.KW;
      t_kv = live & (p < g.npos);
      t_aoff = (t_kh * g.W + t_kw) * 16;
    } else {
      // K = npos * C and C is a multiple of BK (launcher): a live tile lies inside one tap, all of its k are valid
      t_fpos = g.flip ? (g.npos - 1 - pos) : pos;
      s_a = live ? (unsigned)(((kh_run * g.W + kw_run) * g.C + ci0) * 4) + p_item_a : kOOB;
      s_sh = (unsigned)(31 - pos);
      s_b = !live ? kOOB : BMODE == 0 ? (PERSIST ? (unsigned)p_kin * (unsigned)b_tile_bytes + p_item_b : (unsigned)kt * (unsigned)b_tile_bytes)
                                      : ((unsigned)t_fpos * (unsigned)g.cin_fwd * (unsigned)g.ldw + (unsigned)ci0) * 4u;
      // raw value; consumed (and replaced by 1 when there is no in_scale: empty descriptor, reads 0) only at the LDS
      // store one tile later -- touching it here would make the wave wait for the load it has just issued
      if (BMODE == 1) st.s = buf_load4s(rscale, (unsigned)a_kc * 16u, live ? (unsigned)ci0 * 4u : kOOB);   // forward never scales its input
      // advance the running position to the following tile
      ci0 += BK;
      const bool wrap = ci0 >= g.C;
      ci0 = wrap ? 0 : ci0;
      if (PERSIST) {                           // a 1x1 problem ends where its channels end: the next tile is the next problem's first
        p_kin = wrap ? 0 : p_kin + 1;
        p_item_a += wrap ? (unsigned)(g.x_bstride * 4) : 0u;
        p_item_b += wrap ? (unsigned)(g.w_bstride * 4) : 0u;
      } else {
        pos += wrap ? 1 : 0;
        kw_run += wrap ? 1 : 0;
        const bool wrap_w = kw_run >= g.KW;
        kw_run = wrap_w ? 0 : kw_run;
        kh_run += wrap_w ? 1 : 0;
      }
    }
  };

  auto load_op = [&](int idx, Stage& st) {
    if (idx < A_ITERS) {
      // ---------------- A: implicit im2col gather (invalid taps load from kOOB -> 0)
      const int i = idx;
      if (SMALLC) {
        const int ih = a_ih0[i] + t_kh, iw = a_iw0[i] + t_kw;
        const bool ok = t_kv & ((unsigned)ih < (unsigned)g.H) & ((unsigned)iw < (unsigned)g.W);
        st.a[i] = buf_load4(rx, ok ? (unsigned)(a_base[i] + t_aoff) : kOOB);
      } else {
        st.a[i] = buf_load4s(rx, a_voff[i] | ((a_inv[i] << s_sh) & kOOB), s_a);       // two VALU: padding bit of this tap -> bit 31
      }
    } else if (SMALLC) {
      constexpr int CPR = BN / 4;            // float4 chunks per k row
      const int i = idx - A_ITERS;
      const int c = tid + NT * i;
      const int kr = c / CPR;
      const bool ok = t_live & (t_kt * BK + kr < g.K) & b_nvalid[i];   // N is a multiple of 4 (launcher checks)
      st.b[i] = buf_load4(rw, ok ? (unsigned)(b_base[i] + t_kt * b_tile_bytes) : kOOB);
    } else {
      st.b[idx - A_ITERS] = buf_load4s(rw, b_voff[idx - A_ITERS], s_b);            // no VALU at all
    }
  };

  // LDS stores: every operation is one ds_write_b128 of the 4 consecutive k (A, dgrad B) or n (forward B) a thread
  // loaded; the gathered operand is scaled on the way (dgrad: frozen-BN factor of the channel).
  auto store_op = [&](int op, int buf, const Stage& st) {
    float* sA = sA0 + buf * kBufA;
    float* sB = sB0 + buf * kBufB;
    if (op < A_ITERS) {
      const int i = op;
      float4 v = st.a[i];
      if (BMODE == 1) {                     // select, not a branch: the K loop stays one basic block
        v.x *= has_in_scale ? st.s.x : 1.f;
        v.y *= has_in_scale ? st.s.y : 1.f;
        v.z *= has_in_scale ? st.s.z : 1.f;
        v.w *= has_in_scale ? st.s.w : 1.f;
      }
      *reinterpret_cast<float4*>(sA + ((tid >> 3) + kRowStep * i) * kRowPitch + a_kc * 4) = v;
    } else if (BMODE == 0) {
      constexpr int CPR = BN / 4;
      const int i = op - A_ITERS;
      const int cc = tid + NT * i;
      const int kr = cc / CPR, n4 = cc - kr * CPR;
      *reinterpret_cast<float4*>(sB + kr * PB + n4 * 4) = st.b[i];
    } else {
      const int i = op - A_ITERS;
      *reinterpret_cast<float4*>(sB + ((tid >> 3) + kRowStep * i) * kRowPitch + a_kc * 4) = st.b[i];
    }
  };

  // independent 32x32 accumulators per wave (see mfma_tile): two K-interleaved sets for the tiles with fewer than 4
  constexpr int CH = (TM * TN >= 4) ? 1 : kChainsSmallTile;
  f32x16 accs[CH][TM][TN];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.f;

  if (kt_begin < kt_end) {
    tile_begin(kt_begin, true, st0);
#pragma unroll
    for (int op = 0; op < kLoadOps; ++op) load_op(op, st0);
    tile_begin(kt_begin + 1, kt_begin + 1 < kt_end, st1);
#pragma unroll
    for (int op = 0; op < kLoadOps; ++op) load_op(op, st1);
#pragma unroll
    for (int op = 0; op < kStoreOps; ++op) store_op(op, 0, st0);
    __syncthreads();
#ifdef RADNET_DIAG_STAMPS
    t_first = __builtin_amdgcn_s_memtime();
#endif
    // fragment offsets (see mfma_tile_rows): row-major operands start at (row, k = 4*hi), the k-major one at row 4*hi
    // (8-wave workgroups: waves 4-7 start at k = 16 of the tile)
    const int a_off = (wm * (BM / WM) + l31) * kRowPitch + 4 * hi + kKPart * khalf;
    const int b_off = (BMODE == 0) ? (4 * hi + kKPart * khalf) * PB + wn * (BN / WN) + l31 : (wn * (BN / WN) + l31) * kRowPitch + 4 * hi + kKPart * khalf;
    // invariant at the top of step(kt, buf): LDS buffer `buf` holds tile kt; stage `nxt` holds tile kt+1 (in flight or
    // landed); stage `cur` is free.  MFMA steps 0 .. kLoadOps-1 each carry one global load of tile kt+2, the steps
    // after them (all but the last, which has no MFMA behind it to hide under) the LDS stores of tile kt+1.
    constexpr int kSteps = kStepsW;
    constexpr int kStoreSteps = kSteps - 1 - kLoadOps;
    constexpr int kStoresPerStep = (kStoreOps + kStoreSteps - 1) / kStoreSteps;
    static_assert(kStoreSteps >= 1, "tile too large for the staging schedule");
    auto step = [&](int kt, int buf, Stage& cur, Stage& nxt) {
      tile_begin(kt + 2, kt + 2 < kt_end, cur);
      mfma_tile_rows<TM, TN, CH, BMODE != 0, kStepsW>(sA0 + buf * kBufA, sB0 + buf * kBufB, PB, a_off, b_off, accs, [&](int s) {
        if (s < kLoadOps) {
#ifndef RADNET_DIAG_SKIP_LOADS
          load_op(s, cur);
#endif
        } else if (s < kSteps - 1) {
#if defined(RADNET_DIAG_SINK_STORES)         // loads stay (consumed by an empty asm after their wait), the LDS stores go
#pragma unroll
          for (int q = 0; q < kStoresPerStep; ++q) {
            const int op = (s - kLoadOps) * kStoresPerStep + q;
            if (op < kStoreOps) {
              const float4 v = op < A_ITERS ? nxt.a[op < A_ITERS ? op : 0] : nxt.b[op < A_ITERS ? 0 : op - A_ITERS];
              asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
          }
#elif !defined(RADNET_DIAG_SKIP_STORES)
#pragma unroll
          for (int q = 0; q < kStoresPerStep; ++q) {
            const int op = (s - kLoadOps) * kStoresPerStep + q;
            if (op < kStoreOps) store_op(op, buf ^ 1, nxt);
          }
#endif
        }
      });
#ifndef RADNET_DIAG_SKIP_BARRIER
      __syncthreads();
#endif
    };
    // PERSIST: a problem's last K tile has been multiplied -- its sums leave while the next problem's first tile sits in the other
    // LDS buffer and its second is in flight (nothing here waits for memory: plain stores, the K-part sums through LDS scratch of
    // their own; the scratch is rewritten one barrier-terminated step later at the earliest)
    [[maybe_unused]] int p_done = 0, p_out = 0;
    auto item_flush = [&]() {
#pragma unroll
      for (int c = 1; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) accs[0][i][j] += accs[c][i][j];
      if (KH > 1) {
        constexpr int kPartFloats = WG * TM * TN * 16 * 64;
        float* red = lds + igemm_lds_floats<BM, BN, BMODE>() + (wgi * TM * TN * 16) * 64 + lane;
        if (khalf > 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) red[(khalf - 1) * kPartFloats + ((i * TN + j) * 16 + r) * 64] = accs[0][i][j][r];
        }
        __syncthreads();
        if (khalf == 0) {
#pragma unroll
          for (int h = 1; h < KH; ++h)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[0][i][j][r] += red[(h - 1) * kPartFloats + ((i * TN + j) * 16 + r) * 64];
        }
      }
      const __amdgpu_buffer_rsrc_t ryp = make_rsrc(g.y + (bz + p_out) * g.y_bstride, g.y_bytes);
      const unsigned ldy4p = (unsigned)g.ldy * 4u;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = m0 + wm * (BM / WM) + i * 32 + 4 * hi;
          const unsigned vy = (khalf == 0 && n < g.N && mb < g.M) ? ((unsigned)mb * (unsigned)g.ldy + (unsigned)n) * 4u : kOOB;
#pragma unroll
          for (int r = 0; r < 16; ++r) buf_store1(ryp, vy + (unsigned)((r & 3) + 8 * (r >> 2)) * ldy4p, accs[0][i][j][r]);
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.f;
      ++p_out;
    };
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
      step(kt, 0, st0, st1);
      if (PERSIST && ++p_done == nk_total) { item_flush(); p_done = 0; }
      if (kt + 1 < kt_end) {
        step(kt + 1, 1, st1, st0);
        if (PERSIST && ++p_done == nk_total) { item_flush(); p_done = 0; }
      }
    }
  }
  if (PERSIST) return;                  // every problem's tile has been stored
  f32x16(&acc)[TM][TN] = accs[0];
#pragma unroll
  for (int c = 1; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += accs[c][i][j];
  // 8-wave workgroups: waves 4-7 hold the sums over the second half of every K tile; they hand them to waves 0-3
  // through the (now free) staging array and take no further part in the output -- every global access below is
  // predicated on live_out (offset kOOB otherwise), the barriers are reached by all eight waves.
  bool live_out = true;
  if (KH > 1) {
    static_assert((KH - 1) * WG * TM * TN * 16 * 64 <= igemm_lds_floats<BM, BN, BMODE>(), "K-part sums do not fit the staging array");
    constexpr int kPartFloats = WG * TM * TN * 16 * 64;
    float* red = lds + (wgi * TM * TN * 16) * 64 + lane;
    if (khalf > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(khalf - 1) * kPartFloats + ((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int h = 1; h < KH; ++h)            // parts added in k order
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(h - 1) * kPartFloats + ((i * TN + j) * 16 + r) * 64];
    }
    __syncthreads();
    live_out = khalf == 0;
  }
  if constexpr (FUSE != 0) {
    static_assert(BN == 64 && BMODE == 0 && !SMALLC && !COH && !PERSIST && KH == 1 && TM == 1 && TN == 1, "the fused tail follows a whole-K [BM x 64] forward tile");
    bneck_tail<BM, WAVES, FUSE == 2>(g, *tz, lds, acc[0][0], m0);
    return;
  }
#ifdef RADNET_DIAG_STAMPS
  t_loop = __builtin_amdgcn_s_memtime();
  // stamps go to a buffer of their own; nothing the kernel outputs is computed from them.  The epilogue stamp is
  // taken by a trailing block below (after the stores have been ISSUED, plus a vmcnt(0) wait so it covers their
  // completion).
  auto write_stamps = [&]() {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (g.stamps != nullptr && tid == 0) {
      unsigned long long* s = g.stamps + 8ull * (bid_x + (unsigned long long)grid_x * bid_y);
      s[0] = t_start; s[1] = t_first; s[2] = t_loop; s[3] = t_end;
      s[4] = rt_start; s[5] = __builtin_amdgcn_s_memrealtime();
      s[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
      s[7] = (unsigned long long)(kt_end - kt_begin);
    }
  };
#endif

  // ---- epilogue: accumulator register r of a 32x32 tile = row (r&3)+8*(r>>2)+4*hi, column lane&31
  if (slot >= 0) {
    // K-split tile: every slice writes its partial sums as a dense BM x BN slab (no bounds: rows past M accumulated
    // zeros), then takes a ticket from the tile's arrival counter; the slice that draws the last ticket sums ALL
    // slabs in slot (= k) order -- its own included, read back from memory, so the result does not depend on who
    // arrived last -- and applies the epilogue.  Hand-off = cdna_hip_programming.md 6 Guideline 16 R1 / 5 'In-launch
    // split-K reduction', write-through form: every slab store carries sc1 and is drained (vmcnt(0)) by its wave
    // before the barrier, one lane takes the relaxed agent-scope ticket, and EVERY slab load of the reducer is an
    // sc1 load -- no release / acquire cache maintenance (the plain-store + fence form cost 5-12 us per workgroup
    // here: each release writes back the XCD's whole L2).  Correct for any placement of the slices on XCDs / CUs.
    // Slab layout is private to this kernel: each lane keeps the 16 registers of a 32x32 accumulator contiguous
    // (64 bytes), so a slab moves with four 16-byte accesses per accumulator instead of sixteen 4-byte ones.
    const unsigned lane_off = live_out ? (unsigned)((wgi * TM * TN * 64 + lane) * 16) * 4u : kOOB;
    const __amdgpu_buffer_rsrc_t rslab = make_rsrc(g.partial + (size_t)slot * (BM * BN), BM * BN * 4u);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          buf_store4_sc1(rslab, lane_off + (unsigned)(((i * TN + j) * 64 * 16 + q * 4) * 4),
                         make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile int* flag = reinterpret_cast<volatile int*>(lds);      // the staging array is free after the K loop
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(g.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(n_slices - 1);
      if (last) __hip_atomic_store(g.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      flag[0] = last;
    }
    __syncthreads();
    if (flag[0] == 0) {
#ifdef RADNET_DIAG_STAMPS
      write_stamps();
#endif
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // compiler-only: keeps the slab loads below the ticket
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // slices are ADDED in slot order (deterministic), but their loads are issued kGroup slices at a time: one memory
    // round trip per group instead of one per slice (a slice past the end reads from kOOB, i.e. zeros)
    constexpr int kGroup = (TM * TN == 1) ? 4 : (TM * TN == 2 ? 2 : 1);
    for (int s0 = 0; s0 < n_slices; s0 += kGroup) {
      float4 v[kGroup][TM * TN * 4];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(g.partial + (size_t)(slot0 + s0 + u) * (BM * BN), BM * BN * 4u);
        const unsigned off = (s0 + u < n_slices) ? lane_off : kOOB;
#pragma unroll
        for (int t = 0; t < TM * TN * 4; ++t) v[u][t] = buf_load4_sc1(rsrc, off + (unsigned)(((t >> 2) * 64 * 16 + (t & 3) * 4) * 4));
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 w = v[u][(i * TN + j) * 4 + q];
              acc[i][j][4 * q] += w.x; acc[i][j][4 * q + 1] += w.y; acc[i][j][4 * q + 2] += w.z; acc[i][j][4 * q + 3] += w.w;
            }
    }
  }
  // Branch-free like the operand loads, and with the same split of the address: per lane ONE offset per 32x32 tile (its
  // column in the tile's first row, kOOB for a column past N or a wave that holds no output), the row of accumulator
  // register r -- (r&3) + 8*(r>>2) rows further down -- in the SGPR offset.  Rows past M need no test: the descriptors end
  // with row M-1 (y_bytes = ((M-1)*ld + N)*4), the hardware drops the store / answers the load with 0.  Only the LOADS use
  // the SGPR operand: buffer stores with a non-zero SGPR offset ran the whole kernel at HALF speed on gfx950 (measured,
  // tools/concurrency_probe.py: 66 -> 33 TFLOP/s on a 1x1 layer), so the stores add the row offset in a VGPR.  All 16 residual
  // (and mask) loads of a 32x32 tile are issued back to back before the first store, so their latency is paid once per
  // tile instead of once per register (a conditional load -> store chain cannot be reordered by the compiler: y may
  // alias the addend).
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(g.y + bz * g.y_bstride, g.y_bytes);
  const __amdgpu_buffer_rsrc_t radd = make_rsrc(g.addend, g.addend ? g.add_bytes : 0u);     // null -> every load returns 0
  const __amdgpu_buffer_rsrc_t rmask = make_rsrc(g.mask, g.mask ? g.mask_bytes : 0u);
  const bool has_mask = g.mask != nullptr;
  const unsigned ldy4 = (unsigned)g.ldy * 4u, lda4 = (unsigned)g.ld_add * 4u, ldm4 = (unsigned)g.ld_mask * 4u;
  // Round 4: straight-line code.  The first form tested `has_mask`, `act == 1`, `act == 2 && n < act_cols` per accumulator
  // register: the compiler kept them as branches -- 16 x (two scalar branches, a re-load of the kernel arguments with its wait,
  // for the mask an s_waitcnt vmcnt(0) per row) per 32x32 tile.  Now everything wave-uniform is decided once: the per-column
  // factors come through descriptors (null -> 0, replaced by 1 / 0 with a select), ReLU is a select on a uniform flag, the
  // sigmoid columns (rpn_out_class only) are a copy of the loop under ONE uniform branch, the mask exists in the dgrad form only.
  const bool has_scale = g.scale != nullptr, relu = g.act == 1;
  const __amdgpu_buffer_rsrc_t rsc = make_rsrc(g.scale, has_scale ? (unsigned)g.N * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsh = make_rsrc(g.shift, g.shift ? (unsigned)g.N * 4u : 0u);
  float scv[TN], shv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / WN) + j * 32 + l31;
    const unsigned off = n < g.N ? (unsigned)n * 4u : kOOB;
    scv[j] = buf_load1(rsc, off);
    shv[j] = buf_load1(rsh, off);
  }
  auto out_tiles = [&](auto sig_tag) {
    constexpr bool SIG = decltype(sig_tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 32 + l31;
      const bool nv = n < g.N;
      const float sc = has_scale ? scv[j] : 1.f, sh = shv[j];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * (BM / WM) + i * 32 + 4 * hi;
        const bool col_ok = live_out & nv & (mb < g.M);        // mb >= M: every row of this lane is past the end
        const unsigned vy = col_ok ? ((unsigned)mb * (unsigned)g.ldy + (unsigned)n) * 4u : kOOB;
        const unsigned va = col_ok ? ((unsigned)mb * (unsigned)g.ld_add + (unsigned)n) * 4u : kOOB;
        const unsigned vm = col_ok ? ((unsigned)mb * (unsigned)g.ld_mask + (unsigned)n) * 4u : kOOB;
        float ad[16], mk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifdef RADNET_DIAG_SKIP_EPILOGUE                 // measurement only: 1 of 16 rows is loaded / stored
          if (r != 0) { ad[r] = 0.f; mk[r] = 1.f; continue; }
#endif
          const unsigned row = (unsigned)((r & 3) + 8 * (r >> 2));
          ad[r] = buf_load1s(radd, va, row * lda4);
          mk[r] = 1.f;
          if (BMODE == 1) mk[r] = buf_load1s(rmask, vm, row * ldm4);      // null mask: empty descriptor, reads 0 (selected away below)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned row = (unsigned)((r & 3) + 8 * (r >> 2));
          float v = acc[i][j][r] * sc + sh + ad[r];
          if (BMODE == 1) v = (has_mask & !(mk[r] > 0.f)) ? 0.f : v;
          if (SIG) {
            const float sg = 1.f / (1.f + __expf(-v));
            v = n < g.act_cols ? sg : v;
          } else {
            const float vr = fmaxf(v, 0.f);
            v = relu ? vr : v;
          }
#ifdef RADNET_DIAG_SKIP_EPILOGUE
          if (r != 0) { asm volatile("" ::"v"(v)); continue; }
#endif
          if (COH) buf_store1_sc1(ry, vy + row * ldy4, v);
          else buf_store1(ry, vy + row * ldy4, v);      // one add; a STORE with a non-zero SGPR offset is slow (see above)
        }
      }
    }
  };
  if (g.act == 2) out_tiles(std::true_type{});
  else out_tiles(std::false_type{});
#ifdef RADNET_DIAG_STAMPS
  write_stamps();
#endif
}

template <int BM, int BN, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) conv_igemm_persist_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[igemm_lds_floats<BM, BN, 0>() + igemm_persist_scratch_floats<BM, BN, WAVES>()];
  if (g.xcd_batch) {                    // see conv_igemm_kernel
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), per = total >> 3;
    const unsigned l2 = lin < (per << 3) ? (lin & 7u) * per + (lin >> 3) : lin;
    const unsigned z = l2 / (gx * gy), r = l2 - z * gx * gy, y = r / gx;
    conv_igemm_body<BM, BN, 0, false, WAVES, false, true>(g, lds, r - y * gx, y, z, gx);
    return;
  }
  conv_igemm_body<BM, BN, 0, false, WAVES, false, true>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

template <int BM, int BN, int BMODE, bool SMALLC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) conv_igemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[igemm_lds_floats<BM, BN, BMODE>()];
  if (g.batch > 1 && g.xcd_batch) {
    // The hardware deals workgroup L (x fastest, then y, z) to XCD L % 8, and every XCD has its own L2: the tiles of ONE problem of
    // the batch -- which share that problem's operands (a Winograd position's filter slice is read by every row tile, its
    // transformed input by every column tile) -- land on eight L2s and each fetches its own copy (rpn_conv1's 36 GEMMs: 306 MB
    // fetched for 111 MB of operands).  Renumbered, XCD k runs the contiguous range [k, k + 1) * total / 8 of (problem, tile).
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), per = total >> 3;
    const unsigned l2 = lin < (per << 3) ? (lin & 7u) * per + (lin >> 3) : lin;
    const unsigned z = l2 / (gx * gy), r = l2 - z * gx * gy, y = r / gx;
    conv_igemm_body<BM, BN, BMODE, SMALLC, WAVES>(g, lds, r - y * gx, y, z, gx);
    return;
  }
  conv_igemm_body<BM, BN, BMODE, SMALLC, WAVES>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// 3x3 conv + the pointwise convs behind it (bneck_tail): one workgroup per BM output rows, all 64 channels of the 3x3
template <int BM, int WAVES, int FUSE>
__global__ void __launch_bounds__(64 * WAVES) conv_bneck_kernel(GemmArgs g, TailArgs tz) {
  constexpr int kLds = igemm_lds_floats<BM, 64, 0>() > bneck_lds_floats<BM>() ? igemm_lds_floats<BM, 64, 0>() : bneck_lds_floats<BM>();
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  conv_igemm_body<BM, 64, 0, false, WAVES, false, false, FUSE>(g, lds, blockIdx.x, 0u, 0u, gridDim.x, &tz);
}

// Two INDEPENDENT forward problems as one launch (round 4: branch2a and the shortcut conv of a conv_block read the same input,
// resnet50.py:100,111): workgroups [0, n1) run the first, [n1, n1 + n2) the second -- one launch floor (~4 us + a lockstep
// prologue / epilogue) instead of two, and the narrow branch2a grid (N = 128 .. 512) no longer has the chip to itself.  Plain grids,
// one tile shape for both.
template <int BM, int BN>
__global__ void __launch_bounds__(256) conv_fwd_pair_kernel(GemmArgs g1, GemmArgs g2, unsigned n1, unsigned gx1, unsigned gx2) {
  __shared__ __attribute__((aligned(16))) float lds[igemm_lds_floats<BM, BN, 0>()];
  const unsigned b = blockIdx.x;
  if (b < n1) {
    const unsigned by = b / gx1;
    conv_igemm_body<BM, BN, 0, false, 4>(g1, lds, b - by * gx1, by, 0u, gx1);
  } else {
    const unsigned c = b - n1, by = c / gx2;
    conv_igemm_body<BM, BN, 0, false, 4>(g2, lds, c - by * gx2, by, 0u, gx2);
  }
}

// ---- wgrad kernel -----------------------------------------------------------------------------------
// dW[k][n] (+)= sum_m im2col(x)[m][k] * (dy[m][n] * gscale[n]).  Output tile BMK (k) x BN (n); the
// reduction runs over output pixels m in steps of 32, optionally split across blockIdx.z (atomics).
struct WgradArgs {
  const float* x;
  const float* dy;
  const float* gscale;
  float* dw;
  float* db;             // bias gradient [N] (atomic adds by the workgroups of the first k tile) or null
  int H, W, C, OH, OW, KW, stride, pad_t, pad_l;
  int M, N, K;
  int ld_dy, ldw;
  int mt_per_split;
  int atomic;
  const int* rowtab;     // [taps][mpad] byte offset of the row's tap in the biased x descriptor, or kOOB; see get_row_table
  int mpad;              // M rounded up to whole 32-row tiles
  int xcd_batch;         // batched launch: XCD-contiguous workgroup numbering (GemmArgs::xcd_batch)
  unsigned x_bias;       // bytes the x descriptor starts ahead of x (halo rows keep non-negative offsets)
  unsigned x_bytes, dy_bytes;
  int batch, splits;     // batch > 1: blockIdx.z = problem * splits + split (radnet_wgrad_batched)
  long long x_bstride, dy_bstride, dw_bstride;   // floats between consecutive problems
  // Ordered reduction of a split launch (radnet_ctx::deterministic): the splits write their partial tiles as slabs, the last
  // one to arrive at a tile sums them in split order -- the forward kernel's in-launch split-K protocol.  slabs == null:
  // fp32 atomics (run-to-run differences in the last bits).
  float* slabs;          // [tile][split][BMK*BN], then the bias partials [n tile][split][BN]
  unsigned* counters;    // arrival counter per tile (zero outside a launch)
  int accumulate;        // ordered form: 1 = add the sum to dw's contents, 0 = store it
  int tiles_x, tiles_y;  // grid.x, grid.y of the launch (the pair kernel has a grid of its own)
};

template <int BMK, int BN>
constexpr int wgrad_lds_floats() { return 2 * BK * ((BMK + 4) + (BN + 4)); }

template <int BMK, int BN>
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& g, float* __restrict__ lds, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z) {
  constexpr int TM = BMK / 64, TN = BN / 64;
  constexpr int PA = BMK + 4, PB = BN + 4;
  constexpr int A_ITERS = BMK / 32, B_ITERS = BN / 32;
  constexpr int CPRA = BMK / 4, CPRB = BN / 4;
  float* sA0 = lds;
  float* sB0 = lds + 2 * BK * PA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int k0 = bid_x * BMK, n0 = bid_y * BN;

  // this block's k range lies inside one kernel position when C % BMK == 0 (launcher guarantees)
  const int pos = k0 / g.C;
  const int cbase = k0 - pos * g.C;

  const int a_k4 = tid % CPRA, a_mr = tid / CPRA;      // A_ITERS rows: a_mr + (NTHREADS/CPRA)*i
  const int b_n4 = tid % CPRB, b_mr = tid / CPRB;
  const bool a_kv = (k0 + a_k4 * 4) < g.K;
  const bool b_nv = (n0 + b_n4 * 4) < g.N;
  float4 gs = make_float4(1, 1, 1, 1);
  if (g.gscale != nullptr && b_nv) gs = *reinterpret_cast<const float4*>(g.gscale + n0 + b_n4 * 4);

  const int nmt = (g.M + BK - 1) / BK;
  const int zsplit = g.batch > 1 ? (int)(bid_z % (unsigned)g.splits) : (int)bid_z;
  const long long bp = g.batch > 1 ? (long long)(bid_z / (unsigned)g.splits) : 0;
  const int mt_begin = zsplit * g.mt_per_split;
  int mt_end = mt_begin + g.mt_per_split;
  if (mt_end > nmt) mt_end = nmt;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(g.x + bp * g.x_bstride) - g.x_bias, g.x_bytes + g.x_bias);
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(g.dy + bp * g.dy_bstride, g.dy_bytes);
  float4 ra[A_ITERS], rb[B_ITERS];
  // bias gradient: the workgroups of the first k tile see every (dy * gscale) row of their m range exactly once on
  // its way into LDS; they keep a running column sum and add it to db at the end (keras Conv2D bias / the beta-free
  // FixedBatchNormalization shift: d/db = sum over pixels of the scaled output gradient)
  const bool do_bias = g.db != nullptr && bid_x == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

  // Row table (host-built once per conv geometry, get_row_table): rowtab[tap][m] = byte offset of input pixel
  // (image, ih0 + kh, iw0 + kw, channel 0) of output row m -- kOOB where the tap falls into the padding or m >= M.  The
  // reduction index m advances by 32 per tile, so decoding m -> (image, oh, ow) inside the loop cost two multiply-high
  // divisions and three 16-cycle multiplies per load; with the table the gather is ONE vector add per load, everything
  // tile-dependent (table row, channel base, dy row, "tile past the end") is wave-uniform and sits in the loads' SGPR
  // offset (see conv_igemm_kernel).  The entries of tile t+2 are fetched while tile t is multiplied, one tile ahead of
  // the loads that use them.
  const __amdgpu_buffer_rsrc_t rtab = make_rsrc(g.rowtab + (size_t)pos * g.mpad, (unsigned)g.mpad * 4u);
  const unsigned s_tap = (unsigned)(cbase * 4);            // the tap is in the table; the k tile adds its first channel
  unsigned b_voff[B_ITERS], e_voff[A_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) b_voff[i] = b_nv ? (unsigned)(((b_mr + (NTHREADS / CPRB) * i) * g.ld_dy + n0 + b_n4 * 4) * 4) : kOOB;
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) e_voff[i] = (unsigned)((a_mr + (NTHREADS / CPRA) * i) * 4);
  const unsigned a_lane = a_kv ? (unsigned)a_k4 * 16u : kOOB;
  const unsigned dy_tile_bytes = (unsigned)(BK * g.ld_dy * 4);
  struct Entries {
    unsigned e[A_ITERS];
  };
  Entries ent0, ent1;

  constexpr int kLoadOps = A_ITERS + B_ITERS, kStoreOps = A_ITERS + B_ITERS;
  auto entry_op = [&](int i, int mt, Entries& en) {
    en.e[i] = __builtin_amdgcn_raw_buffer_load_b32(rtab, (int)e_voff[i], (int)__builtin_amdgcn_readfirstlane(mt < mt_end ? (unsigned)mt * (BK * 4u) : kOOB), 0);
  };
  auto load_op = [&](int idx, int mt, const Entries& en) {
    const bool live = mt < mt_end;         // a dead tile's table entries read 0: its loads go out of range through the SGPR offset
    if (idx < A_ITERS) {
      ra[idx] = buf_load4s(rx, en.e[idx] + a_lane, live ? s_tap : kOOB);
    } else {
      // rows past M lie past the end of the dy descriptor
      rb[idx - A_ITERS] = buf_load4s(rdy, b_voff[idx - A_ITERS], live ? (unsigned)mt * dy_tile_bytes : kOOB);
    }
  };
  auto store_op = [&](int idx, int buf) {
    float* sA = sA0 + buf * BK * PA;
    float* sB = sB0 + buf * BK * PB;
    if (idx < A_ITERS) {
      const int i = idx;
      *reinterpret_cast<float4*>(sA + (a_mr + (NTHREADS / CPRA) * i) * PA + a_k4 * 4) = ra[i];
    } else {
      const int i = idx - A_ITERS;
      float4 v = rb[i];
      v.x *= gs.x; v.y *= gs.y; v.z *= gs.z; v.w *= gs.w;
      *reinterpret_cast<float4*>(sB + (b_mr + (NTHREADS / CPRB) * i) * PB + b_n4 * 4) = v;
      // rows past M / columns past N / tiles past the end were loaded as 0
      csum.x += do_bias ? v.x : 0.f; csum.y += do_bias ? v.y : 0.f; csum.z += do_bias ? v.z : 0.f; csum.w += do_bias ? v.w : 0.f;
    }
  };

  constexpr int CH = (TM * TN >= 4) ? 1 : kChainsSmallTile;     // independent accumulator sets, see mfma_tile
  f32x16 accs[CH][TM][TN];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.f;

  if (mt_begin < mt_end) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) entry_op(i, mt_begin, ent0);
#pragma unroll
    for (int op = 0; op < kLoadOps; ++op) load_op(op, mt_begin, ent0);
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) entry_op(i, mt_begin + 1, ent1);
#pragma unroll
    for (int op = 0; op < kStoreOps; ++op) store_op(op, 0);
    __syncthreads();
    const int a_off = hi * PA + wm * (BMK / 2) + l31;
    const int b_off = hi * PB + wn * (BN / 2) + l31;
    // Same dealing-out of the staging operations between the MFMA steps as the forward kernel (one basic block per
    // tile, past-the-end tiles load from kOOB): steps 0.. carry the loads of tile mt+1 (and the table entries of
    // tile mt+2), the last steps but one its LDS stores -- one register stage, the loads have 7+ MFMA steps to land.
    constexpr int kSteps = BK / 2;
    constexpr int kStoreSteps = (kStoreOps < kSteps - 1 - kLoadOps) ? kStoreOps : kSteps - 1 - kLoadOps;
    constexpr int kStoresPerStep = (kStoreOps + kStoreSteps - 1) / kStoreSteps;
    constexpr int kFirstStoreStep = kSteps - 1 - kStoreSteps;
    static_assert(kStoreSteps >= 1, "tile too large for the 16-step staging schedule");
    auto step = [&](int mt, int buf, Entries& cur, Entries& nxt) {   // cur: entries of tile mt+1, nxt: receives mt+2
      mfma_tile<TM, TN, CH>(sA0 + buf * BK * PA, sB0 + buf * BK * PB, PA, PB, a_off, b_off, accs, [&](int s) {
        if (s < kLoadOps) {
          load_op(s, mt + 1, cur);
          if (s < A_ITERS) entry_op(s, mt + 2, nxt);
        } else if (s >= kFirstStoreStep && s < kSteps - 1) {
#pragma unroll
          for (int q = 0; q < kStoresPerStep; ++q) {
            const int op = (s - kFirstStoreStep) * kStoresPerStep + q;
            if (op < kStoreOps) store_op(op, buf ^ 1);
          }
        }
      });
      __syncthreads();
    };
    for (int mt = mt_begin; mt < mt_end; mt += 2) {
      step(mt, 0, ent1, ent0);
      if (mt + 1 < mt_end) step(mt + 1, 1, ent0, ent1);
    }
  }
  f32x16(&acc)[TM][TN] = accs[0];
#pragma unroll
  for (int c = 1; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += accs[c][i][j];

  const bool ordered = g.slabs != nullptr;      // uniform for the launch
  float4 bias_t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (do_bias) {             // uniform per workgroup; the staging array is free after the loop's last barrier
    float4* red = reinterpret_cast<float4*>(lds);
    red[tid] = csum;
    __syncthreads();
    if (tid < CPRB) {
      float4 t = red[tid];
      for (int q = 1; q < NTHREADS / CPRB; ++q) {
        const float4 u = red[tid + q * CPRB];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      bias_t = t;
      const int n = n0 + tid * 4;
      if (!ordered && n < g.N) {           // N is a multiple of 4
        atomicAdd(g.db + n, t.x);
        atomicAdd(g.db + n + 1, t.y);
        atomicAdd(g.db + n + 2, t.z);
        atomicAdd(g.db + n + 3, t.w);
      }
    }
    __syncthreads();
  }

  if (ordered) {
    // Same hand-off as the forward kernel's split-K (see there): sc1 slab stores drained before the barrier, one relaxed
    // agent-scope ticket per workgroup, the last arriver reads every slab back with sc1 loads and adds them in split order,
    // so the sum does not depend on which split came last.  The bias partials of the first k tile's workgroups travel the
    // same way.
    const unsigned tile = ((unsigned)bp * (unsigned)g.tiles_y + bid_y) * (unsigned)g.tiles_x + bid_x;
    const unsigned lane_off = (unsigned)((wave * TM * TN * 64 + lane) * 16) * 4u;
    const size_t total_tiles = (size_t)(g.batch > 1 ? g.batch : 1) * g.tiles_y * g.tiles_x;
    const __amdgpu_buffer_rsrc_t rslab = make_rsrc(g.slabs + ((size_t)tile * g.splits + zsplit) * (BMK * BN), BMK * BN * 4u);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          buf_store4_sc1(rslab, lane_off + (unsigned)(((i * TN + j) * 64 * 16 + q * 4) * 4),
                         make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
    float* bias_slabs = g.slabs + total_tiles * g.splits * (size_t)(BMK * BN) + ((size_t)((unsigned)bp * g.tiles_y + bid_y) * g.splits) * BN;
    if (do_bias && tid < CPRB) {
      const __amdgpu_buffer_rsrc_t rb = make_rsrc(bias_slabs + (size_t)zsplit * BN, BN * 4u);
      buf_store4_sc1(rb, (unsigned)tid * 16u, bias_t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile int* flag = reinterpret_cast<volatile int*>(lds);
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(g.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(g.splits - 1);
      if (last) __hip_atomic_store(g.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      flag[0] = last;
    }
    __syncthreads();
    if (flag[0] == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // compiler-only: keeps the slab loads below the ticket
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int kGroup = (TM * TN == 1) ? 4 : 1;      // slabs in flight per round trip, bounded by the register budget of the K loop
    for (int s0 = 0; s0 < g.splits; s0 += kGroup) {
      float4 v[kGroup][TM * TN * 4];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(g.slabs + ((size_t)tile * g.splits + (s0 + u < g.splits ? s0 + u : 0)) * (BMK * BN), BMK * BN * 4u);
        const unsigned off = (s0 + u < g.splits) ? lane_off : kOOB;
#pragma unroll
        for (int t = 0; t < TM * TN * 4; ++t) v[u][t] = buf_load4_sc1(rsrc, off + (unsigned)(((t >> 2) * 64 * 16 + (t & 3) * 4) * 4));
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 w = v[u][(i * TN + j) * 4 + q];
              acc[i][j][4 * q] += w.x; acc[i][j][4 * q + 1] += w.y; acc[i][j][4 * q + 2] += w.z; acc[i][j][4 * q + 3] += w.w;
            }
    }
    if (do_bias && tid < CPRB) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z = 0; z < g.splits; ++z) {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(bias_slabs + (size_t)z * BN, BN * 4u);
        const float4 u = buf_load4_sc1(rb, (unsigned)tid * 16u);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const int n = n0 + tid * 4;
      if (n < g.N) {           // this workgroup is the only writer of db[n0 .. n0+BN) in the launch
        g.db[n] += t.x; g.db[n + 1] += t.y; g.db[n + 2] += t.z; g.db[n + 3] += t.w;
      }
    }
  }

  if (!ordered && g.atomic) {                    // RADNET_DETERMINISTIC=0: the splits add with fp32 atomics
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = k0 + wm * (BMK / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (n < g.N && k < g.K) atomicAdd(g.dw + bp * g.dw_bstride + (size_t)k * g.ldw + n, acc[i][j][r]);
        }
      }
    }
    return;
  }
  // Plain stores / ordered accumulate, straight-line (round 4; the first form decided ordered / accumulate / atomic and the two
  // bounds per accumulator register -- five branches each, and in accumulate mode a load + wait + store round trip per register):
  // rows past K and columns past N fall outside the descriptor, accumulate mode reads a tile's 16 old values in one round trip.
  const __amdgpu_buffer_rsrc_t rdw = make_rsrc(g.dw + bp * g.dw_bstride, (unsigned)((size_t)g.K * (size_t)g.ldw * 4u));
  const bool rmw = ordered && g.accumulate;
  const unsigned ldw4 = (unsigned)g.ldw * 4u;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int kb = k0 + wm * (BMK / 2) + i * 32 + 4 * hi;
      const unsigned voff = (n < g.N && kb < g.K) ? ((unsigned)kb * (unsigned)g.ldw + (unsigned)n) * 4u : kOOB;
      float old[16];
      if (rmw) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = buf_load1(rdw, voff + (unsigned)((r & 3) + 8 * (r >> 2)) * ldw4);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) buf_store1(rdw, voff + (unsigned)((r & 3) + 8 * (r >> 2)) * ldw4, rmw ? old[r] + acc[i][j][r] : acc[i][j][r]);
    }
  }
}

template <int BMK, int BN>
__global__ void __launch_bounds__(NTHREADS) conv_wgrad_kernel(WgradArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[wgrad_lds_floats<BMK, BN>()];
  if (g.batch > 1 && g.xcd_batch) {             // see conv_igemm_kernel: the tiles of one problem share its two operands
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), per = total >> 3;
    const unsigned l2 = lin < (per << 3) ? (lin & 7u) * per + (lin >> 3) : lin;
    const unsigned z = l2 / (gx * gy), r = l2 - z * gx * gy, y = r / gx;
    conv_wgrad_body<BMK, BN>(g, lds, r - y * gx, y, z);
    return;
  }
  conv_wgrad_body<BMK, BN>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- data gradient and weight gradient of one layer in ONE launch ---------------------------------------------------------
// Both read the same dy and are independent of each other; as two launches on one stream they run one after the other, each
// with its own lockstep prologue / epilogue phases and launch gap.  Here the workgroups of the two problems alternate in the
// grid (even linear id: dgrad, odd: wgrad, the longer one fills the rest), so a CU holds workgroups of both and the phases of
// one hide under the K loops of the other.  64x64 tiles, 4 waves, for both (what the tuner picks for the classifier's layers).
struct PairMap {
  unsigned n_a, n_w;            // workgroups of the dgrad / wgrad problem
  unsigned ax, ay;              // dgrad grid (x, y); z = 1
  unsigned wx, wy;              // wgrad grid (x, y); z = n_w / (wx * wy)
};

// ABM: rows of the data gradient's output tile (64, or 32 since round 4: M = 980 rows are 31 tiles of 32 with no K slices to reduce)
template <int ABM>
__global__ void __launch_bounds__(NTHREADS) conv_bwd_pair_kernel(GemmArgs ga, WgradArgs gw, PairMap pm) {
  constexpr int kLds = igemm_lds_floats<ABM, 64, 1>() > wgrad_lds_floats<64, 64>() ? igemm_lds_floats<ABM, 64, 1>() : wgrad_lds_floats<64, 64>();
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  const unsigned b = blockIdx.x, both = 2u * (pm.n_a < pm.n_w ? pm.n_a : pm.n_w);
  bool is_a;
  unsigned idx;
  if (b < both) {
    is_a = (b & 1u) == 0u;
    idx = b >> 1;
  } else {
    is_a = pm.n_a > pm.n_w;
    idx = b - both + (both >> 1);
  }
  if (is_a) {
    const unsigned by = idx / pm.ax;
    conv_igemm_body<ABM, 64, 1, false, 4>(ga, lds, idx - by * pm.ax, by, 0u, pm.ax);
  } else {
    const unsigned plane = pm.wx * pm.wy, bz = idx / plane, r = idx - bz * plane, by = r / pm.wx;
    conv_wgrad_body<64, 64>(gw, lds, r - by * pm.wx, by, bz);
  }
}

// ---- launch helpers -----------------------------------------------------------------------------------
constexpr int kNumCU = 256;

struct TileChoice {
  int bm, bn, splits;
  int waves = 4;         // waves per workgroup (4, or 8 = K tile halved between two wave grids)
};

// Pick the output tile and split-K factor.  Cost model (CU-time in MAC units, 128 MAC/clk/CU):
//   rounds = ceil(units / 256 CUs); unit = tile MACs / tile efficiency + fixed prologue/epilogue cost;
//   split-K adds the partial-sum write + reduce pass and one more launch.
// ctx->tune (radnet_conv_autotune) replaces this guess with measured choices per problem shape.
TileChoice choose_tiles(int M, int N, int K, bool allow_split) {
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  const double eff[4] = {1.0, 0.95, 0.95, 0.85};
  const double kMacPerUs = 128.0 * 2100.0;          // per CU
  const double kFixed = 2.0 * kMacPerUs;            // ~2 us per work unit
  TileChoice best{64, 64, 1};
  double best_cost = 1e30;
  const int nk = radnet_cdiv(K, BK);
  for (int c = 0; c < 4; ++c) {
    const int bm = cand[c][0], bn = cand[c][1];
    if (bn > 64 && N <= 64) continue;
    if (bm > 64 && M <= 64) continue;
    const long long tiles = (long long)radnet_cdiv(M, bm) * radnet_cdiv(N, bn);
    for (int s = 1; s <= (allow_split ? 16 : 1); ++s) {
      if (s > 1 && nk / s < 6) break;
      const int kt = radnet_cdiv(nk, s);
      if (s > 1 && radnet_cdiv(nk, kt) != s) continue;   // would leave an empty split
      const long long units = tiles * s;
      const double rounds = (double)((units + kNumCU - 1) / kNumCU);
      double cost = rounds * ((double)bm * bn * kt * BK / eff[c] + kFixed);
      if (s > 1) cost += (2.0 + 8.0 * (double)M * N * s / 3.0e6) * kMacPerUs;
      if (cost < best_cost) {
        best_cost = cost;
        best = TileChoice{bm, bn, s};
      }
    }
  }
  return best;
}

// Work-unit table for a K-split launch: a unit = (output tile, K slice).  Whole-tile units apply the epilogue
// themselves (slot -1); the slices of a split tile write partial slabs and the last to arrive reduces them in the
// same launch (arrival counter per tile).  Tables live in device memory owned by the context (cached per shape).
radnet_unit_table* get_unit_table(radnet_ctx* ctx, int M, int N, int K, int bm, int bn, int chunks) {
  const std::array<int, 6> key{M, N, K, bm, bn, chunks};
  auto it = ctx->unit_tables.find(key);
  if (it != ctx->unit_tables.end()) return &it->second;
  const int Mt = radnet_cdiv(M, bm), Nt = radnet_cdiv(N, bn), nk = radnet_cdiv(K, BK);
  const long long T = (long long)Mt * Nt;
  std::vector<int> units;
  // Measured on MI355X: cutting the linearised iteration space into equal chunks (true stream-K) balances the CUs but
  // lost 15-20 % against the uniform split below, because the order of the units decides what the XCD L2s can
  // share: units that run together must read the SAME weight k-range (consecutive M tiles of one N tile and one
  // k slice).  So: `chunks` = slices per tile, units ordered (slice, tile_n, tile_m) -- tile_m fastest.
  const int S = chunks < 0 ? -chunks : chunks, kt = radnet_cdiv(nk, S);
  const int S_eff = radnet_cdiv(nk, kt);                      // no empty slice
  int slots = (int)T * S_eff;
  for (int s = 0; s < S_eff; ++s)
    for (int tn = 0; tn < Nt; ++tn)
      for (int tm = 0; tm < Mt; ++tm) {
        const int tile = tn * Mt + tm;
        // {tile_m, tile_n, kt_begin, kt_end | slot (-1 = whole tile, direct epilogue), first slot, slices, tile index}
        const int u[8] = {tm, tn, s * kt, std::min(nk, (s + 1) * kt), S_eff > 1 ? tile * S_eff + s : -1, tile * S_eff, S_eff, tile};
        units.insert(units.end(), u, u + 8);
      }
  if (S_eff <= 1) slots = 0;
  if (chunks < 0) {
    // XCD-aware order (chunks < 0): the hardware deals workgroup b to XCD b % 8, each XCD has its own 4 MiB L2.
    // Give every XCD a CONTIGUOUS run of the logical unit list, so the workgroups sharing one L2 are consecutive
    // M tiles of one (slice, N tile) and read the same weight rows.  Bijective for any unit count.
    const int U = (int)units.size() / 8, q = U / 8, r = U % 8;
    std::vector<int> hw(units.size());
    for (int b = 0; b < U; ++b) {
      const int x = b % 8, j = b / 8;
      const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
      std::copy(units.begin() + 8 * (size_t)(base + j), units.begin() + 8 * (size_t)(base + j) + 8, hw.begin() + 8 * (size_t)b);
    }
    units.swap(hw);
  }
  radnet_unit_table tb{};
  tb.n_units = (int)units.size() / 8;
  tb.n_split_tiles = S_eff > 1 ? (int)T : 0;
  tb.n_slots = slots;
  if (hipMalloc(&tb.d_units, units.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(tb.d_units, units.data(), units.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  if (tb.n_split_tiles) {
    // arrival counters: zeroed here once; every launch leaves them at zero again (the reducer resets its tile's)
    if (hipMalloc(&tb.d_counters, (size_t)T * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(tb.d_counters, 0, (size_t)T * sizeof(unsigned)) != hipSuccess) return nullptr;
  }
  auto ins = ctx->unit_tables.emplace(key, tb);
  return &ins.first->second;
}

// Row table of a convolution geometry for the wgrad kernel: entry m = output pixel (image, oh, ow) holds the byte offset
// of its window origin (image, oh*stride - pad_t, ow*stride - pad_l, channel 0) in x -- negative inside the halo -- and
// the origin's (ih0, iw0) packed into 16 + 16 bits.  Built on the host at the first use of a geometry, cached on the
// context (device memory, freed with it).
const int* get_row_table(radnet_ctx* ctx, const radnet_conv_desc* d) {
  const std::array<int, 11> key{d->nb, d->h, d->w_, d->c, d->oh, d->ow, d->stride, d->pad_t, d->pad_l, d->kh, d->kw};
  auto it = ctx->row_tables->m.find(key);
  if (it != ctx->row_tables->m.end()) return (const int*)it->second;
  const int M = d->nb * d->oh * d->ow, mpad = radnet_cdiv(M, BK) * BK, taps = d->kh * d->kw;
  const int64_t bias = ((int64_t)d->pad_t * d->w_ + d->pad_l) * d->c * 4;
  std::vector<uint32_t> host((size_t)taps * mpad, 0x80000000u);
  for (int kh = 0; kh < d->kh; ++kh)
    for (int kw = 0; kw < d->kw; ++kw) {
      uint32_t* row = host.data() + (size_t)(kh * d->kw + kw) * mpad;
      size_t m = 0;
      for (int img = 0; img < d->nb; ++img)
        for (int oh = 0; oh < d->oh; ++oh)
          for (int ow = 0; ow < d->ow; ++ow, ++m) {
            const int ih = oh * d->stride - d->pad_t + kh, iw = ow * d->stride - d->pad_l + kw;
            if ((unsigned)ih < (unsigned)d->h && (unsigned)iw < (unsigned)d->w_)
              row[m] = (uint32_t)((((int64_t)img * d->h + ih) * d->w_ + iw) * d->c * 4 + bias);
          }
    }
  void* dev = nullptr;
  if (hipMalloc(&dev, host.size() * sizeof(uint32_t)) != hipSuccess) return nullptr;
  if (hipMemcpy(dev, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  ctx->row_tables->m.emplace(key, dev);
  return (const int*)dev;
}

template <int BMODE, bool SMALLC, int WAVES>
void launch_igemm_w(hipStream_t st, const GemmArgs& g, const TileChoice& tc, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
  dim3 block(64 * WAVES);
  // (capping workgroups per CU with extra dynamic LDS was measured: 7-25 % slower on every layer -- co-residency wins)
  if constexpr (WAVES == 4 && !SMALLC) {        // the 32-row tiles exist in the 4-wave form only (8 waves: fewer than one A chunk per thread)
    if (tc.bm == 32 && tc.bn == 64) { RADNET_LAUNCH((conv_igemm_kernel<32, 64, BMODE, SMALLC, WAVES>), grid, block, 0, st, e0, e1, g); return; }
    if (tc.bm == 32 && tc.bn == 32) { RADNET_LAUNCH((conv_igemm_kernel<32, 32, BMODE, SMALLC, WAVES>), grid, block, 0, st, e0, e1, g); return; }
  }
  if (tc.bm == 128 && tc.bn == 128) RADNET_LAUNCH((conv_igemm_kernel<128, 128, BMODE, SMALLC, WAVES>), grid, block, 0, st, e0, e1, g);
  else if (tc.bm == 128 && tc.bn == 64) RADNET_LAUNCH((conv_igemm_kernel<128, 64, BMODE, SMALLC, WAVES>), grid, block, 0, st, e0, e1, g);
  else if (tc.bm == 64 && tc.bn == 128) RADNET_LAUNCH((conv_igemm_kernel<64, 128, BMODE, SMALLC, WAVES>), grid, block, 0, st, e0, e1, g);
  else RADNET_LAUNCH((conv_igemm_kernel<64, 64, BMODE, SMALLC, WAVES>), grid, block, 0, st, e0, e1, g);
}

// persistent batched launch: the tile shapes that have the PERSIST form (forward, channel-tiled, 4 waves)
inline bool persist_shape(const TileChoice& tc) {
  return tc.waves != 8 && ((tc.bm == 64 && tc.bn == 64) || (tc.bm == 32 && tc.bn == 64) || (tc.bm == 64 && tc.bn == 128) || (tc.bm == 32 && tc.bn == 32));
}
void launch_igemm_persist(hipStream_t st, const GemmArgs& g, const TileChoice& tc, hipEvent_t e0, hipEvent_t e1) {
  dim3 grid(radnet_cdiv(g.M, tc.bm), radnet_cdiv(g.N, tc.bn), radnet_cdiv(g.batch, g.zper)), block(256);
  if (tc.bm == 64 && tc.bn == 64) RADNET_LAUNCH((conv_igemm_persist_kernel<64, 64, 4>), grid, block, 0, st, e0, e1, g);
  else if (tc.bm == 32 && tc.bn == 64) RADNET_LAUNCH((conv_igemm_persist_kernel<32, 64, 4>), grid, block, 0, st, e0, e1, g);
  else if (tc.bm == 64 && tc.bn == 128) RADNET_LAUNCH((conv_igemm_persist_kernel<64, 128, 4>), grid, block, 0, st, e0, e1, g);
  else RADNET_LAUNCH((conv_igemm_persist_kernel<32, 32, 4>), grid, block, 0, st, e0, e1, g);
}

template <int BMODE, bool SMALLC>
void launch_igemm(hipStream_t st, const GemmArgs& g, const TileChoice& tc, int n_units, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  dim3 grid(radnet_cdiv(g.M, tc.bm), radnet_cdiv(g.N, tc.bn), g.batch > 1 ? g.batch : 1);
  if (g.units != nullptr) grid = dim3(n_units, 1, 1);
  if (tc.waves == 8) launch_igemm_w<BMODE, SMALLC, 8>(st, g, tc, grid, e0, e1);
  else launch_igemm_w<BMODE, SMALLC, 4>(st, g, tc, grid, e0, e1);
}

// radnet_conv_bwd: the final launch of run_igemm (dgrad) / run_wgrad lands here instead of on the stream while
// ctx->pair_capture is set; the measuring launches of a first, autotuned call are issued as usual (PairPause).
struct PairCapture {
  bool have_a = false, a_ok = false, have_w = false, w_ok = false;
  int a_bm = 64;
  GemmArgs ga;
  unsigned ax = 0, ay = 0;
  WgradArgs gw;
  unsigned wx = 0, wy = 0, wz = 0;
  uint64_t a_slab_bytes = 0, w_slab_bytes = 0;      // split-K slabs at the start / ordered wgrad slabs at the end of the workspace
  double flops = 0.0;
};
struct PairPause {
  radnet_ctx* ctx;
  void* saved;
  explicit PairPause(radnet_ctx* c) : ctx(c), saved(c->pair_capture) { c->pair_capture = nullptr; }
  ~PairPause() { ctx->pair_capture = saved; }
};

// checks + derived fields of a forward / data-gradient problem (descriptor extents, division constants)
int prepare_igemm(radnet_ctx* ctx, GemmArgs& g, int bmode, bool smallc) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  if (g.M >= (1 << 20) || g.OHOW >= (1 << 20)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv: M=%d exceeds 2^20 rows", g.M);
  if ((g.ldw & 3) || (g.N & 3)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv: N=%d and ldw=%d must be multiples of 4", g.N, g.ldw);
  if (((uintptr_t)g.x & 15) || ((uintptr_t)g.w & 15)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv: x / w must be 16-byte aligned");
  if (!smallc && (g.C % BK) != 0) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv: channels %d not a multiple of %d (pad, or use c=4)", g.C, BK);
  if (!smallc && g.npos > 32) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv: %d kernel taps (the padding mask of a row holds 32)", g.npos);
  g.magic_ohow = radnet_div_magic((uint32_t)g.OHOW);
  g.magic_ow = radnet_div_magic((uint32_t)g.OW);
  {
    const uint64_t xb = (uint64_t)(g.M / g.OHOW) * g.H * g.W * g.C * 4ull;      // images * H * W * C floats
    const uint64_t wb = (bmode == 0) ? (uint64_t)g.K * g.ldw * 4ull : (uint64_t)g.npos * g.cin_fwd * g.ldw * 4ull;
    // x is addressed through a descriptor that starts one halo (pad_t rows + pad_l pixels) early: that too stays below 2^31
    const uint64_t halo = ((uint64_t)g.pad_t * g.W + g.pad_l) * g.C * 4ull;
    if (xb + halo >= (1ull << 31) || wb >= (1ull << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv: tensor larger than 2 GiB");
    g.x_bytes = (unsigned)xb;
    g.w_bytes = (unsigned)wb;
    const uint64_t ld_max = (uint64_t)std::max(g.ldy, std::max(g.addend ? g.ld_add : 0, g.mask ? g.ld_mask : 0));
    if ((uint64_t)g.M * ld_max * 4ull >= (1ull << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv: output larger than 2 GiB");
    if (g.ldy < g.N || (g.addend && g.ld_add < g.N) || (g.mask && g.ld_mask < g.N))
      RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv: output / addend / mask row pitch smaller than n=%d", g.N);
    // last row ends at column N, not at the pitch: the tensor may be a column block of a wider one
    g.y_bytes = (unsigned)(((uint64_t)(g.M - 1) * g.ldy + g.N) * 4ull);
    g.add_bytes = g.addend ? (unsigned)(((uint64_t)(g.M - 1) * g.ld_add + g.N) * 4ull) : 0u;
    g.mask_bytes = g.mask ? (unsigned)(((uint64_t)(g.M - 1) * g.ld_mask + g.N) * 4ull) : 0u;
  }
  g.stamps = ctx->diag_stamps;
#ifdef RADNET_DIAG_STAMPS
  if (getenv("RADNET_DIAG_NOMEM")) g.x_bytes = g.w_bytes = 0;   // every operand load out of range: returns 0 without touching memory
#endif
  return RADNET_OK;
}

int run_igemm(radnet_ctx* ctx, GemmArgs& g, int bmode, bool smallc, int cls) {
  {
    const int rcp = prepare_igemm(ctx, g, bmode, smallc);
    if (rcp != RADNET_OK) return rcp;
  }
  const int nk = radnet_cdiv(g.K, BK);
  // TileChoice.splits = number of equal work chunks the iteration space is cut into (0/1 = one workgroup per tile)
  auto launch = [&](const TileChoice& t) -> int {
    radnet_unit_table* tb = nullptr;
    g.units = nullptr;
    g.partial = nullptr;
    g.counters = nullptr;
    // a batch is its own source of workgroups: no K slices.  |slices| = z > 1 on a batch = the persistent form, z consecutive problems
    // per workgroup (plain epilogue, dense problem strides, forward only, the tile shapes of persist_shape)
    g.zper = 0;
    if (g.batch > 1 && t.splits != 1 && t.splits != -1) {
      const int z = t.splits < 0 ? -t.splits : t.splits;
      const bool plain = !g.scale && !g.shift && !g.addend && !g.mask && !g.in_scale && g.act == 0 && g.npos == 1 && g.stride == 1;
      if (bmode != 0 || smallc || !plain || !persist_shape(t) || z > 12 || z > g.batch || g.x_bstride != (long long)g.M * g.C ||
          g.w_bstride != (long long)g.K * g.ldw || (uint64_t)z * (uint64_t)std::max(g.x_bstride, g.w_bstride) * 4ull >= (1ull << 31))
        return RADNET_ERR_UNSUPPORTED;
      g.zper = z;
    }
    if (t.bm < 64 || t.bn < 64) {               // 32-row tiles: 4 waves, channel-tiled layers, 32x64 and 32x32 only
      if (t.waves == 8 || smallc || t.bm != 32 || (t.bn != 64 && t.bn != 32)) return RADNET_ERR_UNSUPPORTED;
    }
    g.xcd_batch = (g.batch > 1 && t.splits < 0) ? 1 : 0;          // negative: the same grid, XCD-contiguous numbering
    if (g.batch <= 1 && (t.splits > 1 || t.splits < 0)) {
      tb = get_unit_table(ctx, g.M, g.N, g.K, t.bm, t.bn, t.splits);
      if (!tb) RADNET_FAIL(ctx, RADNET_ERR_HIP, "conv: cannot build the work-unit table");
      if ((uint64_t)tb->n_slots * t.bm * t.bn * sizeof(float) > ctx->ws_bytes) return RADNET_ERR_UNSUPPORTED;   // candidate skipped
      if (tb->n_split_tiles > 0 || t.splits < 0) {  // an un-split, un-swizzled table is just the plain launch
        g.units = tb->d_units;
        g.partial = (float*)ctx->ws;
        g.counters = tb->d_counters;
      } else {
        tb = nullptr;
      }
    }
    const int n_units = tb ? tb->n_units : 0;
    if (ctx->pair_capture != nullptr) {
      PairCapture* pc = (PairCapture*)ctx->pair_capture;
      pc->have_a = true;
      pc->a_ok = bmode == 1 && !smallc && (t.bm == 64 || t.bm == 32) && t.bn == 64 && t.waves != 8 && g.batch <= 1;
      pc->a_bm = t.bm;
      pc->ga = g;
      pc->a_slab_bytes = tb ? (uint64_t)tb->n_slots * t.bm * t.bn * sizeof(float) : 0;
      pc->ax = g.units != nullptr ? (unsigned)n_units : (unsigned)radnet_cdiv(g.M, t.bm);
      pc->ay = g.units != nullptr ? 1u : (unsigned)radnet_cdiv(g.N, t.bn);
      pc->flops += 2.0 * g.M * g.N * g.K;
      return RADNET_OK;
    }
    if (g.zper > 1) {
      launch_igemm_persist(ctx->stream, g, t, ctx->arm0, ctx->arm1);
    } else if (bmode == 0) {
      if (smallc) launch_igemm<0, true>(ctx->stream, g, t, n_units, ctx->arm0, ctx->arm1);
      else launch_igemm<0, false>(ctx->stream, g, t, n_units, ctx->arm0, ctx->arm1);
    } else {
      launch_igemm<1, false>(ctx->stream, g, t, n_units, ctx->arm0, ctx->arm1);
    }
    RADNET_CHECK_LAUNCH(ctx, "conv_igemm");
    return RADNET_OK;
  };
  // tile / split-K choice: measured once per problem shape when autotuning is on, else the cost model
  const radnet_shape_key key{g.batch > 1 ? 8 : (cls == 1 ? 1 : 0), g.M, g.N, g.K, g.C, g.npos, g.batch > 1 ? g.batch : g.stride};
  TileChoice tc{64, 64, 1};
  auto it = ctx->tuned->find(key);
  if (ctx->force_a > 0) {                      // radnet_force_config: tests sweep every tile / slice / order variant
    tc = TileChoice{ctx->force_a, ctx->force_b, ctx->force_splits, ctx->force_waves == 8 ? 8 : 4};
    if (g.batch <= 1 && tc.splits > 1 && nk / tc.splits < 1) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv: forced %d K slices but only %d K tiles", tc.splits, nk);
  } else if (it != ctx->tuned->end()) {
    tc = TileChoice{it->second.a, it->second.b, it->second.splits, it->second.waves == 8 ? 8 : 4};
  } else if (const radnet_tuned* nb = ctx->autotune == 2 ? radnet_tuned_neighbour(*ctx->tuned, key) : nullptr) {
    tc = TileChoice{nb->a, nb->b, nb->splits, nb->waves == 8 ? 8 : 4};
    (*ctx->tuned)[key] = *nb;
  } else if (ctx->autotune) {
    PairPause pause(ctx);                       // trial launches are real launches
    // 32-row tiles (round 4): M = 980 / 2 394 / 160 rows fill 256 CUs without K slices (no slab seam, no last-arriver tail) and
    // waste no rows where M is a multiple of 32 but not of 64; tried only where the 64x64 grid is small (they re-read B twice as often)
    const int cand[6][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {32, 64}, {32, 32}};
    const long long tiles64 = (long long)radnet_cdiv(g.M, 64) * radnet_cdiv(g.N, 64) * (g.batch > 1 ? g.batch : 1);
    const int chunks[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16};      // K slices per tile
    std::vector<std::pair<float, TileChoice>> seen;
    // A data gradient measured on behalf of radnet_conv_bwd is tuned WITHIN the shapes its one-launch form takes (64x64 tiles,
    // 4 waves): alone on the chip the 8-wave form often wins by a few per cent, but next to other lanes' launches the paired
    // launch beat "fastest dgrad + wgrad, one after the other" for every classifier layer measured in situ (tools/insitu_tune.py:
    // +2.7 % and +1.3 % on the whole step for the two layers the isolated choice had unpaired)
    const bool for_pair = pause.saved != nullptr && bmode == 1 && !smallc && g.batch <= 1;
    for (int c = 0; c < 6; ++c) {
      if ((cand[c][1] > 64 && g.N <= 64) || (cand[c][0] > 64 && g.M <= 64)) continue;
      if (for_pair && ((cand[c][0] != 64 && cand[c][0] != 32) || cand[c][1] != 64)) continue;
      if (cand[c][0] < 64 && (smallc || tiles64 > 6 * kNumCU)) continue;
      const long long tiles = (long long)radnet_cdiv(g.M, cand[c][0]) * radnet_cdiv(g.N, cand[c][1]);
      for (int s : chunks) {
        if (g.batch > 1 ? (s > 12 || s > g.batch) : (s > 1 && (ctx->ws == nullptr || nk / s < 2))) continue;      // slices shorter than 2 k-tiles (a batch: s = problems per workgroup)
        for (int sign = 1; sign >= -1; sign -= 2) {                             // -s = same slices, XCD-aware unit order
          if (sign < 0 && tiles * s * (g.batch > 1 ? g.batch : 1) < 16) continue;
          for (int waves = 4; waves <= ((for_pair || cand[c][0] < 64) ? 4 : 8); waves += 4) {        // 8 = K tile halved between two wave grids
            TileChoice t{cand[c][0], cand[c][1], sign * s, waves};
            float ms = 0.f;
            int rc = radnet_time_launches(ctx, [&]() { return launch(t); }, 3, &ms);
            if (rc == RADNET_ERR_UNSUPPORTED) continue;
            if (rc != RADNET_OK) return rc;
            seen.push_back({ms, t});
          }
        }
      }
    }
    // The 3-launch screening above is noisy (neighbouring candidates differ by a few per cent, a shared host by more):
    // the finalists are measured again, longer and twice, and the smaller figure of each decides.  Without this, two
    // processes tuned different tables and the same commit benched 3.4 or 3.7 ms per step.
    std::sort(seen.begin(), seen.end(), [](const std::pair<float, TileChoice>& a, const std::pair<float, TileChoice>& b) { return a.first < b.first; });
    float best = 1e30f;
    for (size_t i = 0; i < seen.size() && i < 6; ++i) {
      float m1 = 0.f, m2 = 0.f;
      int rc = radnet_time_launches(ctx, [&]() { return launch(seen[i].second); }, 12, &m1);
      if (rc == RADNET_OK) rc = radnet_time_launches(ctx, [&]() { return launch(seen[i].second); }, 12, &m2);
      if (rc != RADNET_OK) return rc;
      const float ms = std::min(m1, m2);
      if (ms < best) { best = ms; tc = seen[i].second; }
    }
    (*ctx->tuned)[key] = radnet_tuned{tc.bm, tc.bn, tc.splits, best, tc.waves};
    if (getenv("RADNET_TUNE_LOG"))
      fprintf(stderr, "[radnet tune] %s M=%d N=%d K=%d C=%d -> tile %dx%d chunks %d waves %d : %.1f us (%.1f TFLOP/s)\n", cls == 1 ? "dgrad" : "fwd",
              g.M, g.N, g.K, g.C, tc.bm, tc.bn, tc.splits, tc.waves, best * 1e3, 2.0 * g.M * g.N * g.K / (best * 1e9));
  } else {
    tc = choose_tiles(g.M, g.N, g.K, ctx->ws != nullptr && g.batch <= 1);
  }
  radnet_timing_arm(ctx);
  int rc = launch(tc);
  if (rc == RADNET_ERR_UNSUPPORTED && ctx->force_a <= 0) {
    // a shared / loaded / adopted choice this launch cannot use: slabs larger than THIS context's workspace, or a K-split /
    // XCD-ordered unit table for a problem that is now launched as a batch (a batch is one plain grid per problem)
    tc.splits = (tc.splits < 0 ? -1 : 1);
    rc = launch(tc);
  }
  if (rc == RADNET_ERR_UNSUPPORTED)
    RADNET_FAIL(ctx, rc, "conv: launch shape tile %dx%d slices %d waves %d cannot run M=%d N=%d K=%d batch=%d (workspace %llu bytes)", tc.bm, tc.bn,
                tc.splits, tc.waves, g.M, g.N, g.K, g.batch, (unsigned long long)ctx->ws_bytes);
  if (rc != RADNET_OK) return rc;
  radnet_timing_end_armed(ctx, cls, 2.0 * g.M * g.N * g.K * (g.batch > 1 ? g.batch : 1));
  return RADNET_OK;
}


// =====================================================================================================================
// Chain kernel: a run of dependent layers as ONE persistent launch
// =====================================================================================================================
// nn_base (resnet50.py:150-228) at batch 1 is ~50 dependent launches of 10-25 us.  Each loses 2 us to the gap behind
// the previous launch, 2 + 3 us to prologue / epilogue phases that all of its workgroups pass through together, and a tail in
// which the CUs with one workgroup fewer idle (profiles/r02_workgroup_stamps.txt) -- about half of the 0.95 ms the chain
// takes alone on the chip.  Here the whole run is one launch of `grid` persistent workgroups that draw WORK ITEMS -- one
// output tile of a conv (conv_igemm_body: the same code the layer launches run), or a block of a Winograd transform -- from
// a list the host wrote in dependency order, and start an item as soon as the items it reads from have finished:
//   * every stage (conv, Winograd input transform, batched Winograd GEMM, output transform) owns arrival counters over
//     blocks of its output (64 rows of a conv output, 64 tiles of a transformed operand, one tile row of a Winograd layer's
//     output); a finished item adds 1 to the counters of the blocks it wrote, an item waits until the blocks it reads have
//     reached the count the host computed for them (`need`);
//   * hand-off between workgroups is the split-K protocol of conv_igemm_body: outputs are written through (sc1 stores),
//     the writer drains them (vmcnt(0)) and bumps the counter with a relaxed agent-scope atomic, the reader polls the counter
//     with agent-scope loads.  Consumers read the data with ordinary loads: a line of an activation tensor is complete before
//     any workgroup may touch it (the counters cover whole rows of whole tiles, tensors are not shared, a launch starts with
//     clean caches), so no cache on the reader's side can hold an older copy;
//   * items are DEALT statically (workgroup b runs items b, b + grid, b + 2 grid, ...: a shared queue head cost one same-address
//     atomic per item) and the list is topologically sorted, so the lowest unfinished item belongs to a workgroup whose earlier
//     items are finished -- i.e. it is being run -- PROVIDED EVERY WORKGROUP OF THE GRID IS RESIDENT: the deal is deadlock-free
//     only while grid (x the number of chains running at the same time, plus whatever other launches hold CU slots) fits the
//     chip's 4 workgroups per CU.  radnet_chain_build caps one grid at 4 * 256; the engine divides that by the chains it runs
//     side by side.  A poll that does not see its counters move within 1 s raises `error` and every workgroup leaves (the grid
//     always drains); the launch's outputs are then INVALID and radnet_chain_run of the NEXT launch of that chain returns
//     RADNET_ERR_HIP (the sticky hdr->last_error travels to a mapped host word), as does radnet_chain_status;
//   * the last workgroup to leave zeroes the counters and the queue head: the launch can be replayed (hipGraph).
// A narrow grid (1-2 workgroups per CU) leaves CU slots to the other lanes' launches: the frozen base forward of an announced
// batch is background work in the pipelined step.
struct ChainStage {
  GemmArgs g;                                   // type 0: conv tile / tile of a batched GEMM
  const float* t_src;                           // type 1: x [nb][h][w][c] -> V;  type 2: M [36][T][n] -> y
  float* t_dst;
  const float* t_scale;
  const float* t_shift;
  int t_nb, t_h, t_w, t_c, t_th, t_tw, t_act, t_ldy;
  unsigned t_dst_bytes;
  int type;
};
struct ChainItem {
  int stage, bx, by, bz;
  int d0_first, d0_count, d1_first, d1_count;   // counter ranges that must have reached their `need`
  int sig0, sig1, pad0, pad1;                   // counters this item bumps when done (-1: none)
};
struct ChainHeader {
  unsigned next, exited, error, last_error;
  unsigned runs, host_lo, host_hi, pad;       // host_lo/hi: a mapped host word that receives the first error (radnet_chain_error: no sync)
};

typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void buf_store2_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float2 v) {
  f32x2v f = {v.x, v.y};
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f), r, (int)off, 0, 16);
}

// One block (256 units: a unit = one tile x 1 channel: 36 live registers, so the item does not
// raise the register budget of the GEMM items it shares the kernel with) of the F(4x4,3x3) input transform, write-through stores.
__device__ __forceinline__ void chain_wino4_input(const ChainStage& st, unsigned block) {
  typedef float VT;
  const int cv = st.t_c, H = st.t_h, W_ = st.t_w, C = st.t_c, TH = st.t_th, TW = st.t_tw;
  const unsigned T = (unsigned)(st.t_nb * TH * TW), total = T * (unsigned)cv;
  const unsigned i = block * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned tile = i / (unsigned)cv;
  const int cq = (int)(i - tile * (unsigned)cv);
  const unsigned trow = tile / (unsigned)TW;
  const int tj = (int)(tile - trow * (unsigned)TW);
  const int img = (int)(trow / (unsigned)TH);
  const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
  const float* x = st.t_src;
  VT t[6][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const int iw = 4 * tj - 1 + b;
    VT col[6], o[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const int ih = 4 * ti - 1 + a;
      col[a] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W_)
                   ? *reinterpret_cast<const VT*>(x + (((long long)img * H + ih) * W_ + iw) * C + cq)
                   : vzero<VT>();
    }
    bt6(col, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][b] = o[a];
  }
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(st.t_dst, st.t_dst_bytes);
  const unsigned off0 = (tile * (unsigned)cv + (unsigned)cq) * 4u, ps = total * 4u;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    VT o[6];
    bt6(t[a], o);
#pragma unroll
    for (int b = 0; b < 6; ++b) buf_store1_sc1(rd, off0 + (unsigned)(6 * a + b) * ps, o[b]);
  }
}

// One block of the output transform (+ folded BN scale / shift, ReLU), write-through stores.
__device__ __forceinline__ void chain_wino4_output(const ChainStage& st, unsigned block) {
  typedef float VT;
  const int N = st.t_c, nv = N, OH = st.t_h, OW = st.t_w, TH = st.t_th, TW = st.t_tw, ldy = st.t_ldy;
  const unsigned T = (unsigned)(st.t_nb * TH * TW), total = T * (unsigned)nv;
  const unsigned i = block * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned tile = i / (unsigned)nv;
  const int nq = (int)(i - tile * (unsigned)nv);
  const unsigned trow = tile / (unsigned)TW;
  const int tj = (int)(tile - trow * (unsigned)TW);
  const int img = (int)(trow / (unsigned)TH);
  const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
  const VT* src = reinterpret_cast<const VT*>(st.t_src) + tile * (unsigned)nv + nq;
  const size_t ps = total;
  VT t[4][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    VT col[6], o[4];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = src[(size_t)(6 * a + b) * ps];
    at6(col, o);
#pragma unroll
    for (int a = 0; a < 4; ++a) t[a][b] = o[a];
  }
  VT sc = 1.f, sh = 0.f;
  if (st.t_scale) sc = *reinterpret_cast<const VT*>(st.t_scale + nq);
  if (st.t_shift) sh = *reinterpret_cast<const VT*>(st.t_shift + nq);
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(st.t_dst, st.t_dst_bytes);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int oh = 4 * ti + a;
    VT o[4];
    at6(t[a], o);
    if (oh >= OH) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ow = 4 * tj + b;
      if (ow >= OW) continue;
      VT v = st.t_scale ? o[b] * sc + sh : o[b] + sh;
      if (st.t_act == 1) v = vmax0(v);
      buf_store1_sc1(rd, (unsigned)((((unsigned)img * OH + oh) * OW + ow) * (unsigned)ldy + nq) * 4u, v);
    }
  }
}

// arrival counters sit 64 bytes apart: the counters one item polls, and the ones neighbouring items bump, spread over cache
// lines and memory channels instead of queueing on one
constexpr int kCtrStride = 16;
constexpr unsigned long long kChainGiveUpTicks = 100000000ull;      // 1 s of the 100 MHz real-time counter: a workgroup gives up waiting

template <bool COHV, int DBG = 0>
__device__ __forceinline__ void chain_body(ChainHeader* __restrict__ hdr, const ChainStage* __restrict__ stages,
                                                         const ChainItem* __restrict__ items, unsigned* __restrict__ counters,
                                                         const unsigned* __restrict__ need, unsigned n_items, unsigned n_counters,
                                                         unsigned* __restrict__ marks, float* __restrict__ lds, unsigned* __restrict__ s_ctl) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // progress mark of every WAVE (diagnosis, radnet_chain_peek): phase in the low byte, item above it
#define CHAIN_MARK(phase, item) do { if (marks != nullptr && lane == 0) __hip_atomic_store(marks + blockIdx.x * 4 + wave, ((item) << 8) | (phase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
  // Control flow: every barrier of this loop must be reached by all four waves the same number of times.  The queue draw is
  // one lane's work; it sits at the END of the loop body (and once in front of the loop), not at its head -- a lane-divergent
  // branch at the head of a loop makes the compiler split the loop so that the other lanes of that wave run on to the barrier
  // first and the drawing lane arrives at it a second time (seen as a hang: waves of one workgroup in different phases).
  // The loop's own conditions are wave-uniform scalars (readfirstlane).
  // Static deal: workgroup b runs items b, b + grid, b + 2 grid, ... (a shared queue head costs one same-address atomic per
  // item: 38 720 of them serialised to ~2 ms for a 1000x600 base forward, more than the launches they replace).  Still
  // deadlock-free while every workgroup of the grid is resident (radnet_chain_build caps the grid at the chip's capacity for
  // this kernel): the lowest unfinished item belongs to a workgroup whose earlier items are finished, i.e. it is being run.
  if (threadIdx.x == 0) s_ctl[1] = 0u;
  __syncthreads();
  unsigned idx = blockIdx.x;
  bool gave_up = false;
  for (; idx < n_items; idx += gridDim.x) {
    CHAIN_MARK(1u, idx);
    const ChainItem* ip = items + idx;
    const int it_stage = __builtin_amdgcn_readfirstlane(ip->stage);
    const int it_bx = __builtin_amdgcn_readfirstlane(ip->bx), it_by = __builtin_amdgcn_readfirstlane(ip->by), it_bz = __builtin_amdgcn_readfirstlane(ip->bz);
    const int d0f = __builtin_amdgcn_readfirstlane(ip->d0_first), d0n = __builtin_amdgcn_readfirstlane(ip->d0_count);
    const int d1f = __builtin_amdgcn_readfirstlane(ip->d1_first), d1n = __builtin_amdgcn_readfirstlane(ip->d1_count);
    const int sig0 = __builtin_amdgcn_readfirstlane(ip->sig0), sig1 = __builtin_amdgcn_readfirstlane(ip->sig1);
    // ---- wait until the blocks this item reads are complete: one counter per lane, one round trip per poll
    if (DBG != 2 && wave == 0 && d0n + d1n > 0) {      // DBG 2 (diagnosis): nobody waits -- results are garbage, the time is the items' own
      const int ci = lane < d0n ? d0f + lane : (lane < d0n + d1n ? d1f + (lane - d0n) : -1);
      const unsigned want = ci >= 0 ? need[ci] : 0u;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, constant
      unsigned polls = 0;
      for (;;) {
        const unsigned have = ci >= 0 ? __hip_atomic_load(counters + (size_t)ci * kCtrStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (__all(have >= want)) break;
        if ((++polls & 63u) == 0u) {             // now and then: has somebody given up / have we waited kChainGiveUpTicks
          const bool late = __builtin_amdgcn_s_memrealtime() - t0 > kChainGiveUpTicks;
          if (late || __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (late && lane == 0) __hip_atomic_store(&hdr->error, idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) s_ctl[1] = 1u;
            break;
          }
        }
        // back off: a workgroup that is early for its inputs must not crowd out the atomics that would complete them
        if (polls < 4u) __builtin_amdgcn_s_sleep(16);
        else if (polls < 16u) __builtin_amdgcn_s_sleep(48);
        else __builtin_amdgcn_s_sleep(127);
      }
    }
    __syncthreads();
    gave_up = __builtin_amdgcn_readfirstlane(s_ctl[1]) != 0u;      // some wait of the launch timed out: leave (uniform for the workgroup)
    if (gave_up) break;
    CHAIN_MARK(2u, idx);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // compiler-only: the item's loads stay below the poll
    const ChainStage& st = stages[it_stage];
    const int type = __builtin_amdgcn_readfirstlane(st.type);
    if (DBG == 1) {
      // diagnosis: the queue / counter machinery without any item work
    } else if (type == 0) {
      GemmArgs g = st.g;
      conv_igemm_body<64, 64, 0, false, 4, COHV>(g, lds, (unsigned)it_bx, (unsigned)it_by, (unsigned)it_bz, 0u);
    } else if (type == 1) {
      chain_wino4_input(st, (unsigned)it_bx);
    } else {
      chain_wino4_output(st, (unsigned)it_bx);
    }
    // ---- publish: every store of this workgroup has left (write-through), then the counters move
    CHAIN_MARK(3u, idx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CHAIN_MARK(4u, idx);
    if (wave == 0) {
      if (lane == 0) {
        if (sig0 >= 0) __hip_atomic_fetch_add(counters + (size_t)sig0 * kCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sig1 >= 0) __hip_atomic_fetch_add(counters + (size_t)sig1 * kCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // ---- the last workgroup to leave restores the initial state (replay), keeping the first error for the host
  CHAIN_MARK(5u, 0u);
  if (wave == 0) {
    if (lane == 0) s_ctl[2] = __hip_atomic_fetch_add(&hdr->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (s_ctl[2] != 0u) {
    // (the K-split tile counters inside the spans reset themselves; after an aborted launch they may not have: clear everything)
    const bool aborted = __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (aborted)
      for (size_t i = tid; i < (size_t)n_counters * kCtrStride; i += NTHREADS) __hip_atomic_store(counters + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      for (unsigned i = tid; i < n_counters; i += NTHREADS) __hip_atomic_store(counters + (size_t)i * kCtrStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
      const unsigned e = __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0u && hdr->last_error == 0u) {
        hdr->last_error = e;
        unsigned* hp = reinterpret_cast<unsigned*>(((unsigned long long)hdr->host_hi << 32) | (unsigned long long)hdr->host_lo);
        if (hp != nullptr) __hip_atomic_store(hp, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      hdr->runs += 1u;
      __hip_atomic_store(&hdr->error, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hdr->next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hdr->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


#define CHAIN_KERNEL(name, attr, coh, dbg)                                                                                         \
  __global__ void __launch_bounds__(NTHREADS) attr name(ChainHeader* __restrict__ hdr, const ChainStage* __restrict__ stages, \
                                                        const ChainItem* __restrict__ items, unsigned* __restrict__ counters,  \
                                                        const unsigned* __restrict__ need, unsigned n_items, unsigned n_counters, \
                                                        unsigned* __restrict__ marks) {                                        \
    __shared__ __attribute__((aligned(16))) float lds[igemm_lds_floats<64, 64, 0>()];                                          \
    __shared__ unsigned s_ctl[4];                                                                                              \
    chain_body<coh, dbg>(hdr, stages, items, counters, need, n_items, n_counters, marks, lds, s_ctl);                          \
  }
CHAIN_KERNEL(chain_kernel, __attribute__((amdgpu_waves_per_eu(4, 4))), true, 0)
CHAIN_KERNEL(chain_kernel_noattr, , true, 1)
CHAIN_KERNEL(chain_kernel_nocoh, __attribute__((amdgpu_waves_per_eu(4, 4))), true, 2)

}  // namespace

#ifdef RADNET_DIAG_STAMPS
// diagnostic library only (not in include/radnet_hip.h): device buffer of 8 x u64 per workgroup of the next launches
extern "C" int radnet_diag_set_stamps(radnet_ctx* ctx, unsigned long long* dev_buf) {
  if (!ctx) return RADNET_ERR_ARG;
  ctx->diag_stamps = dev_buf;
  return RADNET_OK;
}
#endif

static int fwd_args(radnet_ctx* ctx, const radnet_conv_desc* d, GemmArgs& g) {
  if (!d->x || !d->w || !d->y) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_fwd: null tensor");
  g = GemmArgs{};
  g.x = d->x; g.w = d->w; g.y = d->y;
  g.scale = d->scale; g.shift = d->shift; g.addend = d->addend; g.mask = nullptr; g.in_scale = nullptr;
  g.H = d->h; g.W = d->w_; g.C = d->c; g.OH = d->oh; g.OW = d->ow;
  g.KW = d->kw; g.npos = d->kh * d->kw; g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
  g.M = d->nb * d->oh * d->ow; g.N = d->n; g.K = g.npos * d->c;
  g.ldw = d->ldw; g.ldy = d->ldy; g.ld_add = d->ld_add; g.ld_mask = 0;
  g.act = d->act; g.act_cols = d->act_cols; g.flip = 0; g.cin_fwd = 0;
  g.OHOW = d->oh * d->ow;
  // geometry sanity: every output pixel's window must be addressable by the gather's bounds checks
  if ((d->oh - 1) * d->stride - d->pad_t >= d->h || (d->ow - 1) * d->stride - d->pad_l >= d->w_)
    RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_fwd: output %dx%d inconsistent with input %dx%d", d->oh, d->ow, d->h, d->w_);
  return RADNET_OK;
}

extern "C" int radnet_conv_fwd(radnet_ctx* ctx, const radnet_conv_desc* d) {
  if (!ctx || !d) return RADNET_ERR_ARG;
  GemmArgs g;
  const int rc = fwd_args(ctx, d, g);
  return rc != RADNET_OK ? rc : run_igemm(ctx, g, 0, d->c == 4, 0);
}

// Two independent forward convolutions (same output grid and reduction depth: branch2a and the shortcut conv of a conv_block) as ONE
// launch where that measured faster than the two launches with their own tuned shapes (conv_fwd_pair_kernel; decided once per pair
// of shapes, kind 32 in the tuning table: slices 1 = paired with that tile, 2 = two launches); the two launches otherwise.
extern "C" int radnet_conv_fwd_pair(radnet_ctx* ctx, const radnet_conv_desc* d1, const radnet_conv_desc* d2) {
  if (!ctx || !d1 || !d2) return RADNET_ERR_ARG;
  GemmArgs g1, g2;
  int rc = fwd_args(ctx, d1, g1);
  if (rc == RADNET_OK) rc = fwd_args(ctx, d2, g2);
  if (rc != RADNET_OK) return rc;
  auto separate = [&]() -> int {
    GemmArgs a = g1, b = g2;
    int r = run_igemm(ctx, a, 0, d1->c == 4, 0);
    return r != RADNET_OK ? r : run_igemm(ctx, b, 0, d2->c == 4, 0);
  };
  static const bool disabled = radnet_env_flag("RADNET_NO_FWD_PAIR");
  const bool same_grid = g1.M == g2.M && g1.K == g2.K && g1.C == g2.C && g1.npos == g2.npos && g1.stride == g2.stride;
  if (disabled || !same_grid || d1->c == 4 || (g1.C % BK) != 0 || ctx->force_a > 0 || ctx->pair_capture != nullptr || !ctx->autotune) return separate();
  rc = prepare_igemm(ctx, g1, 0, false);
  if (rc == RADNET_OK) rc = prepare_igemm(ctx, g2, 0, false);
  if (rc != RADNET_OK) return rc;
  g1.units = g2.units = nullptr; g1.partial = g2.partial = nullptr; g1.counters = g2.counters = nullptr;
  g1.batch = g2.batch = 0; g1.xcd_batch = g2.xcd_batch = 0; g1.zper = g2.zper = 0;
  auto paired = [&](int bm, int bn) -> int {
    const unsigned gx1 = (unsigned)radnet_cdiv(g1.M, bm), gx2 = (unsigned)radnet_cdiv(g2.M, bm);
    const unsigned n1 = gx1 * (unsigned)radnet_cdiv(g1.N, bn), n2 = gx2 * (unsigned)radnet_cdiv(g2.N, bn);
    dim3 grid(n1 + n2), block(256);
    if (bm == 64 && bn == 64) RADNET_LAUNCH((conv_fwd_pair_kernel<64, 64>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g1, g2, n1, gx1, gx2);
    else if (bm == 32 && bn == 64) RADNET_LAUNCH((conv_fwd_pair_kernel<32, 64>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g1, g2, n1, gx1, gx2);
    else if (bm == 32 && bn == 32) RADNET_LAUNCH((conv_fwd_pair_kernel<32, 32>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g1, g2, n1, gx1, gx2);
    else return RADNET_ERR_UNSUPPORTED;
    RADNET_CHECK_LAUNCH(ctx, "conv_fwd_pair");
    return RADNET_OK;
  };
  const radnet_shape_key key{32, g1.M, g1.N * 65536 + g2.N, g1.K, g1.C, g1.npos, g1.stride};
  auto it = ctx->tuned->find(key);
  if (it == ctx->tuned->end()) {
    // first use of this pair of shapes: the two launches (which measures each of them, if new) against the pair on every tile it has
    rc = separate();
    if (rc != RADNET_OK) return rc;
    float best = 0.f;
    rc = radnet_time_launches(ctx, separate, 12, &best);
    if (rc != RADNET_OK) return rc;
    radnet_tuned choice{64, 64, 2, best, 4};
    const int tiles[3][2] = {{64, 64}, {32, 64}, {32, 32}};
    for (const auto& t : tiles) {
      float m1 = 0.f, m2 = 0.f;
      rc = radnet_time_launches(ctx, [&]() { return paired(t[0], t[1]); }, 12, &m1);
      if (rc == RADNET_OK) rc = radnet_time_launches(ctx, [&]() { return paired(t[0], t[1]); }, 12, &m2);
      if (rc != RADNET_OK) return rc;
      const float ms = std::min(m1, m2);
      if (ms < choice.ms) choice = radnet_tuned{t[0], t[1], 1, ms, 4};
    }
    (*ctx->tuned)[key] = choice;
    if (getenv("RADNET_TUNE_LOG"))
      fprintf(stderr, "[radnet tune] fwd pair M=%d N=%d+%d K=%d -> %s (%.1f us; the two launches %.1f us)\n", g1.M, g1.N, g2.N, g1.K,
              choice.splits == 1 ? (choice.a == 64 ? "one launch, 64x64" : choice.b == 64 ? "one launch, 32x64" : "one launch, 32x32") : "two launches",
              choice.ms * 1e3, best * 1e3);
    it = ctx->tuned->find(key);
  }
  if (it->second.splits != 1) return separate();
  radnet_timing_arm(ctx);
  rc = paired(it->second.a, it->second.b);
  if (rc == RADNET_ERR_UNSUPPORTED) return separate();
  if (rc != RADNET_OK) return rc;
  radnet_timing_end_armed(ctx, 0, 2.0 * g1.M * (double)(g1.N + g2.N) * g1.K);
  return RADNET_OK;
}

// The back of a bottleneck block as one launch (conv_bneck_kernel): db = its 3x3 conv (stride 1, 'same', 64 output channels, ReLU), dc = its
// 1x1 expand on db's output (+ shortcut, ReLU), da = the NEXT block's 1x1 reduce on dc's output (64 columns, ReLU) or null.  db->y is NOT
// written by the fused launch (frozen layers only).  Decided once per shape (kind 33 in the tuning table: slices 1 = fused with tile_a rows
// per workgroup, 2 = the separate launches); anything the fused kernel does not take runs as the separate launches.
extern "C" int radnet_conv_bottleneck(radnet_ctx* ctx, const radnet_conv_desc* db, const radnet_conv_desc* dc, const radnet_conv_desc* da) {
  if (!ctx || !db || !dc) return RADNET_ERR_ARG;
  auto separate = [&]() -> int {
    int r = radnet_conv_fwd(ctx, db);
    if (r == RADNET_OK) r = radnet_conv_fwd(ctx, dc);
    if (r == RADNET_OK && da) r = radnet_conv_fwd(ctx, da);
    return r;
  };
  static const bool disabled = radnet_env_flag("RADNET_NO_BNECK_FUSE");
  const int M = db->nb * db->oh * db->ow;
  auto pointwise = [&](const radnet_conv_desc* d, const radnet_conv_desc* src) {
    return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->x == src->y && d->c == src->n &&
           d->nb * d->oh * d->ow == M && d->act == 1 && d->y != nullptr && d->w != nullptr && (d->ldw & 3) == 0 && !((uintptr_t)d->w & 15);
  };
  bool ok = !disabled && ctx->autotune && ctx->force_a <= 0 && ctx->pair_capture == nullptr;
  ok = ok && db->kh == 3 && db->kw == 3 && db->stride == 1 && db->pad_t == 1 && db->pad_l == 1 && db->n == 64 && db->act == 1 && !db->addend && db->y &&
       db->c % BK == 0 && db->oh == db->h && db->ow == db->w_;
  ok = ok && pointwise(dc, db) && dc->n % 64 == 0 && dc->ldw >= dc->n && dc->ldy >= dc->n && (!dc->addend || dc->ld_add >= dc->n);
  ok = ok && (!da || (pointwise(da, dc) && da->n == 64 && !da->addend && da->ldw >= 64 && da->ldy >= 64));
  ok = ok && (uint64_t)M * (uint64_t)std::max(dc->ldy, dc->ld_add) * 4ull < (1ull << 31);
  if (!ok) return separate();
  GemmArgs g;
  int rc = fwd_args(ctx, db, g);
  if (rc == RADNET_OK) rc = prepare_igemm(ctx, g, 0, false);
  if (rc != RADNET_OK) return rc;
  g.y = nullptr; g.y_bytes = 0;
  TailArgs tz{};
  tz.w2 = dc->w; tz.sc2 = dc->scale; tz.sh2 = dc->shift; tz.add = dc->addend; tz.y = dc->y;
  tz.N2 = dc->n; tz.ldw2 = dc->ldw; tz.ldy2 = dc->ldy; tz.ld_add2 = dc->ld_add;
  tz.w2_bytes = (unsigned)(((uint64_t)63 * dc->ldw + dc->n) * 4ull);
  tz.y_bytes = (unsigned)(((uint64_t)(M - 1) * dc->ldy + dc->n) * 4ull);
  tz.add_bytes = dc->addend ? (unsigned)(((uint64_t)(M - 1) * dc->ld_add + dc->n) * 4ull) : 0u;
  if (da) {
    tz.w3 = da->w; tz.sc3 = da->scale; tz.sh3 = da->shift; tz.t = da->y; tz.ldw3 = da->ldw; tz.ldt = da->ldy;
    tz.w3_bytes = (unsigned)(((uint64_t)(dc->n - 1) * da->ldw + 64) * 4ull);
    tz.t_bytes = (unsigned)(((uint64_t)(M - 1) * da->ldy + 64) * 4ull);
  }
  auto fused = [&](int bm) -> int {
    dim3 grid(radnet_cdiv(M, bm));
    if (bm == 64) {
      if (da) RADNET_LAUNCH((conv_bneck_kernel<64, 4, 2>), grid, dim3(256), 0, ctx->stream, ctx->arm0, ctx->arm1, g, tz);
      else RADNET_LAUNCH((conv_bneck_kernel<64, 4, 1>), grid, dim3(256), 0, ctx->stream, ctx->arm0, ctx->arm1, g, tz);
    } else if (bm == 32) {
      if (da) RADNET_LAUNCH((conv_bneck_kernel<32, 2, 2>), grid, dim3(128), 0, ctx->stream, ctx->arm0, ctx->arm1, g, tz);
      else RADNET_LAUNCH((conv_bneck_kernel<32, 2, 1>), grid, dim3(128), 0, ctx->stream, ctx->arm0, ctx->arm1, g, tz);
    } else {
      return RADNET_ERR_UNSUPPORTED;
    }
    RADNET_CHECK_LAUNCH(ctx, "conv_bneck");
    return RADNET_OK;
  };
  const radnet_shape_key key{33, M, dc->n * 65536 + (da ? 64 : 0), g.K, g.C, g.npos, g.stride};
  auto it = ctx->tuned->find(key);
  if (it == ctx->tuned->end()) {
    rc = separate();                     // measures the layers' own shapes, if new
    if (rc != RADNET_OK) return rc;
    float best = 0.f;
    rc = radnet_time_launches(ctx, separate, 12, &best);
    if (rc != RADNET_OK) return rc;
    radnet_tuned choice{64, 64, 2, best, 4};
    for (int bm : {64, 32}) {
      float m1 = 0.f, m2 = 0.f;
      rc = radnet_time_launches(ctx, [&]() { return fused(bm); }, 12, &m1);
      if (rc == RADNET_OK) rc = radnet_time_launches(ctx, [&]() { return fused(bm); }, 12, &m2);
      if (rc != RADNET_OK) return rc;
      const float ms = std::min(m1, m2);
      if (getenv("RADNET_TUNE_LOG")) fprintf(stderr, "[radnet tune] bottleneck tail M=%d N2=%d%s: fused %d rows %.1f us\n", M, dc->n, da ? "+64" : "", bm, ms * 1e3);
      if (ms < choice.ms) choice = radnet_tuned{bm, 64, 1, ms, 4};
    }
    (*ctx->tuned)[key] = choice;
    if (getenv("RADNET_TUNE_LOG"))
      fprintf(stderr, "[radnet tune] bottleneck tail M=%d N2=%d%s -> %s (%.1f us; the separate launches %.1f us)\n", M, dc->n, da ? "+64" : "",
              choice.splits == 1 ? (choice.a == 64 ? "one launch, 64 rows" : "one launch, 32 rows") : "separate launches", choice.ms * 1e3, best * 1e3);
    it = ctx->tuned->find(key);
  }
  if (it->second.splits != 1) return separate();
  radnet_timing_arm(ctx);
  rc = fused(it->second.a);
  if (rc == RADNET_ERR_UNSUPPORTED) return separate();
  if (rc != RADNET_OK) return rc;
  radnet_timing_end_armed(ctx, 0, 2.0 * M * (64.0 * g.K + 64.0 * dc->n + (da ? 64.0 * dc->n : 0.0)));
  return RADNET_OK;
}

extern "C" int radnet_gemm_batched(radnet_ctx* ctx, const float* a, const float* b, float* y, int32_t batch, int32_t m, int32_t n, int32_t k) {
  if (!ctx || !a || !b || !y) return RADNET_ERR_ARG;
  if (batch < 1 || batch > 65535) RADNET_FAIL(ctx, RADNET_ERR_ARG, "gemm_batched: batch %d", batch);
  GemmArgs g{};
  g.x = a; g.w = b; g.y = y;
  g.H = 1; g.W = m; g.C = k; g.OH = 1; g.OW = m;           // a 1x1 convolution over m 'pixels' of k channels
  g.KW = 1; g.npos = 1; g.stride = 1; g.pad_t = 0; g.pad_l = 0;
  g.M = m; g.N = n; g.K = k;
  g.ldw = n; g.ldy = n; g.ld_add = 0; g.ld_mask = 0;
  g.OHOW = m;
  g.batch = batch;
  g.x_bstride = (long long)m * k; g.w_bstride = (long long)k * n; g.y_bstride = (long long)m * n;
  return run_igemm(ctx, g, 0, false, 0);
}

extern "C" int radnet_conv_dgrad(radnet_ctx* ctx, const radnet_conv_desc* d) {
  if (!ctx || !d) return RADNET_ERR_ARG;
  if (!d->dy || !d->w || !d->dx) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_dgrad: null tensor");
  if (d->stride != 1) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv_dgrad: stride %d (only 1)", d->stride);
  if (d->ld_dy != d->n) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_dgrad: dy must be dense (ld_dy == n)");
  GemmArgs g{};
  g.x = d->dy; g.w = d->w; g.y = d->dx;
  g.scale = nullptr; g.shift = nullptr; g.addend = d->dx_add; g.mask = d->dx_mask; g.in_scale = d->gscale;
  // gather from dy (nb, oh, ow, n) with the kernel flipped; output pixels = forward input pixels
  g.H = d->oh; g.W = d->ow; g.C = d->n; g.OH = d->h; g.OW = d->w_;
  g.KW = d->kw; g.npos = d->kh * d->kw; g.stride = 1; g.pad_t = d->kh - 1 - d->pad_t; g.pad_l = d->kw - 1 - d->pad_l;
  g.M = d->nb * d->h * d->w_; g.N = d->c; g.K = g.npos * d->n;
  g.ldw = d->ldw; g.ldy = d->ld_dx; g.ld_add = d->ld_dx_add; g.ld_mask = d->ld_dx_mask;
  g.act = 0; g.act_cols = 0; g.flip = 1; g.cin_fwd = d->c;
  g.OHOW = d->h * d->w_;
  if (d->n % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_dgrad: n must be a multiple of 4");
  return run_igemm(ctx, g, 1, false, 1);
}

static int run_wgrad(radnet_ctx* ctx, const radnet_conv_desc* d, int batch, long long x_bs, long long dy_bs, long long dw_bs) {
  if (!ctx || !d) return RADNET_ERR_ARG;
  if (!d->x || !d->dy || !d->dw) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_wgrad: null tensor");
  WgradArgs g{};
  g.batch = batch; g.x_bstride = x_bs; g.dy_bstride = dy_bs; g.dw_bstride = dw_bs;
  g.x = d->x; g.dy = d->dy; g.gscale = d->gscale; g.dw = d->dw; g.db = nullptr;
  g.H = d->h; g.W = d->w_; g.C = d->c; g.OH = d->oh; g.OW = d->ow; g.KW = d->kw;
  g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
  g.M = d->nb * d->oh * d->ow; g.N = d->n; g.K = d->kh * d->kw * d->c;
  g.ld_dy = d->ld_dy; g.ldw = d->ldw;
  if (g.M >= (1 << 20)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv_wgrad: M=%d exceeds 2^20", g.M);
  if ((g.N & 3) || (g.ld_dy & 3) || (g.ldw & 3)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_wgrad: n, ld_dy, ldw must be multiples of 4");
  if (d->h >= 32768 || d->w_ >= 32768) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv_wgrad: input %dx%d exceeds the 16-bit row table", d->h, d->w_);
  g.rowtab = get_row_table(ctx, d);
  if (!g.rowtab) RADNET_FAIL(ctx, RADNET_ERR_HIP, "conv_wgrad: cannot build the row table");
  g.mpad = radnet_cdiv(g.M, BK) * BK;
  g.x_bias = (unsigned)(((int64_t)d->pad_t * d->w_ + d->pad_l) * d->c * 4);
  {
    const uint64_t xb = (uint64_t)d->nb * d->h * d->w_ * d->c * 4ull, db = (uint64_t)g.M * g.ld_dy * 4ull;
    if (xb + g.x_bias >= (1ull << 31) || db >= (1ull << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv_wgrad: tensor larger than 2 GiB");
    g.x_bytes = (unsigned)xb;
    g.dy_bytes = (unsigned)db;
  }
  if (d->c % 64) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv_wgrad: channels %d not a multiple of 64", d->c);
  if ((uint64_t)g.K * (uint64_t)g.ldw * 4ull >= (1ull << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "conv_wgrad: weight tensor larger than 2 GiB");
  const int nmt = radnet_cdiv(g.M, BK);
  uint64_t wgrad_slab_bytes = 0;
  auto launch = [&](int bmk, int bn, int splits) -> int {
    g.xcd_batch = (batch > 1 && splits < 0) ? 1 : 0;      // a batch: -s = the same grid, XCD-contiguous numbering
    if (splits < 0) {
      if (batch <= 1) return RADNET_ERR_UNSUPPORTED;
      splits = -splits;
    }
    wgrad_slab_bytes = 0;
    g.mt_per_split = radnet_cdiv(nmt, splits);
    g.splits = splits;
    // dw_accumulate: 0 = overwrite, 1 = add to existing contents, 2 = destination is pre-zeroed by the caller
    // (plain stores when un-split, atomics without the memset when split)
    g.atomic = (splits > 1 || d->dw_accumulate == 1) ? 1 : 0;
    dim3 grid(radnet_cdiv(g.K, bmk), radnet_cdiv(g.N, bn), splits * (batch > 1 ? batch : 1)), block(NTHREADS);
    g.slabs = nullptr;
    g.counters = nullptr;
    g.tiles_x = (int)grid.x; g.tiles_y = (int)grid.y;
    g.accumulate = d->dw_accumulate == 1;
    if (splits > 1 && ctx->deterministic) {
      // ordered reduction: slabs at the END of the workspace (a dgrad launch paired with this one keeps its split-K slabs at the start)
      const uint64_t tiles = (uint64_t)grid.x * grid.y * (batch > 1 ? batch : 1);
      const uint64_t need = (tiles * splits * (uint64_t)(bmk * bn) + (uint64_t)grid.y * (batch > 1 ? batch : 1) * splits * bn) * sizeof(float);
      if (tiles > kAuxWgradCounterCount || ctx->ws == nullptr || need > ctx->ws_bytes) return RADNET_ERR_UNSUPPORTED;      // candidate skipped
      g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->ws) + ((ctx->ws_bytes - need) & ~(uint64_t)255));
      g.counters = reinterpret_cast<unsigned*>(ctx->aux + kAuxWgradCounters);
      g.atomic = 0;
      wgrad_slab_bytes = need + 256;
    } else if (splits > 1 && d->dw_accumulate == 0) {  // atomics need a zeroed destination
      RADNET_CHECK_HIP(ctx, hipMemsetAsync(d->dw, 0, (batch > 1 ? (size_t)batch * dw_bs : (size_t)g.K * g.ldw) * sizeof(float), ctx->stream));
    }
    if (ctx->pair_capture != nullptr) {
      PairCapture* pc = (PairCapture*)ctx->pair_capture;
      pc->have_w = true;
      pc->w_ok = bmk == 64 && bn == 64 && batch <= 1;
      pc->gw = g;
      pc->w_slab_bytes = wgrad_slab_bytes;
      pc->wx = grid.x; pc->wy = grid.y; pc->wz = grid.z;
      pc->flops += 2.0 * g.M * g.N * g.K;
      return RADNET_OK;
    }
    if (bmk == 128 && bn == 128) RADNET_LAUNCH((conv_wgrad_kernel<128, 128>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g);
    else if (bmk == 128 && bn == 64) RADNET_LAUNCH((conv_wgrad_kernel<128, 64>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g);
    else if (bmk == 64 && bn == 128) RADNET_LAUNCH((conv_wgrad_kernel<64, 128>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g);
    else RADNET_LAUNCH((conv_wgrad_kernel<64, 64>), grid, block, 0, ctx->stream, ctx->arm0, ctx->arm1, g);
    RADNET_CHECK_LAUNCH(ctx, "conv_wgrad");
    return RADNET_OK;
  };
  int bmk = (d->c % 128 == 0) ? 128 : 64, bn = g.N > 64 ? 128 : 64, splits = 1;
  const radnet_shape_key key{2 + (d->dw_accumulate == 1 ? 1 : 0) + (batch > 1 ? 16 : 0), g.M, g.N, g.K, g.C, d->kh * d->kw, batch > 1 ? batch : g.stride};
  auto it = ctx->tuned->find(key);
  if (ctx->force_a > 0) {
    bmk = ctx->force_a; bn = ctx->force_b; splits = ctx->force_splits < 1 ? 1 : ctx->force_splits;
    if (bmk < 64) bmk = 64;      // the 32-row tiles are the forward / data-gradient kernel's: a forced 32x64 leaves the weight gradient at 64x64
    if (bn < 64) bn = 64;        // (radnet_conv_bwd then pairs a 32x64 data gradient with a 64x64 weight gradient)
    if (d->c % bmk) RADNET_FAIL(ctx, RADNET_ERR_ARG, "conv_wgrad: forced k tile %d does not divide c=%d", bmk, d->c);
  } else if (it != ctx->tuned->end()) {
    bmk = it->second.a; bn = it->second.b; splits = it->second.splits;
  } else if (const radnet_tuned* nb = (ctx->autotune == 2 && d->dw_accumulate != 1) ? radnet_tuned_neighbour(*ctx->tuned, key) : nullptr;
             nb && (nb->splits <= 1 || (nmt / nb->splits >= 2 && radnet_cdiv(nmt, radnet_cdiv(nmt, nb->splits)) == nb->splits))) {
    bmk = nb->a; bn = nb->b; splits = nb->splits;
    (*ctx->tuned)[key] = *nb;
  } else if (ctx->autotune && d->dw_accumulate != 1) {
    PairPause pause(ctx);                       // trial launches are real launches
    struct WCand { float ms; int bmk, bn, s; };
    std::vector<WCand> seen;
    const bool for_pair = pause.saved != nullptr && batch <= 1 && d->dx != nullptr;      // see run_igemm
    for (int cb = 128; cb >= 64; cb -= 64) {
      if (d->c % cb) continue;
      if (for_pair && cb != 64) continue;
      for (int cn = 128; cn >= 64; cn -= 64) {
        if (cn > 64 && g.N <= 64) continue;
        if (for_pair && cn != 64) continue;
        for (int s0 : {1, 2, 3, 4, 6, 8, 12, 16}) {
          if (s0 > 1 && (nmt / s0 < 2 || radnet_cdiv(nmt, radnet_cdiv(nmt, s0)) != s0)) continue;
          for (int s = s0; s >= (batch > 1 ? -s0 : s0); s -= 2 * s0) {      // a batch: also -s, the XCD-contiguous numbering
            float ms = 0.f;
            int rc = radnet_time_launches(ctx, [&]() { return launch(cb, cn, s); }, 3, &ms);
            if (rc == RADNET_ERR_UNSUPPORTED) continue;      // ordered reduction: slabs larger than the workspace
            if (rc != RADNET_OK) return rc;
            seen.push_back(WCand{ms, cb, cn, s});
          }
        }
      }
    }
    std::sort(seen.begin(), seen.end(), [](const WCand& a, const WCand& b) { return a.ms < b.ms; });
    float best = 1e30f;
    for (size_t i = 0; i < seen.size() && i < 4; ++i) {        // finalists again, longer and twice (see run_igemm)
      float m1 = 0.f, m2 = 0.f;
      int rc = radnet_time_launches(ctx, [&]() { return launch(seen[i].bmk, seen[i].bn, seen[i].s); }, 12, &m1);
      if (rc == RADNET_OK) rc = radnet_time_launches(ctx, [&]() { return launch(seen[i].bmk, seen[i].bn, seen[i].s); }, 12, &m2);
      if (rc != RADNET_OK) return rc;
      const float ms = std::min(m1, m2);
      if (ms < best) { best = ms; bmk = seen[i].bmk; bn = seen[i].bn; splits = seen[i].s; }
    }
    (*ctx->tuned)[key] = radnet_tuned{bmk, bn, splits, best, 4};
    if (getenv("RADNET_TUNE_LOG"))
      fprintf(stderr, "radnet tune: wgrad M=%d N=%d K=%d C=%d -> tile %dx%d slices %d : %.1f us (%.1f TFLOP/s)\n", g.M, g.N, g.K, g.C,
              bmk, bn, splits, best * 1e3, 2.0 * g.M * g.N * g.K / (best * 1e9));
    if (d->dw_accumulate == 2)               // the trial launches added into the pre-zeroed buffer: restore it
      RADNET_CHECK_HIP(ctx, hipMemsetAsync(d->dw, 0, (size_t)g.K * g.ldw * sizeof(float), ctx->stream));
  } else {
    // accumulate mode reuses the overwrite-mode measurement when there is one
    const radnet_shape_key k0{2, g.M, g.N, g.K, g.C, d->kh * d->kw, g.stride};
    auto it0 = ctx->tuned->find(k0);
    if (it0 != ctx->tuned->end()) {
      bmk = it0->second.a; bn = it0->second.b; splits = it0->second.splits;
    } else {
      long long tiles = (long long)radnet_cdiv(g.K, bmk) * radnet_cdiv(g.N, bn);
      if (tiles < kNumCU && bmk == 128 && bn == 128) {
        bn = 64;
        tiles = (long long)radnet_cdiv(g.K, bmk) * radnet_cdiv(g.N, bn);
      }
      while (tiles * splits < 2 * kNumCU && nmt / (splits * 2) >= 4 && splits < 16) splits *= 2;
    }
  }
  radnet_timing_arm(ctx);
  {
    // bias gradient in the same launch (the measurement launches above ran without it); atomics need zeros to add to
    g.db = d->db;
    if (d->db && d->dw_accumulate == 0) RADNET_CHECK_HIP(ctx, hipMemsetAsync(d->db, 0, (size_t)g.N * sizeof(float), ctx->stream));
    int rc = launch(bmk, bn, splits);
    if (rc == RADNET_ERR_UNSUPPORTED && splits < 0) {      // a loaded / shared choice this launch cannot use as it is
      splits = -splits;
      rc = launch(bmk, bn, splits);
    }
    if (rc == RADNET_ERR_UNSUPPORTED && splits > 1) {      // a forced / shared / loaded choice whose slabs exceed THIS context's workspace
      while (rc == RADNET_ERR_UNSUPPORTED && splits > 1) {
        splits = splits > 2 ? splits / 2 : 1;
        while (splits > 1 && radnet_cdiv(nmt, radnet_cdiv(nmt, splits)) != splits) --splits;      // no empty split
        rc = launch(bmk, bn, splits);
      }
    }
    if (rc == RADNET_ERR_UNSUPPORTED) RADNET_FAIL(ctx, rc, "conv_wgrad: no launch shape fits (tile %dx%d, workspace %llu bytes)", bmk, bn, (unsigned long long)ctx->ws_bytes);
    if (rc != RADNET_OK) return rc;
  }
  RADNET_CHECK_LAUNCH(ctx, "conv_wgrad");
  radnet_timing_end_armed(ctx, 2, 2.0 * g.M * g.N * g.K * (batch > 1 ? batch : 1));
  return RADNET_OK;
}

extern "C" int radnet_conv_wgrad(radnet_ctx* ctx, const radnet_conv_desc* d) { return run_wgrad(ctx, d, 1, 0, 0, 0); }

// Weight gradient and data gradient of one layer (the same descriptor: both read dy) as ONE launch when both problems run as
// 64x64-tile, 4-wave workgroups (conv_bwd_pair_kernel); otherwise -- other tile choices, or RADNET_NO_BWD_PAIR=1 -- the two
// launches in the order wgrad, dgrad.  Results are those of the separate launches (same kernels' code, same launch shapes).
extern "C" int radnet_conv_bwd(radnet_ctx* ctx, const radnet_conv_desc* d) {
  if (!ctx || !d) return RADNET_ERR_ARG;
  static const bool disabled = radnet_env_flag("RADNET_NO_BWD_PAIR");
  if (disabled || ctx->pair_capture != nullptr || !d->dx) {
    int rc = radnet_conv_wgrad(ctx, d);
    return rc != RADNET_OK || !d->dx ? rc : radnet_conv_dgrad(ctx, d);
  }
  PairCapture pc;
  const int timed = ctx->timing;
  ctx->timing = 0;
  ctx->pair_capture = &pc;
  int rc = radnet_conv_wgrad(ctx, d);           // host-side preparation (tables, memsets of overwrite mode) happens here
  if (rc == RADNET_OK) rc = radnet_conv_dgrad(ctx, d);
  ctx->pair_capture = nullptr;
  ctx->timing = timed;
  if (rc != RADNET_OK) return rc;
  if (pc.a_slab_bytes + pc.w_slab_bytes > ctx->ws_bytes) pc.a_ok = false;      // the two problems' slabs would overlap in the workspace
  if (!(pc.have_a && pc.have_w && pc.a_ok && pc.w_ok)) {      // not the fusable shapes: issue them one after the other
    rc = radnet_conv_wgrad(ctx, d);
    return rc != RADNET_OK ? rc : radnet_conv_dgrad(ctx, d);
  }
  PairMap pm{pc.ax * pc.ay, pc.wx * pc.wy * pc.wz, pc.ax, pc.ay, pc.wx, pc.wy};
  radnet_timing_arm(ctx);
  if (pc.a_bm == 32) RADNET_LAUNCH(conv_bwd_pair_kernel<32>, dim3(pm.n_a + pm.n_w), dim3(NTHREADS), 0, ctx->stream, ctx->arm0, ctx->arm1, pc.ga, pc.gw, pm);
  else RADNET_LAUNCH(conv_bwd_pair_kernel<64>, dim3(pm.n_a + pm.n_w), dim3(NTHREADS), 0, ctx->stream, ctx->arm0, ctx->arm1, pc.ga, pc.gw, pm);
  RADNET_CHECK_LAUNCH(ctx, "conv_bwd_pair");
  radnet_timing_end_armed(ctx, 4, pc.flops);
  return RADNET_OK;
}

extern "C" int radnet_wgrad_batched(radnet_ctx* ctx, const float* a, const float* dy, float* dw, int32_t batch, int32_t m, int32_t k, int32_t n,
                                    int32_t accumulate) {
  if (!ctx || !a || !dy || !dw) return RADNET_ERR_ARG;
  if (batch < 1 || batch > 4096) RADNET_FAIL(ctx, RADNET_ERR_ARG, "wgrad_batched: batch %d", batch);
  radnet_conv_desc d{};
  d.x = a; d.dy = dy; d.dw = dw;
  d.nb = 1; d.h = 1; d.w_ = m; d.c = k; d.oh = 1; d.ow = m;      // a 1x1 convolution over m 'pixels' of k channels
  d.kh = 1; d.kw = 1; d.stride = 1; d.pad_t = 0; d.pad_l = 0; d.n = n;
  d.ldw = n; d.ld_dy = n;
  d.dw_accumulate = accumulate;
  return run_wgrad(ctx, &d, batch, (long long)m * k, (long long)m * n, (long long)k * n);
}

// ---- host side: radnet_op[] -> stages, items, counters ----------------------------------------------------------------
struct radnet_chain {
  ChainHeader* d_hdr = nullptr;
  ChainStage* d_stages = nullptr;
  ChainItem* d_items = nullptr;
  unsigned* d_counters = nullptr;
  unsigned* d_need = nullptr;
  int* d_units = nullptr;
  float* d_slabs = nullptr;
  unsigned* d_marks = nullptr;      // RADNET_CHAIN_DEBUG=1: one word per wave (phase, item)
  unsigned* h_err = nullptr;        // mapped host word: first 'gave up waiting' error of any launch (1 + item), sticky
  unsigned n_items = 0, n_counters = 0, n_stages = 0;
  int grid = 0;
  std::vector<ChainItem> h_items;       // host copies for radnet_chain_peek / diagnosis
  std::vector<unsigned> h_need;
  double flops = 0.0;            // executed by the matrix cores
  double flops_algorithmic = 0.0;   // 2 M N K of the layers as direct convolutions (Winograd layers credited 9 C per output)
};

namespace {

// where a stage's output can be waited for
struct ChainOut {
  enum Kind { ROWS64, TILES64, TILEROWS } kind = ROWS64;
  int first = 0, count = 0;       // its counters
  int h = 0, w = 0, th = 0, tw = 0;   // TILEROWS: output geometry (pixels, tiles)
};

template <typename E>
int chain_conv_args(E* ctx, const radnet_conv_desc* d, GemmArgs& g) {
  if (!d->x || !d->w || !d->y) RADNET_FAIL(ctx, RADNET_ERR_ARG, "chain: conv with a null tensor");
  g = GemmArgs{};
  g.x = d->x; g.w = d->w; g.y = d->y;
  g.scale = d->scale; g.shift = d->shift; g.addend = d->addend;
  g.H = d->h; g.W = d->w_; g.C = d->c; g.OH = d->oh; g.OW = d->ow;
  g.KW = d->kw; g.npos = d->kh * d->kw; g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
  g.M = d->nb * d->oh * d->ow; g.N = d->n; g.K = g.npos * d->c;
  g.ldw = d->ldw; g.ldy = d->ldy; g.ld_add = d->ld_add;
  g.act = d->act; g.act_cols = d->act_cols;
  g.OHOW = d->oh * d->ow;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M >= (1 << 20) || (g.ldw & 3) || (g.N & 3) || (g.C % BK) != 0 || g.npos > 32)
    RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "chain: conv M=%d N=%d K=%d C=%d taps=%d cannot run as chain items", g.M, g.N, g.K, g.C, g.npos);
  if (((uintptr_t)g.x & 15) || ((uintptr_t)g.w & 15)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "chain: x / w must be 16-byte aligned");
  g.magic_ohow = radnet_div_magic((uint32_t)g.OHOW);
  g.magic_ow = radnet_div_magic((uint32_t)g.OW);
  const uint64_t xb = (uint64_t)d->nb * g.H * g.W * g.C * 4ull, wb = (uint64_t)g.K * g.ldw * 4ull;
  const uint64_t halo = ((uint64_t)g.pad_t * g.W + g.pad_l) * g.C * 4ull;
  const uint64_t ld_max = (uint64_t)std::max(g.ldy, g.addend ? g.ld_add : 0);
  if (xb + halo >= (1ull << 31) || wb >= (1ull << 31) || (uint64_t)g.M * ld_max * 4ull >= (1ull << 31))
    RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "chain: tensor larger than 2 GiB");
  if (g.ldy < g.N || (g.addend && g.ld_add < g.N)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "chain: row pitch smaller than n=%d", g.N);
  g.x_bytes = (unsigned)xb;
  g.w_bytes = (unsigned)wb;
  g.y_bytes = (unsigned)(((uint64_t)(g.M - 1) * g.ldy + g.N) * 4ull);
  g.add_bytes = g.addend ? (unsigned)(((uint64_t)(g.M - 1) * g.ld_add + g.N) * 4ull) : 0u;
  return RADNET_OK;
}

}  // namespace

extern "C" void radnet_chain_destroy(radnet_chain* ch) {
  if (!ch) return;
  for (void* p : {(void*)ch->d_hdr, (void*)ch->d_stages, (void*)ch->d_items, (void*)ch->d_counters, (void*)ch->d_need, (void*)ch->d_units, (void*)ch->d_slabs, (void*)ch->d_marks})
    if (p) (void)hipFree(p);
  if (ch->h_err) (void)hipHostFree(ch->h_err);
  delete ch;
}

struct ChainPlan {
  std::vector<ChainStage> stages;
  std::vector<ChainItem> items;
  std::vector<unsigned> need;                   // per counter
  std::vector<int> units;                       // all K-split unit tables, 8 ints per unit
  std::vector<size_t> unit_base;                // per stage: first int of its table (or ~0)
  std::vector<size_t> slab_base;                // per stage (floats)
  size_t slabs_total = 0;
  double flops = 0.0, flops_alg = 0.0;
};
struct ErrSink {                                // RADNET_FAIL needs ->err
  char err[512];
};

// Host-only: the work-item list of a program (no device call; radnet_chain_check runs it without a GPU).
static int chain_plan(const radnet_op* ops, int32_t n_ops, ChainPlan& pl, ErrSink& ec) {
  std::vector<ChainStage>& stages = pl.stages;
  std::vector<ChainOut> outs;                   // per stage
  std::vector<ChainItem>& items = pl.items;
  std::vector<unsigned>& need = pl.need;
  std::vector<int>& units = pl.units;
  std::vector<size_t>& unit_base = pl.unit_base;
  std::vector<size_t>& slab_base = pl.slab_base;
  std::vector<size_t> slab_floats;              // per stage (floats)
  std::map<const void*, int> producer;          // pixel tensor -> stage that writes it
  size_t& slabs_total = pl.slabs_total;
  double& flops = pl.flops;
  double& flops_alg = pl.flops_alg;

  auto new_counters = [&](int n, unsigned want) {
    const int first = (int)need.size();
    need.insert(need.end(), (size_t)n, want);
    return first;
  };
  // counters of `p` that cover rows [r0, r1] of the pixel tensor it writes -> (first, count)
  auto rows_dep = [&](int p, int r0, int r1, int& first, int& count) {
    const ChainOut& o = outs[p];
    if (o.kind == ChainOut::ROWS64) {
      first = o.first + r0 / 64;
      count = r1 / 64 - r0 / 64 + 1;
    } else {                                    // TILEROWS: one counter per (image, tile row)
      const int y0 = r0 / o.w, y1 = r1 / o.w;   // global pixel row = image * h + oh
      const int t0 = (y0 / o.h) * o.th + (y0 % o.h) / 4, t1 = (y1 / o.h) * o.th + (y1 % o.h) / 4;
      first = o.first + t0;
      count = t1 - t0 + 1;
    }
  };
  auto push_stage = [&](const ChainStage& st, const ChainOut& o) {
    stages.push_back(st);
    outs.push_back(o);
    unit_base.push_back(~(size_t)0);
    slab_base.push_back(0);
    slab_floats.push_back(0);
    return (int)stages.size() - 1;
  };
  // items of one conv / batched-GEMM stage; dep(tm, first, count) gives the counters tile row tm waits for
  auto emit_gemm = [&](int si, int batch, int dep_stage_main, int dep_stage_add, bool batched_dep) -> int {
    GemmArgs& g = stages[si].g;
    const int Mt = radnet_cdiv(g.M, 64), Nt = radnet_cdiv(g.N, 64), nk = radnet_cdiv(g.K, BK);
    int S = 1;
    if (batch <= 1) {
      const long long tiles = (long long)Mt * Nt;
      if (tiles < 384) S = (int)std::min<long long>(std::max(nk / 4, 1), (512 + tiles - 1) / tiles);
      const int kt = radnet_cdiv(nk, S);
      S = radnet_cdiv(nk, kt);
    }
    const int kt = radnet_cdiv(nk, S);
    ChainOut& o = outs[si];
    o.kind = batch > 1 ? ChainOut::TILES64 : ChainOut::ROWS64;
    o.count = Mt;
    o.first = new_counters(Mt, (unsigned)(Nt * S * (batch > 1 ? batch : 1)));
    int split_counters = -1;
    if (S > 1) {
      unit_base[si] = units.size();
      slab_base[si] = slabs_total;
      slab_floats[si] = (size_t)Mt * Nt * S * 4096;
      slabs_total += slab_floats[si];
      split_counters = new_counters(Mt * Nt, 0u);      // arrival counters of the in-launch reduction (never polled)
    }
    for (int tm = 0; tm < Mt; ++tm) {
      int d0f = 0, d0n = 0, d1f = 0, d1n = 0;
      if (batched_dep) {                        // batched Winograd GEMM: tile block tm of the transformed operand
        d0f = outs[dep_stage_main].first + tm;
        d0n = 1;
      } else {
        if (dep_stage_main >= 0) {
          // rows of the producer this tile's windows touch
          int r0 = INT32_MAX, r1 = -1;
          for (int m = tm * 64; m < std::min(g.M, tm * 64 + 64); ++m) {
            const int img = m / g.OHOW, rem = m % g.OHOW, oh = rem / g.OW, ow = rem % g.OW;
            const int ih0 = std::max(oh * g.stride - g.pad_t, 0), iw0 = std::max(ow * g.stride - g.pad_l, 0);
            const int ih1 = std::min(oh * g.stride - g.pad_t + (g.npos / g.KW) - 1, g.H - 1), iw1 = std::min(ow * g.stride - g.pad_l + g.KW - 1, g.W - 1);
            r0 = std::min(r0, (img * g.H + ih0) * g.W + iw0);
            r1 = std::max(r1, (img * g.H + ih1) * g.W + iw1);
          }
          rows_dep(dep_stage_main, r0, r1, d0f, d0n);
        }
        if (dep_stage_add >= 0) rows_dep(dep_stage_add, tm * 64, std::min(g.M, tm * 64 + 64) - 1, d1f, d1n);
      }
      if (d0n + d1n > 64) RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: an item would wait for %d blocks (64 at most)", d0n + d1n);
      for (int bz = 0; bz < (batch > 1 ? batch : 1); ++bz)
        for (int tn = 0; tn < Nt; ++tn)
          for (int s = 0; s < S; ++s) {
            ChainItem it{};
            it.stage = si;
            it.d0_first = d0f; it.d0_count = d0n; it.d1_first = d1f; it.d1_count = d1n;
            it.sig0 = o.first + tm; it.sig1 = -1;
            if (S > 1) {
              const int tile = tn * Mt + tm;
              const int u[8] = {tm, tn, s * kt, std::min(nk, (s + 1) * kt), tile * S + s, tile * S, S, tile};
              it.bx = (int)((units.size() - unit_base[si]) / 8);
              units.insert(units.end(), u, u + 8);
            } else {
              it.bx = tm; it.by = tn; it.bz = bz;
            }
            items.push_back(it);
          }
    }
    if (S > 1) g.counters = reinterpret_cast<unsigned*>((uintptr_t)split_counters);      // index for now, pointer once allocated
    return RADNET_OK;
  };

  // Every tensor is written ONCE per launch and never after it has been read (consumers use ordinary loads: a cache line is complete
  // before any workgroup touches it; the counters order a reader behind its producer -- RAW -- and nothing else).  A list that
  // re-uses a buffer (ping-pong activations, an output written twice, an output that an earlier op read) has WAR / WAW hazards the
  // counters do not cover: refused here, the caller keeps the launch list.
  std::set<const void*> touched;
  auto claim_output = [&](const void* p, int k, const char* what) -> int {
    if (p != nullptr && touched.count(p)) RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: op %d writes %s that an earlier op of the list reads or writes (buffer re-use inside a chain)", k, what);
    touched.insert(p);
    return RADNET_OK;
  };
  for (int k = 0; k < n_ops; ++k) {
    const radnet_op& op = ops[k];
    if (op.kind == RADNET_OP_NOP) continue;
    if (op.kind == RADNET_OP_CONV_FWD) {
      ChainStage st{};
      st.type = 0;
      int rc = chain_conv_args(&ec, &op.conv, st.g);
      if (rc != RADNET_OK) return rc;
      touched.insert(op.conv.x);
      if (op.conv.addend) touched.insert(op.conv.addend);
      rc = claim_output(op.conv.y, k, "its output");
      if (rc != RADNET_OK) return rc;
      const int si = push_stage(st, ChainOut{});
      auto pm = producer.find(op.conv.x), pa = op.conv.addend ? producer.find(op.conv.addend) : producer.end();
      rc = emit_gemm(si, 1, pm != producer.end() ? pm->second : -1, pa != producer.end() ? pa->second : -1, false);
      if (rc != RADNET_OK) return rc;
      producer[op.conv.y] = si;
      flops += 2.0 * stages[si].g.M * stages[si].g.N * stages[si].g.K;
      flops_alg += 2.0 * stages[si].g.M * stages[si].g.N * stages[si].g.K;
    } else if (op.kind == RADNET_OP_WINO && op.i[8] == 4) {
      const float* x = (const float*)op.p[0];
      float* V = (float*)op.p[1];
      const float* U = (const float*)op.p[2];
      float* Mw = (float*)op.p[3];
      touched.insert(x);
      for (int q : {1, 3, 6}) {
        const int rcq = claim_output(op.p[q], k, q == 1 ? "its transformed input" : q == 3 ? "its product buffer" : "its output");
        if (rcq != RADNET_OK) return rcq;
      }
      const int nb = op.i[0], h = op.i[1], w = op.i[2], c = op.i[3], n = op.i[4], T = op.i[5], act = op.i[6], ldy = op.i[7];
      const int th = (h + 3) / 4, tw = (w + 3) / 4;
      if (T != nb * th * tw || (c & 63) || (n & 63) || !(256 % c == 0 || c % 256 == 0) || !(256 % n == 0 || n % 256 == 0))
        RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: Winograd layer c=%d n=%d tiles=%d", c, n, T);
      if ((uint64_t)36 * T * std::max(c, n) * 4ull >= (1ull << 32) || (uint64_t)nb * h * w * ldy * 4ull >= (1ull << 32))
        RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: Winograd operand larger than 4 GiB");
      auto pm = producer.find(x);
      const int dep_x = pm != producer.end() ? pm->second : -1;
      // (1) input transform: blocks of 256 units (tile, 2 channels); counters per 64 tiles
      ChainStage s1{};
      s1.type = 1;
      s1.t_src = x; s1.t_dst = V; s1.t_nb = nb; s1.t_h = h; s1.t_w = w; s1.t_c = c; s1.t_th = th; s1.t_tw = tw;
      s1.t_dst_bytes = (unsigned)((uint64_t)36 * T * c * 4ull);
      ChainOut o1;
      o1.kind = ChainOut::TILES64;
      o1.count = radnet_cdiv(T, 64);
      const int cv = c;                          // units per tile: one per channel (chain_wino4_input)
      const int n_blk1 = radnet_cdiv((long long)T * cv, 256);
      const int si1 = push_stage(s1, o1);
      outs[si1].first = new_counters(o1.count, 0u);
      for (int b = 0; b < n_blk1; ++b) {
        const int t0 = (int)(((long long)b * 256) / cv), t1 = (int)(std::min<long long>((long long)b * 256 + 255, (long long)T * cv - 1) / cv);
        ChainItem it{};
        it.stage = si1; it.bx = b;
        if (dep_x >= 0) {
          int r0 = INT32_MAX, r1 = -1;
          for (int t = t0; t <= t1; ++t) {
            const int img = t / (th * tw), ti = (t / tw) % th, tj = t % tw;
            const int ih0 = std::max(4 * ti - 1, 0), ih1 = std::min(4 * ti + 4, h - 1), iw0 = std::max(4 * tj - 1, 0), iw1 = std::min(4 * tj + 4, w - 1);
            r0 = std::min(r0, (img * h + ih0) * w + iw0);
            r1 = std::max(r1, (img * h + ih1) * w + iw1);
          }
          rows_dep(dep_x, r0, r1, it.d0_first, it.d0_count);
          if (it.d0_count > 64) RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: a transform block would wait for %d blocks", it.d0_count);
        }
        it.sig0 = outs[si1].first + t0 / 64;
        it.sig1 = t1 / 64 != t0 / 64 ? outs[si1].first + t1 / 64 : -1;
        if (t1 / 64 > t0 / 64 + 1) RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: a transform block spans three tile blocks");
        need[(size_t)it.sig0] += 1u;
        if (it.sig1 >= 0) need[(size_t)it.sig1] += 1u;
        items.push_back(it);
      }
      // (2) 36 GEMMs [T x c] . [c x n] as one batched stage
      ChainStage s2{};
      s2.type = 0;
      GemmArgs& g = s2.g;
      g.x = V; g.w = U; g.y = Mw;
      g.H = 1; g.W = T; g.C = c; g.OH = 1; g.OW = T;
      g.KW = 1; g.npos = 1; g.stride = 1;
      g.M = T; g.N = n; g.K = c;
      g.ldw = n; g.ldy = n;
      g.OHOW = T;
      g.batch = 36;
      g.x_bstride = (long long)T * c; g.w_bstride = (long long)c * n; g.y_bstride = (long long)T * n;
      g.magic_ohow = radnet_div_magic((uint32_t)g.OHOW);
      g.magic_ow = radnet_div_magic((uint32_t)g.OW);
      g.x_bytes = (unsigned)((uint64_t)T * c * 4ull);
      g.w_bytes = (unsigned)((uint64_t)c * n * 4ull);
      g.y_bytes = (unsigned)(((uint64_t)(T - 1) * n + n) * 4ull);
      const int si2 = push_stage(s2, ChainOut{});
      int rc = emit_gemm(si2, 36, si1, -1, true);
      if (rc != RADNET_OK) return rc;
      // (3) output transform: counters per (image, tile row)
      ChainStage s3{};
      s3.type = 2;
      s3.t_src = Mw; s3.t_dst = (float*)op.p[6]; s3.t_scale = (const float*)op.p[4]; s3.t_shift = (const float*)op.p[5];
      s3.t_nb = nb; s3.t_h = h; s3.t_w = w; s3.t_c = n; s3.t_th = th; s3.t_tw = tw; s3.t_act = act; s3.t_ldy = ldy;
      s3.t_dst_bytes = (unsigned)((uint64_t)nb * h * w * ldy * 4ull);
      ChainOut o3;
      o3.kind = ChainOut::TILEROWS;
      o3.count = nb * th;
      o3.h = h; o3.w = w; o3.th = th; o3.tw = tw;
      const int si3 = push_stage(s3, o3);
      outs[si3].first = new_counters(o3.count, 0u);
      const int nv = n;
      const int n_blk3 = radnet_cdiv((long long)T * nv, 256);
      for (int b = 0; b < n_blk3; ++b) {
        const int t0 = (int)(((long long)b * 256) / nv), t1 = (int)(std::min<long long>((long long)b * 256 + 255, (long long)T * nv - 1) / nv);
        ChainItem it{};
        it.stage = si3; it.bx = b;
        it.d0_first = outs[si2].first + t0 / 64;
        it.d0_count = t1 / 64 - t0 / 64 + 1;
        const int row0 = t0 / tw, row1 = t1 / tw;         // (image * th + tile row)
        if (row1 > row0 + 1) RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: a transform block spans three tile rows");
        it.sig0 = outs[si3].first + row0;
        it.sig1 = row1 != row0 ? outs[si3].first + row1 : -1;
        need[(size_t)it.sig0] += 1u;
        if (it.sig1 >= 0) need[(size_t)it.sig1] += 1u;
        items.push_back(it);
      }
      producer[op.p[6]] = si3;
      flops += 2.0 * 36.0 * T * (double)n * c;
      flops_alg += 2.0 * nb * h * w * (double)n * 9.0 * c;
    } else {
      RADNET_FAIL(&ec, RADNET_ERR_UNSUPPORTED, "chain: op kind %d at position %d cannot run as chain items", op.kind, k);
    }
  }
  if (pl.items.empty()) RADNET_FAIL(&ec, RADNET_ERR_ARG, "chain: empty program");
  return RADNET_OK;
}

// In list order, with every earlier item finished, each item must find its input blocks complete: then a single workgroup
// can run the list, and any number of workgroups drawing from it in order cannot deadlock.  Also: every counter reaches
// exactly its `need`.  Returns the first offending item (or -1).
static int chain_first_unrunnable(const ChainPlan& pl, int* bad_counter) {
  std::vector<unsigned> c(pl.need.size(), 0u);
  for (size_t i = 0; i < pl.items.size(); ++i) {
    const ChainItem& it = pl.items[i];
    for (int k = 0; k < it.d0_count; ++k)
      if (c[(size_t)it.d0_first + k] < pl.need[(size_t)it.d0_first + k]) { *bad_counter = it.d0_first + k; return (int)i; }
    for (int k = 0; k < it.d1_count; ++k)
      if (c[(size_t)it.d1_first + k] < pl.need[(size_t)it.d1_first + k]) { *bad_counter = it.d1_first + k; return (int)i; }
    if (it.sig0 >= 0) c[(size_t)it.sig0] += 1u;
    if (it.sig1 >= 0) c[(size_t)it.sig1] += 1u;
  }
  for (size_t k = 0; k < c.size(); ++k)
    if (pl.need[k] != 0u && c[k] != pl.need[k]) { *bad_counter = (int)k; return (int)pl.items.size(); }
  return -1;
}

extern "C" int radnet_chain_check(const radnet_op* ops, int32_t n_ops, int32_t* n_items, int32_t* n_stages, int32_t* n_counters, int32_t* first_bad_item,
                                  int32_t* bad_counter, char* err, int32_t err_len) {
  if (!ops || n_ops <= 0) return RADNET_ERR_ARG;
  ChainPlan pl;
  ErrSink ec{};
  const int rc = chain_plan(ops, n_ops, pl, ec);
  if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", ec.err);
  if (rc != RADNET_OK) return rc;
  int bc = -1;
  const int bad = chain_first_unrunnable(pl, &bc);
  if (n_items) *n_items = (int32_t)pl.items.size();
  if (n_stages) *n_stages = (int32_t)pl.stages.size();
  if (n_counters) *n_counters = (int32_t)pl.need.size();
  if (first_bad_item) *first_bad_item = bad;
  if (bad_counter) *bad_counter = bc;
  return RADNET_OK;
}

extern "C" int radnet_chain_build(radnet_ctx* ctx, const radnet_op* ops, int32_t n_ops, int32_t workgroups, radnet_chain** out) {
  if (!ctx || !ops || n_ops <= 0 || !out) return RADNET_ERR_ARG;
  *out = nullptr;
  // every workgroup of the grid must be resident (static deal, see chain_body): 4 per CU is what LDS and registers allow
  const int grid = std::min(workgroups > 0 ? workgroups : 2 * kNumCU, 4 * kNumCU);
  ChainPlan pl;
  {
    ErrSink ec{};
    const int rc = chain_plan(ops, n_ops, pl, ec);
    if (rc != RADNET_OK) RADNET_FAIL(ctx, rc, "%s", ec.err);
    int bc = -1;
    const int bad = chain_first_unrunnable(pl, &bc);
    if (bad >= 0) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "chain: item %d of %d cannot run in list order (counter %d)", bad, (int)pl.items.size(), bc);
  }
  std::vector<ChainStage>& stages = pl.stages;
  std::vector<ChainItem>& items = pl.items;
  std::vector<unsigned>& need = pl.need;
  std::vector<int>& units = pl.units;
  std::vector<size_t>& unit_base = pl.unit_base;
  std::vector<size_t>& slab_base = pl.slab_base;
  const size_t slabs_total = pl.slabs_total;
  const double flops = pl.flops, flops_alg = pl.flops_alg;

  radnet_chain* ch = new radnet_chain();
  ch->grid = grid;
  ch->n_items = (unsigned)items.size();
  ch->n_counters = (unsigned)need.size();
  ch->n_stages = (unsigned)stages.size();
  ch->flops = flops;
  ch->flops_algorithmic = flops_alg;
  auto fail = [&](const char* what) {
    radnet_chain_destroy(ch);
    snprintf(ctx->err, sizeof(ctx->err), "chain: %s", what);
    return RADNET_ERR_HIP;
  };
  if (hipMalloc((void**)&ch->d_hdr, sizeof(ChainHeader)) != hipSuccess || hipMemset(ch->d_hdr, 0, sizeof(ChainHeader)) != hipSuccess) return fail("header");
  {
    if (hipHostMalloc((void**)&ch->h_err, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return fail("error word");
    *ch->h_err = 0u;
    ChainHeader h0{};
    h0.host_lo = (unsigned)((unsigned long long)(uintptr_t)ch->h_err & 0xffffffffull);
    h0.host_hi = (unsigned)((unsigned long long)(uintptr_t)ch->h_err >> 32);
    if (hipMemcpy(ch->d_hdr, &h0, sizeof(h0), hipMemcpyHostToDevice) != hipSuccess) return fail("header");
  }
  if (hipMalloc((void**)&ch->d_counters, need.size() * 4 * kCtrStride) != hipSuccess || hipMemset(ch->d_counters, 0, need.size() * 4 * kCtrStride) != hipSuccess) return fail("counters");
  if (hipMalloc((void**)&ch->d_need, need.size() * 4) != hipSuccess || hipMemcpy(ch->d_need, need.data(), need.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("need");
  if (!units.empty() && (hipMalloc((void**)&ch->d_units, units.size() * 4) != hipSuccess || hipMemcpy(ch->d_units, units.data(), units.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) return fail("units");
  if (slabs_total && hipMalloc((void**)&ch->d_slabs, slabs_total * 4) != hipSuccess) return fail("slabs");
  for (size_t s = 0; s < stages.size(); ++s) {
    if (unit_base[s] == ~(size_t)0) continue;
    GemmArgs& g = stages[s].g;
    g.units = ch->d_units + unit_base[s];
    g.partial = ch->d_slabs + slab_base[s];
    g.counters = ch->d_counters + (uintptr_t)g.counters * kCtrStride;      // dense tile counters inside this stage's strided span
  }
  if (hipMalloc((void**)&ch->d_stages, stages.size() * sizeof(ChainStage)) != hipSuccess ||
      hipMemcpy(ch->d_stages, stages.data(), stages.size() * sizeof(ChainStage), hipMemcpyHostToDevice) != hipSuccess) return fail("stages");
  if (hipMalloc((void**)&ch->d_items, items.size() * sizeof(ChainItem)) != hipSuccess ||
      hipMemcpy(ch->d_items, items.data(), items.size() * sizeof(ChainItem), hipMemcpyHostToDevice) != hipSuccess) return fail("items");
  ch->h_items = items;
  ch->h_need = need;
  if (getenv("RADNET_CHAIN_DEBUG") && (hipMalloc((void**)&ch->d_marks, (size_t)grid * 16) != hipSuccess || hipMemset(ch->d_marks, 0, (size_t)grid * 16) != hipSuccess)) return fail("marks");
  *out = ch;
  return RADNET_OK;
}

// Diagnosis WHILE a chain launch is (or seems to be) running: copies the header {next item, workgroups gone, error, first
// error, runs} and, for item `item` (>= 0), its record and the current values / expected values of the counters it waits
// for, through a stream of its own (does not wait for the launch).  out: 8 header words, 12 item words, then up to 64
// (have, need) pairs; returns the number of pairs.
extern "C" int radnet_chain_peek(radnet_chain* ch, int32_t item, uint32_t* out, int32_t out_words) {
  if (!ch || !out || out_words < 20 + 128) return RADNET_ERR_ARG;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return RADNET_ERR_HIP;
  int pairs = 0;
  bool ok = hipMemcpyAsync(out, ch->d_hdr, 8 * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (ok && item >= 0 && (unsigned)item < ch->n_items) {
    const ChainItem& it = ch->h_items[(size_t)item];
    memcpy(out + 8, &it, 12 * 4);
    for (int r = 0; r < 2 && ok; ++r) {
      const int f = r == 0 ? it.d0_first : it.d1_first, n = r == 0 ? it.d0_count : it.d1_count;
      for (int k = 0; k < n && pairs < 64 && ok; ++k, ++pairs) {
        ok = hipMemcpyAsync(out + 20 + 2 * pairs, ch->d_counters + (size_t)(f + k) * kCtrStride, 4, hipMemcpyDeviceToHost, st) == hipSuccess;
        out[20 + 2 * pairs + 1] = ch->h_need[(size_t)f + k];
      }
    }
  }
  ok = ok && hipStreamSynchronize(st) == hipSuccess;
  if (ok && ch->d_marks && item == -2) {        // debug build of the chain: histogram of the waves' phases into out[20..27], a stuck wave's mark in out[28]
    std::vector<unsigned> m((size_t)ch->grid * 4);
    ok = hipMemcpyAsync(m.data(), ch->d_marks, m.size() * 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    for (int k = 0; k < 9; ++k) out[20 + k] = 0;
    for (unsigned v : m) {
      out[20 + std::min(v & 255u, 7u)] += 1;
      if ((v & 255u) >= 1 && (v & 255u) <= 4) out[28] = v;
    }
  }
  (void)hipStreamDestroy(st);
  return ok ? pairs : RADNET_ERR_HIP;
}

extern "C" uint32_t radnet_chain_error(radnet_chain* ch) { return (ch && ch->h_err) ? *(volatile unsigned*)ch->h_err : 0u; }

extern "C" int radnet_chain_run(radnet_ctx* ctx, radnet_chain* ch) {
  if (!ctx || !ch) return RADNET_ERR_ARG;
  if (const uint32_t e = radnet_chain_error(ch))
    RADNET_FAIL(ctx, RADNET_ERR_HIP, "chain: an earlier launch of this chain gave up waiting at item %u (not every workgroup of its grid was resident?): "
                "its outputs were invalid", e - 1u);
  radnet_timing_arm(ctx);
  static const int variant = getenv("RADNET_CHAIN_VARIANT") ? atoi(getenv("RADNET_CHAIN_VARIANT")) : 0;      // diagnosis
  if (variant == 1)
    RADNET_LAUNCH(chain_kernel_noattr, dim3(ch->grid), dim3(NTHREADS), 0, ctx->stream, ctx->arm0, ctx->arm1, ch->d_hdr, ch->d_stages, ch->d_items,
                  ch->d_counters, ch->d_need, ch->n_items, ch->n_counters, ch->d_marks);
  else if (variant == 2)
    RADNET_LAUNCH(chain_kernel_nocoh, dim3(ch->grid), dim3(NTHREADS), 0, ctx->stream, ctx->arm0, ctx->arm1, ch->d_hdr, ch->d_stages, ch->d_items,
                  ch->d_counters, ch->d_need, ch->n_items, ch->n_counters, ch->d_marks);
  else
    RADNET_LAUNCH(chain_kernel, dim3(ch->grid), dim3(NTHREADS), 0, ctx->stream, ctx->arm0, ctx->arm1, ch->d_hdr, ch->d_stages, ch->d_items, ch->d_counters,
                  ch->d_need, ch->n_items, ch->n_counters, ch->d_marks);
  RADNET_CHECK_LAUNCH(ctx, "chain");
  radnet_timing_end_armed(ctx, 0, ch->flops_algorithmic);
  return RADNET_OK;
}

// Synchronises the context's stream.  last_error: 0, or 1 + the index of the first item that gave up waiting (sticky).
extern "C" int radnet_chain_status(radnet_ctx* ctx, radnet_chain* ch, int32_t* last_error, int32_t* runs, int32_t* n_items, int32_t* n_stages,
                                   double* flops_executed, double* flops_algorithmic) {
  if (!ctx || !ch) return RADNET_ERR_ARG;
  RADNET_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ChainHeader h;
  RADNET_CHECK_HIP(ctx, hipMemcpy(&h, ch->d_hdr, sizeof(h), hipMemcpyDeviceToHost));
  if (last_error) *last_error = (int32_t)h.last_error;
  if (runs) *runs = (int32_t)h.runs;
  if (n_items) *n_items = (int32_t)ch->n_items;
  if (n_stages) *n_stages = (int32_t)ch->n_stages;
  if (flops_executed) *flops_executed = ch->flops;
  if (flops_algorithmic) *flops_algorithmic = ch->flops_algorithmic;
  return RADNET_OK;
}
