// Window value of up to 256 buckets in one block, with every point addition spread over
// four lanes.
//
// The kernels that turn bucket sums into a window value are chains of dependent point
// additions on a few lanes while the rest of the GPU idles, so what counts is the time of
// ONE addition.  An XYZZ addition is 14 field multiplications, but only four of them are
// sequential: {U1, U2, S1, S2}, then {P^2, R^2, ZZ1*ZZ2, ZZZ1*ZZZ2}, then {P^3, U1*P^2,
// ZZ1*ZZ2*P^2}, then {R*(Q - X3), S1*P^3, ZZZ1*ZZZ2*P^3}.  Four adjacent lanes (a quad) take
// one multiplication of each stage and exchange the products through a private piece of LDS
// (all four sit in one wave, so a wait on the LDS counter orders the exchange; no block
// barrier).  Which operand a lane reads is a matter of its address, not of a branch, so the
// four lanes run one instruction stream.  Doubling is three stages the same way.
//
// The algorithm around it is the one of k_window_reduce_small (bp_msm.hip): a binary sum
// tree built in place, one tree per weight bit, then sum_b 2^b S_b.
// Stands in for the window sums inside ark-ec 0.5.0 VariableBaseMSM::msm_unchecked
// (reference call sites: src/msm.rs:84-87, src/ipa/mod.rs:54,70).
#include "bp_curve.h"
#include "bp_internal.h"
#include "bp_quad.h"

namespace bp {

#define DISPATCH_CURVE_R(curve, STMT) \
  switch (curve) {                    \
    case BP_SECP256K1: {              \
      using C = Secp256k1P;           \
      STMT;                           \
    } break;                          \
    case BP_VESTA: {                  \
      using C = Vesta;                \
      STMT;                           \
    } break;                          \
    default:                          \
      return hipErrorInvalidValue;    \
  }

constexpr u32 kQuadMaxBuckets = 256;
// 96 quads: the 96 additions of the widest step at 128 buckets in one pass, the 192 at 256 buckets
// in two (points, exchange space and terms together stay below 64 KB of LDS)
constexpr u32 kQuadThreads = 384;
constexpr int kQuadPasses = 2;
static_assert(kQuadPasses * (kQuadThreads / 4) * 4 >= 3 * kQuadMaxBuckets, "passes cover the widest step");

// Addition number `task` of step t over nb slots: first the nb >> t additions of the sum tree
// (level t), then for every weight bit b < t one step of the S_b tree (nb >> (t+1) additions each).
BP_DEV bool quad_task(u32 task, int t, u32 nb, u32& dst, u32& src) {
  const u32 chain = nb >> t, per = nb >> (t + 1);
  dst = src = 0;
  if (task < chain) {
    dst = task << t;
    src = dst + (1u << (t - 1));
    return true;
  }
  const u32 q = task - chain;
  const u32 b = q / per, j = q % per;
  if (b >= (u32)t) return false;
  const int r = t - (int)b;  // step number inside the S_b tree
  const u32 k = j << r;
  dst = (2 * k + 1) << b;
  src = (2 * (k + (1u << (r - 1))) + 1) << b;
  return true;
}

// V = sum_{m=1}^{nb} m * B_{m-1} over buckets base .. base + nb - 1 of an SoA array with plane
// stride nbt, nb a power of two in 2..256, doubled post_dbl times, to out (XYZZ, 32 words).
// One block of kQuadThreads lanes; see k_window_reduce_small in bp_msm.hip for the two trees.
// EXPORT: the result leaves the pipeline (a window sum): coordinates go back to the boundary form
// JOIN: the block's result is one of the two parts of window join_w (parts[join_w] and
// parts[n_windows + join_w]); the block that finishes second (counted in done[join_w], zero before
// the launch) adds the two and writes the window sum to windows[join_w], in the boundary form.
template <class C, bool EXPORT, bool JOIN = false>
BP_DEV void window_reduce_block_quad(const u32* __restrict__ buckets, size_t nbt, size_t base, u32 nb, int log_nb,
                                     int post_dbl, u32* __restrict__ out, u32* __restrict__ done = nullptr,
                                     const u32* __restrict__ parts = nullptr, u32 n_windows = 0, u32 join_w = 0,
                                     u32* __restrict__ windows = nullptr) {
  using F = typename C::Fq;
  __shared__ u32 slot[kQuadMaxBuckets * 32];
  __shared__ u32 term[16 * 32];
  __shared__ u32 scr_all[(kQuadThreads / 4) * kQuadScratch * 8];
  __shared__ u32 zero[8];
  const u32 tid = threadIdx.x, quad = tid >> 2;
  const int role = (int)(tid & 3u), coord = quad_coord(role);
  u32* scr = scr_all + quad * (kQuadScratch * 8);
  if (tid < 8) zero[tid] = 0;
  // slot m = B_{m-1}, slot 0 = identity; one coordinate per lane
  for (u32 i = tid; i < nb * 4; i += blockDim.x) {
    const u32 s = i >> 2, c = i & 3;
    Fe v = s == 0 ? quad_identity_coord((int)c, fe_one<F>()) : fe_load(buckets + ((size_t)c * nbt + base + s - 1) * 8);
    fe_store(slot + s * 32 + c * 8, v);
  }
  __syncthreads();
  const u32 n_quads = blockDim.x >> 2;
  static_assert(kQuadPasses == 2, "the step below is written out for two passes");
  for (int t = 1; t <= log_nb - 1; t++) {
    Fe v0, v1;
    u32 dst0, src0, dst1, src1;
    const bool work0 = quad_task(quad, t, nb, dst0, src0);
    const bool work1 = quad_task(quad + n_quads, t, nb, dst1, src1);
    const bool st0 = work0 ? quad_add<F>(slot, scr, zero, dst0, src0, role, v0) : false;
    const bool st1 = work1 ? quad_add<F>(slot, scr, zero, dst1, src1, role, v1) : false;
    __syncthreads();
    if (st0) fe_store(slot + dst0 * 32 + coord * 8, v0);
    if (st1) fe_store(slot + dst1 * 32 + coord * 8, v1);
    __syncthreads();
  }
  // terms: T_b = S_b = slot[2^b] for b < log_nb, T_{log_nb} = B_{nb-1}; term b is doubled
  // b + post_dbl times by its own quad, then a tree of additions joins the terms
  const int n_terms = log_nb + 1;
  if ((int)quad < n_terms) {
    Fe v = (int)quad < log_nb ? fe_load(slot + ((size_t)1 << quad) * 32 + coord * 8)
                              : fe_load(buckets + ((size_t)coord * nbt + base + nb - 1) * 8);
    fe_store(term + quad * 32 + coord * 8, v);
    quad_sync();
    for (int d = 0; d < (int)quad + post_dbl; d++) quad_dbl<F>(term + quad * 32, scr, zero, role);
  }
  __syncthreads();
  for (int sft = 1; sft < n_terms; sft <<= 1) {
    const bool work = (int)quad < n_terms && (quad % (2 * sft)) == 0 && (int)quad + sft < n_terms;
    Fe v;
    bool st = false;
    if (work) st = quad_add<F>(term, scr, zero, quad, quad + (u32)sft, role, v);
    __syncthreads();
    if (st) fe_store(term + quad * 32 + coord * 8, v);
    __syncthreads();
  }
  if (quad == 0) {
    Fe v = fe_load(term + coord * 8);
    if constexpr (EXPORT) v = fe_export<C>(v);
    fe_store(out + coord * 8, v);
  }
  if constexpr (JOIN) {
    __shared__ u32 arrival;
    __threadfence();  // the part is visible before the count says so
    __syncthreads();
    if (tid == 0) arrival = atomicAdd(&done[join_w], 1u);
    __syncthreads();
    if (arrival == 1 && quad == 0) {
      __threadfence();  // acquire: the other block's part (another XCD's L2 may hold it) after its count
      // both parts are in memory; read past this CU's cache (the other part was written by another block)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const u32* src = parts + ((size_t)h * n_windows + join_w) * 32 + coord * 8;
        Fe v;
#pragma unroll
        for (int k = 0; k < 8; k++) v.v[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fe_store(term + h * 32 + coord * 8, v);
      }
      quad_sync();
      Fe v;
      if (!quad_add<F>(term, scr, zero, 0, 1, role, v)) v = fe_load(term + coord * 8);
      fe_store(windows + (size_t)join_w * 32 + coord * 8, fe_export<C>(v));
    }
  }
}

template <class C>
__global__ __launch_bounds__(kQuadThreads) void k_window_reduce_quad(const u32* __restrict__ buckets, size_t nbt,
                                                                     u32 n_buckets, int log_nb,
                                                                     u32* __restrict__ windows) {
  window_reduce_block_quad<C, true>(buckets, nbt, (size_t)blockIdx.x * n_buckets, n_buckets, log_nb, 0,
                                    windows + (size_t)blockIdx.x * 32);
}

// Blocks 0 .. n_windows-1: parts[w] = sum_q (q+1)*Csum_q over the 256 column sums of window w;
// blocks n_windows .. 2 n_windows-1: parts[n_windows + w] = 256 * sum_r r*Rsum_r over its row
// sums (see k_rowcol_sums in bp_msm.hip).  Window value = sum of the two parts.
template <class C>
__global__ __launch_bounds__(kQuadThreads) void k_rowcol_final_quad(const u32* __restrict__ cols,
                                                                    const u32* __restrict__ rows, int n_windows,
                                                                    u32 n_rows, int log_rows,
                                                                    u32* __restrict__ parts, u32* __restrict__ done,
                                                                    u32* __restrict__ windows) {
  const bool is_rows = blockIdx.x >= (u32)n_windows;
  const u32 w = is_rows ? blockIdx.x - (u32)n_windows : blockIdx.x;
  const u32 nb = is_rows ? n_rows : 256u;
  window_reduce_block_quad<C, false, true>(is_rows ? rows : cols, (size_t)n_windows * nb, (size_t)w * nb, nb,
                                           is_rows ? log_rows : 8, is_rows ? 8 : 0, parts + (size_t)blockIdx.x * 32, done,
                                           parts, (u32)n_windows, w, windows);
}

hipError_t launch_window_reduce_quad(int curve, const u32* buckets, size_t nbt, u32 n_buckets, int n_windows,
                                     u32* windows, hipStream_t st) {
  if (n_buckets < 2 || n_buckets > kQuadMaxBuckets || (n_buckets & (n_buckets - 1))) return hipErrorInvalidValue;
  int log_nb = 0;
  while ((1u << log_nb) < n_buckets) log_nb++;
  DISPATCH_CURVE_R(curve, hipLaunchKernelGGL(k_window_reduce_quad<C>, dim3((unsigned)n_windows), dim3(kQuadThreads), 0,
                                             st, buckets, nbt, n_buckets, log_nb, windows));
  return hipGetLastError();
}

// done: n_windows counters that are zero (k_rowcol_sums clears them); the window sums go
// to `windows` in the boundary form
hipError_t launch_rowcol_final_quad(int curve, const u32* cols, const u32* rows, int n_windows, u32 n_rows,
                                    u32* parts, u32* done, u32* windows, hipStream_t st) {
  if (n_rows < 2 || n_rows > kQuadMaxBuckets || (n_rows & (n_rows - 1))) return hipErrorInvalidValue;
  int log_rows = 0;
  while ((1u << log_rows) < n_rows) log_rows++;
  DISPATCH_CURVE_R(curve, hipLaunchKernelGGL(k_rowcol_final_quad<C>, dim3(2 * (unsigned)n_windows), dim3(kQuadThreads),
                                             0, st, cols, rows, n_windows, n_rows, log_rows, parts, done, windows));
  return hipGetLastError();
}

}  // namespace bp
