// Curve-vector kernels for gfx950: generator fold, per-point scaling, batch
// normalisation.  One thread owns one output point; the double-and-add schedule
// of the generator fold is shared by all lanes (the two scalars are the same for
// every output), so there is no divergence on the common path.
#include <cstdlib>
#include "bp_curve.h"
#include "bp_internal.h"
#include "bp_quad.h"

namespace bp {

#define DISPATCH_CURVE(curve, STMT)              \
  switch (curve) {                               \
    case BP_SECP256K1: {                         \
      using C = Secp256k1;                       \
      STMT;                                      \
    } break;                                     \
    case BP_VESTA: {                             \
      using C = Vesta;                           \
      STMT;                                      \
    } break;                                     \
    default:                                     \
      return hipErrorInvalidValue;               \
  }

// out[i] = x*L[i] + y*R[i]  (src/ipa/mod.rs:215-227).  x, y arrive as canonical
// 256-bit integers; Shamir's trick over their joint bits with the per-thread
// table {L, R, L+R} (L+R normalised to affine so every add is a mixed add).
// Up to two independent folds (the G and H folds of one IPA round) share a launch:
// the kernel is bound by the latency of its ~450-step dependent chain, not by
// throughput, until h reaches several 10^5.
template <class C>
__global__ __launch_bounds__(256) void k_fold_generators(FoldJob j0, FoldJob j1, size_t h) {
  using F = typename C::Fq;
  // a block belongs to one job, so the bit schedule stays uniform within a wave
  const unsigned blocks_per_job = (unsigned)((h + blockDim.x - 1) / blockDim.x);
  const bool second = blockIdx.x >= blocks_per_job;
  const FoldJob& jb = second ? j1 : j0;
  size_t i = (size_t)(blockIdx.x - (second ? blocks_per_job : 0)) * blockDim.x + threadIdx.x;
  if (i >= h || jb.lx == nullptr) return;
  bool l_inf = jb.linf && jb.linf[i];
  bool r_inf = jb.rinf && jb.rinf[i];
  Fe Lx = fe_load(jb.lx + i * 8), Ly = fe_load(jb.ly + i * 8);
  Fe Rx = fe_load(jb.rx + i * 8), Ry = fe_load(jb.ry + i * 8);
  // T = L + R in affine (or flagged identity)
  Fe Tx, Ty;
  bool t_inf;
  {
    Jac tj = jac_identity<F>();
    if (!l_inf) jac_madd<F>(tj, Lx, Ly);
    if (!r_inf) jac_madd<F>(tj, Rx, Ry);
    t_inf = jac_to_affine<F>(tj, Tx, Ty);
  }
  Jac acc = jac_identity<F>();
  for (int w = 7; w >= 0; w--) {
    u32 xw = second ? j1.x.v[w] : j0.x.v[w], yw = second ? j1.y.v[w] : j0.y.v[w];
    for (int b = 31; b >= 0; b--) {
      acc = jac_dbl<F>(acc);
      u32 sel = ((xw >> b) & 1) | (((yw >> b) & 1) << 1);
      if (sel) {  // uniform within a job
        Fe qx = fe_select(sel == 1, Lx, fe_select(sel == 2, Rx, Tx));
        Fe qy = fe_select(sel == 1, Ly, fe_select(sel == 2, Ry, Ty));
        bool q_inf = sel == 1 ? l_inf : (sel == 2 ? r_inf : t_inf);
        if (!q_inf) jac_madd<F>(acc, qx, qy);
      }
    }
  }
  Fe X, Y;
  bool inf = jac_to_affine<F>(acc, X, Y);
  fe_store(jb.ox + i * 8, X);
  fe_store(jb.oy + i * 8, Y);
  if (jb.oinf) jb.oinf[i] = inf ? 1 : 0;
}

hipError_t launch_fold_generators2(int curve, const FoldJob& j0, const FoldJob& j1, size_t h, hipStream_t st) {
  if (h == 0) return hipSuccess;
  unsigned blocks = (unsigned)((h + 255) / 256) * (j1.lx ? 2u : 1u);
  DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_fold_generators<C>, dim3(blocks), dim3(256), 0, st, j0, j1, h));
  return hipGetLastError();
}

hipError_t launch_fold_generators(int curve, const u32* lx, const u32* ly, const uint8_t* linf,
                                  const u32* rx, const u32* ry, const uint8_t* rinf, size_t h,
                                  FeArg x_canon, FeArg y_canon, u32* ox, u32* oy, uint8_t* oinf,
                                  hipStream_t st) {
  FoldJob j{lx, ly, linf, rx, ry, rinf, ox, oy, oinf, x_canon, y_canon};
  return launch_fold_generators2(curve, j, FoldJob{}, h, st);
}

// out[i] = s[i] * P[i]  (src/range/aggregate.rs:148-153).  Per-point scalars, so
// the add branch diverges; the doublings stay uniform.
template <class C>
__global__ __launch_bounds__(256) void k_scale_points(const u32* __restrict__ px,
                                                      const u32* __restrict__ py,
                                                      const uint8_t* __restrict__ pinf,
                                                      const u32* __restrict__ s, size_t n,
                                                      u32* __restrict__ ox, u32* __restrict__ oy,
                                                      uint8_t* __restrict__ oinf) {
  using F = typename C::Fq;
  using Fr = typename C::Fr;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool p_inf = pinf && pinf[i];
  Fe Px = fe_load(px + i * 8), Py = fe_load(py + i * 8);
  Fe k = fe_from_mont<Fr>(fe_load(s + i * 8));
  Jac acc = jac_identity<F>();
  if (!p_inf) {
    for (int w = 7; w >= 0; w--) {
      u32 kw = k.v[w];
      for (int b = 31; b >= 0; b--) {
        acc = jac_dbl<F>(acc);
        if ((kw >> b) & 1) jac_madd<F>(acc, Px, Py);
      }
    }
  }
  Fe X, Y;
  bool inf = jac_to_affine<F>(acc, X, Y);
  fe_store(ox + i * 8, X);
  fe_store(oy + i * 8, Y);
  if (oinf) oinf[i] = inf ? 1 : 0;
}

hipError_t launch_scale_points(int curve, const u32* px, const u32* py, const uint8_t* pinf,
                               const u32* s, size_t n, u32* ox, u32* oy, uint8_t* oinf,
                               hipStream_t st) {
  if (n == 0) return hipSuccess;
  unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_scale_points<C>, dim3(blocks), dim3(256), 0, st, px, py,
                                           pinf, s, n, ox, oy, oinf));
  return hipGetLastError();
}

// Window tables of a fixed base set: record (w, i) = 2^(c*w) * P_i as affine x|y (16 u32) at
// index w*n + i.  With them an MSM over these bases needs one bucket set for all windows and
// no doublings afterwards.  One lane per base; built once per base set.
template <class C>
__global__ __launch_bounds__(256) void k_window_tables(const u32* __restrict__ px, const u32* __restrict__ py,
                                                       const uint8_t* __restrict__ pinf, size_t n, int c,
                                                       int n_windows, u32* __restrict__ table) {
  using F = typename C::Fq;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool inf = pinf && pinf[i];
  Jac acc;
  acc.x = fe_load(px + i * 8);
  acc.y = fe_load(py + i * 8);
  acc.z = fe_one<F>();
  for (int w = 0; w < n_windows; w++) {
    if (w)
      for (int k = 0; k < c; k++) acc = jac_dbl<F>(acc);
    Fe x = fe_zero(), y = fe_zero();
    if (!inf) jac_to_affine<F>(acc, x, y);  // never the identity: the group order is an odd prime
    // the records are read by the MSM pipeline as packed points: in its form (see CurveT)
    u32* rec = table + ((size_t)w * n + i) * 16;
    fe_store(rec, fe_import<typename C::Pipeline>(x));
    fe_store(rec + 8, fe_import<typename C::Pipeline>(y));
  }
}

// The same tables with ONE field inversion per base instead of one per record: the lane keeps the chain
// 2^(c w) P in Jacobian coordinates in `scratch` (x, y, z and the product of the z before it, 128 bytes per
// record), inverts the product of all its z once and unwinds.  `scratch`: n_windows * n * 128 bytes.
template <class C>
__global__ __launch_bounds__(256) void k_window_tables_inv1(const u32* __restrict__ px, const u32* __restrict__ py,
                                                            const uint8_t* __restrict__ pinf, size_t n, int c, int n_windows,
                                                            u32* __restrict__ scratch, u32* __restrict__ table) {
  using F = typename C::Fq;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (pinf && pinf[i]) {
    for (int w = 0; w < n_windows; w++) {
      u32* rec = table + ((size_t)w * n + i) * 16;
      fe_store(rec, fe_zero());
      fe_store(rec + 8, fe_zero());
    }
    return;
  }
  Jac acc;
  acc.x = fe_load(px + i * 8);
  acc.y = fe_load(py + i * 8);
  acc.z = fe_one<F>();
  Fe run = fe_one<F>();
  auto plane = [&](int w, int p) { return scratch + (((size_t)w * 4 + (size_t)p) * n + i) * 8; };
#pragma unroll 1
  for (int w = 0; w < n_windows; w++) {
    if (w) {
#pragma unroll 1
      for (int k = 0; k < c; k++) acc = jac_dbl<F>(acc);
    }
    fe_store(plane(w, 0), acc.x);
    fe_store(plane(w, 1), acc.y);
    fe_store(plane(w, 2), acc.z);
    fe_store(plane(w, 3), run);
    run = fe_mul<F>(run, acc.z);  // never zero: the group order is an odd prime, no multiple of P is the identity
  }
  Fe inv = fe_inv<F>(run);
  // the record below is loaded while this one is normalised: the loads do not depend on the running inverse
  Fe nx = fe_load(plane(n_windows - 1, 0)), ny = fe_load(plane(n_windows - 1, 1)), nz = fe_load(plane(n_windows - 1, 2)),
     np = fe_load(plane(n_windows - 1, 3));
#pragma unroll 1
  for (int w = n_windows - 1; w >= 0; w--) {
    const Fe cx = nx, cy = ny, z = nz, pre = np;
    if (w > 0) {
      nx = fe_load(plane(w - 1, 0));
      ny = fe_load(plane(w - 1, 1));
      nz = fe_load(plane(w - 1, 2));
      np = fe_load(plane(w - 1, 3));
    }
    const Fe zi = fe_mul<F>(inv, pre);
    inv = fe_mul<F>(inv, z);
    const Fe zi2 = fe_sqr<F>(zi);
    const Fe x = fe_mul<F>(cx, zi2);
    const Fe y = fe_mul<F>(fe_mul<F>(cy, zi2), zi);
    u32* rec = table + ((size_t)w * n + i) * 16;
    fe_store(rec, fe_import<typename C::Pipeline>(x));
    fe_store(rec + 8, fe_import<typename C::Pipeline>(y));
  }
}

hipError_t launch_window_tables_inv1(int curve, const u32* px, const u32* py, const uint8_t* pinf, size_t n, int c,
                                     int n_windows, u32* scratch, u32* table, hipStream_t st) {
  if (n == 0) return hipSuccess;
  unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_window_tables_inv1<C>, dim3(blocks), dim3(256), 0, st, px, py, pinf, n, c,
                                           n_windows, scratch, table));
  return hipGetLastError();
}

hipError_t launch_window_tables(int curve, const u32* px, const u32* py, const uint8_t* pinf, size_t n, int c,
                                int n_windows, u32* table, hipStream_t st) {
  if (n == 0) return hipSuccess;
  unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_window_tables<C>, dim3(blocks), dim3(256), 0, st, px, py, pinf, n, c,
                                           n_windows, table));
  return hipGetLastError();
}

// Jacobian (X,Y,Z interleaved, 24 u32 per point) -> affine planes
template <class C>
__global__ __launch_bounds__(256) void k_normalize_jac(const u32* __restrict__ xyz, size_t n,
                                                       u32* __restrict__ ox, u32* __restrict__ oy,
                                                       uint8_t* __restrict__ oinf) {
  using F = typename C::Fq;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac p;
  p.x = fe_load(xyz + i * 24);
  p.y = fe_load(xyz + i * 24 + 8);
  p.z = fe_load(xyz + i * 24 + 16);
  Fe X, Y;
  bool inf = jac_to_affine<F>(p, X, Y);
  fe_store(ox + i * 8, X);
  fe_store(oy + i * 8, Y);
  if (oinf) oinf[i] = inf ? 1 : 0;
}

hipError_t launch_normalize_jac(int curve, const u32* xyz, size_t n, u32* ox, u32* oy, uint8_t* oinf,
                                hipStream_t st) {
  if (n == 0) return hipSuccess;
  unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_normalize_jac<C>, dim3(blocks), dim3(256), 0, st, xyz, n,
                                           ox, oy, oinf));
  return hipGetLastError();
}

// ---- folded generators in one pass ("rebase") ------------------------------------------------
// After R rounds an inner-product argument over n0 = m * n generators (m = 2^R) works, in this
// library, with the original generators and one coefficient per generator; the folded generators
// themselves are G'_i = sum_{t<m} s_t * G0[i + n t], i < n, with the SAME m coefficients s_t for
// every i (products of the challenges chosen by the bits of t: src/ipa/mod.rs:80-83 applied R
// times).  Since the scalars are shared, so are their digits: lane (i, w) walks t = 0 .. m-1, adds
// G0[i + n t] into the bucket that digit w of s_t names -- the same bucket for all lanes of the
// block, so the bucket choice is a uniform branch over sums kept in registers -- and leaves the
// window sum.  3-bit signed digits (4 buckets, 86 windows).  A second launch combines the windows
// of each output (258 doublings) and normalises.
constexpr int kRebaseBits = 3;

// up to two generator sets (G and H of one session) share the launches
struct RebaseJob {
  const u32* gx[2];
  const u32* gy[2];
  const uint8_t* ginf[2];
  const int8_t* digits[2];  // m x kRebaseWindows each
  u32* win[2];              // n x kRebaseWindows XYZZ points each
  u32* ox[2];
  u32* oy[2];
  uint8_t* oinf[2];
};

template <class C>
__global__ __launch_bounds__(256) void k_rebase_windows(RebaseJob job, size_t n, size_t m) {
  using P = typename C::Pipeline;
  using F = typename P::Fq;
  const int w = (int)blockIdx.y, set = (int)blockIdx.z;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32* __restrict__ gx = job.gx[set];
  const u32* __restrict__ gy = job.gy[set];
  const uint8_t* __restrict__ ginf = job.ginf[set];
  const int8_t* __restrict__ digits = job.digits[set];
  // The four bucket sums live in four register slots; the bucket a step adds to is first swapped
  // into slot 0 (the digit is the same for the whole block, so which slot holds which bucket is
  // scalar state and the swap a uniform branch), and ONE inlined mixed addition works on slot 0.
  // With an addition per bucket the loop body was four copies of it, 0.4 MB of code against a
  // 64 KiB instruction cache.
  Xyzz s0 = xyzz_identity<F>(), s1 = s0, s2 = s0, s3 = s0;
  int at0 = 1, at1 = 2, at2 = 3, at3 = 4;  // bucket held by each slot
  auto swap_slots = [](Xyzz& a, Xyzz& b) {
    const Xyzz tmp = a;
    a = b;
    b = tmp;
  };
  for (size_t t = 0; t < m; t++) {
    const int d = digits[t * kRebaseWindows + w];  // the same for every lane of the block
    if (d == 0) continue;
    const size_t k = i + n * t;
    const int a = d < 0 ? -d : d;
    if (at1 == a) {
      swap_slots(s0, s1);
      at1 = at0;
      at0 = a;
    } else if (at2 == a) {
      swap_slots(s0, s2);
      at2 = at0;
      at0 = a;
    } else if (at3 == a) {
      swap_slots(s0, s3);
      at3 = at0;
      at0 = a;
    }
    if (ginf && ginf[k]) continue;
    const Fe x = fe_import<P>(fe_load(gx + k * 8));
    Fe y = fe_import<P>(fe_load(gy + k * 8));
    if (d < 0) y = fe_neg<F>(y);
    xyzz_madd<F>(s0, x, y);
  }
  // back to the buckets by number: a sorting network over the four slots (uniform branches again)
  auto order = [&](Xyzz& a, int& ka, Xyzz& b, int& kb) {
    if (ka > kb) {
      swap_slots(a, b);
      const int k = ka;
      ka = kb;
      kb = k;
    }
  };
  order(s0, at0, s1, at1);
  order(s2, at2, s3, at3);
  order(s0, at0, s2, at2);
  order(s1, at1, s3, at3);
  order(s1, at1, s2, at2);
  const Xyzz &b1 = s0, &b2 = s1, &b3 = s2, &b4 = s3;
  // b1 + 2 b2 + 3 b3 + 4 b4 by running sums
  Xyzz run = b4, sum = b4;
  xyzz_add<F>(run, b3);
  xyzz_add<F>(sum, run);
  xyzz_add<F>(run, b2);
  xyzz_add<F>(sum, run);
  xyzz_add<F>(run, b1);
  xyzz_add<F>(sum, run);
  u32* o = job.win[set] + ((size_t)w * n + i) * 32;
  fe_store(o, sum.x);
  fe_store(o + 8, sum.y);
  fe_store(o + 16, sum.zz);
  fe_store(o + 24, sum.zzz);
}

// sum_w 8^w * window sum, one output per quad of lanes (bp_quad.h): the chain of 258 doublings and
// 86 additions per output is what this launch costs, and a quad runs it in about half the time of
// one lane.  Lane 0 of the quad then normalises.
template <class C>
__global__ __launch_bounds__(256) void k_rebase_finish(RebaseJob job, size_t n, int sets) {
  using P = typename C::Pipeline;
  using F = typename P::Fq;
  constexpr int kQuads = 64;
  __shared__ u32 pts[kQuads * 2 * 32];  // per quad: the running value, the window sum being added
  __shared__ u32 scr_all[kQuads * kQuadScratch * 8];
  __shared__ u32 zero[8];
  const u32 tid = threadIdx.x, quad = tid >> 2;
  const int role = (int)(tid & 3u), coord = quad_coord(role);
  u32* scr = scr_all + quad * (kQuadScratch * 8);
  u32* acc = pts + quad * 64;
  if (tid < 8) zero[tid] = 0;
  const size_t o = (size_t)blockIdx.x * kQuads + quad;
  const bool live = o < n * (size_t)sets;
  const int set = live ? (int)(o / n) : 0;
  const size_t i = live ? o % n : 0;
  fe_store(acc + role * 8, quad_identity_coord(role, fe_one<F>()));
  __syncthreads();
  if (!live) return;
  const u32* __restrict__ win = job.win[set];
  for (int w = kRebaseWindows - 1; w >= 0; w--) {
#pragma unroll 1
    for (int k = 0; k < kRebaseBits; k++) quad_dbl<F>(acc, scr, zero, role);
    fe_store(acc + 32 + role * 8, fe_load(win + ((size_t)w * n + i) * 32 + role * 8));
    quad_sync();
    Fe out;
    const bool changed = quad_add<F>(pts, scr, zero, quad * 2, quad * 2 + 1, role, out);
    quad_sync();
    if (changed) fe_store(acc + coord * 8, out);
    quad_sync();
  }
  if (role == 0) {
    Xyzz v;
    v.x = fe_load(acc);
    v.y = fe_load(acc + 8);
    v.zz = fe_load(acc + 16);
    v.zzz = fe_load(acc + 24);
    Fe x, y;
    const bool inf = xyzz_to_affine<F>(v, x, y);
    fe_store(job.ox[set] + i * 8, inf ? x : fe_export<P>(x));
    fe_store(job.oy[set] + i * 8, inf ? y : fe_export<P>(y));
    job.oinf[set][i] = inf ? 1 : 0;
  }
}

// The same combination with one lane per output, for launches that fill the machine with lanes
// anyway: a quad finishes one output in half the time but issues 2.2 times the instructions, which
// pays only while SIMDs would otherwise idle (65 536 outputs: 3.3 ms on quads, a chain of about
// 3 500 dependent multiplications per lane at one wave per SIMD here).
template <class C>
__global__ __launch_bounds__(256) void k_rebase_finish_lane(RebaseJob job, size_t n, int sets) {
  using P = typename C::Pipeline;
  using F = typename P::Fq;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n * (size_t)sets) return;
  const int set = (int)(o / n);
  const size_t i = o % n;
  const u32* __restrict__ win = job.win[set];
  Xyzz acc = xyzz_identity<F>();
#pragma unroll 1
  for (int w = kRebaseWindows - 1; w >= 0; w--) {
#pragma unroll 1
    for (int k = 0; k < kRebaseBits; k++) acc = xyzz_dbl<F>(acc);
    const u32* src = win + ((size_t)w * n + i) * 32;
    Xyzz v;
    v.x = fe_load(src);
    v.y = fe_load(src + 8);
    v.zz = fe_load(src + 16);
    v.zzz = fe_load(src + 24);
    xyzz_add<F>(acc, v);
  }
  Fe x, y;
  const bool inf = xyzz_to_affine<F>(acc, x, y);
  fe_store(job.ox[set] + i * 8, inf ? x : fe_export<P>(x));
  fe_store(job.oy[set] + i * 8, inf ? y : fe_export<P>(y));
  job.oinf[set][i] = inf ? 1 : 0;
}

// outputs from which the combination runs one lane per output (about a wave per SIMD)
constexpr size_t kRebaseLaneFinishMin = 32768;

hipError_t launch_ipa_rebase(int curve, const RebaseSet* sets, int n_sets, size_t n, size_t m, hipStream_t st) {
  if (n == 0 || m == 0 || n_sets < 1 || n_sets > 2) return hipErrorInvalidValue;
  RebaseJob job{};
  for (int k = 0; k < n_sets; k++) {
    job.gx[k] = sets[k].gx; job.gy[k] = sets[k].gy; job.ginf[k] = sets[k].ginf; job.digits[k] = sets[k].d_digits;
    job.win[k] = sets[k].tmp_windows; job.ox[k] = sets[k].ox; job.oy[k] = sets[k].oy; job.oinf[k] = sets[k].oinf;
  }
  const unsigned bx = (unsigned)((n + 255) / 256);
  const unsigned fb = (unsigned)((n * (size_t)n_sets + 63) / 64);
  DISPATCH_CURVE(curve, {
    hipLaunchKernelGGL(k_rebase_windows<C>, dim3(bx, kRebaseWindows, (unsigned)n_sets), dim3(256), 0, st, job, n, m);
    if (n * (size_t)n_sets >= kRebaseLaneFinishMin && !getenv("BP_REBASE_QUAD_FINISH"))
      hipLaunchKernelGGL(k_rebase_finish_lane<C>, dim3((unsigned)((n * (size_t)n_sets + 255) / 256)), dim3(256), 0, st,
                         job, n, n_sets);
    else
      hipLaunchKernelGGL(k_rebase_finish<C>, dim3(fb), dim3(256), 0, st, job, n, n_sets);
  });
  return hipGetLastError();
}

// ---- folded generators from window tables ------------------------------------------------------
struct RebaseTabJob {
  const u32* table;
  size_t stride;
  RebaseTabSet set[2];
};

constexpr u32 kRebaseTabThreads = 256;

// point `which` of lane `lane` of a block: word k at (which * 32 + k) * kRebaseTabThreads + lane
BP_DEV Xyzz rebase_lds_load(const u32* pts, u32 which, u32 lane) {
  Xyzz p;
  const u32* b = pts + which * 32 * kRebaseTabThreads + lane;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    p.x.v[k] = b[k * kRebaseTabThreads];
    p.y.v[k] = b[(8 + k) * kRebaseTabThreads];
    p.zz.v[k] = b[(16 + k) * kRebaseTabThreads];
    p.zzz.v[k] = b[(24 + k) * kRebaseTabThreads];
  }
  return p;
}
BP_DEV void rebase_lds_store(u32* pts, u32 which, u32 lane, const Xyzz& p) {
  u32* b = pts + which * 32 * kRebaseTabThreads + lane;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    b[k * kRebaseTabThreads] = p.x.v[k];
    b[(8 + k) * kRebaseTabThreads] = p.y.v[k];
    b[(16 + k) * kRebaseTabThreads] = p.zz.v[k];
    b[(24 + k) * kRebaseTabThreads] = p.zzz.v[k];
  }
}

// UNIFY: a table record is added as a full point, so that lanes of a wave which run different lists
// (log_s > 0) still run one instruction stream; without it (one list for all lanes) a record is a mixed
// addition and the choice between the two additions a uniform branch
template <class C, bool UNIFY>
__global__ __launch_bounds__(kRebaseTabThreads) void k_rebase_tab(RebaseTabJob job, size_t n, int log_s, int len) {
  using P = typename C::Pipeline;
  using F = typename P::Fq;
  __shared__ u32 pts[3 * 32 * kRebaseTabThreads];
  extern __shared__ u32 prog[];  // the lists: (len << log_s) words
  // blockIdx.z: one of several problems over the same table (their own lists and outputs, z_stride apart)
  RebaseTabSet set = job.set[blockIdx.y];
  set.prog += (size_t)blockIdx.z * set.prog_z_stride;
  set.ox += (size_t)blockIdx.z * set.out_z_stride * 8;
  set.oy += (size_t)blockIdx.z * set.out_z_stride * 8;
  set.oinf += (size_t)blockIdx.z * set.out_z_stride;
  const u32 tid = threadIdx.x, lanes = 1u << log_s;
  for (u32 k = tid; k < lanes * (u32)len; k += kRebaseTabThreads) prog[k] = set.prog[k];
  {
    const Xyzz id = xyzz_identity<F>();
    for (u32 which = 0; which < 3; which++) rebase_lds_store(pts, which, tid, id);
  }
  __syncthreads();
  const size_t g = (size_t)blockIdx.x * kRebaseTabThreads + tid;
  const size_t i = g >> log_s;
  const u32 s = (u32)g & (lanes - 1);
  const bool live = i < n;  // the lanes of an output are live together
  const u32* __restrict__ table = job.table;
#pragma unroll 1
  for (int slot = 0; slot < len; slot++) {
    const u32 op = prog[s * (u32)len + (u32)slot];
    const u32 dst = op >> 30, src = (op >> 28) & 3u;
    if (!live || dst == 3u) continue;
    Xyzz p = rebase_lds_load(pts, dst, tid);
    Xyzz q = xyzz_identity<F>();
    bool mixed = false;
    Fe x = fe_zero(), y = fe_zero();
    if (src == 3u) {
      const size_t j = (size_t)set.off + i + n * (size_t)(op & 0x3FFFFFu);
      if (!(set.inf && set.inf[j])) {
        const u32* rec = table + ((size_t)((op >> 22) & 31u) * job.stride + j) * 16;
        x = fe_load(rec);
        y = fe_load(rec + 8);
        if ((op >> 27) & 1u) y = fe_neg<F>(y);
        if constexpr (UNIFY) {
          q = xyzz_from_affine<F>(x, y);
        } else {
          mixed = true;
        }
      }
    } else {
      q = rebase_lds_load(pts, src, tid);
    }
    // one addition for every kind of step when the lanes of a wave run different lists
    if (mixed)
      xyzz_madd<F>(p, x, y);
    else
      xyzz_add<F>(p, q);
    rebase_lds_store(pts, dst, tid, p);
  }
  // the lanes of an output sit side by side in one wave: their totals are added by a tree
  for (u32 d = lanes >> 1; d > 0; d >>= 1) {
    quad_sync();
    if (live && s < d) {
      Xyzz p = rebase_lds_load(pts, 1, tid);
      const Xyzz q = rebase_lds_load(pts, 1, tid + d);
      xyzz_add<F>(p, q);
      quad_sync();
      rebase_lds_store(pts, 1, tid, p);
    }
  }
  if (live && s == 0) {
    const Xyzz v = rebase_lds_load(pts, 1, tid);
    Fe x, y;
    const bool inf = xyzz_to_affine<F>(v, x, y);
    fe_store(set.ox + i * 8, inf ? x : fe_export<P>(x));
    fe_store(set.oy + i * 8, inf ? y : fe_export<P>(y));
    set.oinf[i] = inf ? 1 : 0;
  }
}

hipError_t launch_ipa_rebase_tab(int curve, const u32* table, size_t stride, const RebaseTabSet* sets, int n_sets, size_t n,
                                 int log_s, int len, hipStream_t st, int n_problems) {
  if (!table || n == 0 || n_sets < 1 || n_sets > 2 || log_s < 0 || log_s > kRebaseTabMaxLogS || len < 1 || n_problems < 1 ||
      ((size_t)len << log_s) > kRebaseTabProgMax)
    return hipErrorInvalidValue;
  RebaseTabJob job{};
  job.table = table;
  job.stride = stride;
  for (int k = 0; k < n_sets; k++) job.set[k] = sets[k];
  const unsigned bx = (unsigned)((((size_t)n << log_s) + kRebaseTabThreads - 1) / kRebaseTabThreads);
  const size_t prog_bytes = ((size_t)len << log_s) * sizeof(u32);
  DISPATCH_CURVE(curve, {
    if (log_s > 0)
      hipLaunchKernelGGL((k_rebase_tab<C, true>), dim3(bx, (unsigned)n_sets, (unsigned)n_problems), dim3(kRebaseTabThreads),
                         prog_bytes, st, job, n, log_s, len);
    else
      hipLaunchKernelGGL((k_rebase_tab<C, false>), dim3(bx, (unsigned)n_sets, (unsigned)n_problems), dim3(kRebaseTabThreads),
                         prog_bytes, st, job, n, log_s, len);
  });
  return hipGetLastError();
}

// dst[r] = src[r * stride_elems] (field elements): one column of a matrix of scalars
__global__ __launch_bounds__(256) void k_gather_column(const u32* __restrict__ src, size_t stride_elems, size_t count,
                                                       u32* __restrict__ dst) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 8) return;
  dst[t] = src[(t / 8) * stride_elems * 8 + (t % 8)];
}

hipError_t launch_gather_column(const u32* src, size_t stride_elems, size_t count, u32* dst, hipStream_t st) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_column, dim3((unsigned)((count * 8 + 255) / 256)), dim3(256), 0, st, src, stride_elems, count, dst);
  return hipGetLastError();
}

// Array of 72-byte affine records {x: 8 u32, y: 8 u32, infinity: 1 byte, 7 bytes of padding} ->
// the x, y and infinity planes the pipeline reads.  One lane per 32-bit word: the reads are
// contiguous over the whole array and each plane's writes are contiguous over a record's words.
__global__ __launch_bounds__(256) void k_aos_split(const u32* __restrict__ aos, size_t n, u32* __restrict__ x,
                                                   u32* __restrict__ y, uint8_t* __restrict__ inf) {
  const size_t words = n * 18;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < words; t += (size_t)gridDim.x * blockDim.x) {
    const size_t i = t / 18;
    const u32 w = (u32)(t % 18);
    if (w >= 17) continue;  // padding
    const u32 v = aos[t];
    if (w < 8)
      x[i * 8 + w] = v;
    else if (w < 16)
      y[i * 8 + (w - 8)] = v;
    else
      inf[i] = (v & 0xffu) ? 1 : 0;
  }
}

hipError_t launch_aos_split(const u32* aos, size_t n, u32* x, u32* y, uint8_t* inf, hipStream_t st) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n * 18 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_aos_split, dim3((unsigned)blocks), dim3(256), 0, st, aos, n, x, y, inf);
  return hipGetLastError();
}

}  // namespace bp
