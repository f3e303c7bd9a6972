// Curve-vector kernels for gfx950: generator fold, per-point scaling, batch
// normalisation.  One thread owns one output point; the double-and-add schedule
// of the generator fold is shared by all lanes (the two scalars are the same for
// every output), so there is no divergence on the common path.
#include "bp_curve.h"
#include "bp_internal.h"

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
