// Micro-benchmarks of the arithmetic building blocks, exported for bench/profiling
// scripts through bp_microbench() (not part of the product path).
#include "bp_curve.h"
#include "bp_internal.h"

namespace bp {

template <class F>
__global__ __launch_bounds__(256) void k_mb_mul(int iters, u32* __restrict__ io) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fe x = fe_load(io + (t & 1023) * 8), y = fe_load(io + ((t + 1) & 1023) * 8);
  for (int i = 0; i < iters; i++) {
    x = fe_mul<F>(x, y);
    y = fe_mul<F>(y, x);
  }
  if (x.v[0] == 0x12345 && y.v[3] == 0x54321) fe_store(io, x);
}

// a*b + c*d with one reduction (fe_mul2), two per step
template <class F>
__global__ __launch_bounds__(256) void k_mb_mul2(int iters, u32* __restrict__ io) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fe x = fe_load(io + (t & 1023) * 8), y = fe_load(io + ((t + 1) & 1023) * 8);
  for (int i = 0; i < iters; i++) {
    x = fe_mul2<F>(x, y, y, x);
    y = fe_mul2<F>(y, x, x, y);
  }
  if (x.v[0] == 0x12345 && y.v[3] == 0x54321) fe_store(io, x);
}

template <class F>
__global__ __launch_bounds__(256) void k_mb_addsub(int iters, u32* __restrict__ io) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fe x = fe_load(io + (t & 1023) * 8), y = fe_load(io + ((t + 1) & 1023) * 8);
  for (int i = 0; i < iters; i++) {
    x = fe_add<F>(x, y);
    y = fe_sub<F>(y, x);
  }
  if (x.v[0] == 0x12345 && y.v[3] == 0x54321) fe_store(io, x);
}

template <class F>
__global__ __launch_bounds__(128) void k_mb_madd(int iters, u32* __restrict__ io) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fe qx = fe_load(io + (t & 1023) * 8), qy = fe_load(io + ((t + 1) & 1023) * 8);
  Xyzz acc = xyzz_from_affine<F>(qy, qx);
  for (int i = 0; i < iters; i++) {
    xyzz_madd<F>(acc, qx, qy);
    qx.v[0] ^= acc.x.v[0] & 1;
  }
  if (acc.x.v[0] == 0x12345 && acc.y.v[3] == 0x54321) fe_store(io, acc.x);
}

template <class F>
__global__ __launch_bounds__(128) void k_mb_jdbl(int iters, u32* __restrict__ io) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Jac a;
  a.x = fe_load(io + (t & 1023) * 8);
  a.y = fe_load(io + ((t + 1) & 1023) * 8);
  a.z = fe_load(io + ((t + 2) & 1023) * 8);
  for (int i = 0; i < iters; i++) a = jac_dbl<F>(a);
  if (a.x.v[0] == 0x12345 && a.y.v[3] == 0x54321) fe_store(io, a.x);
}

// Self-check of the plain secp256k1 multiply: the straight-line device form against the portable
// form on a chain of products; io[0] receives the number of lanes that disagreed.
__global__ __launch_bounds__(256) void k_mb_mul_check(int iters, u32* __restrict__ io) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fe x = fe_load(io + (8 + (t & 511)) * 8), y = fe_load(io + (8 + ((t + 1) & 511)) * 8);
  if ((t & 3) == 1) { for (int i = 0; i < 8; i++) x.v[i] = 0xffffffffu; }   // non-canonical, all ones
  if ((t & 3) == 2) { for (int i = 0; i < 8; i++) y.v[i] = Secp256k1FqP::mod(i); y.v[0] -= 1; }  // p - 1
  bool bad = false;
  for (int i = 0; i < iters; i++) {
    Fe d = fe_mul<Secp256k1FqP>(x, y), p = fe_mul_plain_secp_portable(x, y);
    bad |= !fe_eq(d, p);
    // canonical: d < p
    uint32_t bw = 0;
    for (int k = 0; k < 8; k++) (void)__builtin_subc(d.v[k], Secp256k1FqP::mod(k), bw, &bw);
    bad |= bw == 0;
    x = y;
    y = d;
    y.v[0] ^= (uint32_t)i;  // leaves the canonical range now and then: inputs need not be reduced
  }
  if (bad) atomicAdd(io, 1u);
}

// Reads every 64-byte record of a table exactly once, in a scattered order (lane t reads
// record t*mult mod records, mult odd), with the four 16-byte loads per lane that the MSM
// accumulation kernel uses for its point gathers: a known byte count in that access
// pattern, against which the memory-side counters can be calibrated.
__global__ __launch_bounds__(128) void k_mb_gather64(const u32* __restrict__ table, size_t records,
                                                     u32* __restrict__ sink) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= records) return;
  size_t idx = (t * 0x9E3779B1ull) & (records - 1);
  const uint4* q = reinterpret_cast<const uint4*>(table + idx * 16);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  u32 x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  if (x == 0x1234567u) sink[0] = x;
}

// ---- batch-affine additions ------------------------------------------------------------------
// T independent affine additions P_i + Q_i per lane with ONE field inversion for all of them (the product
// trick): pass 1 multiplies the differences dx_i = x2_i - x1_i into a running product and parks the product
// before each factor in memory (32 bytes per addition, written and read back once; T of them do not fit a
// lane's registers); one inversion; pass 2 walks back, peels 1/dx_i off the running inverse (2 M), and
// finishes the addition: lambda = dy/dx (1 M), x3 = lambda^2 - x1 - x2 (1 S), y3 = lambda (x1 - x3) - y1
// (1 M).  6 products per addition plus 1/T of an inversion, against 10 for the XYZZ mixed addition, at the
// price of the staging traffic -- and this kernel leaves out the rest of that price: its operands are made
// in registers (x1_i = x1_0 + i c), where an MSM would read and write two points per addition.
template <class F>
__global__ __launch_bounds__(128) void k_mb_batch_affine(int iters, const u32* __restrict__ io, u32* __restrict__ park,
                                                         u32* __restrict__ sink) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (size_t)gridDim.x * blockDim.x;
  const Fe c1 = fe_load(io + ((t + 3) & 1023) * 8), c2 = fe_load(io + ((t + 5) & 1023) * 8);
  Fe x1 = fe_load(io + (t & 1023) * 8), y1 = fe_load(io + ((t + 1) & 1023) * 8);
  Fe x2 = fe_load(io + ((t + 2) & 1023) * 8), y2 = fe_load(io + ((t + 7) & 1023) * 8);
  Fe run = fe_load(io + ((t + 9) & 1023) * 8);
  for (int i = 0; i < iters; i++) {
    fe_store(park + ((size_t)i * lanes + t) * 8, run);
    run = fe_mul<F>(run, fe_sub<F>(x2, x1));
    x1 = fe_add<F>(x1, c1);
    x2 = fe_add<F>(x2, c2);
  }
  Fe inv = fe_inv<F>(run);
  Fe chk = fe_zero();
  for (int i = iters - 1; i >= 0; i--) {
    x1 = fe_sub<F>(x1, c1);
    x2 = fe_sub<F>(x2, c2);
    const Fe dx = fe_sub<F>(x2, x1);
    const Fe dxi = fe_mul<F>(inv, fe_load(park + ((size_t)i * lanes + t) * 8));
    inv = fe_mul<F>(inv, dx);
    const Fe lam = fe_mul<F>(fe_sub<F>(y2, y1), dxi);
    const Fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(lam), x1), x2);
    const Fe y3 = fe_sub<F>(fe_mul<F>(lam, fe_sub<F>(x1, x3)), y1);
    chk = fe_add<F>(chk, fe_add<F>(x3, y3));
    y1 = fe_add<F>(y1, c2);  // other operands for the next addition
    y2 = fe_add<F>(y2, c1);
  }
  if (chk.v[0] == 0x12345 && chk.v[3] == 0x54321) fe_store(sink, chk);
}

// The same arithmetic on real points, checked: lane t builds the chain P_0 = G, Q_0 = 2G, P_{i+1} = P_i + Q_i,
// Q_{i+1} = P_i with the XYZZ formulas, then adds every pair again the batch-affine way (registers only) and
// counts the sums that differ; io[0] receives the number of lanes with a difference.
__global__ __launch_bounds__(64) void k_mb_batch_affine_check(u32* __restrict__ io) {
  using F = Secp256k1FqP;
  constexpr int T = 12;
  const u32 gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
  const u32 gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
  Fe px[T + 1], py[T + 1], qx[T], qy[T];
  for (int k = 0; k < 8; k++) { px[0].v[k] = gx[k]; py[0].v[k] = gy[k]; }
  {
    // Q_0 = (2 + lane) G by repeated addition of G: a different chain per lane
    Xyzz d = xyzz_dbl_affine<F>(px[0], py[0]);
    for (u32 k = 0; k < (threadIdx.x & 7u); k++) xyzz_madd<F>(d, px[0], py[0]);
    xyzz_to_affine<F>(d, qx[0], qy[0]);
  }
  for (int i = 0; i < T; i++) {
    Xyzz s = xyzz_from_affine<F>(px[i], py[i]);
    xyzz_madd<F>(s, qx[i], qy[i]);
    xyzz_to_affine<F>(s, px[i + 1], py[i + 1]);
    if (i + 1 < T) { qx[i + 1] = px[i]; qy[i + 1] = py[i]; }
  }
  Fe pre[T];
  Fe run = fe_one<F>();
  for (int i = 0; i < T; i++) {
    pre[i] = run;
    run = fe_mul<F>(run, fe_sub<F>(qx[i], px[i]));
  }
  Fe inv = fe_inv<F>(run);
  bool bad = false;
  for (int i = T - 1; i >= 0; i--) {
    const Fe dx = fe_sub<F>(qx[i], px[i]);
    const Fe dxi = fe_mul<F>(inv, pre[i]);
    inv = fe_mul<F>(inv, dx);
    const Fe lam = fe_mul<F>(fe_sub<F>(qy[i], py[i]), dxi);
    const Fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(lam), px[i]), qx[i]);
    const Fe y3 = fe_sub<F>(fe_mul<F>(lam, fe_sub<F>(px[i], x3)), py[i]);
    bad |= !fe_eq(fe_sub<F>(x3, px[i + 1]), fe_zero()) || !fe_eq(fe_sub<F>(y3, py[i + 1]), fe_zero());
  }
  if (bad) atomicAdd(io, 1u);
}

hipError_t launch_batch_affine_bench(int blocks, int iters, const u32* io, u32* park, u32* sink, hipStream_t st) {
  hipLaunchKernelGGL(k_mb_batch_affine<Secp256k1FqP>, dim3(blocks), dim3(128), 0, st, iters, io, park, sink);
  return hipGetLastError();
}

hipError_t launch_batch_affine_check(int blocks, u32* io, hipStream_t st) {
  hipLaunchKernelGGL(k_mb_batch_affine_check, dim3(blocks), dim3(64), 0, st, io);
  return hipGetLastError();
}

hipError_t launch_gather_bench(const u32* table, size_t records, u32* sink, hipStream_t st) {
  hipLaunchKernelGGL(k_mb_gather64, dim3((unsigned)((records + 127) / 128)), dim3(128), 0, st, table, records, sink);
  return hipGetLastError();
}

hipError_t launch_microbench(int id, int blocks, int iters, u32* io, hipStream_t st) {
  using F = Secp256k1Fq;
  switch (id) {
    case 0: hipLaunchKernelGGL(k_mb_mul<F>, dim3(blocks), dim3(256), 0, st, iters, io); break;
    case 1: hipLaunchKernelGGL(k_mb_addsub<F>, dim3(blocks), dim3(256), 0, st, iters, io); break;
    case 2: hipLaunchKernelGGL(k_mb_madd<F>, dim3(blocks), dim3(128), 0, st, iters, io); break;
    case 3: hipLaunchKernelGGL(k_mb_jdbl<F>, dim3(blocks), dim3(128), 0, st, iters, io); break;
    case 4: hipLaunchKernelGGL(k_mb_mul<VestaFq>, dim3(blocks), dim3(256), 0, st, iters, io); break;
    case 6: hipLaunchKernelGGL(k_mb_mul<Secp256k1FqP>, dim3(blocks), dim3(256), 0, st, iters, io); break;
    case 7: hipLaunchKernelGGL(k_mb_madd<Secp256k1FqP>, dim3(blocks), dim3(128), 0, st, iters, io); break;
    case 8: hipLaunchKernelGGL(k_mb_mul_check, dim3(blocks), dim3(256), 0, st, iters, io); break;
    case 9: hipLaunchKernelGGL(k_mb_mul<Secp256k1Fr>, dim3(blocks), dim3(256), 0, st, iters, io); break;
    case 10: hipLaunchKernelGGL(k_mb_mul2<Secp256k1Fr>, dim3(blocks), dim3(256), 0, st, iters, io); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace bp
