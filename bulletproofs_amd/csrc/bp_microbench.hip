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
