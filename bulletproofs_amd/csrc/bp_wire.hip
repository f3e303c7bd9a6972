// Wire format of curve points on the device: the compressed encoding that the reference's proof and
// CRS types derive from ark-serialize 0.5 (`CanonicalSerialize` with `Compress::Yes` behind the
// `Encoding` / `NargDeserialize` derives, e.g. src/ipa/types.rs:48,140).  The reference holds no byte
// vector of it, so the layout below restates the published format of the arkworks crates:
//   * a base-field element is its canonical integer in 32 little-endian bytes; with the two flag
//     bits of a short-Weierstrass point, 256 + 2 (secp256k1) and 255 + 2 (Vesta) bits both round up
//     to 33 bytes, and the flags sit in the two top bits of byte 32;
//   * bit 7 = "y is negative" (y > p - y as integers), bit 6 = point at infinity (x written as 0);
//     both set is refused; the six low bits of byte 32 are not read (ark-ff reads 32 bytes of integer);
//   * decoding recomputes y = sqrt(x^3 + b) and takes the smaller root unless bit 7 is set; x >= p or
//     a non-residue is an error.
// Both curves have cofactor one, so a decoded point needs no subgroup check.
#include "bp_curve.h"
#include "bp_internal.h"

namespace bp {

namespace {

// x^3 + b: the curve constant in Montgomery form
template <class F> struct CurveB;
template <> struct CurveB<Secp256k1Fq> {  // b = 7
  __host__ __device__ static constexpr uint32_t b(int i) { constexpr uint32_t t[8] = {0x00001ab7u, 0x00000007u, 0, 0, 0, 0, 0, 0}; return t[i]; }
};
template <> struct CurveB<VestaFq> {  // b = 5
  __host__ __device__ static constexpr uint32_t b(int i) { constexpr uint32_t t[8] = {0xffffffedu, 0x96bc8c8cu, 0x49f7778eu, 0x74c2a54bu, 0xfffffffdu, 0xffffffffu, 0xffffffffu, 0x3fffffffu}; return t[i]; }
};

// square roots: p = 3 (mod 4) takes a^((p+1)/4); otherwise Tonelli-Shanks over p - 1 = 2^s * t with
// a fixed generator z of the 2^s-th roots of unity (z = 5^t for Vesta's base field, s = 32)
template <class F> struct SqrtConsts;
template <> struct SqrtConsts<Secp256k1Fq> {
  static constexpr int kS = 1;
  // (p + 1) / 4
  __host__ __device__ static constexpr uint32_t exp(int i) { constexpr uint32_t t[8] = {0xbfffff0cu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu}; return t[i]; }
  __host__ __device__ static constexpr uint32_t z(int) { return 0; }
};
template <> struct SqrtConsts<VestaFq> {
  static constexpr int kS = 32;
  // (t - 1) / 2
  __host__ __device__ static constexpr uint32_t exp(int i) { constexpr uint32_t t[8] = {0xc6237590u, 0x04ca546eu, 0x11234c7eu, 0x00000000u, 0x00000000u, 0x00000000u, 0x20000000u, 0x00000000u}; return t[i]; }
  // 5^t, Montgomery form
  __host__ __device__ static constexpr uint32_t z(int i) { constexpr uint32_t t[8] = {0x8c9942deu, 0x21807742u, 0x21b60494u, 0xcc495789u, 0xb2efbee2u, 0xac2e5d27u, 0x7f2db056u, 0x0b79fa89u}; return t[i]; }
};

// a^e for the constant exponent of SqrtConsts<F>, left to right
template <class F>
BP_DEV Fe fe_pow_sqrt_exp(const Fe& a) {
  Fe r = fe_one<F>();
  bool started = false;
#pragma unroll 1
  for (int i = 255; i >= 0; i--) {
    if (started) r = fe_sqr<F>(r);
    if ((SqrtConsts<F>::exp(i >> 5) >> (i & 31)) & 1) {
      r = started ? fe_mul<F>(r, a) : a;
      started = true;
    }
  }
  return r;
}

// false: a is not a square
template <class F>
BP_DEV bool fe_sqrt(const Fe& a, Fe& out) {
  if (fe_is_zero(a)) {
    out = a;
    return true;
  }
  if constexpr (SqrtConsts<F>::kS == 1) {
    out = fe_pow_sqrt_exp<F>(a);
  } else {
    const Fe one = fe_one<F>();
    Fe w = fe_pow_sqrt_exp<F>(a);  // a^((t-1)/2)
    Fe x = fe_mul<F>(a, w);        // a^((t+1)/2)
    Fe b = fe_mul<F>(x, w);        // a^t
    Fe z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = SqrtConsts<F>::z(i);
    int v = SqrtConsts<F>::kS;
#pragma unroll 1
    while (!fe_eq(b, one)) {
      int k = 0;
      Fe b2 = b;
#pragma unroll 1
      while (!fe_eq(b2, one) && k < v) {
        b2 = fe_sqr<F>(b2);
        k++;
      }
      if (k >= v) return false;  // the order of b does not divide 2^(v-1): no root
      Fe w2 = z;
#pragma unroll 1
      for (int j = 0; j < v - k - 1; j++) w2 = fe_sqr<F>(w2);
      z = fe_sqr<F>(w2);
      b = fe_mul<F>(b, z);
      x = fe_mul<F>(x, w2);
      v = k;
    }
    out = x;
  }
  return fe_eq(fe_sqr<F>(out), a);
}

// canonical integers
template <class F>
BP_DEV bool below_modulus(const Fe& a) {
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (a.v[i] < F::mod(i)) return true;
    if (a.v[i] > F::mod(i)) return false;
  }
  return false;
}

// a > p - a for a canonical a (a = 0: no)
template <class F>
BP_DEV bool above_negation(const Fe& a) {
  if (fe_is_zero(a)) return false;
  Fe n;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t d = (uint64_t)F::mod(i) - a.v[i] - borrow;
    n.v[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (a.v[i] > n.v[i]) return true;
    if (a.v[i] < n.v[i]) return false;
  }
  return false;
}

template <class C>
__global__ __launch_bounds__(256) void k_points_compress(const u32* __restrict__ x, const u32* __restrict__ y,
                                                         const uint8_t* __restrict__ inf, size_t n,
                                                         uint8_t* __restrict__ out) {
  using F = typename C::Fq;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t* o = out + i * 33;
  if (inf && inf[i]) {
    for (int k = 0; k < 32; k++) o[k] = 0;
    o[32] = 0x40;
    return;
  }
  const Fe xc = fe_from_mont<F>(fe_load(x + i * 8));
  const Fe yc = fe_from_mont<F>(fe_load(y + i * 8));
#pragma unroll
  for (int k = 0; k < 8; k++) {
    o[4 * k] = (uint8_t)xc.v[k];
    o[4 * k + 1] = (uint8_t)(xc.v[k] >> 8);
    o[4 * k + 2] = (uint8_t)(xc.v[k] >> 16);
    o[4 * k + 3] = (uint8_t)(xc.v[k] >> 24);
  }
  o[32] = above_negation<F>(yc) ? 0x80 : 0x00;
}

template <class C>
__global__ __launch_bounds__(256) void k_points_decompress(const uint8_t* __restrict__ in, size_t n,
                                                           u32* __restrict__ x, u32* __restrict__ y,
                                                           uint8_t* __restrict__ inf, uint8_t* __restrict__ ok) {
  using F = typename C::Fq;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = in + i * 33;
  Fe xc;
#pragma unroll
  for (int k = 0; k < 8; k++)
    xc.v[k] = (u32)p[4 * k] | ((u32)p[4 * k + 1] << 8) | ((u32)p[4 * k + 2] << 16) | ((u32)p[4 * k + 3] << 24);
  const uint8_t flags = p[32];
  const bool negative = (flags & 0x80) != 0, infinity = (flags & 0x40) != 0;
  Fe ox = fe_zero(), oy = fe_zero();
  bool good = below_modulus<F>(xc) && !(negative && infinity);
  bool is_inf = false;
  if (good && infinity) {
    is_inf = true;
  } else if (good) {
    const Fe xm = fe_to_mont<F>(xc);
    Fe b;
#pragma unroll
    for (int k = 0; k < 8; k++) b.v[k] = CurveB<F>::b(k);
    const Fe rhs = fe_add<F>(fe_mul<F>(fe_sqr<F>(xm), xm), b);
    Fe root;
    good = fe_sqrt<F>(rhs, root);
    if (good) {
      // the smaller root unless the flag asks for the larger one
      const bool root_is_larger = above_negation<F>(fe_from_mont<F>(root));
      ox = xm;
      oy = root_is_larger == negative ? root : fe_neg<F>(root);
    }
  }
  fe_store(x + i * 8, ox);
  fe_store(y + i * 8, oy);
  inf[i] = is_inf ? 1 : 0;
  ok[i] = good ? 1 : 0;
}

// Scalars: the canonical integer in 32 little-endian bytes (ark-serialize writes a field element
// without flags in ceil(bits / 8) bytes: 32 for both scalar fields); a value >= the modulus is no scalar.
template <class F>
__global__ __launch_bounds__(256) void k_scalars_to_bytes(const u32* __restrict__ v, size_t n, u32* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store(out + i * 8, fe_from_mont<F>(fe_load(v + i * 8)));  // little-endian limbs are little-endian bytes
}

template <class F>
__global__ __launch_bounds__(256) void k_scalars_from_bytes(const u32* __restrict__ in, size_t n, u32* __restrict__ v,
                                                            uint8_t* __restrict__ ok) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe c = fe_load(in + i * 8);
  const bool good = below_modulus<F>(c);
  fe_store(v + i * 8, good ? fe_to_mont<F>(c) : fe_zero());
  ok[i] = good ? 1 : 0;
}

}  // namespace

#define DISPATCH_CURVE_W(curve, STMT) \
  switch (curve) {                    \
    case BP_SECP256K1: {              \
      using C = Secp256k1;            \
      STMT;                           \
    } break;                          \
    case BP_VESTA: {                  \
      using C = Vesta;                \
      STMT;                           \
    } break;                          \
    default:                          \
      return hipErrorInvalidValue;    \
  }

hipError_t launch_points_compress(int curve, const u32* x, const u32* y, const uint8_t* inf, size_t n, uint8_t* out,
                                  hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE_W(curve, hipLaunchKernelGGL(k_points_compress<C>, dim3(blocks), dim3(256), 0, st, x, y, inf, n, out));
  return hipGetLastError();
}

hipError_t launch_points_decompress(int curve, const uint8_t* in, size_t n, u32* x, u32* y, uint8_t* inf, uint8_t* ok,
                                    hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE_W(curve,
                   hipLaunchKernelGGL(k_points_decompress<C>, dim3(blocks), dim3(256), 0, st, in, n, x, y, inf, ok));
  return hipGetLastError();
}

hipError_t launch_scalars_to_bytes(int curve, const u32* v, size_t n, u32* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE_W(curve, hipLaunchKernelGGL(k_scalars_to_bytes<typename C::Fr>, dim3(blocks), dim3(256), 0, st, v, n, out));
  return hipGetLastError();
}

hipError_t launch_scalars_from_bytes(int curve, const u32* in, size_t n, u32* v, uint8_t* ok, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  DISPATCH_CURVE_W(curve,
                   hipLaunchKernelGGL(k_scalars_from_bytes<typename C::Fr>, dim3(blocks), dim3(256), 0, st, in, n, v, ok));
  return hipGetLastError();
}

}  // namespace bp
