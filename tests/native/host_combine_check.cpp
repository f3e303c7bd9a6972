// CPU check of the Jacobian window combination (hj_combine_windows: plain residues for secp256k1,
// Montgomery form for Vesta) against the XYZZ chain it replaces, on random multiples of a point,
// with identities and repeated points among the windows; prints the time of one combination.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <cstring>

#include "bp_host_math.h"
using namespace bp;
using namespace bp::host;

// the standard generators: secp256k1's G (Montgomery form), Vesta's (-1, 2)
template <class F>
HXyzz generator();
template <>
HXyzz generator<Secp256k1Fq>() {
  HXyzz g;
  g.x = HFe{{0xd7362e5a487e2097ull, 0x231e295329bc66dbull, 0x979f48c033fd129cull, 0x9981e643e9089f48ull}};
  g.y = HFe{{0xb15ea6d2d3dbabe2ull, 0x8dfc5d5d1f1dc64dull, 0x70b6b59aac19c136ull, 0xcf3f851fd4a582d6ull}};
  g.zz = g.zzz = h_one<Secp256k1Fq>();
  return g;
}
template <>
HXyzz generator<VestaFq>() {
  typedef VestaFq F;
  const HFe r2{{F::r264(0), F::r264(1), F::r264(2), F::r264(3)}};
  HXyzz g;
  g.x = h_neg<F>(h_one<F>());
  g.y = h_mul<F>(HFe{{2, 0, 0, 0}}, r2);
  g.zz = g.zzz = h_one<F>();
  return g;
}

template <class F, class O>
int run(const char* name) {
  HXyzz g = generator<F>();
  srand(11);
  int bad = 0;
  double us = 0;
  for (int c : {4, 8, 16}) {
    const int nw = 256 / c;
    for (int t = 0; t < 40; t++) {
      uint64_t win[64 * 16];
      HXyzz p = g;
      for (int w = 0; w < nw; w++) {
        int steps = 1 + rand() % 5;
        for (int k = 0; k < steps; k++) p = (rand() & 1) ? hx_dbl<F>(p) : hx_add<F>(p, g);
        HXyzz q = p;
        if (t % 7 == 3 && w % 3 == 0) q = hx_identity<F>();       // empty windows
        if (t % 7 == 4 && w == nw - 1) q = hx_identity<F>();      // empty top window
        if (t % 7 == 5) q = g;                                    // the same point everywhere
        if (t == 6) q = hx_identity<F>();                         // everything empty
        memcpy(win + w * 16, &q, 128);
      }
      uint64_t a[12], b[12];
      HXyzz r = hx_combine_windows<F>(win, nw, c);
      hx_to_xyz<F>(r, a);
      auto t0 = std::chrono::steady_clock::now();
      hj_combine_windows<O>(win, nw, c, b);
      us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (memcmp(a, b, sizeof a)) bad++;
    }
  }
  printf("%s: %d mismatches, %.1f us per combination\n", name, bad, us / 120);
  return bad;
}

int main() {
  return (run<Secp256k1Fq, SecpPlainOps>("secp256k1 plain") + run<Secp256k1Fq, MontOps<Secp256k1Fq>>("secp256k1 Montgomery") +
          run<VestaFq, MontOps<VestaFq>>("vesta Montgomery")) != 0;
}
