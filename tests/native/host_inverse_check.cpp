// CPU check of the host field inverse used for challenges and result normalisation:
// h_inv_kaliski against the Fermat inverse h_inv and against a * a^-1 = 1, all four fields.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bp_host_math.h"
using namespace bp;
using namespace bp::host;

template <class F>
int run(const char* name) {
  srand(7);
  int bad = 0;
  for (int t = 0; t < 3000; t++) {
    HFe a;
    for (int i = 0; i < 4; i++) a.v[i] = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand();
    a.v[3] &= 0x3fffffffffffffffull;  // below every modulus
    if (t == 0) a = h_one<F>();
    if (t == 1) a = HFe{{F::mod64(0) - 1, F::mod64(1), F::mod64(2), F::mod64(3)}};  // p - 1
    if (t == 2) a = HFe{{1, 0, 0, 0}};
    if (t == 3) a = HFe{{2, 0, 0, 0}};
    if (t == 4) a = h_zero();
    if (t >= 5 && t < 5 + 256) {  // powers of two that are below the modulus
      a = h_zero();
      int bit = (t - 5) % 254;
      a.v[bit >> 6] = 1ull << (bit & 63);
    }
    HFe x = h_inv<F>(a), y = h_inv_kaliski<F>(a);
    if (memcmp(x.v, y.v, 32)) bad++;
    if (!h_is_zero(a) && !h_eq(h_mul<F>(a, y), h_one<F>())) bad++;
  }
  printf("%s: %d mismatches\n", name, bad);
  return bad;
}

int main() {
  return (run<Secp256k1Fq>("secp256k1 Fq") + run<Secp256k1Fr>("secp256k1 Fr") + run<VestaFq>("vesta Fq") +
          run<VestaFr>("vesta Fr")) != 0;
}
