// CPU check of the GLV set-up and scalar split (accumulation_amd/csrc/host_glv.h) against the host group law:
// for both curves, lambda / beta pair up on the generator, the lattice vectors are short, and for random and edge-case
// scalars k: k1 + k2 lambda = k (mod r), |k1|, |k2| < 2^131, and [k] P == [k1] P + [k2] phi(P) for P = [7] G.
// Built and run by tests/test_host_glv_cpu.py (no GPU, no libamsm.so).
#include <stdio.h>

#include "host_glv.h"

using namespace amsm;

template <class Fq, class Fr>
static int run(const char* name, int curve) {
  using namespace host;
  constexpr int NQ = HFe<Fq>::N;
  std::vector<u32> g32 = generator_mont<Fq>(curve);
  u64 gen[2 * NQ];
  memcpy(gen, g32.data(), sizeof(gen));
  Glv<Fq, Fr> glv;
  glv.setup(gen);
  if (!glv.ok) {
    printf("%s: set-up failed\n", name);
    return 1;
  }
  printf("%s: basis bits a1 %d b1 %d a2 %d b2 %d\n", name, big_bits(glv.a1), big_bits(glv.b1), big_bits(glv.a2), big_bits(glv.b2));
  HXYZZ<Fq> G = hx_from_affine<Fq>(gen, false);
  u64 seven[4] = {7, 0, 0, 0};
  HXYZZ<Fq> P = hx_mul<Fq>(G, seven);
  u64 pxy[2 * NQ];
  uint8_t inf;
  hx_to_affine<Fq>(P, pxy, &inf);
  HFe<Fq> px;
  memcpy(px.v, pxy, 8 * NQ);
  HFe<Fq> bx = h_mul<Fq>(glv.beta, px);
  u64 phixy[2 * NQ];
  memcpy(phixy, bx.v, 8 * NQ);
  memcpy(phixy + NQ, pxy + NQ, 8 * NQ);
  HXYZZ<Fq> Pa = hx_from_affine<Fq>(pxy, false), Phi = hx_from_affine<Fq>(phixy, false);
  u64 st = 0x243f6a8885a308d3ull;
  auto next = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  int bad = 0, maxbits = 0;
  Big r = big_modulus<Fr>();
  for (int it = 0; it < 300; it++) {
    u64 k[4];
    for (int i = 0; i < 4; i++) k[i] = next();
    if (it < 8) {
      k[0] = it;
      k[1] = k[2] = k[3] = 0;
    } else if (it < 16) {  // r - 1, r - 2, ...
      Big t = big_sub(r, big_small((u32)(it - 7)));
      for (int i = 0; i < 4; i++) k[i] = (u64)t.w[2 * i] | ((u64)t.w[2 * i + 1] << 32);
    } else {
      HFe<Fr> t;
      for (int i = 0; i < 4; i++) t.v[i] = k[i];
      t.v[3] &= 0x0fffffffffffffffull;
      t = h_from_mont<Fr>(h_to_mont<Fr>(t));  // reduce below r
      for (int i = 0; i < 4; i++) k[i] = t.v[i];
    }
    Big k1, k2;
    if (!glv.decompose(k, k1, k2)) {
      bad++;
      continue;
    }
    maxbits = std::max(maxbits, std::max(big_bits(k1), big_bits(k2)));
    GlvDigits d;
    if (!glv_digits(glv, k, d)) {
      bad++;
      continue;
    }
    if (it >= 40) continue;  // the group-law comparison on a sample
    // evaluate the digit masks exactly as the kernel does
    HXYZZ<Fq> acc = hx_inf<Fq>();
    auto neg = [](const HXYZZ<Fq>& p) {
      HXYZZ<Fq> q = p;
      q.y = h_neg<Fq>(p.y);
      return q;
    };
    for (int bit = (int)d.nd - 1; bit >= 0; bit--) {
      acc = hx_dbl<Fq>(acc);
      const u32 w = bit >> 5, m = 1u << (bit & 31);
      if (d.pos1[w] & m) acc = hx_add<Fq>(acc, Pa);
      if (d.neg1[w] & m) acc = hx_add<Fq>(acc, neg(Pa));
      if (d.pos2[w] & m) acc = hx_add<Fq>(acc, Phi);
      if (d.neg2[w] & m) acc = hx_add<Fq>(acc, neg(Phi));
    }
    u64 got[2 * NQ], want[2 * NQ];
    uint8_t gi, wi;
    hx_to_affine<Fq>(acc, got, &gi);
    hx_to_affine<Fq>(hx_mul<Fq>(Pa, k), want, &wi);
    if (gi != wi || memcmp(got, want, sizeof(got))) bad++;
  }
  printf("%s: mismatches %d, largest half-scalar %d bits\n", name, bad, maxbits);
  return bad != 0 || maxbits > 131;
}

int main() {
  int rc = run<PallasFq, PallasFr>("pallas", 0);
  rc |= run<Bls12381Fq, Bls12381Fr>("bls12_381_g1", 1);
  printf(rc ? "FAIL\n" : "OK\n");
  return rc;
}
