// Poseidon sponge over the curve's BASE field (the reference's `S = PoseidonSponge<ConstraintF<G>>`: every scheme test
// instantiates it, src/hp_as/mod.rs:1047-1055, examples/scaling-as.rs:27,36), host side.  The crate is ark-sponge @ git branch
// `accumulation-experimental` (Cargo.toml:18): NOT in /root/reference and not pinned, so this is a restatement of that
// branch's `poseidon/mod.rs` and `lib.rs` AS RECALLED -- PARITY UNPINNED; every constant below is stated so that it can be
// checked against the crate:
//   * `PoseidonSponge::new()`: rate 2, capacity 1 (state = [rate_0, rate_1, capacity]), alpha = 17, 8 full + 31 partial rounds,
//     mds = [[1,0,1],[1,1,0],[0,1,1]], round constants ark[round][i] = F::rand(&mut ChaChaRng::seed_from_u64(123456789)) in
//     round-major order.  `F::rand` (ark-ff `impl Distribution<Fp> for Standard`): four / six `next_u64()` words taken AS the
//     Montgomery representation, top word masked to the modulus' bit length, resampled while >= modulus.
//     ChaChaRng = ChaCha20, 64-bit block counter from 0, stream 0, key = the 32 seed bytes `seed_from_u64` derives with PCG32.
//   * permutation: for each round: add round constants, S-box (x^alpha on every element in the 4 + 4 outer rounds, on
//     state[0] only in the 31 inner ones), multiply by mds.
//   * duplex sponge: absorbing adds elements into state[0..rate) and permutes when the rate is full; the first squeeze after an
//     absorb permutes first; squeezing reads state[0..rate) and permutes when exhausted; absorbing after a squeeze permutes first.
//   * `squeeze_bits(n)`: ceil(n / (8 * floor(CAPACITY / 8))) native elements, the low floor(CAPACITY / 8) bytes of each
//     (canonical, little-endian), bits little-endian within a byte.  `squeeze_nonnative_field_elements_with_sizes(
//     [Truncated(b); k])`: ONE squeeze_bits(k * b), consecutive b-bit windows, each read little-endian (src/hp_as/mod.rs:233-275).
//   * Absorbable encodings (`to_sponge_field_elements`): native element -> itself; usize -> one element; bool / Option tag ->
//     0 / 1; affine point -> x, y, infinity (ark-ec `ToConstraintField for GroupAffine`; the identity: 0, 1, 1); byte string -> floor(CAPACITY / 8)-byte
//     little-endian chunks, one element each; a Vec -> its items in order, no length; `fork(domain)` = clone + absorb the bytes
//     (domain.len() as u64 little-endian || domain).
#pragma once
#include <memory>

#include "host_serialize.h"

namespace amsm {
namespace host {

// ---- ChaCha20 keystream as rand_chacha::ChaCha20Rng yields it -----------------------------------------------------------
struct ChaCha20Rng {
  uint32_t key[8];
  uint64_t counter = 0;
  uint32_t buf[16];
  int idx = 16;
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  static ChaCha20Rng seed_from_u64(uint64_t state) {  // rand_core::SeedableRng::seed_from_u64 (PCG32 expansion)
    ChaCha20Rng r;
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; i++) {
      state = state * MUL + INC;
      uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      uint32_t rot = (uint32_t)(state >> 59);
      r.key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    return r;
  }
  void block() {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4],      key[5],      key[6],      key[7],      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int i = 0; i < 10; i++) {
      qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
      qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) buf[i] = x[i] + s[i];
    counter++;
    idx = 0;
  }
  uint32_t next_u32() {
    if (idx >= 16) block();
    return buf[idx++];
  }
  uint64_t next_u64() {  // BlockRng::next_u64: low word first
    uint64_t lo = next_u32();
    uint64_t hi = next_u32();
    return lo | (hi << 32);
  }
};

template <class Fq>
struct PoseidonParams {
  static constexpr int RATE = 2, CAPACITY = 1, T = 3, FULL = 8, PARTIAL = 31, ALPHA = 17;
  HFe<Fq> ark[FULL + PARTIAL][T];
  static const PoseidonParams& get() {
    static const PoseidonParams p = [] {
      PoseidonParams q;
      constexpr int N = HFe<Fq>::N;
      const int shave = N * 64 - h_modulus_bits<Fq>();
      ChaCha20Rng rng = ChaCha20Rng::seed_from_u64(123456789ull);
      for (int r = 0; r < FULL + PARTIAL; r++)
        for (int i = 0; i < T; i++) {
          HFe<Fq> v;
          do {  // `Fp::rand`: the sampled integer IS the Montgomery representation
            for (int k = 0; k < N; k++) v.v[k] = rng.next_u64();
            v.v[N - 1] &= ~0ull >> shave;
          } while (h_geq_mod<Fq>(v));
          q.ark[r][i] = v;
        }
      return q;
    }();
    return p;
  }
};

template <class Fq>
struct PoseidonSponge {
  using P = PoseidonParams<Fq>;
  HFe<Fq> state[P::T];
  bool squeezing = false;
  int next_index = 0;  // next absorb / squeeze position inside the rate
  PoseidonSponge() {
    for (auto& s : state) s = h_zero<Fq>();
  }
  static constexpr size_t usable_bytes() { return (size_t)(h_modulus_bits<Fq>() - 1) / 8; }  // floor(CAPACITY / 8)

  static HFe<Fq> pow_alpha(const HFe<Fq>& x) {  // x^17
    HFe<Fq> x2 = h_sqr<Fq>(x), x4 = h_sqr<Fq>(x2), x8 = h_sqr<Fq>(x4), x16 = h_sqr<Fq>(x8);
    return h_mul<Fq>(x16, x);
  }
  void permute() {
    const P& p = P::get();
    for (int r = 0; r < P::FULL + P::PARTIAL; r++) {
      for (int i = 0; i < P::T; i++) state[i] = h_add<Fq>(state[i], p.ark[r][i]);
      const bool full = r < P::FULL / 2 || r >= P::FULL / 2 + P::PARTIAL;
      if (full) {
        for (int i = 0; i < P::T; i++) state[i] = pow_alpha(state[i]);
      } else {
        state[0] = pow_alpha(state[0]);
      }
      // mds = [[1,0,1],[1,1,0],[0,1,1]]: new[i] = sum_j mds[i][j] * state[j]
      HFe<Fq> n0 = h_add<Fq>(state[0], state[2]), n1 = h_add<Fq>(state[0], state[1]), n2 = h_add<Fq>(state[1], state[2]);
      state[0] = n0;
      state[1] = n1;
      state[2] = n2;
    }
  }
  void absorb(const HFe<Fq>* elems, size_t n) {
    if (n == 0) return;
    int start;
    if (squeezing) {
      permute();
      start = 0;
    } else {
      start = next_index;
      if (start == P::RATE) {
        permute();
        start = 0;
      }
    }
    squeezing = false;
    size_t done = 0;
    for (;;) {
      if ((size_t)start + (n - done) <= (size_t)P::RATE) {
        const size_t k = n - done;
        for (size_t i = 0; i < k; i++) state[start + i] = h_add<Fq>(state[start + i], elems[done + i]);
        next_index = start + (int)k;
        return;
      }
      int take = P::RATE - start;
      for (int i = 0; i < take; i++) state[start + i] = h_add<Fq>(state[start + i], elems[done + i]);
      done += take;
      permute();
      start = 0;
    }
  }
  void squeeze(HFe<Fq>* out, size_t n) {
    if (n == 0) return;
    int start;
    if (!squeezing) {
      permute();
      start = 0;
    } else {
      start = next_index;
      if (start == P::RATE) {
        permute();
        start = 0;
      }
    }
    squeezing = true;
    size_t done = 0;
    for (;;) {
      if ((size_t)start + (n - done) <= (size_t)P::RATE) {
        size_t k = n - done;
        for (size_t i = 0; i < k; i++) out[done + i] = state[start + i];
        next_index = start + (int)k;
        return;
      }
      int take = P::RATE - start;
      for (int i = 0; i < take; i++) out[done + i] = state[start + i];
      done += take;
      permute();
      start = 0;
    }
  }
  // elements given / returned as raw Montgomery limbs (the C ABI's format)
  void absorb_words(const u64* e, size_t n) {
    constexpr int N = HFe<Fq>::N;
    std::vector<HFe<Fq>> el(n);
    for (size_t i = 0; i < n; i++) memcpy(el[i].v, e + i * N, 8 * N);
    absorb(el.data(), n);
  }
  void squeeze_words(size_t n, u64* out) {
    constexpr int N = HFe<Fq>::N;
    std::vector<HFe<Fq>> el(n);
    squeeze(el.data(), n);
    for (size_t i = 0; i < n; i++) memcpy(out + i * N, el[i].v, 8 * N);
  }
  void absorb_u64(u64 v) {  // usize / bool / Option tag: one element
    HFe<Fq> c = h_zero<Fq>();
    c.v[0] = v;
    c = h_to_mont<Fq>(c);
    absorb(&c, 1);
  }
  // x, y, infinity per point (ark-ec: ToConstraintField for GroupAffine reads the three fields as they are).  The identity is
  // (0, 1, 1): ark-ec ^0.2.0's short-Weierstrass `GroupAffine::zero()` is `new(zero, ONE, true)` -- every identity the schemes absorb
  // comes from it (`into_affine()` of a zero sum, `InputInstance::zero()`, a deserialised flag); ark-ec 0.4 moved to (0, 0).  As
  // recalled, like the rest of this header; rounds 4-5 absorbed (0, 0, 1).
  void absorb_points(const u64* xy, const uint8_t* inf, size_t n) {
    constexpr int N = HFe<Fq>::N;
    std::vector<HFe<Fq>> el(3 * n);
    for (size_t i = 0; i < n; i++) {
      if (inf && inf[i]) {
        el[3 * i] = h_zero<Fq>();
        el[3 * i + 1] = h_one<Fq>();
        el[3 * i + 2] = h_one<Fq>();
      } else {
        memcpy(el[3 * i].v, xy + i * 2 * N, 8 * N);
        memcpy(el[3 * i + 1].v, xy + i * 2 * N + N, 8 * N);
        el[3 * i + 2] = h_zero<Fq>();
      }
    }
    absorb(el.data(), el.size());
  }
  // bytes -> floor(CAPACITY / 8)-byte little-endian chunks, one element each (`ToConstraintField<F> for [u8]`)
  void absorb_bytes(const uint8_t* b, size_t n) {
    const size_t ub = usable_bytes();
    std::vector<HFe<Fq>> el;
    for (size_t off = 0; off < n; off += ub) {
      HFe<Fq> c = h_zero<Fq>();
      memcpy(c.v, b + off, std::min(ub, n - off));
      el.push_back(h_to_mont<Fq>(c));
    }
    absorb(el.data(), el.size());
  }
  // n_bits bits, little-endian, packed into bytes (bit i of the stream = bit (i % 8) of byte i / 8)
  std::vector<uint8_t> squeeze_bits(size_t n_bits) {
    const size_t ub = usable_bytes();
    const size_t n_el = (n_bits + ub * 8 - 1) / (ub * 8);
    std::vector<HFe<Fq>> el(n_el);
    squeeze(el.data(), n_el);
    std::vector<uint8_t> bytes(n_el * ub);
    for (size_t i = 0; i < n_el; i++) {
      HFe<Fq> c = h_from_mont<Fq>(el[i]);
      memcpy(bytes.data() + i * ub, c.v, ub);
    }
    bytes.resize((n_bits + 7) / 8);
    if (n_bits % 8) bytes.back() &= (uint8_t)((1u << (n_bits % 8)) - 1u);
    return bytes;
  }
};

}  // namespace host
}  // namespace amsm
