// C++17 restatement of the R1CS NARK (reference: src/r1cs_nark_as/r1cs_nark/mod.rs -- matrix_vec_mul :443-462, index
// :78-124, compute_challenge :49-72, prove :127-332, verify :335-419; data structures r1cs_nark/data_structures.rs) above
// the C ABI of include/amsm.h: every SpMV, vector loop and Pedersen commitment on the GPU, the O(1) point arithmetic of
// the verifier on the host.  SURVEY.md section 8(a) row a7.  Constraint synthesis (ark-relations) is outside this path:
// callers hand over the R1CS matrices and the assignment (input = instance variables incl. the leading one, witness).
// Same structure, sponge and matrix hash as accumulation_amd/r1cs_nark.py; tests compare the two byte for byte.
#pragma once
#include "amsm_hp_as.hpp"

namespace amsm {
namespace r1cs_nark {

using hp_as::FrOps;
using hp_as::Sha256Sponge;

constexpr unsigned CHALLENGE_SIZE = 128;
inline const char* protocol_name() { return "R1CS-NARK-2020"; }  // :27

// ---- BLAKE2b-256 (RFC 7693), unkeyed: the digest of the matrices (:422-440) -----------------------------------
class Blake2b256 {
 public:
  Blake2b256() {
    for (int i = 0; i < 8; i++) h_[i] = iv(i);
    h_[0] ^= 0x01010000ull ^ 32ull;  // digest length 32, no key, fanout 1, depth 1
  }
  void update(const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) {
      if (fill_ == 128) {
        t_ += 128;
        compress(false);
        fill_ = 0;
      }
      buf_[fill_++] = p[i];
    }
  }
  void update(const std::vector<uint8_t>& v) { update(v.data(), v.size()); }
  std::array<uint8_t, 32> finish() {
    t_ += fill_;
    while (fill_ < 128) buf_[fill_++] = 0;
    compress(true);
    std::array<uint8_t, 32> out;
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(h_[i / 8] >> (8 * (i % 8)));
    return out;
  }

 private:
  static uint64_t iv(int i) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    return IV[i];
  }
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(bool last) {
    static const uint8_t S[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) {
      m[i] = 0;
      for (int j = 0; j < 8; j++) m[i] |= (uint64_t)buf_[8 * i + j] << (8 * j);
    }
    for (int i = 0; i < 8; i++) {
      v[i] = h_[i];
      v[i + 8] = iv(i);
    }
    v[12] ^= t_;
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
      const uint8_t* s = S[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]);
      G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      G(2, 6, 10, 14, m[s[4]], m[s[5]]);
      G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]);
      G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      G(2, 7, 8, 13, m[s[12]], m[s[13]]);
      G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[i + 8];
  }
  uint64_t h_[8];
  uint64_t t_ = 0;
  uint8_t buf_[128];
  size_t fill_ = 0;
};

// ---- Matrix: `Vec<Vec<(F, usize)>>` resident in HBM as CSR --------------------------------------------------------
class Matrix {
 public:
  using Row = std::vector<std::pair<Fr, size_t>>;  // (coefficient, canonical limbs; column index)
  Matrix(Context& ctx, std::vector<Row> rows) : ctx_(&ctx), rows_(std::move(rows)) {
    FrOps fr{amsm_ctx_curve(ctx.get())};
    std::vector<uint32_t> row_ptr{0}, col;
    std::vector<uint64_t> val;
    for (auto& row : rows_) {
      for (auto& e : row) {
        Fr m = fr.to_mont(e.first);
        val.insert(val.end(), m.begin(), m.end());
        col.push_back((uint32_t)e.second);
      }
      row_ptr.push_back((uint32_t)col.size());
    }
    check(amsm_matrix_load(ctx.get(), row_ptr.data(), col.empty() ? nullptr : col.data(), val.empty() ? nullptr : val.data(),
                           rows_.size(), col.size(), &h_),
          "amsm_matrix_load");
  }
  ~Matrix() { amsm_matrix_free(h_); }
  Matrix(const Matrix&) = delete;
  size_t n_rows() const { return rows_.size(); }
  // matrix_vec_mul(matrix, input, witness) :443-462
  FrVector vec_mul(const FrVector& input, const FrVector& witness) const {
    FrVector out(*ctx_, rows_.size());
    check(amsm_matrix_vec_mul(ctx_->get(), h_, input.ptr(), input.len(), witness.ptr(), witness.len(), out.ptr()),
          "amsm_matrix_vec_mul");
    return out;
  }
  void serialize_into(Blake2b256& h) const {  // the same canonical serialisation as accumulation_amd/r1cs_nark.py
    auto u64 = [&](uint64_t x) {
      uint8_t b[8];
      for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
      h.update(b, 8);
    };
    u64(rows_.size());
    for (auto& row : rows_) {
      u64(row.size());
      for (auto& e : row) {
        for (uint64_t w : e.first) u64(w);
        u64(e.second);
      }
    }
  }

 private:
  Context* ctx_;
  std::vector<Row> rows_;
  amsm_matrix* h_ = nullptr;
};

inline std::array<uint8_t, 32> hash_matrices(const char* domain, const Matrix& a, const Matrix& b, const Matrix& c) {  // :422-440
  Blake2b256 h;
  h.update(reinterpret_cast<const uint8_t*>(domain), strlen(domain));
  a.serialize_into(h);
  b.serialize_into(h);
  c.serialize_into(h);
  return h.finish();
}

// ---- data structures (r1cs_nark/data_structures.rs) ------------------------------------------------------------
struct IndexInfo {  // :14-30
  size_t num_variables, num_constraints, num_instance_variables;
  std::array<uint8_t, 32> matrices_hash;
};
struct IndexProverKey {  // :33-48; IndexVerifierKey is the same type (:51)
  IndexInfo index_info;
  std::unique_ptr<Matrix> a, b, c;
  std::unique_ptr<CommitterKey> ck;
};
struct FirstRoundMessageRandomness {
  Affine comm_r_a, comm_r_b, comm_r_c, comm_1, comm_2;
};
struct FirstRoundMessage {  // :101-113
  Affine comm_a, comm_b, comm_c;
  std::optional<FirstRoundMessageRandomness> randomness;
  template <class S>
  void absorb_into(S& sponge) const {
    sponge.absorb_point(comm_a);
    sponge.absorb_point(comm_b);
    sponge.absorb_point(comm_c);
    if (!randomness) {
      sponge.absorb_bytes({0});
    } else {
      sponge.absorb_bytes({1});
      for (const Affine* p : {&randomness->comm_r_a, &randomness->comm_r_b, &randomness->comm_r_c, &randomness->comm_1,
                              &randomness->comm_2})
        sponge.absorb_point(*p);
    }
  }
};
struct SecondRoundMessageRandomness {  // Montgomery form
  Fr sigma_a, sigma_b, sigma_c, sigma_o;
};
struct SecondRoundMessage {  // :171-177
  std::shared_ptr<FrVector> blinded_witness;
  std::optional<SecondRoundMessageRandomness> randomness;
};
struct Proof {
  FirstRoundMessage first_msg;
  SecondRoundMessage second_msg;
};

template <class Sponge = Sha256Sponge>
struct R1CSNark {
  // index (:78-124): matrices + a Pedersen key with num_constraints generators
  static IndexProverKey index(Context& ctx, std::vector<Matrix::Row> a, std::vector<Matrix::Row> b, std::vector<Matrix::Row> c,
                              size_t num_instance_variables, size_t num_variables, uint64_t key_seed = 0x5EED1001ull) {
    IndexProverKey k;
    k.a = std::make_unique<Matrix>(ctx, std::move(a));
    k.b = std::make_unique<Matrix>(ctx, std::move(b));
    k.c = std::make_unique<Matrix>(ctx, std::move(c));
    size_t n_con = k.a->n_rows();
    k.ck = std::make_unique<CommitterKey>(PedersenCommitment::setup(ctx, n_con, key_seed));
    k.index_info = IndexInfo{num_variables, n_con, num_instance_variables, hash_matrices(protocol_name(), *k.a, *k.b, *k.c)};
    return k;
  }

  // compute_challenge (:49-72).  input: canonical scalars.  Returns gamma in Montgomery form.
  static Fr compute_challenge(const FrOps& fr, const std::array<uint8_t, 32>& matrices_hash, const std::vector<Fr>& input,
                              const FirstRoundMessage& msg, Sponge& sponge) {
    sponge.absorb_bytes(std::vector<uint8_t>(matrices_hash.begin(), matrices_hash.end()));
    std::vector<uint8_t> b;
    for (auto& x : input)
      for (uint64_t w : x)
        for (int i = 0; i < 8; i++) b.push_back((uint8_t)(w >> (8 * i)));
    sponge.absorb_bytes(b);
    msg.absorb_into(sponge);
    return fr.to_mont(sponge.squeeze_bits(CHALLENGE_SIZE));
  }

  // prove (:127-332).  input: instance assignment, canonical, incl. the leading one; witness: device vector (Montgomery).
  // rng: empty = no zk, else returns canonical scalars.
  static Proof prove(const IndexProverKey& ipk, const std::vector<Fr>& input, std::shared_ptr<FrVector> witness,
                     const hp_as::Rng& rng = hp_as::Rng(), Sponge sponge = Sponge()) {
    const CommitterKey& ck = *ipk.ck;
    Context& ctx = ck.ctx();
    FrOps fr{amsm_ctx_curve(ctx.get())};
    hp_as::sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    const bool make_zk = (bool)rng;
    std::vector<Fr> in_m;
    for (auto& x : input) in_m.push_back(fr.to_mont(x));
    FrVector d_input(ctx, in_m);
    FrVector z_a = ipk.a->vec_mul(d_input, *witness), z_b = ipk.b->vec_mul(d_input, *witness), z_c = ipk.c->vec_mul(d_input, *witness);
    if (!make_zk) {
      auto c = MsmBatch::same_bases(ck, {&z_a, &z_b, &z_c});
      FirstRoundMessage first{c[0], c[1], c[2], {}};
      compute_challenge(fr, ipk.index_info.matrices_hash, input, first, sponge);  // gamma is squeezed but unused
      return Proof{first, SecondRoundMessage{witness, {}}};
    }
    // :168-172 -- drawn in canonical form and taken to Montgomery form on the device (one Montgomery product with R^2 per element,
    // one kernel: a witness-length loop of single-element amsm_fr_to_mont calls was ~25 ns per variable on the host)
    std::vector<Fr> r_canon(witness->len());
    for (size_t i = 0; i < r_canon.size(); i++) r_canon[i] = rng();
    FrVector r_raw(ctx, r_canon);
    FrVector r = hp_as::combine_vectors(ctx, {&r_raw}, {fr.to_mont(fr.one())});
    FrVector zeros(ctx, std::vector<Fr>(input.size(), fr.zero()));
    FrVector r_a = ipk.a->vec_mul(zeros, r), r_b = ipk.b->vec_mul(zeros, r), r_c = ipk.c->vec_mul(zeros, r);
    Fr a_bl = fr.to_mont(rng()), b_bl = fr.to_mont(rng()), c_bl = fr.to_mont(rng());
    Fr ra_bl = fr.to_mont(rng()), rb_bl = fr.to_mont(rng()), rc_bl = fr.to_mont(rng());
    FrVector x1 = hp_as::compute_hp(z_a, r_b), x2 = hp_as::compute_hp(z_b, r_a);
    FrVector cross = hp_as::combine_vectors(ctx, {&x1, &x2}, {fr.one(), fr.one()});
    Fr bl1 = fr.to_mont(rng());
    Fr bl2 = fr.to_mont(rng());
    FrVector rr = hp_as::compute_hp(r_a, r_b);
    // eight independent commitments: one pipelined MSM batch (same points as eight commit() calls, :216-261)
    auto cm = PedersenCommitment::commit_batch(ck, {&z_a, &z_b, &z_c, &r_a, &r_b, &r_c, &cross, &rr},
                                               {&a_bl, &b_bl, &c_bl, &ra_bl, &rb_bl, &rc_bl, &bl1, &bl2});
    const Affine &comm_a = cm[0], &comm_b = cm[1], &comm_c = cm[2], &comm_r_a = cm[3], &comm_r_b = cm[4], &comm_r_c = cm[5],
                 &comm_1 = cm[6], &comm_2 = cm[7];
    FirstRoundMessage first{comm_a, comm_b, comm_c, FirstRoundMessageRandomness{comm_r_a, comm_r_b, comm_r_c, comm_1, comm_2}};
    Fr gamma = compute_challenge(fr, ipk.index_info.matrices_hash, input, first, sponge);
    auto blinded = std::make_shared<FrVector>(hp_as::combine_vectors(ctx, {witness.get(), &r}, {fr.one(), gamma}));  // w + gamma r
    Fr g2 = fr.mul(gamma, gamma);
    SecondRoundMessageRandomness rnd{fr.add(a_bl, fr.mul(gamma, ra_bl)), fr.add(b_bl, fr.mul(gamma, rb_bl)),
                                     fr.add(c_bl, fr.mul(gamma, rc_bl)),
                                     fr.add(c_bl, fr.add(fr.mul(gamma, bl1), fr.mul(g2, bl2)))};
    return Proof{first, SecondRoundMessage{blinded, rnd}};
  }

  // verify (:335-419): 3 SpMV + 4 commitments on the GPU, O(1) point arithmetic on the host
  static bool verify(const IndexProverKey& ivk, const std::vector<Fr>& input, const Proof& proof, Sponge sponge = Sponge()) {
    const CommitterKey& ck = *ivk.ck;
    Context& ctx = ck.ctx();
    FrOps fr{amsm_ctx_curve(ctx.get())};
    hp_as::sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    const FirstRoundMessage& first = proof.first_msg;
    const SecondRoundMessage& second = proof.second_msg;
    if (first.randomness.has_value() != second.randomness.has_value()) return false;
    Fr gamma = compute_challenge(fr, ivk.index_info.matrices_hash, input, first, sponge);
    std::vector<Fr> in_m;
    for (auto& x : input) in_m.push_back(fr.to_mont(x));
    FrVector d_input(ctx, in_m);
    FrVector za = ivk.a->vec_mul(d_input, *second.blinded_witness), zb = ivk.b->vec_mul(d_input, *second.blinded_witness),
             zc = ivk.c->vec_mul(d_input, *second.blinded_witness);
    FrVector zab = hp_as::compute_hp(za, zb);
    const SecondRoundMessageRandomness* rnd = second.randomness ? &*second.randomness : nullptr;
    std::vector<Affine> lhs = PedersenCommitment::commit_batch(  // four independent commitments, one batch (:375-403)
        ck, {&za, &zb, &zc, &zab},
        {rnd ? &rnd->sigma_a : nullptr, rnd ? &rnd->sigma_b : nullptr, rnd ? &rnd->sigma_c : nullptr, rnd ? &rnd->sigma_o : nullptr});
    Affine rhs[4];
    if (!rnd) {
      rhs[0] = first.comm_a;
      rhs[1] = first.comm_b;
      rhs[2] = first.comm_c;
      rhs[3] = first.comm_c;
    } else {
      const FirstRoundMessageRandomness& f = *first.randomness;
      Fr one = fr.one(), g2 = fr.mul(gamma, gamma);
      rhs[0] = lincomb(ctx, {&first.comm_a, &f.comm_r_a}, {one, gamma});
      rhs[1] = lincomb(ctx, {&first.comm_b, &f.comm_r_b}, {one, gamma});
      rhs[2] = lincomb(ctx, {&first.comm_c, &f.comm_r_c}, {one, gamma});
      rhs[3] = lincomb(ctx, {&first.comm_c, &f.comm_1, &f.comm_2}, {one, gamma, g2});
    }
    for (int i = 0; i < 4; i++)
      if (!(lhs[i] == rhs[i])) return false;
    return true;
  }

 private:
  static Affine lincomb(Context& ctx, const std::vector<const Affine*>& points, const std::vector<Fr>& scalars) {
    size_t k = points.size(), w = 2 * (size_t)ctx.fq_limbs();
    Affine out;
    out.xy.assign(w, 0);
    std::vector<uint64_t> xy(k * w);
    std::vector<uint8_t> inf(k);
    for (size_t i = 0; i < k; i++) {
      std::copy(points[i]->xy.begin(), points[i]->xy.end(), xy.begin() + (long)(i * w));
      inf[i] = points[i]->infinity ? 1 : 0;
    }
    uint8_t oinf = 0;
    check(amsm_host_lincomb(amsm_ctx_curve(ctx.get()), xy.data(), inf.data(), reinterpret_cast<const uint64_t*>(scalars.data()), k,
                            out.xy.data(), &oinf),
          "amsm_host_lincomb");
    out.infinity = oinf != 0;
    if (out.infinity) std::fill(out.xy.begin(), out.xy.end(), 0);
    return out;
  }
};

}  // namespace r1cs_nark
}  // namespace amsm
