// profile_as -- the reference's accumulation benchmark harness (examples/scaling-as.rs:38-138 `profile_as`, :282-311 `main`)
// on the C++ scheme drivers of include/amsm_*.hpp, i.e. through the C ABI only.
//
// Workload shape, exactly the harness's (:76-104): ONE input; a first accumulation of that input alone; then the TIMED
// accumulation of the same input plus THE SAME ACCUMULATOR TWICE with MakeZK::Enabled; verify; decide; the three
// `serialized_size()` figures (:123-131).  Beside it (--shape n2): 1 input + 1 old accumulator without zk, the lighter shape of
// SURVEY.md section 8(d).  The reference's main() runs trivial_pc_as and ipa_pc_as; hp_as and r1cs_nark_as (BASELINE.json
// configs 4, 5) get the same treatment here.
//
//   profile_as <scheme: trivial_pc_as | ipa_pc_as | hp_as | r1cs_nark_as | all> <log_min> <log_max>
//              [--shape harness|n2|both] [--reps R] [--sponge sha256|poseidon] [--curve 0|1] [--constant] [--uniform] [--no-roundtrip]
//              [--device D | --devices a,b,..] [--seed S] [--dump FILE] [--cold]
//   --cold       no warm-up repetitions (the prove before the timed ones, the decide before the timed one): the CPU legs of
//                bench.py, where a repetition costs seconds and there are no clocks or caches to warm.
//   --devices 0,1,2,3  one context over four GPUs (sharded keys; a repeated id puts two shards on one GPU).
//   --replicate-below L  (with --devices) keys of up to 2^L generators are replicated on every device, batches of MSMs dealt round-robin.
//   --device -1  runs on the library's host backend (amsm.h AMSM_DEVICE_HOST: BASELINE.json config 1 "plumbing, no GPU").
//   --seed S     varies the harness's random stream and the synthetic vectors (0: the bench's inputs).
//   --dump FILE  ONE (scheme, size, shape): exactly one prove after the first accumulation, then FILE receives the serialised new
//                accumulator and proof (ark-serialize layout, include/amsm_serialize.hpp; two records of u64 LE length + bytes) --
//                tests/test_profile_as_dump.py rebuilds the same accumulation on the Python mirrors (the oracle-checked ones)
//                and compares the bytes, at the sizes of BASELINE.json's configs on the GPU box.
//
// Prints the reference's lines ("Indexer:", "Prover:", ...) and one JSON object per (scheme, size, shape) on lines starting
// with "JSON ".  Times are wall-clock milliseconds of the blocking calls (median of R proves after the warm-up accumulation).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "amsm_poseidon.hpp"
#include "amsm_serialize.hpp"

using namespace amsm;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Opt {
  std::string scheme = "all", shape = "both", sponge = "sha256";
  int log_min = 10, log_max = 10, reps = 3, curve = AMSM_PALLAS;
  bool constant = false, roundtrip = true, cold = false;
  bool uniform = false;  // r1cs_nark_as: a circuit whose A z, B z, C z are uniform random vectors (see profile_nark_as)
  int device = 0;
  std::vector<int> devices;  // --devices a,b,..: one multi-device context (sharded keys), amsm.h amsm_ctx_create_multi
  uint64_t seed = 0;
  int replicate_log2 = -1;  // --replicate-below
  std::string dump;
};

static Context make_context(const Opt& o) {
  // --replicate-below L: keys of up to 2^L generators live WHOLE on every device and the independent MSMs of a commit round are
  // dealt to the devices (amsm.h AMSM_BASES_REPLICATE) instead of point-sharding a small key into latency-bound slivers
  if (o.devices.size() >= 2) return Context(o.curve, o.devices, o.replicate_log2 >= 0 ? (size_t)1 << o.replicate_log2 : 0);
  return Context(o.curve, o.device);
}
static void dump_records(const Opt& o, const std::vector<uint8_t>& acc, const std::vector<uint8_t>& proof) {
  if (o.dump.empty()) return;
  FILE* f = fopen(o.dump.c_str(), "wb");
  if (!f) throw Error(AMSM_E_INVALID_ARG, "--dump: cannot open the file");
  for (const std::vector<uint8_t>* r : {&acc, &proof}) {
    uint64_t n = r->size();
    fwrite(&n, 8, 1, f);
    if (n) fwrite(r->data(), 1, n, f);
  }
  fclose(f);
}
// the timed proves: R of them after one more warm-up -- or, with --dump, exactly one (so that the rng stream a mirror has to
// replay is: first accumulation, one prove)
template <class F>
static double timed_proves(const Opt& o, F&& prove_once);

struct HarnessRng {  // ark_std::test_rng() stand-in: a fixed stream
  uint64_t seed, i = 0;
  explicit HarnessRng(uint64_t s) : seed(s) {}
  Fr field() {
    Fr x;
    for (uint64_t k = 0; k < 4; k++) {
      uint64_t z = seed * 0xD1342543DE82EF95ull + (4 * i + k) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      x[k] = z;
    }
    i++;
    x[3] &= (1ull << 62) - 1;
    return x;
  }
};

struct Result {
  double index_ms = 0, prove_ms = 0, verify_ms = 0, decide_ms = 0;
  size_t acc_bytes = 0, inst_bytes = 0, wit_bytes = 0;
  bool verified = false, decided = false, roundtrip = true;
};

static void report(const Opt& o, const char* scheme, int log_size, const char* size_name, const char* shape, bool zk,
                   const Result& r) {
  printf("Indexer: %.0f\nProver: %.0f\nVerifier: %.0f\nDecider: %.0f\n\n", r.index_ms, r.prove_ms, r.verify_ms, r.decide_ms);
  printf("Accumulator size: %zu\nAccumulator instance size: %zu\nAccumulator witness size: %zu\n\n\n\n", r.acc_bytes,
         r.inst_bytes, r.wit_bytes);
  printf("JSON {\"kind\": \"profile_as\", \"scheme\": \"%s\", \"%s\": %d, \"shape\": \"%s\", \"zk\": %s, \"sponge\": \"%s\", "
         "\"index_ms\": %.3f, \"prove_ms\": %.3f, \"verify_ms\": %.3f, \"decide_ms\": %.3f, \"accumulations_per_s\": %.3f, "
         "\"accumulator_bytes\": %zu, \"instance_bytes\": %zu, \"witness_bytes\": %zu, \"verified\": %s, \"decided\": %s, "
         "\"serialize_roundtrip_decides\": %s, \"reps\": %d, \"curve\": %d, \"host_backend\": %s, \"host_pool_threads\": %d}\n",
         scheme, size_name, log_size, shape, zk ? "true" : "false", o.sponge.c_str(), r.index_ms, r.prove_ms, r.verify_ms,
         r.decide_ms, 1000.0 / r.prove_ms, r.acc_bytes, r.inst_bytes, r.wit_bytes, r.verified ? "true" : "false",
         r.decided ? "true" : "false", !o.roundtrip ? "null" : (r.roundtrip ? "true" : "false"), o.reps, o.curve,
         (o.devices.size() < 2 && o.device < 0) ? "true" : "false", amsm_host_threads());
  fflush(stdout);
}

template <class F>
static double median_ms(int reps, F&& f) {
  std::vector<double> t;
  for (int i = 0; i < reps; i++) {
    (void)amsm_ctx_is_host(nullptr);  // a marker for tools/abi_trace.py (a getter without side effects): one timed call starts here
    auto t0 = Clock::now();
    f();
    t.push_back(ms_since(t0));
  }
  (void)amsm_ctx_is_host(nullptr);
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
template <class F>
static double timed_proves(const Opt& o, F&& prove_once) {
  if (!o.dump.empty()) return median_ms(1, prove_once);
  if (!o.cold) prove_once();
  return median_ms(o.reps, prove_once);
}

// ---- hp_as (src/hp_as/mod.rs; inputs like :980-1044 at vector length 2^k) ---------------------------------------------------
template <class Sponge>
static void profile_hp(const Opt& o, int lg, bool harness_shape) {
  using AS = hp_as::ASForHadamardProducts<Sponge>;
  Context ctx = make_context(o);
  hp_as::FrOps fr{o.curve};
  const size_t n = (size_t)1 << lg;
  HarnessRng hr(0xA11CE ^ o.seed);
  hp_as::Rng zk_rng = harness_shape ? hp_as::Rng([&hr]() { return hr.field(); }) : hp_as::Rng();
  Result r;
  auto t0 = Clock::now();
  CommitterKey ck = PedersenCommitment::setup(ctx, n, 0x5EED1001ull, AMSM_BASES_PRECOMPUTE);
  auto keys = AS::index(ck);
  r.index_ms = ms_since(t0);
  auto make_input = [&](uint64_t seed) {
    std::shared_ptr<FrVector> a, b;
    if (o.constant) {  // the reference's inputs: vec![rand; len] (src/hp_as/mod.rs:991-992)
      a = hp_as::filled(ctx, fr.to_mont(hr.field()), n);
      b = hp_as::filled(ctx, fr.to_mont(hr.field()), n);
    } else {
      a = std::make_shared<FrVector>(FrVector::random(ctx, seed, n, true));
      b = std::make_shared<FrVector>(FrVector::random(ctx, seed + 1, n, true));
    }
    FrVector prod = hp_as::compute_hp(*a, *b);
    std::optional<hp_as::InputWitnessRandomness> rnd;
    if (harness_shape) rnd = hp_as::InputWitnessRandomness{fr.to_mont(hr.field()), fr.to_mont(hr.field()), fr.to_mont(hr.field())};
    Affine c1 = PedersenCommitment::commit(ck, *a, rnd ? &rnd->rand_1 : nullptr);
    Affine c2 = PedersenCommitment::commit(ck, *b, rnd ? &rnd->rand_2 : nullptr);
    Affine c3 = PedersenCommitment::commit(ck, prod, rnd ? &rnd->rand_3 : nullptr);
    return hp_as::Accumulator{hp_as::InputInstance{c1, c2, c3}, hp_as::InputWitness{a, b, rnd}};
  };
  std::vector<hp_as::Accumulator> inputs{make_input(100 + o.seed)};
  auto first = AS::prove(*keys.prover_key, inputs, {}, zk_rng);
  std::vector<hp_as::Accumulator> old{first.first};
  if (harness_shape) old.push_back(first.first);
  std::pair<hp_as::Accumulator, hp_as::Proof> res;
  r.prove_ms = timed_proves(o, [&] { res = AS::prove(*keys.prover_key, inputs, old, zk_rng); });
  std::vector<hp_as::InputInstance> ii{inputs[0].instance}, oi;
  for (auto& a : old) oi.push_back(a.instance);
  t0 = Clock::now();
  r.verified = AS::verify(ctx, keys.verifier_key, ii, oi, res.first.instance, res.second);
  r.verify_ms = ms_since(t0);
  if (!o.cold) AS::decide(*keys.decider_key, res.first);
  t0 = Clock::now();
  r.decided = AS::decide(*keys.decider_key, res.first);
  r.decide_ms = ms_since(t0);
  r.acc_bytes = ser::serialized_size(ctx, res.first);
  r.inst_bytes = ser::serialized_size(ctx, res.first.instance);
  r.wit_bytes = ser::serialized_size(ctx, res.first.witness);
  if (o.roundtrip) {
    auto bytes = ser::serialize(ctx, res.first);
    auto back = ser::deserialize<hp_as::Accumulator>(ctx, bytes);
    r.roundtrip = bytes.size() == r.acc_bytes && AS::decide(*keys.decider_key, back) && back.instance == res.first.instance &&
                  ser::serialize(ctx, back) == bytes;
    auto pb = ser::serialize(ctx, res.second);
    r.roundtrip = r.roundtrip && ser::serialize(ctx, ser::deserialize<hp_as::Proof>(ctx, pb)) == pb;
  }
  if (!o.dump.empty()) dump_records(o, ser::serialize(ctx, res.first), ser::serialize(ctx, res.second));
  printf("Vector length: %zu\n", n);
  report(o, "hp_as", lg, "log2_len", harness_shape ? "harness: 1 input + 2x the same accumulator" : "n2: 1 input + 1 accumulator",
         harness_shape, r);
}

// ---- r1cs_nark_as (src/r1cs_nark_as/mod.rs; DummyCircuit :1159-1188 with 2^k constraints, 5 inputs) -------------------------
template <class Sponge>
static void profile_nark_as(const Opt& o, int lg, bool harness_shape) {
  using AS = r1cs_nark_as::ASForR1CSNark<Sponge>;
  using Nark = r1cs_nark::R1CSNark<Sponge>;
  Context ctx = make_context(o);
  hp_as::FrOps fr{o.curve};
  const size_t n_con = (size_t)1 << lg, n_inputs = 5, n_inst = n_inputs + 1;
  const Fr one = {1, 0, 0, 0};
  HarnessRng hr(0xB0B ^ o.seed);
  hp_as::Rng zk_rng = harness_shape ? hp_as::Rng([&hr]() { return hr.field(); }) : hp_as::Rng();
  Result r;
  auto t0 = Clock::now();
  // The reference's DummyCircuit (num_constraints - 1 copies of a * b = c over TWO witness variables) makes A z, B z, C z one value
  // per row: every commitment of the NARK and of the accumulation is a two-valued vector's (amsm_ctx_two_valued_msms).  --uniform
  // runs the same schemes over a circuit whose vectors are uniform: constraint i is w_i * w_i = v_i over 2 (n - 1) witness variables
  // (w random, v their squares), so A z = B z = w and C z = v are random field elements -- what a real circuit's rows look like to
  // the MSM engine.  (Not the reference's harness: reported under its own key by bench.py.)
  const size_t m = n_con - 1, n_wit = o.uniform ? 2 * m : 2;
  std::vector<r1cs_nark::Matrix::Row> A, B, C;
  for (size_t k = 0; k + 1 < n_con; k++) {
    A.push_back({{one, n_inst + (o.uniform ? k : 0)}});
    B.push_back({{one, n_inst + (o.uniform ? k : 1)}});
    C.push_back({{one, o.uniform ? n_inst + m + k : 1}});
  }
  A.push_back({});
  B.push_back({});
  C.push_back({});
  r1cs_nark::IndexProverKey ipk = Nark::index(ctx, A, B, C, n_inst, n_inst + n_wit, 31337);
  auto keys = AS::index(ipk);
  r.index_ms = ms_since(t0);
  uint64_t wit_seed = 0x77A0 + o.seed;
  auto make_input = [&]() {
    Fr a = hr.field(), b = hr.field();
    Fr am = fr.to_mont(a), bm = fr.to_mont(b), abm = fr.mul(am, bm), ab;
    check(amsm_fr_from_mont(o.curve, abm.data(), 1, ab.data()), "from_mont");
    std::vector<Fr> inst{one, ab};
    for (size_t k = 1; k < n_inputs; k++) inst.push_back(a);
    std::shared_ptr<FrVector> wit;
    if (o.uniform) {
      wit = std::make_shared<FrVector>(ctx, n_wit);
      char* w = static_cast<char*>(wit->ptr());
      check(amsm_vec_random(ctx.get(), wit_seed++, m, 1, w), "amsm_vec_random");
      check(amsm_vec_hadamard(ctx.get(), w, w, w + m * 32, m), "amsm_vec_hadamard");
    } else {
      wit = std::make_shared<FrVector>(ctx, std::vector<Fr>{am, bm});
    }
    auto sp = AS::sponges(hp_as::fresh_sponge<Sponge>(o.curve));  // (a sponge over THIS curve's base field before it is forked)
    r1cs_nark::Proof proof = Nark::prove(ipk, inst, wit, zk_rng, sp.nark);
    return r1cs_nark_as::Input{r1cs_nark_as::InputInstance{inst, proof.first_msg}, proof.second_msg};
  };
  std::vector<r1cs_nark_as::Input> inputs{make_input()};
  auto first = AS::prove(keys.pk, inputs, {}, zk_rng);
  std::vector<r1cs_nark_as::Accumulator> old{first.first};
  if (harness_shape) old.push_back(first.first);
  std::pair<r1cs_nark_as::Accumulator, r1cs_nark_as::Proof> res;
  r.prove_ms = timed_proves(o, [&] { res = AS::prove(keys.pk, inputs, old, zk_rng); });
  std::vector<r1cs_nark_as::InputInstance> ii{inputs[0].instance};
  std::vector<r1cs_nark_as::AccumulatorInstance> oi;
  for (auto& a : old) oi.push_back(a.instance);
  t0 = Clock::now();
  r.verified = AS::verify(ctx, keys.vk, ii, oi, res.first.instance, res.second);
  r.verify_ms = ms_since(t0);
  if (!o.cold) AS::decide(*keys.dk, res.first);
  t0 = Clock::now();
  r.decided = AS::decide(*keys.dk, res.first);
  r.decide_ms = ms_since(t0);
  r.acc_bytes = ser::serialized_size(ctx, res.first);
  r.inst_bytes = ser::serialized_size(ctx, res.first.instance);
  r.wit_bytes = ser::serialized_size(ctx, res.first.witness);
  if (o.roundtrip) {
    auto bytes = ser::serialize(ctx, res.first);
    auto back = ser::deserialize<r1cs_nark_as::Accumulator>(ctx, bytes);
    r.roundtrip = bytes.size() == r.acc_bytes && AS::decide(*keys.dk, back) && ser::serialize(ctx, back) == bytes;
    auto pb = ser::serialize(ctx, res.second);
    r.roundtrip = r.roundtrip && ser::serialize(ctx, ser::deserialize<r1cs_nark_as::Proof>(ctx, pb)) == pb;
    auto ib = ser::serialize(ctx, inputs[0]);
    r.roundtrip = r.roundtrip && ser::serialize(ctx, ser::deserialize<r1cs_nark_as::Input>(ctx, ib)) == ib;
  }
  if (!o.dump.empty()) dump_records(o, ser::serialize(ctx, res.first), ser::serialize(ctx, res.second));
  printf("Constraints: %zu\n", n_con);
  report(o, "r1cs_nark_as", lg, "log2_constraints",
         o.uniform ? (harness_shape ? "harness: 1 input + 2x the same accumulator, uniform witness" : "n2: 1 input + 1 accumulator, uniform witness")
                   : (harness_shape ? "harness: 1 input + 2x the same accumulator" : "n2: 1 input + 1 accumulator"),
         harness_shape, r);
}

// ---- the R1CS NARK on its own: examples/scaling-nark.rs:58-110 (profile_nark: index / prove / verify, proof size; main runs it
// with zk off, then on).  DummyCircuit as at :21-56: witness = [a, b, a, ..., a] (num_constraints - 5 + 1 variables), instance
// = [1, a b, a, a, a, a], num_constraints - 1 copies of a * b = c and one empty constraint.  `harness_shape` = make_zk.
template <class Sponge>
static void profile_nark(const Opt& o, int lg, bool make_zk) {
  using Nark = r1cs_nark::R1CSNark<Sponge>;
  Context ctx = make_context(o);
  hp_as::FrOps fr{o.curve};
  const size_t n_con = (size_t)1 << lg, n_inputs = 5, n_inst = n_inputs + 1;
  const size_t n_wit = (n_con > 5 ? n_con - 5 : 1) + 1;  // a, b, then num_witness_variables - 1 copies of a
  const Fr one = {1, 0, 0, 0};
  HarnessRng hr(0xB0B ^ o.seed);
  hp_as::Rng zk_rng = make_zk ? hp_as::Rng([&hr]() { return hr.field(); }) : hp_as::Rng();
  Result r;
  Fr a = hr.field(), b = hr.field();
  Fr am = fr.to_mont(a), bm = fr.to_mont(b), abm = fr.mul(am, bm), ab;
  check(amsm_fr_from_mont(o.curve, abm.data(), 1, ab.data()), "from_mont");
  auto t0 = Clock::now();
  std::vector<r1cs_nark::Matrix::Row> A, B, C;
  for (size_t k = 0; k + 1 < n_con; k++) {
    A.push_back({{one, n_inst + 0}});
    B.push_back({{one, n_inst + 1}});
    C.push_back({{one, 1}});
  }
  A.push_back({});
  B.push_back({});
  C.push_back({});
  r1cs_nark::IndexProverKey ipk = Nark::index(ctx, A, B, C, n_inst, n_inst + n_wit, 31337);
  r.index_ms = ms_since(t0);
  std::vector<Fr> inst{one, ab};
  for (size_t k = 1; k < n_inputs; k++) inst.push_back(a);
  std::vector<Fr> w(n_wit, am);
  w[1] = bm;
  auto wit = std::make_shared<FrVector>(ctx, w);
  r1cs_nark::Proof proof = Nark::prove(ipk, inst, wit, zk_rng, Sponge());
  r.prove_ms = median_ms(o.reps, [&] { proof = Nark::prove(ipk, inst, wit, zk_rng, Sponge()); });
  t0 = Clock::now();
  r.verified = Nark::verify(ipk, inst, proof, Sponge());
  r.verify_ms = ms_since(t0);
  std::vector<Fr> bad = inst;
  bad[1] = a;
  r.decided = !Nark::verify(ipk, bad, proof, Sponge());  // (no decider in a NARK: the slot reports that a wrong input is rejected)
  r.acc_bytes = ser::serialized_size(ctx, proof);
  if (o.roundtrip) {
    auto pb = ser::serialize(ctx, proof);
    r.roundtrip = pb.size() == r.acc_bytes && ser::serialize(ctx, ser::deserialize<r1cs_nark::Proof>(ctx, pb)) == pb;
  }
  printf("(num_constraints, index_time, prover_time, verifier_time):\n(%zu, %.0f, %.0f, %.0f)\nProof size: %zu\n", n_con, r.index_ms,
         r.prove_ms, r.verify_ms, r.acc_bytes);
  report(o, "r1cs_nark", lg, "log2_constraints", make_zk ? "harness: scaling-nark.rs, make_zk" : "n2: scaling-nark.rs, no zk", make_zk, r);
}

// ---- ipa_pc_as (examples/scaling-as.rs:199-280 dl_param_gen / dl_input_gen) -------------------------------------------------
template <class Sponge>
static void profile_ipa(const Opt& o, int lg, bool harness_shape) {
  using AS = ipa_pc_as::AtomicASForInnerProductArgPC<Sponge>;
  using Ipa = ipa_pc::InnerProductArgPC<Sponge>;
  Context ctx = make_context(o);
  ipa_pc::FrX fr(o.curve);
  const size_t degree = ((size_t)1 << lg) - 1;
  HarnessRng hr(0xD1 ^ o.seed);
  hp_as::Rng prng([&hr]() { return hr.field(); });
  hp_as::Rng zk_rng = harness_shape ? prng : hp_as::Rng();
  Result r;
  ipa_pc::CommitterKey pp = Ipa::setup(ctx, degree, 0x1BA5EED);
  auto t0 = Clock::now();
  auto keys = AS::index(pp, degree);
  r.index_ms = ms_since(t0);
  // dl_input_gen: a random polynomial of the supported degree, committed (hiding when zk), opened at a random point
  FrVector poly = FrVector::random(ctx, 77 + o.seed, degree + 1, true);
  auto cr = Ipa::commit(keys.pk.ipa_ck, poly, harness_shape, prng);
  Fr point = fr.to_mont(hr.field());
  FrVector z(ctx, degree + 1);
  check(amsm_vec_powers(ctx.get(), point.data(), degree + 1, z.ptr()), "amsm_vec_powers");
  Fr value;
  check(amsm_vec_inner_product(ctx.get(), poly.ptr(), z.ptr(), degree + 1, value.data()), "amsm_vec_inner_product");
  ipa_pc::Proof proof = Ipa::open(keys.pk.ipa_ck, poly, cr.first, point, cr.second, harness_shape, prng);
  std::vector<ipa_pc_as::InputInstance> inputs{ipa_pc_as::InputInstance{cr.first, point, value, proof}};
  auto first = AS::prove(keys.pk, inputs, {}, zk_rng);
  std::vector<ipa_pc_as::Accumulator> old{first.first};
  if (harness_shape) old.push_back(first.first);
  std::pair<ipa_pc_as::Accumulator, ipa_pc_as::Proof> res;
  r.prove_ms = timed_proves(o, [&] { res = AS::prove(keys.pk, inputs, old, zk_rng); });
  t0 = Clock::now();
  r.verified = AS::verify(ctx, keys.vk, inputs, old, res.first, res.second);
  r.verify_ms = ms_since(t0);
  if (!o.cold) AS::decide(keys.dk, res.first);
  t0 = Clock::now();
  r.decided = AS::decide(keys.dk, res.first);
  r.decide_ms = ms_since(t0);
  r.acc_bytes = r.inst_bytes = ser::serialized_size(ctx, res.first);  // the witness is ()
  r.wit_bytes = 0;
  if (o.roundtrip) {
    auto bytes = ser::serialize(ctx, res.first);
    auto back = ser::deserialize<ipa_pc_as::InputInstance>(ctx, bytes);
    r.roundtrip = bytes.size() == r.acc_bytes && AS::decide(keys.dk, back) && ser::serialize(ctx, back) == bytes;
    auto pb = ser::serialize(ctx, res.second);
    r.roundtrip = r.roundtrip && ser::serialize(ctx, ser::deserialize<ipa_pc_as::Proof>(ctx, pb)) == pb;
  }
  if (!o.dump.empty()) dump_records(o, ser::serialize(ctx, res.first), ser::serialize(ctx, res.second));
  printf("Degree: %zu\n", degree);
  report(o, "ipa_pc_as", lg, "log2_degree_plus_1",
         harness_shape ? "harness: 1 input + 2x the same accumulator" : "n2: 1 input + 1 accumulator", harness_shape, r);
}

// ---- trivial_pc_as (examples/scaling-as.rs:145-197 lh_param_gen / lh_input_gen; the scheme has no zk mode) -------------------
template <class Sponge>
static void profile_trivial(const Opt& o, int lg, bool harness_shape) {
  using namespace trivial_pc_as;
  using AS = ASForTrivialPC<Sponge>;
  Context ctx = make_context(o);
  hp_as::FrOps fr{o.curve};
  const size_t degree = ((size_t)1 << lg) - 1;
  HarnessRng hr(0x7121A1 ^ o.seed);
  Result r;
  CommitterKey pp = TrivialPC::setup(ctx, degree, 0x7121A1);
  CommitterKey ck = TrivialPC::trim(pp, degree);
  auto t0 = Clock::now();
  auto keys = AS::index(pp, degree);
  r.index_ms = ms_since(t0);
  LabeledPolynomial poly;
  for (size_t i = 0; i <= degree; i++) poly.coeffs.push_back(fr.to_mont(hr.field()));
  LabeledCommitment comm = TrivialPC::commit(ck, poly);
  Fr point = fr.to_mont(hr.field());
  std::vector<Input> inputs{Input{InputInstance{comm, point, poly.evaluate(fr, point)}, poly}};
  auto first = AS::prove(keys.prover_key, inputs, {});
  std::vector<Accumulator> old{first.first};
  if (harness_shape) old.push_back(first.first);
  std::pair<Accumulator, Proof> res;
  r.prove_ms = timed_proves(o, [&] { res = AS::prove(keys.prover_key, inputs, old); });
  std::vector<InputInstance> ii{inputs[0].instance}, oi;
  for (auto& a : old) oi.push_back(a.instance);
  t0 = Clock::now();
  r.verified = AS::verify(ctx, keys.verifier_key, ii, oi, res.first.instance, res.second);
  r.verify_ms = ms_since(t0);
  if (!o.cold) AS::decide(keys.prover_key, res.first);
  t0 = Clock::now();
  r.decided = AS::decide(keys.prover_key, res.first);
  r.decide_ms = ms_since(t0);
  r.acc_bytes = ser::serialized_size(ctx, res.first);
  r.inst_bytes = ser::serialized_size(ctx, res.first.instance);
  r.wit_bytes = ser::serialized_size(ctx, res.first.witness);
  if (o.roundtrip) {
    auto bytes = ser::serialize(ctx, res.first);
    auto back = ser::deserialize<Accumulator>(ctx, bytes);
    r.roundtrip = bytes.size() == r.acc_bytes && AS::decide(keys.prover_key, back) && ser::serialize(ctx, back) == bytes;
    auto pb = ser::serialize(ctx, res.second);
    r.roundtrip = r.roundtrip && ser::serialize(ctx, ser::deserialize<Proof>(ctx, pb)) == pb;
  }
  if (!o.dump.empty()) dump_records(o, ser::serialize(ctx, res.first), ser::serialize(ctx, res.second));
  printf("Degree: %zu\n", degree);
  report(o, "trivial_pc_as", lg, "log2_degree_plus_1",
         harness_shape ? "harness: 1 input + 2x the same accumulator" : "n2: 1 input + 1 accumulator", false, r);
}

template <class Sponge>
static void run_all(const Opt& o) {
  struct S {
    const char* name;
    void (*fn)(const Opt&, int, bool);
  } schemes[] = {{"trivial_pc_as", profile_trivial<Sponge>},
                 {"ipa_pc_as", profile_ipa<Sponge>},
                 {"hp_as", profile_hp<Sponge>},
                 {"r1cs_nark_as", profile_nark_as<Sponge>},
                 {"r1cs_nark", profile_nark<Sponge>}};
  for (auto& s : schemes) {
    if (o.scheme != "all" && o.scheme != s.name) continue;
    printf("\n\n\n================ Benchmarking %s ================\n", s.name);
    for (int lg = o.log_min; lg <= o.log_max; lg++) {
      if (o.shape == "harness" || o.shape == "both") s.fn(o, lg, true);
      if (o.shape == "n2" || o.shape == "both") s.fn(o, lg, false);
    }
  }
}

int main(int argc, char** argv) {
  Opt o;
  if (argc < 4) {
    fprintf(stderr, "usage: %s <scheme|all> <log_min> <log_max> [--shape harness|n2|both] [--reps R] [--sponge sha256|poseidon] "
                    "[--curve 0|1] [--constant] [--uniform] [--no-roundtrip] [--device D | --devices a,b,..] [--seed S] [--dump FILE] [--cold] [--replicate-below L]\n", argv[0]);
    return 2;
  }
  o.scheme = argv[1];
  o.log_min = atoi(argv[2]);
  o.log_max = atoi(argv[3]);
  for (int i = 4; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--shape" && i + 1 < argc) o.shape = argv[++i];
    else if (a == "--reps" && i + 1 < argc) o.reps = std::max(1, atoi(argv[++i]));
    else if (a == "--sponge" && i + 1 < argc) o.sponge = argv[++i];
    else if (a == "--curve" && i + 1 < argc) o.curve = atoi(argv[++i]);
    else if (a == "--constant") o.constant = true;
    else if (a == "--uniform") o.uniform = true;
    else if (a == "--no-roundtrip") o.roundtrip = false;
    else if (a == "--cold") o.cold = true;
    else if (a == "--device" && i + 1 < argc) o.device = atoi(argv[++i]);
    else if (a == "--devices" && i + 1 < argc) {
      for (const char* p = argv[++i]; *p;) {
        o.devices.push_back((int)strtol(p, (char**)&p, 10));
        if (*p == ',') p++;
      }
    }
    else if (a == "--replicate-below" && i + 1 < argc) o.replicate_log2 = atoi(argv[++i]);
    else if (a == "--seed" && i + 1 < argc) o.seed = strtoull(argv[++i], nullptr, 0);
    else if (a == "--dump" && i + 1 < argc) o.dump = argv[++i];
    else {
      fprintf(stderr, "unknown option %s\n", a.c_str());
      return 2;
    }
  }
  if (!o.dump.empty() && (o.scheme == "all" || o.log_min != o.log_max || o.shape == "both")) {
    fprintf(stderr, "--dump takes one scheme, one size and one shape\n");
    return 2;
  }
  try {
    if (o.sponge == "poseidon") run_all<poseidon::PoseidonSponge>(o);
    else run_all<hp_as::Sha256Sponge>(o);
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
