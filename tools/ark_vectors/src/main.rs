//! Emits golden vectors from the REAL arkworks stack (ark-ec / ark-ff / ark-serialize 0.2, ark-sponge and ark-poly-commit on
//! their `accumulation-experimental` branches, the reference crate itself) in the JSON layout tests/test_ark_vectors_*.py read.
//! NOT COMPILED in the build image (no Rust toolchain there) -- see Cargo.toml.  One file per vector kind:
//!
//!   ark_msm.json        VariableBaseMSM::multi_scalar_mul on both curves: seeded cases in the synthetic stream of
//!                       accumulation_amd/csrc/rng.h (scalars: uniform in [0, r) by rejection of 255-bit candidates, `rng_frs`;
//!                       points: 254-bit multiplier * generator)
//!                       and explicit edge cases (0, 1, r - 1, duplicate and opposite bases, the identity)
//!   ark_serialize.json  CanonicalSerialize of scalars and points (compressed and uncompressed), both curves
//!   ark_poseidon.json   PoseidonSponge::<Fq>::new(): absorb / squeeze transcripts over every Absorbable the schemes use
//!   ark_pedersen.json   PedersenCommitment::{setup, trim, commit} incl. hiding: the key is written out with the results
//!   ark_hp_as.json      ASForHadamardProducts::prove without zk (deterministic): key, inputs, serialized accumulator + proof
//!
//! All integers are lower-case hex strings of the canonical value ("0x..."), points are [x, y] or null for the identity,
//! byte strings are hex without prefix.
use ark_ec::msm::VariableBaseMSM;
use ark_ec::{AffineCurve, ProjectiveCurve};
use ark_ff::{BigInteger, PrimeField, Zero};
use ark_serialize::CanonicalSerialize;
use std::fmt::Write as _;
use std::fs;
use std::path::Path;

// ---- the synthetic stream (accumulation_amd/csrc/rng.h == oracle/pyref.py rng_word / rng_scalar) ----
fn mix64(mut z: u64) -> u64 {
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}
fn rng_word(seed: u64, j: u64) -> u64 {
    mix64(
        seed.wrapping_mul(0xD1342543DE82EF95)
            .wrapping_add(j.wrapping_mul(0x9E3779B97F4A7C15))
            .wrapping_add(0x632BE59BD9B4E019),
    )
}
fn rng_scalar_words(seed: u64, i: u64) -> [u64; 4] {
    let mut w = [0u64; 4];
    for k in 0..4 {
        w[k] = rng_word(seed, 4 * i + k as u64);
    }
    w[3] &= (1u64 << 62) - 1;
    w
}

/// candidate `t` of scalar `i` of the SCALAR stream (rng.h: rng_scalar_fr; oracle/pyref.py: rng_fr): 255 bits
fn rng_fr_words(seed: u64, i: u64, t: u64) -> [u64; 4] {
    let mut w = [0u64; 4];
    for k in 0..4 {
        w[k] = rng_word(seed, (t << 40) + 4 * i + k as u64);
    }
    w[3] &= (1u64 << 63) - 1;
    w
}

fn hex_be(bytes_be: &[u8]) -> String {
    let mut s = String::from("0x");
    let mut started = false;
    for b in bytes_be {
        if !started && *b == 0 {
            continue;
        }
        if !started {
            write!(s, "{:x}", b).unwrap();
            started = true;
        } else {
            write!(s, "{:02x}", b).unwrap();
        }
    }
    if !started {
        s.push('0');
    }
    s
}
fn hex_field<F: PrimeField>(x: &F) -> String {
    hex_be(&x.into_repr().to_bytes_be())
}
fn hex_bytes(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}
fn json_point<G: AffineCurve>(p: &G, xy: impl Fn(&G) -> Option<(String, String)>) -> String {
    match xy(p) {
        None => "null".to_string(),
        Some((x, y)) => format!("[\"{}\", \"{}\"]", x, y),
    }
}

macro_rules! curve_impl {
    ($modname:ident, $name:expr, $affine:ty, $fr:ty, $fq:ty, $bigint:ty) => {
        mod $modname {
            use super::*;
            pub type G = $affine;
            pub type Fr = $fr;
            pub type Fq = $fq;
            pub fn xy(p: &G) -> Option<(String, String)> {
                if p.is_zero() {
                    None
                } else {
                    Some((hex_field(&p.x), hex_field(&p.y)))
                }
            }
            pub fn scalar_from_words(w: [u64; 4]) -> Fr {
                // the multiplier stream is 254 bits wide: below r on both curves (Pallas r = 2^254 + ..., BLS12-381 r = 0x73ed... x 2^240),
                // so the reduction below never changes a value
                let mut bytes = [0u8; 32];
                for k in 0..4 {
                    bytes[8 * k..8 * k + 8].copy_from_slice(&w[k].to_le_bytes());
                }
                Fr::from_le_bytes_mod_order(&bytes)
            }
            pub fn rng_scalars(seed: u64, n: usize) -> Vec<Fr> {
                (0..n).map(|i| scalar_from_words(rng_scalar_words(seed, i as u64))).collect()
            }
            /// the scalar stream of amsm_vec_random: uniform in [0, r) -- the first 255-bit candidate below r (ark-ff 0.2's
            /// `from_repr` returns None for a value that is not below the modulus); after 64 rejections candidate 63 with bit
            /// 254 cleared (never reached in practice: probability < 2^-64)
            pub fn rng_frs(seed: u64, n: usize) -> Vec<Fr> {
                (0..n as u64)
                    .map(|i| {
                        for t in 0..64u64 {
                            if let Some(x) = Fr::from_repr(<$bigint>::new(rng_fr_words(seed, i, t))) {
                                return x;
                            }
                        }
                        let mut w = rng_fr_words(seed, i, 63);
                        w[3] &= (1u64 << 62) - 1;
                        scalar_from_words(w)
                    })
                    .collect()
            }
            pub fn rng_points(seed: u64, n: usize) -> Vec<G> {
                let g = G::prime_subgroup_generator();
                let proj: Vec<_> = rng_scalars(seed, n).iter().map(|k| g.mul(k.into_repr())).collect();
                <G as AffineCurve>::Projective::batch_normalization_into_affine(&proj)
            }
            pub fn msm(bases: &[G], scalars: &[Fr]) -> G {
                let reprs: Vec<$bigint> = scalars.iter().map(|s| s.into_repr()).collect();
                VariableBaseMSM::multi_scalar_mul(bases, &reprs).into_affine()
            }
            pub fn msm_cases() -> String {
                let mut cases: Vec<String> = Vec::new();
                for (i, n) in [1usize, 2, 31, 32, 33, 255, 1000, 4096, 65536].iter().enumerate() {
                    let (sp, ss) = (0x5EED_A000u64 + i as u64, 0x5EED_B000u64 + i as u64);
                    let r = msm(&rng_points(sp, *n), &rng_frs(ss, *n));
                    cases.push(format!(
                        "{{\"kind\": \"seeded\", \"n\": {}, \"seed_points\": {}, \"seed_scalars\": {}, \"expected\": {}}}",
                        n, sp, ss, json_point(&r, xy)
                    ));
                }
                // explicit edge cases over eight generators
                let pts = rng_points(0x5EED_C000, 8);
                let one = Fr::from(1u64);
                let rm1 = -one;
                let edge: Vec<(&str, Vec<G>, Vec<Fr>)> = vec![
                    ("all_zero_scalars", pts.clone(), vec![Fr::zero(); 8]),
                    ("all_one_scalars", pts.clone(), vec![one; 8]),
                    ("all_r_minus_1", pts.clone(), vec![rm1; 8]),
                    ("duplicate_bases", vec![pts[0]; 8], rng_scalars(0x5EED_C001, 8)),
                    ("opposite_bases_cancel", vec![pts[1], -pts[1]], vec![Fr::from(7u64), Fr::from(7u64)]),
                    ("identity_among_bases", vec![pts[2], G::zero(), pts[3]], rng_scalars(0x5EED_C002, 3)),
                    ("more_bases_than_scalars", pts.clone(), rng_scalars(0x5EED_C003, 5)),
                    ("powers_of_two", pts.clone(), (0..8u64).map(|k| Fr::from(2u64).pow([31 * k + 1])).collect()),
                ];
                for (name, b, s) in edge {
                    let n = b.len().min(s.len());
                    let r = msm(&b[..n], &s[..n]);
                    let bp: Vec<String> = b.iter().map(|p| json_point(p, xy)).collect();
                    let sp: Vec<String> = s.iter().map(|x| format!("\"{}\"", hex_field(x))).collect();
                    cases.push(format!(
                        "{{\"kind\": \"explicit\", \"name\": \"{}\", \"points\": [{}], \"scalars\": [{}], \"expected\": {}}}",
                        name, bp.join(", "), sp.join(", "), json_point(&r, xy)
                    ));
                }
                format!("\"{}\": [\n    {}\n  ]", $name, cases.join(",\n    "))
            }
            pub fn serialize_cases() -> String {
                let mut rows: Vec<String> = Vec::new();
                let scalars = vec![Fr::zero(), Fr::from(1u64), -Fr::from(1u64), rng_scalars(0x5EED_D000, 1)[0]];
                for s in &scalars {
                    let mut b = Vec::new();
                    s.serialize(&mut b).unwrap();
                    rows.push(format!("{{\"type\": \"fr\", \"value\": \"{}\", \"bytes\": \"{}\"}}", hex_field(s), hex_bytes(&b)));
                }
                let g = G::prime_subgroup_generator();
                let mut pts = vec![g, -g, G::zero()];
                pts.extend(rng_points(0x5EED_D001, 4));
                for p in &pts {
                    let (mut c, mut u) = (Vec::new(), Vec::new());
                    p.serialize(&mut c).unwrap();
                    p.serialize_uncompressed(&mut u).unwrap();
                    rows.push(format!(
                        "{{\"type\": \"point\", \"value\": {}, \"compressed\": \"{}\", \"uncompressed\": \"{}\"}}",
                        json_point(p, xy), hex_bytes(&c), hex_bytes(&u)
                    ));
                }
                // Vec<Fr> and Option<Fr>: the length prefix and the tag byte
                let v = rng_scalars(0x5EED_D002, 3);
                let mut b = Vec::new();
                v.serialize(&mut b).unwrap();
                rows.push(format!(
                    "{{\"type\": \"vec_fr\", \"values\": [{}], \"bytes\": \"{}\"}}",
                    v.iter().map(|x| format!("\"{}\"", hex_field(x))).collect::<Vec<_>>().join(", "), hex_bytes(&b)
                ));
                for o in [None, Some(v[0])].iter() {
                    let mut b = Vec::new();
                    o.serialize(&mut b).unwrap();
                    rows.push(format!(
                        "{{\"type\": \"option_fr\", \"value\": {}, \"bytes\": \"{}\"}}",
                        o.map(|x| format!("\"{}\"", hex_field(&x))).unwrap_or("null".to_string()), hex_bytes(&b)
                    ));
                }
                format!("\"{}\": [\n    {}\n  ]", $name, rows.join(",\n    "))
            }
        }
    };
}
curve_impl!(pallas, "pallas", ark_pallas::Affine, ark_pallas::Fr, ark_pallas::Fq, ark_ff::BigInteger256);
curve_impl!(bls, "bls12_381_g1", ark_bls12_381::G1Affine, ark_bls12_381::Fr, ark_bls12_381::Fq, ark_ff::BigInteger256);

// ---- Poseidon transcripts: every step is written with its outputs, the consumer replays the absorbs and compares the squeezes ----
fn poseidon_transcripts() -> String {
    use ark_pallas::{Affine as G, Fq, Fr};
    use ark_sponge::poseidon::PoseidonSponge;
    use ark_sponge::{CryptographicSponge, FieldElementSize};
    let mut out: Vec<String> = Vec::new();
    let fq = |x: &Fq| format!("\"{}\"", hex_field(x));
    // 1. squeeze from the fresh sponge, absorb field elements, squeeze again (duplex mode switches)
    {
        let mut s = PoseidonSponge::<Fq>::new();
        let mut steps: Vec<String> = Vec::new();
        let a = s.squeeze_field_elements(3);
        steps.push(format!("{{\"squeeze_fq\": [{}]}}", a.iter().map(fq).collect::<Vec<_>>().join(", ")));
        let inp: Vec<Fq> = (1..=5u64).map(Fq::from).collect();
        s.absorb(&inp);
        steps.push(format!("{{\"absorb_fq\": [{}]}}", inp.iter().map(fq).collect::<Vec<_>>().join(", ")));
        let b = s.squeeze_field_elements(4);
        steps.push(format!("{{\"squeeze_fq\": [{}]}}", b.iter().map(fq).collect::<Vec<_>>().join(", ")));
        let bits = s.squeeze_bits(300);
        steps.push(format!("{{\"squeeze_bits\": \"{}\"}}", bits.iter().map(|b| if *b { '1' } else { '0' }).collect::<String>()));
        out.push(format!("{{\"name\": \"native\", \"steps\": [{}]}}", steps.join(", ")));
    }
    // 2. bytes, usize, a point, then the truncated non-native challenges the schemes draw (src/hp_as/mod.rs:233-262)
    {
        let mut s = PoseidonSponge::<Fq>::new();
        let mut steps: Vec<String> = Vec::new();
        let bytes: Vec<u8> = (0u8..77).collect();
        s.absorb(&bytes);
        steps.push(format!("{{\"absorb_bytes\": \"{}\"}}", hex_bytes(&bytes)));
        s.absorb(&11usize);
        steps.push("{\"absorb_usize\": 11}".to_string());
        let p = pallas::rng_points(0x5EED_E000, 1)[0];
        s.absorb(&p);
        steps.push(format!("{{\"absorb_point\": {}}}", json_point(&p, pallas::xy)));
        let sizes = vec![FieldElementSize::Truncated(128); 3];
        let ch: Vec<Fr> = s.squeeze_nonnative_field_elements_with_sizes(sizes.as_slice());
        steps.push(format!(
            "{{\"squeeze_nonnative_truncated_128\": [{}]}}",
            ch.iter().map(|x| format!("\"{}\"", hex_field(x))).collect::<Vec<_>>().join(", ")
        ));
        let one: Vec<Fr> = s.squeeze_nonnative_field_elements_with_sizes(&[FieldElementSize::Truncated(128)]);
        steps.push(format!("{{\"squeeze_nonnative_truncated_128\": [\"{}\"]}}", hex_field(&one[0])));
        let full: Vec<Fr> = s.squeeze_nonnative_field_elements_with_sizes(&[FieldElementSize::Full]);
        steps.push(format!("{{\"squeeze_nonnative_full\": [\"{}\"]}}", hex_field(&full[0])));
        out.push(format!("{{\"name\": \"encodings\", \"steps\": [{}]}}", steps.join(", ")));
    }
    // 3. ONE encoding per case, two native squeezes behind it: a mismatch then names the rule (DESIGN.md section 7 lists what the pin
    //    has to settle: byte strings bare or behind a length, the identity's coordinates, the Option tag, fork)
    {
        let sq = |s: &mut PoseidonSponge<Fq>| -> String {
            let v = s.squeeze_field_elements(2);
            format!("{{\"squeeze_fq\": [{}]}}", v.iter().map(fq).collect::<Vec<_>>().join(", "))
        };
        let bytes: Vec<u8> = (0u8..33).collect();
        let mut s = PoseidonSponge::<Fq>::new();
        s.absorb(&bytes);
        out.push(format!("{{\"name\": \"bytes_only\", \"steps\": [{{\"absorb_bytes\": \"{}\"}}, {}]}}", hex_bytes(&bytes), sq(&mut s)));
        let mut s = PoseidonSponge::<Fq>::new();
        let id = G::zero();
        s.absorb(&id);
        out.push(format!("{{\"name\": \"identity_point\", \"steps\": [{{\"absorb_point\": {}}}, {}]}}", json_point(&id, pallas::xy), sq(&mut s)));
        let mut s = PoseidonSponge::<Fq>::new();
        let some: Option<Vec<u8>> = Some(bytes.clone());
        s.absorb(&some);
        out.push(format!("{{\"name\": \"option_some_bytes\", \"steps\": [{{\"absorb_option_bytes\": \"{}\"}}, {}]}}", hex_bytes(&bytes), sq(&mut s)));
        let mut s = PoseidonSponge::<Fq>::new();
        let none: Option<Vec<u8>> = None;
        s.absorb(&none);
        out.push(format!("{{\"name\": \"option_none\", \"steps\": [{{\"absorb_option_bytes\": null}}, {}]}}", sq(&mut s)));
        let s = PoseidonSponge::<Fq>::new();
        let mut f = s.fork(b"AS-FOR-HP-2020");
        out.push(format!("{{\"name\": \"fork\", \"steps\": [{{\"fork\": \"{}\"}}, {}]}}", hex_bytes(b"AS-FOR-HP-2020"), sq(&mut f)));
    }
    let _ = G::prime_subgroup_generator();
    format!("[\n  {}\n]", out.join(",\n  "))
}

// ---- Pedersen commitments (ark_poly_commit::trivial_pc::PedersenCommitment; call sites src/hp_as/mod.rs:196-214,377) ----
fn pedersen_cases() -> String {
    use ark_pallas::Affine as G;
    use ark_poly_commit::trivial_pc::PedersenCommitment;
    let mut rows: Vec<String> = Vec::new();
    for n in [1usize, 8, 33].iter() {
        let pp = PedersenCommitment::<G>::setup(*n);
        let ck = PedersenCommitment::<G>::trim(&pp, *n);
        let v = pallas::rng_scalars(0x5EED_F000 + *n as u64, *n);
        let r = pallas::rng_scalars(0x5EED_F100, 1)[0];
        let plain: G = PedersenCommitment::<G>::commit(&ck, v.as_slice(), None);
        let hiding: G = PedersenCommitment::<G>::commit(&ck, v.as_slice(), Some(r));
        rows.push(format!(
            "{{\"n\": {}, \"generators\": [{}], \"hiding_generator\": {}, \"elems\": [{}], \"rand\": \"{}\", \"commit\": {}, \"commit_hiding\": {}}}",
            n,
            ck.generators.iter().map(|p| json_point(p, pallas::xy)).collect::<Vec<_>>().join(", "),
            json_point(&ck.hiding_generator, pallas::xy),
            v.iter().map(|x| format!("\"{}\"", hex_field(x))).collect::<Vec<_>>().join(", "),
            hex_field(&r),
            json_point(&plain, pallas::xy),
            json_point(&hiding, pallas::xy)
        ));
    }
    format!("[\n  {}\n]", rows.join(",\n  "))
}

// ---- a whole hp_as accumulation without zk (src/hp_as/mod.rs:646-813): deterministic given the inputs and the sponge ----
fn hp_as_cases() -> String {
    use ark_accumulation::hp_as::{ASForHadamardProducts, InputInstance, InputWitness};
    use ark_accumulation::{AccumulationScheme, Input, MakeZK};
    use ark_pallas::{Affine as G, Fq};
    use ark_poly_commit::trivial_pc::PedersenCommitment;
    use ark_sponge::poseidon::PoseidonSponge;
    type AS = ASForHadamardProducts<G, PoseidonSponge<Fq>>;
    let mut rows: Vec<String> = Vec::new();
    for (n, num_inputs) in [(8usize, 1usize), (11, 2), (64, 3)].iter() {
        let pp = PedersenCommitment::<G>::setup(*n);
        let ck = PedersenCommitment::<G>::trim(&pp, *n);
        let (pk, _vk, dk) = AS::index(&(), &(), n).unwrap();
        let mut inputs = Vec::new();
        let mut inputs_json: Vec<String> = Vec::new();
        for i in 0..*num_inputs {
            let a = pallas::rng_scalars(0x5EED_1A00 + (16 * n + i) as u64, *n);
            let b = pallas::rng_scalars(0x5EED_1B00 + (16 * n + i) as u64, *n);
            let prod: Vec<_> = a.iter().zip(b.iter()).map(|(x, y)| *x * *y).collect();
            let instance = InputInstance {
                comm_1: PedersenCommitment::<G>::commit(&ck, a.as_slice(), None),
                comm_2: PedersenCommitment::<G>::commit(&ck, b.as_slice(), None),
                comm_3: PedersenCommitment::<G>::commit(&ck, prod.as_slice(), None),
            };
            let witness = InputWitness { a_vec: a.clone(), b_vec: b.clone(), randomness: None };
            inputs_json.push(format!(
                "{{\"a\": [{}], \"b\": [{}]}}",
                a.iter().map(|x| format!("\"{}\"", hex_field(x))).collect::<Vec<_>>().join(", "),
                b.iter().map(|x| format!("\"{}\"", hex_field(x))).collect::<Vec<_>>().join(", ")
            ));
            inputs.push(Input::<Fq, PoseidonSponge<Fq>, AS> { instance, witness });
        }
        let (acc, proof) = AS::prove(
            &pk,
            Input::<Fq, PoseidonSponge<Fq>, AS>::map_to_refs(&inputs),
            vec![],
            MakeZK::Disabled,
            None::<PoseidonSponge<Fq>>,
        )
        .unwrap();
        assert!(AS::decide(&dk, acc.as_ref(), None::<PoseidonSponge<Fq>>).unwrap());
        let (mut bi, mut bw, mut bp) = (Vec::new(), Vec::new(), Vec::new());
        acc.instance.serialize(&mut bi).unwrap();
        acc.witness.serialize(&mut bw).unwrap();
        proof.serialize(&mut bp).unwrap();
        rows.push(format!(
            "{{\"n\": {}, \"generators\": [{}], \"hiding_generator\": {}, \"inputs\": [{}], \"accumulator_instance\": \"{}\", \"accumulator_witness\": \"{}\", \"proof\": \"{}\"}}",
            n,
            ck.generators.iter().map(|p| json_point(p, pallas::xy)).collect::<Vec<_>>().join(", "),
            json_point(&ck.hiding_generator, pallas::xy),
            inputs_json.join(", "),
            hex_bytes(&bi), hex_bytes(&bw), hex_bytes(&bp)
        ));
    }
    format!("[\n  {}\n]", rows.join(",\n  "))
}

fn write(dir: &Path, name: &str, body: String) {
    let head = "\"generator\": \"tools/ark_vectors (ark-ec / ark-ff / ark-serialize 0.2, ark-sponge + ark-poly-commit @ accumulation-experimental)\"";
    let text = if body.trim_start().starts_with('[') {
        format!("{{\n{},\n\"cases\": {}\n}}\n", head, body)
    } else {
        format!("{{\n{},\n  {}\n}}\n", head, body)
    };
    fs::write(dir.join(name), text).expect("write");
    eprintln!("wrote {}", dir.join(name).display());
}

fn main() {
    let dir = std::env::args().nth(1).unwrap_or_else(|| "../../tests/golden".to_string());
    let dir = Path::new(&dir);
    fs::create_dir_all(dir).unwrap();
    write(dir, "ark_msm.json", format!("{},\n  {}", pallas::msm_cases(), bls::msm_cases()));
    write(dir, "ark_serialize.json", format!("{},\n  {}", pallas::serialize_cases(), bls::serialize_cases()));
    write(dir, "ark_poseidon.json", poseidon_transcripts());
    write(dir, "ark_pedersen.json", pedersen_cases());
    write(dir, "ark_hp_as.json", hp_as_cases());
}
