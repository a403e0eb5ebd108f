"""TEST INFRASTRUCTURE (oracle/): the Fiat-Shamir TRANSCRIPTS of the four accumulation schemes, restated from the reference's
`absorb!` item lists and `Absorbable` impls -- what is absorbed, in which order, and what is squeezed -- over the big-integer Poseidon
sponge of oracle/pyref_poseidon.py.  oracle/pyref_as.py checks the schemes' ALGEBRA with the product's challenges as inputs; this
file derives those challenges independently from the public data, so that tests can compare the two (tests/test_transcripts_vs_oracle.py).
Only tests/ may import it.

PARITY UNPINNED, twice over: the sponge parameters and the `Absorbable` encodings of ark-sponge @ `accumulation-experimental`
(Cargo.toml:18; not in the reference tree) are restated as recalled (pyref_poseidon.py header); what this file pins down is the
part the reference tree itself defines -- the item lists.

Data forms: a point is (x, y) as canonical integers or None for the identity; scalars are canonical integers; `c` is an
oracle/pyref.py curve (c.p: the base field = the sponge field, c.r: the scalar field)."""
from typing import List, Optional, Sequence, Tuple

from oracle.pyref_poseidon import PoseidonSponge

CHALLENGE_SIZE = 128                       # src/hp_as/mod.rs:29, src/r1cs_nark_as/mod.rs:41, r1cs_nark/mod.rs:30
IPA_LINEAR_COMBINATION_CHALLENGE_SIZE = 128      # src/ipa_pc_as/mod.rs:42
TRIVIAL_LINEAR_COMBINATION_CHALLENGE_SIZE = 126  # src/trivial_pc_as/mod.rs:31
CHALLENGE_POINT_SIZE = 184                       # src/ipa_pc_as/mod.rs:43, src/trivial_pc_as/mod.rs:32


def base_sponge(c) -> PoseidonSponge:
    """`S::new()` for S = PoseidonSponge<ConstraintF<G>> (src/hp_as/mod.rs:1047-1055)"""
    return PoseidonSponge(c.p)


def fr_bytes(c, xs: Sequence[int]) -> bytes:
    """`to_bytes!(..)` of scalars: 32 little-endian bytes of the canonical value each, no length"""
    return b"".join((int(x) % c.r).to_bytes(32, "little") for x in xs)


def _tag(sp: PoseidonSponge, some: bool):
    """`Option<T>`: the tag is one element, then the item's own elements"""
    sp.absorb([1 if some else 0])


def _points(sp: PoseidonSponge, pts):
    for P in pts:
        sp.absorb_point(P)


# ---- hp_as -------------------------------------------------------------------------------------------------------------------------
def hp_as(c, sponge: PoseidonSponge, supported_num_elems: int, instances: Sequence[Tuple], hiding_comms: Optional[Tuple],
          low: Sequence, high: Sequence, num_all: int) -> Tuple[List[int], int]:
    """src/hp_as/mod.rs:753-780 (prove) = :863-875 (verify).  instances: (comm_1, comm_2, comm_3) per input then per accumulator, the
    default / placeholder zero instances included (:685-710); hiding_comms: the proof's three commitments or None.
    Returns (the num_all - 1 squeezed mu values, the squeezed nu)."""
    sp = sponge
    sp.absorb([supported_num_elems])                      # `prover_key.supported_num_elems() as u64`
    for inst in instances:                                # data_structures.rs:44-46: comm_1, comm_2, comm_3
        _points(sp, inst)
    _tag(sp, hiding_comms is not None)                    # Option<ProofHidingCommitments>, data_structures.rs:155-157
    if hiding_comms is not None:
        _points(sp, hiding_comms)
    mu = sp.squeeze_nonnative(CHALLENGE_SIZE, num_all - 1) if num_all > 1 else []   # :233-253: ONE squeeze of num_all - 1 sizes
    _points(sp, list(low) + list(high))                   # `absorb(&proof.product_poly_comm)`: low, high (data_structures.rs:125-127)
    nu = sp.squeeze_nonnative(CHALLENGE_SIZE, 1)[0]       # :256-263
    return mu, nu


# ---- r1cs_nark / r1cs_nark_as ---------------------------------------------------------------------------------------------------------
def _first_round_message(sp: PoseidonSponge, msg):
    """r1cs_nark/data_structures.rs:145-147 (comm_a, comm_b, comm_c, Option<randomness>), :88-96 (comm_r_a, comm_r_b, comm_r_c,
    comm_1, comm_2).  msg = (comm_a, comm_b, comm_c, None | five points)"""
    _points(sp, msg[:3])
    _tag(sp, msg[3] is not None)
    if msg[3] is not None:
        _points(sp, msg[3])


def nark_gamma(c, nark_sponge: PoseidonSponge, matrices_hash: bytes, r1cs_input: Sequence[int], first_msg) -> int:
    """r1cs_nark/mod.rs:49-72 `compute_challenge`: the hash, then (input bytes, msg) as one list, one squeeze"""
    sp = nark_sponge
    sp.absorb_bytes(matrices_hash)
    sp.absorb_bytes(fr_bytes(c, r1cs_input))
    _first_round_message(sp, first_msg)
    return sp.squeeze_nonnative(CHALLENGE_SIZE, 1)[0]


def hash_matrices(c, domain_separator: bytes, a, b, cc) -> bytes:
    """r1cs_nark/mod.rs:422-440: Blake2b-256 over the domain separator and `a.serialize() || b.serialize() || c.serialize()`, a matrix
    being `Vec<Vec<(F, usize)>>` (ark-serialize 0.2: u64 little-endian lengths, 32-byte canonical elements, usize as u64).
    Matrices: rows of (coefficient, column) pairs."""
    import hashlib
    h = hashlib.blake2b(digest_size=32)
    h.update(domain_separator)
    for m in (a, b, cc):
        h.update(len(m).to_bytes(8, "little"))
        for row in m:
            h.update(len(row).to_bytes(8, "little"))
            for cf, col in row:
                h.update((int(cf) % c.r).to_bytes(32, "little") + int(col).to_bytes(8, "little"))
    return h.digest()


def nark_as_sponges(c, sponge: PoseidonSponge):
    """src/r1cs_nark_as/mod.rs:112-125: (nark, as, hp) forks of the base sponge"""
    return sponge.fork(b"R1CS-NARK-2020"), sponge.fork(b"AS-FOR-R1CS-NARK-2020"), sponge.fork(b"AS-FOR-HP-2020")


def nark_as_beta(c, as_sponge: PoseidonSponge, as_matrices_hash: bytes, acc_instances: Sequence, input_instances: Sequence,
                 proof_randomness: Optional[Tuple], num: int) -> List[int]:
    """src/r1cs_nark_as/mod.rs:423-448.  acc_instances: (r1cs_input, comm_a, comm_b, comm_c, (hp comm_1, comm_2, comm_3))
    (data_structures.rs:202-209); input_instances: (r1cs_input, first_round_message) (:137-141); proof_randomness: (r1cs_r_input,
    comm_r_a, comm_r_b, comm_r_c) (:342-348) or None.  Returns the num - 1 squeezed values."""
    sp = as_sponge
    sp.absorb_bytes(as_matrices_hash)
    for r1cs_input, ca, cb, cc, hp in acc_instances:      # accumulators FIRST (:433-434)
        sp.absorb_bytes(fr_bytes(c, r1cs_input))
        _points(sp, (ca, cb, cc))
        _points(sp, hp)
    for r1cs_input, msg in input_instances:
        sp.absorb_bytes(fr_bytes(c, r1cs_input))
        _first_round_message(sp, msg)
    _tag(sp, proof_randomness is not None)
    if proof_randomness is not None:
        sp.absorb_bytes(fr_bytes(c, proof_randomness[0]))
        _points(sp, proof_randomness[1:4])
    return sp.squeeze_nonnative(CHALLENGE_SIZE, num - 1) if num > 1 else []


# ---- ipa_pc_as ---------------------------------------------------------------------------------------------------------------------
def ipa_as_sponge(c) -> PoseidonSponge:
    """`DomainSeparatedSponge::<_, S, ASForIpaPCDomain>::new()` (src/ipa_pc_as/mod.rs:572, :696; the domain: :47-57)"""
    return base_sponge(c).fork(b"AS-FOR-IPA-PC-2020")


def ipa_as_alphas(c, as_sponge: PoseidonSponge, checks: Sequence[Tuple[Sequence[int], Tuple]], randomness: Optional[Tuple]) -> List[int]:
    """src/ipa_pc_as/mod.rs:267-296.  checks: (the succinct check polynomial's challenges, final_comm_key) per input then accumulator;
    randomness: (random linear polynomial coefficients, its commitment) or None.  Returns the squeezed linear-combination challenges."""
    sp = as_sponge.clone()
    if randomness is not None:
        co = list(randomness[0]) + [0, 0]
        for i in range(2):
            sp.absorb_bytes(fr_bytes(c, [co[i]]))         # two absorbs of one scalar's bytes each (:270-277)
        sp.absorb_point(randomness[1])
    for poly, comm in checks:
        sp.absorb_bytes(fr_bytes(c, poly))                # :239-250
        sp.absorb_point(comm)
    return sp.squeeze_nonnative(IPA_LINEAR_COMBINATION_CHALLENGE_SIZE, len(checks))


def ipa_as_challenge_point(c, as_sponge: PoseidonSponge, combined_commitment, alphas: Sequence[int], polys: Sequence[Sequence[int]],
                           random_linear_polynomial: Optional[Sequence[int]]) -> int:
    """src/ipa_pc_as/mod.rs:349-388 `compute_new_challenge`"""
    sp = as_sponge.clone()
    sp.absorb_point(combined_commitment)
    _tag(sp, random_linear_polynomial is not None)        # Option<Vec<u8>> (:360-367)
    if random_linear_polynomial is not None:
        co = list(random_linear_polynomial) + [0, 0]
        sp.absorb_bytes(fr_bytes(c, co[:2]))
    nb = (IPA_LINEAR_COMBINATION_CHALLENGE_SIZE + 7) // 8
    for a, poly in zip(alphas, polys):
        sp.absorb_bytes(int(a).to_bytes(32, "little")[:nb])
        sp.absorb_bytes(fr_bytes(c, poly))
    return sp.squeeze_nonnative(CHALLENGE_POINT_SIZE, 1)[0]


# ---- trivial_pc_as -----------------------------------------------------------------------------------------------------------------
def trivial_as(c, sponge: PoseidonSponge, supported_degree: int, instances: Sequence[Tuple], witness_commitments: Sequence,
               evals) -> Tuple[int, List[int]]:
    """src/trivial_pc_as/mod.rs:372-428 (prove) = :523-560 (verify).  instances: (commitment, point, eval) (data_structures.rs:49-54);
    evals(z) -> [(input_witness_eval, witness_eval), ...] at the squeezed challenge point z (the prover's step 3 / the proof's fields).
    Returns (the challenge point, the 2 k squeezed linear-combination challenges)."""
    sp = sponge.clone()
    sp.absorb([supported_degree])
    for (comm, point, ev), wc in zip(instances, witness_commitments):
        sp.absorb_point(comm)
        sp.absorb_bytes(fr_bytes(c, [point]))
        sp.absorb_bytes(fr_bytes(c, [ev]))
        sp.absorb_point(wc)
    z = sp.squeeze_nonnative(CHALLENGE_POINT_SIZE, 1)[0]
    lc = sponge.clone()
    lc.absorb_bytes(int(z).to_bytes(32, "little")[: (CHALLENGE_POINT_SIZE + 7) // 8])
    pairs = evals(z)
    for e, w in pairs:
        lc.absorb_bytes(fr_bytes(c, [e]))
        lc.absorb_bytes(fr_bytes(c, [w]))
    return z, lc.squeeze_nonnative(TRIVIAL_LINEAR_COMBINATION_CHALLENGE_SIZE, 2 * len(pairs))
