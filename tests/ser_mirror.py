"""ark-serialize 0.2 encoding (the layout of include/amsm_serialize.hpp) of the PYTHON mirrors' accumulators, inputs and proofs,
so that what a C++ driver serialises (tools/profile_as --dump) can be compared byte for byte with the mirror's objects.
Scalars the mirrors keep as Python ints are written here as canonical little-endian integers; points and device vectors go
through the library's own primitives (amsm_points_serialize / amsm_fr_serialize -- checked against oracle/pyref_ser.py in
tests/test_wire_format_cpu.py), and tests/test_profile_as_dump.py checks this module itself against oracle/pyref_ser.py.
Test infrastructure."""
import ctypes as C

import numpy as np

from accumulation_amd import ffi
from accumulation_amd.scalar_field import MODULI


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Ser:
    def __init__(self, ctx):
        self.ctx, self.lib, self.curve, self.r = ctx, ctx._lib, ctx.curve, MODULI[ctx.curve]
        self.pt_size = self.lib.amsm_point_serialized_size(self.curve, 1)

    # ---- primitives ----
    @staticmethod
    def u64(x):
        return int(x).to_bytes(8, "little")

    def fr(self, x):  # a canonical Python int
        return (int(x) % self.r).to_bytes(32, "little")

    def frs(self, xs):  # Vec<F> of Python ints
        return self.u64(len(xs)) + b"".join(self.fr(x) for x in xs)

    def fr_vector(self, v):  # Vec<F> resident on the device (Montgomery limbs)
        limbs = np.ascontiguousarray(v.download(), dtype=np.uint64)
        n = limbs.shape[0]
        out = np.zeros(32 * n, dtype=np.uint8)
        if n:
            ffi.check(self.lib.amsm_fr_serialize(self.curve, _p(limbs), n, _p(out)), "amsm_fr_serialize")
        return self.u64(n) + out.tobytes()

    def point(self, pt):  # (xy Montgomery limbs, is_inf)
        xy = np.ascontiguousarray(np.asarray(pt[0], dtype=np.uint64).reshape(-1))
        inf = np.array([1 if pt[1] else 0], dtype=np.uint8)
        out = np.zeros(self.pt_size, dtype=np.uint8)
        ffi.check(self.lib.amsm_points_serialize(self.curve, _p(xy), _p(inf), 1, 1, _p(out)), "amsm_points_serialize")
        return out.tobytes()

    def points(self, pts):
        return self.u64(len(pts)) + b"".join(self.point(p) for p in pts)

    @staticmethod
    def option(x, enc):
        return b"\x00" if x is None else b"\x01" + enc(x)

    # ---- hp_as (src/hp_as/data_structures.rs) ----
    def hp_instance(self, x):
        return self.point(x.comm_1) + self.point(x.comm_2) + self.point(x.comm_3)

    def hp_witness(self, x):
        return self.fr_vector(x.a_vec) + self.fr_vector(x.b_vec) + self.option(
            x.randomness, lambda r: self.fr(r.rand_1) + self.fr(r.rand_2) + self.fr(r.rand_3))

    def hp_accumulator(self, x):
        return self.hp_instance(x.instance) + self.hp_witness(x.witness)

    def hp_proof(self, x):
        return self.points(x.product_poly_comm.low) + self.points(x.product_poly_comm.high) + self.option(
            x.hiding_comms, lambda h: self.point(h.comm_1) + self.point(h.comm_2) + self.point(h.comm_3))

    # ---- r1cs_nark / r1cs_nark_as ----
    def nark_first_msg(self, x):
        return self.point(x.comm_a) + self.point(x.comm_b) + self.point(x.comm_c) + self.option(
            x.randomness, lambda m: b"".join(self.point(p) for p in (m.comm_r_a, m.comm_r_b, m.comm_r_c, m.comm_1, m.comm_2)))

    def nark_second_msg(self, x):
        return self.fr_vector(x.blinded_witness) + self.option(
            x.randomness, lambda s: self.fr(s.sigma_a) + self.fr(s.sigma_b) + self.fr(s.sigma_c) + self.fr(s.sigma_o))

    def nark_as_input(self, x):
        return self.frs(x.instance.r1cs_input) + self.nark_first_msg(x.instance.first_round_message) + self.nark_second_msg(x.witness)

    def nark_as_accumulator(self, x):
        i, w = x.instance, x.witness
        inst = self.frs(i.r1cs_input) + self.point(i.comm_a) + self.point(i.comm_b) + self.point(i.comm_c) + self.hp_instance(i.hp_instance)
        wit = self.fr_vector(w.r1cs_blinded_witness) + self.hp_witness(w.hp_witness) + self.option(
            w.randomness, lambda s: self.fr(s.sigma_a) + self.fr(s.sigma_b) + self.fr(s.sigma_c))
        return inst + wit

    def nark_as_proof(self, x):
        return self.hp_proof(x.hp_proof) + self.option(
            x.randomness, lambda p: self.frs(p.r1cs_r_input) + self.point(p.comm_r_a) + self.point(p.comm_r_b) + self.point(p.comm_r_c))

    # ---- ipa_pc / ipa_pc_as ----
    def ipa_labeled_commitment(self, c):
        return self.u64(0) + self.point(c.comm) + self.option(c.shifted_comm, self.point) + b"\x00"

    def ipa_proof(self, x):
        return (self.points(x.l_vec) + self.points(x.r_vec) + self.point(x.final_comm_key) + self.fr(x.c) +
                self.option(x.hiding_comm, self.point) + self.option(x.rand, self.fr))

    def ipa_as_instance(self, x):  # also the accumulator: the witness is ()
        return self.ipa_labeled_commitment(x.ipa_commitment) + self.fr(x.point) + self.fr(x.evaluation) + self.ipa_proof(x.ipa_proof)

    def ipa_as_proof(self, x):  # Option<Randomness>
        return self.option(x, lambda v: self.frs(v.random_linear_polynomial) + self.point(v.random_linear_polynomial_commitment) +
                           self.fr(v.commitment_randomness))

    # ---- trivial_pc_as ----
    def trivial_commitment(self, c):
        return self.u64(0) + self.point(c.elem) + self.option(c.degree_bound, self.u64)

    def trivial_instance(self, x):
        return self.trivial_commitment(x.commitment) + self.fr(x.point) + self.fr(x.eval)

    def trivial_polynomial(self, x):
        return self.u64(0) + self.frs(x.coeffs) + self.option(x.degree_bound, self.u64) + self.option(x.hiding_bound, self.u64)

    def trivial_accumulator(self, x):
        return self.trivial_instance(x.instance) + self.trivial_polynomial(x.witness)

    def trivial_proof(self, x):
        return self.u64(len(x)) + b"".join(self.trivial_commitment(p.witness_commitment) + self.fr(p.witness_eval) + self.fr(p.eval) for p in x)
