"""Loader for the arkworks-generated golden vectors tests/golden/ark_*.json (written by tools/ark_vectors, a Rust crate pinned to
the reference's dependency specifications; it cannot run in the build image -- no cargo -- so the files are ABSENT until someone
with a Rust toolchain generates and commits them).  The consuming tests skip with "parity unpinned: <file> absent" until then:
one `cargo run` turns the repo's parity from "partial / unpinned" to pinned.

ARK_VECTORS_DIR overrides the directory (tools/ark_vectors/emulate.py writes the same layout from the repo's OWN oracle into a
scratch directory, to exercise these consumers -- that is not a pin and its files say so in their "generator" field)."""
import json
import os

import pytest

from oracle import pyref as o

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = os.environ.get("ARK_VECTORS_DIR", GOLDEN_DIR)
    path = os.path.join(d, name)
    if not os.path.exists(path):
        pytest.skip(f"parity unpinned: {name} absent (generate with tools/ark_vectors: cargo run --release -- ../../tests/golden)")
    with open(path) as f:
        return json.load(f)


def pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def ints(xs):
    return [int(x, 16) for x in xs]


def seeded_msm_inputs(c, case):
    """the synthetic stream both sides share (accumulation_amd/csrc/rng.h == oracle/pyref.py): points = rng_scalar * generator"""
    n = case["n"]
    return case["seed_points"], case["seed_scalars"], n
