"""The adversarial-point fixture (tests/golden/adversarial_points.json) is what it claims to be: every point is on its
curve and in the prime-order group, and the advertised coordinate is extreme in the device's internal Montgomery radix.
No device needed; the GPU tests that use the fixture are in tests/test_adversarial_points_gpu.py."""
import json
import os

from oracle import pyref as o

CURVES = [o.PALLAS, o.BLS12_381_G1]
FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "adversarial_points.json")))


def test_fixture_points_are_what_they_claim():
    for c in CURVES:
        bits = FIX["internal_radix_bits"][c.name]
        for kind, pts in FIX["curves"][c.name].items():
            for x, y in pts:
                P = (int(x, 16), int(y, 16))
                assert o.is_on_curve(c, P) and o.mul(c, c.r, P) is None
                v = (P[1] if kind.endswith("y") else P[0]) * (1 << bits) % c.p
                edge = v if kind.startswith("tiny") else c.p - v
                assert edge < (1 << (c.p.bit_length() - 17)), (c.name, kind)
