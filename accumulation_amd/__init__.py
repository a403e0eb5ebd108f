"""accumulation_amd -- MI355X-native MSM engine behind the accumulation-scheme prover hot path.

The product is libamsm.so (hand-written HIP for gfx950 behind the C ABI in include/amsm.h); this package
is the thin host-side mirror of the reference interfaces used by tests, bench.py and examples.
"""
from . import ffi  # noqa: F401
from .engine import (CommitterKey, Context, FrVector, MultiContext, PedersenCommitment,  # noqa: F401
                     PointVector, VariableBaseMSM)
