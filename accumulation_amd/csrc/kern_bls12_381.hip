// Elliptic-curve kernels instantiated for BLS12-381 G1 (381-bit base field, 12 x u32 limbs).
#define AMSM_FQ Bls12381Fq
#define AMSM_FR Bls12381Fr  // the curve's scalar field (GLV split of fold scalars, host_glv.h)
#define AMSM_CURVE_ID 1
#include "kern_ec.inc"
