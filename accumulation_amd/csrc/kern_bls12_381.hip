// Elliptic-curve kernels instantiated for BLS12-381 G1 (381-bit base field, 12 x u32 limbs).
#define AMSM_FQ Bls12381Fq
#include "kern_ec.inc"
