// Elliptic-curve kernels instantiated for Pallas (255-bit base field, 8 x u32 limbs).
#define AMSM_FQ PallasFq
#include "kern_ec.inc"
