// Elliptic-curve kernels instantiated for Pallas (255-bit base field, 8 x u32 limbs).
#define AMSM_FQ PallasFq
#define AMSM_FR PallasFr  // the curve's scalar field (GLV split of fold scalars, host_glv.h)
#define AMSM_CURVE_ID 0
#include "kern_ec.inc"
