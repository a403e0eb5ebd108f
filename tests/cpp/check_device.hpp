// Which device the check programs of this directory build their Context on: 0 (the GPU) unless the TEST sets
// AMSM_CHECK_DEVICE=-1 (include/amsm.h: AMSM_DEVICE_HOST, the library's host backend).  A knob of the test programs, not of the library.
#pragma once
#include <cstdlib>
inline int check_device() {
  const char* e = std::getenv("AMSM_CHECK_DEVICE");
  return e ? std::atoi(e) : 0;
}
