// Which device the check programs of this directory build their Context on: 0 (the GPU) unless the TEST sets
// AMSM_CHECK_DEVICE=-1 (include/amsm.h: AMSM_DEVICE_HOST, the library's host backend).  A knob of the test programs, not of the library.
#pragma once
#include <cstdlib>
inline int check_device() {
  const char* e = std::getenv("AMSM_CHECK_DEVICE");
  return e ? std::atoi(e) : 0;
}

// AMSM_CHECK_SHARDS=N (N >= 2): the same program over a multi-device context of N shards, all on device 0 -- how a one-GPU box
// runs the drivers over sharded keys (include/amsm.h amsm_ctx_create_multi accepts a repeated device id).
#include <vector>

#include "amsm.hpp"
inline amsm::Context check_context(int curve) {
  const char* e = std::getenv("AMSM_CHECK_SHARDS");
  const int shards = e ? std::atoi(e) : 0;
  if (shards >= 2) return amsm::Context(curve, std::vector<int>((size_t)shards, 0));
  return amsm::Context(curve, check_device());
}
