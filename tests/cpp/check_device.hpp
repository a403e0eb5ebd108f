// Which device the check programs of this directory build their Context on: 0 (the GPU) unless the TEST sets
// AMSM_CHECK_DEVICE=-1 (include/amsm.h: AMSM_DEVICE_HOST, the library's host backend).  A knob of the test programs, not of the library.
#pragma once
#include <cstdlib>
inline int check_device() {
  const char* e = std::getenv("AMSM_CHECK_DEVICE");
  return e ? std::atoi(e) : 0;
}

// AMSM_CHECK_ITERATIONS=N: every scenario of the six-scenario template that iterates (src/lib.rs:398-448) runs N iterations instead
// of the 2-3 the GPU suite affords -- the reference runs NUM_ITERATIONS = 50 (src/lib.rs:273); tests/test_cpp_templates_50_cpu.py sets it
// on the host backend.  no_inputs_init stays at 1 like the reference's (src/lib.rs:451-458).
inline size_t check_iterations(size_t dflt) {
  const char* e = std::getenv("AMSM_CHECK_ITERATIONS");
  const long v = e ? std::atol(e) : 0;
  return (v > 0 && dflt > 1) ? (size_t)v : dflt;
}
// AMSM_CHECK_SKIP_CROSS=1: stop after the scenarios (the deterministic cross-check lines and error cases are the short run's job)

// AMSM_CHECK_SHARDS=N (N >= 2): the same program over a multi-device context of N shards, all on device 0 -- how a one-GPU box
// runs the drivers over sharded keys (include/amsm.h amsm_ctx_create_multi accepts a repeated device id).
#include <vector>

#include "amsm.hpp"
// AMSM_CHECK_CURVE=1: the same program over BLS12-381 G1 instead of the curve its source names (the reference's tests instantiate the
// schemes over Pallas only; the drivers are generic).  Lines a test compares with Pallas values are the short run's job: set
// AMSM_CHECK_SKIP_CROSS=1 beside it.
inline amsm::Context check_context(int curve) {
  if (const char* c = std::getenv("AMSM_CHECK_CURVE")) curve = std::atoi(c);
  const char* e = std::getenv("AMSM_CHECK_SHARDS");
  const int shards = e ? std::atoi(e) : 0;
  if (shards >= 2) return amsm::Context(curve, std::vector<int>((size_t)shards, 0));
  return amsm::Context(curve, check_device());
}
