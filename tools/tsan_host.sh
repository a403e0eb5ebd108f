#!/bin/bash
# ThreadSanitizer run of the library's HOST side (the host thread pool behind amsm_host_lincomb[_batch] and the host backend's parallel loops): same host-only build as
# tools/asan_host.sh with -fsanitize=thread; tests/test_host_fr_cpu.py runs against it.  Usage (repo root, CPU only): bash tools/tsan_host.sh
set -eu
R=$(pwd)
D=$R/build/tsan
mkdir -p $D
cd $D
for u in api kern_pallas kern_bls12_381 kern_fr; do
  hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 --offload-host-only -fsanitize=thread -fno-omit-frame-pointer \
    -Wno-unused-result -Wno-pass-failed -c $R/accumulation_amd/csrc/$u.hip -o $u.o &
done
wait
hipcc -shared -fPIC -fsanitize=thread --offload-host-only -o libamsm_tsan.so api.o kern_pallas.o kern_bls12_381.o kern_fr.o 2>/dev/null || true
nm -u api.o kern_pallas.o kern_bls12_381.o kern_fr.o | awk '/__hip_fatbin_/ {print $2}' | sort -u > fatbins.txt
for s in $(cat fatbins.txt); do echo "__attribute__((aligned(4096))) const char $s[4096] = {0};"; done > stub.c
gcc -shared -fPIC -o libstub.so stub.c
hipcc -shared -fPIC -fsanitize=thread --offload-host-only -o libamsm_tsan.so api.o kern_pallas.o kern_bls12_381.o kern_fr.o \
  -L. -lstub -Wl,-rpath,$D
cat > run_host_tests.py <<PY
import sys
sys.path.insert(0, "$R")
import accumulation_amd.ffi as ffi
ffi.LIB_PATH = "$D/libamsm_tsan.so"
import pytest
sys.exit(pytest.main(["-x", "-q", "$R/tests/test_host_fr_cpu.py", "$R/tests/host_backend/test_host_context_cpu.py", "$R/tests/host_backend/test_host_msm_cpu.py", "$R/tests/host_backend/test_host_vec_cpu.py", "$R/tests/host_backend/test_host_ipa_cpu.py", "$R/tests/host_backend/test_host_hp_as_scheme_cpu.py", "$R/tests/host_backend/test_host_r1cs_nark_as_scheme_cpu.py", "-p", "no:cacheprovider"]))
PY
cd $R
TSAN_LIB=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
LD_PRELOAD=$TSAN_LIB TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 python $D/run_host_tests.py
