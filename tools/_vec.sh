mkdir -p gpurun_out/r4v
(python -m pytest tests/test_vec_gpu.py tests/test_hp_as_scheme_gpu.py tests/test_as_layers_vs_oracle_gpu.py tests/test_r1cs_nark_gpu.py -q -m gpu -x 2>&1 | tail -8) > gpurun_out/r4v/vect.log 2>&1
cat gpurun_out/r4v/vect.log
