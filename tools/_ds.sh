mkdir -p gpurun_out/r4d
(python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^  File" | tail -25) > gpurun_out/r4d/full.log 2>&1
cat gpurun_out/r4d/full.log
