mkdir -p gpurun_out/r4d
(python -m pytest tests -q -m gpu 2>&1 | grep -v "^  File" | tail -40) > gpurun_out/r4d/full2.log 2>&1
cat gpurun_out/r4d/full2.log
