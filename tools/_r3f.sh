mkdir -p gpurun_out/r3f
timeout -k 10 300 python -m pytest "tests/test_hip_deferred_reduce.py::test_aggregated_step_identical_with_and_without_deferral" -m gpu -q -k "C3" > gpurun_out/r3f/only.log 2>&1; grep -n "^E  " gpurun_out/r3f/only.log | cut -c1-1500 | head -5
