mkdir -p gpurun_out/r3e
timeout -k 10 900 python -m pytest tests/test_hip_deferred_reduce.py tests/test_hip_models.py tests/test_hip_kgemm.py -m gpu -q -x > gpurun_out/r3e/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3e/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/_r3d.sh
