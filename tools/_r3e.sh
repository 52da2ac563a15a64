mkdir -p gpurun_out/r3e
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_parity_full.py > gpurun_out/r3e/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3e/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/_r3d.sh
