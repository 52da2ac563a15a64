mkdir -p gpurun_out/r3full; timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r3full/t.log 2>&1; tail -6 gpurun_out/r3full/t.log
