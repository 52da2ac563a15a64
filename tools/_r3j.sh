mkdir -p gpurun_out/r3j && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MOVAE_XCD_REMAP=1 timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_models.py tests/test_hip_bf16.py tests/test_hip_fused_bn.py -m gpu -q -x > gpurun_out/r3j/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3j/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
for x in 0 1 0 1; do for c in C3; do
MOVAE_XCD_REMAP=$x timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3j/${c}_$x.json 2> gpurun_out/r3j/err || exit 1
echo $c xcd $x $(python -c "
import json; d=json.loads(open('gpurun_out/r3j/${c}_$x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done; done
for x in 0 1; do for c in C3 C5; do
MOVAE_XCD_REMAP=$x timeout -k 10 200 python bench.py --config $c --dtype bf16 --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3j/${c}_bf_$x.json 2> gpurun_out/r3j/err || exit 1
echo $c bf16 xcd $x $(python -c "
import json; d=json.loads(open('gpurun_out/r3j/${c}_bf_$x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done; done
for x in 0 1; do for c in C4 C5; do
MOVAE_XCD_REMAP=$x timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3j/${c}_$x.json 2> gpurun_out/r3j/err || exit 1
echo $c xcd $x $(python -c "
import json; d=json.loads(open('gpurun_out/r3j/${c}_$x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done; done
