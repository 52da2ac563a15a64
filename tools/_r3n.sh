mkdir -p gpurun_out/r3n && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_kgemm.py tests/test_hip_models.py tests/test_hip_deferred_reduce.py tests/test_hip_ops.py -m gpu -q -x > gpurun_out/r3n/t.log 2>&1; rc=$?; tail -2 gpurun_out/r3n/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 0 2 0 2; do for c in C2; do
MOVAE_PAIR_INTERLEAVE=$v timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3n/${c}_$v.json 2> gpurun_out/r3n/err || exit 1
echo $c order $v $(python -c "
import json; d=json.loads(open('gpurun_out/r3n/${c}_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done; done
for v in 0 2; do for c in C1 C3 C4 C5; do
MOVAE_PAIR_INTERLEAVE=$v timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3n/${c}_$v.json 2> gpurun_out/r3n/err || exit 1
echo $c order $v $(python -c "
import json; d=json.loads(open('gpurun_out/r3n/${c}_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done; done
