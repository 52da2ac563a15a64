mkdir -p gpurun_out/r3i && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 524288 262144 524288 262144; do
MOVAE_THIN_CS_MIN=$v timeout -k 10 200 python bench.py --config C2 --no-cpu-baseline --no-roofline > gpurun_out/r3i/c2_$v.json 2> gpurun_out/r3i/c2.err || exit 1
echo cs_min $v $(python -c "
import json; d=json.loads(open('gpurun_out/r3i/c2_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done
for v in 131072; do
MOVAE_THIN_CS_MIN=$v timeout -k 10 200 python bench.py --config C1 --no-cpu-baseline --no-roofline > gpurun_out/r3i/c1_$v.json 2> gpurun_out/r3i/c1.err || exit 1
echo C1 cs_min $v $(python -c "
import json; d=json.loads(open('gpurun_out/r3i/c1_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done
timeout -k 10 200 python bench.py --config C1 --no-cpu-baseline --no-roofline > gpurun_out/r3i/c1_def.json 2> gpurun_out/r3i/c1.err || exit 1
echo C1 default $(python -c "
import json; d=json.loads(open('gpurun_out/r3i/c1_def.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
