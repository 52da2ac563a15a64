mkdir -p gpurun_out/r3k && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_parity_full.py > gpurun_out/r3k/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3k/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
for x in 0 1 0 1; do for c in C2; do
MOVAE_KGEMM_BN_FIN=$x timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3k/${c}_$x.json 2> gpurun_out/r3k/err || exit 1
echo $c fin $x $(python -c "
import json; d=json.loads(open('gpurun_out/r3k/${c}_$x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")
done; done
for x in 0 1; do MOVAE_KGEMM_BN_FIN=$x timeout -k 10 200 python bench.py --config C1 --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3k/C1_$x.json 2> gpurun_out/r3k/err || exit 1
echo C1 fin $x $(python -c "
import json; d=json.loads(open('gpurun_out/r3k/C1_$x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])"); done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3k/tr -o x -- python3 bench.py --config C2 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1 || exit 1; python tools/step_sequence.py gpurun_out/r3k/tr > gpurun_out/r3k/seq_C2.txt; rm -rf gpurun_out/r3k/tr; head -1 gpurun_out/r3k/seq_C2.txt
