mkdir -p gpurun_out/r3m && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_parity_full.py > gpurun_out/r3m/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3m/t.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3m/t.log | head -20; exit $rc; fi
for c in C1 C2 C5; do timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3m/$c.json 2> gpurun_out/r3m/err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r3m/$c.json').read().strip().splitlines()[-1]); print('$c', d['ms_per_step'])"; done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3m/tr -o x -- python3 bench.py --config C2 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1 || exit 1; python tools/step_sequence.py gpurun_out/r3m/tr > gpurun_out/r3m/seq_C2.txt; rm -rf gpurun_out/r3m/tr; head -1 gpurun_out/r3m/seq_C2.txt
