mkdir -p gpurun_out/r3l && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_parity_full.py > gpurun_out/r3l/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3l/t.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3l/t.log | head -20; exit $rc; fi
timeout -k 10 200 python bench.py --config C2 --no-cpu-baseline --no-roofline --min-gpu-seconds 3 > gpurun_out/r3l/C2.json 2> gpurun_out/r3l/err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r3l/C2.json').read().strip().splitlines()[-1]); print('C2', d['ms_per_step'])"
