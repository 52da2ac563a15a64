mkdir -p gpurun_out/r3h && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_parity_full.py > gpurun_out/r3h/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3h/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
for c in C1 C2; do timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline > gpurun_out/r3h/bench_$c.json 2> gpurun_out/r3h/bench_$c.err || exit 1; done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3h/tr -o x -- python3 bench.py --config C2 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1 || exit 1; python tools/step_sequence.py gpurun_out/r3h/tr > gpurun_out/r3h/seq_C2.txt; rm -rf gpurun_out/r3h/tr
grep -h ms_per_step gpurun_out/r3h/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['workload'][:40], d['ms_per_step'], d['value'])"
head -1 gpurun_out/r3h/seq_C2.txt; grep -c "act_bwd_k\|recon_bwd" gpurun_out/r3h/seq_C2.txt
