mkdir -p gpurun_out/r3d && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
for c in C2 C3 C4 C5; do timeout -k 10 200 python bench.py --config $c --no-cpu-baseline > gpurun_out/r3d/bench_$c.json 2> gpurun_out/r3d/bench_$c.err || exit 1; done &&
for c in C2 C3 C4 C5; do timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3d/tr_$c -o x -- python3 bench.py --config $c --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1 || exit 1; python tools/step_sequence.py gpurun_out/r3d/tr_$c > gpurun_out/r3d/seq_$c.txt; rm -rf gpurun_out/r3d/tr_$c; done; grep -h ms_per_step gpurun_out/r3d/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['workload'][:40], d['ms_per_step'], d['value'])"
