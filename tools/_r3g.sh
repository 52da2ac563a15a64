mkdir -p gpurun_out/r3g && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mb in 1000000000000 33554432 12582912 6291456; do
export MOVAE_DEFER_MAX_BYTES=$mb
for c in C2 C4 C5; do timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-roofline > gpurun_out/r3g/bench_${c}_$mb.json 2> gpurun_out/r3g/bench_${c}_$mb.err || exit 1; done
done
export MOVAE_DEFER_MAX_BYTES=1000000000000
for c in C2 C4; do timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3g/tr_$c -o x -- python3 bench.py --config $c --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1 || exit 1; python tools/step_sequence.py gpurun_out/r3g/tr_$c > gpurun_out/r3g/seq_$c.txt; rm -rf gpurun_out/r3g/tr_$c; done
for f in gpurun_out/r3g/bench_*.json; do echo $f $(python -c "
import sys, json
d = json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'])"); done
