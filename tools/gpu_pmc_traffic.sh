# HBM-side traffic of every kernel of a bench config (two separate --pmc passes, kernel trace only: MI355X_MICROARCH.md, HBM section).
# usage (GPU box): bash tools/gpu_pmc_traffic.sh <tag> <cfg>      -> gpurun_out/<tag>/pmc_traffic_<cfg>.json
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
CFG=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --repeats 1 --min-gpu-seconds 0"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/f -o f -- python3 $R/bench.py $ARGS > $OUT/f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/w -o w -- python3 $R/bench.py $ARGS > $OUT/w.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/f $OUT/w $OUT/pmc_traffic_$CFG.json
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +20M -delete
