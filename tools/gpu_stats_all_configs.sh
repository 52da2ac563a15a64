set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/stats_all
cd /tmp && export TMPDIR=/tmp
for c in C1 C3 C4 C5; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_all/$c -o $c -- python3 $R/bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/stats_all/$c.log 2>&1
  echo $c $(grep -ho "ms_per_step\": [0-9.]*" $R/gpurun_out/stats_all/$c.log)
done
find $R/gpurun_out/stats_all -name '*kernel_trace.csv' -delete; find $R/gpurun_out/stats_all -name '*.db' -delete
du -sh $R/gpurun_out/stats_all
