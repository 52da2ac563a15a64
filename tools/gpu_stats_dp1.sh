# rocprofv3 kernel stats of the data-parallel step driven by one rank over RCCL (captured all-reduce inside the step's graph)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/stats_dp1
cd /tmp && export TMPDIR=/tmp
export MOVAE_FORCE_DP=1
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_dp1 -o dp1 -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $R/gpurun_out/stats_dp1/run.log 2>&1
find $R/gpurun_out/stats_dp1 -name '*kernel_trace.csv' -delete; find $R/gpurun_out/stats_dp1 -name '*.db' -delete
grep -ho "ms_per_step\": [0-9.]*" $R/gpurun_out/stats_dp1/run.log
