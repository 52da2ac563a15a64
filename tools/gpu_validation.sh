# One GPU-box session that produces everything profiles/ holds for a round: `gpurun -- bash tools/gpu_validation.sh`
# (SKIP_TESTS=1 skips the test suite).  Order: tests, rocprofv3 kernel stats, the two PMC passes, then the plain bench run,
# which reads the fresh profiles/pmc_traffic_C2.json for its `roofline.traffic`.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final && cd $R
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1 || { tail -30 gpurun_out/final/gpu_tests.log; exit 1; }
tail -3 gpurun_out/final/gpu_tests.log
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -o c2 -- python3 $R/bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/final/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_f -o f -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $R/gpurun_out/final/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_w -o w -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $R/gpurun_out/final/pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/final/pmc_f gpurun_out/final/pmc_w gpurun_out/final/pmc_traffic_C2.json
cp gpurun_out/final/pmc_traffic_C2.json profiles/pmc_traffic_C2.json
find gpurun_out/final -name '*kernel_trace.csv' -delete; find gpurun_out/final -name '*.db' -delete
timeout -k 10 400 python bench.py --kernel-table gpurun_out/final/c2_table.json > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
tail -1 gpurun_out/final/bench_c2.json | cut -c1-300
du -sh gpurun_out/final
