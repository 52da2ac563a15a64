# PMC passes + kernel durations for the block-internal split-K kernels (csrc/kgemm.h) on the C2 middle layers.
# usage (GPU box): bash tools/gpu_pmc_kgemm.sh <tag> [microbench args]
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/conv_microbench.py --shapes c2 --reps 5 --no-dbias --index 3 4 8 $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- $CMD > $OUT/ks.log 2>&1
cp $(find $OUT/ks -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1 || true
python3 $R/tools/pmc_sq.py $OUT/pmc_kgemm.json k $OUT/p1 $OUT/p2 > $OUT/pmc_summary.txt || true
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
