# PMC passes for the 128x128 implicit-GEMM kernels on a C3 layer (conv 256 -> 256, 3x3, 16x16 images): fwd, dgrad (BWD form)
# and wgrad.  usage (GPU box): bash tools/gpu_pmc_igemm128.sh <tag>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/conv_microbench.py --shapes c3big --reps 5"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
python3 $R/tools/pmc_sq.py $OUT/pmc_igemm128.json igemm2_ $OUT/p1 $OUT/p2 > $OUT/pmc_summary.txt
$CMD > $OUT/microbench.txt 2>&1
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete
