# usage (on the GPU box, via gpurun): bash tools/gpu_bench_profile.sh <tag> <cfg> [<cfg> ...]
# For each config: the bench line (with roofline + per-call table) and a rocprofv3 kernel-stats summary of a short replay run.
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  python3 $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --min-gpu-seconds 2 --kernel-table $OUT/${c}_call_table.json > $OUT/${c}_bench.json 2> $OUT/${c}_bench.err || { tail -5 $OUT/${c}_bench.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$OUT/${c}_bench.json"))
r=d.get("roofline") or {}
print("$c", "ms/step", round(d["ms_per_step"],4), "img/s", round(d["value"]), "| dom", r.get("kernel","")[:34], "frac", r.get("frac"), "exec", r.get("frac_executed_taps"), "| conv", (r.get("conv_family") or {}).get("us_per_step"), "other", r.get("other_families_us_per_step"))
PY
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --repeats 2 --min-gpu-seconds 0 --no-cpu-baseline --no-roofline > $OUT/${c}_prof.log 2>&1 || { tail -5 $OUT/${c}_prof.log; exit 1; }
  f=$(find $OUT/prof_$c -name '*kernel_stats.csv' | head -1)
  cp $f $OUT/${c}_kernel_stats.csv
  rm -rf $OUT/prof_$c
done
