# Round evidence for a set of configs, one call: per config the bench line WITH the CPU leg (bounded: --cpu-seconds, few ELBO-check
# steps for the large ones), the per-call table, a rocprofv3 kernel-stats summary and the per-step kernel sequence; with PMC=1 also
# the FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py).   usage (GPU box): bash tools/gpu_round_evidence.sh <tag> <cfg> [...]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  case $c in C1|C2) CPU="--cpu-seconds 10 --elbo-check-steps 20";; *) CPU="--cpu-seconds 8 --elbo-check-steps 2";; esac
  python3 $R/bench.py --config $c --steps 20 --warmup 5 $CPU --min-gpu-seconds 3 --kernel-table $OUT/${c}_call_table.json > $OUT/${c}_bench.json 2> $OUT/${c}_bench.err || { tail -5 $OUT/${c}_bench.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$OUT/${c}_bench.json"))
r=d.get("roofline") or {}
cb=d.get("cpu_baseline") or {}
print("$c", "ms/step", round(d["ms_per_step"],4), "img/s", round(d["value"]), "| dom", r.get("kernel","")[:34], "frac", r.get("frac"), "| conv", (r.get("conv_family") or {}).get("frac"), "| cpu", round(cb.get("value",0)), cb.get("sample","")[:40], flush=True)
PY
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -o $c -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --repeats 2 --min-gpu-seconds 0 --no-cpu-baseline --no-roofline > $OUT/${c}_prof.log 2>&1 || { tail -5 $OUT/${c}_prof.log; exit 1; }
  cp $(find /tmp/prof_$c -name '*kernel_stats.csv' | head -1) $OUT/${c}_kernel_stats.csv
  python3 $R/tools/step_sequence.py /tmp/prof_$c > $OUT/${c}_step_sequence.txt
  head -1 $OUT/${c}_step_sequence.txt
  rm -rf /tmp/prof_$c
  if [ "$PMC" = "1" ]; then
    ARGS="--config $c --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --repeats 1 --min-gpu-seconds 0"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pf_$c -o f -- python3 $R/bench.py $ARGS > $OUT/${c}_f.log 2>&1
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pw_$c -o w -- python3 $R/bench.py $ARGS > $OUT/${c}_w.log 2>&1
    python3 $R/tools/pmc_traffic.py /tmp/pf_$c /tmp/pw_$c $OUT/pmc_traffic_$c.json > /dev/null
    rm -rf /tmp/pf_$c /tmp/pw_$c
  fi
done
# the opt-in bf16 leg (never the headline): one bench line per config named in BF16_CFGS, e.g. BF16_CFGS="C3 C5"
for c in $BF16_CFGS; do
  python3 $R/bench.py --config $c --dtype bf16 --steps 20 --warmup 5 --cpu-seconds 8 --elbo-check-steps 2 --min-gpu-seconds 3 > $OUT/${c}_bf16_bench.json 2> $OUT/${c}_bf16_bench.err || { tail -5 $OUT/${c}_bf16_bench.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/${c}_bf16_bench.json')); print('$c bf16 ms/step', round(d['ms_per_step'],4), d['dtype'], (d['roofline'] or {}).get('kernel'), (d['roofline'] or {}).get('frac'))"
done
