#!/bin/bash
# measurements (16 channels each) per chain call x chains in flight at C2, K1 launch groups of $GROUP channels inside the call
# (GROUP=0: the whole call is one launch group): bash tools/c2_group_check.sh [GROUP] ["m1 m2 .."] ["l1 l2 .."]
GROUP=${1:-0}; MS=${2:-"1 2 3 4"}; LS=${3:-"2 3 4"}
for m in $MS; do for l in $LS; do
blocks=$((240 / m))
IMPULSE_BENCH_GROUP=$GROUP IMPULSE_BENCH_MEASUREMENTS=$m timeout -k 10 300 python bench.py --steps 8 --warmup 2 --blocks $blocks --lanes $l --no-cpu-baseline --no-pmc > gpurun_out/c2_m${m}_l$l.json 2>gpurun_out/c2_m${m}_l$l.err || { echo "m $m l $l failed"; tail -2 gpurun_out/c2_m${m}_l$l.err; continue; }
python - <<PY
import json
d=json.load(open("gpurun_out/c2_m${m}_l$l.json"))
print("group $GROUP measurements/call $m chains $l:", round(d["value"]), "IR/s chain,", round(d["deconv_only"]["value"]), "K1 alone")
PY
done; done
