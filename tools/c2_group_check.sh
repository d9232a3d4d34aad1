#!/bin/bash
# measurements (16 channels each) per chain call x chains in flight at C2, 5 GiB of inputs in rotation
for m in 1 2 3 4; do for l in 2 3 4; do
blocks=$((240 / m))
IMPULSE_BENCH_MEASUREMENTS=$m timeout -k 10 300 python bench.py --steps 8 --warmup 2 --blocks $blocks --lanes $l --no-cpu-baseline --no-pmc > gpurun_out/c2_m${m}_l$l.json 2>gpurun_out/c2_m${m}_l$l.err || { echo "m $m l $l failed"; tail -2 gpurun_out/c2_m${m}_l$l.err; continue; }
python - <<PY
import json
d=json.load(open("gpurun_out/c2_m${m}_l$l.json"))
print("measurements/call $m chains $l:", round(d["value"]), "IR/s chain,", round(d["deconv_only"]["value"]), "K1 alone")
PY
done; done
