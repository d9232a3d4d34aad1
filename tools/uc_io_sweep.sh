#!/bin/bash
for f in 0 3 1; do for l in 3 4; do
timeout -k 10 300 python tools/uc_io_probe.py $f --steps 8 --warmup 2 --blocks 160 --lanes $l --no-cpu-baseline --no-pmc > gpurun_out/uc_f${f}_l$l.json 2>gpurun_out/uc_f${f}_l$l.err || { echo "flag $f lanes $l failed"; tail -3 gpurun_out/uc_f${f}_l$l.err; continue; }
python - <<PY
import json
d=json.load(open("gpurun_out/uc_f${f}_l$l.json"))
print("alloc flag $f chains $l:", round(d["value"]), "IR/s chain,", round(d["deconv_only"]["value"]), "K1 alone")
PY
done; done
