#!/bin/bash
# chains in flight x channels per K1 launch group at C2 (bench.py's own arrangement): bash tools/c2_arrangement_sweep.sh
for g in 16 32; do for l in 2 3 4; do
IMPULSE_BENCH_GROUP=$g timeout -k 10 200 python bench.py --steps 5 --warmup 2 --blocks 120 --lanes $l --no-cpu-baseline --no-pmc > gpurun_out/c2_g${g}_l${l}.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/c2_g${g}_l${l}.json"))
print("group $g chains $l:", round(d["value"]), "IR/s chain,", round(d["deconv_only"]["value"]), "K1 alone")
PY
done; done
