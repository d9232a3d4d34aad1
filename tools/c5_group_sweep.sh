for g in 4 6 8 12; do for l in 1 2 3 4; do
IMPULSE_BENCH_GROUP=$g timeout -k 10 200 python bench.py --workload c5 --steps 6 --warmup 2 --lanes $l --no-cpu-baseline --no-pmc > gpurun_out/c5_g${g}_l${l}.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/c5_g${g}_l${l}.json"))
print("group $g lanes $l:", round(d["value"]), "IR/s path_frac", round(d["roofline"]["path_frac"],3))
PY
done; done
