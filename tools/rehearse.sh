#!/bin/bash
# End-of-round rehearsal on a one-GPU box: two ranks of bench.py folded onto device 0 (IMPULSE_BENCH_BACKEND=gloo: the
# rehearsal mode for boxes with fewer GPUs than ranks), smoke(), the knee-search tests.
# bash tools/rehearse.sh   (outputs under gpurun_out/)
set -o pipefail
mkdir -p gpurun_out
IMPULSE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --blocks 24 > gpurun_out/rehearse_2rank.json 2> gpurun_out/rehearse_2rank.err || { tail -c 800 gpurun_out/rehearse_2rank.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/rehearse_2rank.json"))
print({k: d[k] for k in ("value", "n_gpus", "ranks_seen", "rccl_ranks_seen", "scaling")}, d.get("strong_c5"))
PY
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
timeout -k 10 300 python -m pytest tests/test_device_knees.py -m gpu -q 2>&1 | tail -2
