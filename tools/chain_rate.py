"""The deconvolution + FIR chain alone (bench.py's `deconv_fir` leg): python tools/chain_rate.py [lanes=3] [reps=400] [channels=16]
Meant to be run under `rocprofv3 --kernel-trace --stats` to see what each stage of the chain costs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
est = bench.make_estimator("c2")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rec, L, pitch, _ = bench.synth_recordings(est, B, 0xC2)
import chain_leg  # noqa: E402
out = chain_leg.deconv_fir_leg(0, est, rec, L, pitch, reps=reps, lanes=lanes, paired=os.environ.get("PAIRED", "0") == "1")
print({k: out[k] for k in ("value", "ms_per_measurement", "one_chain", "parity")}, flush=True)
