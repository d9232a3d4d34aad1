"""bench.ChainTeam alone (the arrangement bench.py times), without the rest of bench.py:
python tools/chain_team_rate.py [chains=3] [blocks=24] [steps=20] [tails=none|own|shared] [channels per call=32]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402
from impulse_hip import Context  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 3
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tails = sys.argv[4] if len(sys.argv) > 4 else "none"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
est = bench.make_estimator("c2")
rec, L, pitch, _ = bench.synth_recordings(est, B, 0xC2)
contexts = [Context(0) for _ in range(chains)]
ctx = contexts[0]
ring = bench.InputRing(ctx, rec, blocks)
team = bench.ChainTeam(contexts, est, np.asarray(est.inverse_filter, dtype=np.float64), ring, L, pitch, B, tails=tails)
for _ in range(3):
    team.step()
team.sync()
t0 = time.perf_counter()
for _ in range(steps):
    team.step()
t_issue = time.perf_counter() - t0
team.sync()
dt = time.perf_counter() - t0
print(f"ChainTeam {chains} chains x {B} channels per call, tails {tails}, {blocks} blocks: {B * blocks * steps / dt / 1e3:.1f} k IR/s, {dt / (blocks * steps) * 1e6:.1f} us per call "
      f"(host issue {t_issue / (blocks * steps) * 1e6:.1f} us per call)", flush=True)
team.release()
ring.release()
for c in reversed(contexts):
    c.close()
