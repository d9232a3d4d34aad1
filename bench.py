#!/usr/bin/env python3
"""bench.py - IRs/s of the batched ESS deconvolution on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (child torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path (K1: deconvolution incl. the 'same' crop) over every input set in
rotation: at the default workload (BASELINE.json configs[1], "C2": 7.1 x 2 ears = 16 channels, 6.15 s sweep
@48 kHz, column L = N + 2 fs = 391 270) that is 40 measurements = 640 IRs per GPU per step, all resident in
HBM before the clock starts (1 GiB of inputs, so no input line survives in the 256 MiB Infinity Cache between
two uses).  Channels shard across ranks with no data-path collective ("weak": one such stream of measurements
per GPU); the only collective is the one-off RCCL broadcast of the prepared inverse-sweep spectrum.

With N > 1 (or --strong) the line also carries `strong_c5`: BASELINE.json configs[4] (1024 channels x 2^20
samples) sharded over the ranks, its IR/s, and the speed-up over rank 0 doing all 1024 channels alone in the
same run (north_star's strong-scaling figure).

The JSON line carries `roofline` (HIP-event time of the dominant kernel over the timed steps, priced in
ALGORITHMIC bytes 8*L per IR) and `cpu_baseline` (the NumPy oracle of the reference's
scipy.signal.convolve(x, inverse_filter, 'same') timed on this box's host cores, rank 0, N=1).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, ROOT)

PITCH_ALIGN = int(os.environ.get("IMPULSE_BENCH_PITCH_ALIGN", "64"))     # samples
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BROADCAST_VIA = None           # set when the in-library RCCL broadcast had to be replaced
METRIC = "impulse responses/sec (sweep deconv+FIR), 7.1×2-ear @48kHz, 1/2/4/8 GPU"

WORKLOADS = {
    # name: (fs, min_duration, channels, description); c2/c3: channels PER RANK per measurement (weak scaling),
    # c4/c5: channels in TOTAL, sharded over the ranks (strong scaling)
    "c2": (48000, 5.0, 16, "C2: 7.1 layout (8 spk x 2 ear = 16 IRs), 6.15 s ESS sweep @48 kHz"),
    "c3": (96000, 5.0, 26, "C3: 13-ch TrueHD layout x 2 ear @96 kHz (deconvolution stage only)"),
    "c4": (48000, None, 256, "C4: synthetic 256-channel batch, 2^20-sample sweeps @48 kHz, channel-sharded"),
    "c5": (48000, None, 1024, "C5: synthetic 1024-channel batch, 2^20-sample sweeps @48 kHz, channel-sharded"),
}
# channels per launch group when groups overlap on 3 lanes (measured sweeps): the
# groups in flight together must still fit the 256 MiB Infinity Cache with their inputs and outputs
GROUP_CHANNELS = {"c2": 32, "c3": 13, "c4": 8, "c5": 8}
# Resident measurements that travel through K1 as ONE launch group (their rows are contiguous in HBM).  At C2 two 7.1
# measurements = 32 channels per group: with three groups in flight the workspaces (96 x 2.2 MB) still fit the 256 MiB
# Infinity Cache, and every launch carries twice the workgroups (tools/k1_rate.py: 16 ch x 3 lanes 413 k, 32 x 3 443 k,
# 48 x 3 359 k IR/s - past 32 the workspaces spill to HBM).  --measurements-per-group 1 is the round-1 shape.
MEASUREMENTS_PER_GROUP = {"c2": 2, "c3": 1, "c4": 1, "c5": 1}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ws-channels", type=int, default=0,
                    help="workspace size in channels (0 = lanes x the per-workload group size)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="independent launch groups in flight (imp_plan_set_overlap); 1 = strictly serial kernels")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect FETCH_SIZE / WRITE_SIZE with rocprofv3 child runs (roofline.traffic then comes from "
                         "the committed profiles/ summary)")
    ap.add_argument("--event-stride", type=int, default=32,
                    help="bracket the passes of every n-th launch group with HIP events (sampling keeps the "
                         "event records from perturbing the throughput being measured)")
    ap.add_argument("--input-sets", type=int, default=0,
                    help="input batches in rotation = measurements per step (0: enough for 1 GiB, at most 40)")
    ap.add_argument("--measurements-per-group", type=int, default=0,
                    help="resident measurements per K1 launch group (default: 2 at C2, 1 elsewhere)")
    ap.add_argument("--strong", action="store_true", help="add the strong_c5 block at N = 1 too")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong_c5 block at N > 1")
    ap.add_argument("--strong-channels", type=int, default=1024)
    ap.add_argument("--strong-passes", type=int, default=8)
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher/collective rehearsal WITHOUT a GPU: ranks rendezvous over gloo, shard, broadcast "
                         "a dummy spectrum and reduce a clock; no compute, value = null (CPU test of the N > 1 plumbing)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# N > 1 invoked plainly: start the ranks as a CHILD torchrun before this process touches HIP or torch.cuda
# (a process that has initialised the GPU must never exec/replace itself on this pool).
# ------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode("utf-8", "replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            line = ln
        elif ln:
            sys.stderr.write(ln + "\n")
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if proc.returncode != 0:
        raise SystemExit(proc.returncode)
    if line is None:
        raise SystemExit("the ranks exited 0 but rank 0 printed no JSON line")
    return 0


def make_estimator(workload):
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    fs, dur, _, _ = WORKLOADS[workload]
    if dur is not None:
        return ImpulseResponseEstimator(min_duration=dur, fs=fs)
    # C4/C5: the phase formula of core/impulse_response_estimator.py:86-147 with
    # L := N := 2^20 and P = 13, entering the way an off-grid WAV does (from_wav :250-254)
    est = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    N, P = 1 << 20, est.n_octaves
    ln2p = np.log(2 ** P)
    n = np.arange(N)
    sig = np.sin(np.pi / 2 ** P * N / ln2p * np.exp(n / N * ln2p))
    m = 2 * int(fs * (N / fs / P) * 0.5)
    sig[: m // 2] *= (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(m) / (m - 1)))[: m // 2]
    est.test_signal = sig
    est.duration = N / fs
    est.inverse_filter = est.generate_inverse_filter()
    return est


def synth_recordings(est, n_channels, seed0, column=None):
    """SURVEY 8(d) recipe: per channel a sparse-tap room (direct sound at 64+37c, three later
    taps) excited by the sweep, plus -70 dBFS noise; fp32, row pitch padded to a multiple of 64 samples (256 B:
    every channel then starts on a cache-line boundary; an odd pitch makes each 512-byte wave access straddle
    an extra 128-byte line, +25 % input lines fetched)."""
    N, fs = len(est), est.fs
    L = N + 2 * fs if column is None else column
    pitch = (L + PITCH_ALIGN - 1) // PITCH_ALIGN * PITCH_ALIGN
    sweep = est.test_signal.astype(np.float32)
    rec = np.zeros((n_channels, pitch), dtype=np.float32)
    delays = []
    for c in range(n_channels):
        rng = np.random.default_rng(seed0 + c)
        d0 = (64 + 37 * c) % 2048
        taps = [(d0, 1.0)]
        for d in rng.integers(100, 24000, size=3):
            taps.append((d0 + int(d), float(0.3 * np.exp(-d / 9600.0) * rng.standard_normal())))
        for d, gain in taps:
            n = min(N, L - d)
            rec[c, d:d + n] += np.float32(0.5 * gain) * sweep[:n]
        rec[c, :L] += (rng.standard_normal(L) * 10 ** (-70 / 20)).astype(np.float32)
        delays.append(d0)
    return rec, L, pitch, delays


def slice_rate(est, rec, L, reps=3):
    """SURVEY 8(d) secondary figure: the whole hot-path slice (ingest K1 -> crop_heads K3/K4 -> crop_tails K7/K4 ->
    EQ curves + FIR design K12/K6 -> equalize K5 -> normalize K2) on ONE 7.1 x 2-ear measurement laid out as a recording
    (2 s lead + one column per speaker): interleaved PCM frames in, float64 host arrays out; the responses stay on the
    device between the stages."""
    from impulse_hip.pipeline_slice import run_slice
    fs = est.fs
    speakers = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]
    tracks = np.zeros((2, 2 * fs + L * 8), dtype=np.float64)
    for i in range(8):
        for ear in range(2):
            tracks[ear, 2 * fs + i * L: 2 * fs + (i + 1) * L] = rec[2 * i + ear, :L]
    # the recording as a capture buffer / WAV data chunk holds it: interleaved 32-bit PCM frames
    frames = np.ascontiguousarray(np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32))
    job = [((fs, frames), speakers)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_slice(est, job)                                   # plans, tables
        t0 = time.perf_counter()
        for _ in range(reps):
            run_slice(est, job)[0].to_host()                  # float64 host arrays out
        dt = (time.perf_counter() - t0) / reps
    return dict(value=16 / dt, unit="IR/s", ms_per_measurement=dt * 1e3,
                note="end to end: PCM frames in host memory -> float64 responses in host memory, incl. PCIe; 16 IRs per measurement; not the headline metric")


def deconv_fir_leg(dev_index, est, rec, L, pitch, reps=300, lanes=3, per_meas=16, paired=False):
    """The metric's "+FIR" on device pointers: 7.1 x 2-ear measurements resident in HBM go through K1 (deconvolution)
    -> K3 (first-peak search) -> K4 (head crop at peak - 1 ms, 0.68 s long, Hann fades, compacted) -> K5 (per-channel
    9 600-tap FIRs whose spectra are cached in the plan) as ONE stream-ordered chain (imp_chain): the crop offsets are taken
    from the peak search on the device, nothing crosses the bus.  `lanes` chains on their own contexts (streams) take the
    measurements round robin so that one measurement's row pass runs beside another's column passes.
    `rec` holds one or more measurements of `per_meas` channels; all of them go through one chain call (two per call is
    the measured optimum, as for K1 alone: 16 ch x 3 chains 291 k, 32 x 3 312 k, 48 x 3 279 k IR/s).
    Algorithmic bytes per IR: 4 L in + 4 (n + K - 1) out."""
    from impulse_hip import Context, ConvPlan
    from impulse_hip._native import FirChain
    from oracle.estimator import estimate
    from oracle.impulse_response import peak_index
    from oracle.scipy_restated import fft_convolve, hann
    fs, B = est.fs, rec.shape[0]
    n, K, head = int(0.68 * fs), 9600, fs // 1000
    fade = 2 * int(fs * (len(est) / fs / est.n_octaves) * (1 / 24)) // 2
    rng = np.random.default_rng(0xF1)
    firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / 400.0) * 0.05
    firs[:, 0] += 1.0
    po = (n + K - 1 + 63) // 64 * 64

    k1_lanes = int(os.environ.get("IMPULSE_BENCH_CHAIN_K1_LANES", "3"))
    n_inputs = int(os.environ.get("IMPULSE_BENCH_CHAIN_INPUTS", "8"))

    class Lane:
        """one chain: K1 on `k1_lanes` streams of one context, the peak search and K5 on the stream of a second context"""

        def __init__(self):
            self.ctx, self.tail = Context(dev_index), Context(dev_index)
            self.plan1 = ConvPlan(self.ctx, np.asarray(est.inverse_filter, dtype=np.float64), L, "same",
                                  ws_channels=B * k1_lanes, paired=paired)
            self.plan1.set_overlap(k1_lanes)
            self.plan5 = ConvPlan(self.tail, firs, n, "full", ws_channels=B)
            self.chain = FirChain(self.plan1, self.plan5, B, head, head, fade)
            self.d_xs = [self.ctx.malloc(B * pitch * 4) for _ in range(n_inputs)]
            self.d_outs = [self.ctx.malloc(B * po * 4) for _ in range(k1_lanes + 2)]
            self.d_pk = self.ctx.malloc(B * 8)
            for d in self.d_xs:
                self.ctx.h2d(d, rec)
            self.k = 0

        def once(self):
            self.chain.execute_device(self.d_xs[self.k % n_inputs], pitch, self.d_outs[self.k % len(self.d_outs)], po, self.d_pk)
            self.last_out = self.d_outs[self.k % len(self.d_outs)]
            self.k += 1

        def synchronize(self):
            self.ctx.synchronize()
            self.tail.synchronize()

        def close(self):
            self.synchronize()
            self.chain.close()
            for p in self.d_xs + self.d_outs + [self.d_pk]:
                self.ctx.free(p)
            self.tail.close()
            self.ctx.close()

    team = [Lane() for _ in range(lanes)]
    try:
        for _ in range(10):
            for ln in team:
                ln.once()
        for ln in team:
            ln.synchronize()
        feeders = os.environ.get("IMPULSE_BENCH_CHAIN_FEEDERS", "0") == "1"
        t0 = time.perf_counter()
        if feeders:                                   # one host thread per chain (ctypes drops the GIL during the calls)
            import threading

            def feed(ln, count):
                for _ in range(count):
                    ln.once()
            ths = [threading.Thread(target=feed, args=(ln, reps // lanes)) for ln in team]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
        else:
            for i in range(reps):
                team[i % lanes].once()
        t_issue = time.perf_counter() - t0
        for ln in team:
            ln.synchronize()
        dt = (time.perf_counter() - t0) / (reps // lanes * lanes)
        sys.stderr.write(f"[chain] host issue time {t_issue / reps * 1e6:.1f} us per call, total {dt * 1e6:.1f} us per call\n")
        t1 = time.perf_counter()
        for _ in range(100):
            team[0].once()
            team[0].synchronize()
        dt_single = (time.perf_counter() - t1) / 100
        y = np.empty((B, po), dtype=np.float32)
        peaks = np.empty(B, dtype=np.int64)
        team[-1].ctx.d2h(y, team[-1].last_out)
        team[-1].ctx.d2h(peaks, team[-1].d_pk)
    finally:
        for ln in team:
            ln.close()
    errs, peaks_ok = [], True
    w = np.ones(n)
    w[:head] *= hann(2 * head)[:head]
    w[n - fade:] *= hann(2 * fade)[fade:]
    for c in (0, B - 1):
        ir = estimate(rec[c, :L].astype(np.float64), np.asarray(est.inverse_filter, dtype=np.float64))
        pk = peak_index(ir)
        peaks_ok &= pk == int(peaks[c])
        s0 = min(max(pk - head, 0), L - n)
        ref = fft_convolve(ir[s0:s0 + n] * w, firs[c], "full")
        errs.append(float(np.max(np.abs(y[c, :n + K - 1] - ref)) / np.max(np.abs(ref))))
    alg = per_meas * (4.0 * L + 4.0 * (n + K - 1))
    mpc = B / per_meas                                            # measurements per chain call
    dt, dt_single = dt / mpc, dt_single / mpc
    return dict(value=per_meas / dt, unit="IR/s", ms_per_measurement=dt * 1e3, channels=per_meas, channels_per_chain_call=B,
                chains_in_flight=lanes,
                one_chain=dict(value=per_meas / dt_single, ms_per_measurement=dt_single * 1e3),
                stages="K1 deconvolution -> K3 first peak -> K4 crop (peak - 1 ms, 0.68 s) + Hann fades -> K5 per-channel "
                       "9 600-tap FIR (spectra cached in the plan): one stream-ordered chain (imp_chain, seven launches) per call of "
                       f"{B} channels, crop offsets taken from the peak search on the device, no host round trip",
                algorithmic_bytes_per_measurement=alg, achieved_GBps=alg / dt / 1e9, frac_of_hbm_peak=alg / dt / 1e9 / HBM_PEAK_GBS,
                parity=dict(peak_indices_exact=bool(peaks_ok), time_max_rel_err=max(errs), tolerance=1e-6, channels_checked=2))


def cpu_baseline(est, rec, L, budget_s=12.0):
    """The oracle's restatement of estimate() (float64, nfft = next_fast_len, rfft(h) recomputed per
    call exactly like core/impulse_response_estimator.py:149-151), serial over channels as the
    reference ingests them (core/hrir.py:307-355).  Bounded sample of the same workload."""
    from oracle.estimator import estimate
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    done, t0 = 0, time.perf_counter()
    outs = {}
    while True:
        c = done % rec.shape[0]
        y = estimate(rec[c, :L].astype(np.float64), inv)
        if c not in outs:
            outs[c] = y
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 512:
            break
    return dict(value=done / el, unit="IR/s", cores=1, kind="port",
                sample=f"{done} IRs of the same synthetic batch, serial float64 NumPy pocketfft "
                       f"(host has {os.cpu_count()} logical cores, {len(os.sched_getaffinity(0))} usable)"), outs


def cpu_pooled(est, rec, L, budget_s=8.0):
    """Same work through a thread pool of min(2*cpu, 32) workers - the reference's
    core/parallel_processing.py:31-48 heuristic (pocketfft releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.estimator import estimate
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    workers = min(2 * (os.cpu_count() or 1), 32)
    xs = [rec[c, :L].astype(np.float64) for c in range(rec.shape[0])]
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as pool:
        while time.perf_counter() - t0 < budget_s:
            list(pool.map(lambda x: estimate(x, inv), xs * max(1, (2 * workers) // len(xs))))
            done += len(xs) * max(1, (2 * workers) // len(xs))
    el = time.perf_counter() - t0
    return dict(value=done / el, unit="IR/s", cores=workers, kind="port", sample=f"{done} IRs, thread pool")


def fp32_fft_floor(est, x, L):
    """Spectrum error of the reference's own FFT backend (pocketfft) run in SINGLE precision on one
    channel: the yardstick for the un-cropped column, where every fp32 transform exceeds 1e-6."""
    try:
        import scipy.fft as sfft
    except ImportError:
        return None
    from oracle.estimator import estimate
    from oracle.scipy_restated import next_fast_len_real
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    M = len(inv)
    nfft = next_fast_len_real(L + M - 1)
    y = sfft.irfft(sfft.rfft(x[:L].astype(np.float32), nfft) * sfft.rfft(inv, nfft).astype(np.complex64), nfft)
    s0 = (M - 1) // 2
    ref = estimate(x[:L].astype(np.float64), inv)
    A, R = np.abs(np.fft.rfft(y[s0:s0 + L].astype(np.float64))), np.abs(np.fft.rfft(ref))
    return float(np.max(np.abs(A - R)) / np.max(R))


def demo_column_error():
    """Whole-column magnitude-spectrum error of THIS path on the one real recording column that ships as a
    fixture (tests/golden/demo_fc.npz: the reference's data/demo/FC.wav, left track, after the 2 s lead) against the
    oracle in float64.  Peak normalised, like every spectrum figure here."""
    path = os.path.join(ROOT, "tests", "golden", "demo_fc.npz")
    if not os.path.exists(path):
        return None
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle.estimator import estimate
    z = np.load(path)
    key = "column_i32" if "column_i32" in z.files else None
    if key is None:
        return None
    col = z[key].astype(np.float64) / 2.0 ** 31
    est = ImpulseResponseEstimator(min_duration=5.0, fs=48000)
    y = est.estimate_batch(col[None, :].astype(np.float32))[0].astype(np.float64)
    ref = estimate(col.astype(np.float32).astype(np.float64), np.asarray(est.inverse_filter, dtype=np.float64))
    A, R = np.abs(np.fft.rfft(y)), np.abs(np.fft.rfft(ref))
    pk = int(np.argmax(np.abs(ref)))
    sl = slice(pk - 48, pk - 48 + int(0.68 * 48000))
    Ac, Rc = np.abs(np.fft.rfft(y[sl])), np.abs(np.fft.rfft(ref[sl]))
    return dict(whole_column_spectrum_max_rel_err=float(np.max(np.abs(A - R)) / np.max(R)),
                cropped_spectrum_max_rel_err=float(np.max(np.abs(Ac - Rc)) / np.max(Rc)),
                peak_index_equal=bool(int(np.argmax(np.abs(y))) == pk), samples=int(len(col)),
                source="tests/golden/demo_fc.npz (reference data/demo/FC.wav, left track, column 0)")


def load_profile_traffic(workload):
    """L2<->fabric bytes per launch from the committed rocprofv3 --pmc summary of this command (profiles/), or
    None.  bench.py cannot collect PMC counters itself; the figure is from an EARLIER profiled run and says so."""
    for name in ("r02_pmc_traffic.json", "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as fh:
                t = json.load(fh).get(workload)
        except (OSError, ValueError):
            continue
        if t:
            return t, "profiles/" + name
    return None, None


def live_pmc_traffic(workload, mpg):
    """FETCH_SIZE and WRITE_SIZE of the three K1 kernels, collected NOW: two child runs of this very script under
    `rocprofv3 --pmc` (one counter per pass, as MI355X_MICROARCH.md prescribes; strictly serial launch groups so a
    counter belongs to one kernel at a time), started after this process has finished its own GPU work.  Returns the
    same dict as load_profile_traffic, or None when rocprofv3 is missing, this process is itself being profiled, or a
    child fails - the caller then falls back to the committed summary and says so."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ)
    if not os.path.exists(rocprof) or profiled:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    out = tempfile.mkdtemp(prefix="impulse_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IMPULSE_BENCH_FORCE_DIST"):
        env.pop(k, None)                                     # the children are plain single-process runs
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(out, counter), "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "1",
                   "--no-cpu-baseline", "--no-pmc", "--lanes", "1", "--no-events", "--input-sets", "4",
                   "--measurements-per-group", str(mpg)]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=60)
            if res.returncode != 0:
                sys.stderr.write(f"[bench] rocprofv3 --pmc {counter} child failed (rc {res.returncode}): {res.stderr[-400:]}\n")
                return None
        pm = pmc_summary.main(out)
        keys = ("rows_kernel", "cols_fwd", "cols_inv")
        if not all(k in pm and "FETCH_SIZE" in pm[k] and "WRITE_SIZE" in pm[k] for k in keys):
            return None
        b = lambda k: (2 * pm[k]["FETCH_SIZE"] + pm[k]["WRITE_SIZE"]) * 1024      # noqa: E731 - gfx950: FETCH_SIZE counts half
        return {"rows_kernel_bytes_per_launch": b("rows_kernel"), "cols_fwd_bytes_per_launch": b("cols_fwd"),
                "cols_inv_bytes_per_launch": b("cols_inv"), "rows_kernel_fetch_kb_raw": pm["rows_kernel"]["FETCH_SIZE"],
                "rows_kernel_write_kb": pm["rows_kernel"]["WRITE_SIZE"]}
    except Exception as exc:                                  # noqa: BLE001 - a reported figure, never fatal
        sys.stderr.write(f"[bench] live PMC collection failed: {exc!r}\n")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


# ------------------------------------------------------------------------------------------------------
def rehearse_launch(args, rank, world):
    """CPU rehearsal of the N > 1 plumbing (no GPU, no compute): rendezvous, sharding, the spectrum broadcast
    helper on dummy bytes, the MAX clock reduction, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from impulse_hip.sharding import broadcast_bytes, shard_channels
    dist.init_process_group("gloo")
    want = torch.from_numpy(np.random.default_rng(7).integers(0, 255, 1 << 16, dtype=np.uint8))
    buf = want.clone() if rank == 0 else torch.zeros_like(want)
    broadcast_bytes(buf, dist, src=0)
    ok = bool(torch.equal(buf, want))
    lo, hi = shard_channels(args.strong_channels, world, rank)
    spans = [None] * world
    dist.all_gather_object(spans, (lo, hi))
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        tiled = spans[0][0] == 0 and spans[-1][1] == args.strong_channels and all(
            spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        line = {"metric": METRIC, "value": None, "unit": "IR/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "rehearsal": "launcher + collectives over gloo, no GPU, no compute",
                "ranks_seen": dist.get_world_size(), "broadcast_ok": bool(flag.item()), "shards_tile": bool(tiled),
                "max_clock_s": float(t.item()), "strong_c5": {"channels": args.strong_channels, "shards": spans}}
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()
    dist.destroy_process_group()
    return 0


def spectrum_broadcast(plan, ctx, dist, torch, device, backend, rank, world, tag):
    """The path's one collective.  With the RCCL backend the LIBRARY does it (imp_comm_* over librccl; the 128-byte
    communicator id is handed round by the launcher's process group) - torch.distributed only provides the launcher's barriers
    and clock reductions; the gloo rehearsal stages the bytes through host memory instead."""
    from impulse_hip.sharding import broadcast_plan_spectrum, broadcast_plan_spectrum_rccl
    if backend != "nccl" or os.environ.get("IMPULSE_BENCH_BCAST", "lib") != "lib":
        return broadcast_plan_spectrum(plan, ctx, dist, torch, device, src=0, via_host=(backend != "nccl"))
    # the communicator id is 128 bytes of control plane: rank 0 makes it, the launcher's process group hands it round
    from impulse_hip._native import comm_unique_id
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    n, err = 0, None
    try:
        if os.environ.get("IMPULSE_BENCH_FAIL_LIB_BCAST") == "1":           # rehearsal of the fallback below
            raise RuntimeError("IMPULSE_BENCH_FAIL_LIB_BCAST")
        n = broadcast_plan_spectrum_rccl(plan, ctx, rank, world, unique_id=box[0])
    except Exception as exc:                                  # noqa: BLE001 - agreed on below, then reported
        err = repr(exc)
    # every rank must take the same road: if the library's communicator failed anywhere (librccl not loadable, say),
    # all ranks repeat the broadcast through the launcher's process group and the line says so
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        sys.stderr.write(f"[bench] rank {rank}: in-library RCCL broadcast failed ({err}); using torch.distributed\n")
        global BROADCAST_VIA
        BROADCAST_VIA = "torch.distributed (the library's own RCCL communicator failed: see stderr)"
        return broadcast_plan_spectrum(plan, ctx, dist, torch, device, src=0, via_host=False)
    dist.barrier()
    return n


class DeviceBatch:
    """A workload resident in HBM: plan, rotating input sets, per-lane output buffers (torch = device memory
    plumbing only)."""

    def __init__(self, torch, device, ctx, plan, rec, L, pitch, M, lanes, n_sets):
        self.torch, self.device, self.ctx, self.plan = torch, device, ctx, plan
        self.B, self.L, self.pitch, self.M = rec.shape[0], L, pitch, M
        self.lanes = max(1, min(lanes, 4, plan.ws_channels))
        plan.set_overlap(self.lanes)
        d_x = torch.from_numpy(rec).to(device)
        self.d_xs = [d_x] + [d_x.clone() for _ in range(n_sets - 1)]
        # output rows: same pitch; the first sample of a row is placed so that the crop offset of the 'same' window
        # lands the stores on 128-byte lines (the caller chooses where results go: here (M-1)/2 mod 32 samples in)
        self.skew = ((M - 1) // 2) % 32 if os.environ.get("IMPULSE_BENCH_OUT_SKEW", "1") == "1" else 0
        # overlapped launch groups must not write the same memory: one output buffer per lane, used round robin
        self.d_ybufs = [torch.empty(self.B * pitch + 64, dtype=torch.float32, device=device) for _ in range(self.lanes)]
        self.d_ys = [b[self.skew: self.skew + self.B * pitch].view(self.B, pitch) for b in self.d_ybufs]
        self.n = 0
        torch.cuda.synchronize(device)

    def measurement(self):
        """one launch-group sequence over one input set (B channels)"""
        out = self.d_ys[self.n % len(self.d_ys)]
        src = self.d_xs[self.n % len(self.d_xs)]
        self.n += 1
        self.plan.execute_device(src.data_ptr(), self.B, self.pitch, out.data_ptr(), self.pitch)

    def step(self):
        for _ in range(len(self.d_xs)):
            self.measurement()

    def sync(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize(self.device)

    def outputs(self):
        return [b.cpu().numpy()[:, :self.L] for b in self.d_ys]

    def release(self):
        self.plan.close()
        self.d_xs = self.d_ys = self.d_ybufs = None


def strong_block(args, torch, dist, device, comm_device, ctx, rank, world, backend):
    """BASELINE.json configs[4]: 1024 channels x 2^20 samples sharded over the ranks (strong scaling), then rank 0
    alone over all of them for the single-GPU figure of the same run."""
    from impulse_hip import ConvPlan
    from impulse_hip.sharding import broadcast_plan_spectrum, shard_channels
    total = args.strong_channels
    est = make_estimator("c5")
    M = len(est)
    # launch groups of 12 channels on two lanes: 24 workspaces of 6.3 MB in flight (tools/k1_rate.py at 2^20 x 2^20:
    # 8 ch x 3 lanes 132 k, 12 x 2 139 k, 12 x 3 126 k, 16 x 2 136 k IR/s)
    grp = 12
    lanes = max(1, min(args.lanes, 2))
    if rank == 0:
        plan = ConvPlan(ctx, np.asarray(est.inverse_filter, dtype=np.float64), M, "same", ws_channels=lanes * grp)
    else:
        plan = ConvPlan(ctx, None, M, "same", ws_channels=lanes * grp, empty_M=M, n_filters=1)
    bcast = 0
    if dist is not None:
        bcast = spectrum_broadcast(plan, ctx, dist, torch, device, backend, rank, world, "c5")
    plan.set_overlap(lanes)
    base, L, pitch, dl = synth_recordings(est, 64, seed0=0xC5, column=M)
    d_base = torch.from_numpy(base).to(device)

    def timed(lo, hi, everyone):
        n = hi - lo
        reps = -(-n // 64)
        # channel c of the 1024-channel batch is recording c % 64 (tiled on the device)
        d_x = torch.roll(d_base, -(lo % 64), 0).repeat(reps, 1)[:n].contiguous()
        skew = ((M - 1) // 2) % 32
        d_ybuf = torch.empty(n * pitch + 64, dtype=torch.float32, device=device)
        d_y = d_ybuf[skew: skew + n * pitch].view(n, pitch)

        def one_pass():
            plan.execute_device(d_x.data_ptr(), n, pitch, d_y.data_ptr(), pitch)

        one_pass()
        ctx.synchronize()
        torch.cuda.synchronize(device)
        if everyone and dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.strong_passes):
            one_pass()
        ctx.synchronize()
        torch.cuda.synchronize(device)
        el = time.perf_counter() - t0
        if everyone and dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        # parity: every channel's peak where the analytic truth puts it (sampled rows come to the host)
        rows = sorted(set(list(range(0, n, max(1, n // 16))) + [n - 1]))
        y = d_y[rows].cpu().numpy()[:, :L]
        ok = all(int(np.argmax(np.abs(y[i]))) == M // 2 + dl[(lo + r) % 64] for i, r in enumerate(rows))
        # tiles bit-equal their twins (same recording 64 channels apart -> same bits whatever launch group / lane)
        if n > 64:
            ok &= bool(torch.equal(d_y[:n - 64, :L], d_y[64:n, :L]))
        del d_x, d_y, d_ybuf
        return el, ok

    lo, hi = shard_channels(total, world, rank)
    el_n, ok = timed(lo, hi, True)
    el_1, ok1 = (None, True)
    if world > 1:
        if rank == 0:
            el_1, ok1 = timed(0, total, False)
        dist.barrier()
    flag_ok = ok and ok1
    if dist is not None:
        flag = torch.tensor([1 if flag_ok else 0], dtype=torch.int32, device=comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        flag_ok = bool(flag.item())
    ranks_seen = dist.get_world_size() if dist is not None else 1
    nfft = plan.nfft
    plan.close()
    out = None
    if rank == 0:
        rate = total * args.strong_passes / el_n
        out = dict(workload=WORKLOADS["c5"][3], channels=total, scaling="strong", n_gpus=world, ranks_seen=ranks_seen,
                   channels_per_rank=hi - lo, passes=args.strong_passes, value=rate, unit="IR/s",
                   ms_per_pass=el_n / args.strong_passes * 1e3, nfft=nfft,
                   path_frac=rate / world * 8.0 * L / 1e9 / HBM_PEAK_GBS,
                   broadcast_bytes=bcast, peaks_exact_and_tiles_bit_equal=bool(flag_ok))
        if el_1 is not None:
            r1 = total * args.strong_passes / el_1
            out["single_gpu_same_run"] = dict(value=r1, unit="IR/s", ms_per_pass=el_1 / args.strong_passes * 1e3,
                                              note="rank 0 alone over all channels while the other ranks wait")
            out["speedup_vs_single_gpu"] = rate / r1
    return out, flag_ok


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)

    # the host driver only supports dmabuf IPC: RCCL between processes needs this before anything initialises HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_launch:
        return rehearse_launch(args, rank, world)

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 at communicator
    # creation, so everything the run writes to stdout is sent to stderr and the JSON line goes to the
    # real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    dist = None
    # IMPULSE_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks: ranks are
    # folded onto the visible devices and the spectrum broadcast is staged through host memory.
    backend = os.environ.get("IMPULSE_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    # IMPULSE_BENCH_FORCE_DIST=1 takes the collective path with a single rank too (a one-GPU box can
    # then exercise RCCL init, the spectrum broadcast and the reductions)
    if world > 1 or os.environ.get("IMPULSE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")

    from impulse_hip import Context, ConvPlan
    from impulse_hip.sharding import broadcast_plan_spectrum, shard_channels

    fs, dur, B, desc = WORKLOADS[args.workload]
    est = make_estimator(args.workload)
    strong = args.workload in ("c4", "c5")
    mpg, B_meas = 1, B
    if strong:
        lo, hi = shard_channels(B, world, rank)
        total_channels, B = B, hi - lo
        # 64 distinct recordings tiled over the shard keep host set-up short; L = N = 2^20
        base, L, pitch, dl = synth_recordings(est, min(B, 64), seed0=0xC5 + lo, column=len(est))
        reps = -(-B // base.shape[0])
        rec = np.tile(base, (reps, 1))[:B]
        delays = (dl * reps)[:B]
    else:
        mpg = max(1, args.measurements_per_group or MEASUREMENTS_PER_GROUP[args.workload])
        B_meas, B = B, B * mpg                       # a resident block = mpg measurements, rows contiguous
        rec, L, pitch, delays = synth_recordings(est, B, seed0=0xC2 + 1000 * rank)
        total_channels = B * world
    M = len(est)

    ctx = Context(dev_index)
    # C2: a resident block (mpg measurements) is one launch group; C3 cuts its 26 channels into two groups of 13
    # (tools/k1_rate.py at C3's sizes: 9 ch x 3 lanes 185 k, 13 x 3 198 k, 26 x 2 204 k IR/s)
    group_channels = B if args.workload == "c2" else GROUP_CHANNELS[args.workload]
    ws_channels = args.ws_channels or max(1, args.lanes) * group_channels
    if rank == 0:
        plan = ConvPlan(ctx, np.asarray(est.inverse_filter, dtype=np.float64), L, "same", ws_channels=ws_channels)
    else:
        plan = ConvPlan(ctx, None, L, "same", ws_channels=ws_channels, empty_M=M, n_filters=1)
    bcast_bytes = 0
    if dist is not None:
        bcast_bytes = spectrum_broadcast(plan, ctx, dist, torch, device, backend, rank, world, "c2")

    # Successive measurements read DIFFERENT copies of the batch, 1 GiB in rotation, so that no input line survives in
    # the 256 MiB Infinity Cache from one use to the next: inputs come from HBM, as a stream of new measurements would.
    # (Re-reading one 25 MB batch every time measured 9 % higher at C2: 429 k vs 391-395 k IR/s.)
    batch_bytes = rec.size * 4
    n_sets = args.input_sets or int(os.environ.get("IMPULSE_BENCH_INPUT_SETS", "0")) or \
        max(1, min(40, -(-(1 << 30) // batch_bytes)))
    wl = DeviceBatch(torch, device, ctx, plan, rec, L, pitch, M, args.lanes, n_sets)
    lanes = wl.lanes

    def barrier():
        wl.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    groups_per_measurement = -(-B // (plan.ws_channels // lanes))
    groups_timed = args.steps * n_sets * groups_per_measurement
    # at least ~8 sampled launch groups however short the run
    plan.set_timing(0 if args.no_events else max(1, min(args.event_stride, groups_timed // 8)))
    plan.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    wl.sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    kernel_ms, launches = plan.get_timing(reset=True)
    plan.set_timing(0)
    # the same kernels with nothing else on the chip (strictly serial launch groups), outside the timed
    # region: under overlap a kernel's event-to-event time includes the share of the chip it cedes to the
    # other groups in flight, so both views are reported
    iso_ms, iso_n = kernel_ms, launches
    if lanes > 1:
        barrier()
        plan.set_overlap(1)
        plan.set_timing(1)
        for _ in range(max(4, min(40, n_sets))):
            wl.measurement()
        ctx.synchronize()
        iso_ms, iso_n = plan.get_timing(reset=True)
        plan.set_timing(0)
        plan.set_overlap(lanes)

    # parity gate on what the timed loop produced (outside the timed region)
    peaks_ok = True
    ys = wl.outputs()
    for y in ys:
        peaks_ok &= all(int(np.argmax(np.abs(y[c]))) == M // 2 + delays[c] for c in range(B))
    y = ys[-1]
    if dist is not None:                      # rank 0 reports the verdict of every rank
        flag = torch.tensor([1 if peaks_ok else 0], dtype=torch.int32, device=comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        peaks_ok = bool(flag.item())
    ranks_seen = dist.get_world_size() if dist is not None else 1
    nfft = plan.nfft
    plan_ws = plan.ws_channels
    wl.release()

    strong_c5 = None
    if (world > 1 and not args.no_strong) or args.strong:
        strong_c5, s_ok = strong_block(args, torch, dist, device, comm_device, ctx, rank, world, backend)
        peaks_ok &= s_ok

    result = None
    if rank == 0:
        irs_per_step = total_channels * n_sets
        value = irs_per_step * args.steps / elapsed
        alg_bytes_per_launch = 8.0 * L * B / groups_per_measurement          # average over this rank's launch groups
        names = ("cols_kernel<fwd> (pass A)", "rows_kernel (pass B)", "cols_kernel<inv> (pass C)")
        roof = None
        if launches > 0:
            avg_ms = [m / launches for m in kernel_ms]
            dom = int(np.argmax(avg_ms))
            achieved = alg_bytes_per_launch / (avg_ms[dom] * 1e-3) / 1e9
            iso_avg = [m / max(iso_n, 1) for m in iso_ms]
            iso_achieved = alg_bytes_per_launch / (iso_avg[dom] * 1e-3) / 1e9
            prof, prof_src, live = None, None, False
            if world == 1 and not args.no_pmc and not strong:
                prof = live_pmc_traffic(args.workload, mpg)
                live = prof is not None
                prof_src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command, made by this run"
            if prof is None:
                prof, prof_src = load_profile_traffic(args.workload)
            roof = dict(bound="hbm", kernel=names[dom], achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS,
                        traffic=(prof or {}).get("rows_kernel_bytes_per_launch"),
                        traffic_source=("live" if live else "committed summary") if prof else None,
                        traffic_note=(f"L2<->fabric bytes per launch of the dominant kernel (FETCH_SIZE x2 + WRITE_SIZE, "
                                      f"Infinity-Cache hits INCLUDED, so not HBM bytes); separate --pmc passes, strictly serial "
                                      f"launch groups; source: {prof_src}") if prof else None,
                        avg_kernel_ms=dict(zip(("pass_a", "pass_b", "pass_c"), avg_ms)),
                        events_sampled=int(launches), launch_groups_in_flight=lanes,
                        note="achieved/frac: algorithmic bytes of one launch group / HIP-event time of the dominant kernel "
                             "over the timed region; with several launch groups in flight that time includes the share of "
                             "the chip the kernel cedes to the others, so `isolated` (strictly serial groups) and "
                             "`path_frac` (whole path, all kernels) are the cleaner figures",
                        isolated=dict(note="same kernel, launch groups strictly serial (nothing else on the chip), "
                                           "measured after the timed region", achieved=iso_achieved,
                                      frac=iso_achieved / HBM_PEAK_GBS,
                                      avg_kernel_ms=dict(zip(("pass_a", "pass_b", "pass_c"), iso_avg))),
                        algorithmic_bytes_per_launch=alg_bytes_per_launch,
                        launch_groups_per_step=groups_per_measurement * n_sets,
                        path_achieved=value / world * 8.0 * L / 1e9,
                        path_frac=value / world * 8.0 * L / 1e9 / HBM_PEAK_GBS)
            if prof and all(k in prof for k in ("rows_kernel_bytes_per_launch", "cols_fwd_bytes_per_launch",
                                                "cols_inv_bytes_per_launch")):
                moved = sum(prof[k] for k in ("rows_kernel_bytes_per_launch", "cols_fwd_bytes_per_launch",
                                              "cols_inv_bytes_per_launch"))
                rate = moved * groups_per_measurement * n_sets / (elapsed / args.steps) / 1e9
                roof["l2_fabric_traffic"] = dict(
                    bytes_per_launch_group=moved, rate=rate, unit="GB/s", source=prof_src,
                    note="bytes passes A+B+C move across the L2<->fabric boundary per launch group (rocprofv3 --pmc, see "
                         "traffic_note) / this run's time per launch group.  Infinity-Cache hits are counted, so this is a "
                         "FABRIC rate, not achieved HBM bandwidth; it is not compared with the HBM peak")
        cpu = None
        parity = dict(peak_indices_exact=bool(peaks_ok))
        if world == 1 and not args.no_cpu_baseline:
            cpu, outs = cpu_baseline(est, rec, L)
            errs, errs_full = [], []
            for c, ref in outs.items():
                pk = int(np.argmax(np.abs(ref)))
                peaks_ok &= pk == int(np.argmax(np.abs(y[c])))
                for sl, acc in ((slice(pk - fs // 1000, pk - fs // 1000 + int(0.68 * fs)), errs), (slice(None), errs_full)):
                    A, R = np.abs(np.fft.rfft(y[c][sl].astype(np.float64))), np.abs(np.fft.rfft(ref[sl]))
                    acc.append(float(np.max(np.abs(A - R)) / np.max(R)))
            floor = fp32_fft_floor(est, rec[0], L)
            parity = dict(peak_indices_exact=bool(peaks_ok), spectrum_max_rel_err=max(errs), tolerance=1e-6,
                          spectrum_window="IR cropped as the pipeline does before any magnitude_response: peak - 1 ms, 0.68 s long",
                          whole_column_spectrum_max_rel_err=max(errs_full),
                          whole_column_meets_1e_6=bool(max(errs_full) <= 1e-6),
                          whole_column_pocketfft_fp32_err=floor,
                          whole_column_note=f"un-cropped {L}-sample column: above 1e-6 for every fp32 transform (the "
                                            "reference's own pocketfft in single precision is listed beside it); reported, "
                                            "not gated",
                          channels_checked=len(errs))
            try:
                parity["real_demo_column"] = demo_column_error()
            except Exception as exc:                          # noqa: BLE001 - reported figure only
                parity["real_demo_column"] = dict(error=repr(exc))
            peaks_ok &= max(errs) <= 1e-6
            cpu["pooled"] = cpu_pooled(est, rec, L)
        whole_slice = None
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline:
            try:
                whole_slice = slice_rate(est, rec[:B_meas], L)
            except Exception as exc:                          # noqa: BLE001 - secondary figure only
                whole_slice = dict(error=repr(exc))
        fir_leg = None
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline:
            try:
                fir_leg = deconv_fir_leg(dev_index, est, rec, L, pitch, per_meas=B_meas)
                peaks_ok &= fir_leg["parity"]["peak_indices_exact"] and fir_leg["parity"]["time_max_rel_err"] <= 1e-6
            except Exception as exc:                          # noqa: BLE001 - secondary figure only
                fir_leg = dict(error=repr(exc))
        result = {
            "metric": METRIC, "value": value, "unit": "IR/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timed_region_s": elapsed, "irs_per_step": irs_per_step, "ranks_seen": ranks_seen,
            "config": {"workload": desc, "stage": "K1 ONLY: batched sweep deconvolution incl. 'same' crop "
                       "(inverse-filter spectrum prepared once, outside the timed region); the FIR stages are not in the "
                       "timed region (`deconv_fir` = K1 -> peak -> crop -> K5 FIR on device pointers; `slice` = the whole hot-path slice end to end)",
                       "step": f"one pass over {n_sets * mpg} resident measurements of {B_meas} channels per GPU, "
                               f"{mpg} measurement(s) = {B} channels per K1 launch group",
                       "channels_per_gpu_per_measurement": B_meas, "measurements_per_step": n_sets * mpg,
                       "measurements_per_launch_group": mpg, "channels_per_launch_group": B // groups_per_measurement,
                       "sweep_samples": M, "column_samples": L,
                       "layout": f"planar fp32, row pitch {pitch} samples (multiple of {PITCH_ALIGN}); output rows start "
                                 f"{wl.skew} samples into 256-byte aligned buffers so the cropped stores fall on cache lines",
                       "nfft": nfft, "launch_groups_in_flight": lanes, "workspace_channels": plan_ws,
                       "sharding": (f"channels x{world}, no data-path collective; "
                                    f"one {(BROADCAST_VIA or 'RCCL (by libimpulse_hip, no torch in the data path)') if backend == 'nccl' else backend + ' (rehearsal)'} broadcast of "
                                    f"{bcast_bytes} B spectrum at plan creation") if dist is not None else
                                   "single rank: no collective"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "slice": whole_slice, "deconv_fir": fir_leg,
            "strong_c5": strong_c5,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not peaks_ok:
        raise SystemExit("parity gate failed: deconvolved peak indices / cropped spectra do not match the truth")
    return 0


if __name__ == "__main__":
    sys.exit(main())
