"""The deconvolution + FIR chain as an experiment harness (tools/chain_rate.py): chains in flight, K1 lanes per chain, CU
masks for the tail context, feeder threads - the knobs behind DESIGN.md's chain measurements.  bench.py times the arrangement
that came out best (three chains, one K1 stream + one tail stream each)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
HBM_PEAK_GBS = 8000.0


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


