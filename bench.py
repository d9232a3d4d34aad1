#!/usr/bin/env python3
"""bench.py - IRs/s of the sweep deconvolution + FIR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (child torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over every measurement resident in HBM.  The hot path is the metric's own wording,
"sweep deconv+FIR": per 7.1 x 2-ear measurement (BASELINE.json configs[1], "C2": 16 channels, 6.15 s sweep @48 kHz,
column L = N + 2 fs = 391 270 samples)

    K1 deconvolution (estimate(): recording (*) inverse filter, 'same')      core/impulse_response_estimator.py:149-151
 -> K3 first significant peak                                                core/impulse_response.py:32-70
 -> K4 crop at peak - 1 ms, 0.68 s long, Hann fades                          core/hrir.py:548-653
 -> K5 per-channel 9 600-tap minimum-phase FIR, 'full'                       core/impulse_response.py:110-119

as ONE device chain per call (imp_chain: five launches, nothing crosses the bus, crop offsets taken from the peak search
on the device).  `value` is that chain; `deconv_only` (K1 alone, last round's headline) is given beside it with the
roofline of its dominant kernel.  The default step holds 320 blocks of two measurements = 10 240 IRs per GPU (15 GiB of
inputs in rotation, so no input line survives in the 256 MiB Infinity Cache between two uses, and 20 steps time ~0.5 s).
Channels shard across ranks with no data-path collective ("weak": one stream of measurements per GPU); the only
collective is the one-off RCCL broadcast of the prepared inverse-sweep spectrum.

With N > 1 (or --strong) the line also carries `strong_c5`: BASELINE.json configs[4] (1024 channels x 2^20 samples)
sharded over the ranks, its IR/s, and the speed-up over rank 0 doing all 1024 channels alone in the same run.

The JSON line carries `roofline` (HIP-event time of the dominant kernel over the timed steps, priced in ALGORITHMIC
bytes 8 L per IR, L2<->fabric bytes from rocprofv3 --pmc child runs) and `cpu_baseline` (the NumPy oracle of the same
chain timed on this box's host cores, rank 0, N = 1).  No torch at N = 1: device memory comes from the library.
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
# the most ANY streaming kernel moves across the L2<->fabric boundary with in-place read + write traffic on an Infinity-Cache
# resident footprint (profiles/r02_stream_probe2.txt: 5.1 - 6.1 TB/s over grids and loads in flight, best 6.07; K1's workspace
# round trips are that pattern, its first read and last write go to HBM and are slower)
FABRIC_CEILING_GBS = 6070.0
# gate on the whole un-cropped column's magnitude spectrum (tests/test_hip_parity.py FULL_COLUMN_TOL): the fp32 floor of
# these column lengths is 1.1 - 2.6e-6 for any single-precision transform
WHOLE_COLUMN_TOL = 3e-6
WHOLE_COLUMN_SLACK = 1.25
BROADCAST_VIA = None           # set when the in-library RCCL broadcast had to be replaced
RCCL_RANKS_SEEN = None         # ranks the library's own communicator counted (ncclCommCount), set by spectrum_broadcast
RCCL_COMM = None               # the rank's communicator (impulse_hip._native.Comm), made at the first broadcast
METRIC = "impulse responses/sec (sweep deconv+FIR), 7.1×2-ear @48kHz, 1/2/4/8 GPU"

WORKLOADS = {
    # name: (fs, min_duration, channels, description); c2/c3: channels PER RANK per measurement (weak scaling),
    # c4/c5: channels in TOTAL, sharded over the ranks (strong scaling)
    "c2": (48000, 5.0, 16, "C2: 7.1 layout (8 spk x 2 ear = 16 IRs), 6.15 s ESS sweep @48 kHz"),
    "c3": (96000, 5.0, 26, "C3: 13-ch TrueHD layout x 2 ear @96 kHz"),
    "c4": (48000, None, 256, "C4: synthetic 256-channel batch, 2^20-sample sweeps @48 kHz, channel-sharded (deconvolution only)"),
    "c5": (48000, None, 1024, "C5: synthetic 1024-channel batch, 2^20-sample sweeps @48 kHz, channel-sharded (deconvolution only)"),
}
# channels per K1 launch group (measured sweeps): the groups in flight together must still fit the
# 256 MiB Infinity Cache with their inputs and outputs.  C2: two measurements = 32 channels; C3: 13
GROUP_CHANNELS = {"c2": 32, "c3": 13, "c4": 8, "c5": 8}
MEASUREMENTS_PER_BLOCK = {"c2": 2, "c3": 1}
DEFAULT_BLOCKS = {"c2": 320, "c3": 96, "c4": 12, "c5": 12}
# calls (chains / K1 launch groups) in flight = streams, measured per workload (IMPULSE_BENCH_GROUP / --lanes; C3: 2 streams
# x 13-channel groups 194 k IR/s K1 and 154 k chain, 3 x 13: 197 k / 135 k, 3 x 26: 160 k / 135 k; C5, tools/c5_group_sweep.sh:
# 8-channel groups x 2 streams 138 k IR/s, x 3: 130 k, x 1: 111 k, x 4: 119 k; groups of 4 / 6 / 12 / 16 / 32 at their best
# 131 / 132 / 129 / 126 / 123 k - three 6.3 MB workspaces per channel in flight no longer fit beside inputs and outputs)
# (round 4, C3 with the pair plan, 26-channel calls: 1 / 2 / 3 / 4 chains = 111 / 118 / 123 / 111 k IR/s chain, 156 / 177 / 183 / 151 k K1)
CHAINS = {"c2": 3, "c3": 3, "c4": 2, "c5": 2}
# K1's plan kind of the headline legs: pair mode (two ears per complex transform: what the classes take for ear pairs) where the
# one-channel-per-transform plan's even/odd unpack costs accuracy on the un-cropped column (C3: 4.1e-6 against 2.3e-6, C5:
# 3.7e-6 against 3.0e-6); at C2 both kinds sit inside the fp32 floor and the mono plan is 4 % faster.
# The other kind runs as the secondary leg.  IMPULSE_BENCH_K1_PLAN=mono|pair overrides.
K1_PAIRED = {"c2": False, "c3": True, "c4": False, "c5": True}


def k1_paired(workload):
    env = os.environ.get("IMPULSE_BENCH_K1_PLAN", "")
    return {"mono": False, "pair": True}.get(env, K1_PAIRED[workload])


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0,
                    help="chains / K1 launch groups in flight (0: 3 at C2, 2 at C3 - C5); 1 = strictly serial kernels")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect FETCH_SIZE / WRITE_SIZE with rocprofv3 child runs (roofline.traffic then comes from "
                         "the committed profiles/ summary)")
    ap.add_argument("--event-stride", type=int, default=32,
                    help="bracket the passes of every n-th K1 launch group with HIP events (sampling keeps the event "
                         "records from perturbing the throughput being measured)")
    ap.add_argument("--blocks", type=int, default=0,
                    help="resident input blocks per step (0: 320 at C2 = 15 GiB; a block = the measurements of one chain call)")
    ap.add_argument("--no-slice", action="store_true", help="skip the slice / slice_resident blocks")
    ap.add_argument("--stage", default="chain", choices=["chain", "deconv", "slice"],
                    help="what `value` times: the deconvolution + FIR chain (the metric) or K1 alone (C4 / C5 always K1)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
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


def fir_stage_shape(est):
    """the FIR stage of the chain at this sampling rate: crop length (0.68 s), taps (fs / 5: 9 600 @48 k, 19 200 @96 k - the
    minimum-phase FIR length the reference designs, autoeq/frequency_response.py:637-681), head (1 ms), fade-out length"""
    fs = est.fs
    n, K, head = int(0.68 * fs), fs // 5, fs // 1000
    fade = 2 * int(fs * (len(est) / fs / est.n_octaves) * (1 / 24)) // 2
    return n, K, head, fade


def synth_firs(B, K, seed=0xF1):
    rng = np.random.default_rng(seed)
    firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / 400.0) * 0.05
    firs[:, 0] += 1.0
    return firs


def slice_rate(est, rec, L, reps=24, workers=None):
    """SURVEY 8(d) secondary figure: the whole hot-path slice end to end over a job of `reps` measurements (7.1 x 2 ears each,
    one recording file per measurement: interleaved PCM frames in host memory) to float64 responses in host memory:
    ingest K1 -> crop_heads K3/K4 -> crop_tails K7c/K4 -> equalize K5 -> normalize K2, the stage sequence of every
    measurement device resident (imp_slice), `workers` host threads with a stream each so that the upload of measurement
    i + 1 overlaps the compute of measurement i; the FIRs are designed once per job (K12 -> K6), inside the timed region."""
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.parallel_workers import process_equalization_batch
    from impulse_hip.resident_slice import Layout, SlicePipeline, SliceRunner
    from impulse_hip.pipeline_slice import run_slice
    fs = est.fs
    speakers = SLICE_SPEAKERS["c2"][:rec.shape[0] // 2]
    frames = measurement_frames(est, rec, L, speakers)
    layout = Layout(est, [(frames.shape[0], 2, speakers)])
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    target = FrequencyResponse(name="target", frequency=common.copy(), raw=0)

    # the runner lives across jobs: three stages (upload | compute | download), or `workers` lanes of one measurement each
    runner = SlicePipeline(est, layout) if workers is None else SliceRunner(est, layout, workers=workers)

    def job(n, to_host=True):
        firs = {(sp, sd): fir for sp, sd, fir in process_equalization_batch(layout.tasks, None, None, None, None, None, target, common, fs,
                                                                            on_device=True)}
        return runner.run([[frames]] * n, firs, to_host=to_host, align=True), firs

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        job(reps)                                             # plans, tables, the allocator's steady state
        jobs_ms, dt, lane_ms, res, firs = [], None, None, None, None
        for _ in range(3):                                    # host memory effects move this figure: three jobs, the best is reported
            runner.times()
            t0 = time.perf_counter()
            r_, f_ = job(reps)
            d_ = (time.perf_counter() - t0) / reps
            jobs_ms.append(d_ * 1e3)
            if dt is None or d_ < dt:
                dt, res, firs = d_, r_, f_
                lane_ms = {k: v / reps * 1e3 for k, v in runner.times().items() if k != "measurements"}
            del r_
        # the same job with the recordings in page-locked buffers the runner handed out (a reader that fills frame_buffers()
        # instead of its own array): no staging copy before the link
        pinned_ms = None
        if workers is None:
            bufs = [runner.frame_buffers() for _ in range(4)]
            for b in bufs:
                b[0][...] = frames
            for _ in range(2):
                t0 = time.perf_counter()
                r_ = runner.run([bufs[i % 4] for i in range(reps)], firs, align=True)
                d_ = (time.perf_counter() - t0) / reps * 1e3
                pinned_ms = d_ if pinned_ms is None else min(pinned_ms, d_)
                del r_
            del bufs
        # the same measurement through the staged class path (one host readback per stage): identical samples
        hrir, gain = run_slice(est, [((fs, frames), speakers)], firs=firs, align=True)
        same = all(np.array_equal(res[k][0].irs[sp][sd].data, hrir.irs[sp][sd].data) for k in (0, reps - 1) for sp in speakers
                   for sd in ("left", "right"))
        t0 = time.perf_counter()
        for _ in range(4):
            run_slice(est, [((fs, frames), speakers)], firs=firs, align=True)[0].to_host()
        staged_ms = (time.perf_counter() - t0) / 4 * 1e3
        runner.close()
    return dict(value=16 / dt, unit="IR/s", ms_per_measurement=dt * 1e3, measurements=reps, workers=workers,
                ms_per_measurement_of_each_job=jobs_ms,
                identical_to_staged_path=bool(same), staged_path_ms_per_measurement=staged_ms, lane_ms_per_measurement=lane_ms,
                runner="pipeline" if workers is None else "lanes", pinned_result_blocks=runner.pool.allocations,
                recordings_in_page_locked_buffers=None if pinned_ms is None else dict(
                    value=16 / pinned_ms * 1e3, unit="IR/s", ms_per_measurement=pinned_ms,
                    note="the same job with every recording in a buffer from SlicePipeline.frame_buffers() (what a WAV reader "
                         "would fill): the upload needs no staging copy; FIRs reused from the job above"),
                pcie_bytes_per_measurement=int(frames.nbytes),
                note="end to end: PCM frames in host memory -> float64 responses in host memory, incl. PCIe "
                     f"({frames.nbytes / 1e6:.1f} MB up per measurement) and the once-per-job FIR design; "
                     + ("three stages - upload | compute (imp_slice, one measurement per call) | download of the float64 rows into "
                        "recycled page-locked memory - a host thread and a stream each; " if workers is None else
                        f"{workers} host threads, a stream and a one-measurement resident slice each: upload i + 1 overlaps compute i; ")
                     + "staged_path_ms_per_measurement = the class path with a host readback per stage, one measurement after "
                     "the other (FIRs given); not the headline metric")


def measurement_frames(est, rec, L, speakers):
    """bench.py's synthetic channels laid out as ONE binaural recording file: 2 s lead + one column per speaker, the two
    ears as the two tracks, interleaved 32-bit PCM frames [n_frames, 2] (what a capture buffer or a WAV data chunk holds)"""
    fs = est.fs
    tracks = np.zeros((2, 2 * fs + L * len(speakers)), dtype=np.float64)
    for i in range(len(speakers)):
        for ear in range(2):
            tracks[ear, 2 * fs + i * L: 2 * fs + (i + 1) * L] = rec[2 * i + ear, :L]
    return np.ascontiguousarray(np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32))


SLICE_SPEAKERS = {"c2": ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"],
                  "c3": ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "TFL", "TFR", "TSL", "TSR", "TBL", "TBR"]}


class SliceTeam:
    """The reference's stage sequence for M measurements per call, device resident (imp_slice): `n_streams` slices, each on
    a context (stream) of its own, fed round robin by one host thread; the recordings of a call sit in HBM (a ring of
    `ring` call blocks per stream, so that a call's inputs were last touched 2+ calls ago)."""

    def __init__(self, est, rec, L, n_streams=2, M=8, ring=3, speakers=None, quiet_setup=False, align=True):
        from impulse_hip import Context
        from impulse_hip._native import using_context
        from impulse_hip.resident_slice import Layout, ResidentSlice
        self.est, self.M = est, M
        n_spk = rec.shape[0] // 2
        speakers = speakers or (SLICE_SPEAKERS["c2"] if n_spk == 8 else SLICE_SPEAKERS["c3"])[:n_spk]
        self.speakers = speakers
        self.frames = measurement_frames(est, rec, L, speakers)
        self.layout = Layout(est, [(self.frames.shape[0], 2, speakers)])
        packed = self.layout.pack([self.frames])
        self.lanes = []
        self.firs = None
        for i in range(n_streams):
            ctx = Context(0)
            with using_context(ctx):
                rs = ResidentSlice(est, self.layout, max_measurements=M)
            if self.firs is None:
                self.firs = synth_firs(len(self.layout.tasks), rs.taps)
            rs.set_firs(self.firs)
            rs.set_alignment(align)
            d_in = []
            for _ in range(ring):
                p = ctx.malloc(M * packed.nbytes)
                for m in range(M):
                    ctx.h2d(p + m * packed.nbytes, packed)
                d_in.append(p)
            self.lanes.append(dict(ctx=ctx, rs=rs, d_in=d_in, d_out=None, k=0))
        self.rows = self.lanes[0]["rs"].slice.rows
        # size every slice for the knees these recordings have (first call: flagged KEEP_CAP with the knees, grown once)
        for ln in self.lanes:
            self._alloc_out(ln)
            ln["rs"].slice.execute_device(ln["d_in"][0], self.layout.samples, M, ln["d_out"], ln["rs"].out_pitch)
            rows, meas = ln["rs"].slice.results()
            if np.any(meas["flags"] & 4) and ln["rs"].grow_for(rows):
                self._alloc_out(ln)

    def _alloc_out(self, ln):
        if ln["d_out"]:
            ln["ctx"].free(ln["d_out"])
        ln["d_out"] = ln["ctx"].malloc(self.M * self.rows * ln["rs"].out_pitch * 4)

    def describe(self):
        rs = self.lanes[0]["rs"]
        return (f"{len(self.lanes)} streams x {self.M} measurements per call, {self.rows} rows each, column {self.layout.column_len}, "
                f"keep_cap {rs.keep_cap}, taps {rs.taps}, normalisation transform {rs.slice.norm_fft_len} points, "
                f"{'pair' if rs.plan.paired else 'mono'} plan of {rs.plan.n1} rows, alignment {'on' if rs.align else 'off'}")

    def step(self, design=False):
        """one call per lane; design: the call's FIRs are designed first (K12 -> K6 on the lane's stream, flat target: the
        curves belong to a job and a call is a job of M measurements) and handed to the slice on the device"""
        for ln in self.lanes:
            if design:
                from impulse_hip._native import using_context
                from impulse_hip.parallel_workers import process_equalization_batch
                with using_context(ln["ctx"]):
                    firs = process_equalization_batch(self.layout.tasks, None, None, None, None, None, self._target(), self._common,
                                                      self.est.fs, on_device=True)
                ln["rs"].set_firs({(sp, sd): f for sp, sd, f in firs})
            ln["rs"].slice.execute_device(ln["d_in"][ln["k"] % len(ln["d_in"])], self.layout.samples, self.M, ln["d_out"],
                                          ln["rs"].out_pitch)
            ln["k"] += 1

    def _target(self):
        from impulse_hip.frequency_response import FrequencyResponse
        if getattr(self, "_tgt", None) is None:
            self._common = FrequencyResponse.generate_frequencies(f_min=10, f_max=self.est.fs / 2, f_step=1.01)
            self._tgt = FrequencyResponse(name="target", frequency=self._common.copy(), raw=0)
        return self._tgt

    def sync(self):
        for ln in self.lanes:
            ln["ctx"].synchronize()

    def flags(self):
        return sorted({int(f) for ln in self.lanes for f in ln["rs"].slice.results()[1]["flags"]})

    def results(self, lane=0):
        return self.lanes[lane]["rs"].slice.results()

    def fetch(self, lane=0):
        ln = self.lanes[lane]
        out = np.empty((self.M * self.rows, ln["rs"].out_pitch), dtype=np.float32)
        ln["ctx"].synchronize()
        ln["ctx"].d2h(out, ln["d_out"])
        return out

    def release(self):
        for ln in self.lanes:
            ln["rs"].close()
            ln["ctx"].close()
        self.lanes = []


# ------------------------------------------------------------------------------------------------------
# resident inputs and the two timed arrangements
# ------------------------------------------------------------------------------------------------------
class InputRing:
    """n_blocks copies of one block of recordings ([B][pitch] fp32) at different addresses of HBM (library memory)."""

    def __init__(self, ctx, rec, n_blocks):
        self.ctx, self.nbytes = ctx, rec.nbytes
        self.ptrs = [ctx.malloc(rec.nbytes) for _ in range(n_blocks)]
        ctx.h2d(self.ptrs[0], rec)
        for p in self.ptrs[1:]:
            ctx.d2d(p, self.ptrs[0], rec.nbytes)
        ctx.synchronize()

    def release(self):
        for p in self.ptrs:
            self.ctx.free(p)
        self.ptrs = []


def copy_spectrum(dst_plan, src_plan, ctx):
    """the prepared filter spectrum of one plan into another plan of the same geometry on the same device"""
    (d, n), (s, m) = dst_plan.spectrum_buffer(), src_plan.spectrum_buffer()
    assert n == m
    ctx.d2d(d, s, n)
    ctx.synchronize()


class ChainTeam:
    """One deconvolution + FIR chain per context, fed round robin with the blocks of the input ring.  Each chain is five
    launches in ONE stream (K1's three passes, the peak search, the fused K5): three chains = three streams, fewer than the
    four hardware queues a HIP process gets, so none of them is multiplexed with another.  (A tail stream per chain for the
    peak search and K5 - tails="own" / "shared", the library supports it - measured 15 % slower: 320 k against 375 k IR/s;
    the event edges between streams cost more than the overlap buys.  tools/chain_team_rate.py.)"""

    def __init__(self, contexts, est, inv, ring, L, pitch, B, firs=None, k1_plan0=None, tails="none", paired=False):
        from impulse_hip import Context, ConvPlan
        from impulse_hip._native import FirChain
        self.ring, self.L, self.pitch, self.B = ring, L, pitch, B
        self.n, self.K, self.head, self.fade = fir_stage_shape(est)
        self.firs = synth_firs(B, self.K) if firs is None else firs
        self.po = (self.n + self.K - 1 + 63) // 64 * 64
        self.lanes, self.owned = [], []
        shared_tail = None
        if tails == "shared":
            shared_tail = Context(contexts[0].device)
            self.owned.append(shared_tail)
        for i, ctx in enumerate(contexts):
            if tails == "none":
                tail = ctx
            elif tails == "shared":
                tail = shared_tail
            else:
                tail = Context(ctx.device)
                self.owned.append(tail)
            if i == 0 and k1_plan0 is not None:
                plan1 = k1_plan0(ctx)                               # rank 0: from the filter; other ranks: empty + broadcast
            elif i == 0:
                plan1 = ConvPlan(ctx, inv, L, "same", ws_channels=B, fused=False, paired=paired)
            else:
                plan1 = ConvPlan(ctx, None, L, "same", ws_channels=B, empty_M=len(inv), n_filters=1, fused=False, paired=paired)
                copy_spectrum(plan1, self.lanes[0]["plan1"], ctx)
            plan5 = ConvPlan(tail, self.firs, self.n, "full", ws_channels=B)
            chain = FirChain(plan1, plan5, B, self.head, self.head, self.fade)
            outs = [ctx.malloc(B * self.po * 4) for _ in range(3)]
            self.lanes.append(dict(ctx=ctx, tail=tail, plan1=plan1, plan5=plan5, chain=chain, outs=outs, d_pk=ctx.malloc(B * 8), k=0))
        self.next = 0

    def call(self, d_x):
        ln = self.lanes[self.next % len(self.lanes)]
        self.next += 1
        out = ln["outs"][ln["k"] % len(ln["outs"])]
        ln["k"] += 1
        ln["last_out"] = out
        ln["chain"].execute_device(d_x, self.pitch, out, self.po, ln["d_pk"])

    def step(self):
        for p in self.ring.ptrs:
            self.call(p)

    def sync(self):
        for ln in self.lanes:
            ln["ctx"].synchronize()
            ln["tail"].synchronize()

    def set_timing(self, every):
        for ln in self.lanes:
            ln["plan1"].set_timing(every)
            ln["plan5"].set_timing(every)
            ln["plan1"].get_timing(reset=True)
            ln["plan5"].get_timing(reset=True)

    def timing(self):
        """(ms of K1 pass A, B, C summed over the sampled launch groups, groups), (ms of the fused K5 launch, launches)"""
        k1, n1, k5, n5 = np.zeros(3), 0, 0.0, 0
        for ln in self.lanes:
            ms, n = ln["plan1"].get_timing(reset=True)
            k1 += np.asarray(ms)
            n1 += n
            ms, n = ln["plan5"].get_timing(reset=True)
            k5 += ms[0]
            n5 += n
        return (k1, n1), (k5, n5)

    def fetch(self, lane=-1):
        """outputs [B][n + K - 1] and peak indices of the last call of one chain"""
        ln = self.lanes[lane]
        y = np.empty((self.B, self.po), dtype=np.float32)
        pk = np.empty(self.B, dtype=np.int64)
        ln["ctx"].d2h(y, ln["last_out"])
        ln["ctx"].d2h(pk, ln["d_pk"])
        return y[:, :self.n + self.K - 1], pk

    def release(self):
        self.sync()
        for ln in self.lanes:
            ln["chain"].close()
            ln["plan1"].close()
            ln["plan5"].close()
            for p in ln["outs"] + [ln["d_pk"]]:
                ln["ctx"].free(p)
        for c in reversed(self.owned):
            c.close()
        self.lanes, self.owned = [], []


class K1Team:
    """K1 alone over the input ring: one plan per context (= one launch group in flight per stream), fed round robin."""

    def __init__(self, contexts, inv, ring, B, L, pitch, M, group_channels, plan0=None, paired=False):
        from impulse_hip import ConvPlan
        self.ring, self.B, self.L, self.pitch = ring, B, L, pitch
        # output rows: same pitch; the first sample of a row is placed so that the crop offset of the 'same' window
        # lands the stores on 128-byte lines (the caller chooses where results go: here (M-1)/2 mod 32 samples in)
        self.skew = ((M - 1) // 2) % 32 if os.environ.get("IMPULSE_BENCH_OUT_SKEW", "1") == "1" else 0
        self.lanes = []
        for i, ctx in enumerate(contexts):
            if i == 0 and plan0 is not None:
                plan = plan0(ctx)
            elif i == 0:
                plan = ConvPlan(ctx, inv, L, "same", ws_channels=group_channels, fused=False, paired=paired)
            else:
                plan = ConvPlan(ctx, None, L, "same", ws_channels=group_channels, empty_M=len(inv), n_filters=1, fused=False,
                                paired=paired)
                copy_spectrum(plan, self.lanes[0]["plan"], ctx)
            self.lanes.append(dict(ctx=ctx, plan=plan, buf=ctx.malloc((B * pitch + 64) * 4)))
        self.plan = self.lanes[0]["plan"]
        self.n = 0

    def call(self, d_x, lane=None):
        ln = self.lanes[(self.n if lane is None else lane) % len(self.lanes)]
        self.n += 1
        ln["plan"].execute_device(d_x, self.B, self.pitch, ln["buf"] + 4 * self.skew, self.pitch)

    def step(self):
        for p in self.ring.ptrs:
            self.call(p)

    def sync(self):
        for ln in self.lanes:
            ln["ctx"].synchronize()

    def set_timing(self, every):
        for ln in self.lanes:
            ln["plan"].set_timing(every)
            ln["plan"].get_timing(reset=True)

    def timing(self, lanes=None):
        ms, n = np.zeros(3), 0
        for ln in (self.lanes if lanes is None else [self.lanes[i] for i in lanes]):
            m, k = ln["plan"].get_timing(reset=True)
            ms += np.asarray(m)
            n += k
        return ms, n

    def outputs(self):
        ys = []
        for ln in self.lanes:
            y = np.empty((self.B, self.pitch), dtype=np.float32)
            ln["ctx"].d2h(y, ln["buf"] + 4 * self.skew)
            ys.append(y[:, :self.L])
        return ys

    def release(self):
        self.sync()
        for ln in self.lanes:
            ln["plan"].close()
            ln["ctx"].free(ln["buf"])
        self.lanes = []


# ------------------------------------------------------------------------------------------------------
# CPU baseline: the NumPy oracle of the same work on this box's host cores
# ------------------------------------------------------------------------------------------------------
def cpu_peak_index(ir):
    """ImpulseResponse.peak_index on the host the way the reference runs it: scipy.signal.find_peaks (compiled) where SciPy
    is present (core/impulse_response.py:52-70), else the oracle's pure-NumPy restatement of it (20x slower: a baseline
    timed with that would flatter the GPU)."""
    try:
        from scipy.signal import find_peaks
    except ImportError:
        from oracle.impulse_response import peak_index
        return peak_index(ir)
    mx = np.max(np.abs(ir))
    if mx < 1e-20:
        return 0
    seg = ir / mx
    peaks = np.concatenate([find_peaks(seg, height=0.12589)[0], find_peaks(-seg, height=0.12589)[0]])
    return int(np.min(peaks)) if len(peaks) else int(np.argmax(np.abs(seg)))


def oracle_chain(x, inv, fir, n, head, fade):
    """estimate -> peak_index -> crop at peak - head (n samples, clamped) with Hann fades -> fir, 'full' (float64)"""
    from oracle.estimator import estimate
    from oracle.scipy_restated import fft_convolve, hann
    ir = estimate(x, inv)
    pk = cpu_peak_index(ir)
    s0 = min(max(pk - head, 0), len(ir) - n)
    w = np.ones(n)
    w[:head] *= hann(2 * head)[:head]
    w[n - fade:] *= hann(2 * fade)[fade:]
    return ir, pk, fft_convolve(ir[s0:s0 + n] * w, fir, "full")


def cpu_baseline(est, rec, L, firs, shape, with_fir, budget_s=10.0):
    """The oracle's restatement of the timed work (float64; estimate() = nfft next_fast_len with rfft(h) recomputed per call,
    exactly like core/impulse_response_estimator.py:149-151), serial over channels as the reference ingests them
    (core/hrir.py:307-355), then through a thread pool of min(2*cpu, 32) workers - the reference's
    core/parallel_processing.py:31-48 heuristic (pocketfft releases the GIL).  Bounded samples of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.estimator import estimate
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    n, K, head, fade = shape

    def work(c):
        x = rec[c, :L].astype(np.float64)
        return oracle_chain(x, inv, firs[c], n, head, fade) if with_fir else (estimate(x, inv), None, None)

    done, t0, outs = 0, time.perf_counter(), {}
    while True:
        c = done % rec.shape[0]
        r = work(c)
        if c not in outs:
            outs[c] = r
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 512:
            break
    serial = dict(value=done / el, unit="IR/s", cores=1, kind="port",
                  sample=f"{done} IRs of the same synthetic batch, serial float64 NumPy pocketfft")
    workers = min(2 * (os.cpu_count() or 1), 32)
    idx = list(range(rec.shape[0])) * max(1, (2 * workers) // rec.shape[0])
    pdone, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as pool:
        while time.perf_counter() - t0 < budget_s:
            list(pool.map(work, idx))
            pdone += len(idx)
    pel = time.perf_counter() - t0
    what = "estimate -> peak_index -> crop + fades -> FIR (the chain `value` times)" if with_fir else "estimate() only"
    return dict(value=pdone / pel, unit="IR/s", cores=workers, kind="port",
                sample=f"{pdone} IRs of the same synthetic batch through a pool of {workers} threads (the reference's "
                       f"parallel_map heuristic) on a host with {os.cpu_count()} logical cores, "
                       f"{len(os.sched_getaffinity(0))} usable; {what}",
                work=what, serial=serial), outs


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
    if "column_i32" not in z.files:
        return None
    col = z["column_i32"].astype(np.float64) / 2.0 ** 31
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


# ------------------------------------------------------------------------------------------------------
# L2<->fabric traffic from the PMC counters
# ------------------------------------------------------------------------------------------------------
PMC_KEYS = ("cols_fwd", "rows_kernel", "cols_inv", "peak_search", "fir_block")


def load_profile_traffic(workload):
    """L2<->fabric bytes per launch from the committed rocprofv3 --pmc summary of this command (profiles/), or None."""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as fh:
                t = json.load(fh).get(workload)
        except (OSError, ValueError):
            continue
        if t:
            return t, "profiles/" + name
    return None, None


def live_pmc_traffic(workload):
    """FETCH_SIZE and WRITE_SIZE of the chain's kernels, collected NOW: two child runs of this very script under
    `rocprofv3 --pmc` (one counter per pass, as MI355X_MICROARCH.md prescribes; strictly serial launches so a counter
    belongs to one kernel at a time), started after this process has finished its own GPU work.  Returns bytes per launch
    per kernel, or None when rocprofv3 is missing, this process is itself being profiled, or a child fails - the caller
    then falls back to the committed summary and says so."""
    import shutil
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
                   sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--pmc-child"]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=90)
            if res.returncode != 0:
                sys.stderr.write(f"[bench] rocprofv3 --pmc {counter} child failed (rc {res.returncode}): {res.stderr[-400:]}\n")
                return None
        pm = pmc_summary.main(out)
        if "rows_kernel" not in pm and "rows_single" in pm:
            pm["rows_kernel"] = pm["rows_single"]              # pair-mode plans: rows_single_kernel is the row pass
        if not all(k in pm and "FETCH_SIZE" in pm[k] and "WRITE_SIZE" in pm[k] for k in ("rows_kernel", "cols_fwd", "cols_inv")):
            return None
        # gfx950: FETCH_SIZE counts half of a coalesced read stream (MI355X_MICROARCH.md)
        t = {k + "_bytes_per_launch": (2 * pm[k]["FETCH_SIZE"] + pm[k]["WRITE_SIZE"]) * 1024 for k in PMC_KEYS
             if k in pm and "FETCH_SIZE" in pm[k] and "WRITE_SIZE" in pm[k]}
        t["rows_kernel_fetch_kb_raw"] = pm["rows_kernel"]["FETCH_SIZE"]
        t["rows_kernel_write_kb"] = pm["rows_kernel"]["WRITE_SIZE"]
        return t
    except Exception as exc:                                  # noqa: BLE001 - a reported figure, never fatal
        sys.stderr.write(f"[bench] live PMC collection failed: {exc!r}\n")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


SLICE_PMC_CALLS, SLICE_PMC_M = 3, 4


def slice_pmc_child(args):
    """--pmc-child --stage slice: set-up, then SLICE_PMC_CALLS serial calls of SLICE_PMC_M measurements on one stream"""
    est = make_estimator(args.workload)
    rec, L, _, _ = synth_recordings(est, WORKLOADS[args.workload][2], seed0=0xC2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        team = SliceTeam(est, rec, L, n_streams=1, M=SLICE_PMC_M, ring=1, quiet_setup=True)
        for _ in range(SLICE_PMC_CALLS):
            team.step()
            team.sync()
        team.release()
    return 0


def live_slice_traffic(workload):
    """L2<->fabric bytes one measurement of the resident slice moves (every kernel of a call: FETCH_SIZE x2 + WRITE_SIZE
    summed over the dispatches of the child's calls / measurements), from two `rocprofv3 --pmc` child runs; None if it cannot
    be collected"""
    import shutil
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
        env.pop(k, None)
    try:
        tot = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--pmc-child", "--stage", "slice"]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            if res.returncode != 0:
                sys.stderr.write(f"[bench] rocprofv3 --pmc {counter} slice child failed (rc {res.returncode}): {res.stderr[-400:]}\n")
                return None
            t = pmc_summary.totals_from(d, "cols_")          # a call starts with K1's forward column pass; set-up has none
            if counter not in t:
                return None
            tot[counter] = t[counter]
        n_meas = SLICE_PMC_CALLS * SLICE_PMC_M + SLICE_PMC_M   # + the sizing call SliceTeam makes
        kb = 2 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]  # gfx950: FETCH_SIZE counts half of a coalesced read stream
        return dict(bytes_per_measurement=kb * 1024 / n_meas, dispatches_per_call=tot["FETCH_SIZE"][1] / (SLICE_PMC_CALLS + 1))
    except Exception as exc:                                  # noqa: BLE001 - a reported figure, never fatal
        sys.stderr.write(f"[bench] live slice PMC collection failed: {exc!r}\n")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def slice_resident_block(est, rec, L, workload, no_pmc=False, n_streams=3, M=8, budget_s=0.5):
    """`slice_resident`: the reference's real stage order (ingest -> crop_heads with the earlier-ear rule -> crop_tails at
    the Lundeby length -> equalize -> normalize) as imp_slice runs it - M measurements per call, no host readback between the
    stages, recordings resident in HBM.  Checked in the same run: no measurement flagged, and measurement 0 bit-identical to
    the staged class path (one readback per stage)."""
    import warnings
    from impulse_hip.pipeline_slice import run_slice
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        team = SliceTeam(est, rec, L, n_streams=n_streams, M=M, ring=3)
        desc = team.describe()
        for _ in range(2):
            team.step()
        team.sync()
        t0 = time.perf_counter()
        team.step()
        team.sync()
        one = max(time.perf_counter() - t0, 1e-4)
        calls = int(max(4, min(400, budget_s / one)))
        t0 = time.perf_counter()
        for _ in range(calls):
            team.step()
        team.sync()
        dt = time.perf_counter() - t0
        flags = team.flags()
        rows, meas = team.results(0)
        # the same with every call designing its FIRs first (a call = a job of M measurements with curves of its own)
        team._target()
        for _ in range(2):
            team.step(design=True)
        team.sync()
        calls_d = max(4, calls // 2)
        t0 = time.perf_counter()
        for _ in range(calls_d):
            team.step(design=True)
        team.sync()
        dt_design = time.perf_counter() - t0
        for ln in team.lanes:                              # back to the job FIRs the checks below use
            ln["rs"].set_firs(team.firs)
        # the sequence without the two alignments (what rounds 3 / early 4 timed)
        for ln in team.lanes:
            ln["rs"].set_alignment(False)
        for _ in range(2):
            team.step()
        team.sync()
        t0 = time.perf_counter()
        for _ in range(calls_d):
            team.step()
        team.sync()
        dt_plain = time.perf_counter() - t0
        for ln in team.lanes:
            ln["rs"].set_alignment(True)
        # the same with the optional stage between equalize and normalize: every row's decay pulled to a 0.3 s RT60
        for ln in team.lanes:
            ln["rs"].set_decay(0.3)
        for _ in range(2):
            team.step()
        team.sync()
        t0 = time.perf_counter()
        for _ in range(calls_d):
            team.step()
        team.sync()
        dt_decay = time.perf_counter() - t0
        d_rows, d_meas = team.results(0)
        decay_states = {int(k): int(v) for k, v in zip(*np.unique(d_rows["decay_state"], return_counts=True))}
        decay_flags = sorted({int(f) for f in d_meas["flags"]})
        for ln in team.lanes:
            ln["rs"].set_decay(None)
        team.step()
        team.sync()
        flags = sorted(set(flags) | set(team.flags()))
        rows, meas = team.results(0)
        out = team.fetch(0)
        rs = team.lanes[0]["rs"]
        firs = {t: team.firs[i] for i, t in enumerate(team.layout.tasks)}
        hrir, gain = run_slice(est, [((est.fs, team.frames), team.speakers)], firs=firs, align=True)
        n = int(meas["out_len"][0])
        same = abs(float(meas["gain_db"][0]) - gain) <= 1e-11
        for q, sp in enumerate(team.speakers):
            for s_, sd in enumerate(("left", "right")):
                want = hrir.irs[sp][sd].peek()
                got = out[2 * q + s_, :n].astype(np.float64)
                same = same and got.shape == want.shape and bool(np.array_equal(got, want))
        rows_per = team.rows
        in_bytes = team.frames.nbytes / rows_per
        out_bytes = 4.0 * n
        team.release()
    irs = calls * n_streams * M * rows_per
    rate = irs / dt
    alg = in_bytes + out_bytes
    block = dict(value=rate, unit="IR/s", timed_region_s=dt, calls=calls * n_streams, measurements_per_call=M, streams=n_streams,
                 ms_per_call_per_stream=dt / calls * 1e3, arrangement=desc,
                 with_fir_design_per_call=dict(value=calls_d * n_streams * M * rows_per / dt_design, unit="IR/s", calls=calls_d * n_streams,
                                               note="every call designs its 16 minimum-phase FIRs first (K12 -> K6 on the call's "
                                                    "stream, left on the device, spectra formed there): a job of M measurements "
                                                    "with equalisation curves of its own per call"),
                 without_alignment=dict(value=calls_d * n_streams * M * rows_per / dt_plain, unit="IR/s", calls=calls_d * n_streams,
                                        note="ingest -> crop_heads -> crop_tails -> equalize -> normalize only (the sequence of "
                                             "earlier rounds' lines)"),
                 shifts_of_one_measurement=dict(ipsilateral=[int(v) for v in rows["shift_ipsilateral"][:rows_per]],
                                                onset=[int(v) for v in rows["shift_onset"][:rows_per]]),
                 with_decay_adjustment=dict(value=calls_d * n_streams * M * rows_per / dt_decay, unit="IR/s", calls=calls_d * n_streams,
                                            target_rt60_s=0.3, row_states_of_one_call=decay_states, flags_seen=decay_flags,
                                            note="the optional stage of core/pipeline.py:694-716 between equalize and normalize, on "
                                                 "every row: decay_params (K3 + K7c) and decay_times (K7b) of the equalized rows, the "
                                                 "window (K8) in place; row states 1 = adjusted, 2 = already faster, 3 = left to the host"),
                 keep=int(meas["keep"][0]), out_len=n, flags_seen=flags, no_measurement_flagged=flags == [0],
                 bit_identical_to_staged_path=bool(same),
                 algorithmic_bytes_per_ir=alg, path_achieved=rate * alg / 1e9, path_frac=rate * alg / 1e9 / HBM_PEAK_GBS,
                 note="the reference's stage sequence (core/pipeline.py:565-573, 585-601 incl. the alignments of :593-597, 647-692, 725-735) per measurement: "
                      "PCM frames in HBM -> K1 (pair mode) -> first peaks -> crop_heads (earlier ear of each pair - 1 ms, fade-in) -> "
                      "align_ipsilateral_all (K10 lags of 30 ms segments) + align_onset_groups_peak_leftref, rows materialised -> "
                      "Lundeby knees (K7c) -> crop_tails at min(shortest row, next_fast_len(latest knee)) + fade-out -> per-channel "
                      f"{rs.taps}-tap FIR (K5) -> normalize (K2 of the ear sums, gain on the device); scalars read back once per call; "
                      "FIRs set once per job.  algorithmic bytes per IR = the recording's PCM bytes / 16 in + 4 (keep + taps - 1) out")
    if not no_pmc:
        live = live_slice_traffic(workload)
        if live:
            per_ir = live["bytes_per_measurement"] / rows_per
            block["l2_fabric_traffic"] = dict(bytes_per_ir=per_ir, rate=rate * per_ir / 1e9, unit="GB/s", source="live",
                                              dispatches_per_call=live["dispatches_per_call"],
                                              over_algorithmic=per_ir / alg,
                                              note="FETCH_SIZE x2 + WRITE_SIZE of every kernel of a call (rocprofv3 --pmc child runs "
                                                   "made by this run, serial calls), per IR; Infinity-Cache hits included: a fabric "
                                                   "figure, not HBM bytes")
    return block


def pmc_child(args):
    """what the --pmc children run: a few strictly serial chain calls (every kernel alone on the chip)"""
    if args.stage == "slice":
        return slice_pmc_child(args)
    from impulse_hip import Context
    est = make_estimator(args.workload)
    if args.workload in ("c4", "c5"):
        # K1 alone, launch groups of GROUP_CHANNELS channels as the timed leg runs them, strictly serial
        g = GROUP_CHANNELS[args.workload]
        rec, L, pitch, _ = synth_recordings(est, g, seed0=0xC5, column=len(est))
        ctx = Context(0)
        ring = InputRing(ctx, rec, 2)
        dec = K1Team([ctx], np.asarray(est.inverse_filter, dtype=np.float64), ring, g, L, pitch, len(est), g,
                     paired=k1_paired(args.workload))
        for k in range(4):
            dec.call(ring.ptrs[k % 2], lane=0)
            dec.sync()
        dec.release()
        ring.release()
        ctx.close()
        return 0
    B = WORKLOADS[args.workload][2] * MEASUREMENTS_PER_BLOCK.get(args.workload, 1)
    rec, L, pitch, _ = synth_recordings(est, B, seed0=0xC2)
    ctx = Context(0)
    ring = InputRing(ctx, rec, 2)
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    team = ChainTeam([ctx], est, inv, ring, L, pitch, B, paired=k1_paired(args.workload))
    for _ in range(3):
        team.step()
        team.sync()
    team.release()
    ring.release()
    ctx.close()
    return 0


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


def spectrum_broadcast(plan, ctx, dist, torch, device, backend, rank, world):
    """The path's one collective.  With the RCCL backend the LIBRARY does it (imp_comm_* over librccl; the 128-byte
    communicator id is handed round by the launcher's process group) - torch.distributed only provides the launcher's barriers
    and clock reductions; the gloo rehearsal stages the bytes through host memory instead."""
    global BROADCAST_VIA, RCCL_RANKS_SEEN, RCCL_COMM
    from impulse_hip.sharding import broadcast_plan_spectrum
    if backend != "nccl" or os.environ.get("IMPULSE_BENCH_BCAST", "lib") != "lib":
        return broadcast_plan_spectrum(plan, ctx, dist, torch, device, src=0, via_host=(backend != "nccl"))
    from impulse_hip._native import comm_probe, comm_unique_id
    # preflight: ncclCommInitRank has no timeout, so every rank first says whether it can load librccl at all; only if
    # ALL can does anyone create a communicator
    bad = 0 if (comm_probe() and os.environ.get("IMPULSE_BENCH_FAIL_LIB_BCAST") != "1") else 1
    flag = torch.tensor([bad], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        sys.stderr.write(f"[bench] rank {rank}: librccl not loadable by the library on some rank; using torch.distributed\n")
        BROADCAST_VIA = "torch.distributed (librccl could not be opened by the library on every rank: see stderr)"
        return broadcast_plan_spectrum(plan, ctx, dist, torch, device, src=0, via_host=False)
    # the communicator id is 128 bytes of control plane: rank 0 makes it, the launcher's process group hands it round.
    # ONE communicator per rank serves every broadcast of the run (the K1 plan, the chain's plan, the strong block's).
    if RCCL_COMM is None:
        from impulse_hip._native import Comm
        box = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        RCCL_COMM = Comm(ctx, box[0], rank, world)
        seen = torch.tensor([RCCL_COMM.nranks_seen()], dtype=torch.int32, device=device)
        dist.all_reduce(seen, op=dist.ReduceOp.MIN)
        RCCL_RANKS_SEEN = int(seen.item())
    dptr, nbytes = plan.spectrum_buffer()
    ctx.synchronize()
    RCCL_COMM.broadcast(dptr, nbytes, root=0)
    dist.barrier()
    return nbytes


def strong_block(args, torch, dist, comm_device, device, ctx, rank, world, backend):
    """BASELINE.json configs[4]: 1024 channels x 2^20 samples sharded over the ranks (strong scaling), then rank 0
    alone over all of them for the single-GPU figure of the same run.  Deconvolution (K1) only, as the config says."""
    from impulse_hip import ConvPlan
    from impulse_hip.sharding import shard_channels
    total = args.strong_channels
    est = make_estimator("c5")
    M = len(est)
    # launch groups of 12 channels on two lanes: 24 workspaces of 6.3 MB in flight (tools/k1_rate.py at 2^20 x 2^20:
    # 8 ch x 3 lanes 132 k, 12 x 2 139 k, 12 x 3 126 k, 16 x 2 136 k IR/s)
    grp = 12
    lanes = max(1, min(args.lanes or 2, 2))
    paired = k1_paired("c5")                              # the 384-row pair plan: inside the fp32 floor on the whole column
    if rank == 0:
        plan = ConvPlan(ctx, np.asarray(est.inverse_filter, dtype=np.float64), M, "same", ws_channels=lanes * grp, paired=paired)
    else:
        plan = ConvPlan(ctx, None, M, "same", ws_channels=lanes * grp, empty_M=M, n_filters=1, paired=paired)
    bcast = 0
    if dist is not None:
        bcast = spectrum_broadcast(plan, ctx, dist, torch, device, backend, rank, world)
    plan.set_overlap(lanes)
    base, L, pitch, dl = synth_recordings(est, 64, seed0=0xC5, column=M)
    d_base = ctx.malloc(base.nbytes)
    ctx.h2d(d_base, base)
    row_bytes = pitch * 4

    def timed(lo, hi, everyone):
        n = hi - lo
        # channel c of the 1024-channel batch is recording c % 64 (tiled on the device)
        d_x = ctx.malloc(n * row_bytes)
        for i in range(n):
            ctx.d2d(d_x + i * row_bytes, d_base + ((lo + i) % 64) * row_bytes, row_bytes)
        skew = ((M - 1) // 2) % 32
        d_ybuf = ctx.malloc((n * pitch + 64) * 4)
        d_y = d_ybuf + 4 * skew
        ctx.synchronize()

        def one_pass():
            plan.execute_device(d_x, n, pitch, d_y, pitch)

        one_pass()
        ctx.synchronize()
        if everyone and dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.strong_passes):
            one_pass()
        ctx.synchronize()
        el = time.perf_counter() - t0
        if everyone and dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        # parity: every sampled channel's peak where the analytic truth puts it, and bit-equal to its twin 64 channels on
        # (same recording -> same bits whatever launch group / lane)
        rows = sorted(set(list(range(0, n, max(1, n // 16))) + [n - 1]))
        ok = True
        y, twin = np.empty(pitch, dtype=np.float32), np.empty(pitch, dtype=np.float32)
        for r in rows:
            ctx.d2h(y, d_y + r * row_bytes)
            ok &= int(np.argmax(np.abs(y[:L]))) == M // 2 + dl[(lo + r) % 64]
            if r + 64 < n:
                ctx.d2h(twin, d_y + (r + 64) * row_bytes)
                ok &= bool(np.array_equal(y[:L], twin[:L]))
        ctx.free(d_x)
        ctx.free(d_ybuf)
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
    nfft = plan.nfft
    plan.close()
    ctx.free(d_base)
    out = None
    if rank == 0:
        rate = total * args.strong_passes / el_n
        out = dict(workload=WORKLOADS["c5"][3], channels=total, scaling="strong", n_gpus=world,
                   ranks_seen=dist.get_world_size() if dist is not None else 1, rccl_ranks_seen=RCCL_RANKS_SEEN,
                   channels_per_rank=hi - lo, passes=args.strong_passes, value=rate, unit="IR/s",
                   ms_per_pass=el_n / args.strong_passes * 1e3, nfft=nfft, plan="pair" if paired else "mono",
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
    if args.pmc_child:
        return pmc_child(args)
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

    torch = dist = device = comm_device = None
    # IMPULSE_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks: ranks are
    # folded onto the visible devices and the spectrum broadcast is staged through host memory.
    backend = os.environ.get("IMPULSE_BENCH_BACKEND", "nccl")
    dev_index = local_rank
    # IMPULSE_BENCH_FORCE_DIST=1 takes the collective path with a single rank too (a one-GPU box can
    # then exercise RCCL init, the spectrum broadcast and the reductions).  torch is the LAUNCHER's control plane
    # (rendezvous, barriers, clock reductions) and is imported for N > 1 only.
    use_dist = world > 1 or os.environ.get("IMPULSE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch
        import torch.distributed as dist
        if backend == "gloo":
            dev_index = local_rank % max(torch.cuda.device_count(), 1)

    from impulse_hip import Context, ConvPlan
    from impulse_hip.sharding import shard_channels

    lanes = max(1, min(args.lanes or CHAINS[args.workload], 4))
    # one context (= one stream) per call in flight, made once - BEFORE the launcher's process group brings its own streams -
    # and used by every leg: K1 alone, pair mode and the chains.  Three streams stay below the four hardware queues a HIP
    # process gets by default, so no two of them are multiplexed.
    contexts = [Context(dev_index) for _ in range(lanes)]
    ctx = contexts[0]
    if use_dist:
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

    fs, dur, B_meas, desc = WORKLOADS[args.workload]
    est = make_estimator(args.workload)
    strong = args.workload in ("c4", "c5")
    stage = "deconv" if strong else args.stage
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    M = len(est)
    if strong:
        lo, hi = shard_channels(B_meas, world, rank)
        B = GROUP_CHANNELS[args.workload] * min(lanes, 3)     # channels per call: one launch group per lane
        rec, L, pitch, delays = synth_recordings(est, B, seed0=0xC5 + lo, column=len(est))
        mpb = 1
        n_blocks = args.blocks or max(1, (hi - lo) // B)      # one step = one pass over this rank's shard
    else:
        mpb = int(os.environ.get("IMPULSE_BENCH_MEASUREMENTS", "0")) or MEASUREMENTS_PER_BLOCK[args.workload]
        B = B_meas * mpb                                      # a resident block = mpb measurements, rows contiguous
        rec, L, pitch, delays = synth_recordings(est, B, seed0=0xC2 + 1000 * rank)
        # C2: 320 blocks x 2 measurements x 16 channels = 10 240 IRs per step: 20 steps time 0.55 s at 370 k IR/s
        n_blocks = args.blocks or int(os.environ.get("IMPULSE_BENCH_BLOCKS", "0")) or DEFAULT_BLOCKS[args.workload]
    irs_per_step_rank = n_blocks * B

    ring = InputRing(ctx, rec, n_blocks)
    group_channels = int(os.environ.get("IMPULSE_BENCH_GROUP", "0")) or (B if args.workload == "c2" else GROUP_CHANNELS[args.workload])
    bcast_bytes = [0]

    def k1_plan(c, ws_channels, paired=False):
        """rank 0 prepares the inverse-sweep spectrum; every other rank receives it (the path's one collective)"""
        if rank == 0:
            plan = ConvPlan(c, inv, L, "same", ws_channels=ws_channels, fused=False, paired=paired)
        else:
            plan = ConvPlan(c, None, L, "same", ws_channels=ws_channels, empty_M=M, n_filters=1, fused=False, paired=paired)
        if dist is not None:
            bcast_bytes[0] = spectrum_broadcast(plan, c, dist, torch, device, backend, rank, world)
        return plan

    def barrier(obj):
        obj.sync()
        if dist is not None:
            dist.barrier()

    def timed_steps(obj, steps, warmup):
        for _ in range(warmup):
            obj.step()
        barrier(obj)
        t0 = time.perf_counter()
        for _ in range(steps):
            obj.step()
        obj.sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            dist.barrier()
        return el

    peaks_ok = True
    shape = fir_stage_shape(est)
    n_fir, K_fir, head, fade = shape
    # ---------------------------------------------------------------- K1 alone (secondary; headline for C4 / C5)
    main_paired = k1_paired(args.workload)
    dec = K1Team(contexts, inv, ring, B, L, pitch, M, group_channels, plan0=lambda c: k1_plan(c, group_channels, main_paired),
                 paired=main_paired)
    dec_steps = args.steps if stage == "deconv" else max(2, args.steps // 4)
    groups_per_call = -(-B // group_channels)
    groups_timed = dec_steps * n_blocks * groups_per_call
    stride = 0 if args.no_events else max(1, min(args.event_stride, groups_timed // 8))
    dec.set_timing(stride if stage == "deconv" else 0)
    dec_elapsed = timed_steps(dec, dec_steps, args.warmup if stage == "deconv" else 1)
    k1_ms, k1_n = dec.timing()
    # the same kernels with nothing else on the chip (launch groups strictly serial: one stream), outside the timed
    # region: under overlap a kernel's event-to-event time includes the share of the chip it cedes to the other groups
    barrier(dec)
    dec.set_timing(1)
    for p in ring.ptrs[:max(4, min(40, n_blocks))]:
        dec.call(p, lane=0)
    dec.sync()
    iso_ms, iso_n = dec.timing(lanes=[0])
    dec.set_timing(0)
    for i, p in enumerate(ring.ptrs[:len(contexts)]):          # every lane's output written by the overlapped form again
        dec.call(p, lane=i)
    dec.sync()
    ys = dec.outputs()
    for y in ys:
        peaks_ok &= all(int(np.argmax(np.abs(y[c]))) == M // 2 + delays[c] for c in range(B))
    y_k1 = ys[-1]
    nfft, plan_ws, skew = dec.plan.nfft, dec.plan.ws_channels * len(contexts), dec.skew
    dec.release()
    dec_rate_rank = irs_per_step_rank * dec_steps / dec_elapsed
    # the other plan kind on the same planar rows (pair mode = two channels per complex transform, where the lengths allow it)
    pair_block, y_pair = None, None
    if rank == 0 and world == 1:
        try:
            from impulse_hip._native import plan_geometry_paired
            if main_paired or plan_geometry_paired(M, L, "same") is not None:
                pdec = K1Team(contexts, inv, ring, B, L, pitch, M, group_channels, paired=not main_paired)
                psteps = max(2, args.steps // 4)
                pel = timed_steps(pdec, psteps, 1)
                pys = pdec.outputs()
                y_pair = pys[-1]
                p_ok = all(int(np.argmax(np.abs(yy[c]))) == M // 2 + delays[c] for yy in pys for c in range(B))
                peaks_ok &= p_ok
                prate = irs_per_step_rank * psteps / pel
                pair_block = dict(plan="mono" if main_paired else "pair", value=prate, unit="IR/s", rows=pdec.plan.n1,
                                  path_frac=prate * 8.0 * L / 1e9 / HBM_PEAK_GBS,
                                  peak_indices_exact=bool(p_ok), max_rel_diff_vs_headline_plan=float(
                                      np.max(np.abs(pys[-1].astype(np.float64) - y_k1)) / np.max(np.abs(y_k1))),
                                  note="K1 with the OTHER plan kind on the same planar rows.  pair = channels (2q, 2q + 1) as ONE "
                                       "complex signal x_L + i x_R, pointwise H in the row pass (rows_single_kernel); mono = one "
                                       "channel per transform, even/odd packed; same circular length, same workspace bytes - "
                                       "DESIGN.md section 7 has the per-pass times and why both meet the same fabric ceiling")
                pdec.release()
        except Exception as exc:                              # noqa: BLE001 - secondary figure only
            pair_block = dict(error=repr(exc))

    # ---------------------------------------------------------------- the chain (the metric)
    team, chain_elapsed, chain_k1, chain_k5, firs = None, None, None, None, None
    y_chain = pk_chain = None
    if stage == "chain":
        team = ChainTeam(contexts, est, inv, ring, L, pitch, B, k1_plan0=lambda c: k1_plan(c, B, main_paired), paired=main_paired,
                         tails=os.environ.get("IMPULSE_BENCH_CHAIN_TAILS", "none"))
        team.step()
        barrier(team)
        calls_timed = args.steps * n_blocks
        team.set_timing(0 if args.no_events else max(1, min(args.event_stride, calls_timed // 8)))
        chain_elapsed = timed_steps(team, args.steps, args.warmup)
        chain_k1, chain_k5 = team.timing()
        team.set_timing(0)
        y_chain, pk_chain = team.fetch(-1)
        firs = team.firs
        peaks_ok &= all(abs(int(pk_chain[c]) - (M // 2 + delays[c])) <= 8 for c in range(B))   # exact check: against the oracle below
    if dist is not None:                      # rank 0 reports the verdict of every rank
        flag = torch.tensor([1 if peaks_ok else 0], dtype=torch.int32, device=comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        peaks_ok = bool(flag.item())
    if team is not None:
        team.release()
    ring.release()

    strong_c5 = None
    if (world > 1 and not args.no_strong) or args.strong:
        strong_c5, s_ok = strong_block(args, torch, dist, comm_device, device, ctx, rank, world, backend)
        peaks_ok &= s_ok

    # every leg's GPU objects are released: close their streams before the host-side report - the slice legs below bring
    # streams of their own, and a HIP process has four hardware queues (more streams than that are multiplexed)
    if RCCL_COMM is not None:
        RCCL_COMM.close()
    for c in reversed(contexts):
        c.close()
    contexts = []
    if rank == 0:
        irs_per_step = irs_per_step_rank * world
        steps_timed = args.steps if stage == "chain" else dec_steps
        elapsed = chain_elapsed if stage == "chain" else dec_elapsed
        value = irs_per_step * steps_timed / elapsed
        dec_value = dec_rate_rank * world
        # channels per K1 launch group: of the leg whose HIP events time the kernels (the chain deconvolves a whole call as ONE
        # group, K1 alone runs groups of `group_channels`), of the serial isolated calls (K1 alone), of the --pmc children
        timed_group = B if stage == "chain" else group_channels
        pmc_group = group_channels if strong else B
        alg_k1_launch = 8.0 * L * timed_group                            # algorithmic bytes of one timed K1 launch group
        alg_chain_ir = 4.0 * L + 4.0 * (n_fir + K_fir - 1)                # recording in, equalised cropped response out
        names = ("cols_kernel<fwd> (K1 pass A)", ("rows_single_kernel" if main_paired else "rows_kernel") + " (K1 pass B)",
                 "cols_kernel<inv> (K1 pass C)")
        # per-kernel times over the TIMED region: from the chains' K1 plans when the chain is what is timed
        t_ms, t_n = (chain_k1 if stage == "chain" else (np.asarray(k1_ms), k1_n))
        roof = None
        if t_n > 0:
            avg_ms = [float(m) / t_n for m in t_ms]
            dom = int(np.argmax(avg_ms))
            achieved = alg_k1_launch / (avg_ms[dom] * 1e-3) / 1e9
            iso_avg = [m / max(iso_n, 1) for m in iso_ms]
            iso_achieved = 8.0 * L * group_channels / (iso_avg[dom] * 1e-3) / 1e9
            prof, prof_src, live = None, None, False
            if world == 1 and not args.no_pmc:
                prof = live_pmc_traffic(args.workload)
                live = prof is not None
                prof_src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command, made by this run"
            if prof is None:
                prof, prof_src = load_profile_traffic(args.workload)
            roof = dict(bound="hbm", kernel=names[dom], achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS,
                        traffic=(prof or {}).get("rows_kernel_bytes_per_launch", 0.0) * timed_group / pmc_group if prof else None,
                        traffic_source=("live" if live else "committed summary") if prof else None,
                        traffic_note=(f"L2<->fabric bytes per launch of the dominant kernel (FETCH_SIZE x2 + WRITE_SIZE, "
                                      f"Infinity-Cache hits INCLUDED, so not HBM bytes); separate --pmc passes, strictly serial "
                                      f"launches; source: {prof_src}") if prof else None,
                        avg_kernel_ms=dict(zip(("pass_a", "pass_b", "pass_c"), avg_ms)),
                        events_sampled=int(t_n), launch_groups_in_flight=lanes, channels_per_timed_launch_group=timed_group,
                        channels_per_isolated_launch_group=group_channels, channels_per_pmc_launch_group=pmc_group,
                        note="achieved/frac: algorithmic bytes (8 L per IR) of one K1 launch group / HIP-event time of the "
                             "dominant kernel over the timed region; with several launch groups in flight that time includes "
                             "the share of the chip the kernel cedes to the others, so `isolated` (strictly serial groups) "
                             "and the `path_frac` figures (whole path, all kernels) are the cleaner ones",
                        isolated=dict(note="same kernel, launch groups strictly serial (nothing else on the chip), "
                                           "measured outside the timed region", achieved=iso_achieved,
                                      frac=iso_achieved / HBM_PEAK_GBS,
                                      avg_kernel_ms=dict(zip(("pass_a", "pass_b", "pass_c"), iso_avg))),
                        algorithmic_bytes_per_launch=alg_k1_launch,
                        deconv_only_path_achieved=dec_value / world * 8.0 * L / 1e9,
                        deconv_only_path_frac=dec_value / world * 8.0 * L / 1e9 / HBM_PEAK_GBS)
            if stage == "chain":
                roof["path_achieved"] = value / world * alg_chain_ir / 1e9
                roof["path_frac"] = roof["path_achieved"] / HBM_PEAK_GBS
                roof["path_note"] = ("path_*: the chain, algorithmic bytes per IR = 4 L in + 4 (n + K - 1) out = "
                                     f"{alg_chain_ir / 1e6:.2f} MB (the 0.68 s equalised response is all that leaves the path); "
                                     "deconv_only_path_*: K1 alone at 8 L per IR (last round's headline figure)")
                if chain_k5 and chain_k5[1]:
                    roof["fused_k5_avg_ms"] = chain_k5[0] / chain_k5[1]
            else:
                roof["path_achieved"], roof["path_frac"] = roof["deconv_only_path_achieved"], roof["deconv_only_path_frac"]
            if prof and all(k + "_bytes_per_launch" in prof for k in ("rows_kernel", "cols_fwd", "cols_inv")):
                moved_k1 = sum(prof[k + "_bytes_per_launch"] for k in ("rows_kernel", "cols_fwd", "cols_inv"))   # per pmc_group channels
                moved_tail = sum(prof.get(k + "_bytes_per_launch", 0.0) for k in ("peak_search", "fir_block"))
                per_call = moved_k1 * B / pmc_group + (moved_tail if stage == "chain" else 0.0)
                rate = per_call * n_blocks / (elapsed / steps_timed) / 1e9
                # the ceiling of this decomposition: the algorithmic share of the bytes K1 must move across the fabric, at the
                # rate the fabric gives such traffic
                roof["ceiling_frac"] = (8.0 * L * pmc_group / moved_k1) * FABRIC_CEILING_GBS / HBM_PEAK_GBS
                roof["ceiling_note"] = (f"algorithmic bytes / live-PMC fabric bytes of K1 ({8.0 * L * pmc_group / 1e6:.1f} / {moved_k1 / 1e6:.1f} MB per "
                                        f"{pmc_group}-channel group) x {FABRIC_CEILING_GBS / 1e3:.2f} TB/s (the best an in-place read + write "
                                        "stream reaches across the L2<->fabric boundary, profiles/r02_stream_probe2.txt) / 8 TB/s: the "
                                        "most K1 alone (`deconv_only_path_frac`) can reach with two workspace round trips; "
                                        "`deconv_only_fabric_rate` = the fabric rate K1 alone reached in this run")
                roof["deconv_only_fabric_rate"] = dec_value / world / pmc_group * moved_k1 / 1e9
                roof["l2_fabric_traffic"] = dict(
                    bytes_per_call=per_call, k1_bytes_per_launch_group=moved_k1,
                    peak_search_plus_fused_k5_bytes_per_call=moved_tail if stage == "chain" else None,
                    rate=rate, unit="GB/s", source=prof_src, k1_channels_per_pmc_launch_group=pmc_group,
                    note=f"bytes the kernels of one call ({B} channels) move across the L2<->fabric boundary (rocprofv3 --pmc, "
                         "see traffic_note) / this run's time per call.  Infinity-Cache hits are counted, so this is a FABRIC "
                         "rate, not achieved HBM bandwidth; it is not compared with the HBM peak")
        cpu, parity = None, dict(peak_indices_exact=bool(peaks_ok))
        if world == 1 and not args.no_cpu_baseline:
            with_fir = stage == "chain"
            cpu, outs = cpu_baseline(est, rec, L, firs, shape, with_fir)
            errs, errs_full, errs_full_pair, chain_errs = [], [], [], []
            for c, (ref, pk_ref, ref_out) in outs.items():
                pk = int(np.argmax(np.abs(ref)))
                peaks_ok &= pk == int(np.argmax(np.abs(y_k1[c])))
                for sl, acc in ((slice(pk - fs // 1000, pk - fs // 1000 + int(0.68 * fs)), errs), (slice(None), errs_full)):
                    A, R = np.abs(np.fft.rfft(y_k1[c][sl].astype(np.float64))), np.abs(np.fft.rfft(ref[sl]))
                    acc.append(float(np.max(np.abs(A - R)) / np.max(R)))
                if y_pair is not None:
                    A, R = np.abs(np.fft.rfft(y_pair[c].astype(np.float64))), np.abs(np.fft.rfft(ref))
                    errs_full_pair.append(float(np.max(np.abs(A - R)) / np.max(R)))
                if with_fir:
                    peaks_ok &= int(pk_chain[c]) == pk_ref
                    chain_errs.append(float(np.max(np.abs(y_chain[c] - ref_out)) / np.max(np.abs(ref_out))))
            floors = {c: fp32_fft_floor(est, rec[c], L) for c in outs}
            floor = max(v for v in floors.values() if v is not None) if any(v is not None for v in floors.values()) else None
            column_gate = max(WHOLE_COLUMN_TOL, WHOLE_COLUMN_SLACK * floor) if floor else WHOLE_COLUMN_TOL
            parity = dict(peak_indices_exact=bool(peaks_ok), spectrum_max_rel_err=max(errs), tolerance=1e-6,
                          spectrum_window="IR cropped as the pipeline does before any magnitude_response: peak - 1 ms, 0.68 s long",
                          whole_column_spectrum_max_rel_err=max(errs_full), whole_column_plan="pair" if main_paired else "mono",
                          whole_column_other_plan_max_rel_err=max(errs_full_pair) if errs_full_pair else None,
                          whole_column_gate=column_gate,
                          whole_column_meets_1e_6=bool(max(errs_full) <= 1e-6),
                          whole_column_pocketfft_fp32_err=floor,
                          whole_column_note=f"un-cropped {L}-sample column (the function's own output): north_star's 1e-6 is not "
                                            "reachable in fp32 - the transform's white rounding noise gains sqrt(L) in the spectrum - "
                                            "every fp32 transform sits at 1 - 2.5e-6 (the reference's own pocketfft in single "
                                            "precision, run on the SAME channels, is listed beside it: maximum over them).  GATED: the "
                                            f"headline plan's maximum <= max({WHOLE_COLUMN_TOL:g}, {WHOLE_COLUMN_SLACK:g} x pocketfft-fp32's "
                                            "maximum) - the statistic is a maximum over 2 - 5 x 10^5 bins of white noise and moves by "
                                            "up to 2x between channels of one transform; the other plan kind is reported",
                          channels_checked=len(errs))
            if with_fir:
                parity["chain_time_max_rel_err"] = max(chain_errs)
                parity["chain_note"] = ("equalised cropped responses of the timed chain against the oracle chain in float64 "
                                        "(max |dy| / max |y|); peak indices of the device search equal the oracle's")
                peaks_ok &= max(chain_errs) <= 1e-6
            try:
                parity["real_demo_column"] = demo_column_error()
            except Exception as exc:                          # noqa: BLE001 - reported figure only
                parity["real_demo_column"] = dict(error=repr(exc))
            peaks_ok &= max(errs) <= 1e-6
            peaks_ok &= max(errs_full) <= column_gate
            parity["peak_indices_exact"] = bool(peaks_ok)
        whole_slice = slice_res = None
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline and not args.no_slice:
            try:
                whole_slice = slice_rate(est, rec[:B_meas], L)
            except Exception as exc:                          # noqa: BLE001 - secondary figure only
                whole_slice = dict(error=repr(exc))
        if world == 1 and args.workload in ("c2", "c3") and not args.no_slice:
            try:
                slice_res = slice_resident_block(est, rec[:B_meas], L, args.workload, no_pmc=args.no_pmc)
                peaks_ok &= bool(slice_res["bit_identical_to_staged_path"])
            except Exception as exc:                          # noqa: BLE001 - secondary figure only
                slice_res = dict(error=repr(exc))
        result = {
            "metric": METRIC, "value": value, "unit": "IR/s", "n_gpus": world, "steps": steps_timed,
            "warmup": args.warmup, "ms_per_step": elapsed / steps_timed * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "timed_region_s": elapsed, "irs_per_step": irs_per_step,
            "ranks_seen": dist.get_world_size() if dist is not None else 1, "rccl_ranks_seen": RCCL_RANKS_SEEN,
            "config": {"workload": desc,
                       "stage": ("sweep deconvolution + FIR as the metric names it: K1 deconvolution ('same' crop) -> K3 first peak -> "
                                 "K4 crop (peak - 1 ms, 0.68 s) + Hann fades -> K5 per-channel "
                                 f"{K_fir}-tap FIR, one device chain per call (imp_chain: K1's three passes, the peak search, one "
                                 "fused overlap-save FIR launch); spectra prepared once, outside the timed region.  `deconv_only` = K1 "
                                 "alone; `slice` = the whole hot-path slice end to end") if stage == "chain" else
                                "K1 ONLY: batched sweep deconvolution incl. 'same' crop (inverse-filter spectrum prepared once, "
                                "outside the timed region)",
                       "step": f"one pass over {n_blocks} resident blocks of {mpb} measurement(s) = {B} channels per GPU "
                               f"({n_blocks * rec.nbytes / 2 ** 30:.1f} GiB of inputs in rotation), one call per block, "
                               f"{lanes} calls in flight",
                       "channels_per_gpu_per_measurement": B_meas, "measurements_per_step": n_blocks * mpb,
                       "measurements_per_call": mpb, "channels_per_call": B, "channels_per_launch_group": B // groups_per_call,
                       "sweep_samples": M, "column_samples": L, "fir_taps": K_fir if stage == "chain" else None,
                       "crop_samples": n_fir if stage == "chain" else None,
                       "layout": f"planar fp32, row pitch {pitch} samples (multiple of {PITCH_ALIGN})",
                       "nfft": nfft, "chains_in_flight": lanes, "workspace_channels": plan_ws,
                       "device_memory": "libimpulse_hip (imp_malloc); torch only as the launcher's control plane at N > 1",
                       "environment_switches": {k: v for k, v in sorted(os.environ.items())
                                                if k.startswith(("IMPULSE_BENCH_", "IMPULSE_HIP_"))},
                       "sharding": (f"channels x{world}, no data-path collective; "
                                    f"one {(BROADCAST_VIA or 'RCCL (by libimpulse_hip, no torch in the data path)') if backend == 'nccl' else backend + ' (rehearsal)'} broadcast of "
                                    f"{bcast_bytes[0]} B spectrum at plan creation") if dist is not None else
                                   "single rank: no collective"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
            "deconv_only": dict(value=dec_value, unit="IR/s", steps=dec_steps, timed_region_s=dec_elapsed,
                                path_frac=dec_value / world * 8.0 * L / 1e9 / HBM_PEAK_GBS,
                                layout=f"planar fp32; output rows start {skew} samples into 256-byte aligned buffers so the "
                                       "cropped stores fall on cache lines",
                                plan="pair" if main_paired else "mono",
                                note="K1 alone over the same resident inputs, algorithmic bytes 8 L per IR",
                                other_plan=pair_block),
            "slice": whole_slice, "slice_resident": slice_res, "strong_c5": strong_c5,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not peaks_ok:
        raise SystemExit("parity gate failed: peak indices / cropped spectra / chain outputs do not match the truth")
    return 0


if __name__ == "__main__":
    sys.exit(main())
