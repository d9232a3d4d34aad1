/* impulse_hip.h - C ABI of libimpulse_hip.so (MI355X / gfx950 impulse-response engine).
 *
 * Drop-in boundary for ONE hot path of 115dkk/Impulcifer-pip313 (reference paths are relative to
 * the reference checkout): per-channel ESS sweep deconvolution, IR peak/crop/decay and
 * minimum-phase FIR equalisation.  The reference is pure Python; what it calls on this path is
 * SciPy/NumPy.  Each entry point below names the reference call site it replaces.  The Python
 * host side (package impulse_hip, ctypes) keeps the reference's class surfaces on top of this.
 *
 * Conventions
 *   - every function returns 0 on success or a negative IMP_ERR_* code; the message for the
 *     calling thread is available from imp_last_error()
 *   - no exceptions and no torch / numpy types cross the boundary: plain pointers and sizes
 *   - "host" pointers are ordinary process memory, "device" pointers are HIP device memory of
 *     the context's GPU (e.g. obtained from imp_malloc or any other HIP allocator)
 *   - a context owns one GPU and one stream; calls on one context are stream ordered.  Every entry
 *     point is thread safe: calls that share a context (or plans of one context) serialise on the
 *     context's lock, distinct contexts run concurrently
 *   - the convolutions (K1, K5) are IEEE fp32 on the device; their filter spectra are prepared in fp64 (on the device)
 *     and rounded once; curve conditioning, FIR design, magnitude responses, decay reductions and IIR filtering are fp64
 */
#ifndef IMPULSE_HIP_H_
#define IMPULSE_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMP_OK 0
#define IMP_ERR_INVALID (-1)      /* bad argument */
#define IMP_ERR_HIP (-2)          /* a HIP runtime call failed */
#define IMP_ERR_UNSUPPORTED (-3)  /* size outside what the kernels cover */
#define IMP_ERR_NO_DEVICE (-4)    /* no usable GPU */
#define IMP_ERR_ALLOC (-5)

#define IMP_MODE_SAME 0           /* scipy.signal.convolve(..., mode='same') window */
#define IMP_MODE_FULL 1           /* mode='full' */

typedef struct imp_ctx imp_ctx;
typedef struct imp_plan imp_plan;

/* ---- library / device ---------------------------------------------------------------------- */
const char* imp_version(void);
const char* imp_last_error(void);
int imp_device_count(int* n);
int imp_ctx_create(int device_id, imp_ctx** out);
/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the context's own.  The outgoing stream
 * (owned or external) is drained first: the staging ring and the blocks imp_free has taken back are only ordered against
 * work on the context's ONE stream, so nothing may still be in flight on the old one when later calls queue on the new. */
int imp_ctx_set_stream(imp_ctx* ctx, void* hip_stream);
int imp_ctx_synchronize(imp_ctx* ctx);
void imp_ctx_destroy(imp_ctx* ctx);

/* device memory helpers (so a NumPy-only host needs no other HIP binding).  imp_free takes only pointers imp_malloc of
 * the same context returned and KEEPS the block for the next request of about that size (hipFree costs ~0.4 ms and
 * drains the device), up to IMPULSE_HIP_POOL_MB (default 2048; 0 = give every block back at once).  The free is ordered
 * on the context's stream: it does not wait for work already queued there, and whoever gets the block next reaches it
 * through that same stream (every entry point of this library queues its kernels and copies there), so that work is
 * finished by then.  A context with overlap lanes (imp_plan_set_overlap: work on side streams) drains all its streams
 * first; so does every context under IMPULSE_HIP_FREE_SYNC=1.  A block that OTHER contexts or streams still use must be
 * synchronised by the caller before it is freed, as with hipFreeAsync.  imp_ctx_destroy releases what is kept and what
 * was never handed back. */
int imp_malloc(imp_ctx* ctx, size_t bytes, void** dptr);
int imp_free(imp_ctx* ctx, void* dptr);
int imp_memcpy_h2d(imp_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int imp_memcpy_d2h(imp_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);
int imp_memcpy_d2d(imp_ctx* ctx, void* dst_device, const void* src_device, size_t bytes);  /* async on the ctx stream */
int imp_memset(imp_ctx* ctx, void* dptr, int value, size_t bytes);

/* ---- K1/K5: batched FFT convolution plans ---------------------------------------------------
 * Replaces scipy.signal.convolve(x, h, mode, method='auto') for the sizes where SciPy picks the
 * FFT method:
 *   core/impulse_response_estimator.py:149-151  ImpulseResponseEstimator.estimate  (mode 'same',
 *        h = inverse_filter, shared by every channel; called per column at core/hrir.py:336-354)
 *   core/impulse_response.py:110-119, 126-135    ImpulseResponse.equalize / convolve (mode 'full')
 *   core/parallel_workers.py:9-21                process_plot_worker (mode 'full')
 *
 * A plan fixes (filter, M, L, mode) and is one of three kinds (imp_plan_kind):
 *   0  three-launch transform: circular length nfft = 8192*N1, N1 in {4,8,16,24,32,40,48,64,66,72,80,96,128,144,160,192,256}:
 *      the smallest that covers L+M-1 ('full') or L + M/2 ('same': wrap-around may fall into the part of the linear
 *      convolution that the window discards).  Beyond 2^21 points the plan runs overlap-add: input blocks x filter
 *      partitions of at most 2^21 points each through the same kernels, accumulated on the device (L, M < 2^29).
 *   1  pair mode (imp_conv_plan_create_paired): two channels per complex transform, see below.
 *   2  fused FIR (what imp_conv_plan_create makes for M <= 24 577): overlap-save blocks in ONE launch, see below.
 * ws_channels = channels processed per launch group (0 = choose so that the workspace stays
 * resident in the 256 MiB Infinity Cache).
 */
int imp_conv_plan_create(imp_ctx* ctx,
                         const double* filter,      /* host, [n_filters][filter_ld] */
                         int64_t M,                 /* taps per filter */
                         int64_t n_filters,         /* 1 = shared by all channels, else one per channel */
                         int64_t filter_ld,
                         int64_t L,                 /* samples per input channel */
                         int mode,                  /* IMP_MODE_SAME | IMP_MODE_FULL */
                         int64_t ws_channels,
                         imp_plan** out);
/* Empty plan whose spectrum is filled later (imp_plan_spectrum + a broadcast from another GPU). */
int imp_conv_plan_create_empty(imp_ctx* ctx, int64_t M, int64_t n_filters, int64_t L, int mode,
                               int64_t ws_channels, imp_plan** out);
/* Pair mode (the recording's own layout: core/hrir.py:326-341 hands estimate() the left- and right-ear track of a
 * speaker, columns of one stereo frame block).  The plan's ONE real filter is shared by all channels, so two channels
 * travel through one complex transform, z = x_L + i x_R: (x_L + i x_R) (*) h = y_L + i y_R.  Channels (2q, 2q + 1) of
 * every execute call form pair q (an odd last channel pairs with silence); with interleaved frames the pair is ONE
 * 8-byte load per sample.  Same entry points, same results to fp32 rounding; the circular length is 4096 * N1 samples
 * with N1 <= 384 rows (L + M/2 <= 1 572 864 for 'same': the 96 kHz / 6.6 s configuration takes 288 rows, 2^20-sample
 * sweeps 384 = 16 x 24): longer plans return
 * IMP_ERR_UNSUPPORTED and the caller uses imp_conv_plan_create.  ws_channels counts channels (rounded up to whole pairs). */
int imp_conv_plan_create_paired(imp_ctx* ctx, const double* filter, int64_t M, int64_t L, int mode,
                                int64_t ws_channels, imp_plan** out);
int imp_conv_plan_create_empty_paired(imp_ctx* ctx, int64_t M, int64_t L, int mode, int64_t ws_channels,
                                      imp_plan** out);
int imp_plan_is_paired(const imp_plan* plan, int* paired);
/* Fused FIR plans.  A filter of at most 24 577 taps - every FIR of the path: the 9 600- / 19 200-tap equalisation filters of
 * core/impulse_response.py:110-119, core/hrir.py:858-888 - makes imp_conv_plan_create build a FUSED plan: the convolution
 * runs as overlap-save blocks of 32 768 samples (taps - 1 of history + up to 23 169 new outputs), each block on one
 * workgroup that keeps its four-row transform in registers and LDS (fir_block_kernel): ONE launch per group, no workspace.
 * Same entry points, results and modes; imp_plan_info reports nfft = 32 768, n1_rows = 4.  The general form:
 * flags = IMP_PLAN_PAIRED (pair mode) | IMP_PLAN_NO_FUSED (keep the three-launch transform over the whole input);
 * filter = NULL makes an empty plan (imp_conv_plan_create_empty).  imp_plan_kind: 0 three-launch, 1 pair mode, 2 fused. */
#define IMP_PLAN_PAIRED 1
#define IMP_PLAN_NO_FUSED 2
int imp_conv_plan_create_ex(imp_ctx* ctx, const double* filter, int64_t M, int64_t n_filters, int64_t filter_ld,
                            int64_t L, int mode, int64_t ws_channels, int flags, imp_plan** out);
int imp_plan_kind(const imp_plan* plan, int* kind);
void imp_plan_destroy(imp_plan* plan);
/* geometry queries (pair mode: nfft = the circular length in samples = 4096 * n1_rows) */
int imp_plan_info(const imp_plan* plan, int64_t* nfft, int64_t* out_len, int64_t* ws_channels,
                  int64_t* n1_rows);
/* Device buffer holding the prepared filter spectrum (alpha/beta planes, fp32).  This is the only
 * datum shared between GPUs: rank 0 builds it, the others receive it with an RCCL broadcast. */
int imp_plan_spectrum(imp_plan* plan, void** dptr, size_t* bytes);

/* Single process, several GPUs (the reference's fan-out is a pool over channels inside ONE process, core/parallel_utils.py:
 * 97-152): one context per device, channel pairs sharded over them by the host (impulse_hip/sharding.py), one host thread
 * per device.  The spectrum is prepared on one device and COPIED to the plans of the others (hipMemcpyPeer: xGMI inside a
 * node); dst must be an empty plan of the same geometry (imp_conv_plan_create_ex with filter = NULL).  Returns when the
 * copy has landed. */
int imp_plan_copy_spectrum(imp_plan* dst, imp_plan* src);
/* bytes between buffers of two contexts (different devices or the same): waits for src_ctx's stream, then copies in the
 * order of dst_ctx's stream (asynchronous there) - how deconvolved rows are gathered on the device that runs the later stages */
int imp_memcpy_peer(imp_ctx* dst_ctx, void* dst, imp_ctx* src_ctx, const void* src, size_t bytes);

/* host in / host out, planar: x[B][ld_in] -> y[B][ld_out] (first out_len samples of each row) */
int imp_conv_execute(imp_plan* plan, const float* x, int64_t B, int64_t ld_in, float* y, int64_t ld_out);
/* host in, interleaved frames[L][C] (WAV wire order, core/hrir.py:202-219 columns) -> planar y[C][ld_out] */
int imp_conv_execute_interleaved(imp_plan* plan, const float* frames, int64_t C, float* y, int64_t ld_out);
/* device in / device out, asynchronous on the context stream.
 * x: channel b, sample i at d_x[b*chan_stride_in + i*elem_stride_in]; y: d_y[b*chan_stride_out + i]. */
int imp_conv_execute_device(imp_plan* plan, const float* d_x, int64_t B, int64_t chan_stride_in,
                            int64_t elem_stride_in, float* d_y, int64_t chan_stride_out);

/* device in = raw PCM as it sits in a WAV data chunk (bits = 16 or 32, little endian): sample i of channel
 * b at d_pcm[b*chan_stride_in + i*elem_stride_in] (interleaved frames: chan_stride_in = 1, elem_stride_in =
 * tracks).  The loader scales by 2^-(bits-1) and de-interleaves while packing, replacing the reference's
 * host-side int->float64 conversion and transpose (core/audio_truehd.py:153-185, core/hrir.py:202-219). */
int imp_conv_execute_device_pcm(imp_plan* plan, const void* d_pcm, int bits, int64_t B, int64_t chan_stride_in,
                                int64_t elem_stride_in, float* d_y, int64_t chan_stride_out);

/* Pair-mode plans only: n_pairs channel pairs with explicit geometry (bits = 0: float32 samples, 16 / 32: PCM as in
 * imp_conv_execute_device_pcm).  Sample i of pair q: left at d_x[q*pair_stride + i*elem_stride], right at
 * d_x[q*pair_stride + right_offset + i*elem_stride] (in samples); outputs d_y[(2q + side)*chan_stride_out + i].
 * The columns of a binaural recording (core/hrir.py:202-219, :326-341: WAV frames [n][2], speaker q's column starting at
 * frame s0 + q*step) are pair_stride = 2*step, right_offset = 1, elem_stride = 2: one 8-byte load per stereo frame. */
int imp_conv_execute_device_pairs(imp_plan* plan, const void* d_x, int bits, int64_t n_pairs, int64_t pair_stride,
                                  int64_t right_offset, int64_t elem_stride, float* d_y, int64_t chan_stride_out);

/* Overlapped execution of independent launch groups.  With lanes = n > 1 the workspace is split into n
 * private slices and successive launch groups of imp_conv_execute_device - within one call and across
 * calls - go round robin to n streams (lane 0 = the context stream), so that one group's column pass
 * runs beside another group's row pass instead of the whole chip moving through the same phase.
 * Contract while lanes > 1: inputs must be complete before the call (they are not ordered against
 * earlier work on the context stream), outputs are complete after imp_ctx_synchronize, and two calls
 * in flight must not write the same output memory.  lanes = 1 restores strict stream order. */
int imp_plan_set_overlap(imp_plan* plan, int lanes);

/* Replace the filter(s) of an existing plan: same M, n_filters, L and mode as at creation; `filter` is host fp64,
 * filter f at filter + f*filter_ld.  The alpha/beta planes are recomputed on the device in place (work in flight is
 * drained first), so a caller can keep ONE plan - workspace, tables, buffers - per shape and refill it whenever the
 * FIRs change: HRIR.equalize_channels gets new FIRs for every measurement (core/pipeline.py:690-691). */
int imp_plan_set_filters(imp_plan* plan, const double* filter, int64_t filter_ld);
/* the same with the filters ALREADY ON THE DEVICE (fp64, the context's device: what imp_curves_equalization_fir_device
 * leaves there), for plans with one channel per transform or fused FIR plans, no overlap-add, lanes = 1: nothing is
 * uploaded, nothing waits - the new planes take effect in stream order.  d_filter must stay valid until the stream has
 * passed this call (imp_free on the same context's stream is ordered behind it). */
int imp_plan_set_filters_device(imp_plan* plan, const double* d_filter, int64_t filter_ld);


/* per-kernel timing with HIP events on the plan's stream (for bench.py's roofline block):
 * every_n = 0 switches it off, n >= 1 brackets the three passes of every n-th launch group. */
int imp_plan_set_timing(imp_plan* plan, int every_n);
/* Synchronises, then returns accumulated milliseconds of pass A / B / C and the number of launch
 * groups measured since the last reset. */
int imp_plan_get_timing(imp_plan* plan, double ms[3], int64_t* launches, int reset);

/* Test hooks that need no GPU: the plan geometry chosen for (M, L, mode), and the fp64 host
 * preparation of a filter spectrum (alpha/beta planes, fp32, in the row kernel's register order:
 * ab[k1][kb2*256 + u][4], k2 = (u>>4) + 16*(u&15) + 256*kb2) for n1_rows = nfft / 8192. */
int imp_debug_plan_geometry(int64_t M, int64_t L, int mode, int64_t* nfft, int64_t* out_start,
                            int64_t* out_len);
int imp_debug_host_spectrum(const double* filter, int64_t M, int n1_rows, float* ab_out);
/* the block geometry of a fused FIR plan: samples of history per block (taps - 1 rounded up to even), outputs per block
 * (32 768 - history), the first block that reaches the kept window and the number of blocks per channel;
 * IMP_ERR_UNSUPPORTED beyond 24 577 taps.  imp_debug_plan_geometry always reports the three-launch geometry. */
int imp_debug_plan_geometry_fused(int64_t M, int64_t L, int mode, int64_t* history, int64_t* valid, int64_t* first_block,
                                  int64_t* blocks);
/* the same for a pair-mode plan: nfft = circular length in samples = 4096 * n1_rows; IMP_ERR_UNSUPPORTED beyond 384 rows */
int imp_debug_plan_geometry_paired(int64_t M, int64_t L, int mode, int64_t* nfft, int64_t* out_start, int64_t* out_len,
                                   int64_t* n1_rows);

/* debug: copy the workspace of the last launch group to the host (complex64 [chunk][N1][4096]) */
int imp_plan_debug_run_stage(imp_plan* plan, const float* x, int64_t B, int64_t ld_in, int stage,
                             float* ws_out_host);

/* ---- K2: magnitude response of arbitrary-length rows, fp64 --------------------------------------
 * core/audio_io.py:100-113 magnitude_response: 20*log10(abs(np.fft.rfft(x)[:ceil(n/2)])), no epsilon
 * (exact zeros give -inf).  x: host [B][n] float64 (IR data is float64 in the reference's classes);
 * db_out: host [B][ceil(n/2)].  Any n up to 2^22 (Bluestein on a power-of-two Stockham FFT). */
int imp_magnitude_db(imp_ctx* ctx, const double* x, int64_t B, int64_t n, double* db_out);

/* ---- K7: reductions of the decay analysis -------------------------------------------------------
 * core/decay.py:44-253 (decay_params, Lundeby knee search): the analysis segment [peak, peak + 2 s) is
 * peak-normalised and squared, then np.mean is taken over windows ((n, w) reshape, :111, :152) and over
 * ranges (:117-118, :189) of it, between scalar decisions that stay on the host.
 * imp_segset_create uploads B fp64 segments (x + off[b], len[b] samples), finds max|x| of each and
 * replaces the segment by e = (x / max)^2 (x^2 when max < 1e-20, as the reference skips the division),
 * all in fp64 on the device; maxabs_out[B] may be NULL.
 * imp_segset_range_means answers Q queries mean(e[q_seg][q_a : q_b]) with NumPy's pairwise summation
 * order, so each mean has the bits np.mean gives; an empty range yields NaN.
 */
typedef struct imp_segset imp_segset;
int imp_segset_create(imp_ctx* ctx, const double* x, const int64_t* off, const int64_t* len, int64_t B,
                      imp_segset** out, double* maxabs_out);
int imp_segset_range_means(imp_segset* s, const int64_t* q_seg, const int64_t* q_a, const int64_t* q_b, int64_t Q,
                           double* mean_out);
void imp_segset_destroy(imp_segset* s);

/* core/decay.py:263-340 (decay_times): Schroeder backward integral of B responses and the four decay times
 * read from it.  x: host fp64, response b at x + off[b] (len[b] samples); peak[b], knee[b] (absolute
 * knee_point_ind), noise_floor[b] (dB) and window[b] are decay_params' results.  out: host [B][4] =
 * EDT, RT20, RT30, RT60 in seconds, NaN where the reference returns None.
 */
int imp_decay_times(imp_ctx* ctx, const double* x, const int64_t* off, const int64_t* len, int64_t B,
                    const int64_t* peak, const int64_t* knee, const double* noise_floor, const int64_t* window,
                    double fs, double* out);
/* the same for fp32 rows that are on the device (response b at d_x + off[b]; converted exactly on load, so the result has
 * the bits imp_decay_times gives for the rows' float64 copies); the tables and out are host memory */
int imp_decay_times_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B,
                           const int64_t* peak, const int64_t* knee, const double* noise_floor, const int64_t* window,
                           double fs, double* out);

/* ---- K12: equalisation-curve conditioning ----------------------------------------------------
 * The EQ worker's front half (core/parallel_workers.py:69-131 -> autoeq/frequency_response.py) for all speaker-ear
 * curves of a measurement at once, fp64, one workgroup per curve: fractional-octave smoothing (scipy.signal.
 * savgol_filter(polyorder 2, mode 'interp') as a fixed linear operator per window, logistic blend towards the treble
 * window, autoeq :1060-1105), smoothen_heavy_light (:1181-1239), equalize (:1241-1310: gain-limited inversion, the
 * samples around every clip on/off transition replaced by FITPACK's quadratic interpolating spline in log10 f), the
 * gain grid of the FIR design (:651-674) and, chained on the device, the minimum-phase FIR (K6).
 * A handle belongs to one frequency grid (frequency[n] Hz, strictly increasing, 8 <= n <= 2048); curves are host
 * fp64 [B][n] in dB.  Grid-only quantities (log10 f, window coefficients, blend weights) are prepared once per handle
 * on the host in fp64; all per-curve arithmetic runs on the device. */
typedef struct imp_curves imp_curves;
int imp_curves_create(imp_ctx* ctx, const double* frequency, int64_t n, imp_curves** out);
void imp_curves_destroy(imp_curves* c);
/* FrequencyResponse._window_size (autoeq :1033-1050): odd number of grid points covering `octaves` */
int imp_curves_window_size(imp_curves* c, double octaves, int* window);
/* y = savgol(x, w(window_oct)) * (1 - k) + savgol(x, w(treble_window_oct)) * k, k = logistic between treble_f_lower and
 * treble_f_upper (one iteration each, as every caller on the path asks) */
int imp_curves_smooth(imp_curves* c, const double* x, int64_t B, double window_oct, double treble_window_oct,
                      double treble_f_lower, double treble_f_upper, double* y);
/* error -> [smoothen_heavy_light if smoothen_first] -> equalize(max_gain, treble_f_lower, treble_f_upper, treble_max_gain,
 * treble_gain_k, smoothen = smoothen_kinks).  error_smoothed_out (the curve the inversion used) and spline_used_out
 * ([B], 1 where the kink-bridging spline ran) may be NULL. */
int imp_curves_equalization(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                            double treble_f_lower, double treble_f_upper, double treble_max_gain, double treble_gain_k,
                            int smoothen_kinks, double* error_smoothed_out, double* equalization_out, int* spline_used_out);
/* number of taps minimum_phase_impulse_response(fs, f_res) yields: next_fast_len(round(fs // 2 / (f_res / 2))) */
int imp_curves_fir_taps(imp_curves* c, double fs, double f_res, int64_t* ntaps);
/* equalization [B][n] -> minimum-phase FIRs [B][ntaps] (autoeq :637-681); gain_out (the linear gains handed to
 * firwin2, [B][ntaps]) or fir_out may be NULL */
int imp_curves_fir(imp_curves* c, const double* equalization, int64_t B, double fs, double f_res, int normalize,
                   double* gain_out, double* fir_out);
/* the whole worker: error -> equalization -> FIR without leaving the device; equalization_out may be NULL */
int imp_curves_equalization_fir(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                                double treble_f_lower, double treble_f_upper, double treble_max_gain, double treble_gain_k,
                                int smoothen_kinks, double fs, double f_res, int normalize, double* equalization_out,
                                double* fir_out);

/* the whole worker with the FIRs LEFT ON THE DEVICE (core/pipeline.py:690-691 hands every FIR straight to
 * ImpulseResponse.equalize: they never need to visit the host): *d_fir_out = fp64 [B][*ntaps_out] in a block of the context's
 * pool - hand it back with imp_free - ready in the order of the context's stream; the call does not wait for the device
 * unless equalization_out (host, may be NULL) is asked for. */
int imp_curves_equalization_fir_device(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                                       double treble_f_lower, double treble_f_upper, double treble_max_gain, double treble_gain_k,
                                       int smoothen_kinks, double fs, double f_res, int normalize, double* equalization_out,
                                       void** d_fir_out, int64_t* ntaps_out);

/* ---- K11: cascaded second-order sections -------------------------------------------------------
 * core/virtual_bass.py:121-176: scipy.signal.sosfilt(sos, x) (zero initial state) over every response.
 * sos: host [n_sections][6] = b0 b1 b2 a0 a1 a2 with a0 = 1 (SciPy's layout);
 * x: host fp64, B rows at x + off[b] (len[b] samples); y: host, same layout (may alias x).
 * fp64 with SciPy's operation order and no fused multiply-adds: bit-identical to scipy.signal.sosfilt.
 */
int imp_sosfilt(imp_ctx* ctx, const double* sos, int64_t n_sections, const double* x, const int64_t* off,
                const int64_t* len, int64_t B, double* y);

/* ---- K10: lag search of the ipsilateral alignment ----------------------------------------------
 * core/hrir.py:934-937 and :946-949 (HRIR.align_ipsilateral_all):
 *     corr = scipy.signal.correlate(a, b, mode="full"); lag = arange(-len(a)+1, len(a))[argmax(corr)]
 * for B pairs at once.  a, b: host fp64, pair p at a + a_off[p] (a_len[p] samples) and b + b_off[p]
 * (b_len[p]); corr[k] = sum_l a[l + k - (b_len-1)] b[l] in fp64, arg_out[p] = first index of the
 * maximum (np.argmax), val_out[p] = that maximum (may be NULL).  The caller turns the index into
 * the reference's lag.  a_len[p] + b_len[p] <= 16384 (both segments live in one CU's LDS; the
 * reference's segments are 30 ms: 1 440 / 2 880 samples).
 */
int imp_xcorr_argmax(imp_ctx* ctx, const double* a, const int64_t* a_off, const int64_t* a_len, const double* b,
                     const int64_t* b_off, const int64_t* b_len, int64_t B, int64_t* arg_out, double* val_out);
/* the same for segments of fp32 rows that are on the device (both of a pair at d_x + offset; converted exactly on load, so
 * the sums have the bits imp_xcorr_argmax forms from the rows' float64 copies); tables and results are host memory */
int imp_xcorr_argmax_device(imp_ctx* ctx, const float* d_x, const int64_t* a_off, const int64_t* a_len, const int64_t* b_off,
                            const int64_t* b_len, int64_t B, int64_t* arg_out, double* val_out);
/* ImpulseResponse.shift (core/impulse_response.py:92-108) for fp32 device rows: row b (len[b] samples at d_src +
 * src_off[b]) delayed by shift[b] > 0 (zeros in front, the tail dropped) or advanced by -shift[b] (the head dropped, zeros
 * behind), length kept, written to d_dst + dst_off[b] (not the source rows).  Asynchronous on the context's stream. */
int imp_shift_rows_device(imp_ctx* ctx, const float* d_src, const int64_t* src_off, const int64_t* len, const int64_t* shift,
                          int64_t B, float* d_dst, const int64_t* dst_off);

/* ---- K3: first significant peak ---------------------------------------------------------------
 * core/impulse_response.py:32-70 ImpulseResponse.peak_index (twin core/decay.py:12-41):
 * normalise by max|x| of the searched range, scipy.signal.find_peaks(+x and -x, height), minimum
 * index; no peak -> argmax|x|; max|x| < 1e-20 -> 0.  Indices are relative to each row's start.
 * The reference's fp64 test x/max >= peak_height is turned, once per row, into the smallest fp32 threshold with the
 * same outcome (division by a positive number is monotone), so it stays exact for fp32 data.  peak_height must be
 * positive (the reference's callers pass 0.12589 = -18 dB).
 * x: host, B rows at x + off[b], lengths len[b].
 */
int imp_peak_index(imp_ctx* ctx, const float* x, const int64_t* off, const int64_t* len, int64_t B,
                   double peak_height, int64_t* idx_out, float* maxabs_out);
int imp_peak_index_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len,
                          int64_t B, double peak_height, int64_t* idx_out, float* maxabs_out);

/* ---- K6: minimum-phase FIR design, batched, fp64 -----------------------------------------------
 * Tail of FrequencyResponse.minimum_phase_impulse_response (autoeq/frequency_response.py:676-680),
 * called per channel by core/parallel_workers.py:129:
 *     ir = scipy.signal.firwin2(2n, linspace(0, fs//2, n), gain, fs=fs)      (Hamming window)
 *     fir = scipy.signal.minimum_phase(ir, n_fft=2n)                         (homomorphic, n taps)
 * gain: host [B][n] LINEAR gains on that grid (gain[n-1] must be 0: type II filter, as SciPy demands);
 * fir_out: host [B][n].  2n must factor into 2, 3 and 5 (n comes from next_fast_len).  fs must be even
 * (the grid ends at fs//2 and firwin2 requires it to end at fs/2). */
int imp_minphase_fir(imp_ctx* ctx, const double* gain, int64_t B, int64_t n, double fs, double* fir_out);
/* debug: intermediates of the design. stage 0: out[B][2n] = firwin2 taps; stage 1: |FFT_2n(taps)| */
int imp_debug_minphase_stage(imp_ctx* ctx, const double* gain, int64_t B, int64_t n, double fs, int stage,
                             double* out);

/* test hook: the batched fp64 complex transform K6, K2 and the filter-spectrum preparation share.  x, y: host [B][N]
 * complex128 (interleaved re, im); dir = -1 forward, +1 inverse, unscaled.  N = 2^a 3^b 5^c 11^d.  Lengths that split into
 * one or two factors of at most 1024 points run as tile transforms held in LDS (csrc/fft64.hip.h: one or two launches);
 * *used_tiles (may be NULL) says whether this one did; IMPULSE_HIP_FFT64_GENERIC=1 forces the launch-per-radix-pass form. */
int imp_debug_fft64(imp_ctx* ctx, const double* x, int64_t B, int64_t N, int dir, double* y, int* used_tiles);

/* ---- K4/K8: in-place gain, fades, decay window (elementwise) ---------------------------------
 * core/hrir.py:530-544 (gain), :591-612 (Hann fade-in), :642-651 (crop + Hann fade-out),
 * core/decay.py:383-403 apply_decay_window.
 * y[i] = x[i] * gain * fade_in(i) * fade_out(i) * decay(i) for i < len[b]; any stage may be
 * disabled (fade length 0, decay half_window < 0).
 */
typedef struct imp_window_params {
  float gain;              /* linear */
  int64_t fade_in;         /* samples: hann(2*fade_in)[:fade_in] on the head */
  int64_t fade_out;        /* samples: hann(2*fade_out)[fade_out:] on the tail ending at len */
  int64_t decay_start;     /* window_start */
  int64_t decay_half;      /* half_window (< 0 disables) */
  int64_t decay_knee;      /* knee_point_index: zero level beyond */
  float decay_level_db;    /* window_level */
} imp_window_params;
int imp_apply_window(imp_ctx* ctx, float* x, const int64_t* off, const int64_t* len, int64_t B,
                     const imp_window_params* params /* [B] */);

/* ---- device-resident responses ----------------------------------------------------------------
 * The stages between ingest and output (crop_heads, crop_tails, equalize, normalize: core/hrir.py:457-653, 858-888)
 * on responses that STAY on the device as fp32 rows: a row is (device pointer, offset, length) and a head / tail crop
 * is a change of offset / length.  Together with imp_conv_execute_device(_pcm) and imp_peak_index_device these let
 * the host class keep a measurement on the GPU from the WAV's PCM block to the final responses. */
/* window parameters as imp_apply_window; row b is read at d_src + src_off[b] and written at d_dst + dst_off[b]
 * (len[b] samples; the two may be the same memory for an in-place window) */
int imp_apply_window_device(imp_ctx* ctx, const float* d_src, const int64_t* src_off, float* d_dst,
                            const int64_t* dst_off, const int64_t* len, int64_t B, const imp_window_params* params);
/* imp_segset_create on fp32 device rows (converted exactly to fp64 on the device).  With maxabs_out = NULL the call does
 * not wait for the device (the first imp_segset_range_means does): one round trip less per knee search. */
int imp_segset_create_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B,
                             imp_segset** out, double* maxabs_out);
/* HRIR.crop_tails (core/hrir.py:585-611) asks every response for decay_params()[1], the Lundeby knee INDEX
 * (core/decay.py:44-260).  For B fp32 device rows this runs the peak search (K3) and the whole knee search on the
 * device, one stream-ordered sequence of launches and one readback: spans [peak, peak + int(2 fs)), e = (x / max)^2,
 * 30 ms window levels, line fit, knee estimate, three windows per 10 dB, up to five refinements.  np.log10 and
 * linregress's BLAS dot product cannot be reproduced to the last bit (and the window means here are plain tree sums,
 * ~1e-14 dB from NumPy's), so every decision of the search carries a guard band from explicit error bounds and
 * flags_out[b] != 0 says "row b has a decision inside its band (1) or a shape outside the device path's limits (2):
 * ask the host search" - for rows with flags_out[b] == 0 knee_out[b] IS the host search's integer.  peak_out[b] is
 * exact for every row; floor_out[b] (dB) is within a few ulp of the host's, window_out[b] its window size. */
int imp_decay_knees_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B, double fs,
                           double peak_height, int64_t* peak_out, int64_t* knee_out, double* floor_out,
                           int64_t* window_out, int32_t* flags_out);
/* HRIR.write_wav (core/hrir.py:426-455 -> core/audio_io.py:82-97) for responses on the device: the interleaved PCM
 * block [n_frames][n_tracks] of a WAV data chunk, track t taken from row row_of_track[t] (-1: silence; samples beyond a
 * row's end: silence), converted as soundfile / libsndfile's clip path does: clip(lrint(x * 2^31), -2^31, 2^31 - 1)
 * >> (32 - bits) (round half to even, saturating; PCM_32 pinned by the sweep WAVs the reference ships, tests/golden/
 * sweep_wavs.npz).
 * pcm_out: host, int16 for bits = 16, int32 for bits = 24 and 32 (24-bit values sign-extended). */
int imp_rows_to_pcm_device(imp_ctx* ctx, const float* d_rows, const int64_t* off, const int64_t* len, int64_t n_rows,
                           const int64_t* row_of_track, int64_t n_tracks, int64_t n_frames, int bits, void* pcm_out);
/* magnitude response (as imp_magnitude_db, n points, ceil(n/2) bins each) of the per-group SUMS of device rows:
 * rows of group g are added in row order in fp64, zero beyond their end - np.sum(np.vstack(padded), axis=0) of
 * HRIR.normalize (core/hrir.py:496-503).  db_out: host [n_groups][ceil(n/2)]. */
int imp_magnitude_db_sum_device(imp_ctx* ctx, const float* d_rows, const int64_t* off, const int64_t* len,
                                const int64_t* group, int64_t n_rows, int64_t n_groups, int64_t n, double* db_out);
/* the same sums, but only np.max of each spectrum comes back (HRIR.normalize with peak_target reads nothing else,
 * core/hrir.py:505): peak_db_out[n_groups]; NaN if a spectrum holds a NaN, -inf for an all-zero sum, as np.max. */
int imp_magnitude_db_sum_peak_device(imp_ctx* ctx, const float* d_rows, const int64_t* off, const int64_t* len,
                                     const int64_t* group, int64_t n_rows, int64_t n_groups, int64_t n,
                                     double* peak_db_out);

/* ---- K1 -> K3 -> K4 -> K5 as one stream-ordered chain ("deconvolution + FIR") -------------------------------
 * recording (device) -> estimate() (core/impulse_response_estimator.py:149-151) -> first peak
 * (core/impulse_response.py:32-70) -> crop at peak - head, fir->L samples, Hann fade-in / fade-out
 * (core/impulse_response.py:82-90, core/hrir.py:591-612, 642-651 at a fixed length) -> per-channel FIR
 * (core/impulse_response.py:110-119), with NO host round trip: the crop offsets are taken from the peak search on the
 * device.  `deconv` is a 'same' plan, `fir` a 'full' plan of length n (its filters may be refilled with
 * imp_plan_set_filters between calls); both must be in stream order (lanes = 1) and outlive the chain.
 * imp_chain_execute_device is asynchronous on the context stream; d_out receives B rows of n + K - 1 samples,
 * d_peaks_out (device, may be NULL) the B peak indices.  Seven launches: the deconvolution's last pass also leaves the
 * row maxima the peak search starts from, and the FIR's first pass reads the cropped, faded responses in place.
 * Overlap-add plans cannot be chained. */
typedef struct imp_chain imp_chain;
int imp_chain_create(imp_plan* deconv, imp_plan* fir, int64_t B, int64_t head, int64_t fade_in, int64_t fade_out,
                     double peak_height, imp_chain** out);
int imp_chain_execute_device(imp_chain* chain, const float* d_x, int64_t chan_stride_in, int64_t elem_stride_in,
                             float* d_out, int64_t chan_stride_out, long long* d_peaks_out);
void imp_chain_destroy(imp_chain* chain);

/* ---- the reference's stage sequence, device resident and batched ----------------------------------------------------
 * core/pipeline.py:565-573 (open measurements) -> :585-601 (crop_heads, crop_tails) -> :647-692 (equalize) -> :725-735
 * (normalize), i.e. core/hrir.py:307-355, :548-612, :614-653, core/impulse_response.py:110-119, core/hrir.py:457-546, for
 * M measurements per call with NO host readback between the stages:
 *   ingest      every column of every recording through K1 (the plan's filter = the estimator's inverse filter)
 *   crop_heads  first peak of every row (K3); per ear pair the earlier peak minus (speaker delay + head), both rows of the
 *               pair cropped there, Hann fade-in of `head` samples (skipped for a pair shorter than head, as the reference)
 *   crop_tails  peak + Lundeby knee of every cropped row (K7c); per measurement min(shortest row, next_fast_len(latest
 *               knee)), rows truncated to it with a Hann fade-out of `fade_out` samples
 *   equalize    row r of every measurement convolved with FIR r ('full': keep + taps - 1 samples, K5 in one launch)
 *   normalize   np.max of the magnitude response of each ear's sum (K2, fp64) -> gain = -max + peak_target, every row
 *               scaled by 10^(gain / 20)
 * Every length and offset a later stage needs is left in device memory by the stage that decides it; the launch sequence
 * is sized for the worst case the slice was made for (rows no longer than keep_cap after crop_tails) and workgroups past a
 * measurement's actual lengths exit at once.  The scalars come back once, with imp_slice_results.
 *
 * A measurement = n_pairs ear pairs; pair q is rows 2q (left) and 2q + 1 (right) of the measurement, in the order
 * HRIR.irs lists the speakers.  Its recording samples: left ear sample i at base[pair_offset[q] + i * elem_stride], right
 * ear at + 1 (adjacent tracks of interleaved frames, core/hrir.py:326-341); the files of a measurement are laid out one
 * after the other in one device block.  Measurement m of a call starts rec_stride samples after measurement m - 1.
 *
 * Decisions the device cannot promise to take as the host would are FLAGGED, never guessed (result.flags != 0): a knee
 * search with a decision inside its guard band (IMP_SLICE_KNEE_GUARD / _RANGE, see imp_decay_knees_device), a crop_tails
 * length above keep_cap, a fade-out longer than the cropped rows (the reference raises), a gain whose fp32 rounding
 * depends on the last ulps of pow() or of the transform (IMP_SLICE_GAIN_GUARD), empty / non-finite spectra.  The rows of
 * a flagged measurement are not valid: the caller runs that measurement through the staged entry points instead (the
 * Python host does).  For measurements with flags == 0 the rows are bit-identical to the staged sequence
 * imp_conv_execute_device_pairs -> imp_peak_index_device -> imp_apply_window_device -> imp_decay_knees_device ->
 * imp_apply_window_device -> imp_conv_execute_device (fused FIR plan) -> imp_apply_window_device(gain). */
#define IMP_SLICE_KNEE_GUARD 1
#define IMP_SLICE_KNEE_RANGE 2
#define IMP_SLICE_KEEP_CAP 4
#define IMP_SLICE_FADE 8
#define IMP_SLICE_GAIN_GUARD 16
#define IMP_SLICE_GAIN_NONFINITE 32
#define IMP_SLICE_SHORT 64          /* informational: a pair shorter than the head fade kept its head un-faded */
#define IMP_SLICE_ALIGN_GUARD 256   /* alignment: a row shorter than the correlation segment, an all-zero row, or a delayed row
                                     * whose first sample is not zero: the host flow decides on the materialised rows */
#define IMP_SLICE_DECAY_GUARD 128   /* decay adjustment: a knee search in its guard band, no decay time defined, or a knee
                                     * before the window's start (the reference raises): the host flow decides */
typedef struct imp_slice imp_slice;
typedef struct imp_slice_geometry {
  int64_t n_pairs;               /* ear pairs per measurement */
  int64_t elem_stride;           /* samples between consecutive frames of one ear (= tracks of the recordings) */
  int bits;                      /* 16 / 32: PCM frames as in imp_conv_execute_device_pcm */
  const int64_t* pair_offset;    /* [n_pairs] sample offset of the left ear's first sample from the measurement's base */
  const int64_t* delay;          /* [n_pairs] int(round(SPEAKER_DELAYS[speaker] * fs)) + head (core/hrir.py:566-567) */
  int64_t head;                  /* int(head_ms * fs / 1000) */
  int64_t fade_out;              /* int(fs * seconds_per_octave / 24): hann(2 fade_out)[fade_out:] (core/hrir.py:636-638) */
  int64_t taps;                  /* FIR length (<= 24 577) */
  int64_t keep_cap;              /* longest crop_tails length the slice is sized for */
  double fs;
  double peak_height;            /* 0.12589 */
  double peak_target_db;         /* normalize(peak_target=...) */
  double gain_guard_rel;         /* 0 = default 1e-10: relative band around fp32 rounding boundaries of the gain */
} imp_slice_geometry;
typedef struct imp_slice_row_result {
  int64_t peak;                  /* peak_index of the deconvolved column */
  int64_t cut;                   /* samples crop_heads removed */
  int64_t len;                   /* length after crop_heads */
  int64_t knee;                  /* decay_params()[1] of the cropped row */
  int32_t knee_flags;            /* flags_out of imp_decay_knees_device */
  int32_t knee_why;              /* diagnostic: which decision of the search fell into its guard band first (0: none) */
  int64_t decay_peak;            /* rows with a decay target (imp_slice_set_decay): decay_params()[0] of the equalized row */
  int64_t decay_knee;            /* ... and [1] */
  double decay_slope;            /* measured slope in dB/s from the longest defined decay time (NaN: none defined) */
  double decay_level_db;         /* the window's level at the knee as applied (core/decay.py:375) */
  int32_t decay_state;           /* 0 no target, 1 adjusted, 2 already faster than the target, 3 left to the host flow */
  int32_t decay_flags;           /* flags_out of imp_decay_knees_device for that search */
  int64_t shift_ipsilateral;     /* alignment (imp_slice_set_alignment): samples align_ipsilateral_all delayed the row by */
  int64_t shift_onset;           /* ... and the signed shift align_onset_groups_peak_leftref gave it afterwards */
} imp_slice_row_result;
typedef struct imp_slice_result {
  int64_t keep;                  /* crop_tails' return value */
  int64_t out_len;               /* keep + taps - 1: samples per output row */
  double peak_db[2];             /* left, right: np.max of the ear sum's magnitude response */
  double gain_db;                /* normalize's return value */
  float gain;                    /* 10^(gain_db / 20) as applied */
  int32_t flags;                 /* IMP_SLICE_* */
} imp_slice_result;
/* deconv: a 'same' plan (mono or pair mode) of the column length on the slice's context, lanes = 1; it must outlive the slice */
int imp_slice_create(imp_plan* deconv, const imp_slice_geometry* geometry, int64_t max_measurements, imp_slice** out);
void imp_slice_destroy(imp_slice* slice);
int imp_slice_info(const imp_slice* slice, int64_t* rows_per_measurement, int64_t* max_measurements, int64_t* out_len_max,
                   int64_t* norm_fft_len);
/* the FIRs of a job (host fp64 [2 n_pairs][ld], row r = FIR of row r of every measurement): the curves are per job,
 * core/pipeline.py:668-688 designs them once.  Drains the stream. */
int imp_slice_set_firs(imp_slice* slice, const double* firs, int64_t ld);
/* the same from FIRs on the device (imp_curves_equalization_fir_device): no upload, no wait */
int imp_slice_set_firs_device(imp_slice* slice, const double* d_firs, int64_t ld);
/* The alignment between crop_heads and crop_tails (core/pipeline.py:593-597 -> core/hrir.py:921-1001), on the device:
 *   align_ipsilateral_all            per ipsilateral pair p = (ear pair ipsi_first[p], ear pair ipsi_second[p]) the lag of the
 *                                    full cross-correlation of the first `segment` samples of the first's left ear and the
 *                                    second's right ear (K10); lag > 0 delays the second (both ears; first == second: its
 *                                    right ear only), lag < 0 the first.  No ear pair may appear in two entries (the
 *                                    reference's IPSILATERAL_PAIRS share no speaker): the searches of a measurement then
 *                                    all read the rows as crop_heads left them.
 *   align_onset_groups_peak_leftref  the rows of ear pair q are shifted by -(peak of leader_of_pair[q]'s left ear - peak of
 *                                    ref_pair's left ear), peaks taken after the first alignment; leader_of_pair[q] < 0:
 *                                    no shift (the reference group, speakers outside the groups, groups whose first
 *                                    speaker is absent)
 * Both shifts keep a row's length (ImpulseResponse.shift); the aligned rows are materialised once and the later stages read
 * them.  n_ipsi = 0 with leader_of_pair = NULL switches the stage off.  A measurement the device cannot promise to align as
 * the host flow would is flagged IMP_SLICE_ALIGN_GUARD.  Bit-identical to imp_xcorr_argmax_device -> imp_shift_rows_device
 * -> imp_peak_index_device -> imp_shift_rows_device.  Drains the stream. */
int imp_slice_set_alignment(imp_slice* slice, int64_t n_ipsi, const int32_t* ipsi_first, const int32_t* ipsi_second,
                            const int32_t* leader_of_pair, int32_t ref_pair, int64_t segment);
/* The optional stage between equalize and normalize (core/pipeline.py:694-716 -> core/parallel_workers.py:24-39 ->
 * core/decay.py:355-403): target_rt60[2 n_pairs], the target 60 dB decay time in seconds of row r of every measurement, NaN
 * for rows to leave alone (the reference's `decay` dict is per speaker); NULL or all NaN switches the stage off.  Per row
 * with a target, on the equalized row and on the device: decay_params (K3 + K7c), decay_times (K7b), the slope of the
 * longest defined decay time, no change if the response already decays faster than the target, else the window
 * ones | falling Hann from peak + 2 ms to the knee | zeros, scaled to the level difference at the knee, applied in place
 * (K8).  Where the reference raises (no decay time defined: TypeError; knee before the window's start: ValueError) or the
 * knee search has a decision inside its guard band, the measurement is flagged IMP_SLICE_DECAY_GUARD.  Bit-identical to
 * imp_decay_knees_device -> imp_decay_times_device -> imp_apply_window_device on the same rows.  Drains the stream. */
int imp_slice_set_decay(imp_slice* slice, const double* target_rt60);
/* asynchronous on the context's stream; d_out: [M * 2 n_pairs][out_pitch] fp32, out_pitch >= keep_cap + taps - 1; row
 * m * 2 n_pairs + r holds result.out_len valid samples */
int imp_slice_execute_device(imp_slice* slice, const void* d_rec, int64_t rec_stride, int64_t M, float* d_out,
                             int64_t out_pitch);
/* waits for the last call and copies its scalars: rows_out [M * 2 n_pairs], meas_out [M] (either may be NULL) */
int imp_slice_results(imp_slice* slice, imp_slice_row_result* rows_out, imp_slice_result* meas_out);
/* The rows of the last call as float64, packed: measurement m's 2 n_pairs rows at d_packed + m * meas_stride, row r at
 * + r * out_len(m) with out_len(m) valid samples each (the layout a [2 n_pairs][out_len] float64 host array has, so that one
 * linear copy of 2 n_pairs * out_len * 8 bytes per measurement brings the responses over as ImpulseResponse.data holds
 * them, core/impulse_response.py:21-27).  meas_stride >= 2 n_pairs * (keep_cap + taps - 1).  The conversion float32 ->
 * float64 is exact; asynchronous on the context's stream, after the call that wrote d_out. */
int imp_slice_pack_f64(imp_slice* slice, const float* d_out, int64_t out_pitch, int64_t M, double* d_packed, int64_t meas_stride);

/* page-locked host memory (results the link writes straight into: imp_memcpy_d2h to it needs no staging and its pages
 * are mapped once, not per job); imp_host_free takes any pointer imp_host_alloc returned, from any thread */
int imp_host_alloc(imp_ctx* ctx, size_t bytes, void** out);
int imp_host_free(void* p);

/* ---- the one collective: RCCL broadcast of the prepared filter spectrum -----------------------------
 * Channels shard across GPUs with no data-path collective; the only shared datum is the inverse-sweep spectrum rank 0
 * prepares.  The library does that broadcast itself over RCCL (xGMI inside a node), so the host side needs no
 * communication package: rank 0 calls imp_comm_unique_id and hands the 128 bytes to the other ranks by whatever the
 * launcher offers (an environment variable, a file, a socket); every rank then calls imp_comm_create with its rank
 * and the world size (collective), imp_plan_broadcast_spectrum on a plan of the same geometry (rank != root: made
 * with imp_conv_plan_create_empty), and imp_comm_destroy.  librccl is opened on first use. */
typedef struct imp_comm imp_comm;
int imp_comm_unique_id(unsigned char id_out[128]);
int imp_comm_create(imp_ctx* ctx, const unsigned char id[128], int rank, int nranks, imp_comm** out);
void imp_comm_destroy(imp_comm* c);
/* 0 if librccl can be loaded and has every entry point used here (no communicator, no device work).  ncclCommInitRank
 * has no timeout, so the ranks should agree on this over the launcher's control plane BEFORE any of them creates a
 * communicator. */
int imp_comm_probe(void);
/* ranks of the communicator as RCCL counts them (ncclCommCount) */
int imp_comm_nranks(imp_comm* c, int* nranks);
/* in-place broadcast of `bytes` at device pointer dptr from rank `root`; returns when the data has arrived */
int imp_comm_broadcast(imp_comm* c, void* dptr, size_t bytes, int root);
/* imp_plan_spectrum + imp_comm_broadcast; *bytes_out (may be NULL) = bytes moved */
int imp_plan_broadcast_spectrum(imp_plan* plan, imp_comm* c, int root, size_t* bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* IMPULSE_HIP_H_ */
