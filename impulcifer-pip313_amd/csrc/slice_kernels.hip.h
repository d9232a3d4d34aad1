// imp_slice: the reference's stage sequence of a measurement (core/pipeline.py:565-573, 585-601, 647-692, 725-735) with
// every decision taken where the data is.  The kernels here are the glue between the stages that already run on the device
// (K1, K3, K7c, K4, K5, K2): each reads the previous stage's results from device memory and leaves the row tables and
// window parameters of the next one there, so a call of M measurements is one stream-ordered sequence of launches and
// the scalars (peaks, crop indices, knees, lengths, gains, flags) come back once, at the end.
//   crop_heads   core/hrir.py:548-612   earlier ear's first peak - (speaker delay + head) per pair, Hann fade-in
//   crop_tails   core/hrir.py:614-653   min(shortest row, next_fast_len(latest Lundeby knee)), Hann fade-out
//   normalize    core/hrir.py:457-546   -max(spectrum of the ear sums) + peak_target, 10^(gain / 20) on every row
// Included after decay_kernels.hip.h: fp contraction is off, the integer and fp64 scalar work below is plain IEEE.
#pragma once

namespace imp {

// flags of a measurement (include/impulse_hip.h IMP_SLICE_*)
enum {
  SLICE_KNEE_GUARD = 1,      // a row's knee search has a decision inside its guard band: the host search decides
  SLICE_KNEE_RANGE = 2,      // a row's knee search is outside the device path's limits
  SLICE_KEEP_CAP = 4,        // crop_tails' length exceeds the capacity the slice was made for
  SLICE_FADE = 8,            // fade-out longer than the cropped response (the reference raises ValueError)
  SLICE_GAIN_GUARD = 16,     // 10^(gain / 20) is within the guard band of an fp32 rounding boundary
  SLICE_GAIN_NONFINITE = 32, // all-zero or NaN spectra
  SLICE_SHORT = 64,          // a row shorter than the head fade (the reference skips the fade for that pair)
  SLICE_DECAY_GUARD = 128,   // decay adjustment: a knee search in its guard band, no decay time defined (the reference raises
                             // TypeError), or a knee before the window's start (ValueError): the host flow decides / raises
  SLICE_ALIGN_GUARD = 256,   // alignment: a row shorter than the correlation segment, an all-zero row, or a delayed row whose
                             // first sample is not zero (the host flow decides on the materialised rows)
};

struct SliceRowOut {         // per row, returned to the host at the end (imp_slice_row_result)
  long long peak;            // ImpulseResponse.peak_index of the deconvolved column
  long long cut;             // samples cropped from its head
  long long len;             // length after crop_heads
  long long knee;            // decay_params()[1] of the cropped row
  int knee_flags;            // KNEE_*
  int knee_why;              // diagnostic: KneeRow::why
  long long decay_peak;      // decay adjustment (rows with a target): decay_params()[0:2] of the equalized row ...
  long long decay_knee;
  double decay_slope;        // ... the measured slope in dB/s (from the longest decay time that is defined; NaN: none) ...
  double decay_level_db;     // ... and the window's level at the knee, as applied (fp32)
  int decay_state;           // 0 no target, 1 adjusted, 2 already faster than the target (left alone), 3 flagged
  int decay_flags;           // KNEE_* of that search
  long long shift_ipsilateral;   // alignment (imp_slice_set_alignment): samples align_ipsilateral_all delayed the row by (>= 0) ...
  long long shift_onset;         // ... and the signed shift align_onset_groups_peak_leftref gave it afterwards
};

struct SliceMeasOut {        // per measurement (imp_slice_result)
  long long keep;            // crop_tails' return value
  long long out_len;         // keep + taps - 1
  double peak_db[2];         // np.max of the left / right ear sum's magnitude response
  double gain_db;            // what normalize returns
  float gain;                // 10^(gain_db / 20) as applied (fp32 rows)
  int flags;                 // SLICE_*
};

// ---- crop_heads -------------------------------------------------------------------------------------------------------
// One thread per ear pair.  Rows 2q / 2q + 1 of d_ir are the left / right ear (pitch_ir apart, row_len samples).
__global__ __launch_bounds__(64) void slice_crop_heads_kernel(const RowPeak* __restrict__ res, const long long* __restrict__ delay,
                                                              int pairs_per_meas, int n_pairs_total, long long pitch_ir,
                                                              long long row_len, long long head,
                                                              int64_t* __restrict__ off2, int64_t* __restrict__ len2,
                                                              int64_t* __restrict__ fade_len, WindowParams* __restrict__ fade_par,
                                                              SliceRowOut* __restrict__ rows, int* __restrict__ meas_flags) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_pairs_total) return;
  long long pk[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const RowPeak rp = res[2 * q + s];
    if (row_len == 0 || !(__uint_as_float(rp.maxabs_bits) >= 1e-20f)) pk[s] = 0;            // EPSILON rule
    else pk[s] = (long long)(rp.first_peak != ~0ull ? rp.first_peak : rp.first_max);     // argmax fallback
  }
  const long long first = pk[0] < pk[1] ? pk[0] : pk[1];       // (equal peaks take the reference's else branch: same index)
  long long at = first - delay[q % pairs_per_meas];
  if (at < 0) at = 0;
  const long long cut = at < row_len ? at : row_len;
  const long long n = row_len - cut;
  const bool fade = n >= head && head > 0;
  if (!fade && head > 0) atomicOr(&meas_flags[q / pairs_per_meas], SLICE_SHORT);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int b = 2 * q + s;
    off2[b] = (long long)b * pitch_ir + cut;
    len2[b] = n;
    fade_len[b] = fade ? head : 0;
    WindowParams p;
    p.gain = 1.0f;
    p.fade_in = fade ? head : 0;
    p.fade_out = 0;
    p.decay_start = 0;
    p.decay_half = -1;
    p.decay_knee = 0;
    p.decay_level_db = 0.f;
    fade_par[b] = p;
    SliceRowOut r = {};
    r.peak = pk[s];
    r.cut = cut;
    r.len = n;
    r.knee = 0;
    r.knee_flags = 0;
    r.knee_why = 0;
    rows[b] = r;
  }
}

// scipy.fftpack.next_fast_len: the smallest 2^a 3^b 5^c >= n (n <= 6: n itself)
__device__ inline long long slice_next_fast_len(long long n) {
  if (n <= 6) return n;
  long long best = 1;
  while (best < n) best <<= 1;
  for (long long p5 = 1; p5 < best; p5 *= 5)
    for (long long p35 = p5; p35 < best; p35 *= 3) {
      long long q = p35;
      while (q < n) q <<= 1;
      if (q < best) best = q;
    }
  return best;
}

// ---- crop_tails: the common length of a measurement (the truncation and the fade-out happen in K5's loader) ------------
// One thread per measurement.
__global__ __launch_bounds__(64) void slice_keep_kernel(const KneeRow* __restrict__ knee, const int64_t* __restrict__ off2,
                                                        const int64_t* __restrict__ len2, int rows_per_meas, int n_meas,
                                                        long long fade_out, long long keep_cap, long long taps,
                                                        long long* __restrict__ keep_out, long long* __restrict__ out_len,
                                                        SliceRowOut* __restrict__ rows, SliceMeasOut* __restrict__ meas,
                                                        int* __restrict__ meas_flags) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_meas) return;
  int flags = meas_flags[m];
  long long knee_max = 0, len_min = 0x7fffffffffffffffll;
  for (int r = 0; r < rows_per_meas; ++r) {
    const int b = m * rows_per_meas + r;
    const KneeRow k = knee[b];
    const int kf = k.done ? k.flags : (k.flags ? k.flags : (int)KNEE_GUARD);
    if (kf & KNEE_GUARD) flags |= SLICE_KNEE_GUARD;
    if (kf & KNEE_RANGE) flags |= SLICE_KNEE_RANGE;
    rows[b].knee = k.knee;
    rows[b].knee_flags = kf;
    rows[b].knee_why = k.why;
    knee_max = k.knee > knee_max ? k.knee : knee_max;
    len_min = len2[b] < len_min ? len2[b] : len_min;
  }
  long long keep = slice_next_fast_len(knee_max);
  if (len_min < keep) keep = len_min;
  if (keep < 0) keep = 0;
  if (fade_out > keep) flags |= SLICE_FADE;
  if (keep > keep_cap) {
    flags |= SLICE_KEEP_CAP;
    keep = keep_cap;                                     // (the outputs of a flagged measurement are not used)
  }
  keep_out[m] = keep;
  out_len[m] = keep > 0 ? keep + taps - 1 : 0;
  meas[m].keep = keep;
  meas[m].out_len = keep > 0 ? keep + taps - 1 : 0;
  meas_flags[m] = flags;
}

// ---- normalize: gain from the two ear maxima, tables of the in-place gain launch ---------------------------------------
// One thread per measurement.  gain = np.max([m_l, m_r]) * -1 + peak_target; rows *= 10 ** (gain / 20) (fp32 rows: the
// gain is rounded to fp32 once).  The device's pow is not Python's: a value whose fp32 rounding depends on the last few
// ulp of the fp64 result, or on the ~1e-12 dB by which two correct transforms of the spectrum differ, is flagged and the
// host path decides that measurement.
__global__ __launch_bounds__(64) void slice_gain_kernel(const double* __restrict__ peak_db /*[n_meas][2]*/,
                                                        const long long* __restrict__ out_len, int rows_per_meas, int n_meas,
                                                        double peak_target, double guard_rel, long long out_pitch,
                                                        int64_t* __restrict__ g_off, int64_t* __restrict__ g_len,
                                                        WindowParams* __restrict__ g_par, SliceMeasOut* __restrict__ meas,
                                                        int* __restrict__ meas_flags) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_meas) return;
  int flags = meas_flags[m];
  const double ml = peak_db[2 * m], mr = peak_db[2 * m + 1];
  const double top = (ml != ml || mr != mr) ? __longlong_as_double(0x7ff8000000000000ll) : (ml > mr ? ml : mr);   // np.max
  const double gain_db = top * -1 + peak_target;
  const double g = pow(10.0, gain_db / 20);
  float g32 = (float)g;
  if (!(g == g) || !(fabs(g) < 1e300) || !(fabs(gain_db) < 1e300) || out_len[m] <= 0) {
    flags |= SLICE_GAIN_NONFINITE;
    g32 = 1.0f;
  } else if ((float)(g * (1.0 + guard_rel)) != g32 || (float)(g * (1.0 - guard_rel)) != g32) {
    flags |= SLICE_GAIN_GUARD;
  }
  for (int r = 0; r < rows_per_meas; ++r) {
    const int b = m * rows_per_meas + r;
    g_off[b] = (long long)b * out_pitch;
    g_len[b] = out_len[m];
    WindowParams p;
    p.gain = g32;
    p.fade_in = 0;
    p.fade_out = 0;
    p.decay_start = 0;
    p.decay_half = -1;
    p.decay_knee = 0;
    p.decay_level_db = 0.f;
    g_par[b] = p;
  }
  meas[m].peak_db[0] = ml;
  meas[m].peak_db[1] = mr;
  meas[m].gain_db = gain_db;
  meas[m].gain = g32;
  meas[m].flags = flags;
  meas_flags[m] = flags;
}

// K5's view of the rows crop_tails leaves: row b starts at off2[b] of the deconvolved columns (crop_heads put it there),
// is keep[b / rows_per_meas] samples long (crop_tails decided that on the device) and ends in a Hann fade-out of `fade_out`
// samples - hann(2 fade_out)[fade_out:], read from a table (a cosine per sample in this kernel would cost it a stack
// frame), applied as the window kernel of the staged path applies it: (float)((double)x * g).  Its FIR is filter
// b % rows_per_meas.  No compacted copy of the rows is made: the truncation is the buffer's range, the fade happens on
// load.  fir_block_kernel reads lengths and the block count through these hooks.
struct LoadRowsDeviceLen {
  static constexpr bool kDeviceLen = true;
  const float* __restrict__ base;           // the deconvolved columns
  const int64_t* __restrict__ off;          // [rows] first sample of every row
  const long long* __restrict__ len_of;     // [measurements]
  int rows_per_meas;
  long long taps;
  long long fade_out;
  const double* __restrict__ win;           // [fade_out]
  __host__ __device__ LoadRowsDeviceLen shifted(long long, long long) const { return *this; }
  struct Row {
    __amdgpu_buffer_rsrc_t r;
    int n, fade;
    const double* __restrict__ win;
    __device__ __forceinline__ cf pair_at(int s) const { return bload_cf<kStreamAux>(r, (unsigned)s * 4u, 0u); }
    __device__ __forceinline__ cf finish(cf v, int s) const {
      if (fade > 0 && s + 1 >= n - fade && s < n) {          // only the tail pays for the window
        if (s >= n - fade) v.x = (float)((double)v.x * win[s - (n - fade)]);
        if (s + 1 < n) v.y = (float)((double)v.y * win[s + 1 - (n - fade)]);
      }
      return v;
    }
  };
  __device__ __forceinline__ long long row_len(int b) const { return len_of[b / rows_per_meas]; }
  __device__ __forceinline__ long long out_len(int b) const {
    const long long n = row_len(b);
    return n > 0 ? n + taps - 1 : 0;
  }
  __device__ __forceinline__ int filter_of(int b) const { return b % rows_per_meas; }
  __device__ __forceinline__ Row open(int b) const {
    const long long n = row_len(b);
    return Row{make_rsrc(base + off[b], (unsigned)(n > 0 ? n : 0) * 4u), (int)n, (int)(fade_out <= n ? fade_out : 0), win};
  }
};

// ---- alignment between crop_heads and crop_tails (core/pipeline.py:593-597 -> core/hrir.py:921-1001) -----------------------
// align_ipsilateral_all: per ipsilateral speaker pair (s1, s2) the lag of the full cross-correlation of the first
// `segment` samples of s1's left ear and s2's right ear (K10); lag > 0 delays s2 (both ears; s1 == s2: the right ear only),
// lag < 0 delays s1 (the left ear only when s1 == s2).  The pairs of IPSILATERAL_PAIRS share no speaker, so all searches of
// a measurement read the rows as crop_heads left them.  align_onset_groups_peak_leftref: every group's rows are shifted by
// -(peak of its leader's left ear - peak of FL's left ear), peaks taken AFTER the first alignment.
// ImpulseResponse.shift (core/impulse_response.py:92-108) keeps the length: a delay prepends zeros and drops the tail, an
// advance drops the head and pads zeros.
//
// tables of the lag searches: job j = (measurement m, ipsilateral pair p); one thread per job
__global__ __launch_bounds__(64) void slice_align_jobs_kernel(const int64_t* __restrict__ off2, const int64_t* __restrict__ len2,
                                                              const int* __restrict__ ipsi_a, const int* __restrict__ ipsi_b, int n_ipsi,
                                                              int rows_per_meas, int n_jobs, long long segment,
                                                              int64_t* __restrict__ a_off, int64_t* __restrict__ a_len,
                                                              int64_t* __restrict__ b_off, int64_t* __restrict__ b_len,
                                                              int* __restrict__ meas_flags) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  const int m = j / n_ipsi, p = j - m * n_ipsi;
  const int ra = m * rows_per_meas + 2 * ipsi_a[p], rb = m * rows_per_meas + 2 * ipsi_b[p] + 1;
  const long long na = len2[ra] < segment ? len2[ra] : segment, nb = len2[rb] < segment ? len2[rb] : segment;
  // scipy raises on an empty segment, and lags[argmax] is indexed with len(a) only: unequal segments go to the host flow
  if (na < 1 || nb < 1 || na != nb) atomicOr(&meas_flags[m], SLICE_ALIGN_GUARD);
  // (an empty row has no sample to read: its job looks at the block's first sample instead; the measurement is flagged)
  a_off[j] = na < 1 ? 0 : off2[ra];
  a_len[j] = na < 1 ? 1 : na;
  b_off[j] = nb < 1 ? 0 : off2[rb];
  b_len[j] = nb < 1 ? 1 : nb;
}

// the delays of align_ipsilateral_all per row, and the rows' lengths without the samples a delay pushes out (the peak
// search of the onset alignment runs on those: peak_index(zeros(d) ++ x[:n - d]) = d + peak_index(x[:n - d]) as long as
// x[0] cannot become a peak by gaining a left neighbour - checked in slice_align_onset_kernel); one thread per measurement
__global__ __launch_bounds__(64) void slice_align_delays_kernel(const long long* __restrict__ arg, const int64_t* __restrict__ a_len,
                                                                const int* __restrict__ ipsi_a, const int* __restrict__ ipsi_b, int n_ipsi,
                                                                int rows_per_meas, int n_meas, const int64_t* __restrict__ len2,
                                                                const int* __restrict__ leads, long long* __restrict__ d1,
                                                                int64_t* __restrict__ len1) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_meas) return;
  const int r0 = m * rows_per_meas;
  for (int r = 0; r < rows_per_meas; ++r) d1[r0 + r] = 0;
  for (int p = 0; p < n_ipsi; ++p) {
    const int j = m * n_ipsi + p;
    const long long lag = arg[j] - (a_len[j] - 1);             // lags = arange(-len(a) + 1, len(a)); lag = lags[argmax]
    const int qa = ipsi_a[p], qb = ipsi_b[p];
    if (qa == qb) {
      if (lag > 0) d1[r0 + 2 * qa + 1] = lag;
      else if (lag < 0) d1[r0 + 2 * qa] = -lag;
    } else if (lag > 0) {
      d1[r0 + 2 * qb] = lag;
      d1[r0 + 2 * qb + 1] = lag;
    } else if (lag < 0) {
      d1[r0 + 2 * qa] = -lag;
      d1[r0 + 2 * qa + 1] = -lag;
    }
  }
  // only the left rows of the onset leaders (and of the reference) are searched for a peak: the others get length 0
  for (int r = 0; r < rows_per_meas; ++r) {
    const long long n = len2[r0 + r] - d1[r0 + r];
    len1[r0 + r] = ((r & 1) == 0 && leads[r >> 1] && n > 0) ? n : 0;
  }
}

// the onset shifts: s2 of the rows of pair q = -(P(leader[q]) - P(ref)), P(q) = d1 + peak of q's left row after the first
// alignment; leader[q] < 0: no shift (the reference group, speakers outside the groups, groups whose leader is absent).
// One thread per measurement.
__global__ __launch_bounds__(64) void slice_align_onset_kernel(const RowPeak* __restrict__ res1, const long long* __restrict__ d1,
                                                               const int64_t* __restrict__ len1, const float* __restrict__ x,
                                                               const int64_t* __restrict__ off2, const int* __restrict__ leader, int ref_pair,
                                                               int rows_per_meas, int n_meas, long long* __restrict__ s2,
                                                               long long shifted_base, long long shifted_pitch, int64_t* __restrict__ off_al,
                                                               SliceRowOut* __restrict__ rows, int* __restrict__ meas_flags) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_meas) return;
  const int r0 = m * rows_per_meas;
  int flags = meas_flags[m];
  auto peak_after = [&](int q) {
    const int b = r0 + 2 * q;
    const RowPeak rp = res1[b];
    long long pk;
    if (len1[b] == 0 || !(__uint_as_float(rp.maxabs_bits) >= 1e-20f)) {          // (a leader's len1 is 0 only if the delay emptied it)
      pk = 0;
      if (d1[b] > 0) flags |= SLICE_ALIGN_GUARD;            // an all-zero row: its peak index does not move with the zeros
    } else {
      pk = (long long)(rp.first_peak != ~0ull ? rp.first_peak : rp.first_max);
    }
    if (d1[b] > 0 && len1[b] > 0 && x[off2[b]] != 0.f) flags |= SLICE_ALIGN_GUARD;   // x[0] gains a zero neighbour: it could become the first peak
    return d1[b] + pk;
  };
  const long long ref = peak_after(ref_pair);
  for (int q = 0; q < rows_per_meas / 2; ++q) {
    long long sh = 0;
    if (leader[q] >= 0) sh = -(peak_after(leader[q]) - ref);
    for (int e = 0; e < 2; ++e) {
      const int b = r0 + 2 * q + e;
      s2[b] = sh;
      // a row neither alignment moved stays where crop_heads left it; the others are materialised (shift_rows_kernel)
      off_al[b] = (d1[b] == 0 && sh == 0) ? off2[b] : shifted_base + (long long)b * shifted_pitch;
      rows[b].shift_ipsilateral = d1[b];
      rows[b].shift_onset = sh;
    }
  }
  meas_flags[m] = flags;
}

// ImpulseResponse.shift twice, materialised: y = shift(shift(x, d1), s2), each step keeping the length n (a delay drops the
// tail, an advance pads zeros).  grid (blocks, rows); dst row b at b * dst_pitch; d1 may be NULL (one shift: s2).
__global__ __launch_bounds__(256) void shift_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ src_off,
                                                         const int64_t* __restrict__ len, const long long* __restrict__ d1,
                                                         const long long* __restrict__ s2, float* __restrict__ dst,
                                                         const int64_t* __restrict__ dst_off) {
  const int b = blockIdx.y;
  const long long n = len[b];
  const long long a = d1 ? d1[b] : 0, s = s2[b];
  if (d1 && a == 0 && s == 0) return;           // (the slice: an unmoved row is read where it is)
  const float* __restrict__ in = src + src_off[b];
  float* __restrict__ out = dst + dst_off[b];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long k = i - s;                   // index into the once-shifted row (s > 0: delayed, s < 0: advanced)
    float v = 0.f;
    if (k >= 0 && k < n) {
      const long long j = k - a;                 // index into the source row
      if (j >= 0 && j < n) v = in[j];
    }
    out[i] = v;
  }
}

// ---- adjust decay (core/pipeline.py:694-716 -> core/parallel_workers.py:24-39 -> core/decay.py:355-403) ------------------
// on the equalized rows, for the rows that have a target RT60: decay_params (K3 + K7c, the launches of crop_tails once
// more on the new rows), decay_times (K7b), the window's parameters here, K8 in place.
// tables of the equalized rows: row b starts at b * out_pitch and is out_len[b / R] long; one thread per row
__global__ __launch_bounds__(64) void slice_decay_rows_kernel(const long long* __restrict__ out_len, int rows_per_meas, int n_rows,
                                                              long long out_pitch, int64_t* __restrict__ off3, int64_t* __restrict__ len3) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_rows) return;
  off3[b] = (long long)b * out_pitch;
  len3[b] = out_len[b / rows_per_meas];
}

// decay_times' jobs from the knee search's results; rows without a target get window 0 (their job returns at once)
__global__ __launch_bounds__(64) void slice_decay_jobs_kernel(const KneeRow* __restrict__ knee, const double* __restrict__ target, int rows_per_meas,
                                                              int n_rows, long long scratch_per_row, DecayJob* __restrict__ jobs) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_rows) return;
  const KneeRow k = knee[b];
  const double tg = target[b % rows_per_meas];
  DecayJob j;
  j.off = k.src_off;
  j.n = k.n;
  j.peak = k.peak;
  j.K = k.knee - k.peak;
  j.window = (tg == tg && k.done && !k.flags) ? k.window : 0;
  j.noise_floor = k.floor;
  j.scratch = (long long)b * scratch_per_row;
  jobs[b] = j;
}

// decay_adjustment_params (core/decay.py:355-380) for every row with a target, one thread per measurement: the slope from the
// longest defined decay time, no window when the response already decays faster than the target, else
// (start, half, knee, level) as WindowParams of an in-place K8 launch (rows left alone get length 0 in that launch).
__global__ __launch_bounds__(64) void slice_decay_params_kernel(const KneeRow* __restrict__ knee, const double* __restrict__ rt /*[rows][4]*/,
                                                                const double* __restrict__ target, const int64_t* __restrict__ len3,
                                                                int rows_per_meas, int n_meas, double fs,
                                                                int64_t* __restrict__ d_len, WindowParams* __restrict__ par,
                                                                SliceRowOut* __restrict__ rows, int* __restrict__ meas_flags) {
#pragma clang fp contract(off)          // wanted * knee_s - measured * knee_s rounds as Python's three operations do
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_meas) return;
  int flags = meas_flags[m];
  const double spans[4] = {-10.0, -20.0, -30.0, -60.0};
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int r = 0; r < rows_per_meas; ++r) {
    const int b = m * rows_per_meas + r;
    const double tg = target[r];
    WindowParams p;
    p.gain = 1.f;
    p.fade_in = p.fade_out = 0;
    p.decay_start = 0;
    p.decay_half = -1;
    p.decay_knee = 0;
    p.decay_level_db = 0.f;
    d_len[b] = 0;
    SliceRowOut& o = rows[b];
    o.decay_peak = o.decay_knee = 0;
    o.decay_slope = nan;
    o.decay_level_db = 0.0;
    o.decay_state = 0;
    o.decay_flags = 0;
    if (tg == tg) {
      const KneeRow k = knee[b];
      const int kf = k.done ? k.flags : (k.flags ? k.flags : (int)KNEE_GUARD);
      o.decay_peak = k.peak;
      o.decay_knee = k.knee;
      o.decay_flags = kf;
      double measured = nan;
      for (int q = 0; q < 4; ++q) {
        const double t = rt[4 * b + q];
        if (!(t == t) || t == 0.0) break;                  // `if not rt_time: break` (None or 0.0)
        measured = spans[q] / t;
      }
      o.decay_slope = measured;
      const double wanted = -60.0 / tg;
      const long long start = k.peak + 2 * (long long)floor(fs / 1000.0);
      const long long half = k.knee - start;
      if (kf || !(measured == measured)) {
        o.decay_state = 3;
        flags |= SLICE_DECAY_GUARD;
      } else if (wanted > measured) {
        o.decay_state = 2;                                 // not adjusting decay and noise floor up
      } else if (half < 0 || k.knee > len3[b]) {
        o.decay_state = 3;                                 // the window does not tile the response: the reference raises
        flags |= SLICE_DECAY_GUARD;
      } else {
        const double knee_s = (double)k.knee / fs;
        const double level = wanted * knee_s - measured * knee_s;
        p.decay_start = start;
        p.decay_half = half;
        p.decay_knee = k.knee;
        p.decay_level_db = (float)level;
        o.decay_level_db = (double)p.decay_level_db;
        o.decay_state = 1;
        d_len[b] = len3[b];
      }
    }
    par[b] = p;
  }
  meas_flags[m] = flags;
}

// The finished rows as float64, packed per measurement as a [rows_per_meas][out_len] host array lies (imp_slice_pack_f64).
// grid (blocks, rows); a measurement the device flagged has out_len as crop_tails left it (possibly 0): its rows are skipped
// by the caller.
__global__ void __launch_bounds__(256) slice_pack_f64_kernel(const float* __restrict__ rows, long long pitch,
                                                             const long long* __restrict__ outlen, int rows_per_meas,
                                                             double* __restrict__ packed, long long meas_stride) {
  const int b = blockIdx.y, m = b / rows_per_meas, r = b - m * rows_per_meas;
  const long long n = outlen[m];
  if (n <= 0 || (long long)rows_per_meas * n > meas_stride) return;
  const float* __restrict__ src = rows + (long long)b * pitch;
  double* __restrict__ dst = packed + (long long)m * meas_stride + (long long)r * n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (double)src[i];
}

}  // namespace imp
