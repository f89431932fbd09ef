/*
 * specengine.h — C ABI of the MI355X (gfx950) spectrum-analysis hot path.
 *
 * This is the drop-in boundary for the numeric core of the Gecesars/FFT analyzer
 *     t = arange(N)/Fs -> _generate_waveform -> _apply_modulation -> fftshift(fft(y)) -> abs -> argmax
 * (reference: main.py:452-479, 536-629, 959-965, 1029, 1318, 1622-1629; versão15.py:425-461, 590-602, 814).
 * The reference has no FFI of its own (pure numpy/scipy methods on a GUI class), so each entry point
 * cites the reference lines whose arithmetic it replaces.  The Python mirror of the reference
 * interface (calculate_fft / compute / dominant_peak) lives in fft_amd/specengine.py and binds
 * exactly these symbols through ctypes; INTEGRATION.md shows the 3-line patch to main.py.
 *
 * Conventions
 *   - plain C, caller owns every buffer, the library never returns memory it allocated
 *     (except through sa_dev_alloc, which the caller frees with sa_dev_free);
 *   - every function returns 0 (SA_OK) or a negative SA_E* code; the message is in sa_last_error()
 *     (thread local);
 *   - "host" entry points take host pointers, stage through device memory and synchronise before
 *     returning; "*_dev" entry points take device pointers, enqueue on `stream` (a hipStream_t cast
 *     to void*, NULL = the library's own per-thread non-blocking stream) and do NOT synchronise; to enqueue
 *     on the legacy default ("null") stream pass hipStreamLegacy, i.e. (void*)1, never a literal 0;
 *   - thread safe: may be called concurrently from several host threads (the reference calls the
 *     path from a ThreadPoolExecutor worker and from the Tk main thread, main.py:448, 745, 1080);
 *   - there is no CPU fallback: without a gfx950 device every compute entry point fails with SA_ENODEV.
 */
#ifndef SPECENGINE_H
#define SPECENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define SA_OK        0
#define SA_EINVAL   -1   /* bad argument (shim raises ValueError, as main.py:531 catches)          */
#define SA_ENODEV   -2   /* no usable HIP device                                                   */
#define SA_EHIP     -3   /* HIP runtime error (shim raises RuntimeError -> main.py:485 handler)    */
#define SA_ENOMEM   -4   /* device or host allocation failed                                       */
#define SA_EUNSUP   -5   /* size / option not supported by this build                              */

/* ---- window ids (a11: README.md:13 claim; formulas = numpy/scipy symmetric windows, which agree
 *      with the firmware's apply_window, Core/Src/main.c:83-99) -------------------------------- */
#define SA_WIN_NONE      0
#define SA_WIN_HANN      1   /* np.hanning(N)   = 0.5 - 0.5 cos(2 pi n/(N-1))                       */
#define SA_WIN_HAMMING   2   /* np.hamming(N)   = 0.54 - 0.46 cos(2 pi n/(N-1))                     */
#define SA_WIN_BLACKMAN  3   /* np.blackman(N)  = 0.42 - 0.5 cos(2 pi n/(N-1)) + 0.08 cos(4 pi n/(N-1)) */
#define SA_WIN_GAUSSIAN  4   /* scipy.signal.windows.gaussian(N, std=win_param, sym=True);
                                win_param <= 0 selects std = (N-1)/5                                */
#define SA_WIN_NUTTALL   5   /* the firmware's third window (Core/Src/main.c:92): 0.355768 - 0.487396 cos(2 pi a)
                                + 0.144232 cos(4 pi a) - 0.012604 cos(6 pi a), a = n/(N-1) (symmetric)   */
/* The firmware's apply_window as a whole (Core/Src/main.c:83-99): its Hann / Blackman / Nuttall blended towards 1 by the
 * taper, w[n] = 1 - (1 - mult[n]) t[n], t = a/e for a < e, (1-a)/e for a > 1-e, else 1, with a = n/(N-1), e = taper/100.
 * win_param = taper percent (the firmware's g_taper_percent, default 50 when win_param <= 0; <= 0.1 % gives all ones as
 * in the firmware).  Formulas evaluated in fp64 here (the firmware uses float); the reference applies them to a DAC
 * look-up table, never to an analysed frame — offered for completeness of a11. */
#define SA_WIN_FW_HANN      6
#define SA_WIN_FW_BLACKMAN  7
#define SA_WIN_FW_NUTTALL   8
#define SA_WIN_LAST      SA_WIN_FW_NUTTALL

/* ---- flags -------------------------------------------------------------------------------- */
#define SA_F_WRITE_MAG       1   /* write the N/2+1 magnitudes per frame (else peak only)          */
#define SA_F_DB              2   /* magnitudes written as 20*log10(max(|Y|, SA_DB_FLOOR)) dB        */
#define SA_F_EXCLUDE_DC_NYQ  4   /* peak search over bins 1..N/2-1 only (mirrors the f>0 mask of
                                    main.py:1622-1627, which drops DC and, for even N, Nyquist)     */
#define SA_DB_FLOOR_F32 1e-30f
#define SA_DB_FLOOR_F64 1e-300

/* ---- runtime ------------------------------------------------------------------------------ */
/* Initialise the library on the calling process' current HIP device (or `device` >= 0).
 * Idempotent; every other entry point calls it lazily.  Replaces nothing in the reference
 * (which is host-only) — it is the device boundary inserted at main.py:460-479. */
int  sa_init(int device /* -1 = keep current */);
void sa_shutdown(void);
int  sa_device_count(void);
int  sa_set_device(int device);
const char* sa_last_error(void);
const char* sa_version(void);

/* Device memory + copies for callers that keep data resident (bench, multi-GPU shards). */
int  sa_dev_alloc(void** dptr, uint64_t bytes);
int  sa_dev_free(void* dptr);
int  sa_memcpy_h2d(void* dst_dev, const void* src_host, uint64_t bytes);
int  sa_memcpy_d2h(void* dst_host, const void* src_dev, uint64_t bytes);
int  sa_sync(void);                       /* synchronise the calling thread's library stream */
/* Number of per-thread state records (stream + staging cache) the library holds.  A record is handed to the next
 * new thread when its thread exits, so this is the largest number of threads that ever used the library at the same
 * time, not the number of threads that ever did.  Diagnostic; no reference counterpart. */
int  sa_thread_state_count(void);

/* Single-process multi-GPU fan-out of the HOST-pointer batch entry points (SURVEY §8(b) "library stages/shards", §8(e)).
 * The reference runs this path from one process (a ThreadPoolExecutor worker, main.py:448), so a drop-in caller cannot
 * start a process per GPU: sa_shard_devices names `count` shard devices (NULL: shard i on device i % sa_device_count();
 * a device may repeat = several shards on one GPU; count 0 or 1 switches the fan-out off) and starts one host thread
 * per shard, bound to its device with its own stream and pinned staging.  From then on sa_rfft_peak_f32/_f64,
 * sa_stft_pcm16 (frame ranges, halo re-read from the host source), sa_synth_analyze and sa_analyze_frames_f32/_f64
 * hand shard s the contiguous block of frames [s B/G, (s+1) B/G) (first B mod G shards one more) and every shard
 * writes its slice of the caller's outputs: results are bit-identical to the unsharded call, no device-side exchange.
 * Sharded calls from several threads run one at a time.  *_dev entry points are not affected. */
int sa_shard_devices(const int32_t* devices, int32_t count);
int sa_shard_count(void);

/* Process-per-GPU form of the frame-sharded paths (one process per device, e.g. under torchrun): the ONLY
 * exchange is the gather of the per-frame peak records (8 bytes per frame) to one rank, natively over RCCL — the
 * reference is one process (main.py:448), this has no counterpart there.  Rank 0 calls sa_comm_unique_id and passes the
 * SA_COMM_ID_BYTES bytes to the other ranks (fft_amd/dist.py: through a file); every rank calls sa_comm_create with its
 * device current.  sa_comm_gather_peaks_dev: rank r contributes counts[r] records (counts: host array [world], the same
 * on every rank — fft_amd.dist.shard_range gives it without an exchange); `root` receives them in rank order into
 * all_bins / all_mags (device memory, sum(counts) entries; ignored elsewhere).  Enqueued on `stream`, not synchronised.
 * SA_EUNSUP when librccl.so cannot be loaded. */
#define SA_COMM_ID_BYTES 128
typedef struct sa_comm sa_comm;
int sa_comm_unique_id(void* id /* SA_COMM_ID_BYTES bytes */);
int sa_comm_create(sa_comm** comm, const void* id, int32_t rank, int32_t world);
int sa_comm_destroy(sa_comm* comm);
int sa_comm_gather_peaks_dev(sa_comm* comm, const int32_t* peak_bin, const float* peak_mag, int64_t count, const int64_t* counts,
                             int32_t root, int32_t* all_bins, float* all_mags, void* stream);

/* ---- a6/a8/a9/a11: window -> real FFT -> |Y| (-> dB) -> dominant peak, batched ----------------
 * Replaces, per frame:  Y = fft(y*w); np.abs(Y); np.argmax(...)   (main.py:475, 479, 1029, 1627)
 * computed as an rfft: bins 0..n/2 (the reference's length-n fftshift layout is rebuilt by the
 * shim from Hermitian symmetry).  Unnormalised forward DFT, e^{-2 pi i k n/N}, like scipy.fft.fft.
 *   x         [batch][n]        real frames
 *   mag       [batch][n/2+1]    |Y| (or dB), only if SA_F_WRITE_MAG; may be NULL otherwise
 *   peak_bin  [batch]           argmax_k |Y[k]|, lowest k on ties (np.argmax semantics)
 *   peak_mag  [batch]           |Y[peak_bin]| (always linear); with SA_F_WRITE_MAG and a linear spectrum it is the very
 *                               value written to mag[peak_bin]
 *   One documented deviation: WITHOUT SA_F_WRITE_MAG, 4096-point fp32 frames are ordered by |Y|^2 and the root is taken
 *   of the winner only.  Two bins whose |Y| round to the same float while their squares differ in the last place are a
 *   tie for np.argmax (lowest k) and not for this mode (larger square); exact ties go to the lowest k in both, and
 *   tone-like frames never differ (tests/test_gpu_rfft.py: exact-tie known answers in both modes).
 * n: power of two >= 16.  n <= 32768 (f32) / 16384 (f64) run in ONE LDS-resident pass, and so does the next size
 *    (65536 f32 / 32768 f64: one kernel, two LDS-resident halves) when frames start on even sample boundaries.
 *    Longer frames, up to 2^29
 *    (the late forks transform up to 1 000 000 points, sab10.py:43, 644), take the multi-pass path: a real-input
 *    four-step in the frame's own arithmetic (window fused into the first pass, |Y| / dB / peak into the last); it
 *    borrows one work buffer from the library, and the *_dev forms then synchronise `stream` before returning.
 *    Other n are served by sa_fft_mag_f64 (any n).                                               */
int sa_rfft_peak_f32(const float* x, int64_t batch, int32_t n, int32_t window, double win_param,
                     int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag);
int sa_rfft_peak_f64(const double* x, int64_t batch, int32_t n, int32_t window, double win_param,
                     int32_t flags, double* mag, int32_t* peak_bin, double* peak_mag);
int sa_rfft_peak_f32_dev(const float* x, int64_t batch, int32_t n, int32_t window, double win_param,
                         int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, void* stream);
int sa_rfft_peak_f64_dev(const double* x, int64_t batch, int32_t n, int32_t window, double win_param,
                         int32_t flags, double* mag, int32_t* peak_bin, double* peak_mag, void* stream);

/* Run `iters` back-to-back launches of sa_rfft_peak_f32_dev on the library stream and return the
 * average kernel time in milliseconds measured with HIP events on that same stream (bench.py's
 * roofline leg; no counterpart in the reference, which has one time.time() log line, sab10.py:614). */
int sa_time_rfft_peak_f32_dev(const float* x, int64_t batch, int32_t n, int32_t window, double win_param,
                              int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag,
                              int32_t iters, float* ms_avg);

/* Synthetic benchmark frames generated on the device:
 * frame b: A_b sin(2 pi f_b n/fs + phi_b) + 0.01 N(0,1), counter-based RNG keyed by (seed, b, n).
 * Same formula is restated on the host in oracle/ for parity tests. */
int sa_fill_tone_noise_f32_dev(float* x, int64_t batch, int32_t n, double fs, uint64_t seed, void* stream);

/* HIP-event timer on a stream (bench.py's roofline legs for the entry points that have no sa_time_* form; the reference's
 * only timing is one time.time() log line, sab10.py:614, 659).  sa_timer_start records an event on `stream` (NULL = the
 * calling thread's library stream — the one the *_dev entry points use when THEIR stream argument is NULL) and returns a
 * handle; sa_timer_stop records the closing event on the same stream, waits for it, returns the elapsed milliseconds
 * between the two events and releases the handle. */
int sa_timer_start(void* stream, void** timer);
int sa_timer_stop(void* timer, float* ms);

/* ---- a6 for lengths outside the batched rfft kernels, and BASELINE config 5 --------------------
 * Complex-to-complex fp64 FFT, interleaved (re, im), unnormalised forward e^{-2 pi i k n/N}
 * (scipy.fft.fft, main.py:475); flags bit0 = inverse (conjugate transform scaled by 1/n, scipy.fft.ifft),
 * bit1 = leave the output in digit-reversed order (skips the final scatter; n > 4096 only).
 * n: power of two, 2 <= n <= 2^30.  1 pass in LDS for n <= 4096, 2 passes to 2^20, 3 passes to 2^30.
 * `work` (device variant): scratch of n*batch complex values, may be NULL when n <= 4096; in/out may alias
 * only when n <= 4096. */
#define SA_FFT_INVERSE        1
#define SA_FFT_DIGITREV_OUT   2
int sa_fft_c2c_f64(const double* in_ri, double* out_ri, int64_t n, int64_t batch, int32_t flags);
int sa_fft_c2c_f64_dev(const double* in_ri, double* out_ri, double* work, int64_t n, int64_t batch,
                       int32_t flags, void* stream);
/* Analytic inputs / closed-form checks for ONE device-resident vector in natural order (SURVEY §8(d) C5 known answers at
 * sizes no host buffer should hold): kinds SA_DIST_KAT_* below (IMPULSE at index `param`, TONE at bin `param`, NOISE keyed by
 * seed).  sa_c2c_kat_check_dev synchronises `stream` and returns max |X[k] - expected| over all n bins. */
int sa_c2c_kat_fill_dev(double* x_ri, int64_t n, int32_t kind, int64_t param, uint64_t seed, void* stream);
int sa_c2c_kat_check_dev(const double* X_ri, int64_t n, int32_t kind, int64_t param, double* max_abs_err, void* stream);

/* ---- BASELINE config 5 across GPUs: ONE fp64 complex FFT of n = 2^L points as a distributed four-step --------------
 * No counterpart in the reference (single process, CPU: SURVEY §0); the arithmetic is scipy.fft.fft's (main.py:475).
 * ONE host process drives P ranks (P a power of two <= SA_DIST_MAX_RANKS).  A rank is a device plus its own streams
 * and buffers; `devices` maps rank -> HIP device (NULL: rank i on device i % sa_device_count()).  Several ranks may
 * share a device ("virtual ranks": the whole path runs on a one-GPU machine), ranks on different devices exchange
 * through peer-mapped memory over xGMI.
 *
 *   n = R1 * S, x viewed as [R1][S] row-major; cols_per_rank = S / P, rows_per_rank = R1 / P (sa_dist_plan_info)
 *   default input   rank r holds columns [r S/P, (r+1) S/P) of that view:  in[i * S/P + c] = x[i * S + r S/P + c]
 *   default output  rank q holds rows k1 in [q R1/P, (q+1) R1/P):         out[j * S + k'] = X[(q R1/P + j) + R1 k']
 *   -> ONE all-to-all, n/P^2 points per ordered pair of ranks.
 *   SA_DIST_NATURAL_IN / _OUT: contiguous slabs x[r n/P ...] / X[q n/P ...] instead, one more all-to-all each.
 * Exchange: SA_DIST_XCHG(mode) — DMA = one hipMemcpyAsync per (source, owner, chunk); KERNEL = the same copies by a
 * copy kernel storing through the peer mapping; FUSED = the first pass' store epilogue writes straight into the owners'
 * receive buffers (no send buffer, no copy pass); RCCL = ncclAllToAll (one chunk) or grouped ncclSend / ncclRecv per chunk
 * over one communicator per rank (needs a device of its own per rank: SA_EUNSUP otherwise, or when librccl.so cannot be
 * loaded); AUTO = FUSED when every rank shares one device, else DMA.
 * DMA / KERNEL / RCCL deliver in SA_DIST_CHUNKS(j) row chunks (default 4) and an owner transforms a chunk while the rest is
 * still on the wire.  sa_dist_execute is synchronous (returns when every rank is done) and fills `stats` if given. */
#define SA_DIST_MAX_RANKS    16
#define SA_DIST_NATURAL_IN   1
#define SA_DIST_NATURAL_OUT  2
#define SA_DIST_XCHG_AUTO    0
#define SA_DIST_XCHG_DMA     1
#define SA_DIST_XCHG_KERNEL  2
#define SA_DIST_XCHG_FUSED   3
#define SA_DIST_XCHG_RCCL    4
#define SA_DIST_XCHG(mode)   ((mode) << 4)
#define SA_DIST_CHUNKS(j)    ((j) << 8)
typedef struct sa_dist_plan sa_dist_plan;
typedef struct sa_dist_info {
    int64_t n, R1, S, cols_per_rank, rows_per_rank, slab_elems /* complex values per rank = n / P */;
    int32_t n_ranks, n_devices, device, exchange_mode, n_chunks, reserved;
    double* in;                 /* this rank's input slab  (device memory, slab_elems interleaved re/im pairs)  */
    double* out;                /* this rank's output slab (device memory)                                      */
} sa_dist_info;
typedef struct sa_dist_stats {
    int64_t n, bytes_per_peer;  /* bytes each rank sends to each other rank in the main exchange: 16 n / P^2    */
    int32_t n_ranks, n_devices, exchange_mode, n_chunks, exchanges /* all-to-alls run: 1, 2 or 3 */;
    int32_t pairs_timed;        /* 1: link_gbps[p][q] is that pair's own copy time (DMA, KERNEL); 0: the pairs of a source are
                                   not timed apart (RCCL: one group per chunk; FUSED: the stores of pass 1) and every
                                   link_gbps[p][q], q != p, holds the source's egress divided evenly over its P-1 links  */
    double  total_ms;           /* host wall clock of the call                                                  */
    double  device_ms;          /* longest first-event -> last-event span of any rank                           */
    double  pass1_ms, phase2_ms;/* longest first pass / longest owner phase of any rank (HIP events)            */
    double  exchange_ms;        /* longest (first copy start -> last copy end) of any inter-rank pair; 0 if FUSED */
    double  link_gbps_min, link_gbps_mean, link_gbps_max;   /* over ordered pairs p != q                        */
    double  aggregate_gbps;     /* all inter-rank bytes / exchange_ms (FUSED: / pass1_ms)                       */
    double  link_gbps[SA_DIST_MAX_RANKS][SA_DIST_MAX_RANKS]; /* [source][owner]: bytes_per_peer / that pair's copy time
                                   (FUSED: / the source's pass-1 time, a lower bound); diagonal = local copies */
    double  rank_exchange_ms[SA_DIST_MAX_RANKS]; /* per source rank: first copy / first RCCL group start -> last one's end   */
    double  rank_egress_gbps[SA_DIST_MAX_RANKS]; /* bytes_per_peer (P-1) / rank_exchange_ms: what left that GPU over xGMI   */
    double  chunk_ms[SA_DIST_MAX_RANKS];         /* RCCL: longest time any rank spent in the group of chunk j (j < 16); else 0 */
} sa_dist_stats;
int sa_dist_plan_create(sa_dist_plan** plan, int64_t n, int32_t n_ranks, const int32_t* devices, int32_t flags);
int sa_dist_plan_info(const sa_dist_plan* plan, int32_t rank, sa_dist_info* info);
int sa_dist_execute(sa_dist_plan* plan, sa_dist_stats* stats /* may be NULL */);
int sa_dist_plan_destroy(sa_dist_plan* plan);
/* Analytic inputs written straight into the ranks' input slabs, and the matching closed-form check of the output slabs
 * (SURVEY §8(d) C5 known answers, for sizes no host buffer should hold): IMPULSE at index `param` -> X[k] =
 * e^{-2 pi i k param / n}; TONE e^{+2 pi i param j / n} -> X[param] = n, 0 elsewhere; NOISE = N(0,1) pairs keyed by
 * (seed, j) (no closed form: compare with another transform of the same data).  *max_abs_err = max |X[k] - expected|. */
#define SA_DIST_KAT_IMPULSE  0
#define SA_DIST_KAT_TONE     1
#define SA_DIST_KAT_NOISE    2
int sa_dist_fill(sa_dist_plan* plan, int32_t kind, int64_t param, uint64_t seed);
int sa_dist_check(sa_dist_plan* plan, int32_t kind, int64_t param, double* max_abs_err);
/* Host arrays in natural order on both sides (the library shards, exchanges three times and gathers): the form
 * SURVEY §8(b) lists.  n_ranks = 0: one rank per visible device.  PCIe-bound for large n; benchmarks use the plan. */
int sa_fft_c2c_f64_dist(const double* in_ri, double* out_ri, int64_t n, int32_t n_ranks, int32_t flags,
                        sa_dist_stats* stats /* may be NULL */);

/* ---- a6/a8/a9 for ARBITRARY frame length (the GUI drop-in path) ----------------------------------
 * |fft(y*w)| of real fp64 frames of any length n >= 1: bins 0..n/2, plus the dominant peak.
 * Replaces scipy.fft.fft + np.abs + np.argmax for the N the GUI produces: int(duration*Fs), e.g. 2000
 * (main.py:105-107, 475, 511), the 1500-sample FNIRSI import (main.py:1080) and any len(y) handed to
 * _calculate_fft (main.py:959-965).  Power-of-two n run on the rfft / c2c kernels, every other n
 * through Bluestein's chirp-z on the c2c engine (fp64, chirp phases exact via k^2 mod 2n).
 *   x [batch][n], mag [batch][n/2+1], peak_bin/peak_mag [batch]; same window ids and flags as above. */
int sa_fft_mag_f64(const double* x, int64_t batch, int64_t n, int32_t window, double win_param,
                   int32_t flags, double* mag, int32_t* peak_bin, double* peak_mag);

/* ---- a2/a3/a4/a5: waveform synthesis + AM/FM on the device, fp64 -----------------------------------
 * One enum value per branch of _generate_waveform (main.py:536-615), in source order; anything else
 * yields zeros like the reference's final else (main.py:614-615). */
enum sa_waveform {
    SA_WF_SENO = 0, SA_WF_COSSENO, SA_WF_QUADRADA, SA_WF_TRIANGULAR, SA_WF_DENTE_DE_SERRA, SA_WF_PULSO,
    SA_WF_RUIDO_BRANCO, SA_WF_EXP_DECAIMENTO, SA_WF_PASSO, SA_WF_RAMPA, SA_WF_PARABOLA, SA_WF_IMPULSO,
    SA_WF_TANGENTE, SA_WF_SINC, SA_WF_GAUSSIANA, SA_WF_CHIRP_LINEAR, SA_WF_CHIRP_QUADRATICO, SA_WF_ONDA_AM,
    SA_WF_ONDA_FM, SA_WF_BATIMENTO, SA_WF_LORENTZIANA, SA_WF_HIPERBOLICA, SA_WF_BESSEL, SA_WF_SINC_MODULADO,
    SA_WF_PULSO_GAUSSIANO, SA_WF_DENTE_DE_SERRA_MOD, SA_WF_ONDA_AM_DSB, SA_WF_ONDA_FM_ESTEREO,
    SA_WF_QUADRADA_MODULADA, SA_WF_TRIANGULAR_MODULADA, SA_WF_PULSO_EXPONENCIAL, SA_WF_ONDA_COMB,
    SA_WF_ONDA_HARMONICA, SA_WF_UNKNOWN = 255
};

/* The dict `p` of _validate_inputs (main.py:492-502, 511) plus the amplitude of main.py:464-468. */
typedef struct sa_gen_params {
    int32_t  waveform;     /* enum sa_waveform                                            */
    int32_t  am_on;        /* p['am_on']                                                  */
    int32_t  fm_on;        /* p['fm_on']                                                  */
    int32_t  apply_vpp;    /* 0 mirrors the reference's "invalid Vpp entry" path (:467)   */
    int64_t  N;            /* p['N'] (even, >= 16 when it comes from _validate_inputs)    */
    double   duration;     /* p['duration'] seconds                                       */
    double   Fc, Fs, Fm;   /* Hz                                                          */
    double   am_depth;     /* p['am_depth']                                               */
    double   fm_dev;       /* p['fm_dev'] Hz                                              */
    double   vpp_half;     /* vpp / 2                                                     */
    uint64_t seed;         /* white-noise stream (reference: unseeded np.random.normal)   */
} sa_gen_params;

/* y = _apply_modulation(p, _generate_waveform(p, t) * vpp/2, t), t = arange(N)/Fs  (main.py:460-472).
 * t_out may be NULL. Host buffers of N doubles. */
int sa_synth_f64(const sa_gen_params* p, double* t_out, double* y_out);

/* The whole of main.py:460-479 in one call: synthesis -> modulation -> |fft(y*w)| (bins 0..N/2) -> peak.
 * window SA_WIN_NONE reproduces the reference (which applies no window). y_out[N], mag[N/2+1]. */
int sa_compute_f64(const sa_gen_params* p, int32_t window, double win_param, int32_t flags, double* t_out,
                   double* y_out, double* mag, int32_t* peak_bin, double* peak_mag);

/* ---- BASELINE config 3: short-time |FFT| + per-frame peak of interleaved PCM16 audio --------------
 * Same per-frame arithmetic as above (window -> rfft -> |Y| -> peak; main.py:475-479, 1029) over
 * overlapping frames f = 0 .. (n_samples - nfft)/hop of every channel; samples are scaled int16/32768
 * as in the reference's own import path (versão15.py:637).  The reference has no STFT.
 *   interleaved [n_samples][channels] int16
 *   mag         [channels][frames][nfft/2+1] float (SA_F_WRITE_MAG), peak_bin / peak_mag [channels][frames]
 * nfft: power of two 16..8192 (stereo fast path: both channels in one complex transform) or ..2^24 (per-channel path;
 *       above 32768 points the overlapping frames go through the multi-pass real FFT).
 * `scratch` (device variant): n_samples floats, only needed when channels != 2 or nfft > 8192. */
int64_t sa_stft_num_frames(int64_t n_samples, int32_t nfft, int32_t hop);
int sa_stft_pcm16(const int16_t* interleaved, int64_t n_samples, int32_t channels, int32_t nfft, int32_t hop,
                  int32_t window, double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag);
int sa_stft_pcm16_dev(const int16_t* interleaved, int64_t n_samples, int32_t channels, int32_t nfft, int32_t hop,
                      int32_t window, double win_param, int32_t flags, float* mag, int32_t* peak_bin,
                      float* peak_mag, float* scratch, void* stream);
/* Synthetic stereo recording generated on the device, interleaved [n_samples][2] int16:
 *   L = 0.35 sin(2 pi 440 t) + 0.25 sin(2 pi 3000 t) + 0.25 sin(2 pi phi(t)), phi a linear sweep 100 Hz -> 20 kHz that
 *       restarts every 2^20 samples;   R = pink-ish noise: 0.08 x the sum of 12 octave-held uniform(-1, 1) streams
 *       (Voss-McCartney), counter-based and keyed by (seed, octave, n >> octave);   t = n / fs.
 * No reference counterpart (the reference has no RIFF audio); parity tests read the samples back. */
int sa_fill_pcm16_stereo_dev(int16_t* interleaved, int64_t n_samples, double fs, uint64_t seed, void* stream);

/* ---- BASELINE config 4: fused generator pipeline, fp32 spectra, no HBM input ------------------------
 * Per parameter record:  synthesis + AM/FM at n_base points (main.py:460-472, 536-629; the record's N is
 * ignored and taken as n_base)  ->  `upsample`x not-a-knot cubic spline evaluated on
 * np.linspace(t[0], t[-1], n_base*upsample) (scipy CubicSpline as used at main.py:1390-1392)
 * ->  window  ->  rfft  ->  |Y| (-> dB)  ->  peak (main.py:475-479, 1029).
 * upsample: 1 or 8; n_base*upsample: power of two 2048..16384; n_base >= 256.
 *   params [batch] (host or device resident for the _dev variant), mag [batch][n/2+1],
 *   y_out  [batch][n] upsampled time signal (may be NULL). */
int sa_synth_analyze(const sa_gen_params* params, int64_t batch, int32_t n_base, int32_t upsample,
                     int32_t window, double win_param, int32_t flags, float* mag, int32_t* peak_bin,
                     float* peak_mag, float* y_out);
int sa_synth_analyze_dev(const sa_gen_params* params_dev, int64_t batch, int32_t n_base, int32_t upsample,
                         int32_t window, double win_param, int32_t flags, float* mag, int32_t* peak_bin,
                         float* peak_mag, float* y_out, void* stream);

/* ---- f2: time-domain statistics per frame --------------------------------------------
 * The reductions behind _calculate_time_analysis (main.py:989-1024) and the time half of
 * update_analysis_panels (main.py:1552-1617): Vpp = max-min, RMS = sqrt(mean_sq), DC = mean, crest =
 * max_abs/RMS, zero-crossing rate = zc_count/(t[-1]-t[0]), frequency estimate from the first/last
 * crossing (np.mean(np.diff(t[zc])) telescopes), duty = above_mid/n, scipy.stats.skew = m3/m2^1.5 and
 * kurtosis = m4/m2^2 - 3 (biased central moments).  All accumulation in fp64. */
typedef struct sa_time_stats {
    double  min, max, mean, mean_sq, max_abs;
    double  m2, m3, m4;                 /* mean((y-mean)^k), k = 2, 3, 4                                  */
    int64_t zc_count;                   /* #{ i : sign(y[i+1]) != sign(y[i]) }  (np.diff(np.sign(y)))     */
    int64_t zc_first, zc_last;          /* first / last such i, -1 when there is none                    */
    int64_t above_mid;                  /* #{ y > (max+min)/2 }  (duty cycle, main.py:1592-1594)          */
    int64_t argmax, argmin;             /* first index of the maximum / minimum (np.argmax / np.argmin)  */
} sa_time_stats;
int sa_time_stats_f64(const double* y, int64_t batch, int64_t n, sa_time_stats* out /*[batch]*/);
int sa_time_stats_f32_dev(const float* y, int64_t batch, int64_t n, int64_t frame_stride, sa_time_stats* out, void* stream);
int sa_time_stats_f64_dev(const double* y, int64_t batch, int64_t n, int64_t frame_stride, sa_time_stats* out, void* stream);

/* ---- f1: scipy.signal.find_peaks(x, height=, distance=) per row -----------------------------------------
 * As called at main.py:1034 (height = 0.05 max, distance = 10) and main.py:1639 (height only):
 * local maxima with scipy's plateau rule (middle sample, edges never peaks), x[peak] >= height, then for
 * distance > 1 suppression of peaks closer than `distance` to a higher one, highest first (equal heights:
 * the later peak is kept).  height = height_rel * max(row) when height_rel > 0, else height_abs
 * (-inf for none).  idx [batch][max_peaks] ascending peak indices, count [batch]; the host variant
 * fails with SA_EINVAL if a row has more than max_peaks candidates (n/2 always suffices); the device
 * variant stores -(candidates) in count for such a row.  work: batch*max_peaks bytes of device scratch. */
int sa_find_peaks_f64(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel,
                      int32_t distance, int32_t max_peaks, int32_t* idx, int32_t* count);
int sa_find_peaks_f64_dev(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel,
                          int32_t distance, int32_t max_peaks, int32_t* idx, int32_t* count, uint8_t* work,
                          void* stream);

/* Allocation-free variant of sa_find_peaks_f64_dev.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory — code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same arguments, results and kernels as sa_find_peaks_f64_dev, but the `work` bytes come from a device array built
 * into the library (SA_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends) instead of the caller:
 *   - batch * max_peaks must be <= SA_NOALLOC_SCRATCH_BYTES, else SA_EUNSUP (use sa_find_peaks_f64_dev);
 *   - sa_init must have succeeded before (SA_EINVAL otherwise) and `stream` must be given ((void*)1 = the null
 *     stream): the call itself creates no stream and calls no hipMalloc / hipFree / allocating routine;
 *   - every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (different streams or
 *     threads) race — serialise them.  Nothing needs clearing between calls (the kernel initialises each state byte
 *     it reads). */
#define SA_NOALLOC_SCRATCH_BYTES (4 << 20)
int sa_find_peaks_f64_noalloc_dev(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel,
                                  int32_t distance, int32_t max_peaks, int32_t* idx, int32_t* count, void* stream);

/* ---- f1: derived spectral metrics of a half spectrum (frequency half of update_analysis_panels,
 *      main.py:1619-1735) ------------------------------------------------------------------------------
 * mag [batch][n/2+1] = |Y| of length-n frames as the rfft entry points write them.  Positive bins are
 * 1..K, K = (n-1)/2 (the f > 0 mask of main.py:1622); f_k = k * freq_step with freq_step = 1/(n d), the
 * value scipy.fft.fftfreq multiplies by, so the masks compare the same doubles as the reference:
 *   fundamental (:1627)              fund_bin, fund_amp
 *   harmonic mask (:1634)            bins harm_lo_bin .. K-1  (f > 0.9 f0 and f < f_K)
 *   peaks >= 0.05 fund_amp (:1639)   n_peaks, harmonic_power = sum amp^2 (THD :1648, level :1652),
 *                                    max_spur / top_peak_bin (SFDR :1680-1682, peak frequency :1731)
 *   noise (:1664-1665)               noise_power = sum mag^2 over positive bins outside the mask (SNR :1668)
 *   -3 dB band (:1691-1696)          bw_lo_bin, bw_hi_bin: first / last positive bin >= fund_amp/sqrt 2
 *   sidebands (:1717-1718)           sideband_amp = max mag with |f - f0| < side_hz
 *   total_power                      sum over the two-sided spectrum of |Y|^2 (np.sum(Y**2), main.py:1043)
 * Bin fields are -1 when undefined (no positive bin, no peak). */
typedef struct sa_spec_metrics {
    double  fund_amp, harmonic_power, max_spur, noise_power, total_power, sideband_amp;
    int32_t fund_bin, harm_lo_bin, n_peaks, top_peak_bin, bw_lo_bin, bw_hi_bin;
} sa_spec_metrics;
int sa_spectral_metrics_f64(const double* mag, int64_t batch, int64_t n, double freq_step, double side_hz,
                            sa_spec_metrics* out /*[batch]*/);
int sa_spectral_metrics_f32_dev(const float* mag, int64_t batch, int64_t n, double freq_step, double side_hz,
                                sa_spec_metrics* out, void* stream);
int sa_spectral_metrics_f64_dev(const double* mag, int64_t batch, int64_t n, double freq_step, double side_hz,
                                sa_spec_metrics* out, void* stream);

/* ---- the second call-site's panel (versão15.py:1038-1180): read-outs its formulas take over the FULL fftshift-ordered
 *      spectrum Y [batch][n] with the shared frequency axis f [n], and over the time samples y [batch][ny] -------------------
 *   argmax, max          np.argmax(Y) (first maximum, the negative-frequency twin; the first NaN if there is one), Y[argmax] (:1101)
 *   median               np.median(Y) (:1127; NaN when Y holds a NaN) — exact order statistics, mean of the two middle ones
 *   bw_first, bw_last    np.where(Y > max / np.sqrt(2))[0][[0, -1]] (:1145-1148), -1 when empty
 *   sideband_max, _count np.max(Y[(f > 0.5 f0) & (f < 1.5 f0) & (f != f0)]), f0 = f[argmax] (:1163-1166); count 0 = empty mask
 *   above_half_max, y_max  np.sum(y > np.max(y) * 0.5), np.max(y) (:1081; -1 / 0 when y is NULL)
 * y and f may be NULL (their read-outs are then skipped).  fp64; integers exact, doubles are elements of the input. */
typedef struct sa_v15_readouts {
    double  max, median, sideband_max, y_max;
    int64_t argmax, bw_first, bw_last, sideband_count, above_half_max, nan_count /* 1 when Y holds a NaN */;
} sa_v15_readouts;
int sa_v15_readouts_f64(const double* y, int64_t ny, const double* f, const double* Y, int64_t n, int64_t batch,
                        sa_v15_readouts* out /*[batch]*/);
int sa_v15_readouts_f64_dev(const double* y, int64_t ny, const double* f, const double* Y, int64_t n, int64_t batch,
                            sa_v15_readouts* out, void* stream);

/* ---- whole read-out of a batch of records: what one "Calcular" does after synthesis (calculate_fft,
 *      main.py:977-987 -> _calculate_time_analysis :989-1024 / update_analysis_panels :1542-1735), per frame --
 * x [batch][n] -> window -> rfft -> |Y| on the device, then sa_time_stats over the frame and
 * sa_spectral_metrics over its half spectrum (freq_step = fs/n); only the per-frame structs and the
 * dominant peak cross PCIe (the spectra stay in device scratch).  n: power of two, 16..32768 (f32) /
 * 16..16384 (f64).  ts, sm, peak_bin, peak_mag: [batch]; any of them may be NULL to skip that output.
 * The _dev form takes device pointers; mag_work is batch*(n/2+1) elements of device scratch.  With mag_work,
 * peak_bin and peak_mag all given it only queues work on `stream`; where one of them is NULL the library
 * borrows the buffer from its per-thread arena and synchronises the stream before returning. */
int sa_analyze_frames_f32(const float* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                          double side_hz, sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, float* peak_mag);
int sa_analyze_frames_f64(const double* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                          double side_hz, sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, double* peak_mag);
int sa_analyze_frames_f32_dev(const float* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                              double side_hz, float* mag_work, sa_time_stats* ts, sa_spec_metrics* sm,
                              int32_t* peak_bin, float* peak_mag, void* stream);
int sa_analyze_frames_f64_dev(const double* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                              double side_hz, double* mag_work, sa_time_stats* ts, sa_spec_metrics* sm,
                              int32_t* peak_bin, double* peak_mag, void* stream);

/* ---- a10: smoothing of the zoomed time plot (update_time_zoom, main.py:1385-1392; INTERP_THRESHOLD /
 *      INTERP_SAMPLES main.py:26-27) -------------------------------------------------------------------------
 * Per row: cs = scipy.interpolate.CubicSpline(t, y) (default not-a-knot ends; 2 knots: the line, 3 knots: the
 * parabola, as scipy), t_out = np.linspace(t[0], t[-1], n_out), y_out = cs(t_out).  fp64; knots may be
 * non-uniform but must be finite and strictly increasing (the host form checks and returns SA_EINVAL with
 * scipy's message, the _dev form trusts the caller).  t, y: [batch][m], 2 <= m <= 2048; outputs [batch][n_out]. */
int sa_cubic_upsample_f64(const double* t, const double* y, int64_t batch, int32_t m, int32_t n_out,
                          double* t_out, double* y_out);
int sa_cubic_upsample_f64_dev(const double* t, const double* y, int64_t batch, int32_t m, int32_t n_out,
                              double* t_out, double* y_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPECENGINE_H */
