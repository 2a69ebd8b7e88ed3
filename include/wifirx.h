/*
 * wifirx.h -- C ABI of libwifirx.so: the MI355X-native IEEE 802.11a/g OFDM PHY receive chain.
 *
 * This is the drop-in boundary for the RX half of the reference's `wifi_phy_hier` hier block
 * (gnu_radio/wifi_phy_hier.grc:100-260,480-569,698-768), which IRS_AP inlines block for block
 * (gnu_radio/IRS_AP.py:267-285,291-311).  The reference has no FFI of its own for this path (its
 * blocks are GNU Radio C++ objects reached through SWIG/pybind); a GNU Radio block that wants the
 * GPU path binds exactly these entry points (ctypes stub in INTEGRATION.md).  Every function
 * replaces the work() of the reference blocks named beside it.
 *
 * Plain C: pointers and sizes only, no torch / HIP types.  All functions return 0 on success or a
 * negative WIFIRX_E* code; nothing throws across the boundary; no global state: every call works
 * on a handle.  One handle = one stream of samples = one HIP stream; calls on one handle must be
 * serialised by the caller, different handles are independent (GNU Radio runs each block on its
 * own thread, so this matches the reference's threading).  Stream mode with WIFIRX_P_STREAM_BATCH runs its
 * device pipeline on a worker thread owned by the handle: wifirx_push returns while a batch is still in flight.
 * That stays internal: every other entry point of the handle (batch calls, decode, set_param, sync, memcpy) first
 * waits for the batch in flight, and wifirx_last_error only ever shows text written on the caller's thread.
 *
 * There is NO CPU fallback in this library: wifirx_create() fails with WIFIRX_ENODEV when no
 * gfx950 device is usable.  The CPU restatement lives in oracle/ and is test infrastructure only.
 */
#ifndef WIFIRX_H
#define WIFIRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WIFIRX_ABI_VERSION 4

/* error codes */
#define WIFIRX_OK        0
#define WIFIRX_EINVAL   -1   /* bad argument */
#define WIFIRX_ENODEV   -2   /* no usable HIP device / kernel image (gfx950) */
#define WIFIRX_ENOMEM   -3   /* device or host allocation failed */
#define WIFIRX_EHIP     -4   /* HIP runtime error (wifirx_last_error() has the text) */
#define WIFIRX_ERANGE   -5   /* buffer too small / index out of range */
#define WIFIRX_EDEAD    -6   /* the handle's stream is dead (its sample buffer was lost): nothing to retry, destroy the handle */

/* Equalizer enum of the reference: ieee802_11.Equalizer (gnu_radio/IRS_AP.py:139-141). */
#define WIFIRX_EQ_LS   0
#define WIFIRX_EQ_LMS  1
#define WIFIRX_EQ_COMB 2
#define WIFIRX_EQ_STA  3

/* Encoding enum of the reference: ieee802_11.Encoding (gnu_radio/IRS_user.py:130-132). */
#define WIFIRX_BPSK_1_2   0
#define WIFIRX_BPSK_3_4   1
#define WIFIRX_QPSK_1_2   2
#define WIFIRX_QPSK_3_4   3
#define WIFIRX_16QAM_1_2  4
#define WIFIRX_16QAM_3_4  5
#define WIFIRX_64QAM_2_3  6
#define WIFIRX_64QAM_3_4  7

/* constants of the upstream blocks the chain follows */
#define WIFIRX_SYNC_LENGTH   320     /* sync_long(sync_length), gnu_radio/wifi_phy_hier.grc:59-75 */
#define WIFIRX_MIN_GAP       480     /* sync_short: re-trigger only after this many copied samples */
#define WIFIRX_MAX_SAMPLES   (540 * 80)
#define WIFIRX_MAX_SYM       511     /* decode_mac: frames with more data symbols are dropped */
#define WIFIRX_MAX_PSDU      1528    /* decode_mac: MAX_PAYLOAD 1500 + 28 */

/* frame flags */
#define WIFIRX_F_DETECTED   0x01u   /* sync_short plateau found (trigger valid) */
#define WIFIRX_F_SYNC       0x02u   /* sync_long found an LTS peak pair (frame_start, cfo_fine valid) */
#define WIFIRX_F_SIGNAL     0x04u   /* SIGNAL field decoded: parity ok and known rate */
#define WIFIRX_F_COMPLETE   0x08u   /* all n_sym data symbols lie inside the available samples and the
                                       output capacity (max_sym) and were written */
#define WIFIRX_F_LLR        0x10u   /* LLRs written for this frame (n_bpsc <= llr_bits) */
#define WIFIRX_F_DECODED    0x20u   /* decode_mac ran (Viterbi + descramble) */
#define WIFIRX_F_CRC_OK     0x40u   /* FCS good: PSDU delivered */
#define WIFIRX_F_TRUNCATED  0x80u   /* demodulation stopped because the samples of this trigger (or the
                                       output capacity max_sym) ran out, not because sync/SIGNAL failed */

/* One record per slot (batch mode) or per detected frame (stream mode): the stream tags
 * `wifi_start` of sync_short / sync_long / frame_equalizer rolled into one. */
typedef struct wifirx_frame {
    uint32_t flags;
    int32_t  trigger;      /* index n (in the slot / the stream) of the sync_short trigger sample of
                              the 16-delayed input, -1 if none; first copied sample is x[trigger-16] */
    int32_t  frame_start;  /* sync_long d_frame_start: offset of LTS1 in the copied samples (0..320) */
    float    cfo_coarse;   /* sync_short:  arg(A[trigger])/16          rad/sample */
    float    cfo_fine;     /* sync_long:   arg(p1*conj(p2))/64         rad/sample (upstream sign) */
    float    snr_db;       /* LS equalizer SNR estimate from LTS1/LTS2 */
    uint16_t psdu_len;     /* SIGNAL LENGTH ("frame_bytes") */
    uint8_t  encoding;     /* WIFIRX_BPSK_1_2 .. WIFIRX_64QAM_3_4 */
    uint8_t  n_bpsc;       /* bits per sub-carrier of `encoding` */
    uint16_t n_sym;        /* data symbols the SIGNAL field announces */
    uint16_t n_sym_out;    /* data symbols actually demodulated into the output buffers */
} wifirx_frame;            /* 32 bytes */

typedef struct wifirx_config {
    uint32_t abi_version;  /* WIFIRX_ABI_VERSION */
    int32_t  device;       /* HIP device ordinal */
    double   bandwidth;    /* sample rate in Hz: wifi_phy_hier `bandwidth` (grc:83-92) */
    double   frequency;    /* carrier in Hz:    wifi_phy_hier `frequency` (grc:501-510) */
    float    sensitivity;  /* sync_short threshold: wifi_phy_hier `sensitivity` (grc:681-690), 0.56 */
    int32_t  min_plateau;  /* sync_short min_plateau, 2 (gnu_radio/IRS_AP.py:268) */
    int32_t  chan_est;     /* WIFIRX_EQ_*: wifi_phy_hier `chan_est` (grc:299-308, IRS_AP.py:139-141) */
    uint32_t max_sym;      /* output capacity per frame in data symbols (<= WIFIRX_MAX_SYM) */
    uint32_t llr_bits;     /* LLR capacity per sub-carrier (0 = no LLR output, else 1,2,4,6) */
    uint32_t want_carrier; /* 1: also write the 48 equalised points per symbol (`carrier` port) */
    uint32_t max_batch;    /* largest n_slots of a wifirx_demod_batch call (workspace size) */
    uint32_t max_slot_len; /* largest slot length in samples */
} wifirx_config;

/* parameter ids for wifirx_set_param: the setters the reference flowgraphs call
 * (gnu_radio/IRS_user.py:229,265,273; gnu_radio/IRS_AP.py:348,373,382) */
#define WIFIRX_P_BANDWIDTH   1
#define WIFIRX_P_FREQUENCY   2
#define WIFIRX_P_SENSITIVITY 3
#define WIFIRX_P_CHAN_EST    4
/* stream mode: wifirx_push only collects samples until this many are waiting, then detects / demodulates /
 * decodes them in one go (0 = on every push).  A GNU Radio scheduler hands work() a few thousand items at a
 * time; one GPU round trip per such call would not keep up with the sample rate.  A push with n = 0 flushes.
 * At most WIFIRX_STREAM_BATCH_MAX (two pinned host buffers of one batch each are kept).  Changing the value while
 * samples are staged first runs those samples as a (short) batch.
 * ERRORS: the stream never gains a gap or a doubled sample through a failed push.  wifirx_push_consumed() tells how
 * many leading samples of the last wifirx_push call the stream has taken over: all of them after WIFIRX_OK; after an
 * error 0 -- repeat the call as it was (an allocation may succeed now) --, except for a call that spanned several
 * batches (n > batch size), which may have taken a prefix: repeat it from there.  Without a batch size a failed pass
 * is undone as a whole.  With one, the failure of a batch on the worker thread is reported by the NEXT push / flush
 * (once, before it takes anything); the failed batch stays staged in the library and the call after that runs it
 * again before going on.  A pass whose frames are queued is committed: a device failure behind that point (the carry of the
 * samples a pending frame still needs) is never reported as a failed push.  If it struck before the sample buffer was
 * touched, the stream goes on unharmed; if the buffer can no longer be trusted, every later push returns WIFIRX_EDEAD
 * with wifirx_push_consumed() = 0 and a text that says the stream is dead -- not a condition to retry: destroy the handle
 * (a GNU Radio work() answers it with WORK_DONE, wifirx/block.py). */
#define WIFIRX_P_STREAM_BATCH 5
#define WIFIRX_STREAM_BATCH_MAX (1u << 27)
/* decode_mac has two kernels with identical results: 128 frames per wave (throughput; a lone wave needs ~4 ms)
 * and one frame per wave (latency; ~0.2 ms per frame).  Batches of up to this many frames take the second one
 * (default 16384; 0 = always the first). */
#define WIFIRX_P_DECODE_SMALL_MAX 6
/* 1: every LLR is multiplied by |H|^2 of its sub-carrier (the LS channel estimate of the two long training
 * symbols; the optional channel-state weight of SURVEY.md section 8 row a7).  Default 0: plain max-log LLRs. */
#define WIFIRX_P_LLR_CSI 7
/* stream mode: 0 = the hard decisions of the frames stay on the device (wifirx_poll* then delivers zeros for `idx`);
 * default 1.  A consumer of PDUs only (the wifi_phy_rx block) saves most of the device-to-host traffic of a push. */
#define WIFIRX_P_STREAM_IDX 8

typedef struct wifirx_handle wifirx_handle;

/* Output buffers of a batch call.  Every pointer is caller-owned DEVICE memory when
 * `on_device` != 0, HOST memory otherwise (the library then stages through its own device
 * buffers; that path is PCIe-bound and is not the measured one).  NULL = do not produce.
 *   frames  [n_slots]
 *   idx     [n_slots][max_sym][48]            hard decisions, one constellation index per byte
 *                                              (output 0 of frame_equalizer, grc:550-569)
 *   llr     [n_slots][max_sym*48*llr_bits]    per frame packed [sym][carrier][bit 0..n_bpsc-1]
 *   carrier [n_slots][max_sym][48][2]         equalised points (re,im): the `symbols` message port
 *   psdu    [n_slots][psdu_stride]            decode_mac output: MAC frame incl. FCS position
 *                                              (bytes 0..psdu_len-1), valid when WIFIRX_F_CRC_OK
 *   csi     [n_slots][52][2]                  channel state (re,im) on the 52 occupied sub-carriers in ascending
 *                                              order: the LS estimate from the two long training symbols (what
 *                                              the reference's disabled extract_csi block taps,
 *                                              gnu_radio/IRS_AP.grc:640-654); written once SYNC is set
 */
typedef struct wifirx_out {
    wifirx_frame* frames;
    uint8_t*      idx;
    float*        llr;
    float*        carrier;
    uint8_t*      psdu;
    uint32_t      psdu_stride;  /* bytes per slot in `psdu` (>= largest psdu_len expected) */
    uint32_t      on_device;
    float*        csi;          /* ABI 2 */
    float*        sym_stats;    /* ABI 3: [n_slots][4] = sum |y|, sum |y|^2, sum |y|^4 over the equalised points of the
                                   frame's n_sym_out data symbols (48 each), 0 -- the statistics a
                                   digital.probe_mpsk_snr_est_c fed from frame_equalizer's `symbols` port accumulates
                                   (gnu_radio/IRS_AP.py:275,312); float32 sums in the order of DESIGN.md rule 13 */
    uint32_t*     hbits;        /* ABI 4: [n_slots][max_sym * 12] -- the hard decisions of `idx` as bit planes, the form
                                   wifirx_decode_batch reads (decode_mac's input, gnu_radio/IRS_AP.py:272).  Data symbol
                                   q of a frame with n_bpsc bits per carrier occupies the 2 * n_bpsc words from
                                   q * 2 * n_bpsc on (symbols are packed; 48 bytes per symbol are reserved, as for
                                   `idx`); word 2 b + h holds bit b of the decisions of the FFT bins 32 h .. 32 h + 31
                                   (shifted order: bin 32 = DC; data carrier c of `idx` is bin c + 6 + the pilots / DC
                                   below it), bin 32 h + k in bit k, 0 for bins that carry no data.  Optional: when
                                   given, wifirx_demod_batch fills it and wifirx_decode_batch reads it instead of
                                   `idx`; words behind a frame's last symbol are not written. */
} wifirx_out;

/* Output buffers of wifirx_poll_ex (host memory; NULL = not wanted).  Row i belongs to frame i of the call. */
typedef struct wifirx_poll_out {
    wifirx_frame* frames;       /* [cap], required */
    uint8_t*      psdu;         /* [cap][psdu_stride] */
    uint32_t      psdu_stride;
    uint32_t      reserved;
    uint8_t*      idx;          /* [cap][max_sym][48] */
    float*        carrier;      /* [cap][max_sym][48][2], handles created with want_carrier */
    float*        csi;          /* [cap][52][2] */
    float*        sym_stats;    /* [cap][4], see wifirx_out */
} wifirx_poll_out;

typedef struct wifirx_stats {
    uint64_t samples_in;     /* samples consumed */
    uint64_t frames_detected;
    uint64_t frames_signal_ok;
    uint64_t frames_complete;
    uint64_t frames_crc_ok;
    uint64_t frames_dropped; /* detected but not delivered (sync / SIGNAL / truncation / CRC) */
} wifirx_stats;

/* lifetime ------------------------------------------------------------------------------------ */
int  wifirx_create(const wifirx_config* cfg, wifirx_handle** out);
int  wifirx_destroy(wifirx_handle* h);
const char* wifirx_last_error(const wifirx_handle* h);      /* never NULL */
int  wifirx_abi_version(void);

/* replaces the generated set_bandwidth/set_frequency/set_chan_est/set_sensitivity setters */
int  wifirx_set_param(wifirx_handle* h, int id, double value);
int  wifirx_get_stats(const wifirx_handle* h, wifirx_stats* st);

/* Batch mode: n_slots independent slots of slot_len samples each, laid out back to back in `iq`
 * (interleaved float re,im = complex64 / GNU Radio gr_complex).  Every slot is treated as its own
 * sample stream that starts in sync_short's SEARCH state; the first frame of every slot is
 * demodulated:  autocorrelation graph + sync_short + sync_long + fft_vcc + frame_equalizer
 * (gnu_radio/IRS_AP.py:268-269,271,273,276-285).  Asynchronous on the handle's stream when all
 * buffers are on the device; call wifirx_sync() before reading results.
 * Slots of unequal length: wifirx_demod_batch_v. */
int  wifirx_demod_batch(wifirx_handle* h, const float* iq, int iq_on_device,
                        uint32_t slot_len, uint32_t n_slots, const wifirx_out* out);

/* The same over slots of unequal length (the slot_off[] form of SURVEY.md 8(b)): slot k is the samples
 * [slot_off[k], slot_off[k+1]) of `iq`; slot_off is a HOST array of n_slots + 1 non-decreasing sample offsets (the
 * library copies it to the device before the launch).  Output row k belongs to slot k; a slot may be empty. */
int  wifirx_demod_batch_v(wifirx_handle* h, const float* iq, int iq_on_device, const uint64_t* slot_off,
                          uint32_t n_slots, const wifirx_out* out);

/* decode_mac over the hard decisions of a previous wifirx_demod_batch on the same buffers
 * (ieee802_11.decode_mac, gnu_radio/IRS_AP.py:272,291-292): de-interleave, de-puncture, Viterbi
 * K=7 (133,171), descramble, CRC-32.  Sets WIFIRX_F_DECODED / WIFIRX_F_CRC_OK in out->frames
 * and writes out->psdu.  Input: out->hbits when given (filled by the demod call: no pass over `idx` at all), else
 * out->idx (packed into bit planes by a pre-pass).  Device buffers only (out->on_device = 1).  The call waits once for the
 * stream (a pre-pass reads back the longest trellis of the batch to size the survivor scratch);
 * the decode kernel itself is then queued asynchronously. */
int  wifirx_decode_batch(wifirx_handle* h, uint32_t n_slots, const wifirx_out* out);

/* Stream mode (what the GNU Radio block's work() calls): append `n` samples of the continuous
 * input stream (host or device memory; the library copies).  Frames that completed inside the
 * samples seen so far are demodulated + decoded and queued for wifirx_poll. */
int  wifirx_push(wifirx_handle* h, const float* iq, size_t n, int iq_on_device);

/* How many leading samples of the last wifirx_push call the stream has taken over: n after WIFIRX_OK; after an error
 * the count from which the caller repeats the call (0 unless the call spanned several batches).  See
 * WIFIRX_P_STREAM_BATCH, ERRORS.  GNU Radio equivalent: what work() would pass to consume_each() on an error path. */
size_t wifirx_push_consumed(const wifirx_handle* h);

/* Fetch up to `cap` queued frames (host memory).  For frame i: record frames[i], its PSDU at
 * psdu + i*psdu_stride, and -- when requested at create time and the pointers are non-NULL --
 * its hard decisions at idx + i*max_sym*48 and equalised points at carrier + i*max_sym*96 floats.
 * *n_out receives the number of frames written. */
int  wifirx_poll(wifirx_handle* h, wifirx_frame* frames, uint8_t* psdu, uint32_t psdu_stride,
                 uint8_t* idx, float* carrier, uint32_t cap, uint32_t* n_out);

/* Number of finished frames waiting for wifirx_poll*: one atomic load, no lock -- cheap enough for every work() call
 * (the block polls only when this is non-zero). */
uint32_t wifirx_queued(const wifirx_handle* h);

/* wifirx_poll that also delivers the channel state of every frame: csi + i*104 floats = the LS estimate on the 52
 * occupied sub-carriers (re, im), the `csi` entry upstream's frame_equalizer puts into the frame's tag dictionary
 * (read by ieee802_11.extract_csi, gnu_radio/IRS_AP.grc:640-654).  csi may be NULL. */
int  wifirx_poll_csi(wifirx_handle* h, wifirx_frame* frames, uint8_t* psdu, uint32_t psdu_stride,
                     uint8_t* idx, float* carrier, float* csi, uint32_t cap, uint32_t* n_out);

/* wifirx_poll with every per-frame output behind one struct (what later ABI versions extend). */
int  wifirx_poll_ex(wifirx_handle* h, const wifirx_poll_out* out, uint32_t cap, uint32_t* n_out);

/* block until everything queued on the handle's stream has finished */
int  wifirx_sync(wifirx_handle* h);

/* The HIP stream of the handle as an opaque pointer (hipStream_t), so that a caller can record
 * events / order its own work against it. */
void* wifirx_stream(wifirx_handle* h);

/* Synthetic channel on the device (test bench of gnu_radio/IRS_tranceiver.py:282-294): builds
 * n_slots slots from n_templates clean frames (host or device memory, frame_len samples each, frame
 * i uses template i % n_templates): slot = noise(unit variance, Philox counter RNG, `seed`) with
 * the frame scaled by sqrt(10^(snr_db/10)) and rotated by a per-slot CFO drawn uniformly from
 * +-cfo_max rad/sample, placed `lead` samples into the slot.  Writes `cfo_out[n_slots]` when
 * non-NULL.  Output `slots` is device memory of n_slots*slot_len complex64.  snr_db = NaN selects the
 * noiseless channel (gain 1, no AWGN; the slot outside the frame is zero). */
int  wifirx_synth_slots(wifirx_handle* h, const float* templates, int templates_on_device,
                        uint32_t n_templates, uint32_t frame_len, float* slots, uint32_t slot_len,
                        uint32_t n_slots, uint32_t lead, float snr_db, float cfo_max,
                        uint64_t seed, float* cfo_out);

/* plain device memory helpers so that a host language without a HIP binding can own buffers */
int  wifirx_dev_alloc(wifirx_handle* h, size_t bytes, void** out);
int  wifirx_dev_free(wifirx_handle* h, void* p);
int  wifirx_memcpy_h2d(wifirx_handle* h, void* dst, const void* src, size_t bytes);
int  wifirx_memcpy_d2h(wifirx_handle* h, void* dst, const void* src, size_t bytes);

/* Time the dominant kernel of wifirx_demod_batch with HIP events on the handle's stream: runs the
 * batch `iters` times and returns the mean kernel time in milliseconds (used by bench.py for
 * roofline.achieved). */
int  wifirx_time_demod(wifirx_handle* h, const float* iq_dev, uint32_t slot_len, uint32_t n_slots,
                       const wifirx_out* out, int iters, float* ms_mean);

#ifdef __cplusplus
}
#endif
#endif /* WIFIRX_H */
