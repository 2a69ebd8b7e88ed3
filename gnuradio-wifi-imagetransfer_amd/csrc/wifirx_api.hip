// wifirx_api.hip -- implementation of the C ABI in include/wifirx.h on top of the HIP kernels.
// Host side only: handles, the per-handle HIP stream, workspaces, host<->device staging.
// No CPU compute fallback exists: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <emmintrin.h>      // SSE2 streaming stores of the staging copy (x86-64 baseline)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "wifirx.h"
#include "wr_kernels.h"

// A finished frame of the stream: its record and where its outputs sit in the (shared) host copy of its batch.
struct PolledFrame {
    wifirx_frame                           fr;
    std::shared_ptr<std::vector<uint8_t>>  blob;
    size_t   o_psdu = 0, o_idx = 0, o_car = 0, o_csi = 0, o_stats = 0;     // byte offsets into *blob
    uint32_t n_psdu = 0, n_idx = 0, n_car = 0;                              // bytes, bytes, floats
};

struct PendingTrig {
    int64_t pos;        // absolute stream index of the trigger
    float   cfo;        // coarse CFO, valid once the frame kernel has seen the trigger (have_cfo)
    bool    have_cfo;
};

struct wifirx_handle {
    wifirx_config cfg;
    int           device = 0;
    hipStream_t   stream = nullptr;
    std::string   err;
    wifirx_stats  stats{};

    // batch staging (host-buffer path only)
    void*  stage_iq = nullptr;      size_t stage_iq_bytes = 0;
    void*  stage_frames = nullptr;  size_t stage_frames_bytes = 0;
    void*  stage_idx = nullptr;     size_t stage_idx_bytes = 0;
    void*  stage_llr = nullptr;     size_t stage_llr_bytes = 0;
    void*  stage_car = nullptr;     size_t stage_car_bytes = 0;
    void*  stage_psdu = nullptr;    size_t stage_psdu_bytes = 0;
    void*  stage_csi = nullptr;     size_t stage_csi_bytes = 0;
    void*  stage_stats = nullptr;   size_t stage_stats_bytes = 0;
    void*  stage_hbits = nullptr;   size_t stage_hbits_bytes = 0;
    void*  stage_off = nullptr;     size_t stage_off_bytes = 0;      // slot offsets of wifirx_demod_batch_v

    // decode workspace
    void*  dec_scratch = nullptr;   size_t dec_scratch_bytes = 0;
    void*  dec_max = nullptr;       size_t dec_max_bytes = 0;
    void*  dec_perm = nullptr;      size_t dec_perm_bytes = 0;    // decode_mac over several rates: frames grouped by rate
    void*  dec_hbits = nullptr;     size_t dec_hbits_bytes = 0;   // decode_mac over `idx` alone: its bit planes (wifirx_out.hbits form)
    void*  s_pack = nullptr;        size_t s_pack_bytes = 0;      // stream outputs, rows cut to their filled width
    void*  s_host = nullptr;        size_t s_host_bytes = 0;      // pinned landing zone of the packed outputs

    // stream mode
    float2*  sbuf = nullptr;        int64_t sbuf_cap = 0;     // device sample buffer
    int64_t  sbase = 0;             // absolute index of sbuf[0] (multiple of 64)
    int64_t  sfill = 0;             // valid samples in sbuf
    int64_t  sdetected = 0;         // absolute index up to which triggers have been selected
    int64_t  last_trig = -(1ll << 40);
    int64_t  stream_batch = 0;      // WIFIRX_P_STREAM_BATCH
    uint32_t decode_small_max = WR_DECODE_SMALL_MAX;   // WIFIRX_P_DECODE_SMALL_MAX
    uint32_t decode_fpw = 0;        // WIFIRX_DECODE_FPW (environment, tests): frames per wave of the throughput decoder
    int      decode_ovl = -1;       // WIFIRX_DECODE_OVL: 0 = trace-back behind each task, 1 = under the wave's next task, 2 = speculative walks under the task's own add-compare-select; -1: by batch size
    int      decode_q = -1;         // WIFIRX_DECODE_Q (environment, tests): 1 / 0 = always / never the four-frames-per-lane decoder; -1: by batch size
    int32_t  llr_csi = 0;           // WIFIRX_P_LLR_CSI
    int32_t  stream_want_idx = 1;   // WIFIRX_P_STREAM_IDX
    int64_t  sprocessed = 0;        // absolute index up to which pushes have been processed
    uint8_t* s_above = nullptr;     float2* s_A = nullptr;    int64_t s_above_cap = 0;
    std::vector<PendingTrig> pending;
    std::deque<PolledFrame>  queue;
    void*  s_trig = nullptr;  void* s_frames = nullptr;  void* s_idx = nullptr;
    void*  s_car = nullptr;   void* s_psdu = nullptr;    uint32_t s_cap = 0;
    void*  s_csi = nullptr;
    void*  s_stats = nullptr;
    void*  s_hbits = nullptr;
    int    test_fail_alloc = 0, test_alloc_count = 0;      // WIFIRX_TEST_FAIL_ALLOC (allocation-failure tests)
    size_t test_decode_budget = 0;                         // WIFIRX_TEST_DECODE_BUDGET: bytes of survivor scratch a decode call may hold (tests)
    int    test_fail_decode_scratch = 0;                   // WIFIRX_TEST_FAIL_DECODE_SCRATCH: the next k scratch allocations fail (tests)
    uint32_t dec_last_waves = 0; bool dec_last_overlap = false; int dec_last_mode = 0;      // what the last throughput decode ran with (WIFIRX_TRACE)
    int    test_fail_carry = 0, test_carry_count = 0;      // WIFIRX_TEST_FAIL_CARRY (a failure behind the commit of a stream pass)
    bool   stream_dead = false;                            // the carry step failed after it had begun to move the sample buffer
    std::string stream_dead_msg;

    // Host-buffer stream path (what a GNU Radio work() drives; wifirx_api_stream.inc): pushes are copied into one of two
    // pinned staging buffers of one batch each; a full one is handed to the worker thread, which runs the device
    // pipeline for it while the caller fills the other.  `mu` guards the job slot, the frame queue and the statistics.
    std::mutex              mu;
    std::condition_variable cv;
    std::thread             worker;
    bool    w_started = false, w_stop = false, w_busy = false, w_has_job = false;
    const float* w_job_ptr = nullptr;   size_t w_job_n = 0;   bool w_job_flush = false;
    int     w_rc = 0;                   std::string w_err;     // first failure of a batch, reported by the next call
    std::string w_err_local;            // the worker's own error text (only the worker writes it; h->err belongs to the caller's thread)
    bool    w_retry = false;            // the failed batch is still staged (its ring buffer untouched): the next push / flush re-submits it
    const float* w_retry_ptr = nullptr; size_t w_retry_n = 0;   bool w_retry_flush = false;
    float2* ring[2] = { nullptr, nullptr };   size_t ring_cap = 0, ring_fill = 0;   int ring_cur = 0;
    size_t  push_consumed = 0;          // wifirx_push_consumed
    std::atomic<uint32_t> n_queued{0};  // frames waiting for wifirx_poll (wifirx_queued: read without the lock)
};

namespace {

thread_local std::string g_err;
thread_local bool t_is_worker = false;      // set by the handle's stream worker thread

// h->err is written by the caller's thread only (one handle = calls serialised by the caller); the stream worker keeps
// its text in w_err_local until the caller's next push / flush takes it over (stream_worker_take_error).
int fail(wifirx_handle* h, int code, const std::string& msg)
{
    if (h) { if (t_is_worker) h->w_err_local = msg; else h->err = msg; }
    g_err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, WIFIRX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

// A failed allocation: HIP keeps the last non-success code until it is read (hipGetLastError), and the launch wrappers of
// wr_kernels.hip / wr_decode.hip end in `return hipGetLastError()` -- a later launch that went fine would report this allocation's
// hipErrorOutOfMemory.  Every ENOMEM path reads it away.
int oom(wifirx_handle* h, const char* what, hipError_t e)
{
    (void)hipGetLastError();
    return fail(h, WIFIRX_ENOMEM, std::string(what) + ": " + hipGetErrorString(e));
}

int ensure(wifirx_handle* h, void** p, size_t* have, size_t need)
{
    if (*have >= need) return WIFIRX_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) return oom(h, "hipMalloc", e);
    *have = need;
    return WIFIRX_OK;
}

int ensure_pinned(wifirx_handle* h, void** p, size_t* have, size_t need)
{
    if (*have >= need) return WIFIRX_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *have = 0; }
    need += need / 2;
    hipError_t e = hipHostMalloc(p, need, hipHostMallocDefault);
    if (e != hipSuccess) return oom(h, "hipHostMalloc", e);
    *have = need;
    return WIFIRX_OK;
}

wr::DemodParams params_of(const wifirx_handle* h)
{
    wr::DemodParams p;
    p.bandwidth = h->cfg.bandwidth;
    p.frequency = h->cfg.frequency;
    p.threshold = h->cfg.sensitivity;
    p.min_plateau = h->cfg.min_plateau;
    p.max_sym = h->cfg.max_sym;
    p.llr_bits = h->cfg.llr_bits;
    p.chan_est = h->cfg.chan_est;
    p.llr_csi = h->llr_csi;
    return p;
}

}  // namespace

static void stream_worker_stop(wifirx_handle* h);
static void stream_worker_wait_idle(wifirx_handle* h);

extern "C" {

int wifirx_abi_version(void) { return WIFIRX_ABI_VERSION; }

const char* wifirx_last_error(const wifirx_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int wifirx_create(const wifirx_config* cfg, wifirx_handle** out)
{
    if (!cfg || !out) return fail(nullptr, WIFIRX_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != WIFIRX_ABI_VERSION) return fail(nullptr, WIFIRX_EINVAL, "abi_version mismatch");
    if (cfg->max_sym == 0 || cfg->max_sym > WIFIRX_MAX_SYM) return fail(nullptr, WIFIRX_EINVAL, "max_sym out of range");
    if (!(cfg->llr_bits == 0 || cfg->llr_bits == 1 || cfg->llr_bits == 2 || cfg->llr_bits == 4 || cfg->llr_bits == 6))
        return fail(nullptr, WIFIRX_EINVAL, "llr_bits must be 0,1,2,4,6");
    if (cfg->chan_est < WIFIRX_EQ_LS || cfg->chan_est > WIFIRX_EQ_STA)
        return fail(nullptr, WIFIRX_EINVAL, "chan_est must be one of WIFIRX_EQ_LS, LMS, COMB, STA");
    if (!(cfg->bandwidth > 0) || !(cfg->frequency > 0)) return fail(nullptr, WIFIRX_EINVAL, "bandwidth/frequency must be > 0");
    if (cfg->min_plateau < 0 || cfg->min_plateau > 32) return fail(nullptr, WIFIRX_EINVAL, "min_plateau out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, WIFIRX_ENODEV, "no HIP device: libwifirx has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, WIFIRX_ENODEV, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, WIFIRX_ENODEV, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, WIFIRX_ENODEV, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    wifirx_handle* h = new (std::nothrow) wifirx_handle();
    if (!h) return fail(nullptr, WIFIRX_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    if (const char* e = std::getenv("WIFIRX_DECODE_SMALL_MAX")) h->decode_small_max = (uint32_t)std::strtoul(e, nullptr, 10);   // tests pick the decode kernel with this
    if (const char* e = std::getenv("WIFIRX_DECODE_FPW")) h->decode_fpw = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = std::getenv("WIFIRX_DECODE_Q")) h->decode_q = std::atoi(e) != 0;
    if (const char* e = std::getenv("WIFIRX_DECODE_OVL")) h->decode_ovl = std::max(0, std::min(2, std::atoi(e)));
    if (const char* e = std::getenv("WIFIRX_TEST_FAIL_ALLOC")) h->test_fail_alloc = std::atoi(e);
    if (const char* e = std::getenv("WIFIRX_TEST_FAIL_CARRY")) h->test_fail_carry = std::atoi(e);
    if (const char* e = std::getenv("WIFIRX_TEST_DECODE_BUDGET")) h->test_decode_budget = (size_t)std::strtoull(e, nullptr, 10);
    if (const char* e = std::getenv("WIFIRX_TEST_FAIL_DECODE_SCRATCH")) h->test_fail_decode_scratch = std::atoi(e);
    // WIFIRX_STREAM_PRIORITY = high | low (environment; measurement: two handles whose kernels share the GPU, tools/coresident_probe.py)
    int prio = 0, prio_lo = 0, prio_hi = 0;
    if (const char* e = std::getenv("WIFIRX_STREAM_PRIORITY")) {
        if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess) prio = e[0] == 'h' ? prio_hi : e[0] == 'l' ? prio_lo : 0;
        else (void)hipGetLastError();
    }
    if (hipSetDevice(h->device) != hipSuccess || hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio) != hipSuccess) {
        delete h;
        return fail(nullptr, WIFIRX_EHIP, "hipStreamCreate failed");
    }
    *out = h;
    return WIFIRX_OK;
}

int wifirx_destroy(wifirx_handle* h)
{
    if (!h) return WIFIRX_EINVAL;
    stream_worker_stop(h);
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (float2* r : h->ring) if (r) (void)hipHostFree(r);
    void* bufs[] = { h->stage_iq, h->stage_frames, h->stage_idx, h->stage_llr, h->stage_car, h->stage_psdu, h->stage_csi, h->stage_stats, h->stage_hbits, h->stage_off, h->s_stats,
                     h->dec_scratch, h->dec_max, h->sbuf, h->s_above, h->s_A, h->s_trig, h->s_frames, h->s_idx,
                     h->s_car, h->s_psdu, h->s_csi, h->s_hbits };
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (h->s_pack) (void)hipFree(h->s_pack);
    if (h->dec_hbits) (void)hipFree(h->dec_hbits);
    if (h->dec_perm) (void)hipFree(h->dec_perm);
    if (h->s_host) (void)hipHostFree(h->s_host);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return WIFIRX_OK;
}

int wifirx_set_param(wifirx_handle* h, int id, double value)
{
    if (!h) return WIFIRX_EINVAL;
    if (!std::isfinite(value)) return fail(h, WIFIRX_EINVAL, "parameter value must be finite");
    stream_worker_wait_idle(h);            // a batch in flight keeps the parameters it started with
    switch (id) {
    case WIFIRX_P_BANDWIDTH:
        if (!(value > 0)) return fail(h, WIFIRX_EINVAL, "bandwidth must be > 0");
        h->cfg.bandwidth = value;
        return WIFIRX_OK;
    case WIFIRX_P_FREQUENCY:
        if (!(value > 0)) return fail(h, WIFIRX_EINVAL, "frequency must be > 0");
        h->cfg.frequency = value;
        return WIFIRX_OK;
    case WIFIRX_P_SENSITIVITY:
        h->cfg.sensitivity = (float)value;
        return WIFIRX_OK;
    case WIFIRX_P_CHAN_EST:
        if (value < WIFIRX_EQ_LS || value > WIFIRX_EQ_STA || value != std::floor(value))
            return fail(h, WIFIRX_EINVAL, "chan_est must be one of WIFIRX_EQ_LS, LMS, COMB, STA");
        h->cfg.chan_est = (int)value;
        return WIFIRX_OK;
    case WIFIRX_P_LLR_CSI:
        h->llr_csi = value != 0;
        return WIFIRX_OK;
    case WIFIRX_P_STREAM_IDX:
        h->stream_want_idx = value != 0;
        return WIFIRX_OK;
    case WIFIRX_P_DECODE_SMALL_MAX:
        if (!(value >= 0) || value > 4e9) return fail(h, WIFIRX_EINVAL, "decode threshold out of range");
        h->decode_small_max = (uint32_t)value;
        return WIFIRX_OK;
    case WIFIRX_P_STREAM_BATCH:
        // two pinned staging buffers of one batch each: at most 2^27 samples (1 GiB) per buffer
        if (!(value >= 0) || value > (double)WIFIRX_STREAM_BATCH_MAX) return fail(h, WIFIRX_EINVAL, "stream batch out of range (0 .. WIFIRX_STREAM_BATCH_MAX)");
        h->stream_batch = (int64_t)value;
        return WIFIRX_OK;
    default:
        return fail(h, WIFIRX_EINVAL, "unknown parameter id");
    }
}

int wifirx_get_stats(const wifirx_handle* h, wifirx_stats* st)
{
    if (!h || !st) return WIFIRX_EINVAL;
    std::lock_guard<std::mutex> lk(const_cast<wifirx_handle*>(h)->mu);
    *st = h->stats;
    return WIFIRX_OK;
}

void* wifirx_stream(wifirx_handle* h) { return h ? (void*)h->stream : nullptr; }

int wifirx_sync(wifirx_handle* h)
{
    if (!h) return WIFIRX_EINVAL;
    stream_worker_wait_idle(h);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WIFIRX_OK;
}

int wifirx_dev_alloc(wifirx_handle* h, size_t bytes, void** out)
{
    if (!h || !out) return WIFIRX_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return oom(h, "hipMalloc", e);
    return WIFIRX_OK;
}

int wifirx_dev_free(wifirx_handle* h, void* p)
{
    if (!h) return WIFIRX_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipFree(p));
    return WIFIRX_OK;
}

int wifirx_memcpy_h2d(wifirx_handle* h, void* dst, const void* src, size_t bytes)
{
    if (!h) return WIFIRX_EINVAL;
    stream_worker_wait_idle(h);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WIFIRX_OK;
}

int wifirx_memcpy_d2h(wifirx_handle* h, void* dst, const void* src, size_t bytes)
{
    if (!h) return WIFIRX_EINVAL;
    stream_worker_wait_idle(h);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return WIFIRX_OK;
}

// ---- batch mode -----------------------------------------------------------------------------

static int check_batch(wifirx_handle* h, uint32_t slot_len, uint32_t n_slots, const wifirx_out* out)
{
    if (!h || !out) return WIFIRX_EINVAL;
    if (!out->frames) return fail(h, WIFIRX_EINVAL, "out->frames is required");
    if (slot_len == 0) return fail(h, WIFIRX_EINVAL, "slot_len must be > 0");
    if (slot_len > 0x7fffffffu) return fail(h, WIFIRX_ERANGE, "slot longer than 2^31 - 1 samples");
    if (h->cfg.max_batch && n_slots > h->cfg.max_batch) return fail(h, WIFIRX_ERANGE, "n_slots exceeds max_batch");
    if (h->cfg.max_slot_len && slot_len > h->cfg.max_slot_len) return fail(h, WIFIRX_ERANGE, "slot_len exceeds max_slot_len");
    if (out->llr && h->cfg.llr_bits == 0) return fail(h, WIFIRX_EINVAL, "llr requested but llr_bits == 0");
    if (out->on_device && (reinterpret_cast<uintptr_t>(out->hbits) & 15)) return fail(h, WIFIRX_EINVAL, "out->hbits must be 16-byte aligned");
    return WIFIRX_OK;
}

static int demod_batch_impl(wifirx_handle* h, const float* iq, int iq_on_device, uint32_t slot_len,
                            uint32_t n_slots, const wifirx_out* out, const uint64_t* slot_off_host);

int wifirx_demod_batch(wifirx_handle* h, const float* iq, int iq_on_device, uint32_t slot_len,
                       uint32_t n_slots, const wifirx_out* out)
{
    return demod_batch_impl(h, iq, iq_on_device, slot_len, n_slots, out, nullptr);
}

int wifirx_demod_batch_v(wifirx_handle* h, const float* iq, int iq_on_device, const uint64_t* slot_off,
                         uint32_t n_slots, const wifirx_out* out)
{
    if (!h) return WIFIRX_EINVAL;
    if (!slot_off) return fail(h, WIFIRX_EINVAL, "slot_off is null");
    uint64_t longest = 1;
    for (uint32_t k = 0; k < n_slots; k++) {
        if (slot_off[k + 1] < slot_off[k]) return fail(h, WIFIRX_EINVAL, "slot_off must not decrease");
        if (slot_off[k + 1] - slot_off[k] > 0x7fffffffu) return fail(h, WIFIRX_ERANGE, "slot longer than 2^31 - 1 samples");
        longest = std::max<uint64_t>(longest, slot_off[k + 1] - slot_off[k]);
    }
    return demod_batch_impl(h, iq, iq_on_device, (uint32_t)longest, n_slots, out, slot_off);
}

static int demod_batch_impl(wifirx_handle* h, const float* iq, int iq_on_device, uint32_t slot_len,
                            uint32_t n_slots, const wifirx_out* out, const uint64_t* slot_off_host)
{
    int rc = check_batch(h, slot_len, n_slots, out);
    if (rc) return rc;
    if (!iq) return fail(h, WIFIRX_EINVAL, "iq is null");
    if (n_slots == 0) return WIFIRX_OK;
    stream_worker_wait_idle(h);            // a stream batch in flight uses the handle's stream and workspaces
    HIP_TRY(h, hipSetDevice(h->device));
    const wr::DemodParams prm = params_of(h);
    const size_t n_iq = slot_off_host ? (size_t)(slot_off_host[n_slots] - slot_off_host[0]) + (size_t)slot_off_host[0]
                                      : (size_t)slot_len * n_slots;
    const uint64_t* d_off = nullptr;
    if (slot_off_host) {
        if ((rc = ensure(h, &h->stage_off, &h->stage_off_bytes, ((size_t)n_slots + 1) * sizeof(uint64_t)))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->stage_off, slot_off_host, ((size_t)n_slots + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
        d_off = reinterpret_cast<const uint64_t*>(h->stage_off);
    }
    const size_t idx_n = (size_t)n_slots * h->cfg.max_sym * 48;
    const float2* d_iq = reinterpret_cast<const float2*>(iq);
    if (!iq_on_device) {
        if ((rc = ensure(h, &h->stage_iq, &h->stage_iq_bytes, n_iq * sizeof(float2)))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->stage_iq, iq, n_iq * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        d_iq = reinterpret_cast<const float2*>(h->stage_iq);
    }
    wifirx_frame* d_fr = out->frames;
    uint8_t* d_idx = out->idx;
    float*   d_llr = out->llr;
    float2*  d_car = reinterpret_cast<float2*>(out->carrier);
    float2*  d_csi = reinterpret_cast<float2*>(out->csi);
    float4*  d_stats = reinterpret_cast<float4*>(out->sym_stats);
    uint32_t* d_hb = out->hbits;
    if (!out->on_device) {
        if ((rc = ensure(h, &h->stage_frames, &h->stage_frames_bytes, n_slots * sizeof(wifirx_frame)))) return rc;
        d_fr = reinterpret_cast<wifirx_frame*>(h->stage_frames);
        if (out->idx) {
            if ((rc = ensure(h, &h->stage_idx, &h->stage_idx_bytes, idx_n))) return rc;
            d_idx = reinterpret_cast<uint8_t*>(h->stage_idx);
            HIP_TRY(h, hipMemsetAsync(d_idx, 0, idx_n, h->stream));
        }
        if (out->llr) {
            if ((rc = ensure(h, &h->stage_llr, &h->stage_llr_bytes, idx_n * h->cfg.llr_bits * sizeof(float)))) return rc;
            d_llr = reinterpret_cast<float*>(h->stage_llr);
            HIP_TRY(h, hipMemsetAsync(d_llr, 0, idx_n * h->cfg.llr_bits * sizeof(float), h->stream));
        }
        if (out->carrier) {
            if ((rc = ensure(h, &h->stage_car, &h->stage_car_bytes, idx_n * sizeof(float2)))) return rc;
            d_car = reinterpret_cast<float2*>(h->stage_car);
            HIP_TRY(h, hipMemsetAsync(d_car, 0, idx_n * sizeof(float2), h->stream));
        }
        if (out->csi) {
            if ((rc = ensure(h, &h->stage_csi, &h->stage_csi_bytes, (size_t)n_slots * 52 * sizeof(float2)))) return rc;
            d_csi = reinterpret_cast<float2*>(h->stage_csi);
            HIP_TRY(h, hipMemsetAsync(d_csi, 0, (size_t)n_slots * 52 * sizeof(float2), h->stream));
        }
        if (out->sym_stats) {
            if ((rc = ensure(h, &h->stage_stats, &h->stage_stats_bytes, (size_t)n_slots * sizeof(float4)))) return rc;
            d_stats = reinterpret_cast<float4*>(h->stage_stats);
        }
        if (out->hbits) {
            if ((rc = ensure(h, &h->stage_hbits, &h->stage_hbits_bytes, idx_n))) return rc;
            d_hb = reinterpret_cast<uint32_t*>(h->stage_hbits);
            HIP_TRY(h, hipMemsetAsync(d_hb, 0, idx_n, h->stream));
        }
    }
    const wr::DemodOut dout = { d_fr, d_idx, d_llr, d_car, d_csi, d_stats, d_hb };
    HIP_TRY(h, wr_launch_demod_batch(h->stream, d_iq, slot_len, n_slots, &prm, &dout, d_off));
    h->stats.samples_in += n_iq;
    if (!out->on_device) {
        HIP_TRY(h, hipMemcpyAsync(out->frames, d_fr, n_slots * sizeof(wifirx_frame), hipMemcpyDeviceToHost, h->stream));
        if (out->idx) HIP_TRY(h, hipMemcpyAsync(out->idx, d_idx, idx_n, hipMemcpyDeviceToHost, h->stream));
        if (out->llr) HIP_TRY(h, hipMemcpyAsync(out->llr, d_llr, idx_n * h->cfg.llr_bits * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        if (out->carrier) HIP_TRY(h, hipMemcpyAsync(out->carrier, d_car, idx_n * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
        if (out->csi) HIP_TRY(h, hipMemcpyAsync(out->csi, d_csi, (size_t)n_slots * 52 * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
        if (out->sym_stats) HIP_TRY(h, hipMemcpyAsync(out->sym_stats, d_stats, (size_t)n_slots * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
        if (out->hbits) HIP_TRY(h, hipMemcpyAsync(out->hbits, d_hb, idx_n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (uint32_t i = 0; i < n_slots; i++) {
            uint32_t f = out->frames[i].flags;
            h->stats.frames_detected += (f & WIFIRX_F_DETECTED) != 0;
            h->stats.frames_signal_ok += (f & WIFIRX_F_SIGNAL) != 0;
            h->stats.frames_complete += (f & WIFIRX_F_COMPLETE) != 0;
        }
    }
    return WIFIRX_OK;
}

int wifirx_time_demod(wifirx_handle* h, const float* iq_dev, uint32_t slot_len, uint32_t n_slots,
                      const wifirx_out* out, int iters, float* ms_mean)
{
    int rc = check_batch(h, slot_len, n_slots, out);
    if (rc) return rc;
    if (!iq_dev || !ms_mean || iters <= 0 || !out->on_device) return fail(h, WIFIRX_EINVAL, "device buffers and iters > 0 required");
    stream_worker_wait_idle(h);
    HIP_TRY(h, hipSetDevice(h->device));
    const wr::DemodParams prm = params_of(h);
    struct Events {                       // destroyed on every exit
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    HIP_TRY(h, hipEventCreate(&ev.e0));
    HIP_TRY(h, hipEventCreate(&ev.e1));
    double total = 0;
    for (int i = 0; i < iters; i++) {
        HIP_TRY(h, hipEventRecord(ev.e0, h->stream));
        const wr::DemodOut dout = { out->frames, out->idx, out->llr, reinterpret_cast<float2*>(out->carrier),
                                    reinterpret_cast<float2*>(out->csi), reinterpret_cast<float4*>(out->sym_stats), out->hbits };
        HIP_TRY(h, wr_launch_demod_batch(h->stream, reinterpret_cast<const float2*>(iq_dev), slot_len, n_slots, &prm, &dout, nullptr));
        HIP_TRY(h, hipEventRecord(ev.e1, h->stream));
        HIP_TRY(h, hipEventSynchronize(ev.e1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, ev.e0, ev.e1));
        total += ms;
    }
    *ms_mean = (float)(total / iters);
    return WIFIRX_OK;
}

int wifirx_synth_slots(wifirx_handle* h, const float* templates, int templates_on_device, uint32_t n_templates,
                       uint32_t frame_len, float* slots, uint32_t slot_len, uint32_t n_slots, uint32_t lead,
                       float snr_db, float cfo_max, uint64_t seed, float* cfo_out)
{
    if (!h || !templates || !slots) return WIFIRX_EINVAL;
    if (n_templates == 0 || frame_len == 0) return fail(h, WIFIRX_EINVAL, "empty templates");
    if (slot_len % 2) return fail(h, WIFIRX_EINVAL, "slot_len must be even");
    stream_worker_wait_idle(h);
    HIP_TRY(h, hipSetDevice(h->device));
    const float2* d_t = reinterpret_cast<const float2*>(templates);
    void* tmp = nullptr;
    if (!templates_on_device) {
        size_t bytes = (size_t)n_templates * frame_len * sizeof(float2);
        hipError_t e = hipMalloc(&tmp, bytes);
        if (e != hipSuccess) return oom(h, "hipMalloc(templates)", e);
        e = hipMemcpyAsync(tmp, templates, bytes, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return fail(h, WIFIRX_EHIP, std::string("template upload: ") + hipGetErrorString(e));
        }
        d_t = reinterpret_cast<const float2*>(tmp);
    }
    // snr_db = NaN: the noiseless channel (gain 1, no AWGN) -- what txgen.impair(snr_db=None) is on the host
    const bool noiseless = std::isnan(snr_db);
    float gain = noiseless ? 1.0f : std::sqrt(std::pow(10.0f, snr_db / 10.0f));
    hipError_t e = wr_launch_synth(h->stream, d_t, n_templates, frame_len, reinterpret_cast<float2*>(slots), slot_len,
                                   n_slots, lead, gain, noiseless ? 0.0f : 1.0f, cfo_max, seed, cfo_out);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return fail(h, WIFIRX_EHIP, std::string("synth launch: ") + hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(h, WIFIRX_EHIP, std::string("synth sync: ") + hipGetErrorString(e2));
    return WIFIRX_OK;
}

}  // extern "C"

#include "wifirx_api_decode.inc"
#include "wifirx_api_stream.inc"
