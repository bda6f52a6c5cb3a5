// C ABI of libxlp_gl.so (declared in include/xlp_gl.h): context management,
// host-pointer drop-in entry points (stage through device memory) and the
// device-pointer entry points.  No CPU fallback exists: every entry point
// fails if no HIP device can be bound.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <algorithm>
#include <string>
#include <vector>
#include <stdlib.h>
#include "../../include/xlp_gl.h"
#include "xlp_internal.hpp"

namespace xlp {

static thread_local std::string g_err;
static Ctx g_ctx;
// One lock for the whole C ABI.  Every entry point below takes it: the context (stream, grow-only scratch slots, the twiddle-table map, the
// device-copy cache) is shared state, and the reference calls this boundary from inside OpenMP regions (FRIProve::prove runs
// NTT_Goldilocks(nX, 1).INTT per group on all threads, friProve.cpp:44-108).  All work goes to one HIP stream anyway, so serialising the
// callers costs nothing the GPU would have overlapped.  Recursive, because entry points call each other (xlp_perm -> xlp_perm_batch,
// the staging copies, ensure_init -> xlp_init).  xlp_last_error() is per thread and takes no lock.
static std::recursive_mutex g_mu;
static hipEvent_t g_marks[8];
static bool g_marked[8];
// XLP_ENTER: the lock of the C ABI; in PLAN MODE (xlp_internal.hpp) the entry point ends here with 0 -- nothing it would do touches the GPU.
// XLP_ENTER_PLAN: the entry points that run in plan mode too: the ones that allocate or size device memory (their HIP calls and kernel launches are
// no-ops there) and the host-only ones.
#define XLP_ENTER_PLAN std::lock_guard<std::recursive_mutex> xlp_api_lock(g_mu)
#define XLP_ENTER XLP_ENTER_PLAN; if (xlp::g_plan_active) return 0

bool g_plan_active = false;
struct PlanState {
    std::map<uintptr_t, size_t> live;
    uint64_t cur = 0, peak = 0, allocs = 0, hbm = 0;
    uintptr_t next = (uintptr_t)1 << 44;                 // fake device addresses: unique, aligned, never dereferenced
    std::vector<size_t> at_peak;                         // sizes of the live allocations when the peak was reached
};
static PlanState g_plan;
hipError_t plan_malloc(void** p, size_t bytes)
{
    if (!g_plan_active) return (hipMalloc)(p, bytes);
    if (!bytes) bytes = 8;
    *p = (void*)g_plan.next;
    g_plan.next += (bytes + 4095) / 4096 * 4096 + 4096;
    g_plan.live[(uintptr_t)*p] = bytes;
    g_plan.cur += bytes; g_plan.allocs++;
    if (g_plan.cur > g_plan.peak) {
        g_plan.peak = g_plan.cur;
        g_plan.at_peak.clear();
        for (const auto& kv : g_plan.live) g_plan.at_peak.push_back(kv.second);
    }
    return hipSuccess;
}
hipError_t plan_free(void* p)
{
    if (!g_plan_active) return (hipFree)(p);
    auto it = g_plan.live.find((uintptr_t)p);
    if (it != g_plan.live.end()) { g_plan.cur -= it->second; g_plan.live.erase(it); }
    return hipSuccess;
}

int xlp_fail(const char* msg)
{
    g_err = msg;
    return -1;
}
int xlp_fail_hip(hipError_t e, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    g_err = buf;
    XLP_HIPV(hipGetLastError());
    return -1;
}

// The scratch slots are shared by every stream of the library (main, side, the slab streams of the step programs): before a slot is freed,
// swapped or lent, ALL of them are drained -- the selected one alone may be the side stream (xlp_stream_select(1)) while a kernel of the main
// stream still reads the slot.  These are rare paths (a slot grows a handful of times per process).
static int ctx_drain_streams(Ctx* ctx)
{
    for (hipStream_t s : {ctx->stream, ctx->main_stream, ctx->side_stream, ctx->aux_stream[0], ctx->aux_stream[1], ctx->aux_stream[2]})
        if (s) XLP_HIP(hipStreamSynchronize(s));
    return 0;
}

int ctx_scratch(Ctx* ctx, size_t bytes, int slot, u64** out)
{
    if (ctx->lent[slot] && ctx->lent_bytes[slot] >= bytes) { *out = (u64*)ctx->lent[slot]; return 0; }     // a lent region that is large enough serves
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            if (ctx_drain_streams(ctx)) return -1;
            XLP_ALLOC(plan_free(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        XLP_ALLOC(plan_malloc((void**)&ctx->scratch[slot], bytes));
        ctx->scratch_bytes[slot] = bytes;
    }
    *out = (u64*)ctx->scratch[slot];
    return 0;
}

static int ensure_init()
{
    if (g_ctx.inited) return 0;
    return xlp_init(0);                                      // (XLP_PLAN=1 in the environment: plan mode, no GPU)
}

// RAII device staging buffer for the host-pointer entry points.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)plan_free(p); }
    int alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 8;
        XLP_ALLOC(plan_malloc((void**)&p, bytes));
        return 0;
    }
    u64* u() { return (u64*)p; }
};

// Events of one pipelined call: all destroyed when the call returns, whichever way it returns.
struct EventPool {
    std::vector<hipEvent_t> all;
    ~EventPool() { for (hipEvent_t e : all) XLP_HIPV(hipEventDestroy(e)); }
    int make(hipEvent_t* out, bool timing = false)
    {
        hipEvent_t e = nullptr;
        if (timing) XLP_HIP(hipEventCreate(&e)); else XLP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        all.push_back(e);
        *out = e;
        return 0;
    }
    // record on `from`; `to` (if any) waits for it
    int mark(hipStream_t from, hipStream_t to, hipEvent_t* out = nullptr)
    {
        hipEvent_t e;
        if (make(&e)) return -1;
        XLP_HIP(hipEventRecord(e, from));
        if (to) XLP_HIP(hipStreamWaitEvent(to, e, 0));
        if (out) *out = e;
        return 0;
    }
};
// A device allocation that is either handed on (release) or freed on return.
struct DevOwner {
    void* p = nullptr;
    ~DevOwner() { if (p) (void)plan_free(p); }
    void* release() { void* q = p; p = nullptr; return q; }
};

// ---- device copies of host buffers (see xlp_cache_enable below)
struct CacheEnt { const char* host; size_t bytes; void* dev; uint64_t tick; };
std::vector<CacheEnt> g_cache;
size_t g_cache_cap = 0, g_cache_used = 0;
uint64_t g_cache_tick = 0;

void cache_drop(size_t i)
{
    (void)plan_free(g_cache[i].dev);
    g_cache_used -= g_cache[i].bytes;
    g_cache.erase(g_cache.begin() + i);
}
void cache_drop_overlapping(const void* host, size_t bytes)
{
    const char* a = (const char*)host;
    for (size_t i = g_cache.size(); i-- > 0;)
        if (a < g_cache[i].host + g_cache[i].bytes && g_cache[i].host < a + bytes) cache_drop(i);
}
// takes ownership of dev (frees it when the cache is off or too small for it)
void cache_put(const void* host, size_t bytes, void* dev)
{
    cache_drop_overlapping(host, bytes);
    if (bytes > g_cache_cap) { (void)plan_free(dev); return; }
    while (g_cache_used + bytes > g_cache_cap && !g_cache.empty()) {
        size_t lru = 0;
        for (size_t i = 1; i < g_cache.size(); i++) if (g_cache[i].tick < g_cache[lru].tick) lru = i;
        cache_drop(lru);
    }
    g_cache.push_back({(const char*)host, bytes, dev, ++g_cache_tick});
    g_cache_used += bytes;
}
// detach the cached copy of exactly this range (the caller becomes the owner), if any
void* cache_take(const void* host, size_t bytes)
{
    for (size_t i = 0; i < g_cache.size(); i++)
        if (g_cache[i].host == (const char*)host && g_cache[i].bytes == bytes) {
            void* d = g_cache[i].dev;
            g_cache_used -= bytes;
            g_cache.erase(g_cache.begin() + i);
            return d;
        }
    return nullptr;
}

}  // namespace xlp

using namespace xlp;
static void commit_times_clear();

#define H2D(buf, host, bytes) do { if (xlp_copy_h2d((buf).p, (host), (bytes))) return -1; } while (0)
#define D2H(host, buf, bytes) do { if (xlp_copy_d2h((host), (buf).p, (bytes))) return -1; } while (0)

extern "C" {

// xlp_scratch_release: give grow-only scratch slots back (bit k of mask = slot k): a prover that shares the card with others releases what its
// last stage grew (the fflonk commitments: 11 GB of sort / bucket scratch at 9 x 2^22 points) instead of holding it for the process' life
// xlp_scratch_lend: let slot `slot` use a caller-owned device region instead of growing its own (ptr = NULL: take it back).  A request the
// region cannot hold falls through to the slot's own scratch.  The caller keeps the region alive and unused until it takes it back.
int xlp_scratch_lend(uint32_t slot, void* dev_ptr, size_t bytes)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (slot >= 9) return xlp_fail("xlp_scratch_lend: slot 0..8");
    if (ctx_drain_streams(&g_ctx)) return -1;
    if (slot == 8) { g_ctx.fft_tw_log = 0; g_ctx.fft_tw_ptr = nullptr; }
    g_ctx.lent[slot] = dev_ptr;
    g_ctx.lent_bytes[slot] = dev_ptr ? bytes : 0;
    return 0;
}
int xlp_scratch_release(uint32_t mask)
{
    XLP_ENTER_PLAN;
    if (!g_ctx.inited) return 0;
    if (ctx_drain_streams(&g_ctx)) return -1;
    for (int k = 0; k < 9; k++)
        if (((mask >> k) & 1) && g_ctx.scratch[k]) {
            XLP_ALLOC(plan_free(g_ctx.scratch[k]));
            g_ctx.scratch[k] = nullptr;
            g_ctx.scratch_bytes[k] = 0;
            if (k == 8) { g_ctx.fft_tw_log = 0; g_ctx.fft_tw_ptr = nullptr; }
        }
    return 0;
}


int xlp_version(void) { return 0x000100; }

const char* xlp_last_error(void) { return g_err.c_str(); }

int xlp_plan_begin(uint64_t hbm_bytes)
{
    XLP_ENTER_PLAN;
    if (g_ctx.inited && !g_plan_active) return xlp_fail("xlp_plan_begin: the library is already bound to a GPU (plan mode starts before any other call)");
    g_plan_active = true;
    g_plan.hbm = hbm_bytes ? hbm_bytes : (uint64_t)288 << 30;          // MI355X: 288 GiB of HBM3E
    g_ctx.device = 0; g_ctx.ncu = 256; g_ctx.inited = true;
    return 0;
}
int xlp_plan_active(void) { return g_plan_active ? 1 : 0; }
int xlp_plan_report(uint64_t* peak_bytes, uint64_t* live_bytes, uint64_t* allocations, uint64_t* largest_at_peak, uint32_t nlargest, uint64_t scratch_slot_bytes[9])
{
    XLP_ENTER_PLAN;
    if (!g_plan_active) return xlp_fail("xlp_plan_report: not in plan mode");
    if (peak_bytes) *peak_bytes = g_plan.peak;
    if (live_bytes) *live_bytes = g_plan.cur;
    if (allocations) *allocations = g_plan.allocs;
    std::vector<size_t> v = g_plan.at_peak;
    std::sort(v.begin(), v.end(), [](size_t a, size_t b) { return a > b; });
    for (uint32_t i = 0; i < nlargest && largest_at_peak; i++) largest_at_peak[i] = i < v.size() ? v[i] : 0;
    for (int k = 0; k < 9 && scratch_slot_bytes; k++) scratch_slot_bytes[k] = g_ctx.scratch_bytes[k];
    return 0;
}

int xlp_init(int device)
{
    XLP_ENTER_PLAN;
    if (!g_plan_active && !g_ctx.inited && getenv("XLP_PLAN")) { if (xlp_plan_begin(strtoull(getenv("XLP_PLAN_HBM_BYTES") ? getenv("XLP_PLAN_HBM_BYTES") : "0", nullptr, 0))) return -1; }
    if (g_plan_active) return 0;
    if (g_ctx.inited && g_ctx.device == device) return 0;
    if (g_ctx.inited) return xlp_fail("xlp_init: already bound to another device");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return xlp_fail("xlp_init: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return xlp_fail("xlp_init: device index out of range");
    XLP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    XLP_HIP(hipGetDeviceProperties(&prop, device));
    g_ctx.device = device;
    g_ctx.ncu = (u32)prop.multiProcessorCount / 8 * 8;
    if (g_ctx.ncu == 0) g_ctx.ncu = 8;
    XLP_HIP(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    g_ctx.own_stream = true;
    if (ntt_init(&g_ctx)) return -1;
    if (poseidon_init(&g_ctx)) return -1;
    g_ctx.inited = true;
    return 0;
}

int xlp_set_stream(void* hip_stream)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (g_ctx.selected != 0) return xlp_fail("xlp_set_stream: stream 1 is selected (xlp_stream_select(0) first)");
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    if (g_ctx.own_stream) { XLP_HIP(hipStreamDestroy(g_ctx.stream)); g_ctx.own_stream = false; }
    g_ctx.stream = (hipStream_t)hip_stream;
    return 0;
}

int xlp_sync(void)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    if (g_ctx.side_stream && g_ctx.selected == 0) XLP_HIP(hipStreamSynchronize(g_ctx.side_stream));
    return 0;
}

// ---- two streams for a caller that overlaps exchanges with kernels (host/starks.hpp, row-sharded commit).  Stream 0 is the library's
// stream (or the one given to xlp_set_stream), stream 1 a second one created on first use.  select: every later xlp_dev_* / xlp_comm_* call
// enqueues on that stream.  wait(a, b): work enqueued on stream a from now on starts after everything enqueued on stream b so far.
static hipStream_t stream_of(int which) { return which == 0 ? (g_ctx.selected == 0 ? g_ctx.stream : g_ctx.main_stream) : g_ctx.side_stream; }
int xlp_stream_select(int which)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (which != 0 && which != 1) return xlp_fail("xlp_stream_select: stream 0 or 1");
    if (which == g_ctx.selected) return 0;
    if (which == 1) {
        if (!g_ctx.side_stream) XLP_HIP(hipStreamCreateWithFlags(&g_ctx.side_stream, hipStreamNonBlocking));
        g_ctx.main_stream = g_ctx.stream;
        g_ctx.stream = g_ctx.side_stream;
    } else g_ctx.stream = g_ctx.main_stream;
    g_ctx.selected = which;
    return 0;
}
// finer than xlp_stream_wait: mark(s, slot) records "everything enqueued on stream s so far" in one of 8 reusable marks, wait_mark(s, slot)
// makes later work of stream s wait for what that mark recorded (a mark never recorded: no wait)
int xlp_stream_mark(int stream, int slot)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if ((stream != 0 && stream != 1) || slot < 0 || slot >= 8) return xlp_fail("xlp_stream_mark: stream 0 or 1, mark 0..7");
    if (stream == 1 && !g_ctx.side_stream) XLP_HIP(hipStreamCreateWithFlags(&g_ctx.side_stream, hipStreamNonBlocking));
    if (!g_marks[slot]) XLP_HIP(hipEventCreateWithFlags(&g_marks[slot], hipEventDisableTiming));
    XLP_HIP(hipEventRecord(g_marks[slot], stream_of(stream)));
    g_marked[slot] = true;
    return 0;
}
int xlp_stream_wait_mark(int stream, int slot)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if ((stream != 0 && stream != 1) || slot < 0 || slot >= 8) return xlp_fail("xlp_stream_wait_mark: stream 0 or 1, mark 0..7");
    if (!g_marked[slot]) return 0;
    if (stream == 1 && !g_ctx.side_stream) XLP_HIP(hipStreamCreateWithFlags(&g_ctx.side_stream, hipStreamNonBlocking));
    XLP_HIP(hipStreamWaitEvent(stream_of(stream), g_marks[slot], 0));
    return 0;
}
int xlp_stream_wait(int waiter, int signaller)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if ((waiter != 0 && waiter != 1) || (signaller != 0 && signaller != 1)) return xlp_fail("xlp_stream_wait: stream 0 or 1");
    if (waiter == signaller) return 0;
    if (!g_ctx.side_stream) XLP_HIP(hipStreamCreateWithFlags(&g_ctx.side_stream, hipStreamNonBlocking));
    EventPool pool;
    return pool.mark(stream_of(signaller), stream_of(waiter));
}

void xlp_shutdown(void)
{
    XLP_ENTER_PLAN;
    if (!g_ctx.inited) return;
    XLP_HIPV(hipStreamSynchronize(g_ctx.stream));
    for (auto& kv : g_ctx.tables) (void)plan_free(kv.second);
    g_ctx.tables.clear();
    for (int i = 0; i < 9; i++) {
        if (g_ctx.scratch[i]) (void)plan_free(g_ctx.scratch[i]);
        g_ctx.scratch[i] = nullptr;
        g_ctx.scratch_bytes[i] = 0;
        g_ctx.lent[i] = nullptr;                               // a caller-owned region is not ours past a shutdown: a re-init must not hand it out
        g_ctx.lent_bytes[i] = 0;
    }
    g_ctx.fft_tw_log = 0; g_ctx.fft_tw_ptr = nullptr;
    if (g_ctx.selected != 0) { g_ctx.stream = g_ctx.main_stream; g_ctx.selected = 0; }
    for (int i = 0; i < 8; i++) { if (g_marks[i]) XLP_HIPV(hipEventDestroy(g_marks[i])); g_marks[i] = nullptr; g_marked[i] = false; }
    commit_times_clear();
    if (g_ctx.side_stream) { XLP_HIPV(hipStreamSynchronize(g_ctx.side_stream)); XLP_HIPV(hipStreamDestroy(g_ctx.side_stream)); g_ctx.side_stream = nullptr; }
    if (g_ctx.own_stream) XLP_HIPV(hipStreamDestroy(g_ctx.stream));
    if (g_ctx.copy_stream) { XLP_HIPV(hipStreamSynchronize(g_ctx.copy_stream)); XLP_HIPV(hipStreamDestroy(g_ctx.copy_stream)); g_ctx.copy_stream = nullptr; }
    for (hipStream_t& a : g_ctx.aux_stream) if (a) { XLP_HIPV(hipStreamSynchronize(a)); XLP_HIPV(hipStreamDestroy(a)); a = nullptr; }
    if (g_ctx.down_stream) { XLP_HIPV(hipStreamSynchronize(g_ctx.down_stream)); XLP_HIPV(hipStreamDestroy(g_ctx.down_stream)); g_ctx.down_stream = nullptr; }
    while (!g_cache.empty()) cache_drop(0);
    g_cache_cap = 0;
    bn128_shutdown();
    comm_shutdown();
    g_ctx.stream = nullptr;
    g_ctx.own_stream = false;
    g_ctx.inited = false;
}

// ------------------------------------------------------------- device layer
int xlp_dev_ntt(uint64_t* dst, const uint64_t* src, uint64_t n, uint64_t ncols, int inverse)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_ntt(&g_ctx, dst, src, n, ncols, inverse);
}
int xlp_dev_extend(uint64_t* out, const uint64_t* in, uint64_t n_ext, uint64_t n, uint64_t ncols, uint64_t* scratch)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_extend(&g_ctx, out, in, n_ext, n, ncols, scratch);
}
int xlp_dev_merkle(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_merkle(&g_ctx, nodes, src, w, h);
}
int xlp_dev_perm_batch(uint64_t* out, const uint64_t* in, uint64_t count)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_perm(&g_ctx, out, in, count);
}
int xlp_dev_linear_hash(uint64_t* out, const uint64_t* src, uint64_t w, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_linear_hash(&g_ctx, out, src, w, h);
}
int xlp_dev_fri_fold(uint64_t* pol2, const uint64_t* pol, uint32_t bits, uint32_t red, const uint64_t x[3], uint64_t shift_inv)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_fri_fold(&g_ctx, pol2, pol, bits, red, x, shift_inv);
}
int xlp_dev_transpose3(uint64_t* dst, const uint64_t* src, uint64_t n, uint32_t tbits)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_transpose3(&g_ctx, dst, src, n, tbits);
}
int xlp_dev_fill(uint64_t* dst, uint64_t n, uint64_t seed, uint64_t start)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_fill(&g_ctx, dst, n, seed, start);
}

int xlp_dev_batch_inverse(uint64_t* res, const uint64_t* src, uint64_t n, uint32_t dim)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_batch_inverse(&g_ctx, res, src, n, dim);
}
int xlp_dev_calculate_z(uint64_t* z, const uint64_t* num, const uint64_t* den, uint64_t n)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_calculate_z(&g_ctx, z, 3, num, 3, den, 3, n);
}
int xlp_dev_calculate_z_strided(uint64_t* z, uint64_t z_stride, const uint64_t* num, uint64_t num_stride, const uint64_t* den, uint64_t den_stride, uint64_t n)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_calculate_z(&g_ctx, z, z_stride, num, num_stride, den, den_stride, n);
}
int xlp_dev_scale3_strided(uint64_t* z, uint64_t z_stride, uint64_t n, const uint64_t s[3])
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n && (!z || !s)) return xlp_fail("xlp_dev_scale3_strided: null pointer");
    return dev_scale3(&g_ctx, z, z_stride, n, s);
}
int xlp_dev_calculate_h1h2(uint64_t* h1, uint64_t* h2, const uint64_t* f, const uint64_t* t, uint64_t n, uint32_t dim)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_calculate_h1h2(&g_ctx, h1, dim, h2, dim, f, dim, t, dim, n, dim);
}
int xlp_dev_calculate_h1h2_strided(uint64_t* h1, uint64_t h1_stride, uint64_t* h2, uint64_t h2_stride, const uint64_t* f, uint64_t f_stride,
                                   const uint64_t* t, uint64_t t_stride, uint64_t n, uint32_t dim)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_calculate_h1h2(&g_ctx, h1, h1_stride, h2, h2_stride, f, f_stride, t, t_stride, n, dim);
}
int xlp_expr_create(xlp_expr_program** out, const uint64_t* code, uint64_t ncode, uint64_t ntmp1, uint64_t ntmp3)
{
    XLP_ENTER_PLAN;
    return expr_create(out, code, ncode, ntmp1, ntmp3, 0);   // host-only: validation + compilation to the device form
}
int xlp_expr_create_ex(xlp_expr_program** out, const uint64_t* code, uint64_t ncode, uint64_t ntmp1, uint64_t ntmp3, uint64_t flags)
{
    XLP_ENTER_PLAN;
    return expr_create(out, code, ncode, ntmp1, ntmp3, flags);
}
int xlp_expr_destroy(xlp_expr_program* prog) { XLP_ENTER_PLAN; return expr_destroy(prog); }
int xlp_expr_codegen(const xlp_expr_program* prog, const char* path) { XLP_ENTER_PLAN; return expr_codegen(prog, path); }
int xlp_expr_attach(xlp_expr_program* prog, const char* path) { XLP_ENTER_PLAN; return expr_attach(prog, path); }
int xlp_dev_expr_run(const xlp_expr_program* prog, const xlp_steps_params* params, uint64_t row0, uint64_t nrows)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_expr_run(&g_ctx, prog, params, row0, nrows);
}
int xlp_dev_merkle_proofs(uint64_t* out, const uint64_t* src, const uint64_t* nodes, uint64_t w, uint64_t h, const uint64_t* idxs, uint64_t nq)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (nq == 0) return 0;
    if (!out || !nodes || !idxs || (w && !src)) return xlp_fail("xlp_dev_merkle_proofs: null pointer");
    for (uint64_t q = 0; q < nq; q++) if (idxs[q] >= h) return xlp_fail("xlp_dev_merkle_proofs: index out of range");
    uint32_t levels = 0;
    while ((1ull << levels) < h) levels++;
    const size_t per = w + 4ull * levels;
    DevBuf di, dout;
    if (di.alloc(nq * sizeof(u64)) || dout.alloc(nq * per * sizeof(u64))) return -1;
    H2D(di, idxs, nq * sizeof(u64));
    if (dev_merkle_proofs(&g_ctx, dout.u(), src, nodes, w, h, di.u(), nq)) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    D2H(out, dout, nq * per * sizeof(u64));
    return 0;
}
// ---- column-major device sections (colmajor.hip, ntt_lde_cm.hip, poseidon.hip; include/xlp_gl.h "column-major sections")
int xlp_dev_rows_to_cm(uint64_t* dst, uint64_t dst_cstride, uint64_t dst_coset_stride, const uint64_t* src, uint64_t ld, uint64_t col0, uint64_t ncols, uint64_t rows, uint32_t ext_bits)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (ncols && rows && (!dst || !src)) return xlp_fail("xlp_dev_rows_to_cm: null pointer");
    return dev_rows_to_cm(&g_ctx, dst, dst_cstride, dst_coset_stride, src, ld, col0, ncols, rows, ext_bits);
}
int xlp_dev_cm_to_rows(uint64_t* dst, uint64_t ld, uint64_t col0, const uint64_t* src, uint64_t src_cstride, uint64_t src_coset_stride, uint64_t ncols, uint64_t rows, uint32_t ext_bits)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (ncols && rows && (!dst || !src)) return xlp_fail("xlp_dev_cm_to_rows: null pointer");
    return dev_cm_to_rows(&g_ctx, dst, ld, col0, src, src_cstride, src_coset_stride, ncols, rows, ext_bits);
}
int xlp_dev_copy_view(uint64_t* dst, uint64_t dst_row_stride, uint64_t dst_comp_stride, const uint64_t* src, uint64_t src_row_stride, uint64_t src_comp_stride, uint64_t n, uint32_t dim)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n && dim && (!dst || !src)) return xlp_fail("xlp_dev_copy_view: null pointer");
    return dev_copy_view(&g_ctx, dst, dst_row_stride, dst_comp_stride, src, src_row_stride, src_comp_stride, n, dim);
}
int xlp_dev_copy_3d(uint64_t* dst, uint64_t dst_stride0, uint64_t dst_stride1, const uint64_t* src, uint64_t src_stride0, uint64_t src_stride1, uint64_t n0, uint64_t n1, uint64_t n)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n0 && n1 && n && (!dst || !src)) return xlp_fail("xlp_dev_copy_3d: null pointer");
    return dev_copy_3d(&g_ctx, dst, dst_stride0, dst_stride1, src, src_stride0, src_stride1, n0, n1, n);
}
int xlp_dev_extend_cm(uint64_t* out, uint64_t out_cstride, uint64_t out_coset_stride, const uint64_t* in, uint64_t in_cstride, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (n && ncols && (!out || !in)) return xlp_fail("xlp_dev_extend_cm: null pointer");
    return dev_extend_cm(&g_ctx, out, out_cstride, out_coset_stride, in, in_cstride, n_ext, n, ncols);
}
int xlp_dev_linear_hash_absorb_cm(uint64_t* state, const uint64_t* src, uint64_t cstride, uint64_t coset_stride, uint64_t col0, uint64_t ncols, uint64_t h, uint32_t ext_bits, int first, int last)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_linear_hash_absorb_cm(&g_ctx, state, src, cstride, coset_stride, col0, ncols, h, ext_bits, first, last);
}
int xlp_dev_merkle_cm(uint64_t* nodes, const uint64_t* src, uint64_t cstride, uint64_t coset_stride, uint64_t w, uint64_t h, uint32_t ext_bits)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_merkle_cm(&g_ctx, nodes, src, cstride, coset_stride, w, h, ext_bits);
}
int xlp_dev_merkle_proofs_cm(uint64_t* out, const uint64_t* src, uint64_t cstride, uint64_t coset_stride, const uint64_t* nodes, uint64_t w, uint64_t h, uint32_t ext_bits, const uint64_t* idxs, uint64_t nq)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (nq == 0) return 0;
    if (!out || !nodes || !idxs || (w && !src)) return xlp_fail("xlp_dev_merkle_proofs_cm: null pointer");
    for (uint64_t q = 0; q < nq; q++) if (idxs[q] >= h) return xlp_fail("xlp_dev_merkle_proofs_cm: index out of range");
    uint32_t levels = 0;
    while ((1ull << levels) < h) levels++;
    const size_t per = w + 4ull * levels;
    DevBuf di, dout;
    if (di.alloc(nq * sizeof(u64)) || dout.alloc(nq * per * sizeof(u64))) return -1;
    H2D(di, idxs, nq * sizeof(u64));
    if (dev_merkle_proofs_cm(&g_ctx, dout.u(), src, cstride, coset_stride, nodes, w, h, ext_bits, di.u(), nq)) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    D2H(out, dout, nq * per * sizeof(u64));
    return 0;
}
int xlp_dev_gather_rows(uint64_t* dst, const uint64_t* src, uint64_t src_stride, uint64_t n, uint32_t dim, uint64_t seed)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_gather_rows(&g_ctx, dst, src, src_stride, n, dim, seed);
}
int xlp_dev_copy_cols(uint64_t* dst, uint64_t dst_stride, const uint64_t* src, uint64_t src_stride, uint64_t n, uint32_t dim)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_copy_cols(&g_ctx, dst, dst_stride, src, src_stride, n, dim);
}
int xlp_dev_q_split(uint64_t* qq2, const uint64_t* qq1, uint64_t n, uint64_t n_ext, uint32_t q_deg, uint32_t q_dim, uint64_t shift_in)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_q_split(&g_ctx, qq2, qq1, n, n_ext, q_deg, q_dim, shift_in);
}
int xlp_dev_geometric(uint64_t* out, uint64_t n, uint32_t dim, const uint64_t* first, const uint64_t* ratio)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!first || !ratio) return xlp_fail("xlp_dev_geometric: null first / ratio");
    return dev_geometric(&g_ctx, out, n, dim, first, ratio);
}
int xlp_dev_x_div_x_sub_xi(uint64_t* out, const uint64_t* x, const uint64_t* xi, uint64_t n)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!xi) return xlp_fail("xlp_dev_x_div_x_sub_xi: null xi");
    return dev_x_div_x_sub_xi(&g_ctx, out, x, xi, n);
}
int xlp_dev_dot_columns(uint64_t* out, const xlp_dot_entry* entries, uint64_t nentries, uint32_t ngroups,
                        const uint64_t* coefs, uint64_t coefs_size, uint64_t nrows)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (!nrows || !ngroups) return 0;
    if (!out || (nentries && (!entries || !coefs))) return xlp_fail("xlp_dev_dot_columns: null pointer");
    return dev_dot_columns(&g_ctx, out, entries, nentries, ngroups, coefs, coefs_size, nrows);
}

int xlp_dev_evmap(uint64_t* evals, const uint64_t* LEv, const uint64_t* LpEv, uint64_t n, uint32_t extend_bits, uint32_t nev,
                  const xlp_evmap_entry* entries)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (nev && !entries) return xlp_fail("xlp_dev_evmap: null entries");
    return dev_evmap(&g_ctx, evals, LEv, LpEv, n, extend_bits, nev, entries);
}

int xlp_dev_evmap_cm(uint64_t* evals, const uint64_t* LEv, const uint64_t* LpEv, uint64_t n, uint32_t nev, const xlp_evmap_entry* entries)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (nev && !entries) return xlp_fail("xlp_dev_evmap_cm: null entries");
    return dev_evmap_cm(&g_ctx, evals, LEv, LpEv, n, nev, entries);
}
int xlp_dev_dot_columns_cm(uint64_t* out, uint64_t out_cstride, const xlp_dot_entry* entries, uint64_t nentries, uint32_t ngroups,
                           const uint64_t* coefs, uint64_t coefs_size, uint64_t nrows)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (!nrows || !ngroups) return 0;
    if (!out || (nentries && (!entries || !coefs))) return xlp_fail("xlp_dev_dot_columns_cm: null pointer");
    return dev_dot_columns_cm(&g_ctx, out, out_cstride, entries, nentries, ngroups, coefs, coefs_size, nrows);
}

int xlp_dev_perm_resident(uint64_t* out, uint64_t count, uint64_t iters)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (count && !out) return xlp_fail("xlp_dev_perm_resident: null pointer");
    return dev_perm_resident(&g_ctx, out, count, iters);
}
int xlp_dev_lde_resident(uint64_t* sink, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!sink) return xlp_fail("xlp_dev_lde_resident: null pointer");
    if (!n || (n & (n - 1)) || n_ext < n || (n_ext & (n_ext - 1))) return xlp_fail("xlp_dev_lde_resident: sizes must be powers of two, n_ext >= n");
    return dev_lde_resident(&g_ctx, sink, (u32)__builtin_ctzll(n), (u32)(__builtin_ctzll(n_ext) - __builtin_ctzll(n)), ncols);
}
int xlp_profile_enable(int on)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    for (hipEvent_t e : g_ctx.prof_events) XLP_HIPV(hipEventDestroy(e));
    g_ctx.prof_events.clear();
    g_ctx.profile = on != 0;
    return 0;
}

int xlp_profile_read(double ms_sum[3], uint64_t* count)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    ms_sum[0] = ms_sum[1] = ms_sum[2] = 0;
    const size_t n = g_ctx.prof_events.size() / 4;
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float ms = 0;
            XLP_HIP(hipEventElapsedTime(&ms, g_ctx.prof_events[4 * i + k], g_ctx.prof_events[4 * i + k + 1]));
            ms_sum[k] += ms;
        }
    *count = n;
    return 0;
}

int xlp_dev_alloc(void** ptr, size_t bytes)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    XLP_ALLOC(plan_malloc((void**)ptr, bytes ? bytes : 8));
    return 0;
}
int xlp_dev_mem_info(size_t* free_bytes, size_t* total_bytes)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    size_t f = 0, t = 0;
    if (g_plan_active) { t = g_plan.hbm; f = g_plan.cur < t ? t - g_plan.cur : 0; }
    XLP_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}
int xlp_dev_free(void* ptr)
{
    XLP_ENTER_PLAN;
    if (!ptr) return 0;
    XLP_ALLOC(plan_free(ptr));
    return 0;
}
int xlp_dev_memset(void* dst_dev, int value, size_t bytes)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!bytes) return 0;
    XLP_HIP(hipMemsetAsync(dst_dev, value, bytes, g_ctx.stream));
    return 0;
}
int xlp_copy_h2d(void* dst_dev, const void* src_host, size_t bytes)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!bytes) return 0;
    XLP_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, g_ctx.stream));
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    return 0;
}
int xlp_copy_d2h(void* dst_host, const void* src_dev, size_t bytes)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (!bytes) return 0;
    if (g_plan_active) { memset(dst_host, 0, bytes); return 0; }        // (plan mode: what the host reads back is zeros, deterministically)
    // the host range is rewritten: device copies of it kept by the host-pointer cache (xlp_cache_enable) are stale from here on.  Every
    // host-pointer entry point returns its results through this function (the D2H macro), so none of them can leave a stale copy behind.
    cache_drop_overlapping(dst_host, bytes);
    XLP_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    return 0;
}

int xlp_host_alloc(void** ptr, size_t bytes)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (g_plan_active) { *ptr = calloc(bytes ? bytes : 8, 1); return *ptr ? 0 : xlp_fail("xlp_host_alloc: out of host memory"); }   // (untouched pages cost nothing)
    XLP_HIP(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault));
    return 0;
}
int xlp_host_free(void* ptr)
{
    XLP_ENTER_PLAN;
    if (!ptr) return 0;
    if (g_plan_active) { free(ptr); return 0; }
    XLP_HIP(hipHostFree(ptr));
    return 0;
}
int xlp_host_register(void* ptr, size_t bytes)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!ptr || !bytes) return xlp_fail("xlp_host_register: empty range");
    XLP_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}
int xlp_host_unregister(void* ptr)
{
    XLP_ENTER;
    if (!ptr) return 0;
    XLP_HIP(hipHostUnregister(ptr));
    return 0;
}

int xlp_dev_extend_cols(uint64_t* out, const uint64_t* in, uint64_t n_ext, uint64_t n, uint64_t ld, uint64_t col0, uint64_t ncols)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return dev_extend_cols(&g_ctx, out, in, n_ext, n, ld, col0, ncols);
}
int xlp_dev_linear_hash_absorb(uint64_t* state, const uint64_t* src, uint64_t ld, uint64_t col0, uint64_t ncols, uint64_t h, int first, int last)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_linear_hash_absorb(&g_ctx, state, src, ld, col0, ncols, h, first, last);
}
int xlp_dev_merkle_levels(uint64_t* nodes, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return dev_merkle_levels(&g_ctx, nodes, h);
}

// extendPol + merkelize of one section, band by band (starks.cpp:53-60 / :134-141 / :215-222 as ONE pipeline): the leaf sponge of
// a row absorbs its columns in order, so a band of columns can be hashed into the 4-word leaf state as soon as it is extended.
// With h_src the bands are first uploaded on the copy stream -- band b+1 crosses PCIe while band b is extended and absorbed.
// the reference times the two halves of a commit apart (TimerStart(STARK_STEP_n_LDE) / (STARK_STEP_n_MERKLETREE), starks.cpp:51-58,133-139);
// here the bands interleave them, so each half is the sum of its kernels' event times on the compute stream (xlp_dev_commit_last_times)
struct CommitTimes {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> lde, merkle;
    void clear()
    {
        for (auto* v : {&lde, &merkle}) { for (auto& pr : *v) { XLP_HIPV(hipEventDestroy(pr.first)); XLP_HIPV(hipEventDestroy(pr.second)); } v->clear(); }
    }
    int begin(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, hipStream_t st)
    {
        hipEvent_t a = nullptr, b = nullptr;
        XLP_HIP(hipEventCreate(&a)); XLP_HIP(hipEventCreate(&b));
        v.emplace_back(a, b);
        XLP_HIP(hipEventRecord(a, st));
        return 0;
    }
    int end(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, hipStream_t st) { XLP_HIP(hipEventRecord(v.back().second, st)); return 0; }
};
static CommitTimes g_commit_times;
static void commit_times_clear() { g_commit_times.clear(); }

static int commit_bands(uint64_t* sec_n, uint64_t* sec_2ns, uint64_t* nodes, const uint64_t* h_src, uint64_t n_ext, uint64_t n, uint64_t ncols, double* upload_ms)
{
    CommitTimes& T = g_commit_times;
    T.clear();
    if (upload_ms) *upload_ms = 0.0;
    if (n == 0 || n_ext == 0) return 0;
    if ((n & (n - 1)) || (n_ext & (n_ext - 1)) || n_ext < n) return xlp_fail("commit: sizes must be powers of two, n_ext >= n");
    Ctx* ctx = &g_ctx;
    u64 band = 64;
    if (const char* e = getenv("XLP_COMMIT_BAND")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 8 && v % 8 == 0) band = v; }
    if (ncols <= 4 || ncols <= band) {                       // one band (or the no-hash rule of narrow rows): nothing to pipeline
        if (h_src && ncols) {
            XLP_HIP(hipMemcpyAsync(sec_n, h_src, n * ncols * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
        }
        if (ncols) {
            if (T.begin(T.lde, ctx->stream) || dev_extend(ctx, sec_2ns, sec_n, n_ext, n, ncols, nullptr) || T.end(T.lde, ctx->stream)) return -1;
        }
        if (T.begin(T.merkle, ctx->stream) || dev_merkle(ctx, nodes, sec_2ns, ncols, n_ext) || T.end(T.merkle, ctx->stream)) return -1;
        return 0;
    }
    // Band plan.  From device memory: uniform bands.  From the host (step 1) the copy stream and the extension + hashing run at nearly the
    // same rate per column (PCIe 1.2-1.4 ms, compute 1.1 ms per column of 2^23 rows); the DMA engine moves wide row pieces faster (2 KB
    // pieces: 55 GB/s, 512 B: 49 GB/s, 256 B: 36 GB/s), but a band is only worked on once it has arrived, and what the pipeline cannot hide is
    // the work on the LAST band: bands that start wide and narrow by 15 % each, so the compute stream keeps up and ends on a small one
    // (751 columns: 168, 128, 112, 88, 80, 64, 56, 55 -- profiles/r3/commit_band_ab.txt).
    // XLP_COMMIT_BAND forces uniform bands.
    std::vector<u64> starts, widths;
    if (h_src && !getenv("XLP_COMMIT_BAND") && ncols >= 320) {
        std::vector<u64> rev;                                 // built from the end: 48, then wider by 1 / 0.85 each, at most 160
        double w = 48.0;
        for (u64 rem = ncols / 8 * 8; rem;) {                 // every band starts on a multiple of 8 columns (one sponge block)
            u64 ww = ((u64)(w + 4.0)) / 8 * 8;
            if (ww > rem) ww = rem;
            rev.push_back(ww); rem -= ww;
            w = w / 0.85 < 160.0 ? w / 0.85 : 160.0;
        }
        rev[0] += ncols % 8;                                  // the odd columns go with the last band
        if (rev.size() > 1 && rev.back() < 64) { rev[rev.size() - 2] += rev.back(); rev.pop_back(); }   // no sliver in front
        u64 c0 = 0;
        for (size_t k = rev.size(); k-- > 0;) { starts.push_back(c0); widths.push_back(rev[k]); c0 += rev[k]; }
    } else {
        for (u64 c0 = 0; c0 < ncols; c0 += band) { starts.push_back(c0); widths.push_back(ncols - c0 < band ? ncols - c0 : band); }
    }
    const u64 nb = starts.size();
    EventPool pool;
    std::vector<hipEvent_t> ev(nb, nullptr);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (h_src) {
        if (!ctx->copy_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        // the copies may not start before earlier work on the compute stream (a previous proof still reading sec_n) is done
        if (pool.mark(ctx->stream, ctx->copy_stream)) return -1;
        if (upload_ms) {
            if (pool.make(&t0, true) || pool.make(&t1, true)) return -1;
            XLP_HIP(hipEventRecord(t0, ctx->copy_stream));
        }
        for (u64 b = 0; b < nb; b++) {
            const u64 c0 = starts[b], wc = widths[b];
            XLP_HIP(hipMemcpy2DAsync(sec_n + c0, ncols * sizeof(u64), h_src + c0, ncols * sizeof(u64), wc * sizeof(u64), n, hipMemcpyHostToDevice, ctx->copy_stream));
            if (pool.mark(ctx->copy_stream, nullptr, &ev[b])) return -1;
        }
        if (upload_ms) XLP_HIP(hipEventRecord(t1, ctx->copy_stream));
    }
    for (u64 b = 0; b < nb; b++) {
        const u64 c0 = starts[b], wc = widths[b];
        if (h_src) XLP_HIP(hipStreamWaitEvent(ctx->stream, ev[b], 0));
        if (T.begin(T.lde, ctx->stream) || dev_extend_cols(ctx, sec_2ns, sec_n, n_ext, n, ncols, c0, wc) || T.end(T.lde, ctx->stream)) return -1;
        if (T.begin(T.merkle, ctx->stream) || dev_linear_hash_absorb(ctx, nodes, sec_2ns, ncols, c0, wc, n_ext, b == 0, b == nb - 1)) return -1;
        if (b == nb - 1 && dev_merkle_levels(ctx, nodes, n_ext)) return -1;
        if (T.end(T.merkle, ctx->stream)) return -1;
    }
    if (t0) {
        float ms = 0.f;
        XLP_HIP(hipEventSynchronize(t1));
        XLP_HIP(hipEventElapsedTime(&ms, t0, t1));
        *upload_ms = ms;
    }
    return 0;
}
// The same pipeline on column-major sections: sec_n = ncols columns of n rows (csn words apart), sec_2ns = ncols columns of n_ext rows,
// coset-major (cs2 apart), bands of whole columns.  With h_src (the executor's row-major witness in host memory, prover.cpp:435) every band
// is uploaded as it is -- a dense row-major [n][band] matrix -- into `stage` (device scratch of >= n * ncols words: in step 1 the dead
// base-domain sections behind cm1_n) on the copy stream and transposed into its columns on the compute stream ahead of its extension.
static int commit_bands_cm(uint64_t* sec_n, uint64_t csn, uint64_t* sec_2ns, uint64_t cs2, uint64_t* nodes, const uint64_t* h_src, uint64_t* stage, uint64_t stage_words,
                           uint64_t n_ext, uint64_t n, uint64_t ncols, double* upload_ms, int clear_stage = 0)
{
    CommitTimes& T = g_commit_times;
    T.clear();
    if (upload_ms) *upload_ms = 0.0;
    if (n == 0 || n_ext == 0) return 0;
    if ((n & (n - 1)) || (n_ext & (n_ext - 1)) || n_ext < n) return xlp_fail("commit (column-major): sizes must be powers of two, n_ext >= n");
    if (csn < n || cs2 < n_ext) return xlp_fail("commit (column-major): column stride smaller than a column");
    if (h_src && ncols && stage_words < n * ncols) return xlp_fail("commit (column-major): staging area smaller than the section");
    u32 e = 0;
    while ((n << e) < n_ext) e++;
    Ctx* ctx = &g_ctx;
    // band plan: as commit_bands (wide first, narrowing by 15 %, for a host source; uniform 64-column bands otherwise)
    u64 band = 64;
    if (const char* ev = getenv("XLP_COMMIT_BAND")) { const u64 v = strtoull(ev, nullptr, 10); if (v >= 8 && v % 8 == 0) band = v; }
    std::vector<u64> starts, widths;
    if (ncols <= 4 || ncols <= band) { starts.push_back(0); widths.push_back(ncols); }
    else if (h_src && !getenv("XLP_COMMIT_BAND") && ncols >= 320) {
        // The row-major plan (wide first, narrowing by 15 %) stays: plans with a small first band, or with uniform wide bands, measured within
        // +-1 % of it and none better (profiles/r4/commit_band_plan_ab.txt: step 1 1.069-1.110 s over eight plans).  XLP_COMMIT_PLAN=1: small first band;
        // XLP_COMMIT_PLAN_WIDTHS: an explicit plan.
        const bool small_first = getenv("XLP_COMMIT_PLAN") && atoi(getenv("XLP_COMMIT_PLAN")) == 1;
        std::vector<u64> rev;
        double w = small_first ? 32.0 : 48.0;
        const u64 head = small_first ? 64 : 0;
        for (u64 rem = ncols / 8 * 8 - head; rem;) {
            u64 ww = ((u64)(w + 4.0)) / 8 * 8;
            if (ww > rem) ww = rem;
            rev.push_back(ww); rem -= ww;
            w = w / 0.85 < 160.0 ? w / 0.85 : 160.0;
        }
        rev[0] += ncols % 8;
        if (rev.size() > 1 && rev.back() < 64) { rev[rev.size() - 2] += rev.back(); rev.pop_back(); }
        if (head) rev.push_back(head);
        if (const char* ev = getenv("XLP_COMMIT_PLAN_WIDTHS")) {          // experiments: an explicit plan "w0,w1,..." (multiples of 8 but the last, summing to ncols)
            std::vector<u64> ws;
            u64 sum = 0; bool ok = true;
            for (const char* q = ev; *q;) { char* e2; const u64 v = strtoull(q, &e2, 10); if (e2 == q || !v) { ok = false; break; } ws.push_back(v); sum += v; q = *e2 == ',' ? e2 + 1 : e2; }
            for (size_t k = 0; k + 1 < ws.size(); k++) if (ws[k] % 8) ok = false;
            if (ok && sum == ncols) { rev.assign(ws.rbegin(), ws.rend()); }
        }
        u64 c0 = 0;
        for (size_t k = rev.size(); k-- > 0;) { starts.push_back(c0); widths.push_back(rev[k]); c0 += rev[k]; }
    } else {
        for (u64 c0 = 0; c0 < ncols; c0 += band) { starts.push_back(c0); widths.push_back(ncols - c0 < band ? ncols - c0 : band); }
    }
    const u64 nb = starts.size();
    EventPool pool;
    std::vector<hipEvent_t> ev(nb, nullptr);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (h_src && ncols) {
        if (!ctx->copy_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        if (pool.mark(ctx->stream, ctx->copy_stream)) return -1;       // the staging area may still be read by earlier work
        if (upload_ms) {
            if (pool.make(&t0, true) || pool.make(&t1, true)) return -1;
            XLP_HIP(hipEventRecord(t0, ctx->copy_stream));
        }
        for (u64 b = 0; b < nb; b++) {                                  // band b lands dense at stage + n * starts[b]: no area is reused
            const u64 c0 = starts[b], wc = widths[b];
            XLP_HIP(hipMemcpy2DAsync(stage + n * c0, wc * sizeof(u64), h_src + c0, ncols * sizeof(u64), wc * sizeof(u64), n, hipMemcpyHostToDevice, ctx->copy_stream));
            if (pool.mark(ctx->copy_stream, nullptr, &ev[b])) return -1;
        }
        if (upload_ms) XLP_HIP(hipEventRecord(t1, ctx->copy_stream));
    }
    for (u64 b = 0; b < nb && ncols; b++) {
        const u64 c0 = starts[b], wc = widths[b];
        if (h_src) {
            XLP_HIP(hipStreamWaitEvent(ctx->stream, ev[b], 0));
            if (dev_rows_to_cm(ctx, sec_n + c0 * csn, csn, 0, stage + n * c0, wc, 0, wc, n, 0)) return -1;
            // the staging area is somebody's section again afterwards: hand it back zeroed, band by band (the compute stream waits for PCIe anyway)
            if (clear_stage) XLP_HIP(hipMemsetAsync(stage + n * c0, 0, n * wc * sizeof(u64), ctx->stream));
        }
        if (T.begin(T.lde, ctx->stream) || dev_extend_cm(ctx, sec_2ns + c0 * cs2, cs2, 0, sec_n + c0 * csn, csn, n_ext, n, wc) || T.end(T.lde, ctx->stream)) return -1;
        if (ncols > 4) {
            if (T.begin(T.merkle, ctx->stream) || dev_linear_hash_absorb_cm(ctx, nodes, sec_2ns, cs2, 0, c0, wc, n_ext, e, b == 0, b == nb - 1)) return -1;
            if (b == nb - 1 && dev_merkle_levels(ctx, nodes, n_ext)) return -1;
            if (T.end(T.merkle, ctx->stream)) return -1;
        }
    }
    if (ncols <= 4) {                                                   // the no-hash rule of narrow rows, and the empty section
        if (T.begin(T.merkle, ctx->stream) || dev_merkle_cm(ctx, nodes, sec_2ns, cs2, 0, ncols, n_ext, e) || T.end(T.merkle, ctx->stream)) return -1;
    }
    if (t0) {
        float ms = 0.f;
        XLP_HIP(hipEventSynchronize(t1));
        XLP_HIP(hipEventElapsedTime(&ms, t0, t1));
        *upload_ms = ms;
    }
    return 0;
}
int xlp_dev_commit_cm(uint64_t* sec_2ns, uint64_t cs2, uint64_t* nodes, const uint64_t* sec_n, uint64_t csn, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return commit_bands_cm(const_cast<uint64_t*>(sec_n), csn, sec_2ns, cs2, nodes, nullptr, nullptr, 0, n_ext, n, ncols, nullptr);
}
int xlp_dev_commit_from_host_cm(uint64_t* sec_n, uint64_t csn, uint64_t* sec_2ns, uint64_t cs2, uint64_t* nodes, const uint64_t* h_src, uint64_t* stage, uint64_t stage_words,
                                int clear_stage, uint64_t n_ext, uint64_t n, uint64_t ncols, double* upload_ms)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    if (ncols && n && (!h_src || !stage)) return xlp_fail("xlp_dev_commit_from_host_cm: null host pointer or staging area");
    return commit_bands_cm(sec_n, csn, sec_2ns, cs2, nodes, h_src, stage, stage_words, n_ext, n, ncols, upload_ms, clear_stage);
}
int xlp_dev_commit_last_times(double* lde_ms, double* merkle_ms)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    double sum[2] = {0.0, 0.0};
    int k = 0;
    for (auto* v : {&g_commit_times.lde, &g_commit_times.merkle}) {
        for (auto& pr : *v) {
            float ms = 0.f;
            XLP_HIP(hipEventSynchronize(pr.second));
            XLP_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            sum[k] += ms;
        }
        k++;
    }
    if (lde_ms) *lde_ms = sum[0];
    if (merkle_ms) *merkle_ms = sum[1];
    return 0;
}
int xlp_dev_commit_from_host(uint64_t* sec_n, uint64_t* sec_2ns, uint64_t* nodes, const uint64_t* h_src, uint64_t n_ext, uint64_t n, uint64_t ncols, double* upload_ms)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (ncols && n && !h_src) return xlp_fail("xlp_dev_commit_from_host: null host pointer");
    return commit_bands(sec_n, sec_2ns, nodes, h_src, n_ext, n, ncols, upload_ms);
}
// ---- witness upload in column bands beside the compute stream, for callers that interleave it with their own stages (the row-sharded
// commit of host/starks.hpp: band k + 1 crosses PCIe while band k is exchanged and extended)
int xlp_dev_upload_begin(void)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    Ctx* ctx = &g_ctx;
    if (!ctx->copy_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    EventPool pool;
    return pool.mark(ctx->stream, ctx->copy_stream);          // the copies may not overtake earlier work on the compute stream (it may still read the target)
}
int xlp_dev_upload_cols_async(uint64_t* dst, const uint64_t* src_host, uint64_t ld, uint64_t col0, uint64_t ncols, uint64_t rows)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!rows || !ncols) return 0;
    if (!dst || !src_host) return xlp_fail("xlp_dev_upload_cols_async: null pointer");
    if (col0 + ncols > ld) return xlp_fail("xlp_dev_upload_cols_async: columns exceed the row stride");
    Ctx* ctx = &g_ctx;
    if (!ctx->copy_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    XLP_HIP(hipMemcpy2DAsync(dst + col0, ld * sizeof(u64), src_host + col0, ld * sizeof(u64), ncols * sizeof(u64), rows, hipMemcpyHostToDevice, ctx->copy_stream));
    return 0;
}
int xlp_dev_upload_fence(void)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    Ctx* ctx = &g_ctx;
    if (!ctx->copy_stream) return 0;
    EventPool pool;
    return pool.mark(ctx->copy_stream, ctx->stream);          // later work on the compute stream sees every upload enqueued so far
}
int xlp_dev_commit(uint64_t* sec_2ns, uint64_t* nodes, const uint64_t* sec_n, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    XLP_ENTER_PLAN;
    if (ensure_init()) return -1;
    return commit_bands(const_cast<uint64_t*>(sec_n), sec_2ns, nodes, nullptr, n_ext, n, ncols, nullptr);
}

// --------------------------------------------------------------- host layer

int xlp_ntt(uint64_t* dst, const uint64_t* src, uint64_t n, uint64_t ncols, int inverse)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0 || ncols == 0) return 0;
    if (!dst || !src) return xlp_fail("xlp_ntt: null pointer");
    const size_t bytes = n * ncols * sizeof(u64);
    DevBuf a;
    if (a.alloc(bytes)) return -1;
    H2D(a, src, bytes);
    if (dev_ntt(&g_ctx, a.u(), a.u(), n, ncols, inverse)) return -1;
    D2H(dst, a, bytes);
    return 0;
}

// ---- device copies of host buffers (opt-in: xlp_cache_enable) -------------------------------------------------------------
// The drop-in call sites hand over HOST pointers into the prover's slab, and consecutive calls name the same sections:
// extendPol writes p_cm1_2ns, the next call (merkelize) reads it (starks.cpp:53-56).  Staging every call separately moves the
// 100.8 GB extended section over PCIe twice more than needed.  With the cache enabled the device buffer a call produced (or
// uploaded) stays registered under its host address range; a later call that READS exactly that range takes the device copy,
// a call that WRITES into a range drops every copy overlapping it.  The caller's side of the contract: between two xlp_* calls
// the host does not modify a range it passed to the library -- or calls xlp_cache_invalidate(ptr) if it did.

int xlp_cache_enable(size_t max_bytes)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    g_cache_cap = max_bytes;
    while (g_cache_used > g_cache_cap && !g_cache.empty()) cache_drop(0);
    return 0;
}
int xlp_cache_invalidate(const void* host_ptr, size_t bytes)
{
    XLP_ENTER;
    if (!g_ctx.inited) return 0;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    if (!host_ptr) { while (!g_cache.empty()) cache_drop(0); return 0; }
    cache_drop_overlapping(host_ptr, bytes ? bytes : 1);
    return 0;
}
size_t xlp_cache_bytes(void) { XLP_ENTER; return g_cache_used; }

// extendPol on host buffers, as a three-stream pipeline over 64-column bands: band b+1 is uploaded (copy stream) while band b is
// extended (compute stream) and band b-1 of the result goes back to the host (download stream; PCIe is full duplex).
int xlp_extend(uint64_t* out, const uint64_t* in, uint64_t n_ext, uint64_t n, uint64_t ncols, uint64_t* scratch)
{
    XLP_ENTER;
    (void)scratch;
    if (ensure_init()) return -1;
    if (n == 0 || ncols == 0) return 0;
    if (!out || !in) return xlp_fail("xlp_extend: null pointer");
    if ((n & (n - 1)) || (n_ext & (n_ext - 1)) || n_ext < n) return xlp_fail("extendPol: sizes must be powers of two, n_ext >= n");
    Ctx* ctx = &g_ctx;
    const size_t in_bytes = n * ncols * sizeof(u64), out_bytes = n_ext * ncols * sizeof(u64);
    // the buffers belong to this call until it succeeds (then the cache takes them, or frees them when it is off); on any failure they
    // are freed -- a cached input copy included: it may have been half overwritten
    DevOwner d_in, d_out;
    d_in.p = cache_take(in, in_bytes);
    const bool in_cached = d_in.p != nullptr;
    cache_drop_overlapping(out, out_bytes);
    if (!d_in.p) XLP_ALLOC(plan_malloc((void**)&d_in.p, in_bytes));
    if (hipMalloc(&d_out.p, out_bytes) != hipSuccess) return xlp_fail("xlp_extend: out of device memory");
    const u64 band = 64;
    if (ncols <= band || n_ext * ncols < (1ull << 22)) {       // small: no pipeline
        if (!in_cached) XLP_HIP(hipMemcpyAsync(d_in.p, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        if (dev_extend(ctx, (u64*)d_out.p, (const u64*)d_in.p, n_ext, n, ncols, nullptr)) return -1;
        XLP_HIP(hipMemcpyAsync(out, d_out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        XLP_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        if (!ctx->copy_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        if (!ctx->down_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
        XLP_HIP(hipStreamSynchronize(ctx->stream));
        const u64 nb = (ncols + band - 1) / band;
        EventPool pool;
        std::vector<hipEvent_t> up(nb, nullptr);
        int rc = 0;
        for (u64 b = 0; b < nb && !rc && !in_cached; b++) {
            const u64 c0 = b * band, wc = ncols - c0 < band ? ncols - c0 : band;
            if (hipMemcpy2DAsync((u64*)d_in.p + c0, ncols * 8, in + c0, ncols * 8, wc * 8, n, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess) rc = xlp_fail("xlp_extend: upload failed");
            if (!rc) rc = pool.mark(ctx->copy_stream, nullptr, &up[b]);
        }
        for (u64 b = 0; b < nb && !rc; b++) {
            const u64 c0 = b * band, wc = ncols - c0 < band ? ncols - c0 : band;
            if (up[b] && hipStreamWaitEvent(ctx->stream, up[b], 0) != hipSuccess) rc = xlp_fail("xlp_extend: stream wait failed");
            if (!rc) rc = dev_extend_cols(ctx, (u64*)d_out.p, (const u64*)d_in.p, n_ext, n, ncols, c0, wc);
            if (!rc) rc = pool.mark(ctx->stream, ctx->down_stream);          // the band's download starts when its kernels are done
            if (!rc && hipMemcpy2DAsync(out + c0, ncols * 8, (u64*)d_out.p + c0, ncols * 8, wc * 8, n_ext, hipMemcpyDeviceToHost, ctx->down_stream) != hipSuccess) rc = xlp_fail("xlp_extend: download failed");
        }
        // nothing of this call may still be in flight when the buffers are released, failed or not
        XLP_HIPV(hipStreamSynchronize(ctx->copy_stream));
        XLP_HIPV(hipStreamSynchronize(ctx->stream));
        if (hipStreamSynchronize(ctx->down_stream) != hipSuccess && !rc) rc = xlp_fail("xlp_extend: download failed");
        if (rc) return rc;
    }
    cache_put(in, in_bytes, d_in.release());                   // (frees them when the cache is off)
    cache_put(out, out_bytes, d_out.release());
    return 0;
}

int xlp_merkle(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (h == 0) return 0;
    if (!nodes || (w && !src)) return xlp_fail("xlp_merkle: null pointer");
    if (h & (h - 1)) return xlp_fail("merkletree: number of rows is not a power of two");
    Ctx* ctx = &g_ctx;
    const size_t src_bytes = w * h * sizeof(u64), node_bytes = (2 * h - 1) * 4 * sizeof(u64);
    DevOwner d_src;                                            // this call's until it succeeds; then the cache's (or freed)
    d_src.p = w ? cache_take(src, src_bytes) : nullptr;
    const bool cached = d_src.p != nullptr;
    cache_drop_overlapping(nodes, node_bytes);
    DevBuf b;
    if (b.alloc(node_bytes)) { if (d_src.p) cache_put(src, src_bytes, d_src.release()); return -1; }
    int rc = 0;
    if (!cached && w) {
        XLP_ALLOC(plan_malloc((void**)&d_src.p, src_bytes));
        // rows are hashed independently: upload row chunks on the copy stream while the previous chunk is hashed
        const u64 chunk_rows = w > 4 && src_bytes > (256ull << 20) ? std::max<u64>(1, (128ull << 20) / (w * 8)) : h;
        if (chunk_rows >= h) {
            XLP_HIP(hipMemcpyAsync(d_src.p, src, src_bytes, hipMemcpyHostToDevice, ctx->stream));
            rc = dev_linear_hash(ctx, b.u(), (const u64*)d_src.p, w, h);
        } else {
            if (!ctx->copy_stream) XLP_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            XLP_HIP(hipStreamSynchronize(ctx->stream));
            EventPool pool;
            std::vector<hipEvent_t> ev;
            for (u64 r0 = 0; r0 < h && !rc; r0 += chunk_rows) {
                const u64 nr = h - r0 < chunk_rows ? h - r0 : chunk_rows;
                if (hipMemcpyAsync((u64*)d_src.p + r0 * w, src + r0 * w, nr * w * 8, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess) rc = xlp_fail("xlp_merkle: upload failed");
                hipEvent_t e = nullptr;
                if (!rc) rc = pool.mark(ctx->copy_stream, nullptr, &e);
                ev.push_back(e);
            }
            u64 k = 0;
            for (u64 r0 = 0; r0 < h && !rc; r0 += chunk_rows, k++) {
                const u64 nr = h - r0 < chunk_rows ? h - r0 : chunk_rows;
                if (hipStreamWaitEvent(ctx->stream, ev[k], 0) != hipSuccess) rc = xlp_fail("xlp_merkle: stream wait failed");
                if (!rc) rc = dev_linear_hash(ctx, b.u() + 4 * r0, (const u64*)d_src.p + r0 * w, w, nr);
            }
            XLP_HIPV(hipStreamSynchronize(ctx->copy_stream));      // no copy into d_src may outlive this call
        }
    } else {
        rc = dev_linear_hash(ctx, b.u(), (const u64*)d_src.p, w, h);
    }
    if (!rc) rc = dev_merkle_levels(ctx, b.u(), h);
    if (!rc && hipMemcpyAsync(nodes, b.p, node_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = xlp_fail("xlp_merkle: download failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = xlp_fail("xlp_merkle: synchronise failed");
    if (d_src.p && !rc) cache_put(src, src_bytes, d_src.release());
    return rc;
}

int xlp_perm_batch(uint64_t* out, const uint64_t* in, uint64_t count)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (count == 0) return 0;
    if (!out || !in) return xlp_fail("xlp_perm: null pointer");
    if (count <= 64) {                                        // the transcript's one-at-a-time calls: a staging buffer that stays allocated
        u64* d = nullptr;
        if (ctx_scratch(&g_ctx, 2 * 64 * 12 * sizeof(u64), 6, &d)) return -1;
        XLP_HIP(hipMemcpyAsync(d, in, count * 12 * sizeof(u64), hipMemcpyHostToDevice, g_ctx.stream));
        if (dev_perm(&g_ctx, d + 64 * 12, d, count)) return -1;
        cache_drop_overlapping(out, count * 12 * sizeof(u64));
        XLP_HIP(hipMemcpyAsync(out, d + 64 * 12, count * 12 * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
        XLP_HIP(hipStreamSynchronize(g_ctx.stream));
        return 0;
    }
    DevBuf a;
    if (a.alloc(count * 12 * sizeof(u64))) return -1;
    H2D(a, in, count * 12 * sizeof(u64));
    if (dev_perm(&g_ctx, a.u(), a.u(), count)) return -1;
    D2H(out, a, count * 12 * sizeof(u64));
    return 0;
}
int xlp_perm(uint64_t out[12], const uint64_t in[12]) { return xlp_perm_batch(out, in, 1); }
int xlp_perm_chain(uint64_t out[12], const uint64_t state[4], const uint64_t* elems, uint64_t nblocks)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!out || !state || (nblocks && !elems)) return xlp_fail("xlp_perm_chain: null pointer");
    if (nblocks == 0) return xlp_fail("xlp_perm_chain: no block");
    u64* d = nullptr;                                         // [out 12 | state 4 | elems 8 * nblocks]
    if (ctx_scratch(&g_ctx, std::max<size_t>((16 + 8 * nblocks) * sizeof(u64), 2 * 64 * 12 * sizeof(u64)), 6, &d)) return -1;   // the transcript's staging slot (xlp_perm_batch)
    XLP_HIP(hipMemcpyAsync(d + 12, state, 4 * sizeof(u64), hipMemcpyHostToDevice, g_ctx.stream));
    XLP_HIP(hipMemcpyAsync(d + 16, elems, 8 * nblocks * sizeof(u64), hipMemcpyHostToDevice, g_ctx.stream));
    if (dev_perm_chain(&g_ctx, d, d + 12, d + 16, nblocks)) return -1;
    cache_drop_overlapping(out, 12 * sizeof(u64));
    XLP_HIP(hipMemcpyAsync(out, d, 12 * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    return 0;
}

// ---- transport between the ranks of a row-sharded proof (comm.hip)
int xlp_comm_init(int rank, int world, const char* backend, const char* rendezvous)
{
    XLP_ENTER_PLAN;
    // RCCL binds the communicator to the context's device.  The shared-memory backend needs no device for its rendezvous, barrier and host-word
    // all-gather (tests/test_abi.py runs those as two processes on a CPU-only box); its device transfers fail like every other GPU call there.
    if (backend && !strcmp(backend, "rccl") && ensure_init()) return -1;
    return comm_init(&g_ctx, rank, world, backend, rendezvous);
}
void xlp_comm_shutdown(void) { XLP_ENTER_PLAN; comm_shutdown(); }
int xlp_comm_rank(void) { return comm_rank(); }
int xlp_comm_world(void) { return comm_world(); }
int xlp_comm_ranks_seen(void) { return comm_ranks_seen(); }
void xlp_comm_abort(const char* why) { comm_abort(why); }    // (no lock: the caller may be on its way out of a failed call of another thread)
int xlp_comm_exchange(uint64_t* const* recv, const uint64_t* recv_words, const uint64_t* const* send, const uint64_t* send_words, uint64_t all_max_words)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!recv || !recv_words || !send || !send_words) return xlp_fail("xlp_comm_exchange: null pointer");
    return comm_exchange(&g_ctx, recv, recv_words, send, send_words, all_max_words);
}
int xlp_comm_all_gather(uint64_t* out, const uint64_t* in, uint64_t words)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (words && (!out || !in)) return xlp_fail("xlp_comm_all_gather: null pointer");
    return comm_all_gather(&g_ctx, out, in, words);
}
int xlp_comm_sendrecv(uint64_t* recv, int from, const uint64_t* send, int to, uint64_t words)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (words && (!recv || !send)) return xlp_fail("xlp_comm_sendrecv: null pointer");
    return comm_sendrecv(&g_ctx, recv, from, send, to, words);
}
int xlp_comm_all_gather_host(uint64_t* out, const uint64_t* in, uint64_t words)
{
    XLP_ENTER;
    if (words && (!out || !in)) return xlp_fail("xlp_comm_all_gather_host: null pointer");
    return comm_all_gather_host(&g_ctx, out, in, words);
}
int xlp_comm_barrier(void)
{
    XLP_ENTER;
    return comm_barrier(&g_ctx);
}
int xlp_comm_stats(double* seconds, uint64_t* bytes_sent, uint64_t* calls, int mode) { XLP_ENTER_PLAN; return comm_stats(seconds, bytes_sent, calls, mode); }

// ---- BN128 (StarkRecursiveF)
int xlp_bn128_init(const char* constants_path)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    return bn128_init(&g_ctx, constants_path);
}
uint64_t xlp_merkle_bn128_num_nodes(uint64_t height) { return bn128_num_nodes(height); }
int xlp_dev_merkle_bn128(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (h && (!nodes || (!src && w))) return xlp_fail("xlp_dev_merkle_bn128: null pointer");
    return dev_merkle_bn128(&g_ctx, nodes, src, w, h);
}
int xlp_merkle_bn128(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (h == 0) return 0;
    if (!nodes || (!src && w)) return xlp_fail("xlp_merkle_bn128: null pointer");
    DevBuf a, b;
    const size_t nb = bn128_num_nodes(h) * 4 * sizeof(u64);
    if (a.alloc((h * w ? h * w : 1) * sizeof(u64)) || b.alloc(nb)) return -1;
    H2D(a, src, h * w * sizeof(u64));
    if (dev_merkle_bn128(&g_ctx, b.u(), a.u(), w, h)) return -1;
    D2H(nodes, b, nb);
    return 0;
}
int xlp_dev_poseidon_bn128(uint64_t* states, uint32_t t, uint64_t count)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (count && !states) return xlp_fail("xlp_dev_poseidon_bn128: null pointer");
    return dev_poseidon_bn128(&g_ctx, states, t, count);
}
int xlp_poseidon_bn128(uint64_t* states, uint32_t t, uint64_t count)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (count == 0) return 0;
    if (!states) return xlp_fail("xlp_poseidon_bn128: null pointer");
    DevBuf a;
    const size_t nb = (size_t)count * t * 4 * sizeof(u64);
    if (a.alloc(nb)) return -1;
    H2D(a, states, nb);
    if (dev_poseidon_bn128(&g_ctx, a.u(), t, count)) return -1;
    D2H(states, a, nb);
    return 0;
}

int xlp_dev_merkle_bn128_proofs(uint64_t* out, const uint64_t* src, const uint64_t* nodes, uint64_t w, uint64_t h, const uint64_t* idxs, uint64_t nq)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (nq == 0) return 0;
    if (!out || !nodes || !idxs || (w && !src)) return xlp_fail("xlp_dev_merkle_bn128_proofs: null pointer");
    for (uint64_t q = 0; q < nq; q++) if (idxs[q] >= h) return xlp_fail("xlp_dev_merkle_bn128_proofs: index out of range");
    uint32_t levels = 0;
    for (u64 c = 1; c < h; c *= 16) levels++;
    const size_t per = w + 64ull * levels;
    DevBuf di, dout;
    if (di.alloc(nq * sizeof(u64)) || dout.alloc(nq * per * sizeof(u64))) return -1;
    H2D(di, idxs, nq * sizeof(u64));
    if (dev_merkle_bn128_proofs(&g_ctx, dout.u(), src, nodes, w, h, di.u(), nq)) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    D2H(out, dout, nq * per * sizeof(u64));
    return 0;
}
int xlp_dev_fft_bn254(uint64_t* data, uint64_t n, int inverse)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n && !data) return xlp_fail("xlp_dev_fft_bn254: null pointer");
    return dev_fft_bn254(&g_ctx, data, n, inverse);
}
int xlp_fft_bn254(uint64_t* data, uint64_t n, int inverse)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0) return 0;
    if (!data) return xlp_fail("xlp_fft_bn254: null pointer");
    DevBuf a;
    if (a.alloc(n * 4 * sizeof(u64))) return -1;
    H2D(a, data, n * 4 * sizeof(u64));
    if (dev_fft_bn254(&g_ctx, a.u(), n, inverse)) return -1;
    D2H(data, a, n * 4 * sizeof(u64));
    return 0;
}

int xlp_dev_msm_bn254(uint64_t* out, const uint64_t* bases, const uint64_t* scalars, uint64_t n)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!out || (n && (!bases || !scalars))) return xlp_fail("xlp_dev_msm_bn254: null pointer");
    return dev_msm_bn254(&g_ctx, out, bases, scalars, n);
}
int xlp_dev_msm_resident(uint64_t* out, const uint64_t* point, uint32_t lanes, uint32_t iters)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (lanes && (!out || !point)) return xlp_fail("xlp_dev_msm_resident: null pointer");
    return dev_msm_resident(&g_ctx, out, point, lanes, iters);
}
int xlp_msm_bn254_window_bits(uint64_t n) { return (int)msm_bn254_window_bits(n); }
int xlp_msm_bn254(uint64_t* out, const uint64_t* bases, const uint64_t* scalars, uint64_t n)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!out || (n && (!bases || !scalars))) return xlp_fail("xlp_msm_bn254: null pointer");
    DevBuf b, s, o;
    if (b.alloc((n ? n : 1) * 8 * sizeof(u64)) || s.alloc((n ? n : 1) * 4 * sizeof(u64)) || o.alloc(8 * sizeof(u64))) return -1;
    if (n) { H2D(b, bases, n * 8 * sizeof(u64)); H2D(s, scalars, n * 4 * sizeof(u64)); }
    if (dev_msm_bn254(&g_ctx, o.u(), b.u(), s.u(), n)) return -1;
    D2H(out, o, 8 * sizeof(u64));
    return 0;
}

// ---- vectors of BN254 Fr elements (fflonk.hip): device pointers, asynchronous on the current stream unless a HOST result is returned
#define XLP_FR_CALL(name, check, call)                         \
    XLP_ENTER;                                                 \
    if (ensure_init()) return -1;                              \
    if (check) return xlp_fail(name ": null pointer");         \
    return call;
int xlp_dev_fr_gather_mont(uint64_t* out, const uint32_t* map, uint64_t nmap, const uint64_t* values, uint64_t nvalues)
{ XLP_FR_CALL("xlp_dev_fr_gather_mont", nmap && (!out || !map || (nvalues && !values)), dev_fr_gather_mont(&g_ctx, out, map, nmap, values, nvalues)) }
int xlp_dev_fr_from_mont(uint64_t* out, const uint64_t* in, uint64_t n)
{ XLP_FR_CALL("xlp_dev_fr_from_mont", n && (!out || !in), dev_fr_from_mont(&g_ctx, out, in, n)) }
int xlp_dev_fr_mul(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n)
{ XLP_FR_CALL("xlp_dev_fr_mul", n && (!out || !a || !b), dev_fr_mul(&g_ctx, out, a, b, n)) }
int xlp_dev_fr_lincomb(uint64_t* out, uint64_t len, uint32_t nterms, const uint64_t* const* pols, const uint64_t* lens, const uint64_t* coefs)
{ XLP_FR_CALL("xlp_dev_fr_lincomb", len && (!out || (nterms && (!pols || !lens || !coefs))), dev_fr_lincomb(&g_ctx, out, len, nterms, pols, lens, coefs)) }
int xlp_dev_fr_interleave(uint64_t* out, uint64_t out_len, uint32_t k, const uint64_t* const* pols, const uint64_t* lens)
{ XLP_FR_CALL("xlp_dev_fr_interleave", out_len && (!out || !pols || !lens), dev_fr_interleave(&g_ctx, out, out_len, k, pols, lens)) }
int xlp_dev_fr_div_zh(uint64_t* pol, uint64_t len, uint64_t n)
{ XLP_FR_CALL("xlp_dev_fr_div_zh", len && !pol, dev_fr_div_zh(&g_ctx, pol, len, n)) }
int xlp_dev_fr_batch_inverse(uint64_t* out, const uint64_t* in, uint64_t n)
{ XLP_FR_CALL("xlp_dev_fr_batch_inverse", n && (!out || !in), dev_fr_batch_inverse(&g_ctx, out, in, n)) }
int xlp_dev_fr_prefix_product(uint64_t* out, const uint64_t* in, uint64_t n)
{ XLP_FR_CALL("xlp_dev_fr_prefix_product", n && (!out || !in), dev_fr_prefix_product(&g_ctx, out, in, n)) }
int xlp_dev_fr_div_xm_minus(uint64_t* out, const uint64_t* in, uint64_t len, uint32_t m, const uint64_t* beta)
{ XLP_FR_CALL("xlp_dev_fr_div_xm_minus", len && (!out || !in || !beta), dev_fr_div_xm_minus(&g_ctx, out, in, len, m, beta)) }
int xlp_dev_fr_eval(uint64_t* values, const uint64_t* pol, uint64_t len, const uint64_t* points, uint32_t npoints)
{ XLP_FR_CALL("xlp_dev_fr_eval", !values || !points || (len && !pol), dev_fr_eval(&g_ctx, values, pol, len, points, npoints)) }
int xlp_dev_fflonk_t0(uint64_t* out, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* ql, const uint64_t* qr, const uint64_t* qm, const uint64_t* qo,
                      const uint64_t* qc, const uint64_t* lagrange, const uint64_t* pub, uint32_t npub, uint64_t n4)
{ XLP_FR_CALL("xlp_dev_fflonk_t0", !out || !a || !b || !c || !ql || !qr || !qm || !qo || !qc || (npub && (!lagrange || !pub)), dev_fflonk_t0(&g_ctx, out, a, b, c, ql, qr, qm, qo, qc, lagrange, pub, npub, n4)) }
int xlp_dev_fflonk_z_numden(uint64_t* num, uint64_t* den, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* s1, const uint64_t* s2, const uint64_t* s3, uint64_t n,
                            const uint64_t* scalars, const uint64_t* wpow, uint32_t nw)
{ XLP_FR_CALL("xlp_dev_fflonk_z_numden", !num || !den || !a || !b || !c || !s1 || !s2 || !s3 || !scalars || !wpow || nw > 32, dev_fflonk_z_numden(&g_ctx, num, den, a, b, c, s1, s2, s3, n, scalars, wpow, nw)) }
int xlp_dev_fflonk_z_rotate(uint64_t* z, const uint64_t* run, uint64_t n)
{ XLP_FR_CALL("xlp_dev_fflonk_z_rotate", !z || !run, dev_fflonk_z_rotate(&g_ctx, z, run, n)) }
int xlp_dev_fflonk_t1(uint64_t* t1, uint64_t* t1z, const uint64_t* z, const uint64_t* lagrange1, uint64_t n2, const uint64_t* blinding, const uint64_t* wpow, uint32_t nw)
{ XLP_FR_CALL("xlp_dev_fflonk_t1", !t1 || !t1z || !z || !lagrange1 || !blinding || !wpow || nw > 32, dev_fflonk_t1(&g_ctx, t1, t1z, z, lagrange1, n2, blinding, wpow, nw)) }
int xlp_dev_fflonk_t2(uint64_t* t2, uint64_t* t2z, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* z, const uint64_t* s1, const uint64_t* s2, const uint64_t* s3,
                      uint64_t n4, const uint64_t* scalars, const uint64_t* wpow, uint32_t nw)
{ XLP_FR_CALL("xlp_dev_fflonk_t2", !t2 || !t2z || !a || !b || !c || !z || !s1 || !s2 || !s3 || !scalars || !wpow || nw > 32, dev_fflonk_t2(&g_ctx, t2, t2z, a, b, c, z, s1, s2, s3, n4, scalars, wpow, nw)) }

int xlp_linear_hash(uint64_t* out, const uint64_t* src, uint64_t w, uint64_t h)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (h == 0) return 0;
    if (!out || (w && !src)) return xlp_fail("xlp_linear_hash: null pointer");
    DevBuf a, b;
    if (a.alloc(w * h * sizeof(u64)) || b.alloc(h * 4 * sizeof(u64))) return -1;
    H2D(a, src, w * h * sizeof(u64));
    if (dev_linear_hash(&g_ctx, b.u(), a.u(), w, h)) return -1;
    D2H(out, b, h * 4 * sizeof(u64));
    return 0;
}

int xlp_fri_fold(uint64_t* pol2, const uint64_t* pol, uint32_t bits, uint32_t red, const uint64_t x[3], uint64_t shift_inv)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (!pol2 || !pol || !x) return xlp_fail("xlp_fri_fold: null pointer");
    if (red > bits) return xlp_fail("fri_fold: reduction exceeds polynomial size");
    if (bits > 34) return xlp_fail("fri_fold: polynomial too large");
    DevBuf a, b;
    const size_t nin = (3ull << bits) * sizeof(u64), nout = (3ull << (bits - red)) * sizeof(u64);
    if (a.alloc(nin) || b.alloc(nout)) return -1;
    H2D(a, pol, nin);
    if (dev_fri_fold(&g_ctx, b.u(), a.u(), bits, red, x, shift_inv)) return -1;
    D2H(pol2, b, nout);
    return 0;
}

int xlp_batch_inverse(uint64_t* res, const uint64_t* src, uint64_t n, uint32_t dim)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0) return 0;
    if (!res || !src) return xlp_fail("xlp_batch_inverse: null pointer");
    if (dim != 1 && dim != 3) return xlp_fail("batchInverse: dim must be 1 or 3");
    DevBuf a;
    const size_t bytes = n * dim * sizeof(u64);
    if (a.alloc(bytes)) return -1;
    H2D(a, src, bytes);
    if (dev_batch_inverse(&g_ctx, a.u(), a.u(), n, dim)) return -1;
    D2H(res, a, bytes);
    return 0;
}
int xlp_calculate_z(uint64_t* z, const uint64_t* num, const uint64_t* den, uint64_t n)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0) return 0;
    if (!z || !num || !den) return xlp_fail("xlp_calculate_z: null pointer");
    DevBuf a, b, c;
    const size_t bytes = n * 3 * sizeof(u64);
    if (a.alloc(bytes) || b.alloc(bytes) || c.alloc(bytes)) return -1;
    H2D(a, num, bytes);
    H2D(b, den, bytes);
    if (dev_calculate_z(&g_ctx, c.u(), 3, a.u(), 3, b.u(), 3, n)) return -1;
    D2H(z, c, bytes);
    return 0;
}
int xlp_calculate_h1h2(uint64_t* h1, uint64_t* h2, const uint64_t* f, const uint64_t* t, uint64_t n, uint32_t dim)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0) return 0;
    if (!h1 || !h2 || !f || !t) return xlp_fail("xlp_calculate_h1h2: null pointer");
    if (dim != 1 && dim != 3) return xlp_fail("calculateH1H2: dim must be 1 or 3");
    DevBuf a, b, c, d;
    const size_t bytes = n * dim * sizeof(u64);
    if (a.alloc(bytes) || b.alloc(bytes) || c.alloc(bytes) || d.alloc(bytes)) return -1;
    H2D(a, f, bytes);
    H2D(b, t, bytes);
    if (dev_calculate_h1h2(&g_ctx, c.u(), dim, d.u(), dim, a.u(), dim, b.u(), dim, n, dim)) return -1;
    XLP_HIP(hipStreamSynchronize(g_ctx.stream));
    D2H(h1, c, bytes);
    D2H(h2, d, bytes);
    return 0;
}

int xlp_field_op(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n, int op)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0) return 0;
    DevBuf da, db, dc;
    if (da.alloc(n * 8) || db.alloc(n * 8) || dc.alloc(n * 8)) return -1;
    H2D(da, a, n * 8);
    H2D(db, b, n * 8);
    if (dev_field_op(&g_ctx, dc.u(), da.u(), db.u(), n, op)) return -1;
    D2H(out, dc, n * 8);
    return 0;
}

int xlp_transpose3(uint64_t* dst, const uint64_t* src, uint64_t n, uint32_t tbits)
{
    XLP_ENTER;
    if (ensure_init()) return -1;
    if (n == 0) return 0;
    if (!dst || !src) return xlp_fail("xlp_transpose3: null pointer");
    DevBuf a, b;
    if (a.alloc(3 * n * sizeof(u64)) || b.alloc(3 * n * sizeof(u64))) return -1;
    H2D(a, src, 3 * n * sizeof(u64));
    if (dev_transpose3(&g_ctx, b.u(), a.u(), n, tbits)) return -1;
    D2H(dst, b, 3 * n * sizeof(u64));
    return 0;
}

}  // extern "C"
