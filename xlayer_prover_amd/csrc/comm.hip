// Transport between the ranks of a row-sharded proof: one process per GPU.  Declared in include/xlp_gl.h (xlp_comm_*).
//
// What it carries (host/starks.hpp, sharded mode): per committed section two all-to-alls (rows -> columns before the per-column LDE,
// columns -> rows before the row-wise leaf sponge: the split of ntt.extendPol + treesGL[i]->merkelize(), starks.cpp:53-60,134-141,
// 215-222), ring halo rows of the step programs' next-row operands, all-gathers of the three-column quotient / FRI polynomials and of the
// plookup tables, and a handful of words per rank (sub-tree roots, grand-product totals, partial evaluations).
//
// Backends:
//   "rccl"  RCCL over xGMI.  librccl.so.1 is loaded on first use (it is ~570 MB: a one-GPU process never pays for it).  The 128-byte
//           ncclUniqueId travels through a file: rank 0 writes <rendezvous>.tmp and renames it to <rendezvous>, the others wait for it.
//           An exchange is a grouped ncclSend / ncclRecv per round of <= 256 MiB per (sender, receiver) message (direct peer writes, every
//           link busy; a zkEVM section at G = 8 is 1.6 GB per message and this image's RCCL was seen to drop the tail of a > 1 GiB message),
//           the rank's own block is a device copy.  Everything is enqueued on the library's stream, so exchanges order with the kernels.
//   "shm"   ranks that SHARE a device or have no peer path (the one-GPU test box; a rehearsal of the N > 1 code path): POSIX shared
//           memory named <rendezvous>, every message staged through the host.  Correct and slow by construction.
//   world == 1 with backend NULL / "self": no transport; exchanges are device copies.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <algorithm>
#include <chrono>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include "xlp_internal.hpp"

namespace xlp {

// ---- the slice of the RCCL API this file uses (rccl/rccl.h:40-43,187,220,260,339,678; ncclSend / ncclRecv / group calls), bound by dlsym
struct NcclId { char internal[128]; };
typedef void* NcclComm;
enum { NCCL_UINT64 = 5 };                        // ncclUint64 of ncclDataType_t (ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};

struct ShmHeader {
    std::atomic<uint32_t> arrived;               // barrier: ranks that reached the current generation
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;              // ranks that mapped the segment
    uint32_t world;
    uint64_t slot_bytes;
};

struct Comm {
    int rank = 0, world = 1;
    enum Kind { NONE, SELF, RCCL, SHM } kind = NONE;
    Rccl rccl;
    NcclComm nccl = nullptr;
    // shm
    std::string shm_name;
    char* shm = nullptr;
    size_t shm_bytes = 0;
    uint64_t slot_bytes = 0;
    bool shm_owner = false;
    char* bounce = nullptr;                      // page-locked staging buffer of one slot: HIP never sees the shared mapping itself (a copy
                                                 // engine writing straight into a MAP_SHARED /dev/shm mapping faulted on MI355X, XNACK off)
    u64* small = nullptr;                        // device staging for the host-word all-gather (rccl)
    size_t small_words = 0;
    // accounting (bench): exchanges only
    double seconds = 0.0;
    u64 bytes_sent = 0, calls = 0;
    bool timing = false;
};
static Comm g_comm;

static const u64 MAX_MESSAGE_WORDS = 1ull << 25;  // 256 MiB per (sender, receiver) message and round

#define XLP_NCCL(call)                                                                                          \
    do {                                                                                                        \
        const int _r = (call);                                                                                  \
        if (_r != 0) return xlp_fail((std::string("RCCL error in " #call ": ") + (g_comm.rccl.GetErrorString ? g_comm.rccl.GetErrorString(_r) : "?")).c_str()); \
    } while (0)

static int rccl_load(Rccl& r)
{
    if (r.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) return xlp_fail((std::string("xlp_comm_init: cannot load librccl.so.1: ") + dlerror()).c_str());
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) ok = false; return p; };
    r.GetUniqueId = (int (*)(NcclId*))sym("ncclGetUniqueId");
    r.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))sym("ncclCommInitRank");
    r.CommDestroy = (int (*)(NcclComm))sym("ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    r.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclRecv");
    r.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))sym("ncclAllGather");
    r.GroupStart = (int (*)())sym("ncclGroupStart");
    r.GroupEnd = (int (*)())sym("ncclGroupEnd");
    if (!ok) return xlp_fail("xlp_comm_init: librccl.so.1 lacks an expected symbol");
    return 0;
}

static bool wait_for(const std::function<bool()>& cond, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (!cond()) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    return true;
}

static ShmHeader* shm_hdr() { return (ShmHeader*)g_comm.shm; }
// device <-> shared-memory slot, through the page-locked bounce buffer
static int shm_bounce()
{
    if (!g_comm.bounce) XLP_HIP(hipHostMalloc((void**)&g_comm.bounce, g_comm.slot_bytes, hipHostMallocDefault));
    return 0;
}
static int shm_put(char* slot, const u64* dev, size_t bytes)
{
    if (shm_bounce()) return -1;
    XLP_HIP(hipMemcpy(g_comm.bounce, dev, bytes, hipMemcpyDeviceToHost));
    memcpy(slot, g_comm.bounce, bytes);
    return 0;
}
static int shm_get(u64* dev, const char* slot, size_t bytes)
{
    if (shm_bounce()) return -1;
    memcpy(g_comm.bounce, slot, bytes);
    XLP_HIP(hipMemcpy(dev, g_comm.bounce, bytes, hipMemcpyHostToDevice));
    return 0;
}
static char* shm_slot(int from, int to) { return g_comm.shm + 4096 + ((size_t)from * g_comm.world + to) * g_comm.slot_bytes; }

static int shm_barrier()
{
    ShmHeader* h = shm_hdr();
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)g_comm.world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return 0;
    }
    if (!wait_for([&] { return h->generation.load(std::memory_order_acquire) != gen; }, 600.0)) return xlp_fail("xlp_comm: a rank did not reach the barrier within 600 s");
    return 0;
}

int comm_init(Ctx* ctx, int rank, int world, const char* backend, const char* rendezvous)
{
    Comm& c = g_comm;
    if (c.kind != Comm::NONE) return xlp_fail("xlp_comm_init: already initialised (xlp_comm_shutdown first)");
    if (world < 1 || rank < 0 || rank >= world) return xlp_fail("xlp_comm_init: rank / world out of range");
    if (world & (world - 1)) return xlp_fail("xlp_comm_init: the number of ranks must be a power of two (row shards of a 2^k-row trace)");
    const std::string be = backend ? backend : "";
    c.rank = rank; c.world = world;
    if (be.empty() || be == "self") {
        if (world != 1) return xlp_fail("xlp_comm_init: more than one rank needs a backend (\"rccl\" or \"shm\")");
        c.kind = Comm::SELF;
        return 0;
    }
    if (!rendezvous || !*rendezvous) return xlp_fail("xlp_comm_init: no rendezvous (a file path for rccl, a shared-memory name for shm)");
    if (be == "rccl") {
        if (rccl_load(c.rccl)) return -1;
        NcclId id;
        memset(&id, 0, sizeof id);
        const std::string path = rendezvous, tmp = path + ".tmp";
        if (rank == 0) {
            XLP_NCCL(c.rccl.GetUniqueId(&id));
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) { if (f) fclose(f); return xlp_fail("xlp_comm_init: cannot write the rendezvous file"); }
            fclose(f);
            if (rename(tmp.c_str(), path.c_str())) return xlp_fail("xlp_comm_init: cannot publish the rendezvous file");
        } else {
            bool got = false;
            if (!wait_for([&] {
                    FILE* f = fopen(path.c_str(), "rb");
                    if (!f) return false;
                    got = fread(&id, sizeof id, 1, f) == 1;
                    fclose(f);
                    return got;
                }, 300.0)) return xlp_fail("xlp_comm_init: rank 0 did not publish the RCCL id within 300 s");
        }
        XLP_HIP(hipSetDevice(ctx->device));
        XLP_NCCL(c.rccl.CommInitRank(&c.nccl, world, id, rank));
        c.kind = Comm::RCCL;
        return 0;
    }
    if (be == "shm") {
        c.shm_name = rendezvous[0] == '/' ? rendezvous : std::string("/") + rendezvous;
        u64 slot = 16ull << 20;
        if (const char* e = getenv("XLP_COMM_SHM_SLOT_MB")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 1 && v <= 1024) slot = v << 20; }
        c.slot_bytes = slot;
        c.shm_bytes = 4096 + (size_t)world * world * slot;
        int fd = -1;
        if (rank == 0) {
            shm_unlink(c.shm_name.c_str());
            fd = shm_open(c.shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)c.shm_bytes)) { if (fd >= 0) close(fd); return xlp_fail("xlp_comm_init: cannot create the shared-memory segment"); }
            c.shm_owner = true;
        } else {
            if (!wait_for([&] {
                    fd = shm_open(c.shm_name.c_str(), O_RDWR, 0600);
                    if (fd < 0) return false;
                    struct stat st;
                    if (fstat(fd, &st) || (size_t)st.st_size < c.shm_bytes) { close(fd); fd = -1; return false; }
                    return true;
                }, 300.0)) return xlp_fail("xlp_comm_init: rank 0 did not create the shared-memory segment within 300 s");
        }
        c.shm = (char*)mmap(nullptr, c.shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (c.shm == (char*)MAP_FAILED) { c.shm = nullptr; return xlp_fail("xlp_comm_init: cannot map the shared-memory segment"); }
        ShmHeader* h = shm_hdr();
        if (rank == 0) { h->world = (uint32_t)world; h->slot_bytes = slot; }     // (a fresh segment is zero-filled: counters start at 0)
        h->attached.fetch_add(1, std::memory_order_acq_rel);
        if (!wait_for([&] { return h->attached.load(std::memory_order_acquire) >= (uint32_t)world; }, 300.0)) return xlp_fail("xlp_comm_init: not every rank attached within 300 s");
        if (h->world != (uint32_t)world || h->slot_bytes != slot) return xlp_fail("xlp_comm_init: ranks disagree on world size / slot size");
        c.kind = Comm::SHM;                                   // (the bounce buffer is allocated by the first device transfer)
        return shm_barrier();
    }
    return xlp_fail("xlp_comm_init: unknown backend (\"rccl\", \"shm\", \"self\")");
}

void comm_shutdown()
{
    Comm& c = g_comm;
    if (c.kind == Comm::RCCL && c.nccl) { (void)c.rccl.CommDestroy(c.nccl); c.nccl = nullptr; }
    if (c.small) { (void)hipFree(c.small); c.small = nullptr; c.small_words = 0; }
    if (c.bounce) { (void)hipHostFree(c.bounce); c.bounce = nullptr; }
    if (c.shm) { munmap(c.shm, c.shm_bytes); c.shm = nullptr; if (c.shm_owner) shm_unlink(c.shm_name.c_str()); c.shm_owner = false; }
    c.kind = Comm::NONE; c.rank = 0; c.world = 1;
}

int comm_rank() { return g_comm.rank; }
int comm_world() { return g_comm.world; }

static void trace(const char* what, u64 words)
{
    static const bool on = getenv("XLP_COMM_TRACE") != nullptr;
    if (on) { fprintf(stderr, "[xlp_comm rank %d] %s %llu words\n", g_comm.rank, what, (unsigned long long)words); fflush(stderr); }
}
static int need_comm(const char* what)
{
    if (g_comm.kind == Comm::NONE) return xlp_fail((std::string(what) + ": xlp_comm_init has not been called").c_str());
    return 0;
}

// recv[p] <- the block rank p holds for me; send[p] -> rank p.  send_words[p] here == recv_words[me] on rank p.
int comm_exchange(Ctx* ctx, u64* const* recv, const u64* recv_words, const u64* const* send, const u64* send_words)
{
    if (need_comm("xlp_comm_exchange")) return -1;
    Comm& c = g_comm;
    const int me = c.rank, G = c.world;
    if (send_words[me] != recv_words[me]) return xlp_fail("xlp_comm_exchange: own block sizes differ");
    trace("exchange (own block)", send_words[me]);
    std::chrono::steady_clock::time_point t0;
    if (c.timing) { XLP_HIP(hipStreamSynchronize(ctx->stream)); t0 = std::chrono::steady_clock::now(); }
    if (send_words[me] && recv[me] != send[me])
        XLP_HIP(hipMemcpyAsync(recv[me], send[me], send_words[me] * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    u64 max_words = 0;
    for (int p = 0; p < G; p++) if (p != me) { max_words = std::max(max_words, send_words[p]); max_words = std::max(max_words, recv_words[p]); }
    if (G > 1 && c.kind == Comm::RCCL) {
        // every rank must run the same number of rounds: the longest message of the WHOLE exchange decides (an all-gather of one word)
        u64 mine = max_words, all_max = 0;
        std::vector<u64> every(G);
        if (comm_all_gather_host(ctx, every.data(), &mine, 1)) return -1;
        for (u64 v : every) all_max = std::max(all_max, v);
        for (u64 o = 0; o < all_max; o += MAX_MESSAGE_WORDS) {
            XLP_NCCL(c.rccl.GroupStart());
            for (int d = 1; d < G; d++) {                         // the same pairing order on every rank; empty tails are left out
                const int to = (me + d) % G, frm = (me - d + G) % G;
                if (send_words[to] > o) XLP_NCCL(c.rccl.Send(send[to] + o, (size_t)std::min(MAX_MESSAGE_WORDS, send_words[to] - o), NCCL_UINT64, to, c.nccl, ctx->stream));
                if (recv_words[frm] > o) XLP_NCCL(c.rccl.Recv(recv[frm] + o, (size_t)std::min(MAX_MESSAGE_WORDS, recv_words[frm] - o), NCCL_UINT64, frm, c.nccl, ctx->stream));
            }
            XLP_NCCL(c.rccl.GroupEnd());
        }
    } else if (G > 1 && c.kind == Comm::SHM) {
        XLP_HIP(hipStreamSynchronize(ctx->stream));               // what is sent was produced on the stream
        u64 mine = max_words, all_max = 0;
        std::vector<u64> every(G);
        if (comm_all_gather_host(ctx, every.data(), &mine, 1)) return -1;
        for (u64 v : every) all_max = std::max(all_max, v);
        const u64 slot_words = c.slot_bytes / sizeof(u64);
        for (u64 o = 0; o < all_max; o += slot_words) {
            for (int p = 0; p < G; p++)
                if (p != me && send_words[p] > o)
                    if (shm_put(shm_slot(me, p), send[p] + o, (size_t)std::min(slot_words, send_words[p] - o) * sizeof(u64))) return -1;
            if (shm_barrier()) return -1;
            for (int p = 0; p < G; p++)
                if (p != me && recv_words[p] > o)
                    if (shm_get(recv[p] + o, shm_slot(p, me), (size_t)std::min(slot_words, recv_words[p] - o) * sizeof(u64))) return -1;
            if (shm_barrier()) return -1;                          // nobody overwrites a slot that is still being read
        }
    }
    c.calls++;
    for (int p = 0; p < G; p++) if (p != me) c.bytes_sent += send_words[p] * sizeof(u64);
    if (c.timing) {
        XLP_HIP(hipStreamSynchronize(ctx->stream));
        c.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// out[p * words ..] <- rank p's `in` (device buffers)
int comm_all_gather(Ctx* ctx, u64* out, const u64* in, u64 words)
{
    if (need_comm("xlp_comm_all_gather")) return -1;
    Comm& c = g_comm;
    if (!words) return 0;
    trace("all_gather", words);
    if (c.world == 1 && c.kind != Comm::RCCL) {
        if (out != in) XLP_HIP(hipMemcpyAsync(out, in, words * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    if (c.kind == Comm::RCCL) {
        for (u64 o = 0; o < words; o += MAX_MESSAGE_WORDS) {      // rounds: every rank's piece [o, o + k) lands at out[p * words + o]
            const u64 k = std::min(MAX_MESSAGE_WORDS, words - o);
            if (k == words) { XLP_NCCL(c.rccl.AllGather(in, out, (size_t)words, NCCL_UINT64, c.nccl, ctx->stream)); break; }
            XLP_NCCL(c.rccl.GroupStart());
            for (int p = 0; p < c.world; p++) {
                XLP_NCCL(c.rccl.Send(in + o, (size_t)k, NCCL_UINT64, p, c.nccl, ctx->stream));
                XLP_NCCL(c.rccl.Recv(out + (u64)p * words + o, (size_t)k, NCCL_UINT64, p, c.nccl, ctx->stream));
            }
            XLP_NCCL(c.rccl.GroupEnd());
        }
        return 0;
    }
    // shm: my piece goes to the slot row "me -> *" (slot me -> 0 is used as the broadcast slot)
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    const u64 slot_words = c.slot_bytes / sizeof(u64);
    for (u64 o = 0; o < words; o += slot_words) {
        const u64 k = std::min(slot_words, words - o);
        if (shm_put(shm_slot(c.rank, 0), in + o, (size_t)k * sizeof(u64))) return -1;
        if (shm_barrier()) return -1;
        for (int p = 0; p < c.world; p++) if (shm_get(out + (u64)p * words + o, shm_slot(p, 0), (size_t)k * sizeof(u64))) return -1;
        if (shm_barrier()) return -1;
    }
    return 0;
}

// ring step: `send` goes to rank `to`, `recv` comes from rank `from` (both device buffers of `words` words)
int comm_sendrecv(Ctx* ctx, u64* recv, int from, const u64* send, int to, u64 words)
{
    if (need_comm("xlp_comm_sendrecv")) return -1;
    Comm& c = g_comm;
    if (from < 0 || from >= c.world || to < 0 || to >= c.world) return xlp_fail("xlp_comm_sendrecv: peer out of range");
    if (!words) return 0;
    trace("sendrecv", words);
    if (c.kind == Comm::SELF || (c.world == 1 && c.kind == Comm::SHM)) {
        XLP_HIP(hipMemcpyAsync(recv, send, words * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    if (c.kind == Comm::RCCL) {
        if (words > MAX_MESSAGE_WORDS) return xlp_fail("xlp_comm_sendrecv: message longer than 256 MiB (halo rows are expected here)");
        XLP_NCCL(c.rccl.GroupStart());
        XLP_NCCL(c.rccl.Send(send, (size_t)words, NCCL_UINT64, to, c.nccl, ctx->stream));
        XLP_NCCL(c.rccl.Recv(recv, (size_t)words, NCCL_UINT64, from, c.nccl, ctx->stream));
        XLP_NCCL(c.rccl.GroupEnd());
        return 0;
    }
    if (words * sizeof(u64) > c.slot_bytes) return xlp_fail("xlp_comm_sendrecv: message longer than a shared-memory slot");
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    if (shm_put(shm_slot(c.rank, to), send, (size_t)words * sizeof(u64))) return -1;
    if (shm_barrier()) return -1;
    if (shm_get(recv, shm_slot(from, c.rank), (size_t)words * sizeof(u64))) return -1;
    return shm_barrier();
}

// a few HOST words per rank (roots, totals, partial sums): out[p * words ..] <- rank p's `in`
int comm_all_gather_host(Ctx* ctx, u64* out, const u64* in, u64 words)
{
    if (need_comm("xlp_comm_all_gather_host")) return -1;
    Comm& c = g_comm;
    if (!words) return 0;
    trace("all_gather_host", words);
    if (c.world == 1 && c.kind != Comm::RCCL) { memmove(out, in, words * sizeof(u64)); return 0; }
    if (c.kind == Comm::RCCL) {                                   // RCCL moves device memory: a round trip through a small device buffer
        const size_t need = (size_t)words * (c.world + 1);
        if (c.small_words < need) {
            if (c.small) { XLP_HIP(hipStreamSynchronize(ctx->stream)); XLP_HIP(hipFree(c.small)); c.small = nullptr; }
            XLP_HIP(hipMalloc((void**)&c.small, need * sizeof(u64)));
            c.small_words = need;
        }
        XLP_HIP(hipMemcpyAsync(c.small, in, words * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
        XLP_NCCL(c.rccl.AllGather(c.small, c.small + words, (size_t)words, NCCL_UINT64, c.nccl, ctx->stream));
        XLP_HIP(hipMemcpyAsync(out, c.small + words, (size_t)words * c.world * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        XLP_HIP(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    if (words * sizeof(u64) > c.slot_bytes) return xlp_fail("xlp_comm_all_gather_host: more words than a shared-memory slot holds");
    memcpy(shm_slot(c.rank, c.rank), in, words * sizeof(u64));   // the diagonal slots are used by nothing else
    if (shm_barrier()) return -1;
    for (int p = 0; p < c.world; p++) memcpy(out + (u64)p * words, shm_slot(p, p), words * sizeof(u64));
    return shm_barrier();
}

int comm_barrier(Ctx* ctx)
{
    if (need_comm("xlp_comm_barrier")) return -1;
    if (ctx->inited) XLP_HIP(hipStreamSynchronize(ctx->stream));
    if (g_comm.world == 1 && g_comm.kind != Comm::RCCL) return 0;
    if (g_comm.kind == Comm::SHM) return shm_barrier();
    u64 one = 1;
    std::vector<u64> all(g_comm.world);
    return comm_all_gather_host(ctx, all.data(), &one, 1);
}

int comm_stats(double* seconds, u64* bytes_sent, u64* calls, int mode)
{
    Comm& c = g_comm;
    if (seconds) *seconds = c.seconds;
    if (bytes_sent) *bytes_sent = c.bytes_sent;
    if (calls) *calls = c.calls;
    if (mode & 1) { c.seconds = 0.0; c.bytes_sent = 0; c.calls = 0; }
    if (mode & 2) c.timing = true;
    if (mode & 4) c.timing = false;
    return 0;
}

}  // namespace xlp
