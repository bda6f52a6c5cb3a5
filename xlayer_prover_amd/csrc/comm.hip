// Transport between the ranks of a row-sharded proof: one process per GPU.  Declared in include/xlp_gl.h (xlp_comm_*).
//
// What it carries (host/starks.hpp, sharded mode): per committed section two all-to-alls (rows -> columns before the per-column LDE,
// columns -> rows before the row-wise leaf sponge: the split of ntt.extendPol + treesGL[i]->merkelize(), starks.cpp:53-60,134-141,
// 215-222), ring halo rows of the step programs' next-row operands, all-gathers of the three-column quotient / FRI polynomials and of the
// plookup tables, and a handful of words per rank (sub-tree roots, grand-product totals, partial evaluations).
//
// Backends:
//   "rccl"  RCCL over xGMI.  librccl.so.1 is loaded on first use (it is ~570 MB: a one-GPU process never pays for it).  The 128-byte
//           ncclUniqueId travels through a file: rank 0 writes <rendezvous>.tmp and renames it to <rendezvous>, the others wait for it;
//           the file carries rank 0's pid and start time, a file whose writer is gone is ignored, and rank 0 removes it once every rank
//           has joined.  An exchange is a grouped ncclSend / ncclRecv per round of <= 256 MiB per (sender, receiver) message (direct peer
//           writes, every link busy; a zkEVM section at G = 8 is 1.6 GB per message and this image's RCCL was seen to drop the tail of a
//           > 1 GiB message), the rank's own block is a device copy.  Everything is enqueued on the library's CURRENT stream
//           (xlp_stream_select), so exchanges order with the kernels of that stream and overlap the kernels of the other.
//           NOTHING of this backend has run with a peer yet: no multi-GPU box has been available to this build.
//   "shm"   ranks that SHARE a device or have no peer path (the one-GPU test box; a rehearsal of the N > 1 code path): POSIX shared
//           memory named <rendezvous>, every message staged through the host.  Correct and slow by construction.
//   world == 1 with backend NULL / "self": no transport; exchanges are device copies.
//
// Rendezvous by name is self-validating: a shared-memory segment is only accepted when the process that created it is alive and the
// name still points to the mapped inode (a killed run leaves its segment behind; rank 0 always replaces it, the others re-check the
// name while they wait and move over).
//
// Failure of one rank ends the others: every rank of a world > 1 maps a small control segment (<rendezvous>.ctl: abort flag, reason,
// one heartbeat per rank) and runs a watchdog thread.  xlp_comm_abort (called by the host's exit path), a failed transport call, or a
// heartbeat that stops while its rank has not shut down sets the flag; every wait of the shm backend returns an error when it is
// set, and a watchdog that sees it while its rank sits in an RCCL call (which has no timeout) aborts the communicator and ends the process.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
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
    int (*CommAbort)(NcclComm) = nullptr;
    int (*CommCount)(NcclComm, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};

static const uint32_t SEG_READY = 0x584c5031u;   // "XLP1": the creator finished initialising the segment
static const int MAX_RANKS = 64;

struct SegHeader {                               // first bytes of every shared segment
    std::atomic<uint32_t> ready;                 // SEG_READY, written last by the creator
    uint32_t world;
    uint64_t owner_pid, owner_start;             // the creator: a segment whose creator is gone is a leftover
    uint64_t bytes;
};
struct ShmHeader {
    SegHeader seg;
    std::atomic<uint32_t> arrived;               // barrier: ranks that reached the current generation
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;              // ranks that mapped the segment
    uint64_t slot_bytes;
};
struct CtlBlock {
    SegHeader seg;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> abort;                 // 1 + the rank that raised it (published after `why` is written)
    std::atomic<uint32_t> abort_claim;           // the first rank to fail takes it and writes the reason
    char why[240];
    std::atomic<uint64_t> beat[MAX_RANKS];       // CLOCK_MONOTONIC ms of the rank's last heartbeat
    std::atomic<uint32_t> done[MAX_RANKS];       // the rank has shut its communicator down (or never needed one)
};

struct Segment {
    std::string name;
    char* p = nullptr;
    size_t bytes = 0;
    bool owner = false;
    uint64_t ino = 0;
};

struct Comm {
    int rank = 0, world = 1;
    enum Kind { NONE, SELF, RCCL, SHM } kind = NONE;
    Rccl rccl;
    NcclComm nccl = nullptr;
    std::string id_file;                         // rccl: the rendezvous file (rank 0 removes it)
    int ranks_seen = 1;                          // ranks the transport itself reports (ncclCommCount / attached)
    bool self_via_rccl = false;                  // test hook: the rank's own block travels through ncclSend / ncclRecv too
    // shm
    Segment data;
    uint64_t slot_bytes = 0;
    char* bounce = nullptr;                      // page-locked staging buffer of one slot: HIP never sees the shared mapping itself (a copy
                                                 // engine writing straight into a MAP_SHARED /dev/shm mapping faulted on MI355X, XNACK off)
    u64* small = nullptr;                        // device staging for the host-word all-gather (rccl)
    size_t small_words = 0;
    u64 max_message_words = 1ull << 25;          // 256 MiB per (sender, receiver) message and round (XLP_COMM_MAX_MESSAGE_WORDS: tests)
    // failure propagation
    Segment ctl;
    std::thread watchdog;
    std::atomic<bool> watchdog_stop{false};
    std::atomic<bool> in_transport{false};       // the rank is inside a call that may block on a peer
    // accounting (bench): exchanges only
    double seconds = 0.0;
    u64 bytes_sent = 0, calls = 0;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;   // rccl / self: events around each exchange on its stream (read by comm_stats)
};
static Comm g_comm;

static CtlBlock* ctl() { return (CtlBlock*)g_comm.ctl.p; }
static ShmHeader* shm_hdr() { return (ShmHeader*)g_comm.data.p; }

static uint64_t now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000 + (uint64_t)ts.tv_nsec / 1000000;
}
// start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat); 0 if the process does not exist
static uint64_t proc_start(uint64_t pid)
{
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%llu/stat", (unsigned long long)pid);
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char* q = strrchr(buf, ')');             // the command name may hold spaces and parentheses
    if (!q) return 0;
    int field = 2;
    for (q++; *q; q++) {
        if (*q == ' ') { field++; if (field == 22) return strtoull(q + 1, nullptr, 10); }
    }
    return 0;
}
static bool owner_alive(const SegHeader* h)
{
    if (h->ready.load(std::memory_order_acquire) != SEG_READY) return false;
    const uint64_t st = proc_start(h->owner_pid);
    return st != 0 && st == h->owner_start;
}

// the first transport error of a call is kept; a raised abort flag ends every wait
static bool aborted()
{
    CtlBlock* c = ctl();
    return c && c->abort.load(std::memory_order_acquire) != 0;
}
// set when THIS rank's main thread raised the abort (comm_fail / comm_abort): it is already on its way out with the error, so the watchdog
// leaves the process alone.  Who raised the flag and who failed are different things: when the watchdog itself raises it for a dead peer the
// main thread may be inside ncclSend / ncclRecv or a stream synchronise for ever, and the watchdog must still end the process.
static std::atomic<bool> g_main_raised{false};
static void raise_abort(const char* why)
{
    CtlBlock* c = ctl();
    if (!c) return;
    uint32_t expect = 0;
    if (c->abort_claim.compare_exchange_strong(expect, 1u, std::memory_order_acq_rel)) {
        char msg[sizeof c->why];
        snprintf(msg, sizeof msg, "rank %d: %s", g_comm.rank, why ? why : "aborted");
        memcpy(c->why, msg, sizeof msg);
        c->abort.store(1u + (uint32_t)g_comm.rank, std::memory_order_release);
    }
}
static int fail_abort(const char* where)
{
    CtlBlock* c = ctl();
    char why[sizeof c->why + 1];
    memcpy(why, c->why, sizeof c->why); why[sizeof c->why] = 0;
    return xlp_fail((std::string(where) + ": another rank failed (" + why + ")").c_str());
}
// an error of this rank inside the transport: tell the others, then report it
static int comm_fail(const char* msg)
{
    g_main_raised.store(true, std::memory_order_release);
    raise_abort(msg);
    return xlp_fail(msg);
}

#define XLP_NCCL(call)                                                                                          \
    do {                                                                                                        \
        const int _r = (call);                                                                                  \
        if (_r != 0) return comm_fail((std::string("RCCL error in " #call ": ") + (g_comm.rccl.GetErrorString ? g_comm.rccl.GetErrorString(_r) : "?")).c_str()); \
    } while (0)
// inside a ncclGroupStart / ncclGroupEnd pair: remember the first failure, keep going to the GroupEnd (a group left open poisons every later call)
#define XLP_NCCL_IN_GROUP(err, call)                                                                            \
    do {                                                                                                        \
        const int _r = (call);                                                                                  \
        if (_r != 0 && (err).empty()) (err) = std::string("RCCL error in " #call ": ") + (g_comm.rccl.GetErrorString ? g_comm.rccl.GetErrorString(_r) : "?"); \
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
    r.CommAbort = (int (*)(NcclComm))sym("ncclCommAbort");
    r.CommCount = (int (*)(NcclComm, int*))sym("ncclCommCount");
    r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    r.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclRecv");
    r.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))sym("ncclAllGather");
    r.GroupStart = (int (*)())sym("ncclGroupStart");
    r.GroupEnd = (int (*)())sym("ncclGroupEnd");
    if (!ok) return xlp_fail("xlp_comm_init: librccl.so.1 lacks an expected symbol");
    return 0;
}

// poll `cond` until it holds; false on timeout or when the abort flag goes up (the caller tells which)
static bool wait_for(const std::function<bool()>& cond, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; !cond(); spins++) {
        if (aborted()) return cond();                             // (what was waited for may have happened just before the flag went up)
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        if (spins < 200) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 50 : 1000));
    }
    return true;
}

static void seg_close(Segment& s)
{
    if (s.p) { munmap(s.p, s.bytes); s.p = nullptr; }
    if (s.owner) shm_unlink(s.name.c_str());
    s.owner = false;
}
// Rank 0 replaces whatever carries the name by a fresh, zero-filled segment and marks it ready LAST; the others take a segment only when
// it is ready, as large as expected, and its creator is a live process -- and keep checking that the name still points to what they mapped
// (see seg_still_current), because a leftover of a killed run whose creator's pid has since been taken again would pass the liveness test.
static int seg_open(Segment& s, const std::string& name, size_t bytes, int rank, int world, double timeout_s)
{
    s.name = name; s.bytes = bytes;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name.c_str());
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes)) { if (fd >= 0) close(fd); return xlp_fail("xlp_comm_init: cannot create the shared-memory segment"); }
        s.owner = true;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (!fstat(fd, &st) && (size_t)st.st_size >= bytes) {
                    char* p = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    if (p != (char*)MAP_FAILED) {
                        const SegHeader* h = (const SegHeader*)p;
                        if (owner_alive(h) && h->world == (uint32_t)world && h->bytes == bytes) { s.p = p; s.ino = (uint64_t)st.st_ino; close(fd); return 0; }
                        munmap(p, bytes);
                    }
                }
                close(fd); fd = -1;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return xlp_fail("xlp_comm_init: rank 0 did not create the shared-memory segment in time (or ranks disagree on world / slot size)");
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    s.p = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    struct stat st;
    if (!fstat(fd, &st)) s.ino = (uint64_t)st.st_ino;
    close(fd);
    if (s.p == (char*)MAP_FAILED) { s.p = nullptr; return xlp_fail("xlp_comm_init: cannot map the shared-memory segment"); }
    SegHeader* h = (SegHeader*)s.p;                          // (a fresh segment is zero-filled: counters start at 0)
    h->world = (uint32_t)world; h->bytes = bytes;
    h->owner_pid = (uint64_t)getpid(); h->owner_start = proc_start(h->owner_pid);
    return 0;                                                // the creator publishes (ready) after filling in the rest of its header
}
static void seg_publish(Segment& s) { ((SegHeader*)s.p)->ready.store(SEG_READY, std::memory_order_release); }
// does the name still lead to the inode this rank mapped?  (false: rank 0 has replaced a leftover this rank had picked up)
static bool seg_still_current(const Segment& s)
{
    const int fd = shm_open(s.name.c_str(), O_RDWR, 0600);
    if (fd < 0) return false;
    struct stat st;
    const bool same = !fstat(fd, &st) && (uint64_t)st.st_ino == s.ino;
    close(fd);
    return same;
}
// attach to a segment and wait until all ranks have: `attached` is the segment's counter.  A rank that finds itself on a replaced
// segment starts over on the new one.
static int seg_join(Segment& s, const std::string& name, size_t bytes, int rank, int world, const std::function<std::atomic<uint32_t>*(char*)>& attached,
                    const std::function<void(char*)>& init_owner)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const double left = 300.0 - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (left <= 0) return xlp_fail("xlp_comm_init: not every rank attached within 300 s");
        if (seg_open(s, name, bytes, rank, world, left)) return -1;
        if (rank == 0) { init_owner(s.p); seg_publish(s); }
        std::atomic<uint32_t>* a = attached(s.p);
        // more ranks than the world on one segment: it is the leftover of a rendezvous that had completed
        bool moved = a->fetch_add(1, std::memory_order_acq_rel) + 1 > (uint32_t)world;
        uint64_t last_check = now_ms();
        while (!moved && a->load(std::memory_order_acquire) < (uint32_t)world) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 300.0) { seg_close(s); return xlp_fail("xlp_comm_init: not every rank attached within 300 s"); }
            if (rank != 0 && now_ms() - last_check >= 50) {
                last_check = now_ms();
                if (!seg_still_current(s)) moved = true;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (!moved && rank != 0 && !seg_still_current(s)) moved = true;
        if (!moved) return 0;
        // a leftover: wait until the name leads somewhere else (rank 0 replaces it when it arrives), then start over on the fresh segment
        while (seg_still_current(s)) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 300.0) { munmap(s.p, s.bytes); s.p = nullptr; return xlp_fail("xlp_comm_init: rank 0 did not replace a leftover segment within 300 s"); }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        munmap(s.p, s.bytes); s.p = nullptr;
    }
}

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
static char* shm_slot(int from, int to) { return g_comm.data.p + 4096 + ((size_t)from * g_comm.world + to) * g_comm.slot_bytes; }

static int shm_barrier()
{
    ShmHeader* h = shm_hdr();
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)g_comm.world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return 0;
    }
    double limit = 600.0;                                       // (XLP_COMM_BARRIER_TIMEOUT_S: harnesses that are themselves watched for silence)
    if (const char* e = getenv("XLP_COMM_BARRIER_TIMEOUT_S")) { const double v = atof(e); if (v >= 1.0) limit = v; }
    if (!wait_for([&] { return h->generation.load(std::memory_order_acquire) != gen; }, limit)) {
        if (aborted()) return fail_abort("xlp_comm");
        return comm_fail(("xlp_comm: rank " + std::to_string(g_comm.rank) + " waited " + std::to_string((int)limit) + " s at barrier " + std::to_string(gen) +
                          " of the shm transport: a rank did not reach it").c_str());
    }
    return 0;
}

// ---- failure propagation
static void watchdog_main()
{
    Comm& c = g_comm;
    CtlBlock* b = ctl();
    uint64_t stale_ms = 20000;                                  // (XLP_COMM_HEARTBEAT_STALE_MS: tests)
    if (const char* e = getenv("XLP_COMM_HEARTBEAT_STALE_MS")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 200) stale_ms = v; }
    while (!c.watchdog_stop.load(std::memory_order_acquire)) {
        const uint64_t t = now_ms();
        b->beat[c.rank].store(t, std::memory_order_release);
        if (!b->abort.load(std::memory_order_acquire)) {
            for (int p = 0; p < c.world; p++) {
                if (p == c.rank || b->done[p].load(std::memory_order_acquire)) continue;
                const uint64_t seen = b->beat[p].load(std::memory_order_acquire);
                if (seen && t > seen + stale_ms) {
                    char why[96];
                    snprintf(why, sizeof why, "rank %d stopped responding (no heartbeat for %llu s)", p, (unsigned long long)((t - seen) / 1000));
                    raise_abort(why);
                }
            }
        }
        const uint32_t ab = b->abort.load(std::memory_order_acquire);
        // (XLP_COMM_WATCHDOG_ENDS_PROCESS: the CPU test of this path runs it over the shm transport, whose waits poll the flag by themselves)
        if (ab && !g_main_raised.load(std::memory_order_acquire) && (c.kind == Comm::RCCL || getenv("XLP_COMM_WATCHDOG_ENDS_PROCESS"))) {
            // RCCL calls and the kernels they enqueue have no timeout: a rank waiting for a peer that is gone waits for ever, inside the call or
            // in the next stream synchronise.  Give the main thread a moment to notice by itself (its next xlp_comm_* call fails), then end the
            // process: the proof cannot complete and the communicator is unusable anyway
            std::this_thread::sleep_for(std::chrono::milliseconds(1500));
            if (c.watchdog_stop.load(std::memory_order_acquire)) break;
            char why[sizeof b->why + 1];
            memcpy(why, b->why, sizeof b->why); why[sizeof b->why] = 0;
            fprintf(stderr, "xlp_comm rank %d: %s (%s): aborting the communicator and ending this process\n", c.rank,
                    ab == 1u + (uint32_t)c.rank ? "this rank's watchdog gave up on a peer" : "another rank failed", why);
            fflush(stderr);
            if (c.nccl && c.rccl.CommAbort) (void)c.rccl.CommAbort(c.nccl);
            _exit(86);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}
static std::string ctl_name(const std::string& rendezvous)
{
    // a shared-memory name derived from the rendezvous (a path for rccl): its characters other than [A-Za-z0-9_.-] become '_', the tail is kept
    std::string n;
    for (char ch : rendezvous) n += (isalnum((unsigned char)ch) || ch == '_' || ch == '.' || ch == '-') ? ch : '_';
    if (n.size() > 200) n = n.substr(n.size() - 200);
    return "/" + n + ".ctl";
}
static int ctl_open(const std::string& rendezvous, int rank, int world)
{
    Comm& c = g_comm;
    if (world > MAX_RANKS) return xlp_fail("xlp_comm_init: more ranks than the control block holds");
    if (seg_join(c.ctl, ctl_name(rendezvous), sizeof(CtlBlock), rank, world, [](char* p) { return &((CtlBlock*)p)->attached; }, [](char*) {})) return -1;
    ctl()->beat[rank].store(now_ms(), std::memory_order_release);
    c.watchdog_stop.store(false);
    g_main_raised.store(false);
    c.watchdog = std::thread(watchdog_main);
    return 0;
}
static void ctl_close()
{
    Comm& c = g_comm;
    if (!c.ctl.p) return;
    ctl()->done[c.rank].store(1, std::memory_order_release);
    c.watchdog_stop.store(true, std::memory_order_release);
    if (c.watchdog.joinable()) c.watchdog.join();
    seg_close(c.ctl);
}
struct InTransport {                                          // the main thread is inside a call that may wait for a peer
    InTransport() { g_comm.in_transport.store(true, std::memory_order_release); }
    ~InTransport() { g_comm.in_transport.store(false, std::memory_order_release); }
};

void comm_abort(const char* why)
{
    g_main_raised.store(true, std::memory_order_release);
    raise_abort(why && *why ? why : "the host gave up");
}

int comm_init(Ctx* ctx, int rank, int world, const char* backend, const char* rendezvous)
{
    Comm& c = g_comm;
    if (c.kind != Comm::NONE) return xlp_fail("xlp_comm_init: already initialised (xlp_comm_shutdown first)");
    if (world < 1 || rank < 0 || rank >= world) return xlp_fail("xlp_comm_init: rank / world out of range");
    if (world & (world - 1)) return xlp_fail("xlp_comm_init: the number of ranks must be a power of two (row shards of a 2^k-row trace)");
    const std::string be = backend ? backend : "";
    c.rank = rank; c.world = world; c.ranks_seen = 1;
    if (plan_active()) {                                          // plan mode (xlp_internal.hpp): this process IS rank `rank` of `world`, nobody is contacted
        c.kind = Comm::SELF; c.ranks_seen = world;
        return 0;
    }
    c.max_message_words = 1ull << 25;
    if (const char* e = getenv("XLP_COMM_MAX_MESSAGE_WORDS")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 1) c.max_message_words = v; }
    c.self_via_rccl = getenv("XLP_COMM_SELF_VIA_RCCL") != nullptr;
    if (be.empty() || be == "self") {
        if (world != 1) return xlp_fail("xlp_comm_init: more than one rank needs a backend (\"rccl\" or \"shm\")");
        c.kind = Comm::SELF;
        return 0;
    }
    if (!rendezvous || !*rendezvous) return xlp_fail("xlp_comm_init: no rendezvous (a file path for rccl, a shared-memory name for shm)");
    if (be != "rccl" && be != "shm") return xlp_fail("xlp_comm_init: unknown backend (\"rccl\", \"shm\", \"self\")");
    if (world > 1 && ctl_open(rendezvous, rank, world)) { c.rank = 0; c.world = 1; return -1; }
    auto undo = [&]() { ctl_close(); c.rank = 0; c.world = 1; return -1; };
    if (be == "rccl") {
        if (rccl_load(c.rccl)) return undo();
        struct IdFile { uint32_t magic; uint32_t world; uint64_t pid, start; NcclId id; } rec;
        memset(&rec, 0, sizeof rec);
        const std::string path = rendezvous, tmp = path + ".tmp";
        if (rank == 0) {
            if (c.rccl.GetUniqueId(&rec.id) != 0) { (void)comm_fail("xlp_comm_init: ncclGetUniqueId failed"); return undo(); }
            rec.magic = SEG_READY; rec.world = (uint32_t)world; rec.pid = (uint64_t)getpid(); rec.start = proc_start(rec.pid);
            unlink(path.c_str());
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&rec, sizeof rec, 1, f) != 1) { if (f) fclose(f); (void)comm_fail("xlp_comm_init: cannot write the rendezvous file"); return undo(); }
            fclose(f);
            if (rename(tmp.c_str(), path.c_str())) { (void)comm_fail("xlp_comm_init: cannot publish the rendezvous file"); return undo(); }
            c.id_file = path;
        } else {
            // only an id whose writer is alive: a leftover of a killed run would make ncclCommInitRank wait for ever
            if (!wait_for([&] {
                    FILE* f = fopen(path.c_str(), "rb");
                    if (!f) return false;
                    const bool got = fread(&rec, sizeof rec, 1, f) == 1;
                    fclose(f);
                    if (!got || rec.magic != SEG_READY || rec.world != (uint32_t)world) return false;
                    const uint64_t st = proc_start(rec.pid);
                    return st != 0 && st == rec.start;
                }, 300.0)) {
                if (aborted()) { (void)fail_abort("xlp_comm_init"); return undo(); }
                (void)comm_fail("xlp_comm_init: rank 0 did not publish the RCCL id within 300 s");
                return undo();
            }
        }
        if (hipSetDevice(ctx->device) != hipSuccess) { (void)comm_fail("xlp_comm_init: hipSetDevice failed"); return undo(); }
        {
            InTransport guard;
            const int r = c.rccl.CommInitRank(&c.nccl, world, rec.id, rank);
            if (r != 0) { (void)comm_fail((std::string("xlp_comm_init: ncclCommInitRank: ") + c.rccl.GetErrorString(r)).c_str()); c.nccl = nullptr; return undo(); }
        }
        int cnt = 0;
        if (c.rccl.CommCount && c.rccl.CommCount(c.nccl, &cnt) == 0) c.ranks_seen = cnt;
        if (rank == 0) unlink(path.c_str());                  // every rank has joined: nobody needs the id any more
        c.kind = Comm::RCCL;
        return 0;
    }
    // shm
    u64 slot = 16ull << 20;
    if (const char* e = getenv("XLP_COMM_SHM_SLOT_MB")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 1 && v <= 1024) slot = v << 20; }
    c.slot_bytes = slot;
    const std::string name = rendezvous[0] == '/' ? rendezvous : std::string("/") + rendezvous;
    if (seg_join(c.data, name, 4096 + (size_t)world * world * slot, rank, world, [](char* p) { return &((ShmHeader*)p)->attached; },
                 [&](char* p) { ((ShmHeader*)p)->slot_bytes = slot; })) return undo();
    if (shm_hdr()->slot_bytes != slot) { seg_close(c.data); (void)comm_fail("xlp_comm_init: ranks disagree on the slot size"); return undo(); }
    c.ranks_seen = (int)shm_hdr()->attached.load(std::memory_order_acquire);
    c.kind = Comm::SHM;                                       // (the bounce buffer is allocated by the first device transfer)
    if (world > 1 && shm_barrier()) { seg_close(c.data); c.kind = Comm::NONE; return undo(); }
    return 0;
}

void comm_shutdown()
{
    Comm& c = g_comm;
    if (c.kind == Comm::RCCL && c.nccl) {
        // after a failure the peers may be gone: ncclCommDestroy would wait for them
        if (aborted() && c.rccl.CommAbort) (void)c.rccl.CommAbort(c.nccl); else (void)c.rccl.CommDestroy(c.nccl);
        c.nccl = nullptr;
    }
    if (!c.id_file.empty()) { unlink(c.id_file.c_str()); c.id_file.clear(); }
    if (c.small) { (void)plan_free(c.small); c.small = nullptr; c.small_words = 0; }
    if (c.bounce) { XLP_HIPV(hipHostFree(c.bounce)); c.bounce = nullptr; }
    seg_close(c.data);
    ctl_close();
    c.kind = Comm::NONE; c.rank = 0; c.world = 1; c.ranks_seen = 1;
}

int comm_rank() { return g_comm.rank; }
int comm_world() { return g_comm.world; }
int comm_ranks_seen() { return g_comm.kind == Comm::NONE ? 0 : g_comm.ranks_seen; }

static void trace(const char* what, u64 words)
{
    static const bool on = getenv("XLP_COMM_TRACE") != nullptr;
    if (on) { fprintf(stderr, "[xlp_comm rank %d] %s %llu words\n", g_comm.rank, what, (unsigned long long)words); fflush(stderr); }
}
static int need_comm(const char* what)
{
    if (g_comm.kind == Comm::NONE) return xlp_fail((std::string(what) + ": xlp_comm_init has not been called").c_str());
    if (aborted()) return fail_abort(what);
    return 0;
}

// recv[p] <- the block rank p holds for me; send[p] -> rank p.  send_words[p] here == recv_words[me] on rank p.
// all_max_words: the longest (sender, receiver) message of the WHOLE exchange, the same number on every rank -- it fixes the number of
// rounds every rank must run; a caller that knows every rank's sizes (host/starks.hpp: the band plan) passes it and the exchange is
// enqueued without a host round trip.  0: the ranks agree on it here (an all-gather of one host word: synchronises the stream).
int comm_exchange(Ctx* ctx, u64* const* recv, const u64* recv_words, const u64* const* send, const u64* send_words, u64 all_max_words)
{
    if (need_comm("xlp_comm_exchange")) return -1;
    Comm& c = g_comm;
    const int me = c.rank, G = c.world;
    if (send_words[me] != recv_words[me]) return xlp_fail("xlp_comm_exchange: own block sizes differ");
    trace("exchange (own block)", send_words[me]);
    // timing (bench): the shm backend is synchronous on the host anyway and is timed there; the others are timed with events on the stream
    // the exchange is enqueued on, so that measuring does not undo the overlap with the kernels of the other stream
    std::chrono::steady_clock::time_point t0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c.timing && c.kind == Comm::SHM) { XLP_HIP(hipStreamSynchronize(ctx->stream)); t0 = std::chrono::steady_clock::now(); }
    else if (c.timing) {
        XLP_HIP(hipEventCreate(&ev0)); XLP_HIP(hipEventCreate(&ev1));
        c.timed.emplace_back(ev0, ev1);
        XLP_HIP(hipEventRecord(ev0, ctx->stream));
    }
    const bool own_by_rccl = c.kind == Comm::RCCL && c.self_via_rccl;
    if (!own_by_rccl && send_words[me] && recv[me] != send[me])
        XLP_HIP(hipMemcpyAsync(recv[me], send[me], send_words[me] * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    u64 max_words = 0;
    for (int p = 0; p < G; p++) if (p != me || own_by_rccl) { max_words = std::max(max_words, send_words[p]); max_words = std::max(max_words, recv_words[p]); }
    if (all_max_words && max_words > all_max_words) return comm_fail("xlp_comm_exchange: a message is longer than the agreed maximum");
    u64 all_max = all_max_words;
    if ((G > 1 || own_by_rccl) && !all_max && (c.kind == Comm::RCCL || c.kind == Comm::SHM)) {
        if (c.kind == Comm::SHM) XLP_HIP(hipStreamSynchronize(ctx->stream));
        u64 mine = max_words;
        std::vector<u64> every(G);
        if (comm_all_gather_host(ctx, every.data(), &mine, 1)) return -1;
        for (u64 v : every) all_max = std::max(all_max, v);
    }
    if ((G > 1 || own_by_rccl) && c.kind == Comm::RCCL) {
        InTransport guard;
        const u64 M = c.max_message_words;
        for (u64 o = 0; o < all_max; o += M) {
            std::string err;
            XLP_NCCL(c.rccl.GroupStart());
            for (int d = own_by_rccl ? 0 : 1; d < G; d++) {       // the same pairing order on every rank; empty tails are left out
                const int to = (me + d) % G, frm = (me - d + G) % G;
                if (send_words[to] > o) XLP_NCCL_IN_GROUP(err, c.rccl.Send(send[to] + o, (size_t)std::min(M, send_words[to] - o), NCCL_UINT64, to, c.nccl, ctx->stream));
                if (recv_words[frm] > o) XLP_NCCL_IN_GROUP(err, c.rccl.Recv(recv[frm] + o, (size_t)std::min(M, recv_words[frm] - o), NCCL_UINT64, frm, c.nccl, ctx->stream));
            }
            XLP_NCCL_IN_GROUP(err, c.rccl.GroupEnd());
            if (!err.empty()) return comm_fail(err.c_str());
        }
    } else if (G > 1 && c.kind == Comm::SHM) {
        XLP_HIP(hipStreamSynchronize(ctx->stream));               // what is sent was produced on the stream
        const u64 slot_words = std::min<u64>(c.slot_bytes / sizeof(u64), c.max_message_words);
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
    if (ev1) XLP_HIP(hipEventRecord(ev1, ctx->stream));
    else if (c.timing) {
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
        InTransport guard;
        const u64 M = c.max_message_words;
        for (u64 o = 0; o < words; o += M) {                      // rounds: every rank's piece [o, o + k) lands at out[p * words + o]
            const u64 k = std::min(M, words - o);
            if (k == words) { XLP_NCCL(c.rccl.AllGather(in, out, (size_t)words, NCCL_UINT64, c.nccl, ctx->stream)); break; }
            std::string err;
            XLP_NCCL(c.rccl.GroupStart());
            for (int p = 0; p < c.world; p++) {
                XLP_NCCL_IN_GROUP(err, c.rccl.Send(in + o, (size_t)k, NCCL_UINT64, p, c.nccl, ctx->stream));
                XLP_NCCL_IN_GROUP(err, c.rccl.Recv(out + (u64)p * words + o, (size_t)k, NCCL_UINT64, p, c.nccl, ctx->stream));
            }
            XLP_NCCL_IN_GROUP(err, c.rccl.GroupEnd());
            if (!err.empty()) return comm_fail(err.c_str());
        }
        return 0;
    }
    // shm: my piece goes to the slot row "me -> *" (slot me -> 0 is used as the broadcast slot)
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    const u64 slot_words = std::min<u64>(c.slot_bytes / sizeof(u64), c.max_message_words);
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
        if (words > (1ull << 25)) return xlp_fail("xlp_comm_sendrecv: message longer than 256 MiB (halo rows are expected here)");
        InTransport guard;
        std::string err;
        XLP_NCCL(c.rccl.GroupStart());
        XLP_NCCL_IN_GROUP(err, c.rccl.Send(send, (size_t)words, NCCL_UINT64, to, c.nccl, ctx->stream));
        XLP_NCCL_IN_GROUP(err, c.rccl.Recv(recv, (size_t)words, NCCL_UINT64, from, c.nccl, ctx->stream));
        XLP_NCCL_IN_GROUP(err, c.rccl.GroupEnd());
        if (!err.empty()) return comm_fail(err.c_str());
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
        InTransport guard;
        const size_t need = (size_t)words * (c.world + 1);
        if (c.small_words < need) {
            if (c.small) { XLP_HIP(hipStreamSynchronize(ctx->stream)); XLP_ALLOC(plan_free(c.small)); c.small = nullptr; }
            XLP_ALLOC(plan_malloc((void**)&c.small, need * sizeof(u64)));
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
    for (auto& pr : c.timed) {                                    // (an exchange that failed half way left its second event unrecorded: skipped)
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c.seconds += ms * 1e-3;
        XLP_HIPV(hipGetLastError());
        XLP_HIPV(hipEventDestroy(pr.first)); XLP_HIPV(hipEventDestroy(pr.second));
    }
    c.timed.clear();
    if (seconds) *seconds = c.seconds;
    if (bytes_sent) *bytes_sent = c.bytes_sent;
    if (calls) *calls = c.calls;
    if (mode & 1) { c.seconds = 0.0; c.bytes_sent = 0; c.calls = 0; }
    if (mode & 2) c.timing = true;
    if (mode & 4) c.timing = false;
    return 0;
}

}  // namespace xlp
