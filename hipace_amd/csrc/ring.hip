// ring.hip -- transport of the time-step ring pipeline: one slice hand-off per rank and slice, RCCL point-to-point
// over xGMI (the reference: MultiBuffer::{make_progress, get_data, put_data}, utils/MultiBuffer.cpp:287-609, MPI
// Isend / Irecv between ring neighbours; in-process copy when a rank sends to itself, :299-308).
//
// One process per GPU.  Every EDGE r -> r+1 of the ring is its own 2-rank communicator (the sender is comm rank 0),
// so a rank's receive path (its predecessor's edge) and its send path (its own edge) never share a communicator or
// a stream: ncclRecv runs on the ring's receive stream, ncclSend on its send stream, and the two progress
// independently -- a rank only ever waits for its predecessor, not for its successor's schedule (a grouped
// send+recv on one communicator would step all ranks in lock-step and pay max-over-ranks jitter on every slice:
// the multigrid's V-cycle count differs from slice to slice).  Ordering against the engine is by events only:
// a send waits (on the device) for the event the engine recorded behind the slice's last kernel, the engine waits
// for the event the ring records behind a receive.  No host synchronisation per slice.
//
// RCCL is opened with dlopen so that the library loads (and the rest of the ABI works) in a process that never
// touches the ring; when torch has already loaded its librccl.so.1 the same object is reused.
//
// Second kind of edge under the same entry points (HPS_RING_EDGE=ipc when the ids are made): a PEER COPY ordered through a
// mailbox in POSIX shared memory -- for processes of one node, on different devices or on the SAME device (RCCL refuses
// two ranks on one device, so this is also the edge a one-GPU box can run a ring of processes on).  The receiver writes a
// descriptor per posted receive (which of its allocations, offset, bytes; allocations are exported once with
// hipIpcGetMemHandle of their BASE -- a handle made from an interior pointer opens at the allocation's base on this
// runtime, measured: scripts/ubench/ipc_torch_probe.py) and lets its receive stream write "buffer k is free" into the
// mailbox behind the event that says so (hipStreamWriteValue64 on the registered page); the sender's host enqueues
// hipMemcpyAsync(peer pointer) + hipStreamWriteValue64("message k has landed") on its send stream once receive k is posted
// and free; the receiver's host reads that word before it enqueues the slice that needs the message -- no kernel sits on
// the device waiting for data, nothing depends on how streams map onto hardware queues, and a side that waits gives up
// after HPS_RING_TIMEOUT_S with the counters in the error text.  The reference's progress engine is of this kind
// (MultiBuffer::make_progress polls MPI_Test from the host, MultiBuffer.cpp:287-442; comms_buffer.on_gpu :26-40,76-81).
#include "common.h"

#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_set>
#include <memory>

#include <dlfcn.h>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <thread>
#include <string>
#include <cstring>
#include <vector>

namespace {

// the few RCCL entry points the ring needs (rccl.h: NCCL 2.x ABI; ncclUniqueId = 128 bytes, ncclChar = 0)
struct NcclId { char internal[HPS_RING_ID_BYTES]; };
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(NcclId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};

Rccl g_rccl;

int load_rccl ()
{
    if (g_rccl.so) return HPS_OK;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) { hps::set_error(std::string("hps_ring: cannot open librccl.so.1: ") + dlerror()); return HPS_ERR_COMM; }
#define HPS_SYM(field, name) \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(so, name); \
    if (!g_rccl.field) { hps::set_error(std::string("hps_ring: librccl lacks ") + name); dlclose(so); return HPS_ERR_COMM; }
    HPS_SYM(GetUniqueId, "ncclGetUniqueId")
    HPS_SYM(CommInitRank, "ncclCommInitRank")
    HPS_SYM(CommDestroy, "ncclCommDestroy")
    HPS_SYM(Send, "ncclSend")
    HPS_SYM(Recv, "ncclRecv")
    HPS_SYM(GroupStart, "ncclGroupStart")
    HPS_SYM(GroupEnd, "ncclGroupEnd")
    HPS_SYM(GetErrorString, "ncclGetErrorString")
#undef HPS_SYM
    *reinterpret_cast<void**>(&g_rccl.GetLastError) = dlsym(so, "ncclGetLastError");      // optional
    *reinterpret_cast<void**>(&g_rccl.CommCount) = dlsym(so, "ncclCommCount");            // optional (hps_ring_info)
    *reinterpret_cast<void**>(&g_rccl.CommUserRank) = dlsym(so, "ncclCommUserRank");
    g_rccl.so = so;
    return HPS_OK;
}

#define HPS_NCCL_CHECK(expr)                                                                          \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != 0) {                                                                                \
            hps::set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(r_) +                    \
                           (g_rccl.GetLastError ? std::string(" / ") + g_rccl.GetLastError(nullptr) : std::string())); \
            return HPS_ERR_COMM;                                                                      \
        }                                                                                             \
    } while (0)

// ---- the mailbox of one ipc edge (one POSIX shared-memory segment, mapped and hipHostRegister'ed by both sides) -------
constexpr uint32_t kBoxMagic = 0x48505342u;     // "HPSB"
constexpr uint32_t kBoxVersion = 2;             // layout of Mailbox (2: 65536 descriptors)
constexpr int kBoxArenas = 64;
constexpr uint64_t kBoxDescs = 65536;           // receives posted and not yet matched by a send (up to two whole steps ahead, <= 2 per slice: boxes of 16 k slices)
struct alignas(64) BoxWord { volatile unsigned long long v; char pad[56]; };
struct BoxArena { hipIpcMemHandle_t handle; unsigned long long base, size; };
struct BoxDesc { unsigned long long offset, bytes; unsigned int arena, pad; };
struct Mailbox {
    uint32_t magic, version;
    std::atomic<int> sender_attached, receiver_attached, sender_gone, receiver_gone;
    char pad0[40];
    BoxWord posted;        // receiver's host: receive descriptors written so far
    BoxWord freed;         // receiver's receive STREAM: receives whose buffer may be written (monotonic, <= posted)
    BoxWord issued;        // sender's host: messages enqueued so far
    BoxWord landed;        // sender's send STREAM: messages whose payload is in the receiver's buffer
    std::atomic<int> n_arenas; char pad1[60];
    BoxArena arenas[kBoxArenas];
    BoxDesc desc[kBoxDescs];
};
constexpr char kIpcTag[] = "HPSIPC1:";
struct Ticket { unsigned long long magic, seq; };
constexpr unsigned long long kTicketMagic = 0x4850535449434b54ull;
// every live receive handle of an ipc edge: hps_engine_wait_event refuses them (a host written against the RCCL edge would
// hand one to hipStreamWaitEvent as if it were a hipEvent_t)
std::mutex g_tickets_mu;
std::unordered_set<const void*> g_tickets;

struct Box {                                    // one side's view of a mailbox
    Mailbox* m = nullptr; std::string name; bool creator = false, registered = false;
    unsigned long long* d_freed = nullptr; unsigned long long* d_landed = nullptr;      // device addresses of the two stream-written words
};
std::map<std::string, Box> g_pending_boxes;     // made by hps_ring_unique_id, claimed by hps_ring_init
// a mailbox whose id was made but never claimed (the host gave up between hps_ring_unique_id and hps_ring_init) must not
// stay behind in /dev/shm when the process ends
struct PendingBoxesCleanup {
    ~PendingBoxesCleanup () {
        for (auto& kv : g_pending_boxes) {
            if (kv.second.m) munmap(kv.second.m, sizeof(Mailbox));
            if (!kv.second.name.empty()) shm_unlink(kv.second.name.c_str());
        }
    }
} g_pending_boxes_cleanup;

struct Ring {
    int rank = 0, world = 1, device = 0;
    int kind = 0;                                          // 0 = RCCL, 1 = ipc peer copies
    Box box_in, box_out;                                   // ipc: edge (rank-1 -> rank): I receive; edge (rank -> rank+1): I send
    std::vector<void*> peer_arena;                         // ipc, sender: the receiver's allocations as opened here
    struct Range { unsigned long long base, size; int id; };
    std::vector<Range> my_arenas;                          // ipc, receiver: my allocations that have been exported
    std::vector<std::unique_ptr<Ticket>> tickets;
    unsigned long long n_posted = 0, n_issued = 0;
    ncclComm_t comm_in = nullptr, comm_out = nullptr;      // edge (rank-1 -> rank): I am comm rank 1; edge (rank -> rank+1): comm rank 0
    ncclComm_t comm_self = nullptr;                        // world == 1: one 1-rank communicator ("send to myself")
    hipStream_t st_recv = nullptr, st_send = nullptr;
    std::vector<hipEvent_t> ev_recv, ev_send;
    long n_sent = 0, n_received = 0; long long bytes_sent = 0, bytes_received = 0;
};

// proper edge colouring of the ring: two colours, three when the ring is odd.  Communicators are created colour by
// colour, so no rank waits in one ncclCommInitRank for a peer that waits for it in another.
int edge_colour (int e, int world) { return (world % 2 == 1 && e == world - 1) ? 2 : e % 2; }

int event_of (std::vector<hipEvent_t>& pool, int slot, hipEvent_t* out)
{
    HPS_REQUIRE(slot >= 0 && slot < (1 << 20), "hps_ring: bad event slot");
    if ((size_t)slot >= pool.size()) pool.resize((size_t)slot + 1, nullptr);
    static const bool nofence = [] { const char* v = std::getenv("HPS_EVENT_FENCE"); return v && std::atoi(v) >= 2; }();
    if (!pool[slot]) HPS_HIP_CHECK(hipEventCreateWithFlags(&pool[slot], hipEventDisableTiming | (nofence ? hipEventDisableSystemFence : 0u)));
    *out = pool[slot];
    return HPS_OK;
}

// ---- ipc edge ------------------------------------------------------------------------------------------------------
int edge_kind_env ()
{
    const char* v = std::getenv("HPS_RING_EDGE");
    if (!v || !*v) return 1;                    // default: peer copies through the mailbox (no resident waiter, no queue requirement)
    return std::strcmp(v, "rccl") == 0 ? 0 : 1;
}
bool is_ipc_id (const char* id) { return id && std::strncmp(id, kIpcTag, sizeof(kIpcTag) - 1) == 0; }

double timeout_env (const char* name, double dflt)
{
    const char* e = std::getenv(name);
    const double v = e ? std::atof(e) : dflt;
    return v > 0.0 ? v : dflt;
}

int box_map (Box& B, const std::string& name, bool create)
{
    const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { hps::set_error("hps_ring: shm_open(" + name + ") failed: " + std::strerror(errno)); return HPS_ERR_COMM; }
    if (create && ftruncate(fd, (off_t)sizeof(Mailbox)) != 0) { close(fd); shm_unlink(name.c_str()); hps::set_error("hps_ring: cannot size the mailbox " + name); return HPS_ERR_COMM; }
    void* p = mmap(nullptr, sizeof(Mailbox), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (create) shm_unlink(name.c_str()); hps::set_error("hps_ring: cannot map the mailbox " + name); return HPS_ERR_COMM; }
    B.m = static_cast<Mailbox*>(p); B.name = name; B.creator = create;
    if (create) { B.m->version = kBoxVersion; B.m->magic = kBoxMagic; }       // (a fresh segment is zero-filled)
    return HPS_OK;
}
void box_unmap (Box& B)
{
    if (!B.m) return;
    if (B.registered) (void)hipHostUnregister(B.m);
    munmap(B.m, sizeof(Mailbox));
    if (B.creator && !B.name.empty()) shm_unlink(B.name.c_str());      // (harmless when already unlinked)
    B = Box();
}
int box_register (Box& B)
{
    HPS_HIP_CHECK(hipHostRegister(B.m, sizeof(Mailbox), hipHostRegisterPortable | hipHostRegisterMapped));
    B.registered = true;
    HPS_HIP_CHECK(hipHostGetDevicePointer((void**)&B.d_freed, (void*)&B.m->freed.v, 0));
    HPS_HIP_CHECK(hipHostGetDevicePointer((void**)&B.d_landed, (void*)&B.m->landed.v, 0));
    return HPS_OK;
}

// host wait with a deadline: `cond` polled; `gone` = the peer has said goodbye
template <class F>
int ipc_wait (Ring* R, F cond, const std::atomic<int>* gone, double seconds, const char* what)
{
    if (cond()) return HPS_OK;
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (!cond()) {
        if ((++spins & 0xfff) == 0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const bool left = gone && gone->load() != 0;
            if (left && cond()) return HPS_OK;          // (what the peer did before it left counts)
            if (left || dt > seconds) {
                char buf[448];
                const Mailbox* i = R->box_in.m; const Mailbox* o = R->box_out.m;
                std::snprintf(buf, sizeof buf, "%s: rank %d of %d %s after %.1f s; incoming edge: posted %llu freed %llu issued %llu landed %llu; "
                              "outgoing edge: posted %llu freed %llu issued %llu landed %llu", what, R->rank, R->world,
                              left ? "-- the peer has left the ring" : "still waiting", dt,
                              i ? i->posted.v : 0ull, i ? i->freed.v : 0ull, i ? i->issued.v : 0ull, i ? i->landed.v : 0ull,
                              o ? o->posted.v : 0ull, o ? o->freed.v : 0ull, o ? o->issued.v : 0ull, o ? o->landed.v : 0ull);
                hps::set_error(buf);
                return HPS_ERR_COMM;
            }
            if (dt > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    return HPS_OK;
}

// receiver: which exported allocation holds p?  Exports the allocation (its base) on first sight.
int ipc_arena_of (Ring* R, const void* p, long bytes, unsigned* id, unsigned long long* offset)
{
    const unsigned long long a = (unsigned long long)(uintptr_t)p;
    void* base = nullptr; size_t size = 0;
    HPS_HIP_CHECK(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)const_cast<void*>(p)));
    // a cached range only counts while it is still the SAME allocation (base and size): a freed receive pool whose address
    // range was handed out again would otherwise be reached through the stale mapping in the sender -- "landed" flagged, the
    // data nowhere.  (An allocation freed and re-made with the same base AND size cannot be told apart here: receive buffers
    // must stay allocated until hps_ring_destroy, include/hpslice.h.)
    for (const auto& r : R->my_arenas)
        if ((unsigned long long)(uintptr_t)base == r.base && (unsigned long long)size == r.size &&
            a >= r.base && a + (unsigned long long)bytes <= r.base + r.size) { *id = (unsigned)r.id; *offset = a - r.base; return HPS_OK; }
    HPS_REQUIRE(base && a + (unsigned long long)bytes <= (unsigned long long)(uintptr_t)base + size, "hps_ring_recv_slice: the buffer is not inside one device allocation");
    Mailbox* m = R->box_in.m;
    const int n = m->n_arenas.load();
    HPS_REQUIRE(n < kBoxArenas, "hps_ring_recv_slice: too many distinct allocations hold receive buffers (64)");
    HPS_HIP_CHECK(hipIpcGetMemHandle(&m->arenas[n].handle, base));
    m->arenas[n].base = (unsigned long long)(uintptr_t)base; m->arenas[n].size = size;
    m->n_arenas.store(n + 1, std::memory_order_release);
    R->my_arenas.push_back({(unsigned long long)(uintptr_t)base, (unsigned long long)size, n});
    *id = (unsigned)n; *offset = a - (unsigned long long)(uintptr_t)base;
    return HPS_OK;
}

// sender: the receiver's allocation `id` as mapped into this process
int ipc_peer_arena (Ring* R, unsigned id, void** base)
{
    Mailbox* m = R->box_out.m;
    HPS_REQUIRE((int)id < m->n_arenas.load(std::memory_order_acquire), "hps_ring_send_slice: the receive descriptor names an allocation that was never exported");
    if (R->peer_arena.size() <= id) R->peer_arena.resize(id + 1, nullptr);
    if (!R->peer_arena[id]) HPS_HIP_CHECK(hipIpcOpenMemHandle(&R->peer_arena[id], m->arenas[id].handle, hipIpcMemLazyEnablePeerAccess));
    *base = R->peer_arena[id];
    return HPS_OK;
}

inline unsigned long long ld_acq (const volatile unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void st_rel (volatile unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

bool ipc_can_send (const Ring* R)
{
    const Mailbox* m = R->box_out.m;
    return ld_acq(&m->posted.v) > R->n_issued && ld_acq(&m->freed.v) > R->n_issued;
}

} // namespace

extern "C" int hps_ring_destroy (void* handle);
namespace hps {
bool ring_is_ticket (const void* p)
{
    std::lock_guard<std::mutex> lk(g_tickets_mu);
    return g_tickets.count(p) != 0;
}
}

extern "C" int hps_ring_unique_id (char* id_out)
{
    HPS_REQUIRE(id_out, "hps_ring_unique_id: null argument");
    if (edge_kind_env() == 1) {
        // ipc edge: the id is the name of the edge's mailbox, made (and mapped) here by the rank that will SEND over the edge
        static std::atomic<unsigned> counter{0};
        char name[96];
        std::snprintf(name, sizeof name, "/hps_ring_%d_%u_%llx", (int)getpid(), counter.fetch_add(1),
                      (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
        Box B;
        if (int e = box_map(B, name, true)) return e;
        g_pending_boxes[name] = B;
        std::memset(id_out, 0, HPS_RING_ID_BYTES);
        std::snprintf(id_out, HPS_RING_ID_BYTES, "%s%s", kIpcTag, name);
        return HPS_OK;
    }
    if (int e = load_rccl()) return e;
    NcclId id;
    HPS_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, id.internal, HPS_RING_ID_BYTES);
    return HPS_OK;
}

extern "C" int hps_ring_init (int rank, int world, int device, const char* id_edge_in, const char* id_edge_out, void** handle)
{
    HPS_REQUIRE(handle && world >= 1 && rank >= 0 && rank < world, "hps_ring_init: bad rank / world");
    HPS_REQUIRE(id_edge_out && (world == 1 || id_edge_in), "hps_ring_init: the ids of both edges are needed");
    if (is_ipc_id(id_edge_out)) {
        HPS_REQUIRE(world == 1 || is_ipc_id(id_edge_in), "hps_ring_init: this rank's outgoing edge is an ipc edge and its incoming edge is not (HPS_RING_EDGE differs between ranks)");
        HPS_HIP_CHECK(hipSetDevice(device));
        Ring* R = new Ring;
        R->rank = rank; R->world = world; R->device = device; R->kind = 1;
        auto fail = [&] (int e) { hps_ring_destroy(R); return e; };
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&R->st_recv, hipStreamNonBlocking, hi) != hipSuccess ||
            hipStreamCreateWithPriority(&R->st_send, hipStreamNonBlocking, hi) != hipSuccess) {
            hps::set_error("hps_ring_init: cannot create the ring's streams"); return fail(HPS_ERR_HIP);
        }
        const std::string out_name = id_edge_out + sizeof(kIpcTag) - 1;
        auto it = g_pending_boxes.find(out_name);
        if (it == g_pending_boxes.end()) { hps::set_error("hps_ring_init: id_edge_out was not made by hps_ring_unique_id of this process"); return fail(HPS_ERR_ARG); }
        R->box_out = it->second; g_pending_boxes.erase(it);
        if (world == 1) { *handle = R; return HPS_OK; }          // the rank is its own neighbour: plain device copies, no mailbox traffic
        if (int e = box_map(R->box_in, id_edge_in + sizeof(kIpcTag) - 1, false)) return fail(e);
        if (R->box_in.m->magic != kBoxMagic || R->box_in.m->version != kBoxVersion) { hps::set_error("hps_ring_init: the incoming edge's mailbox is not one of this library version"); return fail(HPS_ERR_COMM); }
        if (int e = box_register(R->box_in)) return fail(e);
        if (int e = box_register(R->box_out)) return fail(e);
        R->box_out.m->sender_attached.store(1);
        R->box_in.m->receiver_attached.store(1);
        const double tmo = timeout_env("HPS_RING_CONNECT_TIMEOUT_S", 300.0);
        if (int e = ipc_wait(R, [&] { return R->box_out.m->receiver_attached.load() != 0; }, nullptr, tmo, "hps_ring_init (waiting for the next rank to attach)")) return fail(e);
        if (int e = ipc_wait(R, [&] { return R->box_in.m->sender_attached.load() != 0; }, nullptr, tmo, "hps_ring_init (waiting for the previous rank to attach)")) return fail(e);
        shm_unlink(R->box_out.name.c_str());                      // both sides hold the mapping: nothing is left behind if a rank dies
        *handle = R;
        return HPS_OK;
    }
    HPS_REQUIRE(world == 1 || !is_ipc_id(id_edge_in), "hps_ring_init: this rank's incoming edge is an ipc edge and its outgoing edge is not (HPS_RING_EDGE differs between ranks)");
    if (int e = load_rccl()) return e;
    if (world > 1) {
        // A receive posted ahead is an RCCL kernel that sits on the device until its data comes.  If the runtime maps the
        // ring's receive stream, its send stream and the engine's stream onto one hardware queue, a send queued behind such
        // a receive never runs and the ring waits in a circle.  ROCm's default is 4 hardware queues per process; the ring
        // needs every stream of the process on its own queue: refuse to start with fewer than 8 (set before HIP starts).
        // The check can only read the environment: whether the value was there BEFORE the runtime started is the host's
        // business (hipace_amd/_lib.py sets it at import when WORLD_SIZE > 1, examples/pipeline_host.cpp before its first
        // HIP call).  HPS_RING_ALLOW_SHARED_QUEUES=1 overrides the refusal (a host that knows its queue assignment).
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        const char* ovr = std::getenv("HPS_RING_ALLOW_SHARED_QUEUES");
        if ((!q || std::atoi(q) < 8) && !(ovr && std::atoi(ovr) != 0)) {
            hps::set_error("hps_ring_init: set GPU_MAX_HW_QUEUES >= 8 in the environment before the process touches HIP (the ring's "
                           "posted-ahead receives must not share a hardware queue with its sends or the engine's stream); "
                           "HPS_RING_ALLOW_SHARED_QUEUES=1 overrides this check");
            return HPS_ERR_ARG;
        }
    }
    HPS_HIP_CHECK(hipSetDevice(device));
    Ring* R = new Ring;
    R->rank = rank; R->world = world; R->device = device;
    // both above the engine's stream: a hand-off is short and sits on the critical path of the next rank
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&R->st_recv, hipStreamNonBlocking, hi) != hipSuccess ||
        hipStreamCreateWithPriority(&R->st_send, hipStreamNonBlocking, hi) != hipSuccess) {
        hps_ring_destroy(R); hps::set_error("hps_ring_init: cannot create the ring's streams"); return HPS_ERR_HIP;
    }
    NcclId id;
    if (world == 1) {
        std::memcpy(id.internal, id_edge_out, HPS_RING_ID_BYTES);
        ncclResult_t r = g_rccl.CommInitRank(&R->comm_self, 1, id, 0);
        if (r != 0) { hps_ring_destroy(R); hps::set_error(std::string("hps_ring_init: ncclCommInitRank: ") + g_rccl.GetErrorString(r)); return HPS_ERR_COMM; }
    } else {
        const int e_out = rank, e_in = (rank + world - 1) % world;
        for (int colour = 0; colour < 3; ++colour) {
            ncclResult_t r = 0;
            if (edge_colour(e_out, world) == colour) {
                std::memcpy(id.internal, id_edge_out, HPS_RING_ID_BYTES);
                r = g_rccl.CommInitRank(&R->comm_out, 2, id, 0);
            }
            if (r == 0 && edge_colour(e_in, world) == colour) {
                std::memcpy(id.internal, id_edge_in, HPS_RING_ID_BYTES);
                r = g_rccl.CommInitRank(&R->comm_in, 2, id, 1);
            }
            if (r != 0) { hps_ring_destroy(R); hps::set_error(std::string("hps_ring_init: ncclCommInitRank: ") + g_rccl.GetErrorString(r)); return HPS_ERR_COMM; }
        }
    }
    // One message over every edge now, edge colour by edge colour.  RCCL connects two peers on their first send / receive
    // -- a rendezvous of the two hosts inside ncclSend / ncclRecv (the connect data travels over the bootstrap network) --
    // and a ring whose ranks make their first calls in pipeline order would wait in a circle: rank 0 posts its receive from
    // rank N-1 first, rank N-1 its receive from N-2, ..., rank 1 its receive from rank 0.  Colour by colour a rank has at
    // most one edge in play and its partner on that edge is in the same phase, so every first call finds its match; from
    // here on sends and receives only enqueue.  (It also takes the first launch of RCCL's kernel, which makes the runtime
    // set up that queue's scratch, out of the first hand-off.)
    {   const size_t nb = 1 << 16;
        char* tmp = nullptr;
        if (hipMalloc(&tmp, 2*nb) != hipSuccess) { hps_ring_destroy(R); hps::set_error("hps_ring_init: out of device memory"); return HPS_ERR_HIP; }
        (void)hipMemset(tmp, 0, 2*nb);
        bool ok = true;
        if (world == 1) {
            ok = ok && g_rccl.GroupStart() == 0;
            ok = ok && g_rccl.Send(tmp, nb, 0, 0, R->comm_self, R->st_send) == 0;
            ok = ok && g_rccl.Recv(tmp + nb, nb, 0, 0, R->comm_self, R->st_send) == 0;
            ok = ok && g_rccl.GroupEnd() == 0;
            ok = ok && hipStreamSynchronize(R->st_send) == hipSuccess;
        } else {
            const int e_out = rank, e_in = (rank + world - 1) % world;
            for (int colour = 0; colour < 3 && ok; ++colour) {
                if (edge_colour(e_out, world) == colour)
                    ok = ok && g_rccl.Send(tmp, nb, 0, 1, R->comm_out, R->st_send) == 0 && hipStreamSynchronize(R->st_send) == hipSuccess;
                if (edge_colour(e_in, world) == colour)
                    ok = ok && g_rccl.Recv(tmp + nb, nb, 0, 0, R->comm_in, R->st_recv) == 0 && hipStreamSynchronize(R->st_recv) == hipSuccess;
            }
        }
        (void)hipFree(tmp);
        if (!ok) { hps_ring_destroy(R); hps::set_error("hps_ring_init: the warm-up message over the ring's edges failed"); return HPS_ERR_COMM; }
    }
    *handle = R;
    return HPS_OK;
}

// put_data (MultiBuffer.cpp:444-493): `bytes` at msg_dev go to the next rank.  The send stream first waits for
// `after_event` (recorded by the engine behind the slice's last kernel; NULL = nothing to wait for); *done_event is
// recorded behind the send: the buffer may be overwritten once it has fired.
extern "C" int hps_ring_send_slice (void* handle, const void* msg_dev, long bytes, void* after_event, int slot, void** done_event)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R && (R->comm_out || (R->kind == 1 && R->world > 1)), "hps_ring_send_slice: the ring has no outgoing edge (world = 1: use hps_ring_sendrecv_self)");
    HPS_REQUIRE(msg_dev && bytes > 0, "hps_ring_send_slice: empty message");
    if (R->kind == 1) {
        // the host waits until the matching receive is posted and its buffer free (a host that must not block asks
        // hps_ring_can_send first); everything else is enqueued
        Mailbox* m = R->box_out.m;
        if (int e = ipc_wait(R, [&] { return ipc_can_send(R); }, &m->receiver_gone, timeout_env("HPS_RING_TIMEOUT_S", 900.0), "hps_ring_send_slice")) return e;
        const BoxDesc d = m->desc[R->n_issued % kBoxDescs];
        if (d.bytes != (unsigned long long)bytes) {
            char buf[160]; std::snprintf(buf, sizeof buf, "hps_ring_send_slice: message %llu has %ld bytes, the receive posted for it %llu", R->n_issued, bytes, d.bytes);
            hps::set_error(buf); return HPS_ERR_COMM;
        }
        void* base = nullptr;
        if (int e = ipc_peer_arena(R, d.arena, &base)) return e;
        if (after_event) HPS_HIP_CHECK(hipStreamWaitEvent(R->st_send, static_cast<hipEvent_t>(after_event), 0));
        HPS_HIP_CHECK(hipMemcpyAsync(static_cast<char*>(base) + d.offset, msg_dev, (size_t)bytes, hipMemcpyDeviceToDevice, R->st_send));
        HPS_HIP_CHECK(hipStreamWriteValue64(R->st_send, R->box_out.d_landed, R->n_issued + 1, 0));
        ++R->n_issued;
        st_rel(&m->issued.v, R->n_issued);
        hipEvent_t ev;
        if (int e = event_of(R->ev_send, slot, &ev)) return e;
        HPS_HIP_CHECK(hipEventRecord(ev, R->st_send));
        if (done_event) *done_event = ev;
        ++R->n_sent; R->bytes_sent += bytes;
        return HPS_OK;
    }
    if (after_event) HPS_HIP_CHECK(hipStreamWaitEvent(R->st_send, static_cast<hipEvent_t>(after_event), 0));
    HPS_NCCL_CHECK(g_rccl.Send(msg_dev, (size_t)bytes, /*ncclChar*/ 0, /*peer*/ 1, R->comm_out, R->st_send));
    hipEvent_t ev;
    if (int e = event_of(R->ev_send, slot, &ev)) return e;
    HPS_HIP_CHECK(hipEventRecord(ev, R->st_send));
    if (done_event) *done_event = ev;
    ++R->n_sent; R->bytes_sent += bytes;
    return HPS_OK;
}

// get_data (MultiBuffer.cpp:495-609): post the receive of the next message of the previous rank into msg_dev.  The
// receive stream first waits for `after_event` (the buffer's previous contents are no longer needed); *done_event
// fires when the data has landed -- make the engine's stream wait for it (hps_ring_engine_wait; on the RCCL edge it is a
// hipEvent_t and hps_engine_wait_event does the same).
extern "C" int hps_ring_recv_slice (void* handle, void* msg_dev, long bytes, void* after_event, int slot, void** done_event)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R && (R->comm_in || (R->kind == 1 && R->world > 1)), "hps_ring_recv_slice: the ring has no incoming edge (world = 1: use hps_ring_sendrecv_self)");
    HPS_REQUIRE(msg_dev && bytes > 0, "hps_ring_recv_slice: empty message");
    if (R->kind == 1) {
        Mailbox* m = R->box_in.m;
        HPS_REQUIRE(R->n_posted - ld_acq(&m->issued.v) < kBoxDescs, "hps_ring_recv_slice: too many receives posted ahead of their sends (65536)");
        HPS_REQUIRE(slot >= 0 && slot < (1 << 20), "hps_ring: bad event slot");
        BoxDesc d{};
        if (int e = ipc_arena_of(R, msg_dev, bytes, &d.arena, &d.offset)) return e;
        d.bytes = (unsigned long long)bytes;
        m->desc[R->n_posted % kBoxDescs] = d;
        ++R->n_posted;
        st_rel(&m->posted.v, R->n_posted);
        // "buffer free" travels in stream order behind the event that says so (and behind every earlier receive's), so the
        // word stays monotonic
        if (after_event) HPS_HIP_CHECK(hipStreamWaitEvent(R->st_recv, static_cast<hipEvent_t>(after_event), 0));
        HPS_HIP_CHECK(hipStreamWriteValue64(R->st_recv, R->box_in.d_freed, R->n_posted, 0));
        if ((size_t)slot >= R->tickets.size()) R->tickets.resize((size_t)slot + 1);
        if (!R->tickets[slot]) {
            R->tickets[slot].reset(new Ticket{kTicketMagic, 0});
            std::lock_guard<std::mutex> lk(g_tickets_mu);
            g_tickets.insert(R->tickets[slot].get());
        }
        R->tickets[slot]->seq = R->n_posted;
        if (done_event) *done_event = R->tickets[slot].get();
        ++R->n_received; R->bytes_received += bytes;
        return HPS_OK;
    }
    if (after_event) HPS_HIP_CHECK(hipStreamWaitEvent(R->st_recv, static_cast<hipEvent_t>(after_event), 0));
    HPS_NCCL_CHECK(g_rccl.Recv(msg_dev, (size_t)bytes, /*ncclChar*/ 0, /*peer*/ 0, R->comm_in, R->st_recv));
    hipEvent_t ev;
    if (int e = event_of(R->ev_recv, slot, &ev)) return e;
    HPS_HIP_CHECK(hipEventRecord(ev, R->st_recv));
    if (done_event) *done_event = ev;
    ++R->n_received; R->bytes_received += bytes;
    return HPS_OK;
}

// world = 1: the rank is its own neighbour (MultiBuffer.cpp:299-308).  Send and receive of one message as ONE RCCL
// group on the 1-rank communicator (a lone self-send would wait for its receive forever).
extern "C" int hps_ring_sendrecv_self (void* handle, const void* src_dev, void* dst_dev, long bytes, void* after_event, int slot,
                                       void** done_event)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R && (R->comm_self || (R->kind == 1 && R->world == 1)), "hps_ring_sendrecv_self: needs a ring of one rank");
    HPS_REQUIRE(src_dev && dst_dev && bytes > 0, "hps_ring_sendrecv_self: empty message");
    if (after_event) HPS_HIP_CHECK(hipStreamWaitEvent(R->st_send, static_cast<hipEvent_t>(after_event), 0));
    // HPS_RING_SELF_COPY=1 (diagnostic): the same choreography with a device copy in place of the RCCL pair -- what of the
    // ring's cost is the RCCL kernel running beside the engine's, and what is the events and the host calls
    static const bool plain_copy = [] { const char* v = std::getenv("HPS_RING_SELF_COPY"); return v && std::atoi(v) != 0; }();
    if (plain_copy || R->kind == 1) {     // (ipc edge with one rank: MultiBuffer.cpp:299-308's in-process copy, on the ring's stream)
        HPS_HIP_CHECK(hipMemcpyAsync(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice, R->st_send));
    } else {
        HPS_NCCL_CHECK(g_rccl.GroupStart());
        ncclResult_t r1 = g_rccl.Send(src_dev, (size_t)bytes, 0, 0, R->comm_self, R->st_send);
        ncclResult_t r2 = g_rccl.Recv(dst_dev, (size_t)bytes, 0, 0, R->comm_self, R->st_send);
        HPS_NCCL_CHECK(g_rccl.GroupEnd());
        HPS_NCCL_CHECK(r1); HPS_NCCL_CHECK(r2);
    }
    hipEvent_t ev;
    if (int e = event_of(R->ev_send, slot, &ev)) return e;
    HPS_HIP_CHECK(hipEventRecord(ev, R->st_send));
    if (done_event) *done_event = ev;
    ++R->n_sent; ++R->n_received; R->bytes_sent += bytes; R->bytes_received += bytes;
    return HPS_OK;
}

// make the ring's receive (which = 0) or send (which = 1) stream wait for an event of another stream, e.g. "the engine
// has finished the step that last used these receive buffers"
extern "C" int hps_ring_stream_wait (void* handle, int which, void* event)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R && event && (which == 0 || which == 1), "hps_ring_stream_wait: bad argument");
    HPS_HIP_CHECK(hipStreamWaitEvent(which == 0 ? R->st_recv : R->st_send, static_cast<hipEvent_t>(event), 0));
    return HPS_OK;
}

// host waits until every message handed to hps_ring_send_slice so far has left (the receives posted ahead stay posted)
// A ring that cannot make progress (a peer died, a receive posted ahead sits on the hardware queue its matching send is
// queued behind) must fail loudly instead of holding the node: the host waits for the ring's streams by polling, and gives
// up after `seconds` (HPS_RING_TIMEOUT_S in the environment for hps_ring_sync / hps_ring_sync_sends, default 900) with the
// rank's message counters in the error text.
static int ring_wait (Ring* R, bool sends, bool recvs, double seconds, const char* what)
{
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (true) {
        const hipError_t a = sends ? hipStreamQuery(R->st_send) : hipSuccess;
        const hipError_t b = recvs ? hipStreamQuery(R->st_recv) : hipSuccess;
        if (a == hipSuccess && b == hipSuccess) return HPS_OK;
        if ((a != hipSuccess && a != hipErrorNotReady) || (b != hipSuccess && b != hipErrorNotReady)) {
            hps::set_error(std::string(what) + ": " + hipGetErrorString(a != hipSuccess && a != hipErrorNotReady ? a : b));
            return HPS_ERR_HIP;
        }
        if ((++spins & 0x3ff) == 0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > seconds) {
                char buf[384];
                std::snprintf(buf, sizeof buf, "%s: rank %d of %d still waiting after %.0f s (%s%s pending); sent %ld messages / %lld bytes, "
                              "received %ld / %lld -- a peer is gone or the ring's streams share a hardware queue (GPU_MAX_HW_QUEUES)",
                              what, R->rank, R->world, dt, a == hipErrorNotReady ? "sends" : "", b == hipErrorNotReady ? " receives" : "",
                              R->n_sent, R->bytes_sent, R->n_received, R->bytes_received);
                hps::set_error(buf);
                return HPS_ERR_COMM;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
}
static double ring_timeout_env ()
{
    const char* e = std::getenv("HPS_RING_TIMEOUT_S");
    const double v = e ? std::atof(e) : 900.0;
    return v > 0.0 ? v : 900.0;
}

extern "C" int hps_ring_sync_sends (void* handle)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R, "hps_ring_sync_sends: null ring");
    return ring_wait(R, true, false, ring_timeout_env(), "hps_ring_sync_sends");
}

// ipc edge: "the receive stream is done" does not say that the posted receives have their data (no receive sits on the
// device); the end of a run also waits until every posted receive has landed -- the sender writes into this process's
// buffers until then
static int ipc_wait_all_landed (Ring* R, double seconds, const char* what)
{
    if (R->kind != 1 || R->world == 1 || !R->box_in.m) return HPS_OK;
    Mailbox* m = R->box_in.m;
    return ipc_wait(R, [&] { return ld_acq(&m->landed.v) >= R->n_posted; }, &m->sender_gone, seconds, what);
}

extern "C" int hps_ring_sync (void* handle)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R, "hps_ring_sync: null ring");
    if (int e = ring_wait(R, true, true, ring_timeout_env(), "hps_ring_sync")) return e;
    return ipc_wait_all_landed(R, ring_timeout_env(), "hps_ring_sync (receives posted and not yet sent by the previous rank)");
}

extern "C" int hps_ring_sync_timeout (void* handle, double seconds)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R && seconds > 0.0, "hps_ring_sync_timeout: bad argument");
    if (int e = ring_wait(R, true, true, seconds, "hps_ring_sync_timeout")) return e;
    return ipc_wait_all_landed(R, seconds, "hps_ring_sync_timeout (receives posted and not yet sent by the previous rank)");
}

// ---- the non-blocking side of the hand-off, and the wait that works for either kind of edge ---------------------------
extern "C" int hps_engine_wait_event (void* h, void* event);

extern "C" int hps_ring_edge_kind (void* handle)
{
    Ring* R = static_cast<Ring*>(handle);
    return R ? R->kind : -1;
}

// 1: the next hps_ring_send_slice will not make the host wait (RCCL edge: always -- the wait happens on the device)
extern "C" int hps_ring_can_send (void* handle)
{
    Ring* R = static_cast<Ring*>(handle);
    if (!R) return -1;
    if (R->kind != 1 || R->world == 1) return 1;
    return ipc_can_send(R) ? 1 : 0;
}

// 1: the message behind this receive (*done_event of hps_ring_recv_slice) has landed; 0: not yet; < 0: bad argument
extern "C" int hps_ring_recv_landed (void* handle, void* done_event)
{
    Ring* R = static_cast<Ring*>(handle);
    if (!R || !done_event) return -1;
    if (R->kind != 1) {
        const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(done_event));
        return e == hipSuccess ? 1 : (e == hipErrorNotReady ? 0 : -1);
    }
    const Ticket* t = static_cast<const Ticket*>(done_event);
    if (t->magic != kTicketMagic) return -1;
    return ld_acq(&R->box_in.m->landed.v) >= t->seq ? 1 : 0;
}

// Order `engine`'s stream behind a received message.  RCCL edge: the stream waits for the receive's event on the device
// (hps_engine_wait_event).  ipc edge: the HOST waits until the sender's stream has written "landed" (returns at once when it
// has, which is the steady state of a rank that trails its predecessor by two slices) -- whatever the engine is given
// afterwards runs after the payload is in place; gives up after HPS_RING_TIMEOUT_S.
extern "C" int hps_ring_engine_wait (void* handle, void* engine, void* done_event)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R && engine && done_event, "hps_ring_engine_wait: null argument");
    if (R->kind != 1) return hps_engine_wait_event(engine, done_event);
    const Ticket* t = static_cast<const Ticket*>(done_event);
    HPS_REQUIRE(t->magic == kTicketMagic, "hps_ring_engine_wait: not a receive of this ring");
    Mailbox* m = R->box_in.m;
    const unsigned long long seq = t->seq;
    return ipc_wait(R, [&] { return ld_acq(&m->landed.v) >= seq; }, &m->sender_gone, ring_timeout_env(), "hps_ring_engine_wait");
}

extern "C" int hps_ring_stats (void* handle, long* n_sent, long* n_received, long long* bytes_sent, long long* bytes_received)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R, "hps_ring_stats: null ring");
    if (n_sent) *n_sent = R->n_sent;
    if (n_received) *n_received = R->n_received;
    if (bytes_sent) *bytes_sent = R->bytes_sent;
    if (bytes_received) *bytes_received = R->bytes_received;
    return HPS_OK;
}

// What RCCL itself says about the ring's communicators: ranks of the incoming and outgoing edge's communicator (2 and 2 on
// a ring of 2+ ranks, 0 and 1 for the one-rank ring's self communicator; -1 where the library has no ncclCommCount) and
// this rank's place in each (1 on the incoming edge, 0 on the outgoing one).
extern "C" int hps_ring_info (void* handle, int* world, int* comm_in_ranks, int* comm_out_ranks, int* my_rank_in, int* my_rank_out)
{
    Ring* R = static_cast<Ring*>(handle);
    HPS_REQUIRE(R, "hps_ring_info: null ring");
    auto count = [](ncclComm_t c) { int n = -1; if (!c) return 0; if (g_rccl.CommCount && g_rccl.CommCount(c, &n) != 0) n = -1; return n; };
    auto urank = [](ncclComm_t c) { int n = -1; if (!c) return -1; if (g_rccl.CommUserRank && g_rccl.CommUserRank(c, &n) != 0) n = -1; return n; };
    if (R->kind == 1) {
        // ipc edge: 2 = both ends of the edge's mailbox are attached (0 and 1 for the one-rank ring, as the RCCL self ring)
        const bool multi = R->world > 1;
        if (world) *world = R->world;
        if (comm_in_ranks) *comm_in_ranks = multi ? (R->box_in.m && R->box_in.m->sender_attached.load() && R->box_in.m->receiver_attached.load() ? 2 : 1) : 0;
        if (comm_out_ranks) *comm_out_ranks = multi ? (R->box_out.m && R->box_out.m->sender_attached.load() && R->box_out.m->receiver_attached.load() ? 2 : 1) : 1;
        if (my_rank_in) *my_rank_in = multi ? 1 : -1;
        if (my_rank_out) *my_rank_out = 0;
        return HPS_OK;
    }
    if (world) *world = R->world;
    if (comm_in_ranks) *comm_in_ranks = count(R->comm_in);
    if (comm_out_ranks) *comm_out_ranks = count(R->world == 1 ? R->comm_self : R->comm_out);
    if (my_rank_in) *my_rank_in = urank(R->comm_in);
    if (my_rank_out) *my_rank_out = urank(R->world == 1 ? R->comm_self : R->comm_out);
    return HPS_OK;
}

extern "C" int hps_ring_destroy (void* handle)
{
    Ring* R = static_cast<Ring*>(handle);
    if (!R) return HPS_OK;
    (void)hipSetDevice(R->device);
    if (R->st_send) (void)hipStreamSynchronize(R->st_send);
    if (R->st_recv) (void)hipStreamSynchronize(R->st_recv);
    if (R->kind == 1) {
        // receiver: copies the previous rank has already enqueued into my buffers must have landed before the buffers go
        // (the host may free them right after this call): wait, bounded, for landed >= min(posted, issued) or for the sender
        // to have left
        if (R->box_in.m && R->world > 1) {
            Mailbox* m = R->box_in.m;
            const auto t0 = std::chrono::steady_clock::now();
            while (!m->sender_gone.load()) {
                const unsigned long long want = std::min<unsigned long long>(R->n_posted, ld_acq(&m->issued.v));
                if (ld_acq(&m->landed.v) >= want) break;
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) break;
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        {   std::lock_guard<std::mutex> lk(g_tickets_mu);
            for (auto& t : R->tickets) if (t) g_tickets.erase(t.get()); }
        if (R->box_out.m && R->world > 1) R->box_out.m->sender_gone.store(1);
        if (R->box_in.m) R->box_in.m->receiver_gone.store(1);
        for (void* p : R->peer_arena) if (p) (void)hipIpcCloseMemHandle(p);
        box_unmap(R->box_in);
        box_unmap(R->box_out);
    }
    if (R->comm_out) g_rccl.CommDestroy(R->comm_out);
    if (R->comm_in) g_rccl.CommDestroy(R->comm_in);
    if (R->comm_self) g_rccl.CommDestroy(R->comm_self);
    for (auto e : R->ev_recv) if (e) (void)hipEventDestroy(e);
    for (auto e : R->ev_send) if (e) (void)hipEventDestroy(e);
    if (R->st_send) (void)hipStreamDestroy(R->st_send);
    if (R->st_recv) (void)hipStreamDestroy(R->st_recv);
    delete R;
    return HPS_OK;
}
