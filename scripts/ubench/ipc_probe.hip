// micro-benchmark / capability probe: what does a between-PROCESS hand-off on one device cost on this runtime, and which
// of the building blocks work?  Two processes (fork before the first HIP call), both on device 0:
//   receiver: hipMalloc arena, hipIpcGetMemHandle of the base and of an interior pointer, posts both into a shared page
//   sender:   hipIpcOpenMemHandle, hipMemcpyAsync device -> opened pointer, then an 8-byte device -> host copy of the
//             message's sequence number into a flag word in POSIX shared memory that both processes hipHostRegister'ed
//   receiver: polls the flag word on the host, then checks the payload on the device
// Also: hipLaunchHostFunc as the flag writer, hipStreamWriteValue64 on the registered page, IPC events
// (hipEventInterprocess + hipIpcGetEventHandle), and a compute kernel of the receiver running beside the copies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <sys/mman.h>
#include <sys/wait.h>
#include <fcntl.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%s] %s failed: %s\n", who, #x, hipGetErrorString(e_)); fflush(stdout); ok = false; } } while (0)

struct Shared {
    std::atomic<int> stage;                 // rendezvous counter
    hipIpcMemHandle_t h_base, h_inner, h_small;
    hipIpcEventHandle_t h_event;
    std::atomic<int> have_event;
    std::atomic<long long> t_issue_ns;      // sender: host time when the message was enqueued
    std::atomic<int> go;
    std::atomic<int> cnt;
    char pad[256];
};
struct Flags { volatile unsigned long long seq[64]; };       // the page both sides hipHostRegister

static long long now_ns () { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void wait_stage (Shared* S, int v) { while (S->stage.load() < v) usleep(50); }

__global__ void k_fill (double* p, long n, double v) { long i = blockIdx.x*(long)blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void k_check (const double* p, long n, double v, int* bad) { long i = blockIdx.x*(long)blockDim.x + threadIdx.x; if (i < n && p[i] != v + i) atomicAdd(bad, 1); }
__global__ void k_spin (long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
// test D: do kernels of the two processes run side by side on the device?  Each process launches R one-workgroup kernels of
// 100 us on one stream at the same time as the other.
static void two_process_concurrency (Shared* S, const char* who, hipStream_t st, int stage_in)
{
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    const long long ticks = (long long)rate*100/1000;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks/10); (void)hipStreamSynchronize(st);
    for (int both = 0; both < 2; ++both) {
        const bool mine = both == 1 || who[0] == 'r';
        S->cnt.fetch_add(1); while (S->cnt.load() < 2*(both + 1)) {}
        const long long t0 = now_ns();
        if (mine) { for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks); (void)hipStreamSynchronize(st); }
        const double us = 1e-3*(now_ns() - t0);
        if (mine) printf("[%s] 100 kernels of 100 us, %s: %.0f us\n", who, both ? "BOTH processes launching" : "this process alone", us);
        fflush(stdout);
    }
}
__global__ void k_busy (double* p, long n, int reps) { long i = blockIdx.x*(long)blockDim.x + threadIdx.x; if (i >= n) return; double a = p[i]; for (int r = 0; r < reps; ++r) a = a*1.0000001 + 1e-9; p[i] = a; }

int main ()
{
    Shared* S = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    new (S) Shared; S->stage = 0; S->have_event = 0; S->go = 0; S->cnt = 0;
    char name[64]; snprintf(name, sizeof name, "/hps_ipc_probe_%d", (int)getpid());
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) { printf("shm_open failed\n"); return 1; }
    Flags* F = (Flags*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    memset((void*)F, 0, 4096);
    const size_t MB = 1 << 20;
    const size_t arena = 128*MB, inner_off = 3*MB + 256;
    const pid_t child = fork();
    const bool recv = child != 0;
    const char* who = recv ? "recv" : "send";
    bool ok = true;
    CK(hipSetDevice(0));
    CK(hipHostRegister((void*)F, 4096, hipHostRegisterMapped));
    printf("[%s] hipHostRegister of the shared page: %s\n", who, ok ? "ok" : "FAILED"); fflush(stdout);
    unsigned long long* F_dev = nullptr;
    CK(hipHostGetDevicePointer((void**)&F_dev, (void*)F, 0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (recv) {
        char* base = nullptr; char* small = nullptr;
        CK(hipMalloc(&base, arena)); CK(hipMalloc(&small, 4096));
        CK(hipMemset(base, 0, arena));
        CK(hipIpcGetMemHandle(&S->h_base, base));
        bool ok_base = ok;
        CK(hipIpcGetMemHandle(&S->h_inner, base + inner_off));
        printf("[recv] handle of base %s, of interior pointer %s; handles equal: %d\n", ok_base ? "ok" : "FAILED", ok ? "ok" : "FAILED",
               (int)!memcmp(&S->h_base, &S->h_inner, sizeof(hipIpcMemHandle_t)));
        CK(hipIpcGetMemHandle(&S->h_small, small));
        hipEvent_t ev; bool evok = true;
        { bool ok2 = ok; ok = true; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess)); CK(hipIpcGetEventHandle(&S->h_event, ev)); evok = ok; ok = ok2; }
        S->have_event = evok ? 1 : -1;
        printf("[recv] interprocess event + handle: %s\n", evok ? "ok" : "FAILED"); fflush(stdout);
        S->stage = 1;
        // ---- test A: messages of several sizes, latency from the sender's enqueue to the flag seen here
        int* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        unsigned long long seq = 0;
        for (size_t bytes : {MB, 33*MB}) for (int mode = 0; mode < 3; ++mode) {
            const int reps = 20;
            double lat_sum = 0, lat_max = 0;
            for (int r = 0; r < reps; ++r) {
                ++seq;
                S->go = (int)seq;                                   // "receive posted, buffer free"
                long spins = 0;
                while (F->seq[0] < seq) { if (++spins > 2000000000L) break; }
                const long long t = now_ns();
                const double lat = 1e-3*(t - S->t_issue_ns.load());
                lat_sum += lat; if (lat > lat_max) lat_max = lat;
                const long n = (long)(bytes/8);
                hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, (const double*)(base + inner_off), n, (double)seq, bad);
                CK(hipStreamSynchronize(st));
            }
            int hbad = -1; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            printf("[recv] %3zu MB, flag by %-22s: enqueue -> flag seen %7.1f us mean, %7.1f max; payload mismatches so far %d\n", bytes/MB,
                   mode == 0 ? "8-byte D2H copy" : mode == 1 ? "hipLaunchHostFunc" : "hipStreamWriteValue64", lat_sum/reps, lat_max, hbad);
            fflush(stdout);
        }
        // ---- test B: the same 1 MB messages while this process keeps the device busy with a bandwidth kernel
        {   double* w; const long nw = 64*MB/8; CK(hipMalloc(&w, nw*8)); CK(hipMemset(w, 0, nw*8));
            hipStream_t sb; CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
            double lat_sum = 0, lat_max = 0; const int reps = 50;
            for (int r = 0; r < reps; ++r) {
                for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_busy, dim3((unsigned)(nw/256)), dim3(256), 0, sb, w, nw, 64);
                ++seq; S->go = (int)seq;
                long spins = 0; while (F->seq[0] < seq) { if (++spins > 2000000000L) break; }
                const double lat = 1e-3*(now_ns() - S->t_issue_ns.load());
                lat_sum += lat; if (lat > lat_max) lat_max = lat;
                CK(hipStreamSynchronize(sb));
            }
            printf("[recv] 1 MB beside a busy device (D2H flag): enqueue -> flag seen %7.1f us mean, %7.1f max\n", lat_sum/reps, lat_max); fflush(stdout);
        }
        // ---- test C: interprocess event recorded by the sender behind a copy; we wait for it on our stream
        if (evok) {
            ++seq; S->go = (int)seq;
            while (S->stage.load() < 2) usleep(20);                 // the sender has called hipEventRecord
            const long long t0 = now_ns();
            CK(hipStreamWaitEvent(st, ev, 0));
            const long n = (long)(MB/8);
            hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, (const double*)(base + inner_off), n, (double)seq, bad);
            CK(hipStreamSynchronize(st));
            int hbad = -1; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            printf("[recv] interprocess event: wait + check took %.1f us, payload mismatches so far %d (0 = the wait ordered the copy)\n", 1e-3*(now_ns() - t0), hbad);
            fflush(stdout);
        }
        two_process_concurrency(S, who, st, 2 + (evok ? 0 : 0));
        S->go = -1;
        int status = 0; waitpid(child, &status, 0);
        printf("[recv] sender exit status %d; all calls ok: %d\n", WEXITSTATUS(status), (int)ok);
        shm_unlink(name);
        return ok ? 0 : 1;
    }
    // ------------------------------------------------------------------ sender
    wait_stage(S, 1);
    char* rbase = nullptr; char* rinner = nullptr; char* rsmall = nullptr;
    CK(hipIpcOpenMemHandle((void**)&rbase, S->h_base, hipIpcMemLazyEnablePeerAccess));
    {   bool ok2 = ok; ok = true;
        CK(hipIpcOpenMemHandle((void**)&rinner, S->h_inner, hipIpcMemLazyEnablePeerAccess));
        printf("[send] open base -> %p, open interior handle -> %p (%s; difference %ld, expected %zu or a separate mapping)\n", (void*)rbase, (void*)rinner,
               ok ? "ok" : "FAILED", rinner && rbase ? (long)(rinner - rbase) : -1L, inner_off);
        if (!ok) rinner = rbase + inner_off;
        ok = true;
        CK(hipIpcOpenMemHandle((void**)&rsmall, S->h_small, hipIpcMemLazyEnablePeerAccess));
        printf("[send] open handle of a 4 KB allocation: %s\n", ok ? "ok" : "FAILED");
        ok = ok2; fflush(stdout);
    }
    char* dst = rbase + inner_off;                                   // always address through the base mapping
    double* src; CK(hipMalloc(&src, 33*MB));
    unsigned long long* seqs; CK(hipMalloc(&seqs, 8*4096));          // device table of sequence numbers: seqs[k] = k
    {   unsigned long long h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i; CK(hipMemcpy(seqs, h, sizeof h, hipMemcpyHostToDevice)); }
    hipEvent_t ev = nullptr; bool evok = false;
    while (S->have_event.load() == 0) usleep(20);
    if (S->have_event.load() > 0) { bool ok2 = ok; ok = true; CK(hipIpcOpenEventHandle(&ev, S->h_event)); evok = ok; ok = ok2; printf("[send] open event handle: %s\n", evok ? "ok" : "FAILED"); fflush(stdout); }
    unsigned long long seq = 0;
    auto wait_go = [&] (unsigned long long s) { while (S->go.load() != (int)s) { if (S->go.load() < 0) return false; } return true; };
    struct HF { Flags* F; unsigned long long s; };
    static HF hf[256];
    for (size_t bytes : {MB, 33*MB}) for (int mode = 0; mode < 3; ++mode) for (int r = 0; r < 20; ++r) {
        ++seq;
        const long n = (long)(bytes/8);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, src, n, (double)seq);
        CK(hipStreamSynchronize(st));
        if (!wait_go(seq)) return 2;
        S->t_issue_ns = now_ns();
        CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        if (mode == 0) CK(hipMemcpyAsync((void*)&F->seq[0], seqs + seq, 8, hipMemcpyDeviceToHost, st));
        else if (mode == 1) { hf[seq % 256] = HF{F, seq}; CK(hipLaunchHostFunc(st, [] (void* p) { HF* h = (HF*)p; h->F->seq[0] = h->s; }, &hf[seq % 256])); }
        else CK(hipStreamWriteValue64(st, (void*)F_dev, seq, 0));
        if (!ok) { F->seq[0] = seq; }                                 // keep the receiver going whatever failed
        CK(hipStreamSynchronize(st));
    }
    for (int r = 0; r < 50; ++r) {
        ++seq;
        const long n = (long)(MB/8);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, src, n, (double)seq);
        CK(hipStreamSynchronize(st));
        if (!wait_go(seq)) return 2;
        S->t_issue_ns = now_ns();
        CK(hipMemcpyAsync(dst, src, MB, hipMemcpyDeviceToDevice, st));
        CK(hipMemcpyAsync((void*)&F->seq[0], seqs + seq, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
    }
    if (evok) {
        ++seq;
        const long n = (long)(MB/8);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, src, n, (double)seq);
        CK(hipStreamSynchronize(st));
        if (!wait_go(seq)) return 2;
        // a slow kernel in front of the copy, so that a wait that does not order anything is caught
        hipLaunchKernelGGL(k_busy, dim3((unsigned)(MB/256)), dim3(256), 0, st, src + 2*MB, (long)MB, 20000);
        CK(hipMemcpyAsync(dst, src, MB, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(ev, st));
        S->stage = 2;
        CK(hipStreamSynchronize(st));
    }
    two_process_concurrency(S, who, st, 2);
    while (S->go.load() >= 0) usleep(100);
    printf("[send] all calls ok: %d\n", (int)ok); fflush(stdout);
    return ok ? 0 : 1;
}
