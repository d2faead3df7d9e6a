// Inter-GPU exchange of the row-partitioned hierarchy: see comm.hpp.
#include "comm.hpp"
#include "amg_dev.hpp"
#include "../../include/amgcore_hip.h"

#include <dlfcn.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace amg {

#define CHK(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

// ------------------------------------------------------------------ peer transport kernels
struct PeerPtrs {
    int n;                                  // ranks
    double *data[COMM_MAX_RANKS];           // push: (me -> p) staging in p's arena; unpack: (p -> me) staging in mine
    unsigned long long *flag[COMM_MAX_RANKS];   // signal: my flag in p's arena; wait: p's flag in mine
    int start[COMM_MAX_RANKS + 1];          // prefix of the per-peer counts
    unsigned partner;                       // bit p: data travels between this rank and p in EITHER direction on this channel
};

__global__ __launch_bounds__(256) void comm_push_kernel(PeerPtrs P, const double *v, const int *send_idx,
                                                        const unsigned long long *seq_send)
{
    const unsigned long long s = *seq_send + 1;              // the exchange being produced
    const int total = P.start[P.n];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int p = 0;
        while (i >= P.start[p + 1]) ++p;
        const int cnt = P.start[p + 1] - P.start[p];
        const double val = send_idx ? v[send_idx[i]] : v[i];
        P.data[p][(size_t)(s & 1) * cnt + (i - P.start[p])] = val;
    }
}

__global__ void comm_signal_kernel(PeerPtrs P, unsigned long long *seq_send)
{
    const unsigned long long s = *seq_send + 1;
    const int p = threadIdx.x;
    // every PARTNER is signalled, also one that only sends to this rank: its next push into the staging slot of the same
    // parity must not start before this rank has unpacked the previous one, and the only thing that holds it back is
    // its own wait for this flag (one-directional couplings: ADVICE r2)
    if (p < P.n && ((P.partner >> p) & 1u))
        __hip_atomic_store(P.flag[p], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (p == 0) *seq_send = s;
}

__global__ void comm_wait_kernel(PeerPtrs P, unsigned long long *seq_recv, int *timeout_flag, long long budget_ticks)
{
    const unsigned long long s = *seq_recv + 1;
    const int p = threadIdx.x;
    if (p < P.n && ((P.partner >> p) & 1u)) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(P.flag[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > budget_ticks) { *timeout_flag = 1; break; }     // every wave leaves the loop
        }
    }
    __syncthreads();
    if (p == 0) *seq_recv = s;
}

__global__ __launch_bounds__(256) void comm_unpack_kernel(PeerPtrs P, double *dst, const unsigned long long *seq_recv)
{
    const unsigned long long s = *seq_recv;                  // already advanced by the wait kernel
    const int total = P.start[P.n];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int p = 0;
        while (i >= P.start[p + 1]) ++p;
        const int cnt = P.start[p + 1] - P.start[p];
        dst[i] = P.data[p][(size_t)(s & 1) * cnt + (i - P.start[p])];
    }
}

// Fused hand-off (two launches per exchange instead of four): the push kernel's LAST workgroup to finish raises the
// flags -- every workgroup makes its stores visible system-wide (release fence), then takes a ticket; the one that draws the
// last ticket has thereby observed all the others' (acquire fence after the ticket) and stores the flags with system-scope
// release.  On the consumer every workgroup of the unpack kernel polls the flags itself (one lane per partner, system-scope
// acquire loads bounded by the wall-clock budget; the lane's acquire also drops this CU's stale lines), the workgroup
// barrier hands that on to its other waves, then it copies its share; the last workgroup to finish advances the
// sequence number (all of them read it when they start).  Tickets return to zero, so a captured graph replays.
__global__ __launch_bounds__(256) void comm_push_signal_kernel(PeerPtrs P, const double *v, const int *send_idx,
                                                               unsigned long long *seq_send, unsigned *ticket)
{
    __shared__ unsigned last;
    const unsigned long long s = *seq_send + 1;
    const int total = P.start[P.n];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int p = 0;
        while (i >= P.start[p + 1]) ++p;
        const int cnt = P.start[p + 1] - P.start[p];
        const double val = send_idx ? v[send_idx[i]] : v[i];
        P.data[p][(size_t)(s & 1) * cnt + (i - P.start[p])] = val;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence_system();
    const int p = threadIdx.x;
    if (p < P.n && ((P.partner >> p) & 1u))
        __hip_atomic_store(P.flag[p], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *seq_send = s;
    }
}

__global__ __launch_bounds__(256) void comm_wait_unpack_kernel(PeerPtrs P, double *dst, unsigned long long *seq_recv, unsigned *ticket,
                                                               int *timeout_flag, long long budget_ticks)
{
    const unsigned long long s = *seq_recv + 1;
    const int p = threadIdx.x;
    if (p < P.n && ((P.partner >> p) & 1u)) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(P.flag[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > budget_ticks) { *timeout_flag = 1; break; }     // every wave leaves the loop
        }
    }
    __syncthreads();
    const int total = P.start[P.n];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int q = 0;
        while (i >= P.start[q + 1]) ++q;
        const int cnt = P.start[q + 1] - P.start[q];
        dst[i] = P.data[q][(size_t)(s & 1) * cnt + (i - P.start[q])];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *seq_recv = s;
        }
    }
}

// all-reduce of one double: every rank stores its partial into every rank's arena (its own included) and raises
// the flag from the same lane (release order); the consumer adds the partials in RANK order
__global__ void comm_reduce_push_kernel(PeerPtrs P, const double *partial, unsigned long long *seq_send)
{
    const unsigned long long s = *seq_send + 1;
    const int p = threadIdx.x;
    if (p < P.n) {
        P.data[p][s & 1] = *partial;
        __threadfence_system();
        __hip_atomic_store(P.flag[p], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (p == 0) *seq_send = s;
}

__global__ void comm_reduce_wait_kernel(PeerPtrs P, double *result, unsigned long long *seq_recv, int *timeout_flag,
                                        long long budget_ticks)
{
    const unsigned long long s = *seq_recv + 1;
    const int p = threadIdx.x;
    __shared__ double part[COMM_MAX_RANKS];
    if (p < P.n) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(P.flag[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > budget_ticks) { *timeout_flag = 1; break; }
        }
        part[p] = __hip_atomic_load(&P.data[p][s & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (p == 0) {
        double sum = 0.0;
        for (int q = 0; q < P.n; ++q) sum += part[q];
        *result = sqrt(sum);
        *seq_recv = s;
    }
}

__global__ void comm_sqrt_kernel(double *v) { *v = sqrt(*v); }

__global__ __launch_bounds__(256) void comm_pack_kernel(double *out, const double *v, const int *send_idx, int total)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        out[i] = send_idx ? v[send_idx[i]] : v[i];
}

static inline int copy_grid(int total)
{
    int g = (total + 255) / 256;
    return g < 1 ? 1 : (g > 512 ? 512 : g);
}

static int wait_grid_cap()
{
    static const int g = std::getenv("AMG_COMM_WAIT_WGS") ? std::atoi(std::getenv("AMG_COMM_WAIT_WGS")) : 32;
    return g < 1 ? 1 : g;
}

// AMG_COMM_FUSED=1: the two-launch hand-off (push + signal, wait + unpack).  Opt-in: on the one-device rehearsal (two ranks
// sharing the GPU at 500^3) it measured SLOWER than the four-launch hand-off -- 20.9 vs 18.7 ms per step (25.9 with 512
// spinning workgroups): every workgroup's system-scope release fence and the spinning unpack workgroups compete with the
// peer process for the same device -- and it cannot be measured on two physical GPUs from here.
static bool comm_fused()
{
    static const int f = std::getenv("AMG_COMM_FUSED") ? std::atoi(std::getenv("AMG_COMM_FUSED")) : 0;
    return f != 0;
}

static int launch_ok(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return 0;
}

static unsigned partner_mask(const amg_comm *c, const Channel &C)
{
    unsigned m = 0;
    for (int p = 0; p < c->world; ++p)
        if (C.send_start[p + 1] > C.send_start[p] || C.recv_start[p + 1] > C.recv_start[p]) m |= 1u << p;
    return m;
}

static PeerPtrs producer_ptrs(amg_comm *c, int chn)
{
    const Channel &C = c->ch[(size_t)chn];
    PeerPtrs P;
    std::memset(&P, 0, sizeof(P));
    P.n = c->world;
    for (int p = 0; p < c->world; ++p) {
        P.data[p] = reinterpret_cast<double *>(c->peer[(size_t)p] + C.stage_off[(size_t)p * c->world + c->rank]);
        P.flag[p] = reinterpret_cast<unsigned long long *>(c->peer[(size_t)p]) + (size_t)chn * c->world + c->rank;
        P.start[p] = C.send_start[p];
    }
    P.start[c->world] = C.send_start[c->world];
    P.partner = partner_mask(c, C);
    return P;
}

static PeerPtrs consumer_ptrs(amg_comm *c, int chn)
{
    const Channel &C = c->ch[(size_t)chn];
    PeerPtrs P;
    std::memset(&P, 0, sizeof(P));
    P.n = c->world;
    for (int p = 0; p < c->world; ++p) {
        P.data[p] = reinterpret_cast<double *>(c->arena + C.stage_off[(size_t)c->rank * c->world + p]);
        P.flag[p] = reinterpret_cast<unsigned long long *>(c->arena) + (size_t)chn * c->world + p;
        P.start[p] = C.recv_start[p];
    }
    P.start[c->world] = C.recv_start[c->world];
    P.partner = partner_mask(c, C);
    return P;
}

// ------------------------------------------------------------------ rccl transport (dlopen)
struct NcclApi {
    int (*GetUniqueId)(void *);
    int (*CommInitRank)(void **, int, struct NcclId, int);
    int (*CommDestroy)(void *);
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t);
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t);
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char *(*GetErrorString)(int);
};
struct NcclId { char internal[128]; };       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;
static NcclApi g_nccl;
static void *g_nccl_handle = nullptr;

static int load_nccl(const char *libpath)
{
    if (g_nccl_handle) return 0;
    void *h = dlopen((libpath && *libpath) ? libpath : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error(std::string("cannot load RCCL: ") + dlerror()); return AMG_ENODEV; }
#define SYM(field, name)                                                             \
    *(void **)(&g_nccl.field) = dlsym(h, name);                                      \
    if (!g_nccl.field) { set_error(std::string("RCCL symbol missing: ") + name); return AMG_ENODEV; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_nccl_handle = h;
    return 0;
}

static int nccl_fail(int rc, const char *what)
{
    set_error(std::string("RCCL error in ") + what + ": " + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "?"));
    return AMG_ENODEV;
}
#define NCCL_CHK(call, what)                        \
    do {                                            \
        int r__ = (call);                           \
        if (r__ != 0) return nccl_fail(r__, what);  \
    } while (0)

// ------------------------------------------------------------------ channel operations
int comm_exchange_begin(amg_comm *c, int chn, const double *v, const int *send_idx, double *dst_halo, hipStream_t st)
{
    if (!c || chn < 0 || chn >= (int)c->ch.size() || !c->connected) { set_error("exchange on an unconnected communicator"); return AMG_ESTATE; }
    Channel &C = c->ch[(size_t)chn];
    if (c->transport == 1) {
        if (C.send_total)
            hipLaunchKernelGGL(comm_pack_kernel, dim3(copy_grid(C.send_total)), dim3(256), 0, st, C.rccl_sendbuf, v, send_idx, C.send_total);
        CHK(launch_ok("comm pack"));
        NCCL_CHK(g_nccl.GroupStart(), "ncclGroupStart");
        for (int p = 0; p < c->world; ++p) {
            const int ns = C.send_start[p + 1] - C.send_start[p], nr = C.recv_start[p + 1] - C.recv_start[p];
            if (ns) NCCL_CHK(g_nccl.Send(C.rccl_sendbuf + C.send_start[p], (size_t)ns, NCCL_FLOAT64, p, c->nccl_comm, st), "ncclSend");
            if (nr) NCCL_CHK(g_nccl.Recv(dst_halo + C.recv_start[p], (size_t)nr, NCCL_FLOAT64, p, c->nccl_comm, st), "ncclRecv");
        }
        NCCL_CHK(g_nccl.GroupEnd(), "ncclGroupEnd");
        return 0;
    }
    (void)dst_halo;
    PeerPtrs P = producer_ptrs(c, chn);
    unsigned long long *seq_send = c->seq + 2 * (size_t)chn;
    if (comm_fused()) {
        hipLaunchKernelGGL(comm_push_signal_kernel, dim3(copy_grid(C.send_total)), dim3(256), 0, st, P, v, send_idx, seq_send,
                           c->ticket + 2 * (size_t)chn);
        return launch_ok("comm push+signal");
    }
    if (C.send_total)
        hipLaunchKernelGGL(comm_push_kernel, dim3(copy_grid(C.send_total)), dim3(256), 0, st, P, v, send_idx, seq_send);
    hipLaunchKernelGGL(comm_signal_kernel, dim3(1), dim3(64), 0, st, P, seq_send);
    return launch_ok("comm push/signal");
}

int comm_exchange_end(amg_comm *c, int chn, double *dst_halo, hipStream_t st)
{
    if (!c || chn < 0 || chn >= (int)c->ch.size() || !c->connected) { set_error("exchange on an unconnected communicator"); return AMG_ESTATE; }
    if (c->transport == 1) return 0;                        // the grouped receive already targets dst_halo
    Channel &C = c->ch[(size_t)chn];
    PeerPtrs P = consumer_ptrs(c, chn);
    unsigned long long *seq_recv = c->seq + 2 * (size_t)chn + 1;
    if (comm_fused()) {
        // at most 32 workgroups: every one of them spins until the peers' flags arrive, and when ranks share a device
        // (tests, rehearsals) the peer's push kernel needs compute units of its own to get there
        hipLaunchKernelGGL(comm_wait_unpack_kernel, dim3(std::min(copy_grid(C.recv_total), wait_grid_cap())), dim3(256), 0, st, P, dst_halo, seq_recv,
                           c->ticket + 2 * (size_t)chn + 1, c->timeout_flag, c->budget_ticks);
        return launch_ok("comm wait+unpack");
    }
    hipLaunchKernelGGL(comm_wait_kernel, dim3(1), dim3(64), 0, st, P, seq_recv, c->timeout_flag, c->budget_ticks);
    if (C.recv_total)
        hipLaunchKernelGGL(comm_unpack_kernel, dim3(copy_grid(C.recv_total)), dim3(256), 0, st, P, dst_halo, seq_recv);
    return launch_ok("comm wait/unpack");
}

int comm_allreduce_sqrt(amg_comm *c, int chn, const double *partial, double *result, hipStream_t st)
{
    if (!c || chn < 0 || chn >= (int)c->ch.size() || !c->connected) { set_error("all-reduce on an unconnected communicator"); return AMG_ESTATE; }
    if (c->transport == 1) {
        NCCL_CHK(g_nccl.AllReduce(partial, result, 1, NCCL_FLOAT64, NCCL_SUM, c->nccl_comm, st), "ncclAllReduce");
        hipLaunchKernelGGL(comm_sqrt_kernel, dim3(1), dim3(1), 0, st, result);
        return launch_ok("comm sqrt");
    }
    PeerPtrs Pp = producer_ptrs(c, chn), Pc = consumer_ptrs(c, chn);
    hipLaunchKernelGGL(comm_reduce_push_kernel, dim3(1), dim3(64), 0, st, Pp, partial, c->seq + 2 * (size_t)chn);
    hipLaunchKernelGGL(comm_reduce_wait_kernel, dim3(1), dim3(64), 0, st, Pc, result, c->seq + 2 * (size_t)chn + 1, c->timeout_flag,
                       c->budget_ticks);
    return launch_ok("comm all-reduce");
}

int comm_check(amg_comm *c)
{
    if (!c || c->transport != 0 || !c->timeout_flag) return 0;
    int flag = 0;
    AMG_HIP(hipMemcpy(&flag, c->timeout_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) { set_error("inter-GPU exchange timed out waiting for a peer (a rank died or fell out of step)"); return AMG_ESTATE; }
    return 0;
}

}  // namespace amg

using namespace amg;

extern "C" {

amg_comm *amg_comm_create(int rank, int world, int device, int transport)
{
    if (world < 1 || world > COMM_MAX_RANKS || rank < 0 || rank >= world || (transport != 0 && transport != 1)) {
        set_error("bad communicator arguments (at most 16 ranks)");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        set_error("no such HIP device");
        return nullptr;
    }
    amg_comm *c = new amg_comm();
    c->rank = rank; c->world = world; c->device = device; c->transport = transport;
    return c;
}

/* counts[dst * world + src] = doubles rank dst receives from rank src in one exchange on this channel (the same
 * matrix on every rank).  Returns the channel id. */
int amg_comm_add_channel(amg_comm *c, const int *counts)
{
    if (!c || !counts || c->committed) { set_error("channels must be declared before commit"); return AMG_ESTATE; }
    Channel C;
    const int W = c->world;
    C.counts.assign(counts, counts + (size_t)W * W);
    for (int v : C.counts) if (v < 0) { set_error("negative count"); return AMG_EINVAL; }
    C.send_start[0] = C.recv_start[0] = 0;
    for (int p = 0; p < W; ++p) {
        C.send_start[p + 1] = C.send_start[p] + counts[(size_t)p * W + c->rank];
        C.recv_start[p + 1] = C.recv_start[p] + counts[(size_t)c->rank * W + p];
    }
    C.send_total = C.send_start[W]; C.recv_total = C.recv_start[W];
    c->ch.push_back(C);
    return (int)c->ch.size() - 1;
}

/* Fix the layout, allocate this rank's arena (and the sequence counters).  peer transport: handle_out receives the
 * 64-byte IPC handle every other rank needs for amg_comm_connect. */
int amg_comm_commit(amg_comm *c, unsigned char *handle_out)
{
    if (!c || c->committed) { set_error("communicator already committed"); return AMG_ESTATE; }
    AMG_HIP(hipSetDevice(c->device));
    const int W = c->world;
    const size_t nch = c->ch.size();
    c->flag_bytes = ((nch * W * sizeof(unsigned long long)) + 4095) / 4096 * 4096 + 4096;
    // every rank computes every rank's layout: (dst, src) staging = 2 slots of counts[dst][src] doubles
    std::vector<size_t> used((size_t)W, c->flag_bytes);
    for (auto &C : c->ch) {
        C.stage_off.assign((size_t)W * W, 0);
        for (int d = 0; d < W; ++d)
            for (int s = 0; s < W; ++s) {
                C.stage_off[(size_t)d * W + s] = used[(size_t)d];
                used[(size_t)d] += ((size_t)2 * C.counts[(size_t)d * W + s] * sizeof(double) + 127) / 128 * 128;
            }
    }
    c->arena_bytes = used[(size_t)c->rank] + 4096;
    AMG_HIP(hipMalloc((void **)&c->seq, sizeof(unsigned long long) * (2 * nch + 2)));
    AMG_HIP(hipMemset(c->seq, 0, sizeof(unsigned long long) * (2 * nch + 2)));
    AMG_HIP(hipMalloc((void **)&c->ticket, sizeof(unsigned) * (2 * nch + 2)));
    AMG_HIP(hipMemset(c->ticket, 0, sizeof(unsigned) * (2 * nch + 2)));
    AMG_HIP(hipMalloc((void **)&c->timeout_flag, sizeof(int)));
    AMG_HIP(hipMemset(c->timeout_flag, 0, sizeof(int)));
    if (c->transport == 0) {
        AMG_HIP(hipExtMallocWithFlags((void **)&c->arena, c->arena_bytes, hipDeviceMallocFinegrained));
        AMG_HIP(hipMemset(c->arena, 0, c->arena_bytes));
        AMG_HIP(hipDeviceSynchronize());
        if (handle_out) {
            hipIpcMemHandle_t hdl;
            if (W > 1) {
                AMG_HIP(hipIpcGetMemHandle(&hdl, c->arena));
                static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
                std::memcpy(handle_out, &hdl, sizeof(hdl));
            } else {
                std::memset(handle_out, 0, 64);
            }
        }
    } else {
        for (auto &C : c->ch)
            if (C.send_total) AMG_HIP(hipMalloc((void **)&C.rccl_sendbuf, sizeof(double) * (size_t)C.send_total));
    }
    c->committed = true;
    return 0;
}

/* peer transport: map the other ranks' arenas (handles = world x 64 bytes, in rank order).  Collective in the sense
 * that every rank must have committed before any rank connects, and must stay alive until all have disconnected. */
int amg_comm_connect(amg_comm *c, const unsigned char *handles)
{
    if (!c || !c->committed || c->transport != 0) { set_error("connect: peer communicator not committed"); return AMG_ESTATE; }
    AMG_HIP(hipSetDevice(c->device));
    c->peer.assign((size_t)c->world, nullptr);
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) { c->peer[(size_t)p] = c->arena; continue; }
        hipIpcMemHandle_t hdl;
        std::memcpy(&hdl, handles + (size_t)p * 64, sizeof(hdl));
        void *ptr = nullptr;
        AMG_HIP(hipIpcOpenMemHandle(&ptr, hdl, hipIpcMemLazyEnablePeerAccess));
        c->peer[(size_t)p] = static_cast<char *>(ptr);
    }
    c->connected = true;
    return 0;
}

/* rccl transport, rank 0: a fresh unique id (128 bytes) for the caller to distribute */
int amg_comm_rccl_unique_id(const char *libpath, unsigned char *id_out)
{
    CHK(load_nccl(libpath));
    NcclId id;
    NCCL_CHK(g_nccl.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int amg_comm_rccl_init(amg_comm *c, const char *libpath, const unsigned char *id_bytes)
{
    if (!c || !c->committed || c->transport != 1) { set_error("rccl init: communicator not committed for rccl"); return AMG_ESTATE; }
    AMG_HIP(hipSetDevice(c->device));
    CHK(load_nccl(libpath));
    NcclId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    NCCL_CHK(g_nccl.CommInitRank(&c->nccl_comm, c->world, id, c->rank), "ncclCommInitRank");
    c->connected = true;
    return 0;
}

void amg_comm_destroy(amg_comm *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (int p = 0; p < (int)c->peer.size(); ++p)
        if (p != c->rank && c->peer[(size_t)p]) hipIpcCloseMemHandle(c->peer[(size_t)p]);
    if (c->arena) hipFree(c->arena);
    if (c->seq) hipFree(c->seq);
    if (c->ticket) hipFree(c->ticket);
    if (c->timeout_flag) hipFree(c->timeout_flag);
    for (auto &C : c->ch) if (C.rccl_sendbuf) hipFree(C.rccl_sendbuf);
    if (c->nccl_comm && g_nccl.CommDestroy) g_nccl.CommDestroy(c->nccl_comm);
    delete c;
}

/* stand-alone use on caller-supplied device pointers (tests, the Python driver): refresh dst_halo from the peers */
int amg_comm_exchange(amg_comm *c, int channel, const double *v, const int *send_idx, double *dst_halo, void *stream)
{
    if (!c) { set_error("null communicator"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(c->device));
    CHK(comm_exchange_begin(c, channel, v, send_idx, dst_halo, (hipStream_t)stream));
    return comm_exchange_end(c, channel, dst_halo, (hipStream_t)stream);
}

int amg_comm_allreduce_sqrt(amg_comm *c, int channel, const double *partial, double *result, void *stream)
{
    if (!c) { set_error("null communicator"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(c->device));
    return comm_allreduce_sqrt(c, channel, partial, result, (hipStream_t)stream);
}

int amg_comm_check(amg_comm *c) { return comm_check(c); }

}  // extern "C"
