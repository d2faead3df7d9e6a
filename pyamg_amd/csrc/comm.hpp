// Inter-GPU exchange for the row-partitioned hierarchy (one process per GPU, one node).
//
// Transport "peer": every rank owns an ARENA in fine-grained device memory, exported once through HIP IPC and
// mapped by all other ranks.  A hand-off is a PUSH over xGMI -- the producer's kernel stores straight into the
// consumer's arena -- followed by a system-scope flag store; the consumer's stream polls its own flag (bounded by a
// wall-clock budget) and then unpacks: four launches per hand-off (push, signal, wait, unpack).  AMG_COMM_FUSED=1: two -- the
// push kernel's last workgroup raises the flags, every workgroup of the unpack kernel polls them before it copies
// (correct, tested, but slower when ranks share one device; opt-in until measured on separate GPUs).  Everything is an ordinary kernel on the
// hierarchy's stream: no host synchronisation, no library call, capturable in a hipGraph.  Sequence numbers live in
// device memory (a replayed graph keeps counting), staging is double-buffered by the parity of the sequence number:
// a producer can run at most one exchange ahead of its consumer on a channel because its next wait needs the
// consumer's next push, which the consumer issues after its unpack (stream order).
//
// Transport "rccl": the same channel interface on ncclSend / ncclRecv groups and ncclAllReduce (librccl is
// dlopen-ed; the communicator is created from a unique id the caller distributes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace amg {

constexpr int COMM_MAX_RANKS = 16;

// one exchange plan: how many doubles every rank receives from every other rank (the same matrix on all ranks)
struct Channel {
    std::vector<int> counts;              // [dst * world + src]
    std::vector<size_t> stage_off;        // peer transport: byte offset of (dst, src)'s staging (2 slots) in dst's arena
    int send_total = 0, recv_total = 0;
    int send_start[COMM_MAX_RANKS + 1];   // prefix of counts[p][me] over p: where peer p's part of my send list begins
    int recv_start[COMM_MAX_RANKS + 1];   // prefix of counts[me][p] over p: where peer p's part lands in my halo
    double *rccl_sendbuf = nullptr;       // rccl transport: packed send buffer
};

}  // namespace amg

struct amg_comm {
    int rank = 0, world = 1, device = 0;
    int transport = 0;                    // 0 peer, 1 rccl
    std::vector<amg::Channel> ch;
    bool committed = false, connected = false;
    // peer transport
    char *arena = nullptr;
    size_t arena_bytes = 0, flag_bytes = 0;
    std::vector<char *> peer;             // mapped arenas, peer[rank] == arena
    unsigned long long *seq = nullptr;    // device: [2 * nchannels] send / recv sequence numbers
    unsigned *ticket = nullptr;           // device: [2 * nchannels] last-workgroup tickets of the fused push / unpack kernels
    int *timeout_flag = nullptr;          // device: set by a wait kernel whose budget ran out
    long long budget_ticks = 100000000LL * 20;   // 20 s of the 100 MHz wall clock
    // rccl transport
    void *nccl_lib = nullptr;
    void *nccl_comm = nullptr;
    std::string error;
};

namespace amg {

// gather v[send_idx[..]] and deliver it to the peers (peer: push + signal; rccl: pack + grouped send/recv into
// dst_halo); `dst_halo` = where MY incoming data finally goes (recv_total doubles)
int comm_exchange_begin(amg_comm *c, int channel, const double *v, const int *send_idx, double *dst_halo, hipStream_t st);
// wait for the peers' data of the exchange begun last on this channel and copy it to dst_halo
int comm_exchange_end(amg_comm *c, int channel, double *dst_halo, hipStream_t st);
// *result = sqrt(sum over ranks of *partial), ranks added in rank order (identical on every rank); channel of
// per-pair count 1
int comm_allreduce_sqrt(amg_comm *c, int channel, const double *partial, double *result, hipStream_t st);
// peer transport: did any wait run out of its budget?  (host-synchronous read)
int comm_check(amg_comm *c);

}  // namespace amg
