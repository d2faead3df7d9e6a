// Probe: halo hand-off between PROCESSES through IPC-mapped device memory, all on HIP streams.
//   producer:  push kernel (stores into the peer's staging slot) -> signal kernel (system-scope flag store)
//   consumer:  wait kernel (one lane polls its own flag, bounded by a wall-clock timeout) -> unpack kernel
// Run as  ./ipc_probe <nranks> <n doubles> <iters> <finegrained 0|1> [device of rank r = r % ndev]
// Ranks are forked BEFORE any HIP call; handles travel through files in a scratch directory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include <sys/wait.h>
#include <sys/stat.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); exit(3); } } while (0)
static int g_rank = 0;

__global__ void push_kernel(double *remote, const double *src, long n, const unsigned long long *seq_dev, long slot_stride)
{
    const unsigned long long s = *seq_dev + 1;           // sequence number of THIS exchange
    double *dst = remote + (s & 1) * slot_stride;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[i] + (double)s;
}
__global__ void signal_kernel(unsigned long long *remote_flag, unsigned long long *seq_dev)
{
    const unsigned long long s = *seq_dev + 1;
    *seq_dev = s;
    __hip_atomic_store(remote_flag, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void wait_kernel(const unsigned long long *my_flag, unsigned long long *seq_dev, int *timeout_flag, long long budget_ticks)
{
    const unsigned long long s = *seq_dev + 1;
    *seq_dev = s;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(my_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > budget_ticks) { *timeout_flag = 1; break; }
    }
}
__global__ void unpack_check_kernel(const double *staging, const double *expect_base, long n, const unsigned long long *seq_dev,
                                    long slot_stride, double *halo, unsigned long long *errors)
{
    const unsigned long long s = *seq_dev;               // already advanced by wait_kernel
    const double *src = staging + (s & 1) * slot_stride;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = src[i];
        halo[i] = v;
        if (v != expect_base[i] + (double)s) atomicAdd(errors, 1ULL);
    }
}

static void publish(const std::string &dir, int rank, const void *p, size_t n)
{
    std::string tmp = dir + "/t" + std::to_string(rank), fin = dir + "/h" + std::to_string(rank);
    FILE *f = fopen(tmp.c_str(), "wb"); fwrite(p, 1, n, f); fclose(f);
    rename(tmp.c_str(), fin.c_str());
}
static void fetch(const std::string &dir, int rank, void *p, size_t n)
{
    std::string fin = dir + "/h" + std::to_string(rank);
    for (int tries = 0; tries < 60000; ++tries) {
        FILE *f = fopen(fin.c_str(), "rb");
        if (f) { size_t got = fread(p, 1, n, f); fclose(f); if (got == n) return; }
        usleep(1000);
    }
    fprintf(stderr, "rank %d: peer %d never published\n", g_rank, rank); exit(4);
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 2;
    const long n = argc > 2 ? atol(argv[2]) : 250000;
    const int iters = argc > 3 ? atoi(argv[3]) : 200;
    const int fine = argc > 4 ? atoi(argv[4]) : 1;
    char tmpl[] = "/tmp/ipcprobeXXXXXX";
    std::string dir = mkdtemp(tmpl);
    std::vector<pid_t> kids;
    for (int r = 1; r < W; ++r) {
        pid_t p = fork();
        if (p == 0) { g_rank = r; kids.clear(); break; }
        kids.push_back(p);
    }
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    CK(hipSetDevice(g_rank % ndev));
    const long slot = (n + 31) & ~31L;
    // arena: [ flags: W x u64 (one per source rank), padded to 4096 B | staging: W sources x 2 slots x slot doubles ]
    const size_t flag_bytes = 4096;
    const size_t bytes = flag_bytes + sizeof(double) * (size_t)(2 * slot) * W;
    char *arena = nullptr;
    if (fine) CK(hipExtMallocWithFlags((void **)&arena, bytes, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void **)&arena, bytes));
    CK(hipMemset(arena, 0, bytes));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t mine;
    CK(hipIpcGetMemHandle(&mine, arena));
    publish(dir, g_rank, &mine, sizeof(mine));
    std::vector<char *> peer((size_t)W, nullptr);
    for (int p = 0; p < W; ++p) {
        if (p == g_rank) { peer[p] = arena; continue; }
        hipIpcMemHandle_t h;
        fetch(dir, p, &h, sizeof(h));
        CK(hipIpcOpenMemHandle((void **)&peer[p], h, hipIpcMemLazyEnablePeerAccess));
    }
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *src, *expect, *halo;
    unsigned long long *seq_send, *seq_recv, *errors;
    int *tmo;
    CK(hipMalloc(&src, sizeof(double) * n)); CK(hipMalloc(&expect, sizeof(double) * n * W)); CK(hipMalloc(&halo, sizeof(double) * n * W));
    CK(hipMalloc(&seq_send, 8 * W)); CK(hipMalloc(&seq_recv, 8 * W)); CK(hipMalloc(&errors, 8)); CK(hipMalloc(&tmo, 4));
    CK(hipMemset(seq_send, 0, 8 * W)); CK(hipMemset(seq_recv, 0, 8 * W)); CK(hipMemset(errors, 0, 8)); CK(hipMemset(tmo, 0, 4));
    std::vector<double> h((size_t)n);
    for (long i = 0; i < n; ++i) h[i] = 1000.0 * g_rank + (double)(i % 977);
    CK(hipMemcpy(src, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    for (int p = 0; p < W; ++p) {
        for (long i = 0; i < n; ++i) h[i] = 1000.0 * p + (double)(i % 977);
        CK(hipMemcpy(expect + (long)p * n, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());
    // second rendezvous: everybody has mapped everybody
    publish(dir, 100 + g_rank, &g_rank, sizeof(int));
    for (int p = 0; p < W; ++p) { int d; fetch(dir, 100 + p, &d, sizeof(int)); }

    const long long budget = 100000000LL * 5;            // 5 s of the 100 MHz wall clock
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto exchange = [&]() {
        for (int p = 0; p < W; ++p) {
            if (p == g_rank) continue;
            double *rstage = reinterpret_cast<double *>(peer[p] + flag_bytes) + (long)g_rank * 2 * slot;
            hipLaunchKernelGGL(push_kernel, dim3(64), dim3(256), 0, st, rstage, src, n, seq_send + p, slot);
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long *>(peer[p]) + g_rank, seq_send + p);
        }
        for (int p = 0; p < W; ++p) {
            if (p == g_rank) continue;
            const double *mystage = reinterpret_cast<double *>(arena + flag_bytes) + (long)p * 2 * slot;
            hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long *>(arena) + p, seq_recv + p, tmo, budget);
            hipLaunchKernelGGL(unpack_check_kernel, dim3(64), dim3(256), 0, st, mystage, expect + (long)p * n, n, seq_recv + p, slot,
                               halo + (long)p * n, errors);
        }
    };
    for (int k = 0; k < 10; ++k) exchange();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < iters; ++k) exchange();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long herr = 0; int htmo = 0;
    CK(hipMemcpy(&herr, errors, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
    printf("rank %d/%d dev %d fine=%d n=%ld: %d exchanges, %.2f us each, errors=%llu timeout=%d\n", g_rank, W, g_rank % ndev, fine, n,
           iters, 1000.0 * ms / iters, herr, htmo);
    // graph capture of one exchange, replayed
    {
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            exchange();
            e = hipStreamEndCapture(st, &g);
        }
        if (e == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < iters; ++k) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&herr, errors, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
            printf("rank %d graph replay: %.2f us each, errors=%llu timeout=%d\n", g_rank, 1000.0 * ms / iters, herr, htmo);
        } else {
            printf("rank %d: graph capture of the exchange failed (%s)\n", g_rank, hipGetErrorString(e));
        }
    }
    fflush(stdout);
    // third rendezvous before unmapping
    publish(dir, 200 + g_rank, &g_rank, sizeof(int));
    for (int p = 0; p < W; ++p) { int d; fetch(dir, 200 + p, &d, sizeof(int)); }
    for (int p = 0; p < W; ++p) if (p != g_rank) hipIpcCloseMemHandle(peer[p]);
    int rc = (herr || htmo) ? 1 : 0;
    for (pid_t k : kids) { int s = 0; waitpid(k, &s, 0); if (!WIFEXITED(s) || WEXITSTATUS(s)) rc = 1; }
    return rc;
}
