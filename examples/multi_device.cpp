// Driving every GPU of a node from ONE process through the C ABI (include/cntt.h) -- SURVEY 8(e), BASELINE config C4:
// a batch of independent polynomials that starts and ends on device 0 is split into contiguous shards (cntt_shard_bounds),
// each shard is copied to its device over xGMI (hipMemcpyPeerAsync), transformed there, and copied back.  There is no collective
// and no cross-device dependency: one host thread + one stream per device, the plan handle shared by all of them (the library keeps
// one table replica per device, created lazily under a mutex on that device's first call; calls are thread-safe).
//
// This is the shape a Rust binding would use with std::thread::scope and one hipStream per device; the
// Python package does the same across PROCESSES with torch.distributed / RCCL (concrete-ntt_amd/shard.py, bench.py --gpus N).
//
//   make -C examples multi_device
//   ./examples/multi_device [--n 16384] [--batch 4096] [--world W] [--logical]
// --world W     number of shards (default: every visible device).  With --logical the W shards are spread round-robin over the
//               devices that exist (W = 3 on a one-GPU box exercises partition, scatter and gather with three threads and
//               three streams on one device); without it W is clamped to the device count.
// Exit status 0 and "OK" when the gathered result equals the single-device transform of the same batch, bit for bit.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/cntt.h"

#define HIP_OK(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                             \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)
#define CNTT_OK_OR_DIE(call)                                                                       \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != CNTT_OK) {                                                                      \
            fprintf(stderr, "%s: status %d: %s\n", #call, rc_, cntt_last_error());                 \
            std::exit(3);                                                                          \
        }                                                                                          \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    size_t n = 16384, batch = 4096;
    int world = 0;
    bool logical = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--n") && i + 1 < argc) n = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--logical")) logical = true;
        else { fprintf(stderr, "usage: %s [--n N] [--batch B] [--world W] [--logical]\n", argv[0]); return 1; }
    }
    const uint64_t p = 4611686018427322369ull;   // benches/ntt.rs:115 (the headline prime; C4's modulus)
    const int ndev = cntt_device_count();
    if (ndev < 1) { fprintf(stderr, "no GPU: the library has no CPU path\n"); return 1; }
    if (world <= 0) world = ndev;
    if (!logical && world > ndev) world = ndev;

    cntt_plan64_t *plan = nullptr;
    CNTT_OK_OR_DIE(cntt_prime64_plan_new(n, p, &plan));

    // the whole batch on device 0, and its single-device transform as the expected result
    const size_t words = batch * n, bytes = words * sizeof(uint64_t);
    uint64_t *full = nullptr, *expect = nullptr;
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipMalloc(&full, bytes));
    HIP_OK(hipMalloc(&expect, bytes));
    CNTT_OK_OR_DIE(cntt_fill_uniform_u64(full, words, p, 0x5EED0C04ull, nullptr));
    HIP_OK(hipMemcpy(expect, full, bytes, hipMemcpyDeviceToDevice));
    CNTT_OK_OR_DIE(cntt_prime64_fwd_batch(plan, expect, batch, CNTT_MEM_DEVICE, nullptr));
    HIP_OK(hipDeviceSynchronize());

    struct Leg { double scatter = 0, compute = 0, gather = 0; size_t begin = 0, end = 0; int dev = 0; };
    std::vector<Leg> legs((size_t)world);
    const double t0 = now();
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r) {
        threads.emplace_back([&, r] {
            Leg &L = legs[(size_t)r];
            L.dev = r % ndev;
            HIP_OK(hipSetDevice(L.dev));                         // per-thread current device: the library launches THERE
            CNTT_OK_OR_DIE(cntt_shard_bounds(batch, world, r, &L.begin, &L.end));
            const size_t mine = L.end - L.begin, mbytes = mine * n * sizeof(uint64_t);
            if (mine == 0) return;
            hipStream_t st;
            HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            uint64_t *slice = full + L.begin * n, *shard = slice;
            const bool in_place = (r == 0 && L.dev == 0);        // rank 0's shard never leaves device 0
            double t = now();
            if (!in_place) {
                HIP_OK(hipMalloc(&shard, mbytes));
                HIP_OK(hipMemcpyPeerAsync(shard, L.dev, slice, 0, mbytes, st));   // scatter: one xGMI link per peer device
                HIP_OK(hipStreamSynchronize(st));
            }
            L.scatter = now() - t;
            t = now();
            CNTT_OK_OR_DIE(cntt_prime64_fwd_batch(plan, shard, mine, CNTT_MEM_DEVICE, st));
            HIP_OK(hipStreamSynchronize(st));
            L.compute = now() - t;
            t = now();
            if (!in_place) {
                HIP_OK(hipMemcpyPeerAsync(slice, 0, shard, L.dev, mbytes, st));   // gather
                HIP_OK(hipStreamSynchronize(st));
                HIP_OK(hipFree(shard));
            }
            L.gather = now() - t;
            HIP_OK(hipStreamDestroy(st));
        });
    }
    for (auto &t : threads) t.join();
    const double total = now() - t0;

    // bit-exact against the single-device result
    HIP_OK(hipSetDevice(0));
    std::vector<uint64_t> got(words), want(words);
    HIP_OK(hipMemcpy(got.data(), full, bytes, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(want.data(), expect, bytes, hipMemcpyDeviceToHost));
    size_t bad = 0, covered = 0;
    for (size_t i = 0; i < words; ++i) bad += got[i] != want[i];
    for (int r = 0; r < world; ++r) {
        const Leg &L = legs[(size_t)r];
        covered += L.end - L.begin;
        printf("shard %d on device %d: polynomials [%zu, %zu)  scatter %.4f s  fwd %.4f s  gather %.4f s\n", r, L.dev, L.begin,
               L.end, L.scatter, L.compute, L.gather);
    }
    printf("%zu polynomials of %zu points over %d shard(s) on %d device(s): %.4f s end to end = %.3e NTT/s\n", batch, n, world,
           ndev < world ? ndev : world, total, (double)batch / total);
    cntt_prime64_plan_free(plan);
    HIP_OK(hipFree(full));
    HIP_OK(hipFree(expect));
    if (bad || covered != batch) {
        fprintf(stderr, "MISMATCH: %zu words differ, %zu of %zu polynomials covered\n", bad, covered, batch);
        return 4;
    }
    printf("OK\n");
    return 0;
}
