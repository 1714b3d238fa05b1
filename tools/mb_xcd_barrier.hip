// Barrier among workgroups that share one XCD (blockIdx % 8 == xcd): atomics at reduced scope execute in that XCD's L2
// instead of travelling to memory.  Measures the cost and CHECKS data visibility (each block publishes a value before
// the barrier and reads every other block's value after it).  Bounded spins: a mistake cannot hang the device.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mb_xcd_barrier.hip -o /tmp/mb_xcd && /tmp/mb_xcd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define SPIN_CAP 400000

template <int SCOPE_ADD, int SCOPE_LD, bool NT_DATA>
__global__ void xcd_barrier(unsigned* ctr, unsigned* data, unsigned* err, unsigned* bad, int iters, int nxcd, int G) {
    if ((int)(blockIdx.x % nxcd) != 0) return;            // only the blocks of XCD 0 take part
    const int b = blockIdx.x / nxcd;                      // 0..G-1
    if (b >= G) return;
    for (int it = 0; it < iters; ++it) {
        // publish
        if (threadIdx.x == 0) {
            if (NT_DATA) __builtin_nontemporal_store((unsigned)(it * 1000 + b), &data[b * 16]);
            else __hip_atomic_store(&data[b * 16], (unsigned)(it * 1000 + b), __ATOMIC_RELAXED, SCOPE_ADD);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, SCOPE_ADD);
            const unsigned want = (unsigned)G * (unsigned)(it + 1);
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, SCOPE_LD) < want) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
            }
        }
        __syncthreads();
        // check every other block's value
        if ((int)threadIdx.x < G) {
            const unsigned v = __hip_atomic_load(&data[threadIdx.x * 16], __ATOMIC_RELAXED, SCOPE_LD);
            if (v != (unsigned)(it * 1000 + threadIdx.x)) atomicAdd(bad, 1u);
        }
        __syncthreads();
        // second barrier so that nobody overwrites data before all have read it (also timed)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr + 64, 1u, __ATOMIC_RELEASE, SCOPE_ADD);
            const unsigned want = (unsigned)G * (unsigned)(it + 1);
            int spins = 0;
            while (__hip_atomic_load(ctr + 64, __ATOMIC_ACQUIRE, SCOPE_LD) < want) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
            }
        }
        __syncthreads();
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *buf, *err; CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&err, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](const char* name, int G, auto launch) {
        CK(hipMemsetAsync(buf, 0, 1 << 20, s)); CK(hipMemsetAsync(err, 0, 64, s));
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
        printf("%-44s G=%2d: %.3f us per barrier, stale reads %u%s\n", name, G, ms * 1e3 / (2 * iters), h[1], h[0] ? "  (SPIN CAP HIT)" : "");
    };
    constexpr int WG = __HIP_MEMORY_SCOPE_WORKGROUP, AG = __HIP_MEMORY_SCOPE_AGENT, SY = __HIP_MEMORY_SCOPE_SYSTEM;
    for (int G : {8, 16, 32}) {
        const int grid = 8 * G;   // round-robin: block i runs on XCD i % 8
        run("add=agent  load=agent", G, [&] { xcd_barrier<AG, AG, false><<<grid, 256, 0, s>>>(buf, buf + 1024, err, err + 1, iters, 8, G); });
        run("add=system load=system", G, [&] { xcd_barrier<SY, SY, false><<<grid, 256, 0, s>>>(buf, buf + 1024, err, err + 1, iters, 8, G); });
        run("add=wg     load=agent", G, [&] { xcd_barrier<WG, AG, false><<<grid, 256, 0, s>>>(buf, buf + 1024, err, err + 1, iters, 8, G); });
        run("add=wg     load=wg (may be stale/hang-capped)", G, [&] { xcd_barrier<WG, WG, false><<<grid, 256, 0, s>>>(buf, buf + 1024, err, err + 1, iters, 8, G); });
        // control: the same G blocks spread over all XCDs (nxcd = 1 => every block participates, grid = G)
        run("control: G blocks on different XCDs, agent", G, [&] { xcd_barrier<AG, AG, false><<<G, 256, 0, s>>>(buf, buf + 1024, err, err + 1, iters, 1, G); });
    }
    return 0;
}
