// Grid-barrier microbenchmark for a persistent decode kernel (build + run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mb_barrier.hip -o /tmp/mb_barrier && /tmp/mb_barrier
// Every spin loop is bounded (SPIN_CAP) so a mistake cannot hang the device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define SPIN_CAP 2000000

// (a) one counter, every block adds 1 and waits for G*(it+1)
__global__ void bar_single(unsigned* ctr, unsigned* err, int iters) {
    const unsigned G = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);
            const unsigned want = G * (unsigned)(it + 1);
            int spins = 0;
            while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < want) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
            }
        }
        __syncthreads();
    }
}

// (b) two level: blocks of one group (blockIdx % NG) add to the group's counter; the last arriver of a group adds to the
// top counter; everyone polls the top counter.
__global__ void bar_two_level(unsigned* grp, unsigned* top, unsigned* err, int iters, int NG) {
    const unsigned G = gridDim.x;
    const int g = blockIdx.x % NG;
    const unsigned per = G / NG;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __atomic_fetch_add(grp + g * 32, 1u, __ATOMIC_ACQ_REL);
            if (old + 1 == per * (unsigned)(it + 1)) __atomic_fetch_add(top, 1u, __ATOMIC_RELEASE);
            const unsigned want = (unsigned)NG * (unsigned)(it + 1);
            int spins = 0;
            while (__atomic_load_n(top, __ATOMIC_ACQUIRE) < want) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
            }
        }
        __syncthreads();
    }
}

// (c) flags: block b stores epoch into flag[b]; every block's first wave reads all G flags (G <= 256: 4 per lane) and
// spins until all have the epoch.  No atomics RMW at all.
__global__ void bar_flags(unsigned* flags, unsigned* err, int iters) {
    const int G = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned ep = it + 1;
            if (threadIdx.x == 0) __atomic_store_n(flags + blockIdx.x * 16, ep, __ATOMIC_RELEASE);
            int spins = 0;
            for (;;) {
                bool ok = true;
                for (int b = threadIdx.x; b < G; b += 64) ok &= __atomic_load_n(flags + b * 16, __ATOMIC_RELAXED) >= ep;
                if (__all(ok)) break;
                if (++spins > SPIN_CAP / 8) { *err = 1; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
    }
}

// (d) packed flags: like (c) but the G flags are contiguous u32 (one cache line per 16/32 blocks)
__global__ void bar_flags_packed(unsigned* flags, unsigned* err, int iters) {
    const int G = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned ep = it + 1;
            if (threadIdx.x == 0) __atomic_store_n(flags + blockIdx.x, ep, __ATOMIC_RELEASE);
            int spins = 0;
            for (;;) {
                bool ok = true;
                for (int b = threadIdx.x; b < G; b += 64) ok &= __atomic_load_n(flags + b, __ATOMIC_RELAXED) >= ep;
                if (__all(ok)) break;
                if (++spins > SPIN_CAP / 8) { *err = 1; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *buf, *err; CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&err, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](const char* name, int G, int T, auto launch) {
        CK(hipMemsetAsync(buf, 0, 1 << 20, s)); CK(hipMemsetAsync(err, 0, 64, s));
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        printf("%-18s G=%3d T=%4d: %.3f us/barrier%s\n", name, G, T, ms * 1e3 / iters, h ? "  (SPIN CAP HIT)" : "");
    };
    for (int G : {32, 64, 128, 256}) {
        for (int T : {64, 256}) {
            run("single", G, T, [&] { bar_single<<<G, T, 0, s>>>(buf, err, iters); });
            run("two-level NG=8", G, T, [&] { bar_two_level<<<G, T, 0, s>>>(buf, buf + 4096, err, iters, 8); });
            run("flags (64B apart)", G, T, [&] { bar_flags<<<G, T, 0, s>>>(buf, err, iters); });
            run("flags packed", G, T, [&] { bar_flags_packed<<<G, T, 0, s>>>(buf, err, iters); });
        }
    }
    return 0;
}
