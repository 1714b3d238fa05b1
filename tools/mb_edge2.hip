// Hand-off edge microbenchmark, second form (build + run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mb_edge2.hip -o tools/bin/mb_edge2 && tools/bin/mb_edge2
// As tools/mb_edge.hip (role-partitioned chain of dependent phases, one workgroup per CU, the data is the flag), but
// with 4-byte granules {16-bit tag, 16-bit value} (bf16 activations), 16-byte sc1 loads (256 granules per wave
// instruction), NW = 4 or 8 gathering waves, and a one-granule ping-pong between two CUs for calibration.
// Every spin is bounded by s_memrealtime; a timeout raises a global abort word that every spinner polls.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned U4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// three 16-byte sc1 loads (bypass this CU's L1), waited for in the same statement
__device__ __forceinline__ void ld3_sc1(const void* p0, const void* p1, const void* p2, U4& a, U4& b, U4& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
__device__ __forceinline__ void ld1_sc1(const void* p0, U4& a) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p0) : "memory");
}

struct Phase { int c_lo, c_hi, n_in, n_out; };
constexpr int MAXP = 8;
struct Params {
    Phase ph[MAXP];
    int npat, P;
    unsigned* gran;      // [P+1][NMAX] granules
    int NMAX;
    unsigned* ctl;       // [0] epoch base, [1] abort, [2] timeouts, [3] leave counter
    int sleep;           // s_sleep between polls
    int wide_store;      // publish 4 granules per lane (16-byte sc1 stores)
    unsigned long long* stamps;   // [256 blocks][5] summed s_memtime segment lengths (diagnostic build only)
};
#ifndef STAMPS
#define STAMPS 0
#endif
__device__ __forceinline__ unsigned long long mt() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ bool tags_ok(const U4& v, unsigned tag) {
    return (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
}
__device__ __forceinline__ unsigned mk_tag(unsigned e) { return (e & 0x7fffu) | 0x8000u; }

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int NW>
__global__ __launch_bounds__(NW * 64) void edge_kernel(Params q) {
    unsigned* xs = reinterpret_cast<unsigned*>(smem);          // gathered vector (one u32 per value)
    unsigned* flag = xs + q.NMAX;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned base = __hip_atomic_load((gu32*)q.ctl, RLX_AGENT);
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    unsigned long long seg[5] = {0, 0, 0, 0, 0};
    for (int p = 0; p < q.P; ++p) {
        const Phase f = q.ph[p % q.npat];
        if (b < f.c_lo || b >= f.c_hi) continue;
        unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0;
        if (STAMPS) T0 = mt();
        const unsigned tag = mk_tag(base + (unsigned)p + 1u);
        const unsigned* in = q.gran + (size_t)p * q.NMAX;
        // ---- gather: chunk c (256 granules = 1 KiB) belongs to wave c % NW; up to three chunks per wave
        const int nchunk = f.n_in / 256;
        if (wave < nchunk) {
            const int c0 = wave, c1 = wave + NW < nchunk ? wave + NW : wave, c2 = wave + 2 * NW < nchunk ? wave + 2 * NW : wave;
            const unsigned* p0 = in + c0 * 256 + lane * 4;
            const unsigned* p1 = in + c1 * 256 + lane * 4;
            const unsigned* p2 = in + c2 * 256 + lane * 4;
            unsigned long long t0 = 0;
            U4 a, bb, c;
            for (unsigned spins = 0;; ++spins) {
                if (c1 == c0) { ld1_sc1(p0, a); bb = a; c = a; }
                else ld3_sc1(p0, p1, p2, a, bb, c);
                if (__all(tags_ok(a, tag) && tags_ok(bb, tag) && tags_ok(c, tag))) break;
                if ((spins & 63u) == 63u) {          // the clock and the abort word are looked at every 64 polls only
                    const unsigned long long t = rt();
                    if (t0 == 0) t0 = t;
                    if (t - t0 > 300000ull || __hip_atomic_load((gu32*)(q.ctl + 1), RLX_AGENT)) {   // 3 ms
                        if (lane == 0) { __hip_atomic_store((gu32*)(q.ctl + 1), 1u, RLX_AGENT); atomicAdd(q.ctl + 2, 1u); flag[0] = 1; }
                        break;
                    }
                }
                if (q.sleep) __builtin_amdgcn_s_sleep(1);
            }
            const U4 m = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
            *reinterpret_cast<U4*>(xs + c0 * 256 + lane * 4) = a & m;
            if (c1 != c0) *reinterpret_cast<U4*>(xs + c1 * 256 + lane * 4) = bb & m;
            if (c2 != c0) *reinterpret_cast<U4*>(xs + c2 * 256 + lane * 4) = c & m;
        }
        if (STAMPS) T1 = mt();
        __syncthreads();
        if (STAMPS) T2 = mt();
        if (flag[0]) break;
        // ---- compute stand-in: every wave sums the whole vector, exact in u32
        unsigned s = 0;
        for (int i = lane; i < f.n_in; i += 64) s += xs[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (STAMPS) T3 = mt();
        // ---- publish this block's rows of the next vector
        const int nb = f.c_hi - f.c_lo, r = b - f.c_lo;
        const int lo = f.n_out / nb * r, hi = lo + f.n_out / nb;      // n_out % (4 * nb) == 0
        unsigned* out = q.gran + (size_t)(p + 1) * q.NMAX;
        const unsigned ntag = mk_tag(base + (unsigned)p + 2u) << 16;
        if (q.wide_store) {
            for (int i = lo + tid * 4; i < hi; i += NW * 64 * 4) {
                U4 v;
                v.x = ntag | ((s * 2654435761u + (unsigned)i + (unsigned)p) & 0xffffu);
                v.y = ntag | ((s * 2654435761u + (unsigned)(i + 1) + (unsigned)p) & 0xffffu);
                v.z = ntag | ((s * 2654435761u + (unsigned)(i + 2) + (unsigned)p) & 0xffffu);
                v.w = ntag | ((s * 2654435761u + (unsigned)(i + 3) + (unsigned)p) & 0xffffu);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(out + i), "v"(v) : "memory");
            }
        } else if (q.wide_store == 2) {
            // one granule per store instruction, from lane 0 of successive waves (what a row-per-wave epilogue does)
            for (int i = lo + wave; i < hi; i += NW)
                if (lane == 0) __hip_atomic_store((gu32*)(out + i), ntag | ((s * 2654435761u + (unsigned)i + (unsigned)p) & 0xffffu), RLX_AGENT);
        } else {
            for (int i = lo + tid; i < hi; i += NW * 64)
                __hip_atomic_store((gu32*)(out + i), ntag | ((s * 2654435761u + (unsigned)i + (unsigned)p) & 0xffffu), RLX_AGENT);
        }
        if (STAMPS) T4 = mt();
        __syncthreads();   // xs is rewritten by the next phase this block takes part in
        if (STAMPS) { const unsigned long long T5 = mt(); seg[0] += T1 - T0; seg[1] += T2 - T1; seg[2] += T3 - T2; seg[3] += T4 - T3; seg[4] += T5 - T4; }
    }
    if (STAMPS && tid == 0) for (int i = 0; i < 5; ++i) q.stamps[b * 5 + i] = seg[i];
    __syncthreads();
    if (tid == 0) {
        const unsigned old = atomicAdd(q.ctl + 3, 1u);
        if (old + 1 == gridDim.x) { q.ctl[3] = 0; __hip_atomic_store((gu32*)q.ctl, base + (unsigned)q.P + 2u, RLX_AGENT); }
    }
}

// one-granule ping-pong between block 0 and block `peer`
__global__ void pingpong_kernel(unsigned* w, unsigned* ctl, int iters, int peer) {
    const int b = blockIdx.x;
    if ((b != 0 && b != peer) || threadIdx.x != 0) return;
    gu32* mine = (gu32*)(w + (b == 0 ? 0 : 64));
    gu32* theirs = (gu32*)(w + (b == 0 ? 64 : 0));
    for (int it = 1; it <= iters; ++it) {
        if (b == 0) __hip_atomic_store(theirs, (unsigned)it, RLX_AGENT);
        const unsigned long long t0 = rt();
        while (__hip_atomic_load(mine, RLX_AGENT) != (unsigned)it) {
            if (rt() - t0 > 300000ull) { ctl[1] = 1; return; }
        }
        if (b != 0) __hip_atomic_store(theirs, (unsigned)it, RLX_AGENT);
    }
}

__global__ void seed_kernel(unsigned* gran, const unsigned* ctl, int n) {
    const unsigned base = ctl[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) gran[i] = (((base + 1u) & 0x7fffu) | 0x8000u) << 16 | ((unsigned)(i * 7 + 1) & 0xffffu);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NMAX = 4096, P = 112;
    unsigned* gran; CK(hipMalloc(&gran, (size_t)(P + 1) * NMAX * 4)); CK(hipMemset(gran, 0, (size_t)(P + 1) * NMAX * 4));
    unsigned* ctl; CK(hipMalloc(&ctl, 256)); CK(hipMemset(ctl, 0, 256));
    unsigned* pp; CK(hipMalloc(&pp, 1024)); CK(hipMemset(pp, 0, 1024));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 5 * 8)); CK(hipMemset(stamps, 0, 256 * 5 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 104 * 1024;
    CK(hipFuncSetAttribute((const void*)edge_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)edge_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    for (int peer : {1, 8, 9, 17}) {
        const int iters = 2000;
        CK(hipMemset(pp, 0, 1024));
        CK(hipEventRecord(e0, s));
        pingpong_kernel<<<32, 64, 0, s>>>(pp, ctl, iters, peer);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("ping-pong block 0 <-> %2d: %.3f us per one-way hop\n", peer, ms * 1e3 / iters / 2);
    }
    unsigned z4[4] = {0, 0, 0, 0}; CK(hipMemcpy(ctl, z4, 16, hipMemcpyHostToDevice));

    auto host_chain = [&](const Params& q) {
        std::vector<unsigned> x(q.ph[0].n_in);
        for (int i = 0; i < q.ph[0].n_in; ++i) x[i] = (unsigned)(i * 7 + 1) & 0xffffu;
        for (int p = 0; p < q.P; ++p) {
            const Phase f = q.ph[p % q.npat];
            unsigned sum = 0; for (int i = 0; i < f.n_in; ++i) sum += x[i];
            std::vector<unsigned> y(f.n_out);
            for (int i = 0; i < f.n_out; ++i) y[i] = (sum * 2654435761u + (unsigned)i + (unsigned)p) & 0xffffu;
            x.swap(y);
        }
        return x;
    };
    auto run = [&](const char* name, Params q, int NW) {
        q.gran = gran; q.NMAX = NMAX; q.ctl = ctl; q.P = P; q.stamps = stamps;
        float best = 1e9f; bool okall = true; unsigned tmo = 0;
        for (int rep = 0; rep < 8; ++rep) {
            seed_kernel<<<(q.ph[0].n_in + 255) / 256, 256, 0, s>>>(gran, ctl, q.ph[0].n_in);
            CK(hipEventRecord(e0, s));
            if (NW == 4) edge_kernel<4><<<256, 256, lds, s>>>(q); else edge_kernel<8><<<256, 512, lds, s>>>(q);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            unsigned h[4]; CK(hipMemcpy(h, ctl, 16, hipMemcpyDeviceToHost));
            tmo += h[2];
            if (h[1]) { okall = false; unsigned z[4] = {h[0] + 1000u, 0, 0, 0}; CK(hipMemcpy(ctl, z, 16, hipMemcpyHostToDevice)); continue; }
            const Phase fl = q.ph[(P - 1) % q.npat];
            std::vector<unsigned> g(fl.n_out);
            CK(hipMemcpy(g.data(), gran + (size_t)P * NMAX, (size_t)fl.n_out * 4, hipMemcpyDeviceToHost));
            const std::vector<unsigned> want = host_chain(q);
            for (int i = 0; i < fl.n_out; ++i) if ((g[i] & 0xffffu) != want[i]) { okall = false; break; }
        }
        if (STAMPS) {
            std::vector<unsigned long long> h(256 * 5); CK(hipMemcpy(h.data(), stamps, 256 * 5 * 8, hipMemcpyDeviceToHost));
            for (int bb : {0, 1, 100, 255}) {
                int np = 0; for (int p = 0; p < P; ++p) { const Phase f = q.ph[p % q.npat]; if (bb >= f.c_lo && bb < f.c_hi) ++np; }
                if (!np) continue;
                printf("   block %3d (wave 0 of it), cycles per own phase: gather %5.0f  barrier %5.0f  sum %5.0f  store %5.0f  barrier %5.0f\n", bb,
                       (double)h[bb * 5] / np, (double)h[bb * 5 + 1] / np, (double)h[bb * 5 + 2] / np, (double)h[bb * 5 + 3] / np, (double)h[bb * 5 + 4] / np);
            }
        }
        printf("%-60s %7.3f us/phase  (%.1f us per %d-phase pattern)%s%s\n", name, best * 1e3 / P, best * 1e3 / P * q.npat, q.npat,
               okall ? "" : "  WRONG/ABORT", tmo ? "  (timeouts)" : "");
        fflush(stdout);
    };

    for (int NW : {8}) for (int sl : {0}) for (int ws : {0, 2}) {
        char nm[128];
        Params q{}; q.sleep = sl; q.wide_store = ws;
        q.npat = 4;
        q.ph[0] = {0, 64, 1024, 2048}; q.ph[1] = {64, 96, 2048, 1024}; q.ph[2] = {96, 192, 1024, 3072}; q.ph[3] = {192, 256, 3072, 1024};
        snprintf(nm, sizeof nm, "roles 64/32/96/64 NW=%d sleep=%d wide_store=%d", NW, sl, ws); run(nm, q, NW);
        q.ph[0] = {0, 256, 1024, 2048}; q.ph[1] = {0, 256, 2048, 1024}; q.ph[2] = {0, 256, 1024, 3072}; q.ph[3] = {0, 256, 3072, 1024};
        if (ws != 1) { snprintf(nm, sizeof nm, "all 256 CUs       NW=%d sleep=%d wide_store=%d", NW, sl, ws); run(nm, q, NW); }
        q.ph[0] = {0, 16, 1024, 2048}; q.ph[1] = {16, 32, 2048, 1024}; q.ph[2] = {32, 48, 1024, 3072}; q.ph[3] = {48, 64, 3072, 1024};
        snprintf(nm, sizeof nm, "roles 16/16/16/16 NW=%d sleep=%d wide_store=%d", NW, sl, ws); run(nm, q, NW);
        q.npat = 1; q.ph[0] = {0, 64, 1024, 1024};
        snprintf(nm, sizeof nm, "one role of 64, n=1024 NW=%d sleep=%d wide_store=%d", NW, sl, ws); run(nm, q, NW);
    }
    return 0;
}
