// Hand-off edge microbenchmark for a role-partitioned persistent decode kernel (build + run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mb_edge.hip -o /tmp/mb_edge && /tmp/mb_edge
// One 256-thread workgroup per CU (104 KB of LDS forces it).  A chain of P dependent phases; phase p is computed by
// the workgroups of one ROLE (a contiguous range of block ids), which gather the whole vector the previous phase
// published as 8-byte {value, tag} granules (sc1 stores / sc1 loads, the data is the flag), compute, and publish
// their rows of the next vector.  Variants: role-partitioned ring (64/32/96/64 CUs) against every-CU-every-phase,
// one gathering wave against four, with and without a weight stream beside it.
// Every spin is bounded by s_memrealtime; a timeout raises a global abort word that every spinner polls.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned U4 __attribute__((ext_vector_type(4)));
// one 16-byte sc1 load (bypasses this CU's L1), waited for in the same statement
__device__ __forceinline__ U4 ld16_sc1(const void* p) {
    U4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// NL 16-byte sc1 loads, 1 KiB apart, issued together
template <int NL> __device__ __forceinline__ void ld16_sc1_n(const char* p, U4 (&v)[NL]);
template <> __device__ __forceinline__ void ld16_sc1_n<1>(const char* p, U4 (&v)[1]) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld16_sc1_n<2>(const char* p, U4 (&v)[2]) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld16_sc1_n<3>(const char* p, U4 (&v)[3]) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:2048 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(p) : "memory");
}
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Phase { int c_lo, c_hi, n_in, n_out; };
constexpr int MAXP = 16;
struct Params {
    Phase ph[MAXP];      // the pattern, repeated
    int npat, P;
    unsigned long long* gran;   // [P+1][NMAX] granules
    int NMAX;
    unsigned* ctl;       // [0] epoch base, [1] abort, [2] timeouts
    const U4* stream;    // weight stream stand-in
    long stream_stride;  // U4 per block
    int stream_loads;    // 16-byte loads per lane per phase by the loader wave (0 = none)
    int gather_waves;    // 1 or 4
    unsigned* sink;
};

__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__global__ __launch_bounds__(256) void edge_kernel(Params q) {
    unsigned* xs = reinterpret_cast<unsigned*>(smem);          // gathered vector
    unsigned* flag = xs + q.NMAX;                               // [0] abort seen by this block
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned base = __hip_atomic_load((gu32*)q.ctl, RLX_AGENT);
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    unsigned acc = 0;
    long soff = 0;
    for (int p = 0; p < q.P; ++p) {
        const Phase f = q.ph[p % q.npat];
        if (b < f.c_lo || b >= f.c_hi) continue;
        const unsigned tag = base + (unsigned)p + 1u;      // tag of the vector consumed (published by phase p-1)
        const gu64* in = (const gu64*)(q.gran + (size_t)p * q.NMAX);
        // ---- loader stand-in: wave 3 streams beside the gather
        if (q.stream_loads && wave == 3) {
            const U4* s = q.stream + (size_t)b * q.stream_stride + soff;
            U4 a = {0, 0, 0, 0};
            for (int i = 0; i < q.stream_loads; ++i) { const U4 v = __builtin_nontemporal_load(s + i * 64 + lane); a ^= v; }
            acc ^= a.x ^ a.y ^ a.z ^ a.w;
            soff += (long)q.stream_loads * 64;
            if (soff + (long)q.stream_loads * 64 > q.stream_stride) soff = 0;
        }
        // ---- gather: every pass re-reads the granules of this wave until all carry the tag
        if (wave < q.gather_waves) {
            const int per = f.n_in / q.gather_waves;          // granules of this wave (multiple of 128)
            const int g0 = wave * per;
            const unsigned long long t0 = rt();
            bool dead = false;
            for (int i0 = 0; i0 < per; i0 += 128 * 4) {
                for (;;) {
                    bool ok = true;
                    unsigned v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int g = g0 + i0 + k * 128 + lane * 2;
                        if (i0 + k * 128 < per) {
                            const unsigned long long a = __hip_atomic_load(in + g, RLX_AGENT);
                            const unsigned long long c = __hip_atomic_load(in + g + 1, RLX_AGENT);
                            ok &= (unsigned)(a >> 32) == tag && (unsigned)(c >> 32) == tag;
                            v[2 * k] = (unsigned)a; v[2 * k + 1] = (unsigned)c;
                        }
                    }
                    if (__all(ok)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (i0 + k * 128 < per) { const int g = g0 + i0 + k * 128 + lane * 2; xs[g] = v[2 * k]; xs[g + 1] = v[2 * k + 1]; }
                        break;
                    }
                    if (rt() - t0 > 300000ull || __hip_atomic_load((gu32*)(q.ctl + 1), RLX_AGENT)) {   // 3 ms
                        if (lane == 0) { __hip_atomic_store((gu32*)(q.ctl + 1), 1u, RLX_AGENT); atomicAdd(q.ctl + 2, 1u); flag[0] = 1; }
                        dead = true; break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dead) break;
            }
        }
        __syncthreads();
        if (flag[0]) break;
        // ---- compute: every wave sums the whole vector (stand-in for the dot products), exact in u32
        unsigned s = 0;
        for (int i = lane; i < f.n_in; i += 64) s += xs[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        // ---- publish this block's rows of the next vector
        const int nb = f.c_hi - f.c_lo, r = b - f.c_lo;
        const int lo = (int)((long)f.n_out * r / nb), hi = (int)((long)f.n_out * (r + 1) / nb);
        gu64* out = (gu64*)(q.gran + (size_t)(p + 1) * q.NMAX);
        for (int i = lo + tid; i < hi; i += 256) {
            const unsigned val = s * 2654435761u + (unsigned)i + (unsigned)p;
            __hip_atomic_store(out + i, ((unsigned long long)(tag + 1u) << 32) | val, RLX_AGENT);
        }
        __syncthreads();   // xs is rewritten by the next phase this block takes part in
    }
    if (acc == 0x12345678u) q.sink[0] = acc;
    // the last block to leave advances the epoch base for the next launch
    __syncthreads();
    if (tid == 0) {
        const unsigned old = atomicAdd(q.ctl + 3, 1u);
        if (old + 1 == gridDim.x) { q.ctl[3] = 0; __hip_atomic_store((gu32*)q.ctl, base + (unsigned)q.P + 2u, RLX_AGENT); }
    }
}

__global__ void seed_kernel(unsigned long long* gran, const unsigned* ctl, int n) {
    const unsigned base = ctl[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) gran[i] = ((unsigned long long)(base + 1u) << 32) | (unsigned)(i * 7 + 1);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NMAX = 4096, P = 112;
    unsigned long long* gran; CK(hipMalloc(&gran, (size_t)(P + 1) * NMAX * 8)); CK(hipMemset(gran, 0, (size_t)(P + 1) * NMAX * 8));
    unsigned* ctl; CK(hipMalloc(&ctl, 256)); CK(hipMemset(ctl, 0, 256));
    const long stride = 1 << 16;   // 1 MiB per block
    U4* stream; CK(hipMalloc(&stream, (size_t)256 * stride * 16)); CK(hipMemset(stream, 1, (size_t)256 * stride * 16));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 104 * 1024;
    CK(hipFuncSetAttribute((const void*)edge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    auto host_chain = [&](const Params& q) {
        std::vector<unsigned> x(q.ph[0].n_in);
        for (int i = 0; i < q.ph[0].n_in; ++i) x[i] = (unsigned)(i * 7 + 1);
        for (int p = 0; p < q.P; ++p) {
            const Phase f = q.ph[p % q.npat];
            unsigned sum = 0; for (int i = 0; i < f.n_in; ++i) sum += x[i];
            std::vector<unsigned> y(f.n_out);
            for (int i = 0; i < f.n_out; ++i) y[i] = sum * 2654435761u + (unsigned)i + (unsigned)p;
            x.swap(y);
        }
        return x;
    };
    auto run = [&](const char* name, Params q) {
        q.gran = gran; q.NMAX = NMAX; q.ctl = ctl; q.stream = stream; q.stream_stride = stride; q.sink = sink; q.P = P;
        float best = 1e9f; bool okall = true; unsigned tmo = 0;
        for (int rep = 0; rep < 6; ++rep) {
            seed_kernel<<<(q.ph[0].n_in + 255) / 256, 256, 0, s>>>(gran, ctl, q.ph[0].n_in);
            CK(hipEventRecord(e0, s));
            edge_kernel<<<256, 256, lds, s>>>(q);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            unsigned h[4]; CK(hipMemcpy(h, ctl, 16, hipMemcpyDeviceToHost));
            tmo += h[2];
            if (h[1]) { okall = false; unsigned z[3] = {h[0] + 1000u, 0, 0}; CK(hipMemcpy(ctl, z, 12, hipMemcpyHostToDevice)); CK(hipMemset(ctl + 3, 0, 4)); continue; }
            // check the last vector
            const Phase fl = q.ph[(P - 1) % q.npat];
            std::vector<unsigned long long> g(fl.n_out);
            CK(hipMemcpy(g.data(), gran + (size_t)P * NMAX, (size_t)fl.n_out * 8, hipMemcpyDeviceToHost));
            const std::vector<unsigned> want = host_chain(q);
            for (int i = 0; i < fl.n_out; ++i) if ((unsigned)g[i] != want[i]) { okall = false; break; }
        }
        printf("%-46s %7.3f us/phase  (%.1f us per %d-phase pattern)%s%s\n", name, best * 1e3 / P, best * 1e3 / P * q.npat, q.npat,
               okall ? "" : "  WRONG/ABORT", tmo ? "  (timeouts)" : "");
        fflush(stdout);
    };

    for (int gw : {1, 4}) for (int sl : {0, 32, 128}) {
        char nm[128];
        Params q{}; q.gather_waves = gw; q.stream_loads = sl;
        // role ring: QKV 64 CUs (1024 -> 2048), Wo 32 (2048 -> 1024), W13 96 (1024 -> 3072), W2 64 (3072 -> 1024)
        q.npat = 4;
        q.ph[0] = {0, 64, 1024, 2048}; q.ph[1] = {64, 96, 2048, 1024}; q.ph[2] = {96, 192, 1024, 3072}; q.ph[3] = {192, 256, 3072, 1024};
        snprintf(nm, sizeof nm, "roles 64/32/96/64  gather_waves=%d stream=%3dKB", gw, sl); run(nm, q);
        // every CU takes part in every phase
        q.ph[0] = {0, 256, 1024, 2048}; q.ph[1] = {0, 256, 2048, 1024}; q.ph[2] = {0, 256, 1024, 3072}; q.ph[3] = {0, 256, 3072, 1024};
        snprintf(nm, sizeof nm, "all 256 CUs        gather_waves=%d stream=%3dKB", gw, sl); run(nm, q);
        // narrow roles
        q.ph[0] = {0, 32, 1024, 2048}; q.ph[1] = {32, 48, 2048, 1024}; q.ph[2] = {48, 96, 1024, 3072}; q.ph[3] = {96, 128, 3072, 1024};
        snprintf(nm, sizeof nm, "roles 32/16/48/32  gather_waves=%d stream=%3dKB", gw, sl); run(nm, q);
        q.ph[0] = {0, 8, 1024, 2048}; q.ph[1] = {8, 16, 2048, 1024}; q.ph[2] = {16, 24, 1024, 3072}; q.ph[3] = {24, 32, 3072, 1024};
        snprintf(nm, sizeof nm, "roles 8/8/8/8      gather_waves=%d stream=%3dKB", gw, sl); run(nm, q);
    }
    return 0;
}
