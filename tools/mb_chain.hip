// Microbenchmark of the chain launch (fish-tts_amd/csrc/wide_kernels.h: wide_chain_gemm_kernel): the four Linear phases of
// L transformer layers at the decode shapes (Wqkv, Wo, W13, W2; attention left out) as ONE launch whose workgroups hand
// over through per-phase counters, against the same phases as 4 L launches of a captured graph.  The two must produce
// the same bits.
// RESULT (profiles/r04_mb_chain.txt): correct and deadlock-free, but NOT faster - agent-scope release / acquire fences
// (buffer_wbl2 sc1 / buffer_inv sc1) cost ~25 us per phase; without them (timing only) a phase costs 6.4-6.7 us, the same as
// a launch of the captured graph (6.4): the graph's launch gap is not what a lock-step phase waits for.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I fish-tts_amd/csrc tools/mb_chain.hip -o tools/bin/mb_chain
//   tools/bin/mb_chain [M=32] [L=28]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

#include "ar_kernels.h"
#include "wide_kernels.h"
using namespace ft;
#ifndef CHAIN_MIN_WAVES
#define CHAIN_MIN_WAVES 4
#endif

// Hand-over between the phases of ONE launch (wide_chain_kernel below): a phase's workgroups count in when their outputs
// are written; a consumer workgroup requests its weights, THEN waits for the producer phase's count, then reads the
// activations.  Every spin is bounded; a time-out raises the abort word, which ends every later wait at once.
struct ChainSync {
    unsigned* wait_ctr;   // null: nothing to wait for
    unsigned wait_n;      // workgroups of the producer phase
    unsigned* sig_ctr;    // null: nobody waits for this phase
    unsigned* abort;
    unsigned* wait_rep;   // the producer phase's eight per-XCD "complete" words (32 dwords apart), or null
    int relay;            // this workgroup polls the counter itself and sets its XCD's word
    int mode;             // experiment switches: 2 = no fences, 4 = long sleep
    static constexpr bool weights_first = true;
    __device__ __forceinline__ void wait() const;
    __device__ __forceinline__ void signal() const;
};
constexpr unsigned long long CHAIN_TIMEOUT_TICKS = 5000000ull;   // 50 ms of s_memrealtime (100 MHz)
#define CHAIN_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned chain_xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void chain_wait(const struct ChainSync& s);
__device__ __forceinline__ void chain_signal(const struct ChainSync& s);
__device__ __forceinline__ void chain_wait(const ChainSync& s) {
    if (!s.wait_ctr) return;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned it = 0;
        unsigned* rep = s.wait_rep ? s.wait_rep + chain_xcc_id() * 32u : nullptr;
        for (;;) {
            ++it;
            if (!rep || s.relay || (it & 31u) == 0u) {
                if (__hip_atomic_load(s.wait_ctr, CHAIN_RLX) >= s.wait_n) {
                    if (rep) __hip_atomic_store(rep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
            } else if (__hip_atomic_load(rep, CHAIN_RLX)) break;
            if ((it & 15u) == 0u) {
                if (__hip_atomic_load(s.abort, CHAIN_RLX)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > CHAIN_TIMEOUT_TICKS) { __hip_atomic_store(s.abort, 1u, CHAIN_RLX); break; }
            }
            if (s.mode & 4) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (!(s.mode & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void chain_signal(const ChainSync& s) {
    if (!s.sig_ctr) return;
    if (!(s.mode & 2)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(s.sig_ctr, 1u, CHAIN_RLX);
}

__device__ __forceinline__ void ChainSync::wait() const { chain_wait(*this); }
__device__ __forceinline__ void ChainSync::signal() const { chain_signal(*this); }

// ------------------------------------------------------------------------------------------
// Chain launch: the workgroups of MANY dependent phases in one grid, phase after phase in workgroup order.  The hardware
// hands out workgroups in index order, so every workgroup of phase k is resident (or done) before the first of phase
// k + 1 starts: a waiting workgroup can only wait for workgroups that already run.  Workgroups of later phases take
// the free slots of the chip early, request their weights from HBM, and wait on the producer phase's counter
// (ChainSync) - the launch gap and the weight round trip of every phase but the first are hidden behind its predecessors.
// ------------------------------------------------------------------------------------------
enum { CHK_N1 = 0,      // <1,1,KS4,norm,store>      K = 1024
       CHK_N2,          // <1,2,KS4,norm,store>
       CHK_N22,         // <2,2,KS4,norm,store>
       CHK_G2,          // <1,2,KS4,norm,swiglu>
       CHK_G22,         // <2,2,KS4,norm,swiglu>
       CHK_R4,          // <1,1,KS4,resid>           K = 1024
       CHK_R8,          // <1,1,KS8,resid>           K = 2048
       CHK_R12,         // <1,1,KS12,resid>          K = 3072
       CHK_GEMM_KINDS };

struct ChainPhase {
    WideP p;
    int kind;
    int block0, gx;       // first workgroup of the phase; tiles along N (workgroup i of the phase: bx = i % gx, by = i / gx)
    int wait_idx;         // counter of the producer phase, -1: none
    unsigned wait_n;
    int sig_idx;          // this phase's counter, -1: none
};

constexpr int CHAIN_THREADS = 512;
#ifndef CHAIN_MIN_WAVES
#define CHAIN_MIN_WAVES 4      // waves per SIMD the register allocation must leave room for: two 8-wave workgroups per CU
#endif
constexpr int CHAIN_LDS_FLOATS = wide_lds_floats<2, 2, 8>();

static inline int chain_gemm_tiles(int kind, int& ts, int& nt) {
    switch (kind) {
        case CHK_N1: case CHK_R4: case CHK_R8: case CHK_R12: ts = 1; nt = 1; return 0;
        case CHK_N2: case CHK_G2: ts = 1; nt = 2; return 0;
        case CHK_N22: case CHK_G22: ts = 2; nt = 2; return 0;
    }
    return -1;
}

__device__ __forceinline__ void chain_gemm_phase(const ChainPhase& P, int bx, int by, float* lds, const ChainSync& sy) {
    switch (P.kind) {
        case CHK_N1: wide_gemm_body<1, 1, 8, 4, true, WEPI_STORE, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_N2: wide_gemm_body<1, 2, 8, 4, true, WEPI_STORE, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_N22: wide_gemm_body<2, 2, 8, 4, true, WEPI_STORE, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_G2: wide_gemm_body<1, 2, 8, 4, true, WEPI_SWIGLU, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_G22: wide_gemm_body<2, 2, 8, 4, true, WEPI_SWIGLU, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_R4: wide_gemm_body<1, 1, 8, 4, false, WEPI_RESID, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_R8: wide_gemm_body<1, 1, 8, 8, false, WEPI_RESID, ChainSync>(P.p, bx, by, lds, sy); break;
        case CHK_R12: wide_gemm_body<1, 1, 8, 12, false, WEPI_RESID, ChainSync>(P.p, bx, by, lds, sy); break;
        default: break;
    }
}

// phase_of: [workgroups] the phase of every workgroup; ctrs: one counter per phase, zero at launch; reps: 8 x 32 dwords per
// phase (per-XCD completion words, zero at launch) or null
static __global__ __launch_bounds__(CHAIN_THREADS, CHAIN_MIN_WAVES) void wide_chain_gemm_kernel(const ChainPhase* phases, const unsigned short* phase_of,
                                                                                                unsigned* ctrs, unsigned* reps, unsigned* abort, int mode) {
    __shared__ float lds[CHAIN_LDS_FLOATS];
    const int pi = __builtin_amdgcn_readfirstlane((int)phase_of[blockIdx.x]);
    const ChainPhase& P = phases[pi];
    const int i = (int)blockIdx.x - P.block0;
    ChainSync sy;
    sy.wait_ctr = P.wait_idx >= 0 ? ctrs + P.wait_idx : nullptr;
    sy.wait_n = P.wait_n;
    sy.sig_ctr = P.sig_idx >= 0 ? ctrs + P.sig_idx : nullptr;
    sy.abort = abort;
    sy.wait_rep = (reps && P.wait_idx >= 0) ? reps + (size_t)P.wait_idx * 256 : nullptr;
    sy.relay = i < 8;
    sy.mode = mode;
    chain_gemm_phase(P, i % P.gx, i / P.gx, lds, sy);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static hipStream_t s;
static float time_graph(const std::function<void()>& enqueue, int reps = 20) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / reps;
}

static uint32_t rng_state = 12345u;
static float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }
static bf16_t h_bits(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

struct Shape { int N, K, kind, epi; bool norm; };

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32, L = argc > 2 ? atoi(argv[2]) : 28;
    const int ldm = 64;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int D = 1024, QN = 4096, HD = 2048, F = 3072;
    const Shape sh[4] = {{QN, D, M > 16 ? CHK_N2 : CHK_N1, WEPI_STORE, true}, {D, HD, CHK_R8, WEPI_RESID, false},
                         {2 * F, D, M > 16 ? CHK_G22 : CHK_G2, WEPI_SWIGLU, true}, {D, F, CHK_R12, WEPI_RESID, false}};
    size_t wl = 0;
    for (const Shape& q : sh) wl += (size_t)q.N * q.K;
    for (int cached = 0; cached < 2; ++cached) {
        const int nW = cached ? 1 : L;
        std::vector<bf16_t*> w(nW);
        {
            std::vector<bf16_t> hw(wl);
            for (int l = 0; l < nW; ++l) {
                for (auto& v : hw) v = h_bits(rnd() * 0.06f);
                CK(hipMalloc(&w[l], wl * 2)); CK(hipMemcpy(w[l], hw.data(), wl * 2, hipMemcpyHostToDevice));
            }
        }
        bf16_t *xo, *yo, *go, *gain, *xo0; float* qkv;
        const size_t xo_n = (size_t)(D / 8) * ldm * 8, yo_n = (size_t)(HD / 8) * ldm * 8, go_n = (size_t)(F / 8) * ldm * 8;
        CK(hipMalloc(&xo, xo_n * 2)); CK(hipMalloc(&xo0, xo_n * 2)); CK(hipMalloc(&yo, yo_n * 2)); CK(hipMalloc(&go, go_n * 2)); CK(hipMalloc(&gain, 4096 * 2));
        CK(hipMalloc(&qkv, (size_t)64 * QN * 4));
        {
            std::vector<bf16_t> h(yo_n);
            for (auto& v : h) v = h_bits(rnd());
            CK(hipMemcpy(yo, h.data(), yo_n * 2, hipMemcpyHostToDevice));
            h.resize(xo_n);
            for (auto& v : h) v = h_bits(rnd() * 2.f);
            CK(hipMemcpy(xo0, h.data(), xo_n * 2, hipMemcpyHostToDevice));
            h.resize(4096);
            for (auto& v : h) v = h_bits(1.f + 0.2f * rnd());
            CK(hipMemcpy(gain, h.data(), 4096 * 2, hipMemcpyHostToDevice));
        }
        auto mk = [&](int l, int j) {
            const Shape& q = sh[j];
            size_t off = 0;
            for (int i = 0; i < j; ++i) off += (size_t)sh[i].N * sh[i].K;
            WideP p{};
            p.W = w[cached ? 0 : l] + off; p.ldw = q.K; p.gain = q.norm ? gain : nullptr; p.eps = 1e-6f; p.M = M; p.N = q.N; p.K = q.K;
            p.ldm = ldm; p.ldm_o = ldm;
            if (j == 0) { p.X = xo; p.out_f32 = qkv; p.ldo = QN; }
            if (j == 1) { p.X = yo; p.out_xo = xo; p.resid_xo = xo; }
            if (j == 2) { p.X = xo; p.out_xo = go; }
            if (j == 3) { p.X = go; p.out_xo = xo; p.resid_xo = xo; }
            return p;
        };
        // ---- the chain
        std::vector<ChainPhase> ph;
        std::vector<unsigned short> pof;
        for (int l = 0; l < L; ++l)
            for (int j = 0; j < 4; ++j) {
                ChainPhase c{};
                c.p = mk(l, j); c.kind = sh[j].kind;
                int ts, nt;
                chain_gemm_tiles(c.kind, ts, nt);
                c.gx = sh[j].N / (16 * nt);
                const int gy = (M + 16 * ts - 1) / (16 * ts);
                c.block0 = (int)pof.size();
                const int idx = (int)ph.size();
                c.wait_idx = idx > 0 ? idx - 1 : -1;
                c.wait_n = idx > 0 ? (unsigned)((int)pof.size() - ph.back().block0) : 0u;
                c.sig_idx = idx;
                for (int b = 0; b < c.gx * gy; ++b) pof.push_back((unsigned short)idx);
                ph.push_back(c);
            }
        ChainPhase* dph; unsigned short* dpof; unsigned *ctrs, *abortw;
        CK(hipMalloc(&dph, ph.size() * sizeof(ChainPhase))); CK(hipMemcpy(dph, ph.data(), ph.size() * sizeof(ChainPhase), hipMemcpyHostToDevice));
        CK(hipMalloc(&dpof, pof.size() * 2)); CK(hipMemcpy(dpof, pof.data(), pof.size() * 2, hipMemcpyHostToDevice));
        const size_t sync_words = ph.size() * (1 + 256);
        CK(hipMalloc(&ctrs, sync_words * 4)); CK(hipMalloc(&abortw, 4)); CK(hipMemset(abortw, 0, 4));
        int mode = 0;
        auto run_chain = [&] {
            CK(hipMemsetAsync(ctrs, 0, sync_words * 4, s));
            wide_chain_gemm_kernel<<<(unsigned)pof.size(), CHAIN_THREADS, 0, s>>>(dph, dpof, ctrs, (mode & 1) ? ctrs + ph.size() : nullptr, abortw, mode);
        };
        auto run_launches = [&] {
            for (int l = 0; l < L; ++l) {
                bool ok = true;
                if (M > 16) ok &= wide_gemm_launch<1, 2, true, WEPI_STORE>(mk(l, 0), s); else ok &= wide_gemm_launch<1, 1, true, WEPI_STORE>(mk(l, 0), s);
                ok &= wide_gemm_launch<1, 1, false, WEPI_RESID>(mk(l, 1), s);
                if (M > 16) ok &= wide_gemm_launch<2, 2, true, WEPI_SWIGLU>(mk(l, 2), s); else ok &= wide_gemm_launch<1, 2, true, WEPI_SWIGLU>(mk(l, 2), s);
                {   // K = 3072 on eight waves as in the chain (the product launch splits it over twelve: another summation order)
                    const WideP p3 = mk(l, 3);
                    wide_gemm_kernel<1, 1, 8, 12, false, WEPI_RESID><<<dim3(p3.N / 16, (M + 15) / 16), 512, 0, s>>>(p3);
                }
                if (!ok) { printf("launch refused\n"); exit(1); }
            }
        };
        // ---- same bits?
        std::vector<bf16_t> a(xo_n), b(xo_n), ga(go_n), gb(go_n);
        CK(hipMemcpyAsync(xo, xo0, xo_n * 2, hipMemcpyDeviceToDevice, s));
        run_launches();
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(a.data(), xo, xo_n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ga.data(), go, go_n * 2, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpyAsync(xo, xo0, xo_n * 2, hipMemcpyDeviceToDevice, s));
            CK(hipMemsetAsync(go, 0, go_n * 2, s));
            run_chain();
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(b.data(), xo, xo_n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), go, go_n * 2, hipMemcpyDeviceToHost));
            unsigned ab; CK(hipMemcpy(&ab, abortw, 4, hipMemcpyDeviceToHost));
            int diff = 0;
            for (int m = 0; m < M; ++m) {
                for (int k = 0; k < D; ++k) diff += a[xo_index(m, k, ldm)] != b[xo_index(m, k, ldm)];
                for (int k = 0; k < F; ++k) diff += ga[xo_index(m, k, ldm)] != gb[xo_index(m, k, ldm)];
            }
            printf("chain vs launches (%s weights, rep %d): %d differing values, abort word %u, %zu workgroups in %zu phases\n", cached ? "cached" : "streamed", rep, diff,
                   ab, pof.size(), ph.size());
            bad += diff + (int)ab;
        }
        if (bad) { printf("SELF-CHECK FAILED\n"); return 1; }
        const float tl = time_graph(run_launches);
        printf("M=%2d L=%d %s: launches %8.1f us (%.2f per phase)   weights %.1f MB per layer\n", M, L, cached ? "cached  " : "streamed", tl, tl / (4 * L), wl * 2 / 1e6);
        for (int md : {0, 4, 1, 5, 2, 3, 7}) {      // 1 = per-XCD relay, 2 = no fences (timing only), 4 = long sleep between polls
            mode = md;
            const float tc = time_graph(run_chain);
            printf("    chain mode %d (%s%s%s): %8.1f us (%.2f per phase)\n", md, (md & 1) ? "relay " : "", (md & 2) ? "nofence " : "", (md & 4) ? "longsleep" : "", tc, tc / (4 * L));
        }
        unsigned ab; CK(hipMemcpy(&ab, abortw, 4, hipMemcpyDeviceToHost));
        if (ab) { printf("abort word set after timing\n"); return 1; }
        for (auto q : w) (void)hipFree(q);
        (void)hipFree(xo); (void)hipFree(xo0); (void)hipFree(yo); (void)hipFree(go); (void)hipFree(gain); (void)hipFree(qkv); (void)hipFree(dph); (void)hipFree(dpof); (void)hipFree(ctrs); (void)hipFree(abortw);
    }
    return 0;
}
