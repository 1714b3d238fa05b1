// Microbenchmark + self-check of the lock-step batch GEMM (fish-tts_amd/csrc/wide_kernels.h) at the decode shapes:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I fish-tts_amd/csrc tools/mb_wide.hip -o tools/bin/mb_wide
//   tools/bin/mb_wide            (on the GPU box)
// Every variant is first checked against a host computation with the same rounding points, then timed as 28 launches
// per graph on 28 distinct weight buffers (HBM-streamed: the slow stack) and on ONE buffer (cache-resident: the codebook
// loop), beside the skinny kernel it replaces.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

#include "ar_kernels.h"
#include "codec_kernels.h"
#include "wide_kernels.h"
using namespace ft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static hipStream_t s;
static float time_graph(const std::function<void()>& enqueue, int launches_per_graph, int reps = 20) {
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
    return ms * 1e3f / (reps * launches_per_graph);
}

static float h_bf(float x) {   // round to nearest even, as the device cast
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return x;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float y; memcpy(&y, &u, 4);
    return y;
}
static bf16_t h_bits(float x) { float y = h_bf(x); uint32_t u; memcpy(&u, &y, 4); return (bf16_t)(u >> 16); }
static float h_val(bf16_t b) { uint32_t u = (uint32_t)b << 16; float y; memcpy(&y, &u, 4); return y; }
static uint32_t rng_state = 12345u;
static float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }

static int g_nt = 1;
template <int TS, bool NORM, int EPI>
static bool launch_w(const WideP& p) { return g_nt == 2 ? wide_gemm_launch<TS, 2, NORM, EPI>(p, s) : wide_gemm_launch<TS, 1, NORM, EPI>(p, s); }

static int check(int M, int N, int K, bool norm, int epi, int TS) {
    const int ldm = 32 + 16 * (M > 32) * 2, No = epi == WEPI_SWIGLU ? N / 2 : N;
    std::vector<bf16_t> hx((size_t)(K / 8) * ldm * 8, 0), hw((size_t)N * K), hg(K), hr((size_t)(No / 8 + 1) * ldm * 8, 0);
    std::vector<float> hb(N);
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hx[xo_index(m, k, ldm)] = h_bits(rnd() * 4.f);
    for (auto& v : hw) v = h_bits(rnd() * 0.1f);
    for (auto& v : hg) v = h_bits(1.f + rnd());
    for (auto& v : hb) v = h_bf(rnd());
    if (epi == WEPI_RESID) for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) hr[xo_index(m, n, ldm)] = h_bits(rnd() * 4.f);
    bf16_t *dx, *dw, *dg, *dr, *dox; float *db, *dof;
    CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dg, hg.size() * 2)); CK(hipMalloc(&dr, hr.size() * 2));
    CK(hipMalloc(&dox, hr.size() * 2)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dof, (size_t)M * N * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dox, 0, hr.size() * 2));
    WideP p{};
    p.X = dx; p.ldm = ldm; p.W = dw; p.ldw = K; p.gain = norm ? dg : nullptr; p.eps = 1e-6f; p.bias = db; p.M = M; p.N = N; p.K = K;
    p.out_f32 = dof; p.ldo = N; p.out_xo = dox; p.ldm_o = ldm; p.resid_xo = dr;
    bool ok = false;
#define GO(TSV) (norm ? (epi == WEPI_SWIGLU ? launch_w<TSV, true, WEPI_SWIGLU>(p) : launch_w<TSV, true, WEPI_STORE>(p)) \
                      : (epi == WEPI_RESID ? launch_w<TSV, false, WEPI_RESID>(p) : launch_w<TSV, false, WEPI_STORE>(p)))
    if (TS == 1) ok = GO(1); else if (TS == 2) ok = GO(2); else ok = GO(4);
#undef GO
    if (!ok) { printf("no instantiation for K=%d\n", K); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> of((size_t)M * N); std::vector<bf16_t> ox(hr.size());
    CK(hipMemcpy(of.data(), dof, of.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ox.data(), dox, ox.size() * 2, hipMemcpyDeviceToHost));
    // host reference (double accumulation: the device's f32 order differs, so compare with a bf16-step tolerance)
    int bad = 0; double worst = 0;
    for (int m = 0; m < M; ++m) {
        std::vector<float> xn(K);
        double ss = 0;
        for (int k = 0; k < K; ++k) { const float v = h_val(hx[xo_index(m, k, ldm)]); ss += (double)v * v; xn[k] = v; }
        if (norm) {
            const float inv = 1.0f / sqrtf((float)(ss / K) + 1e-6f);
            for (int k = 0; k < K; ++k) xn[k] = h_bf(h_bf(xn[k] * inv) * h_val(hg[k]));
        }
        std::vector<float> lin(N);
        for (int n = 0; n < N; ++n) {
            double a = 0;
            for (int k = 0; k < K; ++k) a += (double)xn[k] * h_val(hw[(size_t)n * K + k]);
            lin[n] = h_bf((float)a + hb[n]);
        }
        for (int n = 0; n < No; ++n) {
            float want, got;
            if (epi == WEPI_SWIGLU) {
                const float gt = lin[2 * n], up = lin[2 * n + 1];
                want = h_bf(h_bf(gt / (1.0f + expf(-gt))) * up); got = h_val(ox[xo_index(m, n, ldm)]);
            } else if (epi == WEPI_RESID) {
                want = h_bf(lin[n] + h_val(hr[xo_index(m, n, ldm)])); got = h_val(ox[xo_index(m, n, ldm)]);
            } else { want = lin[n]; got = of[(size_t)m * N + n]; }
            const double err = fabs((double)want - got), tol = 0.02 * fmax(1.0, fabs(want));
            worst = fmax(worst, err / fmax(1.0, fabs(want)));
            if (!(err <= tol)) { if (bad < 4) printf("  mismatch m=%d n=%d want %g got %g\n", m, n, want, got); ++bad; }
        }
    }
    printf("check M=%2d N=%4d K=%4d norm=%d epi=%d TS=%d: %s (worst rel %.4f)\n", M, N, K, (int)norm, epi, TS, bad ? "FAIL" : "ok", worst);
    hipFree(dx); hipFree(dw); hipFree(dg); hipFree(dr); hipFree(dox); hipFree(db); hipFree(dof);
    return bad;
}

int main(int argc, char** argv) {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int bad = 0;
    for (int M : {5, 16, 19, 32}) {
        const int TS = M <= 16 ? 1 : 2;
        bad += check(M, 64, 1024, true, WEPI_STORE, TS);
        bad += check(M, 64, 1024, true, WEPI_SWIGLU, TS);
        bad += check(M, 48, 2048, false, WEPI_RESID, TS);
        bad += check(M, 32, 3072, false, WEPI_RESID, TS);
        bad += check(M, 32, 1024, false, WEPI_STORE, TS);
    }
    bad += check(32, 32, 3072, false, WEPI_RESID, 1);     // M split over two workgroups
    bad += check(32, 64, 1024, true, WEPI_SWIGLU, 1);
    bad += check(50, 32, 1024, true, WEPI_STORE, 4);
    g_nt = 2;
    bad += check(32, 64, 1024, true, WEPI_STORE, 2);
    bad += check(19, 64, 1024, true, WEPI_SWIGLU, 2);
    bad += check(32, 64, 2048, false, WEPI_RESID, 1);
    bad += check(9, 64, 1024, true, WEPI_SWIGLU, 1);
    g_nt = 1;
    if (bad) { printf("SELF-CHECK FAILED\n"); return 1; }
    if (argc > 1 && atoi(argv[1]) == 0) return 0;

    const int L = 28;
    const size_t WMAX = (size_t)6144 * 1024 + (size_t)1024 * 4096;
    std::vector<bf16_t*> w(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&w[l], WMAX * 2)); CK(hipMemset(w[l], 0x11, WMAX * 2)); }
    bf16_t *xb, *ob, *gain; float* of;
    CK(hipMalloc(&xb, 64 * 4096 * 2)); CK(hipMalloc(&ob, 64 * 6144 * 2)); CK(hipMalloc(&of, 64 * 6144 * 4)); CK(hipMalloc(&gain, 4096 * 2));
    CK(hipMemset(xb, 0x11, 64 * 4096 * 2)); CK(hipMemset(ob, 0x11, 64 * 6144 * 2)); CK(hipMemset(of, 0, 64 * 6144 * 4)); CK(hipMemset(gain, 0x3f, 4096 * 2));
    struct Shape { const char* name; int N, K; bool norm; int epi; };
    const Shape shapes[] = {{"qkv  N=4096 K=1024 norm      ", 4096, 1024, true, WEPI_STORE}, {"wo   N=1024 K=2048 resid     ", 1024, 2048, false, WEPI_RESID},
                            {"w13  N=6144 K=1024 norm swiglu", 6144, 1024, true, WEPI_SWIGLU}, {"w2   N=1024 K=3072 resid     ", 1024, 3072, false, WEPI_RESID},
                            {"fqkv N=2048 K=1024 norm      ", 2048, 1024, true, WEPI_STORE}, {"fwo  N=1024 K=1024 resid     ", 1024, 1024, false, WEPI_RESID},
                            {"fhead N=1024 K=1024 norm     ", 1024, 1024, true, WEPI_STORE}};
    for (int M : {8, 16, 32}) {
        for (const Shape& sh : shapes) {
            for (int same = 0; same < 2; ++same) {
                auto mkw = [&](int l) {
                    WideP p{};
                    p.X = xb; p.ldm = 32; p.W = w[same ? 0 : l]; p.ldw = sh.K; p.gain = sh.norm ? gain : nullptr; p.eps = 1e-6f; p.M = M; p.N = sh.N; p.K = sh.K;
                    p.out_f32 = of; p.ldo = sh.N; p.out_xo = ob; p.ldm_o = 32; p.resid_xo = ob;
                    return p;
                };
                auto run = [&](int TS) {
                    return time_graph([&] {
                        for (int l = 0; l < L; ++l) {
                            WideP p = mkw(l);
#define GO(TSV) (sh.norm ? (sh.epi == WEPI_SWIGLU ? launch_w<TSV, true, WEPI_SWIGLU>(p) : launch_w<TSV, true, WEPI_STORE>(p)) \
                         : (sh.epi == WEPI_RESID ? launch_w<TSV, false, WEPI_RESID>(p) : launch_w<TSV, false, WEPI_STORE>(p)))
                            if (TS == 1) GO(1); else GO(2);
#undef GO
                        }
                    }, L);
                };
                g_nt = 1;
                const float t1 = run(1), t2 = M > 16 ? run(2) : 0.f;
                g_nt = 2;
                const float u1 = run(1), u2 = M > 16 ? run(2) : 0.f;
                g_nt = 1;
                // the skinny kernel on the same shape (row-major operands; norm as its own launch is NOT included)
                const float tk = time_graph([&] {
                    for (int l = 0; l < L; ++l) {
                        TapGemmP q{};
                        q.X = xb; q.ldx = sh.K; q.ldw = sh.K; q.T_in = M; q.W = w[same ? 0 : l]; q.ntap = 1; q.M = M; q.N = sh.N; q.K = sh.K; q.n_mod = sh.N;
                        q.act = sh.epi == WEPI_SWIGLU ? ACT_SWIGLU : ACT_NONE; q.round_lin = 1;
                        if (sh.epi == WEPI_SWIGLU) { q.out_bf = ob; q.ldo = sh.N / 2; }
                        else { q.out_f32 = of; q.ldo = sh.N; if (sh.epi == WEPI_RESID) { q.resid_f32 = of; q.ldr = sh.N; q.round_f32_out = 1; } }
                        if (M <= 16) skinny_gemm_launch<1>(q, 1, s); else skinny_gemm_launch<2>(q, 1, s);
                    }
                }, L);
                printf("M=%2d %s %s: NT=1 TS=1 %5.2f TS=2 %5.2f | NT=2 TS=1 %5.2f TS=2 %5.2f | skinny %5.2f us  (%.1f MB)\n", M, sh.name, same ? "cached " : "streamed",
                       t1, t2, u1, u2, tk, (double)sh.N * sh.K * 2 / 1e6);
            }
        }
    }
    return 0;
}
