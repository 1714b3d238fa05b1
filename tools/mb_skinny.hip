// Microbenchmark of the skinny MFMA GEMM at the decode shapes (build + run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I fish-tts_amd/csrc tools/mb_skinny.hip -o /tmp/mb_skinny && /tmp/mb_skinny
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>

#include "ar_kernels.h"
#include "codec_kernels.h"
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

int main(int argc, char** argv) {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int L = 28;
    const int PAD = argc > 1 ? atoi(argv[1]) : 0;   // row padding in elements (weights and activations)
    const size_t WMAX = (size_t)6144 * (1024 + 512) + (size_t)1024 * 4096;
    std::vector<bf16_t*> w(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&w[l], WMAX * 2)); CK(hipMemset(w[l], 0x11, WMAX * 2)); }
    bf16_t *xb, *ob; float *xf, *of; bf16_t* gain;
    CK(hipMalloc(&xb, 64 * 4096 * 2)); CK(hipMalloc(&ob, 64 * 6144 * 2)); CK(hipMalloc(&xf, 64 * 3072 * 4)); CK(hipMalloc(&of, 64 * 6144 * 4));
    CK(hipMalloc(&gain, 4096 * 2)); CK(hipMemset(gain, 0x3f, 4096 * 2));
    CK(hipMemset(xb, 0x11, 64 * 4096 * 2)); printf("row padding %d elements\n", PAD); CK(hipMemset(xf, 0, 64 * 3072 * 4)); CK(hipMemset(of, 0, 64 * 6144 * 4));
    struct Shape { const char* name; int N, K, act, resid; };
    const Shape shapes[] = {{"qkv  N=4096 K=1024", 4096, 1024, ACT_NONE, 0}, {"wo   N=1024 K=2048", 1024, 2048, ACT_NONE, 1},
                            {"w13  N=6144 K=1024", 6144, 1024, ACT_SWIGLU, 0}, {"w2   N=1024 K=3072", 1024, 3072, ACT_NONE, 1},
                            {"fqkv N=2048 K=1024", 2048, 1024, ACT_NONE, 0}, {"fwo  N=1024 K=1024", 1024, 1024, ACT_NONE, 1}};
    for (int M : {8, 16, 32, 64}) {
        for (const Shape& sh : shapes) {
            auto mk = [&](int l) {
                TapGemmP p{};
                p.X = xb; p.ldx = sh.K + PAD; p.ldw = sh.K + PAD; p.T_in = M; p.W = w[l]; p.ntap = 1; p.M = M; p.N = sh.N; p.K = sh.K; p.n_mod = sh.N;
                p.act = sh.act; p.round_lin = 1;
                if (sh.act == ACT_SWIGLU) { p.out_bf = ob; p.ldo = sh.N / 2; }
                else { p.out_f32 = of; p.ldo = sh.N; if (sh.resid) { p.resid_f32 = of; p.ldr = sh.N; p.round_f32_out = 1; } }
                return p;
            };
            float us = time_graph([&] {
                for (int l = 0; l < L; ++l) {
                    TapGemmP p = mk(l);
                    if (M <= 16) skinny_gemm_launch<1>(p, 1, s);
                    else if (M <= 32) skinny_gemm_launch<2>(p, 1, s);
                    else skinny_gemm_launch<4>(p, (M + 63) / 64, s);
                }
            }, L);
            printf("M=%2d %s: %6.2f us  (%.0f GB/s of weights)\n", M, sh.name, us, (double)sh.N * sh.K * 2 / us / 1e3);
        }
        float us = time_graph([&] {
            for (int l = 0; l < L; ++l) rmsnorm_llama_rows_kernel<bf16_t, true><<<M, 256, 0, s>>>(xf, gain, 1e-6f, 1024, xb);
        }, L);
        printf("M=%2d rmsnorm rows D=1024: %6.2f us\n", M, us);
    }
    return 0;
}
