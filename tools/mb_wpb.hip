// Waves per block for the B=1 GEMV: same wave program as gemv_kernel, blocks of 1/2/4/8/16 waves.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I fish-tts_amd/csrc tools/mb_wpb.hip -o /tmp/mb_wpb && /tmp/mb_wpb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
#include "ar_kernels.h"
using namespace ft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT, int R, int WPB>
__global__ __launch_bounds__(WPB * 64) void gemv_wpb(GemvP p) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * WPB + (threadIdx.x >> 6)) * R;
    if (row0 >= p.N) return;
    const float* x = p.x;
    gemv_rows<bf16_t, NT, R, true>(p, 0, row0, lane, [&](int k, float(&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const float4 f = *reinterpret_cast<const float4*>(x + k + j);
            v[j] = f.x; v[j + 1] = f.y; v[j + 2] = f.z; v[j + 3] = f.w;
        }
    });
}
static hipStream_t s;
static float time_graph(const std::function<void()>& enqueue, int n, int reps = 20) {
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
    return ms * 1e3f / (reps * n);
}
template <int NT, int R, int WPB>
static void go(const char* name, std::vector<bf16_t*>& w, GemvP base) {
    const int L = (int)w.size();
    float us = time_graph([&] {
        for (int l = 0; l < L; ++l) {
            GemvP p = base; p.W = w[l];
            gemv_wpb<NT, R, WPB><<<(p.N + WPB * R - 1) / (WPB * R), WPB * 64, 0, s>>>(p);
        }
    }, L);
    printf("%-22s R=%d waves/block=%2d: %5.2f us\n", name, R, WPB, us);
}
int main() {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int L = 28;
    std::vector<bf16_t*> w(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&w[l], (size_t)6144 * 1024 * 2)); CK(hipMemset(w[l], 0x11, (size_t)6144 * 1024 * 2)); }
    float *x, *out; bf16_t* gain;
    CK(hipMalloc(&x, 3072 * 4)); CK(hipMalloc(&out, 8192 * 4)); CK(hipMalloc(&gain, 4096 * 2));
    CK(hipMemset(x, 0, 3072 * 4)); CK(hipMemset(out, 0, 8192 * 4)); CK(hipMemset(gain, 0x3f, 4096 * 2));
    auto mk = [&](int N, int K, int pro, int epi) {
        GemvP p{}; p.x = x; p.ldx = K; p.out = out; p.ldo = N; p.N = N; p.K = K; p.pro = pro; p.epi = epi; p.gain = gain; p.eps = 1e-6f;
        p.resid = out; p.ldr = N; p.nt = 1; return p;
    };
#define SWEEP(NAME, NT, R, P) go<NT, R, 1>(NAME, w, P); go<NT, R, 2>(NAME, w, P); go<NT, R, 4>(NAME, w, P); go<NT, R, 8>(NAME, w, P); go<NT, R, 16>(NAME, w, P);
    SWEEP("qkv N=4096 K=1024 norm", 2, 2, mk(4096, 1024, PRO_RMSNORM, EPI_STORE))
    SWEEP("qkv N=4096 K=1024 norm", 2, 1, mk(4096, 1024, PRO_RMSNORM, EPI_STORE))
    SWEEP("w13 N=6144 swiglu", 2, 2, mk(6144, 1024, PRO_RMSNORM, EPI_SWIGLU))
    SWEEP("wo N=1024 K=2048 res", 4, 1, mk(1024, 2048, PRO_NONE, EPI_RESID))
    SWEEP("w2 N=1024 K=3072 res", 6, 1, mk(1024, 3072, PRO_NONE, EPI_RESID))
    return 0;
}
