// How fast can a FEW workgroups stream weights?  (candidate: fast-stack QKV + attention in one kernel, one block per
// head.)  G blocks x 1024 threads, each block streams RPB rows of a [N][1024] bf16 matrix (GEMV with x), all loads of a
// wave issued up front.   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I fish-tts_amd/csrc tools/mb_fewblocks.hip -o /tmp/mb_few
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
#include "common.h"
using namespace ft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int RPW>   // rows per wave
__global__ __launch_bounds__(1024) void few_gemv(const bf16_t* W, const float* x, float* out, int rows_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * rows_per_block + wave * RPW;
    U4 raw[RPW][2];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) raw[r][t] = *reinterpret_cast<const U4*>(W + (size_t)(row0 + r) * 1024 + t * 512 + lane * 8);
    float xv[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const float4 f = *reinterpret_cast<const float4*>(x + t * 512 + lane * 8 + j);
            xv[t][j] = f.x; xv[t][j + 1] = f.y; xv[t][j + 2] = f.z; xv[t][j + 3] = f.w;
        }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float wv[8];
            Vec<bf16_t>::unpack(raw[r][t], wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(wv[j], xv[t][j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[row0 + r] = acc;
    }
}

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
    return ms * 1e3f / (reps * launches_per_graph);
}

int main() {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int L = 28;
    std::vector<bf16_t*> w(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&w[l], (size_t)8192 * 1024 * 2)); CK(hipMemset(w[l], 0x11, (size_t)8192 * 1024 * 2)); }
    float *x, *out; CK(hipMalloc(&x, 4096)); CK(hipMalloc(&out, 8192 * 4)); CK(hipMemset(x, 0, 4096));
    auto run = [&](const char* name, int G, int rpb, auto launch, bool hot) {
        float us = time_graph([&] { for (int l = 0; l < L; ++l) launch(hot ? w[l % 4] : w[l]); }, L);
        printf("%-28s G=%3d blocks x %3d rows (%4d KB/block, %5.1f MB total) %s: %6.2f us  -> %6.0f GB/s total, %5.0f GB/s per block\n", name, G, rpb,
               rpb * 2, G * rpb * 2048 / 1e6, hot ? "cache-resident" : "HBM          ", us, G * rpb * 2048.0 / us / 1e3, rpb * 2048.0 / us / 1e3);
    };
    for (bool hot : {false, true}) {
        run("fast qkv+attn per q head", 16, 192, [&](bf16_t* W) { few_gemv<12><<<16, 1024, 0, s>>>(W, x, out, 192); }, hot);
        run("fast qkv+attn per kv group", 8, 256, [&](bf16_t* W) { few_gemv<16><<<8, 1024, 0, s>>>(W, x, out, 256); }, hot);
        run("32 blocks", 32, 96, [&](bf16_t* W) { few_gemv<6><<<32, 1024, 0, s>>>(W, x, out, 96); }, hot);
        run("slow qkv+attn per q head", 16, 384, [&](bf16_t* W) { few_gemv<24><<<16, 1024, 0, s>>>(W, x, out, 384); }, hot);
        run("64 blocks", 64, 64, [&](bf16_t* W) { few_gemv<4><<<64, 1024, 0, s>>>(W, x, out, 64); }, hot);
        run("256 blocks (reference)", 256, 16, [&](bf16_t* W) { few_gemv<1><<<256, 1024, 0, s>>>(W, x, out, 16); }, hot);
    }
    return 0;
}
