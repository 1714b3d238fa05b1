// PMC target: the decode GEMV launches of one s1-mini frame, eagerly (no hipGraph: rocprofv3 --pmc crashes on
// graph launches on this stack), distinct weight buffers per layer so every byte comes from HBM.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I fish-tts_amd/csrc tools/pmc_gemv.hip -o /tmp/pmc_gemv
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/pmc_gemv
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "ar_kernels.h"
using namespace ft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT, int R>
static void launch(GemvP p) { gemv_kernel<bf16_t, NT, R, true><<<dim3((p.N + 4 * R - 1) / (4 * R), 1), 256>>>(p); }

int main() {
    const int L = 28, D = 1024, FF = 3072, QKV = 4096, V = 155776;
    std::vector<bf16_t*> wqkv(L), wo(L), w13(L), w2(L);
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&wqkv[l], (size_t)QKV * D * 2)); CK(hipMalloc(&wo[l], (size_t)D * 2048 * 2));
        CK(hipMalloc(&w13[l], (size_t)2 * FF * D * 2)); CK(hipMalloc(&w2[l], (size_t)D * FF * 2));
        CK(hipMemset(wqkv[l], 0x11, (size_t)QKV * D * 2)); CK(hipMemset(wo[l], 0x11, (size_t)D * 2048 * 2));
        CK(hipMemset(w13[l], 0x11, (size_t)2 * FF * D * 2)); CK(hipMemset(w2[l], 0x11, (size_t)D * FF * 2));
    }
    bf16_t* head; CK(hipMalloc(&head, (size_t)V * D * 2)); CK(hipMemset(head, 0x11, (size_t)V * D * 2));
    float *x, *qkv, *y, *g, *logits; bf16_t* gain;
    CK(hipMalloc(&x, D * 4)); CK(hipMalloc(&qkv, QKV * 4)); CK(hipMalloc(&y, 2048 * 4)); CK(hipMalloc(&g, FF * 4));
    CK(hipMalloc(&logits, (size_t)V * 4)); CK(hipMalloc(&gain, 4096 * 2)); CK(hipMemset(gain, 0x3f, 4096 * 2));
    CK(hipMemset(x, 0, D * 4)); CK(hipMemset(y, 0, 2048 * 4)); CK(hipMemset(g, 0, FF * 4));
    auto mk = [&](const void* W, const float* xin, int ldx, float* out, int ldo, int N, int K, int pro, int epi) {
        GemvP p{}; p.W = W; p.x = xin; p.ldx = ldx; p.out = out; p.ldo = ldo; p.N = N; p.K = K; p.pro = pro; p.epi = epi;
        p.gain = gain; p.eps = 1e-6f; p.resid = out; p.ldr = ldo; p.nt = 1; return p;
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int l = 0; l < L; ++l) {
            launch<2, 2>(mk(wqkv[l], x, D, qkv, QKV, QKV, D, PRO_RMSNORM, EPI_STORE));
            launch<4, 1>(mk(wo[l], y, 2048, x, D, D, 2048, PRO_NONE, EPI_RESID));
            launch<2, 2>(mk(w13[l], x, D, g, FF, 2 * FF, D, PRO_RMSNORM, EPI_SWIGLU));
            launch<6, 1>(mk(w2[l], g, FF, x, D, D, FF, PRO_NONE, EPI_RESID));
        }
        launch<2, 4>(mk(head, x, D, logits, V, V, D, PRO_RMSNORM, EPI_STORE));
        CK(hipDeviceSynchronize());
    }
    printf("done\n");
    return 0;
}
