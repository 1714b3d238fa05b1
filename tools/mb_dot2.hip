// v_dot2c_f32_bf16 on gfx950: which arithmetic is it, and how fast?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/mb_dot2.hip -o tools/bin/mb_dot2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
__device__ __forceinline__ float dot2(unsigned w, unsigned x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, w), __builtin_bit_cast(bf2, x), acc, false);
}
__global__ void sem_kernel(const unsigned* w, const unsigned* x, const float* c, float* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = dot2(w[i], x[i], c[i]);
}
template <int MODE>
__global__ void rate_kernel(const unsigned* w, float* o, int iters) {
    unsigned a = w[threadIdx.x], b = w[threadIdx.x + 64];
    float acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = (float)k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) acc[k] = dot2(a, b, acc[k]);
            else {
                acc[k] = fmaf(__uint_as_float(a << 16), __uint_as_float(b << 16), acc[k]);
                acc[k] = fmaf(__uint_as_float(a & 0xffff0000u), __uint_as_float(b & 0xffff0000u), acc[k]);
            }
        }
        a += 0x10001u;
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float bf(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 20;
    std::vector<unsigned> w(n), x(n);
    std::vector<float> c(n), o(n);
    srand(1);
    auto rbf = [&]() {   // a random finite bf16 of moderate exponent
        unsigned e = 110 + rand() % 30, m = rand() & 127, s = rand() & 1;
        return (s << 15) | (e << 7) | m;
    };
    for (int i = 0; i < n; ++i) {
        w[i] = rbf() | (rbf() << 16); x[i] = rbf() | (rbf() << 16);
        c[i] = (i % 3 == 0) ? 0.f : ldexpf((float)(rand() % 2000001 - 1000000) / 1000000.f, rand() % 20 - 10);
    }
    unsigned *dw, *dx; float *dc, *dout;
    hipMalloc(&dw, n * 4); hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    sem_kernel<<<n / 256, 256>>>(dw, dx, dc, dout, n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long eq[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const float a0 = bf(w[i] & 0xffff), a1 = bf(w[i] >> 16), b0 = bf(x[i] & 0xffff), b1 = bf(x[i] >> 16);
        const float f0 = fmaf(a1, b1, fmaf(a0, b0, c[i]));                 // lo first, two fused steps
        const float f1 = fmaf(a0, b0, fmaf(a1, b1, c[i]));                 // hi first
        const float f2 = (float)((double)a0 * b0 + (double)a1 * b1 + (double)c[i]);   // exact sum, one rounding (double is exact enough here)
        const float f3 = (a0 * b0 + a1 * b1) + c[i];                       // products exact in f32; sum of products rounded, then + c
        const float f4 = fmaf(a0, b0, a1 * b1) + c[i];
        const float f5 = c[i] + a0 * b0 + a1 * b1;
        const float cand[6] = {f0, f1, f2, f3, f4, f5};
        for (int k = 0; k < 6; ++k) eq[k] += memcmp(&cand[k], &o[i], 4) == 0;
    }
    printf("dot2c == fma(hi, fma(lo, c)) %ld / %d; fma(lo, fma(hi, c)) %ld; exact sum one rounding %ld; (p0 + p1) + c %ld; fma(p0, p1') + c %ld; (c + p0) + p1 %ld\n",
           eq[0], n, eq[1], eq[2], eq[3], eq[4], eq[5]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate_kernel<0><<<256, 64>>>(dw, dout, iters); else rate_kernel<1><<<256, 64>>>(dw, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms for %d x 8 independent chains per wave (one wave per CU) = %.2f ns per step of 8\n", mode == 0 ? "dot2c      " : "2 x v_fmac ", ms, iters, ms * 1e6 / iters);
    }
    return 0;
}
