// Lock-step batches of 5..64 utterances (bf16): every Linear of a decode frame as ONE weight-streaming MFMA launch with
// the neighbouring row operations folded in, five launches per transformer layer (llama.py:229-283, 322-331):
//     RMSNorm + Wqkv | qk-norm, RoPE, K/V append, attention | Wo + residual | RMSNorm + W13 + SwiGLU | W2 + residual
//
// Activation operands live in HBM in OCTET-MAJOR form  Xo[k / 8][ldm][8]  (bf16): the 16-byte piece (8 consecutive k) of
// row m sits beside the same piece of row m + 1.  A v_mfma_f32_16x16x32_bf16 A-fragment (16 rows x 8 k per 16 lanes)
// is then ONE contiguous 256-byte run per 16 lanes, loaded straight into the operand registers: no LDS staging, no
// barrier in front of the matrix instructions, and a producer's 16-column output tile is two contiguous 16 B x M runs.
// The residual stream itself is kept in this form (its values are bf16-exact at every reference rounding point), so the
// fused RMSNorm (llama.py:172-177) reads the very operand registers: sum of squares over the wave's K slice, one LDS
// exchange between the K-split waves, x -> round(round(x / rms) * gain) in registers.
//
// Work split: one workgroup per 16 weight rows (N / 16 workgroups x M splits), NW waves split K (KS steps of 32 each,
// compile-time: every load of the launch is issued before the first use = one memory round trip per launch); the NW
// partial tiles meet in LDS and are summed in wave order (deterministic).  Weight bytes are read once per M split.
#pragma once
#include "common.h"

namespace ft {

typedef short wk_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wk_f32x4 __attribute__((ext_vector_type(4)));

enum { WEPI_STORE = 0, WEPI_SWIGLU = 1, WEPI_RESID = 2 };

struct WideP {
    const bf16_t* X;        // Xo[K / 8][ldm][8], row 0 = the batch's first row
    int ldm;
    const bf16_t* W;        // [N][ldw]
    long ldw;
    const bf16_t* gain;     // NORM: RMSNorm gain [K]
    float eps;
    const float* bias;      // [N] or null
    int M, N, K;
    float* out_f32;         // WEPI_STORE: [M][ldo] f32 holding bf16-rounded values
    long ldo;
    bf16_t* out_xo;         // WEPI_SWIGLU: Xo[(N / 2) / 8][ldm_o][8]; WEPI_RESID: Xo[N / 8][ldm_o][8]
    int ldm_o;
    const bf16_t* resid_xo; // WEPI_RESID: the residual stream (same form and stride as out_xo; may alias it)
};

// octet-major element address
__device__ __host__ __forceinline__ size_t xo_index(int m, int k, int ldm) { return ((size_t)(k >> 3) * ldm + m) * 8 + (k & 7); }

template <int TS, int NW, int KS, bool NORM, int EPI>
__global__ __launch_bounds__(NW * 64) void wide_gemm_kernel(WideP p) {
    __shared__ float ssw[NW][TS * 16];
    __shared__ float Cs[NW][TS * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (TS * 16);
    const int kw = wave * (KS * 32);
    // ---- every load of the launch, issued back to back
    U4 w[KS], x[TS][KS], g[KS];
    {
        const bf16_t* wrow = p.W + (size_t)(n0 + fr) * p.ldw + kw + fq * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) w[s] = *reinterpret_cast<const U4*>(wrow + s * 32);
    }
#pragma unroll
    for (int j = 0; j < TS; ++j) {
        const int row = m0 + j * 16 + fr;
        const bool on = row < p.M;
        const bf16_t* xr = p.X + ((size_t)((kw >> 3) + fq) * p.ldm + (on ? row : 0)) * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            x[j][s] = on ? *reinterpret_cast<const U4*>(xr + (size_t)s * 4 * p.ldm * 8) : U4{0u, 0u, 0u, 0u};
    }
    if constexpr (NORM) {
#pragma unroll
        for (int s = 0; s < KS; ++s) g[s] = *reinterpret_cast<const U4*>(p.gain + kw + s * 32 + fq * 8);
    }
    float inv[TS];
    if constexpr (NORM) {
        // sum of squares of this wave's K slice, per row: lanes fr, fr + 16, fr + 32, fr + 48 hold the four octets of a step
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            float ss = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                float v[8];
                Vec<bf16_t>::unpack(x[j][s], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
            }
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (fq == 0) ssw[wave][j * 16 + fr] = ss;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            float ss = ssw[0][j * 16 + fr];
#pragma unroll
            for (int q = 1; q < NW; ++q) ss += ssw[q][j * 16 + fr];
            inv[j] = rsqrt_exact(ss / (float)p.K + p.eps);
        }
    }
    wk_f32x4 acc[TS];
#pragma unroll
    for (int j = 0; j < TS; ++j) acc[j] = wk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        wk_bf16x8 b;
        __builtin_memcpy(&b, &w[s], 16);
        float gv[8];
        if constexpr (NORM) Vec<bf16_t>::unpack(g[s], gv);
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            wk_bf16x8 a;
            if constexpr (NORM) {   // x -> round(round(x / rms) * gain): the two roundings of llama.py:172-177
                float xv[8];
                Vec<bf16_t>::unpack(x[j][s], xv);
                U4 o;
                uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float y0 = round_bf16(xv[e] * inv[j]) * gv[e];
                    const float y1 = round_bf16(xv[e + 1] * inv[j]) * gv[e + 1];
                    ow[e >> 1] = (uint32_t)f32_to_bf16_bits(y0) | ((uint32_t)f32_to_bf16_bits(y1) << 16);
                }
                __builtin_memcpy(&a, &o, 16);
            } else {
                __builtin_memcpy(&a, &x[j][s], 16);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
    }
    // lane holds C[row = j*16 + 4*fq + r][n = fr]
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[wave][j * 16 + fq * 4 + r][fr] = acc[j][r];
    __syncthreads();
    for (int e = tid; e < TS * 256; e += NW * 64) {
        const int row = e >> 4, c = e & 15;       // 16 consecutive lanes finish one row's 16 columns
        const int m = m0 + row, n = n0 + c;
        float v = Cs[0][row][c];
#pragma unroll
        for (int q = 1; q < NW; ++q) v += Cs[q][row][c];
        if (p.bias) v += p.bias[n];
        v = round_bf16(v);                        // the nn.Linear output of a bf16 model
        if constexpr (EPI == WEPI_SWIGLU) {
            // weight rows (2i, 2i+1) = (gate, up) of column i: the partner sits in the neighbouring lane (llama.py:322-331)
            const float other = dpp_f<DPP_XOR1>(v);
            if ((c & 1) == 0 && m < p.M) {
                const float sg = round_bf16(v / (1.0f + expf(-v)));
                p.out_xo[((size_t)blockIdx.x * p.ldm_o + m) * 8 + (c >> 1)] = f32_to_bf16_bits(sg * other);
            }
        } else if constexpr (EPI == WEPI_RESID) {
            if (m < p.M) {
                const size_t oi = ((size_t)(n >> 3) * p.ldm_o + m) * 8 + (n & 7);
                v += bf16_bits_to_f32(p.resid_xo[oi]);
                p.out_xo[oi] = f32_to_bf16_bits(v);
            }
        } else {
            if (m < p.M) p.out_f32[(size_t)m * p.ldo + n] = v;
        }
    }
}

// K split: 128..256 contraction elements per wave where the width allows (one memory round trip, registers for every load)
template <int TS, bool NORM, int EPI>
static inline bool wide_gemm_launch(const WideP& p, hipStream_t st) {
    if (p.N % 16 != 0 || p.M < 1) return false;
    const dim3 grid(p.N / 16, (p.M + TS * 16 - 1) / (TS * 16));
    if (p.K == 1024) wide_gemm_kernel<TS, 8, 4, NORM, EPI><<<grid, 512, 0, st>>>(p);
    else if (p.K == 2048) wide_gemm_kernel<TS, 8, 8, NORM, EPI><<<grid, 512, 0, st>>>(p);
    else if (p.K == 3072) wide_gemm_kernel<TS, 12, 8, NORM, EPI><<<grid, 768, 0, st>>>(p);
    else if (p.K == 512) wide_gemm_kernel<TS, 4, 4, NORM, EPI><<<grid, 256, 0, st>>>(p);
    else if (p.K == 4096) wide_gemm_kernel<TS, 16, 8, NORM, EPI><<<grid, 1024, 0, st>>>(p);
    else return false;
    return true;
}
static inline bool wide_k_ok(int K) { return K == 512 || K == 1024 || K == 2048 || K == 3072 || K == 4096; }

}  // namespace ft
