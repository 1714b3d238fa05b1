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
#include "ar_kernels.h"   // xo_index

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

typedef float wk_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 wk_b2 __attribute__((ext_vector_type(2)));

// two packed bf16 -> two f32 lanes of a packed-f32 operand; and back with one v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ wk_f2 wk_unpack2(uint32_t w) { return wk_f2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ uint32_t wk_pack2(wk_f2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wk_b2)); }

// NT = 16-row weight tiles per workgroup (they share the workgroup's normalised operand registers: the RMSNorm arithmetic,
// ~5 vector instructions per element, is paid once per NT * 16 weight rows)
template <int TS, int NT, int NW>
constexpr int wide_lds_floats() { return NW * TS * 16 + NW * TS * 16 * (NT * 16 + 1); }

// bx, by: the workgroup's tile (16 * NT weight rows, 16 * TS batch rows); lds: wide_lds_floats<TS, NT, NW>() floats
// SYNC: hooks of an experiment that ran many dependent phases in one launch (tools/mb_chain.hip: weights requested, then a
// wait for the producer phase, then the activations); WideNoSync in the product
struct WideNoSync {
    static constexpr bool weights_first = false;
    __device__ __forceinline__ void wait() const {}
    __device__ __forceinline__ void signal() const {}
};
template <int TS, int NT, int NW, int KS, bool NORM, int EPI, typename SYNC = WideNoSync>
__device__ __forceinline__ void wide_gemm_body(const WideP& p, const int bx, const int by, float* lds, const SYNC& sync = SYNC()) {
    float (*ssw)[TS * 16] = reinterpret_cast<float (*)[TS * 16]>(lds);
    float (*Cs)[TS * 16][NT * 16 + 1] = reinterpret_cast<float (*)[TS * 16][NT * 16 + 1]>(lds + NW * TS * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = bx * (NT * 16), m0 = by * (TS * 16);
    const int kw = wave * (KS * 32);
    // ---- every load of the launch, issued back to back, in the order of use: the activations (L2) and the gain first, so
    // that the fused norm runs while the weights are still on their way from HBM (the counted waits follow issue order)
    U4 w[NT][KS], x[TS][KS], g[KS];
    auto issue_w = [&] {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16_t* wrow = p.W + (size_t)(n0 + t * 16 + fr) * p.ldw + kw + fq * 8;
#pragma unroll
            for (int s = 0; s < KS; ++s) w[t][s] = *reinterpret_cast<const U4*>(wrow + s * 32);
        }
    };
    if constexpr (SYNC::weights_first) { issue_w(); sync.wait(); }
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
    if constexpr (!SYNC::weights_first) issue_w();
    // the residual values this thread adds in the epilogue (element e of the tile -> thread e % threads), requested now
    constexpr int EPT = (TS * NT * 256 + NW * 64 - 1) / (NW * 64);
    bf16_t rs[EPT];
    if constexpr (EPI == WEPI_RESID) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * NW * 64;
            const int row = e / (NT * 16), c = e % (NT * 16);
            const int m = m0 + row, n = n0 + c;
            rs[i] = (e < TS * NT * 256 && m < p.M) ? p.resid_xo[((size_t)(n >> 3) * p.ldm_o + m) * 8 + (n & 7)] : (bf16_t)0;
        }
    }
    if constexpr (NORM) {
        // sum of squares of this wave's K slice, per row: lanes fr, fr + 16, fr + 32, fr + 48 hold the four octets of a step
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            wk_f2 ss2 = wk_f2{0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const uint32_t* xw = reinterpret_cast<const uint32_t*>(&x[j][s]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const wk_f2 v = wk_unpack2(xw[e]); ss2 = __builtin_elementwise_fma(v, v, ss2); }
            }
            float ss = ss2.x + ss2.y;
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
            const float inv = rsqrt_exact(ss / (float)p.K + p.eps);
            const wk_f2 inv2 = wk_f2{inv, inv};
            // x -> round(round(x / rms) * gain), the two roundings of llama.py:172-177, in place in the operand registers
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                uint32_t* xw = reinterpret_cast<uint32_t*>(&x[j][s]);
                const uint32_t* gw = reinterpret_cast<const uint32_t*>(&g[s]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xw[e] = wk_pack2(wk_unpack2(wk_pack2(wk_unpack2(xw[e]) * inv2)) * wk_unpack2(gw[e]));
            }
        }
    }
    wk_f32x4 acc[TS][NT];
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[j][t] = wk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            wk_bf16x8 a;
            __builtin_memcpy(&a, &x[j][s], 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wk_bf16x8 b;
                __builtin_memcpy(&b, &w[t][s], 16);
                acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j][t], 0, 0, 0);
            }
        }
    }
    // lane holds C[row = j*16 + 4*fq + r][n = t*16 + fr]
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[wave][j * 16 + fq * 4 + r][t * 16 + fr] = acc[j][t][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NW * 64;
        if (e >= TS * NT * 256) break;
        const int row = e / (NT * 16), c = e % (NT * 16);       // consecutive lanes finish one row's columns
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
                const int gc = n >> 1;
                p.out_xo[((size_t)(gc >> 3) * p.ldm_o + m) * 8 + (gc & 7)] = f32_to_bf16_bits(sg * other);
            }
        } else if constexpr (EPI == WEPI_RESID) {
            if (m < p.M) {
                const size_t oi = ((size_t)(n >> 3) * p.ldm_o + m) * 8 + (n & 7);
                v += bf16_bits_to_f32(rs[i]);
                p.out_xo[oi] = f32_to_bf16_bits(v);
            }
        } else {
            if (m < p.M) p.out_f32[(size_t)m * p.ldo + n] = v;
        }
    }
    sync.signal();
}

template <int TS, int NT, int NW, int KS, bool NORM, int EPI>
__global__ __launch_bounds__(NW * 64) void wide_gemm_kernel(WideP p) {
    __shared__ float lds[wide_lds_floats<TS, NT, NW>()];
    wide_gemm_body<TS, NT, NW, KS, NORM, EPI>(p, blockIdx.x, blockIdx.y, lds);
}

// ------------------------------------------------------------------------------------------
// The vocabulary head of a wide batch (llama.py:446-451: final RMSNorm + tied embedding, 155 776 rows at s1-mini): ONE
// normalisation of the <= 32 activation rows per workgroup into LDS (octet-major bf16, the A-operand layout), then every
// wave streams its own 16-row weight tiles over the whole contraction - B fragments straight from HBM, two halves of the
// K-steps in flight per wave (128 KB per CU), no cross-wave sum.  The general launch (16 NT rows per workgroup, K split
// over the waves) re-reads the activations once per 32 weight rows: 311 MB of L2 traffic beside 319 MB of weights, 103 us;
// this form reads them once per workgroup.
// grid: any (256 = one workgroup per CU), 512 threads; dynamic LDS: (K / 8) * WH_ROW * 16 bytes + 8 * 32 floats.  An octet row
// of the LDS image is padded from 32 to 36 pieces: the four octets a wave reads at once (lanes fq = 0..3) then start 576
// bytes apart and fall into four different bank groups (at 512 bytes they share one: a four-way conflict on every read).
// ------------------------------------------------------------------------------------------
constexpr int WH_ROW = 36;
template <int K>
__global__ __launch_bounds__(512) void wide_head_kernel(WideP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hl[];
    constexpr int KS = K / 32, HALF = KS / 2, OCT = K / 8;
    static_assert(KS % 2 == 0, "two halves of the K-steps per tile");
    U4* xn = reinterpret_cast<U4*>(hl);                       // [OCT][WH_ROW] 16-byte pieces (32 used)
    float* ssw = reinterpret_cast<float*>(hl + (size_t)OCT * WH_ROW * 16);   // [8][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // ---- the fused RMSNorm, once per workgroup: thread t owns row t % 32 of octets t / 32, t / 32 + 16, ..
    {
        const int row = tid & 31;
        const bool on = row < p.M;
        constexpr int PER = OCT / 16;
        U4 xr[PER];
        wk_f2 ss2 = wk_f2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int o = (tid >> 5) + 16 * i;
            xr[i] = on ? *reinterpret_cast<const U4*>(p.X + ((size_t)o * p.ldm + row) * 8) : U4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t* xw = reinterpret_cast<const uint32_t*>(&xr[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const wk_f2 v = wk_unpack2(xw[e]); ss2 = __builtin_elementwise_fma(v, v, ss2); }
        }
        float ss = ss2.x + ss2.y;
        ss += __shfl_xor(ss, 32);
        if (lane < 32) ssw[wave * 32 + lane] = ss;
        __syncthreads();
        float tot = ssw[row];
#pragma unroll
        for (int q = 1; q < 8; ++q) tot += ssw[q * 32 + row];
        const float inv = rsqrt_exact(tot / (float)K + p.eps);
        const wk_f2 inv2 = wk_f2{inv, inv};
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int o = (tid >> 5) + 16 * i;
            const U4 g = *reinterpret_cast<const U4*>(p.gain + o * 8);
            uint32_t* xw = reinterpret_cast<uint32_t*>(&xr[i]);
            const uint32_t* gw = reinterpret_cast<const uint32_t*>(&g);
#pragma unroll
            for (int e = 0; e < 4; ++e) xw[e] = wk_pack2(wk_unpack2(wk_pack2(wk_unpack2(xw[e]) * inv2)) * wk_unpack2(gw[e]));
            xn[o * WH_ROW + row] = xr[i];
        }
        __syncthreads();
    }
    // ---- weight tiles: tile t = weight rows [16 t, 16 t + 16); wave w of workgroup b takes tiles (i * gridDim.x + b) * 8 + w
    const int ntiles = p.N / 16;
    U4 wa[HALF], wb[HALF];
    auto issue = [&](U4 (&w)[HALF], int t, int half) {
        const bf16_t* wrow = p.W + (size_t)(min(t, ntiles - 1) * 16 + fr) * p.ldw + half * HALF * 32 + fq * 8;
#pragma unroll
        for (int s = 0; s < HALF; ++s) w[s] = *reinterpret_cast<const U4*>(wrow + s * 32);
    };
    auto mac = [&](const U4 (&w)[HALF], int half, wk_f32x4 (&acc)[2]) {
        // the activation fragments are re-read from LDS for every tile: an opaque zero in the address keeps the compiler from
        // hoisting all 64 of them out of the tile loop (256 registers: it spilled them to scratch, 154 us per launch)
        int z = 0;
        asm volatile("" : "+v"(z));
        const U4* xz = xn + z;
#pragma unroll
        for (int s = 0; s < HALF; ++s) {
            wk_bf16x8 b;
            __builtin_memcpy(&b, &w[s], 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wk_bf16x8 a;
                const U4 av = xz[((half * HALF + s) * 4 + fq) * WH_ROW + j * 16 + fr];
                __builtin_memcpy(&a, &av, 16);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
            }
        }
    };
    int t = (int)blockIdx.x * 8 + wave;
    const int tstep = (int)gridDim.x * 8;
    issue(wa, t, 0);
    issue(wb, t, 1);
    for (; t < ntiles; t += tstep) {
        wk_f32x4 acc[2] = {wk_f32x4{0.f, 0.f, 0.f, 0.f}, wk_f32x4{0.f, 0.f, 0.f, 0.f}};
        mac(wa, 0, acc);
        issue(wa, t + tstep, 0);            // (past the last tile: re-reads it, never used)
        mac(wb, 1, acc);
        issue(wb, t + tstep, 1);
        // lane holds C[m = j * 16 + 4 fq + r][n = 16 t + fr]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = j * 16 + fq * 4 + r;
                if (m < p.M) {
                    float v = acc[j][r];
                    if (p.bias) v += p.bias[t * 16 + fr];
                    p.out_f32[(size_t)m * p.ldo + t * 16 + fr] = round_bf16(v);
                }
            }
    }
}
template <int K>
static inline size_t wide_head_lds() { return (size_t)(K / 8) * WH_ROW * 16 + 8 * 32 * sizeof(float); }

// K split: 128..256 contraction elements per wave where the width allows (one memory round trip, registers for every load)
template <int TS, int NT, bool NORM, int EPI>
static inline bool wide_gemm_launch(const WideP& p, hipStream_t st) {
    if (p.N % (NT * 16) != 0 || p.M < 1) return false;
    const dim3 grid(p.N / (NT * 16), (p.M + TS * 16 - 1) / (TS * 16));
    if (p.K == 1024) wide_gemm_kernel<TS, NT, 8, 4, NORM, EPI><<<grid, 512, 0, st>>>(p);
    else if (p.K == 2048) wide_gemm_kernel<TS, NT, 8, 8, NORM, EPI><<<grid, 512, 0, st>>>(p);
    else if (p.K == 3072) wide_gemm_kernel<TS, NT, 12, 8, NORM, EPI><<<grid, 768, 0, st>>>(p);
    else return false;
    return true;
}
static inline bool wide_k_ok(int K) { return K == 1024 || K == 2048 || K == 3072; }

}  // namespace ft
