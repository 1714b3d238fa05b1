// DAC codec decode kernels for gfx950.  Activations are time-major [T][C] bf16 in HBM (the
// transformer's residual stream is f32); every convolution / linear layer is one "tap GEMM":
//     Y[t][n] = sum_tap sum_c X[t + off_tap][c] * W[tap][n][c]          (rows t + off < 0 read as 0)
// on bf16 MFMA (v_mfma_f32_16x16x32_bf16, f32 accumulate) with fused epilogues:
//   * Conv1d k=7 dilation d (vocoder.py:394-421): 7 taps, off = (kk-6)*d
//   * ConvTranspose1d k=2s stride s, right-trimmed (vocoder.py:432-455): N = s*Cout, 2 taps (0, -1),
//     the [T][s][Cout] result IS the [T*s][Cout] output
//   * Linear / 1x1 conv: one tap
// Snake (x + sin^2(ax)/a) is applied once per element in the producer's epilogue, never per tap.
#pragma once
#include <algorithm>

#include "common.h"

namespace ft {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SWIGLU = 2, ACT_TANH = 3 };

struct TapGemmP {
    const bf16_t* X;      // [B][T_in][ldx]
    long ldx, x_bstride;
    int T_in;
    int t_min;            // lowest readable row of X (0; negative in a streamed decode: rows [t_min, 0) hold the previous
                          // chunk's last rows instead of the causal zero padding)
    const bf16_t* W;      // [ntap][N][K]
    int ntap;
    int offs[8];
    int M, N, K;          // output rows per batch item, output columns, contraction per tap
    const float* bias;    // [n_mod] or null
    int n_mod;            // bias/alpha/gamma index = n % n_mod
    int act;
    const float* gamma;   // per-column scale (LayerScale / ConvNeXt gamma) or null
    const float* resid_f32;
    const bf16_t* resid_bf;
    long ldr, r_bstride;
    float* out_f32;
    bf16_t* out_bf;
    bf16_t* out_act;      // snake(v, alpha) or null
    const float* alpha;
    long ldo, o_bstride;
    int round_lin;      // round acc+bias to bf16 first (an nn.Linear output of a bf16 model; also the SwiGLU steps)
    int round_f32_out;  // out_f32 receives bf16-rounded values
    long ldw;           // skinny kernel: row stride of W in elements (0 = K, dense)
    // skinny kernel, fused RMSNorm (llama.py:172-177) of the X operand: X holds the un-normalised rows (exact bf16
    // copies of the residual stream), ss_in[row][b] the partial sums of squares the producer GEMM's blocks left
    // (ss_nblk partials per row, row stride ss_ld); ss_out: this GEMM's own partials of its f32 output rows
    const bf16_t* gain;
    const float* ss_in;
    float* ss_out;
    int ss_nblk, ss_ld;
    float eps;
    // (measured and removed: the RMSNorm of the OUTPUT rows by the block that finishes last - write-through stores, a
    // returning ticket atomic and the last block's trip to the memory side cost more than the ~5 us norm launch they
    // replaced: 4.67 against 3.54 ms per 32-row frame)
};

// offs[] lives in the kernel arguments: a runtime index would force the whole struct into scratch
__device__ __forceinline__ int tap_off(const TapGemmP& p, int tap) {
    int o = p.offs[0];
#pragma unroll
    for (int t = 1; t < 8; ++t) o = (t == tap) ? p.offs[t] : o;
    return o;
}

__device__ __forceinline__ float snake_f(float v, float a) {
    // hardware sine (v_sin_f32 after the 1/2pi scaling): ~1e-6 absolute error, far below the bf16 step of the
    // value it is stored in; the argument is O(10) here
    const float s = __sinf(a * v);
    return v + (1.0f / (a + 1e-9f)) * (s * s);
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <int TM, int TN>
__device__ __forceinline__ void tapgemm_epilogue(const TapGemmP& p, f32x4 (&acc)[TM][TN], const int mw, const int nw,
                                                 const int b, const int fr, const int fq) {
    // epilogue: lane holds C[row = 4*fq + r][col = fr] of each 16x16 tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nw + j * 16 + fr;
            const bool nv = n < p.N;
            const int nm = nv ? n % p.n_mod : 0;
            const float bias = (p.bias && nv) ? p.bias[nm] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = mw + i * 16 + fq * 4 + r;
                float v = acc[i][j][r] + bias;
                if (p.round_lin) v = round_bf16(v);
                if (p.act == ACT_SWIGLU) {
                    // columns (2i, 2i+1) = (gate, up): partner value sits in the neighbouring lane
                    const float other = dpp_f<DPP_XOR1>(v);
                    if ((fr & 1) == 0 && nv && t < p.M) {
                        const float gate = v, up = other;
                        float sg = gate / (1.0f + expf(-gate));
                        if (p.round_lin) sg = round_bf16(sg);
                        const float o = sg * up;
                        const size_t oi = (size_t)b * p.o_bstride + (size_t)t * p.ldo + (n >> 1);
                        if (p.out_bf) p.out_bf[oi] = f32_to_bf16_bits(o);
                        if (p.out_f32) p.out_f32[oi] = p.round_f32_out ? round_bf16(o) : o;
                    }
                    continue;
                }
                if (!nv || t >= p.M) continue;
                if (p.act == ACT_GELU) v = gelu_f(v);
                else if (p.act == ACT_TANH) v = tanhf(v);
                if (p.gamma) v *= p.gamma[nm];
                const size_t ri = (size_t)b * p.r_bstride + (size_t)t * p.ldr + n;
                if (p.resid_f32) v += p.resid_f32[ri];
                if (p.resid_bf) v += bf16_bits_to_f32(p.resid_bf[ri]);
                const size_t oi = (size_t)b * p.o_bstride + (size_t)t * p.ldo + n;
                if (p.out_f32) p.out_f32[oi] = p.round_f32_out ? round_bf16(v) : v;
                if (p.out_bf) p.out_bf[oi] = f32_to_bf16_bits(v);
                if (p.out_act) p.out_act[oi] = f32_to_bf16_bits(snake_f(v, p.alpha[nm]));
            }
        }
    }
}

// Block tile BM x BN, BK = 32, 256 threads = 4 waves arranged WGM x WGN; each wave owns
// (BM/WGM) x (BN/WGN) as 16x16 MFMA tiles.  LDS rows are padded to 40 bf16 (80 B) so the 16-byte
// fragment reads of 16 consecutive rows spread over the banks.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void tapgemm_kernel(TapGemmP p) {
    constexpr int BK = 32, LDS_LD = 40;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 16, TN = WN / 16;
    __shared__ __attribute__((aligned(16))) bf16_t As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, b = blockIdx.z;
    const bf16_t* X = p.X + (size_t)b * p.x_bstride;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    for (int tap = 0; tap < p.ntap; ++tap) {
        const int off = tap_off(p, tap);
        const bf16_t* Wt = p.W + (size_t)tap * p.N * p.K;
        for (int k0 = 0; k0 < p.K; k0 += BK) {
            // stage A (BM x 32) and B (BN x 32): 16-byte chunks, 4 per row
            for (int c = tid; c < BM * 4; c += 256) {
                const int r = c >> 2, q = c & 3;
                const int t = m0 + r + off;
                U4 v = U4{0u, 0u, 0u, 0u};
                if (m0 + r < p.M && t >= p.t_min && t < p.T_in)
                    v = *reinterpret_cast<const U4*>(X + (size_t)t * p.ldx + k0 + q * 8);
                *reinterpret_cast<U4*>(&As[r * LDS_LD + q * 8]) = v;
            }
            for (int c = tid; c < BN * 4; c += 256) {
                const int r = c >> 2, q = c & 3;
                U4 v = U4{0u, 0u, 0u, 0u};
                if (n0 + r < p.N) v = *reinterpret_cast<const U4*>(Wt + (size_t)(n0 + r) * p.K + k0 + q * 8);
                *reinterpret_cast<U4*>(&Bs[r * LDS_LD + q * 8]) = v;
            }
            __syncthreads();
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * WM + i * 16 + fr) * LDS_LD + fq * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * WN + j * 16 + fr) * LDS_LD + fq * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __syncthreads();
        }
    }
    tapgemm_epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, b, fr, fq);
}

// Epilogue through LDS: the accumulators (column-per-lane) are transposed so that each lane finishes 8 consecutive
// columns of one row: 16-byte loads of the residual, 16-byte stores of every output (the late decoder stages are
// bandwidth-bound: 2-byte scattered stores were the bottleneck).  NWM passes of BM/NWM rows.
template <int BM, int BN, int TM, int TN, int NWM = 2, int NWN = 2>
__device__ __forceinline__ void tapgemm_epilogue_lds(const TapGemmP& p, f32x4 (&acc)[TM][TN], float* Cs, const int m0,
                                                     const int n0, const int b, const int wm, const int wn,
                                                     const int fr, const int fq) {
    constexpr int LDC = BN + 4;
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int NTHR = 64 * NWM * NWN;
    const int tid = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < NWM; ++pass) {
        __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(i * 16 + fq * 4 + r) * LDC + wn * WN + j * 16 + fr] = acc[i][j][r];
        }
        __syncthreads();
        for (int v = tid; v < WM * (BN / 8); v += NTHR) {
            const int row = v / (BN / 8), c8 = (v % (BN / 8)) * 8;
            const int t = m0 + pass * WM + row, n = n0 + c8;
            if (t >= p.M || n >= p.N) continue;
            const int nm = n % p.n_mod;
            float x[8];
            const float4 c0 = *reinterpret_cast<const float4*>(&Cs[row * LDC + c8]);
            const float4 c1 = *reinterpret_cast<const float4*>(&Cs[row * LDC + c8 + 4]);
            x[0] = c0.x; x[1] = c0.y; x[2] = c0.z; x[3] = c0.w; x[4] = c1.x; x[5] = c1.y; x[6] = c1.z; x[7] = c1.w;
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += p.bias[nm + j];
            }
            if (p.round_lin) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = round_bf16(x[j]);
            }
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = gelu_f(x[j]);
            } else if (p.act == ACT_TANH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = tanhf(x[j]);
            }
            if (p.gamma) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] *= p.gamma[nm + j];
            }
            const size_t ri = (size_t)b * p.r_bstride + (size_t)t * p.ldr + n;
            if (p.resid_f32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += p.resid_f32[ri + j];
            }
            if (p.resid_bf) {
                float rv[8];
                Vec<bf16_t>::load(p.resid_bf + ri, rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += rv[j];
            }
            const size_t oi = (size_t)b * p.o_bstride + (size_t)t * p.ldo + n;
            if (p.out_f32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) p.out_f32[oi + j] = p.round_f32_out ? round_bf16(x[j]) : x[j];
            }
            if (p.out_bf) {
                U4 o;
                o.x = f32_to_bf16_bits(x[0]) | ((uint32_t)f32_to_bf16_bits(x[1]) << 16);
                o.y = f32_to_bf16_bits(x[2]) | ((uint32_t)f32_to_bf16_bits(x[3]) << 16);
                o.z = f32_to_bf16_bits(x[4]) | ((uint32_t)f32_to_bf16_bits(x[5]) << 16);
                o.w = f32_to_bf16_bits(x[6]) | ((uint32_t)f32_to_bf16_bits(x[7]) << 16);
                *reinterpret_cast<U4*>(p.out_bf + oi) = o;
            }
            if (p.out_act) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = snake_f(x[j], p.alpha[nm + j]);
                U4 o;
                o.x = f32_to_bf16_bits(y[0]) | ((uint32_t)f32_to_bf16_bits(y[1]) << 16);
                o.y = f32_to_bf16_bits(y[2]) | ((uint32_t)f32_to_bf16_bits(y[3]) << 16);
                o.z = f32_to_bf16_bits(y[4]) | ((uint32_t)f32_to_bf16_bits(y[5]) << 16);
                o.w = f32_to_bf16_bits(y[6]) | ((uint32_t)f32_to_bf16_bits(y[7]) << 16);
                *reinterpret_cast<U4*>(p.out_act + oi) = o;
            }
        }
    }
}

// Pipelined variant for K % 64 == 0 (every layer of the real codec).  Per 64-channel chunk the A rows of the
// block *and its tap halo* (BM + max|off| rows) are staged once and shared by all taps; the B (weight) tile of
// the next tap is fetched into registers while the current one feeds the MFMAs and is written to the other LDS
// buffer afterwards: one barrier per (tap, chunk) step, 32 (BN=128) MFMAs per wave between barriers.
// NWM x NWN waves (default 2 x 2 = 256 threads).  The 8-wave 128-row tiles exist because the small tile is bound by the L2:
// a BM x BN tile reads (BM + BN) K bf16 per BM BN K MACs = BM BN / (BM + BN) flop per byte - 32 at 64 x 64, and 1024
// co-resident blocks re-reading their weight tiles at ~10-13 TB/s of L2 bandwidth is the 320-420 TFLOP/s the k = 7
// convolutions of the decoder ran at (12.8 % MFMA); 128 x 128 doubles it with the same waves per CU.
template <int BM, int BN, int BK, int NWM = 2, int NWN = 2>
__global__ __launch_bounds__(64 * NWM * NWN) void tapgemm64_kernel(TapGemmP p) {
    constexpr int NTHR = 64 * NWM * NWN;
    constexpr int LD = BK + 8;                      // LDS row stride (bf16): keeps 16-byte alignment, spreads banks
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int TM = WM / 16, TN = WN / 16;
    static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile = whole 16 x 16 MFMA tiles");
    constexpr int MAXH = 56;                        // largest tap halo (k=7, dilation 9 -> 54)
    constexpr int QPR = BK / 8;                     // 16-byte chunks per tile row
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    bf16_t* As = lds;                               // [(BM + MAXH)][LD]
    bf16_t* Bs0 = As + (BM + MAXH) * LD;            // [BN][LD] x 2
    bf16_t* Bs1 = Bs0 + BN * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, b = blockIdx.z;
    const bf16_t* X = p.X + (size_t)b * p.x_bstride;
    int offmin = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) offmin = (t < p.ntap) ? min(offmin, p.offs[t]) : offmin;
    const int srows = BM - offmin;                  // stripe rows: t in [m0 + offmin, m0 + BM)
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int BCH = (BN * QPR + NTHR - 1) / NTHR;     // 16-byte chunks of a B tile per thread
    // weight tiles are fetched TWO steps ahead into two named register sets (the MFMA work of one step is far
    // shorter than an L2 round trip)
    U4 bregA[BCH], bregB[BCH];
    auto load_b = [&](U4 (&breg)[BCH], int step) {
        const int tap = step % p.ntap, k0 = (step / p.ntap) * BK;
        const bf16_t* Wt = p.W + (size_t)tap * p.N * p.K;
#pragma unroll
        for (int u = 0; u < BCH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            breg[u] = (r < BN && n0 + r < p.N) ? *reinterpret_cast<const U4*>(Wt + (size_t)(n0 + r) * p.K + k0 + q * 8) : U4{0u, 0u, 0u, 0u};
        }
    };
    auto store_b = [&](const U4 (&breg)[BCH], bf16_t* Bs) {
#pragma unroll
        for (int u = 0; u < BCH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            if (r < BN) *reinterpret_cast<U4*>(&Bs[r * LD + q * 8]) = breg[u];
        }
    };
    const int nsteps = p.ntap * (p.K / BK);
    const int nchunks = p.K / BK;
    // the A stripe of the NEXT K chunk also travels in registers while this chunk's taps run (with one tap - linears,
    // 1x1 convs, the prompt GEMMs - its load latency would otherwise be exposed once per step)
    constexpr int ACH = ((BM + MAXH) * QPR + NTHR - 1) / NTHR;
    U4 areg[ACH];
    auto load_a = [&](int kc) {
#pragma unroll
        for (int u = 0; u < ACH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            const int t = m0 + offmin + r;
            areg[u] = (c < srows * QPR && t >= p.t_min && t < p.T_in)
                          ? *reinterpret_cast<const U4*>(X + (size_t)t * p.ldx + kc * BK + q * 8) : U4{0u, 0u, 0u, 0u};
        }
    };
    load_a(0);
    auto do_step = [&](int step, U4 (&breg)[BCH], bf16_t* Bcur) {
        const int kc = step / p.ntap, tap = step % p.ntap;
        if (tap == 0) {
            // previous chunk's MFMAs are done (barrier at the end of its last step): restage the A stripe
#pragma unroll
            for (int u = 0; u < ACH; ++u) {
                const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
                if (c < srows * QPR) *reinterpret_cast<U4*>(&As[r * LD + q * 8]) = areg[u];
            }
            if (kc + 1 < nchunks) load_a(kc + 1);
        }
        store_b(breg, Bcur);
        __syncthreads();
        if (step + 2 < nsteps) load_b(breg, step + 2);  // in flight during two steps of MFMAs
        const int arow = tap_off(p, tap) - offmin;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[(arow + wm * WM + i * 16 + fr) * LD + kk * 32 + fq * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(&Bcur[(wn * WN + j * 16 + fr) * LD + kk * 32 + fq * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        // the next step writes the OTHER B buffer; the A stripe is rewritten only at tap == 0 of the next chunk,
        // which must wait for every wave's reads of this chunk: barrier only then
        if (tap == p.ntap - 1) __syncthreads();
    };
    load_b(bregA, 0);
    if (nsteps > 1) load_b(bregB, 1);
    for (int step = 0; step < nsteps; step += 2) {
        do_step(step, bregA, Bs0);
        if (step + 1 < nsteps) do_step(step + 1, bregB, Bs1);
    }
    // vector epilogue whenever rows are 16-byte addressable (every real layer); SwiGLU pairs keep the lane epilogue
    const bool vec_ok = p.act != ACT_SWIGLU && (p.N % 8) == 0 && (p.n_mod % 8) == 0 && (p.ldo % 8) == 0 && (p.ldr % 8) == 0;
    if constexpr (TM * TN >= 16) {
        // 128x128: instantiating the lane epilogue here makes the compiler keep all 16 accumulator tiles in scratch for
        // the whole kernel (272 B/lane); the host selects this tile only for layers the vector epilogue covers
        tapgemm_epilogue_lds<BM, BN, TM, TN, NWM, NWN>(p, acc, reinterpret_cast<float*>(lds), m0, n0, b, wm, wn, fr, fq);
    } else {
        if (vec_ok) tapgemm_epilogue_lds<BM, BN, TM, TN, NWM, NWN>(p, acc, reinterpret_cast<float*>(lds), m0, n0, b, wm, wn, fr, fq);
        else tapgemm_epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, b, fr, fq);
    }
}

// Plain Linear (one tap, no halo) on the same tiles, for the prompt pass: A and B tiles of DEPTH K-steps travel in registers
// at any time (the pipelined kernel above has its A stripe one step ahead only - enough behind seven taps, not for a
// linear: at 780 prompt rows every step waited ~2 us for its rows, 38 us per product).  Two LDS buffers, one barrier per
// step: step k's tiles are written to buffer k % 2 while buffer (k + 1) % 2 may still be read by waves one barrier behind.
template <int BM, int BN, int NWM, int NWN, int DEPTH, int MINW = 1>
__global__ __launch_bounds__(64 * NWM * NWN, MINW) void lingemm_kernel(TapGemmP p) {
    constexpr int BK = 64, NTHR = 64 * NWM * NWN, LD = BK + 8, QPR = BK / 8;
    constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 16, TN = WN / 16;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && DEPTH % 2 == 0, "whole MFMA tiles per wave; buffer parity follows the unrolled step");
    static_assert((BM * QPR) % NTHR == 0 && (BN * QPR) % NTHR == 0, "every thread moves whole 16-byte pieces of both tiles");
    constexpr int ACH = (BM * QPR) / NTHR, BCH = (BN * QPR) / NTHR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    bf16_t* const A0 = lds;
    bf16_t* const B0 = lds + 2 * BM * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    U4 ar[DEPTH][ACH], br[DEPTH][BCH];
    auto load = [&](U4 (&a)[ACH], U4 (&b)[BCH], int step) {
        const int k0 = step * BK;
#pragma unroll
        for (int u = 0; u < ACH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            // rows past the end re-read the last row (their products are never stored): every load is unconditional, so the
            // counted waits of the pipeline stay exact (a predicated load costs a branch and a full drain per step)
            a[u] = *reinterpret_cast<const U4*>(p.X + (size_t)min(m0 + r, p.T_in - 1) * p.ldx + k0 + q * 8);
        }
#pragma unroll
        for (int u = 0; u < BCH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            b[u] = *reinterpret_cast<const U4*>(p.W + (size_t)min(n0 + r, p.N - 1) * p.K + k0 + q * 8);
        }
    };
    auto stage = [&](const U4 (&a)[ACH], const U4 (&b)[BCH], bf16_t* As, bf16_t* Bs) {
#pragma unroll
        for (int u = 0; u < ACH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            *reinterpret_cast<U4*>(&As[r * LD + q * 8]) = a[u];
        }
#pragma unroll
        for (int u = 0; u < BCH; ++u) {
            const int c = tid + NTHR * u, r = c / QPR, q = c % QPR;
            *reinterpret_cast<U4*>(&Bs[r * LD + q * 8]) = b[u];
        }
    };
    // K / 64 is a multiple of DEPTH (the host checks): no branch sits between a load and its use, and the loads past the end
    // re-read the last step's tiles, so the compiler's counted waits (vmcnt) never have to drain the pipeline
    const int nsteps = p.K / BK;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(ar[d], br[d], d);
    for (int base = 0; base < nsteps; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int step = base + d;
            bf16_t* As = A0 + (d & 1) * BM * LD;
            bf16_t* Bs = B0 + (d & 1) * BN * LD;
            stage(ar[d], br[d], As, Bs);
            __syncthreads();
            load(ar[d], br[d], min(step + DEPTH, nsteps - 1));
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * WM + i * 16 + fr) * LD + kk * 32 + fq * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * WN + j * 16 + fr) * LD + kk * 32 + fq * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    const bool vec_ok = p.act != ACT_SWIGLU && (p.N % 8) == 0 && (p.n_mod % 8) == 0 && (p.ldo % 8) == 0 && (p.ldr % 8) == 0;
    if (vec_ok) tapgemm_epilogue_lds<BM, BN, TM, TN, NWM, NWN>(p, acc, reinterpret_cast<float*>(lds), m0, n0, 0, wm, wn, fr, fq);
    else { __syncthreads(); tapgemm_epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, 0, fr, fq); }
}
template <int BM, int BN, int NWM>
constexpr size_t lingemm_lds_bytes() {
    return std::max((size_t)2 * (BM + BN) * (64 + 8) * 2, (size_t)(BM / NWM) * (BN + 4) * 4);
}

// ---- prompt-pass attention on the matrix cores (llama.py:229-283 with the causal mask of 437) -----------------------
// One block = 64 query positions of one query head (4 waves x 16 rows); K and V^T tiles of 32 cached positions are staged
// in LDS once per block and shared by the waves.  S = Q K^T and O += P V run on v_mfma_f32_16x16x32_bf16; the softmax is
// the online form in f32; P enters the second product as two bf16 planes (hi = bf16(p), lo = bf16(p - hi)), i.e. with
// f32-like precision, so the result follows the f32 SDPA of the reference up to summation order (one rounding at the end).
struct FlashP {
    const bf16_t* q;    // [S][H*hd] normalised + rotated queries
    const bf16_t* kc;   // [Hkv][n_slots][hd]
    const bf16_t* vc;
    bf16_t* y_bf;       // [S][H*hd]
    int S, H, Hkv, hd, n_slots, pos0;
    float scale;
    // ragged pass over the prompts of several slots (grid.z = sequence): seqs[z] = {first row, rows, first cache position,
    // slot}; q / y_bf rows are the concatenated prompts, kc / vc the layer's base pointers, cache_m_stride the elements
    // between slots.  Blocks past a sequence's last query tile leave at once.
    const int4* seqs;
    size_t cache_m_stride;
};
template <int HD, int NG>
__global__ __launch_bounds__(512) void flash_prefill_kernel(FlashP p) {
    constexpr int KT = 32;                  // keys per tile and key group
    constexpr int QW = 8 / NG;              // NG key groups of QW waves: group g takes keys [kt + 32 g, kt + 32 g + 32) of a step,
    constexpr int QROWS = 16 * QW;          // its waves 16 query rows each - a block covers QROWS query rows of one head
    static_assert(NG == 2 || NG == 4, "eight waves: 2 key groups x 64 query rows or 4 x 32");
    constexpr int LDK = HD + 8;             // K tile row stride (bf16)
    constexpr int LDV = KT + 8;             // V^T tile row stride
    constexpr int LDP = KT + 8;
    constexpr int NS = HD / 32;             // k-steps of Q K^T
    constexpr int NT = HD / 16;             // 16-wide output column tiles
    // The key groups walk the SAME query rows with their own online-softmax state and are merged once at the end: the
    // dependent chain of a block (a 780-position prompt: 26 tiles for the last query tile, the kernel's critical path with
    // fewer blocks than CUs) shrinks to 26 / NG steps of NG tiles side by side.
    constexpr int KS_N = KT * LDK, VT_N = HD * LDV, PS_N = 16 * LDP, XW = NT * 4 + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_f[];      // flash_prefill_lds<HD, NG>() bytes
    static_assert((size_t)(NG * VT_N + 8 * 2 * PS_N) * sizeof(bf16_t) >= (size_t)(NG - 1) * QW * 64 * XW * sizeof(float), "the merge buffer reuses Vt + Ps");
    bf16_t (*Ks)[KS_N] = reinterpret_cast<bf16_t (*)[KS_N]>(smem_f);
    bf16_t (*Vt)[VT_N] = reinterpret_cast<bf16_t (*)[VT_N]>(smem_f + NG * KS_N);
    bf16_t (*Ps)[2][PS_N] = reinterpret_cast<bf16_t (*)[2][PS_N]>(smem_f + NG * KS_N + NG * VT_N);   // [8 waves]
    const int h = blockIdx.x, q0 = blockIdx.y * QROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave / QW, qw = wave % QW;            // key group, query sub-tile
    const int fr = lane & 15, fq = lane >> 4;
    const int G = p.H / p.Hkv, kvh = h / G;
    const bf16_t* kc = p.kc + (size_t)kvh * p.n_slots * HD;
    const bf16_t* vc = p.vc + (size_t)kvh * p.n_slots * HD;
    const bf16_t* qbase = p.q;
    bf16_t* ybase = p.y_bf;
    int S = p.S, pos0 = p.pos0;
    if (p.seqs) {
        const int4 sq = p.seqs[blockIdx.z];
        S = sq.y; pos0 = sq.z;
        if (q0 >= S) return;                               // the whole block: no barrier has been reached yet
        qbase += (size_t)sq.x * p.H * HD; ybase += (size_t)sq.x * p.H * HD;
        kc += (size_t)sq.w * p.cache_m_stride; vc += (size_t)sq.w * p.cache_m_stride;
    }
    // Q fragments of this wave's 16 rows (A operand: row = lane & 15, 8 consecutive d per lane)
    const int qrow = min(q0 + qw * 16 + fr, S - 1);
    bf16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        qf[s] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * p.H * HD + (size_t)h * HD + s * 32 + fq * 8);
    f32x4 O[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) O[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
    const int last_row = min(q0 + QROWS - 1, S - 1);
    const int kmax = pos0 + last_row;                    // last visible key of the block
    // K/V rows of a step (NG * KT keys) travel in registers TWO steps ahead of their use, in two named register sets; every
    // load is unconditional (rows past the last visible key re-read that key's row: they are masked below), so no branch sits
    // between a load and its use and the counted waits never drain the pipeline.  Each thread owns CPT 16-byte pieces.
    constexpr int CPT = NG * KT * (HD / 8) / 512;
    U4 kregA[CPT], vregA[CPT], kregB[CPT], vregB[CPT];
    auto fetch = [&](U4 (&kreg)[CPT], U4 (&vreg)[CPT], int kt) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 512;
            const int kr = c / (HD / 8), d8 = (c % (HD / 8)) * 8;      // kr in [0, NG * KT)
            const int j = min(kt + kr, kmax);
            kreg[i] = *reinterpret_cast<const U4*>(kc + (size_t)j * HD + d8);
            vreg[i] = *reinterpret_cast<const U4*>(vc + (size_t)j * HD + d8);
        }
    };
    fetch(kregA, vregA, 0);
    fetch(kregB, vregB, NG * KT);
    // V^T in LDS: the 8-key group g of row d sits at group g ^ ((d / 8) & 3) - the 16 lanes that write one key of 16 different
    // 8-row bands then spread over four banks instead of one (their rows are 8 x LDV halfs = a multiple of 32 dwords apart)
    auto tile = [&](U4 (&kreg)[CPT], U4 (&vreg)[CPT], const int kt0) {
        __syncthreads();                                    // previous step fully consumed
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 512;
            const int kr2 = c / (HD / 8), d8 = (c % (HD / 8)) * 8;
            const int g2 = kr2 / KT, kr = kr2 % KT;
            *reinterpret_cast<U4*>(&Ks[g2][kr * LDK + d8]) = kreg[i];
            const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vreg[i]);
            const int col = (((kr >> 3) ^ ((d8 >> 3) & 3)) << 3) + (kr & 7);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[g2][(d8 + e) * LDV + col] = ve[e];
        }
        __syncthreads();
        fetch(kreg, vreg, kt0 + 2 * NG * KT);
        const int kt = kt0 + grp * KT;                      // this group's keys
        // S = Q K^T for two 16-key sub-tiles: lane holds S[q = 4 fq + r][key = sub * 16 + fr]
        f32x4 sc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            sc[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[grp][(sub * 16 + fr) * LDK + s * 32 + fq * 8]);
                sc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], kf, sc[sub], 0, 0, 0);
            }
        }
        float pr[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qabs = pos0 + q0 + qw * 16 + fq * 4 + r;
            float s0 = sc[0][r] * p.scale, s1 = sc[1][r] * p.scale;
            if (kt + fr > qabs) s0 = -INFINITY;             // causal mask (also hides the rows re-read past the last key)
            if (kt + 16 + fr > qabs) s1 = -INFINITY;
            float mx = fmaxf(s0, s1);
            mx = fmaxf(mx, dpp_f<DPP_XOR1>(mx));
            mx = fmaxf(mx, dpp_f<DPP_XOR2>(mx));
            mx = fmaxf(mx, dpp_f<DPP_HALF_MIRROR>(mx));
            mx = fmaxf(mx, dpp_f<DPP_MIRROR>(mx));          // max over the row's 32 keys
            const float mn = fmaxf(mrun[r], mx);
            float corr = 1.f, p0 = 0.f, p1 = 0.f;
            if (mn > -INFINITY) {                           // rows past the prompt end may see nothing yet
                corr = expf(mrun[r] - mn);
                p0 = expf(s0 - mn);
                p1 = expf(s1 - mn);
            }
            mrun[r] = mn;
            lrun[r] = lrun[r] * corr + (p0 + p1);           // per-lane share of the row sum, reduced at the end
#pragma unroll
            for (int t = 0; t < NT; ++t) O[t][r] *= corr;
            pr[0][r] = p0; pr[1][r] = p1;
        }
        // P to the A-operand layout through this wave's LDS patch, split into two bf16 planes
        bf16_t* Ph = Ps[wave][0];
        bf16_t* Pl = Ps[wave][1];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = pr[sub][r];
                const bf16_t hi = f32_to_bf16_bits(v);
                const bf16_t lo = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
                Ph[(fq * 4 + r) * LDP + sub * 16 + fr] = hi;
                Pl[(fq * 4 + r) * LDP + sub * 16 + fr] = lo;
            }
        __syncthreads();
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ph[fr * LDP + fq * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Pl[fr * LDP + fq * 8]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int d = t * 16 + fr;
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[grp][d * LDV + ((fq ^ ((d >> 3) & 3)) << 3)]);
            O[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, vf, O[t], 0, 0, 0);
            O[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, vf, O[t], 0, 0, 0);
        }
    };
    // steps in pairs (one per register set); a tile wholly past the last visible key adds nothing (every score masked)
    for (int kt = 0; kt <= kmax; kt += 2 * NG * KT) {
        tile(kregA, vregA, kt);
        tile(kregB, vregB, kt + NG * KT);
    }
    // ---- merge of the two key groups: group 1 leaves (m, l, O) of its rows in LDS (the tile buffers are free now), group 0
    // rescales both to the common maximum and writes y
    __syncthreads();
    float* xbase = reinterpret_cast<float*>(smem_f + NG * KS_N);          // [NG - 1][QW][64 lanes][XW]
    if (grp > 0) {
        float* xch = xbase + ((size_t)((grp - 1) * QW + qw) * 64 + lane) * XW;
#pragma unroll
        for (int r = 0; r < 4; ++r) { xch[r] = mrun[r]; xch[4 + r] = lrun[r]; }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[8 + t * 4 + r] = O[t][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mn = mrun[r];
#pragma unroll
            for (int g = 1; g < NG; ++g) mn = fmaxf(mn, xbase[((size_t)((g - 1) * QW + qw) * 64 + lane) * XW + r]);
            const float c0 = mrun[r] > -INFINITY ? expf(mrun[r] - mn) : 0.f;
            float lsum = lrun[r] * c0;
            float cg[NG];
            cg[0] = c0;
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                const float* xg = xbase + ((size_t)((g - 1) * QW + qw) * 64 + lane) * XW;
                cg[g] = xg[r] > -INFINITY ? expf(xg[r] - mn) : 0.f;
                lsum += xg[4 + r] * cg[g];
            }
            const float l = row16_sum(lsum);
            const int row = q0 + qw * 16 + fq * 4 + r;
            if (row < S) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float o = O[t][r] * c0;
#pragma unroll
                    for (int g = 1; g < NG; ++g) o += xbase[((size_t)((g - 1) * QW + qw) * 64 + lane) * XW + 8 + t * 4 + r] * cg[g];
                    ybase[(size_t)row * p.H * HD + (size_t)h * HD + t * 16 + fr] = f32_to_bf16_bits(o / l);
                }
            }
        }
    }
}
template <int HD, int NG>
constexpr size_t flash_prefill_lds() { return (size_t)(NG * 32 * (HD + 8) + NG * HD * 40 + 8 * 2 * 16 * 40) * sizeof(bf16_t); }

// ---- skinny GEMM: few rows (prompt positions / lock-step utterances), weights streamed once ----------------------
// out[t][n] = sum_k X[t][k] W[n][k] for M <= 16*TS rows per block column.  The problem is weight-bandwidth bound,
// so the grid is cut along N only (16 weight rows per block => N/16 blocks) and the four waves of a block split K;
// fragments go straight from global memory into the MFMA operand registers (each lane's 16 bytes are one operand),
// the four partial tiles meet in LDS and are summed in wave order (deterministic).  Epilogue = the nn.Linear
// rounding points of the prefill (bias, rounded; residual add, rounded; SwiGLU on interleaved (gate, up) rows).
// Requires K % 128 == 0; act in {ACT_NONE, ACT_GELU, ACT_SWIGLU}; ntap == 1; one batch item (b = 0).
// XLDS: the X rows reach the MFMA operand registers through a per-wave LDS patch - loaded in full 256-byte row pieces
// (4 rows per load instruction) and re-read fragment-shaped with ds_read_b128 - instead of 16 rows x 64 bytes per load
// instruction straight from L2 (the per-CU address path was the limit of that form: tools/mb_skinny.hip).
constexpr int SKINNY_LDX = 136;   // LDS row stride of an X chunk (128 k + 8 pad, bf16): fragment reads spread over the banks
template <int TS, int NW>
constexpr size_t skinny_lds_bytes(bool xlds) {
    const size_t cs = (size_t)NW * TS * 16 * 17 * sizeof(float);
    const size_t xs = xlds ? (size_t)NW * TS * 16 * SKINNY_LDX * 2 : 0;
    return TS * 16 * sizeof(float) + (cs > xs ? cs : xs);
}
template <int TS, int NW, bool NORM, bool XLDS>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(TapGemmP p) {
    constexpr int CH = 4;                          // k-steps (of 32) per register chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char skinny_smem[];
    float* inv_s = reinterpret_cast<float*>(skinny_smem);                                  // [TS*16]
    float (*Cs)[TS * 16][17] = reinterpret_cast<float (*)[TS * 16][17]>(skinny_smem + TS * 16 * sizeof(float));   // [NW]
    bf16_t* Xs = reinterpret_cast<bf16_t*>(skinny_smem + TS * 16 * sizeof(float));         // [NW][TS*16][SKINNY_LDX], aliases Cs
    const int kper = p.K / NW;                     // this wave's share of K (a multiple of 32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * TS);
    const int kw = wave * kper;
    const int nch = kper / (32 * CH);
    const int rem = (kper / 32) % CH;              // k-steps of a last partial chunk
    const bool nv = n0 + fr < p.N;
    const bf16_t* wrow = p.W + (size_t)(nv ? n0 + fr : 0) * (p.ldw ? p.ldw : p.K) + kw + fq * 8;
    const bf16_t* xrow[TS];
    bool tv[TS];
#pragma unroll
    for (int j = 0; j < TS; ++j) {
        const int t = m0 + j * 16 + fr;
        tv[j] = t < p.M;
        xrow[j] = p.X + (size_t)(tv[j] ? t : 0) * p.ldx + kw + fq * 8;
    }
    f32x4 acc[TS];
#pragma unroll
    for (int j = 0; j < TS; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const U4 zero = U4{0u, 0u, 0u, 0u};
    const bf16_t* grow = NORM ? p.gain + kw + fq * 8 : nullptr;
    auto load = [&](U4 (&w)[CH], U4 (&x)[TS][CH], U4 (&g)[CH], int k0, int steps) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (s < steps) {
                w[s] = nv ? *reinterpret_cast<const U4*>(wrow + k0 + s * 32) : zero;  // plain loads measured faster than nt here
                if constexpr (NORM) g[s] = *reinterpret_cast<const U4*>(grow + k0 + s * 32);
                if constexpr (!XLDS) {
#pragma unroll
                    for (int j = 0; j < TS; ++j) x[j][s] = tv[j] ? *reinterpret_cast<const U4*>(xrow[j] + k0 + s * 32) : zero;
                }
            }
        }
        if constexpr (XLDS) {
            // load instruction i = j*CH + s: rows 4i .. 4i+3 of the block column, 16 lanes x 16 B = one 256-byte row piece each
#pragma unroll
            for (int j = 0; j < TS; ++j)
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    const int row = (j * CH + s) * 4 + fq;
                    const bool ok = m0 + row < p.M && fr * 8 < steps * 32;
                    x[j][s] = ok ? *reinterpret_cast<const U4*>(p.X + (size_t)(m0 + row) * p.ldx + kw + k0 + fr * 8) : zero;
                }
        }
    };
    float inv[TS];
    if constexpr (NORM) {
        // 1/rms of every row of this block column from the producer's per-block partial sums of squares
        // one wave per row: its lanes fetch the row's partials in one coalesced load, DPP tree sum (fixed order)
        // (all of a wave's rows are fetched before the first reduction: one memory round trip, not one per row)
        constexpr int RPW = (TS * 16 + NW - 1) / NW;
        float ssr[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int t = wave + i * NW, row = m0 + t;
            float ss = 0.f;
            if (t < TS * 16 && row < p.M)
                for (int b = lane; b < p.ss_nblk; b += 64) ss += p.ss_in[(size_t)row * p.ss_ld + b];
            ssr[i] = ss;
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int t = wave + i * NW;
            const float ss = wave_sum(ssr[i]);
            if (t < TS * 16 && lane == 0) inv_s[t] = rsqrt_exact(ss / (float)p.K + p.eps);
        }
    }
    U4 wa[CH], wb[CH], xa[TS][CH], xb[TS][CH], ga[CH], gb[CH];
    if (nch > 0) load(wa, xa, ga, 0, CH);   // in flight while the norms are summed
    if constexpr (NORM) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TS; ++j) inv[j] = inv_s[j * 16 + fr];
    }
    bf16_t* xs_w = Xs + (size_t)wave * TS * 16 * SKINNY_LDX;   // this wave's patch
    auto compute = [&](const U4 (&w)[CH], const U4 (&x)[TS][CH], const U4 (&g)[CH], int steps) {
        if constexpr (XLDS) {
            // LDS operations of one wave execute in order: the fragment reads below see these writes without a barrier
#pragma unroll
            for (int j = 0; j < TS; ++j)
#pragma unroll
                for (int s = 0; s < CH; ++s)
                    *reinterpret_cast<U4*>(&xs_w[((j * CH + s) * 4 + fq) * SKINNY_LDX + fr * 8]) = x[j][s];
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (s < steps) {
                bf16x8 b;
                __builtin_memcpy(&b, &w[s], 16);
                float gv[8];
                if constexpr (NORM) Vec<bf16_t>::unpack(g[s], gv);
#pragma unroll
                for (int j = 0; j < TS; ++j) {
                    bf16x8 a;
                    U4 xf;
                    if constexpr (XLDS) xf = *reinterpret_cast<const U4*>(&xs_w[(j * 16 + fr) * SKINNY_LDX + s * 32 + fq * 8]);
                    else xf = x[j][s];
                    if constexpr (NORM) {   // x -> round(round(x / rms) * gain), the two roundings of llama.py:172-177
                        float xv[8];
                        Vec<bf16_t>::unpack(xf, xv);
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
                        __builtin_memcpy(&a, &xf, 16);
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    };
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) load(wb, xb, gb, (c + 1) * 32 * CH, CH);
        compute(wa, xa, ga, CH);
        if (c + 2 < nch) load(wa, xa, ga, (c + 2) * 32 * CH, CH);
        if (c + 1 < nch) compute(wb, xb, gb, CH);
    }
    if (rem) {
        load(wa, xa, ga, nch * 32 * CH, rem);
        compute(wa, xa, ga, rem);
    }
    if constexpr (XLDS) __syncthreads();   // Cs aliases the X patches: every wave is done reading
    // lane holds C[token = j*16 + 4*fq + r][n = fr]
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[wave][j * 16 + fq * 4 + r][fr] = acc[j][r];
    __syncthreads();
    if (p.act == ACT_SWIGLU) {
        for (int e = tid; e < TS * 16 * 8; e += NW * 64) {
            const int row = e >> 3, c2 = (e & 7) * 2;
            const int t = m0 + row, n = n0 + c2;
            if (t >= p.M || n + 1 >= p.N) continue;
            float g = Cs[0][row][c2], u = Cs[0][row][c2 + 1];
#pragma unroll
            for (int w = 1; w < NW; ++w) { g += Cs[w][row][c2]; u += Cs[w][row][c2 + 1]; }
            if (p.bias) { g += p.bias[n % p.n_mod]; u += p.bias[(n + 1) % p.n_mod]; }
            if (p.round_lin) { g = round_bf16(g); u = round_bf16(u); }
            float sg = g / (1.0f + expf(-g));
            if (p.round_lin) sg = round_bf16(sg);
            const float o = sg * u;
            const size_t oi = (size_t)t * p.ldo + (n >> 1);
            if (p.out_bf) p.out_bf[oi] = f32_to_bf16_bits(o);
            if (p.out_f32) p.out_f32[oi] = p.round_f32_out ? round_bf16(o) : o;
        }
    } else {
        for (int e = tid; e < TS * 16 * 16; e += NW * 64) {
            const int row = e >> 4, c = e & 15;      // 16 consecutive lanes finish one row's 16 columns
            const int t = m0 + row, n = n0 + c;
            const bool ok = t < p.M && n < p.N;
            float stored = 0.f;
            if (ok) {
                float v = Cs[0][row][c];
#pragma unroll
                for (int w = 1; w < NW; ++w) v += Cs[w][row][c];
                if (p.bias) v += p.bias[n % p.n_mod];
                if (p.round_lin) v = round_bf16(v);
                if (p.act == ACT_GELU) v = gelu_f(v);                       // (codec: ConvNeXt pwconv1)
                if (p.gamma) v *= p.gamma[n % p.n_mod];                     // (codec: LayerScale / ConvNeXt gamma)
                if (p.resid_f32) v += p.resid_f32[(size_t)t * p.ldr + n];
                if (p.resid_bf) v += bf16_bits_to_f32(p.resid_bf[(size_t)t * p.ldr + n]);
                const size_t oi = (size_t)t * p.ldo + n;
                stored = p.round_f32_out ? round_bf16(v) : v;
                if (p.out_f32) p.out_f32[oi] = stored;
                if (p.out_bf) p.out_bf[oi] = f32_to_bf16_bits(v);
            }
            if (p.ss_out) {   // this block's share of the row's sum of squares, for the next GEMM's fused RMSNorm
                const float sq = row16_sum(stored * stored);
                if (c == 0 && t < p.M) p.ss_out[(size_t)t * p.ss_ld + blockIdx.x] = sq;
            }
        }
    }
}

// K split: 128-256 contraction steps per wave (one or two register chunks = a single memory round trip per wave)
static inline int skinny_waves(int N, int K, int TS) {
    if (TS < 4 && K % 384 == 0 && K / 12 >= 128) return 12;   // 3072 -> 12 x 256 (TS = 4 would spill registers)
    if (K % 256 == 0 && (K / 8 >= 256 || (N <= 2048 && K / 8 >= 128))) return 8;
    return 4;
}
template <int TS>
static inline void skinny_gemm_launch(const TapGemmP& p, int gy, hipStream_t st) {
    const dim3 grid((p.N + 15) / 16, gy);
    int nw = skinny_waves(p.N, p.K, TS);
    if (p.gain && TS == 4) nw = 4;   // the fused norm's extra registers: keep the 64-row variant off the spill edge
#define FT_SK(NWV, NORMV, XV)                                                                                       \
    do {                                                                                                              \
        constexpr size_t lds_ = skinny_lds_bytes<TS, NWV>(XV);                                                        \
        static DevOnce once_;                                                                                         \
        if (lds_ > 48 * 1024)                                                                                         \
            once_.run([] { hipFuncSetAttribute((const void*)skinny_gemm_kernel<TS, NWV, NORMV, XV>,                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); });            \
        skinny_gemm_kernel<TS, NWV, NORMV, XV><<<grid, NWV * 64, lds_, st>>>(p);                                      \
    } while (0)
#define FT_SK_X(NWV, NORMV) FT_SK(NWV, NORMV, true)      /* X rows staged through LDS (the direct fragment loads were slower) */
    if (p.gain) {
        switch (nw) {
            case 12: FT_SK_X(12, true); break;
            case 8: FT_SK_X(8, true); break;
            default: FT_SK_X(4, true);
        }
    } else {
        switch (nw) {
            case 12: FT_SK_X(12, false); break;
            case 8: FT_SK_X(8, false); break;
            default: FT_SK_X(4, false);
        }
    }
#undef FT_SK_X
#undef FT_SK
}

// ---- residual vector quantiser decode (vocoder.py:800-811): x[t][:] = sum_i table_i[code_i[t]][:]
// table_i = out_proj_i(codebook_i) + bias_i, precomputed in f32 at load time.
struct RvqP {
    const int* codes;        // [B][ncb+1][T]
    const float* tables;     // semantic table [S0][D] followed by ncb tables [S][D]
    int ncb, S0, S, D, T;
    float* x;                // [B][T][D] f32
};
static __global__ __launch_bounds__(256) void rvq_gather_kernel(RvqP p) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int* cd = p.codes + (size_t)b * (p.ncb + 1) * p.T;
    for (int d = threadIdx.x; d < p.D; d += 256) {
        int c0 = cd[t];
        c0 = c0 < 0 ? 0 : (c0 >= p.S0 ? p.S0 - 1 : c0);
        float zs = p.tables[(size_t)c0 * p.D + d];
        float zr = 0.f;
        for (int i = 0; i < p.ncb; ++i) {
            int c = cd[(size_t)(i + 1) * p.T + t];
            c = c < 0 ? 0 : (c >= p.S ? p.S - 1 : c);
            zr += p.tables[((size_t)p.S0 + (size_t)i * p.S + c) * p.D + d];
        }
        p.x[((size_t)b * p.T + t) * p.D + d] = zs + zr;
    }
}
static __global__ void rvq_table_kernel(const float* cb, const float* w, const float* bias, float* table, int S, int D, int cd) {
    // table[s][d] = sum_j w[d][j] * cb[s][j] + bias[d]
    const long n = (long)S * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i / D), d = (int)(i % D);
        float a = 0.f;
        for (int j = 0; j < cd; ++j) a += w[(size_t)d * cd + j] * cb[(size_t)s * cd + j];
        table[i] = a + bias[d];
    }
}

// ---- RMSNorm over rows (vocoder.py:94-102), f32 in -> bf16 (and optionally f32) out
struct RowNormP {
    const float* x;
    const float* w;
    float eps;
    int D;
    bf16_t* out_bf;
    float* out_f32;
};
static __global__ __launch_bounds__(256) void rmsnorm_rows_kernel(RowNormP p) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const float* x = p.x + row * p.D;
    float ss = 0.f;
    for (int d = threadIdx.x; d < p.D; d += 256) ss = fmaf(x[d], x[d], ss);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = rsqrt_exact((((red[0] + red[1]) + red[2]) + red[3]) / (float)p.D + p.eps);
    for (int d = threadIdx.x; d < p.D; d += 256) {
        const float v = (x[d] * inv) * p.w[d];
        if (p.out_bf) p.out_bf[row * p.D + d] = f32_to_bf16_bits(v);
        if (p.out_f32) p.out_f32[row * p.D + d] = v;
    }
}

// ---- RoPE on the q and k thirds of a [T][3*H*hd] bf16 buffer, in place (vocoder.py:145-156)
static __global__ void rope_qk_kernel(bf16_t* qkv, const float* tab, int T, int H, int hd, int pos0 = 0) {
    const int hp = hd >> 1;
    const long n = (long)T * 2 * H * hp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int pr = (int)(i % hp);
        const int h = (int)((i / hp) % (2 * H));  // q heads then k heads
        const int t = (int)(i / ((long)hp * 2 * H));
        bf16_t* v = qkv + (size_t)t * 3 * H * hd + (size_t)h * hd + 2 * pr;
        const float x0 = bf16_bits_to_f32(v[0]), x1 = bf16_bits_to_f32(v[1]);
        const float c = tab[((size_t)(t + pos0) * hp + pr) * 2], s = tab[((size_t)(t + pos0) * hp + pr) * 2 + 1];
        v[0] = f32_to_bf16_bits(x0 * c - x1 * s);
        v[1] = f32_to_bf16_bits(x1 * c + x0 * s);
    }
}

// ---- window-limited causal attention (vocoder.py:210-214 with the band mask of 325-332):
// one wave per (query t, head); lanes over keys for the scores, lanes over dims for P.V
struct WinAttnP {
    const bf16_t* qkv;  // [T][3*H*hd], q and k already rotated
    bf16_t* y;          // [T][H*hd]
    int T, H, hd, window;
    float scale;
    int t0;             // first query row (0; streamed decode: rows [0, t0) are the carried K/V of earlier chunks); y row = t - t0
};
static __global__ __launch_bounds__(256) void window_attn_kernel(WinAttnP p) {
    __shared__ float q_s[4][128];
    __shared__ float p_s[4][512];   // window <= 512 (the encoder's transformer, vocoder.py:516)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)(p.T - p.t0) * p.H) return;
    const int t = p.t0 + (int)(item / p.H), h = (int)(item % p.H);
    const int hd = p.hd, ld = 3 * p.H * hd;
    const bf16_t* q = p.qkv + (size_t)t * ld + (size_t)h * hd;
    for (int d = lane; d < hd; d += 64) q_s[wave][d] = bf16_bits_to_f32(q[d]);
    __builtin_amdgcn_wave_barrier();
    const int j0 = max(0, t - p.window + 1);
    const int nk = t - j0 + 1;
    float mx = -INFINITY;
    for (int jj = lane; jj < nk; jj += 64) {
        const bf16_t* k = p.qkv + (size_t)(j0 + jj) * ld + (size_t)(p.H + h) * hd;
        float s = 0.f;
        for (int d = 0; d < hd; d += 8) {
            float kv[8];
            Vec<bf16_t>::load(k + d, kv);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(q_s[wave][d + e], kv[e], s);
        }
        s *= p.scale;
        p_s[wave][jj] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int jj = lane; jj < nk; jj += 64) {
        const float e = expf(p_s[wave][jj] - mx);
        p_s[wave][jj] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < hd; d += 64) {
        float o = 0.f;
        for (int jj = 0; jj < nk; ++jj)
            o = fmaf(p_s[wave][jj], bf16_bits_to_f32(p.qkv[(size_t)(j0 + jj) * ld + (size_t)(2 * p.H + h) * hd + d]), o);
        p.y[(size_t)(t - p.t0) * p.H * hd + (size_t)h * hd + d] = f32_to_bf16_bits(o / sum);
    }
}

// ---- ConvNeXt front half (vocoder.py:667-671): depthwise causal conv k=7 + LayerNorm(eps 1e-6)
struct DwLnP {
    const bf16_t* x;   // [T][C]
    const float* w;    // [C][7]
    const float* b;    // [C]
    const float* lw;
    const float* lb;
    int T, C;
    bf16_t* out;       // [T][C]
    int t_min;         // lowest readable row of x (as TapGemmP::t_min)
};
static __global__ __launch_bounds__(256) void dwconv_ln_kernel(DwLnP p) {
    __shared__ float red[8];
    extern __shared__ float ybuf[];  // [C]
    const int t = blockIdx.x;
    float s1 = 0.f;
    for (int c = threadIdx.x; c < p.C; c += 256) {
        float a = p.b[c];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int tt = t - 6 + k;
            if (tt >= p.t_min) a = fmaf(p.w[c * 7 + k], bf16_bits_to_f32(p.x[(long)tt * p.C + c]), a);
        }
        ybuf[c] = a;
        s1 += a;
    }
    s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s1;
    __syncthreads();
    const float mean = (((red[0] + red[1]) + red[2]) + red[3]) / (float)p.C;
    float s2 = 0.f;
    for (int c = threadIdx.x; c < p.C; c += 256) { const float d = ybuf[c] - mean; s2 = fmaf(d, d, s2); }
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s2;
    __syncthreads();
    const float var = (((red[4] + red[5]) + red[6]) + red[7]) / (float)p.C;
    const float inv = rsqrt_exact(var + 1e-6f);
    for (int c = threadIdx.x; c < p.C; c += 256)
        p.out[(size_t)t * p.C + c] = f32_to_bf16_bits((ybuf[c] - mean) * inv * p.lw[c] + p.lb[c]);
}

// ---- last layer (vocoder.py:631-635): conv k=7 C->1 on the snake'd input, tanh -> f32 audio
struct FinalConvP {
    const bf16_t* xs;  // [T][C]
    const float* w;    // [7][C]
    float bias;
    int T, C;
    float* audio;      // [T]
    int t_min;         // lowest readable row of xs (as TapGemmP::t_min)
};
static __global__ __launch_bounds__(256) void final_conv_tanh_kernel(FinalConvP p) {
    // 16 lanes per output sample, 16 samples per block step.  C % 8 == 0 and C <= 128: lane `sub` owns channels
    // 8 sub .. 8 sub + 7 - one 16-byte load per tap and sample, its 7 x 8 weights stay in registers (the first version read
    // 2-byte elements with a 32-byte stride: 150 us for 85 MB)
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    if (p.C % 8 == 0 && p.C <= 128) {
        const bool on = sub * 8 < p.C;
        float wv[7][8];
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[k][j] = on ? p.w[k * p.C + sub * 8 + j] : 0.f;
        for (long t = (long)blockIdx.x * 16 + grp; t < p.T; t += (long)gridDim.x * 16) {
            U4 x[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const long tt = t - 6 + k;
                x[k] = (on && tt >= p.t_min) ? *reinterpret_cast<const U4*>(p.xs + tt * p.C + sub * 8) : U4{0u, 0u, 0u, 0u};
            }
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                float xv[8];
                Vec<bf16_t>::unpack(x[k], xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) a = fmaf(wv[k][j], xv[j], a);
            }
            a = row16_sum(a);
            if (sub == 0) p.audio[t] = tanhf(a + p.bias);
        }
        return;
    }
    for (long t = (long)blockIdx.x * 16 + grp; t < p.T; t += (long)gridDim.x * 16) {
        float a = 0.f;
        for (int k = 0; k < 7; ++k) {
            const long tt = t - 6 + k;
            if (tt < p.t_min) continue;
            for (int c = sub; c < p.C; c += 16) a = fmaf(p.w[k * p.C + c], bf16_bits_to_f32(p.xs[tt * p.C + c]), a);
        }
        a = row16_sum(a);
        if (sub == 0) p.audio[t] = tanhf(a + p.bias);
    }
}

// ---- weight repacks
static __global__ void pack_conv_kernel(const float* w, bf16_t* o, int Cout, int Cin, int k) {
    // [Cout][Cin][k] -> [k][Cout][Cin]
    const long n = (long)Cout * Cin * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % k), ci = (int)((i / k) % Cin), co = (int)(i / ((long)k * Cin));
        o[((size_t)kk * Cout + co) * Cin + ci] = f32_to_bf16_bits(w[i]);
    }
}
static __global__ void pack_convT_kernel(const float* w, bf16_t* o, int Cin, int Cout, int k, int s) {
    // [Cin][Cout][k] -> [k/s taps][s*Cout][Cin]; tap j, column (r, co) <- w[ci][co][r + j*s]
    const long n = (long)Cin * Cout * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % k), co = (int)((i / k) % Cout), ci = (int)(i / ((long)k * Cout));
        const int j = kk / s, r = kk % s;
        o[((size_t)j * s * Cout + (size_t)r * Cout + co) * Cin + ci] = f32_to_bf16_bits(w[i]);
    }
}
static __global__ void pack_rows_kernel(const float* w, bf16_t* o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        o[i] = f32_to_bf16_bits(w[i]);
}
static __global__ void pack_interleave_kernel(const float* a, const float* b, bf16_t* o, long rows, long K) {
    const long n = rows * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / K, k = i % K;
        o[(2 * r) * K + k] = f32_to_bf16_bits(a[i]);
        o[(2 * r + 1) * K + k] = f32_to_bf16_bits(b[i]);
    }
}
// ---- encode side (DAC.encode, vocoder.py:885-904) ------------------------------------------------------------------
// strided causal conv (CausalConvNet with stride s, kernel k = taps*s, left pad k - s; vocoder.py:394-421) as a tap GEMM
// over the [T/s][s*Cin] view of the time-major input: [Cout][Cin][k] -> [taps][Cout][s*Cin],
// tap a (row offset a - (taps-1)), column jj*Cin + ci <- w[co][ci][a*s + jj]
static __global__ void pack_strided_conv_kernel(const float* w, bf16_t* o, int Cout, int Cin, int k, int s) {
    const long n = (long)Cout * Cin * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % k), ci = (int)((i / k) % Cin), co = (int)(i / ((long)k * Cin));
        const int a = kk / s, jj = kk % s;
        o[((size_t)a * Cout + co) * ((size_t)s * Cin) + (size_t)jj * Cin + ci] = f32_to_bf16_bits(w[i]);
    }
}
// first encoder conv: 1 input channel, k = 7, causal (vocoder.py:552): raw output and the Snake'd copy the first
// residual unit reads
struct EncInP {
    const float* audio;  // [T]
    const float* w;      // [C][7]
    const float* b;      // [C]
    const float* alpha;  // [C]
    long T;
    int C;
    bf16_t* raw;         // [T][C]
    bf16_t* act;         // [T][C]
};
static __global__ __launch_bounds__(256) void enc_conv_in_kernel(EncInP p) {
    const long n = p.T * p.C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long t = i / p.C;
        const int c = (int)(i % p.C);
        float acc = p.b[c];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const long tt = t - 6 + j;
            if (tt >= 0) acc = fmaf(p.w[c * 7 + j], p.audio[tt], acc);
        }
        p.raw[i] = f32_to_bf16_bits(acc);
        p.act[i] = f32_to_bf16_bits(snake_f(acc, p.alpha[c]));
    }
}
static __global__ void snake_bf_rows_kernel(const bf16_t* x, const float* alpha, bf16_t* out, long n, int C) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16_bits(snake_f(bf16_bits_to_f32(x[i]), alpha[i % C]));
}
static __global__ void bf16_rows_to_f32_kernel(const bf16_t* x, float* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = bf16_bits_to_f32(x[i]);
}
// L2-normalised codebook rows and their squared norms (dac VectorQuantize.decode_latents)
static __global__ void normalize_codebook_kernel(const float* cb, float* cbn, float* cn2, int N, int cd) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        float ss = 0.f;
        for (int c = 0; c < cd; ++c) ss += cb[(size_t)i * cd + c] * cb[(size_t)i * cd + c];
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize eps
        float s2 = 0.f;
        for (int c = 0; c < cd; ++c) {
            const float v = cb[(size_t)i * cd + c] * inv;
            cbn[(size_t)i * cd + c] = v;
            s2 += v * v;
        }
        cn2[i] = s2;
    }
}
// Residual vector quantiser search (vocoder.py:765-779 + dac ResidualVectorQuantize.forward): per frame, for every
// codebook in turn: e = in_proj(residual); nearest codebook row by distance between the L2-normalised vectors
// (first index on ties, as torch.max); residual -= out_proj(codebook[idx]) (the folded decode table).  One block per frame.
struct RvqEncP {
    const float* z;       // [T][D]
    const float* inw;     // [R][cd][D]
    const float* inb;     // [R][cd]
    const float* cbn;     // normalised codebooks: [S0][cd] then (R-1) x [S][cd]
    const float* cn2;     // their squared norms, same order
    const float* tables;  // decode tables [S0][D] then (R-1) x [S][D]
    int R, S0, S, D, cd, T;
    int* codes;           // [R][T]
};
static __global__ __launch_bounds__(256) void rvq_encode_kernel(RvqEncP p) {
    extern __shared__ float res[];           // [D]
    __shared__ double part[4][16];
    __shared__ float e_s[16];
    __shared__ float bestv[4];
    __shared__ int besti[4];
    __shared__ int pick;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, cd = p.cd;
    for (int d = tid; d < D; d += 256) res[d] = p.z[(size_t)t * D + d];
    __syncthreads();
    for (int q = 0; q < p.R; ++q) {
        const float* W = p.inw + (size_t)q * cd * D;
        // the projection is accumulated in f64 (f32 operands): its f32 result is then the correctly rounded one at any width
        // (1024 at the real shapes), so an index can differ from an f32 evaluation only where that evaluation's own
        // accumulation error decides
        for (int c = 0; c < cd; ++c) {
            double a = 0.0;
            for (int d = tid; d < D; d += 256) a = fma((double)W[(size_t)c * D + d], (double)res[d], a);
            for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft);
            if (lane == 0) part[wave][c] = a;
        }
        __syncthreads();
        if (tid < cd) e_s[tid] = (float)((((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]) + (double)p.inb[q * cd + tid]);
        __syncthreads();
        float nn = 0.f;
        for (int c = 0; c < cd; ++c) nn += e_s[c] * e_s[c];
        const float inv = 1.0f / fmaxf(sqrtf(nn), 1e-12f);
        float en[16];
        float l2 = 0.f;
        for (int c = 0; c < cd; ++c) { en[c] = e_s[c] * inv; l2 += en[c] * en[c]; }
        const int N = q == 0 ? p.S0 : p.S;
        const size_t off = q == 0 ? 0 : (size_t)p.S0 + (size_t)(q - 1) * p.S;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < N; i += 256) {
            float dot = 0.f;
            for (int c = 0; c < cd; ++c) dot = fmaf(en[c], p.cbn[(off + i) * cd + c], dot);
            const float score = -((l2 - 2.0f * dot) + p.cn2[off + i]);
            if (score > bv) { bv = score; bi = i; }          // ascending i per thread: first maximum stays
        }
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft);
            const int oi = __shfl_xor(bi, sft);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = bestv[0]; int ix = besti[0];
            for (int w = 1; w < 4; ++w) if (bestv[w] > v || (bestv[w] == v && besti[w] < ix)) { v = bestv[w]; ix = besti[w]; }
            pick = ix;
            p.codes[(size_t)q * p.T + t] = ix;
        }
        __syncthreads();
        const float* row = p.tables + (off + (size_t)pick) * D;
        for (int d = tid; d < D; d += 256) res[d] -= row[d];
        __syncthreads();
    }
}

static __global__ void snake_rows_kernel(const float* x, const float* alpha, bf16_t* out, long T, int C) {
    const long n = T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16_bits(snake_f(x[i], alpha[i % C]));
}

// ---- streamed decode: carried context (ft_codec_stream_*)
// A causal convolution of halo H reads rows [-H, 0) of its input: the last H rows of the input of all earlier chunks.
// tail_roll_kernel (a) copies the carried rows in front of the chunk (x[-H .. 0)) and (b) leaves the carry of the NEXT chunk:
// the last H rows of (carried rows ++ this chunk's T rows).  16-byte pieces; C % 8 == 0.
static __global__ void tail_roll_kernel(bf16_t* x, const bf16_t* tail_in, bf16_t* tail_out, int T, int H, int C) {
    const int per = C / 8;
    const long n = (long)H * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per), q = (int)(i % per);
        const U4 old = *reinterpret_cast<const U4*>(tail_in + (size_t)r * C + q * 8);
        *reinterpret_cast<U4*>(x + ((long)r - H) * C + q * 8) = old;
        const int src = r + T - H;          // row of the chunk that becomes carried row r (negative: still a carried row)
        const U4 nv = src >= 0 ? *reinterpret_cast<const U4*>(x + (long)src * C + q * 8)
                               : *reinterpret_cast<const U4*>(tail_in + (size_t)(r + T) * C + q * 8);
        *reinterpret_cast<U4*>(tail_out + (size_t)r * C + q * 8) = nv;
    }
}
// The K and V of the last nh rows before a chunk (window attention): kv_in [W1][2 * HD] -> rows [0, nh) of the qkv work
// buffer (its k and v thirds; carried rows sit at the END of kv_in), and the carry of the next chunk from rows
// [nh + T - nh2, nh + T) of the work buffer.  Two launches (the second reads what the GEMM wrote after the first).
static __global__ void kv_carry_in_kernel(bf16_t* qkv, const bf16_t* kv_in, int nh, int W1, int HD) {
    const int per = 2 * HD / 8;
    const long n = (long)nh * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per), q = (int)(i % per);
        *reinterpret_cast<U4*>(qkv + (size_t)r * 3 * HD + HD + q * 8) =
            *reinterpret_cast<const U4*>(kv_in + (size_t)(W1 - nh + r) * 2 * HD + q * 8);
    }
}
static __global__ void kv_carry_out_kernel(const bf16_t* qkv, bf16_t* kv_out, int rows, int nh2, int W1, int HD) {
    const int per = 2 * HD / 8;
    const long n = (long)nh2 * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per), q = (int)(i % per);
        *reinterpret_cast<U4*>(kv_out + (size_t)(W1 - nh2 + r) * 2 * HD + q * 8) =
            *reinterpret_cast<const U4*>(qkv + (size_t)(rows - nh2 + r) * 3 * HD + HD + q * 8);
    }
}

}  // namespace ft
