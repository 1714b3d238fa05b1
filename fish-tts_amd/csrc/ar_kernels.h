// Decode-step kernels of the dual-AR transformer for gfx950 (M = 1..few lock-step rows).
//
// Storage convention: activations live in HBM as f32.  In the bf16 model precision every
// value stored is bf16-representable: it is rounded (rb<ROUND>) at exactly the points where
// the reference's eager bf16 path rounds, so greedy indices can match.
// Weights and KV cache are bf16 (or f32 in the f32 precision).
//
// Build with -ffp-contract=off: fused multiply-adds appear only where fmaf() is written
// (dot products); element-wise formulas keep the reference's separate roundings.
#pragma once
#include "common.h"

namespace ft {

enum { PRO_NONE = 0, PRO_RMSNORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

// ------------------------------------------------------------------------------------------
// Weight-streaming GEMV: out[m][n] = epi( sum_k W[n][k] * pro(x[m])[k] + bias[n] )
//   reference: nn.Linear calls of llama.py:240 (wqkv), 283 (wo), 190 (w1/w3/w2),
//   449-451 (vocab head), 578 (fast_output), 590 (fast_project_in); RMSNorm prologue =
//   llama.py:172-177; SwiGLU epilogue = llama.py:190; residual epilogue = llama.py:329-330.
// One wave owns R consecutive rows; lanes stride K in 16-byte pieces (1 KiB per wave
// instruction, fully coalesced); all row loads are issued before anything is consumed.
// ------------------------------------------------------------------------------------------
// Octet-major activation form of the lock-step batch GEMMs (wide_kernels.h): element (row m, column k) of Xo[k / 8][ldm][8]
__device__ __host__ __forceinline__ size_t xo_index(int m, int k, int ldm) { return ((size_t)(k >> 3) * ldm + m) * 8 + (k & 7); }

struct GemvP {
    const void* W;
    const void* bias;
    const float* x;
    int ldx;
    const void* gain;
    float eps;
    float* out;
    int ldo;
    const float* resid;
    int ldr;
    int N, K;
    int pro, epi;
    int nt;  // non-temporal weight loads
};

template <typename WT, int NT, int R>
__device__ __forceinline__ void gemv_issue(const GemvP& p, const int row0, const int lane, U4 (&raw)[R][NT]) {
    constexpr int VEC = Vec<WT>::N;
    constexpr int TILE = 64 * VEC;
    const WT* W = reinterpret_cast<const WT*>(p.W);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = t * TILE + lane * VEC;
            if (row < p.N && k < p.K) {
                const U4* src = reinterpret_cast<const U4*>(W + (size_t)row * p.K + k);
                // streamed-once weights (slow layers, vocab head) bypass the caches' retention;
                // the fast stack's weights are re-read 10x per frame and stay default-policy
                raw[r][t] = p.nt ? __builtin_nontemporal_load(src) : *src;
            } else {
                raw[r][t] = U4{0u, 0u, 0u, 0u};
            }
        }
    }
}

template <typename WT, int NT, int R, int ROUND, typename XLoad>
__device__ __forceinline__ void gemv_finish(const GemvP& p, const int m, const int row0, const int lane,
                                            U4 (&raw)[R][NT], XLoad xload) {
    constexpr int VEC = Vec<WT>::N;
    constexpr int TILE = 64 * VEC;
    const int K = p.K, N = p.N;

    float xv[NT][VEC];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = t * TILE + lane * VEC;
        if (k < K) xload(k, xv[t]);
        else Vec<WT>::zero(xv[t]);
    }

    if (p.pro == PRO_RMSNORM) {
        const WT* gain = reinterpret_cast<const WT*>(p.gain);
        U4 graw[NT];  // gains are fetched before the reduction so their latency overlaps it
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = t * TILE + lane * VEC;
            graw[t] = k < K ? *reinterpret_cast<const U4*>(gain + k) : U4{0u, 0u, 0u, 0u};
        }
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < VEC; ++j) ss = fmaf(xv[t][j], xv[t][j], ss);
        ss = wave_sum(ss);
        const float inv = rsqrt_exact(ss / (float)K + p.eps);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float gv[VEC];
            Vec<WT>::unpack(graw[t], gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) xv[t][j] = rb<ROUND>(rb<ROUND>(xv[t][j] * inv) * gv[j]);
        }
    }

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float wv[VEC];
            Vec<WT>::unpack(raw[r][t], wv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) a = fmaf(wv[j], xv[t][j], a);
        }
        acc[r] = wave_sum(a);
    }

    const WT* bias = reinterpret_cast<const WT*>(p.bias);
    if (p.epi == EPI_SWIGLU) {
        // rows (2i, 2i+1) = (w1_i, w3_i), interleaved at load time
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) {
            const int row = row0 + r;
            if (lane == (r >> 1) && row + 1 < N) {
                const float a = rb<ROUND>(acc[r]);
                const float b = rb<ROUND>(acc[r + 1]);
                const float s = rb<ROUND>(a / (1.0f + expf(-a)));
                p.out[(size_t)m * p.ldo + (row >> 1)] = rb<ROUND>(s * b);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (lane == r && row < N) {
                float v = acc[r];
                if (bias) v += ld_elem(bias, row);
                v = rb<ROUND>(v);
                if (p.epi == EPI_RESID) v = rb<ROUND>(p.resid[(size_t)m * p.ldr + row] + v);
                p.out[(size_t)m * p.ldo + row] = v;
            }
        }
    }
}

template <typename WT, int NT, int R, int ROUND, typename XLoad>
__device__ __forceinline__ void gemv_rows(const GemvP& p, const int m, const int row0, const int lane, XLoad xload) {
    U4 raw[R][NT];
    gemv_issue<WT, NT, R>(p, row0, lane, raw);
    gemv_finish<WT, NT, R, ROUND>(p, m, row0, lane, raw, xload);
}

template <typename WT, int NT, int R, int ROUND>
__global__ __launch_bounds__(256) void gemv_kernel(GemvP p) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= p.N) return;
    const int m = blockIdx.y;
    const float* x = p.x + (size_t)m * p.ldx;
    constexpr int VEC = Vec<WT>::N;
    gemv_rows<WT, NT, R, ROUND>(p, m, row0, lane, [&](int k, float(&v)[VEC]) {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 f = *reinterpret_cast<const float4*>(x + k + j);
            v[j] = f.x; v[j + 1] = f.y; v[j + 2] = f.z; v[j + 3] = f.w;
        }
    });
}

// Lock-step batches: MB utterance rows share one pass over the weights (each wave keeps MB activation vectors in
// registers).  Per row the arithmetic (order of the fma chain, wave reduction, roundings) is exactly that of
// gemv_kernel, so a batched run reproduces the single-utterance run bit for bit.
template <typename WT, int NT, int R, int MB, int ROUND>
__global__ __launch_bounds__(256) void gemv_mb_kernel(GemvP p, int Mrows) {
    constexpr int VEC = Vec<WT>::N;
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= p.N) return;
    const int m0 = blockIdx.y * MB;
    const int K = p.K, N = p.N;
    U4 raw[R][NT];
    gemv_issue<WT, NT, R>(p, row0, lane, raw);
    float xv[MB][NT][VEC];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = min(m0 + mb, Mrows - 1);
        const float* x = p.x + (size_t)m * p.ldx;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = t * TILE + lane * VEC;
            if (k < K) {
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const float4 f = *reinterpret_cast<const float4*>(x + k + j);
                    xv[mb][t][j] = f.x; xv[mb][t][j + 1] = f.y; xv[mb][t][j + 2] = f.z; xv[mb][t][j + 3] = f.w;
                }
            } else {
                Vec<WT>::zero(xv[mb][t]);
            }
        }
    }
    if (p.pro == PRO_RMSNORM) {
        const WT* gain = reinterpret_cast<const WT*>(p.gain);
        U4 graw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = t * TILE + lane * VEC;
            graw[t] = k < K ? *reinterpret_cast<const U4*>(gain + k) : U4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j) ss = fmaf(xv[mb][t][j], xv[mb][t][j], ss);
            ss = wave_sum(ss);
            const float inv = rsqrt_exact(ss / (float)K + p.eps);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float gv[VEC];
                Vec<WT>::unpack(graw[t], gv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) xv[mb][t][j] = rb<ROUND>(rb<ROUND>(xv[mb][t][j] * inv) * gv[j]);
            }
        }
    }
    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[r][mb] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float wv[VEC];
            Vec<WT>::unpack(raw[r][t], wv);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[r][mb] = fmaf(wv[j], xv[mb][t][j], acc[r][mb]);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[r][mb] = wave_sum(acc[r][mb]);
    }
    const WT* bias = reinterpret_cast<const WT*>(p.bias);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + mb;
        if (m >= Mrows) continue;
        if (p.epi == EPI_SWIGLU) {
#pragma unroll
            for (int r = 0; r + 1 < R; r += 2) {
                const int row = row0 + r;
                if (lane == (r >> 1) * MB + mb && row + 1 < N) {
                    const float a = rb<ROUND>(acc[r][mb]);
                    const float b = rb<ROUND>(acc[r + 1][mb]);
                    const float s = rb<ROUND>(a / (1.0f + expf(-a)));
                    p.out[(size_t)m * p.ldo + (row >> 1)] = rb<ROUND>(s * b);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row0 + r;
                if (lane == r * MB + mb && row < N) {
                    float v = acc[r][mb];
                    if (bias) v += ld_elem(bias, row);
                    v = rb<ROUND>(v);
                    if (p.epi == EPI_RESID) v = rb<ROUND>(p.resid[(size_t)m * p.ldr + row] + v);
                    p.out[(size_t)m * p.ldo + row] = v;
                }
            }
        }
    }
}

// RMSNorm of S rows for the MFMA prefill (llama.py:172-177): f32 normalise, round, * gain, round -> bf16
template <typename WT, int ROUND>
__global__ __launch_bounds__(256) void rmsnorm_llama_rows_kernel(const float* x, const void* gain_, float eps, int D,
                                                                 bf16_t* out) {
    __shared__ float red[4];
    const WT* gain = reinterpret_cast<const WT*>(gain_);
    const size_t row = blockIdx.x;
    const float* xr = x + row * D;
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) ss = fmaf(xr[d], xr[d], ss);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = rsqrt_exact((((red[0] + red[1]) + red[2]) + red[3]) / (float)D + eps);
    for (int d = threadIdx.x; d < D; d += 256)
        out[row * D + d] = f32_to_bf16_bits(rb<ROUND>(xr[d] * inv) * ld_elem(gain, d));
}

// ------------------------------------------------------------------------------------------
// Embedding of one input column (llama.py:409-429): token row + masked sum of the codebook
// rows, optional 1/sqrt(ncb+1) at VQ positions.  toks[r*tstride + col], r = 0..ncb.
// ------------------------------------------------------------------------------------------
struct EmbedP {
    const void* emb;
    const void* cb_emb;
    const int* toks;
    long tok_row_stride;  // between codebook rows
    long tok_m_stride;    // between batch rows
    int col;
    float* x;
    int ldx, D, ncb, cbsize, vocab, sem_begin, sem_end, scale;
    float inv_div;  // (float)sqrt(ncb+1), used as a divisor
    bf16_t* xo;     // optional octet-major bf16 copy of x (lock-step batches: the residual stream of wide_kernels.h)
    int xo_ldm;
};

template <typename WT, int ROUND>
__global__ __launch_bounds__(256) void embed_kernel(EmbedP p) {
    const int m = blockIdx.y;
    const int* tk = p.toks + (size_t)m * p.tok_m_stride + p.col;
    const WT* emb = reinterpret_cast<const WT*>(p.emb);
    const WT* cbe = reinterpret_cast<const WT*>(p.cb_emb);
    int t0 = tk[0];
    const bool is_vq = t0 >= p.sem_begin && t0 <= p.sem_end;
    t0 = min(max(t0, 0), p.vocab - 1);
    for (int d = blockIdx.x * 256 + threadIdx.x; d < p.D; d += gridDim.x * 256) {
        float vq = 0.f;
        if (is_vq) {
            // codes first, then all rows (two memory round trips instead of two per codebook); summed in codebook order
            constexpr int MAXCB = 16;
            for (int i0 = 0; i0 < p.ncb; i0 += MAXCB) {
                int c[MAXCB];
                float e[MAXCB];
#pragma unroll
                for (int i = 0; i < MAXCB; ++i) c[i] = i0 + i < p.ncb ? tk[(size_t)(i0 + i + 1) * p.tok_row_stride] : 0;
#pragma unroll
                for (int i = 0; i < MAXCB; ++i) {
                    const int cc = min(max(c[i], 0), p.cbsize - 1);
                    e[i] = i0 + i < p.ncb ? ld_elem(cbe, (size_t)(cc + (i0 + i) * p.cbsize) * p.D + d) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < MAXCB; ++i) if (i0 + i < p.ncb) vq += e[i];
            }
            vq = rb<ROUND>(vq);
        }
        float x = rb<ROUND>(ld_elem(emb, (size_t)t0 * p.D + d) + vq);
        if (p.scale && is_vq) x = rb<ROUND>(x / p.inv_div);
        p.x[(size_t)m * p.ldx + d] = x;
        if (p.xo) p.xo[xo_index(m, d, p.xo_ldm)] = f32_to_bf16_bits(x);
    }
}

// ------------------------------------------------------------------------------------------
// Slow-layer decode attention (S = 1): q/k nn.RMSNorm (llama.py:207-209,246-248), interleaved
// RoPE with the bf16 table (llama.py:594-618), KV-cache append (llama.py:142-149), GQA softmax
// attention over [0, pos] (the reference masks all max_seq_len slots, llama.py:258-274,437 —
// identical result).  grid = (Hkv, nsplit, M): one block per kv head and KV split; groups of
// hd/8 lanes own one cached position each (16-byte K/V pieces), online softmax per group,
// merged through LDS.  nsplit > 1 writes (O, m, l) partials for attn_combine_kernel.
// ------------------------------------------------------------------------------------------
struct AttnP {
    const float* qkv;
    int ldq;
    const void* qn;
    const void* kn;
    const float* rope;  // [n_pos][hd/2][2]
    void* kc;
    void* vc;
    size_t cache_m_stride;  // elements per batch row
    const int* pos;
    int pos_off;
    int H, Hkv, hd, n_slots, nsplit;
    float eps, scale;
    float* y;
    int ldy;
    float* part_o;   // [M][H][nsplit][hd]
    float* part_ml;  // [M][H][nsplit][2]
    // prefill: grid.z indexes prompt positions (pos = pos_off + z, one shared cache); pass 1 appends K/V
    // for every position (kv_only), pass 2 attends reading every key from the cache (no_append)
    int row_is_pos, kv_only, no_append;
    bf16_t* y_bf;    // optional bf16 copy of y
    int y_xo_ldm;    // > 0: y_bf is octet-major Xo[H * hd / 8][y_xo_ldm][8] (wide_kernels.h), and y may be null
    bf16_t* q_out;   // kv_only pass: the normalised, rotated queries [row][H*hd] for the MFMA prompt attention
    const int2* row_sp;  // ragged prompt pass of several slots (prefill_rope_append_kernel): row -> (slot, cache position);
                         // kc / vc then are the layer's base pointers and cache_m_stride the elements between slots
};

template <typename WT, int G, int ROUND>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int kvh = blockIdx.x, split = blockIdx.y, m = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = p.hd, hp = hd >> 1;
    const int pos = p.row_is_pos ? p.pos_off + m : p.pos[m] + p.pos_off;
    const int LPP = hd >> 3;       // lanes per cached position
    const int PPW = 64 / LPP;      // positions per wave step
    const int NSLOT = 4 * PPW;
    float* q_s = smem;                   // [G][hd]
    float* k_new = q_s + G * hd;         // [hd]
    float* v_new = k_new + hd;           // [hd]
    float* ml_s = v_new + hd;            // [NSLOT][G][2]
    float* acc_s = ml_s + NSLOT * G * 2; // [NSLOT][G][hd]
    const float* qkv = p.qkv + (size_t)m * p.ldq;
    const WT* qn = reinterpret_cast<const WT*>(p.qn);
    const WT* kn = reinterpret_cast<const WT*>(p.kn);
    const int chunk = (pos + p.nsplit) / p.nsplit;  // ceil((pos+1)/nsplit)
    const int lo = split * chunk;
    const int hi = min(lo + chunk, pos + 1);
    const size_t crow = p.row_is_pos ? 0 : (size_t)m * p.cache_m_stride;
    WT* kc = reinterpret_cast<WT*>(p.kc) + crow + (size_t)kvh * p.n_slots * hd;
    WT* vc = reinterpret_cast<WT*>(p.vc) + crow + (size_t)kvh * p.n_slots * hd;
    const int grp = lane / LPP, gl = lane % LPP;
    const int slot = wave * PPW + grp;
    // K/V rows travel in chunks of U block steps: every load of a chunk is issued before the first row is used (one memory
    // round trip per U * NSLOT positions instead of one per NSLOT), the arithmetic keeps the order of the plain loop.  The
    // first chunk is requested now, so it travels while q/k/v are built.
    constexpr int U = sizeof(WT) == 2 ? 8 : 4;
    typename Vec<WT>::Raw kraw[U], vraw[U];
    auto load_chunk = [&](int base0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base0 + u * NSLOT + grp;
            const bool on = j < hi && (j != pos || p.no_append);
            typename Vec<WT>::Raw zr{};
            kraw[u] = on ? Vec<WT>::load_raw(kc + (size_t)j * hd + gl * 8) : zr;
            vraw[u] = on ? Vec<WT>::load_raw(vc + (size_t)j * hd + gl * 8) : zr;
        }
    };
    load_chunk(lo + wave * PPW);

    // phase 1: q heads of this group, new k, new v
    for (int item = wave; item < G + 2; item += 4) {
        const float* src;
        const WT* gain = nullptr;
        float* dst;
        if (item < G) { src = qkv + (size_t)(kvh * G + item) * hd; gain = qn; dst = q_s + item * hd; }
        else if (item == G) { src = qkv + (size_t)(p.H + kvh) * hd; gain = kn; dst = k_new; }
        else { src = qkv + (size_t)(p.H + p.Hkv + kvh) * hd; dst = v_new; }
        if (item == G + 1) {
            for (int e = lane; e < hd; e += 64) dst[e] = src[e];
        } else {
            float x0 = 0.f, x1 = 0.f;
            if (lane < hp) { x0 = src[2 * lane]; x1 = src[2 * lane + 1]; }
            if (gain) {
                const float ss = wave_sum(x0 * x0 + x1 * x1);
                const float inv = rsqrt_exact(ss / (float)hd + p.eps);
                if (lane < hp) {
                    x0 = rb<ROUND>((x0 * inv) * ld_elem(gain, 2 * lane));
                    x1 = rb<ROUND>((x1 * inv) * ld_elem(gain, 2 * lane + 1));
                }
            }
            if (lane < hp) {
                const float c = p.rope[((size_t)pos * hp + lane) * 2];
                const float s = p.rope[((size_t)pos * hp + lane) * 2 + 1];
                dst[2 * lane] = rb<ROUND>(x0 * c - x1 * s);
                dst[2 * lane + 1] = rb<ROUND>(x1 * c + x0 * s);
            }
        }
    }
    __syncthreads();

    if (pos >= lo && pos < hi && !p.no_append) {
        for (int e = tid; e < hd; e += 256) {
            st_elem(kc, (size_t)pos * hd + e, k_new[e]);
            st_elem(vc, (size_t)pos * hd + e, v_new[e]);
        }
    }
    if (p.kv_only) {
        if (p.q_out)
            for (int idx = tid; idx < G * hd; idx += 256)
                p.q_out[(size_t)m * p.H * hd + (size_t)kvh * G * hd + idx] = f32_to_bf16_bits(q_s[idx]);
        return;
    }

    // phase 2: this block's share of the cached positions
    float qr[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) qr[g][e] = q_s[g * hd + gl * 8 + e];
    float mrun[G], lrun[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        mrun[g] = -INFINITY; lrun[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }
    for (int base0 = lo + wave * PPW; base0 < hi; base0 += NSLOT * U) {
        if (base0 != lo + wave * PPW) load_chunk(base0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int base = base0 + u * NSLOT;
            if (base >= hi) break;
            const int j = base + grp;
            const bool valid = j < hi;
            float kv[8], vv[8];
            Vec<WT>::unpack_raw(kraw[u], kv);
            Vec<WT>::unpack_raw(vraw[u], vv);
            if (valid && j == pos && !p.no_append) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { kv[e] = k_new[gl * 8 + e]; vv[e] = v_new[gl * 8 + e]; }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(qr[g][e], kv[e], d);
                if (LPP > 1) d = group_sum_rt(d, LPP);
                if (valid) {
                    const float s = d * p.scale;
                    const float mn = fmaxf(mrun[g], s);
                    const float corr = expf(mrun[g] - mn);
                    const float pj = expf(s - mn);
                    lrun[g] = lrun[g] * corr + pj;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[g][e] = acc[g][e] * corr + pj * vv[e];
                    mrun[g] = mn;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (gl == 0) { ml_s[(slot * G + g) * 2] = mrun[g]; ml_s[(slot * G + g) * 2 + 1] = lrun[g]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc_s[(size_t)(slot * G + g) * hd + gl * 8 + e] = acc[g][e];
    }
    __syncthreads();
    for (int idx = tid; idx < G * hd; idx += 256) {
        const int g = idx / hd, e = idx % hd;
        float M = -INFINITY;
        for (int s = 0; s < NSLOT; ++s) M = fmaxf(M, ml_s[(s * G + g) * 2]);
        float L = 0.f, O = 0.f;
        if (M > -INFINITY) {
            for (int s = 0; s < NSLOT; ++s) {
                const float w = expf(ml_s[(s * G + g) * 2] - M);
                L += ml_s[(s * G + g) * 2 + 1] * w;
                O += acc_s[(size_t)(s * G + g) * hd + e] * w;
            }
        }
        const int head = kvh * G + g;
        if (p.nsplit == 1) {
            const float yo = rb<ROUND>(O / L);
            if (p.y) p.y[(size_t)m * p.ldy + head * hd + e] = yo;
            if (p.y_bf) p.y_bf[p.y_xo_ldm ? xo_index(m, head * hd + e, p.y_xo_ldm) : (size_t)m * p.ldy + head * hd + e] = f32_to_bf16_bits(yo);
        } else {
            const size_t pi = ((size_t)m * p.H + head) * p.nsplit + split;
            p.part_o[pi * hd + e] = O;
            if (e == 0) { p.part_ml[pi * 2] = M; p.part_ml[pi * 2 + 1] = L; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Prompt pass (bf16, MFMA attention): q/k nn.RMSNorm, interleaved RoPE and the K/V append of EVERY prompt position in one
// launch - one block per position, a wave per head, the arithmetic of attn_decode_kernel's first phase (llama.py:207-209,
// 246-248, 594-618, 142-149).  Leaves the finished queries [row][H * hd] in bf16 for flash_prefill_kernel.
// (Before: the decode attention kernel in its kv_only mode, one block per position AND kv head: 35 us per layer at 780
// positions.)
// ------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void prefill_rope_append_kernel(AttnP p) {
    constexpr int HP = HD / 2;
    static_assert(HP <= 64, "one lane per rotated pair");
    const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pos = p.pos_off + row;
    size_t coff = 0;
    if (p.row_sp) { const int2 sp = p.row_sp[row]; pos = sp.y; coff = (size_t)sp.x * p.cache_m_stride; }
    const float* qkv = p.qkv + (size_t)row * p.ldq;
    const bf16_t* qn = reinterpret_cast<const bf16_t*>(p.qn);
    const bf16_t* kn = reinterpret_cast<const bf16_t*>(p.kn);
    bf16_t* kc = reinterpret_cast<bf16_t*>(p.kc) + coff;
    bf16_t* vc = reinterpret_cast<bf16_t*>(p.vc) + coff;
    float c = 1.f, sn = 0.f;
    if (lane < HP) { c = p.rope[((size_t)pos * HP + lane) * 2]; sn = p.rope[((size_t)pos * HP + lane) * 2 + 1]; }
    const int items = p.H + 2 * p.Hkv;
    for (int item = wave; item < items; item += 4) {
        const float* src = qkv + (size_t)item * HD;
        float x0 = 0.f, x1 = 0.f;
        if (lane < HP) { x0 = src[2 * lane]; x1 = src[2 * lane + 1]; }
        if (item >= p.H + p.Hkv) {           // a value head: stored as it is
            const int kvh = item - p.H - p.Hkv;
            if (lane < HP) {
                bf16_t* dst = vc + ((size_t)kvh * p.n_slots + pos) * HD;
                dst[2 * lane] = f32_to_bf16_bits(x0); dst[2 * lane + 1] = f32_to_bf16_bits(x1);
            }
            continue;
        }
        const bf16_t* gain = item < p.H ? qn : kn;
        if (gain) {
            const float ss = wave_sum(x0 * x0 + x1 * x1);
            const float inv = rsqrt_exact(ss / (float)HD + p.eps);
            if (lane < HP) {
                x0 = round_bf16((x0 * inv) * ld_elem(gain, 2 * lane));
                x1 = round_bf16((x1 * inv) * ld_elem(gain, 2 * lane + 1));
            }
        }
        if (lane < HP) {
            const float r0 = round_bf16(x0 * c - x1 * sn), r1 = round_bf16(x1 * c + x0 * sn);
            bf16_t* dst = item < p.H ? p.q_out + (size_t)row * p.H * HD + (size_t)item * HD
                                     : kc + ((size_t)(item - p.H) * p.n_slots + pos) * HD;
            dst[2 * lane] = f32_to_bf16_bits(r0); dst[2 * lane + 1] = f32_to_bf16_bits(r1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Slow-layer decode attention of WIDE lock-step batches (bf16, wide_kernels.h): one block per (kv head, row) walks the row's
// whole context.  Same inputs, rounding points and cache append as attn_decode_kernel (q/k nn.RMSNorm, interleaved RoPE,
// f32 scores and probabilities, one rounding of y), but the softmax is taken in two passes - all scores to LDS, the
// exact maximum, then the weighted sum - instead of an online softmax per lane group: no exponentials or rescaling on
// the per-position chain, every K row of up to 128 positions (and the first V rows) in flight from the first
// instruction.  Sums run in another order than the single-utterance kernel; wide batches are judged against the oracle
// with the bf16 margin (tests/test_ar_gpu.py: test_wide_batch_vs_oracle), not bit for bit.
// LDS: (G + 2) HD + 16 G + NSLOT G HD + G n_sc floats, n_sc >= the longest context + 1.
// ------------------------------------------------------------------------------------------
template <int G, int HD>
__global__ __launch_bounds__(256) void attn_wide_kernel(AttnP p, int n_sc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LPP = HD / 8, PPW = 64 / LPP, NSLOT = 4 * PPW, U = 8, HP = HD / 2;
    const int kvh = blockIdx.x, m = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPP, gl = lane % LPP, slot = wave * PPW + grp;
    float* q_s = smem;                      // [G][HD]
    float* k_new = q_s + G * HD;            // [HD]
    float* v_new = k_new + HD;              // [HD]
    float* red = v_new + HD;                // [2][4][G]
    float* acc_s = red + 8 * G;             // [NSLOT][G][HD]
    float* sc = acc_s + NSLOT * G * HD;     // [G][n_sc]
    const int pos = p.pos[m] + p.pos_off;   // cached positions 0 .. pos-1, the new one is pos
    const int n = pos + 1;
    bf16_t* kc = reinterpret_cast<bf16_t*>(p.kc) + (size_t)m * p.cache_m_stride + (size_t)kvh * p.n_slots * HD;
    bf16_t* vc = reinterpret_cast<bf16_t*>(p.vc) + (size_t)m * p.cache_m_stride + (size_t)kvh * p.n_slots * HD;
    U4 kraw[U], vraw[U];
    auto load_rows = [&](const bf16_t* base, U4 (&r)[U], int base0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base0 + u * NSLOT + slot;
            r[u] = j < pos ? *reinterpret_cast<const U4*>(base + (size_t)j * HD + gl * 8) : U4{0u, 0u, 0u, 0u};
        }
    };
    load_rows(kc, kraw, 0);
    load_rows(vc, vraw, 0);
    // ---- q heads of this group, new k, new v (attn_decode_kernel phase 1; the rotation entries are requested with the inputs)
    const float* qkv = p.qkv + (size_t)m * p.ldq;
    const bf16_t* qn = reinterpret_cast<const bf16_t*>(p.qn);
    const bf16_t* kn = reinterpret_cast<const bf16_t*>(p.kn);
    for (int item = wave; item < G + 2; item += 4) {
        const float* src;
        const bf16_t* gain = nullptr;
        float* dst;
        if (item < G) { src = qkv + (size_t)(kvh * G + item) * HD; gain = qn; dst = q_s + item * HD; }
        else if (item == G) { src = qkv + (size_t)(p.H + kvh) * HD; gain = kn; dst = k_new; }
        else { src = qkv + (size_t)(p.H + p.Hkv + kvh) * HD; dst = v_new; }
        if (item == G + 1) {
            for (int e = lane; e < HD; e += 64) dst[e] = src[e];
        } else {
            float x0 = 0.f, x1 = 0.f, c = 1.f, sn = 0.f, g0 = 1.f, g1 = 1.f;
            if (lane < HP) {
                x0 = src[2 * lane]; x1 = src[2 * lane + 1];
                c = p.rope[((size_t)pos * HP + lane) * 2]; sn = p.rope[((size_t)pos * HP + lane) * 2 + 1];
                if (gain) { g0 = ld_elem(gain, 2 * lane); g1 = ld_elem(gain, 2 * lane + 1); }
            }
            if (gain) {
                const float ss = wave_sum(x0 * x0 + x1 * x1);
                const float inv = rsqrt_exact(ss / (float)HD + p.eps);
                x0 = round_bf16((x0 * inv) * g0);
                x1 = round_bf16((x1 * inv) * g1);
            }
            if (lane < HP) {
                dst[2 * lane] = round_bf16(x0 * c - x1 * sn);
                dst[2 * lane + 1] = round_bf16(x1 * c + x0 * sn);
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < HD; e += 256) {      // llama.py:142-149
        kc[(size_t)pos * HD + e] = f32_to_bf16_bits(k_new[e]);
        vc[(size_t)pos * HD + e] = f32_to_bf16_bits(v_new[e]);
    }
    // ---- pass 1: scores of every position, the maximum per head
    float qr[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) qr[g][e] = q_s[g * HD + gl * 8 + e];
    float mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mx[g] = -INFINITY;
    for (int base0 = 0; base0 < n; base0 += NSLOT * U) {
        if (base0) load_rows(kc, kraw, base0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base0 + u * NSLOT >= n) break;
            const int j = base0 + u * NSLOT + slot;
            float kv[8];
            Vec<bf16_t>::unpack(kraw[u], kv);
            if (j == pos) {
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[e] = k_new[gl * 8 + e];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(qr[g][e], kv[e], d);
                if (LPP > 1) d = group_sum_rt(d, LPP);
                const float sv = d * p.scale;
                if (j < n) {
                    mx[g] = fmaxf(mx[g], sv);
                    if (gl == 0) sc[g * n_sc + j] = sv;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float w = wave_max(mx[g]);
        if (lane == 0) red[wave * G + g] = w;
    }
    __syncthreads();
    float Mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) Mx[g] = fmaxf(fmaxf(red[g], red[G + g]), fmaxf(red[2 * G + g], red[3 * G + g]));
    // ---- pass 2: probabilities (one exponential per head and position over the whole block), their sum
    float ls[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float a = 0.f;
        for (int j = tid; j < n; j += 256) {
            const float e = expf(sc[g * n_sc + j] - Mx[g]);
            sc[g * n_sc + j] = e;
            a += e;
        }
        ls[g] = wave_sum(a);
        if (lane == 0) red[4 * G + wave * G + g] = ls[g];
    }
    __syncthreads();
    // ---- pass 3: the weighted sum of the V rows
    float acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    for (int base0 = 0; base0 < n; base0 += NSLOT * U) {
        if (base0) load_rows(vc, vraw, base0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base0 + u * NSLOT >= n) break;
            const int j = base0 + u * NSLOT + slot;
            if (j < n) {
                float vv[8];
                Vec<bf16_t>::unpack(vraw[u], vv);
                if (j == pos) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = v_new[gl * 8 + e];
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float pj = sc[g * n_sc + j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(pj, vv[e], acc[g][e]);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc_s[(size_t)(slot * G + g) * HD + gl * 8 + e] = acc[g][e];
    __syncthreads();
    for (int idx = tid; idx < G * HD; idx += 256) {
        const int g = idx / HD, e = idx % HD;
        float O = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) O += acc_s[(size_t)(sl * G + g) * HD + e];
        const float L = ((red[4 * G + g] + red[5 * G + g]) + red[6 * G + g]) + red[7 * G + g];
        const float yo = round_bf16(O / L);
        const int head = kvh * G + g;
        if (p.y) p.y[(size_t)m * p.ldy + head * HD + e] = yo;
        if (p.y_bf) p.y_bf[p.y_xo_ldm ? xo_index(m, head * HD + e, p.y_xo_ldm) : (size_t)m * p.ldy + head * HD + e] = f32_to_bf16_bits(yo);
    }
}
constexpr int attn_wide_lds_floats(int G, int HD, int n_sc) { return (G + 2) * HD + 8 * G + 4 * (64 / (HD / 8)) * G * HD + G * n_sc; }

// Merge of the split-KV partials of one head at 4 consecutive elements e..e+3: y = sum_s O_s w_s / sum_s l_s w_s,
// w_s = exp(m_s - max m).  Splits are taken 8 at a time (all loads of a chunk issued before use); up to 8 splits this
// is a single pass, beyond that the running (max, sum, acc) are rescaled per chunk.
__device__ __forceinline__ void merge_splits4(const AttnP& a, size_t base, int e, float (&y)[4]) {
    float M = -INFINITY, L = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c0 = 0; c0 < a.nsplit; c0 += 8) {
        float ms[8], ls[8];
        float4 O[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool on = c0 + s < a.nsplit;
            const size_t bi = base + (on ? c0 + s : 0);
            ms[s] = on ? a.part_ml[bi * 2] : -INFINITY;
            ls[s] = on ? a.part_ml[bi * 2 + 1] : 0.f;
            O[s] = *reinterpret_cast<const float4*>(a.part_o + bi * a.hd + e);
        }
        float Mc = -INFINITY;
#pragma unroll
        for (int s = 0; s < 8; ++s) Mc = fmaxf(Mc, ms[s]);
        if (!(Mc > -INFINITY)) continue;              // nothing visible in this chunk
        const float Mn = fmaxf(M, Mc);
        if (c0 > 0 && M > -INFINITY) {                // rescale what earlier chunks gathered
            const float r = expf(M - Mn);
            L *= r; a0 *= r; a1 *= r; a2 *= r; a3 *= r;
        }
        M = Mn;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float w = ms[s] > -INFINITY ? expf(ms[s] - M) : 0.f;
            L += ls[s] * w;
            a0 += O[s].x * w; a1 += O[s].y * w; a2 += O[s].z * w; a3 += O[s].w * w;
        }
    }
    y[0] = a0 / L; y[1] = a1 / L; y[2] = a2 / L; y[3] = a3 / L;
}

// Wo GEMV (+ residual) whose input vector is assembled on the fly from the split-KV partials of
// attn_decode_kernel: y[head][e] = sum_s O_s w_s / sum_s l_s w_s, w_s = exp(m_s - max m).  Saves the
// separate combine launch; each lane merges only the 8 (4) consecutive elements it multiplies.
template <typename WT, int NT, int R, int ROUND>
__global__ __launch_bounds__(256) void gemv_attn_combine_kernel(GemvP p, AttnP a) {
    extern __shared__ __attribute__((aligned(16))) float y_s[];  // [H*hd]
    const int tid = threadIdx.x, lane = tid & 63;
    const int row0 = (blockIdx.x * 4 + (tid >> 6)) * R;
    const int m = blockIdx.y;
    U4 raw[R][NT];
    gemv_issue<WT, NT, R>(p, row0, lane, raw);  // Wo rows stream in while the partials are merged
    const int K = a.H * a.hd;
    for (int k = 4 * tid; k < K; k += 1024) {   // 4 consecutive elements of one head per thread
        const int head = k / a.hd, e = k % a.hd;
        const size_t base = ((size_t)m * a.H + head) * a.nsplit;
        float yv[4];
        merge_splits4(a, base, e, yv);
        y_s[k] = rb<ROUND>(yv[0]); y_s[k + 1] = rb<ROUND>(yv[1]);
        y_s[k + 2] = rb<ROUND>(yv[2]); y_s[k + 3] = rb<ROUND>(yv[3]);
    }
    __syncthreads();
    if (row0 >= p.N) return;
    constexpr int VEC = Vec<WT>::N;
    gemv_finish<WT, NT, R, ROUND>(p, m, row0, lane, raw, [&](int k, float(&v)[VEC]) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = y_s[k + j];
    });
}

// Split-KV merge as its own launch (wide lock-step batches: the Wo product is an MFMA GEMM over all rows, so the merge
// cannot ride inside it).  Same arithmetic as above; writes the bf16 operand copy.  grid (M), 256 threads.
template <int ROUND>
__global__ __launch_bounds__(256) void attn_combine_rows_kernel(AttnP a) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int K = a.H * a.hd;
    for (int k = 4 * tid; k < K; k += 1024) {
        const int head = k / a.hd, e = k % a.hd;
        const size_t base = ((size_t)m * a.H + head) * a.nsplit;
        float yv[4];
        merge_splits4(a, base, e, yv);
        const float y0 = rb<ROUND>(yv[0]), y1 = rb<ROUND>(yv[1]), y2 = rb<ROUND>(yv[2]), y3 = rb<ROUND>(yv[3]);
        if (a.y) {
            float* y = a.y + (size_t)m * a.ldy + k;
            y[0] = y0; y[1] = y1; y[2] = y2; y[3] = y3;
        }
        if (a.y_bf) {
            bf16_t* yb = a.y_bf + (a.y_xo_ldm ? xo_index(m, k, a.y_xo_ldm) : (size_t)m * a.ldy + k);   // k % 4 == 0: inside one octet
            yb[0] = f32_to_bf16_bits(y0); yb[1] = f32_to_bf16_bits(y1); yb[2] = f32_to_bf16_bits(y2); yb[3] = f32_to_bf16_bits(y3);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fast-layer attention fused into the Wo GEMV (+ residual).  The fast transformer attends
// over <= num_codebooks positions (llama.py:544-580); its attention is the explicit
// matmul/softmax path of llama.py:285-309 whose three intermediate roundings are mirrored
// here.  Every block rebuilds the (tiny) attention output in LDS, then streams its Wo rows.
// ------------------------------------------------------------------------------------------
struct FastAttnP {
    const float* qkv;
    int ldq;
    const void* qn;
    const void* kn;
    const float* rope;  // [ncb][hd/2][2]
    void* kc;
    void* vc;           // [M][Hkv][ncb][hd]
    size_t cache_m_stride;
    int c;              // codebook position of this step (0..ncb-1)
    int H, Hkv, hd, ncb;
    float eps, scale;
    bf16_t* y_bf;       // optional bf16 copy of y (operand of the MFMA Wo GEMM in wide batches)
    int y_xo_ldm;       // > 0: y_bf is octet-major Xo[H * hd / 8][y_xo_ldm][8] (wide_kernels.h), and y may be null
    // paired pass of a wide batch (pair_M > 0, c == 0): grid.y = 2 pair_M; block y < pair_M is utterance y at position 0 (row y
    // of qkv / y_bf), block y >= pair_M is utterance y - pair_M at position 1 (row pair_off + y - pair_M).  The position-1
    // block rebuilds position 0's key from the utterance's position-0 row (the cache row is written by another block of
    // this launch) with the arithmetic of the block that appends it.
    int pair_M, pair_off;
};

constexpr int FAST_MAXCB = 16;

// One wave per query head: lane d owns dimension d (hd <= 64) or dimensions d and d+64 (hd = 128).
// All K/V rows of the <= num_codebooks cached positions are fetched in one round trip.
template <typename WT, int ROUND>
__global__ __launch_bounds__(64) void fast_attn_kernel(FastAttnP a, float* y, int ldy) {
    const int h = blockIdx.x, lane = threadIdx.x;
    const bool second = a.pair_M > 0 && (int)blockIdx.y >= a.pair_M;      // position 1 of the paired pass
    const int u = second ? blockIdx.y - a.pair_M : blockIdx.y;           // utterance (cache row)
    const int m = second ? a.pair_off + u : u;                           // row of qkv / y
    const int hd = a.hd, hp = hd >> 1, H = a.H, Hkv = a.Hkv, G = H / Hkv, c = second ? 1 : a.c, ncb = a.ncb;
    const int kvh = h / G;
    const float* qkv = a.qkv + (size_t)m * a.ldq;
    const WT* qn = reinterpret_cast<const WT*>(a.qn);
    const WT* kn = reinterpret_cast<const WT*>(a.kn);
    WT* kc = reinterpret_cast<WT*>(a.kc) + (size_t)u * a.cache_m_stride + (size_t)kvh * ncb * hd;
    WT* vc = reinterpret_cast<WT*>(a.vc) + (size_t)u * a.cache_m_stride + (size_t)kvh * ncb * hd;
    constexpr int EPL = 2;  // dims per lane: d = lane + 64 e, valid while d < hd
    // issue every load first: cached rows, the new q/k/v, the rotation entries, the norm gains
    float kj[FAST_MAXCB][EPL], vj[FAST_MAXCB][EPL];
#pragma unroll
    for (int j = 0; j < FAST_MAXCB; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int d = lane + 64 * e;
            const bool on = j < c && d < hd && !second;
            kj[j][e] = on ? ld_elem(kc, (size_t)j * hd + d) : 0.f;
            vj[j][e] = on ? ld_elem(vc, (size_t)j * hd + d) : 0.f;
        }
    float q[EPL], kx[EPL], vx[EPL], cs[EPL], sn[EPL], gq[EPL], gk[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int d = lane + 64 * e;
        const bool on = d < hd;
        q[e] = on ? qkv[(size_t)h * hd + d] : 0.f;
        kx[e] = on ? qkv[(size_t)(H + kvh) * hd + d] : 0.f;
        vx[e] = on ? qkv[(size_t)(H + Hkv + kvh) * hd + d] : 0.f;
        cs[e] = on ? a.rope[((size_t)c * hp + (d >> 1)) * 2] : 1.f;
        sn[e] = on ? a.rope[((size_t)c * hp + (d >> 1)) * 2 + 1] : 0.f;
        gq[e] = (on && qn) ? ld_elem(qn, d) : 1.f;
        gk[e] = (on && kn) ? ld_elem(kn, d) : 1.f;
    }
    if (second) {   // position 0's key and value of this utterance, from its position-0 row
        const float* qkv0 = a.qkv + (size_t)u * a.ldq;
        float k0[EPL], c0[EPL], s0[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int d = lane + 64 * e;
            const bool on = d < hd;
            k0[e] = on ? qkv0[(size_t)(H + kvh) * hd + d] : 0.f;
            vj[0][e] = on ? qkv0[(size_t)(H + Hkv + kvh) * hd + d] : 0.f;
            c0[e] = on ? a.rope[(size_t)(d >> 1) * 2] : 1.f;
            s0[e] = on ? a.rope[(size_t)(d >> 1) * 2 + 1] : 0.f;
        }
        if (kn) {
            const float ss = wave_sum(k0[0] * k0[0] + k0[1] * k0[1]);
            const float inv = rsqrt_exact(ss / (float)hd + a.eps);
#pragma unroll
            for (int e = 0; e < EPL; ++e) k0[e] = rb<ROUND>((k0[e] * inv) * gk[e]);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float ko = dpp_f<DPP_XOR1>(k0[e]);
            kj[0][e] = rb<ROUND>((lane & 1) == 0 ? k0[e] * c0[e] - ko * s0[e] : k0[e] * c0[e] + ko * s0[e]);
        }
    }
    if (qn) {  // per-head nn.RMSNorm, one rounding (llama.py:207-209)
        const float ss = wave_sum(q[0] * q[0] + q[1] * q[1]);
        const float inv = rsqrt_exact(ss / (float)hd + a.eps);
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = rb<ROUND>((q[e] * inv) * gq[e]);
    }
    if (kn) {
        const float ss = wave_sum(kx[0] * kx[0] + kx[1] * kx[1]);
        const float inv = rsqrt_exact(ss / (float)hd + a.eps);
#pragma unroll
        for (int e = 0; e < EPL; ++e) kx[e] = rb<ROUND>((kx[e] * inv) * gk[e]);
    }
    // interleaved-pair rotation: the partner element sits in the neighbouring lane
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float qo = dpp_f<DPP_XOR1>(q[e]), ko = dpp_f<DPP_XOR1>(kx[e]);
        const bool even = (lane & 1) == 0;
        // even lane holds x0: x0*c - x1*s ; odd lane holds x1: x1*c + x0*s
        q[e] = rb<ROUND>(even ? q[e] * cs[e] - qo * sn[e] : q[e] * cs[e] + qo * sn[e]);
        kx[e] = rb<ROUND>(even ? kx[e] * cs[e] - ko * sn[e] : kx[e] * cs[e] + ko * sn[e]);
    }
    if (h % G == 0) {  // one head of the group appends to the KV cache (llama.py:142-149)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int d = lane + 64 * e;
            if (d < hd) { st_elem(kc, (size_t)c * hd + d, kx[e]); st_elem(vc, (size_t)c * hd + d, vx[e]); }
        }
    }
    // scores with the three roundings of the explicit path (llama.py:304-309)
    float s[FAST_MAXCB];
#pragma unroll
    for (int j = 0; j < FAST_MAXCB; ++j) {
        if (j <= c) {  // c is uniform: a scalar branch
            const float k0 = j == c ? kx[0] : kj[j][0], k1 = j == c ? kx[1] : kj[j][1];
            const float d = wave_sum(fmaf(q[1], k1, q[0] * k0));
            s[j] = rb<ROUND>(rb<ROUND>(d) * a.scale);
        } else {
            s[j] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < FAST_MAXCB; ++j) mx = fmaxf(mx, s[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < FAST_MAXCB; ++j) {
        if (j <= c) { s[j] = expf(s[j] - mx); sum += s[j]; } else s[j] = 0.f;
    }
    float o[EPL] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FAST_MAXCB; ++j) {
        if (j <= c) {
            const float pj = rb<ROUND>(s[j] / sum);
            o[0] = fmaf(pj, j == c ? vx[0] : vj[j][0], o[0]);
            o[1] = fmaf(pj, j == c ? vx[1] : vj[j][1], o[1]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int d = lane + 64 * e;
        if (d < hd) {
            const float yo = rb<ROUND>(o[e]);
            if (y) y[(size_t)m * ldy + (size_t)h * hd + d] = yo;
            if (a.y_bf) a.y_bf[a.y_xo_ldm ? xo_index(m, h * hd + d, a.y_xo_ldm) : (size_t)m * ldy + (size_t)h * hd + d] = f32_to_bf16_bits(yo);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Sampling head (inference.py:24-80) for one row of logits, one 1024-thread block per row.
// Mirrors: repetition penalty on the window ids (gather all, then scatter), top-p on the
// descending order with the *inclusive* cumulative sum (rank 0 always kept), temperature,
// softmax, argmax(p / q) with q ~ Exp(1).  In the bf16 precision the intermediate roundings
// of the reference (softmax output, cumulative sum, top_p itself, logits/T, p/q) are applied.
// No sort: the cut is located by a bitwise search on the order-preserving integer image of
// the logits (mass of {key >= k} is monotone in k); ties inside the cut class are resolved
// by index (the reference's sort is unstable, so the survivor of an exact tie is unspecified).
// ------------------------------------------------------------------------------------------
struct RowCtl {
    float temperature, top_p, rep;
    int ban_eos;
    unsigned long long seed;
};

struct SampP {
    float* logits;
    int ldl, V;
    const RowCtl* ctl;
    int* tokn;        // [M][ncb+1] frame under construction
    int* seq;         // [M][ncb+1][cap]
    int cap;
    int* nf;          // [M] frames generated so far
    int cb;           // 0: semantic token from the slow head; >= 1: fast codebook cb
    int ncb, sem_begin, im_end, cbsize;
    const void* fast_emb;
    float* femb;
    int Df;
    const float* noise;
    long noise_row_len, noise_off;
    long noise_rows;
    int last;         // finalize the frame after this draw
    int* tok;         // [M][ncb+1] input column of the next slow step
    int* pos;
    int* done;
    bf16_t* femb_xo;  // optional octet-major bf16 copy of femb (lock-step batches, wide_kernels.h)
    int femb_ldm;
    // wide batches: layer 0's q k v of the NEXT codebook step is a row of a table indexed by the drawn code (its input is
    // that code's embedding): the draw leaves the row where the step's attention reads it, and the step skips that launch
    const bf16_t* qkv0_tab;   // [codes][qkv0_n] or null
    float* qkv0_out;          // [M][qkv0_n]
    int qkv0_n;
};

__device__ __forceinline__ uint32_t order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Philox4 { uint32_t w[4]; };
__device__ __forceinline__ Philox4 philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}
__device__ __forceinline__ uint32_t philox_word(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
    return philox4(c0 >> 2, c1, c2, c3, k0, k1).w[c0 & 3];
}
__device__ __forceinline__ float exp1_from_word(uint32_t w) {
    const float u = ((float)(w >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    return fmaxf(-logf(u), 1e-30f);
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// block-wide deterministic reductions over 1024 threads (16 waves)
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#define FT_AM_STEP(C) { ArgMax b_; b_.v = dpp_f<C>(a.v); b_.i = dpp_i<C>(a.i); a = better(a, b_); }
    FT_AM_STEP(DPP_XOR1) FT_AM_STEP(DPP_XOR2) FT_AM_STEP(DPP_HALF_MIRROR) FT_AM_STEP(DPP_MIRROR)
#undef FT_AM_STEP
    ArgMax r{lane_f(a.v, 0), __builtin_amdgcn_readlane(a.i, 0)};
    r = better(r, ArgMax{lane_f(a.v, 16), __builtin_amdgcn_readlane(a.i, 16)});
    r = better(r, ArgMax{lane_f(a.v, 32), __builtin_amdgcn_readlane(a.i, 32)});
    r = better(r, ArgMax{lane_f(a.v, 48), __builtin_amdgcn_readlane(a.i, 48)});
    return r;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, float* redv, int* redi) {
    a = wave_argmax(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { redv[wave] = a.v; redi[wave] = a.i; }
    __syncthreads();
    ArgMax t{redv[0], redi[0]};
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = better(t, ArgMax{redv[w], redi[w]});
    return t;
}

// frame bookkeeping after a draw (inference.py:123-126, 148-155, 206-210); called by all threads of
// the (single) finishing block of row m
template <typename WT>
__device__ __forceinline__ void finish_draw(const SampP& p, const int m, const int winner, const int nfv) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int R = p.ncb + 1;
    int* tokn = p.tokn + (size_t)m * R;
    int* seq = p.seq + (size_t)m * R * p.cap;
    int code = winner;
    if (p.cb == 0) {
        if (tid == 0) tokn[0] = winner;
        code = winner - p.sem_begin;
        code = code < 0 ? 0 : code;
        code = code >= p.cbsize ? p.cbsize - 1 : code;  // reference would raise IndexError here
        if (tid == 0) tokn[1] = code;
    } else if (tid == 0) {
        tokn[p.cb + 1] = code;
    }
    const WT* fe = reinterpret_cast<const WT*>(p.fast_emb);
    // the table row of the next step's layer-0 q k v (wide batches) travels with the embedding row: both depend on the drawn
    // code only, so their loads are issued together (one round trip) before anything is stored; 16-byte pieces
    constexpr int TQ = 2;
    U4 tq[TQ];
    const bool tab16 = p.qkv0_tab && (p.qkv0_n & 7) == 0;
    if (tab16) {
#pragma unroll
        for (int i = 0; i < TQ; ++i) {
            const int d = (tid + i * T) * 8;
            tq[i] = d < p.qkv0_n ? *reinterpret_cast<const U4*>(p.qkv0_tab + (size_t)code * p.qkv0_n + d) : U4{0u, 0u, 0u, 0u};
        }
    }
    for (int d = tid; d < p.Df; d += T) {
        const float v = ld_elem(fe, (size_t)code * p.Df + d);
        p.femb[(size_t)m * p.Df + d] = v;
        if (p.femb_xo) p.femb_xo[xo_index(m, d, p.femb_ldm)] = f32_to_bf16_bits(v);
    }
    if (tab16) {
        float* qo = p.qkv0_out + (size_t)m * p.qkv0_n;
        auto put8 = [&](int d, const U4& r) {
            float v[8];
            Vec<bf16_t>::unpack(r, v);
            *reinterpret_cast<float4*>(qo + d) = float4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4*>(qo + d + 4) = float4{v[4], v[5], v[6], v[7]};
        };
#pragma unroll
        for (int i = 0; i < TQ; ++i) {
            const int d = (tid + i * T) * 8;
            if (d < p.qkv0_n) put8(d, tq[i]);
        }
        for (int d = (tid + TQ * T) * 8; d < p.qkv0_n; d += T * 8)
            put8(d, *reinterpret_cast<const U4*>(p.qkv0_tab + (size_t)code * p.qkv0_n + d));
    } else if (p.qkv0_tab) {
        for (int d = tid; d < p.qkv0_n; d += T)
            p.qkv0_out[(size_t)m * p.qkv0_n + d] = bf16_bits_to_f32(p.qkv0_tab[(size_t)code * p.qkv0_n + d]);
    }
    if (p.last) {
        // a slot that has emitted <|im_end|> (or is parked) stays frozen while the rest of the lock-step batch
        // goes on: its position, frame count and frame store no longer move
        const int frozen = p.done[m];
        __syncthreads();
        if (tid < R) {
            const int v = tokn[tid];
            p.tok[(size_t)m * R + tid] = v;
            if (nfv < p.cap && !frozen) seq[(size_t)tid * p.cap + nfv] = v;
        }
        if (tid == 0 && !frozen) {
            p.pos[m] += 1;
            p.nf[m] = nfv + 1;
            if (tokn[0] == p.im_end) p.done[m] = 1;
        }
    }
}

template <typename WT, int ROUND>
__global__ __launch_bounds__(1024) void sample_block_kernel(SampP p) {
    __shared__ float red[16];
    __shared__ int redi[16];
    __shared__ float pen_val[32];
    __shared__ int pen_id[32];
    const int m = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    float* L = p.logits + (size_t)m * p.ldl;
    const int V = p.V;
    const RowCtl ctl = p.ctl[m];
    const int nfv = p.nf[m];
    const int R = p.ncb + 1;
    int* seq = p.seq + (size_t)m * R * p.cap;

    // -- repetition penalty (inference.py:39-45, window rule 187-191, id choice 109-111/141-145)
    if (nfv > 0) {
        const int i = nfv - 1;
        const int ws = i < 16 ? 0 : i - 16;
        const int npen = p.cb == 0 ? R : 16;
        if (tid < npen) {
            const int id = p.cb == 0 ? seq[(size_t)tid * p.cap + ws + 1]
                                     : seq[(size_t)(p.cb + 1) * p.cap + ws + 1 + tid];
            pen_id[tid] = id;
            if (id >= 0 && id < V) {
                const float s = L[id];
                pen_val[tid] = s < 0.f ? rb<ROUND>(s * ctl.rep) : rb<ROUND>(s / ctl.rep);
            }
        }
        __syncthreads();
        if (tid < npen) {
            const int id = pen_id[tid];
            if (id >= 0 && id < V) L[id] = pen_val[tid];
        }
    }
    if (p.cb == 0 && ctl.ban_eos && tid == 0 && p.im_end < V) L[p.im_end] = -INFINITY;
    __syncthreads();

    // -- max / argmax and the softmax normaliser of the sorted logits (inference.py:48-51)
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int i = tid; i < V; i += T) am = better(am, ArgMax{L[i], i});
    am = block_argmax(am, red, redi);
    const float Lmax = am.v;
    float z = 0.f;
    for (int i = tid; i < V; i += T) z += expf(L[i] - Lmax);
    const float Z = block_sum(z, red);

    const float tp = rb<ROUND>(ctl.top_p);
    auto removed = [&](float cum) { return rb<ROUND>(cum) > tp; };
    auto prob = [&](float l) { return rb<ROUND>(expf(l - Lmax) / Z); };

    // kstar: largest key k with removed(mass{key >= k}); none (0) => everything kept
    uint32_t kstar = 0;
    long istar = -1;  // elements of class kstar with index <= istar are kept
    bool all_kept = false, only_top = false;
    if (removed(prob(Lmax))) {
        only_top = true;  // rank 0 is always kept, every later rank is cut
    } else {
        float tot = 0.f;
        for (int i = tid; i < V; i += T) tot += prob(L[i]);
        tot = block_sum(tot, red);
        if (!removed(tot)) {
            all_kept = true;
        } else {
            const int lowbit = ROUND == 1 ? 16 : 0;
            for (int bit = 31; bit >= lowbit; --bit) {
                const uint32_t cand = kstar | (1u << bit);
                float ms = 0.f;
                for (int i = tid; i < V; i += T) {
                    const float l = L[i];
                    if (order_key(l) >= cand) ms += prob(l);
                }
                ms = block_sum(ms, red);
                if (removed(ms)) kstar = cand;
            }
            // class kstar: member count, mass strictly above; a member's mass follows from its
            // (unique) logit value, recovered by inverting the order-preserving key
            const uint32_t cmask = ROUND == 1 ? 0xffff0000u : 0xffffffffu;
            float above = 0.f, cnt = 0.f;
            for (int i = tid; i < V; i += T) {
                const float l = L[i];
                const uint32_t k = order_key(l) & cmask;
                if (k > kstar) above += prob(l);
                else if (k == kstar) cnt += 1.f;
            }
            above = block_sum(above, red);
            const int icnt = (int)block_sum(cnt, red);
            const uint32_t ubits = (kstar & 0x80000000u) ? (kstar & 0x7fffffffu) : ~(kstar | ~cmask);
            const float pk = prob(__uint_as_float(ubits));
            // members kept = largest n with the inclusive cumulative mass still <= top_p
            int nk = 0;
            {
                int lo_n = 0, hi_n = icnt;
                while (lo_n < hi_n) {
                    const int mid = (lo_n + hi_n + 1) >> 1;
                    if (removed(fmaf((float)mid, pk, above))) hi_n = mid - 1; else lo_n = mid;
                }
                nk = lo_n;
            }
            if (nk >= icnt) istar = V;
            else if (nk == 0) istar = -1;
            else {
                // smallest index bound with exactly nk class members at or below it
                long lo_i = 0, hi_i = V - 1;
                while (lo_i < hi_i) {
                    const long mid = (lo_i + hi_i) >> 1;
                    float cc = 0.f;
                    for (int i = tid; i < V; i += T)
                        if (i <= mid && (order_key(L[i]) & cmask) == kstar) cc += 1.f;
                    cc = block_sum(cc, red);
                    if ((int)cc >= nk) hi_i = mid; else lo_i = mid + 1;
                }
                istar = lo_i;
            }
        }
    }

    // -- temperature, softmax over the kept set, exponential race (inference.py:57-61, 24-27)
    int winner = am.i;
    if (!only_top) {
        const uint32_t cmask = ROUND == 1 ? 0xffff0000u : 0xffffffffu;
        const float Tc = fmaxf(ctl.temperature, 1e-5f);
        auto kept = [&](int i, float l) {
            if (all_kept) return true;
            const uint32_t k = order_key(l) & cmask;
            return k > kstar || (k == kstar && (long)i <= istar);
        };
        const float Mt = rb<ROUND>(Lmax / Tc);
        float z2 = 0.f;
        for (int i = tid; i < V; i += T) {
            const float l = L[i];
            if (kept(i, l)) z2 += expf(rb<ROUND>(l / Tc) - Mt);
        }
        const float Z2 = block_sum(z2, red);
        const float* qrow = nullptr;
        if (p.noise && nfv < p.noise_rows) qrow = p.noise + (size_t)nfv * p.noise_row_len + p.noise_off;
        ArgMax best{-1.f, 0x7fffffff};
        for (int i = tid; i < V; i += T) {
            const float l = L[i];
            float pr = 0.f;
            if (kept(i, l)) pr = rb<ROUND>(expf(rb<ROUND>(l / Tc) - Mt) / Z2);
            float q;
            if (qrow) q = qrow[i];
            else {
                q = exp1_from_word(philox_word((uint32_t)i, (uint32_t)p.cb, (uint32_t)nfv, 0u /* not the slot: a draw depends on (seed, frame, codebook, index) only */,
                                               (uint32_t)ctl.seed, (uint32_t)(ctl.seed >> 32)));
            }
            q = rb<ROUND>(q);
            best = better(best, ArgMax{rb<ROUND>(pr / q), i});
        }
        best = block_argmax(best, red, redi);
        winner = best.i;
    }

    finish_draw<WT>(p, m, winner, nfv);
}


// ------------------------------------------------------------------------------------------
// Same draw for V <= 1024 (the fast codebooks, inference.py:134) by ONE wave: 16 logits per lane
// live in registers, every reduction is a wave butterfly, no barriers.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float draw_noise(const SampP& p, const RowCtl& ctl, const float* qrow, int i, int nfv, int m) {
    if (qrow) return qrow[i];
    return exp1_from_word(philox_word((uint32_t)i, (uint32_t)p.cb, (uint32_t)nfv, 0u /* not the slot: a draw depends on (seed, frame, codebook, index) only */,
                                      (uint32_t)ctl.seed, (uint32_t)(ctl.seed >> 32)));
}
// the four draws of elements 4g .. 4g+3 from one Philox call
__device__ __forceinline__ void draw_noise4(const SampP& p, const RowCtl& ctl, const float* qrow, int i4, int nfv,
                                            int m, int V, float (&q)[4]) {
    if (qrow) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = (i4 + e) < V ? qrow[i4 + e] : 1.f;
        return;
    }
    const Philox4 r = philox4((uint32_t)(i4 >> 2), (uint32_t)p.cb, (uint32_t)nfv, 0u /* not the slot: a draw depends on (seed, frame, codebook, index) only */,
                              (uint32_t)ctl.seed, (uint32_t)(ctl.seed >> 32));
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = exp1_from_word(r.w[e]);
}

// ------------------------------------------------------------------------------------------
// V <= 1024 with one 256-thread block: 4 consecutive logits per thread in registers, wave
// reductions on DPP, one barrier per block-wide reduction (partials alternate between two LDS
// slots).  This is the kernel the nine codebook draws of every frame use.
// ------------------------------------------------------------------------------------------
struct Red4 {
    float* buf;  // [2][4]
    int phase;
    __device__ __forceinline__ float sum(float v) {
        v = wave_sum(v);
        float* slot = buf + 4 * (phase & 1);
        ++phase;
        if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
        __syncthreads();
        return ((slot[0] + slot[1]) + slot[2]) + slot[3];
    }
};

template <typename WT, int ROUND>
__global__ __launch_bounds__(256) void sample_small_kernel(SampP p) {
    __shared__ float redbuf[8];
    __shared__ int pen_id[32];
    __shared__ float amv[4];
    __shared__ int ami[4];
    __shared__ int wcnt[4];
    __shared__ float prL[1024];
    __shared__ uint32_t keyL[1024];
    __shared__ uint32_t cut_k;
    __shared__ int cut_nk, cut_all;
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* L = p.logits + (size_t)m * p.ldl;
    const int V = p.V;
    const RowCtl ctl = p.ctl[m];
    const int nfv = p.nf[m];
    const int R = p.ncb + 1;
    const int* seq = p.seq + (size_t)m * R * p.cap;
    Red4 red{redbuf, 0};
    const int i0 = 4 * tid;
    // repetition penalty (inference.py:38-46): the window's ids are fetched beside the logits (one round trip), the owner of
    // a penalised logit rewrites it from the PRE-penalty value in its registers - what the reference's gather-then-scatter
    // does (duplicate ids write the same value)
    const int npen = nfv > 0 ? (p.cb == 0 ? (R < 32 ? R : 32) : 16) : 0;
    if (tid < npen) {
        const int it = nfv - 1;
        const int ws = it < 16 ? 0 : it - 16;
        pen_id[tid] = p.cb == 0 ? seq[(size_t)tid * p.cap + ws + 1] : seq[(size_t)(p.cb + 1) * p.cap + ws + 1 + tid];
    }
    float l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = (i0 + e) < V ? L[i0 + e] : -INFINITY;
    if (npen > 0) {
        __syncthreads();
        float lp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lp[e] = l[e] < 0.f ? rb<ROUND>(l[e] * ctl.rep) : rb<ROUND>(l[e] / ctl.rep);
        for (int k = 0; k < npen; ++k) {
            const int id = pen_id[k];
            if (id >= i0 && id < i0 + 4 && id < V) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (id == i0 + e) l[e] = lp[e];
            }
        }
    }
    if (p.cb == 0 && ctl.ban_eos && p.im_end < V) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i0 + e == p.im_end) l[e] = -INFINITY;
    }
    auto block_best = [&](ArgMax a) {
        a = wave_argmax(a);
        __syncthreads();
        if (lane == 0) { amv[wave] = a.v; ami[wave] = a.i; }
        __syncthreads();
        ArgMax t{amv[0], ami[0]};
        for (int w = 1; w < 4; ++w) t = better(t, ArgMax{amv[w], ami[w]});
        return t;
    };
    ArgMax am{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (i0 + e < V) am = better(am, ArgMax{l[e], i0 + e});
    am = block_best(am);
    const float Lmax = am.v;
    float ex[4], z = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ex[e] = (i0 + e) < V ? expf(l[e] - Lmax) : 0.f; z += ex[e]; }
    const float Z = red.sum(z);
    const float tp = rb<ROUND>(ctl.top_p);
    auto removed = [&](float cum) { return rb<ROUND>(cum) > tp; };
    constexpr uint32_t cmask = ROUND == 1 ? 0xffff0000u : 0xffffffffu;
    uint32_t key[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool v = (i0 + e) < V;
        key[e] = v ? (order_key(l[e]) & cmask) : 0u;
        prL[i0 + e] = v ? rb<ROUND>(ex[e] / Z) : 0.f;
        keyL[i0 + e] = key[e];
    }
    const bool only_top = removed(rb<ROUND>(1.0f / Z));  // block-uniform
    int winner = am.i;
    if (!only_top) {
        __syncthreads();
        // ---- the top-p cut is searched by ONE wave on all 1024 (mass, key) pairs: 16 per lane, DPP reductions,
        // no barriers inside the 16 (32) dependent steps
        if (wave == 0) {
            float pr[16];
            uint32_t ky[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { pr[e] = prL[lane + 64 * e]; ky[e] = keyL[lane + 64 * e]; }
            float tot = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) tot += pr[e];
            tot = wave_sum(tot);
            uint32_t kstar = 0;
            int nk = 0, all_kept = 0;
            if (!removed(tot)) {
                all_kept = 1;
            } else {
                constexpr int lowbit = ROUND == 1 ? 16 : 0;
                for (int bit = 31; bit >= lowbit; --bit) {
                    const uint32_t cand = kstar | (1u << bit);
                    float ms = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ms += ky[e] >= cand ? pr[e] : 0.f;
                    if (removed(wave_sum(ms))) kstar = cand;
                }
                float above = 0.f, cnt = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (ky[e] > kstar) above += pr[e];
                    else if (ky[e] == kstar && (lane + 64 * e) < V) cnt += 1.f;
                }
                above = wave_sum(above);
                const int icnt = (int)wave_sum(cnt);
                const uint32_t ubits = (kstar & 0x80000000u) ? (kstar & 0x7fffffffu) : ~(kstar | ~cmask);
                const float pk = rb<ROUND>(expf(__uint_as_float(ubits) - Lmax) / Z);
                int lo_n = 0, hi_n = icnt;
                while (lo_n < hi_n) {
                    const int mid = (lo_n + hi_n + 1) >> 1;
                    if (removed(fmaf((float)mid, pk, above))) hi_n = mid - 1; else lo_n = mid;
                }
                nk = lo_n;
            }
            if (lane == 0) { cut_k = kstar; cut_nk = nk; cut_all = all_kept; }
        }
        __syncthreads();
        const uint32_t kstar = cut_k;
        const int nk = cut_nk;
        const bool all_kept = cut_all != 0;
        const float Tc = fmaxf(ctl.temperature, 1e-5f);
        const float Mt = rb<ROUND>(Lmax / Tc);
        // members of the cut class stay in index order = thread order, then element order
        bool mem[4];
        int mine = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mem[e] = (i0 + e) < V && !all_kept && key[e] == kstar; mine += mem[e] ? 1 : 0; }
        int below = 0, wtot = 0;
        const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
        for (int c = 1; c <= 4; ++c) {  // lanes with >= c members
            const unsigned long long bal = __ballot(mine >= c);
            below += __popcll(bal & lower);
            wtot += __popcll(bal);
        }
        if (lane == 0) wcnt[wave] = wtot;
        __syncthreads();
        int rank = below;
        for (int w = 0; w < wave; ++w) rank += wcnt[w];
        bool keep[4];
        float et[4], z2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            keep[e] = (i0 + e) < V && (all_kept || key[e] > kstar || (mem[e] && rank < nk));
            rank += mem[e] ? 1 : 0;
            et[e] = keep[e] ? expf(rb<ROUND>(l[e] / Tc) - Mt) : 0.f;
            z2 += et[e];
        }
        const float Z2 = red.sum(z2);
        const float* qrow = nullptr;
        if (p.noise && nfv < p.noise_rows) qrow = p.noise + (size_t)nfv * p.noise_row_len + p.noise_off;
        float q4[4];
        draw_noise4(p, ctl, qrow, i0, nfv, m, V, q4);
        ArgMax best{-1.f, 0x7fffffff};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i0 + e < V) {
                const float prob = keep[e] ? rb<ROUND>(et[e] / Z2) : 0.f;
                best = better(best, ArgMax{rb<ROUND>(prob / rb<ROUND>(q4[e])), i0 + e});
            }
        }
        winner = block_best(best).i;
    }
    __syncthreads();
    finish_draw<WT>(p, m, winner, nfv);
}

// ------------------------------------------------------------------------------------------
// The same draw for a large vocabulary in bf16 precision, in four short launches (one CU cannot evaluate
// 155 776 exponentials several times per frame in time):
//   1 samp_cut       penalty + ban; COUNT histogram over the 65 536 possible bf16 logit values in the LDS of one block
//                    (every member of a class has the same probability, so counts are enough and integer atomics keep
//                    it deterministic); the same block then walks the histogram from the top: max, softmax normaliser,
//                    inclusive cumulative mass, the cut class k*, how many of its members stay (nk), the normaliser of
//                    the kept set after temperature
//   2 samp_count     members of class k* per 1024-logit chunk (ranks tied members by index)
//   3 samp_race      p/q for every kept logit, best per chunk
//   4 samp_finish    best over chunks + frame bookkeeping
// (measured and removed: the histogram by ~V global atomics + the cut search on its read-back, 40 + 39 us per row; the
// histogram spread over 39 blocks with the search in the last-arriving one, 75 us; count + race + finish as one launch with
// a chained look-back, the same 0.104 ms per frame as three launches; this kernel beside the head GEMV on a forked stream,
// walking behind per-chunk completion counters, 1.416-1.419 against 1.412 ms per frame; round 3: eight blocks counting
// slices of the row into their own LDS images, block 0 adding their non-empty groups - 0.1068 against 0.1062 ms for head +
// draw: the walk is not what the 35 us of samp_cut are, the search on the image is)
// ------------------------------------------------------------------------------------------
struct SampCut {
    unsigned kstar;   // 16-bit class of the cut (valid unless all_kept)
    int nk;           // members of class kstar that stay
    int all_kept;
    int argmax;       // lowest index is not tracked here; filled by samp_race when needed
    float Lmax, Mt, Z2, Tc;
};

struct SampBigP {
    SampP s;
    SampCut* cut;        // [M]
    int* chunk_cnt;      // [M][nchunk]
    float* part_score;   // [M][nchunk]
    int* part_idx;       // [M][nchunk]
    int nchunk;
};

__device__ __forceinline__ float key16_value(unsigned k16) {
    const uint32_t k = k16 << 16;
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~(k | 0xffffu);
    return __uint_as_float(u);
}

// LDS image of the histogram: thread t owns classes [64 t, 64 t + 64) as 32 dwords of two u16 counts,
// rows padded to 33 dwords so the 64 lanes of a wave hit 32 different banks.  Counts >= 65535 (only
// possible for a handful of classes) are kept exactly in a side table.
constexpr int SAMP_TH_THREADS = 1024;
constexpr int SAMP_TH_ROW = 33;
constexpr int SAMP_TH_OVF = 8;
constexpr size_t SAMP_TH_LDS = (size_t)SAMP_TH_THREADS * SAMP_TH_ROW * 4;

// Shared-memory scratch of the cut search (one 1024-thread block per row)
struct SampThShared {
    float red[16];
    int redi[16];
    float wsum[16];
    int wact[16];
    int AL[1024];  // active groups, ascending
    unsigned ovf_key[SAMP_TH_OVF];
    unsigned ovf_cnt[SAMP_TH_OVF];
    int ovf_n;
    SampCut cut_s;
};

// The top-p cut of one row from the LDS image of its class histogram (thread t owns classes [64 t, 64 t + 64),
// `has` = the group holds anything, kmax_t = its highest occupied class or -1).  Writes b.cut[m].
__device__ __forceinline__ void samp_cut_from_image(const SampBigP& b, const int m, const uint32_t* cimg, SampThShared& sh,
                                                    const bool has, const int kmax_t) {
    const SampP& p = b.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RowCtl ctl = p.ctl[m];
    float* red = sh.red;
    int* redi = sh.redi;
    float* wsum = sh.wsum;
    int* wact = sh.wact;
    int* AL = sh.AL;
    unsigned* ovf_key = sh.ovf_key;
    unsigned* ovf_cnt = sh.ovf_cnt;
    int& ovf_n = sh.ovf_n;
    SampCut& cut_s = sh.cut_s;
    // ---- compact the active groups (ascending) so the arithmetic below is spread over all threads
    const unsigned long long hb = __ballot(has);
    if (lane == 0) wact[wave] = __popcll(hb);
    __syncthreads();
    int abase = 0, n_active = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) abase += wact[w]; n_active += wact[w]; }
    if (has) AL[abase + __popcll(hb & ((1ull << lane) - 1ull))] = tid;
    ArgMax km = block_argmax(ArgMax{(float)kmax_t, tid}, red, redi);  // (contains the barriers AL needs)
    const unsigned kmax = (unsigned)(int)km.v;
    const float Lmax = key16_value(kmax);
    const int n_items = n_active * 64;
    const int ipt = (n_items + 1023) / 1024;
    const int i0 = tid * ipt, i1 = min(i0 + ipt, n_items);
    auto item_count = [&](int i, unsigned& k) -> float {
        const int g = AL[i >> 6], j = i & 63;
        k = 64u * g + j;
        const uint32_t w = cimg[g * SAMP_TH_ROW + (j >> 1)];
        unsigned c = (j & 1) ? (w >> 16) : (w & 0xffffu);
        if (c == 65535u)
            for (int q = 0; q < ovf_n && q < SAMP_TH_OVF; ++q) if (ovf_key[q] == k) c = ovf_cnt[q];
        return (float)c;
    };
    float z = 0.f;
    for (int i = i0; i < i1; ++i) {
        unsigned k;
        const float c = item_count(i, k);
        if (c > 0.f) z = fmaf(c, expf(key16_value(k) - Lmax), z);
    }
    const float Z = block_sum(z, red);
    const float tp = round_bf16(ctl.top_p);
    auto removed = [&](float cum) { return round_bf16(cum) > tp; };
    auto prob = [&](unsigned k16) { return round_bf16(expf(key16_value(k16) - Lmax) / Z); };
    float mt = 0.f;
    for (int i = i0; i < i1; ++i) {
        unsigned k;
        const float c = item_count(i, k);
        if (c > 0.f) mt = fmaf(c, prob(k), mt);
    }
    float suf = mt;  // inclusive suffix within the wave (lanes >= mine own higher classes)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
    }
    __syncthreads();
    if (lane == 0) wsum[wave] = suf;
    __syncthreads();
    float above = suf - mt;
    for (int w = wave + 1; w < 16; ++w) above += wsum[w];
    // walk my items from the top; the cut is the first class whose inclusive mass is removed
    int found = -1;
    unsigned f_key = 0;
    float f_above = 0.f, f_cnt = 0.f;
    float run = above;
    for (int i = i1 - 1; i >= i0; --i) {
        unsigned k;
        const float c = item_count(i, k);
        if (c > 0.f) {
            const float nxt = fmaf(c, prob(k), run);
            if (found < 0 && removed(nxt)) { found = i; f_key = k; f_above = run; f_cnt = c; }
            run = nxt;
        }
    }
    ArgMax who = block_argmax(ArgMax{found >= 0 ? (float)tid : -1.f, tid}, red, redi);
    if (tid == 0) { cut_s.all_kept = who.v < 0.f ? 1 : 0; cut_s.kstar = 0; cut_s.nk = 0; }
    __syncthreads();
    if (who.v >= 0.f && tid == (int)who.v) {
        const float pk = prob(f_key);
        int lo_n = 0, hi_n = (int)f_cnt;
        while (lo_n < hi_n) {
            const int mid = (lo_n + hi_n + 1) >> 1;
            if (removed(fmaf((float)mid, pk, f_above))) hi_n = mid - 1; else lo_n = mid;
        }
        if (f_key == kmax && lo_n < 1) lo_n = 1;  // rank 0 is always kept (inference.py:53)
        cut_s.kstar = f_key;
        cut_s.nk = lo_n;
    }
    __syncthreads();
    const unsigned kstar = cut_s.kstar;
    const int nk = cut_s.nk, all_kept = cut_s.all_kept;
    const float Tc = fmaxf(ctl.temperature, 1e-5f);
    const float Mt = round_bf16(Lmax / Tc);
    float z2 = 0.f;
    for (int i = i0; i < i1; ++i) {
        unsigned k;
        const float c = item_count(i, k);
        if (c > 0.f) {
            float n = 0.f;
            if (all_kept || k > kstar) n = c;
            else if (k == kstar) n = (float)nk;
            if (n > 0.f) z2 = fmaf(n, expf(round_bf16(key16_value(k) / Tc) - Mt), z2);
        }
    }
    const float Z2 = block_sum(z2, red);
    if (tid == 0) {
        SampCut o;
        o.kstar = kstar; o.nk = nk; o.all_kept = all_kept; o.argmax = 0;
        o.Lmax = Lmax; o.Mt = Mt; o.Z2 = Z2; o.Tc = Tc;
        b.cut[m] = o;
    }
}

// Histogram and cut search of one row in ONE block: the 65 536 class counters live in LDS as packed u16 pairs (the
// image samp_cut_from_image reads), filled with LDS atomics.  A packed counter wraps only if >= 65 536 logits of the row share
// one bf16 value; the check sum(counts) == V catches that (any wrap changes the total) and the row is recounted exactly
// with saturating updates.  Also applies the repetition penalty / EOS ban to the row.
static __global__ __launch_bounds__(1024) void samp_cut_kernel(SampBigP b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cimg[];
    __shared__ SampThShared sh;
    __shared__ int pen_id[32];
    __shared__ float pen_val[32];
    const SampP& p = b.s;
    const int m = blockIdx.x, tid = threadIdx.x;
    float* L = p.logits + (size_t)m * p.ldl;
    const int V = p.V;
    const RowCtl ctl = p.ctl[m];
    const int nfv = p.nf[m];
    const int R = p.ncb + 1;
    const int* seq = p.seq + (size_t)m * R * p.cap;
    for (int i = tid; i < SAMP_TH_THREADS * SAMP_TH_ROW; i += 1024) cimg[i] = 0u;
    if (tid == 0) sh.ovf_n = 0;
    const int npen = nfv > 0 ? (p.cb == 0 ? R : 16) : 0;
    auto count1 = [&](float v) {
        const unsigned k = order_key(v) >> 16;
        atomicAdd(&cimg[(k >> 6) * SAMP_TH_ROW + ((k & 63u) >> 1)], (k & 1u) ? 0x10000u : 1u);
    };
    {
        if (nfv > 0) {   // repetition penalty (inference.py:38-46): gather all, then scatter (duplicates write the same value)
            const int it = nfv - 1;
            const int ws = it < 16 ? 0 : it - 16;
            if (tid < npen) {
                const int id = p.cb == 0 ? seq[(size_t)tid * p.cap + ws + 1] : seq[(size_t)(p.cb + 1) * p.cap + ws + 1 + tid];
                pen_id[tid] = -1;
                if (id >= 0 && id < V) {
                    const float sv = L[id];
                    pen_id[tid] = id;
                    pen_val[tid] = sv < 0.f ? round_bf16(sv * ctl.rep) : round_bf16(sv / ctl.rep);
                }
            }
            __syncthreads();
            if (tid < npen && pen_id[tid] >= 0) L[pen_id[tid]] = pen_val[tid];
        }
        if (p.cb == 0 && ctl.ban_eos && tid == 0 && p.im_end < V) L[p.im_end] = -INFINITY;
        __syncthreads();
        // 16 logits per thread and step, all four 16-byte loads issued before the first counter update (one block walks
        // the whole row: a load-use chain per element would cost a memory round trip 152 times)
        const int V16 = ((reinterpret_cast<uintptr_t>(L) & 15) == 0) ? (V / 16384) * 16384 : 0;
        // two steps in flight: the loads of step i + 1 are issued before step i's counters are updated
        float4 fa[4], fb[4];
        auto ld16 = [&](float4 (&f)[4], int base) {
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = *reinterpret_cast<const float4*>(L + base + u * 4096 + tid * 4);
        };
        auto cnt16 = [&](const float4 (&f)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { count1(f[u].x); count1(f[u].y); count1(f[u].z); count1(f[u].w); }
        };
        if (V16 > 0) ld16(fa, 0);
        for (int base = 0; base < V16; base += 32768) {
            if (base + 16384 < V16) ld16(fb, base + 16384);
            cnt16(fa);
            if (base + 32768 < V16) ld16(fa, base + 32768);
            if (base + 16384 < V16) cnt16(fb);
        }
        for (int base = V16; base < V; base += 4096) {   // tail (and unaligned rows): 4 scalar loads in flight
            float f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * 1024 + tid; f[u] = i < V ? L[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (base + u * 1024 + tid < V) count1(f[u]);
        }
    }
    __syncthreads();
    // my group: occupancy, highest occupied class, and the row total for the wrap check
    const uint32_t* row = cimg + tid * SAMP_TH_ROW;
    int kmax_t = -1;
    unsigned tot = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        const uint32_t w = row[j];
        const unsigned c0 = w & 0xffffu, c1 = w >> 16;
        if (c0) kmax_t = 64 * tid + 2 * j;
        if (c1) kmax_t = 64 * tid + 2 * j + 1;
        tot += c0 + c1;
    }
    const int total = (int)block_sum((float)tot, sh.red);   // exact: V < 2^24
    if (total != V) {   // block-uniform; a packed counter wrapped: some class holds >= 65 536 logits of this row
        // recount with saturating updates (compare-and-swap) and an exact side table for what exceeds 65 535 - slow, and
        // only ever taken for such degenerate rows
        __shared__ unsigned exc_key[SAMP_TH_OVF];
        __shared__ unsigned exc_cnt[SAMP_TH_OVF];
        for (int i = tid; i < SAMP_TH_THREADS * SAMP_TH_ROW; i += 1024) cimg[i] = 0u;
        if (tid < SAMP_TH_OVF) { exc_key[tid] = 0xffffffffu; exc_cnt[tid] = 0u; }
        __syncthreads();
        for (int i = tid; i < V; i += 1024) {
            const unsigned k = order_key(L[i]) >> 16;
            uint32_t* d = &cimg[(k >> 6) * SAMP_TH_ROW + ((k & 63u) >> 1)];
            const unsigned shft = (k & 1u) * 16u;
            for (;;) {
                const uint32_t old = *reinterpret_cast<volatile uint32_t*>(d);
                if (((old >> shft) & 0xffffu) == 65535u) {          // saturated: the excess goes to the side table
                    for (int q = 0; q < SAMP_TH_OVF; ++q) {
                        const unsigned prev = atomicCAS(&exc_key[q], 0xffffffffu, k);
                        if (prev == 0xffffffffu || prev == k) { atomicAdd(&exc_cnt[q], 1u); break; }
                    }
                    break;
                }
                if (atomicCAS(d, old, old + (1u << shft)) == old) break;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int q = 0; q < SAMP_TH_OVF; ++q)
                if (exc_key[q] != 0xffffffffu) { sh.ovf_key[n] = exc_key[q]; sh.ovf_cnt[n] = 65535u + exc_cnt[q]; ++n; }
            sh.ovf_n = n;
        }
        kmax_t = -1;
        tot = 0;
        for (int j = 0; j < 32; ++j) {
            const uint32_t w = row[j];
            if (w & 0xffffu) kmax_t = 64 * tid + 2 * j;
            if (w >> 16) kmax_t = 64 * tid + 2 * j + 1;
            tot += (w & 0xffffu) + (w >> 16);
        }
        __syncthreads();
    }
    samp_cut_from_image(b, m, cimg, sh, tot != 0u, kmax_t);
}

static __global__ __launch_bounds__(256) void samp_count_kernel(SampBigP b) {
    __shared__ float red[4];
    const SampP& p = b.s;
    const int m = blockIdx.y, tid = threadIdx.x;
    const int c0 = blockIdx.x * 1024;
    const float* L = p.logits + (size_t)m * p.ldl;
    const SampCut cut = b.cut[m];
    float c = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = c0 + 4 * tid + e;
        if (i < p.V && (order_key(L[i]) >> 16) == cut.kstar) c += 1.f;
    }
    c = wave_sum(c);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) b.chunk_cnt[(size_t)m * b.nchunk + blockIdx.x] = (int)(red[0] + red[1] + red[2] + red[3]);
}

static __global__ __launch_bounds__(256) void samp_race_kernel(SampBigP b) {
    __shared__ float red[4];
    __shared__ int redi[4];
    __shared__ int wcnt[4];
    const SampP& p = b.s;
    const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * 1024;
    const float* L = p.logits + (size_t)m * p.ldl;
    const int V = p.V;
    const RowCtl ctl = p.ctl[m];
    const int nfv = p.nf[m];
    const SampCut cut = b.cut[m];
    // rank base: members of the cut class in earlier chunks
    float basef = 0.f;
    for (int i = tid; i < (int)blockIdx.x; i += 256) basef += (float)b.chunk_cnt[(size_t)m * b.nchunk + i];
    basef = wave_sum(basef);
    if (lane == 0) red[wave] = basef;
    __syncthreads();
    const int base = (int)(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    // this thread owns 4 consecutive logits, so index order = thread order
    float l[4];
    bool member[4];
    int mine = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = c0 + 4 * tid + e;
        l[e] = i < V ? L[i] : -INFINITY;
        member[e] = i < V && !cut.all_kept && (order_key(l[e]) >> 16) == cut.kstar;
        mine += member[e] ? 1 : 0;
    }
    int incl = mine;  // inclusive prefix over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wcnt[wave] = incl;
    __syncthreads();
    int rank = base + incl - mine;
    for (int w = 0; w < wave; ++w) rank += wcnt[w];
    const float* qrow = nullptr;
    if (p.noise && nfv < p.noise_rows) qrow = p.noise + (size_t)nfv * p.noise_row_len + p.noise_off;
    ArgMax best{-1.f, 0x7fffffff};
    float q4[4];
    draw_noise4(p, ctl, qrow, c0 + 4 * tid, nfv, m, V, q4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = c0 + 4 * tid + e;
        if (i < V) {
            bool keep = cut.all_kept || (order_key(l[e]) >> 16) > cut.kstar;
            if (member[e]) { keep = rank < cut.nk; ++rank; }
            const float prob = keep ? round_bf16(expf(round_bf16(l[e] / cut.Tc) - cut.Mt) / cut.Z2) : 0.f;
            best = better(best, ArgMax{round_bf16(prob / round_bf16(q4[e])), i});
        }
    }
    best = wave_argmax(best);
    if (lane == 0) { red[wave] = best.v; redi[wave] = best.i; }
    __syncthreads();
    if (tid == 0) {
        ArgMax t{red[0], redi[0]};
        for (int w = 1; w < 4; ++w) t = better(t, ArgMax{red[w], redi[w]});
        b.part_score[(size_t)m * b.nchunk + blockIdx.x] = t.v;
        b.part_idx[(size_t)m * b.nchunk + blockIdx.x] = t.i;
    }
}

template <typename WT>
__global__ __launch_bounds__(256) void samp_finish_kernel(SampBigP b) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const SampP& p = b.s;
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ArgMax best{-2.f, 0x7fffffff};
    for (int i = tid; i < b.nchunk; i += 256)
        best = better(best, ArgMax{b.part_score[(size_t)m * b.nchunk + i], b.part_idx[(size_t)m * b.nchunk + i]});
    best = wave_argmax(best);
    if (lane == 0) { red[wave] = best.v; redi[wave] = best.i; }
    __syncthreads();
    ArgMax t{red[0], redi[0]};
    for (int w = 1; w < 4; ++w) t = better(t, ArgMax{red[w], redi[w]});
    __syncthreads();
    finish_draw<WT>(p, m, t.i, p.nf[m]);
}

}  // namespace ft
