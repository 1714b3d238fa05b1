// The codebook loop of ONE frame for a lock-step batch of 2..32 utterances as one persistent launch (gfx950, bf16, the
// s1-mini fast widths of frame_engine.h: ENG_FD / ENG_FH / ENG_FHKV / ENG_FHD / ENG_FF_DIM / ENG_FV).
//
// reference: fish_tts/models/inference.py:116-149 (the loop over codebooks), llama.py:561-580 (forward_generate_fast),
// 229-331 (block), 172-190 (norm, SwiGLU); the launch path it replaces is enqueue_fast_step (engine.hip): ~36 launches
// per codebook position, 5-12 us each whatever the row count.
//
// Same machinery as the batch-1 frame engine (frame_engine.h): one 512-thread workgroup per CU, vectors handed over as
// self-validating {tag, bf16} granules, sc1 polls, the XCD relay, bounded spins, epoch tags.  What differs:
//  * a vector is a MATRIX [row][N] of granules (row = utterance); it is gathered into LDS as bf16 [row][EB_LDX] by all
//    eight waves with eight 1 KiB pieces in flight per wave;
//  * the matrix-vector phases are 16-row weight tiles on v_mfma_f32_16x16x32_bf16: A = the batch rows from LDS, B = the
//    tile's rows straight from the registers they were prefetched into (lane = row fr, 8 contraction steps fq * 8 ..),
//    the eight waves split K and their partial tiles meet in LDS (summed in wave order);
//  * workgroups have ROLES, because a matrix has fewer 16-row tiles than there are workgroups: QKV tiles on workgroups
//    0..127, Wo and head tiles on 128..191, W2 tiles on 192..255, W13 tiles on all (128 of them take two);
//  * attention: workgroup b = (row b / 8, kv head b % 8) keeps that pair's K/V rows of the frame in its LDS and hands over
//    the pair's 128 output values (one more hand-off per layer than batch 1, where every workgroup rebuilds all heads);
//  * the draw of row m runs on workgroup m (eng_sample_small, the batch-1 engine's), the codes travel in one small hop.
// The sums run in another order than the single-row paths (MFMA tiles, K split eight ways), as the MFMA batch path of the
// launches does: rows are judged against the oracle with margins, not bit for bit against their single runs.
#pragma once
#include "frame_engine.h"

namespace ft {

constexpr int EB_M = 32;                      // rows of a launch (lock-step utterances), MFMA-padded
constexpr int EB_LDX = ENG_FD + 8;            // LDS row stride (bf16) of a [row][1024] matrix: 516 dwords, 4 mod 64
constexpr int EB_PB = 8;                      // 1 KiB pieces in flight per wave and poll round
typedef short eb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float eb_f32x4 __attribute__((ext_vector_type(4)));
static_assert(ENG_FD == 1024 && ENG_FH * ENG_FHD == 1024 && ENG_FF_DIM == 3072 && ENG_FV == 1024 && ENG_FHKV == 8 && ENG_FHD == 64 && ENG_NB == 256,
              "roles and tiles below are laid out for these widths");

struct FastBEngP {
    const EngLayer* layers;       // fast layers
    int n_layer, ncb, M;          // M = rows in use (2..32)
    float eps, scale;
    const float* rope;            // [ncb][hd/2][2]
    const bf16_t* fast_norm; const bf16_t* fast_out; const bf16_t* fast_emb;
    const float* hid;             // plain f32 [M][D]: step 0 inputs
    const float* femb;            // plain f32 [M][D]: step 1 inputs (embeddings of the semantic codes)
    // granule matrices, each [2 parities][..][EB_M][N]
    unsigned* gx;                 // [2][n_layer + 1][EB_M * D]
    unsigned* gqkv;               // [2][n_layer][EB_M * QKVN]
    unsigned* gy;                 // [2][n_layer][EB_M * HD]
    unsigned* gxb;                // [2][n_layer][EB_M * D]
    unsigned* gg;                 // [2][n_layer][EB_M * F]
    unsigned* glog;               // [2][EB_M * V]
    unsigned* gcode;              // [ncb][EB_M] raw 16-bit codes
    unsigned* ctl;
    long rep_delta0, rep_stride;  // as in FastEngP
    SampP samp;                   // row 0's sampling state (rows follow at the launch path's strides)
    long noise_cb_stride, noise_off1;
    unsigned long long* stamps;   // diagnostics (tools/batch_engine_probe.py): [3 workgroups 0 / 128 / 192][ncb][n_layer][16] ticks, or nullptr
};
#define EB_STAMP(k) do { if (p.stamps && tid == 0 && (b == 0 || b == 128 || b == 192)) \
    p.stamps[((((size_t)(b == 0 ? 0 : (b == 128 ? 1 : 2)) * p.ncb + cb) * nL + li) * 16) + (k)] = eng_rt(); } while (0)

// eight polls in flight, waited for at once
__device__ __forceinline__ void engb_ld8_sc1(const unsigned* const (&p)[EB_PB], U4 (&v)[EB_PB]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}

// Columns [c0, c0 + 1024) of rows 0..M-1 of the granule matrix g ([row][N]) into dst (LDS, bf16 [row][EB_LDX]).  Called by
// ALL eight waves of EVERY workgroup: each first does its share of the XCD's import (piece p of the rectangle belongs to the
// workgroup of rank p % nr and there to wave (p / nr) % 8), then - if `consume` - polls the XCD's replica for everything.
// engb_import: this workgroup's share of the XCD's import of columns [c0, c0 + 256 * ppr) of rows 0..M-1 (piece p of the
// rectangle belongs to the workgroup of rank p % nr and there to wave (p / nr) % 8).
__device__ __forceinline__ void engb_import(const EngRelay& rl, const unsigned* g, int N, int c0, int ppr, int M, unsigned tag,
                                            int wave, int lane, unsigned* ctl, int* dead, int where) {
    if (!rl.on) return;
    const int npiece = M * ppr;
    unsigned* rep = const_cast<unsigned*>(g) + rl.delta;
    for (int p = rl.rank + wave * rl.nr; p < npiece; p += 8 * rl.nr) {
        const size_t o = (size_t)(p / ppr) * N + c0 + (p % ppr) * 256 + lane * 4;
        EngSpin sp{ctl, dead, 0, 0, where};
        U4 a;
        for (;;) {
            eng_ld1_sc1(g + o, a);
            if (__all(eng_tags_ok(a, tag))) break;
            if (sp.give_up(lane)) return;
        }
        *reinterpret_cast<U4*>(rep + o) = a;      // plain store: stays in this XCD's L2
    }
}
__device__ __forceinline__ void engb_gather(const EngRelay& rl, const unsigned* g, int N, int c0, int M, unsigned tag, bf16_t* dst,
                                            bool consume, int wave, int lane, unsigned* ctl, int* dead, int where, bool import = true) {
    const int npiece = M * 4;                     // 1024 columns = four 1 KiB pieces per row
    if (import) engb_import(rl, g, N, c0, 4, M, tag, wave, lane, ctl, dead, where);
    const unsigned* src = rl.on ? g + rl.delta : g;
    if (!consume) return;
    EngSpin sp{ctl, dead, 0, 0, where};
    for (int p0 = wave; p0 < npiece; p0 += 8 * EB_PB) {
        const unsigned* ad[EB_PB];
        int pc[EB_PB];
#pragma unroll
        for (int i = 0; i < EB_PB; ++i) {
            const int p = p0 + 8 * i < npiece ? p0 + 8 * i : p0;        // (past the end: the batch's first piece again)
            pc[i] = p;
            ad[i] = src + (size_t)(p >> 2) * N + c0 + (p & 3) * 256 + lane * 4;
        }
        U4 v[EB_PB];
        for (;;) {
            engb_ld8_sc1(ad, v);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < EB_PB; ++i) ok = ok && eng_tags_ok(v[i], tag);
            if (__all(ok)) break;
            if (sp.give_up(lane)) return;
        }
#pragma unroll
        for (int i = 0; i < EB_PB; ++i) {
            if (i == 0 || p0 + 8 * i < npiece) {
                uint2 w;
                w.x = (v[i].x & 0xffffu) | (v[i].y << 16);
                w.y = (v[i].z & 0xffffu) | (v[i].w << 16);
                *reinterpret_cast<uint2*>(dst + (size_t)(pc[i] >> 2) * EB_LDX + (pc[i] & 3) * 256 + lane * 4) = w;
            }
        }
    }
}

// 16 weight rows [n0, n0 + 16) x this wave's 128 contraction steps [k0 + wave * 128, +128) into four operand registers
__device__ __forceinline__ void engb_issue(U4 (&w)[4], const bf16_t* W, int K, int n0, int k0, int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w[kk] = eng_ldg16<true>(W + (size_t)(n0 + fr) * K + k0 + wave * 128 + kk * 32 + fq * 8);
}

// acc[mt] += X[mt * 16 .. +16][this wave's 128 steps] x W-tile^T     (X: LDS bf16 [row][EB_LDX], columns 0..1023 = the chunk)
__device__ __forceinline__ void engb_mma(eb_f32x4 (&acc)[2], const U4 (&w)[4], const bf16_t* X, int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const eb_bf16x8 bw = __builtin_bit_cast(eb_bf16x8, w[kk]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const eb_bf16x8 ax = *reinterpret_cast<const eb_bf16x8*>(X + (size_t)(mt * 16 + fr) * EB_LDX + wave * 128 + kk * 32 + fq * 8);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, bw, acc[mt], 0, 0, 0);
        }
    }
}
// the wave's partial tile to red[wave][row m][n] (lane holds C[m = mt * 16 + 4 * fq + r][n = fr])
__device__ __forceinline__ void engb_red_put(float* red, const eb_f32x4 (&acc)[2], int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 512 + (mt * 16 + 4 * fq + r) * 16 + fr] = acc[mt][r];
}
__device__ __forceinline__ float engb_red_get(const float* red, int i) {
    float s = red[i];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += red[w * 512 + i];
    return s;
}

// RMSNorm of the rows of X (LDS bf16 [row][EB_LDX]) in place: x <- round(round(x * inv) * gain) (llama.py:172-177).  Thread t
// owns columns (t & 15) * 64 .. + 64 of row t >> 4; the row's sum of squares meets on the DPP crossbar of its 16 lanes.
__device__ __forceinline__ void engb_norm(bf16_t* X, const bf16_t* gain, float eps, int M, int tid) {
    const int m = tid >> 4, c = (tid & 15) * 64;
    if (m >= M) return;           // (a row's 16 lanes leave together: the DPP sum below never crosses rows)
    bf16_t* x = X + (size_t)m * EB_LDX + c;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float e8[8];
        Vec<bf16_t>::unpack(*reinterpret_cast<const U4*>(x + j * 8), e8);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(e8[e], e8[e], ss);
    }
    ss = row16_sum(ss);
    const float inv = rsqrt_exact(ss / (float)ENG_FD + eps);
#pragma unroll 2
    for (int j = 0; j < 8; ++j) {           // (the row piece is read from LDS a second time rather than kept in 64 registers)
        float e8[8], g8[8];
        Vec<bf16_t>::unpack(*reinterpret_cast<const U4*>(x + j * 8), e8);
        Vec<bf16_t>::unpack(eng_ldg16<false>(gain + c + j * 8), g8);
        U4 o;
        unsigned* ow = reinterpret_cast<unsigned*>(&o);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float a = round_bf16(round_bf16(e8[e] * inv) * g8[e]);
            const float b = round_bf16(round_bf16(e8[e + 1] * inv) * g8[e + 1]);
            ow[e >> 1] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
        }
        *reinterpret_cast<U4*>(x + j * 8) = o;
    }
}

inline size_t engb_fast_lds_bytes(int nL, int ncb) {
    return (size_t)2 * EB_M * EB_LDX * 2 + 8 * 512 * 4 + (size_t)nL * 2 * ncb * ENG_FHD * 2 + 2048;
}

template <int MAXCB>
__global__ __launch_bounds__(ENG_THREADS) void fastb_engine_kernel(FastBEngP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    constexpr int nb = ENG_NB, D = ENG_FD, F = ENG_FF_DIM, hd = ENG_FHD, H = ENG_FH, Hkv = ENG_FHKV, HD = H * hd, QKVN = (H + 2 * Hkv) * hd, V = ENG_FV;
    const int M = p.M, nL = p.n_layer;
    bf16_t* bufA = reinterpret_cast<bf16_t*>(smem);                       // [EB_M][EB_LDX]
    bf16_t* bufB = bufA + (size_t)EB_M * EB_LDX;                           // [EB_M][EB_LDX]
    float* red = reinterpret_cast<float*>(bufB + (size_t)EB_M * EB_LDX);   // [8][512] partial tiles; between QKV and Wo: q k v + y of the pair
    float* qkvS = red;                                                     // [QKVN] (only this pair's slots are filled)
    float* yS = red + QKVN;                                                // [128]
    bf16_t* kvh = reinterpret_cast<bf16_t*>(red + 8 * 512);                // [nL][2][ncb][hd] K / V rows of this (row, kv head)
    int* misc = reinterpret_cast<int*>(kvh + (size_t)nL * 2 * p.ncb * hd);
    int* dead = misc;            // [1]
    int* out_count = misc + 1;
    int* sub_count = misc + 2;
    int* reg_s = misc + 4;       // [4]
    int* codes_s = misc + 8;     // [EB_M] codes of the step just drawn (every workgroup)
    int* mycodes = misc + 8 + EB_M;   // [MAXCB] this workgroup's row, step by step (drawing workgroups)
    float* codesf = reinterpret_cast<float*>(misc + 8 + EB_M + 16);   // [EB_M] landing zone of the code hop
    // the draw's scratch lives in bufB (free between the head phase and the next step's first gather into it)
    float* logS = reinterpret_cast<float*>(bufB);                  // [V]
    float* prL = logS + V;
    uint32_t* keyL = reinterpret_cast<uint32_t*>(prL + 1024);
    float* redbuf = reinterpret_cast<float*>(keyL + 1024);         // [8]
    int* pen_id = reinterpret_cast<int*>(redbuf + 8);              // [32]
    float* pen_val = reinterpret_cast<float*>(pen_id + 32);        // [32]
    float* amv = pen_val + 32;                                     // [8]
    int* ami = reinterpret_cast<int*>(amv + 8);                    // [8]
    int* wcnt = ami + 8;                                           // [4]
    uint32_t* cut = reinterpret_cast<uint32_t*>(wcnt + 4);         // [4]

    if (tid == 0) { *dead = eng_fault_here(p.ctl, ENG_FAULT_FAST, b) ? 1 : 0; *out_count = 0; *sub_count = 0; }
    if (tid == 64) {
        reg_s[0] = 0; reg_s[1] = 0; reg_s[2] = 1;
        if (p.rep_stride) eng_register(p.ctl, nb, reg_s, dead);
    }
    // rows >= M of both activation buffers stay zero for the whole launch (the tiles are 32 rows)
    for (int i = tid; i < 2 * EB_M * EB_LDX / 2; i += ENG_THREADS) reinterpret_cast<unsigned*>(bufA)[i] = 0u;
    const unsigned epoch = __hip_atomic_load((eng_gu32*)(p.ctl + ENG_CTL_EPOCH), ENG_RLX);
    // hand-off matrices of (step parity, layer)
    auto bx = [&](int par, int l) { return p.gx + ((size_t)par * (nL + 1) + l) * ((size_t)EB_M * D); };
    auto bq = [&](int par, int l) { return p.gqkv + ((size_t)par * nL + l) * ((size_t)EB_M * QKVN); };
    auto by = [&](int par, int l) { return p.gy + ((size_t)par * nL + l) * ((size_t)EB_M * HD); };
    auto bxb = [&](int par, int l) { return p.gxb + ((size_t)par * nL + l) * ((size_t)EB_M * D); };
    auto bg = [&](int par, int l) { return p.gg + ((size_t)par * nL + l) * ((size_t)EB_M * F); };
    auto blog = [&](int par) { return p.glog + (size_t)par * ((size_t)EB_M * V); };
    // roles
    const bool r_qkv = b < 128, r_wo = b >= 128 && b < 192, r_w2 = b >= 192;
    const int t_qkv = b, t_wo = b - 128, t_w2 = b - 192;                 // tile numbers (16 rows each)
    const int a_row = b >> 3, a_kvh = b & 7;                              // attention pair of this workgroup
    const bool a_on = a_row < M;
    const bool drawer = b < M;                                           // draws row b
    // this thread's output of a tile: row m_o of the batch, column n_o of the tile
    const int m_o = tid >> 4, n_o = tid & 15;
    eng_barrier();                                                        // registration, zeroed buffers
    const EngRelay rl{p.rep_stride != 0, reg_s[1], reg_s[2], p.rep_delta0 + (long)reg_s[0] * p.rep_stride};
    bool alive = *dead == 0;

    // weight operand registers: wa = QKV tile | Wo tile | W2 tile (chunk 0), wb = W13 tiles (one or two), wc = W2 chunks 1, 2 | head tile
    U4 wa[4], wb[2][4], wc[2][4];
    auto issue_layer = [&](const EngLayer& l) {
        if (r_qkv) engb_issue(wa, l.wqkv, D, t_qkv * 16, 0, wave, lane);
        if (r_wo) engb_issue(wa, l.wo, HD, t_wo * 16, 0, wave, lane);
        if (r_w2) {
            engb_issue(wa, l.w2, F, t_w2 * 16, 0, wave, lane);
            engb_issue(wc[0], l.w2, F, t_w2 * 16, 1024, wave, lane);
            engb_issue(wc[1], l.w2, F, t_w2 * 16, 2048, wave, lane);
        }
        engb_issue(wb[0], l.w13, D, b * 16, 0, wave, lane);
        if (b < 128) engb_issue(wb[1], l.w13, D, (256 + b) * 16, 0, wave, lane);
    };

    EngSub sub{sub_count, 0};
    SampP sp = p.samp;              // this workgroup's row of the sampling state (drawing workgroups)
    {
        const int R = p.ncb + 1;
        const int m = drawer ? b : 0;
        sp.ctl += m; sp.nf += m; sp.seq += (size_t)m * R * sp.cap; sp.tokn += (size_t)m * R; sp.tok += (size_t)m * R; sp.pos += m; sp.done += m;
    }

    for (int cb = 0; cb < p.ncb && alive; ++cb) {
        const int par = cb & 1;
        const unsigned tag = eng_tag16(epoch + (unsigned)cb);
        const float rcs = p.rope[((size_t)cb * (hd >> 1) + (lane >> 1)) * 2], rsn = p.rope[((size_t)cb * (hd >> 1) + (lane >> 1)) * 2 + 1];
        EngDrawPre pre{};
        if (cb >= 1 && drawer && wave >= ENG_CW) {
            sp.cb = cb;
            sp.noise_off = p.noise_off1 + (long)(cb - 1) * p.noise_cb_stride;
            pre = eng_draw_pre(sp, tid - ENG_CW * 64);
        }
        // layer 0's tiles of this step are requested here, not behind the previous step's last layer: the draw in between
        // needs the registers (its ~60 beside 80 of operands spill), at the price of one exposed load latency per step
        issue_layer(eng_layer(p.layers, 0));
        for (int li = 0; li < nL && alive; ++li) {
            const EngLayer l = eng_layer(p.layers, li);
            const int wh = 2000 + cb * 64 + li * 8;
            // ---- x into bufA: the step's input rows (layer 0) or the previous layer's output
            float resid_o = 0.f;            // Wo role: x[m_o][tile column n_o] for the residual
            if (li == 0) {
                const int m = tid >> 4, c = (tid & 15) * 64;
                if (m < M && (r_qkv || r_wo)) {
                    bf16_t* x = bufA + (size_t)m * EB_LDX + c;
                    if (cb <= 1) {
                        const float* s = (cb == 0 ? p.hid : p.femb) + (size_t)m * D + c;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float4 f = *reinterpret_cast<const float4*>(s + j * 4);
                            uint2 w;
                            w.x = (__float_as_uint(f.x) >> 16) | (__float_as_uint(f.y) & 0xffff0000u);
                            w.y = (__float_as_uint(f.z) >> 16) | (__float_as_uint(f.w) & 0xffff0000u);
                            *reinterpret_cast<uint2*>(x + j * 4) = w;
                        }
                    } else {
                        const bf16_t* s = p.fast_emb + (size_t)codes_s[m] * D + c;
#pragma unroll
                        for (int j = 0; j < 8; ++j) *reinterpret_cast<U4*>(x + j * 8) = eng_ldg16<false>(s + j * 8);
                    }
                }
                eng_barrier();
                if (r_wo && m_o < M) resid_o = bf16_bits_to_f32(bufA[(size_t)m_o * EB_LDX + t_wo * 16 + n_o]);
            } else {
                engb_gather(rl, bx(par, li), D, 0, M, tag, bufA, r_qkv || r_wo, wave, lane, p.ctl, dead, wh + 0);
                eng_barrier(); if (*dead) { alive = false; break; }
                if (r_wo && m_o < M) resid_o = bf16_bits_to_f32(bufA[(size_t)m_o * EB_LDX + t_wo * 16 + n_o]);
            }
            EB_STAMP(0);      // x in LDS
            // ---- QKV tile (workgroups 0..127): attention_norm, rows t_qkv * 16 .. of wqkv, bias
            if (r_qkv) {
                engb_norm(bufA, l.attn_norm, p.eps, M, tid);
                eng_barrier();
                eb_f32x4 acc[2] = {eb_f32x4{0.f, 0.f, 0.f, 0.f}, eb_f32x4{0.f, 0.f, 0.f, 0.f}};
                engb_mma(acc, wa, bufA, wave, lane);
                engb_red_put(red, acc, wave, lane);
                eng_barrier();
                if (m_o < M) {
                    float v = engb_red_get(red, tid);
                    const int n = t_qkv * 16 + n_o;
                    if (l.bqkv) v += eng_ldg_bf16(l.bqkv, n);
                    eng_put(bq(par, li), m_o * QKVN + n, round_bf16(v), tag);
                }
                eng_barrier();                                          // red is reused by the attention below
            }
            EB_STAMP(1);      // q k v published
            // ---- attention of (row a_row, kv head a_kvh): own q k v polled directly, K / V rows of the frame in LDS
            if (a_on && wave == 0) {
                const unsigned* g = bq(par, li) + (size_t)a_row * QKVN;
                const int col = lane < 32 ? a_kvh * 128 + lane * 4 : (lane < 48 ? H * hd + a_kvh * 64 + (lane - 32) * 4 : (H + Hkv) * hd + a_kvh * 64 + (lane - 48) * 4);
                EngSpin spn{p.ctl, dead, 0, 0, wh + 1};
                U4 a;
                bool got = true;
                for (;;) {
                    eng_ld1_sc1(g + col, a);
                    if (__all(eng_tags_ok(a, tag))) break;
                    if (spn.give_up(lane)) { got = false; break; }
                }
                if (got) {
                    eng_unpack_to_lds(qkvS + col, a);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    bf16_t* kL = kvh + (size_t)(li * 2) * p.ncb * hd - (size_t)a_kvh * p.ncb * hd;         // (eng_fast_attn64 adds kvh * ncb * hd)
                    bf16_t* vL = kvh + (size_t)(li * 2 + 1) * p.ncb * hd - (size_t)a_kvh * p.ncb * hd;
                    eng_fast_attn64<MAXCB, 2>(qkvS, kL, vL, yS - a_kvh * 128, l.qn, l.kn, rcs, rsn, cb, p.ncb, H, Hkv, p.eps, p.scale, a_kvh, Hkv, lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    unsigned* gy = by(par, li) + (size_t)a_row * HD + a_kvh * 128;
                    eng_put(gy, lane, yS[lane], tag);
                    eng_put(gy, 64 + lane, yS[64 + lane], tag);
                }
            }
            eng_barrier(); if (*dead) { alive = false; break; }
            EB_STAMP(2);      // own attention done
            // ---- y into bufB, Wo tile + residual (workgroups 128..191) -> x'
            engb_gather(rl, by(par, li), HD, 0, M, tag, bufB, r_wo, wave, lane, p.ctl, dead, wh + 2);
            eng_barrier(); if (*dead) { alive = false; break; }
            EB_STAMP(3);      // y gathered
            if (r_wo) {
                eb_f32x4 acc[2] = {eb_f32x4{0.f, 0.f, 0.f, 0.f}, eb_f32x4{0.f, 0.f, 0.f, 0.f}};
                engb_mma(acc, wa, bufB, wave, lane);
                engb_red_put(red, acc, wave, lane);
                eng_barrier();
                if (m_o < M) {
                    float v = engb_red_get(red, tid);
                    const int n = t_wo * 16 + n_o;
                    if (l.bo) v += eng_ldg_bf16(l.bo, n);
                    v = round_bf16(v);
                    eng_put(bxb(par, li), m_o * D + n, round_bf16(resid_o + v), tag);
                }
            }
            // the next user of this workgroup's first operand set: the next layer's (or the next step's first layer's) tile
            const bool more = li + 1 < nL;
            const EngLayer ln = eng_layer(p.layers, li + 1 < nL ? li + 1 : 0);
            if (more && r_qkv) engb_issue(wa, ln.wqkv, D, t_qkv * 16, 0, wave, lane);
            if (more && r_wo) engb_issue(wa, ln.wo, HD, t_wo * 16, 0, wave, lane);
            // ---- x' into bufA (every workgroup), ffn_norm, W13 tiles + SwiGLU -> g
            EB_STAMP(4);      // Wo published
            engb_gather(rl, bxb(par, li), D, 0, M, tag, bufA, true, wave, lane, p.ctl, dead, wh + 3);
            eng_barrier(); if (*dead) { alive = false; break; }
            EB_STAMP(5);      // x' gathered
            float resid2 = 0.f;             // W2 role: x'[m_o][tile column n_o]
            if (r_w2 && m_o < M) resid2 = bf16_bits_to_f32(bufA[(size_t)m_o * EB_LDX + t_w2 * 16 + n_o]);
            eng_barrier();
            engb_norm(bufA, l.ffn_norm, p.eps, M, tid);
            eng_barrier();
            EB_STAMP(6);      // x' normalised
            for (int ti = 0; ti < (b < 128 ? 2 : 1); ++ti) {
                eb_f32x4 acc[2] = {eb_f32x4{0.f, 0.f, 0.f, 0.f}, eb_f32x4{0.f, 0.f, 0.f, 0.f}};
                engb_mma(acc, wb[ti], bufA, wave, lane);
                engb_red_put(red, acc, wave, lane);
                eng_barrier();
                if (m_o < M && (n_o & 1) == 0) {
                    // rows (2 i, 2 i + 1) of the interleaved matrix = (w1_i, w3_i): llama.py:190
                    const float a = round_bf16(engb_red_get(red, tid));
                    const float bb = round_bf16(engb_red_get(red, tid + 1));
                    const float sg = round_bf16(a / (1.0f + expf(-a)));
                    const int tile = ti == 0 ? b : 256 + b;
                    eng_put(bg(par, li), m_o * F + tile * 8 + (n_o >> 1), round_bf16(sg * bb), tag);
                }
                eng_barrier();
            }
            if (more) {
                engb_issue(wb[0], ln.w13, D, b * 16, 0, wave, lane);
                if (b < 128) engb_issue(wb[1], ln.w13, D, (256 + b) * 16, 0, wave, lane);
            }
            EB_STAMP(7);      // g published
            // ---- g in three chunks of 1024 columns through bufB / bufA / bufB, W2 tile + residual (workgroups 192..255) -> x
            eb_f32x4 acc2[2] = {eb_f32x4{0.f, 0.f, 0.f, 0.f}, eb_f32x4{0.f, 0.f, 0.f, 0.f}};
            bool ok3 = true;
            engb_import(rl, bg(par, li), F, 0, 12, M, tag, wave, lane, p.ctl, dead, wh + 4);      // the whole matrix in one import pass
            if (r_w2) {
                engb_gather(rl, bg(par, li), F, 0, M, tag, bufB, true, wave, lane, p.ctl, dead, wh + 4, false);
                engb_gather(rl, bg(par, li), F, 1024, M, tag, bufA, true, wave, lane, p.ctl, dead, wh + 4, false);
            }
            eng_barrier(); if (*dead) ok3 = false;
            if (ok3 && r_w2) {
                engb_mma(acc2, wa, bufB, wave, lane);
                engb_mma(acc2, wc[0], bufA, wave, lane);
                eng_barrier();
                engb_gather(rl, bg(par, li), F, 2048, M, tag, bufB, true, wave, lane, p.ctl, dead, wh + 4, false);
                eng_barrier(); if (*dead) ok3 = false;
                if (ok3) engb_mma(acc2, wc[1], bufB, wave, lane);
            }
            if (!ok3) { alive = false; break; }
            EB_STAMP(8);      // g gathered (three chunks)
            if (r_w2) {
                engb_red_put(red, acc2, wave, lane);
                eng_barrier();
                if (m_o < M) {
                    const float v = round_bf16(engb_red_get(red, tid));
                    eng_put(bx(par, li + 1), m_o * D + t_w2 * 16 + n_o, round_bf16(resid2 + v), tag);
                }
                if (more) {
                    engb_issue(wa, ln.w2, F, t_w2 * 16, 0, wave, lane);
                    engb_issue(wc[0], ln.w2, F, t_w2 * 16, 1024, wave, lane);
                    engb_issue(wc[1], ln.w2, F, t_w2 * 16, 2048, wave, lane);
                }
            }
            eng_barrier();          // bufA / bufB / red free for the next layer
            EB_STAMP(9);      // layer done
        }
        if (!alive) break;
        if (cb == 0) continue;      // logits of position 0 are discarded (inference.py:122)
        // ---- head: the stack's output rows into bufA, fast_norm, head tiles (workgroups 128..191) -> logits
        if (r_wo) engb_issue(wc[0], p.fast_out, D, t_wo * 16, 0, wave, lane);
        engb_gather(rl, bx(par, nL), D, 0, M, tag, bufA, r_wo, wave, lane, p.ctl, dead, 2000 + cb * 64 + 56);
        eng_barrier(); if (*dead) { alive = false; break; }
        { const int li = nL - 1; EB_STAMP(10); }
        if (r_wo) {
            engb_norm(bufA, p.fast_norm, p.eps, M, tid);
            eng_barrier();
            eb_f32x4 acc[2] = {eb_f32x4{0.f, 0.f, 0.f, 0.f}, eb_f32x4{0.f, 0.f, 0.f, 0.f}};
            engb_mma(acc, wc[0], bufA, wave, lane);
            engb_red_put(red, acc, wave, lane);
            eng_barrier();
            if (m_o < M) eng_put(blog(par), m_o * V + t_wo * 16 + n_o, round_bf16(engb_red_get(red, tid)), tag);
        }
        eng_barrier();
        { const int li = nL - 1; EB_STAMP(11); }
        // ---- the draw of row b's codebook cb on workgroup b (inference.py:134-149), then the codes to everybody
        if (drawer) {
            if (wave >= ENG_CW) {
                const int gw = wave - ENG_CW, atid = tid - ENG_CW * 64;
                eng_gather(blog(par) + (size_t)b * V, EngLayout{0}, 0, V, tag, logS, gw, ENG_GW, lane, p.ctl, dead, 2000 + cb * 64 + 57);
                sub.n = 0;
                if (atid == 0) *sub_count = *dead ? 1 : 0;
                sub.sync(lane);
                if (*reinterpret_cast<volatile int*>(sub_count) == 0) {
                    const int last = cb == p.ncb - 1;
                    const int nfv = pre.nfv;
                    EngSampLds S{redbuf, pen_id, pen_val, amv, ami, wcnt, prL, keyL, cut};
                    const int code = eng_sample_small(sp, pre, logS, S, sub, atid, lane, gw);
                    const int R = p.ncb + 1;
                    if (atid == 0) { mycodes[cb] = code; *sub_count = sub.n + 1; }
                    sub.sync(lane);
                    if (atid == 0) {
                        sp.tokn[cb + 1] = code;
                        eng_put_raw(p.gcode + (size_t)cb * EB_M, b, (unsigned)code, tag);
                    }
                    if (last) {
                        const int frozen = sp.done[0];
                        if (atid < R) {
                            const int v = atid < 2 ? sp.tokn[atid] : (atid - 1 == cb ? code : mycodes[atid - 1]);
                            sp.tok[atid] = v;
                            if (nfv < sp.cap && !frozen) sp.seq[(size_t)atid * sp.cap + nfv] = v;
                        }
                        if (atid == 0 && !frozen) {
                            sp.pos[0] += 1;
                            sp.nf[0] = nfv + 1;
                            if (sp.tokn[0] == sp.im_end) sp.done[0] = 1;
                        }
                    }
                }
            } else {
                eng_draw_follow(sub_count);
            }
        } else if (b < ((M + 3) & ~3) && tid == 0) {
            eng_put_raw(p.gcode + (size_t)cb * EB_M, b, 0u, tag);     // (the code hop moves whole groups of four granules)
        }
        eng_barrier(); if (*dead) { alive = false; break; }
        { const int li = nL - 1; EB_STAMP(12); }
        if (cb + 1 < p.ncb) {
            // the codes of this step: the next step's input rows are those rows of the codebook-embedding table
            if (wave >= ENG_CW)
                eng_gather_x(rl, p.gcode + (size_t)cb * EB_M, EngLayout{0}, 0, (M + 3) & ~3, tag, codesf, wave - ENG_CW, ENG_GW, lane, p.ctl, dead, 2000 + cb * 64 + 58);
            eng_barrier(); if (*dead) { alive = false; break; }
            if (tid < EB_M) codes_s[tid] = (int)(__float_as_uint(codesf[tid]) >> 16);
            eng_barrier();
            { const int li = nL - 1; EB_STAMP(13); }
        }
    }

    // ---- leave: the last workgroup out advances the epoch for the next launch
    eng_barrier();
    if (tid == 0) {
        const unsigned old = atomicAdd(p.ctl + ENG_CTL_EXIT, 1u);
        if (old + 1 == (unsigned)nb) {
            __hip_atomic_store((eng_gu32*)(p.ctl + ENG_CTL_EXIT), 0u, ENG_RLX);
            __hip_atomic_store((eng_gu32*)(p.ctl + ENG_CTL_ARRIVED), 0u, ENG_RLX);
            for (int x = 0; x < 8; ++x) __hip_atomic_store((eng_gu32*)(p.ctl + ENG_CTL_XCD + x), 0u, ENG_RLX);
            __hip_atomic_store((eng_gu32*)(p.ctl + ENG_CTL_EPOCH), epoch + ENG_EPOCH_STEP, ENG_RLX);
        }
    }
}

}  // namespace ft
