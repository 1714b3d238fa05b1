// Persistent frame engine for the batch-1 decode step (gfx950): the whole slow stack, and the whole fast
// codebook loop, each as ONE launch of one 512-thread workgroup per CU.
//
// Why: at batch 1 a frame is ~300 dependent matrix-vector phases; as separate launches each phase costs
// 4.5-5 us (launch boundary + first-byte latency), 10x the time its weights need on HBM.  Inside one launch a
// phase hands its output vector to every CU as 4-byte granules {16-bit tag, bf16 value} written with sc1
// (write-through) stores and polled with sc1 (L1-bypassing) loads: the data is the flag, no fence, no
// barrier, no atomics (tools/mb_edge2.hip: 0.8 us from "last producer stored" to "every CU has the vector").
// Every CU takes part in every matrix-vector phase with its slice of the rows; weights for the next phase
// are requested before the wave waits for its input.
//
// Arithmetic: every phase reproduces the per-row operation order of the launch-path kernels in
// ar_kernels.h (one wave per weight row, lanes stride K in 16-byte pieces, the same fma chain, DPP wave
// reduction, rounding points and epilogues), so a frame from the engine is bit-identical to a frame from the
// launches (tests/test_engine_gpu.py) and a lock-step batch row still reproduces its single run.
//   reference: fish_tts/models/llama.py:400-453 (slow pass), 561-580 (fast pass), 193-331 (block),
//   fish_tts/models/inference.py:83-155 (one frame), 24-80 (sampling).
//
// Hand-off rules followed (MI355X guide, "Inter-workgroup communication"): a granule is ONE aligned 4- or 8-byte
// sc1 store; every load of handed-off bytes is an sc1 vector load to registers (never scalar, never plain);
// tags are derived from an epoch word in device memory that the last workgroup to leave advances, never from a
// kernel argument (frozen under graph replay); every spin is bounded by s_memrealtime and raises a global
// abort word that all spinners poll.
#pragma once
#include "ar_kernels.h"

namespace ft {

constexpr int ENG_NB = 256;              // workgroups of an engine launch = CUs of the chip (8 XCDs x 32)
constexpr int ENG_WAVES = 8;             // waves per workgroup (one workgroup per CU; 256 VGPRs per wave)
constexpr int ENG_THREADS = ENG_WAVES * 64;
constexpr int ENG_CW = 4;                // waves 0..3 own weight rows
constexpr int ENG_GW = 4;                // waves 4..7 gather the input vector of a phase

// The widths are compile-time constants of the kernels (section "scalar hygiene" of DESIGN.md: run-time widths cost ~300
// scalar-register spills and the divisions on the hand-off chains); the engine is INSTANTIATED for a short list of shape
// classes (engine.hip: eng_shapes) and the host's gate picks the one a model's config.json matches - any depth.
// Slow stack: dim, query heads, kv heads, head_dim, intermediate_size.  Derived: 16-byte weight pieces per lane and row
// (NT*: 512 contraction elements each), rows / (w1, w3) pairs per compute wave at 256 workgroups (S*), and whether the
// SwiGLU vector travels three values per 8-byte granule (PK3: whole triples per workgroup, whole 1 KiB pieces).
template <int D_, int H_, int HKV_, int HDIM_, int F_>
struct EngSlowShape {
    static constexpr int D = D_, H = H_, HKV = HKV_, HDIM = HDIM_, F = F_;
    static constexpr int HD = H * HDIM, QKVN = (H + 2 * HKV) * HDIM, G = H / HKV;
    static constexpr int NTD = D / 512, NTA = HD / 512, NTF = F / 512;
    static constexpr int SQ = (QKVN / ENG_NB + ENG_CW - 1) / ENG_CW, SF = (F / ENG_NB + ENG_CW - 1) / ENG_CW, SO = (D / ENG_NB + ENG_CW - 1) / ENG_CW;
    static constexpr bool PK3 = F % 384 == 0 && (F / ENG_NB) % 3 == 0;
    static_assert(D % 512 == 0 && HD % 512 == 0 && F % 512 == 0, "weight rows are read as whole 16-byte pieces by 64 lanes");
    static_assert(QKVN % (4 * ENG_NB) == 0 && D % (4 * ENG_NB) == 0 && F % (4 * ENG_NB) == 0, "whole 16-byte granule groups per workgroup");
    static_assert(H % HKV == 0 && HDIM % 8 == 0 && HDIM <= 128, "grouped-query attention, 16-byte K/V pieces");
};
// Fast stack: fast_dim, heads, kv heads, head_dim, intermediate_size, codes drawn per codebook
template <int D_, int H_, int HKV_, int HDIM_, int F_, int V_>
struct EngFastShape {
    static constexpr int D = D_, H = H_, HKV = HKV_, HDIM = HDIM_, F = F_, V = V_;
    static constexpr int HD = H * HDIM, QKVN = (H + 2 * HKV) * HDIM, KVW = HKV * HDIM;
    static constexpr int NTD = D / 512, NTA = HD / 512, NTF = F / 512;
    static constexpr int SQ = (QKVN / ENG_NB + ENG_CW - 1) / ENG_CW, SF = (F / ENG_NB + ENG_CW - 1) / ENG_CW, SO = (D / ENG_NB + ENG_CW - 1) / ENG_CW;
    static constexpr bool PK3 = F % 384 == 0 && (F / ENG_NB) % 3 == 0;
    static_assert(D % 512 == 0 && HD % 512 == 0 && F % 512 == 0 && V == 1024 && HD == D, "fast widths");
    static_assert(QKVN % (4 * ENG_NB) == 0 && D % (4 * ENG_NB) == 0 && F % (4 * ENG_NB) == 0 && V % (4 * ENG_NB) == 0, "whole granule groups per workgroup");
};
typedef EngSlowShape<1024, 16, 8, 128, 3072> EngSlowS1;          // openaudio-s1-mini (the BASELINE shapes)
typedef EngFastShape<1024, 16, 8, 64, 3072, 1024> EngFastS1;
// slow stack: the hand-off vectors of layer li live in the buffers of parity li & 1 (tag = epoch + li), so the lines that
// are polled were written two layers ago and are still in the L2 / memory-side cache instead of cold in HBM
constexpr int ENG_EPOCH_STEP = 64;       // tags used per launch (>= num_codebooks, >= slow layers)
constexpr unsigned long long ENG_TIMEOUT_TICKS = 20000000ull;   // 200 ms of s_memrealtime (100 MHz)

typedef __attribute__((address_space(1))) unsigned eng_gu32;
typedef __attribute__((address_space(1))) unsigned long long eng_gu64;
#define ENG_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// control words (device memory, zeroed at creation)
enum { ENG_CTL_EPOCH = 0, ENG_CTL_ABORT = 1, ENG_CTL_EXIT = 2, ENG_CTL_WHERE = 3, ENG_CTL_ARRIVED = 4, ENG_CTL_FAULT = 5, ENG_CTL_FAULT_SKIP = 6,
       ENG_CTL_XCD = 16, ENG_CTL_WORDS = 32 };
// ENG_CTL_FAULT (test hook, ft_test_engine_fault): 1 + b = workgroup b of the next slow-stack launch, ENG_FAULT_FAST + 1 + b =
// of the next codebook-loop launch, starts with its `dead` word set and clears the fault word: it publishes nothing.
// ENG_CTL_FAULT_SKIP launches of that kind pass first (only the named workgroup reads and counts the word down).
constexpr unsigned ENG_FAULT_FAST = 0x10000u;
// true when this workgroup is the one told to play dead (thread 0 only; the word is cleared)
__device__ __forceinline__ bool eng_fault_here(unsigned* ctl, unsigned base, int b) {
    const unsigned f = __hip_atomic_load((__attribute__((address_space(1))) unsigned*)(ctl + ENG_CTL_FAULT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f != base + 1u + (unsigned)b) return false;
    const unsigned skip = __hip_atomic_load((__attribute__((address_space(1))) unsigned*)(ctl + ENG_CTL_FAULT_SKIP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (skip > 0u) {
        __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(ctl + ENG_CTL_FAULT_SKIP), skip - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(ctl + ENG_CTL_FAULT), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

struct EngLayer {   // device-resident table, one entry per transformer block
    const bf16_t *wqkv, *bqkv, *attn_norm, *qn, *kn, *wo, *bo, *ffn_norm, *w13, *w2;
    bf16_t *kc, *vc;   // KV cache of the slot this launch serves (slow stack only)
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would wait for every
// prefetched weight row at every phase.
__device__ __forceinline__ void eng_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The per-layer pointer table is read through the constant address space: scalar loads (s_load), which do not sit
// in the vector-memory counter the prefetched weight rows are counted in (a global_load of a table entry would
// force vmcnt(0) at every phase).  The table is written once by the host before any launch.
// Pointers that come out of the table are generic; loads through them must be global_load (counted in vmcnt, in
// order), not flat_load (vmcnt AND lgkmcnt, out of order: the compiler then waits for everything at every use).
typedef const __attribute__((address_space(1))) U4* eng_gU4;
// (a run-time choice between the two forms collapses into ONE plain load: the hint has to be a compile-time one)
template <bool NT> __device__ __forceinline__ U4 eng_ldg16(const void* p) {
    if constexpr (NT) return __builtin_nontemporal_load((eng_gU4)p);
    else return *(eng_gU4)p;
}
__device__ __forceinline__ float eng_ldg_bf16(const bf16_t* p, size_t i) {
    return bf16_bits_to_f32(((const __attribute__((address_space(1))) bf16_t*)p)[i]);
}
typedef const __attribute__((address_space(4))) unsigned long long* eng_c64;
__device__ __forceinline__ EngLayer eng_layer(const EngLayer* table, int i) {
    static_assert(sizeof(EngLayer) == 12 * sizeof(unsigned long long), "EngLayer is twelve pointers");
    eng_c64 q = (eng_c64)(unsigned long long)(table + i);
    EngLayer l;
    l.wqkv = (const bf16_t*)q[0]; l.bqkv = (const bf16_t*)q[1]; l.attn_norm = (const bf16_t*)q[2]; l.qn = (const bf16_t*)q[3];
    l.kn = (const bf16_t*)q[4]; l.wo = (const bf16_t*)q[5]; l.bo = (const bf16_t*)q[6]; l.ffn_norm = (const bf16_t*)q[7];
    l.w13 = (const bf16_t*)q[8]; l.w2 = (const bf16_t*)q[9]; l.kc = (bf16_t*)q[10]; l.vc = (bf16_t*)q[11];
    return l;
}

__device__ __forceinline__ unsigned eng_tag16(unsigned e) { return (e & 0x7fffu) | 0x8000u; }
__device__ __forceinline__ unsigned eng_tag32(unsigned e) { return e | 0x80000000u; }
__device__ __forceinline__ unsigned long long eng_rt() { return __builtin_amdgcn_s_memrealtime(); }

// publish one bf16-representable value as a granule (an sc1 store; atomic exchange and sc0 sc1 stores measured no faster)
__device__ __forceinline__ void eng_put(unsigned* g, int i, float v, unsigned tag16) {
    __hip_atomic_store((eng_gu32*)(g + i), (tag16 << 16) | (__float_as_uint(v) >> 16), ENG_RLX);
}
__device__ __forceinline__ void eng_put_raw(unsigned* g, int i, unsigned v16, unsigned tag16) {
    __hip_atomic_store((eng_gu32*)(g + i), (tag16 << 16) | (v16 & 0xffffu), ENG_RLX);
}
// four consecutive granules in one 16-byte store (every dword carries its own tag, so a torn store is harmless)
__device__ __forceinline__ void eng_put4(unsigned* g, float v0, float v1, float v2, float v3, unsigned tag16) {
    U4 w;
    w.x = (tag16 << 16) | (__float_as_uint(v0) >> 16); w.y = (tag16 << 16) | (__float_as_uint(v1) >> 16);
    w.z = (tag16 << 16) | (__float_as_uint(v2) >> 16); w.w = (tag16 << 16) | (__float_as_uint(v3) >> 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(g), "v"(w) : "memory");
}
// publish one f32 as an 8-byte granule {value, tag}
__device__ __forceinline__ void eng_put64(unsigned long long* g, size_t i, float v, unsigned tag32) {
    __hip_atomic_store((eng_gu64*)(g + i), ((unsigned long long)tag32 << 32) | __float_as_uint(v), ENG_RLX);
}

// polls of handed-off bytes: 16-byte sc1 loads to registers, waited for at once (sc0 sc1 and nt polls measured slower
// on buffers other XCDs write; on an XCD's replica and on XCD-local buffers nt measured the same as sc1)
template <int FL = 0>
__device__ __forceinline__ void eng_ld3_sc1(const void* p0, const void* p1, const void* p2, U4& a, U4& b, U4& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
template <int FL = 0>
__device__ __forceinline__ void eng_ld1_sc1(const void* p0, U4& a) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p0) : "memory");
}
// (measured and removed: two polls in flight per importer wave, the second issued half a round trip after the first -
// 597 -> 618 us for the slow stack, 721 -> 746 us for the codebook loop: every variant that polls more is slower)

__device__ __forceinline__ bool eng_tags_ok(const U4& v, unsigned tag) {
    return (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
}
__device__ __forceinline__ void eng_unpack_to_lds(float* dst, const U4& v) {
    float4 f;
    f.x = __uint_as_float(v.x << 16); f.y = __uint_as_float(v.y << 16);
    f.z = __uint_as_float(v.z << 16); f.w = __uint_as_float(v.w << 16);
    *reinterpret_cast<float4*>(dst) = f;
}

// Spin bookkeeping shared by every poll loop: the clock and the abort word are looked at every 256 polls only.
struct EngSpin {
    unsigned* ctl;
    int* dead;          // LDS word: this workgroup gives up
    unsigned spins = 0;
    unsigned long long t0 = 0;
    int where;
    __device__ __forceinline__ bool give_up(int lane) {
        if ((++spins & 255u) != 0u) return false;
        const unsigned long long t = eng_rt();
        if (t0 == 0) t0 = t;
        if (t - t0 > ENG_TIMEOUT_TICKS || __hip_atomic_load((eng_gu32*)(ctl + ENG_CTL_ABORT), ENG_RLX)) {
            if (lane == 0) {
                if (!__hip_atomic_load((eng_gu32*)(ctl + ENG_CTL_ABORT), ENG_RLX))
                    __hip_atomic_store((eng_gu32*)(ctl + ENG_CTL_WHERE), (unsigned)where, ENG_RLX);
                __hip_atomic_store((eng_gu32*)(ctl + ENG_CTL_ABORT), 1u, ENG_RLX);
                *dead = 1;
            }
            return true;
        }
        return false;
    }
};

// Layout of a vector in its hand-off buffer.  ENG_LINE: every producing workgroup owns one 128-byte line and writes its
// `per` consecutive units there with ONE store instruction, so a line is written once per phase (the memory side serves
// the accesses to one line one after the other, and every write of a polled line also recalls its copies from the
// XCDs' L2s: 32 four-byte writes per line cost 2.5-3 us per hand-off, 8 cost 1.6-2, one costs about 1).
// (measured and not kept: every producer's outputs padded to its own 128-byte line, -5 %: vectors are stored linearly)
constexpr int ENG_LINE = 32;   // dwords
// where workgroup b publishes its units [u_lo, ..) of a vector
__device__ __forceinline__ size_t eng_pub(int b, int u_lo) { (void)b; return (size_t)u_lo; }
struct EngLayout {
    int per;        // units per producing workgroup (0: the vector is stored linearly)
    __device__ __forceinline__ int off(int u) const { return u; }
};
struct EngIdent { __device__ __forceinline__ int operator()(int i) const { return i; } };

// Gather units [u0, u0 + n) (n % 4 == 0, u0 % 4 == 0, per % 4 == 0) of the vector at g into LDS: unit u0 + i lands at
// dst[dmap(i)] (dmap is given the first of 4 consecutive units and must keep them consecutive).  Called by ngw waves
// (gw = 0..ngw-1); pieces of 256 units are dealt round-robin to the waves, at most three per wave and pass.
template <typename DMap = EngIdent, int FL = 0>
__device__ __forceinline__ void eng_gather(const unsigned* g, EngLayout lay, int u0, int n, unsigned tag, float* dst, int gw, int ngw,
                                           int lane, unsigned* ctl, int* dead, int where, DMap dmap = DMap(),
                                           unsigned long long* dbg = nullptr) {
    const int npiece = (n + 255) >> 8;
    EngSpin sp{ctl, dead, 0, 0, where};
    unsigned long long n_full = 0, t_first = 0;
    for (int p0 = gw; p0 < npiece; p0 += 3 * ngw) {
        const int c0 = p0;
        const int c1 = p0 + ngw < npiece ? p0 + ngw : c0;
        const int c2 = p0 + 2 * ngw < npiece ? p0 + 2 * ngw : c0;
        // lanes beyond the end of a short last piece re-read the piece's first units (always in range)
        const int i0 = c0 * 256 + lane * 4 < n ? c0 * 256 + lane * 4 : c0 * 256;
        const int i1 = c1 * 256 + lane * 4 < n ? c1 * 256 + lane * 4 : c1 * 256;
        const int i2 = c2 * 256 + lane * 4 < n ? c2 * 256 + lane * 4 : c2 * 256;
        const int o0 = lay.off(u0 + i0), o1 = lay.off(u0 + i1), o2 = lay.off(u0 + i2);
        U4 a, b, c;
        for (;;) {
            if (c1 == c0) { eng_ld1_sc1<FL>(g + o0, a); b = a; c = a; }
            else eng_ld3_sc1<FL>(g + o0, g + o1, g + o2, a, b, c);
            const bool oa = eng_tags_ok(a, tag), ob = eng_tags_ok(b, tag), oc = eng_tags_ok(c, tag);
            if (dbg && n_full == 0) t_first = eng_rt();
            ++n_full;
            if (__all(oa && ob && oc)) break;
            if (sp.give_up(lane)) return;
        }
        if (c0 * 256 + lane * 4 < n) eng_unpack_to_lds(dst + dmap(i0), a);
        if (c1 != c0 && c1 * 256 + lane * 4 < n) eng_unpack_to_lds(dst + dmap(i1), b);
        if (c2 != c0 && c2 * 256 + lane * 4 < n) eng_unpack_to_lds(dst + dmap(i2), c);
    }
    if (dbg && lane == 0) { dbg[0] = t_first; dbg[1] = n_full; dbg[2] = 0; }
}

// ------------------------------------------------------------------------------------------
// XCD relay.  Every CU needs every vector, but 256 CUs polling the same memory lines is what makes a hand-off cost
// 2.4-3 us (each poll and each producer store queues behind ~1000 other requests for those lines).  So only ONE
// workgroup per XCD and 1 KiB piece polls memory (8 pollers per line instead of 256); it copies the piece into its XCD's
// replica of the buffer with PLAIN stores, which stay in that XCD's L2, and the 32 CUs of the XCD poll the replica
// (sc1 loads: past their L1, served by the L2 they share with the importer).  The XCD of a workgroup is read from the
// hardware (XCC_ID), never assumed from blockIdx; replicas carry the same self-validating {tag, value} granules.
// ------------------------------------------------------------------------------------------
struct EngRelay {
    int on;          // 0: every workgroup polls the source buffer itself
    int rank, nr;    // this workgroup's rank among the nr workgroups of its XCD
    long delta;      // words from a source address to the same vector in this XCD's replica
};

// Called by thread 0 of every workgroup at kernel entry (results go to LDS r[0..2]); all workgroups of a launch are
// co-resident, so waiting for all of them to have registered is safe (and bounded).
__device__ __forceinline__ void eng_register(unsigned* ctl, int nb, int* r, int* dead) {
    const int xcd = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u);    // HW_REG_XCC_ID[3:0]
    const unsigned rank = atomicAdd(ctl + ENG_CTL_XCD + xcd, 1u);
    atomicAdd(ctl + ENG_CTL_ARRIVED, 1u);
    EngSpin sp{ctl, dead, 0, 0, 9999};
    bool ok = true;
    while (__hip_atomic_load((eng_gu32*)(ctl + ENG_CTL_ARRIVED), ENG_RLX) < (unsigned)nb) {
        if (sp.give_up(0)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    r[0] = xcd; r[1] = (int)rank;
    r[2] = ok ? (int)__hip_atomic_load((eng_gu32*)(ctl + ENG_CTL_XCD + xcd), ENG_RLX) : 1;
}

template <typename DMap = EngIdent>
__device__ __forceinline__ void eng_gather_x(const EngRelay& rl, const unsigned* g, EngLayout lay, int u0, int n, unsigned tag, float* dst,
                                             int gw, int ngw, int lane, unsigned* ctl, int* dead, int where, DMap dmap = DMap(),
                                             unsigned long long* dbg = nullptr) {
    if (!rl.on) { eng_gather(g, lay, u0, n, tag, dst, gw, ngw, lane, ctl, dead, where, dmap, dbg); return; }
    unsigned* rep = const_cast<unsigned*>(g) + rl.delta;
    // import duty: piece p belongs to the workgroup of rank p % nr of every XCD, and there to wave p % ngw
    const int npiece = (n + 255) >> 8;
    for (int p = rl.rank; p < npiece; p += rl.nr) {
        if (p % ngw != gw) continue;
        const int i = p * 256 + lane * 4 < n ? p * 256 + lane * 4 : p * 256;
        const int o = lay.off(u0 + i);
        EngSpin sp{ctl, dead, 0, 0, where};
        U4 a;
        for (;;) {
            eng_ld1_sc1(g + o, a);
            if (__all(eng_tags_ok(a, tag))) break;
            if (sp.give_up(lane)) return;
        }
        if (p * 256 + lane * 4 < n) *reinterpret_cast<U4*>(rep + o) = a;      // plain store: stays in this XCD's L2
    }
    eng_gather<DMap, 1>(rep, lay, u0, n, tag, dst, gw, ngw, lane, ctl, dead, where, dmap, dbg);
}

// The packed form (three values per 8-byte granule, eng_gemv<.., PK3>): n values (n % 384 == 0) = n / 384 pieces of 1 KiB;
// lane l of a piece holds two granules = values 6 l .. 6 l + 5 of the piece.
__device__ __forceinline__ bool eng_tags3_ok(const U4& v, unsigned tag) { return (v.y >> 16) == tag && (v.w >> 16) == tag; }
__device__ __forceinline__ void eng_unpack3_to_lds(float* dst, const U4& v) {
    float2 a, b, c;
    a.x = __uint_as_float(v.x << 16); a.y = __uint_as_float(v.x & 0xffff0000u);
    b.x = __uint_as_float(v.y << 16); b.y = __uint_as_float(v.z << 16);
    c.x = __uint_as_float(v.z & 0xffff0000u); c.y = __uint_as_float(v.w << 16);
    reinterpret_cast<float2*>(dst)[0] = a; reinterpret_cast<float2*>(dst)[1] = b; reinterpret_cast<float2*>(dst)[2] = c;
}
template <int FL = 0>
__device__ __forceinline__ void eng_gather3(const unsigned* g, int n, unsigned tag, float* dst, int gw, int ngw, int lane, unsigned* ctl,
                                            int* dead, int where) {
    const int npiece = n / 384;
    EngSpin sp{ctl, dead, 0, 0, where};
    for (int p0 = gw; p0 < npiece; p0 += 3 * ngw) {
        const int c0 = p0;
        const int c1 = p0 + ngw < npiece ? p0 + ngw : c0;
        const int c2 = p0 + 2 * ngw < npiece ? p0 + 2 * ngw : c0;
        const unsigned* a0 = g + c0 * 256 + lane * 4;
        const unsigned* a1 = g + c1 * 256 + lane * 4;
        const unsigned* a2 = g + c2 * 256 + lane * 4;
        U4 a, b, c;
        for (;;) {
            if (c1 == c0) { eng_ld1_sc1<FL>(a0, a); b = a; c = a; }
            else eng_ld3_sc1<FL>(a0, a1, a2, a, b, c);
            if (__all(eng_tags3_ok(a, tag) && eng_tags3_ok(b, tag) && eng_tags3_ok(c, tag))) break;
            if (sp.give_up(lane)) return;
        }
        eng_unpack3_to_lds(dst + c0 * 384 + lane * 6, a);
        if (c1 != c0) eng_unpack3_to_lds(dst + c1 * 384 + lane * 6, b);
        if (c2 != c0) eng_unpack3_to_lds(dst + c2 * 384 + lane * 6, c);
    }
}
__device__ __forceinline__ void eng_gather3_x(const EngRelay& rl, const unsigned* g, int n, unsigned tag, float* dst, int gw, int ngw,
                                              int lane, unsigned* ctl, int* dead, int where) {
    if (!rl.on) { eng_gather3(g, n, tag, dst, gw, ngw, lane, ctl, dead, where); return; }
    unsigned* rep = const_cast<unsigned*>(g) + rl.delta;
    const int npiece = n / 384;
    for (int p = rl.rank; p < npiece; p += rl.nr) {      // import duty, as in eng_gather_x
        if (p % ngw != gw) continue;
        const int o = p * 256 + lane * 4;
        EngSpin sp{ctl, dead, 0, 0, where};
        U4 a;
        for (;;) {
            eng_ld1_sc1(g + o, a);
            if (__all(eng_tags3_ok(a, tag))) break;
            if (sp.give_up(lane)) return;
        }
        *reinterpret_cast<U4*>(rep + o) = a;
    }
    eng_gather3<1>(rep, n, tag, dst, gw, ngw, lane, ctl, dead, where);
}

// Two vectors of the same length n (n % 256 == 0) gathered as ONE list of 2 * npiece pieces (the two rows of the fast
// loop's first pass): piece P < npiece belongs to vector 0, the others to vector 1.
struct EngSrc2 { const unsigned* g[2]; float* dst[2]; unsigned tag[2]; };
template <int FL = 0>
__device__ __forceinline__ void eng_gather2(const EngSrc2& S, long delta, int n, int gw, int ngw, int lane, unsigned* ctl, int* dead,
                                            int where) {
    const int npiece = n >> 8, NP = 2 * npiece;
    EngSpin sp{ctl, dead, 0, 0, where};
    for (int p0 = gw; p0 < NP; p0 += 3 * ngw) {
        const int c0 = p0;
        const int c1 = p0 + ngw < NP ? p0 + ngw : c0;
        const int c2 = p0 + 2 * ngw < NP ? p0 + 2 * ngw : c0;
        const int s0 = c0 >= npiece, s1 = c1 >= npiece, s2 = c2 >= npiece;
        const int i0 = (c0 - s0 * npiece) * 256 + lane * 4, i1 = (c1 - s1 * npiece) * 256 + lane * 4, i2 = (c2 - s2 * npiece) * 256 + lane * 4;
        const unsigned* a0 = S.g[s0] + delta + i0;
        const unsigned* a1 = S.g[s1] + delta + i1;
        const unsigned* a2 = S.g[s2] + delta + i2;
        const unsigned t0 = S.tag[s0], t1 = S.tag[s1], t2 = S.tag[s2];
        U4 a, b, c;
        for (;;) {
            if (c1 == c0) { eng_ld1_sc1<FL>(a0, a); b = a; c = a; }
            else eng_ld3_sc1<FL>(a0, a1, a2, a, b, c);
            const bool oa = eng_tags_ok(a, t0), ob = c1 == c0 || eng_tags_ok(b, t1), oc = c2 == c0 || eng_tags_ok(c, t2);
            if (__all(oa && ob && oc)) break;
            if (sp.give_up(lane)) return;
        }
        eng_unpack_to_lds(S.dst[s0] + i0, a);
        if (c1 != c0) eng_unpack_to_lds(S.dst[s1] + i1, b);
        if (c2 != c0) eng_unpack_to_lds(S.dst[s2] + i2, c);
    }
}
__device__ __forceinline__ void eng_gather_x2(const EngRelay& rl, const EngSrc2& S, int n, int gw, int ngw, int lane, unsigned* ctl,
                                              int* dead, int where) {
    if (!rl.on) { eng_gather2(S, 0, n, gw, ngw, lane, ctl, dead, where); return; }
    const int npiece = n >> 8, NP = 2 * npiece;
    for (int p = rl.rank; p < NP; p += rl.nr) {
        if (p % ngw != gw) continue;
        const int s = p >= npiece;
        const int i = (p - s * npiece) * 256 + lane * 4;
        const unsigned* src = S.g[s] + i;
        const unsigned tg = S.tag[s];
        EngSpin sp{ctl, dead, 0, 0, where};
        U4 a;
        for (;;) {
            eng_ld1_sc1(src, a);
            if (__all(eng_tags_ok(a, tg))) break;
            if (sp.give_up(lane)) return;
        }
        *reinterpret_cast<U4*>(const_cast<unsigned*>(src) + rl.delta) = a;
    }
    eng_gather2<1>(S, rl.delta, n, gw, ngw, lane, ctl, dead, where);
}

// ------------------------------------------------------------------------------------------
// One matrix-vector phase on this workgroup's units [u_lo, u_hi) of a weight matrix [N][K] (a unit is RPU
// consecutive rows: 1, or 2 for the interleaved (w1_i, w3_i) pairs).  Unit u_lo + cw + s * ENG_CW belongs to
// compute wave cw (s < MAXS).  eng_issue requests the rows (and the norm gains) into registers, eng_gemv
// consumes them with the arithmetic of gemv_finish (ar_kernels.h).
// ------------------------------------------------------------------------------------------
template <int NT, int RPU, int MAXS>
struct EngW {
    U4 w[MAXS][RPU][NT];
    U4 gain[NT];
};

// which row of the matrix a unit is: the identity, or (slow stack, XCD-local attention) the rows of ONE kv head's q, k and v
// that the workgroups of one XCD share out - workgroup r of the XCD takes NQ query rows, NK key rows and NK value rows
struct EngRowIdent { __device__ __forceinline__ int operator()(int u) const { return u; } };
struct EngRowQkvX {
    int q0, k0, v0, nq, nk;      // first q / k / v row of this workgroup; q rows and k (= v) rows per workgroup
    __device__ __forceinline__ int operator()(int u) const { return u < nq ? q0 + u : (u < nq + nk ? k0 + (u - nq) : v0 + (u - nq - nk)); }
};

template <bool NTL = true, int NT, int RPU, int MAXS, typename RM = EngRowIdent>
__device__ __forceinline__ void eng_issue(EngW<NT, RPU, MAXS>& r, const bf16_t* W, const bf16_t* gain, int K, int u_lo, int u_hi,
                                          int cw, int lane, int nt, RM rm = RM()) {
    // every compute wave issues the same number of loads on every path (units past the workgroup's last one re-read
    // it), so the counted vmcnt waits the compiler derives stay exact
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        int u = u_lo + cw + s * ENG_CW;
        u = u < u_hi ? u : u_hi - 1;
#pragma unroll
        for (int rr = 0; rr < RPU; ++rr)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                r.w[s][rr][t] = (nt & 2) ? U4{0u, 0u, 0u, 0u} : eng_ldg16<NTL>(W + (size_t)(rm(u) * RPU + rr) * K + t * 512 + lane * 8);
            }
    }
    if (gain) {
#pragma unroll
        for (int t = 0; t < NT; ++t) r.gain[t] = eng_ldg16<false>(gain + t * 512 + lane * 8);
    }
}

// Collection point of a workgroup's outputs of one phase: the compute waves leave their values in LDS and count in; the
// wave that arrives last publishes ALL of the workgroup's outputs with ONE store instruction (consecutive lanes,
// consecutive granules).  One coalesced write per workgroup instead of one partial write per row: every 128-byte line
// of the vector is written by 2-8 stores instead of 32, and the memory side serves the writes of a line one after
// the other (measured: acknowledgements of the 4-byte stores of a phase spread over 2.5 us).
struct EngOut {
    float* vals;     // LDS [ENG_MAX_OUT]
    int* count;      // LDS arrival counter (never reset: phase k is complete at (k + 1) * ENG_CW)
    int seq;         // phases done so far (per-wave copy, all compute waves agree)
};
constexpr int ENG_MAX_OUT = 64;
typedef __attribute__((address_space(3))) int* eng_lds_int;

// xs: the input vector in LDS (f32).  resid: LDS vector the residual epilogue adds (indexed by row), or nullptr.
// gout: this workgroup's line of the output vector's hand-off buffer, plain: optional plain f32 copy of the whole
// vector (last phase of a launch).
// Called by EVERY compute wave (also one without rows in this matrix).
// the arithmetic of one input row: this wave's outputs go to vals[u - u_lo] (LDS)
template <int NT, int RPU, int MAXS, int PRO, int EPI, typename RM = EngRowIdent>
__device__ __forceinline__ void eng_gemv_rows(const EngW<NT, RPU, MAXS>& r, const float* xs, int K, float eps, const bf16_t* bias,
                                              const float* resid, float* vals, int u_lo, int u_hi, int cw, int lane, RM rm = RM()) {
    static_assert(EPI == EPI_SWIGLU ? RPU == 2 : RPU == 1, "a unit is a (w1, w3) pair for SwiGLU, one row otherwise");
    if (u_lo + cw < u_hi) {
        float xv[NT][8];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 f0 = *reinterpret_cast<const float4*>(xs + t * 512 + lane * 8);
            const float4 f1 = *reinterpret_cast<const float4*>(xs + t * 512 + lane * 8 + 4);
            xv[t][0] = f0.x; xv[t][1] = f0.y; xv[t][2] = f0.z; xv[t][3] = f0.w;
            xv[t][4] = f1.x; xv[t][5] = f1.y; xv[t][6] = f1.z; xv[t][7] = f1.w;
        }
        if (PRO == PRO_RMSNORM) {
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) ss = fmaf(xv[t][j], xv[t][j], ss);
            ss = wave_sum(ss);
            const float inv = rsqrt_exact(ss / (float)K + eps);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float gv[8];
                Vec<bf16_t>::unpack(r.gain[t], gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[t][j] = round_bf16(round_bf16(xv[t][j] * inv) * gv[j]);
            }
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int u = u_lo + cw + s * ENG_CW;
            if (u >= u_hi) break;
            float acc[RPU];
#pragma unroll
            for (int rr = 0; rr < RPU; ++rr) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float wv[8];
                    Vec<bf16_t>::unpack(r.w[s][rr][t], wv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a = fmaf(wv[j], xv[t][j], a);
                }
                acc[rr] = wave_sum(a);
            }
            float o;
            if (EPI == EPI_SWIGLU) {
                const float a = round_bf16(acc[0]);
                const float b = round_bf16(acc[RPU - 1]);
                const float sg = round_bf16(a / (1.0f + expf(-a)));
                o = round_bf16(sg * b);
            } else {
                float v = acc[0];
                if (bias) v += eng_ldg_bf16(bias, rm(u));
                v = round_bf16(v);
                if (EPI == EPI_RESID) v = round_bf16(resid[rm(u)] + v);
                o = v;
            }
            if (lane == 0) vals[u - u_lo] = o;
        }
    }
}

// Granules that stay inside ONE XCD (slow stack, XCD-local attention): written with PLAIN stores - they stay in the XCD's L2 -
// and polled by workgroups of the same XCD with sc1 loads (past their L1, served by that L2): the mechanism of the relay's
// replicas.  Who is in which XCD is read from the hardware at run time (eng_register), never assumed.
__device__ __forceinline__ void eng_put_local(unsigned* g, int i, float v, unsigned tag16) {
    __hip_atomic_store((eng_gu32*)(g + i), (tag16 << 16) | (__float_as_uint(v) >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void eng_put64_local(unsigned long long* g, size_t i, float v, unsigned tag32) {
    __hip_atomic_store((eng_gu64*)(g + i), ((unsigned long long)tag32 << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// QKV phase of the XCD-local form: this workgroup's units are rows rm(0 .. n-1) of the matrix; unit i's granule goes to
// gvec[rm(i)] (the vector keeps its row order) with a plain store
template <int NT, int MAXS, typename RM>
__device__ __forceinline__ void eng_gemv_qkv_local(const EngW<NT, 1, MAXS>& r, const float* xs, int K, float eps, const bf16_t* bias,
                                                   unsigned* gvec, unsigned tag, int n, int cw, int lane, EngOut& eo, RM rm) {
    eng_gemv_rows<NT, 1, MAXS, PRO_RMSNORM, EPI_STORE>(r, xs, K, eps, bias, nullptr, eo.vals, 0, n, cw, lane, rm);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((eng_lds_int)eo.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    eo.seq += 1;
    if (old + 1 == eo.seq * ENG_CW) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        if (lane < n) eng_put_local(gvec, rm(lane), eo.vals[lane], tag);
    }
}

// PK3: the vector is handed over as 8-byte granules of THREE values {v0 | v1 << 16, v2 | tag << 16} (the SwiGLU vector: 3072
// values are 8 KB instead of 12, and its hand-off is the longest of a layer); gout then points at the vector's base and the
// workgroup's units must be a multiple of three (12 at 256 workgroups).
template <int NT, int RPU, int MAXS, int PRO, int EPI, bool PK3 = false>
__device__ __forceinline__ void eng_gemv(const EngW<NT, RPU, MAXS>& r, const float* xs, int K, float eps, const bf16_t* bias,
                                         const float* resid, unsigned* gout, unsigned tag, float* plain, int u_lo, int u_hi,
                                         int cw, int lane, EngOut& eo) {
    eng_gemv_rows<NT, RPU, MAXS, PRO, EPI>(r, xs, K, eps, bias, resid, eo.vals, u_lo, u_hi, cw, lane);
    // count in; the last wave of the workgroup to arrive publishes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((eng_lds_int)eo.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    eo.seq += 1;
    if (old + 1 == eo.seq * ENG_CW) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const int n = u_hi - u_lo;
        if constexpr (PK3) {
            if (lane * 3 < n) {
                const unsigned w0 = (__float_as_uint(eo.vals[3 * lane]) >> 16) | (__float_as_uint(eo.vals[3 * lane + 1]) & 0xffff0000u);
                const unsigned w1 = (__float_as_uint(eo.vals[3 * lane + 2]) >> 16) | (tag << 16);
                __hip_atomic_store((eng_gu64*)(reinterpret_cast<unsigned long long*>(gout) + (u_lo / 3 + lane)),
                                   ((unsigned long long)w1 << 32) | w0, ENG_RLX);
            }
        } else if (lane < n) {
            const float o = eo.vals[lane];
            eng_put(gout, lane, o, tag);
            if (plain) plain[u_lo + lane] = o;
        }
    }
}

// eng_gemv_rows for two input rows at once (each weight register unpacked once; per row the operations of eng_gemv_rows)
template <int NT, int RPU, int MAXS, int PRO, int EPI>
__device__ __forceinline__ void eng_gemv_rows2(const EngW<NT, RPU, MAXS>& r, const float* xs0, const float* xs1, int K, float eps,
                                               const bf16_t* bias, const float* resid0, const float* resid1, float* vals0, float* vals1,
                                               int u_lo, int u_hi, int cw, int lane) {
    static_assert(EPI == EPI_SWIGLU ? RPU == 2 : RPU == 1, "a unit is a (w1, w3) pair for SwiGLU, one row otherwise");
    if (u_lo + cw < u_hi) {
        float xv[2][NT][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* xs = q ? xs1 : xs0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 f0 = *reinterpret_cast<const float4*>(xs + t * 512 + lane * 8);
                const float4 f1 = *reinterpret_cast<const float4*>(xs + t * 512 + lane * 8 + 4);
                xv[q][t][0] = f0.x; xv[q][t][1] = f0.y; xv[q][t][2] = f0.z; xv[q][t][3] = f0.w;
                xv[q][t][4] = f1.x; xv[q][t][5] = f1.y; xv[q][t][6] = f1.z; xv[q][t][7] = f1.w;
            }
        }
        if (PRO == PRO_RMSNORM) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float ss = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss = fmaf(xv[q][t][j], xv[q][t][j], ss);
                ss = wave_sum(ss);
                const float inv = rsqrt_exact(ss / (float)K + eps);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float gv[8];
                    Vec<bf16_t>::unpack(r.gain[t], gv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[q][t][j] = round_bf16(round_bf16(xv[q][t][j] * inv) * gv[j]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int u = u_lo + cw + s * ENG_CW;
            if (u >= u_hi) break;
            float acc[2][RPU];
#pragma unroll
            for (int rr = 0; rr < RPU; ++rr) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float wv[8];
                    Vec<bf16_t>::unpack(r.w[s][rr][t], wv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a0 = fmaf(wv[j], xv[0][t][j], a0); a1 = fmaf(wv[j], xv[1][t][j], a1); }
                }
                acc[0][rr] = wave_sum(a0);
                acc[1][rr] = wave_sum(a1);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float o;
                if (EPI == EPI_SWIGLU) {
                    const float a = round_bf16(acc[q][0]);
                    const float bb = round_bf16(acc[q][RPU - 1]);
                    const float sg = round_bf16(a / (1.0f + expf(-a)));
                    o = round_bf16(sg * bb);
                } else {
                    float v = acc[q][0];
                    if (bias) v += eng_ldg_bf16(bias, u);
                    v = round_bf16(v);
                    if (EPI == EPI_RESID) v = round_bf16((q ? resid1 : resid0)[u] + v);
                    o = v;
                }
                if (lane == 0) (q ? vals1 : vals0)[u - u_lo] = o;
            }
        }
    }
}

// Two input rows against the same rows of the matrix (the fast loop's first pass: codebook positions 0 and 1 are both
// known when the launch starts).  Row r's outputs go to gout[r] with tag[r]; still ONE store instruction per workgroup
// (lanes 0.. carry row 0, lanes 32.. row 1; a workgroup owns at most 32 units of a vector).
template <int NT, int RPU, int MAXS, int PRO, int EPI>
__device__ __forceinline__ void eng_gemv2(const EngW<NT, RPU, MAXS>& r, const float* xs0, const float* xs1, int K, float eps,
                                          const bf16_t* bias, const float* resid0, const float* resid1, unsigned* gout0, unsigned* gout1,
                                          unsigned tag0, unsigned tag1, int u_lo, int u_hi, int cw, int lane, EngOut& eo) {
    if (NT <= 2) {
        eng_gemv_rows2<NT, RPU, MAXS, PRO, EPI>(r, xs0, xs1, K, eps, bias, resid0, resid1, eo.vals, eo.vals + 32, u_lo, u_hi, cw, lane);
    } else {
        eng_gemv_rows<NT, RPU, MAXS, PRO, EPI>(r, xs0, K, eps, bias, resid0, eo.vals, u_lo, u_hi, cw, lane);
        __builtin_amdgcn_sched_barrier(0);       // one row's inputs in registers at a time
        eng_gemv_rows<NT, RPU, MAXS, PRO, EPI>(r, xs1, K, eps, bias, resid1, eo.vals + 32, u_lo, u_hi, cw, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((eng_lds_int)eo.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    eo.seq += 1;
    if (old + 1 == eo.seq * ENG_CW) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const int n = u_hi - u_lo;
        const int i = lane & 31;
        if (i < n) {
            const float o = eo.vals[lane];
            eng_put(lane < 32 ? gout0 : gout1, i, o, lane < 32 ? tag0 : tag1);
        }
    }
}

__device__ __forceinline__ void eng_units(int U, int b, int nb, int& lo, int& hi) {
    lo = (int)((long)U * b / nb);
    hi = (int)((long)U * (b + 1) / nb);
}

// ------------------------------------------------------------------------------------------
// Slow stack: embedding + n_layer blocks for ONE utterance row (llama.py:400-453 at S = 1).
// Per layer: [x] -> QKV -> split-KV attention on Hkv*nsplit workgroups -> split merge on the same workgroups
// -> [y] -> Wo + residual -> [x'] -> W13 + SwiGLU -> [g] -> W2 + residual -> [x''].
// ------------------------------------------------------------------------------------------
struct SlowEngP {
    const EngLayer* layers;
    int n_layer;
    int D, H, Hkv, hd, F, qkvN;
    float eps, scale;
    // embedding (llama.py:409-429)
    const bf16_t* emb; const bf16_t* cb_emb; const int* toks; long tok_row_stride;
    int ncb, cbsize, vocab, sem_begin, sem_end, scale_cb; float inv_div;
    // attention
    const float* rope; const int* pos; int pos_off; int n_slots, nsplit;
    size_t cache_off;             // elements from the layer's cache base to this slot's rows
    // granule buffers
    unsigned* gx;                 // [n_layer + 1][D]   layer inputs (entry n_layer = the stack's output)
    unsigned* gqkv;               // [n_layer][qkvN]
    unsigned long long* gpart;    // [n_layer][H][nsplit][hd + 2]  split-KV partials (O, then m, l)
    unsigned* gy;                 // [n_layer][H * hd]
    unsigned* gxb;                // [n_layer][D]
    unsigned* gg;                 // [n_layer][F]
    unsigned* ctl;
    long rep_delta0, rep_stride;  // words from a source buffer to XCD 0's replica, and between replicas (0, 0: no relay)
    float* x_out;                 // plain f32 [D]: input of the vocabulary head launch and of the fast stack
    int nt;                       // bit 1 (timing experiments only): skip the weight loads
    unsigned long long* stamps;   // diagnostic builds only (tools/mb_engine.hip): [workgroup][layer][16] s_memrealtime ticks, or nullptr
};
#define ENG_STAMP(k) do { if (p.stamps && tid == 0) p.stamps[((size_t)b * p.n_layer + li) * 16 + (k)] = eng_rt(); } while (0)
// attention-turn stamps of the gathering waves (second region of the stamp array), written by lane 0 of wave `wv`
#define ENG_ASTAMP(wv, k) do { if (p.stamps && gw == (wv) && lane == 0) \
    p.stamps[((size_t)nb * p.n_layer + (size_t)b * p.n_layer + li) * 16 + (k)] = eng_rt(); } while (0)

constexpr int ENG_KVST = 6;   // K/V steps of an attention workgroup held in registers (16 positions each at hd = 128)

// which (kv head, split) workgroup b takes in layer li, or -1: the Hkv * nsplit attention workgroups rotate with the layer
__device__ __forceinline__ int eng_att_role(int b, int nb, int li, int natt) {
    int a = b - (int)(((unsigned)li * (unsigned)natt) % (unsigned)nb);
    if (a < 0) a += nb;
    return a < natt ? a : -1;
}

// Waves 0..3 (compute waves) keep this workgroup's rows of ALL FOUR matrices of the next layer in registers and
// re-request a matrix the moment its rows have been used, so ~120 KB per CU are always in flight and HBM streams
// continuously instead of in one burst per phase (a CU sustains ~24 GB/s only while requests are outstanding).
// Waves 4..7 gather the input vectors, and run the attention of the layers in which this workgroup holds a
// (kv head, split) role, with the cached K/V rows of their next turn prefetched into registers.
// XL (XCD-local attention): kv head h lives in XCD h.  The XCD's workgroups compute the rows of that head's q, k, v, every
// one of them takes ONE split of the head's cached positions (nsplit = workgroups per XCD), and the q k v vector and the
// split partials travel inside the XCD (plain stores into its L2, sc1 polls) instead of through the memory side; only y
// leaves the XCD.  Needs Hkv XCDs with nb / Hkv workgroups each (checked at run time; a failed census raises the abort
// word and the host falls back to the launch path).  Per row the arithmetic is that of the launch path at the same split
// count (attn_decode_kernel + merge_splits4), so the frames stay bit-identical to it.
template <typename S, bool XL = false>
__global__ __launch_bounds__(ENG_THREADS) void slow_engine_kernel(SlowEngP p) {
    // units per compute wave and matrix at 256 workgroups (s1-mini: QKV 16 rows -> 4, W13 12 pairs -> 3, Wo / W2 4 rows -> 1)
    constexpr int NTD = S::NTD, NTA = S::NTA, NTF = S::NTF, G = S::G;
    constexpr int SQ = S::SQ, SF = S::SF, SO = S::SO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: uniform branches, row offsets in SGPRs)
    const int b = blockIdx.x;
    constexpr int nb = ENG_NB;            // workgroups = CUs (the host launches exactly this many, one per CU)
    const int cw = wave;                  // compute-wave index (valid when < ENG_CW)
    const int gw = wave - ENG_CW;         // gather-wave index (valid when >= 0)
    const int atid = tid - ENG_CW * 64;   // thread index inside the attention group (waves 4..7 = 256 threads)
    // The widths are compile-time constants: the host's shape gate (engine.hip: eng_setup_try) admits exactly these.  With
    // run-time widths the kernel spilled ~300 scalar registers into vector lanes and divided by run-time values on its
    // chains (the y layout map alone: six integer divisions per gathered piece).
    constexpr int D = S::D, F = S::F, hd = S::HDIM, hp = hd >> 1, H = S::H, Hkv = S::HKV, HD = H * hd, QKVN = (H + 2 * Hkv) * hd;
    static_assert(!XL || (ENG_NB % Hkv == 0 && QKVN % ENG_NB == 0), "XCD-local attention: the workgroups of an XCD share one kv head's rows");
    float* xA = smem;                     // layer input
    float* yS = xA + D;                   // attention output
    float* xB = yS + HD;                  // x' = x + Wo y
    float* gS = xB + D;                   // SwiGLU output
    float* qS = gS + F;                   // this workgroup's q heads, new k, new v : [(G + 2) * hd]
    float* q_s = qS + (G + 2) * hd;       // [G][hd] normalised, rotated
    float* k_new = q_s + G * hd;          // [hd]
    float* v_new = k_new + hd;            // [hd]
    constexpr int LPP = hd >> 3, PPW = 64 / LPP, NSLOT = 4 * PPW;
    float* ml_s = v_new + hd;             // [NSLOT][G][2]
    float* acc_s = ml_s + NSLOT * G * 2;  // [NSLOT][G][hd]
    float* mscr = acc_s + NSLOT * G * hd; // [64][6] split-merge exchange
    float* outS = mscr + 64 * 6;          // [ENG_MAX_OUT] this workgroup's outputs of the current phase
    int* dead = reinterpret_cast<int*>(outS + ENG_MAX_OUT);
    int* out_count = dead + 1;
    int* reg_s = dead + 4;                // [3] XCD, rank in it, workgroups in it
    if (tid == 0) { *dead = eng_fault_here(p.ctl, 0u, b) ? 1 : 0; *out_count = 0; }
    if (tid == ENG_CW * 64) {             // one thread owns the registration words
        reg_s[0] = 0; reg_s[1] = 0; reg_s[2] = 1;
        if (XL || p.rep_stride) eng_register(p.ctl, nb, reg_s, dead);
        if (XL && (reg_s[2] * Hkv != nb || reg_s[0] >= Hkv || reg_s[1] >= reg_s[2] || p.nsplit != reg_s[2])) {
            // census failed (the workgroups are not spread Hkv x nb / Hkv over the XCDs): nobody may rely on XCD-local data
            if (!__hip_atomic_load((eng_gu32*)(p.ctl + ENG_CTL_ABORT), ENG_RLX)) __hip_atomic_store((eng_gu32*)(p.ctl + ENG_CTL_WHERE), 9998u, ENG_RLX);
            __hip_atomic_store((eng_gu32*)(p.ctl + ENG_CTL_ABORT), 1u, ENG_RLX);
            reg_s[0] = 0; reg_s[1] = 0;
            *dead = 1;
        }
    }
    const unsigned epoch = __hip_atomic_load((eng_gu32*)(p.ctl + ENG_CTL_EPOCH), ENG_RLX);
    const int pos = p.pos[0] + p.pos_off;
    unsigned long long clk0 = 0, rt0 = 0;
    if (p.stamps) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = eng_rt(); }
    const size_t VSTR = (size_t)nb * ENG_LINE;       // dwords per padded vector buffer (one line per workgroup)
    const EngLayout layD{D / nb}, layF{F / nb}, layQ{QKVN / nb}, layLin{0};
    // (XL: one split per workgroup of the XCD - a compile-time 32; the host passes the same number, checked with the census)
    const int nsplit = XL ? nb / Hkv : p.nsplit;
    const int natt = Hkv * nsplit;
    const int chunk = (pos + nsplit) / nsplit;   // positions per KV split (attn_decode_kernel's rule)
    const int grp = lane / LPP, gl = lane % LPP;

    // ---- embedding of the input column, every workgroup for itself (embed_kernel's arithmetic), all threads; then the
    // two wave roles part ways.  Both roles execute the same sequence of workgroup barriers (B0; per layer: B1 from
    // layer 1 on, BA BB BC in an attention turn, B2 B3 B4) and leave at the same barrier when the workgroup gives up.
    {
        const int* tk = p.toks;
        int t0 = tk[0];
        const bool is_vq = t0 >= p.sem_begin && t0 <= p.sem_end;
        t0 = min(max(t0, 0), p.vocab - 1);
        for (int d = tid; d < D; d += ENG_THREADS) {
            float vq = 0.f;
            if (is_vq) {
                constexpr int MAXCB = 16;
                for (int i0 = 0; i0 < p.ncb; i0 += MAXCB) {
                    int c[MAXCB];
                    float e[MAXCB];
#pragma unroll
                    for (int i = 0; i < MAXCB; ++i) c[i] = i0 + i < p.ncb ? tk[(size_t)(i0 + i + 1) * p.tok_row_stride] : 0;
#pragma unroll
                    for (int i = 0; i < MAXCB; ++i) {
                        const int cc = min(max(c[i], 0), p.cbsize - 1);
                        e[i] = i0 + i < p.ncb ? ld_elem(p.cb_emb, (size_t)(cc + (i0 + i) * p.cbsize) * D + d) : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < MAXCB; ++i) if (i0 + i < p.ncb) vq += e[i];
                }
                vq = round_bf16(vq);
            }
            float x = round_bf16(ld_elem(p.emb, (size_t)t0 * D + d) + vq);
            if (p.scale_cb && is_vq) x = round_bf16(x / p.inv_div);
            xA[d] = x;
        }
    }

    if (wave < ENG_CW) {
        // =============================== compute waves: rows of the four matrices ===============================
        int q_lo, q_hi, o_lo, o_hi, f_lo, f_hi, d_lo, d_hi;
        eng_units(QKVN, b, nb, q_lo, q_hi);
        if (XL) { q_lo = 0; q_hi = QKVN / nb; }          // local units of this workgroup (rows: rmq below)
        eng_units(D, b, nb, o_lo, o_hi);
        eng_units(F, b, nb, f_lo, f_hi);     // (w1_i, w3_i) pairs
        eng_units(D, b, nb, d_lo, d_hi);
        EngW<NTD, 1, SQ> wq;
        EngW<NTA, 1, SO> wo;
        EngW<NTD, 2, SF> wf;
        EngW<NTF, 1, SO> wd;
        EngOut eo{outS, out_count, 0};
        {
            const EngLayer l0 = eng_layer(p.layers, 0);
            if (!XL) eng_issue(wq, l0.wqkv, l0.attn_norm, D, q_lo, q_hi, cw, lane, p.nt);
            eng_issue(wo, l0.wo, (const bf16_t*)nullptr, HD, o_lo, o_hi, cw, lane, p.nt);
            eng_issue(wf, l0.w13, l0.ffn_norm, D, f_lo, f_hi, cw, lane, p.nt);
            eng_issue(wd, l0.w2, (const bf16_t*)nullptr, F, d_lo, d_hi, cw, lane, p.nt);
            __builtin_amdgcn_sched_barrier(0);
        }
        eng_barrier();                                              // (registration results in LDS)
        // XL: which rows of Wqkv are this workgroup's is known only now (kv head = its XCD, share = its rank there)
        const int xnq = G * hd * Hkv / nb, xnk = hd * Hkv / nb;
        const EngRowQkvX rmq{reg_s[0] * G * hd + reg_s[1] * xnq, (H + reg_s[0]) * hd + reg_s[1] * xnk,
                             (H + Hkv + reg_s[0]) * hd + reg_s[1] * xnk, xnq, xnk};
        if (XL) {
            const EngLayer l0 = eng_layer(p.layers, 0);
            eng_issue(wq, l0.wqkv, l0.attn_norm, D, q_lo, q_hi, cw, lane, p.nt, rmq);
            __builtin_amdgcn_sched_barrier(0);
        }
        eng_barrier();                                              // B0
        for (int li = 0; li < p.n_layer; ++li) {
            const int par = li & 1, parn = (li + 1) & 1;
            const unsigned tag = eng_tag16(epoch + (unsigned)li);
            const unsigned tagn = eng_tag16(epoch + (unsigned)li + 1u);    // the next layer reads W2's output
            const EngLayer l = eng_layer(p.layers, li);
            const bool more = li + 1 < p.n_layer;
            const EngLayer ln = eng_layer(p.layers, more ? li + 1 : li);
            if (li > 0) { eng_barrier(); if (*dead) break; }        // B1
            ENG_STAMP(0);
            if (XL)
                eng_gemv_qkv_local<NTD, SQ>(wq, xA, D, p.eps, l.bqkv, p.gqkv + (size_t)par * VSTR, tag, q_hi, cw, lane, eo, rmq);
            else
                eng_gemv<NTD, 1, SQ, PRO_RMSNORM, EPI_STORE>(wq, xA, D, p.eps, l.bqkv, nullptr, p.gqkv + (size_t)par * VSTR + eng_pub(b, q_lo), tag, nullptr,
                                                             q_lo, q_hi, cw, lane, eo);
            __builtin_amdgcn_sched_barrier(0);
            ENG_STAMP(1);
            if (more) {
                if (XL) eng_issue(wq, ln.wqkv, ln.attn_norm, D, q_lo, q_hi, cw, lane, p.nt, rmq);
                else eng_issue(wq, ln.wqkv, ln.attn_norm, D, q_lo, q_hi, cw, lane, p.nt);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (XL || eng_att_role(b, nb, li, natt) >= 0) {
                eng_barrier(); if (*dead) break;                    // BA
                eng_barrier();                                      // BB
                eng_barrier();                                      // BC
            }
            ENG_STAMP(2);
            eng_barrier(); if (*dead) break;                        // B2
            ENG_STAMP(3);
            eng_gemv<NTA, 1, SO, PRO_NONE, EPI_RESID>(wo, yS, HD, p.eps, l.bo, xA, p.gxb + (size_t)par * VSTR + eng_pub(b, o_lo), tag, nullptr, o_lo, o_hi, cw, lane, eo);
            __builtin_amdgcn_sched_barrier(0);
            ENG_STAMP(4);
            if (p.stamps && (p.nt & 2)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ENG_STAMP(9); }   // write-through acknowledged
            if (more) eng_issue(wo, ln.wo, (const bf16_t*)nullptr, HD, o_lo, o_hi, cw, lane, p.nt);
            __builtin_amdgcn_sched_barrier(0);
            eng_barrier(); if (*dead) break;                        // B3
            ENG_STAMP(5);
            eng_gemv<NTD, 2, SF, PRO_RMSNORM, EPI_SWIGLU, S::PK3>(wf, xB, D, p.eps, nullptr, nullptr,
                                                                   p.gg + (size_t)par * VSTR + (S::PK3 ? 0 : eng_pub(b, f_lo)), tag, nullptr, f_lo, f_hi, cw, lane, eo);
            __builtin_amdgcn_sched_barrier(0);
            ENG_STAMP(6);
            if (more) eng_issue(wf, ln.w13, ln.ffn_norm, D, f_lo, f_hi, cw, lane, p.nt);
            __builtin_amdgcn_sched_barrier(0);
            eng_barrier(); if (*dead) break;                        // B4
            ENG_STAMP(7);
            eng_gemv<NTF, 1, SO, PRO_NONE, EPI_RESID>(wd, gS, F, p.eps, nullptr, xB, p.gx + (size_t)parn * VSTR + eng_pub(b, d_lo), tagn,
                                                      li == p.n_layer - 1 ? p.x_out : nullptr, d_lo, d_hi, cw, lane, eo);
            __builtin_amdgcn_sched_barrier(0);
            ENG_STAMP(8);
            if (more) eng_issue(wd, ln.w2, (const bf16_t*)nullptr, F, d_lo, d_hi, cw, lane, p.nt);
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.nt & 2)) ENG_STAMP(9);
        }
    } else {
        // ====================== gathering waves: input vectors, attention turns, split merge ======================
        // K/V rows of this workgroup's next attention turn: step st covers positions lo + st * NSLOT + ..
        U4 kpf[ENG_KVST], vpf[ENG_KVST];
        float gq0 = 1.f, gq1 = 1.f, gk0 = 1.f, gk1 = 1.f;   // q / k norm gains (elements 2 lane, 2 lane + 1) of that layer
        int att_next = -1;    // the layer those rows belong to
        // planning (which layer is this workgroup's next turn, where are its cache rows and gains) is done EARLY - at kernel
        // entry and at the start of a turn for the turn after it - so that the loads can be issued at B2 without any
        // dependent scalar fetch or division in front of them: every workgroup produces rows in every phase, a
        // microsecond lost by one of them here is lost by all (measured: the x' hand-off completed 0.9 us later)
        int plan_layer = -1;
        const bf16_t *plan_kc = nullptr, *plan_vc = nullptr, *plan_qn = nullptr, *plan_kn = nullptr;
        const int period = (natt > 0 && nb % natt == 0) ? nb / natt : 0;
        int x_role = 0;       // XL: this workgroup's (kv head, split) in EVERY layer = (its XCD, its rank there); set after registration
        auto role = [&](int layer) { return XL ? x_role : eng_att_role(b, nb, layer, natt); };
        auto kv_plan = [&](int from_layer) {
            plan_layer = -1;
            if (from_layer >= p.n_layer) return;
            if (XL) {
                plan_layer = from_layer;
            } else if (period > 0) {
                const int phase = b / natt;                          // this workgroup holds a role in layers == phase (mod period)
                int d = phase - from_layer % period;
                if (d < 0) d += period;
                if (from_layer + d < p.n_layer) plan_layer = from_layer + d;
            } else {
                for (int l2 = from_layer; l2 < p.n_layer; ++l2)
                    if (role(l2) >= 0) { plan_layer = l2; break; }
            }
            if (plan_layer < 0) return;
            const EngLayer l2 = eng_layer(p.layers, plan_layer);
            plan_kc = l2.kc; plan_vc = l2.vc; plan_qn = l2.qn; plan_kn = l2.kn;
        };
        auto kv_issue = [&]() {
            att_next = plan_layer;
#pragma unroll
            for (int st = 0; st < ENG_KVST; ++st) { kpf[st] = U4{0u, 0u, 0u, 0u}; vpf[st] = U4{0u, 0u, 0u, 0u}; }
            if (att_next < 0) return;
            const int a = role(att_next);
            const int kvh = a / nsplit, split = a % nsplit;
            const int lo = split * chunk, hi = min(lo + chunk, pos + 1);
            if (lane < hp) {
                if (plan_qn) { gq0 = eng_ldg_bf16(plan_qn, 2 * lane); gq1 = eng_ldg_bf16(plan_qn, 2 * lane + 1); }
                if (plan_kn) { gk0 = eng_ldg_bf16(plan_kn, 2 * lane); gk1 = eng_ldg_bf16(plan_kn, 2 * lane + 1); }
            }
            const bf16_t* kc = plan_kc + p.cache_off + (size_t)kvh * p.n_slots * hd;
            const bf16_t* vc = plan_vc + p.cache_off + (size_t)kvh * p.n_slots * hd;
#pragma unroll
            for (int st = 0; st < ENG_KVST; ++st) {
                const int j = lo + st * NSLOT + gw * PPW + grp;
                if (j < hi && j != pos) {
                    kpf[st] = eng_ldg16<false>(kc + (size_t)j * hd + gl * 8);
                    vpf[st] = eng_ldg16<false>(vc + (size_t)j * hd + gl * 8);
                }
            }
        };
        if (!XL) { kv_plan(0); kv_issue(); }
        eng_barrier();                                              // (registration results in LDS)
        const EngRelay rl{p.rep_stride != 0, reg_s[1], reg_s[2], p.rep_delta0 + (long)reg_s[0] * p.rep_stride};
        if (XL) { x_role = reg_s[0] * nsplit + reg_s[1]; kv_plan(0); kv_issue(); }
        // rotation entries of this position (the same for every layer)
        float rope_c = 1.f, rope_s = 0.f;
        if (lane < hp) { rope_c = p.rope[((size_t)pos * hp + lane) * 2]; rope_s = p.rope[((size_t)pos * hp + lane) * 2 + 1]; }
        eng_barrier();                                              // B0
        for (int li = 0; li < p.n_layer; ++li) {
            const int par = li & 1;
            const unsigned tag = eng_tag16(epoch + (unsigned)li), tag32 = eng_tag32(epoch + (unsigned)li);
            const EngLayer l = eng_layer(p.layers, li);
            if (li > 0) {
                eng_gather_x(rl, p.gx + (size_t)par * VSTR, layD, 0, D, tag, xA, gw, ENG_GW, lane, p.ctl, dead, li * 8 + 0);
                eng_barrier(); if (*dead) break;                    // B1
            }
            // ---- attention (attn_decode_kernel's arithmetic; these four waves stand in for its 256 threads)
            const int a = role(li);
            if (a >= 0) {     // (workgroup-uniform)
                kv_plan(li + 1);                                    // the turn after this one (scalar fetches land during this turn)
                const int kvh = a / nsplit, split = a % nsplit;
                const int lo = split * chunk;
                const int hi = min(lo + chunk, pos + 1);
                bf16_t* kc = l.kc + p.cache_off + (size_t)kvh * p.n_slots * hd;
                bf16_t* vc = l.vc + p.cache_off + (size_t)kvh * p.n_slots * hd;
                const int slot = gw * PPW + grp;
                const unsigned* gq = p.gqkv + (size_t)par * VSTR;
                ENG_ASTAMP(0, 0);
                // q heads of the group (G*hd granules), new k, new v (hd each): one gathering wave per piece
                constexpr int QFL = XL ? 2 : 0;      // (XL: written inside this XCD)
                if (gw == 0) eng_gather<EngIdent, QFL>(gq, layQ, kvh * G * hd, G * hd, tag, qS, 0, 1, lane, p.ctl, dead, li * 8 + 1);
                if (gw == 1) eng_gather<EngIdent, QFL>(gq, layQ, (H + kvh) * hd, hd, tag, qS + G * hd, 0, 1, lane, p.ctl, dead, li * 8 + 1);
                if (gw == 2) eng_gather<EngIdent, QFL>(gq, layQ, (H + Hkv + kvh) * hd, hd, tag, qS + (G + 1) * hd, 0, 1, lane, p.ctl, dead, li * 8 + 1);
                eng_barrier(); if (*dead) break;                    // BA
                ENG_ASTAMP(0, 1);
                if (gw >= 0) {
                    for (int item = gw; item < G + 2; item += 4) {
                        const float* src = qS + item * hd;
                        const bf16_t* gain = item < G ? l.qn : (item == G ? l.kn : nullptr);
                        float* dst = item < G ? q_s + item * hd : (item == G ? k_new : v_new);
                        if (item == G + 1) {
                            for (int e = lane; e < hd; e += 64) dst[e] = src[e];
                        } else {
                            float x0 = 0.f, x1 = 0.f;
                            if (lane < hp) { x0 = src[2 * lane]; x1 = src[2 * lane + 1]; }
                            if (gain) {
                                const float ss = wave_sum(x0 * x0 + x1 * x1);
                                const float inv = rsqrt_exact(ss / (float)hd + p.eps);
                                if (lane < hp) {
                                    const bool have = att_next == li;     // the gains came with the prefetched rows
                                    const float g0 = have ? (item < G ? gq0 : gk0) : eng_ldg_bf16(gain, 2 * lane);
                                    const float g1 = have ? (item < G ? gq1 : gk1) : eng_ldg_bf16(gain, 2 * lane + 1);
                                    x0 = round_bf16((x0 * inv) * g0);
                                    x1 = round_bf16((x1 * inv) * g1);
                                }
                            }
                            if (lane < hp) {
                                dst[2 * lane] = round_bf16(x0 * rope_c - x1 * rope_s);
                                dst[2 * lane + 1] = round_bf16(x1 * rope_c + x0 * rope_s);
                            }
                        }
                    }
                }
                eng_barrier();
                ENG_ASTAMP(0, 2);
                if (gw >= 0) {
                    if (pos >= lo && pos < hi) {
                        for (int e = atid; e < hd; e += 256) {
                            st_elem(kc, (size_t)pos * hd + e, k_new[e]);
                            st_elem(vc, (size_t)pos * hd + e, v_new[e]);
                        }
                    }
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
                    auto step = [&](const int j, const float (&kv)[8], const float (&vv)[8]) {
                        const bool valid = j < hi;
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
                    };
                    const bool pre = att_next == li;     // the registers hold this layer's first ENG_KVST steps
                    // The K/V registers are a rolling window: while step st is consumed, the rows of step st + ENG_KVST start
                    // their trip into the registers it frees, so a walk of any length keeps ENG_KVST steps in flight (long
                    // contexts, and the unsplit walk of the XCD-local form) with the register count of a short one.
                    for (int blk = 0; lo + gw * PPW + blk * (ENG_KVST * NSLOT) < hi; ++blk) {      // wave-uniform
#pragma unroll
                        for (int st = 0; st < ENG_KVST; ++st) {
                            const int base = lo + gw * PPW + (blk * ENG_KVST + st) * NSLOT;
                            if (base < hi) {                 // wave-uniform
                                const int j = base + grp;
                                float kv[8], vv[8];
                                if (j < hi && j == pos) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) { kv[e] = k_new[gl * 8 + e]; vv[e] = v_new[gl * 8 + e]; }
                                } else if (pre || blk > 0) {
                                    Vec<bf16_t>::unpack(kpf[st], kv); Vec<bf16_t>::unpack(vpf[st], vv);
                                } else if (j < hi) {
                                    Vec<bf16_t>::unpack(eng_ldg16<false>(kc + (size_t)j * hd + gl * 8), kv);
                                    Vec<bf16_t>::unpack(eng_ldg16<false>(vc + (size_t)j * hd + gl * 8), vv);
                                } else {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) { kv[e] = 0.f; vv[e] = 0.f; }
                                }
                                const int jn = j + ENG_KVST * NSLOT;
                                if (jn < hi && jn != pos) {
                                    kpf[st] = eng_ldg16<false>(kc + (size_t)jn * hd + gl * 8);
                                    vpf[st] = eng_ldg16<false>(vc + (size_t)jn * hd + gl * 8);
                                } else {
                                    kpf[st] = U4{0u, 0u, 0u, 0u}; vpf[st] = U4{0u, 0u, 0u, 0u};
                                }
                                step(j, kv, vv);
                            }
                        }
                    }
                    ENG_ASTAMP(0, 7);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (gl == 0) { ml_s[(slot * G + g) * 2] = mrun[g]; ml_s[(slot * G + g) * 2 + 1] = lrun[g]; }
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc_s[(size_t)(slot * G + g) * hd + gl * 8 + e] = acc[g][e];
                    }
                    ENG_ASTAMP(0, 8);
                }
                eng_barrier();
                ENG_ASTAMP(0, 3);
                if (gw >= 0) {
                    // the combination of the NSLOT running (max, sum, acc) triples (attn_decode_kernel's): the weight of a
                    // slot does not depend on the element, so lane s computes slot s's once and every lane reads it from
                    // there (was: 16 exponentials per element and lane, 2.6 us on the path of every layer)
                    const bool quick = NSLOT == 16 && (hd & 63) == 0;
                    for (int idx0 = gw * 64; idx0 < G * hd; idx0 += 256) {
                        const int idx = idx0 + lane;
                        const int g = quick ? idx0 / hd : idx / hd, e = idx % hd;
                        float M = -INFINITY, L = 0.f, O = 0.f;
                        if (quick) {
                            const int sl = lane & 15;
                            const float m_l = ml_s[(sl * G + g) * 2], l_l = ml_s[(sl * G + g) * 2 + 1];
                            M = m_l;
                            M = fmaxf(M, dpp_f<DPP_XOR1>(M));
                            M = fmaxf(M, dpp_f<DPP_XOR2>(M));
                            M = fmaxf(M, dpp_f<DPP_HALF_MIRROR>(M));
                            M = fmaxf(M, dpp_f<DPP_MIRROR>(M));
                            if (M > -INFINITY) {
                                const float w_l = expf(m_l - M);
#pragma unroll
                                for (int s2 = 0; s2 < 16; ++s2) {
                                    const float w = lane_f(w_l, s2);
                                    L += lane_f(l_l, s2) * w;
                                    O += acc_s[(size_t)(s2 * G + g) * hd + e] * w;
                                }
                            }
                        } else if (idx < G * hd) {
                            for (int s2 = 0; s2 < NSLOT; ++s2) M = fmaxf(M, ml_s[(s2 * G + g) * 2]);
                            if (M > -INFINITY) {
                                for (int s2 = 0; s2 < NSLOT; ++s2) {
                                    const float w = expf(ml_s[(s2 * G + g) * 2] - M);
                                    L += ml_s[(s2 * G + g) * 2 + 1] * w;
                                    O += acc_s[(size_t)(s2 * G + g) * hd + e] * w;
                                }
                            }
                        }
                        if (idx < G * hd) {
                            const int head = kvh * G + g;
                            if (nsplit == 1) {
                                eng_put(p.gy + (size_t)par * HD, head * hd + e, round_bf16(O / L), tag);
                            } else {
                                unsigned long long* gp = p.gpart + (((size_t)par * H + head) * nsplit + split) * (hd + 2);
                                if (XL) {       // the mergers are this XCD's workgroups: the partials stay in its L2
                                    eng_put64_local(gp, e, O, tag32);
                                    if (e == 0) { eng_put64_local(gp, hd, M, tag32); eng_put64_local(gp, hd + 1, L, tag32); }
                                } else {
                                    eng_put64(gp, e, O, tag32);
                                    if (e == 0) { eng_put64(gp, hd, M, tag32); eng_put64(gp, hd + 1, L, tag32); }
                                }
                            }
                        }
                    }
                }
                // ---- merge of the split partials (merge_splits4's arithmetic): this workgroup merges elements
                // [split * hd / nsplit, (split + 1) * hd / nsplit) of its G heads.  An item = 4 consecutive elements of
                // one head; lane 8 * i + s polls split c0 + s of item i (O[0..3], m, l = three 16-byte loads), the
                // values cross to the item's first lane through LDS, which merges in split order.
                ENG_ASTAMP(0, 4);
                if (XL && gw == 3 && nsplit == 32 && hd == 128 && G == 2) {
                    // ---- XL merge, ONE poll round: this workgroup merges elements [4 split, 4 split + 4) of its kv head's two query
                    // heads; lane 32 i + s polls split s of head i (O[0..3], m, l = three 16-byte loads).  merge_splits4's
                    // arithmetic: splits in chunks of 8, running maximum, rescale at each chunk, sums taken split by split.
                    // The per-split weights and products are formed by the polling lanes in parallel; five chains per head
                    // (l, a0..a3; lanes 8 i + q) then add them in split order from LDS.
                    const int item = lane >> 5, s = lane & 31, e = split * 4;
                    const int head = kvh * G + item;
                    const unsigned long long* gs = p.gpart + (((size_t)par * H + head) * nsplit + s) * (hd + 2);
                    EngSpin sp{p.ctl, dead, 0, 0, li * 8 + 2};
                    U4 A, B, C;
                    bool alive = true;
                    for (;;) {
                        eng_ld3_sc1<2>(gs + e, gs + e + 2, gs + hd, A, B, C);
                        const bool ok = A.y == tag32 && A.w == tag32 && B.y == tag32 && B.w == tag32 && C.y == tag32 && C.w == tag32;
                        if (__all(ok)) break;
                        if (sp.give_up(lane)) { alive = false; break; }
                    }
                    ENG_ASTAMP(3, 5);
                    if (alive) {
                        const float m_ = __uint_as_float(C.x), l_ = __uint_as_float(C.z);
                        float Mc = m_;                                  // maximum of this lane's chunk of 8 splits
                        Mc = fmaxf(Mc, dpp_f<DPP_XOR1>(Mc));
                        Mc = fmaxf(Mc, dpp_f<DPP_XOR2>(Mc));
                        Mc = fmaxf(Mc, dpp_f<DPP_HALF_MIRROR>(Mc));
                        float* mx = mscr;                               // [2 heads][4 chunks] chunk maxima, then [2][32][5] products
                        float* pr = mscr + 8;
                        if ((lane & 7) == 0) mx[lane >> 3] = Mc;
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        // running maximum after this lane's chunk (empty chunks - maximum -inf - leave it unchanged)
                        float Mrun = -INFINITY;
                        const int ch = s >> 3;
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (c <= ch) Mrun = fmaxf(Mrun, mx[item * 4 + c]);
                        const float w = m_ > -INFINITY ? expf(m_ - Mrun) : 0.f;
                        float* q5 = pr + (size_t)(item * 32 + s) * 5;
                        q5[0] = l_ * w;
                        q5[1] = __uint_as_float(A.x) * w; q5[2] = __uint_as_float(A.z) * w;
                        q5[3] = __uint_as_float(B.x) * w; q5[4] = __uint_as_float(B.z) * w;
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        // chains: lane 8 i + q (q < 5) sums quantity q of head i over the 32 splits in split order.  Branch-free:
                        // an empty chunk's products are exact zeros and its rescale factor an exact 1 (as is the factor of the
                        // first visible chunk), so adding / multiplying them changes no bit; all 32 LDS reads and the three
                        // exponentials are issued before the dependent chain of additions starts.
                        const int ci = (lane >> 3) & 1, cq = lane & 7;
                        float acc = 0.f;
                        {
                            const float* src = pr + (size_t)ci * 32 * 5 + (cq < 5 ? cq : 0);
                            float v[32];
#pragma unroll
                            for (int k = 0; k < 32; ++k) v[k] = src[k * 5];
                            float mc[4], rr[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) mc[c] = mx[ci * 4 + c];
                            float M = -INFINITY;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float Mn = fmaxf(M, mc[c]);
                                rr[c] = (c > 0 && M > -INFINITY && mc[c] > -INFINITY) ? expf(M - Mn) : 1.0f;
                                M = Mn;
                            }
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                acc *= rr[c];
#pragma unroll
                                for (int k = 0; k < 8; ++k) acc += v[c * 8 + k];
                            }
                        }
                        // the head's l sits in lane 8 i, a0..a3 in the four lanes after it
                        const float a0 = dpp_f<0x101>(acc), a1 = dpp_f<0x102>(acc), a2 = dpp_f<0x103>(acc), a3 = dpp_f<0x104>(acc);
                        if (lane == 0 || lane == 8) {
                            unsigned* gy = p.gy + (size_t)par * HD + (size_t)a * (G * 4) + ci * 4;
                            eng_put4(gy, round_bf16(a0 / acc), round_bf16(a1 / acc), round_bf16(a2 / acc), round_bf16(a3 / acc), tag);
                        }
                    }
                    ENG_ASTAMP(3, 6);
                } else
                if (nsplit > 1 && gw == 3) {
                    const int epb = hd / nsplit;            // elements per workgroup and head (multiple of 4)
                    const int e4n = epb >> 2;
                    const int nitem = G * e4n;
                    EngSpin sp{p.ctl, dead, 0, 0, li * 8 + 2};
                    bool alive = true;
                    for (int ib = 0; ib < nitem && alive; ib += 8) {
                        const int it = ib + (lane >> 3), s8 = lane & 7;
                        const bool item_on = it < nitem;
                        const int g = item_on ? it / e4n : 0;
                        const int e = split * epb + (item_on ? (it % e4n) * 4 : 0);
                        const int head = kvh * G + g;
                        const unsigned long long* gp = p.gpart + ((size_t)par * H + head) * nsplit * (hd + 2);
                        float M = -INFINITY, L = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                        for (int c0 = 0; c0 < nsplit && alive; c0 += 8) {
                            const bool son = item_on && c0 + s8 < nsplit;
                            const unsigned long long* gs = gp + (size_t)(son ? c0 + s8 : 0) * (hd + 2);
                            U4 A, B, C;
                            for (;;) {
                                eng_ld3_sc1(gs + e, gs + e + 2, gs + hd, A, B, C);
                                const bool ok = !son || (A.y == tag32 && A.w == tag32 && B.y == tag32 && B.w == tag32 &&
                                                         C.y == tag32 && C.w == tag32);
                                if (__all(ok)) break;
                                if (sp.give_up(lane)) { alive = false; break; }
                            }
                            if (!alive) break;
                            if (ib + 8 >= nitem && c0 + 8 >= nsplit) ENG_ASTAMP(3, 5);      // last partials seen
                            // lane 8 i + s holds split c0 + s of item i: the weights are formed in parallel, the sums are
                            // taken in split order by the item's first lane (row_shl reads lane + s of the 16-lane row)
                            const float m_ = son ? __uint_as_float(C.x) : -INFINITY;
                            const float l_ = son ? __uint_as_float(C.z) : 0.f;
                            float Mc = m_;
                            Mc = fmaxf(Mc, dpp_f<DPP_XOR1>(Mc));
                            Mc = fmaxf(Mc, dpp_f<DPP_XOR2>(Mc));
                            Mc = fmaxf(Mc, dpp_f<DPP_HALF_MIRROR>(Mc));
                            if (Mc > -INFINITY) {
                                const float Mn = fmaxf(M, Mc);
                                if (c0 > 0 && M > -INFINITY) {
                                    const float rr = expf(M - Mn);
                                    L *= rr; a0 *= rr; a1 *= rr; a2 *= rr; a3 *= rr;
                                }
                                M = Mn;
                                const float w = m_ > -INFINITY ? expf(m_ - M) : 0.f;
                                const float pl = l_ * w;
                                const float p0 = __uint_as_float(A.x) * w, p1 = __uint_as_float(A.z) * w;
                                const float p2 = __uint_as_float(B.x) * w, p3 = __uint_as_float(B.z) * w;
                                L += pl; a0 += p0; a1 += p1; a2 += p2; a3 += p3;
#define ENG_MSTEP(n) L += dpp_f<0x100 + n>(pl); a0 += dpp_f<0x100 + n>(p0); a1 += dpp_f<0x100 + n>(p1); \
                     a2 += dpp_f<0x100 + n>(p2); a3 += dpp_f<0x100 + n>(p3);
                                ENG_MSTEP(1) ENG_MSTEP(2) ENG_MSTEP(3) ENG_MSTEP(4) ENG_MSTEP(5) ENG_MSTEP(6) ENG_MSTEP(7)
#undef ENG_MSTEP
                            }
                        }
                        if (alive && s8 == 0 && item_on) {
                            unsigned* gy = p.gy + (size_t)par * HD + (size_t)a * (G * epb) + g * epb + (e - split * epb);
                            eng_put4(gy, round_bf16(a0 / L), round_bf16(a1 / L), round_bf16(a2 / L), round_bf16(a3 / L), tag);
                        }
                    }
                    ENG_ASTAMP(3, 6);
                }
            }
            {
                // y is stored per attention workgroup: [kvh][split][g][hd / nsplit] (each 128-byte line written by one of them)
                const int epb = hd / nsplit, ns = nsplit;
                auto ymap = [=](int i) { const int a2 = i / (G * epb), r2 = i % (G * epb), g2 = r2 / epb, eo = r2 % epb;
                                         return ((a2 / ns) * G + g2) * hd + (a2 % ns) * epb + eo; };
                eng_gather_x(rl, p.gy + (size_t)par * HD, layLin, 0, HD, tag, yS, gw, ENG_GW, lane, p.ctl, dead, li * 8 + 3, ymap);
            }
            eng_barrier(); if (*dead) break;                        // B2
            // the K/V rows (and norm gains) of this workgroup's next attention turn start their trip now: the polls
            // for x' wait at least a Wo phase anyway (vmcnt is in order: requested right after the partials they would
            // sit in front of the merge polls and of the y polls)
            if (a >= 0) kv_issue();
            const unsigned long long t_poll0 = p.stamps ? eng_rt() : 0ull;   // kept in a register: a store here would sit in front of the polls
            eng_gather_x(rl, p.gxb + (size_t)par * VSTR, layD, 0, D, tag, xB, gw, ENG_GW, lane, p.ctl, dead, li * 8 + 4, EngIdent(),
                       (p.stamps && gw == 0) ? p.stamps + ((size_t)b * p.n_layer + li) * 16 + 13 : nullptr);
            const unsigned long long t_poll1 = p.stamps ? eng_rt() : 0ull;
            eng_barrier(); if (*dead) break;                        // B3
            if (p.stamps && tid == ENG_CW * 64) {
                unsigned long long* q = p.stamps + ((size_t)b * p.n_layer + li) * 16;
                q[10] = t_poll0; q[11] = t_poll1; q[12] = eng_rt();
            }
            if constexpr (S::PK3) eng_gather3_x(rl, p.gg + (size_t)par * VSTR, F, tag, gS, gw, ENG_GW, lane, p.ctl, dead, li * 8 + 5);
            else eng_gather_x(rl, p.gg + (size_t)par * VSTR, layF, 0, F, tag, gS, gw, ENG_GW, lane, p.ctl, dead, li * 8 + 5);
            eng_barrier(); if (*dead) break;                        // B4
        }
    }

    // ---- leave: the last workgroup out advances the epoch for the next launch
    eng_barrier();
    if (p.stamps && tid == 0 && b == 0) {   // shader clock of this launch = d(s_memtime) / d(s_memrealtime) x 100 MHz
        p.stamps[14] = __builtin_amdgcn_s_memtime() - clk0; p.stamps[15] = eng_rt() - rt0;
    }
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

// ==========================================================================================
// Fast codebook loop (inference.py:115-149, llama.py:561-580): num_codebooks steps of n_fast_layer blocks, the
// 1024-row codebook head and the draw of each codebook, ONE launch.
// Per step and layer: [x] -> QKV -> every workgroup rebuilds the (<= num_codebooks positions) attention for all heads
// from the gathered q/k/v, its K/V history kept in its own LDS -> Wo + residual -> [x'] -> W13 + SwiGLU -> [g] ->
// W2 + residual -> [x''].  After the last layer (steps >= 1): fast_norm + head rows -> [logits] -> ONE workgroup
// draws the code (sample_small_kernel's arithmetic on its four gathering waves) -> [code] -> every workgroup
// reads that code's embedding row as the next step's input.
// Hand-off buffers are reused by the steps, double-buffered by step parity (step 1 does not wait for a draw, so a
// fast workgroup may start it while a slow one still reads step 0's vectors); tags are epoch + step.
// ==========================================================================================
struct FastEngP {
    const EngLayer* layers;       // fast layers (kc / vc unused)
    int n_layer, ncb;
    int D, H, Hkv, hd, F, qkvN, V;     // V = rows of the head that are used (min(1024, codebook_size))
    float eps, scale;
    const float* rope;            // [ncb][hd/2][2]
    const bf16_t* fast_norm; const bf16_t* fast_out; const bf16_t* fast_emb;
    const float* hid;             // plain f32 [D]: step 0 input (the slow stack's pre-norm hidden state)
    const float* femb;            // plain f32 [D]: step 1 input (embedding of the semantic code, left by the slow draw)
    // hand-off buffers, each [2 parities][...]
    unsigned* gx;                 // [2][n_layer + 1][nb lines]  layer inputs / stack output
    unsigned* gqkv;               // [2][n_layer][nb lines]
    unsigned* gxb;                // [2][n_layer][nb lines]
    unsigned* gg;                 // [2][n_layer][nb lines]
    unsigned* glog;               // [2][nb lines]
    unsigned* gcode;              // [ncb][ENG_LINE]
    unsigned* ctl;
    long rep_delta0, rep_stride;  // as in SlowEngP
    SampP samp;                   // sampling state (cb, noise_off, last are set per step in the kernel)
    long noise_cb_stride;         // fastV
    long noise_off1;              // offset of codebook 1's noise in a row (vocab_size)
    int pair;                     // 1: codebook positions 0 and 1 (both inputs known at launch) run as two rows of ONE pass
    const bf16_t* qkv0_tab;       // [V][qkvN] bf16 or nullptr: layer 0's q k v of every codebook-embedding row (eng_qkv0_table_kernel):
                                  // from position 2 on the layer-0 input is one of V table rows, so its QKV phase and hand-off are a lookup
    unsigned long long* stamps;   // diagnostic builds only (tools/mb_engine.hip): [workgroup][step][layer][16] ticks, or nullptr
};
#define ENG_FSTAMP(k) do { if (p.stamps && tid == 0) p.stamps[(((size_t)b * p.ncb + cb) * nL + li) * 16 + (k)] = eng_rt(); } while (0)

#define ENG_GSTAMP(k) do { if (p.stamps && atid == 0) p.stamps[(((size_t)b * p.ncb + cb) * nL + nL - 1) * 16 + (k)] = eng_rt(); } while (0)
constexpr int ENG_FV = 1024;    // codes drawn per codebook (inference.py:134); every fast shape class has this many

// The draw's barriers.  The four gathering waves draw; the compute waves have nothing to do meanwhile, so they FOLLOW: they
// sit in the workgroup's hardware barrier once per barrier of the draw (eng_draw_follow) and leave when the draw says it
// is over.  A barrier among the four drawing waves alone had to be an LDS counter they spin on (~0.2 us each, eight per
// draw); the hardware barrier costs a few dozen cycles.
// `end_at` (LDS): 0 while the number of barriers of the current draw is not known yet, then that number - written by the
// drawing waves in front of their last barrier; a follower leaves when its own barrier count equals it (so it does not
// matter whether it reads the word before or after that last barrier).
struct EngSub {
    int* end_at;
    int n;           // barriers of the current draw so far
    __device__ __forceinline__ void sync(int) { ++n; eng_barrier(); }
};
__device__ __forceinline__ void eng_draw_follow(const int* end_at) {
    for (int c = 1;; ++c) {
        eng_barrier();
        if (*reinterpret_cast<const volatile int*>(end_at) == c) break;
    }
}

// The attention of one fast layer at codebook position c for the heads h = w, w + nw, .. (fast_attn_kernel's arithmetic:
// one wave per head, lane d owns dimension d (and d + 64)); q/k/v of this position come from qkvS (LDS, f32), the
// earlier positions' K/V from kvS (LDS, bf16 bits; row j of layer-local cache: [j][Hkv * hd] K then V).
template <int MAXCB, int HD>
__device__ __forceinline__ void eng_fast_attn(const float* qkvS, bf16_t* kS, bf16_t* vS, float* yS, const bf16_t* qn, const bf16_t* kn,
                                              const float (&cs)[2], const float (&sn)[2], int c, int ncb, int H, int Hkv, float eps, float scale,
                                              int w, int nw, int lane) {
    // head_dim is a compile-time constant here (64 or 128): lane d owns dimension d (and d + 64 when HD = 128); the K/V
    // history of a layer is laid out [kv head][position][HD] so a position is an immediate offset from the head's base
    constexpr int EPL = HD > 64 ? 2 : 1;
    const int G = H / Hkv;
    for (int h = w; h < H; h += nw) {
        const int kvh = h / G;
        bf16_t* kh = kS + (size_t)kvh * ncb * HD + lane;
        bf16_t* vh = vS + (size_t)kvh * ncb * HD + lane;
        const float* qp = qkvS + h * HD + lane;
        const float* kp = qkvS + (H + kvh) * HD + lane;
        const float* vp = qkvS + (H + Hkv + kvh) * HD + lane;
        float q[EPL], kx[EPL], vx[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) { q[e] = qp[64 * e]; kx[e] = kp[64 * e]; vx[e] = vp[64 * e]; }
        if (c == 0) {
            // one visible position: the softmax weight is round_bf16(exp(0) / exp(0)) = 1 and the output fma(1, v, 0) = v
            // whatever q is, so only the K/V rows of the history are produced (same operations on k as below)
            if (h % G == 0) {
                if (kn) {
                    float ss = kx[0] * kx[0];
                    if (EPL > 1) ss = kx[0] * kx[0] + kx[EPL - 1] * kx[EPL - 1];
                    ss = wave_sum(ss);
                    const float inv = rsqrt_exact(ss / (float)HD + eps);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) kx[e] = round_bf16((kx[e] * inv) * eng_ldg_bf16(kn, lane + 64 * e));
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float ko = dpp_f<DPP_XOR1>(kx[e]);
                    kx[e] = round_bf16((lane & 1) == 0 ? kx[e] * cs[e] - ko * sn[e] : kx[e] * cs[e] + ko * sn[e]);
                    kh[64 * e] = f32_to_bf16_bits(kx[e]); vh[64 * e] = f32_to_bf16_bits(vx[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) yS[h * HD + lane + 64 * e] = round_bf16(vx[e]);
            continue;
        }
        if (qn) {
            float ss = q[0] * q[0];
            if (EPL > 1) ss = q[0] * q[0] + q[EPL - 1] * q[EPL - 1];
            ss = wave_sum(HD < 64 ? (lane < HD ? ss : 0.f) : ss);
            const float inv = rsqrt_exact(ss / (float)HD + eps);
#pragma unroll
            for (int e = 0; e < EPL; ++e) q[e] = round_bf16((q[e] * inv) * eng_ldg_bf16(qn, lane + 64 * e));
        }
        if (kn) {
            float ss = kx[0] * kx[0];
            if (EPL > 1) ss = kx[0] * kx[0] + kx[EPL - 1] * kx[EPL - 1];
            ss = wave_sum(ss);
            const float inv = rsqrt_exact(ss / (float)HD + eps);
#pragma unroll
            for (int e = 0; e < EPL; ++e) kx[e] = round_bf16((kx[e] * inv) * eng_ldg_bf16(kn, lane + 64 * e));
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float qo = dpp_f<DPP_XOR1>(q[e]), ko = dpp_f<DPP_XOR1>(kx[e]);
            const bool even = (lane & 1) == 0;
            q[e] = round_bf16(even ? q[e] * cs[e] - qo * sn[e] : q[e] * cs[e] + qo * sn[e]);
            kx[e] = round_bf16(even ? kx[e] * cs[e] - ko * sn[e] : kx[e] * cs[e] + ko * sn[e]);
        }
        if (h % G == 0) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { kh[c * HD + 64 * e] = f32_to_bf16_bits(kx[e]); vh[c * HD + 64 * e] = f32_to_bf16_bits(vx[e]); }
        }
        // scores (every lane gets every score: the wave reductions are fast_attn_kernel's); the score of position j is
        // then kept by lane j alone, so the exponentials and the divisions of the softmax run ONCE per head instead of
        // once per position: same operations on the same values, the sum is taken in position order as there
        float sl = -INFINITY;      // lane j: score j
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < MAXCB; ++j) {
            if (j <= c) {
                float pr;
                if (EPL > 1) {
                    const float k0 = j == c ? kx[0] : bf16_bits_to_f32(kh[j * HD]), k1 = j == c ? kx[EPL - 1] : bf16_bits_to_f32(kh[j * HD + 64 * (EPL - 1)]);
                    pr = fmaf(q[EPL - 1], k1, q[0] * k0);
                } else {
                    const float k0 = j == c ? kx[0] : bf16_bits_to_f32(kh[j * HD]);
                    pr = fmaf(0.f, 0.f, q[0] * k0);       // fast_attn_kernel's second element is an exact zero term at HD <= 64
                }
                const float d = wave_sum(pr);
                const float sj = round_bf16(round_bf16(d) * scale);
                mx = fmaxf(mx, sj);
                if (lane == j) sl = sj;
            }
        }
        const float el = lane <= c ? expf(sl - mx) : 0.f;      // lane j: exp(score j - max)
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MAXCB; ++j)
            if (j <= c) sum += lane_f(el, j);
        const float pl = lane <= c ? round_bf16(el / sum) : 0.f;   // lane j: probability j
        float o[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < MAXCB; ++j) {
            if (j <= c) {
                const float pj = lane_f(pl, j);
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] = fmaf(pj, j == c ? vx[e] : bf16_bits_to_f32(vh[j * HD + 64 * e]), o[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) yS[h * HD + lane + 64 * e] = round_bf16(o[e]);
    }
}

// eng_fast_attn at head_dim 64 with far fewer instructions (the phase is issue-bound: ~450 per head above), same bits:
//  * a wave takes ONE kv head and its G query heads (the K/V row of the new position is normalised, rotated and stored once,
//    and every K/V row is written and read by the same wave);
//  * the scores of 4 positions are formed at a time: lane 16 r + k holds dimensions 4k..4k+3 of position 4 ps + r and
//    builds wave_sum's tree from them - (p0 + p1) + (p2 + p3) in the lane = its xor-1 / xor-2 steps, xor-1 / xor-2 across
//    the lanes = its half-mirror / mirror steps (8- and 16-dimension blocks), then the four 16-blocks in order
//    ((R0 + R1) + R2) + R3 = its readlane chain;
//  * q (dimension per lane after the rotation) reaches that layout through the head's own slot of yS, which also
//    carries the scores to "lane j holds score j"; the slot receives the head's output last.
constexpr int DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108, DPP_ROW_SHL12 = 0x10C;   // lane i <- lane i + n of its 16-lane row
template <int MAXCB, int GQ>
__device__ __forceinline__ void eng_fast_attn64(const float* qkvS, bf16_t* kS, bf16_t* vS, float* yS, const bf16_t* qn, const bf16_t* kn,
                                                float cs, float sn, int c, int ncb, int H, int Hkv, float eps, float scale,
                                                int w, int nw, int lane) {
    // GQ = query heads per kv head (compile time): their independent chains are written stage by stage so they interleave
    constexpr int HD = 64;
    const bool even = (lane & 1) == 0;
    const int r = lane >> 4, k4 = (lane & 15) * 4;
    for (int kvh = w; kvh < Hkv; kvh += nw) {
        bf16_t* kh = kS + (size_t)kvh * ncb * HD;
        bf16_t* vh = vS + (size_t)kvh * ncb * HD;
        float kx = qkvS[(H + kvh) * HD + lane];
        const float vx = qkvS[(H + Hkv + kvh) * HD + lane];
        float* slot = yS + (size_t)kvh * GQ * HD;            // the GQ heads' slots are consecutive
        float q[GQ];
#pragma unroll
        for (int g = 0; g < GQ; ++g) q[g] = qkvS[(kvh * GQ + g) * HD + lane];
        if (kn) {
            const float ss = wave_sum(kx * kx);
            const float inv = rsqrt_exact(ss / (float)HD + eps);
            kx = round_bf16((kx * inv) * eng_ldg_bf16(kn, lane));
        }
        {
            const float ko = dpp_f<DPP_XOR1>(kx);
            kx = round_bf16(even ? kx * cs - ko * sn : kx * cs + ko * sn);
        }
        kh[c * HD + lane] = f32_to_bf16_bits(kx);
        vh[c * HD + lane] = f32_to_bf16_bits(vx);
        if (c == 0) {       // one visible position: weight round_bf16(1 / 1) = 1, output fma(1, v, 0) = v
#pragma unroll
            for (int g = 0; g < GQ; ++g) slot[g * HD + lane] = round_bf16(vx);
            continue;
        }
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            if (qn) {
                const float ss = wave_sum(q[g] * q[g]);
                const float inv = rsqrt_exact(ss / (float)HD + eps);
                q[g] = round_bf16((q[g] * inv) * eng_ldg_bf16(qn, lane));
            }
            const float qo = dpp_f<DPP_XOR1>(q[g]);
            q[g] = round_bf16(even ? q[g] * cs - qo * sn : q[g] * cs + qo * sn);
            slot[g * HD + lane] = q[g];
        }
        __builtin_amdgcn_wave_barrier();
        float4 q4[GQ];
#pragma unroll
        for (int g = 0; g < GQ; ++g) q4[g] = *reinterpret_cast<const float4*>(slot + g * HD + k4);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < (MAXCB + 3) / 4; ++ps) {
            if (ps * 4 <= c) {                                  // wave-uniform
                const int j = ps * 4 + r;
                const int jj = j <= c ? j : c;                  // rows past the last position re-read it (not stored)
                const uint2 kb = *reinterpret_cast<const uint2*>(kh + jj * HD + k4);
                const float k0 = __uint_as_float(kb.x << 16), k1 = __uint_as_float(kb.x & 0xffff0000u);
                const float k2 = __uint_as_float(kb.y << 16), k3 = __uint_as_float(kb.y & 0xffff0000u);
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    // fast_attn_kernel's per-lane product carries an exact zero second term at head_dim <= 64
                    const float p0 = fmaf(0.f, 0.f, q4[g].x * k0), p1 = fmaf(0.f, 0.f, q4[g].y * k1);
                    const float p2 = fmaf(0.f, 0.f, q4[g].z * k2), p3 = fmaf(0.f, 0.f, q4[g].w * k3);
                    float sdot = (p0 + p1) + (p2 + p3);
                    sdot += dpp_f<DPP_XOR1>(sdot);
                    sdot += dpp_f<DPP_XOR2>(sdot);
                    float a = sdot + dpp_f<DPP_ROW_SHL4>(sdot);
                    a = a + dpp_f<DPP_ROW_SHL8>(sdot);
                    a = a + dpp_f<DPP_ROW_SHL12>(sdot);
                    const float sj = round_bf16(round_bf16(a) * scale);
                    if ((lane & 15) == 0 && j <= c) slot[g * HD + j] = sj;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        float sl[GQ], el[GQ], sum[GQ], pl[GQ], o[GQ];
#pragma unroll
        for (int g = 0; g < GQ; ++g) sl[g] = lane <= c ? slot[g * HD + lane] : -INFINITY;   // lane j: score j
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            float mx = sl[g];                                       // positions live in lanes 0..15 (MAXCB <= 16)
            mx = fmaxf(mx, dpp_f<DPP_XOR1>(mx));
            mx = fmaxf(mx, dpp_f<DPP_XOR2>(mx));
            mx = fmaxf(mx, dpp_f<DPP_HALF_MIRROR>(mx));
            mx = fmaxf(mx, dpp_f<DPP_MIRROR>(mx));
            mx = lane_f(mx, 0);
            el[g] = lane <= c ? expf(sl[g] - mx) : 0.f;
            sum[g] = 0.f; o[g] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < MAXCB; ++j)
            if (j <= c) {
#pragma unroll
                for (int g = 0; g < GQ; ++g) sum[g] += lane_f(el[g], j);
            }
#pragma unroll
        for (int g = 0; g < GQ; ++g) pl[g] = lane <= c ? round_bf16(el[g] / sum[g]) : 0.f;
#pragma unroll
        for (int j = 0; j < MAXCB; ++j)
            if (j <= c) {
                const float vj = bf16_bits_to_f32(vh[j * HD + lane]);
#pragma unroll
                for (int g = 0; g < GQ; ++g) o[g] = fmaf(lane_f(pl[g], j), vj, o[g]);
            }
#pragma unroll
        for (int g = 0; g < GQ; ++g) slot[g * HD + lane] = round_bf16(o[g]);
    }
}
template <int MAXCB, int HD>
__device__ __forceinline__ void eng_fast_attn_any(const float* qkvS, bf16_t* kS, bf16_t* vS, float* yS, const bf16_t* qn, const bf16_t* kn,
                                                  const float (&cs)[2], const float (&sn)[2], int c, int ncb, int H, int Hkv, float eps,
                                                  float scale, int w, int nw, int lane) {
    if constexpr (HD == 64 && MAXCB <= 16) {
        const int G = H / Hkv;
        if (G == 2) { eng_fast_attn64<MAXCB, 2>(qkvS, kS, vS, yS, qn, kn, cs[0], sn[0], c, ncb, H, Hkv, eps, scale, w, nw, lane); return; }
        if (G == 1) { eng_fast_attn64<MAXCB, 1>(qkvS, kS, vS, yS, qn, kn, cs[0], sn[0], c, ncb, H, Hkv, eps, scale, w, nw, lane); return; }
    }
    eng_fast_attn<MAXCB, HD>(qkvS, kS, vS, yS, qn, kn, cs, sn, c, ncb, H, Hkv, eps, scale, w, nw, lane);
}

// The draw of one codebook from V <= 1024 logits in LDS by the four gathering waves (256 threads): sample_small_kernel's
// arithmetic (ar_kernels.h), its workgroup barriers replaced by the four-wave barrier.  Returns the drawn index.
struct EngSampLds {
    float* redbuf;   // [8]
    int* pen_id;     // [32]
    float* pen_val;  // [32]
    float* amv;      // [2][4]
    int* ami;        // [2][4]
    int* wcnt;       // [4]
    float* prL;      // [1024]
    uint32_t* keyL;  // [1024]
    uint32_t* cut;   // [3]: k, nk, all
};
// What a draw needs besides the logits - sampling controls, frame number, this thread's id of the repetition window, this
// thread's four noise values - is fetched / generated at the START of the codebook step, not between the arrival of the
// logits and the draw (two dependent global loads and a Philox call were on that path)
struct EngDrawPre {
    RowCtl ctl;
    int nfv;
    int pen;        // thread < 16: its id of the 16-frame window of codebook cb (or -1)
    float q4[4];
};
__device__ __forceinline__ EngDrawPre eng_draw_pre(const SampP& p, int tid) {
    EngDrawPre d;
    d.ctl = p.ctl[0];
    d.nfv = p.nf[0];
    d.pen = -1;
    if (d.nfv > 0 && p.cb != 0 && tid < 16) {
        const int it = d.nfv - 1;
        const int ws = it < 16 ? 0 : it - 16;
        d.pen = p.seq[(size_t)(p.cb + 1) * p.cap + ws + 1 + tid];
    }
    const float* qrow = nullptr;
    if (p.noise && d.nfv < p.noise_rows) qrow = p.noise + (size_t)d.nfv * p.noise_row_len + p.noise_off;
    draw_noise4(p, d.ctl, qrow, 4 * tid, d.nfv, 0, ENG_FV, d.q4);
    return d;
}
__device__ __forceinline__ int eng_sample_small(const SampP& p, const EngDrawPre& pre, const float* L, const EngSampLds& S, EngSub& sub, int tid, int lane, int wave,
                                                unsigned long long* stp = nullptr) {
    constexpr int V = ENG_FV;          // (= p.V: the host's gate admits only this many used codes)
    const RowCtl ctl = pre.ctl;
    const int nfv = pre.nfv;
    const int R = p.ncb + 1;
    const int* seq = p.seq;
    int red_phase = 0;
    auto red_sum = [&](float v) {
        v = wave_sum(v);
        float* slot = S.redbuf + 4 * (red_phase & 1);
        ++red_phase;
        if (lane == 0) slot[wave] = v;
        sub.sync(lane);
        return ((slot[0] + slot[1]) + slot[2]) + slot[3];
    };
    const int i0 = 4 * tid;
    float l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = (i0 + e) < V ? L[i0 + e] : -INFINITY;
    if (nfv > 0) {
        const int it = nfv - 1;
        const int ws = it < 16 ? 0 : it - 16;
        const int npen = p.cb == 0 ? R : 16;
        (void)ws;
        if (tid < npen) {
            const int id = p.cb == 0 ? seq[(size_t)tid * p.cap + ws + 1] : pre.pen;
            S.pen_id[tid] = -1;
            if (id >= 0 && id < V) {
                const float sv = L[id];
                S.pen_id[tid] = id;
                S.pen_val[tid] = sv < 0.f ? round_bf16(sv * ctl.rep) : round_bf16(sv / ctl.rep);
            }
        }
        sub.sync(lane);
        if (p.cb != 0) {        // the 16-frame window of one codebook: ids and values as eight 16-byte LDS reads, applied in order
            int ids[16]; float nvs[16];
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                const int4 a = *reinterpret_cast<const int4*>(S.pen_id + k);
                const float4 v = *reinterpret_cast<const float4*>(S.pen_val + k);
                ids[k] = a.x; ids[k + 1] = a.y; ids[k + 2] = a.z; ids[k + 3] = a.w;
                nvs[k] = v.x; nvs[k + 1] = v.y; nvs[k + 2] = v.z; nvs[k + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned d = (unsigned)(ids[k] - i0);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (d == (unsigned)e) l[e] = nvs[k];
            }
        } else {
            for (int k = 0; k < npen; ++k) {
                const int id = S.pen_id[k];
                if (id >= i0 && id < i0 + 4) {
                    const float nv = S.pen_val[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (i0 + e == id) l[e] = nv;
                }
            }
        }
    }
    if (p.cb == 0 && ctl.ban_eos && p.im_end < V) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i0 + e == p.im_end) l[e] = -INFINITY;
    }
    int bb_phase = 0;
    auto block_best = [&](ArgMax a) {       // the per-wave results alternate between two slot sets: one barrier per call
        a = wave_argmax(a);
        float* av = S.amv + 4 * (bb_phase & 1);
        int* ai = S.ami + 4 * (bb_phase & 1);
        ++bb_phase;
        if (lane == 0) { av[wave] = a.v; ai[wave] = a.i; }
        sub.sync(lane);
        ArgMax t{av[0], ai[0]};
#pragma unroll
        for (int w = 1; w < 4; ++w) t = better(t, ArgMax{av[w], ai[w]});
        return t;
    };
    ArgMax am{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (i0 + e < V) am = better(am, ArgMax{l[e], i0 + e});
    am = block_best(am);
    if (stp && tid == 0) stp[12] = eng_rt();
    const float Lmax = am.v;
    float ex[4], z = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ex[e] = (i0 + e) < V ? expf(l[e] - Lmax) : 0.f; z += ex[e]; }
    const float Z = red_sum(z);
    if (stp && tid == 0) stp[-16 + 10] = eng_rt();
    const float tp = round_bf16(ctl.top_p);
    auto removed = [&](float cum) { return round_bf16(cum) > tp; };
    constexpr uint32_t cmask = 0xffff0000u;
    uint32_t key[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool v = (i0 + e) < V;
        key[e] = v ? (order_key(l[e]) & cmask) : 0u;
        S.prL[i0 + e] = v ? round_bf16(ex[e] / Z) : 0.f;
        S.keyL[i0 + e] = key[e];
    }
    const bool only_top = removed(round_bf16(1.0f / Z));
    int winner = am.i;
    if (!only_top) {
        sub.sync(lane);
        const float Tc = fmaxf(ctl.temperature, 1e-5f);
        const float Mt = round_bf16(Lmax / Tc);
        float etp[4];           // expf of the tempered logits: needed for the kept ones, formed while wave 0 finds the cut
        if (wave != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) etp[e] = expf(round_bf16(l[e] / Tc) - Mt);
        }
        if (stp && tid == 0) stp[-16 + 11] = eng_rt();
        if (wave == 0) {
            float pr[16];
            uint32_t ky[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { pr[e] = S.prL[lane + 64 * e]; ky[e] = S.keyL[lane + 64 * e]; }
            float tot = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) tot += pr[e];
            tot = wave_sum(tot);
            if (stp && tid == 0) stp[-16 + 12] = eng_rt();
            uint32_t kstar = 0;
            int nk = 0, all_kept = 0;
            if (!removed(tot)) {
                all_kept = 1;
            } else {
                // the highest key present: a candidate above it selects nothing, its mass is an exact 0 and removed(0) is
                // false whatever top_p is - those steps of the search (about half of the 16) are decided without the sum
                uint32_t kmx = 0u;
#pragma unroll
                for (int e = 0; e < 16; ++e) kmx = ky[e] > kmx ? ky[e] : kmx;
                {
                    int km = (int)(kmx >> 16);      // keys carry 16 significant bits: compare as small non-negative ints
                    km = max(km, dpp_i<DPP_XOR1>(km));
                    km = max(km, dpp_i<DPP_XOR2>(km));
                    km = max(km, dpp_i<DPP_HALF_MIRROR>(km));
                    km = max(km, dpp_i<DPP_MIRROR>(km));
                    km = max(max(__builtin_amdgcn_readlane(km, 0), __builtin_amdgcn_readlane(km, 16)),
                             max(__builtin_amdgcn_readlane(km, 32), __builtin_amdgcn_readlane(km, 48)));
                    kmx = (uint32_t)km << 16;
                }
                // (measured and not kept: two bits per step, the three candidates' masses summed side by side - the step is
                // bound by its ~65 select / add instructions per candidate, not by the dependent sums: 3.7 against 3.4 us)
                for (int bit = 31; bit >= 16; --bit) {
                    const uint32_t cand = kstar | (1u << bit);
                    if (cand > kmx) continue;
                    float ms = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ms += ky[e] >= cand ? pr[e] : 0.f;
                    if (removed(wave_sum(ms))) kstar = cand;
                }
                if (stp && tid == 0) stp[-16 + 13] = eng_rt();
                float above = 0.f, cnt = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (ky[e] > kstar) above += pr[e];
                    else if (ky[e] == kstar && (lane + 64 * e) < V) cnt += 1.f;
                }
                above = wave_sum(above);
                const int icnt = (int)wave_sum(cnt);
                const uint32_t ubits = (kstar & 0x80000000u) ? (kstar & 0x7fffffffu) : ~(kstar | ~cmask);
                const float pk = round_bf16(expf(__uint_as_float(ubits) - Lmax) / Z);
                int lo_n = 0, hi_n = icnt;
                while (lo_n < hi_n) {
                    const int mid = (lo_n + hi_n + 1) >> 1;
                    if (removed(fmaf((float)mid, pk, above))) hi_n = mid - 1; else lo_n = mid;
                }
                nk = lo_n;
            }
            if (stp && tid == 0) stp[-16 + 14] = eng_rt();
            if (lane == 0) { S.cut[0] = kstar; S.cut[1] = (uint32_t)nk; S.cut[2] = (uint32_t)all_kept; }
        }
        sub.sync(lane);
        if (stp && tid == 0) stp[13] = eng_rt();
        const uint32_t kstar = S.cut[0];
        const int nk = (int)S.cut[1];
        const bool all_kept = S.cut[2] != 0;
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) etp[e] = expf(round_bf16(l[e] / Tc) - Mt);
        }
        bool mem[4];
        int mine = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mem[e] = (i0 + e) < V && !all_kept && key[e] == kstar; mine += mem[e] ? 1 : 0; }
        int below = 0, wtot = 0;
        const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
        for (int c = 1; c <= 4; ++c) {
            const unsigned long long bal = __ballot(mine >= c);
            below += __popcll(bal & lower);
            wtot += __popcll(bal);
        }
        if (lane == 0) S.wcnt[wave] = wtot;
        sub.sync(lane);
        int rank = below;
        for (int w = 0; w < wave; ++w) rank += S.wcnt[w];
        bool keep[4];
        float et[4], z2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            keep[e] = (i0 + e) < V && (all_kept || key[e] > kstar || (mem[e] && rank < nk));
            rank += mem[e] ? 1 : 0;
            et[e] = keep[e] ? etp[e] : 0.f;
            z2 += et[e];
        }
        const float Z2 = red_sum(z2);
        ArgMax best{-1.f, 0x7fffffff};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i0 + e < V) {
                const float prob = keep[e] ? round_bf16(et[e] / Z2) : 0.f;
                best = better(best, ArgMax{round_bf16(prob / round_bf16(pre.q4[e])), i0 + e});
            }
        }
        winner = block_best(best).i;
    }
    return winner;      // every thread of the four waves holds it (no barrier: the LDS scratch is next touched a step later)
}

// q k v of fast layer 0 for every row of the codebook-embedding table, with the fast engine's own phase code (same
// loads, eng_gemv_rows, same rounding): grid (qkvN / (S::SQ * ENG_CW), code chunks), 256 threads.
template <typename S>
__global__ __launch_bounds__(ENG_CW * 64) void eng_qkv0_table_kernel(const bf16_t* wqkv, const bf16_t* bqkv, const bf16_t* attn_norm,
                                                                      const bf16_t* fast_emb, bf16_t* tab, int D, int qkvN, int ncodes,
                                                                      float eps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;            // [D]
    float* vals = xs + D;        // [ENG_MAX_OUT]
    const int tid = threadIdx.x, lane = tid & 63, cw = tid >> 6;
    int q_lo, q_hi;
    eng_units(qkvN, blockIdx.x, gridDim.x, q_lo, q_hi);
    constexpr int NTD = S::NTD;
    EngW<NTD, 1, S::SQ> wq;
    eng_issue<false>(wq, wqkv, attn_norm, D, q_lo, q_hi, cw, lane, 0);
    const int per = (ncodes + gridDim.y - 1) / gridDim.y;
    const int c_lo = blockIdx.y * per, c_hi = min(c_lo + per, ncodes);
    for (int code = c_lo; code < c_hi; ++code) {
        for (int d = tid * 8; d < D; d += ENG_CW * 64 * 8) {
            float e8[8];
            Vec<bf16_t>::unpack(eng_ldg16<false>(fast_emb + (size_t)code * D + d), e8);
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[d + j] = e8[j];
        }
        __syncthreads();
        eng_gemv_rows<NTD, 1, S::SQ, PRO_RMSNORM, EPI_STORE>(wq, xs, D, eps, bqkv, nullptr, vals, q_lo, q_hi, cw, lane);
        __syncthreads();
        if (tid < q_hi - q_lo) tab[(size_t)code * qkvN + q_lo + tid] = f32_to_bf16_bits(vals[tid]);
        __syncthreads();
    }
}

// dynamic LDS of fast_engine_kernel (host and harness use this one formula)
inline size_t eng_fast_lds_bytes(int D, int qkvN, int HD, int F, int V, int nL, int ncb, int KVW, bool pair) {
    size_t fl = (size_t)D * 2 + qkvN + HD + F + V + ENG_MAX_OUT + 8 + 32 + 32 + 20 + 2048 + 4 + 4 + 16;
    size_t by = fl * sizeof(float) + (size_t)nL * 2 * ncb * KVW * 2 + 64;
    if (pair) by += ((size_t)2 * D + (size_t)(F > qkvN + HD ? F : qkvN + HD)) * sizeof(float) + 16;
    return by;
}

template <typename FS, int MAXCB>
__global__ __launch_bounds__(ENG_THREADS) void fast_engine_kernel(FastEngP p) {
    // units per compute wave (s1-mini: QKV 8 rows -> 2, W13 12 pairs -> 3, Wo / W2 / head 4 rows -> 1)
    constexpr int NTD = FS::NTD, NTA = FS::NTA, NTF = FS::NTF, HDIM = FS::HDIM;
    constexpr int SQ = FS::SQ, SF = FS::SF, SO = FS::SO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: uniform branches, row offsets in SGPRs)
    const int b = blockIdx.x;
    constexpr int nb = ENG_NB;            // workgroups = CUs (the host launches exactly this many, one per CU)
    const int cw = wave, gw = wave - ENG_CW, atid = tid - ENG_CW * 64;
    // compile-time widths, as in slow_engine_kernel (the host's gate admits exactly these)
    constexpr int D = FS::D, F = FS::F, hd = HDIM, H = FS::H, Hkv = FS::HKV, HD = H * hd, KVW = Hkv * hd, QKVN = (H + 2 * Hkv) * hd, V = FS::V;
    float* xA = smem;                          // layer input
    float* qkvS = xA + D;                      // gathered q, k, v of this position
    float* yS = qkvS + QKVN;                 // attention output
    float* xB = yS + HD;                       // x' = x + Wo y
    float* gS = xB + D;                        // SwiGLU output
    float* logS = gS + F;                      // [V] logits (drawing workgroup)
    float* outS = logS + V;                  // [ENG_MAX_OUT]
    float* redbuf = outS + ENG_MAX_OUT;        // sampling scratch ...
    int* pen_id = reinterpret_cast<int*>(redbuf + 8);
    float* pen_val = reinterpret_cast<float*>(pen_id + 32);
    float* amv = pen_val + 32;                 // [2][4]
    int* ami = reinterpret_cast<int*>(amv + 8);   // [2][4]
    int* wcnt = ami + 8;
    float* prL = reinterpret_cast<float*>(wcnt + 4);
    uint32_t* keyL = reinterpret_cast<uint32_t*>(prL + 1024);
    uint32_t* cut = keyL + 1024;
    int* dead = reinterpret_cast<int*>(cut + 4);
    int* out_count = dead + 1;
    int* sub_count = dead + 2;
    int* codes_s = dead + 4;                   // [MAXCB] codes of this frame as they become known
    int* reg_s = codes_s + MAXCB;              // [4] XCD, rank in it, workgroups in it
    bf16_t* kvS = reinterpret_cast<bf16_t*>(reg_s + 4);    // [n_layer][2][ncb][Hkv * hd] bf16 bits
    // second row of the paired first pass: x, x' and ONE region that holds (q k v | y) until Wo has run and g afterwards
    float* xA1 = smem + ((((size_t)(reinterpret_cast<float*>(kvS + (size_t)p.n_layer * 2 * p.ncb * KVW) - smem)) + 3) & ~(size_t)3);   // 16-byte aligned
    float* xB1 = xA1 + D;
    float* gS1 = xB1 + D;
    float* qkvS1 = gS1;
    float* yS1 = gS1 + QKVN;
    if (tid == 0) { *dead = eng_fault_here(p.ctl, ENG_FAULT_FAST, b) ? 1 : 0; *out_count = 0; *sub_count = 0; }
    if (tid == ENG_CW * 64) {                  // one thread owns the registration words
        reg_s[0] = 0; reg_s[1] = 0; reg_s[2] = 1;
        if (p.rep_stride) eng_register(p.ctl, nb, reg_s, dead);
    }
    const unsigned epoch = __hip_atomic_load((eng_gu32*)(p.ctl + ENG_CTL_EPOCH), ENG_RLX);
    const size_t VSTR = (size_t)nb * ENG_LINE;
    const EngLayout layD{D / nb}, layF{F / nb}, layQ{QKVN / nb}, layV{V / nb};
    const int nL = p.n_layer;
    // hand-off buffers of (step parity, layer)
    auto bx = [&](int par, int l) { return p.gx + ((size_t)par * (nL + 1) + l) * VSTR; };
    auto bq = [&](int par, int l) { return p.gqkv + ((size_t)par * nL + l) * VSTR; };
    auto bxb = [&](int par, int l) { return p.gxb + ((size_t)par * nL + l) * VSTR; };
    auto bg = [&](int par, int l) { return p.gg + ((size_t)par * nL + l) * VSTR; };
    auto blog = [&](int par) { return p.glog + (size_t)par * VSTR; };
    auto drawer = [&](int cb) { return (cb * 37) % nb; };   // the workgroup that draws codebook cb
    bool alive = true;
    const bool pair = p.pair != 0 && p.ncb >= 2;

    if (wave < ENG_CW) {
        // =============================== compute waves ===============================
        int q_lo, q_hi, o_lo, o_hi, f_lo, f_hi, h_lo, h_hi;
        eng_units(QKVN, b, nb, q_lo, q_hi);
        eng_units(D, b, nb, o_lo, o_hi);
        eng_units(F, b, nb, f_lo, f_hi);
        eng_units(V, b, nb, h_lo, h_hi);
        EngW<NTD, 1, SQ> wq;
        EngW<NTA, 1, SO> wo;
        EngW<NTD, 2, SF> wf;
        EngW<NTF, 1, SO> wd;
        EngW<NTD, 1, SO> wh;
        EngOut eo{outS, out_count, 0};
        {
            const EngLayer l0 = eng_layer(p.layers, 0);
            eng_issue<false>(wq, l0.wqkv, l0.attn_norm, D, q_lo, q_hi, cw, lane, 0);
            eng_issue<false>(wo, l0.wo, (const bf16_t*)nullptr, HD, o_lo, o_hi, cw, lane, 0);
            eng_issue<false>(wf, l0.w13, l0.ffn_norm, D, f_lo, f_hi, cw, lane, 0);
            eng_issue<false>(wd, l0.w2, (const bf16_t*)nullptr, F, o_lo, o_hi, cw, lane, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        eng_barrier();                                                  // (registration results in LDS)
        eng_barrier();                                                  // B0
        if (pair) {
            // ---- codebook positions 0 and 1 as two rows of one pass: row 0 = (parity 0, tag of step 0), row 1 = (parity 1,
            // tag of step 1); per row the arithmetic of the single-row pass.  Row 0 only feeds the K/V history, so its
            // Wo / W13 / W2 phases of the last layer (and its logits, inference.py:122) are not computed.
            const unsigned tag0 = eng_tag16(epoch), tag1 = eng_tag16(epoch + 1u);
            float r0c[2], r0s[2], r1c[2], r1s[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                r0c[e] = d < hd ? p.rope[((size_t)(d >> 1)) * 2] : 1.f;
                r0s[e] = d < hd ? p.rope[((size_t)(d >> 1)) * 2 + 1] : 0.f;
                r1c[e] = d < hd ? p.rope[((size_t)(hd >> 1) + (d >> 1)) * 2] : 1.f;
                r1s[e] = d < hd ? p.rope[((size_t)(hd >> 1) + (d >> 1)) * 2 + 1] : 0.f;
            }
            for (int li = 0; li < nL; ++li) {
                const int cb = 1;        // (for the stamp macro)
                // buffer bases re-read per layer: keeps the compiler from hoisting ~20 per-lane publish addresses out of the
                // loop (they would not fit beside the weight registers and spill)
                unsigned *gxL = p.gx, *gqL = p.gqkv, *gxbL = p.gxb, *ggL = p.gg;
                asm volatile("" : "+s"(gxL), "+s"(gqL), "+s"(gxbL), "+s"(ggL));
                auto bx = [&](int par, int l_) { return gxL + ((size_t)par * (nL + 1) + l_) * VSTR; };
                auto bq = [&](int par, int l_) { return gqL + ((size_t)par * nL + l_) * VSTR; };
                auto bxb = [&](int par, int l_) { return gxbL + ((size_t)par * nL + l_) * VSTR; };
                auto bg = [&](int par, int l_) { return ggL + ((size_t)par * nL + l_) * VSTR; };
                const EngLayer l = eng_layer(p.layers, li);
                const bool more = !(p.ncb == 2 && li == nL - 1);
                const bool tail0 = li + 1 < nL;                         // row 0 goes on after the attention
                const EngLayer ln = eng_layer(p.layers, li + 1 < nL ? li + 1 : 0);
                bf16_t* kL = kvS + (size_t)(li * 2) * p.ncb * KVW;
                bf16_t* vL = kvS + (size_t)(li * 2 + 1) * p.ncb * KVW;
                eng_barrier(); if (*dead) { alive = false; break; }     // B1: xA, xA1
                ENG_FSTAMP(0);
                eng_gemv2<NTD, 1, SQ, PRO_RMSNORM, EPI_STORE>(wq, xA, xA1, D, p.eps, l.bqkv, nullptr, nullptr, bq(0, li) + eng_pub(b, q_lo),
                                                              bq(1, li) + eng_pub(b, q_lo), tag0, tag1, q_lo, q_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(1);
                if (more && !(li == nL - 1 && p.qkv0_tab != nullptr)) eng_issue<false>(wq, ln.wqkv, ln.attn_norm, D, q_lo, q_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B1b: qkvS, qkvS1
                ENG_FSTAMP(2);
                eng_fast_attn_any<MAXCB, HDIM>(qkvS, kL, vL, yS, l.qn, l.kn, r0c, r0s, 0, p.ncb, H, Hkv, p.eps, p.scale, wave, ENG_WAVES, lane);
                eng_barrier();                                          // position 0's K/V rows in LDS
                eng_fast_attn_any<MAXCB, HDIM>(qkvS1, kL, vL, yS1, l.qn, l.kn, r1c, r1s, 1, p.ncb, H, Hkv, p.eps, p.scale, wave, ENG_WAVES, lane);
                ENG_FSTAMP(9);
                eng_barrier();                                          // B2: yS, yS1
                ENG_FSTAMP(3);
                if (tail0)
                    eng_gemv2<NTA, 1, SO, PRO_NONE, EPI_RESID>(wo, yS, yS1, HD, p.eps, l.bo, xA, xA1, bxb(0, li) + eng_pub(b, o_lo),
                                                               bxb(1, li) + eng_pub(b, o_lo), tag0, tag1, o_lo, o_hi, cw, lane, eo);
                else
                    eng_gemv<NTA, 1, SO, PRO_NONE, EPI_RESID>(wo, yS1, HD, p.eps, l.bo, xA1, bxb(1, li) + eng_pub(b, o_lo), tag1, nullptr,
                                                              o_lo, o_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(4);
                if (more) eng_issue<false>(wo, ln.wo, (const bf16_t*)nullptr, HD, o_lo, o_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B3: xB, xB1
                ENG_FSTAMP(5);
                if (tail0)
                    eng_gemv2<NTD, 2, SF, PRO_RMSNORM, EPI_SWIGLU>(wf, xB, xB1, D, p.eps, nullptr, nullptr, nullptr, bg(0, li) + eng_pub(b, f_lo),
                                                                   bg(1, li) + eng_pub(b, f_lo), tag0, tag1, f_lo, f_hi, cw, lane, eo);
                else
                    eng_gemv<NTD, 2, SF, PRO_RMSNORM, EPI_SWIGLU, FS::PK3>(wf, xB1, D, p.eps, nullptr, nullptr, bg(1, li) + (FS::PK3 ? 0 : eng_pub(b, f_lo)), tag1, nullptr,
                                                                    f_lo, f_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(6);
                if (more) eng_issue<false>(wf, ln.w13, ln.ffn_norm, D, f_lo, f_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B4: gS, gS1
                ENG_FSTAMP(7);
                if (tail0)
                    eng_gemv2<NTF, 1, SO, PRO_NONE, EPI_RESID>(wd, gS, gS1, F, p.eps, nullptr, xB, xB1, bx(0, li + 1) + eng_pub(b, o_lo),
                                                               bx(1, li + 1) + eng_pub(b, o_lo), tag0, tag1, o_lo, o_hi, cw, lane, eo);
                else
                    eng_gemv<NTF, 1, SO, PRO_NONE, EPI_RESID>(wd, gS1, F, p.eps, nullptr, xB1, bx(1, li + 1) + eng_pub(b, o_lo), tag1, nullptr,
                                                              o_lo, o_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(8);
                if (more) eng_issue<false>(wd, ln.w2, (const bf16_t*)nullptr, F, o_lo, o_hi, cw, lane, 0);
                if (li == nL - 1) eng_issue<false>(wh, p.fast_out, p.fast_norm, D, h_lo, h_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (alive) {
                eng_barrier();                                          // B5: xA = stack output of position 1
                if (*dead) alive = false;
                else {
                    eng_gemv<NTD, 1, SO, PRO_RMSNORM, EPI_STORE>(wh, xA, D, p.eps, nullptr, nullptr, blog(1) + eng_pub(b, h_lo), tag1, nullptr,
                                                                 h_lo, h_hi, cw, lane, eo);
                    __builtin_amdgcn_sched_barrier(0);
                    eng_draw_follow(sub_count);
                }
            }
        }
        for (int cb = pair ? 2 : 0; cb < p.ncb && alive; ++cb) {
            const int par = cb & 1;
            const unsigned tag = eng_tag16(epoch + (unsigned)cb);
            float rcs[2], rsn[2];      // rotation entries of this codebook position (dimensions lane, lane + 64)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                rcs[e] = d < hd ? p.rope[((size_t)cb * (hd >> 1) + (d >> 1)) * 2] : 1.f;
                rsn[e] = d < hd ? p.rope[((size_t)cb * (hd >> 1) + (d >> 1)) * 2 + 1] : 0.f;
            }
            const bool tab0 = p.qkv0_tab != nullptr && cb >= 2;
            for (int li = 0; li < nL; ++li) {
                const EngLayer l = eng_layer(p.layers, li);
                const bool more = !(cb == p.ncb - 1 && li == nL - 1);
                const EngLayer ln = eng_layer(p.layers, li + 1 < nL ? li + 1 : 0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B1: xA
                ENG_FSTAMP(0);
                if (!(li == 0 && tab0))      // (table steps: q k v of layer 0 were looked up by the gathering waves)
                    eng_gemv<NTD, 1, SQ, PRO_RMSNORM, EPI_STORE>(wq, xA, D, p.eps, l.bqkv, nullptr, bq(par, li) + eng_pub(b, q_lo), tag,
                                                                 nullptr, q_lo, q_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(1);
                if (more && !(li == nL - 1 && p.qkv0_tab != nullptr && cb + 1 >= 2))
                    eng_issue<false>(wq, ln.wqkv, ln.attn_norm, D, q_lo, q_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B1b: qkvS
                ENG_FSTAMP(2);
                eng_fast_attn_any<MAXCB, HDIM>(qkvS, kvS + (size_t)(li * 2) * p.ncb * KVW, kvS + (size_t)(li * 2 + 1) * p.ncb * KVW, yS, l.qn, l.kn,
                                         rcs, rsn, cb, p.ncb, H, Hkv, p.eps, p.scale, wave, ENG_WAVES, lane);
                ENG_FSTAMP(9);
                eng_barrier();                                          // B2: yS
                ENG_FSTAMP(3);
                eng_gemv<NTA, 1, SO, PRO_NONE, EPI_RESID>(wo, yS, HD, p.eps, l.bo, xA, bxb(par, li) + eng_pub(b, o_lo), tag, nullptr,
                                                          o_lo, o_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(4);
                if (more) eng_issue<false>(wo, ln.wo, (const bf16_t*)nullptr, HD, o_lo, o_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B3: xB
                ENG_FSTAMP(5);
                eng_gemv<NTD, 2, SF, PRO_RMSNORM, EPI_SWIGLU, FS::PK3>(wf, xB, D, p.eps, nullptr, nullptr, bg(par, li) + (FS::PK3 ? 0 : eng_pub(b, f_lo)), tag, nullptr,
                                                                f_lo, f_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(6);
                if (more) eng_issue<false>(wf, ln.w13, ln.ffn_norm, D, f_lo, f_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
                eng_barrier(); if (*dead) { alive = false; break; }     // B4: gS
                ENG_FSTAMP(7);
                eng_gemv<NTF, 1, SO, PRO_NONE, EPI_RESID>(wd, gS, F, p.eps, nullptr, xB, bx(par, li + 1) + eng_pub(b, o_lo), tag, nullptr,
                                                          o_lo, o_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                ENG_FSTAMP(8);
                if (more) eng_issue<false>(wd, ln.w2, (const bf16_t*)nullptr, F, o_lo, o_hi, cw, lane, 0);
                // the head's rows are requested one hand-off before their use (held only across it)
                if (li == nL - 1 && cb >= 1) eng_issue<false>(wh, p.fast_out, p.fast_norm, D, h_lo, h_hi, cw, lane, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!alive) break;
            if (cb >= 1) {      // logits of position 0 are discarded (inference.py:122)
                eng_barrier(); if (*dead) { alive = false; break; }     // B5: xA = stack output
                eng_gemv<NTD, 1, SO, PRO_RMSNORM, EPI_STORE>(wh, xA, D, p.eps, nullptr, nullptr, blog(par) + eng_pub(b, h_lo), tag, nullptr,
                                                             h_lo, h_hi, cw, lane, eo);
                __builtin_amdgcn_sched_barrier(0);
                eng_draw_follow(sub_count);
            }
        }
    } else {
        // ====================== gathering waves: inputs, attention share, the draws ======================
        EngSub sub{sub_count, 0};
        SampP sp = p.samp;
        int prev_code = 0;
        eng_barrier();                                                  // (registration results in LDS)
        const EngRelay rl{p.rep_stride != 0, reg_s[1], reg_s[2], p.rep_delta0 + (long)reg_s[0] * p.rep_stride};
        eng_barrier();                                                  // B0
        if (pair) {
            const unsigned tag0 = eng_tag16(epoch), tag1 = eng_tag16(epoch + 1u);
            float r0c[2], r0s[2], r1c[2], r1s[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                r0c[e] = d < hd ? p.rope[((size_t)(d >> 1)) * 2] : 1.f;
                r0s[e] = d < hd ? p.rope[((size_t)(d >> 1)) * 2 + 1] : 0.f;
                r1c[e] = d < hd ? p.rope[((size_t)(hd >> 1) + (d >> 1)) * 2] : 1.f;
                r1s[e] = d < hd ? p.rope[((size_t)(hd >> 1) + (d >> 1)) * 2 + 1] : 0.f;
            }
            for (int li = 0; li < nL; ++li) {
                const EngLayer l = eng_layer(p.layers, li);
                const bool tail0 = li + 1 < nL;
                const int wh = 1000 + 64 + li * 8;
                bf16_t* kL = kvS + (size_t)(li * 2) * p.ncb * KVW;
                bf16_t* vL = kvS + (size_t)(li * 2 + 1) * p.ncb * KVW;
                if (li > 0) {
                    eng_gather_x2(rl, EngSrc2{{bx(0, li), bx(1, li)}, {xA, xA1}, {tag0, tag1}}, D, gw, ENG_GW, lane, p.ctl, dead, wh + 0);
                } else {
                    for (int d = atid * 4; d < D; d += ENG_GW * 64 * 4) {
                        *reinterpret_cast<float4*>(xA + d) = *reinterpret_cast<const float4*>(p.hid + d);
                        *reinterpret_cast<float4*>(xA1 + d) = *reinterpret_cast<const float4*>(p.femb + d);
                    }
                }
                eng_barrier(); if (*dead) { alive = false; break; }     // B1
                eng_gather_x2(rl, EngSrc2{{bq(0, li), bq(1, li)}, {qkvS, qkvS1}, {tag0, tag1}}, QKVN, gw, ENG_GW, lane, p.ctl, dead, wh + 1);
                eng_barrier(); if (*dead) { alive = false; break; }     // B1b
                eng_fast_attn_any<MAXCB, HDIM>(qkvS, kL, vL, yS, l.qn, l.kn, r0c, r0s, 0, p.ncb, H, Hkv, p.eps, p.scale, wave, ENG_WAVES, lane);
                eng_barrier();
                eng_fast_attn_any<MAXCB, HDIM>(qkvS1, kL, vL, yS1, l.qn, l.kn, r1c, r1s, 1, p.ncb, H, Hkv, p.eps, p.scale, wave, ENG_WAVES, lane);
                eng_barrier();                                          // B2
                if (tail0) eng_gather_x2(rl, EngSrc2{{bxb(0, li), bxb(1, li)}, {xB, xB1}, {tag0, tag1}}, D, gw, ENG_GW, lane, p.ctl, dead, wh + 2);
                else eng_gather_x(rl, bxb(1, li), layD, 0, D, tag1, xB1, gw, ENG_GW, lane, p.ctl, dead, wh + 2);
                eng_barrier(); if (*dead) { alive = false; break; }     // B3
                if (tail0) eng_gather_x2(rl, EngSrc2{{bg(0, li), bg(1, li)}, {gS, gS1}, {tag0, tag1}}, F, gw, ENG_GW, lane, p.ctl, dead, wh + 3);
                else if constexpr (FS::PK3) eng_gather3_x(rl, bg(1, li), F, tag1, gS1, gw, ENG_GW, lane, p.ctl, dead, wh + 3);
                else eng_gather_x(rl, bg(1, li), layF, 0, F, tag1, gS1, gw, ENG_GW, lane, p.ctl, dead, wh + 3);
                eng_barrier(); if (*dead) { alive = false; break; }     // B4
            }
        }
        for (int cb = pair ? 1 : 0; cb < p.ncb && alive; ++cb) {
            const int par = cb & 1;
            const unsigned tag = eng_tag16(epoch + (unsigned)cb);
            float rcs[2], rsn[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                rcs[e] = d < hd ? p.rope[((size_t)cb * (hd >> 1) + (d >> 1)) * 2] : 1.f;
                rsn[e] = d < hd ? p.rope[((size_t)cb * (hd >> 1) + (d >> 1)) * 2 + 1] : 0.f;
            }
            EngDrawPre pre{};
            if (cb >= 1) {
                sp.cb = cb;
                sp.noise_off = p.noise_off1 + (long)(cb - 1) * p.noise_cb_stride;
                pre = eng_draw_pre(sp, atid);
            }
            for (int li = (pair && cb == 1) ? nL : 0; li < nL; ++li) {
                const EngLayer l = eng_layer(p.layers, li);
                if (li > 0) {
                    eng_gather_x(rl, bx(par, li), layD, 0, D, tag, xA, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + li * 8 + 0);
                } else if (cb <= 1) {
                    const float* src = cb == 0 ? p.hid : p.femb;        // plain f32 left by the launches before this one
                    for (int d = atid * 4; d < D; d += ENG_GW * 64 * 4) *reinterpret_cast<float4*>(xA + d) = *reinterpret_cast<const float4*>(src + d);
                } else {
                    // the code this workgroup drew in the previous step: that row of the codebook-embedding table
                    const int code = prev_code;
                    const int d0 = atid * 8;
                    if (p.qkv0_tab && D <= ENG_GW * 64 * 8 && QKVN <= ENG_GW * 64 * 8) {
                        // the embedding row and layer 0's q k v of that row (a lookup instead of a phase and a hand-off):
                        // both loads in flight together
                        const U4 ue = eng_ldg16<false>(p.fast_emb + (size_t)code * D + (d0 < D ? d0 : 0));
                        const U4 uq = eng_ldg16<false>(p.qkv0_tab + (size_t)code * QKVN + (d0 < QKVN ? d0 : 0));
                        float e8[8], q8[8];
                        Vec<bf16_t>::unpack(ue, e8);
                        Vec<bf16_t>::unpack(uq, q8);
                        if (d0 < D) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) xA[d0 + j] = e8[j];
                        }
                        if (d0 < QKVN) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) qkvS[d0 + j] = q8[j];
                        }
                    } else {
                        for (int d = d0; d < D; d += ENG_GW * 64 * 8) {
                            float e8[8];
                            Vec<bf16_t>::unpack(eng_ldg16<false>(p.fast_emb + (size_t)code * D + d), e8);
#pragma unroll
                            for (int j = 0; j < 8; ++j) xA[d + j] = e8[j];
                        }
                        if (p.qkv0_tab) {
                            for (int d = d0; d < QKVN; d += ENG_GW * 64 * 8) {
                                float e8[8];
                                Vec<bf16_t>::unpack(eng_ldg16<false>(p.qkv0_tab + (size_t)code * QKVN + d), e8);
#pragma unroll
                                for (int j = 0; j < 8; ++j) qkvS[d + j] = e8[j];
                            }
                        }
                    }
                }
                eng_barrier(); if (*dead) { alive = false; break; }     // B1
                if (!(li == 0 && cb >= 2 && p.qkv0_tab))
                    eng_gather_x(rl, bq(par, li), layQ, 0, QKVN, tag, qkvS, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + li * 8 + 1);
                eng_barrier(); if (*dead) { alive = false; break; }     // B1b
                eng_fast_attn_any<MAXCB, HDIM>(qkvS, kvS + (size_t)(li * 2) * p.ncb * KVW, kvS + (size_t)(li * 2 + 1) * p.ncb * KVW, yS, l.qn, l.kn,
                                         rcs, rsn, cb, p.ncb, H, Hkv, p.eps, p.scale, wave, ENG_WAVES, lane);
                eng_barrier();                                          // B2
                eng_gather_x(rl, bxb(par, li), layD, 0, D, tag, xB, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + li * 8 + 2);
                eng_barrier(); if (*dead) { alive = false; break; }     // B3
                if constexpr (FS::PK3) eng_gather3_x(rl, bg(par, li), F, tag, gS, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + li * 8 + 3);
                else eng_gather_x(rl, bg(par, li), layF, 0, F, tag, gS, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + li * 8 + 3);
                eng_barrier(); if (*dead) { alive = false; break; }     // B4
            }
            if (!alive) break;
            if (cb >= 1) {
                eng_gather_x(rl, bx(par, nL), layD, 0, D, tag, xA, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + 56);
                eng_barrier(); if (*dead) { alive = false; break; }     // B5
                ENG_GSTAMP(10);
                {
                    // ---- the draw of codebook cb (inference.py:134-149).  EVERY workgroup gathers the logits and draws (the
                    // draw is a deterministic function of logits, frame and seed): no hand-off of the code, the next step's
                    // embedding row can be fetched at once.  Workgroup drawer(cb) alone does the frame bookkeeping (finish_draw).
                    eng_gather_x(rl, blog(par), layV, 0, V, tag, logS, gw, ENG_GW, lane, p.ctl, dead, 1000 + cb * 64 + 57);
                    sub.n = 0;
                    if (atid == 0) *sub_count = *dead ? 1 : 0;       // (a workgroup that gave up draws nothing: one barrier, then on)
                    sub.sync(lane);
                    ENG_GSTAMP(11);
                    if (*reinterpret_cast<volatile int*>(sub_count) == 0) {
                        const int last = cb == p.ncb - 1;
                        const int nfv = pre.nfv;
                        EngSampLds S{redbuf, pen_id, pen_val, amv, ami, wcnt, prL, keyL, cut};
                        const int code = eng_sample_small(sp, pre, logS, S, sub, atid, lane, gw,
                                                          p.stamps ? p.stamps + (((size_t)b * p.ncb + cb) * nL + nL - 1) * 16 : nullptr);
                        ENG_GSTAMP(14);
                        const int R = p.ncb + 1;
                        if (atid == 0) { codes_s[cb] = code; *sub_count = sub.n + 1; }
                        sub.sync(lane);      // the draw's last barrier: the compute waves stop following
                        prev_code = code;    // (every thread holds the drawn code: the next step's embedding row is fetched from it)
                        if (b == drawer(cb)) {
                            if (atid == 0) sp.tokn[cb + 1] = code;
                            if (last) {
                                const int frozen = sp.done[0];
                                if (atid < R) {
                                    const int v = atid < 2 ? sp.tokn[atid] : (atid - 1 == cb ? code : codes_s[atid - 1]);   // rows 0, 1: the slow draw's launch
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
                    }
                }
            }
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
