// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace ft {

// hipFuncSetAttribute applies to the CURRENT device's function object, and a process may hold contexts on several
// GPUs (FishTTS(gpu_index=...)) driven from several threads: remember per device which opt-ins were made.  A thread
// publishes its device's bit only after it has set the attribute itself (two threads may both set it: harmless).
struct DevOnce {
    std::atomic<unsigned long long> mask{0};
    template <typename F> void run(F&& set) {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (mask.load(std::memory_order_acquire) & bit) return;
        set();
        mask.fetch_or(bit, std::memory_order_release);
    }
};

typedef uint16_t bf16_t;  // raw bf16 bits

typedef uint32_t U4 __attribute__((ext_vector_type(4)));  // one 16-byte load

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// round-to-nearest-even f32 -> bf16 precision, result kept in an f32 register.  The plain cast lowers to
// v_cvt_pk_bf16_f32 on gfx950 (2 instructions with the shift back, NaN stays NaN) instead of the 6 of
// the integer formulation.
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ bf16_t f32_to_bf16_bits(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(bf16_t, h);
}

// precision="fp16" (synthesizer.py:125-126): IEEE half storage, its own type so the templates can tell it from bf16 bits
struct f16_t { uint16_t bits; };
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
__device__ __forceinline__ float round_f16(float x) { return (float)(_Float16)x; }     // RNE, overflow -> inf as torch's cast
__device__ __forceinline__ uint16_t f32_to_f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    return __builtin_bit_cast(uint16_t, h);
}
// the model precision's rounding: ROUND = 0 none (f32), 1 bf16, 2 fp16 (a bool true converts to 1)
enum { RND_NONE = 0, RND_BF16 = 1, RND_F16 = 2 };
template <int ROUND> __device__ __forceinline__ float rb(float x) { return ROUND == 1 ? round_bf16(x) : ROUND == 2 ? round_f16(x) : x; }

// element load/store for the storage types the engine supports (bf16 bits, fp16, f32)
__device__ __forceinline__ float ld_elem(const bf16_t* p, size_t i) { return bf16_bits_to_f32(p[i]); }
__device__ __forceinline__ float ld_elem(const f16_t* p, size_t i) { return f16_bits_to_f32(p[i].bits); }
__device__ __forceinline__ float ld_elem(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ void st_elem(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16_bits(v); }
__device__ __forceinline__ void st_elem(f16_t* p, size_t i, float v) { p[i].bits = f32_to_f16_bits(v); }
__device__ __forceinline__ void st_elem(float* p, size_t i, float v) { p[i] = v; }

// 16-byte vector of weights -> VEC f32 values (VEC = 8 for bf16, 4 for f32)
template <typename T> struct Vec;
// Raw / load_raw / unpack_raw: EIGHT consecutive elements kept in their storage form until they are used (K/V rows in flight)
struct F8Raw { float4 a, b; };
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    typedef U4 Raw;
    __device__ static __forceinline__ U4 load_raw(const bf16_t* p) { return *reinterpret_cast<const U4*>(p); }
    __device__ static __forceinline__ void unpack_raw(const U4& r, float (&v)[8]) { unpack(r, v); }
    __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const U4 r = *reinterpret_cast<const U4*>(p);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
        v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    __device__ static __forceinline__ void unpack(const U4& r, float (&v)[8]) {
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
        v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    __device__ static __forceinline__ void zero(float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
};
template <> struct Vec<f16_t> {
    static constexpr int N = 8;
    typedef U4 Raw;
    __device__ static __forceinline__ U4 load_raw(const f16_t* p) { return *reinterpret_cast<const U4*>(p); }
    __device__ static __forceinline__ void unpack_raw(const U4& r, float (&v)[8]) { unpack(r, v); }
    __device__ static __forceinline__ void unpack(const U4& r, float (&v)[8]) {
        v[0] = f16_bits_to_f32(r.x & 0xffffu); v[1] = f16_bits_to_f32(r.x >> 16);
        v[2] = f16_bits_to_f32(r.y & 0xffffu); v[3] = f16_bits_to_f32(r.y >> 16);
        v[4] = f16_bits_to_f32(r.z & 0xffffu); v[5] = f16_bits_to_f32(r.z >> 16);
        v[6] = f16_bits_to_f32(r.w & 0xffffu); v[7] = f16_bits_to_f32(r.w >> 16);
    }
    __device__ static __forceinline__ void load(const f16_t* p, float (&v)[8]) { unpack(*reinterpret_cast<const U4*>(p), v); }
    __device__ static __forceinline__ void zero(float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef F8Raw Raw;
    __device__ static __forceinline__ F8Raw load_raw(const float* p) {
        return F8Raw{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)};
    }
    __device__ static __forceinline__ void unpack_raw(const F8Raw& r, float (&v)[8]) {
        v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
    }
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 r = *reinterpret_cast<const float4*>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    __device__ static __forceinline__ void unpack(const U4& r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y);
        v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
    }
    __device__ static __forceinline__ void zero(float (&v)[4]) { v[0] = v[1] = v[2] = v[3] = 0.f; }
};

// Wave reductions on the DPP crossbar (no LDS round trips): four row steps leave every lane of a
// 16-lane row with the row total, the four row totals are then combined in a fixed order, so every
// lane ends with the same bits on every run.  All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7-i within 8
constexpr int DPP_MIRROR = 0x140;       // lane i <-> 15-i within 16
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ float row16_sum(float v) {  // every lane: sum over its 16-lane row
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return ((lane_f(v, 0) + lane_f(v, 16)) + lane_f(v, 32)) + lane_f(v, 48);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
// sum over aligned groups of W lanes (W = 2, 4, 8, 16), result in every lane of the group
__device__ __forceinline__ float group_sum_rt(float v, int W) {
    v += dpp_f<DPP_XOR1>(v);
    if (W >= 4) v += dpp_f<DPP_XOR2>(v);
    if (W >= 8) v += dpp_f<DPP_HALF_MIRROR>(v);
    if (W >= 16) v += dpp_f<DPP_MIRROR>(v);
    return v;
}

// exact-ish f32 helpers (no fast-math): IEEE sqrt/div as torch's CPU kernels use
__device__ __forceinline__ float rsqrt_exact(float v) { return 1.0f / sqrtf(v); }

}  // namespace ft
