// libfishtts_hip.so: context, weight ingestion, the AR prefill/decode drivers (hipGraph-captured
// frame step) and the C ABI declared in include/fishtts_hip.h.
#include "engine.h"
#include "ar_kernels.h"
#include "frame_engine.h"
#include "wide_kernels.h"
#include "codec_kernels.h"

#include <math.h>

#include <algorithm>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

using namespace ft;

static std::string g_create_err;
static void eng_release(ft_ctx* ctx);

ft_status ft_fail(ft_ctx* ctx, ft_status code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

// ------------------------------------------------------------------------------------------ utils
template <typename S, typename D>
__global__ void convert_kernel(const S* s, D* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st_elem(d, (size_t)i, ld_elem(s, (size_t)i));
}
template <typename T>
__global__ void interleave_rows_kernel(const T* a, const T* b, T* o, int64_t rows, int64_t K) {
    const int64_t n = rows * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K, k = i % K;
        o[(2 * r) * K + k] = a[i];
        o[(2 * r + 1) * K + k] = b[i];
    }
}

template <typename T>
static ft_status dmalloc(ft_ctx* ctx, T** p, size_t n) {
    void* q = nullptr;
    const size_t bytes = n * sizeof(T) ? n * sizeof(T) : 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return ft_fail(ctx, FT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    hipMemset(q, 0, bytes);
    *p = (T*)q;
    return FT_OK;
}
#define FT_TRY(x) do { ft_status s_ = (x); if (s_ != FT_OK) return s_; } while (0)

// ------------------------------------------------------------------------------------------ names
void ft_expect(ft_ctx* ctx, const std::string& name, std::vector<int64_t> shape, int dtype) {
    FtTensor t;
    t.shape = std::move(shape);
    t.dtype = dtype;
    ctx->expected[name] = t;
}

static void ar_expected(ft_ctx* ctx) {
    const ft_ar_config& c = ctx->c;
    const int dt = c.dtype;
    ft_expect(ctx, "embeddings.weight", {c.vocab_size, c.dim}, dt);
    ft_expect(ctx, "codebook_embeddings.weight", {(int64_t)c.codebook_size * c.num_codebooks, c.dim}, dt);
    auto block = [&](const std::string& p, int dim, int nh, int nkv, int hd, int ffn, int qkvb, int ob, int qkn) {
        const int64_t tot = (int64_t)(nh + 2 * nkv) * hd;
        ft_expect(ctx, p + ".attention.wqkv.weight", {tot, dim}, dt);
        if (qkvb) ft_expect(ctx, p + ".attention.wqkv.bias", {tot}, dt);
        ft_expect(ctx, p + ".attention.wo.weight", {dim, (int64_t)nh * hd}, dt);
        if (ob) ft_expect(ctx, p + ".attention.wo.bias", {dim}, dt);
        if (qkn) {
            ft_expect(ctx, p + ".attention.q_norm.weight", {hd}, dt);
            ft_expect(ctx, p + ".attention.k_norm.weight", {hd}, dt);
        }
        ft_expect(ctx, p + ".feed_forward.w1.weight", {ffn, dim}, dt);
        ft_expect(ctx, p + ".feed_forward.w3.weight", {ffn, dim}, dt);
        ft_expect(ctx, p + ".feed_forward.w2.weight", {dim, ffn}, dt);
        ft_expect(ctx, p + ".ffn_norm.weight", {dim}, dt);
        ft_expect(ctx, p + ".attention_norm.weight", {dim}, dt);
    };
    for (int i = 0; i < c.n_layer; ++i)
        block("layers." + std::to_string(i), c.dim, c.n_head, c.n_local_heads, c.head_dim, c.intermediate_size,
              c.attention_qkv_bias, c.attention_o_bias, c.attention_qk_norm);
    ft_expect(ctx, "norm.weight", {c.dim}, dt);
    if (!c.tie_word_embeddings) ft_expect(ctx, "output.weight", {c.vocab_size, c.dim}, dt);
    if (c.fast_dim != c.dim) {
        ft_expect(ctx, "fast_project_in.weight", {c.fast_dim, c.dim}, dt);
        ft_expect(ctx, "fast_project_in.bias", {c.fast_dim}, dt);
    }
    ft_expect(ctx, "fast_embeddings.weight", {c.codebook_size, c.fast_dim}, dt);
    for (int i = 0; i < c.n_fast_layer; ++i)
        block("fast_layers." + std::to_string(i), c.fast_dim, c.fast_n_head, c.fast_n_local_heads, c.fast_head_dim,
              c.fast_intermediate_size, c.fast_attention_qkv_bias, c.fast_attention_o_bias,
              c.fast_attention_qk_norm);
    ft_expect(ctx, "fast_norm.weight", {c.fast_dim}, dt);
    ft_expect(ctx, "fast_output.weight", {c.codebook_size, c.fast_dim}, dt);
    // RoPE tables exactly as the reference stores them (llama.py:594-603): bf16-rounded cos/sin,
    // supplied by the host so the table is bit-identical; kept as f32 in HBM.
    ft_expect(ctx, "rope.slow", {c.max_seq_len, c.head_dim / 2, 2}, FT_F32);
    ft_expect(ctx, "rope.fast", {c.num_codebooks, c.fast_head_dim / 2, 2}, FT_F32);
}

// ------------------------------------------------------------------------------------------ create
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static ft_status ar_validate(ft_ctx* ctx) {
    const ft_ar_config& c = ctx->c;
    auto bad = [&](const char* m) { return ft_fail(ctx, FT_ERR_UNSUPPORTED, m); };
    if (c.dtype != FT_BF16 && c.dtype != FT_F32 && c.dtype != FT_F16) return bad("dtype must be FT_BF16, FT_F16 or FT_F32");
    if (c.dim % 8 || c.fast_dim % 8 || c.intermediate_size % 8 || c.fast_intermediate_size % 8)
        return bad("dim / intermediate sizes must be multiples of 8");
    if (!pow2(c.head_dim) || c.head_dim < 8 || c.head_dim > 128) return bad("head_dim must be a power of two in [8,128]");
    if (!pow2(c.fast_head_dim) || c.fast_head_dim < 16 || c.fast_head_dim > 128)
        return bad("fast_head_dim must be a power of two in [16,128]");
    if (c.n_head % c.n_local_heads || c.fast_n_head % c.fast_n_local_heads) return bad("n_head % n_local_heads != 0");
    const int G = c.n_head / c.n_local_heads;
    if (G != 1 && G != 2 && G != 4 && G != 8) return bad("n_head / n_local_heads must be 1, 2, 4 or 8");
    if (c.num_codebooks < 1 || c.num_codebooks > FAST_MAXCB) return bad("num_codebooks must be in [1,16]");
    if (c.max_batch < 1 || c.max_batch > 256) return bad("max_batch must be in [1,256]");
    if (c.max_new_tokens < 1) return bad("max_new_tokens must be >= 1");
    if (c.vocab_size <= c.semantic_end_id || c.semantic_begin_id > c.semantic_end_id) return bad("bad semantic id range");
    return FT_OK;
}

static ft_status ar_alloc(ft_ctx* ctx) {
    const ft_ar_config& c = ctx->c;
    const size_t M = c.max_batch;
    ctx->esz = c.dtype == FT_F32 ? 4 : 2;
    ctx->n_slots = c.max_seq_len + ((8 - c.max_seq_len % 8) % 8);  // llama.py:387
    const char* ns = getenv("FT_ATTN_NSPLIT");
    ctx->nsplit = ns ? atoi(ns) : (ctx->n_slots > 512 ? 8 : 1);
    ctx->nsplit_fixed = ns != nullptr;
    ctx->nsplit_max = ctx->nsplit_fixed ? ctx->nsplit : (ctx->n_slots > 3072 ? 32 : ctx->n_slots > 768 ? 16 : ctx->nsplit);
    // One kv head per XCD (frame_engine.h, XL): 8 kv heads of 2 x 128-wide query heads on a chip of 8 XCDs x 32 CUs split every
    // head's cached positions 32 ways - one split per CU of the head's XCD - at every context length.  The split count is a
    // property of the shapes and the chip, not of the frame path: launches and engine then sum in the same order.
    // (Measured and not kept: short contexts walked UNSPLIT by every CU of the XCD for itself - no partials, no merge: 704 us
    // per slow-stack launch at 150 positions and 825 at 260 against 589 for the 32-way split; a 16-position step of the walk
    // costs ~0.45 us - two exact exponentials per head and step - and 10-17 of them sit on every layer's path.)
    {
        hipDeviceProp_t prop;
        ctx->xl_shape = !ns && !getenv("FT_NO_XL") && c.n_local_heads == 8 && c.n_head == 16 && c.head_dim == 128 &&
                        hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount == 256;
        if (ctx->xl_shape) { ctx->nsplit = ctx->nsplit_max = 32; ctx->nsplit_fixed = true; }
    }
    if (ctx->nsplit < 1) ctx->nsplit = 1;
    if (ctx->nsplit > 32) ctx->nsplit = 32;
    ctx->nsplit_max = std::max(1, std::min(32, std::max(ctx->nsplit_max, ctx->nsplit)));
    ctx->cap = c.max_new_tokens + 24;
    ctx->fastV = c.codebook_size < 1024 ? c.codebook_size : 1024;  // inference.py:134
    const size_t qkvN = (size_t)(c.n_head + 2 * c.n_local_heads) * c.head_dim;
    const size_t fqkvN = (size_t)(c.fast_n_head + 2 * c.fast_n_local_heads) * c.fast_head_dim;
    FT_TRY(dmalloc(ctx, &ctx->x, M * c.dim));
    FT_TRY(dmalloc(ctx, &ctx->qkv, M * qkvN));
    ctx->y_ld = std::max(c.n_head * c.head_dim, c.fast_n_head * c.fast_head_dim);
    FT_TRY(dmalloc(ctx, &ctx->y, M * (size_t)ctx->y_ld));
    FT_TRY(dmalloc(ctx, &ctx->g, M * c.intermediate_size));
    FT_TRY(dmalloc(ctx, &ctx->logits, M * c.vocab_size));
    if (c.fast_dim != c.dim) FT_TRY(dmalloc(ctx, &ctx->hid, M * c.fast_dim));
    else ctx->hid = ctx->x;
    FT_TRY(dmalloc(ctx, &ctx->femb, M * c.fast_dim));
    FT_TRY(dmalloc(ctx, &ctx->xf, M * c.fast_dim));
    FT_TRY(dmalloc(ctx, &ctx->qkvf, 2 * ((M + 15) / 16 * 16) * fqkvN));   // 2 x: the paired codebook pass of wide batches (positions 0 and 1 as 2 M rows)
    FT_TRY(dmalloc(ctx, &ctx->gf, M * c.fast_intermediate_size));
    FT_TRY(dmalloc(ctx, &ctx->flog, M * ctx->fastV));
    FT_TRY(dmalloc(ctx, &ctx->part_o, M * c.n_head * ctx->nsplit_max * c.head_dim));
    FT_TRY(dmalloc(ctx, &ctx->part_ml, M * c.n_head * ctx->nsplit_max * 2));
    ctx->cache_m_stride = (size_t)c.n_local_heads * ctx->n_slots * c.head_dim;
    ctx->fcache_m_stride = (size_t)c.fast_n_local_heads * c.num_codebooks * c.fast_head_dim;
    ctx->layers.resize(c.n_layer);
    ctx->flayers.resize(c.n_fast_layer);
    for (auto& l : ctx->layers) {
        FT_TRY(dmalloc(ctx, (char**)&l.kc, M * ctx->cache_m_stride * ctx->esz));
        FT_TRY(dmalloc(ctx, (char**)&l.vc, M * ctx->cache_m_stride * ctx->esz));
    }
    for (auto& l : ctx->flayers) {
        FT_TRY(dmalloc(ctx, (char**)&l.kc, M * ctx->fcache_m_stride * ctx->esz));
        FT_TRY(dmalloc(ctx, (char**)&l.vc, M * ctx->fcache_m_stride * ctx->esz));
    }
    const size_t R = c.num_codebooks + 1;
    FT_TRY(dmalloc(ctx, &ctx->d_pos, M));
    FT_TRY(dmalloc(ctx, &ctx->d_tok, M * R));
    FT_TRY(dmalloc(ctx, &ctx->d_tokn, M * R));
    FT_TRY(dmalloc(ctx, &ctx->d_seq, M * R * ctx->cap));
    FT_TRY(dmalloc(ctx, &ctx->d_nf, M));
    FT_TRY(dmalloc(ctx, &ctx->d_done, M));
    FT_TRY(dmalloc(ctx, &ctx->d_prompt, R * (size_t)c.max_seq_len));
    FT_TRY(dmalloc(ctx, &ctx->d_ctl, M));
    ctx->prefill_gemm_mode = getenv("FT_PREFILL_GEMM") ? atoi(getenv("FT_PREFILL_GEMM")) : 2;
    ctx->prefill_v0 = getenv("FT_PREFILL_V0") != nullptr || c.dtype != FT_BF16 || c.dim % 32 || (c.n_head * c.head_dim) % 32 ||
                      c.intermediate_size % 32;
    if (!ctx->prefill_v0) {
        const size_t S = c.max_seq_len;
        FT_TRY(dmalloc(ctx, &ctx->pf_x, S * c.dim));
        FT_TRY(dmalloc(ctx, &ctx->pf_qkv, S * qkvN));
        FT_TRY(dmalloc(ctx, &ctx->pf_y, S * c.n_head * c.head_dim));
        FT_TRY(dmalloc(ctx, &ctx->pf_xn, S * c.dim));
        FT_TRY(dmalloc(ctx, &ctx->pf_ybf, S * c.n_head * c.head_dim));
        FT_TRY(dmalloc(ctx, &ctx->pf_g, S * c.intermediate_size));
        FT_TRY(dmalloc(ctx, &ctx->pf_qbf, S * c.n_head * c.head_dim));
        FT_TRY(dmalloc(ctx, &ctx->pf_seqs, M));
        FT_TRY(dmalloc(ctx, &ctx->pf_rows, S));
        // wide lock-step batches (wide_kernels.h): no fast-layer f32 bias copies exist; every contraction width must be one
        // the kernel is instantiated for, every output width a whole number of its tiles
        const int HD = c.n_head * c.head_dim, HDf = c.fast_n_head * c.fast_head_dim;
        ctx->wide_ok = !getenv("FT_NO_WIDE") && !c.fast_attention_qkv_bias && !c.fast_attention_o_bias && c.fast_dim == c.dim &&
                       wide_k_ok(c.dim) && wide_k_ok(HD) && wide_k_ok(c.intermediate_size) && wide_k_ok(HDf) && wide_k_ok(c.fast_intermediate_size) &&
                       qkvN % 32 == 0 && fqkvN % 32 == 0 && c.intermediate_size % 16 == 0 && c.fast_intermediate_size % 16 == 0 &&
                       c.dim % 16 == 0 && c.vocab_size % 32 == 0 && ctx->fastV % 16 == 0;
        if (ctx->wide_ok) {
            // rows [0, xo_pair) = the batch, rows [xo_pair, 2 xo_pair) = the same utterances at codebook position 1 in the
            // paired first pass of the codebook loop
            ctx->xo_pair = (int)((M + 15) / 16 * 16);
            ctx->no_pair = getenv("FT_NO_PAIR") != nullptr;
            ctx->no_attn_wide = getenv("FT_NO_ATTN_WIDE") != nullptr;
            ctx->no_head_stream = getenv("FT_NO_HEAD_STREAM") != nullptr;
            ctx->xo_ldm = 2 * ctx->xo_pair;
            const size_t P = ctx->xo_ldm;
            FT_TRY(dmalloc(ctx, &ctx->xo_x, P * c.dim));
            FT_TRY(dmalloc(ctx, &ctx->xo_xf, P * c.fast_dim));
            FT_TRY(dmalloc(ctx, &ctx->xo_femb, P * c.fast_dim));
            FT_TRY(dmalloc(ctx, &ctx->xo_y, P * (size_t)std::max(HD, HDf)));
            FT_TRY(dmalloc(ctx, &ctx->xo_g, P * (size_t)std::max(c.intermediate_size, c.fast_intermediate_size)));
        }
    }
    const size_t nchunk = ((size_t)c.vocab_size + 1023) / 1024;
    FT_TRY(dmalloc(ctx, &ctx->samp_cut, M));
    FT_TRY(dmalloc(ctx, &ctx->samp_chunk_cnt, M * nchunk));
    FT_TRY(dmalloc(ctx, &ctx->samp_part_score, M * nchunk));
    FT_TRY(dmalloc(ctx, &ctx->samp_part_idx, M * nchunk));
    FT_HIP(ctx, hipHostMalloc((void**)&ctx->h_pin, (3 * M + 8) * sizeof(int), hipHostMallocDefault));
    return FT_OK;
}

extern "C" ft_status ft_create(const ft_ar_config* ar, const ft_codec_config* codec, int32_t device, ft_ctx** out) {
    if (!out || (!ar && !codec)) return ft_fail(nullptr, FT_ERR_ARG, "ft_create: need a config and an out pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return ft_fail(nullptr, FT_ERR_HIP, "ft_create: no HIP device visible (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= ndev) return ft_fail(nullptr, FT_ERR_ARG, "ft_create: bad device index");
    ft_ctx* ctx = new ft_ctx();
    ctx->device = device;
    ft_status st = FT_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { st = ft_fail(ctx, FT_ERR_HIP, "hipSetDevice failed"); break; }
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            st = ft_fail(ctx, FT_ERR_HIP, "hipStreamCreate failed"); break;
        }
        if (ar) {
            ctx->c = *ar;
            ctx->has_ar = true;
            if ((st = ar_validate(ctx)) != FT_OK) break;
            if ((st = ar_alloc(ctx)) != FT_OK) break;
            ar_expected(ctx);
        }
        if (codec) {
            ctx->cc = *codec;
            ctx->has_codec = true;
            if ((st = codec_create(ctx)) != FT_OK) break;
            codec_expected(ctx);
        }
    } while (0);
    if (st != FT_OK) {
        g_create_err = ctx->err;
        ft_destroy(ctx);
        return st;
    }
    *out = ctx;
    return FT_OK;
}

extern "C" void ft_destroy(ft_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->graphs) hipGraphExecDestroy(kv.second);
    for (auto& kv : ctx->expected) if (kv.second.p) hipFree(kv.second.p);
    for (auto& l : ctx->layers) { hipFree(l.kc); hipFree(l.vc); if (l.w13) hipFree(l.w13); }
    for (auto& l : ctx->flayers) { hipFree(l.kc); hipFree(l.vc); if (l.w13) hipFree(l.w13); }
    float* bufs[] = {ctx->x, ctx->qkv, ctx->y, ctx->g, ctx->logits, ctx->femb, ctx->xf, ctx->qkvf, ctx->gf,
                     ctx->flog, ctx->part_o, ctx->part_ml, ctx->noise};
    for (float* b : bufs) if (b) hipFree(b);
    if (ctx->hid && ctx->hid != ctx->x) hipFree(ctx->hid);
    int* ibufs[] = {ctx->d_pos, ctx->d_tok, ctx->d_tokn, ctx->d_seq, ctx->d_nf, ctx->d_done, ctx->d_prompt};
    for (int* b : ibufs) if (b) hipFree(b);
    if (ctx->d_ctl) hipFree(ctx->d_ctl);
    if (ctx->wide_qkv0_tab) hipFree(ctx->wide_qkv0_tab);
    { void* pf[] = {ctx->pf_x, ctx->pf_qkv, ctx->pf_y, ctx->pf_xn, ctx->pf_ybf, ctx->pf_g, ctx->xo_x, ctx->xo_xf, ctx->xo_femb, ctx->xo_y, ctx->xo_g, ctx->pf_qbf}; for (void* q : pf) if (q) hipFree(q); }
    for (auto& l : ctx->layers) { if (l.bqkv_f32) hipFree(l.bqkv_f32); if (l.bo_f32) hipFree(l.bo_f32); }
    if (ctx->samp_cut) hipFree(ctx->samp_cut);
    if (ctx->samp_chunk_cnt) hipFree(ctx->samp_chunk_cnt);
    if (ctx->samp_part_score) hipFree(ctx->samp_part_score);
    if (ctx->samp_part_idx) hipFree(ctx->samp_part_idx);
    eng_release(ctx);
    if (ctx->h_pin) hipHostFree(ctx->h_pin);
    for (auto e : ctx->prof_ev) hipEventDestroy(e);
    codec_destroy(ctx);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* ft_last_error(const ft_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" ft_status ft_sync(ft_ctx* ctx) {
    if (!ctx) return FT_ERR_ARG;
    FT_HIP(ctx, hipSetDevice(ctx->device));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ weights
extern "C" ft_status ft_load_weight(ft_ctx* ctx, const char* name, const void* src, int32_t src_dtype,
                                    const int64_t* shape, int32_t ndim) {
    if (!ctx || !name || !src || !shape) return ft_fail(ctx, FT_ERR_ARG, "ft_load_weight: null argument");
    if (ctx->finalized) return ft_fail(ctx, FT_ERR_STATE, "ft_load_weight: weights already finalized");
    auto it = ctx->expected.find(name);
    if (it == ctx->expected.end()) return ft_fail(ctx, FT_ERR_ARG, std::string("unknown weight name: ") + name);
    FtTensor& t = it->second;
    if ((int)t.shape.size() != ndim) return ft_fail(ctx, FT_ERR_ARG, std::string("rank mismatch for ") + name);
    for (int i = 0; i < ndim; ++i)
        if (t.shape[i] != shape[i]) {
            char buf[256];
            snprintf(buf, sizeof buf, "shape mismatch for %s: dim %d is %lld, expected %lld", name, i,
                     (long long)shape[i], (long long)t.shape[i]);
            return ft_fail(ctx, FT_ERR_ARG, buf);
        }
    if (src_dtype != FT_F32 && src_dtype != FT_BF16 && src_dtype != FT_F16)
        return ft_fail(ctx, FT_ERR_ARG, "src_dtype must be FT_F32, FT_BF16 or FT_F16");
    FT_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = t.numel();
    const size_t ssz = src_dtype == FT_F32 ? 4 : 2, dsz = t.dtype == FT_F32 ? 4 : 2;
    if (!t.p) FT_HIP(ctx, hipMalloc(&t.p, (size_t)n * dsz + 64));
    if (src_dtype == t.dtype) {
        FT_HIP(ctx, hipMemcpy(t.p, src, (size_t)n * dsz, hipMemcpyDefault));
        return FT_OK;
    }
    void* stage = nullptr;
    FT_HIP(ctx, hipMalloc(&stage, (size_t)n * ssz));
    hipError_t e = hipMemcpy(stage, src, (size_t)n * ssz, hipMemcpyDefault);
    if (e == hipSuccess) {
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        auto to = [&](auto* sp) {
            if (t.dtype == FT_F32) convert_kernel<<<blocks, 256, 0, ctx->stream>>>(sp, (float*)t.p, n);
            else if (t.dtype == FT_BF16) convert_kernel<<<blocks, 256, 0, ctx->stream>>>(sp, (bf16_t*)t.p, n);
            else convert_kernel<<<blocks, 256, 0, ctx->stream>>>(sp, (f16_t*)t.p, n);
        };
        if (src_dtype == FT_F32) to((const float*)stage);
        else if (src_dtype == FT_BF16) to((const bf16_t*)stage);
        else to((const f16_t*)stage);
        e = hipStreamSynchronize(ctx->stream);
    }
    hipFree(stage);
    if (e != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("ft_load_weight copy: ") + hipGetErrorString(e));
    return FT_OK;
}

static void* wp(ft_ctx* ctx, const std::string& n) {
    auto it = ctx->expected.find(n);
    return it == ctx->expected.end() ? nullptr : it->second.p;
}

static ft_status eng_setup(ft_ctx* ctx);
static ft_status wide_qkv0_build(ft_ctx* ctx);

static ft_status ar_finalize(ft_ctx* ctx) {
    const ft_ar_config& c = ctx->c;
    ctx->emb = wp(ctx, "embeddings.weight");
    ctx->cb_emb = wp(ctx, "codebook_embeddings.weight");
    ctx->norm = wp(ctx, "norm.weight");
    ctx->head = c.tie_word_embeddings ? ctx->emb : wp(ctx, "output.weight");
    ctx->fproj_w = wp(ctx, "fast_project_in.weight");
    ctx->fproj_b = wp(ctx, "fast_project_in.bias");
    ctx->fast_emb = wp(ctx, "fast_embeddings.weight");
    ctx->fast_norm = wp(ctx, "fast_norm.weight");
    ctx->fast_out = wp(ctx, "fast_output.weight");
    ctx->rope = (float*)wp(ctx, "rope.slow");
    ctx->frope = (float*)wp(ctx, "rope.fast");
    auto fill = [&](std::vector<FtLayer>& ls, const char* pre, int ffn, int dim) -> ft_status {
        for (size_t i = 0; i < ls.size(); ++i) {
            const std::string p = std::string(pre) + std::to_string(i);
            FtLayer& l = ls[i];
            l.attn_norm = wp(ctx, p + ".attention_norm.weight");
            l.wqkv = wp(ctx, p + ".attention.wqkv.weight");
            l.bqkv = wp(ctx, p + ".attention.wqkv.bias");
            l.qn = wp(ctx, p + ".attention.q_norm.weight");
            l.kn = wp(ctx, p + ".attention.k_norm.weight");
            l.wo = wp(ctx, p + ".attention.wo.weight");
            l.bo = wp(ctx, p + ".attention.wo.bias");
            l.ffn_norm = wp(ctx, p + ".ffn_norm.weight");
            l.w2 = wp(ctx, p + ".feed_forward.w2.weight");
            // w1/w3 rows interleaved so one wave owns a (gate, up) pair (SwiGLU epilogue)
            FtTensor& w1 = ctx->expected[p + ".feed_forward.w1.weight"];
            FtTensor& w3 = ctx->expected[p + ".feed_forward.w3.weight"];
            FT_HIP(ctx, hipMalloc(&l.w13, (size_t)2 * ffn * dim * ctx->esz));
            const int64_t n = (int64_t)ffn * dim;
            const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
            if (ctx->esz == 2)
                interleave_rows_kernel<bf16_t><<<blocks, 256, 0, ctx->stream>>>((bf16_t*)w1.p, (bf16_t*)w3.p, (bf16_t*)l.w13, ffn, dim);
            else
                interleave_rows_kernel<float><<<blocks, 256, 0, ctx->stream>>>((float*)w1.p, (float*)w3.p, (float*)l.w13, ffn, dim);
            FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(w1.p); w1.p = nullptr;
            hipFree(w3.p); w3.p = nullptr;
        }
        return FT_OK;
    };
    FT_TRY(fill(ctx->layers, "layers.", c.intermediate_size, c.dim));
    if (!ctx->prefill_v0) {
        const int qkvN = (c.n_head + 2 * c.n_local_heads) * c.head_dim;
        for (auto& l : ctx->layers) {
            if (l.bqkv) {
                FT_HIP(ctx, hipMalloc((void**)&l.bqkv_f32, qkvN * sizeof(float)));
                convert_kernel<bf16_t, float><<<(qkvN + 255) / 256, 256, 0, ctx->stream>>>((const bf16_t*)l.bqkv, l.bqkv_f32, qkvN);
            }
            if (l.bo) {
                FT_HIP(ctx, hipMalloc((void**)&l.bo_f32, c.dim * sizeof(float)));
                convert_kernel<bf16_t, float><<<(c.dim + 255) / 256, 256, 0, ctx->stream>>>((const bf16_t*)l.bo, l.bo_f32, c.dim);
            }
        }
        FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    FT_TRY(fill(ctx->flayers, "fast_layers.", c.fast_intermediate_size, c.fast_dim));
    FT_TRY(eng_setup(ctx));
    FT_TRY(wide_qkv0_build(ctx));
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ frame engine (host side)
// The engine is instantiated for a short list of shape classes (eng_slow_shapes / eng_fast_shapes below: openaudio-s1-mini's
// widths, ffn 4096, four kv heads; bf16; any depth); every other configuration keeps the launch path.  It needs every workgroup resident at once:
// one per CU, sized by the device's CU count - so only ONE context per device and process may run it (two would each
// hold part of the CUs and time each other out), and nothing in here is fatal: whatever fails leaves the launch path.
constexpr int ENG_MAX_STRIKES = 2;      // hand-off time-outs after which a context stops using the engine
static std::mutex g_eng_mu;
static std::map<int, ft_ctx*> g_eng_owner;      // device -> the context whose engine runs there

static void eng_release(ft_ctx* ctx) {
    void* eb[] = {ctx->eng_layers, ctx->eng_flayers, ctx->eng_gx /* pool: gxb, gg, gy, gqkv live in it */,
                  ctx->eng_gpart, ctx->eng_fast_g, ctx->eng_ctl, ctx->eng_qkv0_tab};
    for (void* q : eb) if (q) hipFree(q);
    ctx->eng_layers = ctx->eng_flayers = nullptr;
    ctx->eng_gx = ctx->eng_gqkv = ctx->eng_gy = ctx->eng_gxb = ctx->eng_gg = ctx->eng_fast_g = ctx->eng_ctl = nullptr;
    ctx->eng_gpart = nullptr;
    ctx->eng_qkv0_tab = nullptr;
    ctx->eng_on = ctx->eng_fast_on = false;
    if (ctx->eng_owner) {
        std::lock_guard<std::mutex> lk(g_eng_mu);
        auto it = g_eng_owner.find(ctx->device);
        if (it != g_eng_owner.end() && it->second == ctx) g_eng_owner.erase(it);
        ctx->eng_owner = false;
    }
}

// One 512-thread workgroup of `fn` with `lds` bytes of dynamic LDS on one CU: computed from the kernel's own attributes -
// waves per SIMD = 512 registers / the wave's (gfx950: one 512-entry file per SIMD lane, allocated in steps of 8), four SIMDs,
// eight waves to place.  hipOccupancyMaxActiveBlocksPerMultiprocessor is only logged: measured on this runtime it answers 0
// or 1 for the SAME kernel and attributes depending on what the process ran before (it then budgets 256 registers per lane),
// and a false 0 would silently cost the engine.  Should the answer here ever be wrong the other way, the launches time out
// and the recovery turns the engine off after ENG_MAX_STRIKES.
static bool eng_fits_cu(const void* fn, size_t lds, size_t lds_cap, const char* name, std::string& why) {
    hipFuncAttributes fa{};
    const hipError_t fe = hipFuncGetAttributes(&fa, fn);
    if (fe != hipSuccess) { (void)hipGetLastError(); why = std::string("hipFuncGetAttributes: ") + hipGetErrorString(fe); return false; }
    const int regs = std::max(fa.numRegs, 1), per_simd = 512 / (((regs + 7) / 8) * 8);
    const bool fits = per_simd * 4 >= ENG_THREADS / 64 && lds + fa.sharedSizeBytes <= lds_cap && fa.maxThreadsPerBlock >= ENG_THREADS;
    if (getenv("FT_LOG")) {
        int occ = -1;
        const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, ENG_THREADS, lds);
        if (oe != hipSuccess) (void)hipGetLastError();
        fprintf(stderr, "fish_tts_amd: %s engine kernel: %d registers (%d waves per SIMD), %zu B of LDS of %zu: %s (the occupancy query says %d)\n",
                name, fa.numRegs, per_simd, lds + fa.sharedSizeBytes, lds_cap, fits ? "fits a CU" : "does not fit a CU", occ);
    }
    if (!fits) why = std::string("a ") + name + " workgroup does not fit one CU (registers / LDS)";
    return fits;
}

// The shape classes the engine kernels are instantiated for (frame_engine.h: EngSlowShape / EngFastShape; any depth).  The
// first entries are openaudio-s1-mini's widths (config.json: llama.py:74-86); the others cover the two fields SURVEY.md could
// not verify against a real config.json - the feed-forward width and the number of kv heads (four kv heads lose the
// one-kv-head-per-XCD form of the attention and take the general split form).
typedef void (*eng_slow_fn_t)(SlowEngP);
typedef void (*eng_fast_fn_t)(FastEngP);
typedef void (*eng_qkv0_fn_t)(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, float);
struct EngSlowEntry { int D, H, HKV, HDIM, F, SQ, SF, SO; eng_slow_fn_t xl, plain; };
struct EngFastEntry { int D, H, HKV, HDIM, F, V, SQ, SF, SO; eng_fast_fn_t loop; eng_qkv0_fn_t qkv0; };
template <typename S, bool WITH_XL>
static EngSlowEntry eng_slow_entry() {
    eng_slow_fn_t xl = nullptr;
    if constexpr (WITH_XL) xl = slow_engine_kernel<S, true>;
    return EngSlowEntry{S::D, S::H, S::HKV, S::HDIM, S::F, S::SQ, S::SF, S::SO, xl, slow_engine_kernel<S, false>};
}
template <typename S>
static EngFastEntry eng_fast_entry() {
    return EngFastEntry{S::D, S::H, S::HKV, S::HDIM, S::F, S::V, S::SQ, S::SF, S::SO, fast_engine_kernel<S, 10>, eng_qkv0_table_kernel<S>};
}
static const EngSlowEntry* eng_slow_shapes(int* n) {
    static const EngSlowEntry tab[] = {eng_slow_entry<EngSlowS1, true>(), eng_slow_entry<EngSlowShape<1024, 16, 8, 128, 4096>, true>(),
                                       eng_slow_entry<EngSlowShape<1024, 16, 4, 128, 3072>, false>()};
    *n = (int)(sizeof tab / sizeof tab[0]);
    return tab;
}
static const EngFastEntry* eng_fast_shapes(int* n) {
    // (four fast kv heads of 64 would leave a workgroup 6 rows of Wqkv: not a whole number of 4-granule groups - such a model
    // keeps its codebook loop on launches)
    static const EngFastEntry tab[] = {eng_fast_entry<EngFastS1>(), eng_fast_entry<EngFastShape<1024, 16, 8, 64, 4096, 1024>>()};
    *n = (int)(sizeof tab / sizeof tab[0]);
    return tab;
}

// false: `why` says what kept the engine off (allocations made so far are released by the caller)
static bool eng_setup_try(ft_ctx* ctx, std::string& why) {
    const ft_ar_config& c = ctx->c;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        why = std::string(what) + ": " + hipGetErrorString(e);
        (void)hipGetLastError();     // (not sticky for the launch path)
        return false;
    };
    if (getenv("FT_NO_ENGINE")) { why = "FT_NO_ENGINE is set"; return false; }
    if (c.dtype != FT_BF16) { why = "precision is not bf16"; return false; }
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, ctx->device), "hipGetDeviceProperties")) return false;
    const int nb = prop.multiProcessorCount;
    const int HD = c.n_head * c.head_dim;
    const int qkvN = (c.n_head + 2 * c.n_local_heads) * c.head_dim;
    auto per = [&](int units) { return (units + nb - 1) / nb; };
    // (the kernels carry the widths as compile-time constants: one instantiation per shape class)
    const EngSlowEntry* se = nullptr;
    {
        int n = 0;
        const EngSlowEntry* tab = eng_slow_shapes(&n);
        for (int i = 0; i < n && !se; ++i)
            if (c.dim == tab[i].D && c.n_head == tab[i].H && c.n_local_heads == tab[i].HKV && c.head_dim == tab[i].HDIM && c.intermediate_size == tab[i].F)
                se = &tab[i];
    }
    const bool shape_ok = se && c.fast_dim == c.dim && c.n_layer >= 1 && c.n_layer < ENG_EPOCH_STEP &&
                          per(qkvN) <= se->SQ * ENG_CW && per(c.dim) <= se->SO * ENG_CW && per(c.intermediate_size) <= se->SF * ENG_CW &&
                          qkvN % (4 * nb) == 0 && c.dim % (4 * nb) == 0 && c.intermediate_size % (4 * nb) == 0 && per(qkvN) <= ENG_LINE &&
                          per(c.intermediate_size) <= ENG_LINE && c.n_local_heads * ctx->nsplit_max <= nb && c.head_dim % (4 * ctx->nsplit_max) == 0 && nb == ENG_NB;
    if (!shape_ok) {
        char buf[320];
        snprintf(buf, sizeof buf, "widths outside the engine's instantiations (dim %d, heads %d / %d x %d, ffn %d, fast_dim %d on %d CUs; "
                 "built for dim 1024, 16 heads x 128 with 8 kv heads and ffn 3072 or 4096, or 4 kv heads and ffn 3072, on 256 CUs)", c.dim, c.n_head,
                 c.n_local_heads, c.head_dim, c.intermediate_size, c.fast_dim, nb);
        why = buf;
        return false;
    }
    {
        std::lock_guard<std::mutex> lk(g_eng_mu);
        auto it = g_eng_owner.find(ctx->device);
        if (it != g_eng_owner.end() && it->second != ctx) {
            why = "another context of this process already runs the frame engine on this device (its workgroups need every CU)";
            return false;
        }
        g_eng_owner[ctx->device] = ctx;
        ctx->eng_owner = true;
    }
    ctx->eng_nb = nb;
    std::vector<EngLayer> h(c.n_layer);
    for (int i = 0; i < c.n_layer; ++i) {
        const FtLayer& l = ctx->layers[i];
        h[i] = EngLayer{(const bf16_t*)l.wqkv, (const bf16_t*)l.bqkv, (const bf16_t*)l.attn_norm, (const bf16_t*)l.qn,
                        (const bf16_t*)l.kn, (const bf16_t*)l.wo, (const bf16_t*)l.bo, (const bf16_t*)l.ffn_norm,
                        (const bf16_t*)l.w13, (const bf16_t*)l.w2, (bf16_t*)l.kc, (bf16_t*)l.vc};
    }
    if (!hip_ok(hipMalloc((void**)&ctx->eng_layers, h.size() * sizeof(EngLayer)), "hipMalloc(engine layer table)")) return false;
    if (!hip_ok(hipMemcpy(ctx->eng_layers, h.data(), h.size() * sizeof(EngLayer), hipMemcpyHostToDevice), "hipMemcpy(engine layer table)")) return false;
    auto zalloc = [&](void** q, size_t bytes, const char* what) {
        return hip_ok(hipMalloc(q, bytes), what) && hip_ok(hipMemset(*q, 0, bytes), what);
    };
    const size_t L = c.n_layer;
    const size_t VB = (size_t)nb * ENG_LINE * 4;   // bytes reserved per hand-off vector (one 128-byte line per workgroup)
    // one pool [gx | gxb | gg | gy | gqkv], followed by the eight per-XCD replicas of it (XCD relay)
    const size_t pool_words = ((L + 1) * VB + 3 * L * VB + L * HD * 4) / 4;
    ctx->eng_relay = getenv("FT_NO_RELAY") == nullptr;
    ctx->eng_pool_bytes = pool_words * 4 * (ctx->eng_relay ? 9 : 1);
    if (!zalloc((void**)&ctx->eng_gx, ctx->eng_pool_bytes, "hipMalloc(engine hand-off pool)")) return false;
    ctx->eng_pool_words = pool_words;
    ctx->eng_gxb = ctx->eng_gx + (L + 1) * (VB / 4);
    ctx->eng_gg = ctx->eng_gxb + L * (VB / 4);
    ctx->eng_gy = ctx->eng_gg + L * (VB / 4);
    ctx->eng_gqkv = ctx->eng_gy + L * HD;
    ctx->eng_gpart_bytes = L * c.n_head * ctx->nsplit_max * (size_t)(c.head_dim + 2) * 8;
    if (!zalloc((void**)&ctx->eng_gpart, ctx->eng_gpart_bytes, "hipMalloc(engine split partials)")) return false;
    if (!zalloc((void**)&ctx->eng_ctl, ENG_CTL_WORDS * 4, "hipMalloc(engine control words)")) return false;
    const int G = c.n_head / c.n_local_heads, hd = c.head_dim, NSLOT = 4 * (64 / (hd >> 3));
    size_t fl = (size_t)c.dim * 2 + HD + c.intermediate_size + (size_t)(G + 2) * hd + (size_t)G * hd + 2 * hd +
                (size_t)NSLOT * G * 2 + (size_t)NSLOT * G * hd + 64 * 6 + ENG_MAX_OUT + 8;
    ctx->eng_lds_slow = std::max(fl * sizeof(float), (size_t)82 * 1024);   // > half the CU's LDS: one workgroup per CU
    const size_t lds_cap = prop.maxSharedMemoryPerMultiProcessor;
    if (ctx->eng_lds_slow > lds_cap) { why = "the slow stack's LDS need exceeds the CU's"; return false; }
    ctx->eng_xl = ctx->xl_shape && nb == 256 && ctx->nsplit_max == 32 && c.n_local_heads * 32 == nb;
    ctx->eng_xl = ctx->eng_xl && se->xl != nullptr;
    ctx->eng_slow_fn = (const void*)(ctx->eng_xl ? se->xl : se->plain);
    const void* slow_fn = ctx->eng_slow_fn;
    if (!hip_ok(hipFuncSetAttribute(slow_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->eng_lds_slow),
                "hipFuncSetAttribute(slow_engine_kernel)")) return false;
    // every workgroup must be resident at once, one per CU: does one fit a CU at all (registers, LDS, waves)?
    if (!eng_fits_cu(slow_fn, ctx->eng_lds_slow, lds_cap, "slow-stack", why)) return false;
    ctx->eng_on = true;
    why = ctx->eng_xl ? "slow stack on the frame engine (one kv head per XCD)" : "slow stack on the frame engine";

    // ---- fast codebook loop
    const int HDf = c.fast_n_head * c.fast_head_dim;
    const int fqkvN = (c.fast_n_head + 2 * c.fast_n_local_heads) * c.fast_head_dim;
    const EngFastEntry* fe = nullptr;
    {
        int n = 0;
        const EngFastEntry* tab = eng_fast_shapes(&n);
        for (int i = 0; i < n && !fe; ++i)
            if (c.fast_dim == tab[i].D && c.fast_n_head == tab[i].H && c.fast_n_local_heads == tab[i].HKV && c.fast_head_dim == tab[i].HDIM &&
                c.fast_intermediate_size == tab[i].F && ctx->fastV == tab[i].V)
                fe = &tab[i];
    }
    const bool fast_ok = !getenv("FT_NO_FAST_ENGINE") && fe && HDf == c.fast_dim && c.n_fast_layer >= 1 &&
                         c.num_codebooks >= 2 && c.num_codebooks <= 10 && fqkvN % (4 * nb) == 0 && ctx->fastV % (4 * nb) == 0 &&
                         per(fqkvN) <= fe->SQ * ENG_CW && per(c.fast_dim) <= fe->SO * ENG_CW && per(c.fast_intermediate_size) <= fe->SF * ENG_CW &&
                         per(ctx->fastV) <= fe->SO * ENG_CW && per(fqkvN) <= ENG_LINE && per(c.fast_intermediate_size) <= ENG_LINE && ctx->fastV <= 1024 &&
                         c.codebook_size <= 65536;
    if (!fast_ok) { why += "; fast loop on launches (FT_NO_FAST_ENGINE, or fast widths outside the instantiations: 1024 / 16 + 8 heads x 64 / ffn 3072 or 4096 / 1024 codes used, <= 10 codebooks)"; return true; }
    ctx->eng_fast_fn = (const void*)fe->loop;
    // from here on a failure keeps the slow engine and leaves the fast loop on launches
    auto fast_off = [&](const std::string& w2) { why += "; fast loop on launches (" + w2 + ")"; return true; };
    std::vector<EngLayer> hf(c.n_fast_layer);
    for (int i = 0; i < c.n_fast_layer; ++i) {
        const FtLayer& l = ctx->flayers[i];
        hf[i] = EngLayer{(const bf16_t*)l.wqkv, (const bf16_t*)l.bqkv, (const bf16_t*)l.attn_norm, (const bf16_t*)l.qn,
                         (const bf16_t*)l.kn, (const bf16_t*)l.wo, (const bf16_t*)l.bo, (const bf16_t*)l.ffn_norm,
                         (const bf16_t*)l.w13, (const bf16_t*)l.w2, nullptr, nullptr};
    }
    std::string w2;
    auto hip_ok2 = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        w2 = std::string(what) + ": " + hipGetErrorString(e);
        (void)hipGetLastError();
        return false;
    };
    if (!hip_ok2(hipMalloc((void**)&ctx->eng_flayers, hf.size() * sizeof(EngLayer)), "hipMalloc(fast layer table)")) return fast_off(w2);
    if (!hip_ok2(hipMemcpy(ctx->eng_flayers, hf.data(), hf.size() * sizeof(EngLayer), hipMemcpyHostToDevice), "hipMemcpy(fast layer table)")) return fast_off(w2);
    const size_t nLf = c.n_fast_layer;
    // [2 parities]: gx (nLf + 1), gqkv, gxb, gg (nLf each), glog (1); then one line per codebook for the drawn codes
    ctx->eng_fast_words = (2 * ((nLf + 1) + 3 * nLf + 1)) * (VB / 4) + (size_t)c.num_codebooks * ENG_LINE;
    ctx->eng_fast_bytes = ctx->eng_fast_words * 4 * (ctx->eng_relay ? 9 : 1);
    if (!hip_ok2(hipMalloc((void**)&ctx->eng_fast_g, ctx->eng_fast_bytes), "hipMalloc(fast hand-off pool)") ||
        !hip_ok2(hipMemset(ctx->eng_fast_g, 0, ctx->eng_fast_bytes), "hipMemset(fast hand-off pool)")) return fast_off(w2);
    const int kvw = c.fast_n_local_heads * c.fast_head_dim;
    // codebook positions 0 and 1 as two rows of one pass when the second row's buffers fit the CU's LDS
    ctx->eng_pair = getenv("FT_NO_PAIR") == nullptr &&
                    eng_fast_lds_bytes(c.fast_dim, fqkvN, HDf, c.fast_intermediate_size, ctx->fastV, (int)nLf, c.num_codebooks, kvw, true) <= lds_cap;
    ctx->eng_lds_fast = eng_fast_lds_bytes(c.fast_dim, fqkvN, HDf, c.fast_intermediate_size, ctx->fastV, (int)nLf, c.num_codebooks, kvw, ctx->eng_pair);
    if (ctx->eng_lds_fast > lds_cap) return fast_off("the codebook loop's LDS need exceeds the CU's");
    ctx->eng_lds_fast = std::max(ctx->eng_lds_fast, (size_t)82 * 1024);
    if (!hip_ok2(hipFuncSetAttribute(ctx->eng_fast_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)ctx->eng_lds_fast), "hipFuncSetAttribute(fast_engine_kernel)")) return fast_off(w2);
    if (!eng_fits_cu(ctx->eng_fast_fn, ctx->eng_lds_fast, lds_cap, "codebook-loop", w2)) return fast_off(w2);
    // layer 0's q k v of every codebook-embedding row a draw can select (codes < fastV): 4 MB at the s1-mini widths
    if (getenv("FT_NO_QKV0") == nullptr && c.num_codebooks > 2) {
        const size_t tb = (size_t)ctx->fastV * fqkvN * sizeof(bf16_t);
        if (!hip_ok2(hipMalloc((void**)&ctx->eng_qkv0_tab, tb), "hipMalloc(layer-0 q k v table)")) return fast_off(w2);
        const FtLayer& l0 = ctx->flayers[0];
        const size_t lds = ((size_t)c.fast_dim + ENG_MAX_OUT) * sizeof(float);
        fe->qkv0<<<dim3(fqkvN / (fe->SQ * ENG_CW), 16), ENG_CW * 64, lds, ctx->stream>>>(
            (const bf16_t*)l0.wqkv, (const bf16_t*)l0.bqkv, (const bf16_t*)l0.attn_norm, (const bf16_t*)ctx->fast_emb,
            (bf16_t*)ctx->eng_qkv0_tab, c.fast_dim, fqkvN, ctx->fastV, c.norm_eps);
        if (!hip_ok2(hipGetLastError(), "eng_qkv0_table_kernel") || !hip_ok2(hipStreamSynchronize(ctx->stream), "eng_qkv0_table_kernel")) {
            hipFree(ctx->eng_qkv0_tab); ctx->eng_qkv0_tab = nullptr;
            return fast_off(w2);
        }
    }
    ctx->eng_fast_on = true;
    why = ctx->eng_xl ? "slow stack (one kv head per XCD) and codebook loop on the frame engine" : "slow stack and codebook loop on the frame engine";
    return true;
}

static ft_status eng_setup(ft_ctx* ctx) {
    ctx->eng_on = ctx->eng_fast_on = false;
    std::string why;
    if (!eng_setup_try(ctx, why)) {
        eng_release(ctx);
        ctx->eng_why = "launch path: " + why;
    } else {
        char buf[64];
        snprintf(buf, sizeof buf, " (%d workgroups)", ctx->eng_nb);
        ctx->eng_why = why + buf;
    }
    if (getenv("FT_LOG")) fprintf(stderr, "fish_tts_amd: batch-1 decode frames: %s\n", ctx->eng_why.c_str());
    return FT_OK;
}

extern "C" ft_status ft_finalize_weights(ft_ctx* ctx) {
    if (!ctx) return FT_ERR_ARG;
    if (ctx->finalized) return FT_OK;
    FT_HIP(ctx, hipSetDevice(ctx->device));
    for (auto& kv : ctx->expected)
        if (!kv.second.p) return ft_fail(ctx, FT_ERR_MISSING_WEIGHT, "missing weight: " + kv.first);
    if (ctx->has_ar) FT_TRY(ar_finalize(ctx));
    if (ctx->has_codec) FT_TRY(codec_finalize(ctx));
    ctx->finalized = true;
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ launches
struct Launch {
    ft_ctx* ctx;
    hipStream_t s;
    int m0, M;       // batch rows [m0, m0+M)
    int pos_off;     // added to the device position (token-by-token prefill)
    hipError_t err = hipSuccess;
    bool gemv_only = false;  // measurement: enqueue only the weight-streaming GEMV launches of the frame
    bool tail_only = false;  // the frame tail follows a prompt pass: the rows' residual stream is in ctx->x (f32 rows)
    void chk() { hipError_t e = hipGetLastError(); if (e != hipSuccess && err == hipSuccess) err = e; }
};

struct PfX {   // fused RMSNorm hooks of the skinny kernel (codec_kernels.h TapGemmP)
    const void* gain = nullptr;   // normalise the X rows with this gain ...
    const float* ss_in = nullptr; // ... and the producer's partial sums of squares (nblk per row)
    int nblk = 0;
    float* ss_out = nullptr;      // leave this GEMM's own partials for the next consumer
};
static void pf_gemm(Launch& L, const bf16_t* X, long ldx, int S, const void* W, const float* bias, int N, int K,
                    int act, const float* resid, float* out_f32, bf16_t* out_bf, long ldo, int round_out,
                    const PfX& fx = PfX());
// A lock-step batch of >= wide_min rows (bf16): every Linear is ONE M-row MFMA launch (weights read once for the whole
// batch) with the RMSNorm / SwiGLU / residual add folded in (wide_kernels.h): five launches per layer.
static bool wide_batch(const Launch& L) {
    return L.ctx->wide_ok && L.ctx->c.dtype == FT_BF16 && L.M >= L.ctx->wide_min && L.M <= 64 && !L.gemv_only && !L.ctx->prof;
}

// Codebook positions 0 and 1 of a wide batch in ONE pass of 2 M rows (both inputs are known once the semantic token is drawn:
// the slow stack's hidden state and the drawn code's embedding, inference.py:116-131): rows xo_pair + m hold position 1.
static bool wide_pair(const Launch& L) {
    const ft_ctx* ctx = L.ctx;
    return wide_batch(L) && !ctx->no_pair && ctx->c.num_codebooks >= 2 && ctx->c.n_fast_layer > 0 && ctx->xo_pair + L.M <= 64;
}

// One Linear of a lock-step batch.  Tile choices per shape class measured with tools/mb_wide.hip (profiles/r04_mb_wide.txt):
// two 16-row weight tiles per workgroup where the fused norm's arithmetic would otherwise be paid per 16 rows of a long
// matrix (W13 always, Wqkv from 17 batch rows), both 16-row batch tiles in one workgroup only where the weights
// dominate (W13, the vocabulary head); everything else splits the batch rows over workgroups.
static void wide_gemm(Launch& L, const bf16_t* X, const void* W, const float* bias, int N, int K, const void* gain, int epi,
                      float* out_f32, long ldo, bf16_t* out_xo, const bf16_t* resid_xo, int rows = 0, bool vocab_head = false) {
    const int M = rows > 0 ? rows : L.M;
    WideP p{};
    p.X = X; p.ldm = L.ctx->xo_ldm; p.W = (const bf16_t*)W; p.ldw = K; p.gain = (const bf16_t*)gain; p.eps = L.ctx->c.norm_eps;
    p.bias = bias; p.M = M; p.N = N; p.K = K; p.out_f32 = out_f32; p.ldo = ldo; p.out_xo = out_xo; p.ldm_o = L.ctx->xo_ldm;
    p.resid_xo = resid_xo;
    bool ok;
    if (epi == WEPI_RESID) ok = wide_gemm_launch<1, 1, false, WEPI_RESID>(p, L.s);
    else if (epi == WEPI_SWIGLU) ok = M > 16 ? wide_gemm_launch<2, 2, true, WEPI_SWIGLU>(p, L.s) : wide_gemm_launch<1, 2, true, WEPI_SWIGLU>(p, L.s);
    else if (vocab_head && M <= 32 && K == 1024 && N % 16 == 0 && N >= 4096 && !L.ctx->no_head_stream) {
        // the vocabulary head: activations normalised once per workgroup, weights streamed per wave (wide_head_kernel)
        static DevOnce once;
        once.run([] { hipFuncSetAttribute((const void*)wide_head_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_head_lds<1024>()); });
        wide_head_kernel<1024><<<256, 512, wide_head_lds<1024>(), L.s>>>(p);
        ok = true;
    }
    else if (N >= 32768) ok = M > 16 ? wide_gemm_launch<2, 2, true, WEPI_STORE>(p, L.s) : wide_gemm_launch<1, 2, true, WEPI_STORE>(p, L.s);
    else if (N >= 4096 && M > 16) ok = wide_gemm_launch<1, 2, true, WEPI_STORE>(p, L.s);
    else ok = wide_gemm_launch<1, 1, true, WEPI_STORE>(p, L.s);
    if (!ok && L.err == hipSuccess) L.err = hipErrorInvalidValue;
    L.chk();
}

// f32 rows -> octet-major bf16 (the values are bf16-exact): the residual stream a prompt pass left in ctx->x
static __global__ __launch_bounds__(256) void xo_from_rows_kernel(const float* x, int ldx, int D, bf16_t* xo, int ldm) {
    const int m = blockIdx.y;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < D; d += gridDim.x * 256) xo[xo_index(m, d, ldm)] = f32_to_bf16_bits(x[(size_t)m * ldx + d]);
}

// bf16 rows [row][D] -> octet-major operand rows (table build below)
static __global__ __launch_bounds__(256) void xo_from_bf16_rows_kernel(const bf16_t* x, int D, bf16_t* xo, int ldm, int rows) {
    const int m = blockIdx.y;
    if (m >= rows) return;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < D; d += gridDim.x * 256) xo[xo_index(m, d, ldm)] = x[(size_t)m * D + d];
}
static __global__ void f32_to_bf16_kernel(const float* x, bf16_t* o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = f32_to_bf16_bits(x[i]);
}

// Wide batches: the table of layer 0's q k v for every code a codebook draw can select (codes < fastV), built at load with
// the very launch that would compute those rows in a frame (the rows of a lock-step GEMM do not depend on each other, so a
// table row carries the bits the launch would produce): 4 MB at the s1-mini widths, one launch less per codebook step.
static ft_status wide_qkv0_build(ft_ctx* ctx) {
    const ft_ar_config& c = ctx->c;
    if (!ctx->wide_ok || c.dtype != FT_BF16 || getenv("FT_NO_QKV0") || c.num_codebooks <= 2 || c.n_fast_layer < 1) return FT_OK;
    const int Df = c.fast_dim, qkvN = (c.fast_n_head + 2 * c.fast_n_local_heads) * c.fast_head_dim, V = ctx->fastV;
    const FtLayer& l0 = ctx->flayers[0];
    if (l0.bqkv) return FT_OK;
    constexpr int CH = 64;
    bf16_t* xo = nullptr; float* out = nullptr;
    if (hipMalloc((void**)&xo, (size_t)Df * CH * sizeof(bf16_t)) != hipSuccess || hipMalloc((void**)&out, (size_t)CH * qkvN * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&ctx->wide_qkv0_tab, (size_t)V * qkvN * sizeof(bf16_t)) != hipSuccess) {
        (void)hipGetLastError();
        if (xo) hipFree(xo);
        if (out) hipFree(out);
        if (ctx->wide_qkv0_tab) { hipFree(ctx->wide_qkv0_tab); ctx->wide_qkv0_tab = nullptr; }
        return FT_OK;      // the frames then keep their layer-0 launch
    }
    bool ok = true;
    for (int c0 = 0; c0 < V && ok; c0 += CH) {
        const int rows = std::min(CH, V - c0);
        xo_from_bf16_rows_kernel<<<dim3((Df + 255) / 256, rows), 256, 0, ctx->stream>>>((const bf16_t*)ctx->fast_emb + (size_t)c0 * Df, Df, xo, CH, rows);
        WideP p{};
        p.X = xo; p.ldm = CH; p.W = (const bf16_t*)l0.wqkv; p.ldw = Df; p.gain = (const bf16_t*)l0.attn_norm; p.eps = c.norm_eps;
        p.M = rows; p.N = qkvN; p.K = Df; p.out_f32 = out; p.ldo = qkvN;
        ok = wide_gemm_launch<1, 1, true, WEPI_STORE>(p, ctx->stream);
        f32_to_bf16_kernel<<<256, 256, 0, ctx->stream>>>(out, ctx->wide_qkv0_tab + (size_t)c0 * qkvN, (long)rows * qkvN);
    }
    ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
    hipFree(xo); hipFree(out);
    if (!ok) { hipFree(ctx->wide_qkv0_tab); ctx->wide_qkv0_tab = nullptr; (void)hipGetLastError(); }
    return FT_OK;
}

static bool eng_slow_ok(const Launch& L) {
    const ft_ctx* ctx = L.ctx;
    return ctx->eng_on && !ctx->eng_suspended && L.M == 1 && !L.gemv_only && !ctx->prof && ctx->c.n_local_heads * ctx->nsplit <= ctx->eng_nb &&
           ctx->c.head_dim % (4 * ctx->nsplit) == 0;
}

static bool eng_fast_ok(const Launch& L) {
    const ft_ctx* ctx = L.ctx;
    return ctx->eng_fast_on && !ctx->eng_suspended && L.M == 1 && !L.gemv_only && !ctx->prof;
}

// The whole codebook loop of one frame (steps 0 .. num_codebooks-1 with their draws) as one launch; runs after the
// vocabulary head + semantic draw launches, which leave the hidden state, the first code's embedding and tokn[0..1].
static void enqueue_fast_engine(Launch& L) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    const int m0 = L.m0, R = c.num_codebooks + 1, nb = ctx->eng_nb;
    const size_t nLf = c.n_fast_layer, VW = (size_t)nb * ENG_LINE;
    FastEngP p{};
    p.layers = ctx->eng_flayers; p.n_layer = c.n_fast_layer; p.ncb = c.num_codebooks;
    p.D = c.fast_dim; p.H = c.fast_n_head; p.Hkv = c.fast_n_local_heads; p.hd = c.fast_head_dim; p.F = c.fast_intermediate_size;
    p.qkvN = (c.fast_n_head + 2 * c.fast_n_local_heads) * c.fast_head_dim; p.V = ctx->fastV;
    p.eps = c.norm_eps; p.scale = (float)(1.0 / sqrt((double)c.fast_head_dim));
    p.rope = ctx->frope; p.fast_norm = (const bf16_t*)ctx->fast_norm; p.fast_out = (const bf16_t*)ctx->fast_out;
    p.fast_emb = (const bf16_t*)ctx->fast_emb;
    p.qkv0_tab = (const bf16_t*)ctx->eng_qkv0_tab;
    p.hid = ctx->hid + (size_t)m0 * c.fast_dim; p.femb = ctx->femb + (size_t)m0 * c.fast_dim;
    unsigned* g = ctx->eng_fast_g;
    p.gx = g; g += 2 * (nLf + 1) * VW;
    p.gqkv = g; g += 2 * nLf * VW;
    p.gxb = g; g += 2 * nLf * VW;
    p.gg = g; g += 2 * nLf * VW;
    p.glog = g; g += 2 * VW;
    p.gcode = g;
    p.ctl = ctx->eng_ctl;
    p.rep_delta0 = ctx->eng_relay ? (long)ctx->eng_fast_words : 0; p.rep_stride = ctx->eng_relay ? (long)ctx->eng_fast_words : 0;
    SampP s{};
    s.logits = nullptr; s.ldl = ctx->fastV; s.V = ctx->fastV;
    s.ctl = ctx->d_ctl + m0; s.tokn = ctx->d_tokn + (size_t)m0 * R; s.seq = ctx->d_seq + (size_t)m0 * R * ctx->cap;
    s.cap = ctx->cap; s.nf = ctx->d_nf + m0; s.cb = 1; s.ncb = c.num_codebooks; s.sem_begin = c.semantic_begin_id;
    s.im_end = c.im_end_id; s.cbsize = c.codebook_size; s.fast_emb = ctx->fast_emb;
    s.femb = ctx->femb + (size_t)m0 * c.fast_dim; s.Df = c.fast_dim; s.noise = ctx->noise;
    s.noise_row_len = ctx->noise_row_len; s.noise_rows = ctx->noise_rows; s.noise_off = 0; s.last = 0;
    s.tok = ctx->d_tok + (size_t)m0 * R; s.pos = ctx->d_pos + m0; s.done = ctx->d_done + m0;
    p.samp = s;
    p.noise_cb_stride = ctx->fastV; p.noise_off1 = c.vocab_size;
    p.pair = ctx->eng_pair ? 1 : 0;
    ((eng_fast_fn_t)ctx->eng_fast_fn)<<<nb, ENG_THREADS, ctx->eng_lds_fast, L.s>>>(p);
    L.chk();
}

static void enqueue_slow_engine(Launch& L, const int* toks, long tok_row_stride, long tok_m_stride, int col) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    const int m0 = L.m0;
    SlowEngP p{};
    p.layers = ctx->eng_layers; p.n_layer = c.n_layer;
    p.D = c.dim; p.H = c.n_head; p.Hkv = c.n_local_heads; p.hd = c.head_dim; p.F = c.intermediate_size;
    p.qkvN = (c.n_head + 2 * c.n_local_heads) * c.head_dim;
    p.eps = c.norm_eps; p.scale = 1.0f / sqrtf((float)c.head_dim);
    p.emb = (const bf16_t*)ctx->emb; p.cb_emb = (const bf16_t*)ctx->cb_emb;
    p.toks = toks + (size_t)m0 * tok_m_stride + col; p.tok_row_stride = tok_row_stride;
    p.ncb = c.num_codebooks; p.cbsize = c.codebook_size; p.vocab = c.vocab_size; p.sem_begin = c.semantic_begin_id;
    p.sem_end = c.semantic_end_id; p.scale_cb = c.scale_codebook_embeddings; p.inv_div = (float)sqrt((double)(c.num_codebooks + 1));
    p.rope = ctx->rope; p.pos = ctx->d_pos + m0; p.pos_off = L.pos_off; p.n_slots = ctx->n_slots; p.nsplit = ctx->nsplit;
    p.cache_off = (size_t)m0 * ctx->cache_m_stride;
    p.gx = ctx->eng_gx; p.gqkv = ctx->eng_gqkv; p.gpart = ctx->eng_gpart; p.gy = ctx->eng_gy; p.gxb = ctx->eng_gxb; p.gg = ctx->eng_gg;
    p.ctl = ctx->eng_ctl; p.x_out = ctx->x + (size_t)m0 * c.dim; p.nt = 1;
    p.rep_delta0 = ctx->eng_relay ? (long)ctx->eng_pool_words : 0; p.rep_stride = ctx->eng_relay ? (long)ctx->eng_pool_words : 0;
    ((eng_slow_fn_t)ctx->eng_slow_fn)<<<ctx->eng_nb, ENG_THREADS, ctx->eng_lds_slow, L.s>>>(p);
    L.chk();
}


template <typename WT, int ROUND, int R>
static void gemv_nt(Launch& L, const GemvP& p, int nt) {
    const dim3 grid((p.N + 4 * R - 1) / (4 * R), L.M), block(256);
#define FT_NT(n) case n: gemv_kernel<WT, n, R, ROUND><<<grid, block, 0, L.s>>>(p); break;
    switch (nt) { FT_NT(1) FT_NT(2) FT_NT(3) FT_NT(4) FT_NT(6) FT_NT(8) FT_NT(12) default: L.err = hipErrorInvalidValue; }
#undef FT_NT
}

static int pick_nt(int K, int vec) {
    const int need = (K + 64 * vec - 1) / (64 * vec);
    const int opts[] = {1, 2, 3, 4, 6, 8, 12};
    for (int o : opts) if (o >= need) return o;
    return -1;
}

template <typename WT, int ROUND, int R, int MB>
static void gemv_mb_nt(Launch& L, const GemvP& p, int nt) {
    const dim3 grid((p.N + 4 * R - 1) / (4 * R), (L.M + MB - 1) / MB), block(256);
#define FT_NT(n) case n: gemv_mb_kernel<WT, n, R, MB, ROUND><<<grid, block, 0, L.s>>>(p, L.M); break;
    switch (nt) { FT_NT(1) FT_NT(2) FT_NT(4) FT_NT(6) default: L.err = hipErrorInvalidValue; }
#undef FT_NT
}

template <typename WT, int ROUND>
static void gemv(Launch& L, GemvP p, int R) {
    const int nt = pick_nt(p.K, Vec<WT>::N);
    if (nt < 0) { L.err = hipErrorInvalidValue; return; }
    ft_ctx* ctx = L.ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof && !ctx->prof_count_only) {
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, L.s);
    }
    if (p.epi == EPI_SWIGLU && R < 2) R = 2;
    // lock-step batches (bf16): several utterance rows per pass over the weights
    const bool mb_ok = sizeof(WT) == 2 && L.M >= 2 && (nt == 1 || nt == 2 || nt == 4 || nt == 6);
    if (mb_ok) {
        if constexpr (sizeof(WT) == 2) {
            const bool four = L.M >= 3 && nt <= 2 && R <= 2;  // register budget: 4 x NT x 8 activations
            if (R >= 4) { if (nt <= 2) gemv_mb_nt<WT, ROUND, 4, 2>(L, p, nt); else gemv_nt<WT, ROUND, 4>(L, p, nt); }
            else if (R == 2) { if (four) gemv_mb_nt<WT, ROUND, 2, 4>(L, p, nt); else gemv_mb_nt<WT, ROUND, 2, 2>(L, p, nt); }
            else { if (four) gemv_mb_nt<WT, ROUND, 1, 4>(L, p, nt); else gemv_mb_nt<WT, ROUND, 1, 2>(L, p, nt); }
        }
    } else if (R == 1) gemv_nt<WT, ROUND, 1>(L, p, nt);
    else if (R == 2) gemv_nt<WT, ROUND, 2>(L, p, nt);
    else if (R >= 8 && nt <= 2 && p.epi != EPI_SWIGLU) gemv_nt<WT, ROUND, 8>(L, p, nt);
    else gemv_nt<WT, ROUND, 4>(L, p, nt);
    if (ctx->prof) {
        if (!ctx->prof_count_only) {
            hipEventRecord(e1, L.s);
            ctx->prof_ev.push_back(e0); ctx->prof_ev.push_back(e1);
        }
        ctx->prof_bytes += (int64_t)p.N * p.K * sizeof(WT);
        ctx->prof_launches += 1;
    }
    L.chk();
}

static int rows_per_wave(int N, int M) {
    // enough waves to cover the chip (256 CUs x 4 SIMDs) a few times over; big matrices amortise
    const long waves1 = (long)N * M;
    // (the vocabulary head, 155 776 rows at M = 1: 8 rows per wave - 16 KB in flight - measured 0.5 % slower than 4)
    if (waves1 >= 65536) return 4;
    if (waves1 >= 2048) return 2;
    return 1;
}

template <typename WT, int ROUND>
static void attn_decode(Launch& L, const AttnP& p) {
    ft_ctx* ctx = L.ctx;
    const int G = p.H / p.Hkv;
    const int nslot = 2048 / p.hd;
    const size_t lds = ((size_t)G * p.hd + 2 * p.hd + (size_t)nslot * G * 2 + (size_t)nslot * G * p.hd) * sizeof(float);
    const dim3 grid(p.Hkv, p.nsplit, L.M), block(256);
    switch (G) {
        case 1: attn_decode_kernel<WT, 1, ROUND><<<grid, block, lds, L.s>>>(p); break;
        case 2: attn_decode_kernel<WT, 2, ROUND><<<grid, block, lds, L.s>>>(p); break;
        case 4: attn_decode_kernel<WT, 4, ROUND><<<grid, block, lds, L.s>>>(p); break;
        case 8: {
            static DevOnce once;     // per device (the opt-in belongs to the current device's function object)
            once.run([&] { hipFuncSetAttribute((const void*)attn_decode_kernel<WT, 8, ROUND>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            attn_decode_kernel<WT, 8, ROUND><<<grid, block, lds, L.s>>>(p);
            break;
        }
        default: L.err = hipErrorInvalidValue;
    }
    L.chk();
    (void)ctx;
}

template <typename WT, int ROUND, int R>
static void gemv_combine_nt(Launch& L, const GemvP& p, const AttnP& a, int nt) {
    const dim3 grid((p.N + 4 * R - 1) / (4 * R), L.M), block(256);
#define FT_NT(n) case n: gemv_attn_combine_kernel<WT, n, R, ROUND><<<grid, block, (size_t)a.H * a.hd * sizeof(float), L.s>>>(p, a); break;
    switch (nt) { FT_NT(1) FT_NT(2) FT_NT(3) FT_NT(4) FT_NT(6) FT_NT(8) FT_NT(12) default: L.err = hipErrorInvalidValue; }
#undef FT_NT
}

// fast_project_in on the pre-norm hidden state (llama.py:453,590); identity when fast_dim == dim
template <typename WT, int ROUND>
static void enqueue_fproj(Launch& L) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    if (c.fast_dim == c.dim) return;
    GemvP q{};
    q.W = ctx->fproj_w; q.bias = ctx->fproj_b; q.x = ctx->x + (size_t)L.m0 * c.dim; q.ldx = c.dim;
    q.out = ctx->hid + (size_t)L.m0 * c.fast_dim; q.ldo = c.fast_dim; q.N = c.fast_dim; q.K = c.dim;
    q.pro = PRO_NONE; q.epi = EPI_STORE;
    gemv<WT, ROUND>(L, q, rows_per_wave(q.N, L.M));
}

// One slow-transformer pass over the current input column of rows [m0, m0+M) (llama.py:400-453).
template <typename WT, int ROUND>
static void enqueue_slow(Launch& L, const int* toks, long tok_row_stride, long tok_m_stride, int col, bool with_head) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    const int m0 = L.m0;
    const size_t qkvN = (size_t)(c.n_head + 2 * c.n_local_heads) * c.head_dim;
    float* x = ctx->x + (size_t)m0 * c.dim;
    float* qkv = ctx->qkv + (size_t)m0 * qkvN;
    float* y = ctx->y + (size_t)m0 * ctx->y_ld;
    float* g = ctx->g + (size_t)m0 * c.intermediate_size;

    EmbedP e{};
    e.emb = ctx->emb; e.cb_emb = ctx->cb_emb; e.toks = toks; e.tok_row_stride = tok_row_stride;
    e.tok_m_stride = tok_m_stride; e.col = col; e.x = x; e.ldx = c.dim; e.D = c.dim; e.ncb = c.num_codebooks;
    e.cbsize = c.codebook_size; e.vocab = c.vocab_size; e.sem_begin = c.semantic_begin_id;
    e.sem_end = c.semantic_end_id; e.scale = c.scale_codebook_embeddings;
    e.inv_div = (float)sqrt((double)(c.num_codebooks + 1));
    const bool wide = wide_batch(L);
    if (wide) { e.xo = ctx->xo_x + (size_t)m0 * 8; e.xo_ldm = ctx->xo_ldm; }
    if (!L.gemv_only) embed_kernel<WT, ROUND><<<dim3((c.dim + 255) / 256, L.M), 256, 0, L.s>>>(e);
    L.chk();

    for (int li = 0; li < c.n_layer; ++li) {
        const FtLayer& l = ctx->layers[li];
        if (wide) {
            if constexpr (ROUND == RND_BF16) {
                // five launches per layer (wide_kernels.h): the residual stream xo, the attention output and the SwiGLU vector
                // travel octet-major in bf16; q k v stay f32 rows for the attention kernel
                const int D = c.dim, HD = c.n_head * c.head_dim, F = c.intermediate_size, M = L.M, ldm = ctx->xo_ldm;
                bf16_t* xo = ctx->xo_x + (size_t)m0 * 8;
                bf16_t* yo = ctx->xo_y + (size_t)m0 * 8;
                bf16_t* go = ctx->xo_g + (size_t)m0 * 8;
                wide_gemm(L, xo, l.wqkv, l.bqkv_f32, (int)qkvN, D, l.attn_norm, WEPI_STORE, qkv, (long)qkvN, nullptr, nullptr);
                AttnP a{};
                a.qkv = qkv; a.ldq = (int)qkvN; a.qn = l.qn; a.kn = l.kn; a.rope = ctx->rope;
                a.kc = (char*)l.kc + (size_t)m0 * ctx->cache_m_stride * ctx->esz;
                a.vc = (char*)l.vc + (size_t)m0 * ctx->cache_m_stride * ctx->esz;
                a.cache_m_stride = ctx->cache_m_stride; a.pos = ctx->d_pos + m0; a.pos_off = L.pos_off;
                a.H = c.n_head; a.Hkv = c.n_local_heads; a.hd = c.head_dim; a.n_slots = ctx->n_slots;
                a.eps = c.norm_eps; a.scale = 1.0f / sqrtf((float)c.head_dim);
                a.y = nullptr; a.ldy = HD; a.y_bf = yo; a.y_xo_ldm = ldm;
                // >= 128 (row, kv head) blocks: each walks its row's whole context with the two-pass kernel; fewer rows split
                // the cache walk until M x Hkv x nsplit blocks cover the chip (long contexts of few rows: voice prompts)
                const int Gq = c.n_head / c.n_local_heads;
                const size_t wlds = (size_t)attn_wide_lds_floats(Gq, c.head_dim, ctx->n_slots) * sizeof(float);
                if (!ctx->no_attn_wide && (long)M * c.n_local_heads >= 128 && c.head_dim == 128 && (Gq == 1 || Gq == 2 || Gq == 4) && wlds <= 65536) {
                    a.nsplit = 1;
                    const dim3 grid(c.n_local_heads, 1, M);
                    if (Gq == 1) attn_wide_kernel<1, 128><<<grid, 256, wlds, L.s>>>(a, ctx->n_slots);
                    else if (Gq == 2) attn_wide_kernel<2, 128><<<grid, 256, wlds, L.s>>>(a, ctx->n_slots);
                    else attn_wide_kernel<4, 128><<<grid, 256, wlds, L.s>>>(a, ctx->n_slots);
                    L.chk();
                } else {
                    int ns = 1;
                    while (ns < ctx->nsplit && (long)M * c.n_local_heads * ns < 256) ns *= 2;
                    a.nsplit = ns;
                    a.part_o = ctx->part_o + (size_t)m0 * c.n_head * ctx->nsplit * c.head_dim;
                    a.part_ml = ctx->part_ml + (size_t)m0 * c.n_head * ctx->nsplit * 2;
                    attn_decode<WT, ROUND>(L, a);
                    if (ns > 1) { attn_combine_rows_kernel<ROUND><<<M, 256, 0, L.s>>>(a); L.chk(); }
                }
                wide_gemm(L, yo, l.wo, l.bo_f32, D, HD, nullptr, WEPI_RESID, nullptr, 0, xo, xo);
                wide_gemm(L, xo, l.w13, nullptr, 2 * F, D, l.ffn_norm, WEPI_SWIGLU, nullptr, 0, go, nullptr);
                wide_gemm(L, go, l.w2, nullptr, D, F, nullptr, WEPI_RESID, nullptr, 0, xo, xo);
            }
            continue;
        }
        GemvP p{};
        p.W = l.wqkv; p.bias = l.bqkv; p.x = x; p.ldx = c.dim; p.gain = l.attn_norm; p.eps = c.norm_eps;
        p.out = qkv; p.ldo = (int)qkvN; p.N = (int)qkvN; p.K = c.dim; p.pro = PRO_RMSNORM; p.epi = EPI_STORE; p.nt = 1;
        gemv<WT, ROUND>(L, p, rows_per_wave(p.N, L.M));

        AttnP a{};
        a.qkv = qkv; a.ldq = (int)qkvN; a.qn = l.qn; a.kn = l.kn; a.rope = ctx->rope;
        a.kc = (char*)l.kc + (size_t)m0 * ctx->cache_m_stride * ctx->esz;
        a.vc = (char*)l.vc + (size_t)m0 * ctx->cache_m_stride * ctx->esz;
        a.cache_m_stride = ctx->cache_m_stride; a.pos = ctx->d_pos + m0; a.pos_off = L.pos_off;
        a.H = c.n_head; a.Hkv = c.n_local_heads; a.hd = c.head_dim; a.n_slots = ctx->n_slots;
        a.nsplit = ctx->nsplit; a.eps = c.norm_eps; a.scale = 1.0f / sqrtf((float)c.head_dim);
        a.y = y; a.ldy = ctx->y_ld;
        a.part_o = ctx->part_o + (size_t)m0 * c.n_head * ctx->nsplit * c.head_dim;
        a.part_ml = ctx->part_ml + (size_t)m0 * c.n_head * ctx->nsplit * 2;
        if (!L.gemv_only) attn_decode<WT, ROUND>(L, a);

        GemvP o{};
        o.W = l.wo; o.bias = l.bo; o.x = y; o.ldx = ctx->y_ld; o.out = x; o.ldo = c.dim;
        o.resid = x; o.ldr = c.dim; o.N = c.dim; o.K = c.n_head * c.head_dim; o.pro = PRO_NONE; o.epi = EPI_RESID; o.nt = 1;
        if (ctx->nsplit > 1) {  // split-KV partials are merged inside the Wo kernel
            const int nt = pick_nt(o.K, Vec<WT>::N);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (ctx->prof && !ctx->prof_count_only) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, L.s); }
            if (rows_per_wave(o.N, L.M) >= 2) gemv_combine_nt<WT, ROUND, 2>(L, o, a, nt);
            else gemv_combine_nt<WT, ROUND, 1>(L, o, a, nt);
            if (ctx->prof) {
                if (!ctx->prof_count_only) {
                    hipEventRecord(e1, L.s);
                    ctx->prof_ev.push_back(e0); ctx->prof_ev.push_back(e1);
                }
                ctx->prof_bytes += (int64_t)o.N * o.K * sizeof(WT);
                ctx->prof_launches += 1;
            }
            L.chk();
        } else {
            gemv<WT, ROUND>(L, o, rows_per_wave(o.N, L.M));
        }

        GemvP f{};
        f.W = l.w13; f.x = x; f.ldx = c.dim; f.gain = l.ffn_norm; f.eps = c.norm_eps; f.out = g;
        f.ldo = c.intermediate_size; f.N = 2 * c.intermediate_size; f.K = c.dim; f.pro = PRO_RMSNORM; f.epi = EPI_SWIGLU; f.nt = 1;
        gemv<WT, ROUND>(L, f, rows_per_wave(f.N, L.M));

        GemvP d{};
        d.W = l.w2; d.x = g; d.ldx = c.intermediate_size; d.out = x; d.ldo = c.dim; d.resid = x; d.ldr = c.dim;
        d.N = c.dim; d.K = c.intermediate_size; d.pro = PRO_NONE; d.epi = EPI_RESID; d.nt = 1;
        gemv<WT, ROUND>(L, d, rows_per_wave(d.N, L.M));
    }
    if (with_head) enqueue_fproj<WT, ROUND>(L);
}

// final norm + vocabulary head (llama.py:446-451)
template <typename WT, int ROUND>
static void enqueue_head(Launch& L) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    if (wide_batch(L)) {
        if (L.tail_only) {
            xo_from_rows_kernel<<<dim3((c.dim + 255) / 256, L.M), 256, 0, L.s>>>(ctx->x + (size_t)L.m0 * c.dim, c.dim, c.dim,
                                                                             ctx->xo_x + (size_t)L.m0 * 8, ctx->xo_ldm);
            L.chk();
        }
        if constexpr (ROUND == RND_BF16)
            wide_gemm(L, ctx->xo_x + (size_t)L.m0 * 8, ctx->head, nullptr, c.vocab_size, c.dim, ctx->norm, WEPI_STORE,
                      ctx->logits + (size_t)L.m0 * c.vocab_size, c.vocab_size, nullptr, nullptr, 0, true);
        return;
    }
    GemvP h{};
    h.W = ctx->head; h.x = ctx->x + (size_t)L.m0 * c.dim; h.ldx = c.dim; h.gain = ctx->norm; h.eps = c.norm_eps;
    h.out = ctx->logits + (size_t)L.m0 * c.vocab_size; h.ldo = c.vocab_size; h.N = c.vocab_size; h.K = c.dim;
    h.pro = PRO_RMSNORM; h.epi = EPI_STORE; h.nt = 1;
    gemv<WT, ROUND>(L, h, rows_per_wave(h.N, L.M));
}

template <typename WT, int ROUND>
static void enqueue_sample(Launch& L, int cb, bool last) {
    if (L.gemv_only) return;
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    const int m0 = L.m0, R = c.num_codebooks + 1;
    SampP s{};
    if (cb == 0) { s.logits = ctx->logits + (size_t)m0 * c.vocab_size; s.ldl = c.vocab_size; s.V = c.vocab_size; }
    else { s.logits = ctx->flog + (size_t)m0 * ctx->fastV; s.ldl = ctx->fastV; s.V = ctx->fastV; }
    s.ctl = ctx->d_ctl + m0; s.tokn = ctx->d_tokn + (size_t)m0 * R; s.seq = ctx->d_seq + (size_t)m0 * R * ctx->cap;
    s.cap = ctx->cap; s.nf = ctx->d_nf + m0; s.cb = cb; s.ncb = c.num_codebooks; s.sem_begin = c.semantic_begin_id;
    s.im_end = c.im_end_id; s.cbsize = c.codebook_size; s.fast_emb = ctx->fast_emb;
    s.femb = ctx->femb + (size_t)m0 * c.fast_dim; s.Df = c.fast_dim; s.noise = ctx->noise;
    s.noise_row_len = ctx->noise_row_len; s.noise_rows = ctx->noise_rows;
    s.noise_off = cb == 0 ? 0 : (long)c.vocab_size + (long)(cb - 1) * ctx->fastV;
    s.last = last ? 1 : 0; s.tok = ctx->d_tok + (size_t)m0 * R; s.pos = ctx->d_pos + m0; s.done = ctx->d_done + m0;
    if (wide_batch(L)) {
        // the semantic code's embedding is position 1 of the paired pass: it joins the slow stack's residual stream
        s.femb_xo = (cb == 0 && wide_pair(L) ? ctx->xo_x + (size_t)ctx->xo_pair * 8 : ctx->xo_femb) + (size_t)m0 * 8;
        s.femb_ldm = ctx->xo_ldm;
        if (cb >= 1 && cb + 1 < c.num_codebooks && ctx->wide_qkv0_tab) {     // the next codebook step's layer-0 q k v (its launch is skipped)
            s.qkv0_tab = ctx->wide_qkv0_tab; s.qkv0_n = (c.fast_n_head + 2 * c.fast_n_local_heads) * c.fast_head_dim;
            s.qkv0_out = ctx->qkvf + (size_t)m0 * s.qkv0_n;
        }
    }
    if (s.V <= 1024) {
        sample_small_kernel<WT, ROUND><<<L.M, 256, 0, L.s>>>(s);
    } else if (ROUND == RND_BF16) {
        // large vocabulary, bf16: the histogram of the 65 536 bf16 classes and the search of the top-p cut on it in ONE block
        // per row (LDS counters), then index-ordered tie handling and the chip-wide race
        SampBigP b{};
        b.s = s; b.nchunk = (s.V + 1023) / 1024;
        b.cut = ctx->samp_cut + m0;
        b.chunk_cnt = ctx->samp_chunk_cnt + (size_t)m0 * b.nchunk;
        b.part_score = ctx->samp_part_score + (size_t)m0 * b.nchunk;
        b.part_idx = ctx->samp_part_idx + (size_t)m0 * b.nchunk;
        const dim3 gridc(b.nchunk, L.M);
        {
            static DevOnce once;     // per device
            once.run([] {
                hipFuncSetAttribute((const void*)samp_cut_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMP_TH_LDS);
            });
        }
        samp_cut_kernel<<<L.M, SAMP_TH_THREADS, SAMP_TH_LDS, L.s>>>(b);
        samp_count_kernel<<<gridc, 256, 0, L.s>>>(b);
        samp_race_kernel<<<gridc, 256, 0, L.s>>>(b);
        samp_finish_kernel<WT><<<L.M, 256, 0, L.s>>>(b);
    } else {
        sample_block_kernel<WT, ROUND><<<L.M, 1024, 0, L.s>>>(s);
    }
    L.chk();
}

// The fast transformer over codebook positions 0..ncb-1 with its sampling (inference.py:115-149).
template <typename WT, int ROUND>
static void enqueue_fast_step(Launch& L, const int cb, const bool pair = false) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    const int m0 = L.m0;
    const int Df = c.fast_dim, Hf = c.fast_n_head, Hkvf = c.fast_n_local_heads, hdf = c.fast_head_dim;
    const size_t qkvN = (size_t)(Hf + 2 * Hkvf) * hdf;
    float* xf = ctx->xf + (size_t)m0 * Df;
    float* qkvf = ctx->qkvf + (size_t)m0 * qkvN;
    float* gf = ctx->gf + (size_t)m0 * c.fast_intermediate_size;
    {
        const float* xin = cb == 0 ? ctx->hid + (size_t)m0 * Df : ctx->femb + (size_t)m0 * Df;
        const bool wide = wide_batch(L);
        for (int li = 0; li < c.n_fast_layer; ++li) {
            const FtLayer& l = ctx->flayers[li];
            const float* xl = li == 0 ? xin : xf;
            if (wide) {
                if constexpr (ROUND == RND_BF16) {
                    // as the slow layers; layer 0 reads the slow stack's residual stream (position 0) or the drawn code's embedding
                    // pair: rows [0, M) at position 0 and rows [xo_pair, xo_pair + M) at position 1 (cb == 1) in one pass
                    const int HDf = Hf * hdf, Ff = c.fast_intermediate_size, M = L.M, ldm = ctx->xo_ldm;
                    const int rows = pair ? ctx->xo_pair + M : M;
                    const bf16_t* xin_o = ((cb == 0 || pair) ? ctx->xo_x : ctx->xo_femb) + (size_t)m0 * 8;
                    bf16_t* xfo = ctx->xo_xf + (size_t)m0 * 8;
                    bf16_t* yo = ctx->xo_y + (size_t)m0 * 8;
                    bf16_t* go = ctx->xo_g + (size_t)m0 * 8;
                    const bf16_t* xlo = li == 0 ? xin_o : xfo;
                    // layer 0 of steps >= 2: the draw of the previous step left this step's q k v (wide_qkv0_build)
                    if (!(li == 0 && cb >= 2 && !pair && ctx->wide_qkv0_tab))
                        wide_gemm(L, xlo, l.wqkv, nullptr, (int)qkvN, Df, l.attn_norm, WEPI_STORE, qkvf, (long)qkvN, nullptr, nullptr, rows);
                    FastAttnP a{};
                    a.qkv = qkvf; a.ldq = (int)qkvN; a.qn = l.qn; a.kn = l.kn; a.rope = ctx->frope;
                    a.kc = (char*)l.kc + (size_t)m0 * ctx->fcache_m_stride * ctx->esz;
                    a.vc = (char*)l.vc + (size_t)m0 * ctx->fcache_m_stride * ctx->esz;
                    a.cache_m_stride = ctx->fcache_m_stride; a.c = pair ? 0 : cb; a.H = Hf; a.Hkv = Hkvf; a.hd = hdf;
                    a.ncb = c.num_codebooks; a.eps = c.norm_eps; a.scale = (float)(1.0 / sqrt((double)hdf));
                    a.y_bf = yo; a.y_xo_ldm = ldm;
                    a.pair_M = pair ? M : 0; a.pair_off = ctx->xo_pair;
                    fast_attn_kernel<WT, ROUND><<<dim3(Hf, pair ? 2 * M : M), 64, 0, L.s>>>(a, nullptr, HDf);
                    L.chk();
                    wide_gemm(L, yo, l.wo, nullptr, Df, HDf, nullptr, WEPI_RESID, nullptr, 0, xfo, xlo, rows);
                    wide_gemm(L, xfo, l.w13, nullptr, 2 * Ff, Df, l.ffn_norm, WEPI_SWIGLU, nullptr, 0, go, nullptr, rows);
                    wide_gemm(L, go, l.w2, nullptr, Df, Ff, nullptr, WEPI_RESID, nullptr, 0, xfo, xfo, rows);
                }
                continue;
            }
            GemvP p{};
            p.W = l.wqkv; p.bias = l.bqkv; p.x = xl; p.ldx = Df; p.gain = l.attn_norm; p.eps = c.norm_eps;
            p.out = qkvf; p.ldo = (int)qkvN; p.N = (int)qkvN; p.K = Df; p.pro = PRO_RMSNORM; p.epi = EPI_STORE;
            gemv<WT, ROUND>(L, p, rows_per_wave(p.N, L.M));

            FastAttnP a{};
            a.qkv = qkvf; a.ldq = (int)qkvN; a.qn = l.qn; a.kn = l.kn; a.rope = ctx->frope;
            a.kc = (char*)l.kc + (size_t)m0 * ctx->fcache_m_stride * ctx->esz;
            a.vc = (char*)l.vc + (size_t)m0 * ctx->fcache_m_stride * ctx->esz;
            a.cache_m_stride = ctx->fcache_m_stride; a.c = cb; a.H = Hf; a.Hkv = Hkvf; a.hd = hdf;
            a.ncb = c.num_codebooks; a.eps = c.norm_eps; a.scale = (float)(1.0 / sqrt((double)hdf));
            float* yf = ctx->y + (size_t)m0 * ctx->y_ld;  // the slow attention's y buffer is free here
            if (!L.gemv_only) fast_attn_kernel<WT, ROUND><<<dim3(Hf, L.M), 64, 0, L.s>>>(a, yf, ctx->y_ld);
            L.chk();
            GemvP o{};
            o.W = l.wo; o.bias = l.bo; o.x = yf; o.ldx = ctx->y_ld; o.out = xf; o.ldo = Df; o.resid = xl;
            o.ldr = Df; o.N = Df; o.K = Hf * hdf; o.pro = PRO_NONE; o.epi = EPI_RESID;
            gemv<WT, ROUND>(L, o, rows_per_wave(o.N, L.M));

            GemvP f{};
            f.W = l.w13; f.x = xf; f.ldx = Df; f.gain = l.ffn_norm; f.eps = c.norm_eps; f.out = gf;
            f.ldo = c.fast_intermediate_size; f.N = 2 * c.fast_intermediate_size; f.K = Df; f.pro = PRO_RMSNORM;
            f.epi = EPI_SWIGLU;
            gemv<WT, ROUND>(L, f, rows_per_wave(f.N, L.M));

            GemvP d{};
            d.W = l.w2; d.x = gf; d.ldx = c.fast_intermediate_size; d.out = xf; d.ldo = Df; d.resid = xf; d.ldr = Df;
            d.N = Df; d.K = c.fast_intermediate_size; d.pro = PRO_NONE; d.epi = EPI_RESID;
            gemv<WT, ROUND>(L, d, rows_per_wave(d.N, L.M));
        }
        if (cb == 0) return;  // logits of position 0 are discarded (inference.py:122)
        if (wide) {
            if constexpr (ROUND == RND_BF16)
                wide_gemm(L, (c.n_fast_layer > 0 ? ctx->xo_xf : ctx->xo_femb) + (size_t)(m0 + (pair ? ctx->xo_pair : 0)) * 8, ctx->fast_out, nullptr,
                          ctx->fastV, Df, ctx->fast_norm, WEPI_STORE, ctx->flog + (size_t)m0 * ctx->fastV, ctx->fastV, nullptr, nullptr);
            enqueue_sample<WT, ROUND>(L, cb, cb == c.num_codebooks - 1);
            return;
        }
        GemvP h{};
        h.W = ctx->fast_out; h.x = xf; h.ldx = Df; h.gain = ctx->fast_norm; h.eps = c.norm_eps;
        h.out = ctx->flog + (size_t)m0 * ctx->fastV; h.ldo = ctx->fastV; h.N = ctx->fastV; h.K = Df;
        h.pro = PRO_RMSNORM; h.epi = EPI_STORE;
        gemv<WT, ROUND>(L, h, rows_per_wave(h.N, L.M));
        enqueue_sample<WT, ROUND>(L, cb, cb == c.num_codebooks - 1);
    }
}

// One full frame: slow pass on the current column, semantic sample, fast codebooks (inference.py:83-155).
template <typename WT, int ROUND>
static void enqueue_frame_tail(Launch& L);

template <typename WT, int ROUND>
static void enqueue_frame_t(Launch& L, const int* toks, long trs, long tms, int col) {
    if (ROUND == RND_BF16 && eng_slow_ok(L)) enqueue_slow_engine(L, toks, trs, tms, col);
    else enqueue_slow<WT, ROUND>(L, toks, trs, tms, col, true);
    enqueue_frame_tail<WT, ROUND>(L);
}

// everything after the slow layers: vocabulary head, semantic draw, the fast codebooks
template <typename WT, int ROUND>
static void enqueue_frame_tail(Launch& L) {
    ft_ctx* ctx = L.ctx;
    const int ncb = ctx->c.num_codebooks;
    if (L.gemv_only) {  // measurement graph: the HBM-streamed launches only (slow layers above + the head)
        enqueue_head<WT, ROUND>(L);
        return;
    }
    enqueue_head<WT, ROUND>(L);
    enqueue_sample<WT, ROUND>(L, 0, ncb == 1);
    if (ROUND == RND_BF16 && eng_fast_ok(L)) { enqueue_fast_engine(L); return; }
    int cb0 = 0;
    if (wide_pair(L)) { enqueue_fast_step<WT, ROUND>(L, 1, true); cb0 = 2; }
    for (int cb = cb0; cb < ncb; ++cb) enqueue_fast_step<WT, ROUND>(L, cb);
}

static void enqueue_frame(Launch& L, const int* toks, long trs, long tms, int col) {
    if (L.ctx->c.dtype == FT_BF16) enqueue_frame_t<bf16_t, RND_BF16>(L, toks, trs, tms, col);
    else if (L.ctx->c.dtype == FT_F16) enqueue_frame_t<f16_t, RND_F16>(L, toks, trs, tms, col);
    else enqueue_frame_t<float, RND_NONE>(L, toks, trs, tms, col);
}
static void enqueue_slow_only(Launch& L, const int* toks, long trs, long tms, int col) {
    if (L.ctx->c.dtype == FT_BF16) enqueue_slow<bf16_t, RND_BF16>(L, toks, trs, tms, col, false);
    else if (L.ctx->c.dtype == FT_F16) enqueue_slow<f16_t, RND_F16>(L, toks, trs, tms, col, false);
    else enqueue_slow<float, RND_NONE>(L, toks, trs, tms, col, false);
}

// ------------------------------------------------------------------------------------------ AR API
static ft_status eng_recover(ft_ctx* ctx, bool* aborted);
static bool eng_in_use(const ft_ctx* ctx);
static ft_status ar_ready(ft_ctx* ctx) {
    if (!ctx) return FT_ERR_ARG;
    if (!ctx->has_ar) return ft_fail(ctx, FT_ERR_STATE, "context was created without an AR config");
    if (!ctx->finalized) return ft_fail(ctx, FT_ERR_STATE, "weights not finalized (ft_finalize_weights)");
    if (hipSetDevice(ctx->device) != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, "hipSetDevice failed");
    return FT_OK;
}

extern "C" ft_status ft_ar_reset(ft_ctx* ctx, int32_t slot) {
    if (!ctx || !ctx->has_ar) return FT_ERR_ARG;
    if (slot < 0 || slot >= ctx->c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_reset: bad slot");
    FT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t R = ctx->c.num_codebooks + 1;
    FT_HIP(ctx, hipMemsetAsync(ctx->d_seq + (size_t)slot * R * ctx->cap, 0, R * ctx->cap * sizeof(int), ctx->stream));
    FT_HIP(ctx, hipMemsetAsync(ctx->d_pos + slot, 0, sizeof(int), ctx->stream));
    FT_HIP(ctx, hipMemsetAsync(ctx->d_nf + slot, 0, sizeof(int), ctx->stream));
    FT_HIP(ctx, hipMemsetAsync(ctx->d_done + slot, 0, sizeof(int), ctx->stream));
    return FT_OK;
}

// KV splits of the decode attention for a call whose longest context ends at pos_end (more blocks walking the cache in
// parallel); measured at s1-mini shapes (tools/longctx_probe.py): 8 splits are fastest up to ~700 positions, 16 up to ~3000
static void pick_nsplit(ft_ctx* ctx, int pos_end) {
    if (ctx->nsplit_fixed || ctx->nsplit_max <= 1) return;
    const int want = pos_end <= 768 ? 8 : pos_end <= 3072 ? 16 : 32;
    ctx->nsplit = std::min(want, ctx->nsplit_max);
}

static ft_status upload_ctl(ft_ctx* ctx, int m0, int n, const ft_sampling* sp) {
    std::vector<RowCtl> h(n);
    for (int i = 0; i < n; ++i) {
        h[i].temperature = sp[i].temperature; h[i].top_p = sp[i].top_p; h[i].rep = sp[i].repetition_penalty;
        h[i].ban_eos = sp[i].ban_eos; h[i].seed = sp[i].seed;
    }
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_ctl + m0, h.data(), n * sizeof(RowCtl), hipMemcpyHostToDevice, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h goes out of scope
    return FT_OK;
}

// MFMA prefill (bf16 precision): the whole prompt goes through every slow layer as S = Lp rows on the
template <int HD, int NG>
static void flash_launch(const FlashP& fp, int Lp, hipStream_t st, int nz = 1) {
    constexpr size_t lds = flash_prefill_lds<HD, NG>();
    static DevOnce once;
    once.run([] { hipFuncSetAttribute((const void*)flash_prefill_kernel<HD, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    constexpr int QROWS = 16 * (8 / NG);
    flash_prefill_kernel<HD, NG><<<dim3(fp.H, (Lp + QROWS - 1) / QROWS, nz), 512, lds, st>>>(fp);
}

template <int BM, int BN, int NWM, int NWN, int DEPTH = 4, int MINW = 1>
static void lingemm_launch(const TapGemmP& p, int S, int N, hipStream_t st) {
    constexpr size_t lds = lingemm_lds_bytes<BM, BN, NWM>();
    static DevOnce once;     // per device: the dynamic-LDS opt-in belongs to the device's function object
    once.run([] { hipFuncSetAttribute((const void*)lingemm_kernel<BM, BN, NWM, NWN, DEPTH, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    lingemm_kernel<BM, BN, NWM, NWN, DEPTH, MINW><<<dim3((S + BM - 1) / BM, N / BN), 64 * NWM * NWN, lds, st>>>(p);
}

// tap-GEMM kernel (v_mfma_f32_16x16x32_bf16), with the reference's rounding points in the epilogues
// (Linear output rounded, residual add rounded, SwiGLU steps rounded; llama.py:172-190,229-283,322-331).
// K/V of all positions are appended first, then every position attends over the cache.
static void pf_gemm(Launch& L, const bf16_t* X, long ldx, int S, const void* W, const float* bias, int N, int K,
                    int act, const float* resid, float* out_f32, bf16_t* out_bf, long ldo, int round_out, const PfX& fx) {
    TapGemmP p{};
    p.gain = (const bf16_t*)fx.gain; p.ss_in = fx.ss_in; p.ss_out = fx.ss_out; p.ss_nblk = fx.nblk;
    p.ss_ld = std::max(L.ctx->c.dim, L.ctx->c.fast_dim) / 16 + 1; p.eps = L.ctx->c.norm_eps;
    p.X = X; p.ldx = ldx; p.T_in = S; p.W = (const bf16_t*)W; p.ntap = 1; p.offs[0] = 0; p.M = S; p.N = N; p.K = K;
    p.bias = bias; p.n_mod = N; p.act = act; p.resid_f32 = resid; p.ldr = ldo; p.out_f32 = out_f32; p.out_bf = out_bf;
    p.ldo = ldo; p.round_lin = 1; p.round_f32_out = round_out;
    const int mode = L.ctx->prefill_gemm_mode;  // FT_PREFILL_GEMM: 0 = first tile kernel only, 1 = no skinny kernel
    const bool fused = fx.gain || fx.ss_out;
    if ((mode >= 2 || fused) && S <= 128 && K % 128 == 0 && N % 2 == 0) {
        // short prompts are weight-bandwidth bound: 16 weight rows per block, K split over the waves
        if (S <= 16) skinny_gemm_launch<1>(p, 1, L.s);
        else if (S <= 32) skinny_gemm_launch<2>(p, 1, L.s);
        else skinny_gemm_launch<4>(p, (S + 63) / 64, L.s);
    } else if (fused) {
        L.err = hipErrorInvalidValue;   // the fused norm exists on the skinny kernel only (callers check the shapes)
    } else if (mode >= 1 && K % 64 == 0 && N % 128 == 0) {
        // long prompts (reference audio): MFMA tiles with four K-steps of both operands in flight (lingemm_kernel) - 128 x 128
        // on 8 waves for the wide products from 512 rows, 64 x 64 on 4 waves otherwise (the N = 1024 products would leave
        // 200 CUs idle on the large tile: 56 workgroups at 780 rows).  Measured at 780 rows (tools/prefill_probe.py): 7.05 ms
        // with the codec's one-step-ahead tile kernel, 4.89 ms with these (3.50 against 5.61 ms at 256 rows).
        constexpr int tile8_s = 512;
        if (K % 256 == 0) {
            // both tiles are held to <= 128 registers so that workgroups share a CU (128 x 128: two K-steps in flight, two
            // workgroups per CU - 4.92 against 5.25 ms per 780-position prefill with four steps and one workgroup; 64 x 64: four
            // steps, four workgroups - another 0.07 ms)
            if (S >= tile8_s && N > 1024) lingemm_launch<128, 128, 2, 4, 2, 4>(p, S, N, L.s);
            else lingemm_launch<64, 64, 2, 2, 4, 4>(p, S, N, L.s);
        } else if (S >= tile8_s) {
            constexpr size_t lds8 = std::max((size_t)((128 + 56) + 2 * 128) * (64 + 8) * 2, (size_t)(128 / 2) * (128 + 4) * 4);
            static DevOnce once8;
            once8.run([] { hipFuncSetAttribute((const void*)tapgemm64_kernel<128, 128, 64, 2, 4>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8); });
            tapgemm64_kernel<128, 128, 64, 2, 4><<<dim3((S + 127) / 128, N / 128, 1), 512, lds8, L.s>>>(p);
        } else {
            const size_t lds = std::max((size_t)((64 + 56) + 2 * 64) * (64 + 8) * 2, (size_t)(64 / 2) * (64 + 4) * 4);
            tapgemm64_kernel<64, 64, 64><<<dim3((S + 63) / 64, N / 64, 1), 256, lds, L.s>>>(p);
        }
    } else if (N >= 128) tapgemm_kernel<128, 128, 2, 2><<<dim3((S + 127) / 128, (N + 127) / 128, 1), 256, 0, L.s>>>(p);
    else tapgemm_kernel<128, 64, 4, 1><<<dim3((S + 127) / 128, (N + 63) / 64, 1), 256, 0, L.s>>>(p);
    L.chk();
}

// rg: the rows are the prompts of rg->n slots back to back (ctx->pf_seqs / pf_rows describe them; Lp = all rows, slot and
// pos0 unused): the Linear products and norms do not care, the K/V append and the attention go by sequence.
struct RaggedPf { int n, max_lp; };
static __global__ __launch_bounds__(256) void gather_last_rows_kernel(const float* pf_x, int D, const int4* seqs, float* x) {
    const int4 sq = seqs[blockIdx.x];
    for (int d = threadIdx.x; d < D; d += 256) x[(size_t)sq.w * D + d] = pf_x[(size_t)(sq.x + sq.y - 1) * D + d];
}
static void prefill_gemm(Launch& L, int slot, int Lp, int pos0, bool with_tail = true, const RaggedPf* rg = nullptr) {
    ft_ctx* ctx = L.ctx;
    const ft_ar_config& c = ctx->c;
    const int D = c.dim, HD = c.n_head * c.head_dim, F = c.intermediate_size;
    const int qkvN = (c.n_head + 2 * c.n_local_heads) * c.head_dim;
    EmbedP e{};
    e.emb = ctx->emb; e.cb_emb = ctx->cb_emb; e.toks = ctx->d_prompt; e.tok_row_stride = Lp; e.tok_m_stride = 1;
    e.col = 0; e.x = ctx->pf_x; e.ldx = D; e.D = D; e.ncb = c.num_codebooks; e.cbsize = c.codebook_size;
    e.vocab = c.vocab_size; e.sem_begin = c.semantic_begin_id; e.sem_end = c.semantic_end_id;
    e.scale = c.scale_codebook_embeddings; e.inv_div = (float)sqrt((double)(c.num_codebooks + 1));
    embed_kernel<bf16_t, true><<<dim3((D + 255) / 256, Lp), 256, 0, L.s>>>(e);
    L.chk();
    const int G = c.n_head / c.n_local_heads;
    const int nslot = 2048 / c.head_dim;
    const size_t lds = ((size_t)G * c.head_dim + 2 * c.head_dim + (size_t)nslot * G * 2 + (size_t)nslot * G * c.head_dim) * sizeof(float);
    for (int li = 0; li < c.n_layer; ++li) {
        const FtLayer& l = ctx->layers[li];
        rmsnorm_llama_rows_kernel<bf16_t, true><<<Lp, 256, 0, L.s>>>(ctx->pf_x, l.attn_norm, c.norm_eps, D, ctx->pf_xn);
        pf_gemm(L, ctx->pf_xn, D, Lp, l.wqkv, l.bqkv_f32, qkvN, D, ACT_NONE, nullptr, ctx->pf_qkv, nullptr, qkvN, 0);
        AttnP a{};
        a.qkv = ctx->pf_qkv; a.ldq = qkvN; a.qn = l.qn; a.kn = l.kn; a.rope = ctx->rope;
        a.kc = (char*)l.kc + (size_t)slot * ctx->cache_m_stride * ctx->esz;
        a.vc = (char*)l.vc + (size_t)slot * ctx->cache_m_stride * ctx->esz;
        a.cache_m_stride = 0; a.pos = nullptr; a.pos_off = pos0; a.row_is_pos = 1;
        if (rg) { a.kc = l.kc; a.vc = l.vc; a.cache_m_stride = ctx->cache_m_stride; a.row_sp = ctx->pf_rows; }
        a.H = c.n_head; a.Hkv = c.n_local_heads; a.hd = c.head_dim; a.n_slots = ctx->n_slots; a.nsplit = 1;
        a.eps = c.norm_eps; a.scale = 1.0f / sqrtf((float)c.head_dim); a.y = ctx->pf_y; a.ldy = HD; a.y_bf = ctx->pf_ybf;
        Launch LA = L;
        LA.M = Lp;
        // pass 0 appends K/V of every position (and leaves the finished queries); pass 1 attends: on the matrix cores
        // for 64- and 128-wide heads, otherwise position by position with the decode kernel
        // (a short tail behind a long restored prefix: with four key groups per block the tiled kernel wins from 16 new
        // positions - 47.7 against 49.5 ms to the first 10 frames of 8 cloned-voice utterances with 49-token tails; round 2's
        // one-group kernel lost below 64)
        const bool flash = rg || (!getenv("FT_PREFILL_ATTN_V0") && (c.head_dim == 64 || c.head_dim == 128) && Lp >= 16);
        a.q_out = flash ? ctx->pf_qbf : nullptr;
        for (int pass = 0; pass < 2; ++pass) {
            a.kv_only = pass == 0; a.no_append = pass == 1;
            if (pass == 0 && flash) {       // every position's q k v in one launch of Lp blocks
                if (c.head_dim == 128) prefill_rope_append_kernel<128><<<Lp, 256, 0, L.s>>>(a);
                else prefill_rope_append_kernel<64><<<Lp, 256, 0, L.s>>>(a);
                L.chk();
                continue;
            }
            if (pass == 1 && flash) {
                FlashP fp{ctx->pf_qbf, (const bf16_t*)a.kc, (const bf16_t*)a.vc, ctx->pf_ybf, Lp, c.n_head, c.n_local_heads,
                          c.head_dim, ctx->n_slots, pos0, a.scale};
                int nz = 1, lp_grid = Lp;
                if (rg) { fp.seqs = ctx->pf_seqs; fp.cache_m_stride = ctx->cache_m_stride; nz = rg->n; lp_grid = rg->max_lp; }
                // key groups per block: four (32 query rows per block: more blocks) up to 320 positions, two beyond (measured:
                // 3.39 against 3.46 ms per 160-position prefill, 5.24 against 4.91 at 780 - the longer walks re-read K/V twice as often)
                const bool ng4 = lp_grid <= 320;
                if (c.head_dim == 128) { if (ng4) flash_launch<128, 4>(fp, lp_grid, L.s, nz); else flash_launch<128, 2>(fp, lp_grid, L.s, nz); }
                else { if (ng4) flash_launch<64, 4>(fp, lp_grid, L.s, nz); else flash_launch<64, 2>(fp, lp_grid, L.s, nz); }
                L.chk();
                continue;
            }
            const dim3 grid(c.n_local_heads, 1, Lp);
            switch (G) {
                case 1: attn_decode_kernel<bf16_t, 1, true><<<grid, 256, lds, L.s>>>(a); break;
                case 2: attn_decode_kernel<bf16_t, 2, true><<<grid, 256, lds, L.s>>>(a); break;
                case 4: attn_decode_kernel<bf16_t, 4, true><<<grid, 256, lds, L.s>>>(a); break;
                default:
                    hipFuncSetAttribute((const void*)attn_decode_kernel<bf16_t, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    attn_decode_kernel<bf16_t, 8, true><<<grid, 256, lds, L.s>>>(a);
            }
            L.chk();
        }
        pf_gemm(L, ctx->pf_ybf, HD, Lp, l.wo, l.bo_f32, D, HD, ACT_NONE, ctx->pf_x, ctx->pf_x, nullptr, D, 1);
        rmsnorm_llama_rows_kernel<bf16_t, true><<<Lp, 256, 0, L.s>>>(ctx->pf_x, l.ffn_norm, c.norm_eps, D, ctx->pf_xn);
        pf_gemm(L, ctx->pf_xn, D, Lp, l.w13, nullptr, 2 * F, D, ACT_SWIGLU, nullptr, nullptr, ctx->pf_g, F, 0);
        pf_gemm(L, ctx->pf_g, F, Lp, l.w2, nullptr, D, F, ACT_NONE, ctx->pf_x, ctx->pf_x, nullptr, D, 1);
    }
    // the last position feeds the head and the fast stack through the decode kernels
    if (rg) { gather_last_rows_kernel<<<rg->n, 256, 0, L.s>>>(ctx->pf_x, D, ctx->pf_seqs, ctx->x); L.chk(); return; }
    hipMemcpyAsync(ctx->x + (size_t)slot * D, ctx->pf_x + (size_t)(Lp - 1) * D, D * sizeof(float), hipMemcpyDeviceToDevice, L.s);
    if (!with_tail) return;   // ft_ar_prefill_slow: the first frames of several slots are drawn together later
    enqueue_fproj<bf16_t, true>(L);
    enqueue_frame_tail<bf16_t, true>(L);
}

extern "C" ft_status ft_ar_prefill_at(ft_ctx* ctx, int32_t slot, const int32_t* prompt, int32_t Lp, int32_t pos0,
                                      const ft_sampling* sp, int32_t* out_frame) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!prompt || !sp || !out_frame) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill: null argument");
    if (slot < 0 || slot >= c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill: bad slot");
    if (Lp < 1) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill: empty prompt");
    if (pos0 < 0) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_at: pos0 < 0");
    if (pos0 + Lp >= c.max_seq_len) {  // inference.py:296-299
        char buf[128];
        snprintf(buf, sizeof buf, "Input sequence length %d exceeds max_seq_len %d", pos0 + Lp, c.max_seq_len);
        return ft_fail(ctx, FT_ERR_TOO_LONG, buf);
    }
    const int R = c.num_codebooks + 1;
    FT_TRY(ft_ar_reset(ctx, slot));
    FT_TRY(upload_ctl(ctx, slot, 1, sp));
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_prompt, prompt, (size_t)R * Lp * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    pick_nsplit(ctx, pos0 + Lp);       // (position-by-position prompt passes: the split count of this prompt's length)
    Launch L{ctx, ctx->stream, slot, 1, 0};
    const bool on_engine = eng_in_use(ctx);
    if (!ctx->prefill_v0) {
        prefill_gemm(L, slot, Lp, pos0);
    } else {
        // f32 precision (and shapes the MFMA tiles do not cover): the prompt is fed through the S=1 decode
        // kernels position by position (same causal arithmetic); only the last position runs the head
        for (int t = 0; t < Lp - 1; ++t) {
            L.pos_off = pos0 + t;
            enqueue_slow_only(L, ctx->d_prompt, Lp, 0, t);
        }
        L.pos_off = pos0 + Lp - 1;
        enqueue_frame(L, ctx->d_prompt, Lp, 0, Lp - 1);
    }
    if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("prefill launch: ") + hipGetErrorString(L.err));
    if (on_engine) {
        // the first frame's codebook loop (and, position by position, its slow pass) ran on the frame engine: a hand-off
        // time-out there must not pass for a first frame - redo that frame on the launch path (the prompt's K/V and the
        // last position's hidden state are untouched by it)
        FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        bool aborted = false;
        FT_TRY(eng_recover(ctx, &aborted));
        if (aborted) {
            FT_TRY(ft_ar_reset(ctx, slot));
            ctx->eng_suspended = true;
            if (!ctx->prefill_v0) { enqueue_fproj<bf16_t, true>(L); enqueue_frame_tail<bf16_t, true>(L); }
            else enqueue_frame(L, ctx->d_prompt, Lp, 0, Lp - 1);
            ctx->eng_suspended = false;
            if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("prefill launch: ") + hipGetErrorString(L.err));
        }
    }
    // finalize() advanced pos 0 -> 1; the next input position is pos0 + Lp
    ctx->h_pin[0] = pos0 + Lp;
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_pos + slot, ctx->h_pin, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(out_frame, ctx->d_tok + (size_t)slot * R, R * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FT_OK;
}

// Prompt pass without the first frame: K/V of the prompt and the last position's hidden state stay on the device.
extern "C" ft_status ft_ar_prefill_slow(ft_ctx* ctx, int32_t slot, const int32_t* prompt, int32_t Lp, int32_t pos0) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!prompt) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow: null argument");
    if (slot < 0 || slot >= c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow: bad slot");
    if (Lp < 1 || pos0 < 0) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow: empty prompt");
    if (pos0 + Lp >= c.max_seq_len) {
        char buf[128];
        snprintf(buf, sizeof buf, "Input sequence length %d exceeds max_seq_len %d", pos0 + Lp, c.max_seq_len);
        return ft_fail(ctx, FT_ERR_TOO_LONG, buf);
    }
    const int R = c.num_codebooks + 1;
    FT_TRY(ft_ar_reset(ctx, slot));
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_prompt, prompt, (size_t)R * Lp * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    pick_nsplit(ctx, pos0 + Lp);
    Launch L{ctx, ctx->stream, slot, 1, 0};
    if (!ctx->prefill_v0) {
        prefill_gemm(L, slot, Lp, pos0, false);
    } else {
        for (int t = 0; t < Lp; ++t) {
            L.pos_off = pos0 + t;
            enqueue_slow_only(L, ctx->d_prompt, Lp, 0, t);
        }
    }
    if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("prefill launch: ") + hipGetErrorString(L.err));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // d_prompt is reused by the next call
    return FT_OK;
}

// Prompt passes of several slots (a batch scheduler's initial fill and its refills; per prompt: inference.py:353-362).
// Where lock-step batches run on the MFMA launches (ctx->wide_ok) and n reaches their width (wide_min), the prompts go
// through the slow stack TOGETHER: their positions back to back as the rows of one pass - every Linear product streams its
// weights once for all of them (a 48-position prompt alone is bound by the weight stream: 32 of them cost 32 streams) -
// while the K/V append goes by (slot, position) of each row and the attention by sequence (grid.z).  Like the lock-step
// frames of that width the products then sum in another order than a prompt pass on its own (another tile kernel from
// 129 rows): judged against the oracle with the bf16 margin.  Fewer prompts, other precisions and shapes: one by one,
// bit-equal to ft_ar_prefill_slow.  More rows than the workspace holds (max_seq_len): several passes.
extern "C" ft_status ft_ar_prefill_slow_many(ft_ctx* ctx, int32_t n, const int32_t* slots, const int32_t* prompts,
                                             const int32_t* Lps, const int32_t* pos0s) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!slots || !prompts || !Lps || !pos0s) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow_many: null argument");
    if (n < 1 || n > c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow_many: bad count");
    const int R = c.num_codebooks + 1;
    std::vector<char> seen(c.max_batch, 0);
    std::vector<size_t> off(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (slots[i] < 0 || slots[i] >= c.max_batch || seen[slots[i]]) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow_many: bad or repeated slot");
        seen[slots[i]] = 1;
        if (Lps[i] < 1 || pos0s[i] < 0) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_prefill_slow_many: empty prompt");
        if (pos0s[i] + Lps[i] >= c.max_seq_len) {
            char buf[128];
            snprintf(buf, sizeof buf, "Input sequence length %d exceeds max_seq_len %d", pos0s[i] + Lps[i], c.max_seq_len);
            return ft_fail(ctx, FT_ERR_TOO_LONG, buf);
        }
        off[i + 1] = off[i] + (size_t)R * Lps[i];
    }
    const bool ragged = !ctx->prefill_v0 && ctx->wide_ok && n >= ctx->wide_min && (c.head_dim == 64 || c.head_dim == 128) &&
                        !getenv("FT_PREFILL_ATTN_V0") && !getenv("FT_NO_RAGGED_PREFILL");
    if (!ragged) {
        for (int i = 0; i < n; ++i) FT_TRY(ft_ar_prefill_slow(ctx, slots[i], prompts + off[i], Lps[i], pos0s[i]));
        return FT_OK;
    }
    std::vector<int> tok;
    std::vector<int4> seqs;
    std::vector<int2> rows;
    for (int i0 = 0; i0 < n;) {
        int i1 = i0, S = 0, max_lp = 0, end = 0;
        while (i1 < n && S + Lps[i1] <= c.max_seq_len) { S += Lps[i1]; max_lp = std::max(max_lp, Lps[i1]); end = std::max(end, pos0s[i1] + Lps[i1]); ++i1; }
        tok.assign((size_t)R * S, 0); seqs.clear(); rows.clear();
        int row0 = 0;
        for (int i = i0; i < i1; ++i) {
            FT_TRY(ft_ar_reset(ctx, slots[i]));
            for (int r = 0; r < R; ++r)
                memcpy(&tok[(size_t)r * S + row0], prompts + off[i] + (size_t)r * Lps[i], (size_t)Lps[i] * sizeof(int));
            seqs.push_back(int4{row0, Lps[i], pos0s[i], slots[i]});
            for (int t = 0; t < Lps[i]; ++t) rows.push_back(int2{slots[i], pos0s[i] + t});
            row0 += Lps[i];
        }
        FT_HIP(ctx, hipMemcpyAsync(ctx->d_prompt, tok.data(), tok.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        FT_HIP(ctx, hipMemcpyAsync(ctx->pf_seqs, seqs.data(), seqs.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
        FT_HIP(ctx, hipMemcpyAsync(ctx->pf_rows, rows.data(), rows.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
        pick_nsplit(ctx, end);
        Launch L{ctx, ctx->stream, 0, 1, 0};
        const RaggedPf rg{i1 - i0, max_lp};
        prefill_gemm(L, 0, S, 0, false, &rg);
        if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("prefill launch: ") + hipGetErrorString(L.err));
        FT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors and d_prompt are reused by the next pass
        i0 = i1;
    }
    return FT_OK;
}

// First frames of slots [slot0, slot0 + n) whose prompts went through ft_ar_prefill_slow: one lock-step pass of the
// vocabulary head, the semantic draw and the fast codebooks over all of them.
extern "C" ft_status ft_ar_first_frames(ft_ctx* ctx, int32_t slot0, int32_t n, const ft_sampling* sp, const int32_t* next_pos,
                                        int32_t* out_frames) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!sp || !next_pos || !out_frames) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_first_frames: null argument");
    if (slot0 < 0 || n < 1 || slot0 + n > c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_first_frames: bad slot range");
    const int R = c.num_codebooks + 1;
    FT_TRY(upload_ctl(ctx, slot0, n, sp));
    Launch L{ctx, ctx->stream, slot0, n, 0};
    L.tail_only = true;
    const bool on_engine = n == 1 && eng_in_use(ctx);
    auto tail = [&]() {
        if (c.dtype == FT_BF16) { enqueue_fproj<bf16_t, RND_BF16>(L); enqueue_frame_tail<bf16_t, RND_BF16>(L); }
        else if (c.dtype == FT_F16) { enqueue_fproj<f16_t, RND_F16>(L); enqueue_frame_tail<f16_t, RND_F16>(L); }
        else { enqueue_fproj<float, RND_NONE>(L); enqueue_frame_tail<float, RND_NONE>(L); }
    };
    tail();
    if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("first-frame launch: ") + hipGetErrorString(L.err));
    if (on_engine) {      // as in ft_ar_prefill_at: a timed-out codebook loop is redone on the launch path
        FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        bool aborted = false;
        FT_TRY(eng_recover(ctx, &aborted));
        if (aborted) {
            FT_TRY(ft_ar_reset(ctx, slot0));
            ctx->eng_suspended = true;
            tail();
            ctx->eng_suspended = false;
            if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("first-frame launch: ") + hipGetErrorString(L.err));
        }
    }
    // finalize() advanced every position 0 -> 1; the next input position of a slot is the end of its prompt
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_pos + slot0, next_pos, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(out_frames, ctx->d_tok + (size_t)slot0 * R, (size_t)n * R * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FT_OK;
}

extern "C" ft_status ft_ar_prefill(ft_ctx* ctx, int32_t slot, const int32_t* prompt, int32_t Lp,
                                   const ft_sampling* sp, int32_t* out_frame) {
    return ft_ar_prefill_at(ctx, slot, prompt, Lp, 0, sp, out_frame);
}

// ---- reference-prefix K/V snapshots: [layer][K|V][Hkv][n_pos][hd] packed, in the cache's element type
struct ft_kv_snapshot {
    char* data = nullptr;
    int n_pos = 0;
    size_t bytes = 0;
};

static ft_status kv_copy(ft_ctx* ctx, char* snap, int n_pos, int slot, bool save) {
    const ft_ar_config& c = ctx->c;
    const size_t row = (size_t)n_pos * c.head_dim * ctx->esz;            // one head's positions [0, n_pos)
    const size_t pitch = (size_t)ctx->n_slots * c.head_dim * ctx->esz;   // head stride inside the cache
    for (int li = 0; li < c.n_layer; ++li) {
        const FtLayer& l = ctx->layers[li];
        for (int kv = 0; kv < 2; ++kv) {
            char* cache = (char*)(kv ? l.vc : l.kc) + (size_t)slot * ctx->cache_m_stride * ctx->esz;
            char* packed = snap + ((size_t)li * 2 + kv) * c.n_local_heads * row;
            if (save) FT_HIP(ctx, hipMemcpy2DAsync(packed, row, cache, pitch, row, c.n_local_heads, hipMemcpyDeviceToDevice, ctx->stream));
            else FT_HIP(ctx, hipMemcpy2DAsync(cache, pitch, packed, row, row, c.n_local_heads, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    return FT_OK;
}

extern "C" ft_status ft_ar_kv_save(ft_ctx* ctx, int32_t slot, int32_t n_pos, ft_kv_snapshot** out) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!out) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_kv_save: null argument");
    if (slot < 0 || slot >= c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_kv_save: bad slot");
    if (n_pos < 1 || n_pos > ctx->n_slots) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_kv_save: bad n_pos");
    ft_kv_snapshot* s = new ft_kv_snapshot();
    s->n_pos = n_pos;
    s->bytes = (size_t)c.n_layer * 2 * c.n_local_heads * n_pos * c.head_dim * ctx->esz;
    if (hipMalloc((void**)&s->data, s->bytes) != hipSuccess) {
        delete s;
        return ft_fail(ctx, FT_ERR_HIP, "ft_ar_kv_save: out of device memory");
    }
    ft_status st = kv_copy(ctx, s->data, n_pos, slot, true);
    if (st == FT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = ft_fail(ctx, FT_ERR_HIP, "ft_ar_kv_save: copy failed");
    if (st != FT_OK) { hipFree(s->data); delete s; return st; }
    *out = s;
    return FT_OK;
}

extern "C" ft_status ft_ar_kv_restore(ft_ctx* ctx, const ft_kv_snapshot* snap, int32_t slot) {
    FT_TRY(ar_ready(ctx));
    if (!snap || !snap->data) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_kv_restore: null snapshot");
    if (slot < 0 || slot >= ctx->c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_kv_restore: bad slot");
    if (snap->n_pos > ctx->n_slots) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_kv_restore: snapshot longer than the cache");
    return kv_copy(ctx, snap->data, snap->n_pos, slot, false);  // stream-ordered before the next prefill
}

extern "C" int32_t ft_ar_kv_positions(const ft_kv_snapshot* snap) { return snap ? snap->n_pos : 0; }

extern "C" void ft_ar_kv_free(ft_ctx* ctx, ft_kv_snapshot* snap) {
    if (!snap) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    if (snap->data) hipFree(snap->data);
    delete snap;
}

// the captured grids depend on the batch width, on the KV split count and on whether the frame engine serves the frame
static int graph_key(const ft_ctx* ctx, int M, int frames) {
    const bool eng = M == 1 && (ctx->eng_on || ctx->eng_fast_on) && !ctx->eng_suspended;   // the engine serves one-slot frames only
    return (M * 64 + ctx->nsplit) + (frames > 1 ? frames * (1 << 20) : 0) + (eng ? (1 << 28) : 0);
}

static ft_status get_graph(ft_ctx* ctx, int M, hipGraphExec_t* out, int frames = 1) {
    // the captured grids depend on the batch width and on the KV split count; `frames` decode frames per graph
    const int key = graph_key(ctx, M, frames);
    auto it = ctx->graphs.find(key);
    if (it != ctx->graphs.end()) { *out = it->second; return FT_OK; }
    const int R = ctx->c.num_codebooks + 1;
    hipGraph_t graph = nullptr;
    FT_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    Launch L{ctx, ctx->stream, 0, M, 0};
    for (int f = 0; f < frames; ++f) enqueue_frame(L, ctx->d_tok, 1, R, 0);
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (L.err != hipSuccess) e = L.err;
    if (e != hipSuccess) {
        if (graph) hipGraphDestroy(graph);
        return ft_fail(ctx, FT_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(e));
    }
    hipGraphExec_t exec = nullptr;
    { size_t nn = 0; if (frames == 1 && hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess) ctx->graph_nodes[key] = (int)nn; }
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("graph instantiate: ") + hipGetErrorString(e));
    ctx->graphs[key] = exec;
    *out = exec;
    return FT_OK;
}

extern "C" ft_status ft_ar_decode(ft_ctx* ctx, int32_t nslots, int32_t n_frames, const ft_sampling* sp,
                                  int32_t poll, int32_t* out_frames, int32_t* out_n) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!sp || !out_frames || !out_n) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_decode: null argument");
    if (nslots < 1 || nslots > c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_decode: bad nslots");
    if (n_frames < 0) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_decode: n_frames < 0");
    if (poll < 1) poll = 1;
    const int R = c.num_codebooks + 1;
    FT_TRY(upload_ctl(ctx, 0, nslots, sp));
    // state at entry
    int* h = ctx->h_pin;
    FT_HIP(ctx, hipMemcpyAsync(h, ctx->d_nf, nslots * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(h + nslots, ctx->d_pos, nslots * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(h + 2 * nslots, ctx->d_done, nslots * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> nf0(h, h + nslots);
    int budget = n_frames;
    for (int m = 0; m < nslots; ++m) {
        if (nf0[m] < 1) return ft_fail(ctx, FT_ERR_STATE, "ft_ar_decode: slot has not been prefilled");
        if (h[2 * nslots + m]) continue;  // finished or parked: frozen on the device, limits nothing
        budget = std::min(budget, ctx->cap - nf0[m]);
        budget = std::min(budget, ctx->n_slots - h[nslots + m]);  // cache positions left
    }
    if (budget < 0) budget = 0;
    {   // KV splits follow the longest context this call will reach
        int pos_end = 0;
        for (int m = 0; m < nslots; ++m)
            if (!h[2 * nslots + m]) pos_end = std::max(pos_end, h[nslots + m] + budget);
        pick_nsplit(ctx, pos_end);
    }
    const bool eager = getenv("FT_NO_GRAPH") != nullptr;
    // several frames per graph launch where a burst allows: the gap between two graph launches is ~7 us (measured: 687 ->
    // 692 tok/s at 16 frames per graph; the frames are the same launches in the same order)
    constexpr int gk = 16;
    const int first_burst = std::min(poll, budget);
    // `burst` frames on the stream, then the done flags: graph replays (16 / 4 / 1 frames per launch) or eager launches
    auto run_burst = [&](int burst) -> ft_status {
        hipGraphExec_t exec = nullptr, exec_k = nullptr, exec_4 = nullptr;
        if (!eager) FT_TRY(get_graph(ctx, nslots, &exec));
        if (!eager && gk > 1 && first_burst >= gk && burst >= gk) FT_TRY(get_graph(ctx, nslots, &exec_k, gk));
        if (!eager && gk > 4 && first_burst >= 4 && burst >= 4) FT_TRY(get_graph(ctx, nslots, &exec_4, 4));
        int i0 = 0;
        if (exec_k) for (; i0 + gk <= burst; i0 += gk) FT_HIP(ctx, hipGraphLaunch(exec_k, ctx->stream));
        if (exec_4) for (; i0 + 4 <= burst; i0 += 4) FT_HIP(ctx, hipGraphLaunch(exec_4, ctx->stream));
        for (int i = i0; i < burst; ++i) {
            if (eager) {
                Launch L{ctx, ctx->stream, 0, nslots, 0};
                enqueue_frame(L, ctx->d_tok, 1, R, 0);
                if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("frame launch: ") + hipGetErrorString(L.err));
            } else {
                FT_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
            }
        }
        FT_HIP(ctx, hipMemcpyAsync(h, ctx->d_done, nslots * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FT_OK;
    };
    const int pos0 = h[nslots], done0 = h[2 * nslots];      // slot 0 at entry (the engine serves one-slot calls only)
    std::vector<char> was_done(nslots);
    for (int m = 0; m < nslots; ++m) was_done[m] = (char)(h[2 * nslots + m] != 0);   // frozen slots produce nothing new
    int done_frames = 0;
    while (done_frames < budget) {
        const int burst = std::min(poll, budget - done_frames);
        const bool on_engine = nslots == 1 && eng_in_use(ctx);
        FT_TRY(run_burst(burst));
        if (on_engine) {
            bool aborted = false;
            FT_TRY(eng_recover(ctx, &aborted));
            if (aborted) {
                // slot 0 back to where this burst began (its frames [nf0 + done_frames, ..) and cache rows are rewritten by
                // the redo; the draws are counter-based, so the redo draws what the engine would have drawn), then the same
                // frames on the launch path
                int* hp = ctx->h_pin + 3 * nslots;
                hp[0] = done0 ? pos0 : pos0 + done_frames; hp[1] = done0 ? nf0[0] : nf0[0] + done_frames; hp[2] = done0;
                FT_HIP(ctx, hipMemcpyAsync(ctx->d_pos, hp, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
                FT_HIP(ctx, hipMemcpyAsync(ctx->d_nf, hp + 1, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
                FT_HIP(ctx, hipMemcpyAsync(ctx->d_done, hp + 2, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
                FT_HIP(ctx, hipMemcpy2DAsync(ctx->d_tok, sizeof(int), ctx->d_seq + (hp[1] - 1), (size_t)ctx->cap * sizeof(int),
                                             sizeof(int), R, hipMemcpyDeviceToDevice, ctx->stream));
                // the frame store beyond the rewound frame count goes back to zero: early in an utterance the repetition
                // penalty's window reaches past the current frame (inference.py:187-191 reads a 16-column window of a
                // zero-initialised store), and the aborted burst left its frames there
                if (!done0 && hp[1] < ctx->cap)
                    FT_HIP(ctx, hipMemset2DAsync(ctx->d_seq + hp[1], (size_t)ctx->cap * sizeof(int), 0,
                                                 (size_t)std::min(burst, ctx->cap - hp[1]) * sizeof(int), R, ctx->stream));
                ctx->eng_suspended = true;
                const ft_status st = run_burst(burst);
                ctx->eng_suspended = false;
                FT_TRY(st);
            }
        }
        done_frames += burst;
        bool all = true;
        for (int m = 0; m < nslots; ++m) all = all && h[m];
        if (all) break;
    }
    // collect: frames [nf0, nf0+done_frames) of each slot, cut after the first <|im_end|>.  Only the columns
    // [nf0-1, nf0+done_frames) of the frame store travel (one strided copy per slot, all issued before the wait).
    const int W = done_frames + 1;
    std::vector<int> seq((size_t)nslots * R * W);
    for (int m = 0; m < nslots; ++m) {
        if (was_done[m] || done_frames == 0) continue;
        const int c0 = std::min(nf0[m] - 1, ctx->cap - W);        // stay inside the store near its end
        FT_HIP(ctx, hipMemcpy2DAsync(seq.data() + (size_t)m * R * W, (size_t)W * sizeof(int),
                                     ctx->d_seq + (size_t)m * R * ctx->cap + std::max(c0, 0), (size_t)ctx->cap * sizeof(int),
                                     (size_t)std::min(W, ctx->cap) * sizeof(int), R, hipMemcpyDeviceToHost, ctx->stream));
    }
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int m = 0; m < nslots; ++m) {
        int n = 0;
        if (!was_done[m] && done_frames > 0) {
            const int c0 = std::max(std::min(nf0[m] - 1, ctx->cap - W), 0);
            const int* sm = seq.data() + (size_t)m * R * W;
            for (int f = 0; f < done_frames; ++f) {
                const int col = nf0[m] + f - c0;
                if (col >= std::min(W, ctx->cap)) break;
                for (int r = 0; r < R; ++r) out_frames[((size_t)m * n_frames + f) * R + r] = sm[(size_t)r * W + col];
                n = f + 1;
                if (sm[col] == c.im_end_id) break;
            }
        }
        out_n[m] = n;
    }
    return FT_OK;
}

// A hand-off time-out of the frame engine (ENG_CTL_ABORT: some workgroup gave up after ENG_TIMEOUT_TICKS, then all did)
// is survivable.  Called after the stream has drained: *aborted tells the caller that the frames since the last check
// are invalid; the control words and every hand-off buffer are cleared (no granule of the aborted launches can carry a
// valid tag afterwards: tag 0 is never valid and the epoch moves on), and after ENG_MAX_STRIKES events the context stops
// using the engine.  The caller then redoes those frames on the launch path (ctx->eng_suspended).
static ft_status eng_recover(ft_ctx* ctx, bool* aborted) {
    *aborted = false;
    if (!ctx->eng_ctl) return FT_OK;
    unsigned w[ENG_CTL_WORDS];
    FT_HIP(ctx, hipMemcpyAsync(w, ctx->eng_ctl, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!w[ENG_CTL_ABORT]) return FT_OK;
    *aborted = true;
    ctx->eng_strikes += 1;
    ctx->eng_last_where = (int)w[ENG_CTL_WHERE];
    const unsigned epoch = w[ENG_CTL_EPOCH] + 4u * ENG_EPOCH_STEP;
    FT_HIP(ctx, hipMemsetAsync(ctx->eng_ctl, 0, ENG_CTL_WORDS * 4, ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(ctx->eng_ctl + ENG_CTL_EPOCH, &epoch, sizeof epoch, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->eng_gx) FT_HIP(ctx, hipMemsetAsync(ctx->eng_gx, 0, ctx->eng_pool_bytes, ctx->stream));
    if (ctx->eng_gpart) FT_HIP(ctx, hipMemsetAsync(ctx->eng_gpart, 0, ctx->eng_gpart_bytes, ctx->stream));
    if (ctx->eng_fast_g) FT_HIP(ctx, hipMemsetAsync(ctx->eng_fast_g, 0, ctx->eng_fast_bytes, ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // (epoch is a stack word)
    char buf[256];
    if (ctx->eng_strikes >= ENG_MAX_STRIKES) {
        // the engine is given up for good: its graphs go, its buffers and the device's engine seat are released (another
        // context of the process may take it), the reason stays readable through ft_ar_frame_path
        for (auto it = ctx->graphs.begin(); it != ctx->graphs.end();) {
            if (it->first & (1 << 28)) { hipGraphExecDestroy(it->second); it = ctx->graphs.erase(it); } else ++it;
        }
        eng_release(ctx);
        snprintf(buf, sizeof buf, "launch path: the frame engine was turned off after %d hand-off time-outs (last in phase %u)",
                 ctx->eng_strikes, w[ENG_CTL_WHERE]);
        ctx->eng_why = buf;
    }
    fprintf(stderr, "fish_tts_amd: frame engine: a hand-off timed out (phase %u, strike %d of %d): the frames are redone on the "
            "launch path%s\n", w[ENG_CTL_WHERE], ctx->eng_strikes, ENG_MAX_STRIKES,
            ctx->eng_strikes >= ENG_MAX_STRIKES ? "; the engine stays off for this context" : "");
    return FT_OK;
}
static bool eng_in_use(const ft_ctx* ctx) { return (ctx->eng_on || ctx->eng_fast_on) && !ctx->eng_suspended; }

extern "C" ft_status ft_ar_engine_state(ft_ctx* ctx, int32_t* flags, int32_t* aborted, int32_t* where) {
    FT_TRY(ar_ready(ctx));
    if (flags) *flags = (ctx->eng_on ? 1 : 0) | (ctx->eng_fast_on ? 2 : 0);
    if (aborted) *aborted = ctx->eng_strikes;
    if (where) *where = ctx->eng_last_where;
    return FT_OK;
}

extern "C" const char* ft_ar_frame_path(const ft_ctx* ctx) {
    if (!ctx) return "";
    // batch-1 frames, then what a lock-step batch of >= wide_min rows runs on
    ft_ctx* c = const_cast<ft_ctx*>(ctx);
    c->path_str = ctx->eng_why + (ctx->wide_ok && ctx->c.dtype == FT_BF16
                                      ? "; lock-step batches of >= 5 rows: MFMA launches with fused row operations (five per layer)"
                                      : "; lock-step batches: multi-row GEMV launches");
    return c->path_str.c_str();
}

// Test hook: workgroup `wg` of a coming slow-stack (which = 0) or codebook-loop (which = 1) engine launch plays dead - it
// publishes nothing, every workgroup that waits for its rows times out (one shot: that workgroup clears the word).  `skip`
// launches of that kind pass first, so the time-out can be placed in a later burst of a call or inside a multi-frame graph.
extern "C" ft_status ft_test_engine_fault(ft_ctx* ctx, int32_t which, int32_t wg, int32_t skip) {
    FT_TRY(ar_ready(ctx));
    if (!ctx->eng_ctl || !(which == 0 ? ctx->eng_on : ctx->eng_fast_on)) return ft_fail(ctx, FT_ERR_STATE, "ft_test_engine_fault: the frame engine is off");
    if (wg < 0 || wg >= ctx->eng_nb || skip < 0) return ft_fail(ctx, FT_ERR_ARG, "ft_test_engine_fault: bad workgroup or skip count");
    const unsigned v[2] = {(which == 0 ? 0u : ENG_FAULT_FAST) + 1u + (unsigned)wg, (unsigned)skip};
    static_assert(ENG_CTL_FAULT_SKIP == ENG_CTL_FAULT + 1, "the two words are written by one copy");
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FT_HIP(ctx, hipMemcpy(ctx->eng_ctl + ENG_CTL_FAULT, v, sizeof v, hipMemcpyHostToDevice));
    return FT_OK;
}

extern "C" ft_status ft_ar_park(ft_ctx* ctx, int32_t slot) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (slot < 0 || slot >= c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_park: bad slot");
    const int R = c.num_codebooks + 1;
    FT_TRY(ft_ar_reset(ctx, slot));
    // one stored frame = <|im_end|>, done = 1, a harmless input column (token 0, codes 0) at position 0
    int* h = ctx->h_pin;
    h[0] = c.im_end_id; h[1] = 1;
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_seq + (size_t)slot * R * ctx->cap, h, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_nf + slot, h + 1, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FT_HIP(ctx, hipMemcpyAsync(ctx->d_done + slot, h + 1, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FT_HIP(ctx, hipMemsetAsync(ctx->d_tok + (size_t)slot * R, 0, R * sizeof(int), ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FT_OK;
}

extern "C" ft_status ft_ar_set_noise(ft_ctx* ctx, const float* q, int64_t n_rows, int64_t row_len) {
    if (!ctx || !ctx->has_ar) return FT_ERR_ARG;
    FT_HIP(ctx, hipSetDevice(ctx->device));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->graphs) hipGraphExecDestroy(kv.second);  // captured launches hold the old pointer
    ctx->graphs.clear();
    if (ctx->noise) { hipFree(ctx->noise); ctx->noise = nullptr; }
    ctx->noise_rows = ctx->noise_row_len = 0;
    if (!q) return FT_OK;
    const int64_t need = (int64_t)ctx->c.vocab_size + (int64_t)(ctx->c.num_codebooks - 1) * ctx->fastV;
    if (row_len < need) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_set_noise: row_len too small");
    FT_HIP(ctx, hipMalloc((void**)&ctx->noise, (size_t)n_rows * row_len * sizeof(float)));
    FT_HIP(ctx, hipMemcpy(ctx->noise, q, (size_t)n_rows * row_len * sizeof(float), hipMemcpyDefault));
    ctx->noise_rows = n_rows;
    ctx->noise_row_len = row_len;
    return FT_OK;
}

extern "C" ft_status ft_ar_get_debug(ft_ctx* ctx, int32_t slot, float* logits, float* hidden) {
    FT_TRY(ar_ready(ctx));
    if (slot < 0 || slot >= ctx->c.max_batch) return ft_fail(ctx, FT_ERR_ARG, "bad slot");
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (logits)
        FT_HIP(ctx, hipMemcpy(logits, ctx->logits + (size_t)slot * ctx->c.vocab_size, ctx->c.vocab_size * sizeof(float), hipMemcpyDeviceToHost));
    if (hidden)
        FT_HIP(ctx, hipMemcpy(hidden, ctx->hid + (size_t)slot * ctx->c.fast_dim, ctx->c.fast_dim * sizeof(float), hipMemcpyDeviceToHost));
    return FT_OK;
}

extern "C" ft_status ft_ar_profile_gemv(ft_ctx* ctx, int32_t frames, const ft_sampling* sp, double* ms,
                                        int64_t* launches, int64_t* bytes) {
    FT_TRY(ar_ready(ctx));
    if (!sp || !ms || !launches || !bytes) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_profile_gemv: null argument");
    const int R = ctx->c.num_codebooks + 1;
    FT_TRY(upload_ctl(ctx, 0, 1, sp));
    for (auto e : ctx->prof_ev) hipEventDestroy(e);
    ctx->prof_ev.clear();
    ctx->prof_bytes = ctx->prof_launches = 0;
    // The GEMV launches of one frame (weights of every layer in order, nothing else) are captured into a graph
    // and replayed `frames` times between two HIP events on the engine's stream: the average includes the
    // dependent-launch boundary exactly as rocprofv3's serialized kernel durations do.  The launch and byte
    // counters come from the same enqueue code as the real frame.
    double tot = 0.0;
    int64_t n_launch = 0, n_bytes = 0;
    bool graph_ok = true;
    if (graph_ok) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
        Launch L{ctx, ctx->stream, 0, 1, 0};
        L.gemv_only = true;
        ctx->prof_count_only = true;
        ctx->prof = true;
        if (e == hipSuccess) {
            enqueue_frame(L, ctx->d_tok, 1, R, 0);
            e = hipStreamEndCapture(ctx->stream, &graph);
        }
        ctx->prof = false;
        ctx->prof_count_only = false;
        if (e == hipSuccess && L.err == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) hipGraphDestroy(graph);
        graph_ok = e == hipSuccess && L.err == hipSuccess && exec != nullptr;
        if (graph_ok) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipGraphLaunch(exec, ctx->stream);  // warm
            hipEventRecord(e0, ctx->stream);
            for (int f = 0; f < frames; ++f) hipGraphLaunch(exec, ctx->stream);
            hipEventRecord(e1, ctx->stream);
            graph_ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
            float t = 0.f;
            if (graph_ok) graph_ok = hipEventElapsedTime(&t, e0, e1) == hipSuccess;
            tot = t;
            n_launch = ctx->prof_launches * frames;
            n_bytes = ctx->prof_bytes * frames;
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        if (exec) hipGraphExecDestroy(exec);
        (void)hipGetLastError();
    }
    if (!graph_ok) {
        for (auto e : ctx->prof_ev) hipEventDestroy(e);
        ctx->prof_ev.clear();
        ctx->prof_bytes = ctx->prof_launches = 0;
        tot = 0.0;
        ctx->prof = true;
        for (int f = 0; f < frames; ++f) {
            Launch L{ctx, ctx->stream, 0, 1, 0};
            enqueue_frame(L, ctx->d_tok, 1, R, 0);
            if (L.err != hipSuccess) { ctx->prof = false; return ft_fail(ctx, FT_ERR_HIP, "profile launch failed"); }
        }
        ctx->prof = false;
        FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i + 1 < ctx->prof_ev.size(); i += 2) {
            float t = 0.f;
            hipEventElapsedTime(&t, ctx->prof_ev[i], ctx->prof_ev[i + 1]);
            tot += t;
        }
        n_launch = ctx->prof_launches;
        n_bytes = ctx->prof_bytes;
    }
    *ms = tot; *launches = n_launch; *bytes = n_bytes;
    for (auto e : ctx->prof_ev) hipEventDestroy(e);
    ctx->prof_ev.clear();
    return FT_OK;
}

extern "C" ft_status ft_ar_profile_frame(ft_ctx* ctx, int32_t frames, const ft_sampling* sp, double* ms_graph,
                                         double* seg_ms, int32_t* nodes_per_frame) {
    FT_TRY(ar_ready(ctx));
    if (!sp || frames < 1) return ft_fail(ctx, FT_ERR_ARG, "ft_ar_profile_frame: bad argument");
    const ft_ar_config& c = ctx->c;
    const int R = c.num_codebooks + 1;
    FT_TRY(upload_ctl(ctx, 0, 1, sp));
    int h[2] = {0, 0};
    FT_HIP(ctx, hipMemcpy(h, ctx->d_nf, sizeof(int), hipMemcpyDeviceToHost));
    FT_HIP(ctx, hipMemcpy(h + 1, ctx->d_pos, sizeof(int), hipMemcpyDeviceToHost));
    if (h[0] < 1) return ft_fail(ctx, FT_ERR_STATE, "ft_ar_profile_frame: slot 0 has not been prefilled");
    if (h[0] + 2 * frames + 2 > ctx->cap || h[1] + 2 * frames + 2 > ctx->n_slots)
        return ft_fail(ctx, FT_ERR_ARG, "ft_ar_profile_frame: not enough frame / cache capacity left in slot 0");
    hipEvent_t ev[4];
    for (auto& e : ev) FT_HIP(ctx, hipEventCreate(&e));
    // (1) the captured frame graph, replayed back to back
    hipGraphExec_t exec = nullptr;
    FT_TRY(get_graph(ctx, 1, &exec));
    FT_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
    FT_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
    for (int f = 0; f < frames; ++f) FT_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
    FT_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
    FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float t = 0.f;
    FT_HIP(ctx, hipEventElapsedTime(&t, ev[0], ev[1]));
    if (ms_graph) *ms_graph = t;
    if (nodes_per_frame) { auto it = ctx->graph_nodes.find(graph_key(ctx, 1, 1)); *nodes_per_frame = it == ctx->graph_nodes.end() ? 0 : it->second; }
    // (2) the same frame launched eagerly, events between its three parts (bf16 only)
    double seg[3] = {0, 0, 0};
    if (seg_ms && c.dtype == FT_BF16) {
        for (int f = 0; f < frames - 1; ++f) {
            Launch L{ctx, ctx->stream, 0, 1, 0};
            FT_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
            if (eng_slow_ok(L)) enqueue_slow_engine(L, ctx->d_tok, 1, R, 0);
            else enqueue_slow<bf16_t, true>(L, ctx->d_tok, 1, R, 0, true);
            FT_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
            enqueue_head<bf16_t, true>(L);
            enqueue_sample<bf16_t, true>(L, 0, c.num_codebooks == 1);
            FT_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
            if (eng_fast_ok(L)) enqueue_fast_engine(L);
            else for (int cb = 0; cb < c.num_codebooks; ++cb) enqueue_fast_step<bf16_t, true>(L, cb);
            FT_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
            FT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (L.err != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, "ft_ar_profile_frame: launch failed");
            for (int k = 0; k < 3; ++k) { float d = 0.f; hipEventElapsedTime(&d, ev[k], ev[k + 1]); seg[k] += d; }
        }
        for (int k = 0; k < 3; ++k) seg_ms[k] = seg[k];
    } else if (seg_ms) {
        seg_ms[0] = seg_ms[1] = seg_ms[2] = 0.0;
    }
    for (auto& e : ev) hipEventDestroy(e);
    bool aborted = false;
    FT_TRY(eng_recover(ctx, &aborted));
    if (aborted) return ft_fail(ctx, FT_ERR_HIP, "ft_ar_profile_frame: a frame-engine hand-off timed out; the timing is invalid");
    return FT_OK;
}

extern "C" ft_status ft_test_sample(ft_ctx* ctx, const float* logits, int32_t cb, const ft_sampling* sp,
                                    const int32_t* window, const float* q, int32_t* out_index) {
    FT_TRY(ar_ready(ctx));
    const ft_ar_config& c = ctx->c;
    if (!logits || !sp || !out_index || cb < 0 || cb >= c.num_codebooks) return ft_fail(ctx, FT_ERR_ARG, "ft_test_sample: bad argument");
    const int R = c.num_codebooks + 1;
    const int V = cb == 0 ? c.vocab_size : ctx->fastV;
    FT_TRY(ft_ar_reset(ctx, 0));
    FT_TRY(upload_ctl(ctx, 0, 1, sp));
    FT_HIP(ctx, hipMemcpy(cb == 0 ? ctx->logits : ctx->flog, logits, (size_t)V * sizeof(float), hipMemcpyHostToDevice));
    int nfv = 0;
    if (window) {
        nfv = 1;  // iteration i = 0: the window is hist[:, 0:16] = seq[:, 1:17]
        for (int r = 0; r < R; ++r)
            FT_HIP(ctx, hipMemcpy(ctx->d_seq + (size_t)r * ctx->cap + 1, window + (size_t)r * 16, 16 * sizeof(int), hipMemcpyHostToDevice));
        FT_HIP(ctx, hipMemcpy(ctx->d_nf, &nfv, sizeof(int), hipMemcpyHostToDevice));
    }
    float* saved = ctx->noise;
    const long srows = ctx->noise_rows, slen = ctx->noise_row_len;
    float* tmp = nullptr;
    if (q) {
        const long need = (long)c.vocab_size + (long)(c.num_codebooks - 1) * ctx->fastV;
        FT_HIP(ctx, hipMalloc((void**)&tmp, (size_t)(nfv + 1) * need * sizeof(float)));
        const long off = cb == 0 ? 0 : (long)c.vocab_size + (long)(cb - 1) * ctx->fastV;
        FT_HIP(ctx, hipMemcpy(tmp + (size_t)nfv * need + off, q, (size_t)V * sizeof(float), hipMemcpyHostToDevice));
        ctx->noise = tmp; ctx->noise_rows = nfv + 1; ctx->noise_row_len = need;
    } else {
        ctx->noise = nullptr; ctx->noise_rows = 0;
    }
    Launch L{ctx, ctx->stream, 0, 1, 0};
    if (c.dtype == FT_BF16) enqueue_sample<bf16_t, RND_BF16>(L, cb, false);
    else if (c.dtype == FT_F16) enqueue_sample<f16_t, RND_F16>(L, cb, false);
    else enqueue_sample<float, RND_NONE>(L, cb, false);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    ctx->noise = saved; ctx->noise_rows = srows; ctx->noise_row_len = slen;
    if (tmp) hipFree(tmp);
    if (L.err != hipSuccess || e != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, "ft_test_sample launch failed");
    int tokn[2] = {0, 0};
    FT_HIP(ctx, hipMemcpy(tokn, ctx->d_tokn + (cb == 0 ? 0 : cb + 1), sizeof(int), hipMemcpyDeviceToHost));
    *out_index = tokn[0];
    return ft_ar_reset(ctx, 0);
}
