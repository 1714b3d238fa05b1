// DAC codec decode (DAC.decode, vocoder.py:906-912) on the tap-GEMM kernels of codec_kernels.h.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <mutex>

#include "codec_kernels.h"
#include "engine.h"

using namespace ft;

#define FT_TRY(x) do { ft_status s_ = (x); if (s_ != FT_OK) return s_; } while (0)

struct ConvW {            // one packed tap-GEMM weight
    bf16_t* w = nullptr;  // [ntap][N][K]
    float* bias = nullptr;
    int ntap = 1, N = 0, K = 0, n_mod = 0;
    int offs[8] = {0};
};
struct TfLayer { ConvW qkv, wo, w13, w2; float *n1, *n2, *g1, *g2; };
struct UpStage { ConvW ct; float *dw_w, *dw_b, *ln_w, *ln_b, *gamma; ConvW pw1, pw2; int f; };
struct ResUnitW { float *a0, *a2; ConvW c7, c1; };
struct DecBlock { float* a0; ConvW ct; ResUnitW u[3]; int s, cin, cout; };

struct ft_codec_stream;
struct CodecState {
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::vector<ft_codec_stream*> streams;   // live streamed decodes of this context (their device state is freed with it)
    float* tables = nullptr;  // RVQ tables
    float* rope = nullptr;
    std::vector<TfLayer> tf;
    float* tf_norm = nullptr;
    std::vector<UpStage> up;
    ConvW conv_in;
    std::vector<DecBlock> blocks;
    float* a_last = nullptr;
    float* w_last = nullptr;  // [7][C]
    float b_last = 0.f;
    int c_last = 0;
    std::vector<void*> owned;
    // activations (one batch item at a time)
    int* codes = nullptr;
    float *x = nullptr, *audio = nullptr;
    bf16_t *xn = nullptr, *qkv = nullptr, *y = nullptr, *g = nullptr;
    bf16_t* big[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t big_elems = 0, big_margin = 0;
    int frame_len = 0, up_total = 1;
    // ---- encode side
    struct EncUnit { float *a0, *a2; ConvW c7, c1; };
    struct EncBlock { EncUnit u[3]; float* a3; ConvW sc; int s, cin, cout; std::vector<TfLayer> tf; float* tf_norm = nullptr; };
    bool has_enc = false;
    float *enc_w0 = nullptr, *enc_b0 = nullptr;   // first conv [C][7], [C]
    std::vector<EncBlock> enc;
    float* enc_a_last = nullptr;
    ConvW enc_out;                                 // k = 3 conv to the latent
    std::vector<UpStage> down;                     // quantizer.downsample (ct = strided conv here)
    std::vector<TfLayer> pre;
    float* pre_norm = nullptr;
    float *rope_enc = nullptr, *inw = nullptr, *inb = nullptr, *cbn = nullptr, *cn2 = nullptr;
    float *enc_audio = nullptr, *enc_x = nullptr, *enc_zq = nullptr;
    bf16_t *ebuf[4] = {nullptr, nullptr, nullptr, nullptr}, *e_xn = nullptr, *e_qkv = nullptr, *e_y = nullptr, *e_g = nullptr;
    int* enc_codes = nullptr;
    int hop = 1, enc_frame_len = 0;
    long max_samples = 0;
};

static std::string cname(const char* fmt, int a = 0, int b = 0) {
    char buf[160];
    snprintf(buf, sizeof buf, fmt, a, b);
    return buf;
}

// samples one encode call may hold, and the most positions an encoder transformer sees (head_dim 64: 32 rope pairs)
static long enc_max_samples(const ft_codec_config& c) {
    long hop = 1;
    for (int i = 0; i < c.n_enc_rates; ++i) hop *= c.enc_rates[i];
    return (long)c.max_enc_frames * hop * 4;
}
static int64_t enc_rope_positions(const ft_codec_config& c) {
    long rate = 1, best = 1;
    for (int i = 0; i < c.n_enc_rates; ++i) {
        rate *= c.enc_rates[i];
        if (c.enc_tf_layers[i] > 0) best = std::max(best, enc_max_samples(c) / rate);
    }
    return best;
}

void codec_expected(ft_ctx* ctx) {
    const ft_codec_config& c = ctx->cc;
    const int D = c.latent_dim, H = c.tf_n_head * c.tf_head_dim;
    auto E = [&](const std::string& n, std::vector<int64_t> s) { ft_expect(ctx, n, std::move(s), FT_F32); };
    E("quantizer.semantic_quantizer.quantizers.0.codebook.weight", {c.semantic_codebook_size, c.codebook_dim});
    E("quantizer.semantic_quantizer.quantizers.0.out_proj.weight", {D, c.codebook_dim, 1});
    E("quantizer.semantic_quantizer.quantizers.0.out_proj.bias", {D});
    for (int i = 0; i < c.n_codebooks; ++i) {
        E(cname("quantizer.quantizer.quantizers.%d.codebook.weight", i), {c.codebook_size, c.codebook_dim});
        E(cname("quantizer.quantizer.quantizers.%d.out_proj.weight", i), {D, c.codebook_dim, 1});
        E(cname("quantizer.quantizer.quantizers.%d.out_proj.bias", i), {D});
    }
    for (int l = 0; l < c.n_tf_layer; ++l) {
        const std::string p = cname("quantizer.post_module.layers.%d", l);
        E(p + ".attention.wqkv.weight", {3 * H, D});
        E(p + ".attention.wo.weight", {D, H});
        E(p + ".feed_forward.w1.weight", {c.tf_ffn, D});
        E(p + ".feed_forward.w3.weight", {c.tf_ffn, D});
        E(p + ".feed_forward.w2.weight", {D, c.tf_ffn});
        E(p + ".ffn_norm.weight", {D});
        E(p + ".attention_norm.weight", {D});
        E(p + ".attention_layer_scale.gamma", {D});
        E(p + ".ffn_layer_scale.gamma", {D});
    }
    E("quantizer.post_module.norm.weight", {D});
    for (int j = 0; j < c.n_upsample; ++j) {
        const std::string p = cname("quantizer.upsample.%d", j);
        E(p + ".0.conv.weight", {D, D, 2});
        E(p + ".0.conv.bias", {D});
        E(p + ".1.dwconv.conv.weight", {D, 1, 7});
        E(p + ".1.dwconv.conv.bias", {D});
        E(p + ".1.norm.weight", {D});
        E(p + ".1.norm.bias", {D});
        E(p + ".1.pwconv1.weight", {4 * D, D});
        E(p + ".1.pwconv1.bias", {4 * D});
        E(p + ".1.pwconv2.weight", {D, 4 * D});
        E(p + ".1.pwconv2.bias", {D});
        E(p + ".1.gamma", {D});
    }
    E("decoder.model.0.conv.weight", {c.decoder_dim, D, 7});
    E("decoder.model.0.conv.bias", {c.decoder_dim});
    for (int i = 0; i < c.n_rates; ++i) {
        const int cin = c.decoder_dim >> i, cout = c.decoder_dim >> (i + 1), r = c.rates[i];
        const std::string p = cname("decoder.model.%d.block", i + 1);
        E(p + ".0.alpha", {1, cin, 1});
        E(p + ".1.conv.weight", {cin, cout, 2 * r});
        E(p + ".1.conv.bias", {cout});
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + cname(".%d.block", u + 2);
            E(q + ".0.alpha", {1, cout, 1});
            E(q + ".1.conv.weight", {cout, cout, 7});
            E(q + ".1.conv.bias", {cout});
            E(q + ".2.alpha", {1, cout, 1});
            E(q + ".3.conv.weight", {cout, cout, 1});
            E(q + ".3.conv.bias", {cout});
        }
    }
    const int last = c.decoder_dim >> c.n_rates;
    E(cname("decoder.model.%d.alpha", c.n_rates + 1), {1, last, 1});
    E(cname("decoder.model.%d.conv.weight", c.n_rates + 2), {1, last, 7});
    E(cname("decoder.model.%d.conv.bias", c.n_rates + 2), {1});
    ft_expect(ctx, "rope.codec", {c.max_frames, c.tf_head_dim / 2, 2}, FT_F32);
    if (c.encoder_dim <= 0) return;
    // encode side: Encoder (vocoder.py:539-575), quantizer.downsample / pre_module / in_proj (683-757)
    int d = c.encoder_dim;
    E("encoder.block.0.conv.weight", {d, 1, 7});
    E("encoder.block.0.conv.bias", {d});
    for (int i = 0; i < c.n_enc_rates; ++i) {
        d *= 2;
        const std::string p = cname("encoder.block.%d.block", i + 1);
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + cname(".%d.block", u);
            E(q + ".0.alpha", {1, d / 2, 1});
            E(q + ".1.conv.weight", {d / 2, d / 2, 7});
            E(q + ".1.conv.bias", {d / 2});
            E(q + ".2.alpha", {1, d / 2, 1});
            E(q + ".3.conv.weight", {d / 2, d / 2, 1});
            E(q + ".3.conv.bias", {d / 2});
        }
        E(p + ".3.alpha", {1, d / 2, 1});
        E(p + ".4.conv.weight", {d, d / 2, 2 * c.enc_rates[i]});
        E(p + ".4.conv.bias", {d});
        for (int l = 0; l < c.enc_tf_layers[i]; ++l) {
            const std::string t = p + cname(".5.layers.%d", l);
            E(t + ".attention.wqkv.weight", {3 * d, d});
            E(t + ".attention.wo.weight", {d, d});
            E(t + ".feed_forward.w1.weight", {3 * d, d});
            E(t + ".feed_forward.w3.weight", {3 * d, d});
            E(t + ".feed_forward.w2.weight", {d, 3 * d});
            E(t + ".ffn_norm.weight", {d});
            E(t + ".attention_norm.weight", {d});
            E(t + ".attention_layer_scale.gamma", {d});
            E(t + ".ffn_layer_scale.gamma", {d});
        }
        if (c.enc_tf_layers[i]) E(p + ".5.norm.weight", {d});
    }
    E(cname("encoder.block.%d.alpha", c.n_enc_rates + 1), {1, d, 1});
    E(cname("encoder.block.%d.conv.weight", c.n_enc_rates + 2), {D, d, 3});
    E(cname("encoder.block.%d.conv.bias", c.n_enc_rates + 2), {D});
    for (int j = 0; j < c.n_upsample; ++j) {
        const std::string p = cname("quantizer.downsample.%d", j);
        E(p + ".0.conv.weight", {D, D, 2});
        E(p + ".0.conv.bias", {D});
        E(p + ".1.dwconv.conv.weight", {D, 1, 7});
        E(p + ".1.dwconv.conv.bias", {D});
        E(p + ".1.norm.weight", {D});
        E(p + ".1.norm.bias", {D});
        E(p + ".1.pwconv1.weight", {4 * D, D});
        E(p + ".1.pwconv1.bias", {4 * D});
        E(p + ".1.pwconv2.weight", {D, 4 * D});
        E(p + ".1.pwconv2.bias", {D});
        E(p + ".1.gamma", {D});
    }
    for (int l = 0; l < c.n_tf_layer; ++l) {
        const std::string p = cname("quantizer.pre_module.layers.%d", l);
        E(p + ".attention.wqkv.weight", {3 * H, D});
        E(p + ".attention.wo.weight", {D, H});
        E(p + ".feed_forward.w1.weight", {c.tf_ffn, D});
        E(p + ".feed_forward.w3.weight", {c.tf_ffn, D});
        E(p + ".feed_forward.w2.weight", {D, c.tf_ffn});
        E(p + ".ffn_norm.weight", {D});
        E(p + ".attention_norm.weight", {D});
        E(p + ".attention_layer_scale.gamma", {D});
        E(p + ".ffn_layer_scale.gamma", {D});
    }
    E("quantizer.pre_module.norm.weight", {D});
    E("quantizer.semantic_quantizer.quantizers.0.in_proj.weight", {c.codebook_dim, D, 1});
    E("quantizer.semantic_quantizer.quantizers.0.in_proj.bias", {c.codebook_dim});
    for (int i = 0; i < c.n_codebooks; ++i) {
        E(cname("quantizer.quantizer.quantizers.%d.in_proj.weight", i), {c.codebook_dim, D, 1});
        E(cname("quantizer.quantizer.quantizers.%d.in_proj.bias", i), {c.codebook_dim});
    }
    ft_expect(ctx, "rope.codec_enc", {enc_rope_positions(c), 32, 2}, FT_F32);
}

ft_status codec_create(ft_ctx* ctx) {
    const ft_codec_config& c = ctx->cc;
    auto bad = [&](const char* m) { return ft_fail(ctx, FT_ERR_UNSUPPORTED, m); };
    if (c.dtype != FT_BF16) return bad("codec: only FT_BF16 contractions (f32 accumulate) are implemented");
    if (c.n_upsample < 0 || c.n_upsample > 4 || c.n_rates < 1 || c.n_rates > 8) return bad("codec: bad stage counts");
    if (c.latent_dim % 32 || (c.tf_n_head * c.tf_head_dim) % 32 || c.tf_ffn % 32 || c.decoder_dim % 32)
        return bad("codec: channel counts must be multiples of 32");
    if ((c.decoder_dim >> c.n_rates) % 32) return bad("codec: decoder_dim / 2^n_rates must be a multiple of 32");
    if (c.tf_head_dim > 128 || c.tf_head_dim % 8 || c.tf_window > 512) return bad("codec: head_dim <= 128, window <= 512");
    if (c.encoder_dim > 0) {
        if (c.n_enc_rates < 1 || c.n_enc_rates > 8 || c.encoder_dim % 32 || c.max_enc_frames < 1 || c.enc_tf_window > 512)
            return bad("codec: encoder_dim must be a multiple of 32, 1..8 encoder rates, enc_tf_window <= 512");
        if ((c.encoder_dim << c.n_enc_rates) != c.latent_dim) return bad("codec: latent_dim must be encoder_dim * 2^n_enc_rates");
        for (int i = 0; i < c.n_enc_rates; ++i)
            if (c.enc_tf_layers[i] > 0 && (c.encoder_dim << (i + 1)) % 64) return bad("codec: encoder transformer width must be a multiple of 64");
        if (c.codebook_dim > 16) return bad("codec: codebook_dim <= 16");
        if (c.max_enc_frames > c.max_frames) return bad("codec: max_enc_frames must not exceed max_frames (shared rope table)");
    }
    if (c.max_frames < 1 || c.max_batch < 1) return bad("codec: max_frames / max_batch");
    CodecState* s = new CodecState();
    ctx->codec = s;
    FT_HIP(ctx, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->up_total = 1 << c.n_upsample;
    s->frame_len = s->up_total;
    for (int i = 0; i < c.n_rates; ++i) s->frame_len *= c.rates[i];
    s->has_enc = c.encoder_dim > 0;
    if (s->has_enc) {
        s->hop = 1;
        for (int i = 0; i < c.n_enc_rates; ++i) s->hop *= c.enc_rates[i];
        s->enc_frame_len = s->hop * 4;   // DAC.frame_length (vocoder.py:872)
        s->max_samples = enc_max_samples(c);
    }
    return FT_OK;
}

static void stream_orphan(ft_codec_stream* sc);
void codec_destroy(ft_ctx* ctx) {
    CodecState* s = ctx->codec;
    if (!s) return;
    if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    // streams the caller has not ended yet: their device state goes with the context, the host handle stays valid for
    // ft_codec_stream_end (which then only deletes it) and is refused by ft_codec_stream_decode
    for (ft_codec_stream* sc : s->streams) stream_orphan(sc);
    s->streams.clear();
    for (void* p : s->owned) hipFree(p);
    delete s;
    ctx->codec = nullptr;
}

template <typename T>
static ft_status cmalloc(ft_ctx* ctx, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 64);
    if (e != hipSuccess) return ft_fail(ctx, FT_ERR_NOMEM, std::string("codec hipMalloc: ") + hipGetErrorString(e));
    ctx->codec->owned.push_back(q);
    *p = (T*)q;
    return FT_OK;
}
static float* W32(ft_ctx* ctx, const std::string& n) { return (float*)ctx->expected[n].p; }
static int gridfor(long n) { long b = (n + 255) / 256; return (int)(b < 4096 ? b : 4096); }

static ft_status pack_linear(ft_ctx* ctx, ConvW& cw, const std::string& wname, const std::string& bname, int N, int K) {
    CodecState* s = ctx->codec;
    FT_TRY(cmalloc(ctx, &cw.w, (size_t)N * K));
    pack_rows_kernel<<<gridfor((long)N * K), 256, 0, s->stream>>>(W32(ctx, wname), cw.w, (long)N * K);
    cw.bias = bname.empty() ? nullptr : W32(ctx, bname);
    cw.ntap = 1; cw.N = N; cw.K = K; cw.n_mod = N; cw.offs[0] = 0;
    return FT_OK;
}
static ft_status pack_conv(ft_ctx* ctx, ConvW& cw, const std::string& pfx, int Cout, int Cin, int k, int dil) {
    CodecState* s = ctx->codec;
    FT_TRY(cmalloc(ctx, &cw.w, (size_t)Cout * Cin * k));
    pack_conv_kernel<<<gridfor((long)Cout * Cin * k), 256, 0, s->stream>>>(W32(ctx, pfx + ".weight"), cw.w, Cout, Cin, k);
    cw.bias = W32(ctx, pfx + ".bias");
    cw.ntap = k; cw.N = Cout; cw.K = Cin; cw.n_mod = Cout;
    for (int kk = 0; kk < k; ++kk) cw.offs[kk] = (kk - (k - 1)) * dil;
    return FT_OK;
}
static ft_status pack_convT(ft_ctx* ctx, ConvW& cw, const std::string& pfx, int Cin, int Cout, int k, int stride) {
    CodecState* s = ctx->codec;
    FT_TRY(cmalloc(ctx, &cw.w, (size_t)Cin * Cout * k));
    pack_convT_kernel<<<gridfor((long)Cin * Cout * k), 256, 0, s->stream>>>(W32(ctx, pfx + ".weight"), cw.w, Cin, Cout, k, stride);
    cw.bias = W32(ctx, pfx + ".bias");
    cw.ntap = k / stride; cw.N = stride * Cout; cw.K = Cin; cw.n_mod = Cout;
    for (int j = 0; j < cw.ntap; ++j) cw.offs[j] = -j;
    return FT_OK;
}

static ft_status pack_strided(ft_ctx* ctx, ConvW& cw, const std::string& pfx, int Cout, int Cin, int k, int stride) {
    CodecState* s = ctx->codec;
    FT_TRY(cmalloc(ctx, &cw.w, (size_t)Cout * Cin * k));
    pack_strided_conv_kernel<<<gridfor((long)Cout * Cin * k), 256, 0, s->stream>>>(W32(ctx, pfx + ".weight"), cw.w, Cout, Cin, k, stride);
    cw.bias = W32(ctx, pfx + ".bias");
    cw.ntap = k / stride; cw.N = Cout; cw.K = stride * Cin; cw.n_mod = Cout;
    for (int a = 0; a < cw.ntap; ++a) cw.offs[a] = a - (cw.ntap - 1);
    return FT_OK;
}
static ft_status pack_tf_layer(ft_ctx* ctx, TfLayer& t, const std::string& p, int D, int H, int ffn) {
    CodecState* s = ctx->codec;
    FT_TRY(pack_linear(ctx, t.qkv, p + ".attention.wqkv.weight", "", 3 * H, D));
    FT_TRY(pack_linear(ctx, t.wo, p + ".attention.wo.weight", "", D, H));
    FT_TRY(pack_linear(ctx, t.w2, p + ".feed_forward.w2.weight", "", D, ffn));
    FT_TRY(cmalloc(ctx, &t.w13.w, (size_t)2 * ffn * D));
    pack_interleave_kernel<<<gridfor((long)ffn * D), 256, 0, s->stream>>>(
        W32(ctx, p + ".feed_forward.w1.weight"), W32(ctx, p + ".feed_forward.w3.weight"), t.w13.w, ffn, D);
    t.w13.ntap = 1; t.w13.N = 2 * ffn; t.w13.K = D; t.w13.n_mod = 2 * ffn; t.w13.bias = nullptr;
    t.n1 = W32(ctx, p + ".attention_norm.weight"); t.n2 = W32(ctx, p + ".ffn_norm.weight");
    t.g1 = W32(ctx, p + ".attention_layer_scale.gamma"); t.g2 = W32(ctx, p + ".ffn_layer_scale.gamma");
    return FT_OK;
}

static ft_status codec_finalize_encoder(ft_ctx* ctx) {
    const ft_codec_config& c = ctx->cc;
    CodecState* s = ctx->codec;
    const int D = c.latent_dim, H = c.tf_n_head * c.tf_head_dim, cd = c.codebook_dim, R = c.n_codebooks + 1;
    s->enc_w0 = W32(ctx, "encoder.block.0.conv.weight");
    s->enc_b0 = W32(ctx, "encoder.block.0.conv.bias");
    s->enc.resize(c.n_enc_rates);
    int d = c.encoder_dim;
    long rate = 1, tf_rows = 0;
    int tf_dim = 0;
    for (int i = 0; i < c.n_enc_rates; ++i) {
        CodecState::EncBlock& b = s->enc[i];
        b.cin = d; b.cout = 2 * d; b.s = c.enc_rates[i];
        d *= 2;
        const std::string p = cname("encoder.block.%d.block", i + 1);
        const int dil[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + cname(".%d.block", u);
            b.u[u].a0 = W32(ctx, q + ".0.alpha"); b.u[u].a2 = W32(ctx, q + ".2.alpha");
            FT_TRY(pack_conv(ctx, b.u[u].c7, q + ".1.conv", b.cin, b.cin, 7, dil[u]));
            FT_TRY(pack_conv(ctx, b.u[u].c1, q + ".3.conv", b.cin, b.cin, 1, 1));
        }
        b.a3 = W32(ctx, p + ".3.alpha");
        FT_TRY(pack_strided(ctx, b.sc, p + ".4.conv", b.cout, b.cin, 2 * b.s, b.s));
        rate *= b.s;
        b.tf.resize(c.enc_tf_layers[i]);
        for (int l = 0; l < c.enc_tf_layers[i]; ++l)
            FT_TRY(pack_tf_layer(ctx, b.tf[l], p + cname(".5.layers.%d", l), b.cout, b.cout, 3 * b.cout));
        if (c.enc_tf_layers[i]) {
            b.tf_norm = W32(ctx, p + ".5.norm.weight");
            tf_rows = std::max(tf_rows, s->max_samples / rate);
            tf_dim = std::max(tf_dim, b.cout);
        }
    }
    s->enc_a_last = W32(ctx, cname("encoder.block.%d.alpha", c.n_enc_rates + 1));
    FT_TRY(pack_conv(ctx, s->enc_out, cname("encoder.block.%d.conv", c.n_enc_rates + 2), D, d, 3, 1));
    s->down.resize(c.n_upsample);
    for (int j = 0; j < c.n_upsample; ++j) {
        const std::string p = cname("quantizer.downsample.%d", j);
        UpStage& u = s->down[j];
        u.f = 2;
        FT_TRY(pack_strided(ctx, u.ct, p + ".0.conv", D, D, 2, 2));
        u.dw_w = W32(ctx, p + ".1.dwconv.conv.weight"); u.dw_b = W32(ctx, p + ".1.dwconv.conv.bias");
        u.ln_w = W32(ctx, p + ".1.norm.weight"); u.ln_b = W32(ctx, p + ".1.norm.bias");
        u.gamma = W32(ctx, p + ".1.gamma");
        FT_TRY(pack_linear(ctx, u.pw1, p + ".1.pwconv1.weight", p + ".1.pwconv1.bias", 4 * D, D));
        FT_TRY(pack_linear(ctx, u.pw2, p + ".1.pwconv2.weight", p + ".1.pwconv2.bias", D, 4 * D));
    }
    s->pre.resize(c.n_tf_layer);
    for (int l = 0; l < c.n_tf_layer; ++l)
        FT_TRY(pack_tf_layer(ctx, s->pre[l], cname("quantizer.pre_module.layers.%d", l), D, H, c.tf_ffn));
    s->pre_norm = W32(ctx, "quantizer.pre_module.norm.weight");
    s->rope_enc = W32(ctx, "rope.codec_enc");
    // quantiser search operands: in_proj (f32), normalised codebooks and their squared norms
    const size_t ncode = (size_t)c.semantic_codebook_size + (size_t)c.n_codebooks * c.codebook_size;
    FT_TRY(cmalloc(ctx, &s->inw, (size_t)R * cd * D));
    FT_TRY(cmalloc(ctx, &s->inb, (size_t)R * cd));
    FT_TRY(cmalloc(ctx, &s->cbn, ncode * cd));
    FT_TRY(cmalloc(ctx, &s->cn2, ncode));
    for (int q = 0; q < R; ++q) {
        const std::string p = q == 0 ? std::string("quantizer.semantic_quantizer.quantizers.0")
                                     : cname("quantizer.quantizer.quantizers.%d", q - 1);
        FT_HIP(ctx, hipMemcpyAsync(s->inw + (size_t)q * cd * D, W32(ctx, p + ".in_proj.weight"), (size_t)cd * D * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
        FT_HIP(ctx, hipMemcpyAsync(s->inb + (size_t)q * cd, W32(ctx, p + ".in_proj.bias"), (size_t)cd * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
        const int N = q == 0 ? c.semantic_codebook_size : c.codebook_size;
        const size_t off = q == 0 ? 0 : (size_t)c.semantic_codebook_size + (size_t)(q - 1) * c.codebook_size;
        normalize_codebook_kernel<<<(N + 255) / 256, 256, 0, s->stream>>>(W32(ctx, p + ".codebook.weight"), s->cbn + off * cd, s->cn2 + off, N, cd);
    }
    // activations: the widest stage is the first (samples x encoder_dim); transformer scratch for the widest transformer
    const size_t big = (size_t)s->max_samples * c.encoder_dim;
    for (int i = 0; i < 4; ++i) FT_TRY(cmalloc(ctx, &s->ebuf[i], std::max(big, (size_t)(s->max_samples / s->hop) * 4 * D)));
    FT_TRY(cmalloc(ctx, &s->enc_audio, (size_t)s->max_samples));
    const size_t rows = std::max<size_t>((size_t)tf_rows, (size_t)c.max_enc_frames);
    const int wd = std::max(tf_dim, D);
    FT_TRY(cmalloc(ctx, &s->enc_x, rows * wd));
    FT_TRY(cmalloc(ctx, &s->e_xn, rows * wd));
    FT_TRY(cmalloc(ctx, &s->e_qkv, rows * 3 * std::max(tf_dim, H)));
    FT_TRY(cmalloc(ctx, &s->e_y, rows * std::max(tf_dim, H)));
    FT_TRY(cmalloc(ctx, &s->e_g, rows * std::max(3 * tf_dim, c.tf_ffn)));
    FT_TRY(cmalloc(ctx, &s->enc_zq, (size_t)c.max_enc_frames * D));
    FT_TRY(cmalloc(ctx, &s->enc_codes, (size_t)R * c.max_enc_frames));
    return FT_OK;
}

ft_status codec_finalize(ft_ctx* ctx) {
    const ft_codec_config& c = ctx->cc;
    CodecState* s = ctx->codec;
    const int D = c.latent_dim, H = c.tf_n_head * c.tf_head_dim;
    // RVQ tables: out_proj folded into the codebooks (vocoder.py:809-810 + dac from_codes)
    FT_TRY(cmalloc(ctx, &s->tables, ((size_t)c.semantic_codebook_size + (size_t)c.n_codebooks * c.codebook_size) * D));
    {
        const std::string p = "quantizer.semantic_quantizer.quantizers.0";
        rvq_table_kernel<<<gridfor((long)c.semantic_codebook_size * D), 256, 0, s->stream>>>(
            W32(ctx, p + ".codebook.weight"), W32(ctx, p + ".out_proj.weight"), W32(ctx, p + ".out_proj.bias"),
            s->tables, c.semantic_codebook_size, D, c.codebook_dim);
        for (int i = 0; i < c.n_codebooks; ++i) {
            const std::string q = cname("quantizer.quantizer.quantizers.%d", i);
            rvq_table_kernel<<<gridfor((long)c.codebook_size * D), 256, 0, s->stream>>>(
                W32(ctx, q + ".codebook.weight"), W32(ctx, q + ".out_proj.weight"), W32(ctx, q + ".out_proj.bias"),
                s->tables + ((size_t)c.semantic_codebook_size + (size_t)i * c.codebook_size) * D, c.codebook_size, D,
                c.codebook_dim);
        }
    }
    s->rope = W32(ctx, "rope.codec");
    s->tf.resize(c.n_tf_layer);
    for (int l = 0; l < c.n_tf_layer; ++l) {
        const std::string p = cname("quantizer.post_module.layers.%d", l);
        TfLayer& t = s->tf[l];
        FT_TRY(pack_linear(ctx, t.qkv, p + ".attention.wqkv.weight", "", 3 * H, D));
        FT_TRY(pack_linear(ctx, t.wo, p + ".attention.wo.weight", "", D, H));
        FT_TRY(pack_linear(ctx, t.w2, p + ".feed_forward.w2.weight", "", D, c.tf_ffn));
        FT_TRY(cmalloc(ctx, &t.w13.w, (size_t)2 * c.tf_ffn * D));
        pack_interleave_kernel<<<gridfor((long)c.tf_ffn * D), 256, 0, s->stream>>>(
            W32(ctx, p + ".feed_forward.w1.weight"), W32(ctx, p + ".feed_forward.w3.weight"), t.w13.w, c.tf_ffn, D);
        t.w13.ntap = 1; t.w13.N = 2 * c.tf_ffn; t.w13.K = D; t.w13.n_mod = 2 * c.tf_ffn; t.w13.bias = nullptr;
        t.n1 = W32(ctx, p + ".attention_norm.weight"); t.n2 = W32(ctx, p + ".ffn_norm.weight");
        t.g1 = W32(ctx, p + ".attention_layer_scale.gamma"); t.g2 = W32(ctx, p + ".ffn_layer_scale.gamma");
    }
    s->tf_norm = W32(ctx, "quantizer.post_module.norm.weight");
    s->up.resize(c.n_upsample);
    for (int j = 0; j < c.n_upsample; ++j) {
        const std::string p = cname("quantizer.upsample.%d", j);
        UpStage& u = s->up[j];
        u.f = 2;
        FT_TRY(pack_convT(ctx, u.ct, p + ".0.conv", D, D, 2, 2));
        u.dw_w = W32(ctx, p + ".1.dwconv.conv.weight"); u.dw_b = W32(ctx, p + ".1.dwconv.conv.bias");
        u.ln_w = W32(ctx, p + ".1.norm.weight"); u.ln_b = W32(ctx, p + ".1.norm.bias");
        u.gamma = W32(ctx, p + ".1.gamma");
        FT_TRY(pack_linear(ctx, u.pw1, p + ".1.pwconv1.weight", p + ".1.pwconv1.bias", 4 * D, D));
        FT_TRY(pack_linear(ctx, u.pw2, p + ".1.pwconv2.weight", p + ".1.pwconv2.bias", D, 4 * D));
    }
    FT_TRY(pack_conv(ctx, s->conv_in, "decoder.model.0.conv", c.decoder_dim, D, 7, 1));
    s->blocks.resize(c.n_rates);
    size_t per_frame_max = (size_t)s->up_total * c.decoder_dim;  // conv_in output
    size_t tmul = s->up_total;
    for (int i = 0; i < c.n_rates; ++i) {
        DecBlock& b = s->blocks[i];
        b.cin = c.decoder_dim >> i; b.cout = c.decoder_dim >> (i + 1); b.s = c.rates[i];
        const std::string p = cname("decoder.model.%d.block", i + 1);
        b.a0 = W32(ctx, p + ".0.alpha");
        FT_TRY(pack_convT(ctx, b.ct, p + ".1.conv", b.cin, b.cout, 2 * b.s, b.s));
        const int dil[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + cname(".%d.block", u + 2);
            b.u[u].a0 = W32(ctx, q + ".0.alpha"); b.u[u].a2 = W32(ctx, q + ".2.alpha");
            FT_TRY(pack_conv(ctx, b.u[u].c7, q + ".1.conv", b.cout, b.cout, 7, dil[u]));
            FT_TRY(pack_conv(ctx, b.u[u].c1, q + ".3.conv", b.cout, b.cout, 1, 1));
        }
        tmul *= b.s;
        per_frame_max = std::max(per_frame_max, tmul * b.cout);
    }
    s->c_last = c.decoder_dim >> c.n_rates;
    s->a_last = W32(ctx, cname("decoder.model.%d.alpha", c.n_rates + 1));
    {   // [1][C][7] -> [7][C]
        const float* w = W32(ctx, cname("decoder.model.%d.conv.weight", c.n_rates + 2));
        std::vector<float> h((size_t)s->c_last * 7), o((size_t)s->c_last * 7);
        FT_HIP(ctx, hipMemcpy(h.data(), w, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int ci = 0; ci < s->c_last; ++ci) for (int k = 0; k < 7; ++k) o[(size_t)k * s->c_last + ci] = h[(size_t)ci * 7 + k];
        FT_TRY(cmalloc(ctx, &s->w_last, o.size()));
        FT_HIP(ctx, hipMemcpy(s->w_last, o.data(), o.size() * sizeof(float), hipMemcpyHostToDevice));
        FT_HIP(ctx, hipMemcpy(&s->b_last, W32(ctx, cname("decoder.model.%d.conv.bias", c.n_rates + 2)), sizeof(float), hipMemcpyDeviceToHost));
    }
    // activation buffers for one utterance of max_frames
    const size_t T = c.max_frames;
    per_frame_max = std::max(per_frame_max, (size_t)s->up_total * 4 * D);  // ConvNeXt hidden
    s->big_elems = T * per_frame_max;
    // 64 rows of the widest conv input in FRONT of every buffer: a streamed decode puts the previous chunk's last rows there
    // (the largest halo is 54 rows: k = 7, dilation 9)
    s->big_margin = (size_t)64 * std::max(c.decoder_dim, D);
    for (int i = 0; i < 4; ++i) {
        FT_TRY(cmalloc(ctx, &s->big[i], s->big_elems + s->big_margin));
        s->big[i] += s->big_margin;
    }
    FT_TRY(cmalloc(ctx, &s->codes, (size_t)(c.n_codebooks + 1) * T));
    FT_TRY(cmalloc(ctx, &s->x, T * D));
    FT_TRY(cmalloc(ctx, &s->xn, T * D));
    FT_TRY(cmalloc(ctx, &s->qkv, (T + (size_t)c.tf_window) * 3 * H));   // (+ window - 1 rows of carried K/V in a streamed decode)
    FT_TRY(cmalloc(ctx, &s->y, T * H));
    FT_TRY(cmalloc(ctx, &s->g, T * c.tf_ffn));
    FT_TRY(cmalloc(ctx, &s->audio, T * s->frame_len));
    if (s->has_enc) FT_TRY(codec_finalize_encoder(ctx));
    FT_HIP(ctx, hipStreamSynchronize(s->stream));
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ launch
struct GemmIO {
    const bf16_t* X; long ldx; int T_in; int M;
    const float* gamma = nullptr; const float* resid_f32 = nullptr; const bf16_t* resid_bf = nullptr; long ldr = 0;
    float* out_f32 = nullptr; bf16_t* out_bf = nullptr; bf16_t* out_act = nullptr; const float* alpha = nullptr;
    long ldo = 0; int act = ACT_NONE;
    long msel = 0;     // rows the kernel VARIANT is chosen for (0 = M): a streamed decode picks, for every chunk length, the variant
                       // a whole utterance of nominal length takes, so that its results do not depend on the chunking
    int t_min = 0;     // TapGemmP::t_min
};

static void gemm(hipStream_t st, const ConvW& w, const GemmIO& io) {
    TapGemmP p{};
    p.X = io.X; p.ldx = io.ldx; p.x_bstride = 0; p.T_in = io.T_in; p.W = w.w; p.ntap = w.ntap;
    for (int i = 0; i < w.ntap; ++i) p.offs[i] = w.offs[i];
    p.M = io.M; p.N = w.N; p.K = w.K; p.bias = w.bias; p.n_mod = w.n_mod; p.act = io.act; p.gamma = io.gamma;
    p.resid_f32 = io.resid_f32; p.resid_bf = io.resid_bf; p.ldr = io.ldr; p.out_f32 = io.out_f32; p.out_bf = io.out_bf;
    p.out_act = io.out_act; p.alpha = io.alpha; p.ldo = io.ldo; p.t_min = io.t_min;
    const long Msel = io.msel > 0 ? io.msel : io.M;
    int halo = 0;
    for (int i = 0; i < w.ntap; ++i) halo = std::max(halo, -w.offs[i]);
    // few rows (the 215-frame transformers, the first up-sampling stage): a 64x64 tile grid leaves most CUs idle and every
    // block walks all of K alone (20-75 us per GEMM); the skinny kernel cuts N into 16-row blocks and splits K over the
    // waves of a block (weights streamed once per 64 rows)
    // (measured per GEMM at 215 / 430 / 860 rows: N = 1024 skinny 8-13 us against 21-75 us; N = 3072 equal; N >= 4096
    // the tile kernel wins, 23 against 35 us: its grid is already >= 256 blocks there)
    constexpr long skinny_m = 1024, skinny_n = 2048;
    if (w.ntap == 1 && w.offs[0] == 0 && Msel <= skinny_m && w.N <= skinny_n && io.T_in >= io.M && w.K % 128 == 0 && w.N % 2 == 0 &&
        (io.act == ACT_NONE || io.act == ACT_SWIGLU || io.act == ACT_GELU) && !io.out_act) {
        p.ldw = 0;
        skinny_gemm_launch<4>(p, (io.M + 63) / 64, st);
        return;
    }
    if (w.K % 32 == 0 && halo <= 56) {  // pipelined kernel: A stripe shared by the taps, B double-buffered
#define FT_TG(BM_, BN_, BK_)                                                                                   \
    tapgemm64_kernel<BM_, BN_, BK_><<<dim3((io.M + BM_ - 1) / BM_, (w.N + BN_ - 1) / BN_, 1), 256,              \
                                      std::max((size_t)((BM_ + 56) + 2 * BN_) * (BK_ + 8) * 2,                  \
                                               (size_t)(BM_ / 2) * (BN_ + 4) * 4), st>>>(p)
        // 64-row tiles on 4 waves below 4096 rows: the 4-wave 128 x 128 instantiation spills registers and, at these sizes,
        // leaves CUs idle (215-frame decode 9.9 -> 6.2 ms)
        const bool vec_ok = io.act != ACT_SWIGLU && w.N % 8 == 0 && w.n_mod % 8 == 0 && io.ldo % 8 == 0 && io.ldr % 8 == 0;
        const bool k64 = w.K % 64 == 0;
        // many rows (the decoder's convolutions after the first up-sampling): 128-row tiles on 8 waves - the 64 x 64 tile is
        // bound by the L2 bandwidth its weight-tile re-reads need (codec_kernels.h)
#define FT_TG8(BM_, BN_, BK_, NWM_, NWN_)                                                                       \
    do {                                                                                                          \
        constexpr size_t lds8_ = std::max((size_t)((BM_ + 56) + 2 * BN_) * (BK_ + 8) * 2,                         \
                                          (size_t)(BM_ / NWM_) * (BN_ + 4) * 4);                                  \
        static DevOnce once8_;                                                                                    \
        once8_.run([] { hipFuncSetAttribute((const void*)tapgemm64_kernel<BM_, BN_, BK_, NWM_, NWN_>,             \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8_); });          \
        tapgemm64_kernel<BM_, BN_, BK_, NWM_, NWN_><<<dim3((io.M + BM_ - 1) / BM_, (w.N + BN_ - 1) / BN_, 1),     \
                                                      64 * NWM_ * NWN_, lds8_, st>>>(p);                          \
    } while (0)
        constexpr long tile8_m = 4096, wide_m = 30000;
        if (vec_ok && Msel >= tile8_m && (w.N % 128 == 0 || w.N % 96 == 0)) {
            // full-width tiles where the whole N fits one block column (A read once): 128 x 192 (N = 192, 384), 256 x 96 (N = 96)
            // (256 x 128 x 32 on 8 waves was measured slower: 4.76 against 4.57 ms per 215-frame decode)
            if (Msel >= wide_m && w.N % 192 == 0) { FT_TG8(128, 192, 32, 2, 4); return; }
            if (Msel >= wide_m && w.N == 96) { FT_TG8(256, 96, 32, 4, 2); return; }
            if (w.N % 128 == 0) { if (k64) FT_TG8(128, 128, 64, 2, 4); else FT_TG8(128, 128, 32, 2, 4); }
            else { if (k64) FT_TG8(128, 96, 64, 4, 2); else FT_TG8(128, 96, 32, 4, 2); }
            return;
        }
#undef FT_TG8
        if (w.N % 128 == 0 || (w.N % 96 != 0 && w.N > 96)) {
            if (k64) FT_TG(64, 64, 64); else FT_TG(64, 64, 32);
        } else if (w.N % 96 == 0) {
            if (k64) FT_TG(64, 96, 64); else FT_TG(64, 96, 32);
        } else {
            if (k64) FT_TG(128, 64, 64); else FT_TG(128, 64, 32);
        }
#undef FT_TG
    } else if (w.N >= 128) {
        const dim3 grid((io.M + 127) / 128, (w.N + 127) / 128, 1);
        tapgemm_kernel<128, 128, 2, 2><<<grid, 256, 0, st>>>(p);
    } else {
        const dim3 grid((io.M + 127) / 128, (w.N + 63) / 64, 1);
        tapgemm_kernel<128, 64, 4, 1><<<grid, 256, 0, st>>>(p);
    }
}

// ---- streamed decode state (ft_codec_stream_*): what the causal codec needs from earlier chunks, two copies of each
// (a chunk reads one and leaves the other).  The codec is strictly causal: the window-128 attention reads the K / V of
// the 127 frames before a chunk (vocoder.py:325-332), every causal convolution the last `halo` rows of its input
// (vocoder.py:411-420, 449-455); with those carried, a chunk's samples are the whole decode's samples, bit for bit.
constexpr int STREAM_NOMINAL_FRAMES = 215;      // kernel variants are those of a 10 s utterance whatever the chunk length
struct ft_codec_stream {
    ft_ctx* owner = nullptr;   // the context whose codec the state belongs to; null once that context is gone (buffers freed)
    int t0 = 0;           // frames decoded so far (rope position of the next chunk's first frame)
    int par = 0;          // which copy is current
    std::vector<bf16_t*> kv[2];                    // per transformer layer: [window - 1][2 * H * hd], newest rows last
    struct Tail { bf16_t* buf[2]; int H, C; };
    std::vector<Tail> tails;                       // in the order decode_one consumes them
    std::vector<void*> owned;
};

static void stream_orphan(ft_codec_stream* sc) {
    for (void* v : sc->owned) hipFree(v);
    sc->owned.clear();
    sc->owner = nullptr;
}

static int halo_of(const ConvW& w) {
    int h = 0;
    for (int i = 0; i < w.ntap; ++i) h = std::max(h, -w.offs[i]);
    return h;
}

static ft_status decode_one(ft_ctx* ctx, const int32_t* codes_host, int Tfull, int T, float* audio_host, ft_codec_stream* sc = nullptr) {
    const ft_codec_config& c = ctx->cc;
    CodecState* s = ctx->codec;
    hipStream_t st = s->stream;
    const int D = c.latent_dim, H = c.tf_n_head, hd = c.tf_head_dim, HD = H * hd, R = c.n_codebooks + 1;
    const long Tn = sc ? STREAM_NOMINAL_FRAMES : 0;           // msel = Tn x rows per frame of the stage (0: by the real row count)
    const int W1 = c.tf_window - 1;
    const int nh = sc ? std::min(sc->t0, W1) : 0;             // carried K/V rows in front of the chunk's
    size_t ti = 0;                                            // next tail of sc->tails
    // the carried rows of x's earlier chunks in front of x (rows [-H, 0)), and the carry for the next chunk
    auto roll = [&](bf16_t* x, int rows, int Hh, int C) {
        if (!sc || Hh == 0) return 0;
        ft_codec_stream::Tail& tl = sc->tails[ti++];
        tail_roll_kernel<<<gridfor((long)Hh * C / 8), 256, 0, st>>>(x, tl.buf[sc->par], tl.buf[sc->par ^ 1], rows, Hh, C);
        return -Hh;
    };
    // codes of this item, compacted to [R][T]
    std::vector<int> hc((size_t)R * T);
    for (int r = 0; r < R; ++r) memcpy(&hc[(size_t)r * T], codes_host + (size_t)r * Tfull, T * sizeof(int));
    FT_HIP(ctx, hipMemcpyAsync(s->codes, hc.data(), hc.size() * sizeof(int), hipMemcpyHostToDevice, st));
    RvqP rq{s->codes, s->tables, c.n_codebooks, c.semantic_codebook_size, c.codebook_size, D, T, s->x};
    rvq_gather_kernel<<<dim3(T, 1), 256, 0, st>>>(rq);
    // post transformer (vocoder.py:338-354): residual stream f32, GEMM operands bf16
    bf16_t* qkv_c = s->qkv + (size_t)nh * 3 * HD;             // this chunk's rows of the q k v work buffer
    for (int l = 0; l < c.n_tf_layer; ++l) {
        const TfLayer& t = s->tf[l];
        rmsnorm_rows_kernel<<<T, 256, 0, st>>>(RowNormP{s->x, t.n1, c.tf_norm_eps, D, s->xn, nullptr});
        { GemmIO io{s->xn, D, T, T}; io.out_bf = qkv_c; io.ldo = 3 * HD; io.msel = Tn; gemm(st, t.qkv, io); }
        rope_qk_kernel<<<gridfor((long)T * 2 * H * (hd / 2)), 256, 0, st>>>(qkv_c, s->rope, T, H, hd, sc ? sc->t0 : 0);
        if (sc) {
            if (nh > 0) kv_carry_in_kernel<<<gridfor((long)nh * 2 * HD / 8), 256, 0, st>>>(s->qkv, sc->kv[sc->par][l], nh, W1, HD);
            const int nh2 = std::min(W1, nh + T);
            if (nh2 > 0) kv_carry_out_kernel<<<gridfor((long)nh2 * 2 * HD / 8), 256, 0, st>>>(s->qkv, sc->kv[sc->par ^ 1][l], nh + T, nh2, W1, HD);
        }
        window_attn_kernel<<<(T * H + 3) / 4, 256, 0, st>>>(WinAttnP{s->qkv, s->y, nh + T, H, hd, c.tf_window, 1.0f / sqrtf((float)hd), nh});
        { GemmIO io{s->y, HD, T, T}; io.gamma = t.g1; io.resid_f32 = s->x; io.ldr = D; io.out_f32 = s->x; io.ldo = D; io.msel = Tn; gemm(st, t.wo, io); }
        rmsnorm_rows_kernel<<<T, 256, 0, st>>>(RowNormP{s->x, t.n2, c.tf_norm_eps, D, s->xn, nullptr});
        { GemmIO io{s->xn, D, T, T}; io.act = ACT_SWIGLU; io.out_bf = s->g; io.ldo = c.tf_ffn; io.msel = Tn; gemm(st, t.w13, io); }
        { GemmIO io{s->g, c.tf_ffn, T, T}; io.gamma = t.g2; io.resid_f32 = s->x; io.ldr = D; io.out_f32 = s->x; io.ldo = D; io.msel = Tn; gemm(st, t.w2, io); }
    }
    bf16_t *z = s->big[0], *u = s->big[1], *n = s->big[2], *h = s->big[3];
    rmsnorm_rows_kernel<<<T, 256, 0, st>>>(RowNormP{s->x, s->tf_norm, c.tf_norm_eps, D, z, nullptr});
    int Tc = T;
    long Tnc = Tn;                                            // nominal rows at the current rate
    for (const UpStage& us : s->up) {  // vocoder.py:737-748: convT k=s=2, then ConvNeXt
        { GemmIO io{z, D, Tc, Tc}; io.out_bf = u; io.ldo = us.ct.N; io.msel = Tnc; gemm(st, us.ct, io); }
        Tc *= us.f;
        Tnc *= us.f;
        const int tm = roll(u, Tc, 6, D);                     // depthwise causal k = 7
        dwconv_ln_kernel<<<Tc, 256, D * sizeof(float), st>>>(DwLnP{u, us.dw_w, us.dw_b, us.ln_w, us.ln_b, Tc, D, n, tm});
        { GemmIO io{n, D, Tc, Tc}; io.act = ACT_GELU; io.out_bf = h; io.ldo = 4 * D; io.msel = Tnc; gemm(st, us.pw1, io); }
        { GemmIO io{h, 4 * D, Tc, Tc}; io.gamma = us.gamma; io.resid_bf = u; io.ldr = D; io.out_bf = z; io.ldo = D; io.msel = Tnc; gemm(st, us.pw2, io); }
    }
    // decoder (vocoder.py:605-640).  Buffers: a = snake'd input of the next conv, r = raw residual
    bf16_t *a = u, *r = n, *hs = h, *a2 = z;
    { GemmIO io{z, D, Tc, Tc}; io.out_act = a; io.alpha = s->blocks[0].a0; io.ldo = c.decoder_dim; io.msel = Tnc;
      io.t_min = roll(z, Tc, halo_of(s->conv_in), D); gemm(st, s->conv_in, io); }
    // note: conv_in reads z and writes a (= big[1]); z (= big[0]) is free afterwards
    for (size_t bi = 0; bi < s->blocks.size(); ++bi) {
        const DecBlock& b = s->blocks[bi];
        // transposed conv: raw -> r, snake'd by unit 0 -> a2
        { GemmIO io{a, b.cin, Tc, Tc}; io.out_bf = r; io.out_act = a2; io.alpha = b.u[0].a0; io.ldo = b.ct.N; io.msel = Tnc;
          io.t_min = roll(a, Tc, halo_of(b.ct), b.cin); gemm(st, b.ct, io); }
        Tc *= b.s;
        Tnc *= b.s;
        for (int ui = 0; ui < 3; ++ui) {
            const ResUnitW& ru = b.u[ui];
            { GemmIO io{a2, b.cout, Tc, Tc}; io.out_act = hs; io.alpha = ru.a2; io.ldo = b.cout; io.msel = Tnc;
              io.t_min = roll(a2, Tc, halo_of(ru.c7), b.cout); gemm(st, ru.c7, io); }
            const float* next_alpha = ui < 2 ? b.u[ui + 1].a0 : (bi + 1 < s->blocks.size() ? s->blocks[bi + 1].a0 : s->a_last);
            bf16_t* act_dst = ui < 2 ? a2 : a;  // the last unit feeds the next block's transposed conv / the output conv
            { GemmIO io{hs, b.cout, Tc, Tc}; io.resid_bf = r; io.ldr = b.cout; io.out_bf = ui < 2 ? r : nullptr;
              io.out_act = act_dst; io.alpha = next_alpha; io.ldo = b.cout; io.msel = Tnc; gemm(st, ru.c1, io); }
        }
    }
    FinalConvP fp{a, s->w_last, s->b_last, Tc, s->c_last, s->audio, roll(a, Tc, 6, s->c_last)};
    final_conv_tanh_kernel<<<2048, 256, 0, st>>>(fp);
    FT_HIP(ctx, hipMemcpyAsync(audio_host, s->audio, (size_t)Tc * sizeof(float), hipMemcpyDeviceToHost, st));
    FT_HIP(ctx, hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("codec launch: ") + hipGetErrorString(e));
    if (sc) {
        if (ti != sc->tails.size()) return ft_fail(ctx, FT_ERR_STATE, "codec stream: carry bookkeeping out of step");
        sc->t0 += T;
        sc->par ^= 1;
    }
    return FT_OK;
}

// Streamed decode (second half): successive chunks of one utterance's codes, each decoded with
// the context its predecessors left.  The reference decodes every chunk from zero state (synthesizer.py:513-528,
// 591-595: audible restarts at chunk borders); carrying the state is exact because the codec is causal.
extern "C" ft_status ft_codec_stream_begin(ft_ctx* ctx, ft_codec_stream** out) {
    if (!ctx || !out) return FT_ERR_ARG;
    if (!ctx->has_codec || !ctx->codec) return ft_fail(ctx, FT_ERR_STATE, "Vocoder not loaded");
    if (!ctx->finalized) return ft_fail(ctx, FT_ERR_STATE, "weights not finalized (ft_finalize_weights)");
    const ft_codec_config& c = ctx->cc;
    CodecState* s = ctx->codec;
    FT_HIP(ctx, hipSetDevice(ctx->device));
    ft_codec_stream* sc = new ft_codec_stream();
    auto zalloc = [&](bf16_t** q, size_t n) -> bool {
        void* v = nullptr;
        if (hipMalloc(&v, n * sizeof(bf16_t) + 64) != hipSuccess) return false;
        sc->owned.push_back(v);
        if (hipMemset(v, 0, n * sizeof(bf16_t) + 64) != hipSuccess) return false;
        *q = (bf16_t*)v;
        return true;
    };
    bool ok = true;
    const int HD = c.tf_n_head * c.tf_head_dim, W1 = std::max(c.tf_window - 1, 1);
    for (int k = 0; k < 2 && ok; ++k) {
        sc->kv[k].resize(c.n_tf_layer);
        for (int l = 0; l < c.n_tf_layer && ok; ++l) ok = zalloc(&sc->kv[k][l], (size_t)W1 * 2 * HD);
    }
    auto tail = [&](int Hh, int C) {
        if (Hh == 0 || !ok) return;
        ft_codec_stream::Tail t{{nullptr, nullptr}, Hh, C};
        ok = zalloc(&t.buf[0], (size_t)Hh * C) && zalloc(&t.buf[1], (size_t)Hh * C);
        sc->tails.push_back(t);
    };
    // the order decode_one consumes them in
    for (size_t j = 0; j < s->up.size(); ++j) tail(6, c.latent_dim);
    tail(halo_of(s->conv_in), c.latent_dim);
    for (const DecBlock& b : s->blocks) {
        tail(halo_of(b.ct), b.cin);
        for (int ui = 0; ui < 3; ++ui) tail(halo_of(b.u[ui].c7), b.cout);
    }
    tail(6, s->c_last);
    for (const auto& t : sc->tails) ok = ok && (size_t)t.H * t.C <= s->big_margin && t.C % 8 == 0;
    if (!ok) {
        for (void* v : sc->owned) hipFree(v);
        delete sc;
        (void)hipGetLastError();
        return ft_fail(ctx, FT_ERR_NOMEM, "ft_codec_stream_begin: could not set up the carried state");
    }
    sc->owner = ctx;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        s->streams.push_back(sc);
    }
    *out = sc;
    return FT_OK;
}

extern "C" ft_status ft_codec_stream_decode(ft_ctx* ctx, ft_codec_stream* sc, const int32_t* codes, int32_t T, float* audio) {
    if (!ctx || !sc) return FT_ERR_ARG;
    if (!ctx->has_codec || !ctx->codec) return ft_fail(ctx, FT_ERR_STATE, "Vocoder not loaded");
    if (!codes || !audio || T < 1) return ft_fail(ctx, FT_ERR_ARG, "ft_codec_stream_decode: bad argument");
    if (sc->owner != ctx) return ft_fail(ctx, FT_ERR_STATE, "ft_codec_stream_decode: the stream belongs to another (or a destroyed) context");
    const ft_codec_config& c = ctx->cc;
    if (T > c.max_frames) return ft_fail(ctx, FT_ERR_TOO_LONG, "ft_codec_stream_decode: chunk longer than max_frames");
    if (sc->t0 + T > c.max_frames) return ft_fail(ctx, FT_ERR_TOO_LONG, "ft_codec_stream_decode: stream longer than max_frames (rope table)");
    CodecState* s = ctx->codec;
    std::lock_guard<std::mutex> lock(s->mu);
    FT_HIP(ctx, hipSetDevice(ctx->device));
    return decode_one(ctx, codes, T, T, audio, sc);
}

// `ctx` may be null or stale: the stream knows its context, and a stream whose context was destroyed first has no device
// state left (codec_destroy freed it) - only the host handle is deleted then.
extern "C" void ft_codec_stream_end(ft_ctx* ctx, ft_codec_stream* sc) {
    (void)ctx;
    if (!sc) return;
    ft_ctx* own = sc->owner;
    if (own && own->codec) {
        CodecState* s = own->codec;
        std::lock_guard<std::mutex> lock(s->mu);
        hipSetDevice(own->device);
        hipStreamSynchronize(s->stream);
        for (void* v : sc->owned) hipFree(v);
        sc->owned.clear();
        for (size_t i = 0; i < s->streams.size(); ++i)
            if (s->streams[i] == sc) { s->streams.erase(s->streams.begin() + (long)i); break; }
    }
    delete sc;
}

extern "C" ft_status ft_codec_decode(ft_ctx* ctx, const int32_t* codes, int32_t B, int32_t T, const int32_t* lens,
                                     float* audio) {
    if (!ctx) return FT_ERR_ARG;
    if (!ctx->has_codec || !ctx->codec) return ft_fail(ctx, FT_ERR_STATE, "Vocoder not loaded");
    if (!ctx->finalized) return ft_fail(ctx, FT_ERR_STATE, "weights not finalized (ft_finalize_weights)");
    if (!codes || !audio || B < 1 || T < 1) return ft_fail(ctx, FT_ERR_ARG, "ft_codec_decode: bad argument");
    const ft_codec_config& c = ctx->cc;
    if (T > c.max_frames) return ft_fail(ctx, FT_ERR_TOO_LONG, "ft_codec_decode: T exceeds max_frames");
    CodecState* s = ctx->codec;
    std::lock_guard<std::mutex> lock(s->mu);
    FT_HIP(ctx, hipSetDevice(ctx->device));
    const int R = c.n_codebooks + 1;
    const size_t alen = (size_t)T * s->frame_len;
    for (int b = 0; b < B; ++b) {
        int Tb = lens ? lens[b] : T;
        if (Tb < 0 || Tb > T) return ft_fail(ctx, FT_ERR_ARG, "ft_codec_decode: bad length");
        float* out = audio + (size_t)b * alen;
        if ((size_t)Tb * s->frame_len < alen) memset(out + (size_t)Tb * s->frame_len, 0, (alen - (size_t)Tb * s->frame_len) * sizeof(float));
        if (Tb == 0) continue;
        FT_TRY(decode_one(ctx, codes + (size_t)b * R * T, T, Tb, out));
    }
    return FT_OK;
}

// One window-limited transformer (vocoder.py:338-354) over the f32 residual stream x [T][D]; output of the final
// RMSNorm goes to out_bf and/or out_f32.
static void run_transformer(ft_ctx* ctx, hipStream_t st, const std::vector<TfLayer>& layers, const float* final_norm,
                            float* x, int T, int D, int H, int hd, int ffn, int window, const float* rope,
                            bf16_t* xn, bf16_t* qkv, bf16_t* y, bf16_t* g, bf16_t* out_bf, float* out_f32) {
    const ft_codec_config& c = ctx->cc;
    const int HD = H * hd;
    for (const TfLayer& t : layers) {
        rmsnorm_rows_kernel<<<T, 256, 0, st>>>(RowNormP{x, t.n1, c.tf_norm_eps, D, xn, nullptr});
        { GemmIO io{xn, D, T, T}; io.out_bf = qkv; io.ldo = 3 * HD; gemm(st, t.qkv, io); }
        rope_qk_kernel<<<gridfor((long)T * 2 * H * (hd / 2)), 256, 0, st>>>(qkv, rope, T, H, hd);
        window_attn_kernel<<<(T * H + 3) / 4, 256, 0, st>>>(WinAttnP{qkv, y, T, H, hd, window, 1.0f / sqrtf((float)hd)});
        { GemmIO io{y, HD, T, T}; io.gamma = t.g1; io.resid_f32 = x; io.ldr = D; io.out_f32 = x; io.ldo = D; gemm(st, t.wo, io); }
        rmsnorm_rows_kernel<<<T, 256, 0, st>>>(RowNormP{x, t.n2, c.tf_norm_eps, D, xn, nullptr});
        { GemmIO io{xn, D, T, T}; io.act = ACT_SWIGLU; io.out_bf = g; io.ldo = ffn; gemm(st, t.w13, io); }
        { GemmIO io{g, ffn, T, T}; io.gamma = t.g2; io.resid_f32 = x; io.ldr = D; io.out_f32 = x; io.ldo = D; gemm(st, t.w2, io); }
    }
    rmsnorm_rows_kernel<<<T, 256, 0, st>>>(RowNormP{x, final_norm, c.tf_norm_eps, D, out_bf, out_f32});
}

static ft_status rvq_search(ft_ctx* ctx, hipStream_t st, const float* z, int T, int* codes_dev) {
    const ft_codec_config& c = ctx->cc;
    CodecState* s = ctx->codec;
    RvqEncP rp{z, s->inw, s->inb, s->cbn, s->cn2, s->tables, c.n_codebooks + 1, c.semantic_codebook_size, c.codebook_size,
               c.latent_dim, c.codebook_dim, T, codes_dev};
    rvq_encode_kernel<<<T, 256, (size_t)c.latent_dim * sizeof(float), st>>>(rp);
    return FT_OK;
}

extern "C" ft_status ft_codec_rvq_encode(ft_ctx* ctx, const float* z, int32_t T, int32_t* codes) {
    if (!ctx) return FT_ERR_ARG;
    if (!ctx->has_codec || !ctx->codec || !ctx->codec->has_enc) return ft_fail(ctx, FT_ERR_STATE, "codec encoder not configured");
    if (!ctx->finalized) return ft_fail(ctx, FT_ERR_STATE, "weights not finalized (ft_finalize_weights)");
    const ft_codec_config& c = ctx->cc;
    if (!z || !codes || T < 1 || T > c.max_enc_frames) return ft_fail(ctx, FT_ERR_ARG, "ft_codec_rvq_encode: bad argument");
    CodecState* s = ctx->codec;
    std::lock_guard<std::mutex> lock(s->mu);
    FT_HIP(ctx, hipSetDevice(ctx->device));
    const int R = c.n_codebooks + 1;
    FT_HIP(ctx, hipMemcpyAsync(s->enc_zq, z, (size_t)T * c.latent_dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
    FT_TRY(rvq_search(ctx, s->stream, s->enc_zq, T, s->enc_codes));
    FT_HIP(ctx, hipMemcpyAsync(codes, s->enc_codes, (size_t)R * T * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    FT_HIP(ctx, hipStreamSynchronize(s->stream));
    return FT_OK;
}

extern "C" ft_status ft_codec_encode(ft_ctx* ctx, const float* audio, int64_t n_samples, int32_t* codes, int32_t* out_frames) {
    if (!ctx) return FT_ERR_ARG;
    if (!ctx->has_codec || !ctx->codec) return ft_fail(ctx, FT_ERR_STATE, "Vocoder not loaded");
    CodecState* s = ctx->codec;
    if (!s->has_enc) return ft_fail(ctx, FT_ERR_STATE, "codec encoder not configured (encoder_dim = 0)");
    if (!ctx->finalized) return ft_fail(ctx, FT_ERR_STATE, "weights not finalized (ft_finalize_weights)");
    if (!audio || !codes || !out_frames || n_samples < 1) return ft_fail(ctx, FT_ERR_ARG, "ft_codec_encode: bad argument");
    const ft_codec_config& c = ctx->cc;
    const long fl = s->enc_frame_len;
    const long Tf = (n_samples + fl - 1) / fl;            // code frames (vocoder.py:891-892, 903)
    if (Tf > c.max_enc_frames) return ft_fail(ctx, FT_ERR_TOO_LONG, "ft_codec_encode: audio longer than max_enc_frames");
    const long T0 = Tf * fl;                              // padded samples
    std::lock_guard<std::mutex> lock(s->mu);
    FT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = s->stream;
    const int D = c.latent_dim, H = c.tf_n_head, hd = c.tf_head_dim, R = c.n_codebooks + 1;
    FT_HIP(ctx, hipMemsetAsync(s->enc_audio, 0, (size_t)T0 * sizeof(float), st));
    FT_HIP(ctx, hipMemcpyAsync(s->enc_audio, audio, (size_t)n_samples * sizeof(float), hipMemcpyHostToDevice, st));
    // Encoder (vocoder.py:539-575).  r = raw residual stream, a = Snake'd operand of the next conv, hs = inner buffer
    bf16_t *r = s->ebuf[0], *a = s->ebuf[1], *hs = s->ebuf[2], *o = s->ebuf[3];
    long T = T0;
    {
        EncInP ep{s->enc_audio, s->enc_w0, s->enc_b0, s->enc[0].u[0].a0, T, c.encoder_dim, r, a};
        enc_conv_in_kernel<<<gridfor(T * c.encoder_dim), 256, 0, st>>>(ep);
    }
    for (size_t bi = 0; bi < s->enc.size(); ++bi) {
        const CodecState::EncBlock& b = s->enc[bi];
        for (int ui = 0; ui < 3; ++ui) {
            const CodecState::EncUnit& ru = b.u[ui];
            { GemmIO io{a, b.cin, (int)T, (int)T}; io.out_act = hs; io.alpha = ru.a2; io.ldo = b.cin; gemm(st, ru.c7, io); }
            const float* next_alpha = ui < 2 ? b.u[ui + 1].a0 : b.a3;
            { GemmIO io{hs, b.cin, (int)T, (int)T}; io.resid_bf = r; io.ldr = b.cin; io.out_bf = ui < 2 ? r : nullptr;
              io.out_act = a; io.alpha = next_alpha; io.ldo = b.cin; gemm(st, ru.c1, io); }
        }
        // strided conv on the [T/s][s*cin] view of a; raw output to o (no transformer) or to the f32 stream
        const long Tn = T / b.s;
        const bool has_tf = !b.tf.empty();
        { GemmIO io{a, (long)b.s * b.cin, (int)Tn, (int)Tn}; if (has_tf) io.out_f32 = s->enc_x; else io.out_bf = o; io.ldo = b.cout; gemm(st, b.sc, io); }
        T = Tn;
        if (has_tf)
            run_transformer(ctx, st, b.tf, b.tf_norm, s->enc_x, (int)T, b.cout, b.cout / 64, 64, 3 * b.cout, c.enc_tf_window,
                            s->rope_enc, s->e_xn, s->e_qkv, s->e_y, s->e_g, o, nullptr);
        const float* alpha_next = bi + 1 < s->enc.size() ? s->enc[bi + 1].u[0].a0 : s->enc_a_last;
        snake_bf_rows_kernel<<<gridfor(T * b.cout), 256, 0, st>>>(o, alpha_next, a, T * b.cout, b.cout);
        std::swap(r, o);  // the raw output is the next block's residual stream
    }
    bf16_t* z = hs;   // [T][D]
    { GemmIO io{a, s->enc[s->enc.size() - 1].cout, (int)T, (int)T}; io.out_bf = z; io.ldo = D; gemm(st, s->enc_out, io); }
    // quantizer.downsample (vocoder.py:724-735): strided conv k = s = 2, ConvNeXt
    bf16_t *u = r, *n = a, *h = o;
    for (size_t j = 0; j < s->down.size(); ++j) {
        const UpStage& ds = s->down[j];
        const long Tn = T / ds.f;
        { GemmIO io{z, (long)ds.f * D, (int)Tn, (int)Tn}; io.out_bf = u; io.ldo = D; gemm(st, ds.ct, io); }
        T = Tn;
        dwconv_ln_kernel<<<(int)T, 256, D * sizeof(float), st>>>(DwLnP{u, ds.dw_w, ds.dw_b, ds.ln_w, ds.ln_b, (int)T, D, n});
        { GemmIO io{n, D, (int)T, (int)T}; io.act = ACT_GELU; io.out_bf = h; io.ldo = 4 * D; gemm(st, ds.pw1, io); }
        const bool last = j + 1 == s->down.size();
        { GemmIO io{h, 4 * D, (int)T, (int)T}; io.gamma = ds.gamma; io.resid_bf = u; io.ldr = D; io.out_bf = z; io.ldo = D;
          if (last) io.out_f32 = s->enc_x; gemm(st, ds.pw2, io); }
    }
    if (s->down.empty()) bf16_rows_to_f32_kernel<<<gridfor(T * D), 256, 0, st>>>(z, s->enc_x, T * D);
    if (T != Tf) return ft_fail(ctx, FT_ERR_STATE, "ft_codec_encode: stage rates do not multiply to the frame length");
    // pre_module (window-limited transformer), then the residual quantiser search
    run_transformer(ctx, st, s->pre, s->pre_norm, s->enc_x, (int)T, D, H, hd, c.tf_ffn, c.tf_window, s->rope,
                    s->e_xn, s->e_qkv, s->e_y, s->e_g, nullptr, s->enc_zq);
    FT_TRY(rvq_search(ctx, st, s->enc_zq, (int)T, s->enc_codes));
    FT_HIP(ctx, hipMemcpyAsync(codes, s->enc_codes, (size_t)R * T * sizeof(int), hipMemcpyDeviceToHost, st));
    FT_HIP(ctx, hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ft_fail(ctx, FT_ERR_HIP, std::string("codec encode launch: ") + hipGetErrorString(e));
    *out_frames = (int32_t)T;
    return FT_OK;
}

extern "C" int32_t ft_codec_enc_frame_len(const ft_ctx* ctx) { return ctx && ctx->codec ? ctx->codec->enc_frame_len : 0; }

extern "C" int32_t ft_codec_frame_len(const ft_ctx* ctx) { return ctx && ctx->codec ? ctx->codec->frame_len : 0; }
