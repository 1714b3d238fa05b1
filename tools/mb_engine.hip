// Timing harness for the persistent slow-stack engine (csrc/frame_engine.h) on synthetic weights at the s1-mini widths:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Ifish-tts_amd/csrc tools/mb_engine.hip -o tools/bin/mb_engine
//   tools/bin/mb_engine [n_layer=28] [pos=150] [nsplit=8]
// Prints the kernel time per launch and, from the in-kernel s_memrealtime stamps of workgroup 0 and of one attention
// workgroup, where a layer's time goes.  Correctness is NOT checked here (tests/test_engine_gpu.py does that).
#include "frame_engine.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ft;

__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = f32_to_bf16_bits(((float)(h & 0xffff) / 65536.0f - 0.5f) * scale);
    }
}
__global__ void fill_rope(float* r, int n_pos, int hp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_pos * hp) { const int pos = i / hp, k = i % hp; const float a = pos * powf(1e6f, -(float)k / hp); r[2 * i] = round_bf16(cosf(a)); r[2 * i + 1] = round_bf16(sinf(a)); }
}

static int run_fast(int nb, hipStream_t s);

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 0) {     // mb_engine 0  -> the fast codebook loop
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return run_fast(prop.multiProcessorCount, s);
    }
    const int n_layer = argc > 1 ? atoi(argv[1]) : 28;
    const int pos0 = argc > 2 ? atoi(argv[2]) : 150;
    const int nsplit = argc > 3 ? atoi(argv[3]) : 8;
    const int D = 1024, H = 16, Hkv = 8, hd = 128, F = 3072, HD = H * hd, qkvN = (H + 2 * Hkv) * hd, n_slots = 1024, V = 8192, ncb = 10, cbs = 4096;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    printf("device %s, %d CUs; %d layers, pos %d, nsplit %d\n", prop.name, nb, n_layer, pos0, nsplit);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto balloc = [&](size_t n, unsigned seed, float scale) { bf16_t* p; CK(hipMalloc(&p, n * 2 + 64)); fill_bf16<<<1024, 256, 0, s>>>(p, n, seed, scale); return p; };
    std::vector<EngLayer> hl(n_layer);
    for (int i = 0; i < n_layer; ++i) {
        EngLayer& l = hl[i];
        l.wqkv = balloc((size_t)qkvN * D, 11 * i + 1, 0.06f); l.bqkv = nullptr; l.attn_norm = balloc(D, 11 * i + 2, 2.0f);
        l.qn = balloc(hd, 11 * i + 3, 2.0f); l.kn = balloc(hd, 11 * i + 4, 2.0f); l.wo = balloc((size_t)D * HD, 11 * i + 5, 0.04f); l.bo = nullptr;
        l.ffn_norm = balloc(D, 11 * i + 6, 2.0f); l.w13 = balloc((size_t)2 * F * D, 11 * i + 7, 0.06f); l.w2 = balloc((size_t)D * F, 11 * i + 8, 0.04f);
        l.kc = balloc((size_t)Hkv * n_slots * hd, 11 * i + 9, 1.0f); l.vc = balloc((size_t)Hkv * n_slots * hd, 11 * i + 10, 1.0f);
    }
    EngLayer* dl; CK(hipMalloc(&dl, hl.size() * sizeof(EngLayer))); CK(hipMemcpy(dl, hl.data(), hl.size() * sizeof(EngLayer), hipMemcpyHostToDevice));
    SlowEngP p{};
    p.layers = dl; p.n_layer = n_layer; p.D = D; p.H = H; p.Hkv = Hkv; p.hd = hd; p.F = F; p.qkvN = qkvN; p.eps = 1e-6f; p.scale = 1.0f / sqrtf((float)hd);
    p.emb = balloc((size_t)V * D, 901, 1.0f); p.cb_emb = balloc((size_t)ncb * cbs * D, 902, 1.0f);
    int htok[11] = {5000, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10};
    int* dtok; CK(hipMalloc(&dtok, sizeof htok)); CK(hipMemcpy(dtok, htok, sizeof htok, hipMemcpyHostToDevice));
    p.toks = dtok; p.tok_row_stride = 1; p.ncb = ncb; p.cbsize = cbs; p.vocab = V; p.sem_begin = 4000; p.sem_end = 8095; p.scale_cb = 1; p.inv_div = sqrtf(11.f);
    float* rope; CK(hipMalloc(&rope, (size_t)n_slots * hd * 4)); fill_rope<<<(n_slots * hd / 2 + 255) / 256, 256, 0, s>>>(rope, n_slots, hd / 2);
    p.rope = rope;
    int* dpos; CK(hipMalloc(&dpos, 4)); CK(hipMemcpy(dpos, &pos0, 4, hipMemcpyHostToDevice));
    p.pos = dpos; p.pos_off = 0; p.n_slots = n_slots; p.nsplit = nsplit; p.cache_off = 0;
    auto zalloc = [&](size_t bytes) { void* q; CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0, bytes)); return q; };
    // every hand-off buffer out of ONE large allocation (one TLB fragment instead of six small mappings)
    const int relay = getenv("NO_RELAY") ? 0 : 1;
    char* pool = (char*)zalloc((size_t)128 << 20);
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* q = pool + off; off += (bytes + 4095) & ~(size_t)4095; return q; };
    const size_t VB = (size_t)nb * ENG_LINE * 4;   // bytes per padded vector buffer
    p.gx = (unsigned*)carve((size_t)(n_layer + 1) * VB); p.gqkv = (unsigned*)carve((size_t)n_layer * VB);
    p.gy = (unsigned*)carve((size_t)n_layer * HD * 4); p.gxb = (unsigned*)carve((size_t)n_layer * VB); p.gg = (unsigned*)carve((size_t)n_layer * VB);
    const size_t pool_words = off / 4;      // [gx | gqkv | gy | gxb | gg]: replicated per XCD
    off = pool_words * 4 * 9;
    p.rep_delta0 = relay ? (long)pool_words : 0; p.rep_stride = relay ? (long)pool_words : 0;
    p.gpart = (unsigned long long*)carve((size_t)n_layer * H * 32 * (hd + 2) * 8);
    p.ctl = (unsigned*)carve(ENG_CTL_WORDS * 4);
    float* xo; CK(hipMalloc(&xo, D * 4)); p.x_out = xo; p.nt = 1;
    unsigned long long* stamps = (unsigned long long*)zalloc((size_t)nb * n_layer * 16 * 8 * 2);
    const size_t lds = 82 * 1024;
    const bool xl = getenv("XL") != nullptr;     // the XCD-local form (needs nsplit = 32 on 8 x 32 CUs)
    CK(hipFuncSetAttribute((const void*)slow_engine_kernel<EngSlowS1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)slow_engine_kernel<EngSlowS1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("XCD-local attention: %d\n", (int)xl);
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int stamp_mode = argc > 4 ? atoi(argv[4]) : 1;    // which mode's stamps are analysed: 1 = with weights, 2 = without
    for (int mode = 0; mode < 3; ++mode) {
        p.stamps = mode == stamp_mode ? stamps : nullptr;
        p.nt = mode == 2 ? 3 : 1;
        float best = 1e9f, sum = 0.f; const int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipEventRecord(e0, s));
            if (xl) slow_engine_kernel<EngSlowS1, true><<<nb, ENG_THREADS, lds, s>>>(p);
            else slow_engine_kernel<EngSlowS1, false><<<nb, ENG_THREADS, lds, s>>>(p);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) { best = std::min(best, ms); sum += ms; }
        }
        unsigned h[4]; CK(hipMemcpy(h, p.ctl, 16, hipMemcpyDeviceToHost));
        printf("%-22s: %.1f us per launch (best %.1f) = %.2f us per layer%s\n", mode == 0 ? "plain" : mode == 1 ? "with stamps" : "NO weight loads", sum / reps * 1e3, best * 1e3,
               sum / reps * 1e3 / n_layer, h[ENG_CTL_ABORT] ? "  ABORTED" : "");
        if (h[ENG_CTL_ABORT]) { printf("abort at phase %u\n", h[ENG_CTL_WHERE]); return 1; }
    }
    std::vector<unsigned long long> st((size_t)nb * n_layer * 16 * 2);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char* names[13] = {"x gathered", "QKV rows done", "attention section left", "y gathered", "Wo rows done", "x' gathered", "W13 rows done", "g gathered", "W2 rows done", "W2 re-issued (mode 2: Wo granule stores acknowledged)", "gw0: starts polling x'", "gw0: x' pieces seen", "gw0: past barrier B3"};
    for (int b : {0, 100, nb - 1}) {
        double seg[9] = {0};
        int cnt = 0;
        for (int li = 2; li < n_layer; ++li) {
            const unsigned long long* a = &st[((size_t)b * n_layer + li) * 16];
            const unsigned long long prev = st[((size_t)b * n_layer + li - 1) * 16 + 8];
            seg[0] += (double)(a[0] - prev);
            for (int k = 1; k < 9; ++k) seg[k] += (double)(a[k] - a[k - 1]);
            ++cnt;
        }
        if (!cnt) continue;
        printf("workgroup %3d, average over layers 2.. (us since the previous stamp):\n", b);
        double tot = 0;
        for (int k = 0; k < 9; ++k) { printf("   %-24s %6.2f\n", names[k], seg[k] / cnt / 100.0); tot += seg[k] / cnt / 100.0; }
        printf("   %-24s %6.2f\n", "layer", tot);
    }
    // chip-wide view (s_memrealtime is one clock for all CUs): per stamp, the earliest / median / latest workgroup,
    // relative to the moment the LAST workgroup finished the previous layer's W2
    {
        double lo[13] = {0}, md[13] = {0}, hi[13] = {0};
        int cnt = 0;
        std::vector<unsigned long long> v(nb);
        for (int li = 4; li < n_layer; ++li) {
            unsigned long long t0 = 0;
            for (int b = 0; b < nb; ++b) t0 = std::max(t0, st[((size_t)b * n_layer + li - 1) * 16 + 8]);
            for (int k = 0; k < 13; ++k) {
                for (int b = 0; b < nb; ++b) v[b] = st[((size_t)b * n_layer + li) * 16 + k];
                std::sort(v.begin(), v.end());
                lo[k] += (double)v[0] - (double)t0; md[k] += (double)v[nb / 2] - (double)t0; hi[k] += (double)v[nb - 1] - (double)t0;
            }
            ++cnt;
        }
        printf("chip-wide, us after the last workgroup finished the previous layer (earliest / median / latest workgroup):\n");
        for (int k = 0; k < 13; ++k) printf("   %-24s %7.2f %7.2f %7.2f\n", names[k], lo[k] / cnt / 100.0, md[k] / cnt / 100.0, hi[k] / cnt / 100.0);
    }
    {   // the attention turn seen by the workgroups that hold a (kv head, split) role in the layer
        const char* an[9] = {"turn starts (polls q k v)", "q k v gathered (BA)", "norm + rotation done (BB)", "scores + values done (BC)", "partials published",
                             "merger: last partials seen", "merger: y published", "  (position walk done)", "  (slot results in LDS)"};
        double lo[9] = {0}, md[9] = {0}, hi[9] = {0};
        int cnt = 0;
        const size_t off = (size_t)nb * n_layer * 16;
        for (int li = 4; li < n_layer; ++li) {
            unsigned long long t0 = 0;
            for (int b = 0; b < nb; ++b) t0 = std::max(t0, st[((size_t)b * n_layer + li - 1) * 16 + 8]);
            std::vector<unsigned long long> v;
            for (int k = 0; k < 9; ++k) {
                v.clear();
                for (int b = 0; b < nb; ++b) { const unsigned long long x = st[off + ((size_t)b * n_layer + li) * 16 + k]; if (x) v.push_back(x); }
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                lo[k] += (double)v[0] - (double)t0; md[k] += (double)v[v.size() / 2] - (double)t0; hi[k] += (double)v.back() - (double)t0;
            }
            ++cnt;
        }
        printf("attention turn (role workgroups only), same reading:\n");
        for (int k = 0; k < 9; ++k) printf("   %-28s %7.2f %7.2f %7.2f\n", an[k], lo[k] / cnt / 100.0, md[k] / cnt / 100.0, hi[k] / cnt / 100.0);
    }
    printf("shader clock during the launch: %.2f GHz\n", (double)st[14] / (double)st[15] * 0.1);
    {   // the x' gather of gw0: first-pass round trip, passes needed
        double rtt = 0, nf = 0, nl = 0; int cnt = 0;
        for (int li = 4; li < n_layer; ++li) for (int b = 0; b < nb; ++b) {
            const unsigned long long* a = &st[((size_t)b * n_layer + li) * 16];
            if (b == 0) continue; rtt += (double)(a[13] - a[10]); nf += (double)a[14]; nl += (double)a[15]; ++cnt;
        }
        printf("x' gather by gw0: first full pass returned %.2f us after polling began; %.2f full passes, %.2f single-piece polls per gather\n",
               rtt / cnt / 100.0, nf / cnt, nl / cnt);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ the fast codebook loop
static int run_fast(int nb, hipStream_t s) {
    const int nL = 4, ncb = 10, D = 1024, H = 16, Hkv = 8, hd = 64, F = 3072, HD = H * hd, qkvN = (H + 2 * Hkv) * hd, V = 1024, cbs = 4096;
    auto balloc = [&](size_t n, unsigned seed, float scale) { bf16_t* p; CK(hipMalloc(&p, n * 2 + 64)); fill_bf16<<<1024, 256, 0, s>>>(p, n, seed, scale); return p; };
    auto zalloc = [&](size_t bytes) { void* q; CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0, bytes)); return q; };
    std::vector<EngLayer> hl(nL);
    for (int i = 0; i < nL; ++i) {
        EngLayer& l = hl[i];
        l.wqkv = balloc((size_t)qkvN * D, 11 * i + 1, 0.06f); l.bqkv = nullptr; l.attn_norm = balloc(D, 11 * i + 2, 2.0f);
        l.qn = nullptr; l.kn = nullptr; l.wo = balloc((size_t)D * HD, 11 * i + 5, 0.04f); l.bo = nullptr;
        l.ffn_norm = balloc(D, 11 * i + 6, 2.0f); l.w13 = balloc((size_t)2 * F * D, 11 * i + 7, 0.06f); l.w2 = balloc((size_t)D * F, 11 * i + 8, 0.04f);
        l.kc = nullptr; l.vc = nullptr;
    }
    EngLayer* dl; CK(hipMalloc(&dl, hl.size() * sizeof(EngLayer))); CK(hipMemcpy(dl, hl.data(), hl.size() * sizeof(EngLayer), hipMemcpyHostToDevice));
    FastEngP p{};
    p.layers = dl; p.n_layer = nL; p.ncb = ncb; p.D = D; p.H = H; p.Hkv = Hkv; p.hd = hd; p.F = F; p.qkvN = qkvN; p.V = V;
    p.eps = 1e-6f; p.scale = 1.0f / sqrtf((float)hd);
    float* rope; CK(hipMalloc(&rope, (size_t)ncb * hd * 4)); fill_rope<<<(ncb * hd / 2 + 255) / 256, 256, 0, s>>>(rope, ncb, hd / 2);
    p.rope = rope; p.fast_norm = balloc(D, 801, 2.0f); p.fast_out = balloc((size_t)V * D, 802, 0.06f); p.fast_emb = balloc((size_t)cbs * D, 803, 1.0f);
    float* hid = (float*)zalloc(D * 4); float* femb = (float*)zalloc(D * 4);
    { std::vector<float> h(D); for (int i = 0; i < D; ++i) h[i] = (float)((i * 37) % 17 - 8) * 0.0625f; CK(hipMemcpy(hid, h.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(femb, h.data(), D * 4, hipMemcpyHostToDevice)); }
    p.hid = hid; p.femb = femb;
    const size_t VW = (size_t)nb * ENG_LINE;
    const size_t words = (2 * ((size_t)(nL + 1) + 3 * nL + 1)) * VW + (size_t)ncb * ENG_LINE;
    const int relay = getenv("NO_RELAY") ? 0 : 1;
    unsigned* g = (unsigned*)zalloc(words * 4 * 9);
    p.rep_delta0 = relay ? (long)words : 0; p.rep_stride = relay ? (long)words : 0;
    p.gx = g; g += 2 * (size_t)(nL + 1) * VW; p.gqkv = g; g += 2 * (size_t)nL * VW; p.gxb = g; g += 2 * (size_t)nL * VW; p.gg = g; g += 2 * (size_t)nL * VW;
    p.glog = g; g += 2 * VW; p.gcode = g;
    p.ctl = (unsigned*)zalloc(ENG_CTL_WORDS * 4);
    const int R = ncb + 1, cap = 256;
    RowCtl hc{0.7f, 0.8f, 1.1f, 0, 1234ull};
    RowCtl* dctl; CK(hipMalloc(&dctl, sizeof hc)); CK(hipMemcpy(dctl, &hc, sizeof hc, hipMemcpyHostToDevice));
    SampP sp{};
    sp.V = V; sp.ldl = V; sp.ctl = dctl; sp.tokn = (int*)zalloc(R * 4); sp.seq = (int*)zalloc((size_t)R * cap * 4); sp.cap = cap;
    int one = 1; sp.nf = (int*)zalloc(4); CK(hipMemcpy(sp.nf, &one, 4, hipMemcpyHostToDevice));
    sp.ncb = ncb; sp.sem_begin = 4000; sp.im_end = 100; sp.cbsize = cbs; sp.noise = nullptr; sp.tok = (int*)zalloc(R * 4);
    sp.pos = (int*)zalloc(4); sp.done = (int*)zalloc(4);
    p.samp = sp; p.noise_cb_stride = V; p.noise_off1 = 8192;
    unsigned long long* stamps = (unsigned long long*)zalloc((size_t)nb * ncb * nL * 16 * 8);
    p.pair = getenv("NO_PAIR") ? 0 : 1;
    p.qkv0_tab = nullptr;
    if (!getenv("NO_QKV0")) {
        bf16_t* tab; CK(hipMalloc(&tab, (size_t)V * qkvN * 2));
        const EngLayer l0 = hl[0];
        eng_qkv0_table_kernel<EngFastS1><<<dim3(qkvN / (EngFastS1::SQ * ENG_CW), 16), ENG_CW * 64, ((size_t)D + ENG_MAX_OUT) * 4, s>>>(
            l0.wqkv, l0.bqkv, l0.attn_norm, p.fast_emb, tab, D, qkvN, V, p.eps);
        CK(hipGetLastError()); CK(hipStreamSynchronize(s));
        p.qkv0_tab = tab;
    }
    size_t ldsb = eng_fast_lds_bytes(D, qkvN, HD, F, V, nL, ncb, Hkv * hd, p.pair != 0);
    printf("paired first pass: %d, LDS %zu bytes\n", p.pair, ldsb);
    CK(hipFuncSetAttribute((const void*)fast_engine_kernel<EngFastS1, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        p.stamps = mode ? stamps : nullptr;
        float sum = 0.f; const int reps = 10;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemcpyAsync(sp.nf, &one, 4, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(e0, s));
            fast_engine_kernel<EngFastS1, 10><<<nb, ENG_THREADS, ldsb, s>>>(p);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) sum += ms;
        }
        unsigned h[4]; CK(hipMemcpy(h, p.ctl, 16, hipMemcpyDeviceToHost));
        printf("fast loop %-12s: %.1f us per launch = %.2f us per layer-step%s\n", mode ? "with stamps" : "plain", sum / reps * 1e3, sum / reps * 1e3 / (nL * ncb),
               h[ENG_CTL_ABORT] ? "  ABORTED" : "");
        if (h[ENG_CTL_ABORT]) { printf("abort at phase %u\n", h[ENG_CTL_WHERE]); return 1; }
    }
    std::vector<unsigned long long> st((size_t)nb * ncb * nL * 16);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char* names[10] = {"x ready (B1)", "QKV rows done", "qkv gathered (B1b)", "y ready (B2)", "Wo rows done", "x' gathered (B3)", "W13 rows done", "g gathered (B4)",
                             "W2 rows done", "own attention done"};
    const int order[10] = {0, 1, 2, 9, 3, 4, 5, 6, 7, 8};
    double lo[10] = {0}, md[10] = {0}, hi[10] = {0};
    int cnt = 0;
    std::vector<unsigned long long> v(nb);
    for (int cb = 2; cb < ncb; ++cb) for (int li = 1; li < nL; ++li) {
        unsigned long long t0 = 0;
        for (int b = 0; b < nb; ++b) t0 = std::max(t0, st[(((size_t)b * ncb + cb) * nL + li - 1) * 16 + 8]);
        for (int k = 0; k < 10; ++k) {
            for (int b = 0; b < nb; ++b) v[b] = st[(((size_t)b * ncb + cb) * nL + li) * 16 + k];
            std::sort(v.begin(), v.end());
            lo[k] += (double)v[0] - (double)t0; md[k] += (double)v[nb / 2] - (double)t0; hi[k] += (double)v[nb - 1] - (double)t0;
        }
        ++cnt;
    }
    printf("chip-wide (steps 2.., layers 1..), us after the last workgroup finished the previous layer's W2 rows (earliest / median / latest):\n");
    for (int kk = 0; kk < 10; ++kk) { const int k = order[kk]; printf("   %-22s %7.2f %7.2f %7.2f\n", names[k], lo[k] / cnt / 100.0, md[k] / cnt / 100.0, hi[k] / cnt / 100.0); }
    if (p.pair) {
        double plo[10] = {0}, pmd[10] = {0}, phi[10] = {0};
        int pc = 0;
        for (int li = 1; li < nL - 1; ++li) {       // the last layer carries one row after the attention
            unsigned long long t0 = 0;
            for (int b = 0; b < nb; ++b) t0 = std::max(t0, st[(((size_t)b * ncb + 1) * nL + li - 1) * 16 + 8]);
            for (int k = 0; k < 10; ++k) {
                for (int b = 0; b < nb; ++b) v[b] = st[(((size_t)b * ncb + 1) * nL + li) * 16 + k];
                std::sort(v.begin(), v.end());
                plo[k] += (double)v[0] - (double)t0; pmd[k] += (double)v[nb / 2] - (double)t0; phi[k] += (double)v[nb - 1] - (double)t0;
            }
            ++pc;
        }
        printf("paired first pass (positions 0 and 1 as two rows), layers 1..%d, same reading:\n", nL - 2);
        for (int kk = 0; kk < 10; ++kk) { const int k = order[kk]; printf("   %-22s %7.2f %7.2f %7.2f\n", names[k], plo[k] / pc / 100.0, pmd[k] / pc / 100.0, phi[k] / pc / 100.0); }
    }
    // per step: time from the last layer's W2 to the next step's first B1 (head + draw + code hand-off)
    double gap = 0; int gc = 0;
    for (int cb = 2; cb < ncb; ++cb) {
        unsigned long long t0 = 0, t1 = ~0ull;
        for (int b = 0; b < nb; ++b) { t0 = std::max(t0, st[(((size_t)b * ncb + cb - 1) * nL + nL - 1) * 16 + 8]); t1 = std::min(t1, st[(((size_t)b * ncb + cb) * nL + 0) * 16 + 0]); }
        gap += (double)t1 - (double)t0; ++gc;
    }
    {   // span of the stamped part of the launch: first B1 of the first pass .. last draw of the last step
        unsigned long long first = ~0ull, last = 0;
        const int cb0 = p.pair ? 1 : 0;
        for (int b = 0; b < nb; ++b) {
            first = std::min(first, st[(((size_t)b * ncb + cb0) * nL + 0) * 16 + 0]);
            last = std::max(last, st[(((size_t)b * ncb + ncb - 1) * nL + nL - 1) * 16 + 14]);
        }
        double pass0 = 0;   // first pass: first B1 -> last W2 of its last layer
        { unsigned long long e = 0; for (int b = 0; b < nb; ++b) e = std::max(e, st[(((size_t)b * ncb + cb0) * nL + nL - 1) * 16 + 8]); pass0 = (double)e - (double)first; }
        printf("first B1 of the launch -> last draw: %.1f us (kernel: see above); first pass (layers only): %.1f us\n",
               ((double)last - (double)first) / 100.0, pass0 / 100.0);
    }
    printf("head + draw + code hand-off between two steps: %.2f us\n", gap / gc / 100.0);
    {   // inside that gap (gathering wave 0): chip-wide medians relative to the last workgroup's W2 of the step
        const char* dn[5] = {"head input gathered (B5)", "logits gathered", "argmax known", "top-p cut known", "code drawn"};
        const int dk[5] = {10, 11, 12, 13, 14};
        double md[5] = {0}, hi2[5] = {0}; int dc = 0;
        for (int cb = 2; cb < ncb; ++cb) {
            unsigned long long t0 = 0;
            for (int b = 0; b < nb; ++b) t0 = std::max(t0, st[(((size_t)b * ncb + cb) * nL + nL - 1) * 16 + 8]);
            for (int k = 0; k < 5; ++k) {
                for (int b = 0; b < nb; ++b) v[b] = st[(((size_t)b * ncb + cb) * nL + nL - 1) * 16 + dk[k]];
                std::sort(v.begin(), v.end());
                md[k] += (double)v[nb / 2] - (double)t0; hi2[k] += (double)v[nb - 1] - (double)t0;
            }
            ++dc;
        }
        for (int k = 0; k < 5; ++k) printf("   %-26s %7.2f %7.2f   (median / latest workgroup)\n", dn[k], md[k] / dc / 100.0, hi2[k] / dc / 100.0);
        {   // finer stamps of the cut search (kept in the previous layer's stamp row)
            const char* fn[5] = {"  normaliser Z known", "  probabilities in LDS", "  wave 0: total mass", "  wave 0: bit search done", "  wave 0: cut found"};
            double fm[5] = {0}; int fc = 0;
            for (int cb = 2; cb < ncb; ++cb) {
                unsigned long long t0 = 0;
                for (int b = 0; b < nb; ++b) t0 = std::max(t0, st[(((size_t)b * ncb + cb) * nL + nL - 1) * 16 + 8]);
                for (int k = 0; k < 5; ++k) {
                    for (int b = 0; b < nb; ++b) v[b] = st[(((size_t)b * ncb + cb) * nL + nL - 2) * 16 + 10 + k];
                    std::sort(v.begin(), v.end());
                    fm[k] += (double)v[nb / 2] - (double)t0;
                }
                ++fc;
            }
            for (int k = 0; k < 5; ++k) printf("   %-26s %7.2f\n", fn[k], fm[k] / fc / 100.0);
        }
    }
    return 0;
}
