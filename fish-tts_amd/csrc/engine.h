// Host-side state of libfishtts_hip.so (one ctx per GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/fishtts_hip.h"
#include "../../include/fishtts_hip_test.h"
#include "common.h"

namespace ft {
struct RowCtl;
struct SampCut;
struct EngLayer;
}  // namespace ft (kernels live in ar_kernels.h, included by engine.hip only)

struct FtTensor {
    void* p = nullptr;
    std::vector<int64_t> shape;
    int dtype = FT_BF16;  // storage type in HBM (FT_F32 | FT_BF16)
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct FtLayer {
    void *attn_norm = nullptr, *wqkv = nullptr, *bqkv = nullptr, *qn = nullptr, *kn = nullptr, *wo = nullptr,
         *bo = nullptr, *ffn_norm = nullptr, *w13 = nullptr, *w2 = nullptr;
    void *kc = nullptr, *vc = nullptr;
    float *bqkv_f32 = nullptr, *bo_f32 = nullptr;  // f32 copies of the biases for the MFMA prefill epilogues
};

struct CodecState;  // codec.hip

struct ft_ctx {
    ft_ar_config c{};
    bool has_ar = false;
    ft_codec_config cc{};
    bool has_codec = false;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool finalized = false;

    std::map<std::string, FtTensor> expected;  // name -> expected shape (p filled once loaded)
    size_t esz = 2;

    // resolved AR weights
    void *emb = nullptr, *cb_emb = nullptr, *norm = nullptr, *head = nullptr, *fproj_w = nullptr,
         *fproj_b = nullptr, *fast_emb = nullptr, *fast_norm = nullptr, *fast_out = nullptr;
    float *rope = nullptr, *frope = nullptr;
    std::vector<FtLayer> layers, flayers;

    // activations [max_batch][...]
    float *x = nullptr, *qkv = nullptr, *y = nullptr, *g = nullptr, *logits = nullptr, *hid = nullptr,
          *femb = nullptr, *xf = nullptr, *qkvf = nullptr, *gf = nullptr, *flog = nullptr, *part_o = nullptr,
          *part_ml = nullptr;
    int n_slots = 0, nsplit = 1, cap = 0, fastV = 0, y_ld = 0;
    // KV splits of the decode attention: nsplit is what the next enqueue uses; it follows the context length
    // (8 up to 768 positions, 16 up to 3072, 32 beyond) unless FT_ATTN_NSPLIT pins it; buffers hold nsplit_max
    int nsplit_max = 1;
    bool nsplit_fixed = false;
    // MFMA prefill workspace (bf16 precision): S = max_seq_len rows
    float *pf_x = nullptr, *pf_qkv = nullptr, *pf_y = nullptr;
    ft::bf16_t *pf_xn = nullptr, *pf_ybf = nullptr, *pf_g = nullptr, *pf_qbf = nullptr;
    // ragged prompt pass of several slots (ft_ar_prefill_slow_many): per sequence {first row, rows, first position, slot},
    // per row (slot, position)
    int4* pf_seqs = nullptr;
    int2* pf_rows = nullptr;
    // lock-step batches of >= wide_min utterances run every Linear as one MFMA launch with the row operations folded in
    // (wide_kernels.h); their activations are octet-major bf16 Xo[width / 8][xo_ldm][8]: residual streams of the slow and
    // the fast stack, the drawn codes' embeddings, the attention output, the SwiGLU vector
    ft::bf16_t *xo_x = nullptr, *xo_xf = nullptr, *xo_femb = nullptr, *xo_y = nullptr, *xo_g = nullptr;
    ft::bf16_t* wide_qkv0_tab = nullptr;   // [fastV][fast qkv width]: layer 0's q k v of the codebook steps >= 2 by drawn code (wide batches)
    int xo_ldm = 0;     // rows of an octet: 2 x xo_pair
    int xo_pair = 0;    // max_batch rounded up to the 16-row MFMA tile; rows from here on hold codebook position 1 of the paired pass
    bool no_attn_wide = false;   // FT_NO_ATTN_WIDE: wide batches keep the online-softmax attention kernel of the single rows
    bool no_head_stream = false; // FT_NO_HEAD_STREAM: the vocabulary head of a wide batch on the general wide launch
    bool no_pair = false;   // FT_NO_PAIR: codebook positions 0 and 1 as two passes (the comparison a parity test makes)
    int wide_min = 5;   // measured: the MFMA path wins from 5 rows (B=5: 2.79 vs 3.33 ms per frame), B <= 4 keeps the bit-exact multi-row GEMV
    bool wide_ok = false;
    bool prefill_v0 = false;
    int prefill_gemm_mode = 2;
    size_t cache_m_stride = 0, fcache_m_stride = 0;

    // per-slot state
    int *d_pos = nullptr, *d_tok = nullptr, *d_tokn = nullptr, *d_seq = nullptr, *d_nf = nullptr,
        *d_done = nullptr, *d_prompt = nullptr;
    ft::RowCtl* d_ctl = nullptr;
    int *h_pin = nullptr;  // pinned scratch (2*max_batch + 4 ints)
    float* noise = nullptr;
    long noise_rows = 0, noise_row_len = 0;

    // large-vocabulary sampler scratch
    ft::SampCut* samp_cut = nullptr;
    int* samp_chunk_cnt = nullptr;
    float* samp_part_score = nullptr;
    int* samp_part_idx = nullptr;

    // persistent frame engine (frame_engine.h): batch-1 decode frames as two launches of one workgroup per CU
    bool eng_on = false;          // shapes fit the instantiated engine and FT_NO_ENGINE is unset
    bool eng_fast_on = false;
    int eng_nb = 0;               // workgroups = CUs of the device
    ft::EngLayer *eng_layers = nullptr, *eng_flayers = nullptr;   // device tables
    unsigned *eng_gx = nullptr, *eng_gqkv = nullptr, *eng_gy = nullptr, *eng_gxb = nullptr, *eng_gg = nullptr;
    unsigned long long* eng_gpart = nullptr;
    unsigned* eng_fast_g = nullptr;   // granule buffers of the fast stack (one allocation)
    unsigned* eng_ctl = nullptr;
    size_t eng_lds_slow = 0, eng_lds_fast = 0, eng_fast_words = 0, eng_pool_words = 0;
    bool eng_relay = true;        // per-XCD replicas of the hand-off buffers (FT_NO_RELAY: every workgroup polls the source)
    void* eng_qkv0_tab = nullptr;   // fast layer 0's q k v per codebook-embedding row (bf16 [fastV][qkvN])
    bool eng_pair = false;          // fast loop: positions 0 and 1 as two rows of the first pass
    bool xl_shape = false;          // 8 kv heads x (2 x 128) on 8 XCDs x 32 CUs: 32 KV splits always (engine.hip: ar_alloc)
    bool eng_xl = false;            // the slow stack's engine runs its XCD-local form (one kv head per XCD)
    std::string path_str;                // what ft_ar_frame_path last returned
    const void* eng_slow_fn = nullptr;   // the instantiations of the two engine kernels this model's widths match (engine.hip: eng_shapes)
    const void* eng_fast_fn = nullptr;
    size_t eng_pool_bytes = 0, eng_gpart_bytes = 0, eng_fast_bytes = 0;   // hand-off allocations (zeroed again after an abort)
    // A hand-off that timed out (ENG_CTL_ABORT) is survivable: the host clears the control words and the hand-off pools,
    // redoes the affected frames on the launch path (eng_suspended) and turns the engine off for this context after
    // ENG_MAX_STRIKES such events.  eng_why says, in words, which frame path this context takes and why.
    bool eng_suspended = false;
    int eng_strikes = 0, eng_last_where = 0;
    bool eng_owner = false;         // this context holds its device's engine slot (one engine context per device and process)
    std::string eng_why;

    std::map<int, hipGraphExec_t> graphs;
    std::map<int, int> graph_nodes;   // nodes of each captured frame graph (launches per frame)

    // measurement hook
    bool prof = false, prof_count_only = false;
    std::vector<hipEvent_t> prof_ev;
    int64_t prof_bytes = 0, prof_launches = 0;

    CodecState* codec = nullptr;
};

ft_status ft_fail(ft_ctx* ctx, ft_status code, const std::string& msg);
#define FT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ft_fail(ctx, FT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

// codec.hip
ft_status codec_create(ft_ctx* ctx);
void codec_destroy(ft_ctx* ctx);
void codec_expected(ft_ctx* ctx);
void ft_expect(ft_ctx* ctx, const std::string& name, std::vector<int64_t> shape, int dtype);
ft_status codec_finalize(ft_ctx* ctx);
