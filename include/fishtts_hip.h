/* libfishtts_hip.so — C ABI of the MI355X (gfx950) hot path: dual-AR semantic-token decode +
 * DAC codec decode (and encode, for encode_reference).  Plain pointers and sizes only; no torch types.
 *
 * The reference (smolGura/fish-tts) is pure Python and has no FFI; the seams this library
 * sits behind are the Python-level operators named in SURVEY.md §8(b).  Each entry point
 * cites the reference interface it replaces (paths relative to /root/reference).
 *
 * Threading: AR calls on one ctx are serialized by the caller (as the reference's
 * synthesize() is not re-entrant: synthesizer.py:431-481 shares one KV cache);
 * ft_codec_decode is re-entrant with respect to AR calls and runs on its own HIP stream
 * (mirrors the decoder thread of synthesizer.py:513-528).  One ctx per GPU.
 *
 * Ownership: the caller owns every buffer it passes; the library copies weights into its
 * own HBM allocations and never frees caller memory.  Errors: integer status + a message
 * from ft_last_error(); the Python host maps them to the reference's ValueError/RuntimeError.
 */
#ifndef FISHTTS_HIP_H
#define FISHTTS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ft_ctx ft_ctx;
typedef int32_t ft_status;

enum { FT_OK = 0, FT_ERR_ARG = 1, FT_ERR_HIP = 2, FT_ERR_STATE = 3, FT_ERR_UNSUPPORTED = 4,
       FT_ERR_NOMEM = 5, FT_ERR_TOO_LONG = 6, FT_ERR_MISSING_WEIGHT = 7 };
enum { FT_F32 = 0, FT_BF16 = 1, FT_F16 = 2 };   /* FT_F16: the AR model only (precision="fp16", synthesizer.py:125-126) */

/* Field names and meaning = config.json / DualARModelArgs (fish_tts/models/llama.py:31-123);
 * the three token ids come from the tokenizer layout (fish_tts/models/tokenizer.py:83-101). */
typedef struct ft_ar_config {
    int32_t dtype;  /* FT_BF16 | FT_F16 | FT_F32: model precision (fish_tts/synthesizer.py:122-128) */
    int32_t vocab_size, n_layer, n_head, dim, intermediate_size, n_local_heads, head_dim;
    float rope_base, norm_eps;
    int32_t max_seq_len, tie_word_embeddings, attention_qkv_bias, attention_o_bias, attention_qk_norm;
    int32_t codebook_size, num_codebooks, scale_codebook_embeddings;
    int32_t n_fast_layer, fast_dim, fast_n_head, fast_n_local_heads, fast_head_dim,
        fast_intermediate_size, fast_attention_qkv_bias, fast_attention_qk_norm, fast_attention_o_bias;
    int32_t semantic_begin_id, semantic_end_id, im_end_id;
    int32_t max_batch;      /* utterance slots decoded in lock step (reference: 1, inference.py:313-317) */
    int32_t max_new_tokens; /* per-slot capacity for generated frames */
} ft_ar_config;

/* Hyper-parameters of the DAC decode path; the reference hard-codes them in
 * fish_tts/synthesizer.py:199-269 (and fish_tts/models/vocoder.py:824-872). */
typedef struct ft_codec_config {
    int32_t dtype;                 /* FT_BF16 only: bf16 MFMA contractions, f32 accumulation, f32 transformer residual stream (anything else is FT_ERR_UNSUPPORTED) */
    int32_t n_codebooks;           /* residual codebooks (9); +1 semantic */
    int32_t codebook_size, semantic_codebook_size, codebook_dim, latent_dim; /* 1024, 4096, 8, 1024 */
    int32_t n_tf_layer, tf_n_head, tf_head_dim, tf_ffn, tf_window;           /* 8, 16, 64, 3072, 128 */
    float tf_rope_base, tf_norm_eps;                                          /* 1e4, 1e-5 */
    int32_t n_upsample;            /* 2 (x2 each) */
    int32_t decoder_dim;           /* 1536 */
    int32_t n_rates;               /* 4 */
    int32_t rates[8];              /* 8,8,4,2 */
    int32_t max_frames;            /* longest code sequence per call */
    int32_t max_batch;
    /* encode side (encode_reference: synthesizer.py:325-357, Encoder vocoder.py:498-575); encoder_dim = 0: decode only */
    int32_t encoder_dim;           /* 64 */
    int32_t n_enc_rates;           /* 4 */
    int32_t enc_rates[8];          /* 2,4,8,8 */
    int32_t enc_tf_layers[8];      /* 0,0,0,4: window-limited transformer layers after each block's strided conv */
    int32_t enc_tf_window;         /* 512 */
    int32_t max_enc_frames;        /* longest reference in code frames (a frame = prod(enc_rates)*4 samples) */
} ft_codec_config;

/* Sampling scalars of synthesize()/generate_long() (synthesizer.py:431-439, inference.py:741-765). */
typedef struct ft_sampling {
    float temperature, top_p, repetition_penalty;
    int32_t ban_eos;   /* 1: mask <|im_end|> before sampling (fixed-length synthetic benches only) */
    uint64_t seed;     /* counter-based RNG stream for the Exp(1) race (inference.py:24-27) */
} ft_sampling;

/* Lifecycle.  Replaces init_model() + setup_caches() (inference.py:387-414, llama.py:378-398,544-559)
 * and _load_vocoder() (synthesizer.py:188-293).  Either config may be NULL. */
ft_status ft_create(const ft_ar_config* ar, const ft_codec_config* codec, int32_t device, ft_ctx** out);
void ft_destroy(ft_ctx* ctx);
const char* ft_last_error(const ft_ctx* ctx); /* ctx may be NULL: last create-time error */

/* Weight ingestion under the reference's state-dict names (llama.py:349-359,510-535 after the
 * wq/wk/wv->wqkv fuse of llama.py:222-227; codec names as vocoder.py modules after weight-norm
 * folding).  `src` may be a host or a device pointer; `src_dtype` FT_F32|FT_BF16|FT_F16; the tensor is
 * converted to the ctx precision and repacked.  Replaces load_state_dict (llama.py:498). */
ft_status ft_load_weight(ft_ctx* ctx, const char* name, const void* src, int32_t src_dtype,
                         const int64_t* shape, int32_t ndim);
ft_status ft_finalize_weights(ft_ctx* ctx); /* checks completeness, builds fused layouts + tables */

/* AR path.  A "slot" is one utterance's state (KV cache, position, penalty window, RNG). */
ft_status ft_ar_reset(ft_ctx* ctx, int32_t slot);
/* Prefill = the un-compiled first decode_one_token_ar call of generate()/generate_streaming()
 * (inference.py:353-362, 709-718): prompt is (num_codebooks+1) x Lp int32 row-major, host memory
 * (ContentSequence.encode_for_inference output, inference.py:611-640).  Writes the first
 * generated frame (num_codebooks+1 int32) to out_frame (host).  No repetition penalty. */
ft_status ft_ar_prefill(ft_ctx* ctx, int32_t slot, const int32_t* prompt, int32_t Lp,
                        const ft_sampling* sp, int32_t* out_frame);
/* Several prompts at once (a batch scheduler's initial fill): ft_ar_prefill_slow runs the prompt pass of one slot
 * without its first frame (K/V + last hidden state stay on the device); ft_ar_first_frames then draws the first frames
 * of the contiguous slots [slot0, slot0+n) in one lock-step pass (head, semantic draw, fast codebooks) - the per-slot
 * equivalent is ft_ar_prefill_at.  next_pos[i] = pos0_i + Lp_i; out_frames: n x (num_codebooks+1) int32 (host). */
ft_status ft_ar_prefill_slow(ft_ctx* ctx, int32_t slot, const int32_t* prompt, int32_t Lp, int32_t pos0);
ft_status ft_ar_first_frames(ft_ctx* ctx, int32_t slot0, int32_t n, const ft_sampling* sp, const int32_t* next_pos,
                             int32_t* out_frames);
/* Prompt passes of n distinct slots in one call (a batch scheduler's initial fill and refills; the reference runs one
 * prompt pass per utterance, inference.py:353-362).  prompts: the n matrices back to back, matrix i = (num_codebooks+1) x
 * Lps[i] int32 row-major; slot i's prompt lands at cache positions [pos0s[i], pos0s[i] + Lps[i]).  From the width at
 * which lock-step batches run on the MFMA launches (5 prompts, bf16) all prompts go through the slow stack as the rows
 * of ONE pass (weights streamed once; K/V append and attention by sequence) - results follow the oracle within the bf16
 * evaluation-order margin, like the lock-step frames of that width; below it (and in fp16 / fp32) the call equals n calls
 * of ft_ar_prefill_slow bit for bit.  Follow with ft_ar_first_frames per contiguous run of slots. */
ft_status ft_ar_prefill_slow_many(ft_ctx* ctx, int32_t n, const int32_t* slots, const int32_t* prompts,
                                  const int32_t* Lps, const int32_t* pos0s);
/* Marks a slot idle for lock-step decoding: it counts as already finished (ft_ar_decode reports 0 frames for
 * it and it limits nothing); a later ft_ar_prefill[_at] on the slot re-activates it.  Slots that emit
 * <|im_end|> freeze the same way on the device, so a host scheduler can refill finished slots between
 * ft_ar_decode bursts (continuous batching; the reference serves one utterance at a time, synthesizer.py:431). */
ft_status ft_ar_park(ft_ctx* ctx, int32_t slot);
/* Reference-prefix KV reuse (the reference keeps the reference tensors in
 * `_prefill_cache` but re-prefills them on every call: synthesizer.py:363-429, inference.py:779-793,
 * 353-362).  The prompt prefix [<|interleave|>, (<|speaker:0|>, ref text, ref codes, <|im_end|>)*] does
 * not depend on the text to speak, and the model is causal, so its K/V are computed once:
 *   ft_ar_prefill(slot, prefix)  ->  ft_ar_kv_save(slot, n_prefix)            (once per voice)
 *   ft_ar_kv_restore(snap, slot) ->  ft_ar_prefill_at(slot, tail, pos0=n_prefix)  (per utterance)
 * ft_ar_prefill_at feeds `Lp` prompt columns at cache positions [pos0, pos0+Lp) of a slot whose
 * positions [0, pos0) already hold K/V; pos0 = 0 is ft_ar_prefill. */
typedef struct ft_kv_snapshot ft_kv_snapshot;
ft_status ft_ar_prefill_at(ft_ctx* ctx, int32_t slot, const int32_t* prompt, int32_t Lp, int32_t pos0,
                           const ft_sampling* sp, int32_t* out_frame);
ft_status ft_ar_kv_save(ft_ctx* ctx, int32_t slot, int32_t n_pos, ft_kv_snapshot** out);
ft_status ft_ar_kv_restore(ft_ctx* ctx, const ft_kv_snapshot* snap, int32_t slot);
int32_t ft_ar_kv_positions(const ft_kv_snapshot* snap);
void ft_ar_kv_free(ft_ctx* ctx, ft_kv_snapshot* snap);
/* Decode loop = decode_n_tokens[_streaming] (inference.py:158-276) driving decode_one_token_ar
 * (inference.py:83-155), for slots [0, nslots) in lock step, up to n_frames more frames each.
 * out_frames: nslots x n_frames x (num_codebooks+1) int32 (host), frame-major; out_n[slot] =
 * frames produced (the <|im_end|> frame included, as the streaming loop yields it).
 * The step is hipGraph-captured; EOS is polled every `poll` frames (>=1). */
ft_status ft_ar_decode(ft_ctx* ctx, int32_t nslots, int32_t n_frames, const ft_sampling* sp,
                       int32_t poll, int32_t* out_frames, int32_t* out_n);

/* Codec path = DAC.decode (vocoder.py:906-912) incl. DownsampleResidualVectorQuantize.decode
 * (vocoder.py:800-814) and Decoder (vocoder.py:605-640).  codes: B x (n_codebooks+1) x T int32
 * (host), right-padded; lens[b] valid frames.  audio: B x (T*frame_len) float32 (host),
 * samples beyond lens[b]*frame_len are written but meaningless (the codec is causal). */
ft_status ft_codec_decode(ft_ctx* ctx, const int32_t* codes, int32_t B, int32_t T, const int32_t* lens,
                          float* audio);
int32_t ft_codec_frame_len(const ft_ctx* ctx); /* samples per code frame (2048) */
/* Streamed decode with carried state (second half).  The reference's synthesize_stream decodes
 * every chunk from zero state (synthesizer.py:513-528, 591-595); the codec is strictly causal (window-128 attention,
 * vocoder.py:325-332; left-padded convolutions, vocoder.py:411-420, 449-455), so carrying the last 127 frames' K/V of
 * every transformer layer and the last `halo` rows of every convolution input makes the chunks of one stream
 * concatenate to exactly the waveform of one decode of all the codes.  codes: (n_codebooks+1) x T int32 (host), audio:
 * T * frame_len float32 (host).  A stream holds at most max_frames frames.  Kernel variants are chosen as for a 215-frame
 * utterance whatever the chunk length, so the result does not depend on the chunking. */
typedef struct ft_codec_stream ft_codec_stream;
ft_status ft_codec_stream_begin(ft_ctx* ctx, ft_codec_stream** out);
ft_status ft_codec_stream_decode(ft_ctx* ctx, ft_codec_stream* st, const int32_t* codes, int32_t T, float* audio);
void ft_codec_stream_end(ft_ctx* ctx, ft_codec_stream* st);
/* Codec encode = vocoder.encode(audio, lengths) of encode_reference (synthesizer.py:325-357, vocoder.py:885-904):
 * mono f32 audio at the codec sample rate (host), right-padded to whole frames -> codes (num_codebooks+1) x T'
 * int32 row-major (host, row stride = T' = ceil(n_samples / ft_codec_enc_frame_len)); *out_frames = T'. */
ft_status ft_codec_encode(ft_ctx* ctx, const float* audio, int64_t n_samples, int32_t* codes, int32_t* out_frames);
int32_t ft_codec_enc_frame_len(const ft_ctx* ctx);

ft_status ft_sync(ft_ctx* ctx);
/* State of the persistent frame engine (csrc/frame_engine.h), the batch-1 form of the decode step
 * (fish_tts/models/inference.py:83-155 as two launches of one workgroup per CU instead of ~325 launches).
 * flags bit 0: the slow stack runs on it, bit 1: the fast codebook loop runs on it (0 = this context takes the launch
 * path: other widths, f32 precision, FT_NO_ENGINE set, another context of the process owns the device's engine, or the
 * engine was turned off after repeated time-outs).  aborted = hand-off time-outs so far: each one was recovered inside
 * the call that hit it (control words and hand-off buffers cleared, the affected frames redone on the launch path, so
 * the call still returns the frames the launch path yields); after two the context stops using the engine.
 * where = the phase that gave up first in the last such event.  Any pointer may be NULL. */
ft_status ft_ar_engine_state(ft_ctx* ctx, int32_t* flags, int32_t* aborted, int32_t* where);
/* One line of text: which path the batch-1 decode frames of this context take and why (for the host's log). */
const char* ft_ar_frame_path(const ft_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
