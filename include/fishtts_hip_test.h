/* Test and measurement hooks of libfishtts_hip.so - NOT part of the drop-in boundary (include/fishtts_hip.h).
 * Nothing in the product path (fish-tts_amd/*.py outside ARHipEngine's test helpers) calls them: tests/ use them to
 * inject noise, read logits back and provoke the frame engine's recovery; bench.py uses the two profile calls. */
#ifndef FISHTTS_HIP_TEST_H
#define FISHTTS_HIP_TEST_H
#include "fishtts_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hooks: inject the Exp(1) noise the sampler divides by (inference.py:26), one row of
 * `row_len` floats per generated frame (slow vocab draws first, then (num_codebooks-1) x 1024);
 * q == NULL restores the RNG.  Read back the last slow logits / pre-norm hidden of a slot. */
ft_status ft_ar_set_noise(ft_ctx* ctx, const float* q, int64_t n_rows, int64_t row_len);
ft_status ft_ar_get_debug(ft_ctx* ctx, int32_t slot, float* logits /*vocab*/, float* hidden /*fast_dim*/);

/* Test hook: the residual vector quantiser search alone on given pre-quantiser latents z [T][latent_dim] f32 (host). */
ft_status ft_codec_rvq_encode(ft_ctx* ctx, const float* z, int32_t T, int32_t* codes);

/* Measurement hook used by bench.py (never by the product path).  The weight-streaming GEMV launches of
 * one decode frame whose weights come from HBM (4 per slow layer + the vocabulary head; the fast stack's
 * 100 MB stay cache-resident and are excluded) are captured into a hipGraph and replayed `frames` times
 * between two HIP events on the engine's own stream.  Returns the elapsed device ms, the number of kernel
 * launches timed and the algorithmic bytes those launches stream. */
ft_status ft_ar_profile_gemv(ft_ctx* ctx, int32_t frames, const ft_sampling* sp, double* ms,
                             int64_t* launches, int64_t* bytes);
/* Measurement hook used by bench.py (never by the product path): `frames` real decode frames of slot 0 (prefilled by
 * the caller, enough frame / cache capacity left) timed with HIP events on the engine's own stream.
 * ms_graph: elapsed ms of `frames` back-to-back replays of the captured frame graph (what ft_ar_decode runs);
 * seg_ms[3]: ms summed over frames-1 further frames launched eagerly with events between the three parts of a frame:
 * the slow stack, the vocabulary head + semantic draw, the fast codebook loop (bf16 only, else zeros);
 * nodes_per_frame: launches in the captured frame.  Reference: one decode_one_token_ar call, inference.py:83-155. */
ft_status ft_ar_profile_frame(ft_ctx* ctx, int32_t frames, const ft_sampling* sp, double* ms_graph,
                              double* seg_ms, int32_t* nodes_per_frame);

/* Test hook: workgroup `wg` of a coming slow-stack (which = 0) or codebook-loop (which = 1) engine launch publishes
 * nothing, so the launch times out (one shot); `skip` launches of that kind pass first (a later burst of a call, a
 * launch inside a multi-frame graph).  Exercises the recovery described above. */
ft_status ft_test_engine_fault(ft_ctx* ctx, int32_t which, int32_t wg, int32_t skip);

/* Test hook: one draw of the sampling kernel (inference.py:30-80) on caller-supplied logits.
 * cb = 0 draws from `vocab_size` logits, cb >= 1 from min(1024, codebook_size); window is the
 * (num_codebooks+1) x 16 penalty window of inference.py:187-191 or NULL (no penalty); q the Exp(1)
 * noise (same length as the logits) or NULL (RNG).  Clobbers slot 0. */
ft_status ft_test_sample(ft_ctx* ctx, const float* logits, int32_t cb, const ft_sampling* sp,
                         const int32_t* window, const float* q, int32_t* out_index);

#ifdef __cplusplus
}
#endif
#endif
