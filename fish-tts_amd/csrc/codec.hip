// DAC codec decode path (placeholder until the kernels land in this file).
#include "engine.h"

ft_status codec_create(ft_ctx* ctx) { return ft_fail(ctx, FT_ERR_UNSUPPORTED, "codec path not built yet"); }
void codec_destroy(ft_ctx*) {}
void codec_expected(ft_ctx*) {}
ft_status codec_finalize(ft_ctx*) { return FT_OK; }
extern "C" ft_status ft_codec_decode(ft_ctx* ctx, const int32_t*, int32_t, int32_t, const int32_t*, float*) {
    return ft_fail(ctx, FT_ERR_UNSUPPORTED, "codec path not built yet");
}
extern "C" int32_t ft_codec_frame_len(const ft_ctx*) { return 2048; }
