"""Codec golden vectors from the reference's own vocoder.py (run via make_golden.py)."""
import os

import numpy as np
import torch
import torch.nn as nn

OUT = os.path.dirname(os.path.abspath(__file__))


def tiny_codec_shape():
    from oracle.codec import CodecShape
    return CodecShape(n_codebooks=3, codebook_size=64, semantic_codebook_size=128, codebook_dim=8, latent_dim=64,
                      n_tf_layer=2, tf_n_head=4, tf_head_dim=16, tf_ffn=96, tf_window=8, tf_block_size=256,
                      tf_rope_base=10000.0, tf_norm_eps=1e-5, upsample=[2, 2], decoder_dim=128, rates=[4, 2])


def build_reference_codec(vocoder, shape, weights):
    from torch.nn.utils.parametrize import remove_parametrizations
    cfg = lambda **kw: vocoder.VocoderModelArgs(  # noqa: E731
        block_size=shape.tf_block_size, n_layer=shape.n_tf_layer, n_head=shape.tf_n_head,
        dim=shape.tf_n_head * shape.tf_head_dim, intermediate_size=shape.tf_ffn, n_local_heads=-1,
        head_dim=shape.tf_head_dim, rope_base=shape.tf_rope_base, norm_eps=shape.tf_norm_eps, dropout_rate=0.1,
        attn_dropout_rate=0.1, channels_first=True)
    post = vocoder.WindowLimitedTransformer(causal=True, window_size=shape.tf_window, input_dim=shape.latent_dim,
                                            config=cfg())
    q = vocoder.DownsampleResidualVectorQuantize(
        input_dim=shape.latent_dim, n_codebooks=shape.n_codebooks, codebook_size=shape.codebook_size,
        codebook_dim=shape.codebook_dim, quantizer_dropout=0.5, downsample_factor=tuple(reversed(shape.upsample)),
        post_module=post, pre_module=None, semantic_codebook_size=shape.semantic_codebook_size)
    dac = vocoder.DAC(sample_rate=44100, encoder_dim=8, encoder_rates=[2, 4], latent_dim=shape.latent_dim,
                      decoder_dim=shape.decoder_dim, decoder_rates=list(shape.rates), quantizer=q, causal=True,
                      encoder_transformer_layers=[0, 0], decoder_transformer_layers=[0] * len(shape.rates),
                      transformer_general_config=cfg)
    # fold away every weight-norm parametrisation so plain (folded) tensors can be assigned by name
    for mod in dac.modules():
        if isinstance(mod, (nn.Conv1d, nn.ConvTranspose1d)) and hasattr(mod, "parametrizations"):
            remove_parametrizations(mod, "weight")
    sd = dac.state_dict()
    for k, v in weights.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    missing, unexpected = dac.load_state_dict(weights, strict=False)
    assert not unexpected, unexpected
    return dac.eval()


def tiny_encode_shape():
    """Encode-side tiny model: the encoder transformer's head_dim is 64 in the reference (synthesizer.py:243-256),
    so the last encoder block and the latent are 128 wide."""
    from oracle.codec import CodecShape
    return CodecShape(n_codebooks=3, codebook_size=64, semantic_codebook_size=128, codebook_dim=8, latent_dim=128,
                      n_tf_layer=2, tf_n_head=2, tf_head_dim=64, tf_ffn=96, tf_window=4, tf_block_size=256,
                      tf_rope_base=10000.0, tf_norm_eps=1e-5, upsample=[2, 2], decoder_dim=64, rates=[2, 4, 2],
                      encoder_dim=8, encoder_rates=[2, 2, 2, 2], encoder_tf_layers=[0, 0, 2, 1], enc_tf_window=8,
                      enc_tf_block_size=512)


def build_reference_codec_full(vocoder, shape, weights):
    """DAC with the encode side populated (encoder transformers, quantizer.downsample / pre_module / in_proj)."""
    from torch.nn.utils.parametrize import remove_parametrizations
    qcfg = lambda **kw: vocoder.VocoderModelArgs(  # noqa: E731
        block_size=shape.tf_block_size, n_layer=shape.n_tf_layer, n_head=shape.tf_n_head,
        dim=shape.tf_n_head * shape.tf_head_dim, intermediate_size=shape.tf_ffn, n_local_heads=-1,
        head_dim=shape.tf_head_dim, rope_base=shape.tf_rope_base, norm_eps=shape.tf_norm_eps, dropout_rate=0.1,
        attn_dropout_rate=0.1, channels_first=True)
    gen_cfg = lambda **kw: vocoder.VocoderModelArgs(  # noqa: E731  (synthesizer.py:243-256)
        block_size=kw.get("block_size", shape.enc_tf_block_size), n_layer=kw.get("n_layer", 8), n_head=kw.get("n_head", 8),
        dim=kw.get("dim", 512), intermediate_size=kw.get("intermediate_size", 1536), n_local_heads=-1, head_dim=64,
        rope_base=10000, norm_eps=1e-5, dropout_rate=0.1, attn_dropout_rate=0.1, channels_first=True)
    mk = lambda: vocoder.WindowLimitedTransformer(causal=True, window_size=shape.tf_window,  # noqa: E731
                                                  input_dim=shape.latent_dim, config=qcfg())
    q = vocoder.DownsampleResidualVectorQuantize(
        input_dim=shape.latent_dim, n_codebooks=shape.n_codebooks, codebook_size=shape.codebook_size,
        codebook_dim=shape.codebook_dim, quantizer_dropout=0.5, downsample_factor=tuple(reversed(shape.upsample)),
        post_module=mk(), pre_module=mk(), semantic_codebook_size=shape.semantic_codebook_size)
    dac = vocoder.DAC(sample_rate=44100, encoder_dim=shape.encoder_dim, encoder_rates=list(shape.encoder_rates),
                      decoder_dim=shape.decoder_dim, decoder_rates=list(shape.rates), quantizer=q, causal=True,
                      encoder_transformer_layers=list(shape.encoder_tf_layers),
                      decoder_transformer_layers=[0] * len(shape.rates), transformer_general_config=gen_cfg)
    import types
    for m in dac.modules():   # the encoder transformers use window_size=512 in the reference; shrink for the tiny case
        if isinstance(m, vocoder.WindowLimitedTransformer) and m.window_size == 512:
            m.window_size = shape.enc_tf_window
    for mod in dac.modules():
        if isinstance(mod, (nn.Conv1d, nn.ConvTranspose1d)) and hasattr(mod, "parametrizations"):
            remove_parametrizations(mod, "weight")
    sd = dac.state_dict()
    for k, v in weights.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    missing, unexpected = dac.load_state_dict(weights, strict=False)
    assert not unexpected, unexpected
    left = [m for m in missing if not m.endswith(("freqs_cis", "causal_mask"))]
    assert not left, left
    del types
    return dac.eval()


def gen_encode(vocoder):
    """DAC.encode (vocoder.py:885-904) of the reference on seeded audio: pins oracle.codec.CodecOracle.encode."""
    from oracle.codec import random_encoder_weights, random_weights
    shape = tiny_encode_shape()
    w = random_weights(shape, seed=0)
    w.update(random_encoder_weights(shape, seed=1))
    dac = build_reference_codec_full(vocoder, shape, w)
    assert dac.frame_length == shape.enc_frame_len == shape.frame_len, (dac.frame_length, shape.enc_frame_len, shape.frame_len)
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, B, T in (("e1", 1, 1000), ("e2", 2, 640)):
        t = torch.arange(T).float()
        audio = 0.4 * torch.sin(2 * np.pi * t / 37.0)[None].repeat(B, 1) + 0.2 * torch.randn(B, T, generator=g)
        lens = torch.tensor([T] + [T - 100] * (B - 1))
        with torch.inference_mode():
            idx, ilens = dac.encode(audio[:, None], lens)
            back, _ = dac.decode(idx, ilens)
        out[f"{name}.audio"] = audio.numpy()
        out[f"{name}.lens"] = lens.numpy()
        out[f"{name}.indices"] = idx.numpy()
        out[f"{name}.indices_lens"] = ilens.numpy()
        out[f"{name}.roundtrip"] = back.numpy()
    np.savez_compressed(os.path.join(OUT, "codec_encode_tiny.npz"), **out)
    print("codec_encode_tiny", {k: v.shape for k, v in out.items()})


def main(vocoder):
    from oracle.codec import random_weights
    shape = tiny_codec_shape()
    w = random_weights(shape, seed=0)
    dac = build_reference_codec(vocoder, shape, w)
    g = torch.Generator().manual_seed(2)
    out = {}
    for name, B, T in (("b1", 1, 23), ("b2", 2, 11)):
        codes = torch.zeros(B, shape.n_codebooks + 1, T, dtype=torch.long)
        codes[:, 0] = torch.randint(0, shape.semantic_codebook_size, (B, T), generator=g)
        codes[:, 1:] = torch.randint(0, shape.codebook_size, (B, shape.n_codebooks, T), generator=g)
        codes[0, 0, 0] = shape.semantic_codebook_size + 5  # clamped from above (vocoder.py:801-807)
        with torch.inference_mode():
            audio, lens = dac.decode(codes, torch.tensor([T] * B))
        out[f"{name}.codes"] = codes.numpy()
        out[f"{name}.audio"] = audio.numpy()
        out[f"{name}.lens"] = lens.numpy()
    np.savez_compressed(os.path.join(OUT, "codec_tiny.npz"), **out)
    print("codec_tiny", {k: v.shape for k, v in out.items()})
