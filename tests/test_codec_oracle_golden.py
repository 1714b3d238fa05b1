"""Pins oracle/codec.py against the reference's own vocoder.py decode (tests/golden/codec_tiny.npz)."""
import os

import numpy as np
import torch

from oracle import codec as C
from tests.golden.make_golden_codec import tiny_codec_shape

G = os.path.join(os.path.dirname(__file__), "golden")


def test_codec_oracle_matches_reference():
    torch.set_num_threads(4)  # as the generator; conv/matmul partial-sum order depends on it
    gold = np.load(os.path.join(G, "codec_tiny.npz"))
    shape = tiny_codec_shape()
    orc = C.CodecOracle(shape, C.random_weights(shape, seed=0))
    for name in ("b1", "b2"):
        codes = torch.from_numpy(gold[f"{name}.codes"])
        audio, lens = orc.decode(codes, torch.tensor([codes.shape[-1]] * codes.shape[0]))
        assert np.array_equal(lens.numpy(), gold[f"{name}.lens"])
        # same ops in the same order on the same machine: bit-equal
        assert np.array_equal(audio.numpy(), gold[f"{name}.audio"]), np.abs(audio.numpy() - gold[f"{name}.audio"]).max()


def test_codec_is_causal_and_batch_independent():
    """Properties SURVEY.md §8a verified on the reference: strict causality, batch independence,
    and chunked decode != full decode (chunks restart from zero context)."""
    shape = tiny_codec_shape()
    orc = C.CodecOracle(shape, C.random_weights(shape, seed=0))
    g = torch.Generator().manual_seed(5)
    T = 20
    codes = torch.zeros(2, shape.n_codebooks + 1, T, dtype=torch.long)
    codes[:, 0] = torch.randint(0, shape.semantic_codebook_size, (2, T), generator=g)
    codes[:, 1:] = torch.randint(0, shape.codebook_size, (2, shape.n_codebooks, T), generator=g)
    full, _ = orc.decode(codes, torch.tensor([T, T]))
    pre, _ = orc.decode(codes[:, :, :12], torch.tensor([12, 12]))
    fl = shape.frame_len
    assert torch.allclose(full[..., : 12 * fl], pre, atol=1e-5)
    one, _ = orc.decode(codes[1:2], torch.tensor([T]))
    assert torch.allclose(one, full[1:2], atol=1e-5)
    tail, _ = orc.decode(codes[:, :, 12:], torch.tensor([T - 12, T - 12]))
    assert not torch.allclose(full[..., 12 * fl:], tail, atol=1e-3)


def test_codec_oracle_encode_matches_reference():
    """DAC.encode of the reference (vocoder.py:885-904; the dac RVQ forward restated in the generator) on seeded audio:
    the oracle's indices must be equal, batch and ragged lengths included, and so must the decode of those indices."""
    from tests.golden.make_golden_codec import tiny_encode_shape
    torch.set_num_threads(4)
    gold = np.load(os.path.join(G, "codec_encode_tiny.npz"))
    shape = tiny_encode_shape()
    w = C.random_weights(shape, seed=0)
    w.update(C.random_encoder_weights(shape, seed=1))
    orc = C.CodecOracle(shape, w)
    for name in ("e1", "e2"):
        audio = torch.from_numpy(gold[f"{name}.audio"])
        idx, lens = orc.encode(audio[:, None], torch.from_numpy(gold[f"{name}.lens"]))
        assert np.array_equal(idx.numpy(), gold[f"{name}.indices"]), name
        assert np.array_equal(lens.numpy(), gold[f"{name}.indices_lens"]), name
        back, _ = orc.decode(idx, lens)
        assert np.allclose(back.numpy(), gold[f"{name}.roundtrip"], atol=1e-5), name


def test_codec_oracle_bf16_encode_mode_stays_close_to_f32():
    """oracle/codec.py's bf16 mode (the precision the reference encodes in, synthesizer.py:289-291): same algorithm, every
    tensor bf16.  Against the f32 oracle on the same audio most indices agree and the rebuilt latents stay close; shapes,
    lengths and index ranges are those of the f32 path.  (Parity of this mode with the reference's bf16 encode is
    unpinned: the reference's fixtures hold no bf16 encode output; the f32 path is pinned by
    test_codec_oracle_encode_matches_reference.)"""
    import numpy as np
    import torch
    from oracle import codec as C
    shape = C.CodecShape(n_codebooks=3, codebook_size=64, semantic_codebook_size=128, codebook_dim=8, latent_dim=512,
                         n_tf_layer=2, tf_n_head=8, tf_head_dim=64, tf_ffn=768, tf_window=8, tf_block_size=256,
                         upsample=[2, 2], decoder_dim=128, rates=[4, 4], encoder_dim=32, encoder_rates=[2, 2, 2, 2],
                         encoder_tf_layers=[0, 0, 1, 1], enc_tf_window=16, enc_tf_block_size=1024)
    w = C.random_weights(shape, seed=0)
    w.update(C.random_encoder_weights(shape, seed=1))
    g = torch.Generator().manual_seed(5)
    n = 37 * shape.enc_frame_len - 11
    t = torch.arange(n).float()
    audio = (0.4 * torch.sin(2 * np.pi * t / 37.0) + 0.2 * torch.randn(n, generator=g))[None, None]
    o32, o16 = C.CodecOracle(shape, w), C.CodecOracle(shape, w, dtype=torch.bfloat16)
    a, la = o32.encode(audio, torch.tensor([n]))
    b, lb = o16.encode(audio, torch.tensor([n]))
    assert a.shape == b.shape and torch.equal(la, lb) and b.dtype == torch.int64
    assert int(b[:, 0].max()) < shape.semantic_codebook_size and int(b[:, 1:].max()) < shape.codebook_size and int(b.min()) >= 0
    assert float((a[0, 0] == b[0, 0]).float().mean()) >= 0.8
    o32.quantizer_decode(a)
    za = o32.taps["rvq"].clone()
    o32.quantizer_decode(b)
    zb = o32.taps["rvq"]
    assert float((za - zb).pow(2).mean().sqrt() / za.pow(2).mean().sqrt()) <= 0.5
