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
