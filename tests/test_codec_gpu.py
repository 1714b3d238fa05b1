"""GPU parity of the HIP DAC decode path (through the C ABI) against the oracle / golden vectors.
The contractions run on bf16 MFMA with f32 accumulation and activations are stored as bf16, the
oracle is f32: the stated tolerance is a relative RMS error of the waveform <= 3e-2 and, for the
golden file produced by the reference's own vocoder.py, the same bound."""
import os

import numpy as np
import pytest
import torch

from oracle import codec as C
from tests.golden.make_golden_codec import tiny_codec_shape

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
REL_RMS_TOL = 3e-2


def args_from_shape(shape: C.CodecShape):
    from fish_tts_amd.config import CodecArgs
    return CodecArgs(n_codebooks=shape.n_codebooks, codebook_size=shape.codebook_size,
                     semantic_codebook_size=shape.semantic_codebook_size, codebook_dim=shape.codebook_dim,
                     latent_dim=shape.latent_dim, n_tf_layer=shape.n_tf_layer, tf_n_head=shape.tf_n_head,
                     tf_head_dim=shape.tf_head_dim, tf_ffn=shape.tf_ffn, tf_window=shape.tf_window,
                     tf_block_size=shape.tf_block_size, tf_rope_base=shape.tf_rope_base, tf_norm_eps=shape.tf_norm_eps,
                     downsample_factor=list(shape.upsample), decoder_dim=shape.decoder_dim, decoder_rates=list(shape.rates))


def make_codec(shape, seed=0, max_frames=64):
    from fish_tts_amd.codec_engine import CodecHipEngine
    w = C.random_weights(shape, seed=seed)
    eng = CodecHipEngine(args_from_shape(shape), device=0, max_frames=max_frames)
    eng.load_state_dict(w)
    return eng, C.CodecOracle(shape, w)


def rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-12))


def test_tiny_codec_matches_reference_golden():
    gold = np.load(os.path.join(G, "codec_tiny.npz"))
    shape = tiny_codec_shape()
    eng, _ = make_codec(shape)
    for name in ("b1", "b2"):
        codes = gold[f"{name}.codes"]
        got = eng.decode(codes)
        want = gold[f"{name}.audio"][:, 0]
        assert got.shape == want.shape
        assert rel_rms(got, want) <= REL_RMS_TOL, rel_rms(got, want)
    eng.close()


def test_codec_properties_and_ragged_batch():
    shape = tiny_codec_shape()
    eng, orc = make_codec(shape)
    g = torch.Generator().manual_seed(9)
    T = 40
    codes = torch.zeros(3, shape.n_codebooks + 1, T, dtype=torch.long)
    codes[:, 0] = torch.randint(0, shape.semantic_codebook_size, (3, T), generator=g)
    codes[:, 1:] = torch.randint(0, shape.codebook_size, (3, shape.n_codebooks, T), generator=g)
    lens = np.array([40, 17, 1], dtype=np.int32)
    got = eng.decode(codes.numpy(), lens)
    fl = shape.frame_len
    for b in range(3):
        n = int(lens[b])
        want, _ = orc.decode(codes[b: b + 1, :, :n], torch.tensor([n]))
        assert rel_rms(got[b, : n * fl], want[0, 0].numpy()) <= REL_RMS_TOL, b
        assert np.all(got[b, n * fl:] == 0)
    # strict causality: the decode of a prefix equals the prefix of the decode (bit-exact on the GPU:
    # same kernels, same tiles, rows only look backwards)
    pre = eng.decode(codes.numpy()[:1, :, :16])
    assert np.array_equal(pre[0], got[0, : 16 * fl])
    eng.close()


def test_s1_mini_codec_shapes_vs_oracle():
    """Real channel widths (1024 / 1536 / strides 8,8,4,2) on a few frames."""
    shape = C.CodecShape()
    eng, orc = make_codec(shape, max_frames=16)
    g = torch.Generator().manual_seed(3)
    T = 6
    codes = torch.zeros(1, 10, T, dtype=torch.long)
    codes[:, 0] = torch.randint(0, 4096, (1, T), generator=g)
    codes[:, 1:] = torch.randint(0, 1024, (1, 9, T), generator=g)
    got = eng.decode(codes.numpy())
    want, lens = orc.decode(codes, torch.tensor([T]))
    assert got.shape[1] == int(lens[0]) == T * 2048
    assert rel_rms(got[0], want[0, 0].numpy()) <= REL_RMS_TOL, rel_rms(got[0], want[0, 0].numpy())
    eng.close()
