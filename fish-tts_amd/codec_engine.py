"""Host driver of the HIP DAC decode path: the S3 seam of SURVEY.md §8b, `vocoder.decode(indices,
feature_lengths)` (fish_tts/models/vocoder.py:906-912), on top of the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from .ar_engine import HipError, _rope_table
from .config import CodecArgs


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """codec.pth stores weight-normed convs as parametrizations.weight.original0 (g) / original1 (v)
    (vocoder.py:423-429,457-463: torch weight_norm, dim=0).  Fold them to plain `.weight` tensors and
    drop the "generator." prefix the reference strips in synthesizer.py:276-282."""
    if "state_dict" in sd:
        sd = sd["state_dict"]
    if any("generator." in k for k in sd):
        sd = {k.replace("generator.", ""): v for k, v in sd.items() if "generator." in k}
    out = {}
    for k, v in sd.items():
        if k.endswith("parametrizations.weight.original0"):
            base = k[: -len("parametrizations.weight.original0")]
            g, w = v.float(), sd[base + "parametrizations.weight.original1"].float()
            norm = w.flatten(1).norm(dim=1).view(-1, *([1] * (w.dim() - 1)))
            out[base + "weight"] = g * w / norm
        elif k.endswith("parametrizations.weight.original1"):
            continue
        else:
            out[k] = v
    return out


class CodecHipEngine:
    def __init__(self, args: Optional[CodecArgs] = None, device: int = 0, max_frames: int = 2048, max_batch: int = 1,
                 with_encoder: bool = False):
        """`with_encoder`: also build the encode side (encode_reference); its tensors are then required at load."""
        self.args = args or CodecArgs()
        a = self.args
        self.with_encoder = bool(with_encoder and a.encoder_dim > 0)
        self.lib = L.load()
        self._streams = set()          # open CodecStreams (ended before the native context goes)
        c = L.ft_codec_config()
        c.dtype = L.FT_BF16
        c.n_codebooks, c.codebook_size, c.semantic_codebook_size = a.n_codebooks, a.codebook_size, a.semantic_codebook_size
        c.codebook_dim, c.latent_dim = a.codebook_dim, a.latent_dim
        c.n_tf_layer, c.tf_n_head, c.tf_head_dim, c.tf_ffn, c.tf_window = a.n_tf_layer, a.tf_n_head, a.tf_head_dim, a.tf_ffn, a.tf_window
        c.tf_rope_base, c.tf_norm_eps = float(a.tf_rope_base), float(a.tf_norm_eps)
        if any(f != 2 for f in a.downsample_factor):
            raise NotImplementedError("upsample stages other than x2 are not implemented")
        c.n_upsample = len(a.downsample_factor)
        c.decoder_dim, c.n_rates = a.decoder_dim, len(a.decoder_rates)
        for i, r in enumerate(a.decoder_rates):
            c.rates[i] = r
        c.max_frames, c.max_batch = int(max_frames), int(max_batch)
        self.max_enc_frames = 0
        if self.with_encoder:
            c.encoder_dim, c.n_enc_rates, c.enc_tf_window = a.encoder_dim, len(a.encoder_rates), a.encoder_tf_window
            for i, (r, nt) in enumerate(zip(a.encoder_rates, a.encoder_transformer_layers)):
                c.enc_rates[i], c.enc_tf_layers[i] = r, nt
            want = int(a.max_reference_seconds * a.sample_rate) // a.encode_frame_length + 1
            self.max_enc_frames = c.max_enc_frames = max(1, min(int(max_frames), want))
        self.cfg = c
        self.max_frames = max_frames
        self.R = a.n_codebooks + 1
        self._h = C.c_void_p()
        st = self.lib.ft_create(None, C.byref(c), device, C.byref(self._h))
        if st != L.FT_OK:
            raise HipError(f"ft_create(codec) failed ({st}): {self.lib.ft_last_error(None).decode()}")
        self.frame_len = self.lib.ft_codec_frame_len(self._h)
        self.enc_frame_len = self.lib.ft_codec_enc_frame_len(self._h)

    def close(self):
        for st in list(getattr(self, "_streams", ())):
            st.close()
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.ft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, what):
        if st != L.FT_OK:
            msg = self.lib.ft_last_error(self._h).decode()
            if st == L.FT_ERR_STATE and "Vocoder not loaded" in msg:
                raise RuntimeError("Vocoder not loaded")  # synthesizer.py:599-600
            raise HipError(f"{what} failed ({st}): {msg}")

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        a = self.args
        sd = fold_weight_norm(sd)
        want = None
        for k, v in sd.items():
            t = v.detach().float().contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            st = self.lib.ft_load_weight(self._h, k.encode(), C.c_void_p(t.data_ptr()), L.FT_F32, shape, t.dim())
            if st == L.FT_ERR_ARG and b"unknown weight name" in self.lib.ft_last_error(self._h):
                continue  # encoder / pre_module / training-only tensors of codec.pth
            self._check(st, f"ft_load_weight({k})")
        tab = _rope_table(self.max_frames, a.tf_head_dim, a.tf_rope_base)
        shape = (C.c_int64 * 3)(*tab.shape)
        self._check(self.lib.ft_load_weight(self._h, b"rope.codec", C.c_void_p(tab.data_ptr()), L.FT_F32, shape, 3), "rope.codec")
        if self.with_encoder:
            rate, npos = 1, 1
            for r, nt in zip(a.encoder_rates, a.encoder_transformer_layers):
                rate *= r
                if nt > 0:
                    npos = max(npos, self.max_enc_frames * a.encode_frame_length // rate)
            tab2 = _rope_table(npos, 64, a.tf_rope_base)   # encoder transformers: head_dim 64 (synthesizer.py:249)
            shape2 = (C.c_int64 * 3)(*tab2.shape)
            self._check(self.lib.ft_load_weight(self._h, b"rope.codec_enc", C.c_void_p(tab2.data_ptr()), L.FT_F32, shape2, 3),
                        "rope.codec_enc")
        self._check(self.lib.ft_finalize_weights(self._h), "ft_finalize_weights")

    @classmethod
    def synthetic(cls, device: int = 0, max_frames: int = 2048, seed: int = 0, args: Optional[CodecArgs] = None,
                  with_encoder: bool = False):
        from .weights import random_codec_state_dict
        eng = cls(args, device=device, max_frames=max_frames, with_encoder=with_encoder)
        eng.load_state_dict(random_codec_state_dict(eng.args, seed, with_encoder=eng.with_encoder))
        return eng

    def encode(self, audio: np.ndarray) -> np.ndarray:
        """vocoder.encode of encode_reference (synthesizer.py:345-352, vocoder.py:885-904): mono float audio at the
        codec sample rate -> (n_codebooks + 1, T') int64 codes, T' = ceil(len / encode_frame_length)."""
        if not self.with_encoder:
            raise RuntimeError("codec encoder not built (CodecHipEngine(..., with_encoder=True))")
        audio = np.ascontiguousarray(np.asarray(audio, dtype=np.float32).reshape(-1))
        T = (audio.shape[0] + self.enc_frame_len - 1) // self.enc_frame_len
        codes = np.zeros((self.R, max(T, 1)), dtype=np.int32)
        n = C.c_int32(0)
        self._check(self.lib.ft_codec_encode(self._h, audio.ctypes.data_as(C.c_void_p), audio.shape[0],
                                             codes.ctypes.data_as(C.c_void_p), C.byref(n)), "ft_codec_encode")
        assert n.value == T, (n.value, T)
        return codes.astype(np.int64)

    def rvq_encode(self, z: np.ndarray) -> np.ndarray:
        """Test hook: the quantiser search alone on pre-quantiser latents z (T, latent_dim) f32."""
        z = np.ascontiguousarray(z, dtype=np.float32)
        codes = np.zeros((self.R, z.shape[0]), dtype=np.int32)
        self._check(self.lib.ft_codec_rvq_encode(self._h, z.ctypes.data_as(C.c_void_p), z.shape[0],
                                                 codes.ctypes.data_as(C.c_void_p)), "ft_codec_rvq_encode")
        return codes

    def stream(self) -> "CodecStream":
        """A streamed decode with carried state: the chunks' waveforms concatenate to the waveform of one decode."""
        return CodecStream(self)

    def decode(self, codes: np.ndarray, lens: Optional[np.ndarray] = None) -> np.ndarray:
        """codes (B, n_codebooks+1, T) or (n_codebooks+1, T) integer -> float32 (B, T*frame_len)."""
        codes = np.asarray(codes)
        if codes.ndim == 2:
            codes = codes[None]
        codes = np.ascontiguousarray(codes, dtype=np.int32)
        B, R, T = codes.shape
        assert R == self.R, codes.shape
        lens_a = np.full(B, T, dtype=np.int32) if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
        audio = np.empty((B, T * self.frame_len), dtype=np.float32)
        self._check(self.lib.ft_codec_decode(self._h, codes.ctypes.data_as(C.c_void_p), B, T,
                                             lens_a.ctypes.data_as(C.c_void_p), audio.ctypes.data_as(C.c_void_p)),
                    "ft_codec_decode")
        return audio


class CodecStream:
    """Successive chunks of ONE utterance's codes (fish_tts/synthesizer.py:513-528 decodes each chunk from zero state;
    here the causal codec's context - the last 127 frames' K/V of every transformer layer, the last rows of every
    convolution input - is carried)."""

    def __init__(self, engine: CodecHipEngine):
        self.engine = engine
        self._h = C.c_void_p()
        engine._check(engine.lib.ft_codec_stream_begin(engine._h, C.byref(self._h)), "ft_codec_stream_begin")
        self.frames = 0
        engine._streams.add(self)      # the engine ends its open streams before it destroys the native context

    def decode(self, codes: np.ndarray) -> np.ndarray:
        """codes (n_codebooks+1, T) integer -> float32 (T * frame_len,): the next T frames of the stream."""
        e = self.engine
        codes = np.ascontiguousarray(np.asarray(codes), dtype=np.int32)
        assert codes.ndim == 2 and codes.shape[0] == e.R, codes.shape
        T = codes.shape[1]
        audio = np.empty(T * e.frame_len, dtype=np.float32)
        e._check(e.lib.ft_codec_stream_decode(e._h, self._h, codes.ctypes.data_as(C.c_void_p), T,
                                              audio.ctypes.data_as(C.c_void_p)), "ft_codec_stream_decode")
        self.frames += T
        return audio

    def close(self) -> None:
        if self._h:
            self.engine.lib.ft_codec_stream_end(self.engine._h, self._h)   # the stream knows its context: a closed engine is fine
            self._h = C.c_void_p()
        self.engine._streams.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
