"""Model configuration, config.json-compatible with the reference
(fish_tts/models/llama.py:31-123: BaseModelArgs / DualARModelArgs) and the codec hyper-parameters the
reference hard-codes in fish_tts/synthesizer.py:199-269."""
from __future__ import annotations

import json
from dataclasses import dataclass, field, fields
from pathlib import Path
from typing import List, Optional


def _round_up(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - n % k


@dataclass
class DualARModelArgs:
    model_type: str = "dual_ar"
    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: Optional[int] = None
    n_local_heads: int = -1
    head_dim: Optional[int] = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5
    max_seq_len: int = 2048
    dropout: float = 0.0
    tie_word_embeddings: bool = True
    attention_qkv_bias: bool = False
    attention_o_bias: bool = False
    attention_qk_norm: bool = False
    codebook_size: int = 160
    num_codebooks: int = 4
    use_gradient_checkpointing: bool = True
    initializer_range: float = 0.02
    is_reward_model: bool = False
    scale_codebook_embeddings: bool = False
    n_fast_layer: int = 4
    fast_dim: Optional[int] = None
    fast_n_head: Optional[int] = None
    fast_n_local_heads: Optional[int] = None
    fast_head_dim: Optional[int] = None
    fast_intermediate_size: Optional[int] = None
    fast_attention_qkv_bias: Optional[bool] = None
    fast_attention_qk_norm: Optional[bool] = None
    fast_attention_o_bias: Optional[bool] = None

    def __post_init__(self):
        # defaulting rules of llama.py:64-72 and 102-123
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            self.intermediate_size = _round_up(int(2 * (4 * self.dim) / 3), 256)
        if self.head_dim is None:
            self.head_dim = self.dim // self.n_head
        self.fast_dim = self.fast_dim or self.dim
        self.fast_n_head = self.fast_n_head or self.n_head
        self.fast_n_local_heads = self.fast_n_local_heads or self.n_local_heads
        self.fast_head_dim = self.fast_head_dim or self.head_dim
        self.fast_intermediate_size = self.fast_intermediate_size or self.intermediate_size
        for name in ("attention_qkv_bias", "attention_qk_norm", "attention_o_bias"):
            if getattr(self, "fast_" + name) is None:
                setattr(self, "fast_" + name, getattr(self, name))

    @staticmethod
    def from_pretrained(path: str) -> "DualARModelArgs":
        p = Path(path)
        if p.is_dir():
            p = p / "config.json"
        with open(p, "r", encoding="utf-8") as f:
            data = json.load(f)
        if data.get("model_type") != "dual_ar":
            raise ValueError(f"Unknown model type: {data.get('model_type')}")
        known = {f.name for f in fields(DualARModelArgs)}
        return DualARModelArgs(**{k: v for k, v in data.items() if k in known})


def s1_mini_args(**over) -> DualARModelArgs:
    """openaudio-s1-mini shapes (SURVEY.md §8 'Shapes')."""
    kw = dict(vocab_size=155776, n_layer=28, n_head=16, dim=1024, intermediate_size=3072, n_local_heads=8,
              head_dim=128, rope_base=1e6, norm_eps=1e-6, max_seq_len=8192, tie_word_embeddings=True,
              attention_qk_norm=True, codebook_size=4096, num_codebooks=10, scale_codebook_embeddings=True,
              n_fast_layer=4, fast_dim=1024, fast_n_head=16, fast_n_local_heads=8, fast_head_dim=64,
              fast_intermediate_size=3072, fast_attention_qk_norm=False)
    kw.update(over)
    return DualARModelArgs(**kw)


@dataclass
class CodecArgs:
    """DAC decode-path hyper-parameters (synthesizer.py:199-269, vocoder.py:824-872)."""
    sample_rate: int = 44100
    n_codebooks: int = 9
    codebook_size: int = 1024
    semantic_codebook_size: int = 4096
    codebook_dim: int = 8
    latent_dim: int = 1024
    n_tf_layer: int = 8
    tf_n_head: int = 16
    tf_head_dim: int = 64
    tf_ffn: int = 3072
    tf_window: int = 128
    tf_block_size: int = 4096
    tf_rope_base: float = 10000.0
    tf_norm_eps: float = 1e-5
    downsample_factor: List[int] = field(default_factory=lambda: [2, 2])
    decoder_dim: int = 1536
    decoder_rates: List[int] = field(default_factory=lambda: [8, 8, 4, 2])
    # encode side (encode_reference; synthesizer.py:255-268).  encoder_dim = 0 builds a decode-only codec.
    encoder_dim: int = 64
    encoder_rates: List[int] = field(default_factory=lambda: [2, 4, 8, 8])
    encoder_transformer_layers: List[int] = field(default_factory=lambda: [0, 0, 0, 4])
    encoder_tf_window: int = 512
    max_reference_seconds: float = 60.0

    @property
    def hop_length(self) -> int:
        n = 1
        for r in self.encoder_rates:
            n *= r
        return n

    @property
    def encode_frame_length(self) -> int:
        """DAC.frame_length = hop_length * 4 (vocoder.py:872)."""
        return self.hop_length * 4

    @property
    def frame_length(self) -> int:
        n = 1
        for r in self.decoder_rates:
            n *= r
        for r in self.downsample_factor:
            n *= r
        return n
