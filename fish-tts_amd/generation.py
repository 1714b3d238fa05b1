"""generate_long (fish_tts/models/inference.py:741-846) on the HIP engine: prompt build, length guard,
batch or streaming generation, GenerateResponse protocol."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Literal, Optional, Sequence, Union

import numpy as np

from .ar_engine import ARHipEngine
from .prompt import build_prompt_split


@dataclass
class GenerateResponse:
    action: Literal["sample", "next"]
    codes: Optional[np.ndarray] = None
    text: Optional[str] = None


def generate_long(*, engine: ARHipEngine, tokenizer, text: str, num_samples: int = 1, max_new_tokens: int = 0,
                  top_p: float = 0.8, repetition_penalty: float = 1.1, temperature: float = 0.8,
                  prompt_text: Optional[Union[str, Sequence[str]]] = None, prompt_tokens=None,
                  streaming: bool = False, seed: int = 0, stream_burst: int = 4,
                  prefix_cache: Optional["PrefixCache"] = None) -> Iterator[GenerateResponse]:
    assert 0 < top_p <= 1, "top_p must be in (0, 1]"
    assert 0 < repetition_penalty < 2, "repetition_penalty must be in (0, 2)"
    assert 0 < temperature < 2, "temperature must be in (0, 2)"
    ncb = engine.args.num_codebooks
    encoded, n_prefix = build_prompt_split(tokenizer, text, prompt_text, prompt_tokens, ncb)
    max_length = engine.args.max_seq_len
    if encoded.shape[1] > max_length - 2048:
        raise ValueError(f"Prompt is too long: {encoded.shape[1]} > {max_length - 2048}")
    kw = dict(temperature=temperature, top_p=top_p, repetition_penalty=repetition_penalty)
    if prefix_cache is not None and n_prefix >= prefix_cache.min_positions:
        kw["prefix"] = prefix_cache.get(engine, encoded[:, :n_prefix])  # K/V of the references, computed once
    for sample_idx in range(num_samples):
        if streaming:
            for block in engine.generate_streaming(encoded, max_new_tokens, seed=seed + sample_idx, chunk=stream_burst, **kw):
                for j in range(block.shape[1]):  # one response per frame, like the reference's per-token yield
                    codes = block[:, j: j + 1].copy()
                    codes[codes < 0] = 0
                    yield GenerateResponse(action="sample", codes=codes, text=text)
        else:
            y = engine.generate(encoded, max_new_tokens, seed=seed + sample_idx, **kw)
            codes = y[1:, encoded.shape[1]: -1].copy()  # the last generated column is dropped (inference.py:839)
            assert (codes >= 0).all(), "Negative code found"
            yield GenerateResponse(action="sample", codes=codes, text=text)
        yield GenerateResponse(action="next")


class PrefixCache:
    """Reference-prefix K/V per voice: keyed by the prefix columns themselves, least recently
    used entry dropped beyond `capacity` (a 30 s reference at s1-mini shapes is ~82 MB of K/V)."""

    def __init__(self, capacity: int = 8, min_positions: int = 32):
        self.capacity, self.min_positions = capacity, min_positions
        self._entries = {}  # key -> KVPrefix, insertion order = recency

    def get(self, engine: ARHipEngine, prefix_cols: np.ndarray):
        import hashlib
        cols = np.ascontiguousarray(prefix_cols, dtype=np.int32)
        key = (id(engine), cols.shape[1], hashlib.sha1(cols.tobytes()).hexdigest())
        hit = self._entries.pop(key, None)
        if hit is None or not hit.handle:
            hit = engine.build_prefix(cols)
        self._entries[key] = hit
        while len(self._entries) > self.capacity:
            self._entries.pop(next(iter(self._entries))).free()
        return hit

    def clear(self) -> None:
        for pf in self._entries.values():
            pf.free()
        self._entries.clear()

    def __len__(self) -> int:
        return len(self._entries)
