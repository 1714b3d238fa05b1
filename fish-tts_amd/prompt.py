"""Prompt assembly for the dual-AR model: the host-side equivalent of ContentSequence /
encode_for_inference (fish_tts/models/inference.py:417-640) and of the prompt recipe in
generate_long (inference.py:779-793).  Pure integer work on the CPU; it defines the prefill input,
so it reproduces the reference's matrix exactly (pinned by tests/golden/prompt.npz)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from .tokenizer import IM_END_TOKEN, MODALITY_TOKENS


@dataclass
class TextPart:
    text: Optional[str] = None
    tokens: Optional[List[int]] = None

    def __post_init__(self):
        if self.text is None and self.tokens is None:
            raise ValueError("Either text or tokens must be provided")


@dataclass
class VQPart:
    codes: np.ndarray  # (num_codebooks, T)

    def __post_init__(self):
        self.codes = np.asarray(self.codes)


class ContentSequence:
    """Interleaved text / VQ parts; `modality` prepends its marker token (inference.py:492-499)."""

    def __init__(self, parts: Optional[Sequence] = None, modality: Optional[str] = None):
        self.parts: List[Union[TextPart, VQPart]] = list(parts or [])
        if modality and not (self.parts and isinstance(self.parts[0], TextPart) and self.parts[0].text is not None
                             and self.parts[0].text.startswith(MODALITY_TOKENS[modality])):
            self.parts.insert(0, TextPart(text=MODALITY_TOKENS[modality]))

    def append(self, parts, add_end: bool = False, speaker=None):
        if speaker is not None:
            self.parts.append(TextPart(text=f"<|speaker:{speaker}|>"))
        self.parts.extend(parts if isinstance(parts, list) else [parts])
        if add_end:
            self.parts.append(TextPart(text=IM_END_TOKEN))

    def encode_for_inference(self, tokenizer, num_codebooks: int) -> np.ndarray:
        """-> (num_codebooks + 1, L) int32: row 0 token ids (VQ positions: code0 + semantic_begin_id),
        rows 1.. the codes at VQ positions, 0 elsewhere (inference.py:611-640)."""
        cols: List[np.ndarray] = []
        for part in self.parts:
            if isinstance(part, TextPart):
                ids = part.tokens if part.tokens is not None else tokenizer.encode(part.text)
                blk = np.zeros((num_codebooks + 1, len(ids)), dtype=np.int32)
                blk[0] = np.asarray(ids, dtype=np.int32)
            elif isinstance(part, VQPart):
                codes = part.codes.astype(np.int32)
                blk = np.zeros((num_codebooks + 1, codes.shape[1]), dtype=np.int32)
                # the reference looks each code up in semantic_id_to_token_id (inference.py:553-559) and then
                # overwrites the row with code + semantic_begin_id (631-633): the same id for a valid code
                _ = [tokenizer.semantic_id_to_token_id[int(i)] for i in codes[0]]
                blk[0] = codes[0] + tokenizer.semantic_begin_id
                blk[1:] = codes[:num_codebooks]
            else:
                raise ValueError(f"Unsupported part type: {type(part)}")
            cols.append(blk)
        return np.concatenate(cols, axis=1) if cols else np.zeros((num_codebooks + 1, 0), dtype=np.int32)


def build_prompt_split(tokenizer, text: str, prompt_text: Optional[Sequence[str]], prompt_tokens: Optional[Sequence],
                       num_codebooks: int):
    """-> (matrix, n_prefix): the prompt of build_prompt and the number of leading columns that do not depend on
    `text` -- [<|interleave|>, (<|speaker:0|>, ref text, ref codes, <|im_end|>)*] -- i.e. the part whose K/V a
    voice can keep (0 without references)."""
    use_prompt = prompt_text is not None and prompt_tokens is not None
    if use_prompt and isinstance(prompt_text, str):
        prompt_text, prompt_tokens = [prompt_text], [prompt_tokens]
    if use_prompt:
        assert len(prompt_text) == len(prompt_tokens)
    seq = ContentSequence(modality="interleave")
    if use_prompt:
        for t, c in zip(prompt_text, prompt_tokens):
            c = c.cpu().numpy() if hasattr(c, "cpu") else np.asarray(c)
            seq.append([TextPart(text=t), VQPart(codes=c)], add_end=True, speaker=0)
    n_parts = len(seq.parts)
    has_refs = use_prompt and len(prompt_text) > 0
    seq.append([TextPart(text=text)], add_end=False, speaker=0)
    full = seq.encode_for_inference(tokenizer, num_codebooks)
    n_prefix = 0
    if has_refs:
        head = ContentSequence(seq.parts[:n_parts])
        n_prefix = head.encode_for_inference(tokenizer, num_codebooks).shape[1]
    return full, n_prefix


def build_prompt(tokenizer, text: str, prompt_text: Optional[Sequence[str]], prompt_tokens: Optional[Sequence],
                 num_codebooks: int) -> np.ndarray:
    """[<|interleave|>, (<|speaker:0|>, ref text, ref codes, <|im_end|>)*, <|speaker:0|>, text]
    (inference.py:767-793)."""
    return build_prompt_split(tokenizer, text, prompt_text, prompt_tokens, num_codebooks)[0]
