"""Helpers shared by the GPU parity tests: build a HIP engine and an oracle on the same seeded weights."""
import numpy as np
import torch

from oracle import ar as O


def args_from_shape(shape: O.ARShape):
    from fish_tts_amd.config import DualARModelArgs
    names = set(DualARModelArgs.__dataclass_fields__)
    return DualARModelArgs(**{k: v for k, v in shape.__dict__.items() if k in names})


_WEIGHTS = {}


def cached_random_weights(shape: O.ARShape, **kw):
    """oracle.ar.random_weights, generated once per (widths, seed, std, dtype, loud rows) and test run: the 700 M seeded
    parameters of the full-depth fixtures take seconds to draw and nine GPU tests share two sets of them.  (The weights
    do not depend on max_seq_len; callers must not modify the tensors.)"""
    dims = tuple(sorted((k, str(v)) for k, v in shape.__dict__.items() if k != "max_seq_len"))
    key = (dims, tuple(sorted((k, str(v)) for k, v in kw.items())))
    if key not in _WEIGHTS:
        _WEIGHTS[key] = O.random_weights(shape, **kw)
    return _WEIGHTS[key]


def make_pair(shape: O.ARShape, precision: str, seed: int = 0, max_batch: int = 1, max_new_tokens: int = 64,
              std=None):
    from fish_tts_amd.ar_engine import ARHipEngine
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(precision, torch.float32)
    w = O.random_weights(shape, seed=seed, std=std)
    orc = O.AROracle(shape, w, dtype)
    eng = ARHipEngine(args_from_shape(shape), shape.semantic_begin_id, shape.semantic_end_id, shape.im_end_id,
                      precision=precision, device=0, max_batch=max_batch, max_new_tokens=max_new_tokens)
    eng.load_state_dict({k: v.to(dtype) for k, v in w.items()})
    return eng, orc


class NoiseTape:
    """Exp(1) draws shared by the oracle (as a callable) and the HIP sampler (as a table)."""

    def __init__(self, shape: O.ARShape, n_frames: int, seed: int):
        self.V = shape.vocab_size
        self.fastV = min(1024, shape.codebook_size)
        self.ncb = shape.num_codebooks
        self.row_len = self.V + (self.ncb - 1) * self.fastV
        g = torch.Generator().manual_seed(seed)
        self.q = torch.empty(n_frames, self.row_len).exponential_(1.0, generator=g).clamp_min_(1e-6)
        self.calls = 0

    def __call__(self, probs: torch.Tensor) -> torch.Tensor:
        f, k = divmod(self.calls, self.ncb)
        self.calls += 1
        off = 0 if k == 0 else self.V + (k - 1) * self.fastV
        n = probs.shape[-1]
        return self.q[f, off: off + n].to(probs.dtype)

    def table(self) -> np.ndarray:
        return self.q.numpy()


def first_divergence(a: np.ndarray, b: np.ndarray):
    """(col, row) of the first differing entry in column-major (frame, then codebook) order, or None."""
    n = min(a.shape[1], b.shape[1])
    for col in range(n):
        for row in range(a.shape[0]):
            if a[row, col] != b[row, col]:
                return col, row
    if a.shape[1] != b.shape[1]:
        return n, 0
    return None
