"""Host driver of the HIP dual-AR path: the Python side of the S2 seam —
init_model / generate / generate_streaming of fish_tts/models/inference.py:281-414,645-738 — on top
of the C ABI (include/fishtts_hip.h).  torch is used only to hold weight tensors."""
from __future__ import annotations

import ctypes as C
import logging
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .config import DualARModelArgs


class HipError(RuntimeError):
    pass


logger = logging.getLogger(__name__)


def _rope_table(n_pos: int, n_elem: int, base: float) -> torch.Tensor:
    """cos/sin pairs rounded to bf16, as the reference builds them (llama.py:594-603), returned as f32."""
    expo = torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem
    inv_freq = 1.0 / (base ** expo)
    ang = torch.outer(torch.arange(n_pos), inv_freq)
    z = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([z.real, z.imag], dim=-1).to(torch.bfloat16).float().contiguous()


def normalise_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Key layout handling of llama.py:484-498 and the wq/wk/wv -> wqkv fuse of llama.py:222-227."""
    if "state_dict" in sd:
        sd = sd["state_dict"]
    if next(iter(sd.keys())).startswith("model."):
        sd = {k.replace("model.", ""): v for k, v in sd.items()}
    sd = {k: v for k, v in sd.items() if "audio_" not in k}
    out = dict(sd)
    for k in list(sd.keys()):
        if k.endswith("attention.wq.weight"):
            pre = k[: -len("wq.weight")]
            out[pre + "wqkv.weight"] = torch.cat([out.pop(pre + "wq.weight"), out.pop(pre + "wk.weight"),
                                                  out.pop(pre + "wv.weight")])
    return out


def contiguous_runs(slots: Sequence[int]) -> list:
    """Indices into `slots`, grouped into runs of consecutive slot numbers in ascending slot order: [5, 2, 3, 9] ->
    [[1, 2], [0], [3]] (ft_ar_first_frames draws the first frames of a contiguous slot range in one lock-step pass)."""
    order = sorted(range(len(slots)), key=lambda i: slots[i])
    runs, a = [], 0
    while a < len(order):
        b = a + 1
        while b < len(order) and slots[order[b]] == slots[order[b - 1]] + 1:
            b += 1
        runs.append(order[a:b])
        a = b
    return runs


class ARHipEngine:
    """One GPU context holding the dual-AR weights, KV caches and the captured frame graph."""

    def __init__(self, args: DualARModelArgs, semantic_begin_id: int, semantic_end_id: int, im_end_id: int,
                 precision: str = "bf16", device: int = 0, max_batch: int = 1, max_new_tokens: int = 2048,
                 codec_cfg: Optional[L.ft_codec_config] = None):
        if precision not in ("bf16", "fp16", "fp32"):
            raise ValueError(f"precision {precision!r}: expected 'bf16', 'fp16' or 'fp32' (synthesizer.py:122-128)")
        self.args = args
        self.precision = precision
        self.lib = L.load()
        c = L.ft_ar_config()
        c.dtype = {"bf16": L.FT_BF16, "fp16": L.FT_F16, "fp32": L.FT_F32}[precision]
        for name in ("vocab_size", "n_layer", "n_head", "dim", "intermediate_size", "n_local_heads", "head_dim",
                     "max_seq_len", "codebook_size", "num_codebooks", "n_fast_layer", "fast_dim", "fast_n_head",
                     "fast_n_local_heads", "fast_head_dim", "fast_intermediate_size"):
            setattr(c, name, int(getattr(args, name)))
        for name in ("tie_word_embeddings", "attention_qkv_bias", "attention_o_bias", "attention_qk_norm",
                     "scale_codebook_embeddings", "fast_attention_qkv_bias", "fast_attention_qk_norm",
                     "fast_attention_o_bias"):
            setattr(c, name, 1 if getattr(args, name) else 0)
        c.rope_base = float(args.rope_base)
        c.norm_eps = float(args.norm_eps)
        c.semantic_begin_id, c.semantic_end_id, c.im_end_id = int(semantic_begin_id), int(semantic_end_id), int(im_end_id)
        c.max_batch, c.max_new_tokens = int(max_batch), int(max_new_tokens)
        self.cfg = c
        self.R = args.num_codebooks + 1
        self.max_batch = max_batch
        self.max_new_tokens = max_new_tokens
        self.im_end_id = im_end_id
        self._h = C.c_void_p()
        st = self.lib.ft_create(C.byref(c), C.byref(codec_cfg) if codec_cfg is not None else None, device,
                                C.byref(self._h))
        if st != L.FT_OK:
            raise HipError(f"ft_create failed ({st}): {self.lib.ft_last_error(None).decode()}")
        self._loaded = False

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            for pf in list(getattr(self, "_prefixes", [])):
                pf.free()
            self.lib.ft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int, what: str):
        if st == L.FT_OK:
            return
        msg = self.lib.ft_last_error(self._h).decode()
        if st == L.FT_ERR_TOO_LONG:
            raise ValueError(msg)  # inference.py:296-299
        raise HipError(f"{what} failed ({st}): {msg}")

    # ------------------------------------------------------------------ weights
    def load_tensor(self, name: str, t: torch.Tensor, strict: bool = True) -> bool:
        """Copies one tensor into the library's HBM.  strict=False mirrors the reference's
        load_state_dict(strict=False, assign=True) (llama.py:498) for EXTRA keys only: a name the model does not
        have is skipped (returns False); a wrong shape or rank still raises."""
        t = t.detach()
        if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            t = t.float()
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        dt = {torch.float32: L.FT_F32, torch.bfloat16: L.FT_BF16, torch.float16: L.FT_F16}[t.dtype]
        st = self.lib.ft_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), dt, shape, t.dim())
        if not strict and st == L.FT_ERR_ARG and self.lib.ft_last_error(self._h).startswith(b"unknown weight name"):
            logger.debug("checkpoint key %s is not a weight of this model: skipped", name)
            return False
        self._check(st, f"ft_load_weight({name})")
        return True

    def load_state_dict(self, sd: Dict[str, torch.Tensor], finalize: bool = True):
        a = self.args
        sd = normalise_state_dict(sd)
        for k, v in sd.items():
            if k.endswith(("freqs_cis", "causal_mask")) or "kv_cache" in k:
                continue
            # extras (an `output.weight` saved beside tied embeddings, `fast_project_in.*` when fast_dim == dim,
            # training-only tensors) are ignored like the reference's strict=False load; a MISSING weight still
            # fails in ft_finalize_weights, a mis-shaped one here
            self.load_tensor(k, v, strict=False)
        self.load_tensor("rope.slow", _rope_table(a.max_seq_len, a.head_dim, a.rope_base))
        self.load_tensor("rope.fast", _rope_table(a.num_codebooks, a.fast_head_dim, a.rope_base))
        if finalize:
            self.finalize()

    def finalize(self):
        self._check(self.lib.ft_finalize_weights(self._h), "ft_finalize_weights")
        self._loaded = True
        logger.info("batch-1 decode frames: %s", self.frame_path())

    def frame_path(self) -> str:
        """Which path the batch-1 frames take (persistent frame engine or launches) and why."""
        return self.lib.ft_ar_frame_path(self._h).decode()

    def inject_engine_fault(self, which: int = 0, workgroup: int = 0, skip: int = 0) -> None:
        """Test hook: a coming slow-stack (0) / codebook-loop (1) engine launch loses one workgroup's rows and times out;
        `skip` launches of that kind pass first."""
        self._check(self.lib.ft_test_engine_fault(self._h, which, workgroup, skip), "ft_test_engine_fault")

    # ------------------------------------------------------------------ primitives
    @staticmethod
    def _sampling(temperature, top_p, repetition_penalty, seed=0, ban_eos=False) -> L.ft_sampling:
        s = L.ft_sampling()
        s.temperature, s.top_p, s.repetition_penalty = float(temperature), float(top_p), float(repetition_penalty)
        s.seed, s.ban_eos = int(seed) & (2 ** 64 - 1), 1 if ban_eos else 0
        return s

    def prefill(self, prompt: np.ndarray, sampling: L.ft_sampling, slot: int = 0, pos0: int = 0) -> np.ndarray:
        """Feeds the (R, Lp) prompt at cache positions [pos0, pos0 + Lp); returns the first generated frame.
        pos0 > 0 continues a restored prefix (kv_restore)."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        assert prompt.ndim == 2 and prompt.shape[0] == self.R, prompt.shape
        out = np.zeros(self.R, dtype=np.int32)
        self._check(self.lib.ft_ar_prefill_at(self._h, slot, prompt.ctypes.data_as(C.c_void_p), prompt.shape[1], pos0,
                                              C.byref(sampling), out.ctypes.data_as(C.c_void_p)), "ft_ar_prefill")
        return out

    def prefill_many(self, prompts: Sequence[np.ndarray], samplings: Sequence[L.ft_sampling], slot0=0,
                     prefixes: Optional[Sequence[Optional["KVPrefix"]]] = None) -> np.ndarray:
        """Prompts of several slots - slot0 = the first of a contiguous range, or the list of (distinct) slots, e.g. the
        ones a scheduler found finished after a burst: the prompt passes in one call (from 5 prompts in bf16 as the rows
        of ONE pass through the slow stack, ft_ar_prefill_slow_many), then the first frames in one lock-step pass per
        contiguous run of slots.  Returns (n, R) first frames in the order of `prompts`."""
        n = len(prompts)
        slots = list(range(slot0, slot0 + n)) if isinstance(slot0, (int, np.integer)) else [int(s) for s in slot0]
        assert len(slots) == n and len(set(slots)) == n, slots
        tails, lps, pos0s = [], np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        for i, prompt in enumerate(prompts):
            prompt = np.ascontiguousarray(prompt, dtype=np.int32)
            assert prompt.ndim == 2 and prompt.shape[0] == self.R, prompt.shape
            pf = prefixes[i] if prefixes is not None else None
            if pf is not None:
                assert 0 < pf.n_pos < prompt.shape[1], (pf.n_pos, prompt.shape)
                self.kv_restore(pf, slots[i])
                pos0s[i], prompt = pf.n_pos, np.ascontiguousarray(prompt[:, pf.n_pos:])
            tails.append(prompt.reshape(-1))
            lps[i] = prompt.shape[1]
        packed = np.ascontiguousarray(np.concatenate(tails)) if n else np.zeros(0, dtype=np.int32)
        slot_arr = np.asarray(slots, dtype=np.int32)
        if n:
            self._check(self.lib.ft_ar_prefill_slow_many(self._h, n, slot_arr.ctypes.data_as(C.c_void_p),
                                                         packed.ctypes.data_as(C.c_void_p), lps.ctypes.data_as(C.c_void_p),
                                                         pos0s.ctypes.data_as(C.c_void_p)), "ft_ar_prefill")
        next_pos = (pos0s + lps).astype(np.int32)
        out = np.zeros((n, self.R), dtype=np.int32)
        for idx in contiguous_runs(slots):             # one lock-step pass per contiguous run of slots
            arr = (L.ft_sampling * len(idx))(*[samplings[i] for i in idx])
            npos = np.ascontiguousarray(next_pos[idx])
            run = np.zeros((len(idx), self.R), dtype=np.int32)
            self._check(self.lib.ft_ar_first_frames(self._h, slots[idx[0]], len(idx), arr, npos.ctypes.data_as(C.c_void_p),
                                                    run.ctypes.data_as(C.c_void_p)), "ft_ar_first_frames")
            out[idx] = run
        return out

    def park(self, slot: int) -> None:
        """Marks a slot idle (finished) for lock-step decoding until its next prefill."""
        self._check(self.lib.ft_ar_park(self._h, slot), "ft_ar_park")

    # ---- reference-prefix K/V reuse
    def kv_save(self, n_pos: int, slot: int = 0) -> "KVPrefix":
        h = C.c_void_p()
        self._check(self.lib.ft_ar_kv_save(self._h, slot, n_pos, C.byref(h)), "ft_ar_kv_save")
        return KVPrefix(self, h, n_pos)

    def kv_restore(self, prefix: "KVPrefix", slot: int = 0) -> None:
        if prefix.engine is not self or not prefix.handle:
            raise ValueError("KV prefix belongs to another engine or was freed")
        self._check(self.lib.ft_ar_kv_restore(self._h, prefix.handle, slot), "ft_ar_kv_restore")

    def build_prefix(self, prefix_cols: np.ndarray, slot: int = 0) -> "KVPrefix":
        """K/V of a prompt prefix (its own prefill; the sampled frame is discarded)."""
        self.prefill(prefix_cols, self._sampling(0.7, 0.7, 1.0), slot)
        return self.kv_save(prefix_cols.shape[1], slot)

    def _start(self, prompt: np.ndarray, sp, prefix: Optional["KVPrefix"], slot: int = 0) -> np.ndarray:
        if prefix is None:
            return self.prefill(prompt, sp, slot)
        assert 0 < prefix.n_pos < prompt.shape[1], (prefix.n_pos, prompt.shape)
        self.kv_restore(prefix, slot)
        return self.prefill(prompt[:, prefix.n_pos:], sp, slot, pos0=prefix.n_pos)

    def decode(self, n_frames: int, samplings: Sequence[L.ft_sampling], poll: int = 8):
        ns = len(samplings)
        arr = (L.ft_sampling * ns)(*samplings)
        frames = np.zeros((ns, max(n_frames, 1), self.R), dtype=np.int32)
        n = np.zeros(ns, dtype=np.int32)
        self._check(self.lib.ft_ar_decode(self._h, ns, n_frames, arr, poll, frames.ctypes.data_as(C.c_void_p),
                                          n.ctypes.data_as(C.c_void_p)), "ft_ar_decode")
        return frames, n

    def set_noise(self, q: Optional[np.ndarray]):
        if q is None:
            self._check(self.lib.ft_ar_set_noise(self._h, None, 0, 0), "ft_ar_set_noise")
            return
        q = np.ascontiguousarray(q, dtype=np.float32)
        self._check(self.lib.ft_ar_set_noise(self._h, q.ctypes.data_as(C.c_void_p), q.shape[0], q.shape[1]),
                    "ft_ar_set_noise")

    def debug_state(self, slot: int = 0):
        logits = np.zeros(self.args.vocab_size, dtype=np.float32)
        hidden = np.zeros(self.args.fast_dim, dtype=np.float32)
        self._check(self.lib.ft_ar_get_debug(self._h, slot, logits.ctypes.data_as(C.c_void_p),
                                             hidden.ctypes.data_as(C.c_void_p)), "ft_ar_get_debug")
        return logits, hidden

    def test_sample(self, logits: np.ndarray, cb: int, sampling: L.ft_sampling, window=None, q=None) -> int:
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        w = None if window is None else np.ascontiguousarray(window, dtype=np.int32)
        qq = None if q is None else np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros(1, dtype=np.int32)
        self._check(self.lib.ft_test_sample(
            self._h, logits.ctypes.data_as(C.c_void_p), cb, C.byref(sampling),
            None if w is None else w.ctypes.data_as(C.c_void_p),
            None if qq is None else qq.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "ft_test_sample")
        return int(out[0])

    def engine_state(self):
        """(flags, time-outs recovered so far, phase of the last one) of the persistent frame engine: flags bit 0 = slow
        stack, bit 1 = fast loop."""
        f, a, w = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self.lib.ft_ar_engine_state(self._h, C.byref(f), C.byref(a), C.byref(w)), "ft_ar_engine_state")
        return f.value, a.value, w.value

    def sync(self):
        self._check(self.lib.ft_sync(self._h), "ft_sync")

    def profile_gemv(self, frames: int, sampling: L.ft_sampling):
        ms, n, b = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self.lib.ft_ar_profile_gemv(self._h, frames, C.byref(sampling), C.byref(ms), C.byref(n),
                                                C.byref(b)), "ft_ar_profile_gemv")
        return ms.value, n.value, b.value

    # ------------------------------------------------------------------ generate (inference.py:281-384)
    def profile_frame(self, frames: int, sampling: L.ft_sampling):
        """(ms of `frames` graph replays, [ms slow stack, ms head + draw, ms fast loop] over frames-1 eager frames,
        launches per captured frame) - slot 0 must be prefilled."""
        ms, n = C.c_double(0), C.c_int32(0)
        seg = (C.c_double * 3)()
        self._check(self.lib.ft_ar_profile_frame(self._h, frames, C.byref(sampling), C.byref(ms), seg, C.byref(n)),
                    "ft_ar_profile_frame")
        return ms.value, [seg[0], seg[1], seg[2]], n.value

    def _clamp_new(self, T: int, max_new_tokens: int) -> int:
        m = self.args.max_seq_len
        if T >= m:
            raise ValueError(f"Input sequence length {T} exceeds max_seq_len {m}")
        if max_new_tokens:
            if T + max_new_tokens > m:
                max_new_tokens = m - T
        else:
            max_new_tokens = m - T
        return min(max_new_tokens, self.max_new_tokens)

    def generate(self, prompt: np.ndarray, max_new_tokens: int, temperature: float = 0.7, top_p: float = 0.7,
                 repetition_penalty: float = 1.5, seed: int = 0, ban_eos: bool = False, poll: int = 8,
                 prefix: Optional["KVPrefix"] = None) -> np.ndarray:
        """(R, T) int32 prompt -> (R, T + n) int32, n <= max_new_tokens, stopping after <|im_end|>.
        `prefix`: saved K/V of the first prefix.n_pos prompt columns (only the rest is prefilled)."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        T = prompt.shape[1]
        n_new = self._clamp_new(T, max_new_tokens)
        sp = self._sampling(temperature, top_p, repetition_penalty, seed, ban_eos)
        first = self._start(prompt, sp, prefix)
        frames, n = self.decode(n_new - 1, [sp], poll)
        return np.concatenate([prompt, first[:, None], frames[0, : n[0]].T], axis=1)

    def generate_streaming(self, prompt: np.ndarray, max_new_tokens: int, temperature: float = 0.7,
                           top_p: float = 0.7, repetition_penalty: float = 1.5, seed: int = 0,
                           ban_eos: bool = False, chunk: int = 8,
                           prefix: Optional["KVPrefix"] = None) -> Iterator[np.ndarray]:
        """Yields (num_codebooks, k) code blocks as they are produced, <|im_end|> frame included
        (inference.py:645-738, 218-276); `chunk` frames per graph burst."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        n_new = self._clamp_new(prompt.shape[1], max_new_tokens)
        sp = self._sampling(temperature, top_p, repetition_penalty, seed, ban_eos)
        first = self._start(prompt, sp, prefix)
        yield first[1:, None]
        left = n_new - 1
        while left > 0:
            k = min(chunk, left)
            frames, n = self.decode(k, [sp], poll=k)
            if n[0] > 0:
                yield frames[0, : n[0], 1:].T
            left -= k
            if n[0] < k or (n[0] > 0 and frames[0, n[0] - 1, 0] == self.im_end_id):
                break


class KVPrefix:
    """Device-resident K/V of the first n_pos prompt positions (one per voice); freed with the engine."""

    def __init__(self, engine: ARHipEngine, handle, n_pos: int):
        self.engine, self.handle, self.n_pos = engine, handle, n_pos
        if not hasattr(engine, "_prefixes"):
            engine._prefixes = []
        engine._prefixes.append(self)

    def free(self) -> None:
        if self.handle:
            self.engine.lib.ft_ar_kv_free(self.engine._h, self.handle)
            self.handle = None
        if self in getattr(self.engine, "_prefixes", []):
            self.engine._prefixes.remove(self)
