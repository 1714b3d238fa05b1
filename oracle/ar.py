"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference's dual-AR semantic-token path in torch CPU eager.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

It follows the reference's op order *and rounding points* so that, on the same
weights, it is bit-identical to the reference's CPU eager path (pinned by
tests/golden/*.npz, generated from the imported reference by
tests/golden/make_golden.py).

Reference files followed (all under /root/reference/fish_tts/models/):
  llama.py:126-149   KVCache                      -> _Cache
  llama.py:164-177   RMSNorm (fp32 norm, cast, *w) -> rms_norm
  llama.py:207-209   q/k nn.RMSNorm                -> F.rms_norm in attention()
  llama.py:180-190   SwiGLU                        -> mlp
  llama.py:229-309   Attention (SDPA / explicit)   -> attention
  llama.py:400-453   slow forward                  -> AROracle.slow_forward
  llama.py:561-580   fast forward                  -> AROracle.fast_forward
  llama.py:594-618   RoPE table (bf16) + rotation  -> rope_table / rope
  inference.py:24-80   sampling                    -> logits_to_probs / sample
  inference.py:83-155  one frame                   -> AROracle.decode_frame
  inference.py:158-276 frame loops                 -> AROracle.generate / generate_stream
  inference.py:281-384 generate                    -> AROracle.generate
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterator, List, Optional

import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel


@dataclass
class ARShape:
    """Field names follow config.json of the reference (llama.py:31-123)."""

    vocab_size: int = 155776
    n_layer: int = 28
    n_head: int = 16
    dim: int = 1024
    intermediate_size: int = 3072
    n_local_heads: int = 8
    head_dim: int = 128
    rope_base: float = 1e6
    norm_eps: float = 1e-6
    max_seq_len: int = 8192
    tie_word_embeddings: bool = True
    attention_qkv_bias: bool = False
    attention_o_bias: bool = False
    attention_qk_norm: bool = True
    codebook_size: int = 4096
    num_codebooks: int = 10
    scale_codebook_embeddings: bool = True
    n_fast_layer: int = 4
    fast_dim: int = 1024
    fast_n_head: int = 16
    fast_n_local_heads: int = 8
    fast_head_dim: int = 64
    fast_intermediate_size: int = 3072
    fast_attention_qkv_bias: bool = False
    fast_attention_qk_norm: bool = False
    fast_attention_o_bias: bool = False
    initializer_range: float = 0.02
    # token-id layout (tokenizer.py:83-101): ids the model math needs
    semantic_begin_id: int = 151658
    semantic_end_id: int = 155753
    im_end_id: int = 151647


def state_dict_keys(c: ARShape) -> Dict[str, tuple]:
    """Names/shapes of the reference's state dict (llama.py:349-359, 510-535)."""
    keys: Dict[str, tuple] = {}
    keys["embeddings.weight"] = (c.vocab_size, c.dim)
    keys["codebook_embeddings.weight"] = (c.codebook_size * c.num_codebooks, c.dim)

    def block(prefix, dim, nh, nkv, hd, ffn, qkv_bias, o_bias, qk_norm):
        tot = (nh + 2 * nkv) * hd
        keys[f"{prefix}.attention.wqkv.weight"] = (tot, dim)
        if qkv_bias:
            keys[f"{prefix}.attention.wqkv.bias"] = (tot,)
        keys[f"{prefix}.attention.wo.weight"] = (dim, nh * hd)
        if o_bias:
            keys[f"{prefix}.attention.wo.bias"] = (dim,)
        if qk_norm:
            keys[f"{prefix}.attention.q_norm.weight"] = (hd,)
            keys[f"{prefix}.attention.k_norm.weight"] = (hd,)
        keys[f"{prefix}.feed_forward.w1.weight"] = (ffn, dim)
        keys[f"{prefix}.feed_forward.w3.weight"] = (ffn, dim)
        keys[f"{prefix}.feed_forward.w2.weight"] = (dim, ffn)
        keys[f"{prefix}.ffn_norm.weight"] = (dim,)
        keys[f"{prefix}.attention_norm.weight"] = (dim,)

    for i in range(c.n_layer):
        block(f"layers.{i}", c.dim, c.n_head, c.n_local_heads, c.head_dim, c.intermediate_size,
              c.attention_qkv_bias, c.attention_o_bias, c.attention_qk_norm)
    keys["norm.weight"] = (c.dim,)
    if not c.tie_word_embeddings:
        keys["output.weight"] = (c.vocab_size, c.dim)
    if c.fast_dim != c.dim:
        keys["fast_project_in.weight"] = (c.fast_dim, c.dim)
        keys["fast_project_in.bias"] = (c.fast_dim,)
    keys["fast_embeddings.weight"] = (c.codebook_size, c.fast_dim)
    for i in range(c.n_fast_layer):
        block(f"fast_layers.{i}", c.fast_dim, c.fast_n_head, c.fast_n_local_heads, c.fast_head_dim,
              c.fast_intermediate_size, c.fast_attention_qkv_bias, c.fast_attention_o_bias,
              c.fast_attention_qk_norm)
    keys["fast_norm.weight"] = (c.fast_dim,)
    keys["fast_output.weight"] = (c.codebook_size, c.fast_dim)
    return keys


def random_weights(c: ARShape, seed: int = 0, dtype=torch.float32, std: Optional[float] = None, loud=None):
    """Seeded synthetic weights: normal(0, std) for matrices/embeddings, ones for norm
    gains (the reference's init rule, llama.py:455-464), generated in fp32 then cast.

    loud = (n, factor): n seeded rows of the vocabulary head inside the semantic id range and n of the codebook head's
    first min(1024, codebook_size) rows are scaled by factors spread over (factor, 2 factor].  With iid normal rows the top-1/top-2 gap of V logits is
    ~1/20 of their range whatever the scale (order statistics), i.e. about the size of a few bf16 steps; a few loud rows
    give the decisions the clear margins a trained model has, so that a bf16 parity test can judge nearly all of them."""
    g = torch.Generator().manual_seed(seed)
    std = c.initializer_range if std is None else std
    out = {}
    for k, shp in state_dict_keys(c).items():
        if k.endswith("norm.weight"):
            # perturb gains a little so that a dropped gain is visible in parity tests
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            w = std * torch.randn(shp, generator=g)
        out[k] = w
    if loud is not None:
        n, factor = int(loud[0]), float(loud[1])
        g2 = torch.Generator().manual_seed(seed + 7919)
        head = "embeddings.weight" if c.tie_word_embeddings else "output.weight"
        n_sem = c.semantic_end_id - c.semantic_begin_id + 1
        rows = c.semantic_begin_id + torch.randperm(n_sem, generator=g2)[:n]
        # (factor, 2 factor]: unequal loudness spreads the leaders apart
        scale = factor * (1.0 + torch.arange(n, dtype=torch.float32) / max(n, 1)).unsqueeze(1)
        out[head][rows] *= scale
        rows = torch.randperm(min(1024, c.codebook_size), generator=g2)[:n]
        out["fast_output.weight"][rows] *= scale
    return {k: w.to(dtype) for k, w in out.items()}


# ----------------------------------------------------------------------------- math

def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """llama.py:172-177: normalise in fp32, cast to x dtype, multiply by gain in x dtype."""
    xf = x.float()
    y = (xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)).type_as(x)
    return y * w


def rope_table(n_pos: int, n_elem: int, base: float) -> torch.Tensor:
    """llama.py:594-603: cos/sin in fp32, stored as bf16 for every model dtype."""
    inv = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(n_pos), inv)
    unit = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([unit.real, unit.imag], dim=-1).to(torch.bfloat16)


def rope(x: torch.Tensor, tab: torch.Tensor) -> torch.Tensor:
    """llama.py:606-618: interleaved pairs, fp32 rotation with the bf16 table, cast back.
    x: (B, S, H, hd); tab: (S, hd/2, 2)."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    t = tab.view(1, xs.size(1), 1, xs.size(3), 2)
    re = xs[..., 0] * t[..., 0] - xs[..., 1] * t[..., 1]
    im = xs[..., 1] * t[..., 0] + xs[..., 0] * t[..., 1]
    return torch.stack([re, im], dim=-1).flatten(3).type_as(x)


class _Cache:
    """llama.py:126-149."""

    def __init__(self, n_kv: int, n_slots: int, hd: int, dtype):
        self.k = torch.zeros(1, n_kv, n_slots, hd, dtype=dtype)
        self.v = torch.zeros(1, n_kv, n_slots, hd, dtype=dtype)

    def put(self, pos: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
        self.k[:, :, pos] = k
        self.v[:, :, pos] = v
        return self.k, self.v


def _explicit_attention(q, k, v, mask):
    """llama.py:285-309 (fast layers): every intermediate is rounded to the model dtype."""
    L, S = q.size(-2), k.size(-2)
    bias = torch.zeros(1, 1, L, S, dtype=q.dtype)
    bias.masked_fill_(mask.logical_not(), float("-inf"))
    w = q @ k.transpose(-2, -1) * (1 / math.sqrt(q.size(-1)))
    w += bias
    w = torch.softmax(w, dim=-1)
    w = torch.dropout(w, 0.0, train=True)
    return w @ v


# ----------------------------------------------------------------------------- sampling

def logits_to_probs(logits, temperature, top_p, repetition_penalty, previous_tokens=None):
    """inference.py:30-61.  `logits` is 1-D and is modified in place by the penalty."""
    if previous_tokens is not None:
        idx = previous_tokens.long()
        s = torch.gather(logits, dim=-1, index=idx)
        s = torch.where(s < 0, s * repetition_penalty, s / repetition_penalty)
        logits.scatter_(dim=-1, index=idx, src=s)
    srt, order = torch.sort(logits, descending=True)
    cum = torch.cumsum(F.softmax(srt, dim=-1), dim=-1)
    drop_sorted = cum > top_p
    drop_sorted[0] = False
    drop = drop_sorted.scatter(dim=-1, index=order, src=drop_sorted)
    logits = logits.masked_fill(drop, -float("Inf"))
    logits = logits / torch.clip(temperature, min=1e-5)
    return F.softmax(logits, dim=-1)


def sample(logits, temperature, top_p, repetition_penalty, previous_tokens=None,
           noise: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
    """inference.py:64-80 + 24-27.  `noise(probs)` may inject the Exp(1) draw (tests);
    by default it is drawn from torch's global CPU generator exactly like the reference."""
    probs = logits_to_probs(logits[0, -1], temperature, top_p, repetition_penalty, previous_tokens)
    q = torch.empty_like(probs).exponential_(1) if noise is None else noise(probs)
    return torch.argmax(probs / q, dim=-1, keepdim=True).to(dtype=torch.int), probs


# ----------------------------------------------------------------------------- model

class AROracle:
    """Functional restatement of DualARTransformer's generation path (batch 1)."""

    def __init__(self, shape: ARShape, weights: Dict[str, torch.Tensor], dtype=torch.float32):
        self.c = shape
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        c = shape
        self.tab = rope_table(c.max_seq_len, c.head_dim, c.rope_base)
        self.fast_tab = rope_table(c.num_codebooks, c.fast_head_dim, c.rope_base)
        self.tril = torch.tril(torch.ones(c.max_seq_len, c.max_seq_len, dtype=torch.bool))
        self.n_slots = c.max_seq_len + (-c.max_seq_len) % 8  # find_multiple(.., 8) llama.py:387
        self.reset()
        # taps for golden vectors / debugging
        self.last_logits: Optional[torch.Tensor] = None
        self.last_hidden: Optional[torch.Tensor] = None

    def reset(self):
        c = self.c
        self.slow_cache = [_Cache(c.n_local_heads, self.n_slots, c.head_dim, self.dtype)
                           for _ in range(c.n_layer)]
        self.fast_cache = [_Cache(c.fast_n_local_heads, c.num_codebooks, c.fast_head_dim, self.dtype)
                           for _ in range(c.n_fast_layer)]

    # -- blocks ---------------------------------------------------------------
    def _attention(self, pfx, x, tab, mask, pos, cache, nh, nkv, hd, qk_norm, sdpa):
        w = self.w
        B, S, _ = x.shape
        qkv = F.linear(x, w[f"{pfx}.wqkv.weight"], w.get(f"{pfx}.wqkv.bias"))
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
        q = q.view(B, S, nh, hd)
        k = k.view(B, S, nkv, hd)
        v = v.view(B, S, nkv, hd)
        if qk_norm:
            q = F.rms_norm(q, (hd,), w[f"{pfx}.q_norm.weight"], self.c.norm_eps)
            k = F.rms_norm(k, (hd,), w[f"{pfx}.k_norm.weight"], self.c.norm_eps)
        q = rope(q, tab)
        k = rope(k, tab)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))
        k, v = cache.put(pos, k, v)
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
        if sdpa:
            y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
        else:
            y = _explicit_attention(q, k, v, mask)
        y = y.transpose(1, 2).contiguous().view(B, S, nh * hd)
        return F.linear(y, w[f"{pfx}.wo.weight"], w.get(f"{pfx}.wo.bias"))

    def _mlp(self, pfx, x):
        w = self.w
        return F.linear(F.silu(F.linear(x, w[f"{pfx}.w1.weight"])) * F.linear(x, w[f"{pfx}.w3.weight"]),
                        w[f"{pfx}.w2.weight"])

    def _block(self, pfx, x, tab, mask, pos, cache, nh, nkv, hd, qk_norm, sdpa):
        eps = self.c.norm_eps
        h = x + self._attention(f"{pfx}.attention", rms_norm(x, self.w[f"{pfx}.attention_norm.weight"], eps),
                                tab, mask, pos, cache, nh, nkv, hd, qk_norm, sdpa)
        return h + self._mlp(f"{pfx}.feed_forward", rms_norm(h, self.w[f"{pfx}.ffn_norm.weight"], eps))

    # -- embeddings (llama.py:409-429) ----------------------------------------
    def embed(self, inp: torch.Tensor) -> torch.Tensor:
        c, w = self.c, self.w
        rows = [F.embedding(inp[:, i + 1] + i * c.codebook_size, w["codebook_embeddings.weight"])
                for i in range(c.num_codebooks)]
        vq = torch.stack(rows, dim=1).sum(dim=1)
        is_vq = (inp[:, 0] >= c.semantic_begin_id) & (inp[:, 0] <= c.semantic_end_id)
        vq[~is_vq] = 0
        x = F.embedding(inp[:, 0], w["embeddings.weight"]) + vq
        if c.scale_codebook_embeddings:
            x = torch.where(is_vq.unsqueeze(-1).expand_as(x), x / math.sqrt(c.num_codebooks + 1), x)
        return x

    # -- slow / fast forward --------------------------------------------------
    def slow_forward(self, inp: torch.Tensor, pos: torch.Tensor):
        """inp (1, ncb+1, S) int; pos (S,) long.  Returns logits (1,1,V), hidden (1,1,fast_dim)."""
        c, w = self.c, self.w
        x = self.embed(inp)
        mask = self.tril[None, None, pos, : self.n_slots]
        tab = self.tab[pos]
        for i in range(c.n_layer):
            x = self._block(f"layers.{i}", x, tab, mask, pos, self.slow_cache[i],
                            c.n_head, c.n_local_heads, c.head_dim, c.attention_qk_norm, True)
        if x.size(1) > 1:
            x = x[:, -1:]
        normed = rms_norm(x, w["norm.weight"], c.norm_eps)
        head = w["embeddings.weight"] if c.tie_word_embeddings else w["output.weight"]
        logits = F.linear(normed, head)
        hidden = x
        if c.fast_dim != c.dim:
            hidden = F.linear(hidden, w["fast_project_in.weight"], w["fast_project_in.bias"])
        return logits, hidden

    def fast_forward(self, x: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        c, w = self.c, self.w
        x = x.view(x.shape[0], 1, -1)
        mask = self.tril[None, None, pos, : c.num_codebooks]
        tab = self.fast_tab[pos]
        for i in range(c.n_fast_layer):
            x = self._block(f"fast_layers.{i}", x, tab, mask, pos, self.fast_cache[i],
                            c.fast_n_head, c.fast_n_local_heads, c.fast_head_dim,
                            c.fast_attention_qk_norm, False)
        return F.linear(rms_norm(x, w["fast_norm.weight"], c.norm_eps), w["fast_output.weight"])

    # -- one frame (inference.py:83-155) --------------------------------------
    def decode_frame(self, x, pos, temperature, top_p, repetition_penalty, window=None, noise=None):
        c = self.c
        logits, hidden = self.slow_forward(x, pos)
        self.last_logits = logits.detach().clone()
        self.last_hidden = hidden.detach().clone()
        out = [sample(logits, temperature, top_p, repetition_penalty,
                      window[:, 0] if window is not None else None, noise)[0]]
        for cache in self.fast_cache:
            cache.k.fill_(0)
            cache.v.fill_(0)
        self.fast_forward(hidden, torch.tensor([0], dtype=torch.long))
        a = out[0] - c.semantic_begin_id
        a[a < 0] = 0
        hidden = F.embedding(a, self.w["fast_embeddings.weight"])
        out.append(a)
        self.last_fast_logits = []
        for cb in range(1, c.num_codebooks):
            fl = self.fast_forward(hidden, torch.tensor([cb], dtype=torch.long))
            short = fl[:, :, :1024]
            self.last_fast_logits.append(short.detach().clone())
            a = sample(short, temperature, top_p, repetition_penalty,
                       window[cb + 1] if window is not None else None, noise)[0]
            hidden = F.embedding(a, self.w["fast_embeddings.weight"])
            out.append(a)
        return torch.stack(out, dim=1).T  # (ncb+1, 1) int32

    # -- loops (inference.py:158-276, 281-384, 645-738) ------------------------
    def _frames(self, first, T, n_more, temperature, top_p, repetition_penalty, noise, on_frame=None):
        c = self.c
        prev = torch.zeros((c.num_codebooks + 1, c.max_seq_len), dtype=torch.int)
        cur = first.view(1, c.num_codebooks + 1, -1)
        pos = torch.tensor([T], dtype=torch.int)
        n = 0
        for i in range(n_more):
            window = prev[:, :16] if i < 16 else prev[:, i - 16: i]
            with sdpa_kernel(SDPBackend.MATH):
                nxt = self.decode_frame(cur, pos, temperature, top_p, repetition_penalty, window, noise).clone()
            pos += 1
            cur = nxt.view(1, c.num_codebooks + 1, -1)
            prev[:, i: i + 1] = nxt.view(c.num_codebooks + 1, -1)
            n = i + 1
            if on_frame is not None:
                on_frame(nxt)
            if cur[0, 0, -1] == c.im_end_id:
                break
        return prev[:, :n]

    @torch.inference_mode()
    def generate(self, prompt: torch.Tensor, max_new_tokens: int, temperature=0.7, top_p=0.7,
                 repetition_penalty=1.5, noise=None, frame_taps: Optional[list] = None) -> torch.Tensor:
        """inference.py:281-384.  prompt (ncb+1, T) int32 -> (ncb+1, T+n) int32."""
        c = self.c
        T = prompt.size(1)
        if T >= c.max_seq_len:
            raise ValueError(f"Input sequence length {T} exceeds max_seq_len {c.max_seq_len}")
        if max_new_tokens:
            if T + max_new_tokens > c.max_seq_len:
                max_new_tokens = c.max_seq_len - T
        else:
            max_new_tokens = c.max_seq_len - T
        t = torch.tensor(temperature, dtype=torch.float)
        p = torch.tensor(top_p, dtype=torch.float)
        r = torch.tensor(repetition_penalty, dtype=torch.float)
        first = self.decode_frame(prompt.view(1, c.num_codebooks + 1, -1), torch.arange(0, T, dtype=torch.long),
                                  t, p, r, None, noise)
        if frame_taps is not None:
            frame_taps.append((self.last_logits, self.last_hidden, self.last_fast_logits))

        def tap(_):
            if frame_taps is not None:
                frame_taps.append((self.last_logits, self.last_hidden, self.last_fast_logits))

        rest = self._frames(first, T, max_new_tokens - 1, t, p, r, noise, tap)
        return torch.cat([prompt.to(torch.int), first, rest], dim=1)

    @torch.inference_mode()
    def generate_stream(self, prompt: torch.Tensor, max_new_tokens: int, temperature=0.7, top_p=0.7,
                        repetition_penalty=1.5, noise=None) -> List[torch.Tensor]:
        """inference.py:645-738: list of (ncb, 1) code columns, EOS frame included."""
        c = self.c
        T = prompt.size(1)
        if T >= c.max_seq_len:
            raise ValueError(f"Input sequence length {T} exceeds max_seq_len {c.max_seq_len}")
        if max_new_tokens:
            if T + max_new_tokens > c.max_seq_len:
                max_new_tokens = c.max_seq_len - T
        else:
            max_new_tokens = c.max_seq_len - T
        t = torch.tensor(temperature, dtype=torch.float)
        p = torch.tensor(top_p, dtype=torch.float)
        r = torch.tensor(repetition_penalty, dtype=torch.float)
        first = self.decode_frame(prompt.view(1, c.num_codebooks + 1, -1), torch.arange(0, T, dtype=torch.long),
                                  t, p, r, None, noise)
        cols = [first[1:, :].clone()]
        self._frames(first, T, max_new_tokens - 1, t, p, r, noise, lambda f: cols.append(f[1:, :].clone()))
        return cols
