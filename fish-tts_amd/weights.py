"""Synthetic weights and checkpoint ingestion for the dual-AR model.

random_state_dict follows the reference's init rule (llama.py:455-464: normal(0, initializer_range)
for Linear and Embedding weights, zero biases, ones for RMSNorm gains) under the reference's
state-dict names; load_checkpoint mirrors BaseTransformer.from_pretrained (llama.py:466-500)."""
from __future__ import annotations

from pathlib import Path
from typing import Dict

import torch

from .config import DualARModelArgs


def state_dict_shapes(a: DualARModelArgs) -> Dict[str, tuple]:
    out: Dict[str, tuple] = {"embeddings.weight": (a.vocab_size, a.dim),
                             "codebook_embeddings.weight": (a.codebook_size * a.num_codebooks, a.dim)}

    def block(p, dim, nh, nkv, hd, ffn, qkv_bias, o_bias, qk_norm):
        tot = (nh + 2 * nkv) * hd
        out[f"{p}.attention.wqkv.weight"] = (tot, dim)
        if qkv_bias:
            out[f"{p}.attention.wqkv.bias"] = (tot,)
        out[f"{p}.attention.wo.weight"] = (dim, nh * hd)
        if o_bias:
            out[f"{p}.attention.wo.bias"] = (dim,)
        if qk_norm:
            out[f"{p}.attention.q_norm.weight"] = (hd,)
            out[f"{p}.attention.k_norm.weight"] = (hd,)
        out[f"{p}.feed_forward.w1.weight"] = (ffn, dim)
        out[f"{p}.feed_forward.w3.weight"] = (ffn, dim)
        out[f"{p}.feed_forward.w2.weight"] = (dim, ffn)
        out[f"{p}.ffn_norm.weight"] = (dim,)
        out[f"{p}.attention_norm.weight"] = (dim,)

    for i in range(a.n_layer):
        block(f"layers.{i}", a.dim, a.n_head, a.n_local_heads, a.head_dim, a.intermediate_size,
              a.attention_qkv_bias, a.attention_o_bias, a.attention_qk_norm)
    out["norm.weight"] = (a.dim,)
    if not a.tie_word_embeddings:
        out["output.weight"] = (a.vocab_size, a.dim)
    if a.fast_dim != a.dim:
        out["fast_project_in.weight"] = (a.fast_dim, a.dim)
        out["fast_project_in.bias"] = (a.fast_dim,)
    out["fast_embeddings.weight"] = (a.codebook_size, a.fast_dim)
    for i in range(a.n_fast_layer):
        block(f"fast_layers.{i}", a.fast_dim, a.fast_n_head, a.fast_n_local_heads, a.fast_head_dim,
              a.fast_intermediate_size, a.fast_attention_qkv_bias, a.fast_attention_o_bias,
              a.fast_attention_qk_norm)
    out["fast_norm.weight"] = (a.fast_dim,)
    out["fast_output.weight"] = (a.codebook_size, a.fast_dim)
    return out


def random_state_dict(a: DualARModelArgs, seed: int = 0, dtype=torch.bfloat16, std=None) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    std = a.initializer_range if std is None else std
    sd = {}
    for k, shp in state_dict_shapes(a).items():
        if k.endswith("norm.weight"):
            sd[k] = torch.ones(shp, dtype=dtype)
        elif k.endswith(".bias"):
            sd[k] = torch.zeros(shp, dtype=dtype)
        else:
            sd[k] = torch.empty(shp, dtype=torch.float32).normal_(0.0, std, generator=g).to(dtype)
    return sd


def load_checkpoint(model_dir) -> Dict[str, torch.Tensor]:
    """model.pth of the reference layout: mmap'd, weights_only (llama.py:476-482)."""
    sd = torch.load(Path(model_dir) / "model.pth", map_location="cpu", mmap=True, weights_only=True)
    return sd["state_dict"] if "state_dict" in sd else sd
