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


def codec_state_dict_shapes(a) -> Dict[str, tuple]:
    """Folded decode-path tensors of the DAC codec under the reference's module paths
    (vocoder.py Decoder 605-640, DownsampleResidualVectorQuantize 683-757, synthesizer.py:199-269)."""
    D = a.latent_dim
    H = a.tf_n_head * a.tf_head_dim
    s: Dict[str, tuple] = {}
    s["quantizer.semantic_quantizer.quantizers.0.codebook.weight"] = (a.semantic_codebook_size, a.codebook_dim)
    s["quantizer.semantic_quantizer.quantizers.0.out_proj.weight"] = (D, a.codebook_dim, 1)
    s["quantizer.semantic_quantizer.quantizers.0.out_proj.bias"] = (D,)
    for i in range(a.n_codebooks):
        s[f"quantizer.quantizer.quantizers.{i}.codebook.weight"] = (a.codebook_size, a.codebook_dim)
        s[f"quantizer.quantizer.quantizers.{i}.out_proj.weight"] = (D, a.codebook_dim, 1)
        s[f"quantizer.quantizer.quantizers.{i}.out_proj.bias"] = (D,)
    for l in range(a.n_tf_layer):
        p = f"quantizer.post_module.layers.{l}"
        s[f"{p}.attention.wqkv.weight"] = (3 * H, D)
        s[f"{p}.attention.wo.weight"] = (D, H)
        s[f"{p}.feed_forward.w1.weight"] = (a.tf_ffn, D)
        s[f"{p}.feed_forward.w3.weight"] = (a.tf_ffn, D)
        s[f"{p}.feed_forward.w2.weight"] = (D, a.tf_ffn)
        for n in ("ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma", "ffn_layer_scale.gamma"):
            s[f"{p}.{n}"] = (D,)
    s["quantizer.post_module.norm.weight"] = (D,)
    for j, f in enumerate(a.downsample_factor):
        p = f"quantizer.upsample.{j}"
        s[f"{p}.0.conv.weight"] = (D, D, f)
        s[f"{p}.0.conv.bias"] = (D,)
        s[f"{p}.1.dwconv.conv.weight"] = (D, 1, 7)
        s[f"{p}.1.dwconv.conv.bias"] = (D,)
        s[f"{p}.1.norm.weight"] = (D,)
        s[f"{p}.1.norm.bias"] = (D,)
        s[f"{p}.1.pwconv1.weight"] = (4 * D, D)
        s[f"{p}.1.pwconv1.bias"] = (4 * D,)
        s[f"{p}.1.pwconv2.weight"] = (D, 4 * D)
        s[f"{p}.1.pwconv2.bias"] = (D,)
        s[f"{p}.1.gamma"] = (D,)
    s["decoder.model.0.conv.weight"] = (a.decoder_dim, D, 7)
    s["decoder.model.0.conv.bias"] = (a.decoder_dim,)
    for i, r in enumerate(a.decoder_rates):
        cin, cout = a.decoder_dim // 2 ** i, a.decoder_dim // 2 ** (i + 1)
        p = f"decoder.model.{i + 1}.block"
        s[f"{p}.0.alpha"] = (1, cin, 1)
        s[f"{p}.1.conv.weight"] = (cin, cout, 2 * r)
        s[f"{p}.1.conv.bias"] = (cout,)
        for u in range(3):
            q = f"{p}.{u + 2}.block"
            s[f"{q}.0.alpha"] = (1, cout, 1)
            s[f"{q}.1.conv.weight"] = (cout, cout, 7)
            s[f"{q}.1.conv.bias"] = (cout,)
            s[f"{q}.2.alpha"] = (1, cout, 1)
            s[f"{q}.3.conv.weight"] = (cout, cout, 1)
            s[f"{q}.3.conv.bias"] = (cout,)
    n = len(a.decoder_rates) + 1
    last = a.decoder_dim // 2 ** len(a.decoder_rates)
    s[f"decoder.model.{n}.alpha"] = (1, last, 1)
    s[f"decoder.model.{n + 1}.conv.weight"] = (1, last, 7)
    s[f"decoder.model.{n + 1}.conv.bias"] = (1,)
    return s


def codec_encoder_state_dict_shapes(a) -> Dict[str, tuple]:
    """Folded encode-path tensors: Encoder (vocoder.py:539-575), quantizer.downsample / pre_module (724-735, 752),
    the quantisers' in_proj (dac VectorQuantize)."""
    s: Dict[str, tuple] = {}
    D = a.latent_dim
    H = a.tf_n_head * a.tf_head_dim
    d = a.encoder_dim
    s["encoder.block.0.conv.weight"] = (d, 1, 7)
    s["encoder.block.0.conv.bias"] = (d,)
    for i, (r, nt) in enumerate(zip(a.encoder_rates, a.encoder_transformer_layers)):
        d *= 2
        p = f"encoder.block.{i + 1}.block"
        for u in range(3):
            q = f"{p}.{u}.block"
            s[f"{q}.0.alpha"] = (1, d // 2, 1)
            s[f"{q}.1.conv.weight"] = (d // 2, d // 2, 7)
            s[f"{q}.1.conv.bias"] = (d // 2,)
            s[f"{q}.2.alpha"] = (1, d // 2, 1)
            s[f"{q}.3.conv.weight"] = (d // 2, d // 2, 1)
            s[f"{q}.3.conv.bias"] = (d // 2,)
        s[f"{p}.3.alpha"] = (1, d // 2, 1)
        s[f"{p}.4.conv.weight"] = (d, d // 2, 2 * r)
        s[f"{p}.4.conv.bias"] = (d,)
        for l in range(nt):
            t = f"{p}.5.layers.{l}"
            s[f"{t}.attention.wqkv.weight"] = (3 * d, d)
            s[f"{t}.attention.wo.weight"] = (d, d)
            s[f"{t}.feed_forward.w1.weight"] = (3 * d, d)
            s[f"{t}.feed_forward.w3.weight"] = (3 * d, d)
            s[f"{t}.feed_forward.w2.weight"] = (d, 3 * d)
            for n in ("ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma", "ffn_layer_scale.gamma"):
                s[f"{t}.{n}"] = (d,)
        if nt:
            s[f"{p}.5.norm.weight"] = (d,)
    n = len(a.encoder_rates) + 1
    s[f"encoder.block.{n}.alpha"] = (1, d, 1)
    s[f"encoder.block.{n + 1}.conv.weight"] = (D, d, 3)
    s[f"encoder.block.{n + 1}.conv.bias"] = (D,)
    for j, f in enumerate(a.downsample_factor):
        p = f"quantizer.downsample.{j}"
        s[f"{p}.0.conv.weight"] = (D, D, f)
        s[f"{p}.0.conv.bias"] = (D,)
        s[f"{p}.1.dwconv.conv.weight"] = (D, 1, 7)
        s[f"{p}.1.dwconv.conv.bias"] = (D,)
        s[f"{p}.1.norm.weight"] = (D,)
        s[f"{p}.1.norm.bias"] = (D,)
        s[f"{p}.1.pwconv1.weight"] = (4 * D, D)
        s[f"{p}.1.pwconv1.bias"] = (4 * D,)
        s[f"{p}.1.pwconv2.weight"] = (D, 4 * D)
        s[f"{p}.1.pwconv2.bias"] = (D,)
        s[f"{p}.1.gamma"] = (D,)
    for l in range(a.n_tf_layer):
        p = f"quantizer.pre_module.layers.{l}"
        s[f"{p}.attention.wqkv.weight"] = (3 * H, D)
        s[f"{p}.attention.wo.weight"] = (D, H)
        s[f"{p}.feed_forward.w1.weight"] = (a.tf_ffn, D)
        s[f"{p}.feed_forward.w3.weight"] = (a.tf_ffn, D)
        s[f"{p}.feed_forward.w2.weight"] = (D, a.tf_ffn)
        for n in ("ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma", "ffn_layer_scale.gamma"):
            s[f"{p}.{n}"] = (D,)
    s["quantizer.pre_module.norm.weight"] = (D,)
    s["quantizer.semantic_quantizer.quantizers.0.in_proj.weight"] = (a.codebook_dim, D, 1)
    s["quantizer.semantic_quantizer.quantizers.0.in_proj.bias"] = (a.codebook_dim,)
    for i in range(a.n_codebooks):
        s[f"quantizer.quantizer.quantizers.{i}.in_proj.weight"] = (a.codebook_dim, D, 1)
        s[f"quantizer.quantizer.quantizers.{i}.in_proj.bias"] = (a.codebook_dim,)
    return s


def random_codec_state_dict(a, seed: int = 0, with_encoder: bool = False) -> Dict[str, torch.Tensor]:
    """Synthetic codec weights with O(1) activations through the stack (for benches / smoke runs)."""
    import math
    g = torch.Generator().manual_seed(seed)
    out = {}
    shapes = codec_state_dict_shapes(a)
    if with_encoder:
        shapes.update(codec_encoder_state_dict_shapes(a))
    for k, shp in shapes.items():
        if k.endswith("alpha"):
            w = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("gamma"):
            w = 0.1 + 0.1 * torch.rand(shp, generator=g)
        elif k.endswith("norm.weight"):
            w = torch.ones(shp)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        elif "codebook.weight" in k:
            w = torch.randn(shp, generator=g)
        else:
            if k.startswith("quantizer.upsample.") and k.endswith(".0.conv.weight"):
                fan_in = shp[0]
            elif k.startswith("decoder.model.") and k.endswith(".block.1.conv.weight"):
                fan_in = 2 * shp[0]
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
            w = torch.randn(shp, generator=g) / math.sqrt(fan_in)
            if k.endswith(".block.3.conv.weight"):
                w = 0.3 * w
            if k.startswith("decoder.model.") and k.endswith(".block.1.conv.weight"):
                w = 0.7 * w
            if k == "encoder.block.0.conv.weight":
                w = 3.0 * w   # audio in [-1, 1] has a small RMS
        out[k] = w.float()
    return out
