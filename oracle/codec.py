"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement (torch eager, fp32) of the reference's DAC decode path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference files followed (/root/reference/fish_tts/models/vocoder.py):
  800-814  DownsampleResidualVectorQuantize.decode  -> CodecOracle.decode (clamp, RVQ sum, post, upsample)
  94-102   RMSNorm (no upcast)                       -> _rms
  132-156  RoPE table (bf16) + rotation              -> rope_table / _rope
  159-241  Attention / TransformerBlock / LayerScale -> _tf_block
  296-354  WindowLimitedTransformer (band mask 325-332)
  394-463  CausalConvNet / CausalTransConvNet        -> causal_conv / causal_convT
  474-495  ResidualUnit, 578-640 DecoderBlock/Decoder, 644-680 ConvNeXtBlock, 906-912 DAC.decode
Third-party pieces absent from /root/reference and unpinned (descript-audio-codec),
restated from their published definitions:
  dac.nn.layers.Snake1d:                x + (alpha + 1e-9)^-1 * sin(alpha x)^2
  dac.nn.quantize.ResidualVectorQuantize.from_codes:  sum_i out_proj_i(codebook_i[codes_i])
Weights are the *folded* tensors (weight-norm g*v/||v|| already applied, vocoder.py:423-429,457-463)
under the reference's module paths.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class CodecShape:
    """Hyper-parameters hard-coded by the reference in synthesizer.py:199-269."""
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
    upsample: List[int] = field(default_factory=lambda: [2, 2])  # applied in this order (vocoder.py:737-748)
    decoder_dim: int = 1536
    rates: List[int] = field(default_factory=lambda: [8, 8, 4, 2])

    @property
    def frame_len(self) -> int:
        n = 1
        for r in self.upsample + self.rates:
            n *= r
        return n


def weight_shapes(c: CodecShape) -> Dict[str, tuple]:
    D = c.latent_dim
    s: Dict[str, tuple] = {}
    s["quantizer.semantic_quantizer.quantizers.0.codebook.weight"] = (c.semantic_codebook_size, c.codebook_dim)
    s["quantizer.semantic_quantizer.quantizers.0.out_proj.weight"] = (D, c.codebook_dim, 1)
    s["quantizer.semantic_quantizer.quantizers.0.out_proj.bias"] = (D,)
    for i in range(c.n_codebooks):
        s[f"quantizer.quantizer.quantizers.{i}.codebook.weight"] = (c.codebook_size, c.codebook_dim)
        s[f"quantizer.quantizer.quantizers.{i}.out_proj.weight"] = (D, c.codebook_dim, 1)
        s[f"quantizer.quantizer.quantizers.{i}.out_proj.bias"] = (D,)
    H = c.tf_n_head * c.tf_head_dim
    for l in range(c.n_tf_layer):
        p = f"quantizer.post_module.layers.{l}"
        s[f"{p}.attention.wqkv.weight"] = (3 * H, D)
        s[f"{p}.attention.wo.weight"] = (D, H)
        s[f"{p}.feed_forward.w1.weight"] = (c.tf_ffn, D)
        s[f"{p}.feed_forward.w3.weight"] = (c.tf_ffn, D)
        s[f"{p}.feed_forward.w2.weight"] = (D, c.tf_ffn)
        s[f"{p}.ffn_norm.weight"] = (D,)
        s[f"{p}.attention_norm.weight"] = (D,)
        s[f"{p}.attention_layer_scale.gamma"] = (D,)
        s[f"{p}.ffn_layer_scale.gamma"] = (D,)
    s["quantizer.post_module.norm.weight"] = (D,)
    for j, f in enumerate(c.upsample):
        p = f"quantizer.upsample.{j}"
        s[f"{p}.0.conv.weight"] = (D, D, f)  # ConvTranspose1d: (in, out, k)
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
    ch = c.decoder_dim
    s["decoder.model.0.conv.weight"] = (ch, D, 7)
    s["decoder.model.0.conv.bias"] = (ch,)
    for i, r in enumerate(c.rates):
        cin, cout = c.decoder_dim // 2 ** i, c.decoder_dim // 2 ** (i + 1)
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
    last = c.decoder_dim // 2 ** len(c.rates)
    n = len(c.rates) + 1
    s[f"decoder.model.{n}.alpha"] = (1, last, 1)
    s[f"decoder.model.{n + 1}.conv.weight"] = (1, last, 7)
    s[f"decoder.model.{n + 1}.conv.bias"] = (1,)
    return s


def random_weights(c: CodecShape, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights with activations of O(1) through the stack (fan-in scaled normals;
    Snake alphas in [0.5, 1.5]; LayerScale / ConvNeXt gammas of O(0.1) so every branch is visible)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in weight_shapes(c).items():
        if k.endswith("alpha"):
            w = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("gamma"):
            w = 0.1 + 0.1 * torch.rand(shp, generator=g)
        elif k.endswith("norm.weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        elif "codebook.weight" in k:
            w = torch.randn(shp, generator=g)
        else:
            is_up = k.startswith("quantizer.upsample.") and k.endswith(".0.conv.weight")
            is_dec_t = k.startswith("decoder.model.") and k.endswith(".block.1.conv.weight")
            if is_up:        # ConvTranspose1d (in, out, k) with k == stride: one tap per output sample
                fan_in = shp[0]
            elif is_dec_t:   # ConvTranspose1d with k == 2*stride: two taps overlap per output sample
                fan_in = 2 * shp[0]
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
            w = torch.randn(shp, generator=g) / math.sqrt(fan_in)
            if k.endswith(".block.3.conv.weight"):
                w = 0.3 * w  # residual branch of a ResidualUnit: keeps the stack's scale O(1)
            if k.startswith("decoder.model.") and k.endswith(".block.1.conv.weight"):
                w = 0.7 * w  # Snake adds a positive offset; the transposed convs rein it in
        out[k] = w.float()
    return out


# ----------------------------------------------------------------------------- math
def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def causal_conv(x, w, b, dilation=1, groups=1):
    """vocoder.py:394-421 for stride 1: left pad (k-1)*dilation, no right pad."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    return F.conv1d(F.pad(x, (k_eff - 1, 0)), w, b, dilation=dilation, groups=groups)


def causal_convT(x, w, b, stride):
    """vocoder.py:432-455: full transposed conv, then drop the last (k - stride) samples."""
    y = F.conv_transpose1d(x, w, b, stride=stride)
    cut = w.shape[-1] - stride
    return y[..., : y.shape[-1] - cut] if cut > 0 else y


def rope_table(n_pos: int, n_elem: int, base: float) -> torch.Tensor:
    inv = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(n_pos), inv)
    z = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([z.real, z.imag], dim=-1).to(torch.bfloat16)  # vocoder.py:132-142


def _rope(x, tab):
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    t = tab.view(1, xs.size(1), 1, xs.size(3), 2)
    re = xs[..., 0] * t[..., 0] - xs[..., 1] * t[..., 1]
    im = xs[..., 1] * t[..., 0] + xs[..., 0] * t[..., 1]
    return torch.stack([re, im], dim=-1).flatten(3).type_as(x)


def _rms(x, w, eps):
    return (x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)).type_as(x) * w


class CodecOracle:
    def __init__(self, shape: CodecShape, weights: Dict[str, torch.Tensor]):
        self.c = shape
        self.w = {k: v.float() for k, v in weights.items()}
        self.tab = rope_table(shape.tf_block_size, shape.tf_head_dim, shape.tf_rope_base)
        self.taps: Dict[str, torch.Tensor] = {}

    def _tf_block(self, p, x, tab, mask):
        c, w = self.c, self.w
        B, T, _ = x.shape
        H, hd = c.tf_n_head, c.tf_head_dim
        xn = _rms(x, w[f"{p}.attention_norm.weight"], c.tf_norm_eps)
        q, k, v = F.linear(xn, w[f"{p}.attention.wqkv.weight"]).split([H * hd] * 3, dim=-1)
        q, k, v = (t.view(B, T, H, hd) for t in (q, k, v))
        q, k = _rope(q, tab), _rope(k, tab)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
        y = y.transpose(1, 2).contiguous().view(B, T, H * hd)
        h = x + F.linear(y, w[f"{p}.attention.wo.weight"]) * w[f"{p}.attention_layer_scale.gamma"]
        hn = _rms(h, w[f"{p}.ffn_norm.weight"], c.tf_norm_eps)
        f = F.linear(F.silu(F.linear(hn, w[f"{p}.feed_forward.w1.weight"])) * F.linear(hn, w[f"{p}.feed_forward.w3.weight"]),
                     w[f"{p}.feed_forward.w2.weight"])
        return h + f * w[f"{p}.ffn_layer_scale.gamma"]

    def post_module(self, z):
        """vocoder.py:338-354 + 271-293."""
        c, w = self.c, self.w
        x = z.transpose(1, 2)
        T = x.shape[1]
        rows = torch.arange(T).view(-1, 1)
        cols = torch.arange(T)
        mask = ((cols >= (rows - c.tf_window + 1).clamp(min=0)) & (cols <= rows))[None, None]
        tab = self.tab[torch.arange(T)]
        for l in range(c.n_tf_layer):
            x = self._tf_block(f"quantizer.post_module.layers.{l}", x, tab, mask)
        x = _rms(x, w["quantizer.post_module.norm.weight"], c.tf_norm_eps)
        return x.transpose(1, 2)

    def convnext(self, p, x):
        w = self.w
        y = causal_conv(x, w[f"{p}.dwconv.conv.weight"], w[f"{p}.dwconv.conv.bias"], groups=x.shape[1])
        y = y.permute(0, 2, 1)
        y = F.layer_norm(y, (y.shape[-1],), w[f"{p}.norm.weight"], w[f"{p}.norm.bias"], 1e-6)
        y = F.linear(y, w[f"{p}.pwconv1.weight"], w[f"{p}.pwconv1.bias"])
        y = F.gelu(y)
        y = F.linear(y, w[f"{p}.pwconv2.weight"], w[f"{p}.pwconv2.bias"])
        y = w[f"{p}.gamma"] * y
        return x + y.permute(0, 2, 1)

    def quantizer_decode(self, indices):
        """vocoder.py:800-814; indices (B, 1+n_codebooks, T) integer."""
        c, w = self.c, self.w
        idx = indices.long().clone()
        idx[:, 0] = torch.clamp(idx[:, 0], max=c.semantic_codebook_size - 1)
        idx[:, 1:] = torch.clamp(idx[:, 1:], max=c.codebook_size - 1)
        p = "quantizer.semantic_quantizer.quantizers.0"
        z = F.conv1d(F.embedding(idx[:, 0], w[f"{p}.codebook.weight"]).transpose(1, 2),
                     w[f"{p}.out_proj.weight"], w[f"{p}.out_proj.bias"])
        zr = 0.0
        for i in range(c.n_codebooks):
            p = f"quantizer.quantizer.quantizers.{i}"
            zr = zr + F.conv1d(F.embedding(idx[:, i + 1], w[f"{p}.codebook.weight"]).transpose(1, 2),
                               w[f"{p}.out_proj.weight"], w[f"{p}.out_proj.bias"])
        z = z + zr
        self.taps["rvq"] = z
        z = self.post_module(z)
        self.taps["post"] = z
        for j, f in enumerate(c.upsample):
            p = f"quantizer.upsample.{j}"
            z = causal_convT(z, w[f"{p}.0.conv.weight"], w[f"{p}.0.conv.bias"], stride=f)
            z = self.convnext(f"{p}.1", z)
        self.taps["upsampled"] = z
        return z

    def decoder(self, z):
        """vocoder.py:605-640 with causal=True."""
        c, w = self.c, self.w
        x = causal_conv(z, w["decoder.model.0.conv.weight"], w["decoder.model.0.conv.bias"])
        for i, r in enumerate(c.rates):
            p = f"decoder.model.{i + 1}.block"
            x = snake(x, w[f"{p}.0.alpha"])
            x = causal_convT(x, w[f"{p}.1.conv.weight"], w[f"{p}.1.conv.bias"], stride=r)
            for u, d in enumerate((1, 3, 9)):
                q = f"{p}.{u + 2}.block"
                y = snake(x, w[f"{q}.0.alpha"])
                y = causal_conv(y, w[f"{q}.1.conv.weight"], w[f"{q}.1.conv.bias"], dilation=d)
                y = snake(y, w[f"{q}.2.alpha"])
                y = causal_conv(y, w[f"{q}.3.conv.weight"], w[f"{q}.3.conv.bias"])
                x = x + y
            self.taps[f"block{i}"] = x
        n = len(c.rates) + 1
        x = snake(x, w[f"decoder.model.{n}.alpha"])
        x = causal_conv(x, w[f"decoder.model.{n + 1}.conv.weight"], w[f"decoder.model.{n + 1}.conv.bias"])
        return torch.tanh(x)

    @torch.inference_mode()
    def decode(self, indices: torch.Tensor, feature_lengths: torch.Tensor):
        """vocoder.py:906-912 -> (audio (B,1,T*frame_len), audio_lengths)."""
        if indices.ndim == 2:
            indices = indices[None]
        z = self.quantizer_decode(indices)
        return self.decoder(z), feature_lengths * self.c.frame_len
