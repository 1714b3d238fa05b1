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
  498-575  EncoderBlock / Encoder, 765-783 quantizer forward (encode side), 885-904 DAC.encode -> CodecOracle.encode
Third-party pieces absent from /root/reference and unpinned (descript-audio-codec),
restated from their published definitions:
  dac.nn.layers.Snake1d:                x + (alpha + 1e-9)^-1 * sin(alpha x)^2
  dac.nn.quantize.ResidualVectorQuantize.from_codes:  sum_i out_proj_i(codebook_i[codes_i])
  dac.nn.quantize.VectorQuantize.forward (encode): in_proj, L2-normalised nearest neighbour, out_proj; residual loop
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
    # encoder side (encode_reference; synthesizer.py:255-268, vocoder.py:498-575)
    encoder_dim: int = 64
    encoder_rates: List[int] = field(default_factory=lambda: [2, 4, 8, 8])
    encoder_tf_layers: List[int] = field(default_factory=lambda: [0, 0, 0, 4])
    enc_tf_window: int = 512
    enc_tf_block_size: int = 16384

    @property
    def frame_len(self) -> int:
        n = 1
        for r in self.upsample + self.rates:
            n *= r
        return n

    @property
    def hop_length(self) -> int:
        n = 1
        for r in self.encoder_rates:
            n *= r
        return n

    @property
    def enc_frame_len(self) -> int:
        """DAC.frame_length = hop_length * 4 (vocoder.py:872): audio samples per code frame on the encode side."""
        return self.hop_length * 4


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


def encoder_weight_shapes(c: CodecShape) -> Dict[str, tuple]:
    """Folded encode-path tensors: Encoder (vocoder.py:539-575), quantizer.downsample / pre_module / in_proj
    (683-757, dac VectorQuantize)."""
    s: Dict[str, tuple] = {}
    D = c.latent_dim
    d = c.encoder_dim
    s["encoder.block.0.conv.weight"] = (d, 1, 7)
    s["encoder.block.0.conv.bias"] = (d,)
    for i, (r, nt) in enumerate(zip(c.encoder_rates, c.encoder_tf_layers)):
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
        H = d  # n_head = dim // 64, head_dim 64
        for l in range(nt):
            t = f"{p}.5.layers.{l}"
            s[f"{t}.attention.wqkv.weight"] = (3 * H, d)
            s[f"{t}.attention.wo.weight"] = (d, H)
            s[f"{t}.feed_forward.w1.weight"] = (3 * d, d)
            s[f"{t}.feed_forward.w3.weight"] = (3 * d, d)
            s[f"{t}.feed_forward.w2.weight"] = (d, 3 * d)
            for n in ("ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma", "ffn_layer_scale.gamma"):
                s[f"{t}.{n}"] = (d,)
        if nt:
            s[f"{p}.5.norm.weight"] = (d,)
    n = len(c.encoder_rates) + 1
    s[f"encoder.block.{n}.alpha"] = (1, d, 1)
    s[f"encoder.block.{n + 1}.conv.weight"] = (D, d, 3)
    s[f"encoder.block.{n + 1}.conv.bias"] = (D,)
    for j, f in enumerate(c.upsample):   # downsample_factor == the upsample factors (vocoder.py:724-748)
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
    H = c.tf_n_head * c.tf_head_dim
    for l in range(c.n_tf_layer):
        p = f"quantizer.pre_module.layers.{l}"
        s[f"{p}.attention.wqkv.weight"] = (3 * H, D)
        s[f"{p}.attention.wo.weight"] = (D, H)
        s[f"{p}.feed_forward.w1.weight"] = (c.tf_ffn, D)
        s[f"{p}.feed_forward.w3.weight"] = (c.tf_ffn, D)
        s[f"{p}.feed_forward.w2.weight"] = (D, c.tf_ffn)
        for n in ("ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma", "ffn_layer_scale.gamma"):
            s[f"{p}.{n}"] = (D,)
    s["quantizer.pre_module.norm.weight"] = (D,)
    s["quantizer.semantic_quantizer.quantizers.0.in_proj.weight"] = (c.codebook_dim, D, 1)
    s["quantizer.semantic_quantizer.quantizers.0.in_proj.bias"] = (c.codebook_dim,)
    for i in range(c.n_codebooks):
        s[f"quantizer.quantizer.quantizers.{i}.in_proj.weight"] = (c.codebook_dim, D, 1)
        s[f"quantizer.quantizer.quantizers.{i}.in_proj.bias"] = (c.codebook_dim,)
    return s


def random_encoder_weights(c: CodecShape, seed: int = 1) -> Dict[str, torch.Tensor]:
    """Seeded synthetic encoder-side weights, same scaling rules as random_weights."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in encoder_weight_shapes(c).items():
        if k.endswith("alpha"):
            w = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("gamma"):
            w = 0.1 + 0.1 * torch.rand(shp, generator=g)
        elif k.endswith("norm.weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            w = torch.randn(shp, generator=g) / math.sqrt(fan_in)
            if k.endswith(".block.3.conv.weight"):
                w = 0.3 * w
            if k == "encoder.block.0.conv.weight":
                w = 3.0 * w   # audio in [-1, 1] has small RMS: lift the first stage to O(1)
        out[k] = w.float()
    return out


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


def causal_conv_strided(x, w, b, stride):
    """vocoder.py:394-421 with stride > 1: left pad k - stride, right pad up to a whole number of frames."""
    k = w.shape[-1]
    pad = k - stride
    length = x.shape[-1]
    n_frames = (length - k + pad) / stride + 1
    extra = (math.ceil(n_frames) - 1) * stride + (k - pad) - length
    return F.conv1d(F.pad(x, (pad, extra)), w, b, stride=stride)


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
    def __init__(self, shape: CodecShape, weights: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float32):
        """dtype=torch.bfloat16 restates the precision the reference ENCODES in (synthesizer.py:289-291, 345-353: the
        codec is moved to bfloat16 on the accelerator, the audio is cast to the model's dtype): every parameter and every
        activation is bf16, each torch op rounds its result once - the like-for-like partner of the HIP encode path
        (bf16 operands, f32 accumulation, bf16 activations).  The decode oracle stays f32 (vocoder.py:906-912 as used by
        the synthesizer's f32 decode)."""
        self.c = shape
        self.dtype = dtype
        self.w = {k: v.float().to(dtype) for k, v in weights.items()}
        self.tab = rope_table(shape.tf_block_size, shape.tf_head_dim, shape.tf_rope_base)
        self.taps: Dict[str, torch.Tensor] = {}

    def _tf_block(self, p, x, tab, mask, n_head=None):
        c, w = self.c, self.w
        B, T, _ = x.shape
        H, hd = (n_head or c.tf_n_head), c.tf_head_dim
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

    def _window_transformer(self, prefix, z, n_layer, window, n_head=None, ffn=None, tab=None):
        """vocoder.py:338-354 + 271-293 (causal band mask 325-332)."""
        c, w = self.c, self.w
        x = z.transpose(1, 2)
        T = x.shape[1]
        rows = torch.arange(T).view(-1, 1)
        cols = torch.arange(T)
        mask = ((cols >= (rows - window + 1).clamp(min=0)) & (cols <= rows))[None, None]
        tab = (self.tab if tab is None else tab)[torch.arange(T)]
        for l in range(n_layer):
            x = self._tf_block(f"{prefix}.layers.{l}", x, tab, mask, n_head)
        x = _rms(x, w[f"{prefix}.norm.weight"], c.tf_norm_eps)
        return x.transpose(1, 2)

    def post_module(self, z):
        return self._window_transformer("quantizer.post_module", z, self.c.n_tf_layer, self.c.tf_window)

    # ------------------------------------------------------------------ encode side
    def _res_unit(self, q, x, d):
        w = self.w
        y = snake(x, w[f"{q}.0.alpha"])
        y = causal_conv(y, w[f"{q}.1.conv.weight"], w[f"{q}.1.conv.bias"], dilation=d)
        y = snake(y, w[f"{q}.2.alpha"])
        y = causal_conv(y, w[f"{q}.3.conv.weight"], w[f"{q}.3.conv.bias"])
        return x + y

    def encoder(self, audio):
        """vocoder.py:539-575 with causal=True; audio (B, 1, T) -> (B, latent_dim, T / hop_length)."""
        c, w = self.c, self.w
        x = causal_conv(audio, w["encoder.block.0.conv.weight"], w["encoder.block.0.conv.bias"])
        enc_tab = rope_table(c.enc_tf_block_size, 64, c.tf_rope_base)
        for i, (r, nt) in enumerate(zip(c.encoder_rates, c.encoder_tf_layers)):
            p = f"encoder.block.{i + 1}.block"
            for u, d in enumerate((1, 3, 9)):
                x = self._res_unit(f"{p}.{u}.block", x, d)
            x = snake(x, w[f"{p}.3.alpha"])
            x = causal_conv_strided(x, w[f"{p}.4.conv.weight"], w[f"{p}.4.conv.bias"], r)
            if nt:
                x = self._window_transformer(f"{p}.5", x, nt, c.enc_tf_window, n_head=x.shape[1] // 64, tab=enc_tab)
            self.taps[f"enc_block{i}"] = x
        n = len(c.encoder_rates) + 1
        x = snake(x, w[f"encoder.block.{n}.alpha"])
        return causal_conv(x, w[f"encoder.block.{n + 1}.conv.weight"], w[f"encoder.block.{n + 1}.conv.bias"])

    def _vq(self, p, z):
        """dac VectorQuantize.forward (third party, restated): in_proj, cosine nearest neighbour, out_proj."""
        w = self.w
        e = F.conv1d(z, w[f"{p}.in_proj.weight"], w[f"{p}.in_proj.bias"])          # (B, cd, T)
        B, cd, T = e.shape
        enc = F.normalize(e.permute(0, 2, 1).reshape(B * T, cd))
        cb = F.normalize(w[f"{p}.codebook.weight"])
        dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
        idx = (-dist).max(1)[1].reshape(B, T)
        top2 = (-dist).topk(2, dim=1).values            # test aid: how close the runner-up row was (per frame)
        self.taps.setdefault("vq_gap", []).append((top2[:, 0] - top2[:, 1]).reshape(B, T))
        zq = F.conv1d(F.embedding(idx, w[f"{p}.codebook.weight"]).transpose(1, 2), w[f"{p}.out_proj.weight"], w[f"{p}.out_proj.bias"])
        return zq, idx

    def quantizer_encode(self, z):
        """vocoder.py:765-783: downsample, pre_module, semantic VQ, residual VQ over the rest -> (B, 1+n, T') indices."""
        c, w = self.c, self.w
        self.taps["vq_gap"] = []                         # one (B, T) tensor of top-1/top-2 score gaps per codebook, in order
        for j, f in enumerate(c.upsample):
            p = f"quantizer.downsample.{j}"
            z = causal_conv_strided(z, w[f"{p}.0.conv.weight"], w[f"{p}.0.conv.bias"], f)
            z = self.convnext(f"{p}.1", z)
        z = self._window_transformer("quantizer.pre_module", z, c.n_tf_layer, c.tf_window)
        self.taps["pre"] = z
        sem_z, sem_idx = self._vq("quantizer.semantic_quantizer.quantizers.0", z)
        residual = z - sem_z
        codes = [sem_idx]
        for i in range(c.n_codebooks):
            zq, idx = self._vq(f"quantizer.quantizer.quantizers.{i}", residual)
            residual = residual - zq
            codes.append(idx)
        return torch.stack(codes, dim=1)

    @torch.inference_mode()
    def encode(self, audio: torch.Tensor, audio_lengths: torch.Tensor = None):
        """vocoder.py:885-904: right-pad to whole frames, encoder, quantizer -> (indices, indices_lens)."""
        if audio.ndim == 2:
            audio = audio.unsqueeze(1)
        length = audio.shape[-1]
        fl = self.c.enc_frame_len
        right = math.ceil(length / fl) * fl - length
        audio = F.pad(audio, (0, right))
        if audio_lengths is None:
            audio_lengths = torch.tensor([length + right])
        z = self.encoder(audio.float().to(self.dtype))
        self.taps["enc_out"] = z
        return self.quantizer_encode(z), torch.ceil(audio_lengths / fl).long()

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
