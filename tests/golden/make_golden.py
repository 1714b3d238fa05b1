"""Generates tests/golden/*.npz by importing the REFERENCE (/root/reference) in this container.

Run here only (`python tests/golden/make_golden.py`); the reference never travels to the GPU
box, the .npz vectors do.  The reference's three absent third-party modules (tiktoken, dac,
audiotools) are registered as empty placeholder modules: they are not
called on the AR path.  For the codec path the two third-party classes the decode path needs
(dac's Snake1d and ResidualVectorQuantize.from_codes, absent and unpinned: SURVEY.md §8-c) are
restated from their published definitions; every other codec op executed is the reference's
own vocoder.py.

Weights are seeded synthetic tensors from oracle.ar.random_weights / oracle.codec.random_weights
(inputs, not reference outputs) loaded into the reference modules with load_state_dict.
"""
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Snake1d(nn.Module):
    """dac.nn.layers.Snake1d (published definition): x + 1/(alpha+1e-9) * sin(alpha x)^2."""

    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))

    def forward(self, x):
        shape = x.shape
        x = x.reshape(shape[0], shape[1], -1)
        x = x + (self.alpha + 1e-9).reciprocal() * torch.sin(self.alpha * x).pow(2)
        return x.reshape(shape)


class _VQ(nn.Module):
    def __init__(self, input_dim, codebook_size, codebook_dim):
        super().__init__()
        from torch.nn.utils.parametrizations import weight_norm
        self.codebook_size = codebook_size
        self.in_proj = weight_norm(nn.Conv1d(input_dim, codebook_dim, 1))
        self.out_proj = weight_norm(nn.Conv1d(codebook_dim, input_dim, 1))
        self.codebook = nn.Embedding(codebook_size, codebook_dim)


class _RVQ(nn.Module):
    """dac.nn.quantize.ResidualVectorQuantize restated from its published definition: from_codes (decode) and the
    inference forward (encode: residual loop over VectorQuantize.forward = in_proj, L2-normalised nearest neighbour
    on the factorised codes, out_proj)."""

    def forward(self, z, n_quantizers=None):
        z_q, residual, codes, latents = 0.0, z, [], []
        for q in self.quantizers:
            z_e = q.in_proj(residual)
            B, D, T = z_e.shape
            enc = torch.nn.functional.normalize(z_e.permute(0, 2, 1).reshape(B * T, D))
            cb = torch.nn.functional.normalize(q.codebook.weight)
            dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
            idx = (-dist).max(1)[1].reshape(B, T)
            z_q_i = q.out_proj(q.codebook(idx).transpose(1, 2))
            z_q = z_q + z_q_i
            residual = residual - z_q_i
            codes.append(idx)
            latents.append(z_e)
        zero = torch.zeros(())
        return z_q, torch.stack(codes, dim=1), torch.cat(latents, dim=1), zero, zero

    def __init__(self, input_dim=512, n_codebooks=9, codebook_size=1024, codebook_dim=8,
                 quantizer_dropout=0.0):
        super().__init__()
        self.n_codebooks = n_codebooks
        self.codebook_size = codebook_size
        self.quantizers = nn.ModuleList(_VQ(input_dim, codebook_size, codebook_dim) for _ in range(n_codebooks))

    def from_codes(self, codes):
        z_q = 0.0
        z_p = []
        for i in range(codes.shape[1]):
            z_p_i = self.quantizers[i].codebook(codes[:, i, :]).transpose(1, 2)
            z_p.append(z_p_i)
            z_q = z_q + self.quantizers[i].out_proj(z_p_i)
        return z_q, torch.cat(z_p, dim=1), codes


class _CodecMixin:
    def get_delay(self):
        return 0


def import_reference():
    sys.path.insert(0, "/root/reference")
    tk = _stub("tiktoken")
    tk.core = _stub("tiktoken.core", Encoding=object)
    _stub("dac"); _stub("dac.nn"); _stub("dac.model")
    _stub("dac.nn.layers", Snake1d=_Snake1d, WNConv1d=nn.Module, WNConvTranspose1d=nn.Module)
    _stub("dac.nn.quantize", ResidualVectorQuantize=_RVQ)
    _stub("dac.model.base", CodecMixin=_CodecMixin)
    _stub("audiotools"); _stub("audiotools.ml", BaseModel=nn.Module)
    from fish_tts.models import inference, llama, vocoder
    return llama, inference, vocoder


class FakeTok:
    """Byte-level stand-in exposing what the model/prompt code touches
    (llama.py:346,418-419; inference.py:123,182,546,555,632), ids laid out as tokenizer.py:83-101."""

    NAMED = ["<|begin_of_text|>", "<|end_of_text|>", "<|pad|>", "<|im_start|>", "<|im_end|>",
             "<|phoneme_start|>", "<|phoneme_end|>", "<|tool_call_start|>", "<|tool_call_end|>",
             "<|text|>", "<|voice|>", "<|interleave|>", "<|audio_start|>", "<|audio_end|>", "<|audio|>"]

    def __init__(self, n_text=256, n_sem=4096):
        self.n_text = n_text
        self.special = {t: n_text + i for i, t in enumerate(self.NAMED)}
        base = n_text + len(self.NAMED)
        self.semantic_id_to_token_id = {i: base + i for i in range(n_sem)}
        self.semantic_begin_id = base
        self.semantic_end_id = base + n_sem - 1

    def get_token_id(self, tok):
        return self.special[tok]

    def encode(self, s):
        import re
        out = []
        pat = "(" + "|".join(re.escape(t) for t in self.NAMED) + ")"
        for piece in re.split(pat, s):
            if not piece:
                continue
            if piece in self.special:
                out.append(self.special[piece])
            else:
                out.extend(b % self.n_text for b in piece.encode("utf-8"))
        return out


def build_reference_model(llama, shape, weights, dtype):
    tok = FakeTok(n_text=shape.semantic_begin_id - 15, n_sem=shape.semantic_end_id - shape.semantic_begin_id + 1)
    fields = {f for f in llama.DualARModelArgs.__dataclass_fields__}
    kw = {k: v for k, v in shape.__dict__.items() if k in fields}
    cfg = llama.DualARModelArgs(**kw)
    model = llama.DualARTransformer(cfg, tokenizer=tok).eval()
    missing, unexpected = model.load_state_dict(weights, strict=False)
    assert not unexpected, unexpected
    assert all(("freqs" in m or "mask" in m or "kv_cache" in m) for m in missing), missing
    return model.to(dtype), tok


from tests.shapes import make_prompt, tiny_shape, tiny_shape_b  # noqa: E402


def gen_ar(llama, inference, name, shape, dtype, T, n_new, seed_w=0):
    from oracle.ar import random_weights
    w = random_weights(shape, seed=seed_w)
    model, tok = build_reference_model(llama, shape, w, dtype)
    prompt = make_prompt(shape, T, seed=1, n_vq=3)
    out = {"prompt": prompt.numpy()}
    cases = [("greedy_rep1.0", dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.0), None),
             ("greedy_rep1.1", dict(temperature=0.7, top_p=1e-6, repetition_penalty=1.1), None),
             ("sampled_seed7", dict(temperature=0.7, top_p=0.8, repetition_penalty=1.1), 7),
             ("sampled_seed11", dict(temperature=1.0, top_p=0.95, repetition_penalty=1.5), 11)]
    for cname, kw, seed in cases:
        model._cache_setup_done = False
        for attr in ("fixed_temperature", "fixed_top_p", "fixed_repetition_penalty"):
            if hasattr(model, attr):
                delattr(model, attr)
        model.max_seq_len = -1
        model.max_batch_size = -1
        if seed is not None:
            torch.manual_seed(seed)
        seq = inference.generate(model=model, prompt=prompt.clone(), max_new_tokens=n_new, audio_masks=None,
                                 audio_parts=None, **kw)
        out[f"{cname}.seq"] = seq.numpy().copy()
        # streaming variant (EOS frame included, inference.py:645-738)
        model._cache_setup_done = False
        model.max_seq_len = -1
        model.max_batch_size = -1
        if seed is not None:
            torch.manual_seed(seed)
        cols = list(inference.generate_streaming(model=model, prompt=prompt.clone(), max_new_tokens=n_new,
                                                 audio_masks=None, audio_parts=None, **kw))
        out[f"{cname}.stream"] = torch.cat(cols, dim=1).numpy().copy()
    # frame-0 logits + hidden (prefill), for the tolerance-based GPU tests
    model._cache_setup_done = False
    model.max_seq_len = -1
    model.max_batch_size = -1
    with torch.inference_mode():
        model.setup_caches(max_batch_size=1, max_seq_len=shape.max_seq_len, dtype=dtype)
        res = model.forward_generate(prompt.view(1, shape.num_codebooks + 1, -1), torch.arange(T))
    out["frame0.logits"] = res.logits.float().numpy().reshape(-1).copy()
    out["frame0.hidden"] = res.hidden_states.float().numpy().reshape(-1).copy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def gen_sampling(inference):
    """inference.py:30-61 on fixed logits: penalty quirks, top-p boundary, bf16 path."""
    out = {}
    g = torch.Generator().manual_seed(3)
    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        for V in (1024, 2320):
            logits = (3.0 * torch.randn(V, generator=g)).to(dtype)
            prev = torch.randint(0, V, (16,), generator=g).int()
            prev[:5] = 0  # early zero padding (inference.py:187-191)
            prev[7] = prev[6]  # duplicate id
            for tp in (0.8, 0.2, 1.0, 1e-6):
                for rep in (1.0, 1.1, 1.5):
                    l = logits.clone()
                    probs = inference.logits_to_probs(l, torch.tensor(0.7), torch.tensor(tp), torch.tensor(rep), prev)
                    key = f"{tag}.V{V}.tp{tp}.rep{rep}"
                    out[key + ".probs"] = probs.float().numpy()
                    out[key + ".penalised"] = l.float().numpy()
            out[f"{tag}.V{V}.logits"] = logits.float().numpy()
            out[f"{tag}.V{V}.prev"] = prev.numpy()
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **out)
    print("sampling", len(out))


def gen_prompt(inference):
    """A1/A7: prompt matrix for (ref transcript, gura_voice.npy, text) under the byte FakeTok."""
    tok = FakeTok(n_text=151643, n_sem=4096)
    codes = np.load("/root/reference/gura_voice.npy")
    out = {}
    for name, refs, text in (("noref", [], "Hello world"),
                             ("oneref", [("ref transcript", codes[:, :37])], "Hello world"),
                             ("tworef", [("first one", codes[:, :5]), ("zweite", codes[:, 100:111])], "Nice to meet you.")):
        seq = inference.ContentSequence(modality="interleave")
        for t, c in refs:
            seq.append([inference.TextPart(text=t), inference.VQPart(codes=torch.from_numpy(c))], add_end=True, speaker=0)
        seq.append([inference.TextPart(text=text)], add_end=False, speaker=0)
        enc, am, ap = seq.encode_for_inference(tok, num_codebooks=10)
        assert am is None and ap is None
        out[f"{name}.prompt"] = enc.numpy()
        for i, (t, c) in enumerate(refs):
            out[f"{name}.ref{i}.codes"] = c
            out[f"{name}.ref{i}.text"] = np.frombuffer(t.encode(), dtype=np.uint8)
        out[f"{name}.text"] = np.frombuffer(text.encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "prompt.npz"), **out)
    print("prompt", {k: v.shape for k, v in out.items()})


def main():
    llama, inference, vocoder = import_reference()
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "codec_encode":   # only the encode-side fixture
        from tests.golden import make_golden_codec
        make_golden_codec.gen_encode(vocoder)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "f16":            # only the precision="fp16" fixtures (synthesizer.py:125-126)
        gen_ar(llama, inference, "ar_tiny_f16", tiny_shape(), torch.float16, T=9, n_new=16)
        gen_ar(llama, inference, "ar_tinyb_f16", tiny_shape_b(), torch.float16, T=12, n_new=12)
        return
    gen_ar(llama, inference, "ar_tiny_f32", tiny_shape(), torch.float32, T=9, n_new=16)
    gen_ar(llama, inference, "ar_tiny_bf16", tiny_shape(), torch.bfloat16, T=9, n_new=16)
    gen_ar(llama, inference, "ar_tinyb_f32", tiny_shape_b(), torch.float32, T=12, n_new=12)
    gen_ar(llama, inference, "ar_tinyb_bf16", tiny_shape_b(), torch.bfloat16, T=12, n_new=12)
    gen_sampling(inference)
    gen_prompt(inference)
    if os.path.exists(os.path.join(ROOT, "oracle", "codec.py")):
        from tests.golden import make_golden_codec
        make_golden_codec.main(vocoder)
        make_golden_codec.gen_encode(vocoder)


if __name__ == "__main__":
    main()
