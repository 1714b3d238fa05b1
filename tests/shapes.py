"""Shapes and seeded inputs shared by the golden-vector generator and the tests."""
import torch


from oracle.ar import ARShape


def tiny_shape():
    n_text, n_sem = 256, 2048
    return ARShape(vocab_size=2319, n_layer=2, n_head=4, dim=64, intermediate_size=128, n_local_heads=2,
                   head_dim=16, rope_base=1e6, norm_eps=1e-6, max_seq_len=128, tie_word_embeddings=True,
                   attention_qk_norm=True, codebook_size=2048, num_codebooks=10,
                   scale_codebook_embeddings=True, n_fast_layer=2, fast_dim=64, fast_n_head=4,
                   fast_n_local_heads=2, fast_head_dim=16, fast_intermediate_size=128,
                   fast_attention_qk_norm=False, initializer_range=0.5,
                   semantic_begin_id=n_text + 15, semantic_end_id=n_text + 15 + n_sem - 1, im_end_id=n_text + 4)


def tiny_shape_b():
    """Variant exercising the other branches: untied head, fast_dim != dim (fast_project_in),
    no slow qk-norm, no codebook scaling, biases on."""
    n_text, n_sem = 256, 1024
    return ARShape(vocab_size=1295, n_layer=2, n_head=4, dim=64, intermediate_size=96, n_local_heads=4,
                   head_dim=16, rope_base=1e4, norm_eps=1e-5, max_seq_len=96, tie_word_embeddings=False,
                   attention_qkv_bias=True, attention_o_bias=True, attention_qk_norm=False,
                   codebook_size=1024, num_codebooks=4, scale_codebook_embeddings=False, n_fast_layer=1,
                   fast_dim=32, fast_n_head=2, fast_n_local_heads=1, fast_head_dim=16,
                   fast_intermediate_size=64, fast_attention_qkv_bias=True, fast_attention_qk_norm=True,
                   fast_attention_o_bias=True, initializer_range=0.5,
                   semantic_begin_id=n_text + 15, semantic_end_id=n_text + 15 + n_sem - 1, im_end_id=n_text + 4)


def make_prompt(shape, T, seed, n_vq=0):
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(shape.num_codebooks + 1, T, dtype=torch.int)
    p[0] = torch.randint(0, shape.semantic_begin_id - 15, (T,), generator=g)
    p[0, 0] = shape.semantic_begin_id - 15 + 11  # <|interleave|>
    for j in range(n_vq):  # a few VQ positions so the codebook-embedding branch is live
        col = 2 + j
        code0 = int(torch.randint(0, shape.semantic_end_id - shape.semantic_begin_id + 1, (1,), generator=g))
        p[0, col] = shape.semantic_begin_id + code0
        p[1, col] = code0
        p[2:, col] = torch.randint(0, min(1024, shape.codebook_size), (shape.num_codebooks - 1,), generator=g)
    return p


def s1mini_shape(max_seq_len=128):
    """openaudio-s1-mini widths and depths (SURVEY.md §8 'Shapes'), ids laid out as tokenizer.py:83-101."""
    n_sem = 4096
    n_text = 155776 - 15 - n_sem
    return ARShape(vocab_size=155776, n_layer=28, n_head=16, dim=1024, intermediate_size=3072, n_local_heads=8,
                   head_dim=128, rope_base=1e6, norm_eps=1e-6, max_seq_len=max_seq_len, tie_word_embeddings=True,
                   attention_qk_norm=True, codebook_size=4096, num_codebooks=10, scale_codebook_embeddings=True,
                   n_fast_layer=4, fast_dim=1024, fast_n_head=16, fast_n_local_heads=8, fast_head_dim=64,
                   fast_intermediate_size=3072, fast_attention_qk_norm=False, initializer_range=0.02,
                   semantic_begin_id=n_text + 15, semantic_end_id=n_text + 15 + n_sem - 1, im_end_id=n_text + 4)
