"""
Synthetic weights and requests (there is no network on the GPU box, so real checkpoints are optional).

Two uses:
  * tests / golden fixtures: tiny models, generated with numpy's PCG64 (bit-stable across platforms),
    every value rounded onto the bf16 grid so the numpy oracle (fp32 arrays) and the HIP path
    (bf16 storage) see *identical* weights.
  * bench.py: LLaVA-v1.6-vicuna-7B-shaped weights created directly on the GPU (torch generator).

Weight names are the reference's state-dict names (vispec/model/cnets_ours.py:683-757 for
the draft, HF Llama names for the target's language model as loaded by vispec/model/modeling_llama_kv.py).

`structured=True` builds a *successor* pair: the target's greedy next token is succ(t) and the draft
(which has its own embedding table, cnets_ours.py:683) agrees with it except for a fraction `rho` of
vocabulary rows.  All layers keep random N(0, 0.02) weights, so the arithmetic per round is that of a real
model, while the accept path (evaluate_posterior / update_inference_inputs) is exercised with a
controllable, *measured* acceptance length instead of the tau≈0 a fully random pair gives.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def bf16_grid(x: np.ndarray) -> np.ndarray:
    """Round float32 values to the nearest-even bf16 value, kept as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32).reshape(x.shape)


def succ_table(vocab: int, lo: int = 3, hi: int | None = None) -> np.ndarray:
    """succ(t): cyclic +1 on [lo, hi); everything else jumps to lo.  Never produces ids < lo (EOS = 2)."""
    hi = vocab if hi is None else hi
    t = np.arange(vocab, dtype=np.int64)
    s = np.where((t >= lo) & (t < hi), lo + (t - lo + 1) % (hi - lo), lo)
    return s


def make_target_weights(D, H, I, V, NL, seed=0, std=0.02, structured=False, layer_gain=None,
                        embed_std=None, succ_hi=None, qkv_bias=False, H_kv=None, head_gain=20.0) -> Dict[str, np.ndarray]:
    """structured: embeddings have unit variance, lm_head[succ(t)] = E[t] * head_gain / D (a confident successor
    model: correct logit ~ head_gain), and the residual-writing projections are damped by `layer_gain` (0.1)."""
    if layer_gain is None:
        layer_gain = 0.1 if structured else 1.0
    if embed_std is None:
        embed_std = 1.0 if structured else std
    rng = np.random.default_rng(seed)
    H_kv = H if H_kv is None else H_kv
    hd = D // H
    n = lambda *s, sd=std: bf16_grid(rng.standard_normal(s, dtype=np.float32) * np.float32(sd))
    w: Dict[str, np.ndarray] = {}
    E = n(V, D, sd=embed_std)
    w["model.embed_tokens.weight"] = E
    for i in range(NL):
        p = f"model.layers.{i}."
        w[p + "self_attn.q_proj.weight"] = n(H * hd, D)
        w[p + "self_attn.k_proj.weight"] = n(H_kv * hd, D)
        w[p + "self_attn.v_proj.weight"] = n(H_kv * hd, D)
        if qkv_bias:
            w[p + "self_attn.q_proj.bias"] = n(H * hd)
            w[p + "self_attn.k_proj.bias"] = n(H_kv * hd)
            w[p + "self_attn.v_proj.bias"] = n(H_kv * hd)
        w[p + "self_attn.o_proj.weight"] = n(D, H * hd, sd=std * layer_gain)
        w[p + "mlp.gate_proj.weight"] = n(I, D)
        w[p + "mlp.up_proj.weight"] = n(I, D)
        w[p + "mlp.down_proj.weight"] = n(D, I, sd=std * layer_gain)
        w[p + "input_layernorm.weight"] = bf16_grid(1.0 + 0.1 * rng.standard_normal(D, dtype=np.float32))
        w[p + "post_attention_layernorm.weight"] = bf16_grid(1.0 + 0.1 * rng.standard_normal(D, dtype=np.float32))
    w["model.norm.weight"] = bf16_grid(1.0 + 0.1 * rng.standard_normal(D, dtype=np.float32))
    if structured:
        s = succ_table(V, hi=succ_hi)
        head = n(V, D)  # rows never reached stay random
        g = np.float32(head_gain / D)
        head[s] = E * g  # lm_head[succ(t)] = E[t]*g  (duplicates of succ(.)==lo overwrite; fixed next line)
        lo_src = 3 + ((succ_hi or V) - 3) - 1  # the in-range predecessor of `lo`
        head[3] = E[lo_src] * g
        w["lm_head.weight"] = bf16_grid(head)
    else:
        w["lm_head.weight"] = n(V, D)
    return w


def make_draft_weights(D, H, I, V, num_q=2, seed=1, std=0.02, bias=True, qkv_bias=False, structured=False,
                       target_embed: np.ndarray | None = None, rho=0.115, layer_gain=None) -> Dict[str, np.ndarray]:
    if layer_gain is None:
        layer_gain = 0.1 if structured else 1.0
    rng = np.random.default_rng(seed)
    hd = D // H
    n = lambda *s, sd=std: bf16_grid(rng.standard_normal(s, dtype=np.float32) * np.float32(sd))
    w: Dict[str, np.ndarray] = {}
    if structured:
        assert target_embed is not None
        E = target_embed.copy()
        wrong = np.nonzero(rng.random(V) < rho)[0]
        E[wrong] = target_embed[rng.permutation(V)[: wrong.size]]
        w["embed_tokens.weight"] = bf16_grid(E)
    else:
        w["embed_tokens.weight"] = n(V, D)
    p = "layers.0."
    for nm in ("q_proj", "k_proj", "v_proj"):
        w[p + f"self_attn.{nm}.weight"] = n(D, D)
        if qkv_bias:
            w[p + f"self_attn.{nm}.bias"] = n(D)
    w[p + "self_attn.o_proj.weight"] = n(D, D, sd=std * layer_gain)
    w[p + "mlp.gate_proj.weight"] = n(I, D)
    w[p + "mlp.up_proj.weight"] = n(I, D)
    w[p + "mlp.down_proj.weight"] = n(D, I, sd=std * layer_gain)
    w[p + "post_attention_layernorm.weight"] = bf16_grid(1.0 + 0.1 * rng.standard_normal(D, dtype=np.float32))
    if structured:
        fcw = n(D, 2 * D, sd=std * layer_gain)
        fcw[:, :D] += np.eye(D, dtype=np.float32)
        w["fc.weight"] = bf16_grid(fcw)
        imw = n(D, 2 * D, sd=std * layer_gain)
        imw[:, :D] += np.eye(D, dtype=np.float32)
        w["img_fc.weight"] = bf16_grid(imw)
    else:
        w["fc.weight"] = n(D, 2 * D)
        w["img_fc.weight"] = n(D, 2 * D)
    if bias:
        w["fc.bias"] = n(D)
        w["img_fc.bias"] = n(D)
    w["imadpt.q"] = n(num_q, H, hd, sd=hd ** -0.5)
    w["imadpt.k_proj.weight"] = n(D, D)
    w["imadpt.v_proj.weight"] = n(D, D)
    if qkv_bias:
        w["imadpt.k_proj.bias"] = n(D)
        w["imadpt.v_proj.bias"] = n(D)
    w["imadpt.o_proj.weight"] = n(D, D)
    return w


# The tiny configuration shared by the golden fixtures, the oracle tests and the GPU parity tests.
# head_dim is 128 (what the HIP attention tiles are written for); every GEMM dim is a multiple of 64/16.
TINY = dict(D=256, H=2, I=704, V=1008, NL=2, max_pos=512)
# Qwen2.5-VL-shaped tiny target: GQA (4 query / 2 kv heads), q/k/v bias, theta 1e6, multimodal rotary sections, eps 1e-6
QWEN_TINY = dict(D=512, H=4, Hkv=2, I=704, V=1024, NL=2, max_pos=512, mrope_section=(16, 24, 24), theta=1e6, eps=1e-6)


def qwen_rope_index(input_ids: np.ndarray, image_token_id: int, grids, spatial_merge: int = 2, video_token_id: int | None = None,
                    video_grids=(), second_per_grid_ts=None, tokens_per_second: float = 2.0):
    """3-component (t, h, w) position ids [3, L] and rope_delta of a Qwen2.5-VL prompt — HF's / the reference's
    Qwen2_5_VLForConditionalGeneration.get_rope_index (modeling_qwen2_5_vl_kv.py:1789-1975, called by the reference's prefill):
    text tokens advance all three components together; a vision run of grid (t, h, w) covers t * (h/merge) * (w/merge) tokens with
    h-index, w-index over the merged grid and temporal index  floor(frame * second_per_grid_t * tokens_per_second)  — 0 for every
    frame of an IMAGE (second_per_grid_t = 0 there, :1899), second_per_grid_ts[v] (default 1.0, :1909-1912) for VIDEO v — all offset
    by the running position; the next text token continues from max + 1 (:1926-1930).
    grids / video_grids: (t, h, w) in patches, one per image / video run, in prompt order."""
    L = input_ids.shape[0]
    pos = np.zeros((3, L), np.int64)
    i, st, gi, vi = 0, 0, 0, 0
    while i < L:
        tok = input_ids[i]
        is_img = tok == image_token_id
        is_vid = video_token_id is not None and tok == video_token_id
        if is_img or is_vid:
            if is_img:
                t, h, w = grids[gi]
                gi += 1
                sec = 0.0
            else:
                t, h, w = video_grids[vi]
                sec = 1.0 if second_per_grid_ts is None else float(second_per_grid_ts[vi])
                vi += 1
            lh, lw = h // spatial_merge, w // spatial_merge
            n = t * lh * lw
            # float32 like torch (arange(int64) * python float / fp32 tensor element -> float32 tensor, then .long() truncates)
            frame_t = (np.arange(t, dtype=np.float32) * np.float32(sec) * np.float32(tokens_per_second)).astype(np.int64)
            tt = np.repeat(frame_t, lh * lw)
            hh = np.tile(np.repeat(np.arange(lh), lw), t)
            ww = np.tile(np.arange(lw), t * lh)
            pos[:, i : i + n] = np.stack([tt, hh, ww]) + st
            st = int(pos[:, i : i + n].max()) + 1
            i += n
        else:
            pos[:, i] = st
            st += 1
            i += 1
    return pos, int(pos.max()) + 1 - L


def make_request(V, D, L_text_pre, n_img, L_text_post, seed, image_token_id=None, embed: np.ndarray | None = None,
                 text_std=0.02, img_std=0.05):
    """One synthetic (image, prompt) request in the SURVEY.md §8(d) recipe:
    ids [L], inputs_embeds [L,D] (bf16 grid), image_mask [L]."""
    rng = np.random.default_rng(1000 + seed)
    L = L_text_pre + n_img + L_text_post
    hi = V if image_token_id is None else min(V, image_token_id)
    ids = rng.integers(3, hi, size=L).astype(np.int64)
    mask = np.zeros(L, bool)
    mask[L_text_pre : L_text_pre + n_img] = True
    if image_token_id is not None:
        ids[mask] = image_token_id
    if embed is not None:
        emb = embed[np.minimum(ids, embed.shape[0] - 1)].copy()
    else:
        emb = rng.standard_normal((L, D), dtype=np.float32) * np.float32(text_std)
    emb[mask] = rng.standard_normal((n_img, D), dtype=np.float32) * np.float32(img_std)
    return ids, bf16_grid(emb), mask
