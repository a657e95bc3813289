"""Synthetic LLaVA-v1.6-vicuna-7B-shaped weights created directly on the GPU (bench.py, smoke): same construction as
vispec_amd/synth.py (random N(0,0.02) layers; optional successor structure so that acceptance is MEASURED, not scripted),
but with the torch generator so 7 B parameters take seconds, already fused in the streaming layout."""
from __future__ import annotations

import torch

from .engine import DraftConfig, DraftWeightsDev, TargetConfig, TargetWeights
from .synth import succ_table


def _n(gen, shape, std, device):
    return (torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std).to(torch.bfloat16)


def make_pair(tcfg: TargetConfig, dcfg: DraftConfig, device, seed=0, structured=True, rho=0.115, num_q=2, layer_gain=0.1,
              head_gain=20.0, succ_hi=32000):
    """-> (TargetWeights, DraftWeightsDev).  rho=0.115 makes the draft agree with the target on ~88.5 % of the tokens,
    which gives a mean accept length ~2.98 at depth 3 (README.md:186 of the reference) — measured, see bench.py."""
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed)
    D, H, Hk, I, V, hd = tcfg.hidden_size, tcfg.num_heads, tcfg.num_kv_heads, tcfg.intermediate_size, tcfg.vocab_size, tcfg.head_dim
    std = 0.02
    lg = layer_gain if structured else 1.0
    tw = TargetWeights(tcfg, device)
    E = _n(g, (V, D), 1.0 if structured else std, device)
    tw.embed = E
    tw.norm = (1.0 + 0.1 * torch.randn(D, generator=g, device=device)).to(torch.bfloat16)
    for _ in range(tcfg.num_layers):
        tw.layers.append(dict(
            wqkv=_n(g, ((H + 2 * Hk) * hd, D), std, device), bqkv=_n(g, ((H + 2 * Hk) * hd,), std, device) if tcfg.qkv_bias else None,
            wo=_n(g, (D, H * hd), std * lg, device), wgu=_n(g, (2 * I, D), std, device), wdown=_n(g, (D, I), std * lg, device),
            ln1=(1.0 + 0.1 * torch.randn(D, generator=g, device=device)).to(torch.bfloat16),
            ln2=(1.0 + 0.1 * torch.randn(D, generator=g, device=device)).to(torch.bfloat16)))
    head = _n(g, (V, D), std, device)
    if structured:
        hi = min(succ_hi, V)
        s = torch.from_numpy(succ_table(V, hi=hi)).to(device)
        t = torch.arange(3, hi, device=device)
        head[s[t]] = (E[t].float() * (head_gain / D)).to(torch.bfloat16)
    tw.lm_head = head.contiguous()
    Dd, Hd, Id = dcfg.hidden_size, dcfg.num_heads, dcfg.intermediate_size
    dw = DraftWeightsDev(dcfg, num_q, device)
    t = dw.t
    if structured:
        Ed = E.clone()
        wrong = torch.nonzero(torch.rand(V, generator=g, device=device) < rho)[:, 0]
        Ed[wrong] = E[torch.randperm(V, generator=g, device=device)[: wrong.numel()]]
        t["embed"] = Ed
    else:
        t["embed"] = _n(g, (V, Dd), std, device)
    eye = torch.eye(Dd, device=device)
    def cat_eye(w):
        w = w.float()
        w[:, :Dd] += eye
        return w.to(torch.bfloat16)
    t["fc_w"] = cat_eye(_n(g, (Dd, 2 * Dd), std * lg, device)) if structured else _n(g, (Dd, 2 * Dd), std, device)
    t["fc_b"] = _n(g, (Dd,), std, device) if dcfg.bias else None
    t["imgfc_w"] = cat_eye(_n(g, (Dd, 2 * Dd), std * lg, device)) if structured else _n(g, (Dd, 2 * Dd), std, device)
    t["imgfc_b"] = _n(g, (Dd,), std, device) if dcfg.bias else None
    t["wqkv"] = _n(g, (3 * Dd, Dd), std, device)
    t["bqkv"] = _n(g, (3 * Dd,), std, device) if dcfg.qkv_bias else None
    t["wo"] = _n(g, (Dd, Dd), std * lg, device)
    t["wgu"] = _n(g, (2 * Id, Dd), std, device)
    t["wdown"] = _n(g, (Dd, Id), std * lg, device)
    t["ln2"] = (1.0 + 0.1 * torch.randn(Dd, generator=g, device=device)).to(torch.bfloat16)
    t["ad_q"] = _n(g, (num_q, Dd), (Dd // Hd) ** -0.5, device)
    t["ad_wkv"] = _n(g, (2 * Dd, Dd), std, device)
    t["ad_bkv"] = _n(g, (2 * Dd,), std, device) if dcfg.qkv_bias else None
    t["ad_wo"] = _n(g, (Dd, Dd), std, device)
    return tw, dw


def make_request_ids(vocab_hi: int, n_pre: int, n_img: int, n_post: int, seed: int, image_token_index: int):
    """input_ids of one synthetic (image, prompt) request: SURVEY.md §8(d) — template tokens, one image run, text tokens."""
    g = torch.Generator().manual_seed(1000 + seed)
    L = n_pre + n_img + n_post
    ids = torch.randint(3, vocab_hi, (L,), generator=g)
    ids[n_pre : n_pre + n_img] = image_token_index
    return ids
