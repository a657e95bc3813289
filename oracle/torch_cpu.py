"""PyTorch-CPU back end of the oracle — TEST / MEASUREMENT INFRASTRUCTURE, never part of the product path (only tests/ and
bench.py's `cpu_baseline` leg import it).

`oracle/vispec_oracle.py` restates the reference's draft-and-verify path in numpy; its elementary ops live in `Ops`.  `TorchOps`
re-implements exactly those ops with PyTorch CPU kernels (oneDNN / MKL GEMMs, fused SDPA) on all host cores, so that the SAME
restatement — same call graph, same functions citing the reference file:line — runs the way the reference itself runs on a CPU:
PyTorch ops, `torch.set_num_threads(os.cpu_count())`.  fp32 arithmetic (optionally bf16 weights for
the linear layers); no bf16 rounding-point emulation — that is the numpy oracle's job.  Pinned like the numpy oracle: the
reference-captured token streams of tests/golden/g8_loop.npz are reproduced with this back end (tests/test_oracle_golden.py).

`timed_request` is the cpu_baseline leg: one request of the bench workload — target prefill, draft prefill with image-token
compression, then a bounded number of draft-and-verify rounds and of plain AR steps at the real context length — with per-phase wall
times and the MEASURED accept lengths of the same synthetic weight pair the GPU runs."""
from __future__ import annotations

import math
import os
import time
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import vispec_oracle as vo


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


class TorchOps(vo.Ops):
    """vispec_oracle.Ops on PyTorch-CPU (fp32; numpy in, numpy out, zero-copy both ways)."""

    def __init__(self, weight_dtype=torch.float32):
        super().__init__(bf16=False)
        self.weight_dtype = weight_dtype
        self._wcache: Dict[int, torch.Tensor] = {}

    def _w(self, W):
        key = id(W)
        t = self._wcache.get(key)
        if t is None:
            t = W if torch.is_tensor(W) else torch.from_numpy(np.asarray(W, np.float32))
            t = t.to(self.weight_dtype)
            self._wcache[key] = t
        return t

    def linear(self, x, W, b=None, a8=False):  # nn.Linear (a8: the numpy back end only)
        if a8:
            raise NotImplementedError("fp8 activations are restated on the numpy back end only")
        if isinstance(W, tuple):
            return super().linear(x, W, b)
        w = self._w(W)
        y = F.linear(_t(x).to(w.dtype), w).float()
        if b is not None:
            y = y + _t(b)
        return y.numpy()

    def rmsnorm(self, x, w, eps):  # cnets_ours.py:522-527 ; modeling_llama_kv.py:118-133
        xf = _t(x)
        var = xf.pow(2).mean(-1, keepdim=True)
        return (_t(w) * (xf * torch.rsqrt(var + eps))).numpy()

    def silu_mul(self, g, u):
        return (F.silu(_t(g)) * _t(u)).numpy()

    def add(self, a, b):
        return (_t(a) + _t(b)).numpy()

    def rope(self, x, cos, sin, pos):  # cnets_ours.py:104-119
        xt = _t(x)
        c = _t(cos[pos])[None]
        s = _t(sin[pos])[None]
        half = xt.shape[-1] // 2
        rot = torch.cat([-xt[..., half:], xt[..., :half]], dim=-1)
        return (xt * c + rot * s).numpy()

    def _attn(self, q, k, v, allow):
        o = F.scaled_dot_product_attention(_t(q)[None], _t(k)[None], _t(v)[None], attn_mask=torch.from_numpy(np.ascontiguousarray(allow))[None, None])
        return o[0].numpy()

    # fp32: the eager path (modeling_llama_kv.py:602-623) and the fused one (cnets_ours.py:428-433) are the same function of their inputs
    attn_sdpa = _attn
    attn_eager = _attn

    def log_softmax(self, x):
        return torch.log_softmax(_t(x), dim=-1).numpy()


def split_fused_target(tw, tcfg) -> Dict[str, np.ndarray]:
    """vispec_amd.engine.TargetWeights (device, fused q|k|v and gate|up rows) -> the reference's state-dict names, fp32 on the host."""
    f = lambda t: t.detach().to("cpu", torch.float32).numpy()
    hd = tcfg.head_dim

    def fw(lw, k):  # fp8 target weights: `k` holds the e4m3 codes, `k_scale` their per-row factors -> the de-quantised matrix
        w = f(lw[k])
        return w if lw.get(k + "_scale") is None else w * f(lw[k + "_scale"])[:, None]

    nq, nk = tcfg.num_heads * hd, tcfg.num_kv_heads * hd
    I = tcfg.intermediate_size
    head = f(tw.lm_head)
    if getattr(tw, "lm_head_scale", None) is not None:
        head = head * f(tw.lm_head_scale)[:, None]
    sd = {"model.embed_tokens.weight": f(tw.embed), "model.norm.weight": f(tw.norm), "lm_head.weight": head}
    for i, lw in enumerate(tw.layers):
        p = f"model.layers.{i}."
        wqkv, wgu = fw(lw, "wqkv"), fw(lw, "wgu")
        sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = wqkv[:nq], wqkv[nq:nq + nk], wqkv[nq + nk:]
        if lw.get("bqkv") is not None:
            b = f(lw["bqkv"])
            sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"] = b[:nq], b[nq:nq + nk], b[nq + nk:]
        sd[p + "self_attn.o_proj.weight"] = fw(lw, "wo")
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = wgu[:I], wgu[I:]
        sd[p + "mlp.down_proj.weight"] = fw(lw, "wdown")
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = f(lw["ln1"]), f(lw["ln2"])
    return sd


def split_fused_draft(dw) -> Dict[str, np.ndarray]:
    f = lambda t: t.detach().to("cpu", torch.float32).numpy()
    t = dw.t
    D, Id = t["wo"].shape[0], t["wdown"].shape[1]
    H = dw.cfg.num_heads
    sd = {"embed_tokens.weight": f(t["embed"]), "fc.weight": f(t["fc_w"]), "img_fc.weight": f(t["imgfc_w"]),
          "imadpt.q": f(t["ad_q"]).reshape(dw.num_q, H, D // H), "imadpt.o_proj.weight": f(t["ad_wo"]),
          "layers.0.post_attention_layernorm.weight": f(t["ln2"]), "layers.0.self_attn.o_proj.weight": f(t["wo"]),
          "layers.0.mlp.down_proj.weight": f(t["wdown"])}
    if t.get("fc_b") is not None:
        sd["fc.bias"] = f(t["fc_b"])
    if t.get("imgfc_b") is not None:
        sd["img_fc.bias"] = f(t["imgfc_b"])
    wqkv, wgu, wkv = f(t["wqkv"]), f(t["wgu"]), f(t["ad_wkv"])
    for j, n in enumerate("qkv"):
        sd[f"layers.0.self_attn.{n}_proj.weight"] = wqkv[j * D:(j + 1) * D]
    sd["layers.0.mlp.gate_proj.weight"], sd["layers.0.mlp.up_proj.weight"] = wgu[:Id], wgu[Id:]
    sd["imadpt.k_proj.weight"], sd["imadpt.v_proj.weight"] = wkv[:D], wkv[D:]
    if t.get("bqkv") is not None:
        b = f(t["bqkv"])
        for j, n in enumerate("qkv"):
            sd[f"layers.0.self_attn.{n}_proj.bias"] = b[j * D:(j + 1) * D]
    if t.get("ad_bkv") is not None:
        b = f(t["ad_bkv"])
        sd["imadpt.k_proj.bias"], sd["imadpt.v_proj.bias"] = b[:D], b[D:]
    return sd


def timed_request(target: "vo.TargetLlama", draft: "vo.DraftModel", input_ids, inputs_embeds, image_mask, rounds: int, ar_steps: int,
                  max_pos: int, position_ids=None, rope_delta: int = 0, prefilled=None, budget_s: float = 1e9, draft_sees_embeds: bool = True):
    """One request of the bench workload on the host cores: SpecModel.specgenerate's call sequence (spec_model_ours.py:247-547) for
    `rounds` greedy draft-and-verify rounds, then `ar_steps` plain AR steps (gen_baseline_answer_coco_caption.py:111-129) continuing from
    the same context.  -> dict of wall times and the measured accept lengths.
    draft_sees_embeds=False is LLaVA-1.5: the target consumes the merged embeddings while the draft embeds the ids with
    its own table and never compresses (no image mask).
    prefilled = (kv [2*layers, H_kv, L, hd], hidden [L, D], last_logits [V]): the target prefill was done elsewhere (bench.py hands over
    the GPU's: a 2704-token prefill is 36 TFLOP, minutes on host cores, and not what the steady-state rate measures) — the cache is
    seeded with it (KVCache.cat semantics) and the timed part starts at the draft prefill.  budget_s bounds the round loop (>= 2 rounds)."""
    c = target.cfg
    tick = time.perf_counter
    input_ids = np.asarray(input_ids, np.int64).copy()
    draft.reset_kv()
    pkv, pkv_data, cur_len = vo.initialize_past_key_values(c.num_layers, c.num_kv_heads, max_pos, c.head_dim)
    target.tree_mask = None
    t0 = tick()
    if prefilled is not None:
        kv, hidden, last_logits = prefilled
        Lp = kv.shape[2]
        pkv_data[0][:, 0, :, :Lp] = kv
        cur_len[...] = Lp
        logits = np.asarray(last_logits, np.float32)[None]
        hidden = np.asarray(hidden, np.float32)
    elif inputs_embeds is None:  # text target: the draft embeds the ids with its own table (cnets_ours.py:1099-1107)
        logits, hidden = target.forward(pkv, input_ids=input_ids, position_ids=position_ids)
    else:
        logits, hidden = target.forward(pkv, inputs_embeds=inputs_embeds, position_ids=position_ids)
    t_prefill = tick() - t0
    token = vo.argmax_first(logits[-1])
    t0 = tick()
    dt, ri, tm, tp = draft.topK_genrate(hidden, np.concatenate([input_ids, [token]]), target.lm_head,
                                        inputs_embeds=inputs_embeds if draft_sees_embeds else None, image_mask=image_mask if draft_sees_embeds else None)
    t_draft_prefill = tick() - t0
    st = vo.LoopState(input_ids, dt, ri, tm, tp)
    t_verify, t_draft = [], []
    t_loop = tick()
    for r_ in range(rounds):
        if r_ >= 2 and tick() - t_loop > budget_s:
            break
        target.tree_mask = st.tree_mask
        t0 = tick()
        lg, hidden_new = vo.tree_decoding(target, pkv, st.draft_tokens, st.tree_position_ids, st.input_ids.shape[0], st.retrieve_indices, rope_delta)
        candidates = np.concatenate([st.draft_tokens, [-1]])[st.retrieve_indices]
        best, acc, sample_p = vo.evaluate_posterior_greedy(lg, candidates)
        t_verify.append(tick() - t0)
        st.accept_lengths.append(acc)
        t0 = tick()
        token = vo.update_inference_inputs(st, candidates, best, acc, pkv_data, cur_len, hidden_new, sample_p, draft, target.lm_head)
        t_draft.append(tick() - t0)
    # plain AR from the same context: the pending token `token` then greedy continuation
    target.tree_mask = None
    t_ar = []
    n = st.input_ids.shape[0]
    for i in range(ar_steps):
        t0 = tick()
        lg, _ = target.forward(pkv, input_ids=np.asarray([token]), position_ids=np.asarray([n + i + rope_delta]))
        token = vo.argmax_first(lg[-1])
        t_ar.append(tick() - t0)
    return dict(prefill_s=t_prefill, draft_prefill_s=t_draft_prefill, verify_s=t_verify, draft_s=t_draft, ar_s=t_ar,
                accept_lengths=list(st.accept_lengths), context=int(n), tokens=st.input_ids)
