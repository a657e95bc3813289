"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not product code.

CPU restatement (numpy, float32) of the reference's ViSpec draft-and-verify hot path
(SURVEY.md §8 rows A4–A12, A14).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; the product path
(`vispec_amd/`) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
against golden vectors captured by importing the reference itself in the build
container (tests/golden/gen_golden.py, fixtures tests/golden/*.npz).

Two numeric modes:
  * bf16=False : plain float32 everywhere — compared with the reference run in fp32.
  * bf16=True  : float32 arithmetic with bf16 round-to-nearest-even applied at the points
                 where the reference's torch-bf16 graph materialises a bf16 tensor — this is
                 what the HIP kernels (bf16 storage, fp32 accumulate) are compared with.

All citations `file:line` are into /root/reference/vispec/model/.
Integer outputs (token ids, tree structure, accept lengths, KV lengths) are exact restatements.
Tie-breaking (not defined by torch.topk/argmax docs): value descending, then index ascending;
argmax = first maximal index.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

F32_MIN = np.finfo(np.float32).min


# --------------------------------------------------------------------------------------
# numeric helpers
# --------------------------------------------------------------------------------------
def bf16_round(x) -> np.ndarray:
    """float32 -> nearest-even bf16 -> float32 (what a torch bf16 tensor can hold)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return out.reshape(x.shape)


def e4m3_round(x) -> np.ndarray:
    """float32 -> nearest-even OCP e4m3fn value (4 exponent bits, bias 7, 3 mantissa bits, max 448, subnormal step 2^-9),
    returned as float32; magnitudes above 448 saturate."""
    x = np.asarray(x, np.float32)
    a = np.minimum(np.abs(x).astype(np.float64), 464.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    ulp = np.where(a >= 2.0 ** -6, 2.0 ** (e - 3), 2.0 ** -9)
    q = np.rint(a / ulp) * ulp  # numpy rint = round-half-to-even
    q = np.minimum(q, 448.0)
    return (np.sign(x) * q).astype(np.float32)


def quantize_fp8(W):
    """per-output-channel e4m3 quantisation used for BASELINE config 5: W ≈ scale[n] * q[n, k]  (fp32 arithmetic throughout)."""
    W = np.asarray(W, np.float32)
    scale = (np.maximum(np.abs(W).max(axis=1), np.float32(1e-12)) / np.float32(448.0)).astype(np.float32)
    return e4m3_round((W / scale[:, None]).astype(np.float32)), scale


def _ident(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float32)


def topk_desc(v: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """top-k of a 1-D array: (values, indices), value descending then index ascending."""
    v = np.asarray(v)
    order = np.lexsort((np.arange(v.shape[0]), -v.astype(np.float64)))[:k]
    return v[order], order.astype(np.int64)


def argmax_first(v: np.ndarray) -> int:
    return int(np.argmax(v))  # numpy returns the first maximal index


def rope_tables(head_dim: int, max_pos: int, theta: float = 10000.0) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin caches [max_pos, head_dim] float32 — cnets_ours.py:122-155, modeling_llama_kv.py:147-181."""
    inv_freq = (1.0 / (np.float32(theta) ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim)))).astype(
        np.float32
    )
    t = np.arange(max_pos, dtype=np.float32)
    freqs = np.outer(t, inv_freq).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


class Ops:
    """Elementary ops with the reference's rounding points (bf16 mode) or none (fp32 mode)."""

    def __init__(self, bf16: bool = False):
        self.bf16 = bf16
        self.rd = bf16_round if bf16 else _ident

    def linear(self, x, W, b=None, a8=False):
        if isinstance(W, tuple) and a8:
            # W8A8 (vispec_set_fp8_activations; no reference counterpart — the reference has no fp8 path): the activations are quantised too,
            # one dynamic scale per row: sx = max|x[m, :]| / 448, q = e4m3(x / sx); y = (q_x . q_w^T) * scale_w[n] * sx[m] (+ b).  Products of
            # e4m3 numbers are exact in fp32; the f8f6f4 MFMA sums the 64 products of a step after aligning them to the step's largest (measured,
            # tools/probe/f8f6f4_probe.hip: 8e-5 of the sum of |products| against the exact sum) and adds the result to an fp32 accumulator, so numpy
            # reproduces it up to that alignment loss and the summation order (the tests' tolerances: tests/test_fp8a8_gpu.py).
            x = np.asarray(x, np.float32)
            sx = (np.maximum(np.abs(x).max(axis=-1, keepdims=True), np.float32(1e-12)) / np.float32(448.0)).astype(np.float32)
            y = ((e4m3_round((x / sx).astype(np.float32)) @ W[0].T) * W[1][None, :]) * sx
            if b is not None:
                y = y + np.asarray(b, np.float32)
            return self.rd(y.astype(np.float32))
        if isinstance(W, tuple):  # (e4m3 values, per-output-channel scales): y = scale * (x · q^T) (+b), W8A16
            y = (np.asarray(x, np.float32) @ W[0].T) * W[1][None, :]
            if b is not None:
                y = y + np.asarray(b, np.float32)
            return self.rd(y.astype(np.float32))
        y = np.asarray(x, np.float32) @ np.asarray(W, np.float32).T
        if b is not None:
            y = y + np.asarray(b, np.float32)
        return self.rd(y)

    def rmsnorm(self, x, w, eps):
        # cnets_ours.py:522-527 ; modeling_llama_kv.py:118-133
        xf = np.asarray(x, np.float32)
        var = np.mean(xf * xf, axis=-1, keepdims=True, dtype=np.float32)
        y = xf * (np.float32(1.0) / np.sqrt(var + np.float32(eps)))
        return self.rd(np.asarray(w, np.float32) * self.rd(y))

    def silu_mul(self, g, u):
        # cnets_ours.py:508 ; act(gate) materialised, then * up
        g = np.asarray(g, np.float32)
        act = self.rd(g / (np.float32(1.0) + np.exp(-g)))
        return self.rd(act * np.asarray(u, np.float32))

    def add(self, a, b):
        return self.rd(np.asarray(a, np.float32) + np.asarray(b, np.float32))

    def rope(self, x, cos, sin, pos):
        """x [H,S,hd]; cos/sin tables [P,hd]; pos [S] -> rotated x  (cnets_ours.py:104-119)."""
        c = self.rd(cos[pos])[None, :, :]
        s = self.rd(sin[pos])[None, :, :]
        half = x.shape[-1] // 2
        rot = np.concatenate([-x[..., half:], x[..., :half]], axis=-1)
        return self.rd(self.rd(x * c) + self.rd(rot * s))

    def attn_sdpa(self, q, k, v, allow):
        """Fused-kernel semantics of F.scaled_dot_product_attention (cnets_ours.py:428-433, 649-654):
        scores and softmax in fp32, probabilities cast to the storage dtype for P·V.
        q [H,Sq,hd], k/v [H,Sk,hd], allow [Sq,Sk] bool."""
        hd = q.shape[-1]
        s = np.einsum("hqd,hkd->hqk", q, k).astype(np.float32) * np.float32(1.0 / math.sqrt(hd))
        s = np.where(allow[None], s, -np.inf).astype(np.float32)
        m = s.max(axis=-1, keepdims=True)
        p = np.exp(s - m).astype(np.float32)
        l = p.sum(axis=-1, keepdims=True, dtype=np.float32)
        if self.bf16:
            o = np.einsum("hqk,hkd->hqd", self.rd(p), v).astype(np.float32) / l
        else:
            o = np.einsum("hqk,hkd->hqd", p / l, v).astype(np.float32)
        return self.rd(o)

    def attn_eager(self, q, k, v, allow):
        """Eager attention of the KV-Llama target (modeling_llama_kv.py:602-623):
        matmul -> /sqrt(hd) -> +mask -> softmax(fp32) -> cast -> matmul."""
        hd = q.shape[-1]
        s = self.rd(np.einsum("hqd,hkd->hqk", q, k).astype(np.float32))
        s = self.rd(s / np.float32(math.sqrt(hd)))
        s = np.where(allow[None], s, F32_MIN).astype(np.float32)
        m = s.max(axis=-1, keepdims=True)
        p = np.exp(s - m).astype(np.float32)
        p = self.rd(p / p.sum(axis=-1, keepdims=True, dtype=np.float32))
        return self.rd(np.einsum("hqk,hkd->hqd", p, v).astype(np.float32))

    def log_softmax(self, x):
        x = np.asarray(x, np.float32)
        m = x.max(axis=-1, keepdims=True)
        lse = m + np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=np.float32))
        return self.rd(x - lse)


# --------------------------------------------------------------------------------------
# Draft model (cnets_ours.py) — weights use the reference state-dict names
# --------------------------------------------------------------------------------------
@dataclass
class DraftConfig:
    hidden_size: int
    num_heads: int
    intermediate_size: int
    vocab_size: int
    max_position_embeddings: int = 4096
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    total_token: int = 30
    depth: int = 3
    top_k: int = 8
    num_q: int = 2

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


class DraftModel:
    """Restatement of cnets_ours.Model (forward :817-1038, topK_genrate :1043-1238)."""

    def __init__(self, cfg: DraftConfig, weights: Dict[str, np.ndarray], bf16: bool = False, cos=None, sin=None):
        self.cfg = cfg
        self.w = {k: np.asarray(v, np.float32) for k, v in weights.items()}
        self.ops = Ops(bf16)
        if cos is None:
            cos, sin = rope_tables(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
        self.cos, self.sin = np.asarray(cos, np.float32), np.asarray(sin, np.float32)
        self.stable_kv = None  # (K [H,n_c,hd], V, real_len)
        self.last_img_hidden = np.zeros((1, cfg.hidden_size), np.float32)

    # -- cnets_ours.py:1040-1041
    def reset_kv(self):
        self.stable_kv = None

    def _b(self, name):
        return self.w.get(name)

    def _heads(self, x):
        S = x.shape[0]
        return x.reshape(S, self.cfg.num_heads, self.cfg.head_dim).transpose(1, 0, 2)

    # -- ImgAdaptor.forward, cnets_ours.py:630-661
    def imgadaptor(self, img_emd: np.ndarray) -> np.ndarray:
        o = self.ops
        k = self._heads(o.linear(img_emd, self.w["imadpt.k_proj.weight"], self._b("imadpt.k_proj.bias")))
        v = self._heads(o.linear(img_emd, self.w["imadpt.v_proj.weight"], self._b("imadpt.v_proj.bias")))
        q = o.rd(self.w["imadpt.q"]).transpose(1, 0, 2)  # [H, num_q, hd]
        allow = np.ones((q.shape[1], k.shape[1]), bool)
        a = o.attn_sdpa(q, k, v, allow)  # [H, num_q, hd]
        a = a.transpose(1, 0, 2).reshape(q.shape[1], self.cfg.hidden_size)
        return o.linear(a, self.w["imadpt.o_proj.weight"])

    # -- fc(cat(emb, img_fc(cat(h, g)))), cnets_ours.py:918-922, 982-988
    def fuse_inputs(self, emb: np.ndarray, hidden: np.ndarray, g: np.ndarray) -> np.ndarray:
        o = self.ops
        gg = np.broadcast_to(g, hidden.shape)
        h2 = o.linear(np.concatenate([hidden, gg], -1), self.w["img_fc.weight"], self._b("img_fc.bias"))
        return o.linear(np.concatenate([emb, h2], -1), self.w["fc.weight"], self._b("fc.bias"))

    # -- LlamaDecoderLayer index 0 (no input norm), cnets_ours.py:545-600, attention :323-463
    def layer(self, x, pos, past, allow):
        o, c = self.ops, self.cfg
        p = "layers.0."
        q = self._heads(o.linear(x, self.w[p + "self_attn.q_proj.weight"], self._b(p + "self_attn.q_proj.bias")))
        k = self._heads(o.linear(x, self.w[p + "self_attn.k_proj.weight"], self._b(p + "self_attn.k_proj.bias")))
        v = self._heads(o.linear(x, self.w[p + "self_attn.v_proj.weight"], self._b(p + "self_attn.v_proj.bias")))
        q = o.rope(q, self.cos, self.sin, pos)
        k = o.rope(k, self.cos, self.sin, pos)
        if past is not None:
            k = np.concatenate([past[0], k], axis=1)
            v = np.concatenate([past[1], v], axis=1)
        kv = (k, v, int(np.max(pos)) + 1)  # :416-418 real_len = position_ids.max()+1
        a = o.attn_sdpa(q, k, v, allow)
        a = a.transpose(1, 0, 2).reshape(x.shape[0], c.hidden_size)
        h = o.add(x, o.linear(a, self.w[p + "self_attn.o_proj.weight"]))
        n = o.rmsnorm(h, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
        g = o.linear(n, self.w[p + "mlp.gate_proj.weight"])
        u = o.linear(n, self.w[p + "mlp.up_proj.weight"])
        d = o.linear(o.silu_mul(g, u), self.w[p + "mlp.down_proj.weight"])
        return o.add(h, d), kv

    # -- prefill / compression branch, cnets_ours.py:879-975
    def compress(self, hidden, embeds, image_mask):
        """hidden/embeds [L,D] (embeds already shifted by one, :1081-1082); image_mask [L] bool (unshifted).
        Returns compressed inputs [L_c,D], their position ids [L_c], and row->compressed index map."""
        L = hidden.shape[0]
        q = self.cfg.num_q
        m1 = np.asarray(image_mask, bool)[1:]  # :880
        ends = np.concatenate([m1[:-1] & ~m1[1:], m1[-1:]]) if m1.size else m1  # :881-883
        last_img_ids = np.nonzero(ends)[0]
        g = np.zeros((1, self.cfg.hidden_size), np.float32)  # :914
        hs, ps = [], []
        start = 0
        for e in last_img_ids:
            end = int(e) + 1
            cur = m1[start:end]
            txt = start + np.nonzero(~cur)[0]
            img = start + np.nonzero(cur)[0]
            hs.append(self.fuse_inputs(embeds[txt], hidden[txt], g))  # :918-922 (previous g)
            adapted = self.imgadaptor(embeds[img])  # :924-927
            hs.append(adapted[:-1])  # :928
            g = adapted[-1:]  # :930
            ps += [txt, np.arange(end - q + 1, end)]  # :932-937
            start = end
        rest = np.arange(start, L)
        hs.append(self.fuse_inputs(embeds[rest], hidden[rest], g))  # :944-948
        ps.append(rest)
        self.last_img_hidden = g
        return np.concatenate(hs, 0), np.concatenate(ps, 0).astype(np.int64)

    def forward_prefill(self, hidden, embeds, image_mask):
        """-> (out_c [L_c,D], kv, pos_c).  out_c[-1] is what topK_genrate consumes (:1109)."""
        if image_mask is None:
            # :976-988 without image: g = 0 (LLaVA-1.5 semantics)
            self.last_img_hidden = np.zeros((1, self.cfg.hidden_size), np.float32)
            x = self.fuse_inputs(embeds, hidden, self.last_img_hidden)
            pos = np.arange(hidden.shape[0], dtype=np.int64)
        else:
            x, pos = self.compress(hidden, embeds, image_mask)
        Lc = x.shape[0]
        allow = np.tril(np.ones((Lc, Lc), bool))  # :971-975
        out, kv = self.layer(x, pos, None, allow)
        return out, kv, pos

    def forward_decode(self, hidden, ids, past, pos=None, tree_mask=None):
        """decode / tree branch (:976-988, mask :781-815).  hidden [S,D], ids [S], past (K,V,real_len);
        tree_mask [S, T1] bool applies to the trailing T1 key columns."""
        S = hidden.shape[0]
        n_past = past[0].shape[1]
        if pos is None:
            pos = np.arange(past[2], past[2] + S, dtype=np.int64)  # :862-867 uses real length
        emb = self.ops.rd(self.w["embed_tokens.weight"][ids])
        x = self.fuse_inputs(emb, hidden, self.last_img_hidden)
        allow = np.ones((S, n_past + S), bool)
        allow[:, n_past:] = np.tril(np.ones((S, S), bool))
        if tree_mask is not None:
            t1 = tree_mask.shape[1]
            allow[:, -t1:] &= tree_mask.astype(bool)
        return self.layer(x, pos, past, allow)

    # -- cnets_ours.py:1043-1238
    def topK_genrate(self, hidden_states, input_ids, head_w, inputs_embeds=None, image_mask=None, sampling=False):
        """hidden_states [S,D]; input_ids [n+1] (last = sampled root token); head_w [V,D] target lm_head.
        Returns draft_tokens [T], retrieve_indices [n_leaf, max_depth], tree_mask [T,T] bool, tree_position_ids [T]."""
        c, o = self.cfg, self.ops
        k, depth, total = c.top_k, c.depth, c.total_token - 1
        input_ids = np.asarray(input_ids, np.int64)
        sample_token = input_ids[-1]
        if inputs_embeds is not None:  # :1066-1082
            new = o.rd(self.w["embed_tokens.weight"][input_ids[inputs_embeds.shape[0]:]])
            inputs_embeds = np.concatenate([inputs_embeds[1:], new], 0)
        ids = input_ids[1:]  # :1084
        len_posi = ids.shape[0]  # :1087
        if self.stable_kv is not None:  # :1090-1097
            S = hidden_states.shape[0]
            out, kv = self.forward_decode(hidden_states, ids[-S:], self.stable_kv)
        else:
            if inputs_embeds is None:
                inputs_embeds = o.rd(self.w["embed_tokens.weight"][ids])
            out, kv, _ = self.forward_prefill(hidden_states, inputs_embeds, image_mask)
        self.stable_kv = kv  # :1108
        last = out[-1:]  # :1109
        last_p = o.log_softmax(o.linear(last, head_w))[0]  # :1111-1113
        p1, tok1 = topk_desc(last_p, k)  # :1114-1115
        scores = p1
        scores_list = [p1]
        parents_list = [np.zeros(1, np.int64)]
        ss_token = [tok1]
        in_ids = tok1
        in_h = np.repeat(last, k, axis=0)  # :1121
        tmask = np.eye(k, dtype=bool)  # :1122
        cs_idx = np.arange(k)
        self.level_debug = []
        for i in range(depth):  # :1126
            pos = np.full(k, len_posi + i, np.int64)  # :1128,1137
            out, kv = self.forward_decode(in_h, in_ids, kv, pos=pos, tree_mask=tmask)
            bias = 1 + k * k * max(0, i - 1) + (k if i > 0 else 0)  # :1139-1141
            parents_list.append(cs_idx + bias)
            lp = o.log_softmax(o.linear(out, head_w))  # [k,V]  :1145-1146
            tp = np.zeros((k, k), np.float32)
            ti = np.zeros((k, k), np.int64)
            for r in range(k):
                tp[r], ti[r] = topk_desc(lp[r], k)  # :1148-1149
            cu = o.rd(tp + scores[:, None])  # :1151
            cs_p, cs_idx = topk_desc(cu.reshape(-1), k)  # :1153-1154
            scores = cs_p
            out_ids = cs_idx // k  # :1157
            in_h = out[out_ids]  # :1158
            in_ids = ti.reshape(-1)[cs_idx]  # :1159
            ss_token.append(ti.reshape(-1))
            scores_list.append(cu.reshape(-1))
            tmask = np.concatenate([tmask[out_ids], np.eye(k, dtype=bool)], axis=1)  # :1163-1165 (dim 2 of [1,1,k,w] = the ROWS: node j inherits its parent's row)
            self.level_debug.append(dict(out=out, cu=cu, tok=ti, cs_idx=cs_idx.copy()))
        scores_all = np.concatenate(scores_list)  # :1167
        tokens_all = np.concatenate(ss_token)  # :1168
        parents_all = np.concatenate(parents_list)
        return build_tree(scores_all, tokens_all, parents_all, sample_token, total, k, sampling)


def build_tree(scores_all, tokens_all, parents_all, sample_token, total, k, sampling=False):
    """Tree post-processing, cnets_ours.py:1169-1238 (integer logic, exact)."""
    _, top_idx = topk_desc(scores_all, total)  # :1169-1170
    top_idx = np.sort(top_idx)  # :1171
    draft_tokens = np.concatenate([[sample_token], tokens_all[top_idx]]).astype(np.int64)  # :1173-1174
    draft_parents = parents_all[top_idx // k]  # :1176
    mask_index = np.searchsorted(top_idx, draft_parents - 1, side="left")  # :1177-1179
    mask_index[draft_parents == 0] = -1  # :1180
    mask_index = mask_index + 1  # :1181
    T = total + 1
    tree_mask = np.eye(T, dtype=bool)  # :1183
    tree_mask[:, 0] = True
    for i in range(total):  # :1185-1186
        tree_mask[i + 1] |= tree_mask[mask_index[i]]
    tree_position_ids = tree_mask.sum(1) - 1  # :1188
    max_depth = int(tree_position_ids.max()) + 1  # :1195
    noleaf = set(np.unique(mask_index).tolist())  # :1196
    leaf_num = total - (len(noleaf) - 1)  # :1197-1198
    retrieve = [[-1] * max_depth for _ in range(leaf_num)]  # :1200-1201
    rid = 0
    for i in range(T):  # :1206-1213
        if i not in noleaf:
            cid = i
            d = int(tree_position_ids[i])
            for j in reversed(range(d + 1)):
                retrieve[rid][j] = cid
                cid = int(mask_index[cid - 1])
            rid += 1
    if sampling:  # :1215-1224
        maxitem = total + 5
        retrieve = sorted(retrieve, key=lambda l: [x if x >= 0 else maxitem for x in l])
    return draft_tokens, np.asarray(retrieve, np.int64), tree_mask, tree_position_ids.astype(np.int64)


# --------------------------------------------------------------------------------------
# Pre-allocated KV cache (kv_cache.py) and KV-Llama target (modeling_llama_kv.py)
# --------------------------------------------------------------------------------------
class KVCache:
    """kv_cache.py:4-66 — view into the shared buffer + a length scalar."""

    def __init__(self, data: np.ndarray, current_length: np.ndarray):
        self.data = data  # [1, H_kv, max_pos, hd]
        self.current_length = current_length  # 0-d view

    @property
    def shape(self):
        return (self.data.shape[0], self.data.shape[1], int(self.current_length), self.data.shape[3])

    def cat(self, t: np.ndarray):
        n = int(self.current_length)
        self.data[:, :, n : n + t.shape[2]] = t
        self.current_length += t.shape[2]
        return self.data[:, :, : int(self.current_length)]


def initialize_past_key_values(num_layers, num_kv_heads, max_pos, head_dim):
    """kv_cache.py:69-166 (single device)."""
    data = np.zeros((2 * num_layers, 1, num_kv_heads, max_pos, head_dim), np.float32)
    cur = np.zeros(2 * num_layers, np.int64)
    pkv = [[KVCache(data[2 * i + j], cur[2 * i + j : 2 * i + j + 1].reshape(())) for j in range(2)] for i in range(num_layers)]
    return pkv, [data], cur


@dataclass
class TargetConfig:
    hidden_size: int
    num_heads: int
    num_kv_heads: int
    intermediate_size: int
    vocab_size: int
    num_layers: int
    max_position_embeddings: int = 8192
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    attn_impl: str = "eager"          # "eager": modeling_llama_kv.py:602-623 ; "sdpa": modeling_qwen2_5_vl_kv.py:1156-1163
    mrope_section: Optional[tuple] = None  # Qwen2.5-VL multimodal rotary sections, e.g. (16, 24, 24)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


def mrope_tables(pos3: np.ndarray, head_dim: int, theta: float, section) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin [S, head_dim] for 3-component (t, h, w) position ids [3, S] — modeling_qwen2_5_vl_kv.py rotary embedding +
    apply_multimodal_rotary_pos_emb: chunk i of sizes section*2 takes component i % 3."""
    inv_freq = (1.0 / (np.float32(theta) ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim)))).astype(np.float32)
    freqs = pos3.astype(np.float32)[:, :, None] * inv_freq[None, None, :]  # [3, S, hd/2]
    emb = np.concatenate([freqs, freqs], axis=-1)  # [3, S, hd]
    cos3, sin3 = np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)
    cos, sin = np.zeros(emb.shape[1:], np.float32), np.zeros(emb.shape[1:], np.float32)
    o = 0
    for i, w in enumerate(list(section) * 2):
        cos[:, o : o + w] = cos3[i % 3][:, o : o + w]
        sin[:, o : o + w] = sin3[i % 3][:, o : o + w]
        o += w
    return cos, sin


class TargetLlama:
    """KV-Llama target: modeling_llama_kv.py:527-653 (attention), :927-1080 (model), lm_head + .float() :1190-1197."""

    def __init__(self, cfg: TargetConfig, weights: Dict[str, np.ndarray], bf16: bool = False, cos=None, sin=None, fp8=False):
        """fp8: True = quantise every streamed GEMM weight (projections + lm_head) here; or a dict name -> (e4m3 values, scales)
        of already-quantised weights (the product's own codes, so that exact .5 ties of w/scale cannot differ)."""
        self.cfg = cfg
        self.w = {k: np.asarray(v, np.float32) for k, v in weights.items()}
        if fp8:
            for k in list(self.w):
                if k.endswith("_proj.weight") or k == "lm_head.weight":
                    self.w[k] = fp8[k] if isinstance(fp8, dict) else quantize_fp8(self.w[k])
        self.ops = Ops(bf16)
        if cos is None:
            cos, sin = rope_tables(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
        self.cos, self.sin = np.asarray(cos, np.float32), np.asarray(sin, np.float32)
        self.tree_mask = None  # [T,T] bool, installed by the loop (spec_model_ours.py:486-489)
        self.a8_decode = False  # True (fp8 weights only; the name is historical): every forward — prefill, tree verify, AR steps — quantises the
                                # activations of the q/k/v, gate/up and down projections too (Ops.linear a8=True; o_proj keeps bf16 activations)

    @property
    def lm_head(self):
        return self.w["lm_head.weight"]

    def embed(self, ids):
        return self.ops.rd(self.w["model.embed_tokens.weight"][np.asarray(ids, np.int64)])

    def forward(self, past_key_values, input_ids=None, inputs_embeds=None, position_ids=None):
        """-> (logits [S,V] fp32, hidden [S,D] post-final-norm).  Appends K/V to the cache (KVCache.cat)."""
        c, o = self.cfg, self.ops
        x = self.embed(input_ids) if inputs_embeds is None else o.rd(inputs_embeds)
        S = x.shape[0]
        n_past = past_key_values[0][0].shape[2]
        pos = np.arange(n_past, n_past + S, dtype=np.int64) if position_ids is None else np.asarray(position_ids, np.int64)
        if pos.ndim == 2:  # [3, S] multimodal rotary positions (Qwen2.5-VL)
            cosT, sinT = mrope_tables(pos, c.head_dim, c.rope_theta, c.mrope_section)
            pos_idx = np.arange(S)
        else:
            cosT, sinT, pos_idx = self.cos, self.sin, pos
        attn = o.attn_sdpa if c.attn_impl == "sdpa" else o.attn_eager
        allow = np.ones((S, n_past + S), bool)
        allow[:, n_past:] = np.tril(np.ones((S, S), bool))  # :892-900
        if self.tree_mask is not None:  # :917-922
            T = self.tree_mask.shape[-1]
            allow[-T:, -T:] &= self.tree_mask.astype(bool)
        rep = c.num_heads // c.num_kv_heads
        a8 = dict(a8=True) if self.a8_decode else {}
        for i in range(c.num_layers):
            p = f"model.layers.{i}."
            h = o.rmsnorm(x, self.w[p + "input_layernorm.weight"], c.rms_norm_eps)
            q = o.linear(h, self.w[p + "self_attn.q_proj.weight"], self.w.get(p + "self_attn.q_proj.bias"), **a8)
            k = o.linear(h, self.w[p + "self_attn.k_proj.weight"], self.w.get(p + "self_attn.k_proj.bias"), **a8)
            v = o.linear(h, self.w[p + "self_attn.v_proj.weight"], self.w.get(p + "self_attn.v_proj.bias"), **a8)
            q = q.reshape(S, c.num_heads, c.head_dim).transpose(1, 0, 2)
            k = k.reshape(S, c.num_kv_heads, c.head_dim).transpose(1, 0, 2)
            v = v.reshape(S, c.num_kv_heads, c.head_dim).transpose(1, 0, 2)
            q = o.rope(q, cosT, sinT, pos_idx)
            k = o.rope(k, cosT, sinT, pos_idx)
            kk = past_key_values[i][0].cat(k[None])[0]  # :583,593
            vv = past_key_values[i][1].cat(v[None])[0]
            if rep > 1:
                kk = np.repeat(kk, rep, axis=0)
                vv = np.repeat(vv, rep, axis=0)
            a = attn(q, kk, vv, allow).transpose(1, 0, 2).reshape(S, c.hidden_size)
            x = o.add(x, o.linear(a, self.w[p + "self_attn.o_proj.weight"]))  # (W8A8 leaves o_proj on bf16 activations: 5 % of a layer's weight
                                                                              #  bytes, and its input has no producer that sees a whole row)
            h = o.rmsnorm(x, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            g = o.linear(h, self.w[p + "mlp.gate_proj.weight"], **a8)
            u = o.linear(h, self.w[p + "mlp.up_proj.weight"], **a8)
            x = o.add(x, o.linear(o.silu_mul(g, u), self.w[p + "mlp.down_proj.weight"], **a8))
        hidden = o.rmsnorm(x, self.w["model.norm.weight"], c.rms_norm_eps)  # :1062
        logits = o.linear(hidden, self.w["lm_head.weight"])  # then .float()
        return logits.astype(np.float32), hidden


# --------------------------------------------------------------------------------------
# utils.py: initialize_tree / tree_decoding / evaluate_posterior / update_inference_inputs
# --------------------------------------------------------------------------------------
def evaluate_posterior_greedy(logits: np.ndarray, candidates: np.ndarray):
    """utils.py:438-451.  logits [n_leaf, m, V], candidates [n_leaf, m] -> (best, accept_length, sample_p [V])."""
    am = np.argmax(logits[:, :-1], axis=-1)
    posterior_mask = (candidates[:, 1:] == am).astype(np.int64)
    cal = np.cumprod(posterior_mask, axis=1).sum(axis=1)
    accept_length = int(cal.max()) if cal.size else 0
    best = 0 if accept_length == 0 else int(np.argmax(cal))
    return best, accept_length, logits[best, accept_length]


def uniform_hash(seed: int, a: int, b: int, c: int) -> float:
    """Counter-based uniform in [0,1): splitmix64 of (seed, a, b, c), top 24 bits.  The HIP sampling kernels use the same
    function (csrc/tree_kernels.h: vs_uniform), so oracle and device draw identical numbers."""
    M = (1 << 64) - 1
    z = (seed * 0x9E3779B97F4A7C15 + a * 0xBF58476D1CE4E5B9 + b * 0x94D049BB133111EB + c * 0xD6E8FEB86659FD93 + 0x2545F4914F6CDD1D) & M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return (z >> 40) / float(1 << 24)


def softmax_T(row: np.ndarray, temperature: float, top_k: int = 0) -> np.ndarray:
    """softmax(logits_processor(row)) in fp32 — utils.py:454-455 with the processor list of utils.py:39-55:
    TemperatureLogitsWarper, then TopKLogitsWarper when top_k > 0 (scores below the k-th largest become -inf; ties with the
    k-th value survive, as in HF).  TopP is not restated: HF's TopPLogitsWarper scatters along dim 1 and raises on the 3-D tree
    logits the reference hands it (RuntimeError: index out of bounds), i.e. the reference itself cannot run it."""
    x = np.asarray(row, np.float32) / np.float32(temperature)
    if top_k and top_k > 0:
        kth = np.sort(x)[::-1][min(int(top_k), x.shape[0]) - 1]
        x = np.where(x < kth, np.float32(-np.inf), x)
    e = np.exp(x - x.max())
    return (e / e.sum(dtype=np.float32)).astype(np.float32)


def multinomial_inverse_cdf(p: np.ndarray, u: float) -> int:
    """One draw from weights p (not necessarily normalised) by inverse CDF with a given uniform u — the deterministic stand-in
    for torch.multinomial(p, 1) (utils.py:288,551): same distribution, explicit randomness."""
    c = np.cumsum(np.asarray(p, np.float64))
    return int(min(np.searchsorted(c, u * c[-1], side="right"), len(p) - 1))


def evaluate_posterior_sampling(logits: np.ndarray, candidates: np.ndarray, temperature: float, uni, top_k: int = 0):
    """utils.py:453-493 — sequential rejection over the tree's children.  `uni(j, i)` supplies uni_dist[j, i]
    (the reference draws torch.rand_like(candidates)).  -> (best, accept_length, sample_p [V])."""
    n_leaf, m = candidates.shape
    logits_p = np.stack([[softmax_T(logits[j, c], temperature, top_k) for c in range(m)] for j in range(n_leaf)])
    accept_length = 1
    accept_cand = candidates[0].copy()
    best = 0
    adjust = False
    gtp = None
    for i in range(1, m):
        if i != accept_length:
            break
        adjust = False
        is_eq = (candidates[:, :accept_length] == accept_cand[:accept_length]).all(axis=1)
        fi = int(np.nonzero(is_eq)[0][0])
        gtp = logits_p[fi, i - 1].copy()
        seen = set()
        for j in range(n_leaf):
            if not is_eq[j]:
                continue
            xi = int(candidates[j, i])
            if xi == -1 or xi in seen:
                continue
            seen.add(xi)
            if uni(j, i) <= gtp[xi]:
                accept_cand[accept_length] = xi
                accept_length += 1
                best = j
                break
            gtp[xi] = 0
            gtp = gtp / gtp.sum(dtype=np.float32)
            adjust = True
    sample_p = gtp if (adjust and accept_length != m) else logits_p[best, accept_length - 1]
    return best, accept_length - 1, sample_p


def tree_decoding(target: TargetLlama, pkv, tree_candidates, tree_position_ids, n_ctx, retrieve_indices, rope_delta=0):
    """utils.py:389-412.  -> (logits [n_leaf, m, V], hidden_state_new [T,D]).  Qwen2.5-VL adds the cached rope_deltas and
    expands to 3 equal components (:397-402), which is ordinary 1-D rotary at the shifted position."""
    position_ids = tree_position_ids + n_ctx + rope_delta
    tree_logits, hidden = target.forward(pkv, input_ids=tree_candidates, position_ids=position_ids)
    return tree_logits[retrieve_indices], hidden  # -1 wraps to the last row exactly as torch indexing does


@dataclass
class LoopState:
    input_ids: np.ndarray
    draft_tokens: np.ndarray
    retrieve_indices: np.ndarray
    tree_mask: np.ndarray
    tree_position_ids: np.ndarray
    new_token: int = 0
    accept_lengths: List[int] = field(default_factory=list)


def update_inference_inputs(st: LoopState, candidates, best, accept_length, pkv_data, cur_len, hidden_state_new, sample_p,
                            draft: DraftModel, head_w, sample_u=None):
    """utils.py:496-593.  sample_u: None = greedy (argmax, :554); a uniform in [0,1) = multinomial(sample_p) (:551)."""
    prev = st.input_ids.shape[0]
    select = st.retrieve_indices[best, : accept_length + 1] + prev  # :516-518
    st.input_ids = np.concatenate([st.input_ids, candidates[best, : accept_length + 1]])  # :520-526
    for d in pkv_data:  # :529-538
        tgt = d[..., select, :].copy()
        d[..., prev : prev + tgt.shape[-2], :] = tgt
    cur_len[...] = prev + accept_length + 1  # :541
    accept_hidden = hidden_state_new[st.retrieve_indices[best, : accept_length + 1]]  # :543-546
    token = argmax_first(sample_p) if sample_u is None else multinomial_inverse_cdf(sample_p, sample_u)  # :554 / :551
    st.draft_tokens, st.retrieve_indices, st.tree_mask, st.tree_position_ids = draft.topK_genrate(
        accept_hidden, np.concatenate([st.input_ids, [token]]), head_w, sampling=sample_u is not None
    )
    st.new_token += accept_length + 1  # :582
    return token


def specgenerate(target: TargetLlama, draft: DraftModel, input_ids, inputs_embeds=None, image_mask=None,
                 max_new_tokens=512, max_length=2048, eos_token_id=2, max_pos=None, scripted_accept=None, position_ids=None,
                 rope_delta=0, temperature=0.0, seed=0, top_k=0, stop_token_id=None):
    """SpecModel.specgenerate, temperature 0 (spec_model_ours.py:247-582).
    -> (input_ids, new_token, idx, accept_lengths).  `scripted_accept` (bench-only knob, never used by
    parity tests) is None."""
    c = target.cfg
    max_length = max_length - (draft.cfg.total_token - 1) - 10  # :270
    input_ids = np.asarray(input_ids, np.int64).copy()
    draft.reset_kv()  # :283
    pkv, pkv_data, cur_len = initialize_past_key_values(c.num_layers, c.num_kv_heads, max_pos or c.max_position_embeddings, c.head_dim)
    input_len = input_ids.shape[0]
    target.tree_mask = None  # reset_tree_mode :456
    # initialize_tree, utils.py:266-327
    if inputs_embeds is None:
        logits, hidden = target.forward(pkv, input_ids=input_ids, position_ids=position_ids)
    else:
        logits, hidden = target.forward(pkv, inputs_embeds=inputs_embeds, position_ids=position_ids)
    sampling = temperature > 1e-5  # spec_model_ours.py:272-277
    if sampling:  # utils.py:284-288
        x = logits[-1] / np.float32(temperature)
        if top_k and top_k > 0:
            x = np.where(x < np.sort(x)[::-1][min(int(top_k), x.shape[0]) - 1], np.float32(-np.inf), x)
        token = multinomial_inverse_cdf(np.exp(x - x.max()), uniform_hash(seed, 0xFFFF, 0, 0))
    else:
        token = argmax_first(logits[-1])  # :290
    ids1 = np.concatenate([input_ids, [token]])
    dt, ri, tm, tp = draft.topK_genrate(hidden, ids1, target.lm_head, inputs_embeds=inputs_embeds, image_mask=image_mask, sampling=sampling)
    st = LoopState(input_ids, dt, ri, tm, tp)
    idx = 0
    for idx in range(max_length):  # :484
        target.tree_mask = st.tree_mask  # :486-489
        logits, hidden_new = tree_decoding(target, pkv, st.draft_tokens, st.tree_position_ids, st.input_ids.shape[0], st.retrieve_indices,
                                           rope_delta)
        ext = np.concatenate([st.draft_tokens, [-1]])  # :503
        candidates = ext[st.retrieve_indices]  # :504
        if sampling:
            rnd = len(st.accept_lengths)
            best, acc, sample_p = evaluate_posterior_sampling(logits, candidates, temperature, lambda j, i, r=rnd: uniform_hash(seed, r, j, i),
                                                              top_k)
            su = uniform_hash(seed, rnd, 255, 255)
        else:
            best, acc, sample_p = evaluate_posterior_greedy(logits, candidates)  # :505-507
            su = None
        st.accept_lengths.append(acc)
        update_inference_inputs(st, candidates, best, acc, pkv_data, cur_len, hidden_new, sample_p, draft, target.lm_head, sample_u=su)
        if stop_token_id is not None and stop_token_id in st.input_ids[input_len:].tolist():  # is_llama3, :540-542
            break
        if eos_token_id in st.input_ids[input_len:].tolist():  # :544
            break
        if st.new_token > max_new_tokens:  # :546
            break
    return st.input_ids, st.new_token, idx, st.accept_lengths


def baseline_forward(target: TargetLlama, input_ids, inputs_embeds=None, max_steps=2048, eos_token_id=2, max_pos=None,
                     position_ids=None, rope_delta=0):
    """evaluation/gen_baseline_answer_coco_caption.py:34-133 — greedy AR with the same KV cache."""
    c = target.cfg
    pkv, _, _ = initialize_past_key_values(c.num_layers, c.num_kv_heads, max_pos or c.max_position_embeddings, c.head_dim)
    target.tree_mask = None
    input_ids = np.asarray(input_ids, np.int64).copy()
    if inputs_embeds is None:
        logits, _ = target.forward(pkv, input_ids=input_ids, position_ids=position_ids)
    else:
        logits, _ = target.forward(pkv, inputs_embeds=inputs_embeds, position_ids=position_ids)
    out = input_ids
    for _ in range(max_steps):
        tok = argmax_first(logits[-1])
        out = np.concatenate([out, [tok]])
        if tok == eos_token_id:
            break
        logits, _ = target.forward(pkv, input_ids=np.asarray([tok]), position_ids=np.asarray([out.shape[0] - 1 + rope_delta]))
    return out
