"""Engine: owns one `vispec_ctx` (one per process/GPU), the device weights in the layout the HIP kernels stream
(q|k|v and gate|up fused once at load) and the KV buffers.  PyTorch is used here for device memory, streams and
the target's prefill GEMMs only; every decode-round op is a call into libvispec_hip."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import lib as L


@dataclass
class TargetConfig:
    hidden_size: int
    num_heads: int
    num_kv_heads: int
    intermediate_size: int
    vocab_size: int
    num_layers: int
    max_position_embeddings: int = 8192  # modeling_llava_next_kv.py:10 forces 8192 for the LLaVA family
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    qkv_bias: bool = False
    architectures: tuple = ("LlavaNextForConditionalGeneration",)
    image_token_index: int = 32000
    eos_token_id: int = 2
    attn_impl: str = "eager"          # LLaVA-family KV-Llama is eager (modeling_llama_kv.py:602-623); Qwen2.5-VL runs SDPA
    mrope_section: Optional[tuple] = None  # Qwen2.5-VL multimodal rotary sections (16, 24, 24)
    video_token_id: int = -1
    tokens_per_second: float = 2.0    # Qwen2.5-VL vision_config.tokens_per_second: temporal rotary index of video frames (get_rope_index)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


@dataclass
class DraftConfig:
    hidden_size: int
    num_heads: int
    intermediate_size: int
    vocab_size: int
    max_position_embeddings: int = 4096  # vispec/train/llava_1.6_7B_config.json
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    qkv_bias: bool = False
    bias: bool = True  # fc / img_fc bias (spec_model_ours.py:59-64)


QWEN25_VL_7B = dict(hidden_size=3584, num_heads=28, num_kv_heads=4, intermediate_size=18944, vocab_size=152064, num_layers=28,
                    max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=1e6, qkv_bias=True,
                    architectures=("Qwen2_5_VLForConditionalGeneration",), image_token_index=151655, video_token_id=151656,
                    eos_token_id=151645, attn_impl="sdpa", mrope_section=(16, 24, 24))
LLAVA_16_13B = dict(hidden_size=5120, num_heads=40, num_kv_heads=40, intermediate_size=13824, vocab_size=32064, num_layers=40)
LLAVA_16_7B = dict(hidden_size=4096, num_heads=32, num_kv_heads=32, intermediate_size=11008, vocab_size=32064, num_layers=32)


def rope_tables(head_dim: int, max_pos: int, theta: float, device, dtype=torch.bfloat16):
    """cos/sin caches exactly as the reference builds them (cnets_ours.py:122-155, modeling_llama_kv.py:147-181):
    fp32 on the host, then cast to the model dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype).to(device).contiguous(), emb.sin().to(dtype).to(device).contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """Row-major nn.Linear weight [N, K] (bf16, on the GPU) -> the W32 tile layout the skinny GEMM streams
    (include/vispec_hip.h: vispec_pack_weight).  One-time, at load."""
    lib = L.load()
    assert w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 2 and w.is_contiguous()
    N, K = w.shape
    out = torch.empty(int(lib.vispec_packed_elems(N, K)), dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        L.check(lib.vispec_pack_weight(None, C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream), _p(w), N, K, _p(out)))
    return out


def qkv_rope_order(w: torch.Tensor, n_rope_heads: int, head_dim: int = 128) -> torch.Tensor:
    """Reorder the q and k rows of a fused q|k|v weight (or code matrix) into the "rope order" the one-launch
    projection + rotary + KV-append kernel expects (include/vispec_hip.h: vispec_gemm_qkv_rope): within each of the first
    `n_rope_heads` heads, row 32t + c takes natural row 16t + c (c < 16) or 64 + 16t + (c - 16).  Identity when the library
    reports that it will run the two-kernel path for this row count (vispec_qkv_rope_fused)."""
    lib = L.load()
    assert head_dim == 128
    if not lib.vispec_qkv_rope_fused(int(w.shape[0])):
        return w
    r = torch.arange(w.shape[0], device=w.device)
    h, t, c = r // 128, (r % 128) // 32, r % 32
    src = h * 128 + torch.where(c < 16, 16 * t + c, 64 + 16 * t + (c - 16))
    src = torch.where(r < n_rope_heads * 128, src, r)
    return w.index_select(0, src).contiguous()


def swiglu_order(w: torch.Tensor) -> torch.Tensor:
    """gate|up weight (or code matrix) [2I, K] -> "SwiGLU order" (include/vispec_hip.h: vispec_pack_weight): packed row 32t + c is
    gate row 16t + c (c < 16) or up row I + 16t + (c - 16)."""
    I = w.shape[0] // 2
    assert w.shape[0] == 2 * I and I % 16 == 0
    r = torch.arange(2 * I, device=w.device)
    t, c = r // 32, r % 32
    src = torch.where(c < 16, 16 * t + c, I + 16 * t + (c - 16))
    return w.index_select(0, src).contiguous()


E4M3_MAX = 448.0


def quantize_fp8(w: torch.Tensor):
    """Row-major bf16 weight [N, K] -> (e4m3fn codes uint8 [N, K], per-output-channel fp32 scales [N]); W ≈ scale[n] * q[n, k]."""
    wf = w.float()
    scale = (wf.abs().amax(dim=1).clamp_min(1e-12) / E4M3_MAX).contiguous()
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale


def pack_weight_fp8(q_u8: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    assert q_u8.is_cuda and q_u8.dtype == torch.uint8 and q_u8.dim() == 2 and q_u8.is_contiguous()
    N, K = q_u8.shape
    out = torch.empty(((N + 31) // 32) * 32 * K, dtype=torch.uint8, device=q_u8.device)
    with torch.cuda.device(q_u8.device):
        L.check(lib.vispec_pack_weight_fp8(None, C.c_void_p(torch.cuda.current_stream(q_u8.device).cuda_stream), _p(q_u8), N, K, _p(out)))
    return out


class TargetWeights:
    """Target language-model weights on the device, fused for streaming: wqkv [ (H+2Hkv)*hd, D ], wgu [2I, D]."""

    def __init__(self, cfg: TargetConfig, device):
        self.cfg, self.device = cfg, device
        self.embed = self.norm = self.lm_head = None
        self.lm_head_scale = None  # fp8 target weights: lm_head then holds the e4m3 codes (as bf16) and this their per-row scales
        self.layers = []  # dicts: wqkv, bqkv, wo, wgu, wdown, ln1, ln2

    @classmethod
    def from_state_dict(cls, cfg: TargetConfig, sd: Dict[str, "np.ndarray | torch.Tensor"], device, prefix="model."):
        self = cls(cfg, device)
        g = lambda k: (torch.from_numpy(np.ascontiguousarray(sd[k])) if isinstance(sd[k], np.ndarray) else sd[k]).to(
            device=device, dtype=torch.bfloat16).contiguous()
        has = lambda k: k in sd
        self.embed = g(prefix + "embed_tokens.weight")
        self.norm = g(prefix + "norm.weight")
        self.lm_head = g("lm_head.weight")
        for i in range(cfg.num_layers):
            p = f"{prefix}layers.{i}."
            lw = dict(
                wqkv=torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"), g(p + "self_attn.v_proj.weight")], 0).contiguous(),
                bqkv=(torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"), g(p + "self_attn.v_proj.bias")], 0).contiguous()
                      if has(p + "self_attn.q_proj.bias") else None),
                wo=g(p + "self_attn.o_proj.weight"),
                wgu=torch.cat([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], 0).contiguous(),
                wdown=g(p + "mlp.down_proj.weight"),
                ln1=g(p + "input_layernorm.weight"),
                ln2=g(p + "post_attention_layernorm.weight"),
            )
            self.layers.append(lw)
        return self

    def tensors(self):
        yield self.embed
        yield self.norm
        yield self.lm_head
        for lw in self.layers:
            for v in lw.values():
                if v is not None:
                    yield v

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.tensors())


class DraftWeightsDev:
    """Draft weights on the device (SURVEY §8 A0 names), fused like the target's."""

    def __init__(self, cfg: DraftConfig, num_q: int, device):
        self.cfg, self.num_q, self.device = cfg, num_q, device
        self.t: Dict[str, Optional[torch.Tensor]] = {}

    @classmethod
    def from_state_dict(cls, cfg: DraftConfig, sd, num_q: int, device):
        self = cls(cfg, num_q, device)
        g = lambda k: (torch.from_numpy(np.ascontiguousarray(sd[k])) if isinstance(sd[k], np.ndarray) else sd[k]).to(
            device=device, dtype=torch.bfloat16).contiguous()
        has = lambda k: k in sd
        p = "layers.0."
        t = self.t
        t["embed"] = g("embed_tokens.weight")
        t["fc_w"], t["fc_b"] = g("fc.weight"), (g("fc.bias") if has("fc.bias") else None)
        t["imgfc_w"], t["imgfc_b"] = g("img_fc.weight"), (g("img_fc.bias") if has("img_fc.bias") else None)
        t["wqkv"] = torch.cat([g(p + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous()
        t["bqkv"] = torch.cat([g(p + f"self_attn.{n}_proj.bias") for n in "qkv"], 0).contiguous() if has(p + "self_attn.q_proj.bias") else None
        t["wo"] = g(p + "self_attn.o_proj.weight")
        t["wgu"] = torch.cat([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], 0).contiguous()
        t["wdown"] = g(p + "mlp.down_proj.weight")
        t["ln2"] = g(p + "post_attention_layernorm.weight")
        t["ad_q"] = g("imadpt.q").reshape(num_q, -1).contiguous()
        t["ad_wkv"] = torch.cat([g("imadpt.k_proj.weight"), g("imadpt.v_proj.weight")], 0).contiguous()
        t["ad_bkv"] = torch.cat([g("imadpt.k_proj.bias"), g("imadpt.v_proj.bias")], 0).contiguous() if has("imadpt.k_proj.bias") else None
        t["ad_wo"] = g("imadpt.o_proj.weight")
        if t["ad_q"].shape[0] != num_q:
            raise ValueError("imadpt.q does not match num_q")
        return self

    def tensors(self):
        for v in self.t.values():
            if v is not None:
                yield v


class Engine:
    """One per (process, GPU).  Not re-entrant."""

    def __init__(self, tcfg: TargetConfig, dcfg: DraftConfig, tw: TargetWeights, dw: DraftWeightsDev, total_token=30, depth=3,
                 top_k=8, num_q=2, kv_max_pos: Optional[int] = None, draft_max_pos: Optional[int] = None, eager_scores=None,
                 target_weight_dtype: str = "bf16", leader: Optional["Engine"] = None):
        """leader: build this engine as a COHORT MEMBER of `leader` (same configs and weights): its activation workspaces alias one
        32-row tile of the leader's (up to seven members per leader), so `leader.cohort_round([members...])` runs all their rounds on one
        weight pass."""
        if eager_scores is None:
            eager_scores = tcfg.attn_impl == "eager"
        if not torch.cuda.is_available():
            raise L.VispecError("no GPU visible: the ViSpec hot path only exists as HIP kernels (no CPU fallback)")
        self.lib = L.load()
        self.tcfg, self.dcfg, self.tw, self.dw = tcfg, dcfg, tw, dw
        self.device = tw.device
        self.total_token, self.depth, self.top_k, self.num_q = total_token, depth, top_k, num_q
        self.kv_max_pos = kv_max_pos or tcfg.max_position_embeddings
        self.draft_max_pos = draft_max_pos or dcfg.max_position_embeddings
        cfg = L.VispecConfig(
            hidden_size=tcfg.hidden_size, num_heads=tcfg.num_heads, num_kv_heads=tcfg.num_kv_heads, head_dim=tcfg.head_dim,
            intermediate_size=tcfg.intermediate_size, vocab_size=tcfg.vocab_size, num_layers=tcfg.num_layers,
            max_pos=self.kv_max_pos, rms_eps=tcfg.rms_norm_eps, qkv_bias=int(tcfg.qkv_bias),
            draft_heads=dcfg.num_heads, draft_intermediate=dcfg.intermediate_size, draft_max_pos=self.draft_max_pos,
            draft_qkv_bias=int(dcfg.qkv_bias), draft_fc_bias=int(dcfg.bias), draft_rms_eps=dcfg.rms_norm_eps,
            total_token=total_token, depth=depth, top_k=top_k, num_q=num_q, eos_token_id=tcfg.eos_token_id,
            eager_scores=int(eager_scores), draft_rope_rows=max(self.kv_max_pos, self.draft_max_pos),
        )
        self.leader = leader
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            if leader is None:
                L.check(self.lib.vispec_ctx_create(C.byref(cfg), C.byref(h)))
            else:
                L.check(self.lib.vispec_ctx_create_member(C.byref(cfg), leader.h, C.byref(h)))
        self.h = h
        self.t_cos, self.t_sin = rope_tables(tcfg.head_dim, self.kv_max_pos, tcfg.rope_theta, self.device)
        # the draft rotates its rows at their UNCOMPRESSED positions (cnets_ours.py:845-868), which run up to the target's context
        # length: only the draft's KV rows are bounded by draft_max_pos, the tables cover the target cache (the reference regrows its
        # rotary cache on demand, cnets_ours.py:157-162)
        self.d_cos, self.d_sin = rope_tables(dcfg.hidden_size // dcfg.num_heads, max(self.kv_max_pos, self.draft_max_pos), dcfg.rope_theta,
                                             self.device)
        # W32-packed copies of every streamed GEMM weight (the row-major originals stay for the PyTorch prefill)
        GEMM_T = ("wqkv", "wo", "wgu", "wdown")
        self.target_weight_dtype = target_weight_dtype
        # "fp8a8" (round 4): fp8 weights AND fp8 (e4m3, per-row dynamic scale) activations for the target's q|k|v, gate|up and down GEMMs (not o_proj) of the
        # verify / AR forwards (vispec_set_fp8_activations) and of the PyTorch prefill (model/target.py: torch._scaled_mm), multiplied on the fp8 MFMA;
        # "fp8" keeps bf16 activations (W8A16)
        fp8_w = target_weight_dtype in ("fp8", "fp8a8")
        if fp8_w:
            # BASELINE config 5: fp8 (e4m3, per-output-channel scales) target weights.  The row-major copies that the PyTorch prefill
            # uses become the e4m3 CODES held in bf16 (exact) next to their scales (`<name>_scale`): the prefill then computes
            # bf16( (x . codes) * scale + bias ) with an fp32 accumulator — the arithmetic of the W8A16 decode GEMMs — instead of
            # multiplying by weights that were rounded once more when de-quantised to bf16.
            if not hasattr(tw, "packed8"):
                tw.packed8, tw.scales8, tw.codes8 = [], [], []  # codes8: row-major e4m3 codes (kept for inspection / tests)
                for lw in tw.layers:
                    pk, sc, cd = {}, {}, {}
                    for k in GEMM_T:
                        q, s_ = quantize_fp8(lw[k])
                        qp = qkv_rope_order(q, tcfg.num_heads + tcfg.num_kv_heads) if k == "wqkv" else (swiglu_order(q) if k == "wgu" else q)
                        pk[k], sc[k], cd[k] = pack_weight_fp8(qp), s_, q
                        lw[k], lw[k + "_scale"] = q.view(torch.float8_e4m3fn).to(torch.bfloat16), s_
                    tw.packed8.append(pk)
                    tw.scales8.append(sc)
                    tw.codes8.append(cd)
                q, s_ = quantize_fp8(tw.lm_head)
                tw.p_lm_head8, tw.s_lm_head8, tw.c_lm_head8 = pack_weight_fp8(q), s_, q
                # a NEW tensor, never an in-place write: a tied checkpoint hands the same storage out as embed_tokens and lm_head
                # (weights_io.py), and the embedding table must keep its real values
                tw.lm_head = q.view(torch.float8_e4m3fn).to(torch.bfloat16)
                tw.lm_head_scale = s_
        elif target_weight_dtype != "bf16":
            raise ValueError("target_weight_dtype must be 'bf16', 'fp8' or 'fp8a8'")
        if target_weight_dtype == "bf16" and hasattr(tw, "packed8"):
            raise ValueError("these TargetWeights were quantised to fp8 by an earlier Engine (their row-major tensors now hold e4m3 codes + "
                             "scales): build bf16 engines on their own TargetWeights")
        if target_weight_dtype == "bf16" and not hasattr(tw, "packed"):
            nrh = tcfg.num_heads + tcfg.num_kv_heads
            order = lambda k, w: qkv_rope_order(w, nrh) if k == "wqkv" else (swiglu_order(w) if k == "wgu" else w)
            tw.packed = [{k: pack_weight(order(k, lw[k])) for k in GEMM_T} for lw in tw.layers]
            tw.p_lm_head = pack_weight(tw.lm_head)
        GEMM_D = ("fc_w", "imgfc_w", "wqkv", "wo", "wgu", "wdown", "ad_wkv", "ad_wo")
        if not hasattr(dw, "packed"):
            dorder = lambda k, w: qkv_rope_order(w, 2 * dcfg.num_heads) if k == "wqkv" else (swiglu_order(w) if k == "wgu" else w)
            dw.packed = {k: pack_weight(dorder(k, dw.t[k])) for k in GEMM_D}
        fp8 = fp8_w
        for i, lw in enumerate(tw.layers):
            pk = tw.packed8[i] if fp8 else tw.packed[i]
            s = L.LayerWeights(**{k: _p(pk[k] if k in GEMM_T else v) for k, v in lw.items()})
            if fp8:
                s.sqkv, s.so, s.sgu, s.sdown = (_p(tw.scales8[i][k]) for k in GEMM_T)
            L.check(self.lib.vispec_set_target_layer(self.h, i, C.byref(s)))
        m = L.TargetMisc(embed=_p(tw.embed), norm=_p(tw.norm), lm_head=_p(tw.p_lm_head8 if fp8 else tw.p_lm_head),
                         lm_head_scale=_p(tw.s_lm_head8) if fp8 else None, rope_cos=_p(self.t_cos), rope_sin=_p(self.t_sin))
        L.check(self.lib.vispec_set_target_misc(self.h, C.byref(m)))
        d = L.DraftWeights(rope_cos=_p(self.d_cos), rope_sin=_p(self.d_sin),
                           **{k: _p(dw.packed[k] if k in GEMM_D else v) for k, v in dw.t.items()})
        L.check(self.lib.vispec_set_draft_weights(self.h, C.byref(d)))
        # KV buffers: target exactly in the reference layout (kv_cache.py:105-126); draft [2, H, max_pos, hd]
        self.target_kv = torch.zeros(2 * tcfg.num_layers, 1, tcfg.num_kv_heads, self.kv_max_pos, tcfg.head_dim,
                                     dtype=torch.bfloat16, device=self.device)
        self.draft_kv = torch.zeros(2, dcfg.num_heads, self.draft_max_pos, dcfg.hidden_size // dcfg.num_heads,
                                    dtype=torch.bfloat16, device=self.device)
        L.check(self.lib.vispec_set_kv(self.h, _p(self.target_kv), _p(self.draft_kv)))
        if target_weight_dtype == "fp8a8":
            L.check(self.lib.vispec_set_fp8_activations(self.h, 1))

    def close(self):
        """Destroy the library context now (idempotent).  A cohort member gives its activation tile back to its leader — garbage collection
        alone is not prompt enough for that: the model objects around an engine hold reference cycles."""
        if getattr(self, "h", None):
            self.lib.vispec_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- stream plumbing ---------------------------------------------------------------------------
    def _stream(self):
        # HIP's current device is per host thread (a fresh thread starts on device 0) and a launch must be issued with the
        # stream's device current: every library call goes through here, so this is where it is enforced
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- the path ------------------------------------------------------------------------------------
    def check_prompt_fits(self, n_prompt: int):
        """Raise BEFORE any kernel writes a KV row when a prompt of n_prompt tokens (+ one tree) cannot fit the target cache — the reference
        fails cleanly inside KVCache.cat (kv_cache.py:40-58); vispec_begin_request repeats the check on the library side."""
        if n_prompt < 1 or n_prompt + self.total_token + 8 > self.kv_max_pos:
            raise L.VispecError(f"prompt does not fit the KV cache ({n_prompt} tokens + a {self.total_token}-node tree > {self.kv_max_pos} rows)")

    def begin_request(self, prompt_ids, max_new_tokens: int):
        ids = np.ascontiguousarray(np.asarray(prompt_ids, dtype=np.int32))
        self._keep = ids
        L.check(self.lib.vispec_begin_request(self.h, self._stream(), ids.ctypes.data_as(C.c_void_p), int(ids.shape[0]), int(max_new_tokens)))

    def draft_prefill(self, hidden: torch.Tensor, embeds: torch.Tensor, image_mask: Optional[np.ndarray], first_token: torch.Tensor):
        assert hidden.dtype == torch.bfloat16 and embeds.dtype == torch.bfloat16 and hidden.is_contiguous() and embeds.is_contiguous()
        assert first_token.dtype == torch.int32
        Ln = hidden.shape[0]
        m = None
        if image_mask is not None:
            m = np.ascontiguousarray(np.asarray(image_mask, dtype=np.uint8))
            self._keep_mask = m
        L.check(self.lib.vispec_draft_prefill(self.h, self._stream(), _p(hidden), _p(embeds),
                                              None if m is None else m.ctypes.data_as(C.c_void_p), int(Ln), _p(first_token)))

    def verify_accept(self, forced_accept: int = -1):
        L.check(self.lib.vispec_verify_accept(self.h, self._stream(), int(forced_accept)))

    def target_forward(self):
        L.check(self.lib.vispec_target_forward(self.h, self._stream()))

    def accept(self, forced_accept: int = -1):
        L.check(self.lib.vispec_accept(self.h, self._stream(), int(forced_accept)))

    def set_tree(self, tokens: np.ndarray, pos: np.ndarray, mask_bits: np.ndarray, retrieve: Optional[np.ndarray] = None):
        """Install a caller-built tree; retrieve None = the table follows through set_retrieve (SpecModel.forward's verify form)."""
        tokens, pos = np.ascontiguousarray(tokens, np.int32), np.ascontiguousarray(pos, np.int32)
        mask_bits = np.ascontiguousarray(mask_bits, np.uint64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        if retrieve is None:
            L.check(self.lib.vispec_set_tree_host(self.h, self._stream(), vp(tokens), vp(pos), vp(mask_bits), None, 0, 0))
            return
        retrieve = np.ascontiguousarray(retrieve, np.int32)
        L.check(self.lib.vispec_set_tree_host(self.h, self._stream(), vp(tokens), vp(pos), vp(mask_bits), vp(retrieve),
                                              int(retrieve.shape[0]), int(retrieve.shape[1])))

    def set_retrieve(self, retrieve: np.ndarray):
        retrieve = np.ascontiguousarray(retrieve, np.int32)
        L.check(self.lib.vispec_set_retrieve_host(self.h, self._stream(), retrieve.ctypes.data_as(C.c_void_p), int(retrieve.shape[0]),
                                                  int(retrieve.shape[1])))

    def draft_round(self):
        L.check(self.lib.vispec_draft_round(self.h, self._stream()))

    def cohort_round(self, members, forced_accept: int = -1):
        """One draft-and-verify round of 2..8 requests (this engine's and its members', created with leader=self) on one weight pass."""
        members = [members] if isinstance(members, Engine) else list(members)
        hs = [self] + members
        if any(e.h is None for e in hs):
            raise L.VispecError("cohort_round: a context of this cohort was closed")
        arr = (C.c_void_p * len(hs))(*[e.h.value for e in hs])
        st = self._stream()
        L.check(self.lib.vispec_cohortn_verify_accept(arr, len(hs), st, int(forced_accept)))
        L.check(self.lib.vispec_cohortn_draft_round(arr, len(hs), st))

    def set_total_token(self, total_token: int):
        L.check(self.lib.vispec_set_total_token(self.h, int(total_token)))
        self.total_token = int(total_token)

    def set_stop_token(self, token_id: int):
        L.check(self.lib.vispec_set_stop_token(self.h, self._stream(), int(token_id)))

    def set_rope_delta(self, delta: int):
        L.check(self.lib.vispec_set_rope_delta(self.h, self._stream(), int(delta)))

    def set_sampling(self, temperature: float, seed: int = 0, top_k: int = 0):
        L.check(self.lib.vispec_set_sampling(self.h, float(temperature), int(seed) & 0xFFFFFFFFFFFFFFFF))
        if top_k:
            L.check(self.lib.vispec_set_top_k(self.h, int(top_k)))

    def sample_row(self, logits_row: torch.Tensor) -> torch.Tensor:
        row = logits_row.reshape(-1).to(torch.bfloat16).contiguous()
        out = torch.zeros(1, dtype=torch.int32, device=row.device)
        self._keep_row = row
        L.check(self.lib.vispec_sample_row(self.h, self._stream(), _p(row), int(row.shape[0]), _p(out)))
        return out

    def set_next_token(self, token: torch.Tensor):
        assert token.dtype == torch.int32 and token.is_cuda
        self._keep_tok = token
        L.check(self.lib.vispec_set_next_token(self.h, self._stream(), _p(token)))

    def ar_step(self):
        L.check(self.lib.vispec_ar_step(self.h, self._stream()))

    def cohort_ar_step(self, members):
        """One greedy AR token for each request of the cohort (this engine's and its members') on one weight pass."""
        hs = [self] + list(members)
        if any(e.h is None for e in hs):
            raise L.VispecError("cohort_ar_step: a context of this cohort was closed")
        arr = (C.c_void_p * len(hs))(*[e.h.value for e in hs])
        L.check(self.lib.vispec_cohortn_ar_step(arr, len(hs), self._stream()))

    PROF_KINDS = ["gemm_none", "gemm_residual", "gemm_swiglu", "gemm_splitk_partial", "gemm_splitk_reduce", "gemm_qkv_rope", "k6", "gemm_prefill_mfma", "k8",
                  "attn_partial", "attn_reduce", "k11", "attn_partial_sdpa", "attn_reduce_sdpa"]

    def set_graphs(self, on: bool):
        L.check(self.lib.vispec_set_graphs(self.h, int(on)))

    def set_wide_row_blocks(self, row_blocks: int):
        """Launch shape of a 3-4 request cohort's GEMMs (leader only): 4 = default, 0 = best for ONE lane, 8 = eight row blocks per workgroup,
        84 = eight for bf16 weights and for W8A8, four for fp8 weights with bf16 activations (several lanes per GPU: what bench.py sets)."""
        L.check(self.lib.vispec_set_wide_row_blocks(self.h, int(row_blocks)))

    def graph_stats(self) -> Dict[str, int]:
        out = (C.c_longlong * 3)()
        L.check(self.lib.vispec_graph_stats(self.h, out))
        return dict(replays=int(out[0]), captures=int(out[1]), direct=int(out[2]))

    def prof_enable(self, on: bool):
        L.check(self.lib.vispec_prof_enable(self.h, int(on)))

    def prof_report(self) -> Dict[str, Dict[str, float]]:
        n = 16
        out = (C.c_double * (4 * n))()
        L.check(self.lib.vispec_prof_report_host(self.h, self._stream(), out, n))
        rep = {}
        for i, name in enumerate(self.PROF_KINDS):
            if out[4 * i] > 0:
                rep[name] = dict(launches=out[4 * i], ms=out[4 * i + 1], bytes=out[4 * i + 2], workgroups=out[4 * i + 3])
        return rep

    # -- blocking read-backs ---------------------------------------------------------------------------
    def state(self) -> Dict[str, int]:
        out = (C.c_int * 8)()
        L.check(self.lib.vispec_get_state_host(self.h, self._stream(), out))
        k = ("n_ctx", "new_token", "rounds", "done", "accept_len", "next_token", "draft_len", "n_leaf")
        return dict(zip(k, list(out)))

    def cohort_states(self, members):
        """state() of this engine and of its cohort members with ONE stream synchronisation."""
        hs = [self] + list(members)
        if any(x.h is None for x in hs):
            raise L.VispecError("cohort_states: a context of this cohort was closed")
        arr = (C.c_void_p * len(hs))(*[x.h.value for x in hs])
        out = (C.c_int * (8 * len(hs)))()
        L.check(self.lib.vispec_cohort_get_state_host(arr, len(hs), self._stream(), out))
        k = ("n_ctx", "new_token", "rounds", "done", "accept_len", "next_token", "draft_len", "n_leaf")
        return [dict(zip(k, list(out[8 * t:8 * t + 8]))) for t in range(len(hs))]

    def cohort_states_enqueue(self, members, slot: int):
        """First half of cohort_states for a loop with one round of lookahead: snapshot every request's state into pinned slot 0 / 1 in
        stream order (no synchronisation); the next round may be launched right behind it."""
        hs = [self] + list(members)
        if any(x.h is None for x in hs):
            raise L.VispecError("cohort_states_enqueue: a context of this cohort was closed")
        arr = (C.c_void_p * len(hs))(*[x.h.value for x in hs])
        L.check(self.lib.vispec_cohort_state_enqueue(arr, len(hs), self._stream(), int(slot)))

    def cohort_states_wait(self, members, slot: int):
        """Second half: wait for that snapshot only (not for the stream) and return the states as cohort_states does."""
        hs = [self] + list(members)
        arr = (C.c_void_p * len(hs))(*[x.h.value for x in hs])
        out = (C.c_int * (8 * len(hs)))()
        L.check(self.lib.vispec_cohort_state_wait(arr, len(hs), int(slot), out))
        k = ("n_ctx", "new_token", "rounds", "done", "accept_len", "next_token", "draft_len", "n_leaf")
        return [dict(zip(k, list(out[8 * t:8 * t + 8]))) for t in range(len(hs))]

    def last_accept(self):
        """(best_candidate, accept_length) of the last accept."""
        out = (C.c_int * 2)()
        L.check(self.lib.vispec_get_last_accept_host(self.h, self._stream(), out))
        return int(out[0]), int(out[1])

    def tokens(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.int32)
        L.check(self.lib.vispec_get_tokens_host(self.h, self._stream(), out.ctypes.data_as(C.c_void_p), int(n)))
        return out

    def accept_log(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.int32)
        L.check(self.lib.vispec_get_accept_log_host(self.h, self._stream(), out.ctypes.data_as(C.c_void_p), int(n)))
        return out

    def tree(self):
        """-> draft_tokens [T], tree_position_ids [T], tree_mask [T,T] bool, retrieve_indices [n_leaf, max_depth] (host)."""
        T = self.total_token
        tok = np.zeros(L.TREE_MAX_T, np.int32)
        pos = np.zeros(L.TREE_MAX_T, np.int32)
        mask = np.zeros(L.TREE_MAX_T, np.uint64)
        ret = np.zeros((L.TREE_MAX_T, L.TREE_RET_W), np.int32)
        nl, md = C.c_int(), C.c_int()
        L.check(self.lib.vispec_get_tree_host(self.h, self._stream(), tok.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                                              mask.ctypes.data_as(C.c_void_p), ret.ctypes.data_as(C.c_void_p), C.byref(nl), C.byref(md)))
        bits = ((mask[:T, None] >> np.arange(T, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        return tok[:T].astype(np.int64), pos[:T].astype(np.int64), bits, ret[: nl.value, : md.value].astype(np.int64)

    def buffer(self, name: str, shape, dtype=torch.bfloat16) -> torch.Tensor:
        """Zero-copy torch view of an internal device buffer (tests / API mirror)."""
        ptr = self.lib.vispec_buffer(self.h, name.encode())
        if not ptr:
            raise KeyError(name)
        # build a tensor from the raw device pointer through the __cuda_array_interface__ protocol
        typestr = {torch.bfloat16: "<i2", torch.int32: "<i4", torch.float32: "<f4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
        holder = type("DevBuf", (), {"__cuda_array_interface__": {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}})()
        t = torch.as_tensor(holder, device=self.device)
        return t.view(torch.bfloat16) if dtype == torch.bfloat16 else t
