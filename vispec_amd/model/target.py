"""Target VLM wrapper (`base_model` of the reference's SpecModel).

What lives where (BASELINE.json north_star): vision tower / projector and the PREFILL forward stay on PyTorch-ROCm
(plain torch ops below: hipBLASLt GEMMs + SDPA), writing K/V straight into the engine's KV buffer; every forward
that happens inside the draft-and-verify loop (tree verify, AR decode step) runs as HIP kernels via the Engine.
This module owns its Llama decode path instead of subclassing HF (transformers 5.x no longer lets
`language_model` be swapped — SURVEY.md §7.2); reference semantics followed: modeling_llama_kv.py:527-653
(attention), :104-133 (RMSNorm), :927-1080 (model), lm_head + .float() (:1190-1197)."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

import ctypes as C
import os

from .. import lib as L
from ..engine import Engine, TargetConfig, TargetWeights


_gemm_lock = __import__("threading").Lock()
_gemm_state = {"mode": None}
# Recorded GEMM solutions of the prefill shapes (torch's TunableOp CSV: one line per (transposition, M, N, K, leading dims) with the library
# solution that won an offline tuning run, preceded by validator lines: PyTorch / ROCm / hipBLASLt / rocBLAS versions and the GPU arch).
# Written by tools/tune_prefill.py on an MI355X; TunableOp ignores the file when a validator does not match the running stack.
PREFILL_GEMM_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tunable", "prefill_gemms_gfx950.csv")


def prefill_gemm_selection():
    """Which library kernel runs the prefill's GEMMs is decided HERE, once per process and identically on every rank and lane — not by a
    timing taken at the first prompt (rounds 2-3 padded the gate|up weight with zero rows until hipBLASLt's heuristic left a slow
    stream-K kernel: a second copy of every gate|up matrix, chosen by whichever lane got there first, gone with the next ROCm).

      VISPEC_PREFILL_GEMMS=recorded (default)  torch's TunableOp in LOOKUP-ONLY mode over the committed table PREFILL_GEMM_FILE (or
                                               $VISPEC_PREFILL_GEMM_FILE): a shape in the table runs the recorded rocBLAS / hipBLASLt solution,
                                               any other shape — and every shape when the table's validators do not match this PyTorch /
                                               ROCm / GPU — runs the libraries' own default.  Deterministic; nothing is timed at run time.
                                               (TunableOp is a process-wide switch of torch: in this mode every torch GEMM of the process does
                                               one table lookup; shapes outside the table are untouched.  A host application that minds sets
                                               VISPEC_PREFILL_GEMMS=default.)
      VISPEC_PREFILL_GEMMS=default             TunableOp stays off.
      VISPEC_PREFILL_GEMMS=tune                tuning on, results written to the file (tools/tune_prefill.py: offline, one process, idle GPU)."""
    if _gemm_state["mode"] is not None:
        return _gemm_state["mode"]
    with _gemm_lock:
        if _gemm_state["mode"] is not None:
            return _gemm_state["mode"]
        mode = os.environ.get("VISPEC_PREFILL_GEMMS", "recorded")
        path = os.environ.get("VISPEC_PREFILL_GEMM_FILE", PREFILL_GEMM_FILE)
        if mode not in ("recorded", "default", "tune"):
            raise ValueError("VISPEC_PREFILL_GEMMS must be recorded, default or tune")
        if mode != "default":
            try:
                import torch.cuda.tunable as tn
                if mode == "tune":
                    tn.enable(True)
                    tn.tuning_enable(True)
                    tn.set_filename(path)
                elif os.path.exists(path):
                    tn.enable(True)
                    tn.tuning_enable(False)
                    tn.set_filename(path)
                    if not tn.read_file(path):  # validators of another stack: the libraries' defaults, and say so once
                        tn.enable(False)
                        # (no warning at import time of a host application — round-4 advice; the mode string says it, bench.py prints it in
                        #  its line's config and VISPEC_PREFILL_GEMMS_STRICT=1 makes it an error)
                        mode = "default (the recorded table does not match this PyTorch / ROCm / GPU: re-run tools/tune_prefill.py)"
                        if os.environ.get("VISPEC_PREFILL_GEMMS_STRICT") == "1":
                            raise RuntimeError(f"{path}: recorded prefill GEMM solutions do not match this PyTorch / ROCm / GPU (VISPEC_PREFILL_GEMMS_STRICT=1)")
                else:
                    mode = "default (no recorded table)"
            except RuntimeError as e:
                if "VISPEC_PREFILL_GEMMS_STRICT" in str(e):
                    raise
                mode = f"default (TunableOp unavailable: {type(e).__name__})"
            except Exception as e:  # TunableOp is an optimisation: never lose a run to it
                mode = f"default (TunableOp unavailable: {type(e).__name__})"
        _gemm_state["mode"] = mode
        return mode


_sdpa_lock = __import__("threading").Lock()
_sdpa_ready = set()


def _sdpa(q, k, v, gqa):
    """Causal SDPA for the prefill.  PyTorch-ROCm's flash backend initialises its per-device kernel table lazily and not
    thread-safely (concurrent first calls from several lanes were seen to fail with "Accelerated SDPA only supports ..."), so the
    first call per device is serialised; if the accelerated backend is unavailable the math backend is used (still PyTorch)."""
    dev = q.device.index
    if dev not in _sdpa_ready:
        with _sdpa_lock:
            out = _sdpa_try(q, k, v, gqa)
            torch.cuda.synchronize(q.device)
            _sdpa_ready.add(dev)
            return out
    return _sdpa_try(q, k, v, gqa)


def _sdpa_try(q, k, v, gqa):
    try:
        return F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=gqa)
    except RuntimeError as e:
        import warnings
        from torch.nn.attention import SDPBackend, sdpa_kernel
        warnings.warn(f"accelerated SDPA unavailable ({str(e)[:120]}); prefill attention falls back to the math backend")
        with sdpa_kernel([SDPBackend.MATH]):
            return F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=gqa)


def scaled_linear(x, w, b=None, scale=None, q8=None):
    """nn.Linear on a row-major weight.  scale != None: `w` holds fp8 (e4m3) codes as bf16 and `scale` their per-output-channel factors —
    bf16( (x . w^T) * scale + b ) on an fp32 accumulator, the rounding points of the library's W8A16 GEMMs (kernels.h epilogues).
    q8 != None (W8A8, target_weight_dtype "fp8a8"): the same codes as a row-major uint8 tensor; the activations are quantised row by row
    (vispec_quant_rows_e4m3: the decode path's kernel) and the product runs on the fp8 MFMA through the library's fp8 x fp8 GEMM
    (torch._scaled_mm, row-wise scales): bf16( (q_x . q_w^T) * sx[m] * scale[n] + b ) — oracle: Ops.linear(a8=True)."""
    if scale is None:
        return F.linear(x, w, b)
    x2 = x.reshape(-1, x.shape[-1])
    if q8 is not None:
        lib = L.load()
        M, K = x2.shape
        x2 = x2.contiguous()
        qx = torch.empty(M, K, dtype=torch.uint8, device=x2.device)
        sx = torch.empty(M, 1, dtype=torch.float32, device=x2.device)
        L.check(lib.vispec_quant_rows_e4m3(None, C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream), C.c_void_p(x2.data_ptr()), K,
                                           C.c_void_p(qx.data_ptr()), K, C.c_void_p(sx.data_ptr()), M, K))
        out = torch._scaled_mm(qx.view(torch.float8_e4m3fn), q8.view(torch.float8_e4m3fn).t(), scale_a=sx, scale_b=scale.reshape(1, -1).float(),
                               bias=None if b is None else b.to(torch.bfloat16), out_dtype=torch.bfloat16)
        return out.reshape(*x.shape[:-1], q8.shape[0])
    acc = torch.mm(x2, w.t(), out_dtype=torch.float32)
    N = w.shape[0]
    if acc.is_cuda and N % 8 == 0 and x.dtype == torch.bfloat16:  # scale, bias and the bf16 rounding in one pass (vispec_scale_bias_cast)
        lib = L.load()
        out = torch.empty(acc.shape[0], N, dtype=torch.bfloat16, device=acc.device)
        sc = scale.reshape(-1).float().contiguous()
        bb = None if b is None else b.to(torch.bfloat16).contiguous()
        L.check(lib.vispec_scale_bias_cast(None, C.c_void_p(torch.cuda.current_stream(acc.device).cuda_stream), C.c_void_p(acc.data_ptr()), N,
                                           C.c_void_p(sc.data_ptr()), None if bb is None else C.c_void_p(bb.data_ptr()), C.c_void_p(out.data_ptr()), N,
                                           acc.shape[0], N))
        return out.reshape(*x.shape[:-1], N)
    y = acc * scale
    if b is not None:
        y = y + b.float()
    return y.to(x.dtype).reshape(*x.shape[:-1], N)


class _Head:
    """stand-in for nn.Linear lm_head: the reference passes `base_model.lm_head` into topK_genrate (utils.py:300)."""

    def __init__(self, weights):
        self._w = weights  # the Engine may quantise the head in place after this object exists: read weight / scale at call time

    @property
    def weight(self):
        return self._w.lm_head

    def __call__(self, x):
        return scaled_linear(x, self._w.lm_head, None, getattr(self._w, "lm_head_scale", None))


class SyntheticVision:
    """Vision front-end used when no vision tower weights exist on the box (no network): deterministic features
    with the statistics of SURVEY.md §8(d) — N(0,1)*0.05 — one row per image placeholder token."""

    def __init__(self, hidden_size: int, std: float = 0.05):
        self.hidden_size, self.std = hidden_size, std

    def features(self, n_tokens: int, seed: int, device, dtype):
        g = torch.Generator(device="cpu").manual_seed(1000 + int(seed))
        return (torch.randn(n_tokens, self.hidden_size, generator=g) * self.std).to(device=device, dtype=dtype)


class TargetLM:
    """`base_model`: config + weights + PyTorch prefill + handles the reference code touches."""

    def __init__(self, cfg: TargetConfig, weights: TargetWeights, vision=None):
        self.cfg, self.w = cfg, weights
        self.device, self.dtype = weights.device, torch.bfloat16
        self.config = SimpleNamespace(
            architectures=list(cfg.architectures), num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
            num_key_value_heads=cfg.num_kv_heads, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
            vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
            image_token_index=cfg.image_token_index, image_token_id=cfg.image_token_index, video_token_id=cfg.video_token_id,
            eos_token_id=cfg.eos_token_id,
            vision_feature_layer=-2, vision_feature_select_strategy="default")
        self.lm_head = _Head(weights)
        self.vision = vision or SyntheticVision(cfg.hidden_size)
        self.engine: Optional[Engine] = None  # attached by SpecModel
        self.tree_mask = None  # reference stores it on base_model.model (spec_model_ours.py:486-489); kept for API parity
        self.model = self  # `base_model.model.tree_mask = ...` (utils.py:334)

    # ---- handles used by spec_model_ours.py:339-376 -------------------------------------------------
    def get_input_embeddings(self):
        return lambda ids: F.embedding(ids, self.w.embed)

    def get_image_features(self, pixel_values, image_sizes=None, image_grid_thw=None, **kw):
        """-> packed image features [n_image_tokens, D] in prompt order.
        A ready [N, D] tensor is passed through (bench / tests keep the request's features resident in HBM); with a real
        checkpoint (`self.vision` is an HFVisionFrontEnd) `pixel_values` are the processor's pixels and the HF vision tower +
        projector (+ anyres packing) run on PyTorch-ROCm, as in the reference (spec_model_ours.py:341-356); the synthetic
        front-end takes (n_image_tokens, seed)."""
        if (torch.is_tensor(pixel_values) and pixel_values.dim() == 2 and pixel_values.shape[-1] == self.cfg.hidden_size
                and not hasattr(self.vision, "tower")):
            return pixel_values.to(self.device, self.dtype)
        if hasattr(self.vision, "tower"):
            kw = {k: v for k, v in kw.items() if k in ("vision_feature_layer", "vision_feature_select_strategy") and v is not None}
            return self.vision.features(pixel_values, image_sizes=image_sizes, image_grid_thw=image_grid_thw, **kw).to(self.device, self.dtype)
        n, seed = pixel_values
        return self.vision.features(int(n), int(seed), self.device, self.dtype)

    def pack_image_features(self, image_features, image_sizes=None, **kw):
        return image_features, torch.tensor([image_features.shape[0]])

    image_newline = None

    # ---- PyTorch prefill -----------------------------------------------------------------------------
    def mrope_cos_sin(self, pos3: torch.Tensor):
        """cos/sin [L, hd] for 3-component positions [3, L] (modeling_qwen2_5_vl_kv.py rotary + apply_multimodal_rotary_pos_emb):
        fp32 angles per component, chunk i of sizes mrope_section*2 takes component i % 3, then cast to the model dtype."""
        c = self.cfg
        hd = c.head_dim
        inv_freq = 1.0 / (c.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
        freqs = pos3.float().cpu()[:, :, None] * inv_freq[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        cos3, sin3 = emb.cos(), emb.sin()
        sec = list(c.mrope_section) * 2
        cos = torch.cat([m[i % 3] for i, m in enumerate(cos3.split(sec, dim=-1))], dim=-1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(sin3.split(sec, dim=-1))], dim=-1)
        return cos.to(self.dtype).to(self.device), sin.to(self.dtype).to(self.device)

    @torch.no_grad()
    def prefill(self, inputs_embeds: torch.Tensor, all_logits: bool = False, position_ids: Optional[torch.Tensor] = None):
        """inputs_embeds [L, D] bf16 -> (logits fp32 [L or 1, V], hidden [L, D] post-final-norm); K/V rows [0, L)
        of every layer are written into the engine's KV buffer (KVCache.cat semantics, modeling_llama_kv.py:583-594).
        position_ids: None (0..L-1) or [3, L] multimodal rotary positions (Qwen2.5-VL).

        The prefill is MFMA-bound: its GEMMs (hipBLASLt) stay PyTorch ops; its attention is the library's causal kernel
        (vispec_prefill_attention, round 3: 185 us per layer at L = 2704 where torch's flash SDPA takes 296 us); the element-wise
        steps between them — RMSNorm, rotary + KV append, SwiGLU — run as the library's kernels (one launch each instead of ~6 / ~12 / 2 torch ops,
        same rounding points as the decode path): 11 launches per layer instead of ~35, which also keeps the host thread of a lane
        from holding the interpreter while other lanes wait to issue their rounds."""
        c, eng = self.cfg, self.engine
        eng.check_prompt_fits(int(inputs_embeds.shape[0]))  # before the first KV row is written
        lib, st = eng.lib, eng._stream()
        x = inputs_embeds.to(self.dtype).contiguous()
        Ln = x.shape[0]
        H, Hk, hd = c.num_heads, c.num_kv_heads, c.head_dim
        QKV = (H + 2 * Hk) * hd
        if position_ids is not None and position_ids.dim() == 2:
            cos, sin = self.mrope_cos_sin(position_ids)  # per-request tables [L, hd]: row m is the rotary of prompt position m
            cos, sin = cos.contiguous(), sin.contiguous()
        else:
            cos, sin = eng.t_cos, eng.t_sin
        kv = eng.target_kv
        S = kv.shape[3]
        p = lambda t: C.c_void_p(t.data_ptr())
        native_attn = hd == 128 and os.environ.get("VISPEC_PREFILL_SDPA", "0") != "1"  # (A/B switch: torch's SDPA instead)
        prefill_gemm_selection()

        def rmsnorm(t, w):
            out = torch.empty_like(t)
            L.check(lib.vispec_rmsnorm(eng.h, st, p(t), p(w), p(out), Ln, c.hidden_size, c.rms_norm_eps))
            return out

        def add_rmsnorm(t, r, w):  # t <- t + r (in place), returns RMSNorm(t) * w: one launch instead of a torch add + the norm
            out = torch.empty_like(t)
            L.check(lib.vispec_add_rmsnorm(eng.h, st, p(t), p(r), p(w), p(out), Ln, c.hidden_size, c.rms_norm_eps))
            return out

        x = x.clone() if x.data_ptr() == inputs_embeds.data_ptr() else x  # the residual stream is updated in place
        y = None  # the previous layer's down_proj output, added to the residual stream by the next norm's launch
        # W8A8 (fp8a8): the q|k|v, gate|up and down inputs are quantised row by row like the decode forwards' and multiplied on the fp8 MFMA
        a8 = getattr(eng, "target_weight_dtype", "bf16") == "fp8a8" and hasattr(self.w, "codes8")
        for i, lw in enumerate(self.w.layers):
            c8 = self.w.codes8[i] if a8 else {}
            h = rmsnorm(x, lw["ln1"]) if y is None else add_rmsnorm(x, y, lw["ln1"])
            qkv = scaled_linear(h, lw["wqkv"], lw["bqkv"], lw.get("wqkv_scale"), c8.get("wqkv"))  # [L, QKV]
            # rotary at position m (bf16 rounding points of the reference) on q in place; k (rotated) and v -> cache rows [0, L)
            L.check(lib.vispec_rope_append(eng.h, st, p(qkv), Ln, H, Hk, hd, p(cos), p(sin), None, None, p(kv[2 * i]), p(kv[2 * i + 1]), S, None))
            if native_attn:  # causal attention of the L rows over the cache rows just written: the library's kernel (no [H, L, L] tensor,
                # the reference's eager score arithmetic for LLaVA — the decode kernel's — instead of a flash kernel's)
                a = torch.empty(Ln, H * hd, dtype=self.dtype, device=x.device)
                L.check(lib.vispec_prefill_attention(eng.h, st, p(qkv), QKV, p(kv[2 * i]), p(kv[2 * i + 1]), S, H, Hk, Ln, p(a), H * hd,
                                                     int(c.attn_impl == "eager")))
            else:
                q = qkv[:, : H * hd].view(Ln, H, hd).transpose(0, 1)
                a = _sdpa(q[None], kv[2 * i, :, :, :Ln], kv[2 * i + 1, :, :, :Ln], H != Hk)[0].transpose(0, 1).reshape(Ln, H * hd)
            h = add_rmsnorm(x, scaled_linear(a, lw["wo"], None, lw.get("wo_scale")), lw["ln2"])
            gu = scaled_linear(h, lw["wgu"], None, lw.get("wgu_scale"), c8.get("wgu"))  # [L, 2I]
            act = torch.empty(Ln, c.intermediate_size, dtype=self.dtype, device=x.device)
            L.check(lib.vispec_silu_mul(eng.h, st, p(gu), gu.shape[1], p(act), c.intermediate_size, Ln, c.intermediate_size))
            y = scaled_linear(act, lw["wdown"], None, lw.get("wdown_scale"), c8.get("wdown"))
        hidden = rmsnorm(x, self.w.norm) if y is None else add_rmsnorm(x, y, self.w.norm)
        logits = scaled_linear(hidden if all_logits else hidden[-1:], self.w.lm_head, None, getattr(self.w, "lm_head_scale", None)).float()
        return logits, hidden.contiguous()

    def eval(self):
        return self
