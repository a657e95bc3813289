"""SpecModel — the reference's spec-decoding API (vispec/model/spec_model_ours.py): from_pretrained (:109-203),
forward (:205-245), specgenerate (:247-582), get_tokenizer (:101-107), and the attributes the harness reads
(base_model, spec_layer, past_key_values, past_key_values_data, current_length_data, tokenizer).

MI355X design: one SpecModel = one process = one GPU = one `vispec_ctx`.  specgenerate() runs the target prefill in
PyTorch, then each draft-and-verify round is three stream-ordered C calls (verify+accept, draft round, state read-back)
with exactly one host sync per round (the reference has >= 10)."""
from __future__ import annotations

import json
import os
import time
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .. import lib as L
from ..engine import DraftConfig, DraftWeightsDev, Engine, TargetConfig, TargetWeights
from .cnets_ours import Model
from .kv_cache import initialize_past_key_values
from .target import TargetLM
from .utils import initialize_tree, reset_tree_mode


class SpecModel:
    def __init__(self, base_model: TargetLM, spec_layer: Model, tokenizer=None, total_token=30, depth=3, top_k=8, num_q=2,
                 kv_max_pos: Optional[int] = None, draft_max_pos: Optional[int] = None, target_weight_dtype: str = "bf16",
                 cohort_leader: Optional["SpecModel"] = None):
        """cohort_leader: build this model as the second member of a two-request cohort (same weights objects as the leader): see
        specgenerate_cohort()."""
        self.base_model = base_model
        self.config = base_model.config
        self.hidden_size = base_model.cfg.hidden_size
        self.vocab_size = base_model.cfg.vocab_size
        self.tokenizer = tokenizer or SimpleNamespace(eos_token_id=base_model.cfg.eos_token_id, vocab_size=base_model.cfg.vocab_size)
        self.spec_layer = spec_layer
        self.engine = Engine(base_model.cfg, spec_layer.config, base_model.w, spec_layer.w, total_token=total_token, depth=depth,
                             top_k=top_k, num_q=num_q, kv_max_pos=kv_max_pos, draft_max_pos=draft_max_pos,
                             target_weight_dtype=target_weight_dtype, leader=None if cohort_leader is None else cohort_leader.engine)
        base_model.engine = self.engine
        spec_layer.engine = self.engine
        spec_layer.init_tree()
        self.training = False
        self._last_embeds = None

    def eval(self):
        return self

    def make_cohort_member(self) -> "SpecModel":
        """A second model over the SAME weight tensors (nothing is copied or re-packed) whose engine is a cohort member of this one's:
        own KV caches, tree and round state; specgenerate_cohort([self, member], [req_a, req_b]) then runs two requests per weight pass."""
        e = self.engine
        base = TargetLM(self.base_model.cfg, self.base_model.w, vision=self.base_model.vision)
        draft = Model(self.spec_layer.config, self.spec_layer.w, total_tokens=e.total_token, depth=e.depth, top_k=e.top_k, num_q=e.num_q)
        return SpecModel(base, draft, tokenizer=self.tokenizer, total_token=e.total_token, depth=e.depth, top_k=e.top_k, num_q=e.num_q,
                         kv_max_pos=e.kv_max_pos, draft_max_pos=e.draft_max_pos, target_weight_dtype=e.target_weight_dtype, cohort_leader=self)

    def get_tokenizer(self):
        return self.tokenizer

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_weights(cls, tcfg: TargetConfig, dcfg: DraftConfig, target_sd, draft_sd, device="cuda:0", **kw):
        """Build from in-memory state dicts (numpy / torch) with the reference's names."""
        device = torch.device(device)
        num_q = kw.get("num_q", 2)
        tw = TargetWeights.from_state_dict(tcfg, target_sd, device)
        dw = DraftWeightsDev.from_state_dict(dcfg, draft_sd, num_q, device)
        base = TargetLM(tcfg, tw)
        draft = Model(dcfg, dw, total_tokens=kw.get("total_token", 30), depth=kw.get("depth", 3), top_k=kw.get("top_k", 8),
                      num_q=num_q)
        return cls(base, draft, **kw)

    @classmethod
    def from_pretrained(cls, Type="LLaMA", base_model_path=None, spec_model_path=None, total_token=30, depth=3, top_k=8,
                        threshold=1.0, num_q=2, device="cuda:0", **kwargs):
        """spec_model_ours.py:109-203.  Local directories only (there is no network on the GPU box):
        base_model_path = HF checkpoint dir of the target (config.json + *.safetensors [+ index]),
        spec_model_path = ViSpec draft dir (config.json + model.safetensors | pytorch_model.bin)."""
        from ..weights_io import load_draft_dir, load_target_dir  # imported lazily: safetensors is optional at import time
        tcfg, target_sd, tokenizer = load_target_dir(base_model_path)
        dcfg, draft_sd = load_draft_dir(spec_model_path, tcfg)
        model = cls.from_weights(tcfg, dcfg, target_sd, draft_sd, device=device, total_token=60 if total_token == -1 else total_token,
                                 depth=depth, top_k=top_k, num_q=num_q, tokenizer=tokenizer,
                                 target_weight_dtype=kwargs.get("target_weight_dtype", "bf16"))  # "fp8" / "fp8a8": quantised at load (engine.py)
        if tcfg.architectures[0] not in ("LlamaForCausalLM", "Qwen2ForCausalLM"):
            # vision tower + projector of the checkpoint (PyTorch-ROCm, HF modules): SURVEY §8 A2
            from .vision import HFVisionFrontEnd
            try:
                model.base_model.vision = HFVisionFrontEnd.from_dir(base_model_path, device, torch.bfloat16)
            except (FileNotFoundError, ImportError) as e:
                import warnings
                warnings.warn(f"{base_model_path}: no usable vision weights ({e}); image prompts need precomputed features")
        if total_token == -1:
            model.autotune_total_token()
        return model

    def autotune_total_token(self, cans=(40, 48, 50, 56, 60), x=(1, 1.05, 1.07, 1.1, 1.13), iters=20):
        """spec_model_ours.py:179-201: time the target forward on `length` tokens for each candidate tree size, weight the times by
        x, keep the cheapest.  Here the timed forward is the one the loop actually runs (the HIP verify pass on a `length`-node chain)."""
        eng, times = self.engine, []
        g = torch.Generator().manual_seed(0)
        for length in cans:
            self.spec_layer.total_tokens = length - 1
            ids = torch.randint(0, max(2, self.vocab_size - 200), (length,), generator=g).numpy().astype(np.int32)
            eng.begin_request(ids[:1], 8)
            causal = np.array([(1 << (i + 1)) - 1 for i in range(length)], np.uint64)
            eng.set_tree(ids, np.arange(length, dtype=np.int32), causal, np.zeros((1, 1), np.int32))
            eng.target_forward()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(iters):
                eng.target_forward()
            torch.cuda.synchronize()
            times.append((time.time() - t0) / x[len(times)])
        best = cans[times.index(min(times))]
        self.spec_layer.total_tokens = best - 1
        return best

    # ------------------------------------------------------------------------------------------------
    def _first_token(self, orig: torch.Tensor) -> torch.Tensor:
        """argmax(orig[:, -1]) (utils.py:290) via the HIP argmax kernel on the bf16-rounded logits (first max wins)."""
        import ctypes as C
        row = orig.reshape(-1, orig.shape[-1])[-1:].to(torch.bfloat16).contiguous()
        out = torch.zeros(1, dtype=torch.int32, device=row.device)
        L.check(self.engine.lib.vispec_argmax_rows(self.engine.h, self.engine._stream(), C.c_void_p(row.data_ptr()), row.shape[-1], 1,
                                                   row.shape[-1], C.c_void_p(out.data_ptr())))
        return out

    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, output_orig=False, position_ids=None,
                inputs_embeds=None, output_real_hidden=False, **kwargs):
        """spec_model_ours.py:205-245: ONE function serving both callers of the reference —
          * utils.initialize_tree (utils.py:280-283): empty KV cache -> the PREFILL form (PyTorch-ROCm GEMMs + SDPA, K/V written in place);
          * utils.tree_decoding (utils.py:404-409): non-empty cache, `input_ids` = the T tree candidates, `position_ids` =
            tree_position_ids + n (x3, + rope_deltas for Qwen2.5-VL), tree mask installed on `base_model.model.tree_mask`
            (spec_model_ours.py:486-489) -> the VERIFY form (vispec_target_forward: HIP kernels, K/V rows appended at [n, n+T)).
        Returns (None[, logits fp32 [1,S,V]], hidden [1,S,D]) (post-final-norm hidden, modeling_llama_kv.py:1062-1066)."""
        n_past = 0
        if past_key_values is not None:
            n_past = int(past_key_values[0][0].current_length.item())  # `past_key_values[0][0].shape[2]` of modeling_llama_kv.py:957
        if n_past > 0:
            return self._forward_verify(input_ids, past_key_values, output_orig, position_ids, inputs_embeds)
        if inputs_embeds is None:
            inputs_embeds = self.base_model.get_input_embeddings()(input_ids.to(self.engine.device))
        emb = inputs_embeds.reshape(-1, inputs_embeds.shape[-1]).to(torch.bfloat16).contiguous()
        self._last_embeds = None if input_ids is not None and kwargs.get("_draft_embeds_from_ids") else emb[None]
        pos3 = position_ids if (position_ids is not None and position_ids.dim() == 3) else None
        logits, hidden = self.base_model.prefill(emb, all_logits=bool(kwargs.get("all_logits", False)),
                                                 position_ids=None if pos3 is None else pos3[:, 0].cpu())
        # device-side round state: context = the L prefilled rows (utils.initialize_tree re-issues this with the prompt ids)
        ids0 = (input_ids.reshape(-1).cpu().numpy() if input_ids is not None and input_ids.numel() == emb.shape[0]
                else np.zeros(emb.shape[0], np.int32))
        self.engine.begin_request(ids0, 1 << 20)
        if past_key_values is not None:
            for kv in past_key_values:
                kv[0].current_length.fill_(emb.shape[0])
                kv[1].current_length.fill_(emb.shape[0])
        if output_orig:
            return None, logits[None], hidden[None]
        return None, hidden[None]

    def _forward_verify(self, input_ids, past_key_values, output_orig, position_ids, inputs_embeds):
        """The verify form of forward(): S <= 64 new tokens against the committed context of the engine."""
        eng = self.engine
        if input_ids is None or inputs_embeds is not None:
            raise NotImplementedError("forward() with a non-empty KV cache takes token ids (the tree candidates); the reference's "
                                      "loop never passes inputs_embeds there (utils.py:404-409)")
        ids = input_ids.reshape(-1).cpu().numpy().astype(np.int32)
        S = int(ids.shape[0])
        if S < 1 or S > L.TREE_MAX_T:
            raise ValueError(f"forward() with a non-empty KV cache handles 1..{L.TREE_MAX_T} tokens per call (one tree), got {S}")
        st = eng.state()
        n = int(st["n_ctx"])  # authoritative context length (the host mirror in past_key_values is refreshed below)
        # relative positions = depth in the tree (utils.py:397: position_ids = tree_position_ids + input_ids.shape[1])
        if position_ids is None:
            pos_rel = np.arange(S, dtype=np.int64)
        else:
            p = position_ids
            if p.dim() == 3:  # Qwen2.5-VL: [3, 1, T], the three components are equal in the decode phase (utils.py:398-402)
                p = p[0]
            pos_rel = p.reshape(-1).cpu().numpy().astype(np.int64) - n - int(getattr(self, "_rope_delta", 0))
        if pos_rel.shape[0] != S or pos_rel.min() < 0 or pos_rel.max() >= S:
            raise ValueError("position_ids of a verify forward must be tree depths offset by the context length "
                             f"(n = {n}{', + rope_deltas' if getattr(self, '_rope_delta', 0) else ''})")
        tm = self.base_model.tree_mask
        bits = np.zeros(L.TREE_MAX_T, np.uint64)
        if tm is None:  # no tree mask installed: plain causal continuation (modeling_llama_kv.py:890-924 without the tree branch)
            for i in range(S):
                bits[i] = np.uint64((1 << (i + 1)) - 1)
        else:
            m = (tm.reshape(tm.shape[-2], tm.shape[-1]).cpu().numpy() > 0)
            if m.shape != (S, S):
                raise ValueError(f"tree_mask {m.shape} does not match the {S} tree tokens")
            for i in range(S):
                bits[i] = np.uint64(sum(1 << j for j in range(S) if m[i, j]))
        restore = None
        if S != eng.total_token:
            restore = eng.total_token
            eng.set_total_token(S)
        try:
            eng.set_tree(ids, pos_rel.astype(np.int32), bits, None)
            eng.target_forward()
            V, D = eng.tcfg.vocab_size, eng.tcfg.hidden_size
            logits = eng.buffer("logits", (64, V))[:S].float()[None]  # `.float()` of modeling_llama_kv.py:1197
            hidden = eng.buffer("hidden_new", (64, D))[:S].clone()[None]
        finally:
            if restore is not None:
                eng.set_total_token(restore)
        for kv in past_key_values:  # KVCache.cat appended S rows (kv_cache.py:40-58); update_inference_inputs resets the lengths
            kv[0].current_length.fill_(n + S)
            kv[1].current_length.fill_(n + S)
        if output_orig:
            return None, logits, hidden
        return None, hidden

    __call__ = forward

    def _merge_vision(self, input_ids, inputs_embeds, kwargs):
        """spec_model_ours.py:309-453: token embeddings, image features scattered over the placeholder tokens, and the
        image mask the draft compresses with.
        -> (inputs_embeds | None, special_image_mask | None, draft_embeds | None, position_ids [3,L] | None, rope_delta)."""
        special_image_mask = None
        arch = self.base_model.config.architectures[0]
        draft_embeds = None
        position_ids, rope_delta = None, 0
        if arch == "LlavaNextForConditionalGeneration":  # :311-378
            pixel_values = kwargs.get("pixel_values")
            image_sizes = kwargs.get("image_sizes")
            if pixel_values is not None and inputs_embeds is not None:
                raise ValueError("You cannot specify both pixel_values and inputs_embeds at the same time, and must specify either one")
            if inputs_embeds is None:
                inputs_embeds = self.base_model.get_input_embeddings()(input_ids)
            if pixel_values is not None:
                image_features = self.base_model.get_image_features(pixel_values, image_sizes)
                image_features, _ = self.base_model.pack_image_features(image_features, image_sizes)
                mask = input_ids == self.base_model.config.image_token_index
                n_tok = int(mask.sum())
                if n_tok != image_features.shape[0]:  # :363-370
                    raise ValueError(f"Image features and image tokens do not match: tokens: {n_tok}, features {image_features.shape[0]}")
                inputs_embeds = inputs_embeds.clone()
                inputs_embeds[mask] = image_features.to(inputs_embeds.dtype)
                special_image_mask = mask
            draft_embeds = inputs_embeds
        elif arch == "LlavaForConditionalGeneration":
            # LLaVA-1.5: the reference's specgenerate has NO vision branch for it: the target merges the
            # image features itself (HF forward with pixel_values), while the draft gets inputs_embeds=None / image_mask=None, i.e.
            # it embeds the ids (placeholder tokens included) with its own table and never compresses (g = 0).
            pixel_values = kwargs.get("pixel_values")
            if inputs_embeds is None:
                inputs_embeds = self.base_model.get_input_embeddings()(input_ids)
                if pixel_values is not None:
                    feats = self.base_model.get_image_features(pixel_values, kwargs.get("image_sizes"))
                    mask = input_ids == self.base_model.config.image_token_index
                    if int(mask.sum()) != feats.shape[0]:
                        raise ValueError(f"Image features and image tokens do not match: tokens: {int(mask.sum())}, features {feats.shape[0]}")
                    inputs_embeds = inputs_embeds.clone()
                    inputs_embeds[mask] = feats.to(inputs_embeds.dtype)
        elif arch == "Qwen2_5_VLForConditionalGeneration":  # :380-453
            pixel_values, grid = kwargs.get("pixel_values"), kwargs.get("image_grid_thw")
            pixel_values_videos, vgrid = kwargs.get("pixel_values_videos"), kwargs.get("video_grid_thw")
            if inputs_embeds is None:
                inputs_embeds = self.base_model.get_input_embeddings()(input_ids)
                for pv, g_, tid, what in ((pixel_values, grid, self.base_model.config.image_token_id, "Image"),
                                          (pixel_values_videos, vgrid, self.base_model.cfg.video_token_id, "Video")):
                    if pv is None:
                        continue
                    feats = self.base_model.get_image_features(pv, image_grid_thw=g_)  # the same visual tower serves both (:393-395, 426-428)
                    mask = input_ids == tid
                    if int(mask.sum()) != feats.shape[0]:  # :403-406 / :435-438
                        raise ValueError(f"{what} features and {what.lower()} tokens do not match: tokens: {int(mask.sum())}, features {feats.shape[0]}")
                    inputs_embeds = inputs_embeds.clone()
                    inputs_embeds[mask] = feats.to(inputs_embeds.dtype)
                    special_image_mask = mask  # the reference keeps the LAST mask it built (:420,453): with a video, the draft compresses
                    #                            the video runs and sees image tokens as ordinary rows
            draft_embeds = inputs_embeds
            # multimodal rotary positions of the prefill + rope_delta of the decode rounds: get_rope_index over image AND video grids,
            # video frames at floor(frame * second_per_grid_ts * tokens_per_second) (modeling_qwen2_5_vl_kv.py:1789-1975, 2157-2163)
            position_ids, rope_delta = self._qwen_rope(input_ids, grid, video_grid=vgrid, second_per_grid_ts=kwargs.get("second_per_grid_ts"))
        elif arch in ("LlamaForCausalLM", "Qwen2ForCausalLM"):
            pass  # text targets (Qwen2 = the same decoder with q/k/v bias, modeling_qwen2_kv.py): the draft embeds the ids itself (cnets_ours.py:1099-1107)
        else:
            raise NotImplementedError(f"target architecture {arch}")
        return inputs_embeds, special_image_mask, draft_embeds, position_ids, rope_delta

    def _qwen_rope(self, input_ids, grid, video_grid=None, second_per_grid_ts=None):
        """Multimodal rotary positions [3, L] + rope_delta of a Qwen2.5-VL prompt (get_rope_index: image and video grids), cached on the
        target like the reference's prefill does (modeling_qwen2_5_vl_kv.py `self.rope_deltas`; read back by utils.tree_decoding, utils.py:398-400)."""
        from ..synth import qwen_rope_index
        as_grids = lambda g: [] if g is None else [tuple(int(v) for v in g3) for g3 in (g.tolist() if torch.is_tensor(g) else g)]
        sec = None
        if second_per_grid_ts is not None:
            sec = [float(v) for v in (second_per_grid_ts.tolist() if torch.is_tensor(second_per_grid_ts) else second_per_grid_ts)]
        cfg = self.base_model.cfg
        pos3, rope_delta = qwen_rope_index(input_ids[0].cpu().numpy(), self.base_model.config.image_token_id, as_grids(grid),
                                           video_token_id=cfg.video_token_id if cfg.video_token_id >= 0 else None,
                                           video_grids=as_grids(video_grid), second_per_grid_ts=sec, tokens_per_second=cfg.tokens_per_second)
        self.base_model.rope_deltas = torch.tensor([[rope_delta]], device=input_ids.device)
        return torch.from_numpy(pos3), int(rope_delta)

    @torch.no_grad()
    def _start_request(self, input_ids, inputs_embeds, kwargs, temperature=0.0, top_k=0.0, seed=0, max_new_tokens=512, is_llama3=False):
        """Everything specgenerate does before its loop (spec_model_ours.py:272-475): sampling config, KV reset, vision merge, target
        prefill, first token, device round state, draft prefill with image-token compression + the first tree.
        Returns (hidden [L,D], draft_embeds [L,D], image_mask numpy | None, first_token int32 [1]) — what the draft prefill consumed."""
        eng = self.engine
        eng.set_sampling(temperature if temperature > 1e-5 else 0.0, seed, top_k=int(top_k) if temperature > 1e-5 else 0)
        dev = eng.device
        input_ids = input_ids.clone().to(dev)
        self.spec_layer.reset_kv()  # :283
        if not hasattr(self, "past_key_values"):  # :286-307
            self.past_key_values, self.past_key_values_data, self.current_length_data = initialize_past_key_values(self.base_model)
        self.current_length_data.zero_()
        inputs_embeds, special_image_mask, draft_embeds, position_ids, rope_delta = self._merge_vision(input_ids, inputs_embeds, kwargs)
        reset_tree_mode(self)  # :456
        # initialize_tree (:458-475): prefill + first token + draft prefill with image-token compression
        emb = (inputs_embeds if inputs_embeds is not None else self.base_model.get_input_embeddings()(input_ids))
        emb = emb.reshape(-1, emb.shape[-1]).to(torch.bfloat16).contiguous()
        logits, hidden = self.base_model.prefill(emb, position_ids=position_ids)  # raises before touching the KV when the prompt cannot fit
        first = eng.sample_row(logits[-1]) if temperature > 1e-5 else self._first_token(logits)
        eng.begin_request(input_ids[0].cpu().numpy(), max_new_tokens)
        self._rope_delta = int(rope_delta)
        if rope_delta:
            eng.set_rope_delta(rope_delta)
        if is_llama3:  # :268-269, 540-542
            eng.set_stop_token(int(self.tokenizer.convert_tokens_to_ids("<|eot_id|>")))
        if draft_embeds is None:
            ids1 = torch.cat([input_ids[0], first.long()])
            demb = torch.nn.functional.embedding(ids1[:-1], self.spec_layer.w.t["embed"]).contiguous()
        else:
            demb = emb
        mask_np = None if special_image_mask is None else special_image_mask.reshape(-1).cpu().numpy()
        eng.draft_prefill(hidden, demb, mask_np, first)
        self.spec_layer.stable_kv = ("device", eng)
        return hidden, demb, mask_np, first

    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def specgenerate(self, input_ids, temperature=0.0, top_p=0.0, top_k=0.0, max_new_tokens=512, max_length=2048, log=False,
                     is_llama3=False, inputs_embeds=None, return_acceptance_len=False, return_decode_time=False,
                     forced_accept=None, seed=0, **kwargs):
        """spec_model_ours.py:247-582.  `forced_accept` (callable round->int, bench-only) scripts the accept length.
        temperature > 1e-5 selects the sampling path (utils.py:453-493) with device-side counter-based randomness (`seed`) and the
        processor list of utils.py:39-55: temperature, then TopK when top_k > 0.  0 < top_p < 1 raises: HF's TopPLogitsWarper fails
        on the 3-D tree logits evaluate_posterior passes it, so the reference cannot run that setting either."""
        if (input_ids is None) ^ (inputs_embeds is not None):  # :263-266 (sic: exactly the reference's condition)
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if temperature > 1e-5 and 1e-8 <= top_p < 1.0:
            raise NotImplementedError("TopPLogitsWarper (utils.py:50-51) cannot run on the tree logits (HF scatters along dim 1: "
                                      "RuntimeError in the reference as well); use temperature / top_k")
        eng = self.engine
        max_length = max_length - self.spec_layer.total_tokens - 10  # :270
        self._start_request(input_ids, inputs_embeds, kwargs, temperature=temperature, top_k=top_k, seed=seed, max_new_tokens=max_new_tokens,
                            is_llama3=is_llama3)
        dev = eng.device
        acceptance_len = []
        if return_decode_time:
            torch.cuda.synchronize()
            start_time = time.time()
        idx = 0
        st = eng.state()  # max_length <= 0 runs no round: the reference then returns the prompt as it is (:484,555)
        for idx in range(max_length):  # :484
            fa = -1 if forced_accept is None else int(forced_accept(idx))
            eng.verify_accept(fa)   # tree_decoding + evaluate_posterior + accept half of update_inference_inputs
            eng.draft_round()       # topK_genrate half of update_inference_inputs
            st = eng.state()        # the round's single host sync
            if return_acceptance_len:
                acceptance_len.append(int(st["accept_len"]))
            if st["done"] & 1:      # :544 eos in generated ids
                break
            if st["new_token"] > max_new_tokens:  # :546
                break
            if st["done"] & 4:  # the next tree would not fit a KV cache: the reference raises in KVCache.cat here (kv_cache.py:40-58)
                import warnings
                warnings.warn(f"specgenerate stopped after {st['new_token']} new tokens: the next round would not fit a KV cache "
                              f"(context {st['n_ctx']} of {eng.kv_max_pos} rows); the returned sequence is truncated", RuntimeWarning)
                break
        n_ctx, new_token = st["n_ctx"], st["new_token"]
        self.current_length_data.fill_(n_ctx)
        toks = torch.from_numpy(eng.tokens(n_ctx).astype(np.int64)).to(dev)[None]
        outputs = (toks,)
        if log:
            outputs += (new_token, idx)
        if return_acceptance_len:
            outputs += (acceptance_len,)
        if return_decode_time:
            torch.cuda.synchronize()
            outputs += (time.time() - start_time,)
        return outputs[0] if len(outputs) == 1 else outputs

    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _start_baseline(self, input_ids, inputs_embeds, kwargs, max_new_tokens):
        """Prefill + first token of an AR request (gen_baseline_answer_coco_caption.py:60-110); -> the prompt ids on the device."""
        eng = self.engine
        eng.set_sampling(0.0, 0)
        input_ids = input_ids.clone().to(eng.device)
        inputs_embeds, _, _, position_ids, rope_delta = self._merge_vision(input_ids, inputs_embeds, kwargs)
        if inputs_embeds is None:
            inputs_embeds = self.base_model.get_input_embeddings()(input_ids)
        emb = inputs_embeds.reshape(-1, inputs_embeds.shape[-1]).to(torch.bfloat16).contiguous()
        logits, _ = self.base_model.prefill(emb, position_ids=position_ids)
        first = self._first_token(logits)
        eng.begin_request(input_ids[0].cpu().numpy(), max_new_tokens)
        if rope_delta:
            eng.set_rope_delta(rope_delta)
        eng.set_next_token(first)
        return input_ids

    def _finish_baseline(self, input_ids, n_ctx, max_new_tokens):
        toks = self.engine.tokens(n_ctx).astype(np.int64)
        eos = self.base_model.cfg.eos_token_id
        gen = toks[input_ids.shape[1]:]
        cut = np.nonzero(gen == eos)[0]
        if cut.size:
            toks = toks[: input_ids.shape[1] + int(cut[0]) + 1]
        elif len(gen) > max_new_tokens + 1:
            toks = toks[: input_ids.shape[1] + max_new_tokens + 1]
        return torch.from_numpy(toks).to(self.engine.device)[None]

    @torch.no_grad()
    def baseline_generate(self, input_ids, inputs_embeds=None, max_new_tokens=512, max_steps=2048, **kwargs):
        """Greedy AR with the same KV cache and kernels — evaluation/gen_baseline_answer_coco_caption.py:34-133."""
        eng = self.engine
        input_ids = self._start_baseline(input_ids, inputs_embeds, kwargs, max_new_tokens)
        sync_every = 16
        for n in range(1, max_steps + 1):
            eng.ar_step()
            if n % sync_every == 0 or n == max_steps:
                st = eng.state()
                if st["done"]:
                    break
        st = eng.state()
        return self._finish_baseline(input_ids, st["n_ctx"], max_new_tokens)


def _sync_tree_size(models):
    """Every request of a cohort round needs the leader's tree size (the C library checks it): a member built before
    `autotune_total_token()` / `spec_layer.total_tokens = ...` changed the leader's follows it here.  Trees of 33..64 nodes take two
    activation tiles per request, so such a cohort has at most four requests (vispec_ctx_create_member / vispec_set_total_token)."""
    lead = models[0]
    T = lead.engine.total_token
    if T > 32 and len(models) > 4:
        raise ValueError(f"a cohort of {len(models)} requests with trees of {T} nodes: trees of more than 32 nodes take two activation tiles per "
                         "request, at most four requests share a weight pass")
    for m in models[1:]:
        if m.engine.total_token != T:
            m.spec_layer.total_tokens = T - 1


@torch.no_grad()
def baseline_generate_cohort(models, requests, max_new_tokens=512, max_steps=2048, stats=None):
    """The AR baseline (gen_baseline_answer_coco_caption.py:34-133) for two to eight requests in LOCKSTEP on one weight pass: what
    specgenerate_cohort is to specgenerate.  models / requests as there; max_new_tokens may be a list.  Returns one [1, L + new] id tensor
    per request — for cohorts of up to four the tokens `m.baseline_generate(ids, ...)` returns for that request alone (a request that
    reaches EOS or its budget freezes on the device while the others go on).  The speed-up bench.py prints divides a cohort's
    speculative tokens/s by THIS rate: like by like (speed.py:56-97).
    Arithmetic class (round 5): cohorts of up to FOUR requests keep every row bit-identical to a run of that request alone; cohorts of
    FIVE to EIGHT run the cohort-8 GEMM (csrc/gemm_c8.h: one fp32 accumulator chain per element) and 768-key attention splits — a request's
    tokens then do not depend on WHAT shares its weight pass (any cohort of 5..8, any tile), but they are the tokens of the c8 summation
    order, which may leave the solo run's at a near tie of the logits.  The class follows len(models) of the call: a deployment that wants
    one class for every request always calls with its full slot count."""
    n = len(models)
    if not 2 <= n <= 8 or len(requests) != n:
        raise ValueError("a cohort is 2..8 models (leader, members...) and one request per model")
    lead = models[0]
    for m in models[1:]:
        if m.engine.leader is not lead.engine:
            raise ValueError("models[1:] must have been built with cohort_leader=models[0]")
    budgets = list(max_new_tokens) if isinstance(max_new_tokens, (list, tuple)) else [max_new_tokens] * n
    _sync_tree_size(models)
    prompts = [m._start_baseline(ids, None, dict(kw), mx) for m, (ids, kw), mx in zip(models, requests, budgets)]
    members = [m.engine for m in models[1:]]
    if stats is not None:
        torch.cuda.current_stream().synchronize()
        t_loop = time.time()
    sync_every, steps = 16, 0
    for steps in range(1, max_steps + 1):
        lead.engine.cohort_ar_step(members)
        if steps % sync_every == 0 or steps == max_steps:
            states = lead.engine.cohort_states(members)
            if all(st["done"] for st in states):
                break
    states = lead.engine.cohort_states(members)
    if stats is not None:
        stats.update(decode_s=time.time() - t_loop, steps=steps)
    return [m._finish_baseline(p, st["n_ctx"], mx) for m, p, st, mx in zip(models, prompts, states, budgets)]


@torch.no_grad()
def specgenerate_cohort(models, requests, temperature=0.0, top_k=0.0, max_new_tokens=512, max_length=2048, is_llama3=False, seeds=None,
                        forced_accept=None, stats=None):
    """Two to eight independent requests through SpecModel.specgenerate's loop (spec_model_ours.py:247-582) in LOCKSTEP on one weight pass.

    models   = [leader, member, ...]  (members built with cohort_leader=leader: one vispec_ctx, KV cache, tree and round state each)
    requests = [(input_ids [1,L], specgenerate kwargs), ...] one per model; max_new_tokens may be a list (one budget per request)
    Returns one (input_ids [1, L+new], new_token, idx, acceptance_len) tuple per request — for cohorts of up to four exactly what
    `m.specgenerate(ids, log=True, return_acceptance_len=True, ...)` returns for that request alone, token for token: the prefills run
    per request, every decode round launches each GEMM once on all requests' rows (Engine.cohort_round), and a request that finishes
    first is frozen on the device while the others complete.
    Arithmetic class (round 5): cohorts of up to FOUR requests keep every row bit-identical to a run of that request alone; cohorts of
    FIVE to EIGHT run the cohort-8 GEMM (csrc/gemm_c8.h: one fp32 accumulator chain per element) and 768-key attention splits — a request's
    tokens then do not depend on WHAT shares its weight pass (any cohort of 5..8, any tile), but they are the tokens of the c8 summation
    order, which may leave the solo run's at a near tie of the logits.  The class follows len(models) of the call: a deployment that wants
    one class for every request always calls with its full slot count.  `stats` (a dict, optional) receives the wall time of the round loop
    (`decode_s`, bracketed by synchronisations like specgenerate's return_decode_time) and the number of lockstep rounds (`rounds`)."""
    n = len(models)
    if not 2 <= n <= 8 or len(requests) != n:
        raise ValueError("a cohort is 2..8 models (leader, members...) and one request per model")
    lead = models[0]
    for m in models[1:]:
        if m.engine.leader is not lead.engine:
            raise ValueError("models[1:] must have been built with cohort_leader=models[0]")
    seeds = seeds or [0] * n
    _sync_tree_size(models)
    budgets = list(max_new_tokens) if isinstance(max_new_tokens, (list, tuple)) else [max_new_tokens] * n  # per request
    for m, (ids, kw), sd, mx in zip(models, requests, seeds, budgets):
        m._start_request(ids, None, dict(kw), temperature=temperature, top_k=top_k, seed=sd, max_new_tokens=mx, is_llama3=is_llama3)
    rounds_cap = max_length - lead.spec_layer.total_tokens - 10  # :270
    alive = [True] * n
    final = [m.engine.state() for m in models]
    idxs, accs = [0] * n, [[] for _ in range(n)]
    member_engines = [m.engine for m in models[1:]]
    if stats is not None:
        torch.cuda.current_stream().synchronize()
        t_loop = time.time()
    for idx in range(rounds_cap):
        fa = -1 if forced_accept is None else int(forced_accept(idx))
        lead.engine.cohort_round(member_engines, fa)
        states = lead.engine.cohort_states(member_engines)  # the round's ONE host synchronisation
        for t, m in enumerate(models):
            if not alive[t]:
                continue
            st = states[t]
            final[t], idxs[t] = st, idx
            accs[t].append(int(st["accept_len"]))
            if (st["done"] & 1) or st["new_token"] > budgets[t] or (st["done"] & 4):  # :544 / :546 / KV full
                if st["done"] & 4 and not (st["done"] & 3):
                    import warnings
                    warnings.warn(f"cohort request {t} stopped after {st['new_token']} new tokens: the next round would not fit a KV cache",
                                  RuntimeWarning)
                alive[t] = False
        if not any(alive):
            break
    if stats is not None:
        torch.cuda.current_stream().synchronize()
        stats.update(decode_s=time.time() - t_loop, rounds=idx + 1)
    outs = []
    for t, m in enumerate(models):
        n_ctx = final[t]["n_ctx"]
        m.current_length_data.fill_(n_ctx)
        toks = torch.from_numpy(m.engine.tokens(n_ctx).astype(np.int64)).to(m.engine.device)[None]
        outs.append((toks, final[t]["new_token"], idxs[t], accs[t]))
    return outs


def specgenerate_stream(models, requests, temperature=0.0, top_k=0.0, max_new_tokens=512, max_length=2048, is_llama3=False, seeds=None,
                        stats=None):
    """Any number of independent requests through the request SLOTS of one cohort (continuous batching): `models` = [leader, member, ...]
    as for specgenerate_cohort, `requests` = [(input_ids [1,L], specgenerate kwargs), ...] in arrival order.  The first len(models)
    requests start together; every lockstep round serves all slots on one weight pass, and the moment a request finishes (EOS, its
    token budget, a full KV cache) its slot takes the next request of the queue — prefill, first token and draft prefill of that request
    on the same stream, then it simply joins the following rounds (the other slots' state is device-resident and waits).  A cohort run
    request by request (specgenerate_cohort) executes max(rounds of its requests) lockstep rounds; the stream executes ≈ mean(rounds).
    Every request keeps the reference's batch-1 semantics: with up to four slots it returns exactly the (input_ids, new_token, idx,
    acceptance_len) tuple of `m.specgenerate(ids, log=True, return_acceptance_len=True, ...)` run alone (spec_model_ours.py:247-582), in
    request order; with five to eight slots the tuple of the c8 arithmetic class (see specgenerate_cohort) — independent of what shares the
    slots.  When len(requests) <= len(models) the call falls back to plain cohorts of len(group) requests, so the class then follows the
    number of requests that arrive together (4 or fewer: solo arithmetic; 5 or more: c8).
    `stats` (dict, optional): `rounds` = lockstep rounds executed, `request_rounds` = rounds summed over the requests."""
    n, R = len(models), len(requests)
    seeds = list(seeds) if seeds is not None else [0] * R
    budgets = list(max_new_tokens) if isinstance(max_new_tokens, (list, tuple)) else [max_new_tokens] * R
    if len(seeds) != R or len(budgets) != R:
        raise ValueError("one seed and one token budget per request")
    lead = models[0]
    rounds_cap = max_length - lead.spec_layer.total_tokens - 10  # :270
    if R <= n or n < 2 or rounds_cap < 1:  # nothing to refill: the plain cohort (or single-request) loops
        outs, st_all = [], dict(rounds=0, request_rounds=0)
        for lo in range(0, R, max(n, 1)):
            grp = list(range(lo, min(R, lo + max(n, 1))))
            if len(grp) >= 2 and n >= 2:
                st = {}
                got = specgenerate_cohort(models[:len(grp)], [requests[i] for i in grp], temperature=temperature, top_k=top_k,
                                          max_new_tokens=[budgets[i] for i in grp], max_length=max_length, is_llama3=is_llama3,
                                          seeds=[seeds[i] for i in grp], stats=st)
                st_all["rounds"] += st["rounds"]
            else:
                got = []
                for i in grp:
                    ids, kw = requests[i]
                    got.append(lead.specgenerate(ids, temperature=temperature, top_k=top_k, max_new_tokens=budgets[i], max_length=max_length,
                                                 log=True, is_llama3=is_llama3, return_acceptance_len=True, seed=seeds[i], **kw))
                st_all["rounds"] += sum(g[2] + 1 for g in got)
            st_all["request_rounds"] += sum(g[2] + 1 for g in got)
            outs += got
        if stats is not None:
            stats.update(st_all)
        return outs
    for m in models[1:]:
        if m.engine.leader is not lead.engine:
            raise ValueError("models[1:] must have been built with cohort_leader=models[0]")
    _sync_tree_size(models)
    member_engines = [m.engine for m in models[1:]]
    outs = [None] * R
    slot = [None] * n  # request index a slot is working on
    rnd, accs = [0] * n, [[] for _ in range(n)]
    nxt = 0

    def start(t):
        nonlocal nxt
        i, nxt = nxt, nxt + 1
        ids, kw = requests[i]
        models[t]._start_request(ids, None, dict(kw), temperature=temperature, top_k=top_k, seed=seeds[i], max_new_tokens=budgets[i],
                                 is_llama3=is_llama3)
        slot[t], rnd[t], accs[t] = i, 0, []

    def finish(t, st):
        m, n_ctx = models[t], st["n_ctx"]
        m.current_length_data.fill_(n_ctx)
        toks = torch.from_numpy(m.engine.tokens(n_ctx).astype(np.int64)).to(m.engine.device)[None]
        outs[slot[t]] = (toks, st["new_token"], rnd[t] - 1, accs[t])
        slot[t] = None

    # One round of LOOKAHEAD (round 4; VISPEC_STREAM_LOOKAHEAD=0 restores the lockstep loop): round k + 1 is launched before round k's states are
    # read, so the stream never waits for the host between rounds (state snapshots travel through two pinned slots in stream order:
    # Engine.cohort_states_enqueue / _wait).  A request that finished in round k is frozen on the device (EOS, budget and cache limits are
    # decided there: tree_kernels.h), so the round already in flight leaves it untouched; its slot rides along idle for that one round and the
    # refill joins from the round after.  valid_from[t] = the first launched round whose snapshot describes slot t's CURRENT request.
    import os
    lookahead = os.environ.get("VISPEC_STREAM_LOOKAHEAD", "1") != "0"
    valid_from = [0] * n
    k_enq = k_done = 0

    def launch():
        nonlocal k_enq
        lead.engine.cohort_round(member_engines, -1)
        lead.engine.cohort_states_enqueue(member_engines, k_enq & 1)
        k_enq += 1

    for t in range(n):
        start(t)
    lockstep = request_rounds = 0
    launch()
    while any(i is not None for i in slot):
        if lookahead and k_enq - k_done < 2:
            launch()
        states = lead.engine.cohort_states_wait(member_engines, k_done & 1)  # waits for that round's snapshot only
        k, k_done = k_done, k_done + 1
        lockstep += 1
        for t in range(n):
            if slot[t] is None or k < valid_from[t]:
                continue  # an idle slot (the queue is empty / its new request joins a later round): its frozen rows ride along
            st = states[t]
            rnd[t] += 1
            request_rounds += 1
            accs[t].append(int(st["accept_len"]))
            if (st["done"] & 1) or st["new_token"] > budgets[slot[t]] or (st["done"] & 4) or rnd[t] >= rounds_cap:  # :544 / :546 / KV full / :484
                if st["done"] & 4 and not (st["done"] & 3):
                    import warnings
                    warnings.warn(f"request {slot[t]} stopped after {st['new_token']} new tokens: the next round would not fit a KV cache",
                                  RuntimeWarning)
                finish(t, st)
        for t in range(n):
            if slot[t] is None and nxt < R:
                start(t)
                valid_from[t] = k_enq  # it is part of the rounds launched from now on
        if k_enq == k_done and any(i is not None for i in slot):
            launch()
    # the round launched ahead of the last snapshot (lookahead) is still in flight: wait for its snapshot too, so that nothing of this call is
    # running on the stream when it returns.  (A request ended by the HOST-only rule `rnd >= rounds_cap` (:484) is not frozen on the device: the
    # round in flight may advance it once more — harmless, finish() took its tokens up to the snapshot's n_ctx, which that round does not
    # rewrite, and the slot's next _start_request resets the device state in stream order.)
    while k_done < k_enq:
        lead.engine.cohort_states_wait(member_engines, k_done & 1)
        k_done += 1
    if stats is not None:
        stats.update(rounds=lockstep, request_rounds=request_rounds)
    return outs
