"""Vision front-end of the target VLM for REAL checkpoints: vision tower + multimodal projector
(+ LLaVA-NeXT anyres packing with `image_newline`, or Qwen2.5-VL's `visual` tower).  It stays on PyTorch-ROCm — HF's own
modules, instantiated from the checkpoint's config and filled from its safetensors — exactly as the reference delegates to
`base_model.get_image_features` / `pack_image_features` (spec_model_ours.py:341-356, 395-401).  Only the vision side is
built: the language model's 7-13 B parameters are never instantiated twice.

Local directories only.  Without such a checkpoint `TargetLM` falls back to `SyntheticVision` (bench / tests)."""
from __future__ import annotations

import json
import os
import threading
from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

_VISION_PREFIXES = {
    "tower": ("model.vision_tower.", "vision_tower."),
    "proj": ("model.multi_modal_projector.", "multi_modal_projector."),
    "newline": ("model.image_newline", "image_newline"),
    "visual": ("model.visual.", "visual."),
}


def _iter_tensors(path):
    from safetensors import safe_open
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        wm = json.load(open(idx))["weight_map"]
        files = sorted({f for k, f in wm.items() if "vision" in k or "visual" in k or "projector" in k or "image_newline" in k})
    else:
        files = [f for f in sorted(os.listdir(path)) if f.endswith(".safetensors")]
    for f in files:
        with safe_open(os.path.join(path, f), framework="pt", device="cpu") as sf:
            for k in sf.keys():
                if "vision" in k or "visual" in k or "projector" in k or "image_newline" in k:
                    yield k, sf.get_tensor(k)


def _strip(key, prefixes):
    for p in prefixes:
        if key.startswith(p):
            return key[len(p):]
    return None


_tables_lock = threading.Lock()  # the per-grid tables below are shared by every lane thread / HIP stream of a process


def _window_groups(cu_seqlens, dev):
    """Chunk table of _qwen_vision_attention_batched for one `cu_seqlens` tensor: per chunk length, the [chunks, length] gather indices.
    Built under the lock and PUBLISHED (as an attribute of the tensor object, which a cached grid entry keeps alive across requests) only after
    the building stream has finished — another lane's stream may gather through it right away, and no stream dependency links the lanes."""
    with _tables_lock:
        groups = getattr(cu_seqlens, "_vispec_groups", None)
        if groups is None:
            bounds = cu_seqlens.tolist()
            by_len = {}
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                if hi > lo:
                    by_len.setdefault(hi - lo, []).append(lo)
            groups = [(torch.tensor(starts, device=dev)[:, None] + torch.arange(n, device=dev)[None, :]) for n, starts in sorted(by_len.items())]
            if groups and groups[0].is_cuda:
                torch.cuda.current_stream(dev).synchronize()
            cu_seqlens._vispec_groups = groups
    return groups


def _qwen_vision_attention_batched(self, hidden_states, cu_seqlens, position_embeddings=None, **kwargs):
    """Drop-in for transformers' Qwen2_5_VLVisionAttention.forward on its non-flash path.  HF splits the sequence at `cu_seqlens` and calls the
    attention once per chunk in a Python loop, after a `.tolist()` of a device tensor — for one 1280x960 image that is ~110 windows x 28 windowed
    layers = ~3 000 attention calls and 32 host synchronisations per image set (measured: the tower costs the Qwen lines of bench.py 17-36 %
    when it runs inside the timed request, profiles/r05_vision_in_region_ab.txt).  Same arithmetic here — qkv, rotary, softmax(q k^T / sqrt(d)) v per
    chunk, projection — with the chunks of equal length attended as ONE batched scaled_dot_product_attention call (windows come in at most a handful
    of lengths), and the chunk table computed once per `cu_seqlens` tensor (the tower hands the same two tensors to all its layers)."""
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import apply_rotary_pos_emb_vision
    S = hidden_states.shape[0]
    q, k, v = self.qkv(hidden_states).reshape(S, 3, self.num_heads, -1).permute(1, 0, 2, 3).unbind(0)  # [S, H, hd] each
    cos, sin = position_embeddings
    q, k = apply_rotary_pos_emb_vision(q, k, cos, sin)
    groups = getattr(cu_seqlens, "_vispec_groups", None)
    if groups is None:
        groups = _window_groups(cu_seqlens, hidden_states.device)
    out = torch.empty_like(q)
    for idx in groups:  # idx [chunks of this length, length]
        o = torch.nn.functional.scaled_dot_product_attention(q[idx].transpose(1, 2), k[idx].transpose(1, 2), v[idx].transpose(1, 2), scale=self.scaling)
        out[idx] = o.transpose(1, 2)
    return self.proj(out.reshape(S, -1))


class HFVisionFrontEnd:
    """`features(pixel_values, image_sizes=None, image_grid_thw=None)` -> [n_image_tokens, D_text] in prompt order.
    batched_windows (Qwen2.5-VL tower, default on): the window attention of every vision block runs as one batched SDPA call per chunk length
    instead of HF's per-window Python loop (_qwen_vision_attention_batched) — the tower stays HF's module on PyTorch-ROCm, only its attention call
    pattern changes; False = HF's own forward, call for call."""

    def __init__(self, arch: str, config, tower: nn.Module, projector: Optional[nn.Module], image_newline: Optional[torch.Tensor],
                 batched_windows: bool = True):
        self.arch, self.config, self.tower, self.projector, self.image_newline = arch, config, tower, projector, image_newline
        self._batched_windows, self._grid_tables = False, {}
        if arch == "Qwen2_5_VLForConditionalGeneration" and batched_windows:
            import types
            vc = getattr(config, "vision_config", None)
            if getattr(vc, "_attn_implementation", "sdpa") in ("sdpa", "eager", None):  # (flash_attention_2 takes HF's own varlen call)
                for blk in getattr(tower, "blocks", []):
                    blk.attn.forward = types.MethodType(_qwen_vision_attention_batched, blk.attn)
                # the cached tables are handed to the forward as keyword arguments ONLY next to the patched attention: HF passes unknown
                # kwargs through every block into its attention interface, where the flash wrapper reads a stray `position_ids` as a
                # packed-sequence hint
                self._batched_windows = True

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_dir(cls, path: str, device="cpu", dtype=torch.bfloat16, batched_windows: bool = True) -> "HFVisionFrontEnd":
        from transformers import AutoConfig, AutoModel
        config = AutoConfig.from_pretrained(path, local_files_only=True)
        arch = (config.architectures or ["?"])[0]
        buckets = {k: {} for k in _VISION_PREFIXES}
        for k, v in _iter_tensors(path):
            for name, pre in _VISION_PREFIXES.items():
                s = _strip(k, pre)
                if s is not None:
                    buckets[name][s] = v
                    break
        if arch in ("LlavaNextForConditionalGeneration", "LlavaForConditionalGeneration"):
            if arch == "LlavaNextForConditionalGeneration":
                from transformers.models.llava_next.modeling_llava_next import LlavaNextMultiModalProjector as Proj
            else:
                from transformers.models.llava.modeling_llava import LlavaMultiModalProjector as Proj
            tower = AutoModel.from_config(config.vision_config)
            proj = Proj(config)
            _load(tower, buckets["tower"], "vision_tower")
            _load(proj, buckets["proj"], "multi_modal_projector")
            newline = buckets["newline"].get("", None)
            if arch == "LlavaNextForConditionalGeneration" and newline is None:
                raise FileNotFoundError(f"{path}: no image_newline tensor in the checkpoint")
            fe = cls(arch, config, tower.to(device, dtype).eval(), proj.to(device, dtype).eval(),
                     None if newline is None else newline.to(device, dtype))
        elif arch == "Qwen2_5_VLForConditionalGeneration":
            from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel as Visual
            visual = Visual._from_config(config.vision_config)
            _load(visual, buckets["visual"], "visual")
            fe = cls(arch, config, visual.to(device, dtype).eval(), None, None, batched_windows=batched_windows)
        else:
            raise NotImplementedError(f"no vision front-end for {arch}")
        return fe

    # ------------------------------------------------------------------------------------------------
    def _qwen_grid_tables(self, grid_thw, device) -> dict:
        """The index tables HF's Qwen2.5-VL tower derives from `grid_thw` at the top of every forward — rotary position ids, the window
        permutation, the window and image boundaries (Python loops over the images with `.tolist()` round trips) — computed ONCE per distinct
        grid with HF's own helpers and handed to the forward through the keyword arguments those helpers look up first (`position_ids`,
        `window_index`, `cu_window_seqlens`, `cu_seqlens`).  Same tables, same forward; a serving process sees a handful of grids.  The cached
        boundary tensors also keep the chunk tables of _qwen_vision_attention_batched across requests: no host synchronisation per image set."""
        if not self._batched_windows:
            return {}
        key = tuple(tuple(int(v) for v in row) for row in (grid_thw.tolist() if torch.is_tensor(grid_thw) else grid_thw))
        hit = self._grid_tables.get((key, str(device)))
        if hit is None:
            # Built under the lock and published only when the building stream has finished: every lane thread (own HIP stream, no dependency
            # on the builder's) reads the entry the moment it is in the dict.  The chunk tables of the two boundary tensors are built here as
            # well, so that no consumer ever runs `.tolist()` on a tensor another stream is still writing.
            with _tables_lock:
                hit = self._grid_tables.get((key, str(device)))
                if hit is None:
                    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as M
                    t = self.tower
                    g = torch.as_tensor(key, device=device)
                    try:
                        window_index, cu_window = M.get_vision_window_index(g, spatial_merge_size=t.spatial_merge_size, window_size=t.window_size,
                                                                            patch_size=t.patch_size)
                        cu, _ = M.get_vision_attention_seqlens(g, t.config)
                        hit = dict(position_ids=M.get_vision_position_ids(g, t.spatial_merge_size), window_index=window_index, cu_window_seqlens=cu_window,
                                   cu_seqlens=cu)
                    except (AttributeError, TypeError):  # another transformers layout: the tower computes its tables itself
                        hit = {}
                    if torch.device(device).type == "cuda":
                        torch.cuda.current_stream(device).synchronize()
                    if len(self._grid_tables) > 64:
                        self._grid_tables.clear()
                    self._grid_tables[(key, str(device))] = hit
            for name in ("cu_window_seqlens", "cu_seqlens"):  # (takes the lock itself; after the entry's tensors are complete)
                if torch.is_tensor(hit.get(name)):
                    _window_groups(hit[name], device)
        return dict(hit)

    @torch.no_grad()
    def features(self, pixel_values, image_sizes=None, image_grid_thw=None, vision_feature_layer=None,
                 vision_feature_select_strategy=None) -> torch.Tensor:
        c = self.config
        if self.arch == "Qwen2_5_VLForConditionalGeneration":
            dev = next(self.tower.parameters())
            out = self.tower(pixel_values.to(dev.device, dev.dtype), grid_thw=image_grid_thw.to(dev.device), **self._qwen_grid_tables(image_grid_thw, dev.device))
            out = getattr(out, "pooler_output", out)  # transformers 5.x wraps the merged tokens
            return out if torch.is_tensor(out) else torch.cat(list(out), dim=0)
        layer = c.vision_feature_layer if vision_feature_layer is None else vision_feature_layer
        strategy = c.vision_feature_select_strategy if vision_feature_select_strategy is None else vision_feature_select_strategy
        dev = next(self.tower.parameters())
        if self.arch == "LlavaNextForConditionalGeneration":
            from transformers.models.llava_next import modeling_llava_next as M
            n_patches = [M.image_size_to_num_patches(image_size=s, grid_pinpoints=c.image_grid_pinpoints, patch_size=c.vision_config.image_size)
                         for s in (image_sizes.tolist() if torch.is_tensor(image_sizes) else image_sizes)]
            if pixel_values.dim() == 5:  # [images, max_patches, C, H, W] padded -> the real patches of every image
                pixel_values = torch.cat([pv[:n] for pv, n in zip(pixel_values, n_patches)], dim=0)
            elif pixel_values.dim() != 4:
                raise ValueError(f"pixel_values of shape {pixel_values.shape}, expect to be of 4 or 5 dimensions")
        hs = self.tower(pixel_values.to(dev.device, dev.dtype), output_hidden_states=True, return_dict=True).hidden_states
        sel = hs[layer] if isinstance(layer, int) else torch.cat([hs[i] for i in layer], dim=-1)
        if strategy == "default":
            sel = sel[:, 1:]  # drop CLS
        feats = self.projector(sel)
        if self.arch == "LlavaForConditionalGeneration":
            return feats.reshape(-1, feats.shape[-1])
        from transformers.models.llava_next.modeling_llava_next import LlavaNextModel
        shim = SimpleNamespace(config=c)
        packed, _ = LlavaNextModel.pack_image_features(shim, torch.split(feats, n_patches, dim=0), image_sizes,
                                                       vision_feature_select_strategy=strategy, image_newline=self.image_newline)
        return packed if torch.is_tensor(packed) else torch.cat(list(packed), dim=0)  # 4.x concatenates, 5.x returns the list


def _load(module: nn.Module, sd: dict, what: str):
    if not sd:
        raise FileNotFoundError(f"no {what} tensors in the checkpoint")
    # The published llava-hf checkpoints carry transformers 4.x names (`vision_tower.vision_model.encoder...`); transformers 5.x builds the
    # same CLIP tower without the inner `vision_model.` level (and a 4.x module reading a 5.x-written file is the reverse): follow the module.
    own = list(module.state_dict().keys())
    inner = "vision_model."
    if own and sd:
        has_own, has_sd = own[0].startswith(inner), next(iter(sd)).startswith(inner)
        if has_sd and not has_own and all(k.startswith(inner) for k in sd):
            sd = {k[len(inner):]: v for k, v in sd.items()}
        elif has_own and not has_sd and all(k.startswith(inner) for k in own):
            sd = {inner + k: v for k, v in sd.items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k and "inv_freq" not in k]
    if missing or unexpected:
        raise RuntimeError(f"{what}: missing {missing[:4]} unexpected {list(unexpected)[:4]}")
