"""On-disk formats: HF safetensors checkpoints of the target's language model (single file or
sharded with model.safetensors.index.json) and the ViSpec draft directory (config.json + model.safetensors |
pytorch_model.bin), as read by reference spec_model_ours.py:147-166 and cnets_ours.py:692-717.  Local paths only."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from .engine import DraftConfig, TargetConfig


def _open_all(path):
    from safetensors import safe_open
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        files = sorted(set(json.load(open(idx))["weight_map"].values()))
    else:
        files = [f for f in sorted(os.listdir(path)) if f.endswith(".safetensors")]
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    for f in files:
        with safe_open(os.path.join(path, f), framework="pt", device="cpu") as sf:
            for k in sf.keys():
                yield k, sf.get_tensor(k)


def load_target_dir(path):
    """-> (TargetConfig, {HF-Llama-named tensors}, tokenizer-or-stub).  Accepts LLaVA(-NeXT) checkpoints in both key
    layouts (`language_model.model.*` of transformers 4.x and `model.language_model.*` of 5.x) and plain Llama."""
    cfg = json.load(open(os.path.join(path, "config.json")))
    tc = cfg.get("text_config", cfg)
    arch = cfg.get("architectures", ["LlamaForCausalLM"])[0]
    H = tc.get("num_attention_heads", 32)
    eos = tc.get("eos_token_id", cfg.get("eos_token_id", 2))
    kw = {}
    if arch == "Qwen2_5_VLForConditionalGeneration":  # modeling_qwen2_5_vl_kv.py: q/k/v bias, SDPA scores, multimodal rotary sections
        rs = tc.get("rope_scaling") or tc.get("rope_parameters") or cfg.get("rope_scaling") or {}
        kw = dict(qkv_bias=True, attn_impl="sdpa", mrope_section=tuple(rs.get("mrope_section", (16, 24, 24))),
                  image_token_index=cfg.get("image_token_id", 151655), video_token_id=cfg.get("video_token_id", 151656),
                  tokens_per_second=float((cfg.get("vision_config") or {}).get("tokens_per_second", 2)),
                  max_position_embeddings=4096)  # kv_cache.py:88-119 sizes Qwen's cache at 4096 rows
        rope_theta = tc.get("rope_theta") or rs.get("rope_theta") or 1e6
    else:
        kw = dict(image_token_index=cfg.get("image_token_index", 32000))
        if arch == "Qwen2ForCausalLM":  # modeling_qwen2_kv.py: Llama decoder with q/k/v bias, eager scores
            kw["qkv_bias"] = True
        rope_theta = tc.get("rope_theta") or (tc.get("rope_parameters") or {}).get("rope_theta") or 10000.0
    tcfg = TargetConfig(
        hidden_size=tc.get("hidden_size", 4096), num_heads=H, num_kv_heads=tc.get("num_key_value_heads", H),
        intermediate_size=tc.get("intermediate_size", 11008), vocab_size=tc.get("vocab_size", 32064),
        num_layers=tc.get("num_hidden_layers", 32), rms_norm_eps=tc.get("rms_norm_eps", 1e-5),
        rope_theta=float(rope_theta), architectures=(arch,), eos_token_id=eos if isinstance(eos, int) else 2, **kw)
    sd = {}
    for k, v in _open_all(path):
        for pre in ("language_model.model.", "model.language_model."):
            if k.startswith(pre):
                sd["model." + k[len(pre):]] = v
                break
        else:
            if k in ("language_model.lm_head.weight", "lm_head.weight"):
                sd["lm_head.weight"] = v
            elif k.startswith("model.") and "vision" not in k and "visual" not in k and "projector" not in k and "image_newline" not in k:
                sd[k] = v
    if "lm_head.weight" not in sd:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    try:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(path, local_files_only=True)
    except Exception:
        tok = SimpleNamespace(eos_token_id=tcfg.eos_token_id, vocab_size=tcfg.vocab_size)
    return tcfg, sd, tok


def load_draft_dir(path, tcfg: TargetConfig):
    cfg = json.load(open(os.path.join(path, "config.json")))
    dcfg = DraftConfig(hidden_size=cfg["hidden_size"], num_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                       vocab_size=cfg["vocab_size"], max_position_embeddings=cfg.get("max_position_embeddings", 4096),
                       rms_norm_eps=cfg.get("rms_norm_eps", 1e-5), rope_theta=cfg.get("rope_theta", 10000.0),
                       qkv_bias=bool(cfg.get("qkv_bias", False)), bias=bool(cfg.get("bias", True)))
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(os.path.join(path, "pytorch_model.bin")):  # spec_model_ours.py:152-160 prefers the .bin
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    elif os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        raise FileNotFoundError(f"no draft weights under {path}")
    return dcfg, sd
