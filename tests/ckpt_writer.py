"""Test infrastructure (round 6): writes a FULL-SIZE checkpoint pair in the layouts the reference loads (spec_model_ours.py:109-166) —

  <root>/<target name>/   config.json (the HF config class of the architecture, serialised by transformers itself),
                          model-0000x-of-0000y.safetensors + model.safetensors.index.json with the PUBLISHED key names
                          (llava-hf/llava-v1.6-*: `language_model.model.layers.N.*`, `language_model.lm_head.weight`, `vision_tower.*`,
                          `multi_modal_projector.*`, `image_newline`; Qwen/Qwen2.5-VL-*: `model.*`, `lm_head.weight`, `visual.*`)
  <root>/<draft name>/    config.json + model.safetensors (the ViSpec draft's key contract, SURVEY.md §8 A0: embed_tokens, fc, img_fc,
                          imadpt.{q,k_proj,v_proj,o_proj}, layers.0.*)

from the device-resident synthetic weight pair bench.py builds (vispec_amd.synth_gpu.make_pair): no published checkpoint can reach the box
(no network), so the VALUES are synthetic while every byte of the on-disk format is the real one.  The vision tower / projector are HF's own
modules at the published architecture, randomly initialised.

    python tests/ckpt_writer.py --model llava7b --out /tmp/vispec_ckpt      # -> the directory tree bench.py --weights-dir expects"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHARD_BYTES = 5 << 30


def _save_sharded(tensors, path):
    """tensors: ordered {name: cpu tensor} -> HF sharded safetensors + index."""
    from safetensors.torch import save_file
    shards, cur, cur_b = [], {}, 0
    for k, v in tensors.items():
        nb = v.numel() * v.element_size()
        if cur and cur_b + nb > SHARD_BYTES:
            shards.append(cur)
            cur, cur_b = {}, 0
        cur[k] = v
        cur_b += nb
    if cur:
        shards.append(cur)
    wm, total = {}, 0
    for i, sh in enumerate(shards):
        fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file({k: v.contiguous() for k, v in sh.items()}, os.path.join(path, fn), metadata={"format": "pt"})
        for k, v in sh.items():
            wm[k] = fn
            total += v.numel() * v.element_size()
    json.dump({"metadata": {"total_size": total}, "weight_map": wm}, open(os.path.join(path, "model.safetensors.index.json"), "w"), indent=1)
    return len(shards), total


def _language_model_tensors(tw, tcfg, prefix, head_key):
    c = lambda t: t.detach().to("cpu", torch.bfloat16).contiguous()
    hd = tcfg.head_dim
    nq, nk, I = tcfg.num_heads * hd, tcfg.num_kv_heads * hd, tcfg.intermediate_size
    out = {prefix + "embed_tokens.weight": c(tw.embed)}
    for i, lw in enumerate(tw.layers):
        p = f"{prefix}layers.{i}."
        wqkv, wgu = lw["wqkv"], lw["wgu"]
        out[p + "self_attn.q_proj.weight"], out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = c(wqkv[:nq]), c(wqkv[nq:nq + nk]), c(wqkv[nq + nk:])
        if lw.get("bqkv") is not None:
            b = lw["bqkv"]
            out[p + "self_attn.q_proj.bias"], out[p + "self_attn.k_proj.bias"], out[p + "self_attn.v_proj.bias"] = c(b[:nq]), c(b[nq:nq + nk]), c(b[nq + nk:])
        out[p + "self_attn.o_proj.weight"] = c(lw["wo"])
        out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = c(wgu[:I]), c(wgu[I:])
        out[p + "mlp.down_proj.weight"] = c(lw["wdown"])
        out[p + "input_layernorm.weight"], out[p + "post_attention_layernorm.weight"] = c(lw["ln1"]), c(lw["ln2"])
    out[prefix + "norm.weight"] = c(tw.norm)
    out[head_key] = c(tw.lm_head)
    return out


def draft_tensors(dw):
    """DraftWeightsDev (fused) -> the ViSpec draft's state-dict names (cnets_ours.py:692-717), bf16 on the host."""
    c = lambda t: t.detach().to("cpu", torch.bfloat16).contiguous()
    t = dw.t
    D, Id, H = t["wo"].shape[0], t["wdown"].shape[1], dw.cfg.num_heads
    sd = {"embed_tokens.weight": c(t["embed"]), "fc.weight": c(t["fc_w"]), "img_fc.weight": c(t["imgfc_w"]),
          "imadpt.q": c(t["ad_q"]).reshape(dw.num_q, H, D // H).contiguous(), "imadpt.o_proj.weight": c(t["ad_wo"]),
          "layers.0.post_attention_layernorm.weight": c(t["ln2"]), "layers.0.self_attn.o_proj.weight": c(t["wo"]),
          "layers.0.mlp.down_proj.weight": c(t["wdown"])}
    if t.get("fc_b") is not None:
        sd["fc.bias"] = c(t["fc_b"])
    if t.get("imgfc_b") is not None:
        sd["img_fc.bias"] = c(t["imgfc_b"])
    for j, n in enumerate("qkv"):
        sd[f"layers.0.self_attn.{n}_proj.weight"] = c(t["wqkv"][j * D:(j + 1) * D])
        if t.get("bqkv") is not None:
            sd[f"layers.0.self_attn.{n}_proj.bias"] = c(t["bqkv"][j * D:(j + 1) * D])
    sd["layers.0.mlp.gate_proj.weight"], sd["layers.0.mlp.up_proj.weight"] = c(t["wgu"][:Id]), c(t["wgu"][Id:])
    sd["imadpt.k_proj.weight"], sd["imadpt.v_proj.weight"] = c(t["ad_wkv"][:D]), c(t["ad_wkv"][D:])
    if t.get("ad_bkv") is not None:
        sd["imadpt.k_proj.bias"], sd["imadpt.v_proj.bias"] = c(t["ad_bkv"][:D]), c(t["ad_bkv"][D:])
    return sd


def write_draft_dir(path, dcfg, dw):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    save_file(draft_tensors(dw), os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    json.dump({"hidden_size": dcfg.hidden_size, "num_attention_heads": dcfg.num_heads, "intermediate_size": dcfg.intermediate_size,
               "vocab_size": dcfg.vocab_size, "max_position_embeddings": dcfg.max_position_embeddings, "rms_norm_eps": dcfg.rms_norm_eps,
               "rope_theta": dcfg.rope_theta, "qkv_bias": bool(dcfg.qkv_bias), "bias": bool(dcfg.bias)}, open(os.path.join(path, "config.json"), "w"), indent=1)


def write_llava_next_dir(path, tcfg, tw, seed=0):
    """llava-hf/llava-v1.6-vicuna-*-hf layout (transformers 4.x key names, which the published files carry)."""
    from transformers import AutoModel, LlamaConfig, LlavaNextConfig
    from transformers.models.llava_next.modeling_llava_next import LlavaNextMultiModalProjector
    os.makedirs(path, exist_ok=True)
    tc = LlamaConfig(vocab_size=tcfg.vocab_size, hidden_size=tcfg.hidden_size, intermediate_size=tcfg.intermediate_size,
                     num_hidden_layers=tcfg.num_layers, num_attention_heads=tcfg.num_heads, num_key_value_heads=tcfg.num_kv_heads,
                     rms_norm_eps=tcfg.rms_norm_eps, max_position_embeddings=4096)
    cfg = LlavaNextConfig(text_config=tc, image_token_index=tcfg.image_token_index)  # vision_config default = CLIP ViT-L/14-336, the published tower
    cfg.architectures = ["LlavaNextForConditionalGeneration"]
    cfg.save_pretrained(path)
    torch.manual_seed(seed)
    tower = AutoModel.from_config(cfg.vision_config).to(torch.bfloat16)
    proj = LlavaNextMultiModalProjector(cfg).to(torch.bfloat16)
    tensors = _language_model_tensors(tw, tcfg, "language_model.model.", "language_model.lm_head.weight")
    for k, v in tower.state_dict().items():  # (transformers 5.x dropped the inner `vision_model.` level of CLIPVisionModel: the published files have it)
        tensors["vision_tower." + (k if k.startswith("vision_model.") else "vision_model." + k)] = v.detach().cpu().contiguous()
    for k, v in proj.state_dict().items():
        tensors["multi_modal_projector." + k] = v.detach().cpu().contiguous()
    tensors["image_newline"] = (torch.randn(tcfg.hidden_size) * 0.02).to(torch.bfloat16)
    return _save_sharded(tensors, path)


def write_qwen25vl_dir(path, tcfg, tw, seed=0):
    """Qwen/Qwen2.5-VL-7B-Instruct layout: `model.*`, `lm_head.weight`, `visual.*`."""
    from transformers import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel as Visual
    os.makedirs(path, exist_ok=True)
    text = dict(vocab_size=tcfg.vocab_size, hidden_size=tcfg.hidden_size, intermediate_size=tcfg.intermediate_size, num_hidden_layers=tcfg.num_layers,
                num_attention_heads=tcfg.num_heads, num_key_value_heads=tcfg.num_kv_heads, rms_norm_eps=tcfg.rms_norm_eps)
    # the published 7B tower: 32 blocks of width 1280 / MLP 3420, merger to the text width, 2 temporal tokens per second
    vision = dict(hidden_size=1280, intermediate_size=3420, num_heads=16, depth=32, out_hidden_size=tcfg.hidden_size, tokens_per_second=int(tcfg.tokens_per_second))
    try:  # transformers 5.x: nested text_config
        cfg = Qwen2_5_VLConfig(text_config=text, vision_config=vision, image_token_id=tcfg.image_token_index, video_token_id=tcfg.video_token_id)
    except TypeError:  # 4.x: the text fields live on the top level
        cfg = Qwen2_5_VLConfig(vision_config=vision, image_token_id=tcfg.image_token_index, video_token_id=tcfg.video_token_id, **text)
    vc = cfg.vision_config
    cfg.architectures = ["Qwen2_5_VLForConditionalGeneration"]
    cfg.save_pretrained(path)
    torch.manual_seed(seed)
    visual = Visual._from_config(vc).to(torch.bfloat16)
    tensors = _language_model_tensors(tw, tcfg, "model.", "lm_head.weight")
    for k, v in visual.state_dict().items():
        tensors["visual." + k] = v.detach().cpu().contiguous()
    return _save_sharded(tensors, path)


def write_pair(root, model, device, seed=0):
    """-> (target dir, draft dir, tcfg, dcfg, tw, dw, info): the bench's synthetic pair of `model` under its hub names below `root`."""
    import bench
    bench.MODEL = model
    tcfg, dcfg, tw, dw = bench.synth_pair(device, seed)
    tname, dname = bench.HUB_NAMES[model.split("-")[0]]
    tdir, ddir = os.path.join(root, tname), os.path.join(root, dname)
    if tcfg.architectures[0] == "Qwen2_5_VLForConditionalGeneration":
        n_shards, nbytes = write_qwen25vl_dir(tdir, tcfg, tw, seed)
    else:
        n_shards, nbytes = write_llava_next_dir(tdir, tcfg, tw, seed)
    write_draft_dir(ddir, dcfg, dw)
    return tdir, ddir, tcfg, dcfg, tw, dw, dict(target_shards=n_shards, target_GB=round(nbytes / 1e9, 2))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llava7b")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    tdir, ddir, *_, info = write_pair(a.out, a.model, torch.device("cuda:0"))
    print(json.dumps(dict(target=tdir, draft=ddir, **info)))
