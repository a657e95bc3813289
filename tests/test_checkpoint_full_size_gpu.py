"""A FULL-SIZE checkpoint pair through the real loader (round 6; reference: SpecModel.from_pretrained, spec_model_ours.py:109-166).

Every earlier load test used tiny random directories.  Here tests/ckpt_writer.py writes, for LLaVA-v1.6-vicuna-7B and for Qwen2.5-VL-7B, an
HF-layout target directory at the published size — sharded `model-0000x-of-0000y.safetensors` + index with the published key names, the HF
config written by transformers, the vision tower / projector / image_newline (or `visual.*`) of the published architecture — and a ViSpec draft
directory (config.json + model.safetensors, the A0 key contract); the VALUES are bench.py's synthetic pair (no network on the box).  Then

    SpecModel.from_pretrained(base_model_path, spec_model_path)  ->  specgenerate(input_ids, pixel_values = PIXELS, ...)

must equal, token for token and accept length for accept length, the same weights handed over in memory (SpecModel via bench.build_models'
path, fed the image features the loaded tower computes).  Load time, peak host RSS and the bytes on disk are printed (the 14-15 GB shard reads,
the W32 packing of 291 tensors and the HF vision tower from the same shards had never run at size).

Needs ~17 GB under $VISPEC_CKPT_TMP (default /tmp) per model, one model at a time; skipped when less than 24 GB are free."""
import gc
import json
import os
import resource
import shutil
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
pytest.importorskip("transformers")
pytest.importorskip("safetensors")


@pytest.mark.parametrize("model", ["llava7b", "qwen7b"])
def test_full_size_checkpoint_through_from_pretrained(model):
    import bench
    from ckpt_writer import write_pair
    from vispec_amd.model import SpecModel
    from vispec_amd.model.cnets_ours import Model
    from vispec_amd.model.target import TargetLM
    base = os.environ.get("VISPEC_CKPT_TMP", "/tmp")
    free_gb = shutil.disk_usage(base).free / 1e9
    if free_gb < 24:
        pytest.skip(f"{free_gb:.0f} GB free under {base}: a full-size checkpoint needs ~17 GB")
    root = os.path.join(base, f"vispec_ckpt_{os.getpid()}_{model}")
    dev = torch.device("cuda:0")
    bench.MODEL = model
    try:
        t0 = time.time()
        tdir, ddir, tcfg, dcfg, tw, dw, info = write_pair(root, model, dev)
        t_write = time.time() - t0
        idx = json.load(open(os.path.join(tdir, "model.safetensors.index.json")))
        keys = list(idx["weight_map"])
        assert info["target_shards"] >= 3 and len(set(idx["weight_map"].values())) == info["target_shards"]
        if model == "llava7b":  # the published llava-hf key names
            assert "language_model.model.layers.31.mlp.down_proj.weight" in keys and "language_model.lm_head.weight" in keys
            assert any(k.startswith("vision_tower.vision_model.encoder.layers.23.") for k in keys) and "image_newline" in keys
            assert "multi_modal_projector.linear_1.weight" in keys
        else:
            assert "model.layers.27.self_attn.q_proj.bias" in keys and "lm_head.weight" in keys and any(k.startswith("visual.blocks.31.") for k in keys)
        # ---- the real loader
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
        t0 = time.time()
        sm = SpecModel.from_pretrained(base_model_path=tdir, spec_model_path=ddir, device=str(dev), **bench.TREE)
        torch.cuda.synchronize()
        t_load = time.time() - t0
        rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
        assert sm.base_model.cfg.num_layers == tcfg.num_layers and sm.base_model.cfg.hidden_size == tcfg.hidden_size
        assert sm.base_model.cfg.architectures[0] == tcfg.architectures[0] and hasattr(sm.base_model.vision, "tower")
        # the loaded tensors are the written ones, bit for bit (fused back by the loader)
        for a, b in zip(sm.engine.tw.tensors(), tw.tensors()):
            assert a.shape == b.shape and torch.equal(a, b)
        for k, v in dw.t.items():
            if v is not None:
                assert torch.equal(sm.engine.dw.t[k], v), k
        # ---- one request through the loaded model, pixels in: HF tower -> features -> merge -> prefill -> rounds
        n_img = {"qwen7b": 1024}.get(model, bench.N_IMG)
        ids, kw = bench._make_request(tcfg, 5, dev)
        g = torch.Generator(device="cpu").manual_seed(5005)
        if model == "qwen7b":  # four 32 x 32-patch images: [patches, channels x temporal patch x patch^2] rows, as HF's processor emits them
            pix, sizes = torch.randn(4 * 32 * 32, 3 * 2 * 14 * 14, generator=g).to(dev, torch.bfloat16), None
        else:  # five anyres tiles of a 640 x 427 image -> 2144 image tokens after unpadding + newline packing
            pix, sizes = torch.randn(1, 5, 3, 336, 336, generator=g).to(dev, torch.bfloat16), torch.tensor([[427, 640]])
        gkw = {k: v for k, v in kw.items() if k != "pixel_values"}
        if sizes is not None:
            gkw["image_sizes"] = sizes
        out, new_token, idx_, acc = sm.specgenerate(ids, pixel_values=pix, max_new_tokens=64, log=True, return_acceptance_len=True, **gkw)
        assert new_token > 64 and len(acc) == idx_ + 1
        feats = sm.base_model.get_image_features(pix, gkw.get("image_sizes"), **({"image_grid_thw": gkw["image_grid_thw"]} if "image_grid_thw" in gkw else {}))
        assert feats.shape == (n_img, tcfg.hidden_size)
        # ---- the same weights handed over in memory, fed the features
        sm2 = SpecModel(TargetLM(tcfg, tw), Model(dcfg, dw, total_tokens=bench.TREE["total_token"], depth=bench.TREE["depth"], top_k=bench.TREE["top_k"],
                                                  num_q=bench.TREE["num_q"]), **bench.TREE)
        out2, new2, idx2, acc2 = sm2.specgenerate(ids, pixel_values=feats, max_new_tokens=64, log=True, return_acceptance_len=True,
                                                  **{k: v for k, v in gkw.items() if k != "image_sizes"})
        np.testing.assert_array_equal(out[0].cpu().numpy(), out2[0].cpu().numpy())
        assert (new_token, idx_, acc) == (new2, idx2, acc2) and np.mean(acc) > 1.0
        print(f"{model}: wrote {info['target_GB']} GB in {info['target_shards']} shards ({t_write:.0f} s); from_pretrained {t_load:.1f} s, peak host RSS "
              f"{rss0:.1f} -> {rss1:.1f} GB; {new_token} tokens, tau {np.mean(acc):.2f}, == in-memory weights")
    finally:
        shutil.rmtree(root, ignore_errors=True)
        bench.MODEL = "llava7b"
        gc.collect()
        torch.cuda.empty_cache()
