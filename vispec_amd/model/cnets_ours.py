"""ViSpec draft (`spec_layer`): vision adaptor + single decoder layer + dynamic tree drafter.
API of reference vispec/model/cnets_ours.py `Model` (:664-1238): topK_genrate / reset_kv / init_tree / reset and the
state attributes stable_kv / last_img_hidden / tree_mask.  All arithmetic is in libvispec_hip
(vispec_draft_prefill / vispec_draft_round); this class only carries weights and converts the device-side tree into
the reference's return tuple when a caller asks for it."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..engine import DraftConfig, DraftWeightsDev, Engine


class Model:
    def __init__(self, config: DraftConfig, weights: DraftWeightsDev, total_tokens=30, depth=3, top_k=8, threshold=1.0, num_q=2):
        self.config, self.w = config, weights
        self.top_k = top_k
        self._total_tokens = total_tokens - 1  # cnets_ours.py:733
        self.depth = depth
        self.threshold = float(np.log(threshold))  # stored, unused by the reference too (:735)
        self.num_q = num_q
        self.engine: Optional[Engine] = None
        self.stable_kv = None
        self.tree_mask = None
        self.embed_tokens = type("E", (), {"weight": weights.t["embed"]})()

    @property
    def total_tokens(self):
        return self._total_tokens

    @total_tokens.setter
    def total_tokens(self, v):
        """`model.spec_layer.total_tokens = total_token - 1` (spec_model_ours.py:201) resizes the tree of later rounds."""
        self._total_tokens = int(v)
        if self.engine is not None:
            self.engine.set_total_token(int(v) + 1)

    # cnets_ours.py:764-779 — the reference registers eye(k) / zeros(k) buffers; here the equivalents live in the ctx
    def init_tree(self):
        return None

    def reset(self):
        self.tree_mask = None

    def reset_kv(self):
        self.stable_kv = None

    @property
    def last_img_hidden(self):
        """global image feature g [1, D] (cnets_ours.py:930)"""
        D = self.config.hidden_size
        return self.engine.buffer("draft_g", (1, D)).clone()

    def _tree_tuple(self, device):
        """(draft_tokens [1,T] dev, retrieve_indices [n_leaf, max_depth] cpu, tree_mask [1,1,T,T] f32 cpu,
        tree_position_ids [T] dev) — the types/devices of cnets_ours.py:1238."""
        tok, pos, mask, ret = self.engine.tree()
        return (torch.from_numpy(tok)[None].to(device), torch.from_numpy(ret), torch.from_numpy(mask.astype(np.float32))[None, None],
                torch.from_numpy(pos).to(device))

    @torch.no_grad()
    def topK_genrate(self, hidden_states, input_ids, head, logits_processor, inputs_embeds=None, embed_weights=None,
                     image_mask=None):
        """cnets_ours.py:1043-1238.  `head` must be the target's lm_head (it is what the ctx streams)."""
        eng = self.engine
        # logits_processor only changes the row order of retrieve_indices here (cnets_ours.py:1215-1224); the ctx applies it when
        # sampling is enabled (Engine.set_sampling)
        if head.weight.data_ptr() != eng.tw.lm_head.data_ptr():
            raise ValueError("head must be base_model.lm_head")
        if inputs_embeds is not None and inputs_embeds.shape[-2] >= input_ids.shape[-1]:
            raise ValueError("inputs_embeds length must be less than input_ids length")  # :1068-1071
        self.reset()
        if self.stable_kv is None:
            hs = hidden_states.reshape(-1, hidden_states.shape[-1]).to(torch.bfloat16).contiguous()
            if inputs_embeds is None:
                emb = torch.nn.functional.embedding(input_ids.reshape(-1)[:-1].to(eng.device), self.w.t["embed"])
                # un-shifted convention of the C-ABI: row i is the embedding of token i (row 0 is never read)
            else:
                emb = inputs_embeds.reshape(-1, inputs_embeds.shape[-1]).to(torch.bfloat16).contiguous()
            first = input_ids.reshape(-1)[-1:].to(device=eng.device, dtype=torch.int32)
            m = None if image_mask is None else image_mask.reshape(-1).cpu().numpy()
            eng.draft_prefill(hs, emb.contiguous(), m, first)
        else:
            eng.draft_round()
        self.stable_kv = ("device", eng)
        return self._tree_tuple(hidden_states.device)
