"""Loop helpers with the names and argument meaning of reference vispec/model/utils.py, implemented over the
device-resident round state of libvispec_hip.  The fused fast path used by SpecModel.specgenerate is
`Engine.verify_accept()` + `Engine.draft_round()`; the functions here expose the same stages one at a time so code
written against the reference (and the parity tests) can drive them individually."""
from __future__ import annotations

import numpy as np
import torch


class LogitsProcessorConfig:
    """What utils.py:39-55 builds as a LogitsProcessorList, as plain numbers for the device-side sampler: temperature, then
    TopK when top_k > 0.  `seed` selects the counter-based random stream (the reference draws from torch's global generator)."""

    def __init__(self, temperature, top_k, seed=0):
        self.temperature, self.top_k, self.seed = float(temperature), int(top_k), int(seed)


def prepare_logits_processor(temperature=0.0, repetition_penalty=0.0, top_p=0.0, top_k=0, seed=0):
    """utils.py:39-55.  The reference calls it only for temperature > 1e-5 and passes None (greedy) otherwise."""
    if repetition_penalty > 1.0:
        raise NotImplementedError("RepetitionPenaltyLogitsProcessor needs input_ids; the reference calls its list with None (utils.py:286,454)")
    if 1e-8 <= top_p < 1.0:
        raise NotImplementedError("TopPLogitsWarper cannot run on the 3-D tree logits (HF scatters along dim 1; the reference raises too)")
    return LogitsProcessorConfig(temperature if temperature > 1e-5 else 0.0, top_k if temperature > 1e-5 else 0, seed)


def _enable(model, logits_processor):
    eng = model.engine
    if logits_processor is None or logits_processor.temperature <= 1e-5:
        eng.set_sampling(0.0, 0)
        return False
    eng.set_sampling(logits_processor.temperature, logits_processor.seed, top_k=logits_processor.top_k)
    return True


def reset_tree_mode(model):  # utils.py:330-338
    model.base_model.tree_mask = None
    model.base_model.tree_mode = None


def reset_past_key_values(passed_key_values):  # utils.py:341-358
    for i in range(len(passed_key_values)):
        for j in range(2):
            passed_key_values[i][j].current_length.fill_(0)
    return passed_key_values


def initialize_tree(input_ids, model, past_key_values, logits_processor, inputs_embeds=None, embed_weights=None,
                    image_mask=None, **kwargs):
    """utils.py:266-327: target prefill -> first token -> first topK_genrate.  For Qwen2.5-VL pass `image_grid_thw` / `video_grid_thw` /
    `second_per_grid_ts` (the processor outputs the reference hands to its prefill as kwargs): the multimodal rotary positions and rope_deltas are built from it as in
    SpecModel.specgenerate."""
    sampling = _enable(model, logits_processor)
    position_ids, rope_delta = None, 0
    if model.base_model.config.architectures[0] == "Qwen2_5_VLForConditionalGeneration":
        pos3, rope_delta = model._qwen_rope(input_ids, kwargs.get("image_grid_thw"), video_grid=kwargs.get("video_grid_thw"),
                                            second_per_grid_ts=kwargs.get("second_per_grid_ts"))
        position_ids = pos3[:, None, :]
    outputs, orig, hidden_states = model(input_ids, past_key_values=past_key_values, output_orig=True, inputs_embeds=inputs_embeds,
                                         position_ids=position_ids)
    # argmax(orig[:, -1]) on the device, first max wins (:290) / multinomial(softmax(lp(orig[:, -1]))) (:284-288)
    token = model.engine.sample_row(orig.reshape(-1, orig.shape[-1])[-1]) if sampling else model._first_token(orig)
    # the device-side round state starts here (the reference keeps it in Python locals): prompt ids, n = L, new_token = 0
    model.engine.begin_request(input_ids[0].cpu().numpy(), int(kwargs.get("max_new_tokens", 1 << 20)))
    model._rope_delta = int(rope_delta)
    if rope_delta:
        model.engine.set_rope_delta(rope_delta)
    input_ids = torch.cat((input_ids, token.to(input_ids.device).long()[None]), dim=1)
    embeds = inputs_embeds if inputs_embeds is not None else model._last_embeds
    draft_tokens, retrieve_indices, tree_mask, tree_position_ids = model.spec_layer.topK_genrate(
        hidden_states, input_ids, model.base_model.lm_head, logits_processor, inputs_embeds=embeds, image_mask=image_mask)
    return draft_tokens, retrieve_indices, tree_mask, tree_position_ids, orig, hidden_states, token.long()[None]


def tree_decoding(model, tree_candidates, past_key_values, tree_position_ids, input_ids, retrieve_indices):
    """utils.py:389-412, statement for statement: positions = tree depth + context length (x3 + rope_deltas for Qwen2.5-VL), the
    target forward of the T tree tokens through `model(...)` (SpecModel.forward's verify form -> vispec_target_forward) with the
    tree mask installed on the target, logits gathered along the candidate paths.
    Returns (logits [n_leaf, m, V] fp32, hidden_state_new [1,T,D], None)."""
    if model.base_model.tree_mask is None:
        raise ValueError("tree_mask must be installed on the target before tree_decoding (spec_model_ours.py:486-489)")
    position_ids = tree_position_ids + input_ids.shape[1]
    if model.base_model.config.architectures[0] == "Qwen2_5_VLForConditionalGeneration":
        position_ids = position_ids.unsqueeze(0) + model.base_model.rope_deltas.to(position_ids.device)
        position_ids = position_ids.unsqueeze(0).expand(3, -1, -1)
    outputs, tree_logits, hidden_state = model(tree_candidates, output_orig=True, past_key_values=past_key_values,
                                               position_ids=position_ids)
    model.engine.set_retrieve(retrieve_indices.cpu().numpy())  # what vispec_accept walks (evaluate_posterior / update_inference_inputs)
    logits = tree_logits[0, retrieve_indices.to(tree_logits.device)]
    return logits, hidden_state, outputs


def evaluate_posterior(logits, candidates, logits_processor, model=None):
    """utils.py:415-493.  Greedy: a pure function of its arguments (torch integer ops; the fused loop uses the HIP verify_accept
    kernel instead).  Sampling: the sequential rejection runs on the device over the logits of the last tree_decoding — pass
    `model`; the accept step is then already done when update_inference_inputs is called (sample_p stays on the device: None)."""
    if logits_processor is not None and logits_processor.temperature > 1e-5:
        if model is None:
            raise ValueError("the sampling branch of evaluate_posterior runs on the device: pass model=")
        eng = model.engine
        eng.accept()
        model._accept_done = True
        best, acc = eng.last_accept()
        return torch.tensor(best), acc, None
    posterior_mask = (candidates[:, 1:].to(logits.device) == torch.argmax(logits[:, :-1], dim=-1)).int()
    cal = torch.cumprod(posterior_mask, dim=1).sum(dim=1)
    accept_length = cal.max()
    best = torch.tensor(0, dtype=torch.long, device=candidates.device) if accept_length == 0 else torch.argmax(cal).to(torch.long)
    return best, accept_length, logits[best, accept_length]


@torch.no_grad()
def update_inference_inputs(input_ids, candidates, best_candidate, accept_length, retrieve_indices, logits_processor,
                            new_token, past_key_values_data_list, current_length_data, model, hidden_state_new, sample_p):
    """utils.py:496-593: commit the accepted path (tokens, KV compaction, lengths), sample the next token and run the
    next topK_genrate — on the device (vispec_accept + vispec_draft_round)."""
    eng = model.engine
    if not getattr(model, "_accept_done", False):
        eng.set_retrieve(retrieve_indices.cpu().numpy())
        eng.accept()
    model._accept_done = False
    eng.draft_round()
    st = eng.state()
    a = int(st["accept_len"])
    if a != int(accept_length):
        raise RuntimeError("device accept length differs from the caller's evaluate_posterior")
    input_ids = torch.cat([input_ids, candidates[None, int(best_candidate), : a + 1].to(input_ids.device)], dim=-1)
    current_length_data.fill_(st["n_ctx"])
    token = torch.tensor([[st["next_token"]]], dtype=torch.long, device=input_ids.device)
    draft_tokens, retrieve_indices, tree_mask, tree_position_ids = model.spec_layer._tree_tuple(input_ids.device)
    new_token += a + 1
    return input_ids, draft_tokens, retrieve_indices, tree_mask, tree_position_ids, new_token, None, token
