"""Python mirror of the reference's object API for the draft-and-verify path:
SpecModel.from_pretrained / forward / specgenerate, spec_layer.topK_genrate / reset_kv / init_tree,
KVCache / initialize_past_key_values, initialize_tree / tree_decoding / evaluate_posterior /
update_inference_inputs.  Everything below the API is libvispec_hip (HIP kernels); there is no eager fallback."""
from .kv_cache import KVCache, initialize_past_key_values  # noqa: F401
from .spec_model_ours import SpecModel  # noqa: F401
