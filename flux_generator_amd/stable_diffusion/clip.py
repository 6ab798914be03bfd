"""CLIP text encoders of the stable_diffusion/ pipelines (mirror of the reference's
stable_diffusion/stable_diffusion/clip.py): the HIP CLIPTextModel of flux/clip.py with the SD-side options —
`projection_dim` (text_projection on the pooled EOS row, "...WithProjection" checkpoints: SDXL's second encoder),
`hidden_act` "gelu" (exact erf: the OpenCLIP towers of SD 2.1 / SDXL) or "quick_gelu", and the full
`hidden_states` list (SDXL conditions on hidden_states[-2] of both encoders, __init__.py:206-229)."""
from ..flux.clip import CLIPOutput, CLIPTextModel, CLIPTextModelConfig  # noqa: F401


def map_clip_text_encoder_weights(key, value):
    """HF CLIPTextModel names -> module tree (stable_diffusion/.../model_io.py:98-123)."""
    for pre in ("text_model.", "embeddings.", "encoder."):
        if key.startswith(pre):
            key = key[len(pre):]
    for a, b in (("self_attn.", "attention."), ("q_proj.", "query_proj."), ("k_proj.", "key_proj."),
                 ("v_proj.", "value_proj."), ("mlp.fc1", "linear1"), ("mlp.fc2", "linear2")):
        if a in key:
            key = key.replace(a, b)
    return [(key, value)]
