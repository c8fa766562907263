"""Architecture facts of the shipped VibeVoice models, in the reference's own
config schema (vibevoice/configs/qwen2.5_1.5b_64k.json, qwen2.5_7b_32k.json;
Streaming-0.5B dims from SURVEY.md 8 -- the reference ships no JSON for it).
Used for synthetic-weight benchmarks when no checkpoint directory is given."""
import copy

_TOKENIZER = {
    "causal": True, "channels": 1, "conv_bias": True, "conv_norm": "none", "disable_last_norm": True,
    "encoder_depths": "3-3-3-3-3-3-8", "encoder_n_filters": 32, "encoder_ratios": [8, 5, 5, 4, 2, 2],
    "layer_scale_init_value": 1e-06, "layernorm": "RMSNorm", "layernorm_elementwise_affine": True,
    "layernorm_eps": 1e-05, "mixer_layer": "depthwise_conv", "pad_mode": "constant", "weight_init_value": 0.01,
}

_BASE = {
    "acoustic_vae_dim": 64,
    "semantic_vae_dim": 128,
    "acoustic_tokenizer_config": dict(_TOKENIZER, decoder_depths=None, decoder_n_filters=32,
                                      decoder_ratios=[8, 5, 5, 4, 2, 2], fix_std=0.5, std_dist_type="gaussian",
                                      vae_dim=64),
    "semantic_tokenizer_config": dict(_TOKENIZER, fix_std=0, std_dist_type="none", vae_dim=128),
    "diffusion_head_config": {
        "ddpm_batch_mul": 4, "ddpm_beta_schedule": "cosine", "ddpm_num_inference_steps": 20,
        "ddpm_num_steps": 1000, "diffusion_type": "ddpm", "head_ffn_ratio": 3.0, "head_layers": 4,
        "latent_size": 64, "prediction_type": "v_prediction", "rms_norm_eps": 1e-05, "speech_vae_dim": 64,
    },
    "decoder_config": {
        "hidden_act": "silu", "initializer_range": 0.02, "model_type": "qwen2", "num_hidden_layers": 28,
        "rms_norm_eps": 1e-06, "rope_theta": 1000000.0,
    },
}


def _mk(hidden, inter, heads, kv, max_pos, vocab, tie, layers=28):
    c = copy.deepcopy(_BASE)
    c["decoder_config"].update(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads,
                               num_key_value_heads=kv, max_position_embeddings=max_pos, vocab_size=vocab,
                               tie_word_embeddings=tie, num_hidden_layers=layers)
    c["diffusion_head_config"]["hidden_size"] = hidden
    return c


CONFIGS = {
    "1.5b": _mk(1536, 8960, 12, 2, 65536, 151936, True),
    "7b": _mk(3584, 18944, 28, 4, 32768, 152064, False),
}

# VibeVoice-Streaming-0.5B: hidden/heads/kv/head_dim from the shipped voice presets (lm KV
# [1,2,74,64] x 4 layers, tts_lm [1,2,251,64] x 20), intermediate/vocab from the public Qwen2.5-0.5B config
# (the reference ships no JSON for it); 8K context (README.md:53).
_s = _mk(896, 4864, 14, 2, 8192, 151936, True, layers=24)
_s["tts_backbone_num_hidden_layers"] = 20
_s["semantic_tokenizer_config"] = None
CONFIGS["0.5b-streaming"] = _s
