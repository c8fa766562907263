"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32) restatement of the VibeVoice inference hot path that
SURVEY.md section 8 scopes: diffusion head (D), DPM-Solver++ sampler (S),
streaming acoustic decoder (A), streaming semantic encoder (E), speech
connectors (C), Qwen2 decode step (L/N) and the generate() loop (G).

Nothing under vibevoice_amd/ may import this package.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
the checker -- never as the thing measured or shipped.

Pinning status ("how do we know the oracle is the reference?"):
  * D, S, A, E, C: pinned.  tests/golden/*.npz were generated in the build
    container by tests/golden/make_golden.py, which imports the reference's
    own classes from /root/reference (through the two import shims in
    oracle/refshim.py) and records their outputs on seeded inputs;
    tests/test_oracle_golden.py checks every oracle function against them.
  * L/N (Qwen2 decode): the arithmetic lives in a third-party dependency
    (transformers==4.51.3 models/qwen2/modeling_qwen2.py, pinned by the
    reference's pyproject.toml:22, NOT vendored under /root/reference).  The
    golden vectors come from the installed transformers 5.15 Qwen2Model (same
    math for dense Qwen2), again via make_golden.py.
  * G (generate loop): PINNED.  oracle/refshim.install_generate_shims() adapts the handful of transformers
    signatures that drifted between 4.51.3 (pinned by the reference) and the installed 5.15 -- and restores
    4.51.3's attention-mask-derived position_ids in prepare_inputs_for_generation -- so the reference's own
    VibeVoiceForConditionalGenerationInference.generate() runs here on a tiny seeded model.  make_golden.py
    records it (forced token plans through a LogitsProcessor, every torch.randn draw captured):
    generate_forced_b1 / generate_forced_b2 (desynchronised batch) / generate_greedy_b1 / generate_cap_b1 (length cap) /
    generate_ragged_voice_b1 (voice sample of 2.5 frames) / generate_sampled_b1
    (do_sample=True: pins the order and shapes in which the torch generator is consumed).  The oracle loop,
    fed the same inputs and noise, reproduces token sequences exactly and waveforms to rel-L2 <= 1e-4.
    (This pinning found and fixed a real deviation: after <speech_start> the reference's negative context
    restarts EMPTY, not with one kept entry.)
  * Z (Streaming-0.5B loop): PINNED the same way (refshim.install_streaming_shims): the reference's
    VibeVoiceStreamingForConditionalGenerationInference.generate() runs on a tiny seeded split model from
    prefilled branches made with its own forward_lm / forward_tts_lm; streaming_text12_cap40 /
    streaming_text3_cap20 record caches, noise and outputs; oracle/generate_streaming.py reproduces token
    count, stop reason and waveform (rel-L2 <= 1e-4).
"""
