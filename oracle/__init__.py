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
  * G (generate loop): restated from modeling_vibevoice_inference.py:326-710.
    The reference's generate() cannot execute under transformers 5.x
     so the loop orchestration itself is "parity unpinned";
    every arithmetic stage it calls is pinned as above.
"""
