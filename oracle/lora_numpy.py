"""Second, independently written restatement of the LoRA adapter arithmetic - TEST INFRASTRUCTURE ONLY.

peft (pinned by the reference at 0.10.0, pyproject.toml:16-23; configured at muffin/train/train_llava15_lora.py:304-318) is
absent offline, so oracle/dpo_oracle.py::lora_linear cannot be pinned against it ("parity unpinned").  This file restates the
same layer from the LoRA paper's equations (Hu et al. 2021, section 4.1: h = W0 x + (alpha / r) B A x, A ~ random, B = 0 at
init, merge W = W0 + (alpha / r) B A) in float64 numpy with hand-derived gradients - no torch, no autograd, written without
looking at lora_linear - so the torch oracle is at least cross-checked by something other than itself
(tests/test_lora_oracle.py::test_lora_oracle_matches_numpy_restatement).

Shapes: x [n, d_in], W0 [d_out, d_in], A [r, d_in], B [d_out, r]; ``mask`` [n, d_in] is the dropout multiplier keep / (1 - p)
applied to the ADAPTER input only (peft: result = base(x) + lora_B(lora_A(dropout(x))) * scaling).
"""
import numpy as np


def forward(x, W0, A, B, alpha, r, mask=None):
    x = np.asarray(x, np.float64)
    xa = x if mask is None else x * np.asarray(mask, np.float64)
    low = xa @ np.asarray(A, np.float64).T                      # [n, r]
    return x @ np.asarray(W0, np.float64).T + (alpha / r) * (low @ np.asarray(B, np.float64).T)


def backward(x, W0, A, B, alpha, r, dh, mask=None):
    """Gradients of sum(h * dh) with respect to x, A, B (W0 frozen)."""
    x, W0, A, B, dh = (np.asarray(t, np.float64) for t in (x, W0, A, B, dh))
    s = alpha / r
    m = 1.0 if mask is None else np.asarray(mask, np.float64)
    xa = x * m
    low = xa @ A.T                                              # [n, r]
    dlow = s * (dh @ B)                                         # [n, r]
    dB = s * (dh.T @ low)                                       # [d_out, r]
    dA = dlow.T @ xa                                            # [r, d_in]
    dx = dh @ W0 + (dlow @ A) * m                               # base path + adapter path (through the mask)
    return dx, dA, dB


def merged_weight(W0, A, B, alpha, r):
    return np.asarray(W0, np.float64) + (alpha / r) * (np.asarray(B, np.float64) @ np.asarray(A, np.float64))
