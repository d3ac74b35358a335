"""The layer-streamed evaluation of the oracle (oracle/streamed.py) equals the oracle's own one-graph step
(dpo_oracle.dpo_train_step): same forward, same gradients, also with several reference-log-prob variants off one forward.
CPU only."""
import torch

from oracle import dpo_oracle as O
from oracle import streamed as S


def _run_full(batch, W, cfg):
    W = {k: v.clone() for k, v in W.items()}
    out, grads, gn = O.dpo_train_step(batch, W, cfg, {}, lr=0.0, step=1, sft_weight=0.0, dpo_weight=1.0)
    return out, grads, gn


def test_streamed_equals_one_graph_step():
    cfg = O.tiny_cfg()
    cfg.layers = 3
    W = O.make_weights(cfg, seed=5)
    batch = O.make_synthetic_batch(cfg, 3, 48, prompt_len=12, seed=9, image_pos=5)
    out, grads, _ = _run_full(batch, W, cfg)
    got = {}
    res = S.dpo_step_streamed(batch, W, cfg, grad_sink=lambda v, n, g: got.__setitem__(n, g.clone()))
    assert torch.equal(res["labels"], out["labels"])
    assert torch.allclose(res["log_prob"], out["log_prob"].detach(), rtol=1e-6, atol=1e-5)
    assert abs(float(res["loss"]) - float(out["loss"])) <= 1e-6 * abs(float(out["loss"]))
    assert set(got) == set(grads)
    for k, g in grads.items():
        assert torch.allclose(got[k], g, rtol=1e-4, atol=1e-6 * float(g.abs().max()) + 1e-12), k


def test_streamed_variants_share_one_forward():
    cfg = O.tiny_cfg()
    W = O.make_weights(cfg, seed=6)
    batch = O.make_synthetic_batch(cfg, 2, 40, prompt_len=12, seed=3, image_pos=5)
    fwd = S.dpo_step_streamed(batch, W, cfg, backward=False)
    pw, pr = fwd["policy_win_logp"], fwd["policy_rej_logp"]
    variants = [dict(ref_win_logp=batch["ref_win_logp"], ref_rej_logp=batch["ref_rej_logp"]),
                dict(ref_win_logp=pw.clone(), ref_rej_logp=pr - torch.tensor([0.0, 5.0]))]      # beta z = 0, 0.5
    got = [{}, {}]
    res = S.dpo_step_streamed(batch, W, cfg, variants=variants, grad_sink=lambda v, n, g: got[v].__setitem__(n, g.clone()))
    assert abs(float(res["variants"][1]["losses"][0]) - 0.6931472) < 1e-5
    for v, var in enumerate(variants):
        b = dict(batch)
        b.update(var)
        _, grads, _ = _run_full(b, W, cfg)
        for k, g in grads.items():
            assert torch.allclose(got[v][k], g, rtol=1e-4, atol=1e-6 * float(g.abs().max()) + 1e-12), (v, k)
