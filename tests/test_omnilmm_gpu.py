"""OmniLMM branch on the GPU against the fixture the REFERENCE'S OWN OmniLMMForCausalLM /
Resampler / forward_DPO produced (tests/golden/omnilmm_tiny.pt, tests/golden/make_omnilmm_golden.py):

  * the Resampler on the HIP kernels, forward and every parameter gradient, both position-table branches;
  * the whole DPO step from tower tokens on: replacement splice (bit exact indexing), Mistral-style GQA decoder, fused
    LM-head log-probs, DPO loss, and the gradients of the language model AND the resampler; reference and packed layouts.

Bars: sequence log-probs / loss 1e-3 relative; gradients per-tensor norm within 3 %, direction cosine >= 0.995."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpo_oracle as O  # noqa: E402
from oracle import omnilmm_oracle as OO  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "omnilmm_tiny.pt")
BF16 = torch.bfloat16


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _cfg(meta):
    from rlaif_v_amd.omnilmm import OmniLMMConfig
    pt, st, en = meta["tokens"]
    return OmniLMMConfig(hidden=512, layers=2, heads=4, kv_heads=2, ffn=768, vocab=325, model_max_length=256,
                         num_query=meta["num_query"], vision_width=meta["kv_dim"], image_size=84,
                         im_patch_token=pt, im_start_token=st, im_end_token=en)


def _weights(meta):
    ocfg = O.LlavaCfg(hidden=512, layers=2, heads=4, kv_heads=2, ffn=768, vocab=325, model_max_length=256)
    W = {k: v for k, v in O.make_weights(ocfg, seed=3).items() if "vision_tower" not in k and "mm_projector" not in k}
    W.update(OO.make_resampler_weights(512, meta["kv_dim"], meta["num_query"]))
    return W


def _check_grads(gold, mine, what):
    worst = 0.0
    for k, g in gold.items():
        m = mine[k].float().cpu()
        if "full" in g:
            ref = g["full"]
            if float(ref.norm()) < 1e-7:
                continue
            c = _cos(m, ref)
            rel = abs(float(m.norm()) - float(ref.norm())) / float(ref.norm())
        else:
            blk = m.reshape(m.shape[0], -1)[:8, :64]
            c = _cos(blk, g["block"])
            rel = abs(float(m.double().norm()) - g["norm"]) / g["norm"]
        worst = max(worst, rel)
        assert c >= 0.995, (what, k, c)
        assert rel <= 3e-2, (what, k, rel)
    print(f"[{what}] worst per-tensor grad-norm rel err {worst:.3e}")


@pytest.mark.parametrize("n_tok", [36, 16])
def test_resampler_matches_reference_golden(n_tok):
    _need_gpu()
    from rlaif_v_amd.resampler import Resampler, param_shapes
    G = torch.load(GOLD, weights_only=False)
    meta, case = G["meta"], G[f"resampler_{n_tok}"]
    W = _weights(meta)
    dev = torch.device("cuda:0")
    rs = Resampler(512, meta["kv_dim"], meta["num_query"], dev)
    P = {k: W[OO.RS + k].to(dev, BF16).contiguous() for k in param_shapes(512, meta["kv_dim"], meta["num_query"])}
    Gd = {k: torch.zeros_like(v) for k, v in P.items()}
    B, N, kv = case["x"].shape
    ctx = {}
    z = rs.forward(case["x"].to(dev, BF16).reshape(B * N, kv).contiguous(), B, N, P.__getitem__, ctx)
    y = z.float().cpu().view(B, meta["num_query"], 512)
    err = float((y - case["y"]).abs().max()) / float(case["y"].abs().max())
    print(f"resampler n_tok={n_tok}: forward max err / max |y| = {err:.3e}")
    assert err <= 2e-2 and _cos(y, case["y"]) >= 0.9995
    rs.backward(case["gy"].to(dev, BF16).reshape(B * meta["num_query"], 512).contiguous(), ctx, P.__getitem__, Gd.__getitem__)
    _check_grads(case["grads"], Gd, f"resampler {n_tok}")


@pytest.mark.parametrize("share_prefix", [False, True])
def test_dpo_step_matches_reference_golden(share_prefix, monkeypatch):
    _need_gpu()
    from rlaif_v_amd.omnilmm import OmniLMMDPOModel
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    monkeypatch.setenv("SFT_weight", "0.0")
    monkeypatch.setenv("DPO_weight", "1.0")
    G = torch.load(GOLD, weights_only=False)
    meta, case = G["meta"], G["dpo"]
    model = OmniLMMDPOModel(_cfg(meta))
    model.share_prefix = share_prefix
    model.load_state_dict(_weights(meta))
    tr = LLaVA15DPOTrainer(model=model, args=TrainingArguments())
    batch = dict(case["batch"])
    batch["images"] = case["tower_features"]                   # precomputed tower tokens [B, N, width]
    loss = tr.compute_loss(model, batch)
    out = model.last_out
    # ---- indexing: bit exact (labels are NOT changed by the OmniLMM splice)
    labels = case["batch"]["concatenated_labels"]
    mask = labels[:, 1:] != -100
    s_idx, l_idx = torch.nonzero(mask, as_tuple=True)
    assert torch.equal(out.plan.tgt.cpu().long(), labels[:, 1:][mask])
    assert torch.equal(out.plan.seq_of_row.cpu().long(), s_idx)
    if not share_prefix:
        assert torch.equal(out.plan.labels.cpu(), labels)
        assert torch.equal(out.plan.sel_idx.cpu().long(), s_idx * labels.shape[1] + l_idx)
        st = int(torch.where(case["batch"]["concatenated_input_ids"][0] == meta["tokens"][1])[0][0])
        src = out.plan.src.cpu().view(out.plan.S, out.plan.L)
        assert src[0, st] == meta["tokens"][1] and src[0, st + 1] == -2 and src[0, st + meta["num_query"]] == -2 - (meta["num_query"] - 1)
        assert src[2, st + 1] == -2                              # rejected row of pair 0 reads the same image
    else:
        assert out.plan.S == 2 and max(out.plan.shared_len) > meta["num_query"]       # the image lies in the shared prefix
    ref_lp = case["logp"]
    err_lp = (out.seq_logp.cpu() - ref_lp).abs()
    print(f"omnilmm share_prefix={share_prefix}: seq logp err {err_lp.tolist()} of {ref_lp.tolist()}; loss {float(loss):.6f} "
          f"vs {float(case['loss']):.6f}")
    assert bool((err_lp <= 1e-3 * ref_lp.abs()).all())
    torch.testing.assert_close(loss.cpu(), case["loss"], rtol=1e-3, atol=0.0)
    torch.testing.assert_close(out.per_pair[1].cpu(), case["chosen_rewards"], rtol=2e-3, atol=1e-2)
    # ---- backward: language model + resampler gradients
    model.backward(out, model.last_coef)
    _check_grads(case["grads"], model.grads_state_dict(), f"dpo share_prefix={share_prefix}")


def test_pixel_input_without_tower_fails_loudly():
    _need_gpu()
    from rlaif_v_amd.omnilmm import OmniLMMDPOModel
    G = torch.load(GOLD, weights_only=False)
    model = OmniLMMDPOModel(_cfg(G["meta"]), with_optimizer=False)
    with pytest.raises(NotImplementedError):
        model.encode_images(torch.zeros(1, 3, 84, 84))


def test_eva_tower_matches_its_restatement_and_feeds_the_model():
    """EVA02 tower: PARITY UNPINNED (see rlaif-v_amd/eva_tower.py) - the HIP tower against the author's own torch restatement
    (head dim 112 on the 128-wide kernels, post-norm blocks, resampled positions), then pixels -> tower -> resampler -> DPO
    forward through ``set_vision_tower`` equals the same forward fed with the tower tokens."""
    _need_gpu()
    from rlaif_v_amd.eva_tower import EvaConfig, EvaTower
    from rlaif_v_amd.omnilmm import OmniLMMDPOModel
    G = torch.load(GOLD, weights_only=False)
    meta, case = G["meta"], G["dpo"]
    ec = EvaConfig(width=448, depth=4, heads=4, mlp=768, patch=14, pretrain_grid=3)     # head dim 112, as EVA02-E
    We = OO.make_eva_weights(ec.width, ec.depth, ec.heads, ec.mlp, ec.patch, ec.pretrain_grid)
    tower = EvaTower(ec)
    tower.load_state_dict(We)
    px = torch.randn(2, 3, 84, 84, generator=torch.Generator().manual_seed(2))        # 6 x 6 grid: positions are resampled from 3 x 3
    ref = OO.eva_forward_features(px, We, ec.heads, ec.patch, ec.pretrain_grid, ec.blocks_used)
    got = tower(px).float().cpu()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print(f"eva tower: max err / max |ref| = {err:.3e}, cosine {_cos(got, ref):.6f}")
    assert got.shape == ref.shape == (2, 36, 448)
    assert err <= 3e-2 and _cos(got, ref) >= 0.9995
    # pixels in, through the model
    cfg = _cfg(meta)
    cfg.vision_width = 448
    model = OmniLMMDPOModel(cfg, with_optimizer=False)
    W = _weights(meta)
    W.update(OO.make_resampler_weights(512, 448, meta["num_query"]))
    model.load_state_dict(W)
    model.set_vision_tower(tower)
    b = case["batch"]
    out_px = model.forward_logps(b["concatenated_input_ids"], b["concatenated_labels"], px, save_for_backward=False)
    out_tok = model.forward_logps(b["concatenated_input_ids"], b["concatenated_labels"], tower(px), save_for_backward=False)
    assert torch.equal(out_px.seq_logp, out_tok.seq_logp)
