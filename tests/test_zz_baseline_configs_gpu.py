"""Numeric parity AT the BASELINE.json configurations (VERDICT r1 "what's missing" #4):

  * config 1 shape - 4 synthetic 336-px pairs, text length T = 512 -> spliced length L = 1087, full 7B widths, CLIP-L/14-336 at
    full depth, 4 language-model layers: HIP forward + backward vs the fp32 CPU oracle run on the GPU box's host cores;
  * FULL DEPTH (32 layers): config 1's whole optimisation step (forward, backward, clip, AdamW) and config 2's sequence shape
    (L = 2048, forward) against what the fp32 oracle produced for the same seeded weights and batch on the GPU box's host
    (tests/full_depth.py, tools/full_depth_parity.py -> tests/golden/fulldepth_*.pt; RV_PARITY_LIVE=1 re-runs the oracle);
  * the full-width golden produced by the REFERENCE ITSELF (tests/golden/fullwidth_l2_b2.pt): HIP forward + backward.

Bars (north_star): token indexing bit exact; sequence log-prob sums and the DPO loss within 1e-3 RELATIVE, at every depth;
per-token log-probs (bf16 activations through the whole stack against an fp32 oracle): at full depth CALIBRATED - no further
from the fp32 oracle than the oracle evaluated the way HF runs under --bf16 (oracle.emulate_bf16) - and at 4 layers mean |err|
within 5e-3 of the mean |log-prob|, worst token within 2e-2; gradients: per-tensor norm within 3 %, direction cosine >= 0.99.
(The file sorts last on purpose: these cases spend minutes in the CPU oracle.)  The measured numbers are written to
gpurun_out/parity_<round>.json (RV_ROUND, default r06; copied to profiles/).
"""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpo_oracle as O  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # tests/full_depth.py (harness shared with tools/)


def _need_big_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2**30:
        pytest.skip("needs the 288 GB part")


def _host_ram_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 2**20
    except OSError:
        pass
    return 0.0


def _record(key, value):
    path = os.path.join(REPO, "gpurun_out", f"parity_{os.environ.get('RV_ROUND', 'r06')}.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    blob = {}
    if os.path.exists(path):
        try:
            blob = json.load(open(path))
        except ValueError:
            blob = {}
    blob[key] = value
    json.dump(blob, open(path, "w"), indent=1)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _model(cfg, W):
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    model = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)), with_optimizer=False)
    model.load_state_dict(W)
    return model


def _trainer(model):
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    return LLaVA15DPOTrainer(model=model, args=TrainingArguments())


def _check_forward(out, loss, ref, tag, loss_rtol=1e-3):
    lp, lp_ref = out.seq_logp.cpu(), ref["log_prob"].detach()
    rel = ((lp - lp_ref).abs() / lp_ref.abs()).max().item()
    mask = ref["labels"][:, 1:] != -100
    tok_ref = ref["per_token_logps"].detach()[mask]
    tok_d = (out.per_token_logp.cpu() - tok_ref).abs()
    tok_err, tok_mean, tok_mag = tok_d.max().item(), tok_d.mean().item(), tok_ref.abs().mean().item()
    loss_rel = abs(float(loss) - float(ref["loss"].detach())) / abs(float(ref["loss"].detach()))
    print(f"[{tag}] seq log-prob {lp.tolist()} vs oracle {lp_ref.tolist()}: max rel err {rel:.2e}; per-token err max "
          f"{tok_err:.2e} mean {tok_mean:.2e} (mean |log-prob| {tok_mag:.2f}, {tok_ref.numel()} tokens); loss {float(loss):.6f} vs "
          f"{float(ref['loss']):.6f} (rel {loss_rel:.2e})")
    assert torch.equal(out.plan.tgt.cpu().long(), ref["labels"][:, 1:][mask])          # token indexing: bit exact
    assert out.seq_cnt.cpu().tolist() == mask.sum(1).float().tolist()
    assert rel <= 1e-3 and loss_rel <= loss_rtol
    assert tok_mean <= 5e-3 * tok_mag and tok_err <= 2e-2 * tok_mag
    return dict(seq_logp=lp.tolist(), seq_logp_oracle=lp_ref.tolist(), seq_logp_max_rel_err=rel, per_token_max_abs_err=tok_err,
                per_token_mean_abs_err=tok_mean, per_token_mean_abs_value=tok_mag, n_tokens=int(tok_ref.numel()),
                loss=float(loss), loss_oracle=float(ref["loss"]), loss_rel_err=float(loss_rel))


@pytest.mark.timeout(1500)
def test_config1_shape_vs_oracle():
    """BASELINE config 1's batch (4 pairs, T = 512 -> L = 1087) at 4 layers of full width: forward AND backward."""
    _need_big_gpu()
    cfg = O.LlavaCfg(layers=4, model_max_length=2048)
    W = O.make_weights(cfg, seed=31)
    model = _model(cfg, W)
    tr = _trainer(model)
    batch = O.make_synthetic_batch(cfg, 4, 512, 64, seed=31, ragged=True)
    assert batch["concatenated_input_ids"].shape == (8, 512)
    loss = tr.compute_loss(model, dict(batch))
    out = model.last_out
    model.backward(out, model.last_coef)
    grads = model.grads_state_dict()
    torch.cuda.synchronize()
    torch.set_num_threads(min(128, os.cpu_count() or 8))
    for k in O.trainable_names(cfg):
        W[k].requires_grad_(True)
    t0 = time.time()
    ref = O.dpo_step_forward(batch, W, cfg, sft_weight=0.0, dpo_weight=1.0)
    ref["loss"].backward()
    t_cpu = time.time() - t0
    assert ref["labels"].shape == (8, 1087)
    rec = _check_forward(out, loss, ref, "config-1 shape, 4 layers")
    worst_norm, worst_cos = 0.0, 1.0
    for k in O.trainable_names(cfg):
        g_ref = W[k].grad
        n_ref = float(g_ref.double().norm())
        if n_ref < 1e-9:
            continue
        rel = abs(float(grads[k].double().norm()) - n_ref) / n_ref
        c = _cos(grads[k], g_ref)
        worst_norm, worst_cos = max(worst_norm, rel), min(worst_cos, c)
        assert rel <= 3e-2 and c >= 0.99, (k, rel, c)
    print(f"  backward: worst per-tensor grad-norm rel err {worst_norm:.2e}, worst cosine {worst_cos:.5f}; oracle fwd+bwd {t_cpu:.1f} s")
    rec.update(grad_worst_norm_rel_err=worst_norm, grad_worst_cosine=worst_cos, oracle_fwd_bwd_s=t_cpu, layers=4, pairs=4,
               L=1087, threads=torch.get_num_threads())
    _record("config1_shape_4layers", rec)


@pytest.fixture(scope="module")
def full_depth():
    """All 32 layers of LLaVA-1.5-7B on the GPU (weights seeded like tools/full_depth_parity.py), shared by the two
    full-depth cases below: cfg2_fwd runs first (forward only), cfg1_step last (it moves the weights)."""
    _need_big_gpu()
    import full_depth as FD
    live = os.environ.get("RV_PARITY_LIVE", "0") != "0"
    if live and _host_ram_gb() < 400:
        pytest.skip("the live fp32 oracle of one full 7B training step needs ~350 GB of host RAM")
    if _host_ram_gb() < 48:
        pytest.skip("the seeded fp32 weights of the 7B model (27 GB on the host) do not fit this box")
    torch.cuda.empty_cache()
    cfg = FD.make_cfg(32)
    t0 = time.time()
    W = O.make_weights(cfg, seed=FD.WEIGHT_SEED)
    model, trainer = FD.build_model(cfg, W, with_optimizer=True)
    print(f"  full-depth weights + model: {time.time() - t0:.0f} s")
    # the pristine parameters, taken BEFORE any case runs: every stepping case starts from (and returns to) them
    yield dict(FD=FD, cfg=cfg, W=W, model=model, trainer=trainer, live=live, snap=FD.snapshot(model))
    del model, trainer
    torch.cuda.empty_cache()


def _fixture_or_oracle(fd, case, golden_dir):
    FD = fd["FD"]
    if fd["live"]:
        torch.set_num_threads(min(128, os.cpu_count() or 8))
        return FD.oracle_case(case, fd["W"], fd["cfg"], emulate=True)
    path = os.path.join(golden_dir, f"fulldepth_{case}.pt")
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: generate it on the GPU box with tools/full_depth_parity.py")
    fx = torch.load(path, weights_only=False)
    assert fx["layers"] == 32 and fx["weight_seed"] == FD.WEIGHT_SEED
    return fx


@pytest.mark.timeout(2400)
def test_full_depth_config2_forward(full_depth, golden_dir):
    """BASELINE config 2's sequence at FULL DEPTH (32 layers, L = 2048, one pair): log-prob sums AND the DPO loss within
    1e-3 of the fp32 oracle; per-token log-probs no further from it than the bf16-emulated oracle (tests/full_depth.py)."""
    FD = full_depth["FD"]
    hip = FD.hip_case("cfg2_fwd", full_depth["model"], full_depth["trainer"], full_depth["cfg"])
    assert hip["plan_S"] == 1 and hip["plan_L"] > 2048                    # one packed pair row
    fx = _fixture_or_oracle(full_depth, "cfg2_fwd", golden_dir)
    assert fx["labels"].shape == (2, 2048)
    m = FD.compare("cfg2_fwd", hip, fx, check=True)
    print("  " + json.dumps({k: v for k, v in m.items() if not isinstance(v, (list, dict))}))
    _record("config2_full_depth_forward", m)


def _fixture(FD, case, golden_dir):
    path = os.path.join(golden_dir, f"fulldepth_{case}.pt")
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: generate it with tools/full_depth_oracle_streamed.py (build container, ~20 min per base case)")
    fx = torch.load(path, weights_only=False)
    assert fx["layers"] == 32 and fx["weight_seed"] == FD.WEIGHT_SEED and fx["case"] == case
    return fx


def _stepping_case(fd, golden_dir, case, key):
    """One stepping case (forward, backward, clip, AdamW) from the shared starting weights; the weights are put back after."""
    FD = fd["FD"]
    fx = _fixture(FD, case, golden_dir)
    FD.restore(fd["model"], fd["trainer"], fd["snap"])
    try:
        hip = FD.hip_case(case, fd["model"], fd["trainer"], fd["cfg"], fx=fx)
    finally:
        FD.restore(fd["model"], fd["trainer"], fd["snap"])
    m = FD.compare(case, hip, fx, W0=fd["W"], check=False)
    print("  " + json.dumps({k: v for k, v in m.items() if not isinstance(v, dict)}))
    _record(key, m)
    FD.compare(case, hip, fx, W0=fd["W"], check=True)
    return m, hip, fx


@pytest.mark.timeout(1800)
def test_full_depth_config2_step(full_depth, golden_dir):
    """BASELINE config 2's PACKED shape at FULL DEPTH, forward AND backward (VERDICT r3 missing 3): two pairs at L = 2048 with
    ragged answers -> two packed rows [shared | chosen tail | rejected tail], rejected-tail tile skipping in dQ and dK/dV at
    27 key blocks per row, 32 layers, clip + AdamW - against the fp32 oracle evaluated layer by layer in the build container
    (oracle/streamed.py, tests/golden/fulldepth_cfg2_step.pt).  Same bars as the config-1 step."""
    m, hip, fx = _stepping_case(full_depth, golden_dir, "cfg2_step", "config2_full_depth_step")
    assert fx["labels"].shape == (4, 2048) and hip["plan_S"] == 2 and hip["plan_L"] > 2048


@pytest.mark.timeout(1800)
def test_full_depth_config2_conditioned(full_depth, golden_dir):
    """The same batch in the regime DPO training STARTS in (VERDICT r3 missing 4): reference log-probs = the oracle's own fp32
    policy log-probs, shifted so that beta*z = 0 / +1 for the two pairs - both pairs contribute gradient, the loss is ~ ln 2
    instead of beta x a large difference.  Asserted: the logit error beta*|d((pw - pr))| per pair within 3 sigma of a sum of
    independent per-token errors of the bf16-EMULATED oracle's size, per-token RMS error below the emulation's, the gradient /
    clip / AdamW bars of the saturated case."""
    m, _, _ = _stepping_case(full_depth, golden_dir, "cfg2_cond", "config2_full_depth_conditioned")
    assert all(abs(z - t) < 0.5 for z, t in zip(m["logit"], m["beta_z"]))          # the batch really is in the conditioned regime


@pytest.mark.timeout(1800)
def test_full_depth_config1_conditioned(full_depth, golden_dir):
    """BASELINE config 1's batch with beta*z = -0.5 / 0 / +0.5 / +1 for its four pairs: all four contribute to the gradient
    (the saturated case has two pairs at sigma(-beta z) ~ 1 and two at ~ 0)."""
    m, _, _ = _stepping_case(full_depth, golden_dir, "cfg1_cond", "config1_full_depth_conditioned")
    assert all(abs(z - t) < 0.5 for z, t in zip(m["logit"], m["beta_z"]))


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("case", ["cfg1_step", "cfg2_step"])
def test_full_depth_step0_self_consistency(full_depth, golden_dir, case):
    """Step 0 of a real run (VERDICT r3 missing 5): the reference log-probs come from the precompute pass (inference_logp: plain
    un-packed rows, all_rows=True, eval mode, one forward per branch; muffin/eval/muffin_inference_logp.py:213-281), the policy
    log-probs from the packed training forward (trainers.py:161-275) - on IDENTICAL weights the rewards must vanish.  Measured
    and asserted at 32 layers: per-row delta in nats, |reward| = beta*|delta| within beta x 3 sigma of the bf16-emulated
    oracle's per-token spread (the two HIP paths differ only in attention tile order, so the delta is expected well inside)."""
    from rlaif_v_amd.inference_logp import get_multimodal_sample_logps
    FD, model, trainer, cfg = full_depth["FD"], full_depth["model"], full_depth["trainer"], full_depth["cfg"]
    fx = _fixture(FD, case, golden_dir)
    batch = FD.make_batch(case, cfg)
    B = batch["win_input_ids"].shape[0]
    rows = [{k: (v[i:i + 1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in batch.items()} for i in range(B)]
    win_lp, _, _, rej_lp, _, _ = get_multimodal_sample_logps(model, rows)          # batch size 1 like the reference (:323)
    b = dict(batch)
    b["ref_win_logp"], b["ref_rej_logp"] = torch.tensor(win_lp), torch.tensor(rej_lp)
    model.train(True)
    loss = trainer.compute_loss(model, dict(b))          # (get_beta_and_logps pops the keys it consumes, like the reference)
    out = model.last_out
    metrics = trainer.pop_metrics()
    pol = out.seq_logp.float().cpu()
    ref = torch.cat([b["ref_win_logp"], b["ref_rej_logp"]])
    delta = pol - ref
    mask = fx["labels"][:, 1:] != -100
    n_row = mask.sum(1).float()
    rms_emu = float((fx["emu_per_token"] - fx["per_token"]).pow(2).mean().sqrt())
    beta = float(batch["beta"])
    rec = dict(case=case, loss=float(loss), ln2=0.6931472, delta_nats=delta.tolist(), seq_logp=pol.tolist(), n_tokens=n_row.tolist(),
               reward_abs_max=float(beta * delta.abs().max()), bar_3sigma=(3 * beta * rms_emu * n_row.sqrt()).tolist(),
               delta_in_sigmas=(delta.abs() / (rms_emu * n_row.sqrt())).tolist(), metrics=metrics)
    print("  " + json.dumps(rec))
    _record(f"step0_self_consistency_{case}", rec)
    assert bool((delta.abs() <= 3 * rms_emu * n_row.sqrt()).all()), rec
    assert abs(float(loss) - 0.6931472) <= float((3 * beta * rms_emu * (n_row[:B] + n_row[B:]).sqrt()).mean())
    assert abs(metrics["rewards_train/chosen"]) <= rec["reward_abs_max"] + 1e-6


@pytest.mark.timeout(3600)
def test_full_depth_config1_step(full_depth, golden_dir):
    """BASELINE config 1 - THE configuration north_star states the 1e-3 bar on - at FULL DEPTH: 4 pairs, T = 512 -> L = 1087,
    32 layers, ONE whole optimisation step (forward, backward, clip, AdamW) against the fp32 oracle: indexing bit exact,
    log-prob sums and loss 1e-3, every per-tensor gradient (norm 3 %, cosine 0.99), total norm / clip factor 1 %, post-step
    fp32 masters and Adam first moment on sampled elements."""
    FD = full_depth["FD"]
    W0 = full_depth["W"]             # the HIP side never touches the CPU dict; only a LIVE oracle run moves it (AdamW in place)
    if full_depth["live"]:
        W0 = {k: v.clone() for k, v in full_depth["W"].items() if not k.startswith(O.VT)}
    FD.restore(full_depth["model"], full_depth["trainer"], full_depth["snap"])
    try:
        hip = FD.hip_case("cfg1_step", full_depth["model"], full_depth["trainer"], full_depth["cfg"], full_grads=full_depth["live"])
    finally:
        FD.restore(full_depth["model"], full_depth["trainer"], full_depth["snap"])
    fx = _fixture_or_oracle(full_depth, "cfg1_step", golden_dir)
    assert fx["labels"].shape == (8, 1087)
    m = FD.compare("cfg1_step", hip, fx, W0=W0, check=True)
    print("  " + json.dumps({k: v for k, v in m.items() if not isinstance(v, (list, dict))}))
    _record("config1_full_depth_step", m)


@pytest.mark.timeout(1800)
def test_full_depth_config1_step_with_margin(full_depth, golden_dir):
    """BASELINE config 1 once more, on a batch whose 1e-3 loss bar is a >= 4-sigma statement instead of a 0.2-sigma one (VERDICT r4
    weak 1): chosen answers 200 - 300 tokens longer than the rejected ones, so the saturated loss is beta x differences of ~2,500
    nats (tests/full_depth.py CASES["cfg1m_step"]).  Whole optimisation step, the bars of the other full-depth cases."""
    m, hip, fx = _stepping_case(full_depth, golden_dir, "cfg1m_step", "config1_full_depth_step_with_margin")
    assert fx["labels"].shape == (8, 1087)
    assert m["loss_one_sigma_rel"] <= 2.5e-4, m["loss_one_sigma_rel"]          # the bar really has >= 4 sigma of margin on this batch


@pytest.mark.timeout(2400)
def test_full_depth_config1_three_steps(full_depth, golden_dir):
    """THREE optimisation steps of config 1 at full depth (VERDICT r5 missing 3): a different batch per step, lr 5e-7 constant,
    against the layer-streamed oracle run in the reference's precision arrangement (bf16 model parameters, fp32 masters:
    script/train/llava15_train.sh:17 + script/zero2.json:11-13; tests/full_depth.py ``oracle_multistep``).  Per step the forward /
    backward bars of the one-step cases; after step 3 Adam's second moment, the bias corrections at t = 2, 3, the fp32 master
    accumulation (exact, float64 on the HIP path's own gradients) and the bf16 parameter refresh (bit exact)."""
    FD = full_depth["FD"]
    fx = _fixture(FD, "cfg1m_3step", golden_dir)
    assert len(fx["steps"]) == 3
    FD.restore(full_depth["model"], full_depth["trainer"], full_depth["snap"])
    try:
        hip = FD.hip_multistep("cfg1m_3step", full_depth["model"], full_depth["trainer"], full_depth["cfg"])
    finally:
        FD.restore(full_depth["model"], full_depth["trainer"], full_depth["snap"])
    m = FD.compare_multistep("cfg1m_3step", hip, fx, W0=full_depth["W"], check=False)
    print("  " + json.dumps(m))
    _record("config1_full_depth_three_steps", m)
    FD.compare_multistep("cfg1m_3step", hip, fx, W0=full_depth["W"], check=True)


@pytest.mark.timeout(2400)
def test_full_depth_config1_outlier_channels(full_depth, golden_dir):
    """Config 1's batch on weights whose residual stream carries OUTLIER CHANNELS >= 100 x the rest from layer 2 on (VERDICT r5
    missing 6: every real Llama checkpoint does; the N(0, 0.02) fixtures do not) - the regime in which a bf16 residual stream
    loses the most: a stream value of ~1400 is stored to +-4 while a layer's branch adds ~2.  Whole optimisation step against the
    fp32 oracle (tests/golden/fulldepth_cfg1_outlier.pt), once per residual-stream precision (``model.resid_fp32``); BOTH are
    recorded, the bars are asserted on the SHIPPED default (DESIGN section 2 records the decision taken on these numbers)."""
    FD = full_depth["FD"]
    model, trainer, cfg = full_depth["model"], full_depth["trainer"], full_depth["cfg"]
    fx = _fixture(FD, "cfg1_outlier", golden_dir)
    st = fx["outlier_stats"]
    assert all(st[i]["outlier_rms"] >= 100 * st[i]["rest_rms"] for i in st if int(i) >= 1), st       # the stream really carries them
    W2 = FD.outlier_weights(full_depth["W"], cfg, "cfg1_outlier")
    default = model.resid_fp32
    rec = {}
    try:
        model.load_state_dict(W2)
        snap2 = FD.snapshot(model)
        for flag in (False, True):
            model.resid_fp32 = flag
            FD.restore(model, trainer, snap2)
            hip = FD.hip_case("cfg1_outlier", model, trainer, cfg, fx=fx)
            m = FD.compare("cfg1_outlier", hip, fx, W0=W2, check=False)
            rec[f"resid_fp32_{int(flag)}"] = m
            print(f"  resid_fp32={int(flag)}: " + json.dumps({k: v for k, v in m.items() if not isinstance(v, dict)}))
            if flag == default:
                hip_default = hip
        del snap2
    finally:
        model.resid_fp32 = default
        model.load_state_dict(full_depth["W"])
        FD.restore(model, trainer, full_depth["snap"])
    rec["shipped_default_resid_fp32"] = bool(default)
    _record("config1_full_depth_outlier_channels", rec)
    FD.compare("cfg1_outlier", hip_default, fx, W0=W2, check=True)


@pytest.mark.parametrize("share_prefix,mi16", [(False, 1), (True, 1), (True, 0)])
def test_fullwidth_reference_golden(golden_dir, share_prefix, mi16, monkeypatch):
    """Production widths through the reference classes themselves (tests/golden/make_golden.py --full-width).  mi16 = 0 runs the
    whole model on the 32x32x16-MFMA main loops (rv_set_gemm_mi16(0)) so that path stays covered end to end."""
    _need_big_gpu()
    from rlaif_v_amd import hip
    hip.lib().call("rv_set_gemm_mi16", mi16)
    try:
        _fullwidth_golden_case(golden_dir, share_prefix, monkeypatch, f"_mi16_{mi16}")
    finally:
        hip.lib().call("rv_set_gemm_mi16", 1)


def _fullwidth_golden_case(golden_dir, share_prefix, monkeypatch, tag):
    g = torch.load(os.path.join(golden_dir, "fullwidth_l2_b2.pt"), weights_only=False)
    monkeypatch.setenv("SFT_weight", str(g["sft_weight"]))
    monkeypatch.setenv("DPO_weight", "1.0")
    cfg = O.LlavaCfg(**g["cfg"])
    model = _model(cfg, O.make_weights(cfg, seed=g["seed"]))
    model.share_prefix = share_prefix
    tr = _trainer(model)
    batch = O.make_synthetic_batch(cfg, g["n_pairs"], g["text_len"], g["prompt_len"], seed=g["seed"])
    loss = tr.compute_loss(model, dict(batch))
    out = model.last_out
    ref = dict(log_prob=g["log_prob"], labels=g["labels"], per_token_logps=g["per_token_logps"], loss=g["loss"])
    rec = _check_forward(out, loss, ref, f"full-width reference golden, share_prefix={share_prefix}")
    if not share_prefix:
        assert torch.equal(out.plan.labels.cpu(), g["labels"])
    model.backward(out, model.last_coef)
    grads = model.grads_state_dict()
    worst = 0.0
    for k, n_ref in g["grad_norms"].items():
        if k not in grads:
            assert "vision_tower" in k, k
            continue
        rel = abs(float(grads[k].double().norm()) - n_ref) / max(n_ref, 1e-12)
        worst = max(worst, rel)
        assert rel <= 3e-2 or n_ref < 1e-6, (k, rel)
    for k, gr in g["grad_full"].items():
        assert _cos(grads[k], gr) >= 0.99, k
    assert _cos(grads["model.embed_tokens.weight"].double().sum(-1), g["grad_embed_rowsum"]) >= 0.99
    rec.update(grad_worst_norm_rel_err=worst)
    _record(f"fullwidth_reference_golden_share{int(share_prefix)}{tag}", rec)
