"""FULL-DEPTH, FULL-LENGTH parity of the two BASELINE configurations that had none:

  * config 5 - RLAIF-V-7B LoRA-DPO, r = 64 / alpha 16 on all seven decoder projections (muffin/train/train_llava15_lora.py:111-116,
    304-318), spliced length L = 4096, all 32 layers: forward, backward (every adapter + projector gradient), clip, AdamW, in
    the saturated and in the conditioned regime (cfg5_step / cfg5_cond: two pairs, packed rows of 6,696 and 5,239 tokens), and
    with adapter dropout 0.05 on the masks the device draws (cfg5_drop: reference row layout, one pair = 8,192 rows);
  * config 4 - OmniLMM-12B's trainable side: precomputed tower tokens -> Resampler (omnilmm/model/resampler.py:96-168) ->
    replacement splice (omnilmm/model/omnilmm.py:221-257) -> the Mistral-7B-shaped decoder (8 key-value heads, f 14336,
    V 32009) at L = 2048 and 32 layers, forward_DPO (muffin/train/trainers.py:66-88), backward incl. every resampler gradient,
    clip, AdamW (cfg4_step / cfg4_cond).  The EVA02-E tower itself stays "parity unpinned" (timm absent).

against what the fp32 oracle produced for the same seeded weights and batch, evaluated layer by layer inside the build
container (oracle/streamed.py through tools/full_depth_oracle_streamed.py --base cfg5 | cfg5_drop | cfg4 ->
tests/golden/fulldepth_cfg{5_step,5_cond,5_drop,4_step,4_cond}.pt).  Same harness and the same bars as configs 1 / 2
(tests/full_depth.py ``compare``): indexing bit exact, log-prob sums 1e-3, the saturated loss 1e-3, logit error within 3 sigma
of the bf16-emulated oracle's spread, per-token error no larger than the emulation's, per-tensor gradient norms 3 % /
cosine 0.99 - or, where the bf16-EMULATED oracle's own backward sits below 0.99 on a tensor (config 4's late q / k projections), no
worse than that emulation -, total norm and clip factor 1 %, the optimizer checked exactly against its own inputs.  Everything goes
through the C ABI.
(Sorts after test_zz_*: the 7B full-fine-tune model of that module must be gone from HBM before these are built.)
"""
import json
import os
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # tests/full_depth.py


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


def _host_ram_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 2**20
    except OSError:
        pass
    return 0.0


def _build(base_case, monkeypatch_env):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2**30:
        pytest.skip("needs the 288 GB part")
    if _host_ram_gb() < 48:
        pytest.skip("the seeded fp32 weights of the 7B model (27 GB on the host) do not fit this box")
    import full_depth as FD
    os.environ["SFT_weight"], os.environ["DPO_weight"] = "0.0", "1.0"
    torch.cuda.empty_cache()
    cfg = FD.make_cfg(32, base_case)
    t0 = time.time()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    W = FD.make_case_weights(base_case, cfg)
    model, trainer = FD.build_model(cfg, W, with_optimizer=True, case=base_case)
    print(f"  {base_case}: weights + model in {time.time() - t0:.0f} s")
    return dict(FD=FD, cfg=cfg, W=W, model=model, trainer=trainer)


_LIVE = {}        # at most ONE 7B model on the device at a time (the LoRA and the OmniLMM model together would not leave room for activations)


def _get(base_case):
    if base_case not in _LIVE:
        import gc
        _LIVE.clear()
        gc.collect()
        torch.cuda.empty_cache()
        _LIVE[base_case] = _build(base_case, None)
    return _LIVE[base_case]


@pytest.fixture
def lora_fd():
    return _get("cfg5_step")


@pytest.fixture
def omni_fd():
    return _get("cfg4_step")


@pytest.fixture(scope="module", autouse=True)
def _free_models_at_module_end():
    yield
    import gc
    _LIVE.clear()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def _fixture(FD, case, golden_dir):
    path = os.path.join(golden_dir, f"fulldepth_{case}.pt")
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: generate it with tools/full_depth_oracle_streamed.py (build container)")
    fx = torch.load(path, weights_only=False)
    assert fx["layers"] == 32 and fx["weight_seed"] == FD.WEIGHT_SEED and fx["case"] == case
    return fx


def _stepping_case(fd, golden_dir, case, key):
    FD = fd["FD"]
    fx = _fixture(FD, case, golden_dir)
    if "snap" not in fd:
        fd["snap"] = FD.snapshot(fd["model"])
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
def test_full_depth_config5_lora_step(lora_fd, golden_dir):
    """config 5 at its stated shape: LoRA r = 64, L = 4096, 32 layers, two pairs packed into rows of 6,696 / 5,239 tokens
    (RoPE positions to 4095, 64 key tiles per chosen branch), one whole optimisation step on the adapters + projector."""
    m, hip, fx = _stepping_case(lora_fd, golden_dir, "cfg5_step", "config5_full_depth_step")
    assert fx["labels"].shape == (4, 4096) and hip["plan_S"] == 2 and hip["plan_L"] >= 6696
    assert m["grad_tensors"] == 32 * 14 + 4                              # every adapter tensor + the projector's four


@pytest.mark.timeout(1800)
def test_full_depth_config5_lora_conditioned(lora_fd, golden_dir):
    """the same batch with beta z = 0 / +1: the regime DPO training starts in (logit error vs the bf16-emulated spread,
    backward on the oracle's coefficients)."""
    m, _, _ = _stepping_case(lora_fd, golden_dir, "cfg5_cond", "config5_full_depth_conditioned")
    assert all(abs(z - t) < 0.5 for z, t in zip(m["logit"], m["beta_z"]))


@pytest.mark.timeout(1800)
def test_full_depth_config5_lora_dropout(lora_fd, golden_dir):
    """config 5 AS IT TRAINS: adapter dropout 0.05.  The oracle fixture was computed in the build container on the masks
    oracle/dropout_mask.py predicts for the model's first training forward; this test first pins that prediction bit-exactly
    against rv_dropout on the device, then runs the step in the reference row layout (the masks index [S L, in] rows)."""
    from oracle import dropout_mask as DM
    from rlaif_v_amd import ops
    model = lora_fd["model"]
    for rows, width, seed in ((4096, 4096, DM.model_seed(1, 0, 31, 0)), (2048, 11008, DM.model_seed(1, 0, 0, 3))):
        dev = ops.dropout(torch.ones(rows, width, dtype=torch.bfloat16, device="cuda:0"), 0.05, seed)
        assert torch.equal((dev != 0).cpu(), torch.from_numpy(DM.keep_mask(rows * width, 0.05, seed).reshape(rows, width)))
    p0, sp0 = model.lora.lora_dropout, model.share_prefix
    model.lora.lora_dropout, model.share_prefix = 0.05, False
    try:
        m, hip, fx = _stepping_case(lora_fd, golden_dir, "cfg5_drop", "config5_full_depth_dropout")
    finally:
        model.lora.lora_dropout, model.share_prefix = p0, sp0
    assert fx["labels"].shape == (2, 4096) and hip["plan_S"] == 2 and hip["plan_L"] == 4096
    assert all(abs(z - t) < 0.5 for z, t in zip(m["logit"], m["beta_z"]))


@pytest.mark.timeout(1800)
def test_full_depth_config4_omnilmm_step(omni_fd, golden_dir):
    """config 4's trainable side at 32 layers, L = 2048: Resampler + grouped-query decoder, whole optimisation step."""
    m, hip, fx = _stepping_case(omni_fd, golden_dir, "cfg4_step", "config4_full_depth_step")
    assert fx["labels"].shape == (4, 2048) and hip["plan_S"] == 2
    assert any(k.startswith("model.resampler.") for k in fx["grad_norms"])


@pytest.mark.timeout(1800)
def test_full_depth_config4_omnilmm_conditioned(omni_fd, golden_dir):
    m, _, _ = _stepping_case(omni_fd, golden_dir, "cfg4_cond", "config4_full_depth_conditioned")
    assert all(abs(z - t) < 0.5 for z, t in zip(m["logit"], m["beta_z"]))
