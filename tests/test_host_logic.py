"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, the splice planner is
bit-identical to the oracle restatement of prepare_inputs_labels_for_multimodal, the collator reproduces the
reference collator's golden batches, schedules / layouts are consistent.  No GPU, no compute calls."""
import importlib.util
import os
import sys

import pytest
import torch

from oracle import dpo_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden_instances(seed):
    spec = importlib.util.spec_from_file_location("mkcol", os.path.join(REPO, "tests", "golden", "make_collator_golden.py"))
    src = open(spec.origin).read()
    # reuse the instance builder without importing the reference (not available on the GPU box)
    ns = {}
    from rlaif_v_amd.data import SyntheticPreferenceDataset
    body = src[src.index("def instances(seed):"):src.index('if __name__ == "__main__":')]
    exec(body, {"torch": torch, "SyntheticPreferenceDataset": SyntheticPreferenceDataset}, ns)
    return ns["instances"](seed)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from rlaif_v_amd import hip
    lib = hip.lib()
    assert len(lib.decls) >= 37
    for name in lib.decls:
        assert hasattr(lib.lib, name), name
    assert lib.lib.rv_abi_version() == 7
    # argument errors are reported through the ABI (no launch happens for an invalid shape)
    assert lib.lib.rv_set_gemm_variant(7) == 1 and "rv_set_gemm_variant" in lib.last_error()
    assert lib.lib.rv_set_gemm_variant(1) == 0


def test_product_path_fails_loudly_without_library(tmp_path):
    from rlaif_v_amd import hip
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.HipLib(str(tmp_path / "missing.so"))
    if not torch.cuda.is_available():
        from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            LlavaDPOModel(LlavaConfig(**O.asdict(O.tiny_cfg())))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "rlaif-v_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn


@pytest.mark.parametrize("max_len", [None, 48, 60, 4096])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_splice_plan_bit_exact(seed, max_len):
    from rlaif_v_amd.splice import build_splice_plan
    cfg = O.tiny_cfg()
    B = 3
    b = O.make_synthetic_batch(cfg, B, 44, 13, seed=seed)
    ids, lab = b["concatenated_input_ids"], b["concatenated_labels"]
    sk, si, nl = O.splice_plan(ids, lab, cfg.n_patches, max_len)
    p = build_splice_plan(ids, lab, cfg.n_patches, B, max_len)
    S, L = sk.shape
    assert (p.S, p.L) == (S, L) and torch.equal(p.labels, nl)
    src = p.src.view(S, L).long()
    kind = torch.where(src >= 0, 1, torch.where(src == -1, 0, 2))
    assert torch.equal(kind, sk) and torch.equal(src[sk == 1], si[sk == 1])
    rows = torch.nonzero(sk == 2)[:, 0]
    assert torch.equal(-2 - src[sk == 2], (rows % B) * cfg.n_patches + si[sk == 2])
    # row selection == get_batch_logps' shift + mask
    mask = nl[:, 1:] != -100
    s_idx, l_idx = torch.nonzero(mask, as_tuple=True)
    assert torch.equal(p.sel_idx.long(), s_idx * L + l_idx) and torch.equal(p.tgt.long(), nl[:, 1:][mask])
    assert p.seq_off.tolist() == [0] + torch.cumsum(mask.sum(1), 0).tolist()
    assert torch.equal(p.seq_of_row.long(), s_idx)
    # embedding-backward segments: every text row appears once, grouped by id
    flat = src.reshape(-1)
    assert sorted(p.pos_sorted.tolist()) == torch.nonzero(flat >= 0)[:, 0].tolist()
    for u in range(p.uniq_ids.numel()):
        seg = p.pos_sorted[p.seg_off[u]:p.seg_off[u + 1]].long()
        assert bool((flat[seg] == p.uniq_ids[u]).all())
    # feature-gradient sources point back at the rows that consumed that feature row
    for r in range(p.feat_src_a.numel()):
        for srcrow in (int(p.feat_src_a[r]), int(p.feat_src_b[r])):
            if srcrow >= 0:
                assert int(flat[srcrow]) == -2 - r


def test_splice_plan_edges():
    from rlaif_v_amd.splice import build_splice_plan
    # a row without an image token still consumes an image index (llava_arch.py:241-247); empty answers
    ids = torch.tensor([[1, 5, 6, 7], [1, -200, 8, 0], [1, -200, 9, 2]])
    lab = torch.tensor([[-100, 5, 6, 7], [-100, -100, -100, -100], [-100, -100, 9, 2]])
    p = build_splice_plan(ids, lab, 3, 3, None)
    assert p.L == 6 and p.src.view(3, 6)[0].tolist() == [1, 5, 6, 7, -1, -1]
    assert p.src.view(3, 6)[1].tolist() == [1, -5, -6, -7, 8, 0]        # image index 1 -> rows 3..5
    assert p.src.view(3, 6)[2].tolist() == [1, -8, -9, -10, 9, 2]       # image index 2 -> rows 6..8
    assert p.seq_off.tolist() == [0, 3, 3, 5]                           # row 1 has no target -> avg logp NaN
    sk, si, nl = O.splice_plan(ids, lab, 3, None)
    assert torch.equal(p.labels, nl)


def _unpack(p, b, B):
    """Reconstruct (chosen, rejected) source / position rows of pair b from a packed plan."""
    L = p.L
    src = p.src.view(p.S, L)[b].long()
    pos = p.pos.view(p.S, L)[b].long()
    sh, e1 = int(p.seg_sh[b]), int(p.seg_e1[b])
    n = int((src != -1).sum()) if bool((src == -1).any()) else L
    n = max(n, e1)
    chosen, cpos = src[:e1], pos[:e1]
    rej, rpos = torch.cat([src[:sh], src[e1:n]]), torch.cat([pos[:sh], pos[e1:n]])
    return chosen, cpos, rej, rpos, sh, e1


@pytest.mark.parametrize("seed,max_len,common_answer", [(1, None, 0), (2, 50, 0), (3, None, 5), (4, 4096, 40)])
def test_packed_plan_is_a_lossless_repacking(seed, max_len, common_answer):
    from rlaif_v_amd.splice import build_packed_plan, build_splice_plan, _splice_rows
    cfg = O.tiny_cfg()
    B = 3
    b = O.make_synthetic_batch(cfg, B, 44, 13, seed=seed)
    ids, lab = b["concatenated_input_ids"].clone(), b["concatenated_labels"].clone()
    if common_answer:      # chosen / rejected share the first answer tokens: those rows carry targets of BOTH sequences
        T = ids.shape[1]
        k = min(common_answer, T - 13)
        ids[B:, 13:13 + k] = ids[:B, 13:13 + k]
        lab[B:, 13:13 + k] = torch.where(lab[B:, 13:13 + k] != -100, ids[B:, 13:13 + k], lab[B:, 13:13 + k])
    ref = build_splice_plan(ids, lab, cfg.n_patches, B, max_len)
    p = build_packed_plan(ids, lab, cfg.n_patches, B, max_len, cfg.pad_token_id)
    assert p.S == B and p.n_seq == 2 * B and p.labels is None
    rows_src, rows_lab = _splice_rows(ids, lab, cfg.n_patches, B, max_len)
    for bi in range(B):
        chosen, cpos, rej, rpos, sh, e1 = _unpack(p, bi, B)
        for got, gpos, r in ((chosen, cpos, bi), (rej, rpos, B + bi)):
            full = rows_src[r]
            assert torch.equal(got, full[:got.numel()])                 # same tokens, only right padding dropped
            rest, rest_lab = full[got.numel():], rows_lab[r][got.numel():]
            assert bool(((rest == cfg.pad_token_id) & (rest_lab == -100)).all())
            assert torch.equal(gpos, torch.arange(got.numel()))         # RoPE positions of the reference layout
        # nothing with a target (as input row) lives in the shared part
        assert sh <= min(int(torch.nonzero(rows_lab[bi] != -100)[0]), int(torch.nonzero(rows_lab[B + bi] != -100)[0])) - 1 \
            or sh == 0
    # same targets in the same (sequence, position) order, same per-sequence counts
    assert torch.equal(p.tgt, ref.tgt) and torch.equal(p.seq_off, ref.seq_off) and torch.equal(p.seq_of_row, ref.seq_of_row)
    # each selected packed row holds the same source token as the reference row it replaces
    assert torch.equal(p.src[p.sel_idx.long()], ref.src[ref.sel_idx.long()])
    assert torch.equal(p.pos[p.sel_idx.long()].long(), ref.sel_idx.long() % ref.L)
    assert p.n_real_tokens < ref.n_real_tokens
    if common_answer == 0 and max_len is None:
        assert all(s == 13 - 1 + cfg.n_patches - 1 for s in p.shared_len)   # whole prompt minus its last token


@pytest.mark.parametrize("seed", [1, 5])
def test_pad_free_plan_is_the_rectangular_plan_without_its_padding(seed):
    """build_packed_plan(pad_free=True): the B packed rows concatenated - every index table maps to the SAME tokens as the
    rectangular packed plan (llava_arch.py:305-313 pads; SURVEY 8a property (i) says dropping pads is exact)."""
    from rlaif_v_amd.splice import build_packed_plan
    cfg = O.tiny_cfg()
    B = 4
    b = O.make_synthetic_batch(cfg, B, 52, 13, seed=seed, ragged=True)
    ids, lab = b["concatenated_input_ids"], b["concatenated_labels"]
    r = build_packed_plan(ids, lab, cfg.n_patches, B, None, cfg.pad_token_id)
    p = build_packed_plan(ids, lab, cfg.n_patches, B, None, cfg.pad_token_id, pad_free=True)
    assert r.row_off is None and r.rows is None and r.n_tokens == r.S * r.L
    assert p.S == r.S and p.L == r.L and p.n_seq == r.n_seq and p.n_sel == r.n_sel
    lens = p.row_len.tolist()
    assert p.row_off.tolist() == [sum(lens[:i]) for i in range(B)] and p.n_tokens == sum(lens) == p.n_real_tokens == r.n_real_tokens
    assert p.n_tokens < r.n_tokens and max(lens) == p.L                      # a ragged batch really loses rows
    src_r, pos_r = r.src.view(B, r.L), r.pos.view(B, r.L)
    for i in range(B):
        a, n = int(p.row_off[i]), lens[i]
        assert torch.equal(p.src[a:a + n], src_r[i, :n]) and bool((src_r[i, n:] == -1).all())
        assert torch.equal(p.pos[a:a + n], pos_r[i, :n])
    assert torch.equal(p.tgt, r.tgt) and torch.equal(p.seq_off, r.seq_off) and torch.equal(p.seq_of_row, r.seq_of_row)
    assert torch.equal(p.seg_sh, r.seg_sh) and torch.equal(p.seg_e1, r.seg_e1)
    # a selected row is the same (packed row, offset) in both layouts
    row_r, off_r = r.sel_idx.long() // r.L, r.sel_idx.long() % r.L
    assert torch.equal(p.sel_idx.long(), p.row_off.long()[row_r] + off_r)
    # embedding-gradient segments and feature users: same token ids / same (row, offset) positions
    assert torch.equal(p.uniq_ids, r.uniq_ids) and torch.equal(p.seg_off, r.seg_off)
    to_pf = lambda t: p.row_off.long()[t.long() // r.L] + t.long() % r.L                                      # noqa: E731
    assert torch.equal(p.pos_sorted.long(), to_pf(r.pos_sorted))
    for fa, fb in ((p.feat_src_a, r.feat_src_a), (p.feat_src_b, r.feat_src_b)):
        used = fb >= 0
        assert torch.equal(fa >= 0, used) and torch.equal(fa[used].long(), to_pf(fb[used]))


@pytest.mark.parametrize("seed", [1, 2])
def test_collator_matches_reference_golden(golden_dir, seed):
    from rlaif_v_amd.data import DataCollatorForDPODataset
    gold = torch.load(os.path.join(golden_dir, "collator.pt"), weights_only=False)[seed]

    class Tok:
        pad_token_id = 0
    batch = DataCollatorForDPODataset(Tok(), beta=0.1, mod_token_weight=1.5)(_golden_instances(seed))
    assert set(batch) == set(gold)
    for k, v in gold.items():
        if torch.is_tensor(v):
            assert batch[k].dtype == v.dtype and torch.equal(batch[k], v), k
        else:
            assert batch[k] == v, k
    assert (batch["win_token_weight"] == 1.5).any()          # the difflib path really fired


def test_param_store_layout_and_schedules():
    from rlaif_v_amd.model import LlavaConfig, ParamStore
    from rlaif_v_amd.trainer import cosine_lr
    cfg = LlavaConfig(**O.asdict(O.tiny_cfg()))
    st = ParamStore(cfg, "cpu")
    sch = st.bucket_schedule()
    assert sch[0][1] == 0 and sch[-1][2] == st.n_total
    assert all(sch[i][2] == sch[i + 1][1] for i in range(len(sch) - 1))
    names = st.hf_slices(cfg)
    assert set(names) == set(O.trainable_names(O.tiny_cfg()))
    for hf, (key, r0, n, step) in names.items():
        off, shp = st.offsets[key]
        assert (off >= st.n_decay) == (not O.is_decay_param(hf)), hf     # decay / no-decay split == HF's
        assert tuple(st.rows(st.p(key), r0, n, step).shape) == tuple(O.weight_shapes(O.tiny_cfg())[hf]), hf
    # the fused MLP weight interleaves gate / up rows (SwiGLU in the GEMM epilogue); together they tile it exactly
    g_, u_ = names["model.layers.0.mlp.gate_proj.weight"], names["model.layers.0.mlp.up_proj.weight"]
    assert g_[0] == u_[0] and (g_[1], g_[3], u_[1], u_[3]) == (0, 2, 1, 2) and g_[2] == u_[2] == cfg.ffn
    for s in range(0, 40):
        assert abs(cosine_lr(s, 40, 5e-7, 0.05) - O.cosine_lr(s, 40, 5e-7, 0.05)) < 1e-18
    # the 7B layout: 6.76 B trainable parameters
    big = LlavaConfig()
    d, f, V = big.hidden, big.ffn, big.vocab
    n = 2 * V * d + 32 * (4 * d * d + 3 * d * f + 2 * d) + d + (d * 1024 + d) + (d * d + d)
    assert n == 6_759_272_448 or n > 6.7e9


def test_param_store_lora_layout():
    """LoRA mode: frozen base in front, adapters (backward-completion order) + projector trainable; gradient buckets
    tile flat_g exactly; peft tensor names map to (padded) slices of the fused q|k|v / gate|up adapter stacks."""
    from rlaif_v_amd.model import LlavaConfig, LoraConfig, ParamStore
    cfg = LlavaConfig(**O.asdict(O.tiny_cfg()))
    st = ParamStore(cfg, "cpu", lora=LoraConfig(r=16, lora_alpha=32))
    assert st.lora.scaling == 2.0 and st.lora.r_pad == 64
    sch = st.bucket_schedule()
    assert sch[0][1] == 0 and sch[-1][2] == st.n_train == st.flat_g.numel()
    assert all(sch[i][2] == sch[i + 1][1] for i in range(len(sch) - 1))
    assert [n for n, _, _ in sch][:2] == [f"layer{cfg.layers - 1}", f"layer{cfg.layers - 2}"]
    with pytest.raises(KeyError):
        st.g("layers.0.wqkv")                                # frozen: no gradient slot
    assert st.g("layers.0.lora_qkv.A").shape == (3 * 64, cfg.hidden)
    assert st.train_p.data_ptr() == st.flat_p[st.t0:].data_ptr()
    W = O.make_lora_weights(O.tiny_cfg(), 16)
    sl = st.lora_slices(cfg)
    assert set(sl) == set(W)
    for name, (key, r0, n, ncol, step) in sl.items():
        assert tuple(st.lora_view(st.p(key), r0, n, ncol, step).shape) == tuple(W[name].shape), name
        assert key in st.trainable and key in st.t_offsets
    # trainable set == what the reference's LoRA run trains (adapters + mm_projector)
    full = dict(O.make_weights(O.tiny_cfg(), seed=0))
    full.update(W)
    ref_names = {n for n in O.lora_trainable_names(full)}
    ours = set(sl) | {k for k in st.trainable if "mm_projector" in k}
    assert ours == ref_names
    # 7B: rank-64 adapters on all seven projections = 159.9 M parameters (+ 21 M projector)
    d, f, r = 4096, 11008, 64
    assert 32 * r * (4 * 2 * d + 3 * (d + f)) == 159_907_840


def test_checkpoint_roundtrip_hf_layout(tmp_path):
    from rlaif_v_amd.checkpoint import (config_from_hf, hf_config_dict, load_state_dict_dir, save_state_dict_sharded)
    from rlaif_v_amd.model import LlavaConfig
    import json
    cfg = LlavaConfig(**O.asdict(O.tiny_cfg()))
    sd = {k: v.to(torch.bfloat16) for k, v in O.make_weights(O.tiny_cfg(), seed=3).items()}
    index = save_state_dict_sharded(sd, str(tmp_path), cfg, max_shard_bytes=1 << 20)     # forces several shards
    files = sorted(os.listdir(tmp_path))
    assert "config.json" in files and "model.safetensors.index.json" in files and sum(f.endswith(".safetensors") for f in files) > 2
    assert set(index["weight_map"]) == set(sd)
    back = load_state_dict_dir(str(tmp_path))
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    cfg2 = config_from_hf(json.load(open(tmp_path / "config.json")), clip_layers=cfg.clip_layers, clip_heads=cfg.clip_heads,
                          clip_ffn=cfg.clip_ffn, image_size=cfg.image_size)
    assert cfg2 == cfg
    assert hf_config_dict(cfg)["model_type"] == "llava_llama"
    # legacy .bin checkpoints load too
    os.makedirs(tmp_path / "bin")
    torch.save(sd, tmp_path / "bin" / "pytorch_model.bin")
    back2 = load_state_dict_dir(str(tmp_path / "bin"))
    assert all(torch.equal(back2[k], sd[k]) for k in sd)


def test_sample_encoding_matches_reference_golden(golden_dir, tmp_path):
    """preprocess_v1 / tokenizer_image_token / encode_multimodal_preference_sample vs the reference's own functions
    (tests/golden/make_preprocess_golden.py), then the parquet -> RLAIFVDataset -> DPODataset -> collator chain."""
    import json
    import sys
    sys.path.insert(0, golden_dir)
    from toy_tokenizer import SAMPLES, ToyTokenizer
    from rlaif_v_amd.dataset import (DPODataset, encode_multimodal_preference_sample, llava_v1_prompt, preprocess_v1,
                                      tokenizer_image_token)
    from rlaif_v_amd.data import DataCollatorForDPODataset
    gold = torch.load(os.path.join(golden_dir, "preprocess.pt"), weights_only=False)
    tok = ToyTokenizer()
    cfg = dict(image_processor=lambda img: torch.full((3, 4, 4), float(img)), keep_image_tag=True, is_multimodal=True)
    for i, s in enumerate(SAMPLES):
        src = dict(image=i, question={"from": "human", "value": f"<image>\n{s['question']}"},
                   chosen={"from": "gpt", "value": s["chosen"]}, rejected={"from": "gpt", "value": s["rejected"]},
                   ref_win_logp=-1.0 - i, ref_rej_logp=-2.0 - i, ref_win_avg_logp=-0.1, ref_rej_avg_logp=-0.2,
                   ref_win_per_token_logp=[0.0, -1.0], ref_rej_per_token_logp=[-2.0])
        rej, win = encode_multimodal_preference_sample(src, tok, cfg)
        for got, ref in ((rej, gold[i]["rej"]), (win, gold[i]["win"])):
            assert set(got) == set(ref)
            for k, v in ref.items():
                if torch.is_tensor(v):
                    assert torch.equal(got[k], v), k
                else:
                    assert got[k] == v, k
        assert int((win["input_ids"] == -200).sum()) == 1
    p = llava_v1_prompt([{"from": "human", "value": "<image>\nhi"}, {"from": "gpt", "value": "yo"}])
    assert p.endswith("USER: <image>\nhi ASSISTANT: yo</s>")
    assert tokenizer_image_token("a <image> b", tok).count(-200) == 1

    # parquet rows (reference schema incl. the JSON logps column) -> dataset -> collator
    import pandas as pd
    from PIL import Image
    import io
    rows = []
    for i, s in enumerate(SAMPLES):
        buf = io.BytesIO()
        Image.new("RGB", (8, 8), (10 * i, 20, 30)).save(buf, format="PNG")
        n_w, n_r = 60, 60
        rows.append(dict(image={"bytes": buf.getvalue()}, question=s["question"], chosen=s["chosen"], rejected=s["rejected"],
                         origin_dataset="toy", origin_split="train", idx=i, image_path=f"{i}.png",
                         logps=json.dumps({"logps": [-5.0 - i, -0.5, [0.0] * n_w, -6.0 - i, -0.6, [0.0] * n_r]})))
    pd.DataFrame(rows).to_parquet(tmp_path / "RLAIF-V-Dataset-withlogp_000-3.parquet")
    ds = DPODataset(tok, str(tmp_path), dict(image_processor=lambda im: torch.zeros(3, 4, 4), is_multimodal=True))
    assert len(ds) == 3
    batch = DataCollatorForDPODataset(tok, beta=0.1, mod_token_weight=1.0)([ds[i] for i in range(3)])
    assert batch["images"].shape == (3, 3, 4, 4) and batch["ref_win_logp"].tolist() == [-5.0, -6.0, -7.0]
    assert batch["concatenated_input_ids"].shape[0] == 6 and (batch["concatenated_input_ids"] == -200).sum() == 6
    with pytest.raises(AssertionError, match="reference_model"):       # muffin/data/datasets.py:39 (same message)
        DPODataset(tok, str(tmp_path / "empty"), {})
    with pytest.raises(FileNotFoundError):                             # reference model given, but no raw rows to score (no hub)
        DPODataset(tok, str(tmp_path / "empty2"), {}, reference_model=object())


def test_entrypoint_flag_surface():
    """script/train/llava15_train.sh and llava15_train_lora.sh flag lines parse into the reference's three argument
    dataclasses (muffin/train/train_llava15.py:32-100, train_llava15_lora.py:111-116); control-plane flags are ignored."""
    from rlaif_v_amd import train_llava15 as T
    common = ("--deepspeed ./script/zero2.json --model_name_or_path liuhaotian/llava-v1.5-7b --data_dir ./RLAIF-V-Dataset_logps/ "
              "--image_folder not_used --vision_tower openai/clip-vit-large-patch14-336 --mm_use_im_start_end False "
              "--mm_use_im_patch_token False --image_aspect_ratio pad --bf16 True --mm_projector_type mlp2x_gelu "
              "--mm_vision_select_layer -2 --output_dir .ckpt/x --num_train_epochs 10 --per_device_train_batch_size 1 "
              "--per_device_eval_batch_size 4 --gradient_accumulation_steps 1 --evaluation_strategy no --save_strategy steps "
              "--save_steps 167 --save_total_limit 50 --data_source_names x --data_source_weights 1 --max_steps 2672 "
              "--weight_decay 0.01 --warmup_ratio 0.05 --lr_scheduler_type cosine --logging_steps 2 --logging_dir .ckpt/log "
              "--tf32 True --model_max_length 2048 --gradient_checkpointing True --lazy_preprocess True --task DPO "
              "--report_to wandb --run_name x --dataloader_num_workers 16 --dpo_use_average False --dpo_token_weighted False "
              "--dpo_token_weight 1.0 --dpo_beta 0.1")
    m, d, t = T.parse_args((common + " --fully_tune True --learning_rate 5e-7").split())
    assert (m.mm_vision_select_layer, m.mm_projector_type, m.version) == (-2, "mlp2x_gelu", "llava_v1")
    assert (d.dpo_beta, d.dpo_token_weight, d.image_aspect_ratio, d.lazy_preprocess) == (0.1, 1.0, "pad", True)
    assert (t.task, t.max_steps, t.learning_rate, t.warmup_ratio, t.model_max_length, t.save_steps, t.bf16) == \
        ("DPO", 2672, 5e-7, 0.05, 2048, 167, True)
    assert t.fully_tune and t.gradient_checkpointing and not t.lora_enable and t.lora_config() is None
    m, d, t = T.parse_args((common + " --fully_tune False --learning_rate 1e-5 --lora_enable True").split())
    lc = t.lora_config()
    assert t.lora_enable and (lc.r, lc.lora_alpha, lc.lora_dropout, lc.bias) == (64, 16, 0.05, "none") and lc.scaling == 0.25
    # dataclass defaults are the reference's
    assert T.DataArguments().dpo_beta == 0.5 and T.DataArguments().dpo_token_weight == 3.0
    assert T.ModelArguments().mm_vision_select_layer == -1 and T.TrainingArguments().task == "LM"
    with pytest.raises(NotImplementedError):
        T.init_model(T.ModelArguments(), T.DataArguments(), T.TrainingArguments(task="LM"))


def _bare_trainer(n, world=1, rank=0, bs=1, seed=42, **kw):
    """A trainer shell for the host-side loop logic (no model, no GPU)."""
    from rlaif_v_amd.trainer import GradReducer, LLaVA15DPOTrainer, TrainingArguments
    tr = LLaVA15DPOTrainer.__new__(LLaVA15DPOTrainer)
    tr.args = TrainingArguments(per_device_train_batch_size=bs, seed=seed, **kw)
    tr.reducer = GradReducer()
    tr.reducer.world_size = world
    tr.train_dataset, tr.data_collator = list(range(n)), (lambda x: x)
    tr.state = dict(global_step=0, log_history=[], epoch=0, batches_in_epoch=0)
    os.environ["RANK"] = str(rank)
    return tr


def test_sampler_reshuffles_every_epoch_and_resumes_mid_epoch(monkeypatch):
    """HF Trainer semantics the shipped script relies on (4 epochs over the data, auto-resume from checkpoint-*): a new
    permutation per epoch, identical on every rank; a resumed run skips the batches its epoch already consumed; ranks get
    equally many full batches; a rank without a single full batch raises instead of spinning."""
    monkeypatch.setenv("RANK", "0")
    tr = _bare_trainer(23, bs=2)
    e0 = [b for b in tr.get_train_dataloader(0)]
    e1 = [b for b in tr.get_train_dataloader(1)]
    assert len(e0) == len(e1) == 11 and e0 != e1                       # reshuffled
    assert [b for b in tr.get_train_dataloader(0)] == e0               # deterministic in (seed, epoch)
    assert [b for b in tr.get_train_dataloader(0, skip_batches=4)] == e0[4:]        # resume position
    flat = sorted(x for b in e0 for x in b)
    assert len(set(flat)) == 22                                        # a permutation, drop_last
    # world 3: disjoint equal shards of ONE permutation
    shards = []
    for r in range(3):
        t = _bare_trainer(23, world=3, rank=r, bs=2)
        shards.append([x for b in t.get_train_dataloader(5) for x in b])
    assert all(len(s) == 6 for s in shards) and len(set(sum(shards, []))) == 18
    os.environ["RANK"] = "0"
    with pytest.raises(ValueError, match="no full batch"):
        _bare_trainer(3, world=2, rank=0, bs=2).get_train_dataloader(0)


def test_lr_schedules_and_rejected_flags():
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments, cosine_lr, lr_at
    assert lr_at("cosine", 10, 100, 1.0, 0.05) == cosine_lr(10, 100, 1.0, 0.05)
    assert lr_at("constant", 0, 100, 2.0, 0.5) == 2.0
    assert lr_at("linear", 0, 100, 1.0, 0.1) == 0.0 and lr_at("linear", 5, 100, 1.0, 0.1) == 0.5
    assert abs(lr_at("linear", 55, 100, 1.0, 0.1) - 0.5) < 1e-12 and lr_at("linear", 100, 100, 1.0, 0.1) == 0.0
    assert lr_at("constant_with_warmup", 50, 100, 3.0, 0.1) == 3.0
    # HF's own schedules agree (transformers is installed)
    import transformers
    p = torch.nn.Parameter(torch.zeros(1))
    for kind in ("linear", "cosine", "constant_with_warmup"):
        opt = torch.optim.SGD([p], lr=1.0)
        sch = transformers.get_scheduler(kind, opt, num_warmup_steps=10, num_training_steps=100)
        for step in range(0, 100, 7):
            assert abs(sch.get_last_lr()[0] - lr_at(kind, step, 100, 1.0, 0.1)) < 1e-9, (kind, step)
            for _ in range(7):
                opt.step(), sch.step()
    with pytest.raises(NotImplementedError, match="lr_scheduler_type"):
        lr_at("polynomial", 1, 10, 1.0, 0.0)

    class _M:                       # constructor-time validation needs no device
        device = "cpu"
        training = True
        lora = None
    with pytest.raises(NotImplementedError):
        LLaVA15DPOTrainer(model=_M(), args=TrainingArguments(lr_scheduler_type="inverse_sqrt"))
    with pytest.raises(ValueError):
        LLaVA15DPOTrainer(model=_M(), args=TrainingArguments(gradient_accumulation_steps=0))


def test_dkv_isa_has_no_compiler_agpr_traffic():
    """attn_bwd_dkv3_kernel / attn_bwd_dkv5_kernel address the accumulator file by hard register numbers (csrc/attn_agpr.inc).
    That is only sound while hipcc itself never touches AGPRs in those kernels (no spill-to-AGPR, no scratch): audit the generated
    ISA.  Version 5 additionally relies on counted lgkmcnt waits: no scalar memory load may sit inside its tile body."""
    import shutil
    import subprocess
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "check_agpr_isa.sh")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-500:]
    assert "instructions in dkv3: 0" in out.stdout and "instructions in dkv5: 0" in out.stdout
    assert "instructions in fwd3: 0" in out.stdout               # round 6: attn_fwd3_kernel owns a[0:255] by number as well
    assert "SMEM loads inside the dkv5 tile body: 0" in out.stdout


def test_dkv5_generated_bodies_are_current():
    """csrc/attn_dkv5_*.inc are generated (tools/gen_attn_dkv5.py): the committed files must be what the generator emits."""
    import subprocess
    import sys
    import tempfile
    csrc = os.path.join(REPO, "rlaif-v_amd", "csrc")
    keep = {n: open(os.path.join(csrc, n)).read() for n in os.listdir(csrc) if n.startswith("attn_dkv5_")}
    with tempfile.TemporaryDirectory() as td:            # generated into a scratch directory: the tree is never touched
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_attn_dkv5.py"), "--out", td], check=True, capture_output=True)
        assert sorted(os.listdir(td)) == sorted(keep)
        for n, txt in keep.items():
            assert open(os.path.join(td, n)).read() == txt, f"{n} is stale: run python tools/gen_attn_dkv5.py"
    assert len(keep) == 3


def test_omnilmm_splice_plan_matches_oracle_bit_exact():
    """The <im_start>/<im_end> replacement rule of the planner against the oracle's restatement of omnilmm.py:221-257
    (itself pinned by the reference golden): identical spliced rows, unchanged labels, both layouts' targets."""
    import importlib
    from oracle import omnilmm_oracle as OO
    from oracle import dpo_oracle as O
    sp = importlib.import_module("rlaif-v_amd.splice")
    tokens = (317, 318, 319)
    cfg = O.LlavaCfg(hidden=512, layers=1, heads=4, kv_heads=2, ffn=768, vocab=320, model_max_length=256)
    b = OO.make_omnilmm_batch(cfg, 3, 60, 16, tokens, seed=2)
    ids, lab = b["concatenated_input_ids"], b["concatenated_labels"]
    emb = torch.arange(cfg.vocab, dtype=torch.float32)[:, None].repeat(1, 2)
    feats = (-2.0 - torch.arange(3 * 16, dtype=torch.float32)).view(3, 16, 1).repeat(1, 1, 2)
    ref = OO.omnilmm_splice(ids, emb, torch.cat([feats, feats], 0), *tokens)[..., 0]        # [6, T]: id or -2 - feature row
    plan = sp.build_splice_plan(ids, lab, 16, 3, 256, splicer=sp.make_omnilmm_splicer(*tokens))
    assert torch.equal(plan.src.view(plan.S, plan.L).float(), ref)
    assert torch.equal(plan.labels, lab)
    packed = sp.build_packed_plan(ids, lab, 16, 3, 256, 0, splicer=sp.make_omnilmm_splicer(*tokens))
    assert torch.equal(packed.tgt, plan.tgt) and packed.S == 3 and min(packed.shared_len) >= 1 + 8 + 18
    bad = ids.clone()
    bad[0, int(torch.where(ids[0] == tokens[2])[0][0])] = 7
    with pytest.raises(ValueError):
        sp.build_splice_plan(bad, lab, 16, 3, 256, splicer=sp.make_omnilmm_splicer(*tokens))


def test_omnilmm_config_json_round_trip():
    """config.json of an OmniLMM checkpoint: the fields the reference's initialize_vision_modules sets (omnilmm.py:76-80)
    plus what is needed to rebuild OmniLMMConfig; a Mistral sliding window of 4096 is written because the path never
    exceeds it (model_max_length 2048)."""
    import importlib
    ck = importlib.import_module("rlaif-v_amd.checkpoint")
    om = importlib.import_module("rlaif-v_amd.omnilmm")
    cfg = om.OmniLMMConfig(layers=3, model_max_length=1024, im_patch_token=32003)
    d = ck.hf_config_dict(cfg)
    assert d["model_type"] == "omnilmm" and d["num_query"] == 64 and d["image_size"] == 448 and d["vocab_size"] == 32009
    assert d["num_key_value_heads"] == 8 and d["intermediate_size"] == 14336
    back = ck.config_from_hf(d)
    assert isinstance(back, om.OmniLMMConfig) and back == cfg and back.vocab_padded == 32064
    # the LLaVA dictionary is untouched
    lc = importlib.import_module("rlaif-v_amd.model").LlavaConfig()
    assert ck.hf_config_dict(lc)["model_type"] == "llava_llama" and ck.config_from_hf(ck.hf_config_dict(lc)) == lc


def test_omni_preprocess_matches_reference_golden(golden_dir):
    """omni_preprocess (OmniLMM chat-template encoding + response-only labels, omnilmm/train/train_utils.py:50-151) and the
    chat.py wrappers vs the reference's OWN function over the toy tokenizer (tests/golden/make_omni_preprocess_golden.py):
    single turn with an image span, two rounds, a trailing question, and a text truncated before any assistant marker."""
    import copy
    import warnings
    sys.path.insert(0, golden_dir)
    from toy_tokenizer import OMNI_CONVERSATIONS, OmniToyTokenizer
    from rlaif_v_amd.omni_data import expand_question_into_multimodal, omni_preprocess, wrap_question_for_omni_lmm
    gold = torch.load(os.path.join(golden_dir, "omni_preprocess.pt"), weights_only=False)
    tok = OmniToyTokenizer()
    for mode, gen in (("train", False), ("generation", True)):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            d = omni_preprocess([copy.deepcopy(c) for c in OMNI_CONVERSATIONS], tok, generation=gen)
        for got_ids, got_lab, ref in zip(d["input_ids"], d["labels"], gold[mode]):
            assert torch.equal(got_ids, ref["input_ids"]) and torch.equal(got_lab, ref["labels"]), mode
        assert any("Could not find key" in str(x.message) for x in w)         # the truncated sample warns, like the reference
    # labels cover exactly the answers: sample 1 has two answers, sample 2 drops the trailing question
    lab = d["labels"][1]
    assert int((gold["train"][1]["labels"] != -100).sum()) == 8 + 8 and int((lab != -100).sum()) >= 16
    # chat.py surface: <image> tag -> <im_start> + patches + <im_end>, generation prompt appended
    q = expand_question_into_multimodal([{"role": "user", "content": "<image>\nWhat ?"}], 3, "<im_start>", "<im_end>", "<im_patch>")
    assert q[0]["content"] == "<im_start><im_patch><im_patch><im_patch><im_end>\nWhat ?"
    q = expand_question_into_multimodal([{"role": "user", "content": "What ?"}], 2, "<im_start>", "<im_end>", "<im_patch>")
    assert q[0]["content"] == "<im_start><im_patch><im_patch><im_end>\nWhat ?"
    out = wrap_question_for_omni_lmm("What is this ?", 4, tok)
    ids = out["input_ids"].tolist()
    assert ids.count(9) == 4 and ids[-2:] == [6, 3]                            # 4 patches; ends with "<|assistant|>\n"
    with pytest.raises(AssertionError):
        omni_preprocess([[{"role": "user", "content": "a"}, {"role": "user", "content": "b"}]], tok)


def test_full_depth_harness_compare_logic():
    """tests/full_depth.py::compare (the checker of the full-depth fixtures) on synthetic records: a HIP record equal to the
    fixture up to bf16-level noise passes every bar; a wrong target id, a 1 % loss error, a rotated gradient or a sign-flipped
    update each trip their own assertion.  (The real records only exist on the GPU box.)"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import copy
    import full_depth as FD
    g = torch.Generator().manual_seed(0)
    S, L = 8, 40
    labels = torch.full((S, L), -100, dtype=torch.int64)
    labels[:, 20:36] = torch.randint(3, 500, (S, 16), generator=g)
    mask = labels[:, 1:] != -100
    per_tok = -11.0 + torch.randn(int(mask.sum()), generator=g)
    seq = torch.stack([per_tok[i * 16:(i + 1) * 16].sum() for i in range(S)])
    lr, clip = 5e-7, 1.2e-3
    names = ["model.layers.0.mlp.down_proj.weight", "model.norm.weight"]
    gs = {k: torch.randn(FD.N_SAMPLE, generator=g) * 1e-2 for k in names}
    w0 = {k: torch.randn(4096, generator=g) * 0.02 for k in names}
    def adam1(k, grad, c):          # AdamW step 1 on the sampled elements (decay on the projection, none on the norm gain)
        p0, gc = w0[k][FD.sample_index(k, 4096)].double(), grad.double() * c
        return (p0 * ((1 - lr * 0.01) if O.is_decay_param(k) else 1.0) - lr * gc / (gc.abs() + 1e-8)).float()

    post = {k: adam1(k, gs[k], clip) for k in names}
    fx = dict(case="cfg1_step", layers=32, labels=labels, per_token=per_tok, log_prob=seq, loss=13.78,
              emu_per_token=per_tok + 0.04 * torch.randn(per_tok.shape, generator=g), emu_log_prob=seq * 1.0002, emu_loss=13.79,
              grad_norms={k: float(v.norm()) for k, v in gs.items()}, grad_samples=gs, post_samples=post,
              grad_norm_total=811.0, clip_coef=clip, lr=lr)
    noise = lambda t, s: t + s * torch.randn(t.shape, generator=g)                      # noqa: E731
    hip_gs = {k: noise(v, 3e-4) for k, v in gs.items()}
    hip = dict(tgt=labels[:, 1:][mask], seq_cnt=mask.sum(1).float(), log_prob=seq * (1 + 1e-4), per_token=noise(per_tok, 0.02),
               loss=13.78 * (1 + 2e-4), grad_norms={k: v * 1.002 for k, v in fx["grad_norms"].items()},
               grad_samples=hip_gs, grad_norm_total=811.5, clip_coef=clip * 0.9995,
               post_samples={k: adam1(k, hip_gs[k], clip * 0.9995) for k in names},      # the path's OWN gradient through AdamW
               m_samples={k: 0.1 * clip * v for k, v in hip_gs.items()})
    m = FD.compare("cfg1_step", hip, fx, W0=w0, check=True)
    assert m["indexing_bit_exact"] and m["master_update_agree_frac_large_grads"] > 0.99 and m["grad_worst_sample_cosine"] > 0.999
    assert m["optimizer_self_consistency_frac"] == 1.0
    bad = copy.deepcopy(hip); bad["tgt"] = hip["tgt"].clone(); bad["tgt"][3] += 1
    with pytest.raises(AssertionError, match="indexing"):
        FD.compare("cfg1_step", bad, fx, W0=w0)
    bad = copy.deepcopy(hip); bad["loss"] = 13.78 * 1.01
    with pytest.raises(AssertionError):
        FD.compare("cfg1_step", bad, fx, W0=w0)
    bad = copy.deepcopy(hip); bad["grad_samples"][names[0]] = torch.randn(FD.N_SAMPLE, generator=g) * 1e-2
    with pytest.raises(AssertionError):
        FD.compare("cfg1_step", bad, fx, W0=w0)
    bad = copy.deepcopy(hip); bad["post_samples"][names[0]] = w0[names[0]][FD.sample_index(names[0], 4096)] + lr * torch.sign(gs[names[0]])
    with pytest.raises(AssertionError):
        FD.compare("cfg1_step", bad, fx, W0=w0)


def test_lora_and_skinny_dispatch_conditions(monkeypatch):
    """Which C-ABI entry points the LoRA input gradient under adapter dropout takes for which shapes (no GPU: the launches are recorded,
    not made): the one-pass adapter-first GEMM only when the 256-tile NN kernel can take the shape and the chip is filled, the plain
    input gradient + rv_gemm_nt_dropout_bf16 otherwise or with RV_LORA_DGRAD_PRE=0; forward fused-LoRA projections stay on the in-ring
    form unless RV_LORA_FWD_PRE=1."""
    from rlaif_v_amd import hip, ops
    calls = []
    monkeypatch.setattr(hip, "call", lambda name, *a: calls.append(name))
    monkeypatch.setattr(ops, "_chk2d", lambda t, name: None)            # the wrappers insist on CUDA tensors; here nothing is launched
    bf = torch.bfloat16

    def dgrad(M, K, N, K2):
        calls.clear()
        dy, w, dt, a = (torch.zeros(M, K, dtype=bf), torch.zeros(K, N, dtype=bf), torch.zeros(M, K2, dtype=bf), torch.zeros(K2, N, dtype=bf))
        out = ops.lora_dgrad_dropout(dy, w, w.t().contiguous(), dt, a, a.t().contiguous(), 0.05, 7)
        assert out.shape == (M, N)
        return list(calls)
    monkeypatch.delenv("RV_LORA_DGRAD_PRE", raising=False)
    assert dgrad(3072, 512, 4096, 64) == ["rv_gemm_nn_lora_pre_bf16"]                   # 12 x 16 = 192 tiles: fills the chip
    assert dgrad(3072, 512, 4096, 192) == ["rv_gemm_nn_lora_pre_bf16"]
    two = dgrad(2816, 512, 4096, 64)                                                     # 11 x 16 tiles: too few
    assert two[-1] == "rv_gemm_nt_dropout_bf16" and two[0] in ("rv_gemm_nn_bf16", "rv_gemm_nt_bf16") and len(two) == 2
    assert dgrad(3072, 448, 4096, 64)[-1] == "rv_gemm_nt_dropout_bf16"                   # K < 512
    assert dgrad(3072, 544, 4096, 64)[-1] == "rv_gemm_nt_dropout_bf16"                   # K % 64 != 0
    assert dgrad(3072, 512, 4096, 96)[-1] == "rv_gemm_nt_dropout_bf16"                   # K2 % 64 != 0
    monkeypatch.setenv("RV_LORA_DGRAD_PRE", "0")
    assert dgrad(3072, 512, 4096, 64)[-1] == "rv_gemm_nt_dropout_bf16"
    monkeypatch.delenv("RV_LORA_DGRAD_PRE")

    def fwd(M, K, N, gc):
        calls.clear()
        G = N // gc if gc else 1
        x, w, a2, b2 = (torch.zeros(M, K, dtype=bf), torch.zeros(N, K, dtype=bf), torch.zeros(M, 64 * G, dtype=bf), torch.zeros(N, 64, dtype=bf))
        ops.linear_lora(x, w, w.t().contiguous(), a2, b2, b2.t().contiguous(), group_cols=gc)
        return list(calls)
    monkeypatch.delenv("RV_LORA_FWD_PRE", raising=False)
    assert fwd(3072, 512, 4096, 2048) == ["rv_gemm_nn_lora_bf16"]
    assert fwd(256, 512, 4096, 2048) == ["rv_gemm_nt_lora_bf16"]
    monkeypatch.setenv("RV_LORA_FWD_PRE", "1")
    assert fwd(3072, 512, 4096, 2048) == ["rv_gemm_nn_lora_pre_bf16"]
