"""Image preprocessing: the oracle against outputs of the real PIL + CLIPImageProcessor chain
(tests/golden/image_preprocess.npz, bit exact), and - on the GPU - the HIP kernels against the oracle (bit exact)."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_image_golden as G  # noqa: E402  (image generator only; no PIL/transformers needed at test time)
from oracle import clip_preprocess_oracle as P  # noqa: E402


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_oracle_matches_pil_and_transformers_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "image_preprocess.npz"))
    for i, (h, w) in enumerate(g["cases"].tolist()):
        img = G.make_image(h, w, i)
        out = P.clip_preprocess(img)
        assert out.dtype == np.float32 and out.shape == (3, 336, 336)
        assert np.array_equal(_sha(out), g[f"sha256_{i}"]), (i, h, w)
        if f"pixel_values_{i}" in g:
            assert np.array_equal(out, g[f"pixel_values_{i}"])
            oh, ow = P.resize_output_size(h, w)
            assert np.array_equal(P.pil_resize_bicubic_u8(img, ow, oh), g[f"resized_{i}"])


def test_coefficient_tables_properties():
    """Fixed-point taps sum to 2^22 (+- rounding), bounds stay inside the source, identity resize is a copy."""
    for n_in, n_out in [(640, 448), (150, 336), (1024, 336), (336, 336), (97, 336)]:
        ks, b, k = P.precompute_coeffs(n_in, 0.0, float(n_in), n_out)
        assert k.shape == (n_out, ks) and b[:, 0].min() >= 0 and (b[:, 0] + b[:, 1]).max() <= n_in
        assert np.abs(k.sum(1) - (1 << P.PRECISION_BITS)).max() <= ks
    img = G.make_image(40, 50, 3)
    assert np.array_equal(P.pil_resize_bicubic_u8(img, 50, 40), img)
    t = P.normalize_table()
    assert t.shape == (3, 256) and np.all(np.diff(t, axis=1) > 0)


def test_product_tap_tables_equal_the_oracle():
    """rlaif-v_amd/image.py builds the fixed-point tap tables itself (the product never imports the oracle): they must
    be the oracle's - and therefore Pillow's - integers, and the lookup table transformers' float32 arithmetic."""
    from rlaif_v_amd import image as img
    for n_in, n_out in [(640, 448), (480, 336), (150, 336), (200, 448), (1024, 336), (97, 336), (411, 1423), (336, 336),
                        (900, 336), (337, 336), (56, 56), (120, 74)]:
        ks, b, k = img._taps(n_in, n_out)
        ks2, b2, k2 = P.precompute_coeffs(n_in, 0.0, float(n_in), n_out)
        assert ks == ks2 and np.array_equal(b, b2) and np.array_equal(k, k2), (n_in, n_out)
    assert np.array_equal(img._norm_table(img.CLIP_MEAN, img.CLIP_STD, 1 / 255), P.normalize_table())
    for h, w in [(480, 640), (640, 480), (336, 336), (97, 411), (1000, 352)]:
        assert img.resize_output_size(h, w, 336) == P.resize_output_size(h, w, 336)
    raw = img.RawImageProcessor()(G.make_image(30, 20, 1))
    assert raw.dtype == np.uint8 and raw.shape == (30, 20, 3)
    with pytest.raises(ValueError):
        img.RawImageProcessor()(np.zeros((4, 4), dtype=np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(range(len(G.CASES))))
def test_hip_preprocess_bit_exact(case, golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from rlaif_v_amd.image import clip_preprocess_batch
    h, w = G.CASES[case]
    img = G.make_image(h, w, case)
    out = clip_preprocess_batch([img, img[:, ::-1].copy()]).cpu().numpy()      # a ragged-free batch of two
    assert out.shape == (2, 3, 336, 336) and out.dtype == np.float32
    assert np.array_equal(out[0], P.clip_preprocess(img))
    assert np.array_equal(out[1], P.clip_preprocess(img[:, ::-1].copy()))
    g = np.load(os.path.join(golden_dir, "image_preprocess.npz"))
    assert np.array_equal(_sha(out[0]), g[f"sha256_{case}"])        # == the real PIL + transformers output


@pytest.mark.gpu
def test_hip_preprocess_ragged_batch_feeds_the_tower():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from rlaif_v_amd.image import clip_preprocess_batch
    imgs = [G.make_image(h, w, 10 + i) for i, (h, w) in enumerate([(480, 640), (900, 337), (336, 336)])]
    out = clip_preprocess_batch(imgs)
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i].cpu().numpy(), P.clip_preprocess(im))


@pytest.mark.gpu
def test_raw_uint8_images_through_the_model_equal_preprocessed_floats():
    """RawImageProcessor -> collator keeps a ragged list -> model resizes on the device: identical log-probs to feeding
    the CLIPImageProcessor-equivalent float tensors."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import dpo_oracle as O
    from rlaif_v_amd.image import RawImageProcessor
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    cfg = O.tiny_cfg()                       # image_size 56, patch 14
    model = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)), with_optimizer=False)
    model.load_state_dict(O.make_weights(cfg, seed=2))
    batch = O.make_synthetic_batch(cfg, 2, 36, 12, seed=4)
    proc = RawImageProcessor(size=cfg.image_size)
    raws = [proc(G.make_image(h, w, 20 + i)) for i, (h, w) in enumerate([(120, 90), (64, 200)])]
    floats = torch.from_numpy(np.stack([P.clip_preprocess(r, size=cfg.image_size) for r in raws]))
    a = model.eval().forward_logps(batch["concatenated_input_ids"], batch["concatenated_labels"], raws, save_for_backward=False)
    b = model.forward_logps(batch["concatenated_input_ids"], batch["concatenated_labels"], floats, save_for_backward=False)
    assert torch.equal(a.per_token_logp, b.per_token_logp)
