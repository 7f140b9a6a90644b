"""GPU parity of the preprocessing kernels against the CPU restatement of the reference chain."""
import numpy as np
import pytest
import torch




@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def synthetic_ct(shape, seed):
    """a blob of soft tissue in air, raw scanner units (so that slope/intercept, the HU clamp and the crop all bite)"""
    rng = np.random.default_rng(seed)
    H, W, D = shape
    yy, xx, zz = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), np.linspace(-1, 1, D), indexing="ij")
    body = (yy / 0.7) ** 2 + (xx / 0.6) ** 2 + (zz / 0.8) ** 2 < 1.0
    vol = np.where(body, 1000.0 + 300.0 * rng.standard_normal(shape), -30.0 + 5.0 * rng.standard_normal(shape))
    vol[H // 2 - 2:H // 2 + 2, W // 2 - 2:W // 2 + 2, :] = 3500.0      # bone / metal: clamps at +200 HU
    return vol.astype(np.float64)


@pytest.mark.parametrize("shape,zs,xys", [((48, 40, 30), 3.0, 1.1), ((64, 64, 21), 1.0, 0.8), ((33, 47, 40), 2.5, 0.7)])
@pytest.mark.gpu
def test_nii_to_volume_matches_reference_chain(dev, shape, zs, xys):
    from hsenet_amd import preprocess as P
    from oracle import prep_cpu as R
    raw = synthetic_ct(shape, 7)
    ref = R.nii_img_to_tensor(raw, 1.0, -1024.0, zs, xys, out_size=(16, 64, 64))
    got = P.nii_to_volume(torch.from_numpy(raw).to(dev), 1.0, -1024.0, zs, xys, out_size=(16, 64, 64)).cpu()
    assert got.shape == ref.shape == (1, 16, 64, 64)
    # float32 trilinear taps summed in a different order than torch's separable kernel: a few ulps of values in [0,1]
    assert (got - ref).abs().max() < 2e-6, (got - ref).abs().max()
    assert got.min() >= 0 and got.max() <= 1


@pytest.mark.parametrize("shape", [(48, 40, 30), (33, 47, 40)])
@pytest.mark.gpu
def test_bimcv_nii_to_volume_matches_reference_chain(dev, shape):
    """the BIMCV-R variant (BIMCV_R/2. BIMCV_R_nii_to_3D_volume_npy_file.py:30-87): float32 voxels already in HU, no slope / intercept,
    no respacing: clamp -> (D,H,W) -> min-max -> crop -> resize."""
    from hsenet_amd import preprocess as P
    from oracle import prep_cpu as R
    raw = (synthetic_ct(shape, 9) - 1024.0).astype(np.float32)
    ref = R.bimcv_nii_to_tensor(raw, out_size=(16, 64, 64))
    got = P.bimcv_nii_to_volume(torch.from_numpy(raw).to(dev), out_size=(16, 64, 64)).cpu()
    assert got.shape == ref.shape == (1, 16, 64, 64)
    assert (got - ref).abs().max() < 2e-6, (got - ref).abs().max()
    assert got.min() >= 0 and got.max() <= 1


@pytest.mark.gpu
def test_pieces(dev):
    from hsenet_amd import preprocess as P
    g = torch.Generator().manual_seed(3)
    v64 = torch.randn(9, 14, 11, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.interpolate(v64[None, None], size=[13, 7, 20], mode="trilinear", align_corners=False)[0, 0]
    got = P.resample(v64.to(dev), (13, 7, 20)).cpu()
    assert (got.double() - ref).abs().max() < 1e-6                       # f64 arithmetic, rounded once to f32
    v32 = v64.float()
    mm = P.minmax(v32.to(dev)).cpu()
    assert mm[0] == v32.min() and mm[1] == v32.max()
    x = torch.zeros(6, 7, 8)
    x[2:5, 1:3, 4:8] = 1.0
    assert P.foreground_bbox(x.to(dev), 0.0).cpu().tolist() == [2, 1, 4, 4, 2, 7]
    assert P.foreground_bbox(torch.zeros(3, 3, 3, device=dev), 0.0).cpu().tolist() == [3, 3, 3, -1, -1, -1]


@pytest.mark.parametrize("k,flips", [(0, 0), (1, 0), (2, 5), (3, 2), (1, 7)])
@pytest.mark.gpu
def test_augment(dev, k, flips):
    from hsenet_amd import preprocess as P
    from oracle import prep_cpu as R
    g = torch.Generator().manual_seed(5)
    img = torch.rand(1, 6, 10, 14, generator=g)
    ref = R.augment(img, k, flips, 1.07, -0.03)
    got = P.augment(img.to(dev), k, flips, 1.07, -0.03).cpu()
    assert got.shape == ref.shape
    assert torch.equal(got, ref), (got - ref).abs().max()                 # one multiply and one add per voxel: bit-exact


def test_draws_follow_the_reference_probabilities():
    from hsenet_amd import preprocess as P
    rng = np.random.RandomState(0)
    draws = [P.draw_augmentation(rng) for _ in range(4000)]
    assert 0.45 < np.mean([d[0] > 0 for d in draws]) < 0.55
    assert 0.07 < np.mean([bool(d[1] & 1) for d in draws]) < 0.13
    assert all(0.9 <= d[2] <= 1.1 and -0.1 <= d[3] <= 0.1 for d in draws)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16", 2e-2), ("f16", 3e-3)])     # (fp16: the biased qkv projection and ln_eps options of the block driver at HS_PREC_F16)
def test_trunk2d_and_slice_features(dev, prec, tol):
    """N4: the 2-D branch (resize to 32x224x224 + frozen BiomedCLIP-shaped ViT trunk) against its CPU restatement."""
    import hsenet_amd as H
    from oracle import trunk2d_cpu as T
    torch.manual_seed(3)
    ref = T.Trunk(depth=2).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn_like(p) * 0.03)
        ref.norm.weight.add_(1.0)
        for b in ref.blocks:
            b.norm1.weight.add_(1.0)
            b.norm2.weight.add_(1.0)
    mine = H.BiomedClipTrunk(depth=2)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev)
    vol = torch.rand(1, 1, 20, 72, 64)
    with torch.no_grad():
        want = T.slice_features(vol, ref, slices=4)
    H.set_precision(prec)
    try:
        got = H.slice_features(vol.to(dev), mine, slices=4).cpu()
    finally:
        H.set_precision("bf16")
    assert got.shape == want.shape == (1, 4, 768)
    err = float((got - want).norm() / want.norm())
    assert err < tol, err
