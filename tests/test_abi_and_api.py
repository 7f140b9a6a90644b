"""CPU tests of the boundary: the shared library loads and exports every symbol include/hsenet_hip.h declares
(no compute calls without a GPU), the ctypes table mirrors the header, the Python mirror of the reference API keeps
its constructor/error/state-dict contract, and the product path fails loudly without a GPU."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hsenet_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", src)) - {"hs_gemm_args", "hs_vit_desc"})


def test_library_exports_every_declared_symbol():
    from hsenet_amd import _lib
    from hsenet_amd.build import build
    lib_path = build()                        # hipcc cross-compiles gfx950 without a GPU
    names = header_functions()
    assert len(names) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hs_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in include/hsenet_hip.h but not exported: {missing}"
    undeclared = sorted(exported - set(names))
    assert not undeclared, f"exported by the library but not declared in include/hsenet_hip.h: {undeclared}"
    lib = _lib.load()
    assert sorted(_lib.SIGNATURES) == names, "hsenet_amd/_lib.py SIGNATURES must mirror the header one to one"
    assert lib.hs_version() >= 1
    names = [lib.hs_prof_name(i).decode() for i in range(lib.hs_prof_count())]
    assert len(names) == len(set(names)) == 20 and names[3] == "attn_fwd" and lib.hs_prof_name(99) == b""
    # one class per HOT kernel function: the dominant class of a run is a kernel, not a sum over a GEMM layout
    assert {"attn_bwd_dkdv", "attn_bwd_dq", "gemm_nt16_plain", "gemm_nt16_bias_resid", "gemm_nt16_bias_gelu", "gemm_nt16_dgelu",
            "gemm_nt16_bias", "gemm_tn_group", "adamw"} <= set(names)


def test_code_object_is_gfx950_only():
    from hsenet_amd.build import LIB
    data = open(LIB, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in data


def test_no_gpu_means_loud_failure():
    import hsenet_amd as H
    from hsenet_amd._lib import HsError
    m = H.ViT_stage1(1, (8, 32, 32), (4, 16, 16), num_layers=1, pos_embed="perceptron", classification=True)
    with pytest.raises(HsError):
        m(torch.zeros(1, 1, 8, 32, 32))
    p = H.VisualPacker_3d_phi_v3((32, 256, 256), (4, 16, 16), 768, 3072, "mlp", 2)
    with pytest.raises(HsError):
        p(torch.zeros(1, 2048, 768))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hsenet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle: product code must not use it"


def test_state_dict_contract(golden):
    import hsenet_amd as H
    from oracle import ref_cpu as R
    kw = dict(in_channels=1, img_size=(32, 256, 256), patch_size=(4, 16, 16), pos_embed="perceptron", classification=True)
    for mine, ref, n in ((H.ViT_stage1(**kw), R.ViT_stage1(**kw), 138), (H.ViT_stage2(**kw), R.ViT_stage2(**kw), 150)):
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys()) and len(a) == n        # order is load-bearing (train_VLM.py:477-503)
        assert all(a[k].shape == b[k].shape for k in a)
        mine.load_state_dict(R.seeded_state_dict(ref, 1))
    g = golden("tiny_encoders.npz")
    kw = dict(in_channels=1, img_size=tuple(g["img"]), patch_size=tuple(g["patch"]), num_layers=int(g["layers"]),
              pos_embed="perceptron", classification=True)
    assert list(H.ViT_stage1(**kw).state_dict().keys()) == list(g["s1_keys"])     # keys recorded from the REFERENCE
    assert list(H.ViT_stage2(**kw).state_dict().keys()) == list(g["s2_keys"])
    gp = golden("packer.npz")
    pk = H.VisualPacker_3d_phi_v3((32, 256, 256), (4, 16, 16), 768, 3072, "mlp", 2)
    assert list(pk.state_dict().keys()) == list(gp["packer_keys"]) and pk.proj_out_num == 128
    pk256 = H.VisualPacker_3d_phi_v3((256,) * 3, (16,) * 3, 768, 3072, "mlp", 2, grid=(16, 16, 16), kernel_size=(2, 4, 4))
    assert pk256.proj_out_num == 128           # 256^3 configs: 2 packers x 128 = 256 tokens


def test_builders_and_error_conventions():
    import hsenet_amd as H

    class C:
        vision_tower = "vit_stage2_dual_encoders"
        vision_select_layer, vision_select_feature, image_channel = -1, "patch", 1
        image_size, patch_size = (32, 256, 256), (4, 16, 16)
        mm_projector_type, mm_hidden_size, hidden_size = "VisualPacker_3d_phi_v3", 768, 3072
        proj_layer_type, proj_layer_num, proj_pooling_type, proj_pooling_size = "mlp", 2, "spatial", 2
    t = H.build_vision_tower(C())
    assert isinstance(t, H.ViT3DTower_dual_encoders) and t.hidden_size == 768 and t.remain_2d3d_ViT_type == "dual_vits"
    assert isinstance(H.build_mm_projector(C()), H.VisualPacker_3d_phi_v3)
    C.vision_tower = "resnet"
    with pytest.raises(ValueError, match="Unknown vision tower"):
        H.build_vision_tower(C())
    C.mm_projector_type = "linear"
    with pytest.raises(ValueError, match="Unknown projector type"):
        H.build_mm_projector(C())
    with pytest.raises(ValueError, match="dropout_rate"):
        H.ViT_stage1(1, (32, 256, 256), (4, 16, 16), dropout_rate=2.0, pos_embed="perceptron")
    with pytest.raises(ValueError, match="divisible"):
        H.ViT_stage2(1, (32, 256, 256), (4, 16, 16), hidden_size=770, pos_embed="perceptron")


def test_wgrad_split_and_bucket_plan():
    from hsenet_amd.functional import _wgrad_split
    from hsenet_amd.engine import bucket_plan
    for shape in ((3072, 768, 16388), (768, 768, 16388), (768, 4096, 32768), (768, 768, 64)):
        s = _wgrad_split(*shape)
        assert 1 <= s <= 32 and s <= (shape[2] + 63) // 64
    names = ["cls_token", "pos"] + [f"blocks.{i}.mlp.linear1.weight" for i in range(12)] + ["norm.weight"]
    offs = [0, 64] + [1000 + 100 * i for i in range(12)] + [2200]
    chunks, bounds = bucket_plan(names, offs, 2264, 12, 4)
    assert chunks == [(12, 9), (9, 6), (6, 3), (3, 0)]
    assert bounds == [(1900, 2264), (1600, 1900), (1300, 1600), (0, 1300)]
    # buckets tile the flat buffer exactly once
    assert sorted(bounds)[0][0] == 0 and all(a[1] == b[0] for a, b in zip(sorted(bounds), sorted(bounds)[1:]))


def test_fused_driver_descriptors_validate_without_a_gpu():
    """hs_embed_* / hs_packer_* / hs_vit_* reject bad descriptors and null pointers with error codes before touching the device
    (no compute call is made here), and their workspace sizes follow the geometry."""
    import ctypes as C
    from hsenet_amd import _lib as L
    lib = L.load()
    e = L.EmbedDesc(2, 4096, 4096, 768, 1, 1, 32, 1, 0.1)
    big = lib.hs_embed_workspace_bytes(C.byref(e))
    e0 = L.EmbedDesc(2, 4096, 4096, 768, 1, 0, 32, 1, 0.0)
    small = lib.hs_embed_workspace_bytes(C.byref(e0))
    assert big > small > 2 * 4096 * 768 * 4                       # the scorer's activations dominate; stage-1 keeps pe + dpe
    e_eval = L.EmbedDesc(2, 4096, 4096, 768, 1, 1, 32, 0, 0.0)
    assert lib.hs_embed_workspace_bytes(C.byref(e_eval)) < big       # forward-only / eval(): no backward scratch, no dropout maps
    for bad in (L.EmbedDesc(0, 4096, 4096, 768, 1, 1, 32, 1, 0.0), L.EmbedDesc(2, 4096, 4095, 768, 1, 0, 32, 1, 0.0),
                L.EmbedDesc(2, 4096, 4096, 770, 1, 0, 32, 1, 0.0), L.EmbedDesc(2, 4096, 4096, 768, 1, 1, 30, 1, 0.0),
                L.EmbedDesc(2, 4096, 4096, 768, 1, 1, 32, 1, 1.5)):
        assert lib.hs_embed_workspace_bytes(C.byref(bad)) == 0
        assert lib.hs_embed_fwd(C.byref(bad), 1, 1, 1, 1, 1, 0, None) < 0
    assert lib.hs_embed_fwd(C.byref(e), None, None, None, None, None, 0, None) == L.HS_ERR_ARG
    assert lib.hs_embed_bwd(C.byref(e_eval), 1, 1, 1, 1, 1, 0, None) == L.HS_ERR_ARG      # backward needs save_for_bwd
    T, d, out = 4096, 768, 3072
    k = L.PackerDesc(2, 16, 16, 16, 2, 4, 4, d, out, 1, (T + 1) * d, 2 * 128 * out, 1, 0.0)
    assert lib.hs_packer_workspace_bytes(C.byref(k)) > 2 * T * 2 * d * 4
    for bad in (L.PackerDesc(2, 16, 16, 16, 4, 8, 8, d, out, 1, (T + 1) * d, 2 * 128 * out, 1, 0.0),       # 256 children > 64
                L.PackerDesc(2, 16, 16, 16, 2, 4, 4, d, out, 1, T * d, 2 * 128 * out, 1, 0.0),             # batch stride too short for the row offset
                L.PackerDesc(2, 16, 16, 16, 2, 4, 4, d, out, 1, (T + 1) * d, 100, 1, 0.0),                 # output stride < cells * out_dim
                L.PackerDesc(2, 16, 16, 16, 2, 4, 4, d, 3070, 1, (T + 1) * d, 2 * 128 * 3070, 1, 0.0)):
        assert lib.hs_packer_workspace_bytes(C.byref(bad)) == 0
        assert lib.hs_packer_fwd(C.byref(bad), 1, 1, 1, 1, 0, None) < 0
    assert lib.hs_packer_fwd(C.byref(k), None, None, None, None, 0, None) == L.HS_ERR_ARG
    ws1 = lib.hs_vit_workspace_bytes(C.byref(L.VitDesc(1, 4097, 768, 12, 3072, 12, L.PREC_BF16, 1, 0.0, None)))
    ws2 = lib.hs_vit_workspace_bytes(C.byref(L.VitDesc(2, 4097, 768, 12, 3072, 12, L.PREC_BF16, 1, 0.0, None)))
    assert 1.0e9 < ws2 - ws1 < 1.4e9                                # ~1.2 GB of saved activations per 256^3 volume
    fixed = 2 * ws1 - ws2                                           # batch-independent part: the bf16 weight copies (W and W^T), 2 x 170 MB
    assert 3.3e8 < fixed < 3.6e8
    fwd_only = lib.hs_vit_workspace_bytes(C.byref(L.VitDesc(1, 4097, 768, 12, 3072, 12, L.PREC_BF16, 0, 0.0, None)))
    assert fwd_only < 0.45e9                                        # forward-only: one layer's activations + the W copies, no W^T
    assert lib.hs_vit_workspace_bytes(C.byref(L.VitDesc(1, 4097, 768, 11, 3072, 12, L.PREC_BF16, 1, 0.0, None))) == 0   # head dim != 64
    assert lib.hs_sumsq(None, None, None, 8, None) == L.HS_ERR_ARG and lib.hs_rows_copy(None, 0, None, None, 0, 1, 1, 4, None) == L.HS_ERR_ARG
