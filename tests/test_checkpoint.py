"""CPU tests of the on-disk formats: the CLIP save layout, the substring filters and the
POSITIONAL tower hand-over of train/train_VLM.py:477-503, checked against the oracle's restatement of the reference
modules (same state-dict order) and against a brute-force restatement of the copy loop."""
import os
import types

import pytest
import torch

from hsenet_amd import checkpoint as C

GEOM = dict(in_channels=1, img_size=(8, 32, 32), patch_size=(4, 16, 16), num_layers=2, pos_embed="perceptron", classification=True)


class _FakeClipStage2(torch.nn.Module):
    """shape of M3DCLIP_stage2 (CLIP_stage2.py:60-95): a frozen stage-1 CLIP inside, the 2E3 encoder, heads, scale."""

    def __init__(self):
        super().__init__()
        from oracle import ref_cpu as R
        inner = torch.nn.Module()
        inner.vision_encoder = R.ViT_stage1(**GEOM)
        inner.mm_vision_proj = torch.nn.Linear(768, 768)
        self.stage1_pretrained_CLIP = inner
        self.vision_encoder = R.ViT_stage2(**GEOM)
        self.language_encoder = torch.nn.Linear(4, 4)
        self.mm_vision_proj = torch.nn.Linear(768, 768)
        self.mm_language_proj = torch.nn.Linear(768, 768)
        self.logit_scale = torch.nn.Parameter(torch.ones([]))


def _randomize(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g))


def test_clip_checkpoint_round_trip(tmp_path):
    src = _FakeClipStage2()
    _randomize(src, 1)
    C.save_clip_checkpoint(str(tmp_path), src.state_dict())
    assert sorted(os.listdir(tmp_path)) == [C.SAFE_WEIGHTS_NAME, C.PARAMS_NAME]
    by_bin = C.load_clip_state_dict(str(tmp_path))                                  # prefers model_params.bin
    by_st = C.load_clip_state_dict(os.path.join(tmp_path, C.SAFE_WEIGHTS_NAME))
    assert list(by_bin) == list(src.state_dict())                                   # module order survives the .bin
    assert sorted(by_st) == sorted(by_bin)
    for k, v in src.state_dict().items():
        assert torch.equal(by_bin[k], v) and torch.equal(by_st[k], v)


def test_positional_tower_handover(tmp_path):
    import hsenet_amd as H
    src = _FakeClipStage2()
    _randomize(src, 2)
    C.save_clip_checkpoint(str(tmp_path), src.state_dict())
    vis = C.filter_visual_checkpoint(C.load_clip_state_dict(str(tmp_path)))
    assert not any(s in k for k in vis for s in ("language_encoder", "mm_vision_proj", "mm_language_proj", "logit_scale"))
    assert sum("stage1_pretrained_CLIP" in k for k in vis) == 1 + 3 + 2 * 11 + 2        # ViT_stage1 tensors, L = 2
    assert sum("stage1_pretrained_CLIP" not in k for k in vis) == 1 + 3 + 2 * 11 + 2 + 12

    cfg = types.SimpleNamespace(vision_select_layer=-1, vision_select_feature="patch", image_channel=1, image_size=GEOM["img_size"],
                                patch_size=GEOM["patch_size"], remain_2d3d_ViT_type="dual_vits", vision_tower="vit_stage2_dual_encoders")
    holder = torch.nn.Module()
    holder.vision_tower = H.build_vision_tower(cfg)
    # the reference towers are built with 12 layers; shrink ours to the fixture's depth by rebuilding the stages
    holder.vision_tower.vision_tower_stage1 = H.ViT_stage1(**GEOM)
    holder.vision_tower.vision_tower_stage2 = H.ViT_stage2(**GEOM)
    n1, n2, skipped = C.load_towers_positional(holder, vis)
    assert (n1, n2, skipped) == (28, 40, [])
    # brute-force restatement of train_VLM.py:479-503 on an oracle-built target
    for (_, a), (_, b) in zip(src.stage1_pretrained_CLIP.vision_encoder.state_dict().items(),
                              holder.vision_tower.vision_tower_stage1.state_dict().items()):
        assert torch.equal(a, b)
    for (_, a), (_, b) in zip(src.vision_encoder.state_dict().items(), holder.vision_tower.vision_tower_stage2.state_dict().items()):
        assert torch.equal(a, b)


def test_stage1_only_checkpoint_falls_back_to_vision_encoder_keys():
    import hsenet_amd as H
    from oracle import ref_cpu as R
    clip1 = torch.nn.Module()
    clip1.vision_encoder = R.ViT_stage1(**GEOM)
    clip1.mm_vision_proj = torch.nn.Linear(768, 768)
    _randomize(clip1, 3)
    vis = C.filter_visual_checkpoint(clip1.state_dict())
    holder = torch.nn.Module()
    holder.vision_tower_stage1 = H.ViT_stage1(**GEOM)
    n1, n2, skipped = C.load_towers_positional(holder, vis, "vit_single_3dvit_encoder")
    assert (n1, n2, skipped) == (28, 0, [])
    for (_, a), (_, b) in zip(clip1.vision_encoder.state_dict().items(), holder.vision_tower_stage1.state_dict().items()):
        assert torch.equal(a, b)


def test_shape_mismatch_is_skipped_like_the_reference():
    import hsenet_amd as H
    from oracle import ref_cpu as R
    clip1 = torch.nn.Module()
    clip1.vision_encoder = R.ViT_stage1(**dict(GEOM, img_size=(8, 64, 64)))      # 8 patches instead of 2: pos-embed differs
    holder = torch.nn.Module()
    holder.vision_tower_stage1 = H.ViT_stage1(**GEOM)
    _, _, skipped = C.load_towers_positional(holder, C.filter_visual_checkpoint(clip1.state_dict()), "vit_single_3dvit_encoder")
    assert skipped == ["vision_encoder.patch_embedding.position_embeddings"]


def test_projector_and_lora_checkpoint(tmp_path):
    import hsenet_amd as H
    m = torch.nn.Module()
    m.mm_projector = H.VisualPacker_3d_phi_v3((32, 256, 256), (4, 16, 16), 768, 64, "mlp", 2)
    m.mm_projector2 = H.VisualPacker_3d_phi_v3((32, 256, 256), (4, 16, 16), 768, 64, "mlp", 2)
    m.other = torch.nn.Linear(3, 3)
    m.lora_A = torch.nn.Linear(2, 2)
    _randomize(m, 4)
    f = C.save_projector_and_lora(str(tmp_path), m)
    saved = torch.load(f, weights_only=True)
    assert all(("mm_projector" in k) or ("lora" in k) for k in saved) and not any(k.startswith("other") for k in saved)
    m2 = torch.nn.Module()
    m2.mm_projector = H.VisualPacker_3d_phi_v3((32, 256, 256), (4, 16, 16), 768, 64, "mlp", 2)
    m2.mm_projector2 = H.VisualPacker_3d_phi_v3((32, 256, 256), (4, 16, 16), 768, 64, "mlp", 2)
    m2.other = torch.nn.Linear(3, 3)
    m2.lora_A = torch.nn.Linear(2, 2)
    res = C.load_projector_and_lora(m2, f)
    assert all(k.startswith("other") for k in res.missing_keys) and not res.unexpected_keys
    for k, v in m.state_dict().items():
        if not k.startswith("other"):
            assert torch.equal(m2.state_dict()[k], v)
