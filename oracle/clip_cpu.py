"""CPU oracle for the CLIP loss head and the reference's stage-2 CLIP train step (TEST INFRASTRUCTURE ONLY — imported by
tests/, smoke() and bench.py's cpu_baseline leg; nothing under hsenet_amd/ imports it).

Plain-PyTorch fp32 restatement of the vision half of M3DCLIP_stage2 (Preprint/LaMed/src/model/CLIP_stage2.py:58-190) at world
size 1 (gather_features is then the identity, utils/dist_utils.py:280-306), with the BERT text towers' outputs given as inputs
(out of scope; BASELINE configs[3] trains against frozen text embeddings), plus the embedding splice of
prepare_inputs_for_multimodal (lamed_arch.py:151-154).

Pinning (round 5): pinned against the reference classes RUN here.  oracle/make_goldens.py --only clip imports
LaMed.src.model.CLIP_stage1 / CLIP_stage2 as the reference's own package (sys.modules stand-ins for the absent third-party monai /
open_clip exactly as for vit.py; BertModel.from_pretrained -> a random-init one-layer BertModel of the same width because no BERT
weights exist offline; AutoModel.from_pretrained -> the stage-1 instance), feeds the reference's own encode_text outputs to the
classes below with identical seeded vision weights, and asserts loss / both logits / every vision gradient equal (<= 1e-6 / 2e-5)
for stage 1 and for stage 2 at global_step in {0, 1000, 5000, 6000, None}, local_loss in {False, True}, gather_loss in {True,
False}, at world size 1 and -- two gloo processes running the reference -- at world size 2.  The reference's outputs are committed as
tests/golden/clip_ref.npz (tiny geometry) and tests/golden/clip_c4a_256.npz (BASELINE configs[3] at full size: the reference's own
forward + autograd); tests/test_oracle.py re-checks this file against both without the reference present.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ref_cpu as R


def relation_loss_weight(global_step, max_weighted_step=5000):
    """CLIP_stage2.py:141-149 (the bare except makes an unknown step weigh 0)."""
    try:
        if global_step < max_weighted_step:
            return 0.1 * (1 - global_step / max_weighted_step)
        return 0.0
    except Exception:
        return 0.0


def image_text_contrastive_learning(image_features, text_features, logit_scale, local_loss=False, world=1, rank=0,
                                    all_image_features=None, all_text_features=None, gather_loss=True):
    """CLIP_stage1.py:141-155 / CLIP_stage2.py:170-185.  At world 1 the gathered features are the inputs; at world > 1 the caller
    passes the features of every rank (rank order) to restate what rank `rank` computes.  Labels are the collator's
    (train/train_CLIP_stage1.py:146-151): arange(global batch) with gather_loss, arange(local batch) without; with local_loss a
    rank's rows carry rank * B_local + arange(B_local)."""
    ai = image_features if all_image_features is None else all_image_features
    at = text_features if all_text_features is None else all_text_features
    if not gather_loss:                                                                                            # :150-152 / :180-182
        lpi = logit_scale * image_features @ text_features.T
        lpt = logit_scale * text_features @ image_features.T
        labels = torch.arange(image_features.shape[0])
    elif local_loss:
        lpi = logit_scale * image_features @ at.T
        lpt = logit_scale * text_features @ ai.T
        labels = rank * image_features.shape[0] + torch.arange(image_features.shape[0])
    else:
        lpi = logit_scale * ai @ at.T
        lpt = lpi.T
        labels = torch.arange(ai.shape[0])
    loss = (F.cross_entropy(lpi, labels) + F.cross_entropy(lpt, labels)) / 2
    return loss, lpi, lpt


def image_text_relation_regulation(lpi, lpt, lpi_masked, lpt_masked):
    """CLIP_stage2.py:187-190."""
    return (F.mse_loss(lpi, lpi_masked) + F.mse_loss(lpt, lpt_masked)) / 2


class M3DCLIPVision_stage1(nn.Module):
    """CLIP_stage1.py:56-102 minus the text tower."""

    def __init__(self, img_size, patch_size, num_layers=12, gather_loss=True, local_loss=False):
        super().__init__()
        self.vision_encoder = R.ViT_stage1(in_channels=1, img_size=img_size, patch_size=patch_size, num_layers=num_layers,
                                           pos_embed="perceptron", classification=True)
        self.mm_vision_proj = nn.Linear(768, 768)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.gather_loss, self.local_loss = gather_loss, local_loss

    def encode_image(self, image):
        return R.clip_encode_image(self.vision_encoder, self.mm_vision_proj, image)

    def forward(self, images, text_features, global_step=None, world=1, rank=0):
        """CLIP_stage1.py:111-139 with the text tower's cls features given.  world > 1: `images` / `text_features` hold the GLOBAL
        batch in rank order and the result is what rank `rank` computes (its own slice as the local features)."""
        img = self.encode_image(images)[:, 0]
        loss, lpi, lpt = _rank_view(self, img, text_features, self.logit_scale, world, rank)                     # :120
        return {"loss": loss, "logits": (lpi + lpt) / 2.0}


def _rank_view(m, img_all, txt_all, logit_scale, world, rank):
    b = img_all.shape[0] // world
    sl = slice(rank * b, (rank + 1) * b)
    return image_text_contrastive_learning(img_all[sl], txt_all[sl], logit_scale, m.local_loss, world, rank, img_all, txt_all,
                                           m.gather_loss)


class M3DCLIPVision_stage2(nn.Module):
    """CLIP_stage2.py:58-169 minus the text towers: text features are inputs."""

    def __init__(self, img_size, patch_size, num_layers=12, gather_loss=True, local_loss=False):
        super().__init__()
        self.vision_encoder = R.ViT_stage2(in_channels=1, img_size=img_size, patch_size=patch_size, num_layers=num_layers,
                                           pos_embed="perceptron", classification=True)
        self.mm_vision_proj = nn.Linear(768, 768)
        self.stage1_pretrained_CLIP = M3DCLIPVision_stage1(img_size, patch_size, num_layers)
        self.stage1_pretrained_CLIP.requires_grad_(False)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.gather_loss, self.local_loss = gather_loss, local_loss

    def encode_image(self, image, images_2d):
        return R.clip_encode_image(self.vision_encoder, self.mm_vision_proj, image, images_2d)

    def forward(self, images, text_features_stage1, text_features, images_2d, global_step=None, world=1, rank=0):
        """CLIP_stage2.py:119-169.  world > 1: every input holds the GLOBAL batch in rank order; returns rank `rank`'s results."""
        with torch.no_grad():       # the reference uses inference_mode (:124); no_grad keeps the tensors usable below
            img1 = self.stage1_pretrained_CLIP.encode_image(images)[:, 0]
        _, lpi1, lpt1 = _rank_view(self, img1, text_features_stage1, self.logit_scale, world, rank)              # :128 (THIS module's scale and flags)
        img2 = self.encode_image(images, images_2d)[:, 0]
        loss_cl, lpi2, lpt2 = _rank_view(self, img2, text_features, self.logit_scale, world, rank)               # :137
        loss_rel = image_text_relation_regulation(lpi1.detach(), lpt1.detach(), lpi2, lpt2)                     # :139
        w = relation_loss_weight(global_step)
        return {"loss": loss_cl + w * loss_rel, "logits": (lpi2 + lpt2) / 2.0, "loss_CL_stage2": loss_cl, "loss_relation": loss_rel}


def splice_image_features(inputs_embeds, image_features):
    """lamed_arch.py:153-154."""
    n = image_features.shape[1]
    return torch.cat((inputs_embeds[:, :1, :], image_features, inputs_embeds[:, (n + 1):, :]), dim=1)


def synthetic_text_features(batch, seed):
    """frozen text embeddings: normalize(N(0,1)) [B, 768]."""
    import numpy as np
    return F.normalize(torch.from_numpy(np.random.default_rng(seed).standard_normal((batch, 768), dtype=np.float32)), dim=-1)


def c4b_cpu_step(img, patch, x):
    """bench.py's cpu_baseline for --workload clip_c4b (SURVEY section 8(d) "C4-B": both encoders trainable): the stage-1 step and the
    stage-2 step of the reference on the same volume, the stage-1 forward shared (its logits enter the relation term detached)."""
    m = M3DCLIPVision_stage2(img, patch)
    m.stage1_pretrained_CLIP.requires_grad_(True)
    s = R.synthetic_slices(x.shape[0], 1235)
    t1, t2 = synthetic_text_features(x.shape[0], 1236), synthetic_text_features(x.shape[0], 1237)

    def step():
        c4b_losses(m, x, t1, t2, s, 0)[0].backward()
    return step, [m]


def c4b_losses(m, x, t1, t2, s, global_step):
    """(loss_stage1 + loss_stage2, loss_stage1, loss_stage2) of the both-trainable step: stage 1 = CLIP_stage1.py:111-139 on
    m.stage1_pretrained_CLIP (its own logit_scale), stage 2 = CLIP_stage2.py:119-151 with the stage-1 features detached."""
    s1 = m.stage1_pretrained_CLIP
    img1 = s1.encode_image(x)[:, 0]
    loss1, _, _ = image_text_contrastive_learning(img1, t1, s1.logit_scale)
    _, lpi1, lpt1 = image_text_contrastive_learning(img1.detach(), t1, m.logit_scale)                            # CLIP_stage2.py:128
    img2 = m.encode_image(x, s)[:, 0]
    loss_cl, lpi2, lpt2 = image_text_contrastive_learning(img2, t2, m.logit_scale)
    loss_rel = image_text_relation_regulation(lpi1.detach(), lpt1.detach(), lpi2, lpt2)
    loss2 = loss_cl + relation_loss_weight(global_step) * loss_rel
    return loss1 + loss2, loss1, loss2


def c4a_cpu_step(img, patch, x):
    """bench.py's cpu_baseline for --workload clip_c4a: returns (step_fn, modules-to-seed)."""
    m = M3DCLIPVision_stage2(img, patch)
    s = R.synthetic_slices(x.shape[0], 1235)
    t1, t2 = synthetic_text_features(x.shape[0], 1236), synthetic_text_features(x.shape[0], 1237)

    def step():
        m(x, t1, t2, s, global_step=0)["loss"].backward()
    return step, [m]
