"""CPU oracle for the CLIP loss head and the reference's stage-2 CLIP train step (TEST INFRASTRUCTURE ONLY — imported by
tests/, smoke() and bench.py's cpu_baseline leg; nothing under hsenet_amd/ imports it).

Plain-PyTorch fp32 restatement of the vision half of M3DCLIP_stage2 (Preprint/LaMed/src/model/CLIP_stage2.py:58-190) at world
size 1 (gather_features is then the identity, utils/dist_utils.py:280-306), with the BERT text towers' outputs given as inputs
(out of scope; BASELINE configs[3] trains against frozen text embeddings), plus the embedding splice of
prepare_inputs_for_multimodal (lamed_arch.py:151-154).

Pinning: CLIP_stage2.py cannot be imported here (BertModel.from_pretrained weights, the LaMed package path and open_clip at module
scope — ordinary import errors, nothing was denied), so these ~40 lines are restated from the source text and are **parity
unpinned** against the reference class itself; the encoders they call are the pinned oracle classes of ref_cpu.py, and the loss
pieces are checked against torch.nn.functional identities in tests/test_oracle.py.
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
                                    all_image_features=None, all_text_features=None):
    """CLIP_stage2.py:170-185.  At world 1 the gathered features are the inputs; for the local_loss branch the caller may pass
    the features of every rank to restate what rank `rank` computes."""
    ai = image_features if all_image_features is None else all_image_features
    at = text_features if all_text_features is None else all_text_features
    if local_loss:
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

    def __init__(self, img_size, patch_size, num_layers=12):
        super().__init__()
        self.vision_encoder = R.ViT_stage1(in_channels=1, img_size=img_size, patch_size=patch_size, num_layers=num_layers,
                                           pos_embed="perceptron", classification=True)
        self.mm_vision_proj = nn.Linear(768, 768)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def encode_image(self, image):
        return R.clip_encode_image(self.vision_encoder, self.mm_vision_proj, image)

    def forward(self, images, text_features, global_step=None):
        """CLIP_stage1.py:111-139 with the text tower's cls features given."""
        img = self.encode_image(images)[:, 0]
        loss, lpi, lpt = image_text_contrastive_learning(img, text_features, self.logit_scale)                   # :121
        return {"loss": loss, "logits": (lpi + lpt) / 2.0}


class M3DCLIPVision_stage2(nn.Module):
    """CLIP_stage2.py:58-169 minus the text towers: text features are inputs."""

    def __init__(self, img_size, patch_size, num_layers=12):
        super().__init__()
        self.vision_encoder = R.ViT_stage2(in_channels=1, img_size=img_size, patch_size=patch_size, num_layers=num_layers,
                                           pos_embed="perceptron", classification=True)
        self.mm_vision_proj = nn.Linear(768, 768)
        self.stage1_pretrained_CLIP = M3DCLIPVision_stage1(img_size, patch_size, num_layers)
        self.stage1_pretrained_CLIP.requires_grad_(False)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def encode_image(self, image, images_2d):
        return R.clip_encode_image(self.vision_encoder, self.mm_vision_proj, image, images_2d)

    def forward(self, images, text_features_stage1, text_features, images_2d, global_step=None):
        with torch.no_grad():       # the reference uses inference_mode (:124); no_grad keeps the tensors usable below
            img1 = self.stage1_pretrained_CLIP.encode_image(images)[:, 0]
        _, lpi1, lpt1 = image_text_contrastive_learning(img1, text_features_stage1, self.logit_scale)            # :128
        img2 = self.encode_image(images, images_2d)[:, 0]
        loss_cl, lpi2, lpt2 = image_text_contrastive_learning(img2, text_features, self.logit_scale)            # :137
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
