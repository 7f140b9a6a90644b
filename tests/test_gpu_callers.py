"""The CALLER's conditions around the hot path ("what's missing" item 2): HF Trainer runs the towers inside
``torch.autocast("cuda", torch.bfloat16)`` under ``DistributedDataParallel`` (``find_unused_parameters=True`` for the CLIP pretraining,
Preprint/LaMed/src/train/train_CLIP_stage1.py:88-92; ``False`` for the VLM, train_VLM.py:147-153) with ``torch.optim.AdamW``
(optim "adamw_torch", :85), and the MLLM finetune keeps the dual tower FROZEN (``requires_grad_(False)``, lamed_arch.py:66) but in
``train()`` while the two packers train (lamed_arch.py:64-76, 122-141).  These tests run the drop-in module API exactly like that —
2 ranks on cuda:0 over gloo — and compare with a single process on the global batch.

What the autocast context does to this API (documented in INTEGRATION.md): nothing inside the modules — their forward is a
torch.autograd.Function over the C ABI, which autocast does not touch; the bf16 policy is the library's own (set_precision("bf16"), the
default).  Outputs are fp32 (the reference's autocast gives bf16 Linear outputs promoted by the fp32 residual stream; its
``encode_images`` result is bf16 and is cast when spliced into the embeddings, lamed_arch.py:153-154 — here the splice receives fp32)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

IMG, PATCH, LAYERS = (8, 64, 64), (4, 16, 16), 2
GRID, KERN, OUT = (2, 4, 4), (1, 2, 2), 256


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------------------------------------------------------------------
# CLIP stage 2: ViT_stage2 + mm_vision_proj (+ the frozen stage-1 CLIP) under DDP(find_unused_parameters=True) + autocast + AdamW
# ------------------------------------------------------------------------------------------------------------------------------
def _clip_model(dev):
    from hsenet_amd.clip import M3DCLIPVision_stage2
    from oracle import clip_cpu as RC, ref_cpu as R
    ref = RC.M3DCLIPVision_stage2(IMG, PATCH, LAYERS)
    sd = R.seeded_state_dict(ref, 187)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)))
    sd["stage1_pretrained_CLIP.logit_scale"] = sd["logit_scale"].clone()
    m = M3DCLIPVision_stage2(img_size=IMG, patch_size=PATCH, num_layers=LAYERS)
    m.load_state_dict(sd)
    return m.to(dev).eval()                      # eval(): the scorer's Dropout(0.1) would draw different masks per rank and per batch split


def _clip_inputs(n):
    from oracle import clip_cpu as RC, ref_cpu as R
    return (R.synthetic_volume((n, 1) + IMG, 188), R.synthetic_slices(n, 189), RC.synthetic_text_features(n, 190),
            RC.synthetic_text_features(n, 191))


def _clip_steps(model, fwd, x, s2d, t1, t2, steps):
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.1)
    losses = []
    for it in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = fwd(x, t1, t2, s2d, global_step=it)
        assert out["loss"].dtype == torch.float32 and out["logits"].dtype == torch.float32
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    return losses


def _clip_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        dev = torch.device("cuda:0")
        model = _clip_model(dev)
        ddp = DDP(model, find_unused_parameters=True)                 # train_CLIP_stage1.py:88-89
        x, s2d, t1, t2 = (t[rank:rank + 1].to(dev) for t in _clip_inputs(world))
        losses = _clip_steps(model, ddp, x, s2d, t1, t2, 2)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "losses": losses}, out_path)
    finally:
        dist.destroy_process_group()


def test_clip_stage2_module_under_ddp_autocast_adamw(dev, tmp_path):
    out_path = str(tmp_path / "ddp_clip.pt")
    mp.spawn(_clip_worker, args=(2, _port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    model = _clip_model(dev)
    before = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, s2d, t1, t2 = (t.to(dev) for t in _clip_inputs(2))
    losses = _clip_steps(model, model, x, s2d, t1, t2, 2)
    torch.cuda.synchronize()
    single = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert list(got["sd"].keys()) == list(single.keys())               # the state-dict order survives DDP (train_VLM.py:477-503 copies by position)
    assert abs(got["losses"][0] - losses[0]) < 2e-3 * abs(losses[0]), (got["losses"], losses)
    moved = diff = 0.0
    for k in single:
        if k.startswith("stage1_pretrained_CLIP."):
            assert torch.equal(got["sd"][k], before[k]) and torch.equal(single[k], before[k]), k      # frozen (CLIP_stage2.py:88)
            continue
        moved += float((single[k] - before[k]).double().pow(2).sum())
        diff += float((single[k] - got["sd"][k]).double().pow(2).sum())
    moved, diff = moved ** 0.5, diff ** 0.5
    print(f"DDP(2 ranks) + autocast + AdamW vs one process: parameters moved {moved:.3e}, differ by {diff:.3e}")
    # AdamW normalises every gradient element to ~lr: bf16-level differences in tiny gradient elements flip their sign, so the parameter
    # difference is a few percent of the movement (the engine's 2-rank test uses the same bound)
    assert moved > 0 and diff < 5e-2 * moved


# ------------------------------------------------------------------------------------------------------------------------------
# MLLM finetune, vision side: frozen dual tower in train() + two trainable packers under DDP(find_unused_parameters=False) + autocast
# ------------------------------------------------------------------------------------------------------------------------------
class _Cfg:
    vision_tower = "vit_stage2_dual_encoders"
    vision_select_layer, vision_select_feature, image_channel = -1, "patch", 1
    remain_2d3d_ViT_type = "dual_vits"
    image_size, patch_size = IMG, PATCH


class _VisionSide(torch.nn.Module):
    """what LamedMetaModel owns on the vision side (lamed_arch.py:64-76): tower (frozen), mm_projector, mm_projector2."""

    def __init__(self):
        super().__init__()
        import hsenet_amd as H
        from oracle import ref_cpu as R
        self.vision_tower = H.build_vision_tower(_Cfg())
        for t in (self.vision_tower.vision_tower_stage1, self.vision_tower.vision_tower_stage2):
            del t.blocks[LAYERS:]
        pargs = (IMG, PATCH, 768, OUT, "mlp", 2)
        self.mm_projector = H.VisualPacker_3d_phi_v3(*pargs, grid=GRID, kernel_size=KERN)
        self.mm_projector2 = H.VisualPacker_3d_phi_v3(*pargs, grid=GRID, kernel_size=KERN)
        rt = R.ViT3DTower_dual_encoders(_Cfg())
        for t in (rt.vision_tower_stage1, rt.vision_tower_stage2):
            del t.blocks[LAYERS:]
        self.vision_tower.load_state_dict(R.seeded_state_dict(rt, 201))
        self.mm_projector.load_state_dict(R.seeded_state_dict(R.VisualPacker_3d_phi_v3(*pargs, grid=GRID, kernel_size=KERN), 202))
        self.mm_projector2.load_state_dict(R.seeded_state_dict(R.VisualPacker_3d_phi_v3(*pargs, grid=GRID, kernel_size=KERN), 203))
        self.vision_tower.requires_grad_(False)                        # lamed_arch.py:66 — and NO .eval(): the tower stays in train()

    def forward(self, images, images_2d):
        import hsenet_amd as H
        return H.encode_images(self.vision_tower, self.mm_projector, self.mm_projector2, images, images_2d)


def _vlm_inputs(n):
    from oracle import ref_cpu as R
    w = torch.from_numpy(np.random.default_rng(204).standard_normal((n, 16, OUT), dtype=np.float32))      # 2 packers x 8 cells
    return R.synthetic_volume((n, 1) + IMG, 205), R.synthetic_slices(n, 206), w


def _vlm_steps(side, fwd, x, s2d, w, steps):
    opt = torch.optim.AdamW([p for p in side.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.0)
    outs = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = fwd(x, s2d)
            loss = (out.float() * w).sum() / w.shape[0]                # stands in for the LM loss through the spliced embeddings
        loss.backward()
        opt.step()
        outs.append(out.detach())
    return outs


def _vlm_worker(rank, world, port, out_path, train_mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        dev = torch.device("cuda:0")
        side = _VisionSide().to(dev)
        side.train(train_mode)
        ddp = DDP(side, find_unused_parameters=False)                  # train_VLM.py:147-149
        x, s2d, w = (t[rank:rank + 1].to(dev) for t in _vlm_inputs(world))
        _vlm_steps(side, ddp, x, s2d, w, 2)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({k: v.cpu() for k, v in side.state_dict().items()}, out_path)
    finally:
        dist.destroy_process_group()


def test_frozen_tower_trainable_packers_under_ddp_autocast(dev, tmp_path):
    # (1) the reference's actual condition: tower frozen but in train() — its Dropout(0.1) sites fire
    side = _VisionSide().to(dev)
    side.train()
    assert side.vision_tower.training and not any(p.requires_grad for p in side.vision_tower.parameters())
    before = {k: v.detach().clone() for k, v in side.state_dict().items()}
    x, s2d, w = (t.to(dev) for t in _vlm_inputs(2))
    torch.manual_seed(3)
    o1 = _vlm_steps(side, side, x, s2d, w, 1)[0]
    assert o1.dtype == torch.float32 and tuple(o1.shape) == (2, 2 * side.mm_projector.proj_out_num, OUT) and bool(torch.isfinite(o1).all())
    after = side.state_dict()
    for k in before:
        same = torch.equal(after[k], before[k])
        assert same == k.startswith("vision_tower."), k                # only the packers moved
    assert all(p.grad is None for p in side.vision_tower.parameters())
    side2 = _VisionSide().to(dev)
    side2.train()
    torch.manual_seed(4)
    with torch.no_grad():
        a = side2(x, s2d)
        b = side2(x, s2d)
    assert not torch.equal(a, b)                                       # train(): the frozen tower's dropout masks change from call to call
    # (2) the DDP arithmetic, with dropout off so that 2 x 1 volume and 1 x 2 volumes are the same function
    out_path = str(tmp_path / "ddp_vlm.pt")
    mp.spawn(_vlm_worker, args=(2, _port(), out_path, False), nprocs=2, join=True)
    got = torch.load(out_path)
    side3 = _VisionSide().to(dev)
    side3.eval()
    start = {k: v.detach().cpu().clone() for k, v in side3.state_dict().items()}
    _vlm_steps(side3, side3, x, s2d, w, 2)
    torch.cuda.synchronize()
    moved = diff = 0.0
    for k, v in side3.state_dict().items():
        v = v.detach().cpu()
        if k.startswith("vision_tower."):
            assert torch.equal(got[k], start[k])
            continue
        moved += float((v - start[k]).double().pow(2).sum())
        diff += float((v - got[k]).double().pow(2).sum())
    moved, diff = moved ** 0.5, diff ** 0.5
    print(f"frozen tower + packers, DDP(2 ranks) vs one process: parameters moved {moved:.3e}, differ by {diff:.3e}")
    assert moved > 0 and diff < 5e-2 * moved
    # (3) the same run with the tower in train() under DDP: completes, finite, tower untouched
    out_path2 = str(tmp_path / "ddp_vlm_train.pt")
    mp.spawn(_vlm_worker, args=(2, _port(), out_path2, True), nprocs=2, join=True)
    got2 = torch.load(out_path2)
    assert all(bool(torch.isfinite(v).all()) for v in got2.values())
    assert all(torch.equal(got2[k], start[k]) for k in got2 if k.startswith("vision_tower."))


# ------------------------------------------------------------------------------------------------------------------------------
# the whole trainable vision side (dual tower + two packers) under DDP over RCCL: the fused node chain forks the 2E3 encoder onto a side
# stream INSIDE its autograd nodes -- to DDP's reducer (bucket copies and all-reduces issued from gradient hooks on the stream of the
# hook) every gradient must be complete on the caller's stream.  World size 1 on the one GPU there is: RCCL's own stream, DDP's buckets
# and hooks all run; the result must equal the same steps without DDP.
# ------------------------------------------------------------------------------------------------------------------------------
class _TrainableSide(torch.nn.Module):
    def __init__(self):
        super().__init__()
        import hsenet_amd as H
        from oracle import ref_cpu as R

        class Cfg:
            vision_tower = "vit_stage2_dual_encoders"
            vision_select_layer, vision_select_feature, image_channel, remain_2d3d_ViT_type = -1, "patch", 1, "dual_vits"
            image_size, patch_size = IMG, PATCH
        self.vision_tower = H.build_vision_tower(Cfg())
        for t in (self.vision_tower.vision_tower_stage1, self.vision_tower.vision_tower_stage2):
            del t.blocks[LAYERS:]
        pk = dict(grid=GRID, kernel_size=KERN)
        self.mm_projector = H.VisualPacker_3d_phi_v3(IMG, PATCH, 768, OUT, "mlp", 2, **pk)
        self.mm_projector2 = H.VisualPacker_3d_phi_v3(IMG, PATCH, 768, OUT, "mlp", 2, **pk)
        for i, m in enumerate((self.vision_tower, self.mm_projector, self.mm_projector2)):
            m.load_state_dict(R.seeded_state_dict(m, 910 + i))

    def forward(self, x, s2d):
        import hsenet_amd as H
        return H.encode_images(self.vision_tower, self.mm_projector, self.mm_projector2, x, s2d)


def _trainable_steps(model, fwd, x, s2d, steps):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.1, fused=True)
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = fwd(x, s2d)
        out.float().square().mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()


def _rccl1_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        from oracle import ref_cpu as R
        model = _TrainableSide().to(dev).eval()
        ddp = DDP(model, device_ids=[0], find_unused_parameters=False, bucket_cap_mb=5)      # small buckets: many all-reduces during the backward
        x, s2d = R.synthetic_volume((2, 1) + IMG, 920).to(dev), R.synthetic_slices(2, 921).to(dev)
        torch.save(_trainable_steps(model, ddp, x, s2d, 3), out_path)
    finally:
        dist.destroy_process_group()


def test_trainable_dual_tower_under_ddp_over_rccl(dev, tmp_path):
    from oracle import ref_cpu as R
    out_path = str(tmp_path / "ddp_rccl1.pt")
    mp.spawn(_rccl1_worker, args=(1, _port(), out_path), nprocs=1, join=True)
    got = torch.load(out_path)
    model = _TrainableSide().to(dev).eval()
    start = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    x, s2d = R.synthetic_volume((2, 1) + IMG, 920).to(dev), R.synthetic_slices(2, 921).to(dev)
    want = _trainable_steps(model, model, x, s2d, 3)
    moved, diff = float((want - start).norm()), float((want - got).norm())
    print(f"trainable dual tower + packers, DDP over RCCL (world 1) vs no DDP: parameters moved {moved:.3e}, differ by {diff:.3e}")
    assert moved > 0 and diff < 5e-2 * moved and bool(torch.isfinite(got).all())
