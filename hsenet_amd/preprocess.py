"""CT volume preprocessing on the GPU: the chain the reference runs once per scan on
the host before the encoder ever sees a volume, and its train-time augmentation.

    nii_to_volume  <->  nii_img_to_tensor (Data/data_processing/CT-RATE/CT-RATE_nii_to_3D_volume_npy_file.py:39-115):
        raw (H,W,D) -> slope*x+intercept -> clamp [-1000,200] HU -> (D,H,W) -> trilinear respacing to (1.5, 0.75, 0.75) mm
        in float64 (resize_array :25-37) -> float32 -> min-max -> CropForeground -> Resize (32,256,256)   (:118-121)
    augment        <->  train_transform (LaMed/src/dataset/multi_dataset.py:148-158): RandRotate90(H,W), RandFlip x3,
        RandScaleIntensity(0.1), RandShiftIntensity(0.1); the random draws are made on the host
        (`draw_augmentation`), the voxel work is one gather kernel.

Everything runs through the C ABI (hs_prep_*); there is no CPU fallback.  MONAI's CropForeground / Resize / Rand* are a
third-party dependency that is absent here: they are restated from their published semantics
(bounding box of x > 0, margin 0; torch trilinear interpolate with align_corners=False; np.rot90 / flip / v*(1+f) / v+o).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .functional import _need, _s

HU_MIN, HU_MAX = -1000.0, 200.0                      # CT-RATE_nii_to_3D_volume_npy_file.py:79
TARGET_SPACING = (1.5, 0.75, 0.75)                   # (z, x, y) mm, :66-72
OUT_SIZE = (32, 256, 256)                            # mtf.Resize(spatial_size=[32, 256, 256]), :120


def resample(vol: torch.Tensor, out_size: Sequence[int], window: Optional[Tuple[Sequence[int], Sequence[int]]] = None,
             sub: float = 0.0, den: float = 1.0) -> torch.Tensor:
    """Trilinear (align_corners=False) resize of a (D,H,W) float32/float64 volume, optionally of a crop window
    (origin, size), with v -> (v - sub)/den applied to the taps.  Returns float32 (oD,oH,oW)."""
    _need(vol)
    if vol.dim() != 3 or vol.dtype not in (torch.float32, torch.float64):
        raise L.HsError("resample expects a (D,H,W) float32/float64 tensor")
    vol = vol.contiguous()
    iD, iH, iW = vol.shape
    (c0, cs) = window if window is not None else ((0, 0, 0), (iD, iH, iW))
    out = torch.empty(tuple(int(v) for v in out_size), dtype=torch.float32, device=vol.device)
    L.call("hs_prep_resample", vol.data_ptr(), int(vol.dtype == torch.float64), out.data_ptr(), iD, iH, iW, *[int(v) for v in c0],
           *[int(v) for v in cs], *out.shape, float(sub), float(den), _s())
    return out


def minmax(x: torch.Tensor) -> torch.Tensor:
    _need(x)
    out = torch.tensor([float("inf"), float("-inf")], dtype=torch.float32, device=x.device)
    xc = x.contiguous()
    L.call("hs_prep_minmax", xc.data_ptr(), xc.numel(), out.data_ptr(), _s())
    return out


def foreground_bbox(x: torch.Tensor, thresh: float) -> torch.Tensor:
    """{min d,h,w, max d,h,w} (inclusive) of x > thresh for a (D,H,W) float32 volume; (D,H,W,-1,-1,-1) if empty."""
    _need(x)
    D, H, W = x.shape
    box = torch.tensor([D, H, W, -1, -1, -1], dtype=torch.int32, device=x.device)
    L.call("hs_prep_foreground_bbox", x.contiguous().data_ptr(), D, H, W, float(thresh), box.data_ptr(), _s())
    return box


def nii_to_volume(raw: torch.Tensor, slope: float, intercept: float, z_spacing: float, xy_spacing: float,
                  target_spacing: Sequence[float] = TARGET_SPACING, out_size: Sequence[int] = OUT_SIZE) -> torch.Tensor:
    """raw: the NIfTI voxel array (H, W, D) as float32/float64 on the GPU.  Returns the (1, 32, 256, 256) float32 volume
    the reference saves as .npy (normalised to [0,1], foreground-cropped, resized)."""
    _need(raw)
    if raw.dim() != 3 or raw.dtype not in (torch.float32, torch.float64):
        raise L.HsError("nii_to_volume expects the (H,W,D) voxel array as float32/float64")
    raw = raw.contiguous()
    H, W, D = raw.shape
    hu = torch.empty(D, H, W, dtype=torch.float64, device=raw.device)
    L.call("hs_prep_hu_window", raw.data_ptr(), int(raw.dtype == torch.float64), hu.data_ptr(), H, W, D, float(slope),
           float(intercept), HU_MIN, HU_MAX, _s())
    current = (z_spacing, xy_spacing, xy_spacing)
    new_shape = [int(s * (current[i] / target_spacing[i])) for i, s in enumerate((D, H, W))]      # resize_array :30-35
    if min(new_shape) < 1:
        raise L.HsError(f"respacing {tuple(current)} -> {tuple(target_spacing)} mm leaves an empty volume {new_shape}")
    vol = resample(hu, new_shape)                                        # float64 arithmetic, float32 result (:95)
    mm = minmax(vol).cpu()                                               # two scalars decide the launch geometry below
    mn, mx = float(mm[0]), float(mm[1])
    den = max(np.float32(mx) - np.float32(mn), np.float32(1e-8))         # torch.clamp(max - min, min=1e-8) in float32
    box = foreground_bbox(vol, mn).cpu().tolist()                        # normalised > 0  <=>  raw > min
    if box[3] < 0:                                                        # constant volume: MONAI keeps the full extent
        box = [0, 0, 0, new_shape[0] - 1, new_shape[1] - 1, new_shape[2] - 1]
    origin, size = box[:3], [box[3 + a] - box[a] + 1 for a in range(3)]
    return resample(vol, out_size, (origin, size), sub=mn, den=float(den)).unsqueeze(0)


def bimcv_nii_to_volume(raw: torch.Tensor, out_size: Sequence[int] = OUT_SIZE) -> torch.Tensor:
    """The BIMCV-R variant of the chain (Data/data_processing/BIMCV_R/2. BIMCV_R_nii_to_3D_volume_npy_file.py:30-87): float32 voxels
    (H, W, D) -> clamp [-1000, 200] -> (D, H, W) -> min-max -> CropForeground -> Resize, with NO rescale slope / intercept and NO
    respacing.  Same kernels as nii_to_volume (HU window + transpose, min/max, bounding box, one crop-normalise-resize gather)."""
    _need(raw)
    if raw.dim() != 3 or raw.dtype not in (torch.float32, torch.float64):
        raise L.HsError("bimcv_nii_to_volume expects the (H,W,D) voxel array as float32/float64")
    raw = raw.float().contiguous()                                       # the script reads the scan as float32 (get_fdata(dtype=np.float32))
    H, W, D = raw.shape
    hu = torch.empty(D, H, W, dtype=torch.float64, device=raw.device)    # (float32 values held exactly; the gather below reads f64)
    L.call("hs_prep_hu_window", raw.data_ptr(), 0, hu.data_ptr(), H, W, D, 1.0, 0.0, HU_MIN, HU_MAX, _s())
    mm = minmax(hu.float()).cpu()
    mn, mx = float(mm[0]), float(mm[1])
    den = max(np.float32(mx) - np.float32(mn), np.float32(1e-8))
    volf = hu.float()
    box = foreground_bbox(volf, mn).cpu().tolist()
    if box[3] < 0:
        box = [0, 0, 0, D - 1, H - 1, W - 1]
    origin, size = box[:3], [box[3 + a] - box[a] + 1 for a in range(3)]
    return resample(volf, out_size, (origin, size), sub=mn, den=float(den)).unsqueeze(0)


def draw_augmentation(rng: np.random.RandomState):
    """The random decisions of train_transform (multi_dataset.py:148-158), in its order: (rot_k, flip_mask, scale, shift)."""
    k = int(rng.randint(3) + 1) if rng.rand() < 0.5 else 0               # RandRotate90(prob=0.5, max_k=3)
    flips = 0
    for bit in range(3):                                                  # RandFlip(prob=0.10) on spatial axis 0, 1, 2
        if rng.rand() < 0.10:
            flips |= 1 << bit
    scale = 1.0 + float(rng.uniform(-0.1, 0.1)) if rng.rand() < 0.5 else 1.0        # RandScaleIntensity(factors=0.1)
    shift = float(rng.uniform(-0.1, 0.1)) if rng.rand() < 0.5 else 0.0              # RandShiftIntensity(offsets=0.1)
    return k, flips, scale, shift


def augment(image: torch.Tensor, rot_k: int = 0, flip_mask: int = 0, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    """image (C,D,H,W) float32 -> rot90(rot_k, axes=(H,W)) -> flips (bit0 D, bit1 H, bit2 W) -> *scale -> +shift."""
    _need(image)
    if image.dim() != 4 or image.dtype != torch.float32:
        raise L.HsError("augment expects a (C,D,H,W) float32 tensor")
    image = image.contiguous()
    Cc, D, H, W = image.shape
    out = torch.empty((Cc, D, W, H) if rot_k & 1 else (Cc, D, H, W), dtype=torch.float32, device=image.device)
    L.call("hs_prep_augment", image.data_ptr(), out.data_ptr(), Cc, D, H, W, int(rot_k), int(flip_mask), float(scale), float(shift), _s())
    return out
