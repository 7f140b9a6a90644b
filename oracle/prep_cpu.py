"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by hsenet_amd/): CPU restatement of the reference's CT
preprocessing chain, Data/data_processing/CT-RATE/CT-RATE_nii_to_3D_volume_npy_file.py:25-124, and of the train-time
augmentation, LaMed/src/dataset/multi_dataset.py:148-158.

Pinning (round 5): `resize_array` and `nii_img_to_tensor` are pinned by RUNNING the reference script's own functions in the build
container (oracle/make_goldens.py --only prep: the script imported with empty stand-ins for the absent monai.transforms / nibabel, nib.load
replaced by a synthetic scan, Tensor.to('cuda') mapped to the CPU for the call): bit-equal, fixture tests/golden/prep_ref.npz.  MONAI's
CropForeground / Resize / RandRotate90 / RandFlip / RandScaleIntensity / RandShiftIntensity are third-party (monai==1.3.0, not under
/root/reference): restated from their published semantics -> parity unpinned for those steps (the reference function takes its
`transform` as an argument, so the pin above passes this file's crop + resize in).
"""
import numpy as np
import torch
import torch.nn.functional as F


def resize_array(array, current_spacing, target_spacing):                 # :25-37
    original_shape = array.shape[2:]
    scaling = [current_spacing[i] / target_spacing[i] for i in range(len(original_shape))]
    new_shape = [int(original_shape[i] * scaling[i]) for i in range(len(original_shape))]
    return F.interpolate(array, size=new_shape, mode="trilinear", align_corners=False).cpu().numpy()


def crop_foreground(t):                                                    # MONAI CropForeground(): select_fn = x > 0, margin 0
    fg = (t > 0).any(0)
    if not fg.any():
        return t
    idx = [torch.nonzero(fg.any(dim=tuple(a for a in range(3) if a != ax))).flatten() for ax in range(3)]
    lo = [int(i.min()) for i in idx]
    hi = [int(i.max()) + 1 for i in idx]
    return t[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]


def monai_resize(t, size):                                                 # MONAI Resize(mode="bilinear") on 3 spatial dims
    return F.interpolate(t.unsqueeze(0), size=list(size), mode="trilinear", align_corners=False).squeeze(0)


def nii_img_to_tensor(img_data, slope, intercept, z_spacing, xy_spacing, out_size=(32, 256, 256)):   # :39-115
    current = (z_spacing, xy_spacing, xy_spacing)
    target = (1.5, 0.75, 0.75)
    x = torch.tensor(img_data)                                             # float64 (nibabel get_fdata)
    x = slope * x + intercept
    x = torch.clamp(x, min=-1000, max=200)
    x = x.numpy().transpose(2, 0, 1)
    tensor = torch.tensor(x).unsqueeze(0).unsqueeze(0)
    arr = resize_array(tensor, current, target)[0][0]
    arr = np.transpose(arr, (1, 2, 0))
    arr = torch.tensor(arr).to(dtype=torch.float32).numpy()
    tensor = torch.tensor(arr).permute(2, 0, 1).unsqueeze(0)
    mn, mx = torch.min(tensor), torch.max(tensor)
    norm = (tensor - mn) / torch.clamp(mx - mn, min=1e-8)
    return monai_resize(crop_foreground(norm), out_size)


def bimcv_nii_to_tensor(img_data, out_size=(32, 256, 256)):
    """Data/data_processing/BIMCV_R/2. BIMCV_R_nii_to_3D_volume_npy_file.py:30-87 (convert_nii_to_npy, the main path): the float32 voxel
    array (H, W, D) -> clamp [-1000, 200] -> (D, H, W) -> min-max -> CropForeground -> Resize.  No rescale slope / intercept and no
    respacing in this script.  Restated from the source text (the function needs a CUDA device object to exist before its first line of
    arithmetic, and builds its MONAI transform internally); every step it shares with the CT-RATE function is pinned through that one."""
    x = torch.tensor(np.asarray(img_data, dtype=np.float32), dtype=torch.float32)
    x = torch.clamp(x, min=-1000, max=200)
    tensor = x.permute(2, 0, 1).unsqueeze(0)
    mn, mx = torch.min(tensor), torch.max(tensor)
    norm = (tensor - mn) / torch.clamp(mx - mn, min=1e-8)
    return monai_resize(crop_foreground(norm), out_size)


def augment(image, k, flips, scale, shift):                                # multi_dataset.py:148-158 with the draws fixed
    x = torch.rot90(image, k, dims=(2, 3)) if k else image
    for bit, ax in ((1, 1), (2, 2), (4, 3)):
        if flips & bit:
            x = torch.flip(x, dims=(ax,))
    return x * scale + shift
