"""fastMRI ingest and the dataset contracts of the fitting loops (the reference's
src/data/nerp_datasets.py:13-441 and models/utils.py:57-135), kept DEVICE-RESIDENT: one slice of one scan is read
once, pre-processed on the GPU with torch.fft (hipFFT) and stays in HBM as flat ``(C*H*W, 2)`` / ``(C*H*W, 3)``
tensors; the loops slice views out of them, there is no DataLoader, no collate, no per-batch H2D copy.

Containers:
  *.h5   fastMRI multicoil file (``kspace`` [slices, C, H, W] complex64 + ``ismrmrd_header`` XML).  Read with h5py
         when it is importable; this image has no h5py, so the call then fails loudly -- nothing is substituted.
  *.npz  the same two members under the same names (``kspace``, and ``ismrmrd_header`` bytes or ``crop_size``):
         what ``tools``-side conversion of a scan produces on a machine that has h5py.
  *.npy  a bare [slices, C, H, W] complex array (no header: crop = the array's own H, W).
  dir    sorted ``*.h5`` (then ``*.npz``) files, entry number ``sample`` (nerp_datasets.py:190-216).  The reference's
         list of malformed scans never matches (it compares Path objects with strings, :203-204), so ``sample``
         indexes ALL sorted files there; the same indexing is kept here.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .evalchain import complex_abs, fft2c, ifft2c
from .synthetic import create_coords
from .undersampling import Undersampler, parse_undersampling_argument

_ISMRMRD_NS = "http://www.ismrm.org/ISMRMRD"


def recon_size_from_ismrmrd(header) -> Tuple[int, int, int]:
    """encoding/reconSpace/matrixSize (x, y, z) of an ISMRMRD XML header (nerp_datasets.py:152-175 returns exactly
    this; the padding arithmetic it also does is unused there)."""
    if isinstance(header, np.ndarray):
        header = header.item() if header.shape == () else header.tobytes()
    if isinstance(header, str):
        header = header.encode()
    root = ET.fromstring(header)
    node = root
    for name in ("encoding", "reconSpace", "matrixSize"):
        nxt = node.find(f"{{{_ISMRMRD_NS}}}{name}")
        if nxt is None:
            nxt = node.find(name)
        if nxt is None:
            raise RuntimeError(f"ismrmrd_header: element {name!r} not found")
        node = nxt
    out = []
    for ax in ("x", "y", "z"):
        v = node.find(f"{{{_ISMRMRD_NS}}}{ax}")
        if v is None:
            v = node.find(ax)
        if v is None:
            raise RuntimeError(f"ismrmrd_header: matrixSize/{ax} not found")
        out.append(int(v.text))
    return tuple(out)


def _read_h5(path: Path):
    try:
        import h5py
    except ImportError as e:  # loud: there is no silent substitute for a scan
        raise ImportError(f"reading {path} needs h5py, which is not installed; convert the scan to .npz "
                          "(members 'kspace' and 'ismrmrd_header') on a machine that has it") from e
    with h5py.File(str(path), "r") as f:
        data = f["kspace"][()]
        crop = recon_size_from_ismrmrd(f["ismrmrd_header"][()])
    return data, crop


def _read_npz(path: Path):
    with np.load(str(path), allow_pickle=False) as z:
        data = z["kspace"]
        if "ismrmrd_header" in z.files:
            crop = recon_size_from_ismrmrd(z["ismrmrd_header"])
        elif "crop_size" in z.files:
            crop = tuple(int(v) for v in z["crop_size"])
        else:
            crop = (data.shape[-2], data.shape[-1], 1)
    return data, crop


def load_kspace_file(path_or_file: str, sample: Optional[int] = None):
    """-> (kspace [slices, C, H, W] complex ndarray, crop_size (x, y[, z]), Path)   (nerp_datasets.py:177-216)."""
    p = Path(path_or_file)
    if p.is_dir():
        files = sorted(p.glob("*.h5")) or sorted(p.glob("*.npz"))
        assert len(files) > 0, f"No files in the path {path_or_file}"
        if sample is None:
            raise NotImplementedError("Multi path loading is not currently supported yet")
        p = files[sample]
    if p.suffix == ".h5":
        data, crop = _read_h5(p)
    elif p.suffix == ".npz":
        data, crop = _read_npz(p)
    elif p.suffix == ".npy":
        data = np.load(str(p))
        crop = (data.shape[-2], data.shape[-1], 1)
    else:
        raise ValueError(f"{p}: expected a .h5 / .npz / .npy scan or a directory of them")
    if data.ndim != 4 or not np.iscomplexobj(data):
        raise ValueError(f"{p}: kspace must be a complex [slices, coils, H, W] array, got {data.dtype} {data.shape}")
    return data, crop, p


def complex_center_crop(data: torch.Tensor, shape: Sequence[int]) -> torch.Tensor:
    """data [..., H, W, 2] -> centre crop to (shape[0], shape[1]); a crop wider than W falls back to a W x W square
    (data/utils.py:65-88)."""
    if data.shape[-2] < shape[1]:
        shape = (data.shape[-2], data.shape[-2])
    assert 0 < shape[0] <= data.shape[-3]
    assert 0 < shape[1] <= data.shape[-2]
    h0 = (data.shape[-3] - shape[0]) // 2
    w0 = (data.shape[-2] - shape[1]) // 2
    return data[..., h0:h0 + shape[0], w0:w0 + shape[1], :]


def normalize_image(data: torch.Tensor, full_norm: bool = False) -> torch.Tensor:
    """data / max |data|  (data/utils.py:90-96; ``full_norm`` is accepted and ignored there as well)."""
    return data / complex_abs(data).max()


def _gaussian_blur(k: torch.Tensor, sigma: float) -> torch.Tensor:
    """3x3 Gaussian, zero 'same' padding, on each of the two components (data/utils.py gaussian_filter_2d)."""
    ax = torch.arange(-1, 2, dtype=k.dtype, device=k.device)
    g = torch.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma ** 2))
    g = (g / g.sum())[None, None]
    x = k.permute(0, 3, 1, 2)
    out = torch.cat([torch.nn.functional.conv2d(x[:, i:i + 1], g, padding=1) for i in range(x.shape[1])], dim=1)
    return out.permute(0, 2, 3, 1)


def normalize_kspace(k: torch.Tensor, type: Optional[str] = "max", eps: float = 1e-9) -> torch.Tensor:
    """k [C,H,W,2] -> normalised k-space, the seven schemes of nerp_datasets.py:108-143 (anything else: unchanged)."""
    if type == "abs_max":
        return k / complex_abs(k).max()
    if type == "max":  # max over the real AND imaginary components, not the magnitude
        return k / k.abs().max()
    if type == "gaussian_blur":
        return _gaussian_blur(k / k.abs().max(), 0.1)
    if type == "max_std":
        k = k / k.abs().max()
        k = (k - k.mean()) / k.std()
        return k / k.max()
    if type == "tonemap":
        k = k / (k + 1)
        k = k / k.max()
        return k - k.mean(dim=(1, 2, 3), keepdim=True)
    if type == "coil":
        mx = complex_abs(k).reshape(k.shape[0], -1).max(dim=-1)[0]
        return k / mx[:, None, None, None]
    if type == "stand":
        return (k - k.mean()) / (k.std() + eps)
    return k


def preprocess_slice(kspace_slice, crop_size, transform: bool, centercrop: bool = True, normalization="max",
                     full_norm: bool = False, device="cpu") -> torch.Tensor:
    """One raw slice [C,H,W] complex -> the normalised [C,H',W',2] fp32 tensor the loops fit
    (nerp_datasets.py:60-76): ifft2c, centre crop in image space, then either normalize_image (image mode) or
    fft2c + normalize_kspace (k-space mode)."""
    z = torch.as_tensor(np.ascontiguousarray(kspace_slice)).to(torch.complex64).to(device)
    data = ifft2c(torch.view_as_real(z))
    if centercrop:
        data = complex_center_crop(data, crop_size)
    if transform:
        return normalize_image(data, full_norm).contiguous()
    return normalize_kspace(fft2c(data), normalization).contiguous()


class MRIDataset:
    """nerp_datasets.py:13-242 with the same constructor arguments (+ ``device``).  ``image`` (C*H*W, 2),
    ``coords`` (C*H*W, 3), ``shape`` (C, H, W, 2) live on ``device``."""

    def __init__(self, data_class="brain", data_root="data", challenge="multicoil", set="train", transform=True,
                 sample=0, slice=0, full_norm=False, custom_file_or_path=None, per_coil_stats=True, centercrop=True,
                 normalization="max", device="cuda"):
        self.challenge, self.transform, self.data_class = challenge, transform, data_class
        self.data_root, self.set, self.device = data_root, set, torch.device(device)
        if custom_file_or_path is None or custom_file_or_path == "":
            self.root = "{}/{}_{}_{}/".format(data_root, data_class, challenge, set)
        else:
            self.root = custom_file_or_path
        kspace, crop_size, self.file_name = load_kspace_file(self.root, sample)
        data = preprocess_slice(kspace[slice], crop_size, transform, centercrop, normalization, full_norm, self.device)
        self.shape = tuple(data.shape)  # (C, H, W, 2)
        self.coil_stats = per_coil_statistics(data) if per_coil_stats else None
        self.flatten_image_and_create_coords(data)

    def flatten_image_and_create_coords(self, data: torch.Tensor) -> None:
        C, H, W, S = data.shape
        self.image = data.reshape(C * H * W, S)
        self.coords = create_coords(C, H, W).to(self.device)

    @property
    def file(self):
        return self.file_name

    @property
    def img_shape(self):
        return self.shape

    def __getitem__(self, idx):
        return self.coords[idx], self.image[idx], list(), list()

    def __len__(self):
        return len(self.image)


def per_coil_statistics(data: torch.Tensor) -> List[tuple]:
    """(coil, mean, std, max, min) rows of the table nerp_datasets.py:80-95 prints."""
    flat = data.reshape(data.shape[0], -1)
    cols = torch.stack([flat.mean(1), flat.std(1), flat.max(1)[0], flat.min(1)[0]], dim=1).cpu().tolist()
    return [(i, *row) for i, row in enumerate(cols)]


class MRIDatasetUndersampling(MRIDataset):
    """nerp_datasets.py:244-344: + ``undersampling`` = "grid-3*3" | "random_line-0.5" | "radial-4" | None; samples
    are (coords, masked image, [], coords_mask)."""

    def __init__(self, data_class="brain", data_root="data", challenge="multicoil", set="train", transform=True,
                 sample=0, slice=0, full_norm=False, custom_file_or_path=None, per_coil_stats=True, centercrop=True,
                 normalization="max", undersampling=None, device="cuda"):
        self.undersampling_argument, self.undersampling_params = parse_undersampling_argument(undersampling)
        super().__init__(data_class, data_root, challenge, set, transform, sample, slice, full_norm,
                         custom_file_or_path, per_coil_stats, centercrop, normalization, device)

    def flatten_image_and_create_coords(self, data: torch.Tensor) -> None:
        C, H, W, S = data.shape
        if self.undersampling_argument is None or self.undersampling_argument.lower() == "none":
            super().flatten_image_and_create_coords(data)
            return
        self.undersampler = Undersampler(self.undersampling_argument)
        masked, coords, coords_mask = self.undersampler.apply(data, self.undersampling_params)
        self.image = masked.reshape(C * H * W, S)
        self.shape = tuple(masked.shape)
        self.coords = coords.to(self.device)
        self.coords_mask = coords_mask.to(self.device)

    def __len__(self):
        return len(self.coords)

    def __getitem__(self, idx):
        return self.coords[idx], self.image[idx], list(), (self.coords_mask[idx] if hasattr(self, "coords_mask") else list())


class MRIDatasetWithDistances(MRIDatasetUndersampling):
    """nerp_datasets.py:349-395: + ``dist_to_center`` = sqrt(y^2 + x^2) per coordinate (``cat_dists`` appends it as
    a fourth coordinate, ``cat_coil`` returns (coil, last) columns in the third slot)."""

    def __init__(self, data_class="brain", data_root="data", challenge="multicoil", set="train", transform=True,
                 sample=0, slice=0, full_norm=False, custom_file_or_path=None, per_coil_stats=True, centercrop=True,
                 normalization="max", cat_coil=False, cat_dists=False, undersampling=None, device="cuda"):
        super().__init__(data_class, data_root, challenge, set, transform, sample, slice, full_norm,
                         custom_file_or_path, per_coil_stats, centercrop, normalization, undersampling, device)
        self.dist_to_center = torch.sqrt(self.coords[..., 1] ** 2 + self.coords[..., 2] ** 2)
        if cat_dists:
            self.coords = torch.cat((self.coords, self.dist_to_center.unsqueeze(-1)), dim=-1)
        self.cat_coil = cat_coil

    def __getitem__(self, idx):
        if self.cat_coil:
            return self.coords[idx], self.image[idx], self.coords[idx][..., [0, -1]], list()
        return self.coords[idx], self.image[idx], self.dist_to_center[idx], list()


class MRICoilWrapperDataset:
    """nerp_datasets.py:397-441: one item = one whole coil (H*W rows), so that Total Variation can be taken over the
    predicted plane.  Items are views of the wrapped dataset's HBM tensors."""

    def __init__(self, dataset, undersampling=True, coord_size=3):
        self.dataset, self.coord_size = dataset, coord_size
        C, H, W, S = dataset.shape
        self.len = C
        self.img_shape, self.file, self.shape = dataset.img_shape, dataset.file, dataset.shape
        self.image = dataset.image.reshape(C, H, W, S)
        self.coords = dataset.coords.reshape(C, H, W, coord_size)
        if hasattr(dataset, "dist_to_center"):
            self.dist_to_center = dataset.dist_to_center.reshape(C, H, W, 1)
        if hasattr(dataset, "coords_mask"):
            self.coords_mask = dataset.coords_mask.reshape(C, H, W, coord_size)
        self.undersampling = undersampling

    def __len__(self):
        return self.len

    def __getitem__(self, idx):
        img = self.image[idx].reshape(-1, 2)
        coords = self.coords[idx].reshape(-1, self.coord_size)
        mask = (self.coords_mask[idx].reshape(-1, self.coord_size)
                if self.undersampling is not None and hasattr(self, "coords_mask") else list())
        if type(self.dataset) is MRIDatasetWithDistances:
            return coords, img, self.dist_to_center[idx].reshape(-1, 1), mask
        if type(self.dataset) is MRIDatasetUndersampling:
            return coords, img, list(), mask
        return coords, img, list(), list()


def get_datasets(data, data_root, set, transform=True, sample=0, slice=0, challenge="multicoil", full_norm=False,
                 normalization="max", use_dists="no", undersampling=None, per_coil=False, custom_file_or_path=None,
                 device="cuda"):
    """The dataset half of models/utils.py:57-135 ``get_data_loader``: -> (dataset, train_dataset).  ``dataset`` is
    the fully sampled one the validation sweep reads; ``train_dataset`` is the undersampled (and, with ``per_coil``,
    coil-wrapped) one the loop trains on -- the same object as ``dataset`` when nothing is undersampled.  The
    loaders themselves have no counterpart: batches are views (``iter_batches``)."""
    assert data in ("brain", "knee"), "Unsupported parameter is provided in the get_data_loader() function"
    dists = use_dists == "yes" or use_dists is True
    common = dict(data_class=data, data_root=data_root, challenge=challenge, set=set, transform=transform,
                  sample=sample, slice=slice, full_norm=full_norm, normalization=normalization,
                  custom_file_or_path=custom_file_or_path, device=device)
    none = undersampling is None or str(undersampling).lower() == "none"
    if dists:
        dataset = MRIDatasetWithDistances(undersampling=None, **common)
    else:
        dataset = MRIDataset(**common)
    if none:
        train = dataset
    elif dists:
        train = MRIDatasetWithDistances(undersampling=undersampling, per_coil_stats=False, **common)
    else:
        train = MRIDatasetUndersampling(undersampling=undersampling, per_coil_stats=False, **common)
    if per_coil:
        train = MRICoilWrapperDataset(train, undersampling=None if none else undersampling)
    return dataset, train


def iter_batches(dataset, batch_size: int):
    """Sequential, unshuffled (coords, gt, dist, mask) batches as views -- what the reference's
    DataLoader(shuffle=False, collate_fn=collate_inr) yields (models/utils.py:84-90), minus the copies.  A
    coil-wrapped dataset yields one coil per batch."""
    if isinstance(dataset, MRICoilWrapperDataset):
        for c in range(len(dataset)):
            yield dataset[c]
        return
    n = len(dataset)
    for lo in range(0, n, batch_size):
        yield dataset[slice(lo, min(lo + batch_size, n))]


def trainer_inputs(dataset):
    """(image, coords, (C, H, W)) of a fully sampled dataset, the form INRTrainer / MultiscaleTrainer take; they apply
    ``config["undersampling"]`` and ``config["per_coil"]`` themselves to these resident tensors."""
    C, H, W, _ = dataset.shape
    return dataset.image, dataset.coords[:, :3].contiguous(), (C, H, W)


def from_config(config: dict, device="cuda"):
    """The fully sampled dataset train.py:271-287 builds from config['data' / 'data_root' / 'set' / 'sample' /
    'slice' / 'transform' / 'full_norm' / 'normalization'] (+ 'custom_file_or_path')."""
    return MRIDataset(data_class=config.get("data", "brain"), data_root=config.get("data_root", "data"),
                      set=config.get("set", "train"), transform=bool(config.get("transform", False)),
                      sample=config.get("sample", 0), slice=config.get("slice", 0),
                      full_norm=config.get("full_norm", False), normalization=config.get("normalization", "max"),
                      custom_file_or_path=config.get("custom_file_or_path"), per_coil_stats=False, device=device)
