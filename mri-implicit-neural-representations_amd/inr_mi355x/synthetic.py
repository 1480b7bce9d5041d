"""Synthetic multi-coil k-space with the tensor contract of the reference's datasets
(data/nerp_datasets.py:101-143; data/utils.py:98-108): a normalised (C,H,W,2) k-space flattened
C-major to image [(C*H*W),2] plus coords [(C*H*W),3] ordered (coil, y, x) in [-1,1]^3.

The reference has no synthetic generator (it reads fastMRI HDF5, which is not available
offline); this one follows SURVEY.md section 8(d): K random ellipses x smooth complex coil
sensitivities + complex Gaussian noise -> centred orthonormal FFT2 per coil -> normalisation.
Deterministic: numpy Generator(PCG64) with an explicit seed, float64 math, one final cast."""
from __future__ import annotations

import numpy as np
import torch


def create_coords(c: int, h: int, w: int) -> torch.Tensor:
    """data/utils.py:98-108."""
    Z, Y, X = torch.meshgrid(torch.linspace(-1, 1, c), torch.linspace(-1, 1, h), torch.linspace(-1, 1, w),
                             indexing="ij")
    return torch.hstack((Z.reshape(-1, 1), Y.reshape(-1, 1), X.reshape(-1, 1)))


def phantom(H: int, W: int, rng: np.random.Generator, n_ellipses: int = 12) -> np.ndarray:
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    img = np.zeros((H, W))
    for _ in range(n_ellipses):
        cx, cy = rng.uniform(-0.5, 0.5, 2)
        a, b = rng.uniform(0.08, 0.55, 2)
        th = rng.uniform(0, np.pi)
        v = rng.uniform(0.2, 1.0)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        # soft edge keeps the spectrum decaying like a real anatomy image
        r = (xr / a) ** 2 + (yr / b) ** 2
        img += v / (1.0 + np.exp(np.minimum(40.0 * (r - 1.0), 60.0)))
    return img


def make_kspace(C: int = 15, H: int = 640, W: int = 368, seed: int = 1234, normalization: str = "coil",
                noise: float = 1e-3, image_space: bool = False):
    """Returns (image [(C*H*W),2] f32, coords [(C*H*W),3] f32, shape (C,H,W))."""
    rng = np.random.default_rng(seed)
    img = phantom(H, W, rng)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    coils = np.empty((C, H, W), dtype=np.complex128)
    for c in range(C):
        ang = 2 * np.pi * c / C
        cx, cy = 0.9 * np.cos(ang), 0.9 * np.sin(ang)
        sens = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 0.7 ** 2))
        phase = np.exp(1j * (1.5 * (xx * np.cos(ang) + yy * np.sin(ang)) + 0.3 * c))
        nz = noise * (rng.standard_normal((H, W)) + 1j * rng.standard_normal((H, W)))
        coils[c] = img * sens * phase + nz
    if image_space:
        data = coils
        data = data / np.abs(data).max()  # normalize_image (data/utils.py:90-96)
    else:
        data = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(coils, axes=(-2, -1)), norm="ortho"), axes=(-2, -1))
        if normalization == "max":  # nerp_datasets.py:113-115: max over real and imaginary components
            data = data / max(np.abs(data.real).max(), np.abs(data.imag).max())
        elif normalization == "coil":  # nerp_datasets.py:134-136: per-coil max magnitude
            data = data / np.abs(data).reshape(C, -1).max(axis=1)[:, None, None]
        elif normalization in (None, "none"):
            pass
        else:
            raise NotImplementedError(normalization)
    k = np.stack([data.real, data.imag], axis=-1).astype(np.float32)
    image = torch.from_numpy(k.reshape(C * H * W, 2).copy())
    return image, create_coords(C, H, W), (C, H, W)
