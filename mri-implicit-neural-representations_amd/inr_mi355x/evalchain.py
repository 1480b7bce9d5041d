"""Validation chain of train.py:199-242 on the device: no_grad forward sweep over all C*H*W
coordinates -> (C,H,W,2) -> centred orthonormal inverse FFT -> |.| -> root-sum-of-squares ->
PSNR (models/utils.py:236-250: max(x), not max(x)^2).  fastmri's ifft2c / complex_abs / rss are
third-party and absent offline; these follow their published definitions via torch.fft (hipFFT)."""
from __future__ import annotations

import torch


def complex_abs(x: torch.Tensor) -> torch.Tensor:
    return (x ** 2).sum(dim=-1).sqrt()


def rss(x: torch.Tensor, dim: int = 0) -> torch.Tensor:
    return torch.sqrt((x ** 2).sum(dim))


def _fftc(x: torch.Tensor, inverse: bool) -> torch.Tensor:
    c = torch.view_as_complex(x.contiguous())
    c = torch.fft.ifftshift(c, dim=(-2, -1))
    c = (torch.fft.ifftn if inverse else torch.fft.fftn)(c, dim=(-2, -1), norm="ortho")
    c = torch.fft.fftshift(c, dim=(-2, -1))
    return torch.view_as_real(c)


def fft2c(x):
    return _fftc(x, False)


def ifft2c(x):
    return _fftc(x, True)


def psnr(x: torch.Tensor, xhat: torch.Tensor, epsilon: float = 1e-10) -> torch.Tensor:
    denom = torch.mean((x - xhat) ** 2)
    return 10 * torch.log10(torch.max(x) / (denom + epsilon))


def reconstruct(flat: torch.Tensor, shape, in_image_space: bool) -> torch.Tensor:
    C, H, W = shape
    im = flat.reshape(C, H, W, 2)
    if not in_image_space:
        im = ifft2c(im)
    return rss(complex_abs(im), dim=0)
