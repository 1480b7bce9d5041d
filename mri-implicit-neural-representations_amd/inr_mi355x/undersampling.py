"""K-space undersampling masks and the masked coordinate grid (host plumbing for the hot path).

Mirrors the reference's ``Undersampler`` (undersampling/undersampler.py:11-186) and the
``"function-params"`` argument grammar of ``MRIDatasetUndersampling.parse_undersampling_argument``
(data/nerp_datasets.py:256-310): ``"grid-3*2"``, ``"random_line-0.5"``, ``"radial-4"``.

The masks are built once per fit on the host and shipped to HBM as one bool per (y, x); the fused
train step consumes them as a row mask (train.py:172-177).  Differences from the reference, both
deliberate:

* the radial generator draws its golden-angle phase ``t`` from an UNSEEDED ``RandomState``
  (undersampler.py:115,123), so two runs of the reference never see the same mask; here ``t`` (or a
  ``seed`` it is drawn from) is an argument, and ``tests/golden/undersampling.npz`` pins the
  generator against the reference with ``t`` fixed;
* nothing is plotted or written to disk (the reference saves ``undersampling_mask.png``).
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

SUPPORTED_UNDERSAMPLING_METHODS = ("grid", "random_line", "radial")
GOLDEN_RATIO = (1 + np.sqrt(5)) / 2


def parse_undersampling_argument(arg: Optional[str]) -> Tuple[Optional[str], list]:
    """nerp_datasets.py:256-310: ``None``/``"none"`` -> (arg, []); else (method, params)."""
    if arg is None or arg.lower() == "none":
        return arg, []
    parts = arg.split("-")
    assert len(parts) == 2, f"Argument {arg} is incorrect"
    method, param = parts
    if method == "grid":
        assert "*" in param, "Please use * symbol for stating grid size"
        dims = param.split("*")
        assert len(dims) == 2, f"Grid dimensions provided ({param}) for undersampling is wrong please provide x*y format"
        return method, [int(dims[0]), int(dims[1])]
    if method == "random_line":
        p = float(param)
        assert 0 <= p <= 1.0, "P value is not in range [0,1]"
        return method, [p]
    if method == "radial":
        return method, [int(param)]
    raise NotImplementedError(f"Undersamping method: {method} not supported")


def grid_mask(H: int, W: int, grid_x: int = 3, grid_y: int = 3) -> torch.Tensor:
    """Every grid_x-th row x every grid_y-th column (undersampler.py:81-93)."""
    mask = torch.zeros((H, W), dtype=torch.bool)
    mask[::grid_x, ::grid_y] = True
    return mask


def random_line_mask(H: int, W: int, p: float, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Whole rows and whole columns kept with probability p each; rows are drawn before columns from
    torch's RNG (undersampler.py:97-113), so ``torch.manual_seed`` reproduces the reference's mask."""
    rows = torch.rand(H, generator=generator) <= p
    cols = torch.rand(W, generator=generator) <= p
    return rows[:, None] | cols[None, :]


def _perimeter_rc(S: int, sq: int, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(row, col) of the idx-th point on the clockwise perimeter, from the top-left corner, of the
    sq-th nested square of an S x S matrix -- closed form of utils.py:29-59 (top J points, right
    J-2, bottom J-1 leftwards, left J-1 upwards)."""
    J = S - 2 * sq
    lo, hi = sq, S - sq - 1
    row = np.empty_like(idx)
    col = np.empty_like(idx)
    top = idx < J
    right = (~top) & (idx < 2 * J - 2)
    bottom = (~top) & (~right) & (idx < 3 * J - 3)
    left = ~(top | right | bottom)
    row[top], col[top] = lo, lo + idx[top]
    row[right], col[right] = lo + 1 + (idx[right] - J), hi
    row[bottom], col[bottom] = hi, hi - (idx[bottom] - (2 * J - 2))
    row[left], col[left] = hi - (idx[left] - (3 * J - 3)), lo
    return row, col


def radial_mask(H: int, W: int, acceleration: int, t: Optional[int] = None, seed: Optional[int] = None) -> torch.Tensor:
    """Golden-angle radial spokes rasterised on nested squares (undersampler.py:115-155).

    M spokes; on each nested square of the even-sided max_dim x max_dim canvas spoke m lands on
    perimeter point floor(frac((m + t M) / phi) K), K = 4 (J - 1).  The canvas is padded by one
    row/column at the top/left for odd H/W and centre-cropped to (H, W).  ``t`` is the phase the
    reference draws as ``RandomState().randint(0, 1e4)``; give ``t`` or a ``seed`` to draw it from."""
    assert acceleration != 0, "Acceleration cannot be zero"
    if t is None:
        t = int(np.random.RandomState(seed).randint(low=0, high=1e4, size=1, dtype=int).item())
    max_dim = max(H, W) - max(H, W) % 2
    min_dim = min(H, W) - min(H, W) % 2
    nsq = max_dim // 2
    M = int(np.prod((H, W)) / (acceleration * (max_dim / 2 - (max_dim - min_dim) * (1 + min_dim / max_dim) / 4)))
    frac = np.mod((np.arange(M) + t * M) / GOLDEN_RATIO, 1)
    canvas = np.zeros((max_dim, max_dim), dtype=bool)
    for sq in range(nsq):
        K = 4 * (2 * (nsq - sq) - 1)
        r, c = _perimeter_rc(max_dim, sq, np.floor(frac * K).astype(np.int64))
        canvas[r, c] = True
    canvas = np.pad(canvas, ((H % 2, 0), (W % 2, 0)), constant_values=False)
    r0, c0 = (canvas.shape[0] - H) // 2, (canvas.shape[1] - W) // 2
    assert 0 < H <= canvas.shape[0] and 0 < W <= canvas.shape[1]
    return torch.from_numpy(np.ascontiguousarray(canvas[r0:r0 + H, c0:c0 + W]))


def acceleration_factor(mask: torch.Tensor) -> float:
    """numel / nonzero (utils.py:62-64)."""
    return float(mask.numel() / torch.count_nonzero(mask))


class Undersampler:
    """Same surface as the reference class: ``apply(kspace [C,H,W,2], params)`` returns the
    zero-filled k-space, the [C*H*W, 3] coordinate grid and its [C*H*W, 3] bool mask
    (undersampler.py:35-72, 159-186).  ``mask_image`` is the [H, W] mask itself."""

    def __init__(self, undersampling_method: str, seed: Optional[int] = None, t: Optional[int] = None):
        assert undersampling_method in SUPPORTED_UNDERSAMPLING_METHODS, \
            f"Undersamping method: {undersampling_method} not supported"
        self.undersampling_method = undersampling_method
        self.seed, self.t = seed, t
        self.mask_image = None
        self._grid = self._grid_mask = None

    def create_mask(self, H: int, W: int, params: List) -> torch.Tensor:
        m = self.undersampling_method
        if m == "grid":
            assert len(params) == 2, "Grid undersampling method's paramaters are not correct, it should have two parameters"
            self.mask_image = grid_mask(H, W, params[0], params[1])
        elif m == "random_line":
            assert len(params) == 1, "Random line undersampling method's paramaters are not correct, it should have one parameters"
            g = None if self.seed is None else torch.Generator().manual_seed(self.seed)
            self.mask_image = random_line_mask(H, W, params[0], g)
        else:
            assert len(params) == 1, "Radial undersampling method's paramaters are not correct, it should have one parameters"
            self.mask_image = radial_mask(H, W, params[0], t=self.t, seed=self.seed)
        return self.mask_image

    def apply(self, images_tensor: torch.Tensor, params: List):
        assert images_tensor.dim() == 4, \
            "For processing, please provide a 4-dimensional tensor as [batch_size, image_x, image_y, channel_n]"
        C, H, W, _ = images_tensor.shape
        mask = self.create_mask(H, W, params)
        masked = images_tensor * mask.to(images_tensor.device)[None, :, :, None]
        Z, Y, X = torch.meshgrid(torch.linspace(-1, 1, C), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W),
                                 indexing="ij")
        self._grid = torch.stack((Z.reshape(-1), Y.reshape(-1), X.reshape(-1)), dim=1)
        self._grid_mask = mask.reshape(1, H * W, 1).expand(C, H * W, 3).reshape(C * H * W, 3).contiguous()
        return masked, self._grid, self._grid_mask

    __call__ = apply

    def get_grid_and_mask(self):
        assert self._grid is not None, "Call apply() function first"
        return self._grid, self._grid_mask
