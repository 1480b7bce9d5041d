"""fastMRI ingest + dataset contracts (inr_mi355x/datasets.py) against an independent float64 numpy restatement of
nerp_datasets.py:60-143 / data/utils.py:65-96 -- CPU only, no GPU library needed."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))

from inr_mi355x import datasets as D  # noqa: E402

HEADER = """<?xml version="1.0" encoding="utf-8"?>
<ismrmrdHeader xmlns="http://www.ismrm.org/ISMRMRD" xmlns:xs="http://www.w3.org/2001/XMLSchema">
  <encoding>
    <encodedSpace><matrixSize><x>{ex}</x><y>{ey}</y><z>1</z></matrixSize></encodedSpace>
    <reconSpace><matrixSize><x>{rx}</x><y>{ry}</y><z>1</z></matrixSize></reconSpace>
    <encodingLimits><kspace_encoding_step_1><minimum>0</minimum><maximum>19</maximum><center>10</center></kspace_encoding_step_1></encodingLimits>
  </encoding>
</ismrmrdHeader>"""


def _scan(S=3, C=4, H=24, W=20, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((S, C, H, W)) + 1j * rng.standard_normal((S, C, H, W))).astype(np.complex64)


def _c(a):  # centred orthonormal transforms, float64
    return lambda x: np.fft.fftshift(a(np.fft.ifftshift(x, axes=(-2, -1)), axes=(-2, -1), norm="ortho"), axes=(-2, -1))


def _ref(kslice, crop, transform, normalization):
    img = _c(np.fft.ifft2)(kslice.astype(np.complex128))
    C, H, W = img.shape
    ch, cw = crop[0], crop[1]
    if W < cw:
        ch = cw = W
    h0, w0 = (H - ch) // 2, (W - cw) // 2
    img = img[:, h0:h0 + ch, w0:w0 + cw]
    if transform:
        z = img / np.abs(img).max()
        return np.stack([z.real, z.imag], -1)
    k = _c(np.fft.fft2)(img)
    k2 = np.stack([k.real, k.imag], -1)
    n = normalization
    if n == "abs_max":
        k2 = k2 / np.abs(k).max()
    elif n == "max":
        k2 = k2 / np.abs(k2).max()
    elif n == "max_std":
        k2 = k2 / np.abs(k2).max()
        k2 = (k2 - k2.mean()) / k2.std(ddof=1)
        k2 = k2 / k2.max()
    elif n == "tonemap":
        k2 = k2 / (k2 + 1)
        k2 = k2 / k2.max()
        k2 = k2 - k2.mean(axis=(1, 2, 3), keepdims=True)
    elif n == "coil":
        k2 = k2 / np.abs(k).reshape(C, -1).max(1)[:, None, None, None]
    elif n == "stand":
        k2 = (k2 - k2.mean()) / (k2.std(ddof=1) + 1e-9)
    elif n == "gaussian_blur":
        k2 = k2 / np.abs(k2).max()
        k1 = np.exp(-np.arange(-1, 2) ** 2 / (2 * 0.1 ** 2))
        k1 /= k1.sum()
        p = np.pad(k2, ((0, 0), (1, 1), (0, 0), (0, 0)))
        k2 = sum(k1[a] * p[:, a:a + k2.shape[1]] for a in range(3))
        p = np.pad(k2, ((0, 0), (0, 0), (1, 1), (0, 0)))
        k2 = sum(k1[a] * p[:, :, a:a + k2.shape[2]] for a in range(3))
    return k2


@pytest.mark.parametrize("normalization", ["abs_max", "max", "max_std", "coil", "stand", "gaussian_blur", "none"])
def test_kspace_pipeline_vs_float64(tmp_path, normalization):
    ks = _scan()
    f = tmp_path / "scan.npz"
    np.savez(f, kspace=ks, ismrmrd_header=np.frombuffer(HEADER.format(ex=24, ey=20, rx=16, ry=12).encode(), np.uint8))
    ds = D.MRIDataset(transform=False, sample=0, slice=1, custom_file_or_path=str(f), normalization=normalization,
                      device="cpu")
    want = _ref(ks[1], (16, 12, 1), False, normalization)
    assert ds.shape == (4, 16, 12, 2) and ds.img_shape == ds.shape and ds.file == f
    assert ds.image.dtype == torch.float32 and ds.image.shape == (4 * 16 * 12, 2) and ds.coords.shape == (4 * 16 * 12, 3)
    np.testing.assert_allclose(ds.image.numpy().reshape(want.shape), want, rtol=2e-4, atol=2e-6)
    c, g, d, m = ds[5:9]
    assert c.shape == (4, 3) and g.shape == (4, 2) and d == [] and m == []
    assert len(ds) == 4 * 16 * 12 and len(ds.coil_stats) == 4
    # coordinate grid: coil, row, column in [-1, 1] (data/utils.py:98-108)
    grid = ds.coords.reshape(4, 16, 12, 3)
    np.testing.assert_allclose(grid[:, 0, 0, 0], np.linspace(-1, 1, 4), atol=1e-6)
    np.testing.assert_allclose(grid[0, :, 0, 1], np.linspace(-1, 1, 16), atol=1e-6)
    np.testing.assert_allclose(grid[0, 0, :, 2], np.linspace(-1, 1, 12), atol=1e-6)


def test_tonemap_and_image_mode_and_crop_fallback(tmp_path):
    ks = _scan(S=2, C=3, H=24, W=10, seed=3) * 0.05  # small values: k + 1 stays away from 0 in the tonemap
    f = tmp_path / "scan.npz"
    np.savez(f, kspace=ks, crop_size=np.array([16, 16, 1]))  # W = 10 < 16 -> 10 x 10 square (data/utils.py:80-81)
    ds = D.MRIDataset(transform=True, slice=0, custom_file_or_path=str(f), device="cpu")
    want = _ref(ks[0], (16, 16, 1), True, None)
    assert ds.shape == (3, 10, 10, 2)
    np.testing.assert_allclose(ds.image.numpy().reshape(want.shape), want, rtol=2e-4, atol=2e-6)
    assert abs(float(D.complex_abs(ds.image).max()) - 1.0) < 1e-6
    ds = D.MRIDataset(transform=False, slice=0, custom_file_or_path=str(f), normalization="tonemap", device="cpu")
    want = _ref(ks[0], (16, 16, 1), False, "tonemap")
    np.testing.assert_allclose(ds.image.numpy().reshape(want.shape), want, rtol=5e-4, atol=5e-6)
    ds = D.MRIDataset(transform=False, slice=0, custom_file_or_path=str(f), centercrop=False, device="cpu")
    assert ds.shape == (3, 24, 10, 2)


def test_h5_route_and_directory_indexing(tmp_path, monkeypatch):
    ks = {n: _scan(S=1, C=2, H=8, W=8, seed=i) for i, n in enumerate(["file_b.h5", "file_a.h5", "file_c.h5"])}
    for n in ks:
        (tmp_path / n).write_bytes(b"")
    monkeypatch.setitem(sys.modules, "h5py", None)  # import h5py -> ImportError
    with pytest.raises(ImportError, match="h5py"):
        D.load_kspace_file(str(tmp_path), 0)

    class File:  # the three things nerp_datasets.py:183-187 touches
        def __init__(self, path, mode):
            assert mode == "r"
            self.m = {"kspace": ks[os.path.basename(path)], "ismrmrd_header": HEADER.format(ex=8, ey=8, rx=6, ry=4).encode()}

        def __getitem__(self, k):
            return _Item(self.m[k])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class _Item:
        def __init__(self, v):
            self.v = v

        def __getitem__(self, idx):
            assert idx == ()
            return self.v

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    data, crop, p = D.load_kspace_file(str(tmp_path), 1)  # sorted: file_a, file_b, file_c
    assert p.name == "file_b.h5" and crop == (6, 4, 1) and np.array_equal(data, ks["file_b.h5"])
    ds = D.MRIDataset(data_class="brain", data_root=str(tmp_path), set="train", transform=False, sample=0, device="cpu",
                      custom_file_or_path=str(tmp_path / "file_c.h5"))
    assert ds.shape == (2, 6, 4, 2)
    with pytest.raises(NotImplementedError):
        D.load_kspace_file(str(tmp_path), None)
    with pytest.raises(ValueError):
        D.load_kspace_file(str(tmp_path / "scan.txt"))


def test_undersampled_distance_and_coil_contracts(tmp_path):
    ks = _scan(S=1, C=3, H=16, W=12, seed=5)
    f = tmp_path / "scan.npy"
    np.save(f, ks)
    full, train = D.get_datasets("brain", "unused", "train", transform=False, custom_file_or_path=str(f),
                                 undersampling="grid-2*3", use_dists="yes", per_coil=True, device="cpu")
    assert type(full) is D.MRIDatasetWithDistances and type(train) is D.MRICoilWrapperDataset and len(train) == 3
    us = train.dataset
    mask = us.coords_mask.reshape(3, 16, 12, 3)
    assert torch.equal(mask[0], mask[2]) and 0 < float(mask.float().mean()) < 1
    # masked image = full image where sampled, zero elsewhere (undersampler.py:163-169)
    sampled = us.coords_mask[:, 0] != 0
    assert torch.equal(us.image[sampled], full.image[sampled]) and float(us.image[~sampled].abs().max()) == 0.0
    np.testing.assert_allclose(full.dist_to_center, torch.sqrt(full.coords[:, 1] ** 2 + full.coords[:, 2] ** 2))
    coords, img, dists, m = train[1]
    assert coords.shape == (192, 3) and img.shape == (192, 2) and dists.shape == (192, 1) and m.shape == (192, 3)
    assert torch.equal(coords, us.coords[192:384]) and torch.equal(img, us.image[192:384])
    batches = list(D.iter_batches(full, 100))
    assert [b[0].shape[0] for b in batches] == [100] * 5 + [76]
    assert torch.equal(torch.cat([b[1] for b in batches]), full.image)
    assert torch.equal(batches[2][2], full.dist_to_center[200:300])
    assert len(list(D.iter_batches(train, 100))) == 3
    # no undersampling: one dataset serves both roles, the wrapper yields empty masks
    full2, train2 = D.get_datasets("brain", "unused", "train", transform=False, custom_file_or_path=str(f), device="cpu")
    assert full2 is train2 and type(full2) is D.MRIDataset
    cat = D.MRIDatasetWithDistances(transform=False, custom_file_or_path=str(f), cat_dists=True, cat_coil=True, device="cpu")
    c4, _, coil_dist, _ = cat[7:9]
    assert c4.shape == (2, 4) and torch.equal(coil_dist, c4[:, [0, 3]])
    image, coords, shape = D.trainer_inputs(full)
    assert shape == (3, 16, 12) and image.shape == (576, 2) and coords.shape == (576, 3)
    with pytest.raises(AssertionError):
        D.get_datasets("abdomen", "x", "train")


def test_recon_size_without_namespace():
    xml = "<h><encoding><reconSpace><matrixSize><x>320</x><y>320</y><z>1</z></matrixSize></reconSpace></encoding></h>"
    assert D.recon_size_from_ismrmrd(xml) == (320, 320, 1)
    with pytest.raises(RuntimeError):
        D.recon_size_from_ismrmrd("<h><encoding/></h>")


# ---- pinned to the reference's own ingest arithmetic (tests/golden/ingest.npz, tools/make_golden.py: ingest_vectors) ----
GOLD = os.path.join(ROOT, "tests", "golden")


def _ingest_checks(device):
    """datasets.py against what MRIDataset.__normalize_kspace (nerp_datasets.py:108-143), complex_center_crop /
    normalize_image / gaussian_filter_2d / create_coords (data/utils.py:19-28,65-108) and retrieve_size
    (nerp_datasets.py:151-170) of the reference returned for the same tensors, rtol 1e-5."""
    import json
    from inr_mi355x.synthetic import create_coords
    z = np.load(os.path.join(GOLD, "ingest.npz"))
    meta = json.load(open(os.path.join(GOLD, "ingest_meta.json")))
    k = torch.from_numpy(z["kspace"]).to(device)
    for n in meta["normalizations"]:
        got = D.normalize_kspace(k.clone(), n).cpu().numpy()
        np.testing.assert_allclose(got, z[f"norm/{n}"], rtol=1e-5, atol=1e-7, err_msg=n)
    img = torch.from_numpy(z["image"]).to(device)
    np.testing.assert_allclose(D.normalize_image(img).cpu().numpy(), z["normalize_image"], rtol=1e-5, atol=1e-7)
    for tag, shp in meta["crops"].items():
        got = D.complex_center_crop(img, shp).cpu().numpy()
        assert got.shape == z[f"crop/{tag}"].shape and np.array_equal(got, z[f"crop/{tag}"]), tag
    # the 3 x 3 blur of the 'gaussian_blur' scheme on its own: [N,1,H,W] in the reference, [C,H,W,2] here
    x = torch.from_numpy(z["blur_in"]).to(device)  # [2,1,9,7]
    got = D._gaussian_blur(x.permute(1, 2, 3, 0).contiguous(), 0.1).permute(3, 0, 1, 2).cpu().numpy()
    np.testing.assert_allclose(got, z["blur_out"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(create_coords(3, 5, 4).numpy(), z["coords_3_5_4"], rtol=0, atol=1e-7)
    assert list(D.recon_size_from_ismrmrd(meta["header"])) == meta["recon_size"]


def test_ingest_matches_reference_fixtures_cpu():
    _ingest_checks("cpu")


@pytest.mark.gpu
def test_ingest_matches_reference_fixtures_on_device():
    """The same on the MI355X (the tensors stay in HBM, reductions and the blur run as device kernels), and the whole
    slice pipeline -- hipFFT ifft2c, crop, fft2c, normalisation -- against the float64 restatement at 1e-5 of the
    largest value (the centred FFT itself is fastmri's published definition: third-party, absent)."""
    _ingest_checks("cuda:0")
    ks = _scan(S=2, C=4, H=48, W=40, seed=5)
    for normalization in ("max", "coil", "abs_max", "stand"):
        got = D.preprocess_slice(ks[1], (32, 24, 1), False, True, normalization, False, "cuda:0").cpu().numpy()
        want = _ref(ks[1], (32, 24, 1), False, normalization)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), (normalization, np.abs(got - want).max())
    got = D.preprocess_slice(ks[0], (32, 24, 1), True, True, None, False, "cuda:0").cpu().numpy()
    want = _ref(ks[0], (32, 24, 1), True, None)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
