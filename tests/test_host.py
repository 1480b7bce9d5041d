"""CPU-side tests (``-m "not gpu"``): the C-ABI library loads and exports every declared symbol,
argument validation, drop-in classes' construction parity, host logic of the trainer."""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_abi_header_symbols_exported():
    """Every function include/inr_abi.h declares is exported by the built library and bound."""
    from inr_mi355x import _lib
    hdr = open(os.path.join(ROOT, "include", "inr_abi.h")).read()
    declared = set(re.findall(r"^int (inr_\w+)\(", hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.inr_abi_version() == 1


def test_plan_validation_and_sizes():
    from inr_mi355x import _lib as L
    lib = L.load()
    plan = C.c_void_p()
    good = L.NetDesc(kind=L.KIND_SIREN, in_features=512, width=256, depth=5, out_features=2, last_act=L.ACT_TANH,
                     input=L.INPUT_GAUSS, enc_size=256, w0=30.0)
    assert lib.inr_plan_create(C.byref(good), C.byref(plan)) == 0
    sz = L.Sizes()
    assert lib.inr_plan_sizes(plan, C.byref(sz)) == 0
    assert sz.n_params == 329218  # SIREN 5x256/in512
    assert sz.tile_rows == 128 and sz.slab_floats >= sz.n_params + 1
    nt, nb = C.c_int64(), C.c_int64()
    assert lib.inr_plan_launch_dims(plan, 25000, C.byref(nt), C.byref(nb)) == 0
    assert (nt.value, nb.value) == (196, 196)
    assert lib.inr_plan_launch_dims(plan, 65536, C.byref(nt), C.byref(nb)) == 0
    assert (nt.value, nb.value) == (512, 256)
    assert lib.inr_plan_launch_dims(plan, 0, C.byref(nt), C.byref(nb)) < 0
    assert "B = 0" in L.last_error()
    lib.inr_plan_destroy(plan)
    for bad, frag in ((dict(width=100), "width"), (dict(depth=1), "depth"), (dict(out_features=9), "out_features"),
                      (dict(in_features=500), "2*enc_size"), (dict(kind=7), "kind")):
        kw = dict(kind=L.KIND_SIREN, in_features=512, width=256, depth=5, out_features=2, last_act=L.ACT_TANH,
                  input=L.INPUT_GAUSS, enc_size=256, w0=30.0)
        kw.update(bad)
        rc = lib.inr_plan_create(C.byref(L.NetDesc(**kw)), C.byref(plan))
        assert rc < 0 and frag in L.last_error(), (bad, L.last_error())
    # null arguments are rejected before anything touches a GPU
    assert lib.inr_forward(None, None, None, None, None, 10, None, None, None) < 0
    assert lib.inr_adam_step(None, None, None, None, None, None, 1e-3, .9, .999, 1e-8, 0, 0, 0, 1, None) < 0


def _sha(t):
    return hashlib.sha256(t.detach().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["SIREN", "SIREN4", "FFN"])
def test_dropin_init_bit_exact(name):
    """Drop-in constructors consume the RNG like the reference: same seeds -> same SHA-256."""
    import inr_mi355x as M
    ent = json.load(open(os.path.join(GOLD, "init_hashes.json")))[name]
    torch.manual_seed(ent["seed"])
    enc = M.Positional_Encoder(ent["encoder"], device="cpu")
    assert _sha(enc.B) == ent["enc_sha256"]
    model = {"SIREN": M.SIREN, "FFN": M.FFN}[ent["model"]](ent["net"])
    sd = model.state_dict()
    assert set(sd) == set(ent["shapes"])
    for k, v in sd.items():
        assert list(v.shape) == ent["shapes"][k] and _sha(v) == ent["sha256"][k], k
    assert sum(p.numel() for p in model.parameters()) == ent["n_params"]
    # parameters are views of one flat buffer, in state_dict order
    flat = model._flat
    for p, (o, n, s, _c) in zip(model.parameters(), model._layout):
        assert p.data_ptr() == flat.data_ptr() + 4 * o and tuple(p.shape) == s


def test_no_cpu_fallback():
    import inr_mi355x as M
    net = dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32)
    model = M.SIREN(net)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(4, 16))


def test_state_dict_roundtrip_and_checkpoint_keys():
    import inr_mi355x as M
    net = dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32)
    torch.manual_seed(0)
    a = M.SIREN(net)
    torch.manual_seed(1)
    b = M.SIREN(net)
    b.load_state_dict(a.state_dict())
    assert torch.equal(a._flat, b._flat)
    assert list(a.state_dict()) == [f"model.{k}.linear.{n}" for k in range(3) for n in ("weight", "bias")]
    f = M.FFN(net)
    assert list(f.state_dict()) == [f"model.{2 * k}.{n}" for k in range(3) for n in ("weight", "bias")]


def test_shard_rows_and_schedule():
    from inr_mi355x.train import shard_rows, lr_factor
    for lo, hi, world in ((0, 25000, 8), (17500, 25000, 4), (3, 10, 3), (0, 5, 8)):
        parts = [shard_rows(lo, hi, r, world) for r in range(world)]
        assert parts[0][0] == lo and parts[-1][1] == hi
        for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
            assert a1 == b0 and a0 <= a1
    assert lr_factor(0, 10) == 1.0 and abs(lr_factor(10, 10) - 0.2) < 1e-12 and lr_factor(50, 10) == lr_factor(10, 10)


def test_synthetic_kspace_contract():
    from inr_mi355x.synthetic import make_kspace
    import oracle as O
    image, coords, shape = make_kspace(3, 32, 24, seed=7, normalization="coil")
    image2, _, _ = make_kspace(3, 32, 24, seed=7, normalization="coil")
    assert torch.equal(image, image2) and shape == (3, 32, 24)
    assert image.shape == (3 * 32 * 24, 2) and coords.shape == (3 * 32 * 24, 3)
    assert torch.equal(coords, O.create_coords(3, 32, 24))
    mag = O.complex_abs(image.reshape(3, 32, 24, 2)).reshape(3, -1).max(dim=1)[0]
    torch.testing.assert_close(mag, torch.ones(3), rtol=1e-6, atol=0)  # 'coil' normalisation
    im_max, _, _ = make_kspace(3, 32, 24, seed=7, normalization="max")
    assert abs(float(im_max.abs().max()) - 1.0) < 1e-6
    # centred FFT: the spectral peak sits next to the centre of the grid, not at a corner
    k = O.complex_abs(image.reshape(3, 32, 24, 2))[0]
    py, px = np.unravel_index(int(k.argmax()), k.shape)
    assert abs(py - 16) <= 3 and abs(px - 12) <= 3
