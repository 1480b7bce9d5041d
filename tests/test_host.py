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
    assert lib.inr_abi_version() == _lib.ABI_VERSION == 7


def test_adam_schedule_table():
    """inr_adam_schedule (host-only): the (step_size, bc2_sqrt) pairs torch.optim.Adam derives in Python doubles
    (torch/optim/adam.py _single_tensor_adam: step_size = lr / (1 - beta1^t), bias_correction2_sqrt), and both
    have converged in fp32 well inside the 32 768 entries the device-resident step uses."""
    from inr_mi355x import _lib
    lib = _lib.load()
    n, lr, b1, b2 = 32768, 3e-5, 0.9, 0.999
    tab = np.empty(2 * n, dtype=np.float32)
    assert lib.inr_adam_schedule(lr, b1, b2, n, tab.ctypes.data) == 0
    for t in list(range(1, 300)) + [1000, 5000, 16000, 20000, n]:
        assert tab[2 * (t - 1)] == np.float32(lr / (1.0 - b1 ** t)), t
        assert tab[2 * (t - 1) + 1] == np.float32((1.0 - b2 ** t) ** 0.5), t
    assert tab[2 * 20000] == np.float32(lr) and tab[2 * 20000 + 1] == np.float32(1.0)
    assert (tab[2 * 20000:].reshape(-1, 2) == tab[-2:]).all()
    assert lib.inr_adam_schedule(lr, b1, b2, 0, tab.ctypes.data) != 0
    assert lib.inr_adam_schedule(lr, b1, b2, 4, None) != 0


def test_bf16_kernel_inline_asm_memory_hazards():
    """The bf16 kernels' phase loads are inline assembly with hand-placed vmcnt waits (csrc/inr_siren_bf16_impl.h, "vector
    loads the compiler does not see"); the compiler neither knows that their registers are written asynchronously nor pads
    hardware hazards inside assembly text.  Builds have gone wrong three ways: a spilled SGPR reloaded (v_readlane) right in
    front of such a load -- gfx9 needs 5 wait states between a VALU write of an SGPR and a vector-memory read of it; a
    load in flight "moved" by a loop-carried copy; a load nobody reads whose destination the allocator reused.
    tools/check_inflight_regs.py walks every built kernel (three modes x six depths): no instruction may touch a
    register of a load still in flight (in-order vmcnt model), and no vector-memory instruction may read an SGPR a VALU
    instruction wrote fewer than 5 wait states earlier."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs llvm-objdump")
    seen = 0
    for m in (0, 1, 2):
        obj = os.path.join(ROOT, "mri-implicit-neural-representations_amd", "build", f"inr_siren_bf16_m{m}.o")
        if not os.path.exists(obj):
            pytest.skip("needs the built objects")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py"), obj,
                            "inr_siren_bf16_kernel"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        seen += r.stdout.count(" 0 touch a register in flight")
        assert r.stdout.count(" 0 touch a register in flight") == r.stdout.count(" 0 vector-memory instructions read an SGPR") == 6
    assert seen == 18


def test_no_valu_sgpr_to_vmem_hazard_in_any_kernel():
    """The same SGPR hazard check over every built kernel object (compiler-generated memory instructions are padded by
    the compiler; this guards the ones it cannot see and costs ten seconds)."""
    import glob
    import subprocess
    import sys
    objs = sorted(glob.glob(os.path.join(ROOT, "mri-implicit-neural-representations_amd", "build", "inr_*.o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built objects and llvm-objdump")
    for obj in objs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py"), obj, "kernel",
                            "--sgpr-only"], capture_output=True, text=True)
        assert r.returncode == 0, os.path.basename(obj) + "\n" + r.stdout + r.stderr


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
    # workspace of a step: this plan's weight gradients come from the batch GEMM -> per-tile stash, chunk slabs
    slots, slabs = C.c_int64(), C.c_int64()
    assert sz.step_save_by_tile == 1
    assert lib.inr_plan_workspace(plan, 25000, C.byref(slots), C.byref(slabs)) == 0
    # 256 x 256 tiles: 5 per chunk -> 51 chunks of 4 tiles = K 512 each: short, so 128 x 256 tiles (10 per chunk) over
    # twice the K: 25 chunks of 8 tiles.  The fused step of this plan is the row-split kernel (inr_mlp_rs_impl.h): the 1568
    # column blocks of 16 coordinates go to 256 workgroups (7 or 6 each), so 256 workgroup slabs, not one per tile
    assert (slots.value, slabs.value) == (196, 256 + 25)
    assert lib.inr_plan_workspace(plan, 65536, C.byref(slots), C.byref(slabs)) == 0
    assert slots.value == 512 and 256 < slabs.value <= 256 + 52
    # which fused kernel: 25 000 rows = 1568 column blocks on 256 workgroups, 32 tiles of 7 and 224 of 6; 65 536 rows fill
    # inr_mlp_kernel's two rounds of 256 tiles exactly and stay there; INR_RS forces either
    info = L.StepInfo()
    assert lib.inr_plan_step_info(plan, 25000, C.byref(info)) == 0
    assert (info.row_split, info.ncb, info.grid, info.rounds, info.hi, info.lo, info.n_hi) == (1, 7, 256, 1, 7, 6, 32)
    assert lib.inr_plan_step_info(plan, 65536, C.byref(info)) == 0
    assert (info.row_split, info.grid, info.rounds) == (0, 256, 2)
    os.environ["INR_RS"] = "1"
    try:
        assert lib.inr_plan_step_info(plan, 65536, C.byref(info)) == 0
        assert (info.row_split, info.ncb, info.rounds, info.hi, info.lo, info.n_hi) == (1, 6, 3, 6, 4, 512)
        for B in (1, 100, 129, 4133, 100000, 235520, 3532800):  # every block of every slot is dealt exactly once
            assert lib.inr_plan_step_info(plan, B, C.byref(info)) == 0
            T = info.grid * info.rounds
            assert 1 <= info.ncb <= 7 and info.hi == info.ncb and (info.lo == info.hi or info.lo % 2 == 0)
            assert info.n_hi * info.hi + (T - info.n_hi) * info.lo == 8 * ((B + 127) // 128)
    finally:
        del os.environ["INR_RS"]
    assert lib.inr_plan_workspace(plan, 100, C.byref(slots), C.byref(slabs)) == 0
    assert (slots.value, slabs.value) == (1, 8 + 1)  # one slot = 8 column blocks = 8 workgroups; one GEMM chunk
    assert lib.inr_plan_workspace(plan, 0, C.byref(slots), C.byref(slabs)) < 0
    lib.inr_plan_destroy(plan)
    small = L.NetDesc(kind=L.KIND_SIREN, in_features=16, width=32, depth=4, out_features=2, last_act=L.ACT_TANH,
                      input=L.INPUT_GAUSS, enc_size=8, w0=30.0)  # narrow nets keep their in-kernel dW passes
    assert lib.inr_plan_create(C.byref(small), C.byref(plan)) == 0
    assert lib.inr_plan_sizes(plan, C.byref(sz)) == 0 and sz.step_save_by_tile == 0
    assert lib.inr_plan_workspace(plan, 65536, C.byref(slots), C.byref(slabs)) == 0
    assert (slots.value, slabs.value) == (256, 256)
    lib.inr_plan_destroy(plan)
    for bad, frag in ((dict(width=513), "width"), (dict(width=0), "width"), (dict(depth=1), "depth"), (dict(out_features=9), "out_features"),
                      (dict(in_features=500), "2*enc_size"), (dict(kind=99), "kind")):
        kw = dict(kind=L.KIND_SIREN, in_features=512, width=256, depth=5, out_features=2, last_act=L.ACT_TANH,
                  input=L.INPUT_GAUSS, enc_size=256, w0=30.0)
        kw.update(bad)
        rc = lib.inr_plan_create(C.byref(L.NetDesc(**kw)), C.byref(plan))
        assert rc < 0 and frag in L.last_error(), (bad, L.last_error())
    # null arguments are rejected before anything touches a GPU
    assert lib.inr_forward(None, None, None, None, None, 10, None, None, None) < 0
    assert lib.inr_adam_step(None, None, None, None, None, None, 1e-3, .9, .999, 1e-8, 0, 0, 0, 1, None) < 0


def test_bf16_plan_sizes_and_chunk_classes(monkeypatch):
    """A bf16 plan is created and sized without a GPU.  Workspace of a step = one stash slot per 128-row tile + the fused
    kernel's slabs (one per workgroup, two tiles each above 256 tiles) + the weight-gradient GEMM's chunk slabs: the larger
    of its two chunk classes (csrc/inr_api.hip: dw_gemm_bf16_setup -- first-layer units 1.5 x the chunks of the others,
    about 256 workgroups in all)."""
    from inr_mi355x import _lib as L
    lib = L.load()
    monkeypatch.delenv("INR_GEMM_ONE_CLASS", raising=False)
    monkeypatch.delenv("INR_GEMM_ENC_COST", raising=False)
    plan = C.c_void_p()
    d = L.NetDesc(kind=L.KIND_SIREN, in_features=512, width=256, depth=5, out_features=2, last_act=L.ACT_TANH,
                  input=L.INPUT_GAUSS, enc_size=256, w0=30.0, precision=L.PRECISION_BF16)
    assert lib.inr_plan_create(C.byref(d), C.byref(plan)) == 0, L.last_error()
    sz = L.Sizes()
    assert lib.inr_plan_sizes(plan, C.byref(sz)) == 0
    assert sz.tile_rows == 128 and sz.save_bytes_per_tile == 4 * (2 * 4 * 8192 + 2 * 128 + 4 * 128)  # 8-bit stash: inr_w2.h
    nt, nb, slots, slabs = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    assert lib.inr_plan_launch_dims(plan, 65536, C.byref(nt), C.byref(nb)) == 0 and (nt.value, nb.value) == (512, 256)
    assert lib.inr_plan_launch_dims(plan, 25000, C.byref(nt), C.byref(nb)) == 0 and (nt.value, nb.value) == (196, 196)

    def chunks(nt, target):
        tpc = -(-nt // max(1, target))
        return -(-nt // tpc)

    for B, ntiles, nblocks in ((65536, 512, 256), (25000, 196, 196), (100, 1, 1)):
        per = 256.0 / (1.5 * 2 + 4)  # two first-layer units (256 frequencies), three hidden + the last layer
        n_enc = chunks(ntiles, int(per * 1.5))
        n_oth = chunks(ntiles, (256 - 2 * n_enc) // 4)
        assert lib.inr_plan_workspace(plan, B, C.byref(slots), C.byref(slabs)) == 0
        assert (slots.value, slabs.value) == (ntiles, nblocks + max(n_enc, n_oth)), (B, slabs.value, n_enc, n_oth)
        assert 2 * n_enc + 4 * n_oth <= 256
    # the chunking knobs are read when a plan is CREATED: an existing plan's workspace sizes do not follow the environment
    monkeypatch.setenv("INR_GEMM_ONE_CLASS", "1")
    assert lib.inr_plan_workspace(plan, 65536, C.byref(slots), C.byref(slabs)) == 0
    assert slabs.value == 256 + max(chunks(512, int(256.0 / 7 * 1.5)), chunks(512, (256 - 2 * chunks(512, int(256.0 / 7 * 1.5))) // 4))
    plan1 = C.c_void_p()
    assert lib.inr_plan_create(C.byref(d), C.byref(plan1)) == 0, L.last_error()
    assert lib.inr_plan_workspace(plan1, 65536, C.byref(slots), C.byref(slabs)) == 0
    assert slabs.value == 256 + chunks(512, 256 // 6)
    lib.inr_plan_destroy(plan1)
    lib.inr_plan_destroy(plan)
    # what the bf16 kernels do not cover is refused at plan creation, not run on something else
    for bad, frag in ((dict(width=512), "bf16"), (dict(width=64, in_features=512), "bf16"), (dict(depth=9), "bf16"),
                      (dict(enc_size=40, in_features=80), "bf16"), (dict(input=L.INPUT_X), "bf16")):
        kw = dict(kind=L.KIND_SIREN, in_features=512, width=256, depth=5, out_features=2, last_act=L.ACT_TANH,
                  input=L.INPUT_GAUSS, enc_size=256, w0=30.0, precision=L.PRECISION_BF16)
        kw.update(bad)
        rc = lib.inr_plan_create(C.byref(L.NetDesc(**kw)), C.byref(plan))
        assert rc < 0 and frag in L.last_error().lower(), (bad, rc, L.last_error())


def test_short_workspaces_are_invalid_arguments():
    """A stash or slab buffer shorter than inr_plan_workspace reports is INR_ERR_INVALID at the ABI -- checked on the
    host, before any launch -- never an out-of-bounds GPU write (pointers here are never dereferenced)."""
    from inr_mi355x import _lib as L
    lib = L.load()
    fake = 0x1000  # non-null; the size checks come before any use
    for kind, kw in ((L.KIND_SIREN, dict(in_features=512, width=256, depth=5, enc_size=256, input=L.INPUT_GAUSS, w0=30.0,
                                        last_act=L.ACT_TANH)),
                     (L.KIND_MSFOURIER, dict(in_features=512, width=512, depth=8, enc_size=256, input=L.INPUT_GAUSS)),
                     (L.KIND_MSFOURIER, dict(in_features=512, width=512, depth=8, input=L.INPUT_X))):
        plan = C.c_void_p()
        assert lib.inr_plan_create(C.byref(L.NetDesc(kind=kind, out_features=2, **kw)), C.byref(plan)) == 0, L.last_error()
        sz = L.Sizes()
        lib.inr_plan_sizes(plan, C.byref(sz))
        B = 25000
        slots, n_slabs = C.c_int64(), C.c_int64()
        assert lib.inr_plan_workspace(plan, B, C.byref(slots), C.byref(n_slabs)) == 0
        need_save = slots.value * (sz.save_bytes_per_tile // 4)
        need_slab = n_slabs.value * sz.slab_floats
        ld = L.LossDesc(kind=L.LOSS_L2_HALF, inv_count=1.0 / B)
        multi = kind != L.KIND_SIREN

        def step(ws):
            if multi:
                return lib.inr_train_step_multi(plan, C.byref(ld), fake, fake, fake, fake, fake, fake, None, B,
                                                C.byref(ws), fake, fake, None)
            return lib.inr_train_step(plan, C.byref(ld), fake, fake, fake, fake, fake, None, B, C.byref(ws), fake, fake,
                                      None)

        for ws, frag in ((L.Workspace(fake, need_save - 1, fake, need_slab), "stash"),
                         (L.Workspace(fake, need_save, fake, need_slab - 1), "slab"),
                         (L.Workspace(None, 0, fake, need_slab), "stash"),
                         (L.Workspace(fake, need_save, None, 0), "slab")):
            assert step(ws) == -1 and frag in L.last_error(), (kind, frag, L.last_error())
        nt, nb = C.c_int64(), C.c_int64()
        lib.inr_plan_launch_dims(plan, B, C.byref(nt), C.byref(nb))
        short = L.Workspace(fake, nt.value * (sz.save_bytes_per_tile // 4) - 1, fake, need_slab)
        if multi:
            assert lib.inr_backward_multi(plan, fake, fake, fake, fake, fake, B, fake, C.byref(short), fake, None) == -1
            assert lib.inr_forward_multi(plan, fake, fake, fake, fake, fake, B, fake, C.byref(short), 0, None) == -1
        else:
            assert lib.inr_backward(plan, fake, fake, fake, fake, B, fake, C.byref(short), fake, None) == -1
            assert lib.inr_forward(plan, fake, fake, fake, fake, B, fake, C.byref(short), None) == -1
        assert "stash" in L.last_error()
        lib.inr_plan_destroy(plan)


def test_mfn_plans_accept_the_encoded_input():
    """INR_INPUT_X for the filter networks (the reference's call contract, mfn.py:34-43): any in_features, no
    encoder; the feature image in the stash spans in_features rounded up to whole 32-row blocks."""
    from inr_mi355x import _lib as L
    lib = L.load()
    for in_f in (3, 16, 42, 512):
        for kind in (L.KIND_FOURIER, L.KIND_GABOR, L.KIND_MSBOUNDED):
            plan = C.c_void_p()
            d = L.NetDesc(kind=kind, in_features=in_f, width=48, depth=3, out_features=2, input=L.INPUT_X)
            assert lib.inr_plan_create(C.byref(d), C.byref(plan)) == 0, L.last_error()
            lib.inr_plan_destroy(plan)
    plan = C.c_void_p()
    bad = L.NetDesc(kind=L.KIND_FOURIER, in_features=500, width=48, depth=3, out_features=2, input=L.INPUT_GAUSS, enc_size=256)
    assert lib.inr_plan_create(C.byref(bad), C.byref(plan)) < 0 and "2*enc_size" in L.last_error()


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


# ---- undersampling masks (undersampling/undersampler.py) vs vectors made by the reference ------------
def _us_gold():
    g = dict(np.load(os.path.join(GOLD, "undersampling.npz")))
    meta = json.load(open(os.path.join(GOLD, "undersampling_meta.json")))
    return g, meta


def test_radial_mask_matches_reference():
    from inr_mi355x import undersampling as U
    g, meta = _us_gold()
    for key, (H, W, acc) in {"radial_640x368_acc4": (640, 368, 4), "radial_65x48_acc2": (65, 48, 2)}.items():
        ref = np.unpackbits(g[key])[:H * W].reshape(H, W).astype(bool)
        got = U.radial_mask(H, W, acc, t=meta["radial_t"])
        assert np.array_equal(got.numpy(), ref), key
        assert np.array_equal(U.radial_mask(H, W, acc, seed=meta["radial_seed"]).numpy(), ref), key
    m = U.radial_mask(640, 368, 4, t=meta["radial_t"])
    assert 3.5 < U.acceleration_factor(m) < 4.5


def test_grid_apply_matches_reference():
    from inr_mi355x import undersampling as U
    from inr_mi355x.synthetic import create_coords
    g, meta = _us_gold()
    C, H, W = meta["shape"]
    k = torch.from_numpy(g["grid_masked"])
    us = U.Undersampler("grid")
    masked, grid, gm = us.apply(k, meta["grid"])  # masking twice is idempotent
    assert torch.equal(masked, k)
    assert torch.equal(grid, torch.from_numpy(g["grid_coords"]))
    assert np.array_equal(gm.numpy(), g["grid_mask"])
    assert int(us.mask_image.sum()) == len(range(0, H, meta["grid"][0])) * len(range(0, W, meta["grid"][1]))
    assert torch.equal(grid, create_coords(C, H, W))


def test_random_line_and_argument_grammar():
    from inr_mi355x import undersampling as U
    assert U.parse_undersampling_argument(None) == (None, [])
    assert U.parse_undersampling_argument("none") == ("none", [])
    assert U.parse_undersampling_argument("grid-3*2") == ("grid", [3, 2])
    assert U.parse_undersampling_argument("random_line-0.5") == ("random_line", [0.5])
    assert U.parse_undersampling_argument("radial-4") == ("radial", [4])
    with pytest.raises(AssertionError):
        U.parse_undersampling_argument("grid-3")
    with pytest.raises(AssertionError):
        U.Undersampler("spiral")
    # rows drawn before columns from torch's global RNG, as the reference does (undersampler.py:100-101)
    torch.manual_seed(3)
    m = U.random_line_mask(12, 9, 0.5)
    torch.manual_seed(3)
    rows, cols = torch.rand(12) <= 0.5, torch.rand(9) <= 0.5
    ref = torch.zeros(12, 9, dtype=torch.bool)
    ref[rows, :] = True
    ref[:, cols] = True
    assert torch.equal(m, ref)
    assert U.random_line_mask(8, 8, 1.0).all()


# ---- ring partition (clustering.py) vs vectors produced by the reference's functions -------------------
def test_ring_partition_matches_reference():
    from inr_mi355x import clustering as Cl
    g = dict(np.load(os.path.join(GOLD, "clustering.npz")))
    meta = json.load(open(os.path.join(GOLD, "clustering_meta.json")))
    for tag, c in meta["cases"].items():
        k, grid = torch.from_numpy(g[f"{tag}/kspace"]), torch.from_numpy(g[f"{tag}/coords"])
        labels, radii = Cl.partition_kspace(k, grid, c["no_steps"], c["no_parts"])
        assert np.array_equal(labels, g[f"{tag}/labels"]), tag
        np.testing.assert_allclose(radii, g[f"{tag}/radii"], rtol=0, atol=0)
        stats, radii2 = Cl.partition_and_stats(k, grid, c["no_steps"], c["no_parts"], "max")
        np.testing.assert_array_equal(radii2, radii)
        torch.testing.assert_close(stats, torch.from_numpy(g[f"{tag}/stats"]), rtol=0, atol=0)
        assert radii[0] == 0 and radii[-1] == 5 and np.all(np.diff(radii) > 0)


def test_ring_ensemble_assignment_and_overwrite_order():
    """Ring -> rank map and the 'outer ring wins on shared boundary points' rule of the evaluation sweep
    (train_variations/train_clustering.py:199-211: sequential batch_rec[ind] = output)."""
    from inr_mi355x.train_ring_ensemble import ring_owner, winning_ring
    assert [ring_owner(i, 4) for i in range(4)] == [0, 1, 2, 3]
    assert [ring_owner(i, 2) for i in range(4)] == [0, 1, 0, 1]
    radii = [0.0, 0.5, 1.0, 5.0]
    dist = torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0, 1.3, 7.0])
    win = winning_ring(dist, radii)
    rec = torch.full((7,), -1)
    for i in range(3):  # the reference's sequential overwrite
        rec[(dist >= radii[i]) & (dist <= radii[i + 1])] = i
    assert torch.equal(win, rec) and win.tolist() == [0, 0, 1, 1, 2, 2, -1]
