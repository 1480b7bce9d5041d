"""Diagnostic: per-phase cycle shares of a fused kernel (needs `make -C csrc dbg`).
Run on the GPU box:  python tools/stamps.py [B] [f32|bf16|mfn|wire|wire2d|siren512]

The stamp buffer holds 64 slots per WAVE of the launch: every fused kernel runs 4 waves per workgroup whatever
its tile (128-coordinate tiles: one wave per 32 coordinates; 64-coordinate tiles: two waves per coordinate
group), so it is sized from that -- not from tile_rows / 32 -- and its length is handed to the library, which
drops any stamp that would fall outside (inr_debug_set_stamp_buffer(buf, entries))."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mri-implicit-neural-representations_amd")
os.environ.setdefault("INR_LIB_PATH", os.path.join(PKG, "lib", "libinr_mi355x_dbg.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
WAVES_PER_WORKGROUP = 4
dev = torch.device("cuda:0")
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
torch.manual_seed(0)
enc = M.Positional_Encoder(enc_cfg, device=dev)
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
lib = L.load()
lib.inr_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_longlong]
if PREC == "mfn":  # BASELINE config 4: MultiscaleKFourier 8x512, LSL + consistency
    from inr_mi355x.mfn import MultiscaleKFourier
    from inr_mi355x.engine import ConsistencySpec
    net = dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512)
    model = MultiscaleKFourier(net).to(dev).bind_encoder(enc)
    eng = model._engine("gauss")
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
    pairs = [(0.0, 0.2), (0.0, 0.45), (0.0, 0.8), (0.0, 5.0)]
    inv = [1.0 / max(1.0, 2.0 * float(((dist < lo) | (dist > hi)).sum())) for lo, hi in pairs[:-1]] + [0.0]
    cons = ConsistencySpec(0.1, pairs, inv, 2)
    spec = M.LossSpec(L.LOSS_LOGSPACE, eps=3e-3)

    def step():
        eng.train_step(coords, enc.B.contiguous(), gt, spec, dist=dist, scale=0.5, cons=cons)
elif PREC == "siren512":  # the reference's shipped config_siren_kspace.yaml: SIREN depth 8 / width 512, gauss-512
    net = dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512, last_tanh=True)
    model = M.SIREN(net).to(dev)
    eng = model.fused_engine(256)

    def step():
        eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
elif PREC == "wire2d":  # the reference's shipped config_wire2d_kspace.yaml: WIRE2D depth 3 / width 256 (256 complex), L2
    net = dict(network_input_size=3, network_output_size=2, network_depth=3, network_width=256, first_omega_0=30,
               hidden_omega_0=30, scale=15)
    model = M.WIRE2D(net).to(dev)
    eng = model._engine()

    def step():
        eng.train_step(coords, None, gt, M.LossSpec(L.LOSS_L2_HALF))
elif PREC == "wire":  # BASELINE config 3: WIRE depth 4 / width 256 (181 complex features), HDR
    net = dict(network_input_size=3, network_output_size=2, network_depth=4, network_width=256, first_omega_0=30,
               hidden_omega_0=30, scale=15)
    model = M.WIRE(net).to(dev)
    eng = model._engine()

    def step():
        eng.train_step(coords, None, gt, M.LossSpec(L.LOSS_HDR), hdr_A=0.3)
else:
    net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
    model = M.SIREN(net).to(dev)
    eng = model.fused_engine(256, precision=PREC)

    def step():
        eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
nt, nb = eng.launch_dims(B)
NWV = 8 if PREC == "bf16" else WAVES_PER_WORKGROUP  # (the bf16 kernel: eight waves, two tile groups)
dbg = torch.zeros(nb * NWV * 64, dtype=torch.int64, device=dev)
lib.inr_debug_set_stamp_buffer(dbg.data_ptr(), dbg.numel())
import time
for _ in range(3):
    step()
torch.cuda.synchronize()
# >= 2.5 s of back-to-back steps on random data before the launch whose stamps are read: the clock a kernel holds under load
# is reached only after seconds (MI355X_MICROARCH.md, DVFS (6)); every launch overwrites the stamps, the last one stays
t_warm = time.perf_counter()
while time.perf_counter() - t_warm < 2.5:
    for _ in range(50):
        step()
    torch.cuda.synchronize()
if PREC == "bf16":  # wall time of the fused kernel alone (grads = NULL) in this build: cycles / time = the clock it ran at
    ws = eng._ws(*eng.workspace(B)); ld = eng.loss_desc(M.LossSpec(L.LOSS_L2_HALF), B)
    def fused_only():
        L.check(lib.inr_train_step(eng.plan, C.byref(ld), eng.params.data_ptr(), eng.packed.data_ptr(), coords.data_ptr(),
                                   enc.B.contiguous().data_ptr(), gt.data_ptr(), None, B, C.byref(ws), None,
                                   eng._loss_word.data_ptr(), eng._stream()))
    for _ in range(200):
        fused_only()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        fused_only()
    e1.record(); torch.cuda.synchronize()
    KERNEL_US = e0.elapsed_time(e1) / 200 * 1e3
lib.inr_debug_set_stamp_buffer(None, 0)
d = dbg.cpu().view(nb, NWV, 64).double()
if PREC == "mfn":
    names = {0: "start", 1: "stage 0"}
    names.update({1 + i: f"fwd stage {i}" for i in range(1, 9)})
    names[12] = "heads+loss"
    order = [0] + list(range(1, 10)) + [12]
    for i in range(8, 0, -1):
        names[12 + 4 * i + 1] = f"bwd {i}: head dX+park"
        names[12 + 4 * i + 2] = f"bwd {i}: dW head, g->stash"
        names[12 + 4 * i + 3] = f"bwd {i}: dX"
        order += [12 + 4 * i + 1, 12 + 4 * i + 2, 12 + 4 * i + 3]
    names[50] = "stage 0 g_u -> stash"
    order.append(50)
    last = 50
elif PREC in ("wire", "wire2d", "siren512"):  # inr_mlp_wide_kernel: stamps 0, 1, 2, ... in program order
    NH = {"wire": 4, "wire2d": 3, "siren512": 6}[PREC]  # hidden layers behind the first
    labels = ["start", "L0 GEMM", "L0 epilogue"]
    for l in range(1, NH + 1):
        labels += [f"sync + L{l} GEMM", f"sync + L{l} epilogue"]
    labels += ["sync + last layer + loss (wave 0 of a pair)", "sync + dZ_last -> image", "sync + dW last", "dX last"]
    for l in range(NH, 0, -1):
        labels += [f"sync + dH_{l} -> image", f"sync + dZ_{l} = J dH_{l}", f"sync + dX L{l}", f"dZ_{l} -> stash"]
    labels += ["sync + dZ_0 = J dH_0 -> image", "sync + dW_0 passes"]
    names = dict(enumerate(labels))
    order = list(range(len(labels)))
    last = len(labels) - 1
else:
    names = {0: "start", 1: "fwd L0", 2: "fwd L1", 3: "fwd L2", 4: "fwd L3", 10: "fwd last+loss", 11: "sync", 12: "dW last",
             13: "dX last+store", 26: "dX L3", 29: "dZ3 -> stash", 22: "dX L2", 25: "dZ2 -> stash", 18: "dX L1",
             21: "dZ1 -> stash", 41: "dZ0 -> stash", 42: "sync"}
    # (the 256-row builds leave dW of the hidden-width layers to inr_dw_gemm.hip: no dW phases in the kernel)
    order = [0, 1, 2, 3, 4, 10, 11, 12, 13, 26, 29, 22, 25, 18, 21, 41, 42]
    last = 42
    if PREC != "f32":  # inr_siren_bf16_kernel: stamps 0, 1, 2, ... in program order
        labels = ["start", "L0 GEMM (+ features)", "L0 epilogue"]
        labels += [f"L{l} (row blocks: GEMM | epilogue)" for l in (1, 2, 3)]
        labels += ["last layer + loss", "dH last + dZ_3 epilogues"]
        labels += [f"dX L{l} (row blocks: GEMM | epilogue)" for l in (3, 2, 1)]
        labels += ["the last epilogue (row block 7 of dZ_0)"]
        names = dict(enumerate(labels))
        order = list(range(len(labels)))
        last = len(labels) - 1
tot = (d[:, :, last] - d[:, :, 0])
rt = d[:, :, 63] - d[:, :, 62]
ok = (rt > 0) & (tot > 0)
ghz = (tot[ok] / rt[ok] * 0.1)
rt0, rt1 = d[:, :, 62][d[:, :, 62] > 0], d[:, :, 63][d[:, :, 63] > 0]
print(f"100 MHz counter, all waves of the launch: first stamp of the earliest wave -> last stamp of the latest "
      f"{(rt1.max() - rt0.min()) / 100:.1f} us; first stamps spread over {(rt0.max() - rt0.min()) / 100:.1f} us, last stamps over "
      f"{(rt1.max() - rt1.min()) / 100:.1f} us")
if PREC == "bf16":  # slots 60 / 61: the counter at kernel entry / exit of each wave
    e0, e1 = d[:, :, 60], d[:, :, 61]
    print(f"kernel entry of the earliest wave -> exit of the latest: {(e1.max() - e0.min()) / 100:.1f} us; entry -> first stamp "
          f"(tables, first panels, barrier): median {float((d[:, :, 62] - e0).median()) / 100:.2f} us; last stamp -> exit (drain, "
          f"scale state, loss word): median {float((e1 - d[:, :, 63]).median()) / 100:.2f} us; entries spread over "
          f"{(e0.max() - e0.min()) / 100:.2f} us")
if d.shape[0] >= 8:  # workgroup b runs on XCD b % 8 (round-robin dispatch): do the slow waves share an XCD?
    q = torch.quantile(tot.flatten(), torch.tensor([0.5, 0.9, 0.99], dtype=tot.dtype))
    print(f"cycles/wave quantiles: median {q[0]:.0f}  p90 {q[1]:.0f}  p99 {q[2]:.0f}  max {tot.max():.0f}")
    wg_end = d[:, :, 63].max(dim=1).values
    for x in range(8):
        sel = torch.arange(d.shape[0]) % 8 == x
        t_x, r_x = tot[sel], rt[sel]
        print(f"  XCD {x}: cycles/wave mean {t_x.mean():.0f} max {t_x.max():.0f}; clock {float((t_x / r_x).mean()) * 0.1:.3f} GHz; "
              f"last workgroup ends {(wg_end[sel].max() - rt0.min()) / 100:.1f} us after the launch's first stamp")
print(f"in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz, first to last stamp of a wave, after >= 2.5 s of launches): "
      f"median {ghz.median():.3f} GHz, min {ghz.min():.3f}, max {ghz.max():.3f} over {int(ok.sum())} waves")
print(f"B={B} blocks={nb} (last tile of each) total cycles/wave: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
if PREC == "bf16":
    span = float((d[:, :, last].max(dim=1).values - d[:, :, 0].min(dim=1).values).mean())
    print(f"fused kernel alone: {KERNEL_US:.1f} us per launch; first stamp -> last stamp of a workgroup {span:.0f} cycles: "
          f">= {span / KERNEL_US / 1e3:.2f} GHz (the launch also loads its tables and drains)")
prev = order[0]
for i in order[1:]:
    seg = d[:, :, i] - d[:, :, prev]
    print(f"  {names[i]:>24s}: mean {seg.mean():9.0f}  per-wave means {[round(float(seg[:, w].mean())) for w in range(NWV)]}")
    if d.shape[0] >= 8:
        xs = [float(seg[torch.arange(d.shape[0]) % 8 == x].mean()) for x in range(8)]
        if max(xs) > 1.08 * sorted(xs)[3]:  # an XCD more than 8 % above the median XCD in this phase
            print(f"  {'':>24s}  by XCD {[round(v) for v in xs]}")
    prev = i
