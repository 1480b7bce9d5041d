"""Diagnostic: where a launch's time goes outside its waves' work -- entry and exit of every wave of the fused kernel and of the
weight-gradient GEMM on the chip-wide 100 MHz counter (diagnostic build: tools/build_dbg_rs.sh inr_dw_gemm inr_dw_gemm_bf16
inr_siren_bf16_m0 inr_siren_bf16_m1 inr_siren_bf16_m2).  Run on the GPU box:  python tools/stamps_launch.py [B] [f32|bf16]

Slots of a wave's 64-entry record: 44 / 45 = entry / exit of the row-split kernel and of both GEMMs (the GEMM of a step runs
behind the fused kernel and overwrites them: the fused kernel is read from launches without the GEMM), 60 / 61 = the bf16 fused
kernel.  Prints, per kernel: first entry -> last exit, the spread of the entries (dispatch) and of the exits (imbalance + drain),
the distribution of per-workgroup durations, and the same by XCD (workgroup b runs on XCD b % 8)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mri-implicit-neural-representations_amd")
os.environ.setdefault("INR_LIB_PATH", os.path.join(PKG, "lib", "libinr_mi355x_dbg.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
enc = M.Positional_Encoder(dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3), device=dev)
torch.manual_seed(0)
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
lib = L.load()
lib.inr_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_longlong]
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
eng = M.SIREN(net).to(dev).fused_engine(256, precision=PREC)
encB = enc.B.contiguous()
spec = M.LossSpec(L.LOSS_L2_HALF)
ws = eng._ws(*eng.workspace(B)); ld = eng.loss_desc(spec, B)
NWV = 8
NBLK = 512
dbg = torch.zeros(NBLK * NWV * 64, dtype=torch.int64, device=dev)


def fused_only():
    L.check(lib.inr_train_step(eng.plan, C.byref(ld), eng.params.data_ptr(), eng.packed.data_ptr(), coords.data_ptr(),
                               encB.data_ptr(), gt.data_ptr(), None, B, C.byref(ws), None, eng._loss_word.data_ptr(),
                               eng._stream()))


def step():
    eng.train_step(coords, encB, gt, spec)


def timed(fn, n=200):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.8:  # (the chip's clocks settle after some 0.5 s of load: a cold figure reads 10 % high)
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def collect(fn, nw, s0, s1):
    dbg.zero_()
    lib.inr_debug_set_stamp_buffer(dbg.data_ptr(), dbg.numel())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.5:  # the clock a kernel holds under load is reached after seconds
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
    lib.inr_debug_set_stamp_buffer(None, 0)
    d = dbg.cpu().view(-1, 64)
    d = d[: (d.shape[0] // nw) * nw].view(-1, nw, 64).double()
    live = (d[:, :, s0] > 0).all(dim=1) & (d[:, :, s1] > 0).all(dim=1)
    if s0 < 60:  # shader-clock counter two slots on: the clock each wave ran at between entry and exit
        clk = ((d[live][:, :, s1 + 2] - d[live][:, :, s0 + 2]) / (d[live][:, :, s1] - d[live][:, :, s0]) * 0.1).flatten()
        print(f"   [entry -> exit clock of the next kernel: median {clk.median():.3f} GHz, min {clk.min():.3f}, max {clk.max():.3f}]")
    return d[live][:, :, s0], d[live][:, :, s1], torch.nonzero(live)[:, 0]


def report(name, us, e0, e1, blocks):
    wg0, wg1 = e0.min(dim=1).values, e1.max(dim=1).values
    dur = (wg1 - wg0) / 100
    q = torch.quantile(dur, torch.tensor([0.5, 0.9, 0.99], dtype=dur.dtype))
    print(f"{name}: {us:.1f} us per launch (HIP events, back to back); {len(blocks)} workgroups; first entry -> last exit "
          f"{(wg1.max() - wg0.min()) / 100:.1f} us; entries spread over {(wg0.max() - wg0.min()) / 100:.1f} us, exits over "
          f"{(wg1.max() - wg1.min()) / 100:.1f} us")
    print(f"   workgroup duration: median {q[0]:.1f}  p90 {q[1]:.1f}  p99 {q[2]:.1f}  max {dur.max():.1f} us; a wave of the "
          f"workgroup exits up to {float(((e1.max(dim=1).values - e1.min(dim=1).values) / 100).max()):.1f} us before its last")
    n = len(blocks)
    tenths = [dur[(blocks >= blocks.min() + (blocks.max() + 1 - blocks.min()) * i // 10) &
                  (blocks < blocks.min() + (blocks.max() + 1 - blocks.min()) * (i + 1) // 10)] for i in range(10)]
    print("   by tenth of the grid, in block order (mean / max us): " +
          "  ".join(f"{float(v.mean()):.1f}/{float(v.max()):.1f}" if len(v) else "-" for v in tenths))
    by = [dur[blocks % 8 == x] for x in range(8)]
    print("   by XCD (mean / max us): " + "  ".join(f"{float(v.mean()):.1f}/{float(v.max()):.1f}" if len(v) else "-" for v in by))


us_fused = timed(fused_only)
us_step = timed(step)
info = L.StepInfo(); L.check(lib.inr_plan_step_info(eng.plan, B, C.byref(info)))
print(f"B={B} {PREC}: fused kernel alone {us_fused:.1f} us, gradient step (fused + GEMM + reduction) {us_step:.1f} us; "
      f"row_split={info.row_split}")
if PREC == "bf16":
    e0, e1, blk = collect(fused_only, 8, 60, 61)
    report("fused bf16 kernel", us_fused, e0, e1, blk)
    e0, e1, blk = collect(step, 8, 44, 45)
    report("dw_gemm_bf16_kernel", us_step - us_fused, e0, e1, blk)
else:
    if info.row_split:
        e0, e1, blk = collect(fused_only, 4, 44, 45)
        report("row-split fused kernel", us_fused, e0, e1, blk)
    e0, e1, blk = collect(step, 4, 44, 45)
    # the row-split kernel stamps the same slots and has the larger grid: its workgroups beyond the GEMM's grid keep their
    # (earlier) stamps -- the GEMM's workgroups are the ones that entered within half a GEMM of the latest entry
    keep = e0.min(dim=1).values > e0.max() - 50.0 * (us_step - us_fused)
    report("dw_gemm_kernel (+ reduction in the per-launch figure)", us_step - us_fused, e0[keep], e1[keep], blk[keep])
