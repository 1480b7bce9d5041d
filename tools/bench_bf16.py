"""Quick timing of the bf16 path (GPU): fused kernel alone, gradient path (fused + dW GEMM + reduction), per batch size.
python tools/bench_bf16.py [B ...]   (HIP events on the current stream, 200 repetitions after 0.6 s of warm-up)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
import ctypes as C
import bench

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = M.Positional_Encoder(bench.CONFIG["encoder"], device=dev)
model = M.SIREN(bench.CONFIG["net"]).to(dev)
eng = model.fused_engine(256, precision="bf16")
encB = enc.B.contiguous()
spec = M.LossSpec(L.LOSS_L2_HALF)


def timed(fn, reps=200, warm=30):
    import time
    t0 = time.perf_counter()
    n = 0
    while n < warm or time.perf_counter() - t0 < 0.6:  # the chip's clocks settle after some 0.5 s of load
        fn()
        n += 1
        if n % 50 == 0:
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


out = {}
for B in [int(x) for x in sys.argv[1:]] or [65536, 25000]:
    g = torch.Generator().manual_seed(B)
    coords = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
    gt = (torch.randn(B, 2, generator=g) * 0.2).to(dev)
    eng.train_step(coords, encB, gt, spec)
    ws = eng._ws(*eng.workspace(B))
    ld = eng.loss_desc(spec, B)

    def fused_only():
        L.check(eng.lib.inr_train_step(eng.plan, C.byref(ld), eng.params.data_ptr(), eng.packed.data_ptr(), coords.data_ptr(),
                                       encB.data_ptr(), gt.data_ptr(), None, B, C.byref(ws), None,
                                       eng._loss_word.data_ptr(), eng._stream()))

    t_f = timed(fused_only)
    t_p = timed(lambda: eng.train_step(coords, encB, gt, spec))
    t_fw = timed(lambda: eng.forward(coords, encB))
    frac = bench.FLOP_PER_SAMPLE * B / (t_p * 1e-6) / 1e12 / 2500.0
    out[B] = dict(fused_us=round(t_f, 1), path_us=round(t_p, 1), gemm_reduce_us=round(t_p - t_f, 1), forward_us=round(t_fw, 1),
                  frac_bf16_peak=round(frac, 4), msamples_s=round(B / t_p, 1))
    print(B, out[B], flush=True)
print(json.dumps(out))
