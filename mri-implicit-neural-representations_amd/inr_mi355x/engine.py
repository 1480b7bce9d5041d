"""Host side of the MI355X INR engine: owns the flat parameter / gradient / Adam buffers and the
workspaces (all torch device tensors -- PyTorch is the allocator and the stream provider) and
calls the kernels through the C-ABI (include/inr_abi.h).  No arithmetic of the hot path happens
in this file; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib as L


@dataclass
class LossSpec:
    """Loss selection of train.py:81-98 (+ loss_opts of the YAML configs)."""
    kind: int = L.LOSS_L2_HALF
    eps: float = 1e-3
    sigma: float = 2.0
    factor: float = 0.5
    min_sample: int = 3000  # CenterLoss: pairs per radial band (loss_opts['min_sample'], losses.py:150)

    @staticmethod
    def from_config(config: dict) -> "LossSpec":
        name = config["loss"]
        opts = config.get("loss_opts", {}) or {}
        kinds = {"L2": L.LOSS_L2_HALF, "L1": L.LOSS_L1_HALF, "tanh": L.LOSS_TANH, "HDR": L.LOSS_HDR,
                 "LogSpace": L.LOSS_LOGSPACE, "MSLE": L.LOSS_MSLE_HALF, "LSL": L.LOSS_CENTER}  # train.py:82-96 ('LSL'
        # there is CenterLoss -- its pairs come from torch.randperm on the CPU generator, as here; 'T' / 'FFL' are broken
        # at their call sites: SURVEY A.4 #7, #8)
        if name not in kinds:
            # the reference evaluates `NotImplementedError` without raising (train.py:98); we raise
            raise NotImplementedError(f"loss {name!r} has no MI355X kernel (supported: {sorted(kinds)})")
        return LossSpec(kinds[name], float(opts.get("hdr_eps", 1e-3)), float(opts.get("hdr_ff_sigma", 2.0)),
                        float(opts.get("hdr_ff_factor", 0.5)), int(opts.get("min_sample", 3000)))


def _shape(t: Optional[torch.Tensor], name: str, *shape) -> None:
    """Raise unless ``t`` has exactly this shape (the kernels index by these extents: a short tensor would be an
    out-of-bounds GPU read, a [B,512] tensor passed where [B,3] is expected garbage without an error)."""
    if t is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


def _ptr(t: Optional[torch.Tensor], name: str, dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the INR engine only runs on an MI355X (no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous {dtype} tensor (got {t.dtype}, contiguous={t.is_contiguous()})")
    return t.data_ptr()


def sharded_exchange(gbuf_pad: torch.Tensor, gchunk: torch.Tensor, pchunk: torch.Tensor, pgather: torch.Tensor,
                     params: torch.Tensor, rank: int, world: int, group, update) -> torch.Tensor:
    """The collectives of a sharded data-parallel update (MLPEngine.adam_step_sharded).
    ``gbuf_pad`` [world * chunk] = [gradient (P) | loss word | zero padding] of this rank, chunk = ceil((P + 1) / world).
    reduce-scatter (SUM) -> ``update(lo, hi, gchunk)`` changes params[lo:hi] from gchunk[:hi - lo] (the rank's entries;
    lo == hi for a rank whose chunk is all padding) -> all-gather of [updated entries | what the chunk holds behind them:
    the summed loss on the rank that owns word P, padding] -> params <- gathered[:P].  Returns the summed loss.
    gloo has no reduce-scatter on device tensors: there (CPU-side rehearsals of the GPU path) the same chunk is cut from
    an all-reduce -- the same sums in the same order per element."""
    import torch.distributed as dist
    P, chunk = params.numel(), gchunk.numel()
    assert gbuf_pad.numel() == pgather.numel() == chunk * world and pchunk.numel() == chunk and chunk * world > P
    if gbuf_pad.is_cuda and dist.get_backend(group) == "gloo":
        dist.all_reduce(gbuf_pad, op=dist.ReduceOp.SUM, group=group)
        gchunk.copy_(gbuf_pad[rank * chunk:(rank + 1) * chunk])
    else:
        dist.reduce_scatter_tensor(gchunk, gbuf_pad, op=dist.ReduceOp.SUM, group=group)
    lo = min(rank * chunk, P)
    hi = min(lo + chunk, P)
    update(lo, hi, gchunk)
    pchunk.copy_(gchunk)
    if hi > lo:
        pchunk[:hi - lo].copy_(params[lo:hi])
    dist.all_gather_into_tensor(pgather, pchunk, group=group)
    params.copy_(pgather[:P])
    return pgather[P].clone()


class MLPEngine:
    """One plan + its device buffers.  Mirrors what nn.Module + torch.optim.Adam hold for SIREN / FFN
    in the reference (models/networks.py:48-124; train.py:75-78)."""

    def __init__(self, kind: int, in_features: int, width: int, depth: int, out_features: int, last_act: int,
                 input_mode: int = L.INPUT_X, enc_size: int = 0, w0: float = 30.0, first_omega_0: float = 0.0,
                 hidden_omega_0: float = 0.0, scale_0: float = 0.0, precision: int = L.PRECISION_F32):
        self.lib = L.load()
        desc = L.NetDesc(kind=kind, in_features=in_features, width=width, depth=depth, out_features=out_features,
                         last_act=last_act, input=input_mode, enc_size=enc_size, w0=w0,
                         first_omega_0=first_omega_0, hidden_omega_0=hidden_omega_0, scale_0=scale_0,
                         precision=precision)
        self.desc = desc
        plan = C.c_void_p()
        L.check(self.lib.inr_plan_create(C.byref(desc), C.byref(plan)))
        self.plan = plan
        sz = L.Sizes()
        L.check(self.lib.inr_plan_sizes(plan, C.byref(sz)))
        self.n_params = int(sz.n_params)
        self.packed_floats = int(sz.packed_floats)
        self.tile_rows = int(sz.tile_rows)
        self.save_floats_per_tile = int(sz.save_bytes_per_tile) // 4
        self.max_blocks = int(sz.max_blocks)
        self.slab_floats = int(sz.slab_floats)
        self.step_save_by_tile = bool(sz.step_save_by_tile)
        self.in_features, self.out_features = in_features, out_features
        self.input_mode = input_mode
        self.always_save = kind == L.KIND_WIRE2D  # its orth terms travel through the save buffer even when not training
        self.params: Optional[torch.Tensor] = None
        self.grads = self.exp_avg = self.exp_avg_sq = self.packed = None
        self._save = self._slabs = self._loss = None
        self._ws_generation = 0
        self._sched = self._sched_key = self._step_dev = None
        self._step_dev_mirror = 0
        self.step = 0

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.inr_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    # ---- buffers ------------------------------------------------------------------------------
    def bind(self, flat_params: torch.Tensor) -> None:
        """Adopt a flat fp32 device tensor [P] (state_dict order) as the master weights."""
        _ptr(flat_params, "flat_params")
        if flat_params.numel() != self.n_params:
            raise RuntimeError(f"flat_params has {flat_params.numel()} elements, plan needs {self.n_params}")
        self.params = flat_params
        dev = flat_params.device
        # one extra word behind the gradient: the data-parallel step all-reduces gradient AND loss in ONE collective
        self.gbuf = torch.zeros(self.n_params + 1, device=dev)
        self.grads = self.gbuf[:self.n_params]
        self._loss_word = self.gbuf[self.n_params:]  # the fused steps write their loss here, next to the gradient
        self.exp_avg = torch.zeros(self.n_params, device=dev)
        self.exp_avg_sq = torch.zeros(self.n_params, device=dev)
        self.packed = torch.zeros(self.packed_floats, device=dev)  # padding entries stay zero forever
        self._loss = torch.zeros(L.LOSS_WORDS, device=dev)
        self.step = 0
        if self.desc.precision == L.PRECISION_BF16:
            # the plan's gradient-scale state is allocated by the first call that needs it: make that call here, not inside
            # a training step (which may be under graph capture)
            with torch.cuda.device(dev):
                self.grad_scale_state()
        if getattr(self, "_shard", None) is not None:  # re-bound (checkpoint load): keep the sharded-update buffers
            self.enable_sharded_update(self._shard[0], self._shard[1])
        self.pack()

    def launch_dims(self, B: int):
        nt, nb = C.c_int64(), C.c_int64()
        L.check(self.lib.inr_plan_launch_dims(self.plan, B, C.byref(nt), C.byref(nb)))
        return int(nt.value), int(nb.value)

    def workspace(self, B: int):
        """(stash slots a fused step needs, slabs behind ``slabs``) for a batch of B rows."""
        ss, ns = C.c_int64(), C.c_int64()
        L.check(self.lib.inr_plan_workspace(self.plan, B, C.byref(ss), C.byref(ns)))
        return int(ss.value), int(ns.value)

    def grad_scale_state(self):
        """bf16 plans: the sixteen gradient-scale words (include/inr_abi.h: inr_plan_grad_scale_state) as a list of floats --
        [0..3] fused steps, [4..7] split steps, [8..11] counts of clipped / flushed steps -- after the stream's queued work
        (synchronises).  Tests and diagnostics."""
        buf = (C.c_float * 16)()
        L.check(self.lib.inr_plan_grad_scale_state(self.plan, buf, self._stream()))
        return [float(v) for v in buf]

    def _take_stash(self, B: int) -> None:
        if getattr(self, "_stash_rows", None) != B:
            raise RuntimeError("backward needs the stash of a forward(save=True) on the same batch, and a backward "
                               "consumes it: run the forward again")
        self._stash_rows = None

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.params.device).cuda_stream

    def _ws_save(self, n_slots: int) -> torch.Tensor:
        need = n_slots * self.save_floats_per_tile
        if self._save is None or self._save.numel() < need:
            self._save = torch.zeros(need, device=self.params.device)
            self._ws_generation += 1  # captured step graphs hold the old address
        return self._save

    def _ws_slabs(self, n_blocks: int) -> torch.Tensor:
        need = n_blocks * self.slab_floats
        if self._slabs is None or self._slabs.numel() < need:
            self._slabs = torch.empty(need, device=self.params.device)
            self._ws_generation += 1
        return self._slabs

    def _ws(self, save_slots: int, n_slabs: int):
        """inr_workspace of a call: the buffers AND their extents, which the library checks against the plan."""
        ws = L.Workspace()
        if save_slots > 0:
            sv = self._ws_save(save_slots)
            ws.save, ws.save_floats = _ptr(sv, "save"), sv.numel()
        if n_slabs > 0:
            sl = self._ws_slabs(n_slabs)
            ws.slabs, ws.slab_floats = _ptr(sl, "slabs"), sl.numel()
        return ws

    def _check_input(self, x: torch.Tensor, enc_B: Optional[torch.Tensor]) -> int:
        B = int(x.shape[0])
        if self.input_mode == L.INPUT_GAUSS:
            _shape(x, "coords", B, 3)
            if enc_B is None:
                raise RuntimeError("enc_B is required: this engine fuses the gauss Positional_Encoder")
            _shape(enc_B, "enc_B", self.in_features // 2, 3)
        else:
            _shape(x, "x", B, self.in_features)
        return B

    # ---- kernels ------------------------------------------------------------------------------
    def pack(self) -> None:
        L.check(self.lib.inr_pack_params(self.plan, _ptr(self.params, "params"), _ptr(self.packed, "packed"),
                                         self._stream()))

    def forward(self, x: torch.Tensor, enc_B: Optional[torch.Tensor] = None, save: bool = False) -> torch.Tensor:
        B = self._check_input(x, enc_B)
        nt, _ = self.launch_dims(B)
        out = torch.empty(B, self.out_features, device=x.device)
        saving = save or self.always_save
        ws = self._ws(nt if saving else 0, 0)
        L.check(self.lib.inr_forward(self.plan, _ptr(self.params, "params"), _ptr(self.packed, "packed"),
                                     _ptr(x, "x"), _ptr(enc_B, "enc_B"), B, _ptr(out, "out"),
                                     C.byref(ws), self._stream()))
        self._stash_rows = B if saving else None
        return out

    def backward(self, x: torch.Tensor, enc_B: Optional[torch.Tensor], dout: torch.Tensor) -> torch.Tensor:
        """d(loss)/d(params) for the most recent forward(save=True) on the same x.  Consumes the stash (the
        weight-gradient GEMM's operands overwrite the stashed activation derivatives): one backward per forward."""
        B = self._check_input(x, enc_B)
        _shape(dout, "dout", B, self.out_features)
        self._take_stash(B)
        nt, _ = self.launch_dims(B)
        ws = self._ws(nt, self.workspace(B)[1])
        L.check(self.lib.inr_backward(self.plan, _ptr(self.params, "params"), _ptr(self.packed, "packed"),
                                      _ptr(x, "x"), _ptr(enc_B, "enc_B"), B, _ptr(dout, "dout"),
                                      C.byref(ws), _ptr(self.grads, "grads"), self._stream()))
        return self.grads

    def loss_desc(self, spec: LossSpec, count: int, hdr_A: float = 0.0) -> L.LossDesc:
        return L.LossDesc(kind=spec.kind, eps=spec.eps, sigma=spec.sigma, factor=spec.factor,
                          inv_count=1.0 / float(count), hdr_A=hdr_A)

    def loss_grad(self, spec: LossSpec, out: torch.Tensor, gt: torch.Tensor, count: int,
                  mask: Optional[torch.Tensor] = None, hdr_A: float = 0.0):
        """Tier-1 loss: returns (loss scalar tensor view, dout [B,2])."""
        B = out.shape[0]
        _shape(out, "out", B, 2)
        _shape(gt, "gt", B, 2)
        _shape(mask, "mask", B)
        dout = torch.empty_like(out)
        ld = self.loss_desc(spec, count, hdr_A)
        L.check(self.lib.inr_loss_grad(C.byref(ld), _ptr(out, "out"), _ptr(gt, "gt"), None,
                                       _ptr(mask, "mask", torch.uint8), B, _ptr(self._loss, "loss"),
                                       _ptr(dout, "dout"), self._stream()))
        return self._loss[0], dout

    def loss_tv_grad(self, spec: LossSpec, out: torch.Tensor, gt: torch.Tensor, count: int, rows_own: int, W: int, H: int,
                     mask: Optional[torch.Tensor] = None, hdr_A: float = 0.0, weight: float = 1e-4):
        """loss_grad + tv_grad of R = out.shape[0] / W image rows in one pass (inr_loss_tv_grad): the masked pointwise loss on
        the first rows_own rows, tv_loss (losses.py:326-343) on the grid.  Returns (loss scalar (device), dout)."""
        R = out.shape[0] // W
        assert R * W == out.shape[0] and out.shape[1] == 2
        _shape(gt, "gt", out.shape[0], 2)
        _shape(mask, "mask", out.shape[0])
        dout = torch.empty_like(out)
        ld = self.loss_desc(spec, count, hdr_A)
        L.check(self.lib.inr_loss_tv_grad(C.byref(ld), _ptr(out, "out"), _ptr(gt, "gt"), _ptr(mask, "mask", torch.uint8), R,
                                          rows_own, W, H, C.c_float(weight), _ptr(self._loss, "loss"), _ptr(dout, "dout"),
                                          self._stream()))
        return self._loss[0], dout

    def tv_grad(self, out: torch.Tensor, dout: torch.Tensor, rows_own: int, W: int, H: int,
                weight: float = 1e-4):
        """tv_loss (losses.py:326-343) on out [R*W,2] = R image rows of width W (first rows_own owned,
        see inr_tv_grad): adds its gradient into dout and its value into the loss scalar of the
        preceding loss_grad call.  Returns the loss scalar (device)."""
        R = out.shape[0] // W
        assert R * W == out.shape[0] and out.shape[1] == 2 and dout.shape == out.shape
        L.check(self.lib.inr_tv_grad(_ptr(out, "out"), R, rows_own, W, H, C.c_float(weight),
                                     _ptr(self._loss, "loss"), _ptr(dout, "dout"), self._stream()))
        return self._loss[0]

    def center_pairs_grad(self, out: torch.Tensor, gt: torch.Tensor, dout: torch.Tensor, idx_a: torch.Tensor,
                          idx_b: torch.Tensor, weight: float = 0.1):
        """One radial band of CenterLoss's random-pair term (losses.py:175-199; see inr_center_pairs_grad): idx_a / idx_b
        are int64 row indices into out / gt [B,2].  Adds weight * mean_p r_p^2 to the loss scalar of the preceding
        loss_grad call and its gradient to dout.  Returns the loss scalar (device)."""
        B, n = out.shape[0], idx_a.shape[0]
        _shape(out, "out", B, 2)
        _shape(gt, "gt", B, 2)
        _shape(dout, "dout", B, 2)
        _shape(idx_b, "idx_b", n)
        L.check(self.lib.inr_center_pairs_grad(_ptr(out, "out"), _ptr(gt, "gt"), _ptr(idx_a, "idx_a", torch.int64),
                                               _ptr(idx_b, "idx_b", torch.int64), n, B, C.c_float(weight),
                                               _ptr(self._loss, "loss"), _ptr(dout, "dout"), self._stream()))
        return self._loss[0]

    def train_step(self, x: torch.Tensor, enc_B: Optional[torch.Tensor], gt: torch.Tensor, spec: LossSpec,
                   count: Optional[int] = None, mask: Optional[torch.Tensor] = None, hdr_A: float = 0.0):
        """Fused encode -> forward -> loss -> backward (stages of train.py:163-189).  Leaves the
        un-reduced-across-ranks gradient in self.grads and returns the loss scalar (device)."""
        B = self._check_input(x, enc_B)
        _shape(gt, "gt", B, self.out_features)
        _shape(mask, "mask", B)
        self._stash_rows = None  # the fused step writes (and consumes) the same stash buffer
        ws = self._ws(*self.workspace(B))
        ld = self.loss_desc(spec, B if count is None else count, hdr_A)
        L.check(self.lib.inr_train_step(self.plan, C.byref(ld), _ptr(self.params, "params"),
                                        _ptr(self.packed, "packed"), _ptr(x, "x"), _ptr(enc_B, "enc_B"),
                                        _ptr(gt, "gt"), _ptr(mask, "mask", torch.uint8), B, C.byref(ws),
                                        _ptr(self.grads, "grads"), self._loss_word.data_ptr(), self._stream()))
        return self._loss_word[0]

    def train_adam_step(self, x: torch.Tensor, enc_B: Optional[torch.Tensor], gt: torch.Tensor, spec: LossSpec,
                        lr: float, count: Optional[int] = None, mask: Optional[torch.Tensor] = None,
                        hdr_A: float = 0.0, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                        weight_decay: float = 0.0, l1: float = 0.0, l2: float = 0.0):
        """train_step + adam_step as one call (single-rank steps): the Adam update rides in the slab reduction's launch.
        Bit-identical to the two calls; self.grads still receives the gradient.  Returns the loss scalar (device)."""
        B = self._check_input(x, enc_B)
        _shape(gt, "gt", B, self.out_features)
        _shape(mask, "mask", B)
        self._stash_rows = None
        ws = self._ws(*self.workspace(B))
        ld = self.loss_desc(spec, B if count is None else count, hdr_A)
        self.step += 1
        L.check(self.lib.inr_train_adam_step(self.plan, C.byref(ld), _ptr(self.params, "params"),
                                             _ptr(self.packed, "packed"), _ptr(x, "x"), _ptr(enc_B, "enc_B"),
                                             _ptr(gt, "gt"), _ptr(mask, "mask", torch.uint8), B, C.byref(ws),
                                             _ptr(self.grads, "grads"), self._loss_word.data_ptr(),
                                             _ptr(self.exp_avg, "exp_avg"), _ptr(self.exp_avg_sq, "exp_avg_sq"), lr,
                                             beta1, beta2, eps, weight_decay, l1, l2, self.step, self._stream()))
        return self._loss_word[0]

    def adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                  weight_decay: float = 0.0, l1: float = 0.0, l2: float = 0.0) -> None:
        """torch.optim.Adam.step (train.py:190) on the flat buffers + weight re-pack."""
        self.step += 1
        L.check(self.lib.inr_adam_step(self.plan, _ptr(self.params, "params"), _ptr(self.grads, "grads"),
                                       _ptr(self.exp_avg, "exp_avg"), _ptr(self.exp_avg_sq, "exp_avg_sq"),
                                       _ptr(self.packed, "packed"), lr, beta1, beta2, eps, weight_decay, l1, l2,
                                       self.step, self._stream()))

    def reg_grad(self, l1: float, l2: float, l2_dir: Optional[torch.Tensor] = None, grads: Optional[torch.Tensor] = None,
                 lo: int = 0, hi: Optional[int] = None) -> None:
        """Adds the gradient of Regularization_L1 / _L2 (models/regularization.py:21-36) to ``grads`` (default: the
        engine's gradient; entries [lo, hi), grads[i - lo]) -- the form that is right for complex64 tensors too
        (inr_reg_grad).  ``l2_dir`` = device tensor (Re, Im) of conj(S) / |S|, S = sum of squares of every Parameter."""
        g = self.grads if grads is None else grads
        hi = self.n_params if hi is None else hi
        L.check(self.lib.inr_reg_grad(self.plan, _ptr(self.params, "params"), _ptr(g, "grads"), lo, hi, l1, l2,
                                      None if l2_dir is None else _ptr(l2_dir, "l2_dir"), self._stream()))

    # ---- data-parallel update with the parameters sharded over the ranks ------------------------
    def enable_sharded_update(self, rank: int, world: int) -> None:
        """Buffers of adam_step_sharded: the gradient buffer padded to `world` equal chunks (the loss word still sits
        behind the gradient and travels with it), the rank's chunk of the reduce-scatter, the chunk it contributes to the
        all-gather and the gathered vector.  Call once, before the first step (the gradient buffer is re-allocated)."""
        P = self.n_params
        chunk = -(-(P + 1) // world)
        dev = self.params.device
        self._gbuf_pad = torch.zeros(chunk * world, device=dev)
        self.gbuf = self._gbuf_pad[:P + 1]
        self.grads = self.gbuf[:P]
        self._loss_word = self.gbuf[P:]
        self._gchunk = torch.zeros(chunk, device=dev)
        self._pchunk = torch.zeros(chunk, device=dev)
        self._pgather = torch.zeros(chunk * world, device=dev)
        self._shard = (rank, world, chunk)

    def shard_bounds(self):
        """[lo, hi) of the flat parameter entries this rank updates (empty for a rank whose chunk is all padding)."""
        rank, world, chunk = self._shard
        lo = min(rank * chunk, self.n_params)
        return lo, min(lo + chunk, self.n_params)

    def adam_step_sharded(self, group, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                          weight_decay: float = 0.0, l1: float = 0.0, l2: float = 0.0,
                          reg: Optional[tuple] = None) -> torch.Tensor:
        """The exchange + update of a data-parallel step with every rank owning 1/N of the parameters: reduce-scatter
        of [gradient | loss] (SUM) -> torch.optim.Adam on the rank's entries (inr_adam_step_shard) -> all-gather of the
        updated entries (the summed loss rides in the word behind the last parameter) -> every rank re-packs its weight
        images.  The replicas' parameters come out of the same gathered buffer: bitwise equal on every rank; equal to
        all-reduce + inr_adam_step up to the collective's summation order.  Returns the global loss; `self.grads` is
        left holding this rank's PARTIAL gradient."""
        self.step += 1

        def update(lo: int, hi: int, gchunk: torch.Tensor) -> None:
            if reg is not None and hi > lo:  # (l1, l2, l2_dir): penalties of a model with complex tensors, see reg_grad
                self.reg_grad(reg[0], reg[1], reg[2], grads=gchunk, lo=lo, hi=hi)
            L.check(self.lib.inr_adam_step_shard(self.plan, _ptr(self.params, "params"), _ptr(gchunk, "grads_shard"),
                                                 _ptr(self.exp_avg, "exp_avg"), _ptr(self.exp_avg_sq, "exp_avg_sq"), lo,
                                                 hi, lr, beta1, beta2, eps, weight_decay, l1, l2, self.step,
                                                 self._stream()))

        rank, world, _ = self._shard
        loss = sharded_exchange(self._gbuf_pad, self._gchunk, self._pchunk, self._pgather, self.params, rank, world,
                                group, update)
        self.pack()
        return loss

    # ---- the step as a HIP graph ----------------------------------------------------------------
    N_SCHED = 32768  # both fp32 bias-correction terms have converged long before (include/inr_abi.h)

    def _adam_schedule(self, lr: float, beta1: float, beta2: float) -> torch.Tensor:
        """Device table of (step_size, bc2_sqrt) per step for this learning rate; rebuilt only when lr changes
        (the per-epoch LambdaLR, train.py:153,251).  The copy is ordered on the current stream."""
        key = (float(lr), float(beta1), float(beta2))
        if self._sched_key != key:
            host = np.empty(2 * self.N_SCHED, dtype=np.float32)
            L.check(self.lib.inr_adam_schedule(key[0], key[1], key[2], self.N_SCHED, host.ctypes.data))
            if self._sched is None:
                self._sched = torch.empty(2 * self.N_SCHED, device=self.params.device)
            self._sched.copy_(torch.from_numpy(host))
            self._sched_key = key
        return self._sched

    def _sync_step_dev(self) -> torch.Tensor:
        """The device copy of ``self.step``; refreshed only after eager adam_step calls moved the host count."""
        if self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=self.params.device)
            self._step_dev_mirror = 0
        if self._step_dev_mirror != self.step:
            self._step_dev.fill_(self.step)
            self._step_dev_mirror = self.step
        return self._step_dev

    def adam_step_dev(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                      weight_decay: float = 0.0, l1: float = 0.0, l2: float = 0.0) -> None:
        """adam_step with the step count and the bias corrections read from device memory: the launch has no
        argument that changes between steps (bit-identical results; tests/test_gpu_parity.py)."""
        sched = self._adam_schedule(lr, beta1, beta2)
        sd = self._sync_step_dev()
        L.check(self.lib.inr_adam_step_dev(self.plan, _ptr(self.params, "params"), _ptr(self.grads, "grads"),
                                           _ptr(self.exp_avg, "exp_avg"), _ptr(self.exp_avg_sq, "exp_avg_sq"),
                                           _ptr(self.packed, "packed"), _ptr(sched, "sched"), self.N_SCHED,
                                           _ptr(sd, "step_dev", torch.int32), beta1, beta2, eps, weight_decay, l1,
                                           l2, self._stream()))
        self.step += 1
        self._step_dev_mirror += 1

    def capture_step(self, gradient_step: Callable[[], object], lr: float, beta1: float = 0.9,
                     beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0, l1: float = 0.0,
                     l2: float = 0.0) -> "StepGraph":
        """Capture ``gradient_step()`` (a closure over train_step on FIXED input views) + the Adam update as one
        HIP graph.  ``gradient_step`` runs once eagerly first, so that workspaces exist and nothing is allocated
        while capturing; it only overwrites the gradient buffer."""
        gradient_step()
        self._adam_schedule(lr, beta1, beta2)
        self._sync_step_dev()
        graph = torch.cuda.CUDAGraph()
        step0 = self.step
        with torch.cuda.graph(graph):
            gradient_step()
            self.adam_step_dev(lr, beta1, beta2, eps, weight_decay, l1, l2)
        self.step = self._step_dev_mirror = step0  # capturing launched nothing
        return StepGraph(self, graph, (beta1, beta2))


class StepGraph:
    """One captured step (fused kernel, weight-gradient GEMM, slab reduction, Adam + re-pack, step advance)."""

    def __init__(self, engine: MLPEngine, graph: "torch.cuda.CUDAGraph", betas):
        self.engine, self.graph, self.betas = engine, graph, betas
        self.generation = engine._ws_generation

    @property
    def stale(self) -> bool:
        return self.generation != self.engine._ws_generation

    def replay(self, lr: float) -> torch.Tensor:
        eng = self.engine
        if self.stale:
            raise RuntimeError("the engine's workspaces were re-allocated after this step was captured")
        eng._adam_schedule(lr, *self.betas)
        eng._sync_step_dev()
        self.graph.replay()
        eng.step += 1
        eng._step_dev_mirror += 1
        return eng._loss_word[0]


def encode_gauss(coords: torch.Tensor, enc_B: torch.Tensor) -> torch.Tensor:
    """Positional_Encoder.embedding, 'gauss' (networks.py:30-33) on the device."""
    lib = L.load()
    B, E = coords.shape[0], enc_B.shape[0]
    out = torch.empty(B, 2 * E, device=coords.device)
    L.check(lib.inr_encode_gauss(_ptr(coords, "coords"), _ptr(enc_B, "enc_B"), B, E, _ptr(out, "out"),
                                 torch.cuda.current_stream(coords.device).cuda_stream))
    return out


def encode_logf(coords: torch.Tensor, bands: torch.Tensor) -> torch.Tensor:
    """Positional_Encoder.embedding, 'LogF' (networks.py:24-29) on the device; bands = encoder.B [n,1]."""
    lib = L.load()
    B, nb = coords.shape[0], bands.numel()
    out = torch.empty(B, 6 * nb, device=coords.device)
    L.check(lib.inr_encode_logf(_ptr(coords, "coords"), _ptr(bands.reshape(-1), "bands"), B, nb, _ptr(out, "out"),
                                torch.cuda.current_stream(coords.device).cuda_stream))
    return out


@dataclass
class ConsistencySpec:
    """ConsistencyLoss(pairs) of train_kspace_multiscale.py:122,179 for one batch."""
    weight: float            # 0.1
    bounds: list             # [(lo, hi)] * n_heads  (create_pairs(radii, 1))
    inv_counts: list         # 1 / number of compared ELEMENTS of pair i over the global batch (0 if empty)
    channels: int = 2        # 1 in per-coil mode (dist [B,1] selects channel 0 only)


class MFNEngine(MLPEngine):
    """Plan + buffers for the multiplicative filter networks (models/mfn.py: FourierNet,
    MultiscaleKFourier).  Heads come back as one [n_heads, B, out] tensor."""

    def __init__(self, kind: int, in_features: int, width: int, depth: int, out_features: int, enc_size: int,
                 bounds=None, input_mode: int = L.INPUT_GAUSS):
        """input_mode INPUT_GAUSS: called on raw coordinates [B,3] with the encoder matrix (the gauss encoder is
        fused into the filters); INPUT_X: called on the encoded x [B,in_features] like the reference's forward
        (mfn.py:34-43), enc_B is None."""
        super().__init__(kind, in_features, width, depth, out_features, L.ACT_ID, input_mode,
                         enc_size if input_mode == L.INPUT_GAUSS else 0, 0.0)
        nh = C.c_int32()
        L.check(self.lib.inr_plan_heads(self.plan, C.byref(nh)))
        self.n_heads = int(nh.value)
        if kind == L.KIND_MSBOUNDED:
            lo = (C.c_float * depth)(*[float(b[0]) for b in bounds])
            hi = (C.c_float * depth)(*[float(b[1]) for b in bounds])
            L.check(self.lib.inr_plan_set_bounds(self.plan, lo, hi, depth))

    def forward(self, coords: torch.Tensor, enc_B: torch.Tensor, save: bool = False,
                dist: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = self._check_input(coords, enc_B)
        _shape(dist, "dist", B)
        nt, nb = self.launch_dims(B)
        out = torch.empty(self.n_heads, B, self.out_features, device=coords.device)
        ws = self._ws(nt if save else nb, 0)
        L.check(self.lib.inr_forward_multi(self.plan, _ptr(self.params, "params"), _ptr(self.packed, "packed"),
                                           _ptr(coords, "coords"), _ptr(enc_B, "enc_B"), _ptr(dist, "dist"), B,
                                           _ptr(out, "out"), C.byref(ws), 0 if save else 1, self._stream()))
        self._stash_rows = B if save else None
        return out

    def backward(self, coords: torch.Tensor, enc_B: torch.Tensor, dout: torch.Tensor,
                 dist: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = self._check_input(coords, enc_B)
        _shape(dist, "dist", B)
        _shape(dout, "dout", self.n_heads, B, self.out_features)
        self._take_stash(B)
        nt, nb = self.launch_dims(B)
        ws = self._ws(nt, self.workspace(B)[1])
        L.check(self.lib.inr_backward_multi(self.plan, _ptr(self.params, "params"), _ptr(self.packed, "packed"),
                                            _ptr(coords, "coords"), _ptr(enc_B, "enc_B"), _ptr(dist, "dist"), B,
                                            _ptr(dout, "dout"), C.byref(ws), _ptr(self.grads, "grads"),
                                            self._stream()))
        return self.grads

    def multi_loss_desc(self, spec: LossSpec, count: int, hdr_A: float = 0.0, scale: float = 1.0,
                        cons: Optional[ConsistencySpec] = None) -> L.LossDesc:
        ld = self.loss_desc(spec, count, hdr_A)
        ld.scale = scale
        if cons is not None:
            ld.cons_w = cons.weight
            ld.cons_chan = cons.channels
            for i, (lo, hi) in enumerate(cons.bounds[:4]):
                ld.cons_lo[i], ld.cons_hi[i] = float(lo), float(hi)
            for i, v in enumerate(cons.inv_counts[:4]):
                ld.cons_inv[i] = float(v)
        return ld

    def loss_grad_multi(self, spec: LossSpec, outs: torch.Tensor, gt: torch.Tensor, count: int,
                        mask: Optional[torch.Tensor] = None, dist: Optional[torch.Tensor] = None, scale: float = 1.0,
                        cons: Optional[ConsistencySpec] = None):
        """Tier-1 multiscale loss (train_kspace_multiscale.py:176-195) on outs [n_heads,B,2]: returns (loss scalar
        view, douts [n_heads,B,2]).  Pointwise terms on rows with mask != 0, consistency on every row."""
        NH, B = outs.shape[0], outs.shape[1]
        _shape(outs, "outs", NH, B, 2)
        _shape(gt, "gt", B, 2)
        _shape(mask, "mask", B)
        _shape(dist, "dist", B)
        douts = torch.empty_like(outs)
        ld = self.multi_loss_desc(spec, count, 0.0, scale, cons)
        L.check(self.lib.inr_loss_grad_multi(C.byref(ld), _ptr(outs, "outs"), _ptr(gt, "gt"), _ptr(dist, "dist"),
                                             _ptr(mask, "mask", torch.uint8), NH, B, _ptr(self._loss, "loss"),
                                             _ptr(douts, "douts"), self._stream()))
        return self._loss[0], douts

    def train_step(self, coords: torch.Tensor, enc_B: torch.Tensor, gt: torch.Tensor, spec: LossSpec,
                   count: Optional[int] = None, mask: Optional[torch.Tensor] = None, hdr_A: float = 0.0,
                   dist: Optional[torch.Tensor] = None, scale: float = 1.0,
                   cons: Optional[ConsistencySpec] = None):
        B = self._check_input(coords, enc_B)
        _shape(gt, "gt", B, self.out_features)
        _shape(mask, "mask", B)
        _shape(dist, "dist", B)
        self._stash_rows = None
        ws = self._ws(*self.workspace(B))
        ld = self.multi_loss_desc(spec, B if count is None else count, hdr_A, scale, cons)
        L.check(self.lib.inr_train_step_multi(self.plan, C.byref(ld), _ptr(self.params, "params"),
                                              _ptr(self.packed, "packed"), _ptr(coords, "coords"),
                                              _ptr(enc_B, "enc_B"), _ptr(gt, "gt"), _ptr(dist, "dist"),
                                              _ptr(mask, "mask", torch.uint8), B, C.byref(ws),
                                              _ptr(self.grads, "grads"), self._loss_word.data_ptr(), self._stream()))
        return self._loss_word[0]
