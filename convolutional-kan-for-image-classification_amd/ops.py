"""Autograd operators over the libkanconv C ABI.

Two differentiable entry points, both launching only hand-written HIP kernels:

* ``kan_conv``           -- the fused "expand to basis planes + convolve" stage
                            (reference: kan_layers.py:199-239, fast_kan_layers.py:103-109,
                            cheby_kan_layers.py:93-97).
* ``kan_conv_in_prelu``  -- the same followed by InstanceNorm2d [+ PReLU]
                            (kan_layers.py:241-243, cheby_kan_layers.py:98).
* ``instance_norm``      -- InstanceNorm2d alone (fast_kan_layers.py:106, norm on the input).

All tensors are fp32, NCHW, on a ROCm device.  Groups (kan_layers.py:249-258 loops over them in
Python) run inside the SAME launches (KanGeom.groups; the group index is folded into the grid):
the per-group weights are stacked once per call, activations are never copied or concatenated.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from collections import OrderedDict
from dataclasses import dataclass
from functools import lru_cache
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L


@dataclass(frozen=True)
class ConvSpec:
    """Static description of one layer's conv stage (hashable: used as a plan-cache key)."""
    kind: int
    n_basis: int
    order: int
    act: int                      # L.ACT_NONE => no base branch
    p0: float
    p1: float
    table: Tuple[float, ...]
    kernel: Tuple[int, int]
    stride: Tuple[int, int]
    padding: Tuple[int, int]
    dilation: Tuple[int, int]
    groups: int = 1

    @property
    def has_base(self) -> bool:
        return self.act != L.ACT_NONE

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = self.kernel, self.stride, self.padding, self.dilation
        return (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1


def _basis_struct(spec: ConvSpec) -> L.KanBasis:
    b = L.KanBasis()
    b.kind, b.n_basis, b.order, b.act, b.p0, b.p1 = spec.kind, spec.n_basis, spec.order, spec.act, spec.p0, spec.p1
    if len(spec.table) > L.KAN_MAX_TABLE:
        raise L.KanConvError(f"basis table of {len(spec.table)} entries exceeds KAN_MAX_TABLE={L.KAN_MAX_TABLE}")
    for i, v in enumerate(spec.table):
        b.table[i] = v
    return b


@lru_cache(maxsize=512)
def _plan_cached(spec: ConvSpec, B: int, Cg: int, H: int, W: int, Og: int, C_total: int, O_total: int):
    Ho, Wo = spec.out_hw(H, W)
    if Ho <= 0 or Wo <= 0:
        raise L.KanConvError(f"empty output ({Ho}x{Wo}) for input {H}x{W}")
    g = L.KanGeom()
    g.B, g.C, g.H, g.W, g.O, g.Ho, g.Wo = B, Cg, H, W, Og, Ho, Wo
    (g.kh, g.kw), (g.sh, g.sw), (g.ph, g.pw), (g.dh, g.dw) = spec.kernel, spec.stride, spec.padding, spec.dilation
    g.x_bstride, g.y_bstride = C_total * H * W, O_total * Ho * Wo
    g.groups = spec.groups
    b = _basis_struct(spec)
    p = L.KanPlan()
    L.check(L.load().kan_plan(C.byref(g), C.byref(b), C.byref(p)), "kan_plan")
    return g, b, p


def _with_phases(basis: L.KanBasis, phases: Optional[torch.Tensor], mode: int = 0) -> L.KanBasis:
    """Per-call copy of a cached basis struct carrying the device pointer of the ReLU-KAN phase table and the plane mode."""
    if phases is None:
        return basis
    b = L.KanBasis.from_buffer_copy(basis)
    b.chan_table, b.order = phases.data_ptr(), mode
    return b


def _ptr(t: Optional[torch.Tensor], offset: int = 0) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr() + 4 * offset)


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise L.KanConvError(f"{name} is on {t.device}: this path runs only on a ROCm device (MI355X); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise L.KanConvError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _split_weights(spec: ConvSpec, weights: Sequence[torch.Tensor]):
    G = spec.groups
    if spec.has_base:
        return list(weights[:G]), list(weights[G:2 * G])
    return [None] * G, list(weights[:G])


# --------------------------------------------------------------------------------------- optional kernel timing
# bench.py sets PROFILE to a list; each conv-kernel launch then appends a KernelSample with HIP events recorded on the
# launch stream (torch's current stream is the stream handed to the C ABI).
PROFILE: Optional[list] = None


@dataclass
class KernelSample:
    name: str                 # kernel family + output-tile tag
    flops: float              # dense algorithmic FLOPs of the launch
    executed: float           # FLOPs the MFMA pipe really runs: structural zeros skipped by position-major launches removed
    layer: str                # "C->O@HxW kKxK" of the launch
    start: "torch.cuda.Event"
    end: "torch.cuda.Event"


def _launch(name: str, flops: float, t: torch.Tensor, fn, executed: Optional[float] = None, layer: str = "") -> None:
    if PROFILE is None:
        L.check(fn(), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(t.device)
    e0.record(stream)
    L.check(fn(), name)
    e1.record(stream)
    PROFILE.append(KernelSample(name, flops, flops if executed is None else executed, layer, e0, e1))


def _conv_flops(geom, plan) -> float:
    """Dense algorithmic FLOPs of one conv-stage launch: 2*B*O*Ho*Wo*C*P*kh*kw per group."""
    return 2.0 * geom.B * geom.O * geom.Ho * geom.Wo * geom.C * plan.P * geom.kh * geom.kw * max(1, geom.groups)


@lru_cache(maxsize=256)
def _live_fraction(which: str, H: int, W: int, Ho: int, Wo: int, kh: int, kw: int, sh: int, sw: int, ph: int, pw: int, dh: int, dw: int) -> float:
    """Share of the (position, tap) products that touch a real input pixel (the rest multiply zero padding).  The
    position-major launches skip exactly those (kanconv.hip: live_taps_out / live_taps_in / live_positions_for_tap)."""
    live = 0
    for ho in range(Ho):
        for wo in range(Wo):
            for r in range(kh):
                for t in range(kw):
                    hi, wi = ho * sh - ph + r * dh, wo * sw - pw + t * dw
                    live += 0 <= hi < H and 0 <= wi < W
    return live / float(Ho * Wo * kh * kw)


def _executed_flops(geom, plan, which: str) -> float:
    """Dense count minus the dead (position, tap) products a position-major launch skips (plan.*_target > 0 marks one)."""
    target = {"fwd": plan.fwd_target, "bwd_data": plan.bwd_data_target, "bwd_weight": plan.bwd_weight_target}[which]
    dense = _conv_flops(geom, plan)
    if target <= 0:
        # row-ordered 4x4 launches skip the blocks of the first / last row under the tap row that leaves the plane: 2/3 * 1/4 of the work
        if plan.row_blocks & {"fwd": 1, "bwd_data": 2}.get(which, 0):
            return dense * (5.0 / 6.0)
        return dense
    g = geom
    return dense * _live_fraction(which, g.H, g.W, g.Ho, g.Wo, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.dh, g.dw)


def _layer_tag(geom) -> str:
    return f"{geom.C * max(1, geom.groups)}->{geom.O * max(1, geom.groups)}@{geom.H}x{geom.W} k{geom.kh}x{geom.kw}"


def _tile_tag(plan) -> str:
    return "128" if plan.Opad % 128 == 0 else "64"


def _fwd_name(plan) -> str:
    if plan.fwd_band:                                 # band kernel tiles: 128 outputs where they fit, 192 for exactly 192, else 64 (kan_direct.hip)
        return "k_band_fwd/o" + ("128" if plan.Opad % 128 == 0 else "192" if plan.Opad == 192 else "64")
    return ("k_conv_fwd_halo/o" if plan.fwd_halo else "k_conv_fwd/o") + _tile_tag(plan)


# --------------------------------------------------------------------------------------- raw stages
def _position_major(t: torch.Tensor, ch_off: int, Cn: int) -> torch.Tensor:
    """[Cn*H*W][B] copy of channels [ch_off, ch_off+Cn) of an NCHW tensor (kan_position_major)."""
    B, Ct, H, W = t.shape
    out = torch.empty(Cn * H * W * B, device=t.device, dtype=torch.float32)
    L.check(L.load().kan_position_major(_ptr(t, ch_off * H * W), _ptr(out), B, Cn, H * W, Ct * H * W, _stream(t)), "kan_position_major")
    return out


def _stack(ws: Sequence[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
    """Per-group weights -> one [G, ...] block (a view for G == 1)."""
    if ws[0] is None:
        return None
    return ws[0].unsqueeze(0) if len(ws) == 1 else torch.stack(list(ws))


# --------------------------------------------------------------------------------------- persistent packed weights
# The kernels read the weights in two GEMM layouts (wp forward, wd bwd-data) that only change when the weights do.  Single-group
# layers keep both layouts alive between calls and re-pack on demand:
#   * host side, a change of tensor identity, storage address or autograd version counter (every in-place op, optimizer step,
#     load_state_dict; FusedAdamW bumps the counters of the parameters it updates through raw pointers) forces a re-pack;
#   * device side, every call fingerprints the reference-layout weights (kan_pack_weights_cached: one pass over the bytes,
#     ~20 us for an 85 MB layer) and the pack kernels return at once when the fingerprint equals the previous call's.  This is
#     what catches writes the version counter cannot see -- `p.data.mul_()`, `dist.broadcast(p.data)`, EMA idioms, any
#     raw-pointer writer.  KAN_PACK_SAMPLE=n (> 1) fingerprints every n-th group of four elements instead of all of them.
# KAN_PACK_CACHE=0 disables the cache (pack on every call); otherwise it is the number of (layer, packed layout) entries kept (LRU),
# bounded in bytes as well by KAN_PACK_CACHE_BYTES (default 16 GiB of the 288 GB).  An entry is keyed by what FIXES the packed layouts
# (channels, outputs, planes, kernel, step shape, halo pair order) -- not by batch size or resolution, so a last partial batch, another
# evaluation batch size or variable-size inputs share one copy of each layer's layouts instead of adding ~0.66 GB per shape for KAN-VGG11.
_PACKED: "OrderedDict[tuple, PackedWeights]" = OrderedDict()
_PACK_CACHE_MAX = int(os.environ.get("KAN_PACK_CACHE", "256"))
_PACK_CACHE_BYTES = int(os.environ.get("KAN_PACK_CACHE_BYTES", str(16 << 30)))
_PACK_SAMPLE = max(1, int(os.environ.get("KAN_PACK_SAMPLE", "1")))
# How often the device-side fingerprint runs for a layer whose HOST stamps (tensor identity, address, version counter) have not moved:
# 1 = on every call (332 MB of weight reads + two early-exit pack launches per KAN-VGG11 step: 0.19 ms of 15.4); N = on every N-th call --
# a write the version counters cannot see (`p.data.mul_()`, a raw copy into the storage) is then picked up at the next verification at the
# latest, and at once after `weights_changed()`; 0 = never (stamps only).  Every write autograd / the optimizers can see (in-place ops,
# load_state_dict, torch.optim, FusedAdamW) moves a stamp and re-packs on the next call whatever this is set to.
_PACK_VERIFY_EVERY = int(os.environ.get("KAN_PACK_VERIFY_EVERY", "16"))
_EPOCH = 0
PACK_STATS = {"calls": 0, "forced": 0, "skipped": 0}  # host-side counters (tests, tools)


_ALWAYS_PACK = False


@contextlib.contextmanager
def always_pack():
    """Inside: every conv stage packs its weights unconditionally (the pack kernels are enqueued whatever the host-side stamps say).  For code that
    RECORDS launches for later replay -- a HIP graph (`train.GraphedStep`): the decision "weights unchanged, skip the pack" is taken on the host at
    record time and would be frozen into the graph, so a replay after an optimizer step would multiply stale layouts."""
    global _ALWAYS_PACK
    prev, _ALWAYS_PACK = _ALWAYS_PACK, True
    try:
        yield
    finally:
        _ALWAYS_PACK = prev


_SPLIT_INFERENCE = False
_SPLIT_NOW = False
_SPLIT_CACHE: "OrderedDict" = OrderedDict()             # (weight addresses) -> (version stamps, epoch, cut weights)


@contextlib.contextmanager
def split_precision_inference():
    """OPT-IN, inference only: inside, a fused conv + InstanceNorm + PReLU stage whose inputs need no gradient (torch.no_grad() / eval serving) and whose
    geometry is in scope of `kan_conv_fwd_split` (default B-spline spec on 8x8 / 16x16 planes, 3x3 / stride 1 / pad 1, one group, C % 8 == 0, O % 128 == 0,
    even batch on 8x8) runs its conv stage in split precision (3 x bf16 pieces, six bf16 MFMA products; ~4e-6 of the largest pre-norm value from the exact result, 1.7x
    faster).  Everything else -- every other layer, and every training step -- stays on exact fp32 MFMA.  Never on by default."""
    global _SPLIT_INFERENCE
    prev, _SPLIT_INFERENCE = _SPLIT_INFERENCE, True
    try:
        yield
    finally:
        _SPLIT_INFERENCE = prev


def _split_stage(spec: ConvSpec, x: torch.Tensor, w_base, w_basis):
    """[1, B, O, H, W] pre-norm slab through kan_conv_fwd_split, or None when the stage is out of its scope.  Cut weights are cached per weight pair."""
    if spec.groups != 1 or not spec.has_base or x.dim() != 4:
        return None
    lib = L.load()
    B, Ct, H, W = x.shape
    wb, ws = w_base[0], w_basis[0]
    geom, basis, _ = _plan_cached(spec, B, Ct, H, W, ws.shape[0], Ct, ws.shape[0])
    if not lib.kan_split_supported(C.byref(geom), C.byref(basis)):
        return None
    key = (wb.data_ptr(), ws.data_ptr(), x.device.index)
    stamps = (_owner(wb)._version, _owner(ws)._version, _EPOCH)
    ent = _SPLIT_CACHE.get(key)
    wc = ent[1] if ent is not None and ent[0] == stamps else None
    z, wc = kan_conv_fwd_split(spec, x, wb, ws, wc)
    _SPLIT_CACHE[key] = (stamps, wc)
    _SPLIT_CACHE.move_to_end(key)
    while len(_SPLIT_CACHE) > 16:
        _SPLIT_CACHE.popitem(last=False)
    return z.unsqueeze(0)


def weights_changed() -> None:
    """Tell the packed-weight cache that weights may have been written through a path the version counters cannot see (`.data` ops,
    `dist.broadcast(p.data)`, raw-pointer kernels): every layer fingerprints its weights on its next call."""
    global _EPOCH
    _EPOCH += 1


class PackedWeights:
    __slots__ = ("wp", "wd", "ring", "cur", "stamps", "epoch", "unverified")

    def __init__(self):
        self.wp = self.wd = self.ring = None
        self.cur, self.stamps = 0, None
        self.epoch, self.unverified = -1, 0           # weights_changed() epoch of the last device-side check; calls since then

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.wp, self.wd) if t is not None)


def _layout_key(spec: "ConvSpec", geom, plan) -> tuple:
    """What the packed layouts wp / wd depend on (kanconv.hip: wp_row, k_pack_bwd_data): the per-group channel / output counts, the
    step shape (IPC, KC), the padded dims, and whether the forward uses the halo kernels' pair order.  B, H, W enter only through those."""
    return (spec, geom.C, geom.O, plan.P, plan.IPC, plan.KC, plan.Kpad, plan.Opad, plan.fwd_halo, plan.fwd_band, plan.packed_weight_bytes, plan.bwd_data_weight_bytes)


@lru_cache(maxsize=512)
def _cacheable(plan_key) -> bool:
    geom, basis, _ = _plan_cached(*plan_key)
    return bool(L.load().kan_pack_cacheable(C.byref(geom), C.byref(basis)))


def _owner(w: torch.Tensor) -> torch.Tensor:
    """The tensor that owns the storage (1-D layers hand in `weight.unsqueeze(2)` views, fresh objects on every call)."""
    return w._base if w._base is not None else w


def _packed_entry(key, owners):
    ent = _PACKED.get(key)
    if ent is None:
        ent = PackedWeights()
        _PACKED[key] = ent
        for o in owners:                              # drop the layouts when a weight tensor dies
            weakref.finalize(o, _PACKED.pop, key, None)
        while len(_PACKED) > _PACK_CACHE_MAX or (len(_PACKED) > 1 and sum(e.nbytes() for e in _PACKED.values()) > _PACK_CACHE_BYTES):
            _PACKED.popitem(last=False)
    else:
        _PACKED.move_to_end(key)
    return ent


def _pack(lib, spec, geom, basis, plan, plan_key, w_base, w_basis, need_dgrad, phases, device, st):
    """Returns (wp, wd or None): packed afresh, or the layer's persistent layouts brought up to date."""
    def fresh():
        wp = torch.empty(plan.packed_weight_bytes // 4, device=device, dtype=torch.float32)
        wd = torch.empty(plan.bwd_data_weight_bytes // 4, device=device, dtype=torch.float32) if need_dgrad else None
        wb_all, ws_all = _stack(w_base), _stack(w_basis)          # keep the stacked copies alive until the pack is enqueued
        L.check(lib.kan_pack_weights(_ptr(wb_all), _ptr(ws_all), _ptr(wp), _ptr(wd), C.byref(geom), C.byref(basis), st), "kan_pack_weights")
        return wp, wd
    if _PACK_CACHE_MAX <= 0 or spec.groups != 1 or phases is not None or not _cacheable(plan_key):
        return fresh()
    ws, wb = w_basis[0], w_base[0]
    if (ws.data_ptr() | (wb.data_ptr() if wb is not None else 0)) & 15:
        return fresh()
    owners = [_owner(w) for w in (wb, ws) if w is not None]
    if not all(o.is_leaf for o in owners):            # a per-call temporary of the autograd graph (weight slices made contiguous, re-ordered
        return fresh()                                # plane-major weights): an entry would be built and dropped on every call
    ent = _packed_entry((_layout_key(spec, geom, plan), device.index) + tuple(id(o) for o in owners), owners)
    stamps = [(w.data_ptr(), o._version, tuple(w.shape)) for w, o in zip((w for w in (wb, ws) if w is not None), owners)]
    force = _ALWAYS_PACK or ent.wp is None or ent.stamps != stamps or (need_dgrad and ent.wd is None)
    if ent.wp is None:
        ent.wp = torch.empty(plan.packed_weight_bytes // 4, device=device, dtype=torch.float32)
        ent.ring = torch.zeros(L.KAN_FP_WORDS, device=device, dtype=torch.int64)
    if need_dgrad and ent.wd is None:
        ent.wd = torch.empty(plan.bwd_data_weight_bytes // 4, device=device, dtype=torch.float32)
    if not force and ent.epoch == _EPOCH and _PACK_VERIFY_EVERY != 1 and (_PACK_VERIFY_EVERY == 0 or ent.unverified + 1 < _PACK_VERIFY_EVERY):
        ent.unverified += 1                           # host stamps unchanged and verified recently: no fingerprint, no pack launch
        PACK_STATS["skipped"] += 1
        return ent.wp, (ent.wd if need_dgrad else None)
    ent.epoch, ent.unverified = _EPOCH, 0
    if force:                                         # a graph that saved the old layouts must not be differentiated any more:
        for t in (ent.wp, ent.wd):                    # autograd's saved-tensor version check raises, as it would for the weights
            if t is not None:
                torch.autograd.graph.increment_version(t)
    # once the bwd-data layout exists it is kept in step with wp on every call (a call that skipped it could re-pack wp alone)
    L.check(lib.kan_pack_weights_cached(_ptr(wb), _ptr(ws), _ptr(ent.wp), _ptr(ent.wd), C.byref(geom), C.byref(basis),
                                        C.c_void_p(ent.ring.data_ptr()), ent.cur, int(force), _PACK_SAMPLE, st), "kan_pack_weights_cached")
    PACK_STATS["calls"] += 1
    PACK_STATS["forced"] += int(force)
    ent.cur = (ent.cur + 1) % 3
    ent.stamps = stamps
    return ent.wp, (ent.wd if need_dgrad else None)


def _conv_forward(spec: ConvSpec, x, xn, w_base, w_basis, need_dgrad: bool = True, phases=None):
    """Returns (z_slabs [S,B,O,Ho,Wo], (bwd-data weight layout or None, position-major x or None), geom, basis, plan)."""
    lib = L.load()
    B, Ct, H, W = x.shape
    G = spec.groups
    Cg, Og = Ct // G, w_basis[0].shape[0]
    Ot = Og * G
    plan_key = (spec, B, Cg, H, W, Og, Ct, Ot)
    geom, basis, plan = _plan_cached(*plan_key)
    basis = _with_phases(basis, phases)
    Ho, Wo = geom.Ho, geom.Wo
    st = _stream(x)
    z = torch.empty((plan.fwd_splits, B, Ot, Ho, Wo), device=x.device, dtype=torch.float32)
    wp, wd = _pack(lib, spec, geom, basis, plan, plan_key, w_base, w_basis, need_dgrad, phases, x.device, st)
    if plan.fwd_expanded and xn is None:
        # small padded planes: the expanded position-major operand (kept for the weight gradient), DMA + MFMA forward
        e_pm = _expand_pm(x, geom, basis, plan, st)
        _launch("k_conv_fwd_pmdma/o" + _tile_tag(plan), _conv_flops(geom, plan), x,
                lambda: lib.kan_conv_fwd_expanded(_ptr(e_pm), _ptr(wp), _ptr(z), C.byref(geom), C.byref(basis), st),
                _executed_flops(geom, plan, "fwd"), _layer_tag(geom))
        # what the backward keeps: the expanded copy (its weight gradient reads it), or -- when that launch still runs the
        # tap-major position-major kernel -- the plain copy it needs
        return z, (wd, _position_major(x, 0, Ct) if plan.x_pm_wanted else e_pm), geom, basis, plan
    x_pm = _position_major(x, 0, Ct) if (plan.x_pm_wanted and xn is None) else None
    _launch(_fwd_name(plan), _conv_flops(geom, plan), x,
            lambda: lib.kan_conv_fwd(_ptr(x), _ptr(xn if xn is not None else x), _ptr(wp), _ptr(z), C.byref(geom), C.byref(basis),
                                     _ptr(x_pm), st), _executed_flops(geom, plan, "fwd") if (x_pm is not None or plan.row_blocks & 1) else None, _layer_tag(geom))
    return z, (wd, x_pm), geom, basis, plan


def _expand_pm(x, geom, basis, plan, st):
    e_pm = torch.empty(plan.e_pm_elems, device=x.device, dtype=torch.float32)
    L.check(L.load().kan_position_major_expanded(_ptr(x), _ptr(e_pm), C.byref(geom), C.byref(basis), st), "kan_position_major_expanded")
    return e_pm


# --------------------------------------------------------------------------------------- gradient sinks (data parallel)
# parallel/dp.py registers, per weight Parameter, the slice of its all-reduce bucket.  The weight-gradient unpack kernel
# then writes the reference-layout gradient straight into the bucket and autograd adopts that view as .grad: the copy
# "gradient -> bucket" (332 MB read + write per KAN-VGG11 step) disappears.  Single-group layers only (grouped layers
# stack their per-group gradients in one tensor and keep the copy).
class GradSink:
    """One registered sink: `view` is the bucket slice shaped like the Parameter.  `used` is set when a launch has written
    through it and cleared by the reducer's finish(): a Parameter that feeds TWO graph nodes of one backward pass (a shared
    layer) must not have both nodes write the same buffer -- autograd would then sum two aliases of the last gradient."""
    __slots__ = ("ref", "view", "used")

    def __init__(self, ref, view):
        self.ref, self.view, self.used = ref, view, False


GRAD_SINKS: "dict[int, GradSink]" = {}             # id(Parameter) -> its sink


def _sink_for(wid, shape, device):
    ent = GRAD_SINKS.get(wid) if wid is not None else None
    if ent is None or ent.used:
        return None
    p, t = ent.ref(), ent.view
    # only a FIRST gradient may be written in place: with .grad already set autograd accumulates, and the sink would
    # have overwritten the running sum
    if p is None or p.grad is not None or tuple(t.shape) != tuple(shape) or t.device != device or not t.is_contiguous():
        return None
    ent.used = True
    return t


def _conv_backward(spec: ConvSpec, x, xn, packed, dz, need_x: bool, need_xn: bool, need_w: bool, phases=None, mode: int = 0, wids=None, dparams=None):
    """dz: [B,O,Ho,Wo] contiguous.  Returns (dx, dxn, dw_base list, dw_basis list) -- per-group views of stacked gradients.
    `wids`: ids of the (base, basis) weight Parameters of a single-group layer, for GRAD_SINKS."""
    lib = L.load()
    B, Ct, H, W = x.shape
    G = spec.groups
    Cg = Ct // G
    Ot = dz.shape[1]
    Og = Ot // G
    geom, basis, plan = _plan_cached(spec, B, Cg, H, W, Og, Ct, Ot)
    basis = _with_phases(basis, phases, mode)
    kh, kw = spec.kernel
    st = _stream(x)
    xs = xn if xn is not None else x
    wd, x_pm = packed
    dw_base: List[Optional[torch.Tensor]] = [None] * G
    dw_basis: List[Optional[torch.Tensor]] = [None] * G
    dz_pm = _position_major(dz, 0, Ot) if (plan.dz_pm_wanted and xn is None) else None
    if need_w:
        dwp = torch.empty(plan.bwd_weight_splits * plan.bwd_weight_slab_elems, device=x.device, dtype=torch.float32)
        if plan.bwd_weight_expanded and xn is None and dz_pm is not None:
            # small padded planes: expanded position-major operand, DMA + MFMA weight gradient (kanconv.h)
            e_pm = x_pm if (plan.fwd_expanded and not plan.x_pm_wanted and x_pm is not None and mode == 0) else _expand_pm(x, geom, basis, plan, st)   # the forward's copy, if it kept one (value planes only)
            _launch("k_conv_bwd_weight_pmdma/o" + _tile_tag(plan), _conv_flops(geom, plan), x,
                    lambda: lib.kan_conv_bwd_weight_expanded(_ptr(dz_pm), _ptr(e_pm), _ptr(dwp), C.byref(geom), C.byref(basis), st),
                    _executed_flops(geom, plan, "bwd_weight"), _layer_tag(geom))
        else:
            _launch(_fwd_name(plan).replace("fwd", "bwd_weight") if plan.bwd_weight_band else
                    ("k_conv_bwd_weight_halo/o" if (plan.bwd_weight_halo and xn is None) else "k_conv_bwd_weight/o") + _tile_tag(plan), _conv_flops(geom, plan), x,
                    lambda: lib.kan_conv_bwd_weight(_ptr(dz), _ptr(x), _ptr(xs), _ptr(dwp), C.byref(geom), C.byref(basis), _ptr(x_pm),
                                                    _ptr(dz_pm), st),
                    _executed_flops(geom, plan, "bwd_weight") if (x_pm is not None and dz_pm is not None) else None, _layer_tag(geom))
        sb = _sink_for(wids[0], (Og, Cg, kh, kw), x.device) if (wids and G == 1 and spec.has_base) else None
        ss = _sink_for(wids[1], (Og, Cg * spec.n_basis, kh, kw), x.device) if (wids and G == 1) else None
        dwb = (sb.unsqueeze(0) if sb is not None else
               torch.empty((G, Og, Cg, kh, kw), device=x.device, dtype=torch.float32)) if spec.has_base else None
        dws = ss.unsqueeze(0) if ss is not None else torch.empty((G, Og, Cg * spec.n_basis, kh, kw), device=x.device, dtype=torch.float32)
        L.check(lib.kan_unpack_wgrad(_ptr(dwp), _ptr(dwb), _ptr(dws), C.byref(geom), C.byref(basis), st), "kan_unpack_wgrad")
        dw_basis = list(dws.unbind(0))
        if spec.has_base:
            dw_base = list(dwb.unbind(0))
    dx = dxn = None
    if need_x or need_xn:
        S = plan.bwd_data_splits
        separate = xn is not None
        dxs = torch.empty((S, B, Ct, H, W), device=x.device, dtype=torch.float32)
        dxns = torch.empty_like(dxs) if separate else None
        _launch("k_conv_bwd_data", _conv_flops(geom, plan), x,
                (lambda: lib.kan_conv_bwd_data(_ptr(dz), _ptr(x), _ptr(xs), _ptr(wd), _ptr(dxs), _ptr(dxns) if separate else C.c_void_p(0),
                                               C.byref(geom), C.byref(basis), _ptr(dz_pm), st)) if dparams is None else
                (lambda: lib.kan_conv_bwd_data_params(_ptr(dz), _ptr(x), _ptr(xs), _ptr(wd), _ptr(dxs), _ptr(dxns) if separate else C.c_void_p(0),
                                                      _ptr(dparams), C.byref(geom), C.byref(basis), _ptr(dz_pm), st)),
                _executed_flops(geom, plan, "bwd_data") if (dz_pm is not None or plan.row_blocks & 2) else None, _layer_tag(geom))
        dx, dxn = _sum_slabs(dxs, B, Ct, H * W), (_sum_slabs(dxns, B, Ct, H * W) if separate else None)
    return dx, dxn, dw_base, dw_basis


def _sum_slabs(slabs: torch.Tensor, B: int, Cn: int, HW: int) -> torch.Tensor:
    """[S,B,Cn,...] partial slabs -> their sum (kan_slab_reduce); S == 1 is a view."""
    if slabs.shape[0] == 1:
        return slabs[0]
    out = torch.empty_like(slabs[0])
    L.check(L.load().kan_slab_reduce(_ptr(slabs), slabs.shape[0], slabs[0].numel(), _ptr(out), B, Cn, HW, Cn * HW, _stream(slabs)),
            "kan_slab_reduce")
    return out


def _unflatten(layout, tensors):
    """Inverse of the save_for_backward flattening of the (wd, x_pm) pair."""
    it = iter(tensors)
    has_wd, has_xp = layout
    return (next(it) if has_wd else None, next(it) if has_xp else None)


def _flat_grads(spec: ConvSpec, dw_base, dw_basis):
    return tuple(dw_base) + tuple(dw_basis) if spec.has_base else tuple(dw_basis)


# --------------------------------------------------------------------------------------- autograd
class _KanConv(torch.autograd.Function):
    """z = conv stage.  args: spec, x, xn (or None), *[w_base_g...], *[w_basis_g...]"""

    @staticmethod
    def forward(ctx, spec: ConvSpec, x, xn, *weights):
        x = _require(x, "x")
        xn = _require(xn, "xn") if xn is not None else None
        weights = [_require(w, "weight") for w in weights]
        w_base, w_basis = _split_weights(spec, weights)
        need_dgrad = bool(ctx.needs_input_grad[1] or (xn is not None and ctx.needs_input_grad[2]))
        with torch.cuda.device(x.device):
            z, packed, geom, _, plan = _conv_forward(spec, x, xn, w_base, w_basis, need_dgrad)
            z = _sum_slabs(z, geom.B, z.shape[2], geom.Ho * geom.Wo)
        ctx.spec, ctx.has_xn = spec, xn is not None
        ctx.layout = (packed[0] is not None, packed[1] is not None)
        ctx.save_for_backward(x, *([xn] if xn is not None else []), *[t for t in packed if t is not None])
        return z

    @staticmethod
    def backward(ctx, dz):
        saved = ctx.saved_tensors
        x = saved[0]
        xn = saved[1] if ctx.has_xn else None
        packed = _unflatten(ctx.layout, list(saved[1 + int(ctx.has_xn):]))
        need_x, need_xn = ctx.needs_input_grad[1], ctx.has_xn and ctx.needs_input_grad[2]
        need_w = any(ctx.needs_input_grad[3:])
        with torch.cuda.device(x.device):
            dx, dxn, dwb, dws = _conv_backward(ctx.spec, x, xn, packed, dz.contiguous(), need_x, need_xn, need_w)
        return (None, dx if need_x else None, dxn if need_xn else None) + _flat_grads(ctx.spec, dwb, dws)


def _is_depthwise(spec: ConvSpec, x, w_basis) -> bool:
    """Layers the library runs on its direct depthwise kernels (C = 1 per group, O <= 2, <= 9 taps): no G tile, no epilogue accumulation."""
    return x.shape[1] // spec.groups == 1 and w_basis[0].shape[0] <= 2 and spec.kernel[0] * spec.kernel[1] <= 9


class _KanConvPhased(torch.autograd.Function):
    """Conv stage of a basis with trainable per-channel parameters (ReLU-KAN: relu_kan_layers.py:118-136).
    args: spec, x, xn (or None: the basis reads x), phases [Cg, 2, n] (phase_low, phase_high per channel), *[w_base_g], *[w_basis_g].

    d phase[c][m][j] = sum_{b,pixel} G_{c,j} * d basis_j / d phase_m, where G = dgrad(dz, W_basis) is never materialised:
    the sum is regrouped as  sum_{group,o,tap} W_basis[o][c*n+j][tap] * wgrad(d basis / d phase_m, dz)[o][c*n+j][tap],
    i.e. the weight-gradient kernel run on the parameter-derivative planes, contracted with the weights."""

    @staticmethod
    def forward(ctx, spec: ConvSpec, x, xn, phases, *weights):
        x = _require(x, "x")
        xn = _require(xn, "xn") if xn is not None else None      # host-applied base activation: x = act(input) feeds the base branch, xn = input the basis
        phases = _require(phases, "phases")
        weights = [_require(w, "weight") for w in weights]
        w_base, w_basis = _split_weights(spec, weights)
        G = spec.groups
        want = (x.shape[1] // G, 2, spec.n_basis) if spec.kind == L.BASIS_RELU else (spec.n_basis,)
        if tuple(phases.shape) != want:
            raise L.KanConvError(f"parameter table {tuple(phases.shape)} != {want} for basis kind {spec.kind}")
        need_dgrad = bool(ctx.needs_input_grad[1]) or (xn is not None and bool(ctx.needs_input_grad[2]))
        with torch.cuda.device(x.device):
            z, packed, geom, _, plan = _conv_forward(spec, x, xn, w_base, w_basis, need_dgrad, phases)
            z = _sum_slabs(z, geom.B, z.shape[2], geom.Ho * geom.Wo)
        ctx.spec = spec
        ctx.two = xn is not None
        ctx.layout = (packed[0] is not None, packed[1] is not None)
        ctx.save_for_backward(x, phases, *w_basis, *[t for t in packed if t is not None], *([xn] if xn is not None else []))
        return z

    @staticmethod
    def backward(ctx, dz):
        spec, G = ctx.spec, ctx.spec.groups
        saved = list(ctx.saved_tensors)
        xn = saved.pop() if ctx.two else None
        x, phases = saved[0], saved[1]
        w_basis = saved[2:2 + G]
        packed = _unflatten(ctx.layout, list(saved[2 + G:]))
        need_x, need_xn = ctx.needs_input_grad[1], (ctx.two and ctx.needs_input_grad[2])
        need_p, need_w = ctx.needs_input_grad[3], any(ctx.needs_input_grad[4:])
        dz = dz.contiguous()
        dph = dxn = None
        with torch.cuda.device(x.device):
            # ReLU-KAN / Gram: when the input gradient is computed anyway, its launch also accumulates the parameter gradients from the same G tiles
            # (kan_conv_bwd_data_params); otherwise (first layer of a model, depthwise groups) two more weight-gradient passes deliver them
            in_epilogue = need_p and (need_x or need_xn) and not _is_depthwise(spec, x, w_basis)
            dpar = None
            if in_epilogue:                                    # ReLU-KAN: [Cg, 2, n] directly; Gram: 64 slot rows of partial sums
                dpar = torch.zeros_like(phases) if spec.kind == L.BASIS_RELU else phases.new_zeros((64, spec.n_basis))
            dx, dxn, dwb, dws = _conv_backward(spec, x, xn, packed, dz, need_x, need_xn, need_w, phases, dparams=dpar)
            if in_epilogue:
                dph = dpar if spec.kind == L.BASIS_RELU else dpar.sum(dim=0)
            if need_p and not in_epilogue:
                Cg, n = x.shape[1] // G, spec.n_basis
                W = torch.stack(list(w_basis)).view(G, -1, Cg, n, spec.kernel[0] * spec.kernel[1])

                def factor(mode):                               # wgrad of the parameter-derivative planes, times the weights
                    _, _, _, dwm = _conv_backward(spec, x, xn, packed, dz, False, False, True, phases, mode)
                    return W * torch.stack(list(dwm)).view_as(W)
                if spec.kind == L.BASIS_RELU:                  # per-channel phases [Cg, 2, n]
                    dph = torch.stack([factor(mode).sum(dim=(0, 1, 4)) for mode in (1, 2)], dim=1)
                else:                                          # Gram: layer-global coefficients c_2..c_degree (entries 0, 1 unused)
                    dph = torch.zeros_like(phases)
                    for mode in range(1, n - 1):
                        dph[mode + 1] = factor(mode).sum()
        return (None, dx if need_x else None, dxn if need_xn else None, dph) + _flat_grads(spec, dwb, dws)


def _cat(ts: Sequence[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
    """Per-group vectors -> one contiguous vector (the tensor itself for one group)."""
    if ts[0] is None:
        return None
    return ts[0] if len(ts) == 1 else torch.cat([t.reshape(-1) for t in ts])


class _KanConvInPrelu(torch.autograd.Function):
    """y = [MaxPool2d(2, 2)]([PReLU](InstanceNorm(conv stage))).
    args: spec, eps, use_affine, use_prelu, pool, x, *[w_base_g], *[w_basis_g], *[gamma_g], *[beta_g], *[prelu_g]"""

    @staticmethod
    def forward(ctx, spec: ConvSpec, eps: float, use_affine: bool, use_prelu: bool, pool: bool, x, *params):
        lib = L.load()
        x = _require(x, "x")
        params = [_require(p, "parameter") for p in params]
        G = spec.groups
        nw = G * (2 if spec.has_base else 1)
        w_base, w_basis = _split_weights(spec, params[:nw])
        rest = params[nw:]
        gammas = rest[:G] if use_affine else [None] * G
        betas = rest[G:2 * G] if use_affine else [None] * G
        prelus = rest[2 * G * int(use_affine):] if use_prelu else [None] * G
        if use_prelu and any(p.numel() != 1 for p in prelus):
            raise L.KanConvError("only scalar-slope PReLU (nn.PReLU()) is supported, as in kan_layers.py:182")
        need_dgrad = bool(ctx.needs_input_grad[5])
        with torch.cuda.device(x.device):
            zs = _split_stage(spec, x, w_base, w_basis) if _SPLIT_NOW else None      # (set by kan_conv_in_prelu: grad mode is always off in here)
            if zs is not None:                           # opt-in inference mode (split_precision_inference): one slab, nothing kept for a backward
                packed = (None, None)
                _, _, plan = _plan_cached(spec, x.shape[0], x.shape[1], x.shape[2], x.shape[3], zs.shape[2], x.shape[1], zs.shape[2])
            else:
                zs, packed, geom, _, plan = _conv_forward(spec, x, None, w_base, w_basis, need_dgrad)
            S, B, Ot, Ho, Wo = zs.shape
            Og, HW = Ot // G, Ho * Wo
            mean = torch.empty(B * Ot, device=x.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            # all groups in one launch: per-channel gamma/beta concatenated, one PReLU slope per Og channels
            gamma, beta, slope = _cat(gammas), _cat(betas), _cat(prelus)
            # summed pre-norm values (saved for the backward): slab 0 itself when there is one slab, else a tensor of their own, so
            # that the S-slab buffer can be freed (was: summed in place into slab 0, then cloned out of the buffer)
            z = zs[0] if S == 1 else torch.empty((B, Ot, Ho, Wo), device=x.device, dtype=torch.float32)
            pidx = None
            if pool and pool is not True and tuple(pool) == (2, 2) and Ho % 2 == 0 and Wo % 2 == 0:
                pool = True                              # even planes: the register-resident 2x2 kernels
            if pool and pool is not True:                # MaxPool2d(k, s), overlapping windows (the AlexNet pattern 3, 2): generic norm kernels
                pk, ps = pool
                if Ho < pk or Wo < pk:
                    raise L.KanConvError(f"fused {pk}x{pk} max-pool on a {Ho}x{Wo} plane")
                Hp, Wp = (Ho - pk) // ps + 1, (Wo - pk) // ps + 1
                y = torch.empty((B, Ot, Hp, Wp), device=x.device, dtype=torch.float32)
                pidx = torch.empty((B, Ot, Hp, Wp), device=x.device, dtype=torch.uint8)
                L.check(lib.kan_instnorm_prelu_poolk_fwd(_ptr(zs), S, plan.fwd_slab_elems, _ptr(z), _ptr(gamma), _ptr(beta), _ptr(slope),
                                                         _ptr(y), C.c_void_p(pidx.data_ptr()), _ptr(mean), _ptr(rstd), B, Ot, Ho, Wo,
                                                         Ot * HW, eps, Og if G > 1 else 0, pk, ps, _stream(x)), "kan_instnorm_prelu_poolk_fwd")
            elif pool:                                   # MaxPool2d(2, 2) fused behind the PReLU: the full-size y is never written
                if Ho % 2 or Wo % 2:
                    raise L.KanConvError(f"fused 2x2 max-pool needs an even plane, got {Ho}x{Wo}")
                y = torch.empty((B, Ot, Ho // 2, Wo // 2), device=x.device, dtype=torch.float32)
                pidx = torch.empty((B, Ot, Ho // 2, Wo // 2), device=x.device, dtype=torch.uint8)
                L.check(lib.kan_instnorm_prelu_pool_fwd(_ptr(zs), S, plan.fwd_slab_elems, _ptr(z), _ptr(gamma), _ptr(beta), _ptr(slope),
                                                        _ptr(y), C.c_void_p(pidx.data_ptr()), _ptr(mean), _ptr(rstd), B, Ot, Ho, Wo,
                                                        Ot * HW, eps, Og if G > 1 else 0, _stream(x)), "kan_instnorm_prelu_pool_fwd")
            else:
                y = torch.empty((B, Ot, Ho, Wo), device=x.device, dtype=torch.float32)
                L.check(lib.kan_instnorm_prelu_fwd(_ptr(zs), S, plan.fwd_slab_elems, _ptr(z), _ptr(gamma), _ptr(beta), _ptr(slope), _ptr(y),
                                                   _ptr(mean), _ptr(rstd), B, Ot, HW, Ot * HW, eps, Og if G > 1 else 0, _stream(x)),
                        "kan_instnorm_prelu_fwd")
        ctx.spec, ctx.flags = spec, (use_affine, use_prelu, pool)
        ctx.wids = (id(w_base[0]) if spec.has_base else None, id(w_basis[0])) if G == 1 else None
        ctx.layout = (packed[0] is not None, packed[1] is not None)
        ctx.save_for_backward(x, z, mean, rstd, *[t for t in packed if t is not None], *[t for t in (gamma, beta, slope) if t is not None],
                              *([pidx] if pool else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        spec = ctx.spec
        use_affine, use_prelu, pool = ctx.flags
        G = spec.groups
        saved = ctx.saved_tensors
        pidx = saved[-1] if pool else None
        if pool:
            saved = saved[:-1]
        x, z, mean, rstd = saved[:4]
        nwd = int(ctx.layout[0]) + int(ctx.layout[1])
        packed = _unflatten(ctx.layout, list(saved[4:4 + nwd]))
        rest = list(saved[4 + nwd:])
        gamma, beta = (rest[0], rest[1]) if use_affine else (None, None)
        slope = rest[2 * int(use_affine)] if use_prelu else None
        dy = dy.contiguous()
        B, Ot, Ho, Wo = z.shape
        Og, HW = Ot // G, Ho * Wo
        with torch.cuda.device(x.device):
            dz = torch.empty_like(z)
            dgam = torch.zeros_like(gamma) if use_affine else None
            dbet = torch.zeros_like(beta) if use_affine else None
            dpre = torch.zeros_like(slope) if use_prelu else None
            if pool and pool is not True:
                L.check(lib.kan_instnorm_prelu_poolk_bwd(_ptr(dy), C.c_void_p(pidx.data_ptr()), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                                         _ptr(beta), _ptr(slope), _ptr(dz), _ptr(dgam), _ptr(dbet), _ptr(dpre), B, Ot, Ho, Wo,
                                                         Ot * HW, Og if G > 1 else 0, pool[0], pool[1], _stream(x)), "kan_instnorm_prelu_poolk_bwd")
            elif pool:
                L.check(lib.kan_instnorm_prelu_pool_bwd(_ptr(dy), C.c_void_p(pidx.data_ptr()), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                                        _ptr(beta), _ptr(slope), _ptr(dz), _ptr(dgam), _ptr(dbet), _ptr(dpre), B, Ot, Ho, Wo,
                                                        Ot * HW, Og if G > 1 else 0, _stream(x)), "kan_instnorm_prelu_pool_bwd")
            else:
                L.check(lib.kan_instnorm_prelu_bwd(_ptr(dy), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(slope), _ptr(dz),
                                                   _ptr(dgam), _ptr(dbet), _ptr(dpre), B, Ot, HW, Ot * HW, Og if G > 1 else 0, _stream(x)),
                        "kan_instnorm_prelu_bwd")
            nw = G * (2 if spec.has_base else 1)
            need_x = ctx.needs_input_grad[5]
            need_w = any(ctx.needs_input_grad[6:6 + nw])
            dx, _, dwb, dws = _conv_backward(spec, x, None, packed, dz, need_x, False, need_w, wids=ctx.wids)
        grads = _flat_grads(spec, dwb, dws)
        if use_affine:
            grads += tuple(dgam.view(G, Og).unbind(0)) + tuple(dbet.view(G, Og).unbind(0))
        if use_prelu:
            grads += tuple(dpre.view(G, 1).unbind(0))
        return (None, None, None, None, None, dx if need_x else None) + grads


class _InstanceNorm(torch.autograd.Function):
    """InstanceNorm2d over NCHW (args: x, gamma|None, beta|None, eps)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        lib = L.load()
        x = _require(x, "x")
        B, Cn, H, W = x.shape
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            mean = torch.empty(B * Cn, device=x.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            L.check(lib.kan_instnorm_prelu_fwd(_ptr(x), 1, 0, _ptr(x), _ptr(gamma), _ptr(beta), C.c_void_p(0), _ptr(y), _ptr(mean), _ptr(rstd),
                                               B, Cn, H * W, Cn * H * W, eps, 0, _stream(x)), "kan_instnorm_prelu_fwd")
        ctx.affine = gamma is not None
        ctx.save_for_backward(x, mean, rstd, *([gamma, beta] if gamma is not None else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        saved = ctx.saved_tensors
        x, mean, rstd = saved[:3]
        gamma, beta = (saved[3], saved[4]) if ctx.affine else (None, None)
        dy = dy.contiguous()
        B, Cn, H, W = x.shape
        with torch.cuda.device(x.device):
            dx = torch.empty_like(x)
            dg = torch.zeros_like(gamma) if ctx.affine else None
            db = torch.zeros_like(beta) if ctx.affine else None
            L.check(lib.kan_instnorm_prelu_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), C.c_void_p(0), _ptr(dx),
                                               _ptr(dg), _ptr(db), C.c_void_p(0), B, Cn, H * W, Cn * H * W, 0, _stream(x)), "kan_instnorm_prelu_bwd")
        return dx, dg, db, None


# --------------------------------------------------------------------------------------- Wav-KAN wavelet stage
class _WavStage(torch.autograd.Function):
    """u = sum_{c, taps} w[o, c, tap] * psi((x - trans[o, c]) / scale[o, c]) for one group (wav_kan_layers.py:186-217 and the two
    'fast' variants), on the direct kernels of csrc/wavkan.inc.  x: [B, C, H, W] contiguous; scale, trans: [O, C]; w: [O, C, kh, kw]."""

    @staticmethod
    def _geom(wavelet, x, w, stride, padding, dilation):
        B, Cn, H, W = x.shape
        O, _, kh, kw = w.shape
        Ho = (H + 2 * padding[0] - dilation[0] * (kh - 1) - 1) // stride[0] + 1
        Wo = (W + 2 * padding[1] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
        if Ho <= 0 or Wo <= 0:
            raise L.KanConvError(f"empty output ({Ho}x{Wo}) for input {H}x{W}")
        return L.KanWavGeom(B, Cn, H, W, O, Ho, Wo, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                            int(wavelet), Cn * H * W, O * Ho * Wo)

    @staticmethod
    def forward(ctx, wavelet, stride, padding, dilation, x, scale, trans, w):
        lib = L.load()
        x, scale, trans, w = (_require(t, n).contiguous() for t, n in ((x, "x"), (scale, "scale"), (trans, "translation"), (w, "wavelet weights")))
        if scale.shape != (w.shape[0], w.shape[1]) or trans.shape != scale.shape or w.shape[1] != x.shape[1]:
            raise L.KanConvError(f"wavelet parameter shapes {tuple(scale.shape)}, {tuple(trans.shape)}, {tuple(w.shape)} do not match input {tuple(x.shape)}")
        geom = _WavStage._geom(wavelet, x, w, stride, padding, dilation)
        u = torch.empty((geom.B, geom.O, geom.Ho, geom.Wo), device=x.device, dtype=torch.float32)
        L.check(lib.kan_wav_fwd(_ptr(x), _ptr(scale), _ptr(trans), _ptr(w), _ptr(u), C.byref(geom), _stream(x)), "kan_wav_fwd")
        ctx.save_for_backward(x, scale, trans, w)
        ctx.geom = geom
        return u

    @staticmethod
    def backward(ctx, du):
        lib = L.load()
        x, scale, trans, w = ctx.saved_tensors
        geom = ctx.geom
        du = du.contiguous()
        st = _stream(x)
        dx = dscale = dtrans = dw = None
        if ctx.needs_input_grad[4]:
            dx = torch.empty_like(x)
            L.check(lib.kan_wav_bwd_input(_ptr(du), _ptr(x), _ptr(scale), _ptr(trans), _ptr(w), _ptr(dx), C.byref(geom), st), "kan_wav_bwd_input")
        if any(ctx.needs_input_grad[5:8]):
            ws = torch.empty((lib.kan_wav_param_workspace(C.byref(geom)),), device=x.device, dtype=torch.float32)
            dw, dscale, dtrans = torch.empty_like(w), torch.empty_like(scale), torch.empty_like(trans)
            L.check(lib.kan_wav_bwd_params(_ptr(du), _ptr(x), _ptr(scale), _ptr(trans), _ptr(w), _ptr(dw), _ptr(dscale), _ptr(dtrans), _ptr(ws),
                                           C.byref(geom), st), "kan_wav_bwd_params")
        return None, None, None, None, dx, dscale, dtrans, dw


def wav_stage(wavelet: int, x: torch.Tensor, scale: torch.Tensor, trans: torch.Tensor, w: torch.Tensor, stride, padding, dilation) -> torch.Tensor:
    return _WavStage.apply(int(wavelet), tuple(stride), tuple(padding), tuple(dilation), x, scale, trans, w)


# --------------------------------------------------------------------------------------- public API
# The kernels address activations with 32-bit byte offsets (buffer loads), so one launch takes tensors below 2 GiB.  Every op
# on this path is independent per image (conv, per-(image, channel) InstanceNorm, PReLU, pooling), so a larger batch is cut
# into the fewest equal-ish runs of whole images that fit and the results are concatenated; autograd sums the weight gradients.
MAX_TENSOR_BYTES = (1 << 31) - 1


def _image_runs(spec: ConvSpec, x: torch.Tensor, out_channels: int) -> Optional[int]:
    """Images per launch if the batch has to be cut (None: it fits)."""
    B, C, H, W = x.shape
    Ho, Wo = spec.out_hw(H, W)
    per_image = 4 * max(C * H * W, out_channels * Ho * Wo)
    if B * per_image <= MAX_TENSOR_BYTES:
        return None
    if per_image > MAX_TENSOR_BYTES:
        raise L.KanConvError(f"one image's activations ({per_image} bytes) exceed the {MAX_TENSOR_BYTES}-byte launch limit")
    fit = MAX_TENSOR_BYTES // per_image               # images one launch can take; equal-ish runs of at most that many
    runs = -(-B // fit)
    return -(-B // runs)


def _by_image_runs(fn, n: int, x: torch.Tensor, xn: Optional[torch.Tensor] = None) -> torch.Tensor:
    xs = x.split(n)
    xns = xn.split(n) if xn is not None else [None] * len(xs)
    return torch.cat([fn(a.contiguous(), b.contiguous() if b is not None else None) for a, b in zip(xs, xns)])


def kan_conv(spec: ConvSpec, x: torch.Tensor, xn: Optional[torch.Tensor], w_base: Sequence[torch.Tensor],
             w_basis: Sequence[torch.Tensor]) -> torch.Tensor:
    ws = (list(w_base) if spec.has_base else []) + list(w_basis)
    n = _image_runs(spec, x, sum(w.shape[0] for w in w_basis))
    if n is not None:
        return _by_image_runs(lambda a, b: _KanConv.apply(spec, a, b, *ws), n, x, xn)
    return _KanConv.apply(spec, x, xn, *ws)


def kan_conv_phased(spec: ConvSpec, x: torch.Tensor, phases: torch.Tensor, w_base: Sequence[torch.Tensor],
                    w_basis: Sequence[torch.Tensor], xn: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Conv stage of a basis with trainable parameters held in device memory, differentiable in them: ReLU-KAN
    (`phases` = [channels per group, 2, n_basis]: low, high) or Gram (`phases` = [n_basis] recurrence coefficients c_k)."""
    ws = (list(w_base) if spec.has_base else []) + list(w_basis)
    n = _image_runs(spec, x, sum(w.shape[0] for w in w_basis))
    if n is not None:
        return _by_image_runs(lambda a, b: _KanConvPhased.apply(spec, a, b, phases, *ws), n, x, xn)
    return _KanConvPhased.apply(spec, x, xn, phases, *ws)


def _norm_pool(pool):
    """False | True (= the even-plane 2x2 kernels) | (k, s)."""
    if pool is True or not pool:
        return bool(pool)
    k, st = int(pool[0]), int(pool[1])
    if not (2 <= k <= 15 and 1 <= st <= k):
        raise L.KanConvError(f"fused max-pool takes 2 <= kernel <= 15 and 1 <= stride <= kernel, got {pool!r}")
    return (k, st)


def kan_conv_in_prelu(spec: ConvSpec, x: torch.Tensor, w_base: Sequence[torch.Tensor], w_basis: Sequence[torch.Tensor],
                      gammas: Optional[Sequence[torch.Tensor]], betas: Optional[Sequence[torch.Tensor]],
                      prelus: Optional[Sequence[torch.Tensor]], eps: float = 1e-5, pool=False) -> torch.Tensor:
    """`pool=True` additionally applies MaxPool2d(kernel 2, stride 2) inside the same kernels (even output planes only); `pool=(k, s)` a general
    MaxPool2d(k, s) without padding (overlapping windows allowed: the AlexNet pattern (3, 2))."""
    pool = _norm_pool(pool)
    global _SPLIT_NOW
    _SPLIT_NOW = _SPLIT_INFERENCE and not torch.is_grad_enabled()        # opt-in inference mode: decided where the caller's grad mode is visible
    ws = (list(w_base) if spec.has_base else []) + list(w_basis)
    aff = gammas is not None
    extra = (list(gammas) + list(betas) if aff else []) + (list(prelus) if prelus is not None else [])
    n = _image_runs(spec, x, sum(w.shape[0] for w in w_basis))
    if n is not None:
        return _by_image_runs(lambda a, _: _KanConvInPrelu.apply(spec, float(eps), aff, prelus is not None, pool, a, *ws, *extra), n, x)
    return _KanConvInPrelu.apply(spec, float(eps), aff, prelus is not None, pool, x, *ws, *extra)


def kan_conv_fwd_split(spec: ConvSpec, x: torch.Tensor, w_base: torch.Tensor, w_basis: torch.Tensor, wc: Optional[torch.Tensor] = None):
    """OPT-IN split-precision conv stage (forward only, no autograd; kanconv.h `kan_conv_fwd_split`): the fp32 result of `kan_conv` to ~4e-6 of its largest
    element, through 3 x bf16 pieces and six bf16 MFMA products per k-block.  Never used by the layers; scope: default B-spline spec on 8x8 or 16x16 planes, 3x3 /
    stride 1 / pad 1, one group, C % 8 == 0, O % 128 == 0, even batch on 8x8 (raises KanConvError otherwise).  Returns (z, wc): pass `wc` back while the
    weights are unchanged to skip the cut (0.05 ms at 256 -> 256)."""
    lib = L.load()
    x, w_base, w_basis = _require(x, "x"), _require(w_base, "w_base"), _require(w_basis, "w_basis")
    B, Ct, H, W = x.shape
    Ot = w_basis.shape[0]
    geom, basis, _ = _plan_cached(spec, B, Ct, H, W, Ot, Ct, Ot)
    if not lib.kan_split_supported(C.byref(geom), C.byref(basis)):
        raise L.KanConvError("split-precision forward: default B-spline spec (grid 5, order 3, SiLU base branch), 8x8 or 16x16 planes, 3x3 / stride 1 / pad 1, one group, C % 8 == 0, O % 128 == 0, "
                             "even batch only")
    with torch.cuda.device(x.device):
        if wc is None:
            wc = torch.empty(lib.kan_split_weight_bytes(C.byref(geom), C.byref(basis)), device=x.device, dtype=torch.uint8)
            L.check(lib.kan_split_pack_weights(_ptr(w_base), _ptr(w_basis), C.c_void_p(wc.data_ptr()), C.byref(geom), C.byref(basis), _stream(x)),
                    "kan_split_pack_weights")
        z = torch.empty((B, Ot, H, W), device=x.device, dtype=torch.float32)
        L.check(lib.kan_conv_fwd_split(_ptr(x), C.c_void_p(wc.data_ptr()), _ptr(z), C.byref(geom), C.byref(basis), _stream(x)), "kan_conv_fwd_split")
    return z, wc


def instance_norm(x: torch.Tensor, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None, eps: float = 1e-5):
    return _InstanceNorm.apply(x, gamma, beta, float(eps))
