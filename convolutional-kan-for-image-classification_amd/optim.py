"""Fused AdamW step over flat parameter blocks -- SURVEY.md section 8(f) rank 4 (the step right after the hot path).

The reference trains with ``optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)`` + ``ExponentialLR``
(generic_train.py:24-25), stepped once per batch (evaluations.py: train()).  torch walks the parameter list; here every
parameter group lives in ONE flat fp32 block (parameters, exp_avg, exp_avg_sq; the parameters are views into it) and a
step is one launch of ``kan_adamw_step_segments`` per group, which reads every gradient where autograd -- or the DP
reducer's all-reduce bucket -- left it (a device table of gradient addresses, re-uploaded only when an address changes).
Nothing is copied, zeroed or accumulated per step: 28 bytes of HBM traffic per element, 2.3 GB = ~0.3 ms at the HBM
roofline for KAN-VGG11's 83 M parameters.

It is a ``torch.optim.Optimizer``: ``param_groups`` / ``lr`` schedulers (ExponentialLR), ``state_dict`` /
``load_state_dict`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` entries, as torch's AdamW stores them) and
``zero_grad`` (torch's own: gradients set to None) work as usual.  The update itself has no CPU path: ``step()`` on CPU parameters raises.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib as L

_ALIGN = 64          # elements: every parameter starts on a 256-byte boundary of the flat block
_CHUNK = 8192        # elements per workgroup of the segment kernel (256 threads x 8 float4)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = []
        for group in self.param_groups:
            self._flat.append(self._flatten(group))

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_flat"):                   # groups added after construction get their own flat block
            self._flat.append(self._flatten(self.param_groups[-1]))

    # ------------------------------------------------------------------ layout
    @staticmethod
    def _flatten(group):
        ps: List[torch.nn.Parameter] = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev:
                raise L.KanConvError("FusedAdamW needs float32 parameters on one device per group")
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        blk = torch.zeros((3, n), dtype=torch.float32, device=dev)          # rows: parameters, exp_avg, exp_avg_sq
        views = []
        with torch.no_grad():
            for p, o in zip(ps, offs):
                pv, mv, vv = (blk[r, o:o + p.numel()].view_as(p) for r in range(3))
                pv.copy_(p.data)
                p.data = pv
                views.append((pv, mv, vv))
        # chunk table of kan_adamw_step_segments: workgroup b -> (segment, first element); built once per layout
        seg, start = [], []
        for i, p in enumerate(ps):
            for s0 in range(0, p.numel(), _CHUNK):
                seg.append(i); start.append(s0)
        tab = dict(seg_off=torch.tensor(offs, dtype=torch.int64, device=dev), seg_n=torch.tensor([p.numel() for p in ps], dtype=torch.int32, device=dev),
                   chunk_seg=torch.tensor(seg, dtype=torch.int32, device=dev), chunk_start=torch.tensor(start, dtype=torch.int32, device=dev),
                   seg_grad=torch.zeros(len(ps), dtype=torch.int64, device=dev), grad_ptrs=None,
                   seg_bias=torch.zeros((len(ps), 2), dtype=torch.float32, device=dev))
        group.setdefault("step", 0)
        return dict(params=ps, block=blk, views=views, n=n, tab=tab, steps=[0] * len(ps), index={id(p): i for i, p in enumerate(ps)})

    def _link_state(self, group, flat):
        for p, (_, mv, vv) in zip(flat["params"], flat["views"]):
            st = self.state[p]
            st["step"] = torch.tensor(float(flat["steps"][flat["index"][id(p)]]))
            st["exp_avg"], st["exp_avg_sq"] = mv, vv

    # ------------------------------------------------------------------ torch.optim surface
    @torch.no_grad()
    def step(self, closure=None):
        """One launch per parameter group; gradients are read where autograd (or the DP reducer's bucket) left them."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for group, flat in zip(self.param_groups, self._flat):
            if flat is None:
                continue
            blk, tab = flat["block"], flat["tab"]
            if not blk.is_cuda:
                raise L.KanConvError(f"FusedAdamW.step: parameters are on {blk.device}; the fused step runs only on a ROCm device "
                                     "(there is no CPU fallback)")
            ptrs, keep = [], []
            for p, (pv, _, _) in zip(flat["params"], flat["views"]):
                if p.data_ptr() != pv.data_ptr():
                    raise L.KanConvError("a parameter was re-allocated after FusedAdamW flattened it (call .to()/.cuda() before "
                                         "building the optimizer)")
                g = p.grad
                if g is None:
                    ptrs.append(0)                                         # torch skips parameters without a gradient
                    continue
                if g.dtype != torch.float32 or g.device != blk.device or not g.is_contiguous():
                    g = g.to(device=blk.device, dtype=torch.float32).contiguous()
                    keep.append(g)
                ptrs.append(g.data_ptr())
            if not any(ptrs):
                continue
            if ptrs != tab["grad_ptrs"]:                                   # the caching allocator hands back the same blocks step
                tab["seg_grad"].copy_(torch.tensor(ptrs, dtype=torch.int64), non_blocking=False)      # after step: usually a no-op
                tab["grad_ptrs"] = ptrs
            b1, b2 = group["betas"]
            steps = flat["steps"]                                          # torch counts steps per parameter
            for i, a in enumerate(ptrs):
                steps[i] += 1 if a else 0
            group["step"] = max(steps)
            bias = None
            if any(a and st != group["step"] for a, st in zip(ptrs, steps)):      # rare: some parameter skipped earlier steps
                bias = tab["seg_bias"]
                bias.copy_(torch.tensor([[-group["lr"] / (1.0 - b1 ** max(st, 1)), 1.0 / (1.0 - b2 ** max(st, 1)) ** 0.5] for st in steps],
                                        dtype=torch.float32))
            row = lambda r: C.c_void_p(blk.data_ptr() + 4 * r * flat["n"])
            with torch.cuda.device(blk.device):
                L.check(lib.kan_adamw_step_segments(row(0), row(1), row(2), C.c_void_p(tab["seg_grad"].data_ptr()),
                                                    C.c_void_p(tab["seg_off"].data_ptr()), C.c_void_p(tab["seg_n"].data_ptr()),
                                                    C.c_void_p(tab["chunk_seg"].data_ptr()), C.c_void_p(tab["chunk_start"].data_ptr()),
                                                    C.c_void_p(bias.data_ptr() if bias is not None else 0), tab["chunk_seg"].numel(), _CHUNK, group["lr"], b1, b2, group["eps"],
                                                    group["weight_decay"], group["step"], 1.0,
                                                    C.c_void_p(torch.cuda.current_stream(blk.device).cuda_stream)), "kan_adamw_step_segments")
            del keep
            for p, a in zip(flat["params"], ptrs):                         # the kernel wrote through raw pointers: tell autograd (and
                if a:                                                      # the packed-weight cache of ops.py) that the values changed
                    torch.autograd.graph.increment_version(p)
        return loss

    def state_dict(self):
        for group, flat in zip(self.param_groups, self._flat):
            if flat is not None:
                self._link_state(group, flat)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group, flat in zip(self.param_groups, self._flat):
            if flat is None:
                continue
            steps_seen = 0                          # one step count per group (torch keeps one per parameter; they agree)
            with torch.no_grad():
                for p, (_, mv, vv) in zip(flat["params"], flat["views"]):
                    st = self.state.get(p, {})
                    if "exp_avg" in st:
                        mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                        flat["steps"][flat["index"][id(p)]] = int(st.get("step", 0))
                        group["step"] = max(int(st.get("step", 0)), steps_seen)
                        steps_seen = group["step"]
            self._link_state(group, flat)
