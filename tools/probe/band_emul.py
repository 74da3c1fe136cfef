#!/usr/bin/env python3
"""CPU emulation (numpy) of the band kernels' index arithmetic (csrc/kan_direct.hip): phase decomposition of strided taps, virtual halo
rows with a gap per image, tap shifts, band weight order, even-padded plane rows.  Checks the decomposition itself against F.conv2d on
random geometries -- the HIP kernels mirror these formulas line by line.  python tools/probe/band_emul.py"""
import itertools
import random

import numpy as np
import torch
import torch.nn.functional as F


def floordiv(a, b):
    return a // b          # python floors (b > 0)


def band_tables(g):
    """Host side of kan_direct.hip: per-tap phase / offset, phases in first-use order, tap lists per phase, shifts."""
    kh, kw, sh, sw, ph, pw, dh, dw = (g[k] for k in ("kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw"))
    er = [dh * r - ph for r in range(kh)]
    et = [dw * t - pw for t in range(kw)]
    a_r, off_r = [e % sh for e in er], [floordiv(e, sh) for e in er]
    b_t, off_t = [e % sw for e in et], [floordiv(e, sw) for e in et]
    OR0, OR1, OC0, OC1 = min(off_r), max(off_r), min(off_t), max(off_t)
    span_r, span_c = OR1 - OR0, OC1 - OC0
    HC = g["Wo"] + span_c
    phases = {}
    for r in range(kh):
        for t in range(kw):
            phases.setdefault((a_r[r], b_t[t]), []).append((r, t, (off_r[r] - OR0) * HC + (off_t[t] - OC0)))
    return dict(OR0=OR0, OC0=OC0, span_r=span_r, span_c=span_c, HC=HC, phases=list(phases.items()))


def tile_layout(g, tb, p0, TP):
    """Device side, per tile: pixel -> cell, and cell -> (image, sub-row i, sub-col j) for the expansion."""
    Ho, Wo = g["Ho"], g["Wo"]
    Mtot = g["B"] * Ho * Wo
    p1 = min(p0 + TP, Mtot) - 1
    b0, r0 = divmod(p0, Ho * Wo); ho0 = r0 // Wo
    b1, r1 = divmod(p1, Ho * Wo); ho1 = r1 // Wo
    n0 = (ho1 if b0 == b1 else Ho - 1) - ho0 + 1            # output rows of the first image
    blk0, blk = n0 + tb["span_r"], Ho + tb["span_r"]          # virtual rows of image 0 / of every later image
    nimg = b1 - b0 + 1
    last_rows = (ho1 + 1) if nimg > 1 else 0
    VR = blk0 + (nimg - 2) * blk * (nimg > 2) + ((last_rows + tb["span_r"]) if nimg > 1 else 0)

    def cell_of(px):
        b, r = divmod(px, Ho * Wo); ho, wo = divmod(r, Wo)
        k = b - b0
        v = (ho - ho0) if k == 0 else blk0 + (k - 1) * blk + ho
        return v * tb["HC"] + wo

    def unit(cell):                                          # -> (image, sub-row i, sub-col j)
        v, jj = divmod(cell, tb["HC"])
        if v < blk0:
            k, lv, first = 0, v, ho0
        else:
            k, lv = divmod(v - blk0, blk); k += 1; first = 0
        return b0 + k, first + lv + tb["OR0"], jj + tb["OC0"]
    return cell_of, unit, VR


def emulate(g, planes, w, TP, NG):
    """planes: [B, C, P, H, W] expanded input (any values), w: [O, C, P, kh, kw] -> z [B, O, Ho, Wo] by the band algorithm."""
    B, C, P, H, W = planes.shape
    O = w.shape[0]
    Ho, Wo, sh, sw = g["Ho"], g["Wo"], g["sh"], g["sw"]
    tb = band_tables(g)
    NPL = NG * P; NPLE = NPL + (NPL & 1)
    NGR = -(-C // NG)
    Mtot = B * Ho * Wo
    z = np.zeros((Mtot, O))
    # band weight order: step = (phase, group, tap-in-phase), row = ch * P + p  (pad rows / missing channels zero)
    for p0 in range(0, Mtot, TP):
        cell_of, unit, VR = tile_layout(g, tb, p0, TP)
        cells = VR * tb["HC"]
        npx = min(TP, Mtot - p0)
        pcell = np.array([cell_of(p0 + i) for i in range(npx)])
        for (a, b), taps in tb["phases"]:
            for gidx in range(NGR):
                halo = np.zeros((cells + max(s for _, _, s in taps) + 1, NPLE))          # [cell][plane row]
                for cell in range(cells):
                    img, i, j = unit(cell)
                    row, col = sh * i + a, sw * j + b
                    if img < B and 0 <= row < H and 0 <= col < W:
                        for ch in range(NG):
                            c = gidx * NG + ch
                            if c < C:
                                halo[cell, ch * P:(ch + 1) * P] = planes[img, c, :, row, col]
                for (r, t, shift) in taps:
                    wstep = np.zeros((NPLE, O))
                    for ch in range(NG):
                        c = gidx * NG + ch
                        if c < C:
                            wstep[ch * P:(ch + 1) * P, :] = w[:, c, :, r, t].T
                    assert (pcell + shift).max() < cells, "tap read past the tile's halo"
                    z[p0:p0 + npx] += halo[pcell + shift] @ wstep
    return z.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2)


def check(seed):
    r = random.Random(seed)
    kh, kw = r.choice([1, 2, 3, 5, 7, 11]), r.choice([1, 2, 3, 5, 7, 11])
    sh, sw = r.choice([1, 1, 2, 3, 4]), r.choice([1, 1, 2, 3, 4])
    dh, dw = r.choice([1, 1, 2]), r.choice([1, 1, 3])
    ph, pw = r.choice([0, 1, 2, 5]), r.choice([0, 1, 2, 5])
    H, W = r.randint(1, 30), r.randint(1, 30)
    Ho, Wo = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    if Ho <= 0 or Wo <= 0:
        return None
    B, C, P, O = r.randint(1, 4), r.choice([1, 2, 3, 4, 5]), r.choice([1, 2, 5]), r.choice([1, 3])
    NG = r.choice([1, 2, 3])
    TP = r.choice([8, 32, 128, 256])
    g = dict(B=B, H=H, W=W, Ho=Ho, Wo=Wo, kh=kh, kw=kw, sh=sh, sw=sw, ph=ph, pw=pw, dh=dh, dw=dw)
    rng = np.random.default_rng(seed)
    planes = rng.standard_normal((B, C, P, H, W))
    w = rng.standard_normal((O, C, P, kh, kw))
    ref = F.conv2d(torch.from_numpy(planes.reshape(B, C * P, H, W)), torch.from_numpy(w.reshape(O, C * P, kh, kw)), None, (sh, sw), (ph, pw), (dh, dw)).numpy()
    got = emulate(g, planes, w, TP, NG)
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
    assert err < 1e-12, (seed, g, C, P, NG, TP, err)
    return err


if __name__ == "__main__":
    n = 0
    for seed in range(400):
        if check(seed) is not None:
            n += 1
    print(f"band emulation == conv2d on {n} random geometries")
