"""Forward-only (torch.no_grad, eval) throughput of KAN-VGG11 bs 256, with and without the packed-weight cache of ops.py.
python tools/infer_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convkan_amd as K  # noqa: E402
from convkan_amd import ops  # noqa: E402
from convkan_amd.models import vggkan  # noqa: E402


def main():
    torch.manual_seed(0)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear").cuda().eval()
    x = torch.randn(256, 3, 32, 32, device="cuda")
    out = {}
    for name, entries in (("cached", 128), ("repack_every_call", 0)):
        ops._PACKED.clear(); ops._PACK_CACHE_MAX = entries
        with torch.no_grad():
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                m(x)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        out[name] = dict(ms_per_batch=round(ms, 3), images_per_s=round(256 / ms * 1e3, 1))
    ops._PACKED.clear(); ops._PACK_CACHE_MAX = 128
    with torch.no_grad(), ops.split_precision_inference():                # OPT-IN: the two 8x8 layers' conv stages in split precision
        ref = None
        for _ in range(5):
            y = m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            m(x)
        e1.record(); torch.cuda.synchronize()
    with torch.no_grad():
        ref = m(x)
    ms = e0.elapsed_time(e1) / 30
    out["split_precision_inference_opt_in"] = dict(ms_per_batch=round(ms, 3), images_per_s=round(256 / ms * 1e3, 1),
                                                   logits_max_diff_vs_exact=float((y - ref).abs().max() / ref.abs().max()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
