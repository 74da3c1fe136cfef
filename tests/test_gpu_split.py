"""OPT-IN split-precision forward (kan_conv_fwd_split) -- its OWN tolerance, apart from the exact fp32 path's:
every operand is cut into three bf16 pieces and six bf16 MFMA products per k-block are accumulated in ONE fp32 chain over the whole depth, so the
error against fp64 is that of a k-ordered fp32 sum (measured 3.6e-6 of the largest output at K = 20 736; the exact kernels, which sum four
shorter chains, sit at 1.1e-6).  Asserted: <= 1e-5 max-normalised against the fp64 oracle -- the forward tolerance SURVEY 8(c) states for the exact
path -- and <= 1e-5 against the exact HIP kernel on the same inputs, on KAN-VGG11's 16x16 and 8x8 shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C_,O_,B,PW", [(256, 256, 4, 8), (128, 256, 6, 8), (8, 128, 2, 8), (64, 128, 3, 16), (16, 256, 2, 16)])
def test_split_forward_vs_fp64_oracle_and_exact_kernel(C_, O_, B, PW, gpu_lib):
    import convkan_amd as K
    from convkan_amd import ops
    from oracle import kan_oracle as O
    torch.manual_seed(C_ + B)
    layer = K.KANConv2DLayer(C_, O_, 3, padding=1, base_activation=torch.nn.SiLU).cuda()      # (as models/kan_vgg.py builds it)
    x = torch.randn(B, C_, PW, PW, device="cuda")
    spec = layer.conv_spec()
    wb, ws = layer.base_conv[0].weight.detach(), layer.spline_conv[0].weight.detach()
    z_split, wc = ops.kan_conv_fwd_split(spec, x, wb, ws)
    z_again, _ = ops.kan_conv_fwd_split(spec, x, wb, ws, wc)          # cut weights reused
    z_exact = ops.kan_conv(spec, x, None, [wb], [ws])
    torch.cuda.synchronize()
    assert torch.equal(z_split, z_again)
    pre = []
    O.kan_conv2d(x.double().cpu(), [wb.double().cpu()], [ws.double().cpu()], [torch.tensor([0.25], dtype=torch.float64)],
                 knots=layer.grid.double().cpu(), spline_order=3, act=torch.nn.functional.silu, padding=1, pre_norm_out=pre)
    ref = pre[0]
    scale = float(ref.abs().max())
    e_split = float((z_split.double().cpu() - ref).abs().max()) / scale
    e_exact = float((z_exact.double().cpu() - ref).abs().max()) / scale
    e_pair = float((z_split - z_exact).abs().max()) / scale
    print(f"[split {C_}->{O_} @{PW}x{PW} B={B}] vs fp64: split {e_split:.2e}  exact kernel {e_exact:.2e};  split vs exact {e_pair:.2e}")
    assert e_split <= 1e-5 and e_pair <= 1e-5, (e_split, e_exact, e_pair)


def test_split_forward_scope_is_enforced(gpu_lib):
    import convkan_amd as K
    from convkan_amd import _lib as L
    from convkan_amd import ops
    for layer, x in ((K.KANConv2DLayer(16, 128, 3, padding=1, base_activation=torch.nn.SiLU), torch.randn(2, 16, 4, 4)),                 # 4x4 planes
                     (K.KANConv2DLayer(16, 128, 3, padding=1, grid_size=8, base_activation=torch.nn.SiLU), torch.randn(2, 16, 8, 8)),     # not the default spec
                     (K.KANConv2DLayer(16, 64, 3, padding=1, base_activation=torch.nn.SiLU), torch.randn(2, 16, 8, 8)),                    # 64 outputs
                     (K.KANConv2DLayer(16, 128, 3, padding=1, base_activation=torch.nn.SiLU), torch.randn(3, 16, 8, 8)),                  # odd batch
                     (K.KANConv2DLayer(16, 128, 3, padding=1), torch.randn(2, 16, 8, 8))):                                              # GELU base branch (the layer's own default)
        layer = layer.cuda()
        with pytest.raises(L.KanConvError, match="split-precision"):
            ops.kan_conv_fwd_split(layer.conv_spec(), x.cuda(), layer.base_conv[0].weight.detach(), layer.spline_conv[0].weight.detach())


def test_split_precision_inference_mode_on_kan_vgg11(gpu_lib):
    """ops.split_precision_inference(): under no_grad the 16x16 layer and the two 8x8 layers of KAN-VGG11 (64 -> 128, 128 -> 256, 256 -> 256) take the split-precision conv stage, every other
    layer and every call that needs gradients stays exact.  Logits within 1e-4 of the exact model's (max-normalised); outside the context, and in a
    training step inside it, the results are the exact path's bit for bit."""
    from convkan_amd import ops
    from convkan_amd.models import vggkan
    torch.manual_seed(1)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", dropout_linear=0.0).cuda().eval()
    x = torch.randn(64, 3, 32, 32, device="cuda")
    calls = []
    orig = ops.kan_conv_fwd_split
    ops.kan_conv_fwd_split = lambda *a, **k: (calls.append(a[1].shape), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            exact = m(x)
            with ops.split_precision_inference():
                split = m(x)
                again = m(x)                                             # cut weights come from the cache
            after = m(x)
        assert [tuple(c) for c in calls] == [(64, 64, 16, 16), (64, 128, 8, 8), (64, 256, 8, 8)] * 2, calls
        assert torch.equal(split, again) and torch.equal(exact, after)
        err = float((split - exact).abs().max() / exact.abs().max())
        print(f"[split inference] logits vs exact model: {err:.2e}")
        assert 0.0 < err <= 1e-4, err
        n = len(calls)
        with ops.split_precision_inference():                             # with grad mode on (a training step) nothing leaves the exact path
            m.train()
            y1 = m(x)
        y0 = m(x)
        assert len(calls) == n and torch.equal(y0, y1)
    finally:
        ops.kan_conv_fwd_split = orig
