/*
 * kanconv.h -- C ABI of libkanconv.so: MI355X (gfx950) kernels for the conv-KAN hot path.
 *
 * The reference (GadGadGad/Convolutional-KAN-for-Image-Classification) has NO native
 * interface for this path: it is ~60 eager ATen ops per layer call.  The entry points
 * below are what a ctypes/cffi binding placed inside the reference's layer classes would
 * call instead of those op sequences.  Each declaration cites the
 * reference lines it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *   - the caller owns every buffer, including workspaces; the library never allocates,
 *     frees or synchronises the device; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*);
 *   - return value: 0 on success, <0 on error; message via kan_last_error() (thread-local);
 *   - no C++ exception crosses the boundary; no mutable global state.
 *
 * Tensor layouts: activations NCHW.  KanGeom describes ONE group of the reference layer
 * (kan_layers.py:249-258 loops over groups in Python); `groups` = G > 1 runs G such groups in
 * the same launches: group j reads channels [j*C, (j+1)*C) of x and writes channels
 * [j*O, (j+1)*O) of z, x / z point at group 0's first channel and the batch strides
 * (`x_bstride`, `y_bstride`) are those of the FULL tensors.  Per-group weights are stacked on a
 * leading axis; every packed / workspace layout is G per-group blocks back to back.
 */
#ifndef KANCONV_H
#define KANCONV_H

#ifdef __cplusplus
extern "C" {
#endif

#define KAN_MAX_PLANES 16   /* basis planes per input channel incl. the base-activation plane */
#define KAN_MAX_TABLE  32   /* knots (B-spline) or centres (RBF) */
#define KAN_FP_WORDS   192  /* 64-bit words of the fingerprint ring of kan_pack_weights_cached (3 slots x 64) */

/* basis families */
enum { KAN_BASIS_BSPLINE = 0, KAN_BASIS_RBF = 1, KAN_BASIS_CHEBY = 2, KAN_BASIS_POLY = 3, KAN_BASIS_FOURIER = 4, KAN_BASIS_RELU = 5, KAN_BASIS_GRAM = 6 };
/* base-branch activations; KAN_ACT_NONE = layer has no base branch (ChebyKAN) */
enum { KAN_ACT_NONE = -1, KAN_ACT_IDENTITY = 0, KAN_ACT_GELU = 1, KAN_ACT_SILU = 2, KAN_ACT_RELU = 3,
       KAN_ACT_TANH = 4, KAN_ACT_SIGMOID = 5, KAN_ACT_GELU_TANH = 6 };

/* Geometry of ONE group's convolution (all fields in elements, not bytes). */
typedef struct KanGeom {
    int B, C, H, W;              /* input block: batch, channels of this group, height, width */
    int O, Ho, Wo;               /* output block of this group */
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int groups;                  /* G groups handled by one call (0 is read as 1); C and O stay PER GROUP */
    long long x_bstride;         /* distance between images in x / dx  (C_total*H*W)   */
    long long y_bstride;         /* distance between images in z / dz  (O_total*Ho*Wo) */
} KanGeom;

/* Basis description.
 *   B-spline : n_basis = grid_size + spline_order, order = spline_order,
 *              table[0 .. n_basis+order] = the fp32 knots of torch.linspace
 *              (layers/kan_layers.py:184-190); must be uniform, as the reference always
 *              builds them (non-uniform knots are rejected)
 *   RBF      : n_basis = grid_size, table[0..n_basis-1] = centres, p0 = denominator
 *              (utils/utils.py:28-30)
 *   Chebyshev: n_basis = degree + 1, p0 / p1 = clamp bounds (-1+1e-7, 1-1e-7 as fp32)
 *              (layers/cheby_kan_layers.py:93-96)
 *   Poly     : three-term-recurrence families (Bessel, Fibonacci, Gegenbauer, Hermite, Laguerre, Lucas, Taylor, Jacobi:
 *              layers/<family>_kan_layers.py compute_*_basis).  n_basis planes T_0..T_{n-1} of t = tanh(x) (order = 1)
 *              or t = x (order = 0):  T_0 = table[0], T_1 = table[1]*t + table[2],
 *              T_k = (table[3k-3]*t + table[3k-2]) * T_{k-1} + table[3k-1] * T_{k-2} for k >= 2;  n_basis <= 11
 *   Fourier  : n_basis = 2*grid_size planes cos(k x), k = 1..grid_size, then sin(k x)
 *              (layers/fourier_kan_layers.py:163-187)
 *   ReLU     : n_basis = g + k planes  (r * relu(x - lo[c][j]) * relu(hi[c][j] - x))^2, p0 = r = 4 g^2 / (k+1)^2
 *              (layers/relu_kan_layers.py:118-136).  The phases are PER CHANNEL and trainable, so they live in device
 *              memory: chan_table[c][0][j] = phase_low, chan_table[c][1][j] = phase_high, c = channel inside its group
 *              (the reference shares one phase tensor between the groups).  `order` selects what the non-derivative
 *              planes hold: 0 the basis; 1 / 2 its derivative w.r.t. phase_low / phase_high (base plane zero), so
 *              that kan_conv_bwd_weight yields the factor of the phase gradient,
 *              d phase[c][j] = sum_{o,tap} W[o][c*n+j][tap] * dW_mode[o][c*n+j][tap].
 *   Gram     : n_basis = degree + 1 planes act(P_k(tanh x)), P_0 = 1, P_1 = t, P_k = t P_{k-1} - c_k P_{k-2}, the planes
 *              passed through the layer's activation `act` (layers/gram_kan_layers.py:150-182).  The c_k derive from the
 *              trainable `beta_weights`, so they are read from device memory: chan_table[k] = c_k, k = 2..degree (entries 0, 1
 *              unused; one row for the whole layer).  `order` = 0: the basis; m >= 1: its derivative w.r.t. c_{m+1}
 *              (base plane zero), for the coefficient gradient through kan_conv_bwd_weight as for ReLU.
 * `act` is the base-branch activation (KAN_ACT_NONE: no base branch, no base weight).
 * Planes per channel P = n_basis + (act != KAN_ACT_NONE); P <= KAN_MAX_PLANES. */
typedef struct KanBasis {
    int kind, n_basis, order, act;
    float p0, p1;
    float table[KAN_MAX_TABLE];
    const float* chan_table;      /* device pointer: KAN_BASIS_RELU [C][2][n_basis] floats, KAN_BASIS_GRAM [n_basis] floats; else NULL */
} KanBasis;

/* Launch plan for one geometry: split counts and workspace sizes (bytes). */
typedef struct KanPlan {
    int P;                        /* planes per channel */
    int K;                        /* GEMM depth of the forward conv: C*kh*kw*P (= rows of the packed weight gradient) */
    int IPC, KC;                  /* forward packing: IPC (c,tap) items per LDS step of KC = even(IPC*P) rows */
    int Kpad, Opad;               /* dims of the packed weight matrix [Kpad][Opad], Kpad = ceil(C*kh*kw / IPC) * KC */
    int fwd_splits;               /* z is written as fwd_splits partial slabs */
    int bwd_data_splits;          /* dx is written as bwd_data_splits partial slabs */
    int bwd_weight_splits;        /* packed dW is written as bwd_weight_splits partial slabs */
    int fwd_target, bwd_data_target, bwd_weight_target;   /* position-major launches: live steps per split (0 = n/a) */
    int x_pm_wanted, dz_pm_wanted;/* small padded planes: pass position-major copies (kan_position_major) of x / dz to
                                     unlock structural-zero tap skipping; optional, NULL keeps the image-major path */
    int bwd_weight_expanded;      /* the weight gradient reads the expanded copy (kan_conv_bwd_weight_expanded) */
    int row_blocks;               /* informational: bit 0 / bit 1 = the forward / bwd-data launch orders 4x4 planes in row blocks and skips
                                     the 1/6 of its MFMA work that multiplies the zero border */
    int e_pm_wanted, fwd_expanded;/* small padded planes: the weight gradient (and, with fwd_expanded, the forward) reads the EXPANDED
                                     position-major copy: build it with kan_position_major_expanded (e_pm_elems floats) and call
                                     kan_conv_bwd_weight_expanded / kan_conv_fwd_expanded */
    int fwd_halo, bwd_weight_halo;/* informational: the forward / weight-gradient launch of this geometry uses the halo-tile kernel
                                     (k_conv_fwd_halo / k_conv_bwd_weight_halo) -- profiling tools name their samples by it */
    int fwd_band, bwd_weight_band;/* informational: the forward / weight-gradient launch uses the band kernels (few input channels, or an
                                     output count that fills no 128-wide tile: k_band_fwd / k_band_bwd_weight); wp is then in band order */
    long long packed_weight_bytes;    /* G*Kpad*Opad*4    : all groups, group j at j*Kpad*Opad floats */
    long long bwd_data_weight_bytes;  /* size of the bwd-data weight layout `wd`, all groups (equal blocks) */
    long long fwd_slab_elems;         /* B*y_bstride      : stride between z slabs  */
    long long bwd_data_slab_elems;    /* B*x_bstride      : stride between dx slabs */
    long long bwd_weight_slab_elems;  /* G*K*Opad         : stride between dW slabs, group j at j*K*Opad inside a slab */
    long long e_pm_elems;             /* floats of the expanded position-major copy (0 unless e_pm_wanted) */
} KanPlan;

const char* kan_version(void);
const char* kan_last_error(void);

/* Fill `plan` for (geom, basis).  Pure host arithmetic, no device work. */
int kan_plan(const KanGeom* geom, const KanBasis* basis, KanPlan* plan);

/* Pack the reference-layout weights of one group into the GEMM layouts of the kernels:
 *   wp (forward):   wp[k(item,p)][o],  item = tap*C + c (tap-major), T = kh*kw,
 *                   k = (item / IPC)*KC + (item % IPC)*P + p;  plane p = 0 is the base branch
 *                   (if any), planes hb.. are basis k = p - hb;  plan.packed_weight_bytes.
 *                   (3x3 / stride 1 / pad 1 layers of the default B-spline, ChebyKAN degree-3 and one-input recurrence
 *                   degree-3 specs on 32x32, 16x16, 8x8, 4x4 planes use the pair order of the halo forward kernel instead:
 *                   k = ((c/2)*T + tap)*2P + 2p + (c&1).  Poly specs with order = 0 -- basis on a second tensor xn != x,
 *                   LegendreKAN -- never do.  Layers served by the band kernels -- plan.fwd_band -- use
 *                   k = ((phase, channel group, tap of the phase) step) * KC + (c % IPC) * P + p with KC = even(IPC * P).
 *                   wp is opaque to the caller either way: it is only ever passed back to kan_conv_fwd of the same geometry and basis.)
 *   wd (bwd-data):  optional (NULL to skip), plan.bwd_data_weight_bytes (for depthwise groups -- C = 1, O <= 2, <= 9 taps,
 *                   which run on direct kernels -- it is a plain copy of wp):
 *                   wd[tap*Opad32 + o][ct*128 + cl*P + p],  c = ct*(128/P) + cl.
 * Replaces nothing in the reference (layout only); sources are  base_conv[g].weight [O,C,kh,kw]  (kan_layers.py:159-166) and
 * spline_conv[g].weight / poly_conv[g].weight [O,C*n_basis,kh,kw], channel c*n_basis+k
 * (kan_layers.py:170-177,237; fast_kan_layers.py:68-75,107; cheby_kan_layers.py:77-84,95).
 * `w_base` may be NULL iff basis->act == KAN_ACT_NONE.  With geom->groups = G the sources are the per-group weights
 * stacked on a leading axis, [G,O,C,kh,kw] and [G,O,C*n_basis,kh,kw]. */
int kan_pack_weights(const float* w_base, const float* w_basis, float* wp, float* wd,
                     const KanGeom* geom, const KanBasis* basis, void* stream);

/* The same with the packed layouts kept by the caller between calls (single-group geometries for which kan_pack_cacheable
 * returns 1; 16-byte aligned weight tensors).  Every call fingerprints the reference-layout weights on the device -- the two 32-bit sums
 * sum_i bits_i * (2 i + 1) and sum_i rotl(bits_i, i mod 32), over all elements or over every sample_stride-th group of
 * four when sample_stride > 1 -- into slot `cur` of `ring`, and the pack kernels return
 * at once when slot cur equals slot (cur + 2) % 3, i.e. when the weights are bit-for-bit what they were at the previous call: no
 * host round trip, and writes that bypass the framework's version counters are still seen.  `ring` is KAN_FP_WORDS device words (three slots), zero
 * before the first call; the caller passes cur = 0, 1, 2, 0, ... on successive calls (the call also clears slot cur + 1).
 * force != 0 packs unconditionally (first call, new buffers, a weight change the host already knows of).  wd may be NULL (no
 * bwd-data layout wanted): a later call that wants it must pass force. */
int kan_pack_cacheable(const KanGeom* geom, const KanBasis* basis);
int kan_pack_weights_cached(const float* w_base, const float* w_basis, float* wp, float* wd,
                            const KanGeom* geom, const KanBasis* basis, unsigned long long* ring, int cur, int force,
                            int sample_stride, void* stream);

/* Fused forward:  z = act(x) (*) W_base + sum_k basis_k(xn) (*) W_basis[:, c*n+k]
 * with zero padding applied to the EXPANDED operand.  Replaces
 *   kan_layers.py:199-200, 203-239   (B-spline: activation, knot broadcast, indicator,
 *                                     Cox-de Boor, moveaxis/flatten, two convs, add)
 *   fast_kan_layers.py:103, 106-109  (RBF; xn = normalised input, x = raw input)
 *   cheby_kan_layers.py:93-97        (Chebyshev; no base branch)
 * `xn` is the tensor the basis is evaluated on (== x except for FastKAN / LegendreKAN, and for a host-applied base
 * activation: x = act(input), xn = input, act = KAN_ACT_IDENTITY).  The compile-time specs of the single-input kinds
 * (default B-spline with SiLU/GELU, ChebyKAN, ReLU-KAN, GRAM) reject xn != x.
 * z holds plan.fwd_splits slabs of plan.fwd_slab_elems elements; their sum is the result
 * (kan_instnorm_prelu_fwd / kan_slab_reduce consume slabs directly). */
int kan_conv_fwd(const float* x, const float* xn, const float* wp, float* z,
                 const KanGeom* geom, const KanBasis* basis, const float* x_pm, void* stream);

/* dst[(c*HW + i)*B + b] = src[b*bstride + c*HW + i]  (B images, Cn channels of HW pixels; Cn = G*C or G*O of a grouped
 * call): the position-major copy the conv kernels read on small padded planes, where whole taps are structurally zero for a given output position
 * (the reference multiplies those zeros: kan_layers.py:239 zero-pads the expanded tensor). */
int kan_position_major(const float* src, float* dst, int B, int Cn, int HW, long long bstride, void* stream);

/* Gradient w.r.t. the input (autograd of the sites listed at kan_conv_fwd):
 *   dx  = act'(x) * dgrad(dz, W_base)              (base branch)
 *   dxn = sum_k basis_k'(xn) * dgrad(dz, W_basis)_k
 * If dxn == NULL the two are summed into dx (valid when xn == x).  Both are written as
 * plan.bwd_data_splits slabs of plan.bwd_data_slab_elems elements. */
int kan_conv_bwd_data(const float* dz, const float* x, const float* xn, const float* wd,
                      float* dx, float* dxn,
                      const KanGeom* geom, const KanBasis* basis, const float* dz_pm, void* stream);

/* kan_conv_bwd_data that also accumulates the gradient of the basis' trainable device-memory parameters from the same launch.  ReLU-KAN
 * (relu_kan_layers.py:127-131): dparams[C][2][n_basis] (d phase_low, d phase_high per channel of a group; groups share the table and add up).
 * GRAM-KAN (gram_kan_layers.py:156-182): dparams[64][n_basis], 64 slot rows of partial sums of d c_k (entries k < 2 stay 0) that the caller
 * adds up.  The caller zeroes dparams beforehand; every split adds its share with float atomics (scalars; dx stays atomic-free).  Replaces
 * the runs of the weight-gradient kernel on the parameter-derivative planes (two for ReLU-KAN, n_basis - 2 for GRAM). */
int kan_conv_bwd_data_params(const float* dz, const float* x, const float* xn, const float* wd, float* dx, float* dxn, float* dparams,
                             const KanGeom* geom, const KanBasis* basis, const float* dz_pm, void* stream);

/* Gradient w.r.t. the weights in the FLAT packed layout dwp[(tap*C+c)*P + p][o]
 * = sum_pixels expanded[k][pixel] * dz[o][pixel], written as plan.bwd_weight_splits slabs of plan.bwd_weight_slab_elems elements. */
int kan_conv_bwd_weight(const float* dz, const float* x, const float* xn, float* dwp,
                        const KanGeom* geom, const KanBasis* basis, const float* x_pm, const float* dz_pm, void* stream);

/* Small padded planes (plan.e_pm_wanted): the expanded operand in position-major order,
 *   e_pm[((c*HW + pos)*P + p)*B + b] = plane_p(x[b][c][pos])        (c over all G*C channels; plan.e_pm_elems floats),
 * materialised once per layer call -- on 4x4 / 2x2 planes it is 19 - 75 MB next to 85 MB of weights -- and the weight gradient on it:
 * a DMA + MFMA kernel (both operands by LDS-DMA, the (position, tap) pairs that read padding skipped), same dwp layout and slab
 * count as kan_conv_bwd_weight.  dz_pm = kan_position_major(dz).  Replaces the same reference lines as kan_conv_bwd_weight. */
int kan_position_major_expanded(const float* x, float* e_pm, const KanGeom* geom, const KanBasis* basis, void* stream);
int kan_conv_fwd_expanded(const float* e_pm, const float* wp, float* z, const KanGeom* geom, const KanBasis* basis, void* stream);   /* plan.fwd_expanded; z slabs as kan_conv_fwd */
int kan_conv_bwd_weight_expanded(const float* dz_pm, const float* e_pm, float* dwp,
                                 const KanGeom* geom, const KanBasis* basis, void* stream);

/* Sum the dwp slabs and scatter back to the reference layouts (inverse of kan_pack_weights; stacked [G,...] when
 * geom->groups = G).  dw_base may be NULL iff there is no base branch.  dwp is scratch: with >= 32 slabs the sum is
 * first folded into slab 0 in place. */
int kan_unpack_wgrad(const float* dwp, float* dw_base, float* dw_basis,
                     const KanGeom* geom, const KanBasis* basis, void* stream);

/* out = sum_s slabs[s]  over an NCHW block (B, Cn channels, HW pixels, batch stride bstride). */
int kan_slab_reduce(const float* slabs, int n_slabs, long long slab_elems, float* out,
                    int B, int Cn, int HW, long long bstride, void* stream);

/* InstanceNorm2d (eps, biased variance, per (b,channel) plane, optional affine gamma/beta)
 * followed by an optional scalar-slope PReLU; replaces kan_layers.py:241-243
 * (layer_norm -> prelus), cheby_kan_layers.py:98 and fast_kan_layers.py:106 (norm only:
 * pass prelu_a = NULL).  `z` holds n_slabs partial slabs (their sum is normalised); the
 * summed pre-norm value is written to z_out (may alias slab 0), which the backward needs.
 * mean / rstd: [B*Cn] outputs saved for the backward.
 * prelu_span: 0 = one slope prelu_a[0] for every channel (nn.PReLU()); k > 0 = channel c uses prelu_a[c / k] -- the
 * per-group slopes prelus[g] of a grouped layer (kan_layers.py:182,243) concatenated, k = channels per group. */
int kan_instnorm_prelu_fwd(const float* z, int n_slabs, long long slab_elems, float* z_out,
                           const float* gamma, const float* beta, const float* prelu_a,
                           float* y, float* mean, float* rstd,
                           int B, int Cn, int HW, long long bstride, float eps, int prelu_span, void* stream);

/* Backward of the above.  dz <- gradient w.r.t. the summed pre-norm value.
 * dgamma/dbeta ([Cn]) and dprelu ([1], or [Cn / prelu_span]) are ACCUMULATED with atomics: zero them first.
 * Any of gamma/prelu_a/dgamma/dbeta/dprelu may be NULL when that feature is off. */
int kan_instnorm_prelu_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, const float* prelu_a,
                           float* dz, float* dgamma, float* dbeta, float* dprelu,
                           int B, int Cn, int HW, long long bstride, int prelu_span, void* stream);

/* The same pair with MaxPool2d(kernel 2, stride 2) fused behind the PReLU -- the VGG pattern (models/kan_vgg.py:97-101: a
 * "M" entry right after the layer).  y_pooled / dy_pooled are dense [B][Cn][H/2][W/2]; pool_idx (same shape, bytes) holds the
 * position 2*dh + dw of each window's maximum (first maximum in scan order, NaN wins: torch's max_pool2d rule) and carries
 * the routing to the backward.  The full-size activation and its gradient never touch HBM.  H and W must be even. */
int kan_instnorm_prelu_pool_fwd(const float* z, int n_slabs, long long slab_elems, float* z_out,
                                const float* gamma, const float* beta, const float* prelu_a,
                                float* y_pooled, unsigned char* pool_idx, float* mean, float* rstd,
                                int B, int Cn, int H, int W, long long bstride, float eps, int prelu_span, void* stream);
int kan_instnorm_prelu_pool_bwd(const float* dy_pooled, const unsigned char* pool_idx, const float* z, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, const float* prelu_a,
                                float* dz, float* dgamma, float* dbeta, float* dprelu,
                                int B, int Cn, int H, int W, long long bstride, int prelu_span, void* stream);

/* The same with a general MaxPool2d(pool_k, pool_s) (no padding, floor mode; 2 <= pool_k <= 15, 1 <= pool_s <= pool_k) -- the AlexNet pattern
 * MaxPool2d(kernel_size=3, stride=2) right after a layer (models/kan_alexnet.py:120-126 `features`).  y_pooled / dy_pooled / pool_idx are dense
 * [B][Cn][Hp][Wp], Hp = (H - pool_k) / pool_s + 1; pool_idx = dh * pool_k + dw of the window's maximum (torch's rule, as above).  Windows overlap:
 * the backward of an element sums the pooled gradients of every window that picked it, as torch's max_pool2d backward does. */
int kan_instnorm_prelu_poolk_fwd(const float* z, int n_slabs, long long slab_elems, float* z_out,
                                 const float* gamma, const float* beta, const float* prelu_a,
                                 float* y_pooled, unsigned char* pool_idx, float* mean, float* rstd,
                                 int B, int Cn, int H, int W, long long bstride, float eps, int prelu_span, int pool_k, int pool_s, void* stream);
int kan_instnorm_prelu_poolk_bwd(const float* dy_pooled, const unsigned char* pool_idx, const float* z, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, const float* prelu_a,
                                 float* dz, float* dgamma, float* dbeta, float* dprelu,
                                 int B, int Cn, int H, int W, long long bstride, int prelu_span, int pool_k, int pool_s, void* stream);

/* OPT-IN split-precision forward: NOT reached from kan_conv_fwd, never the default.  Every fp32 operand is cut into three bf16
 * pieces (hi + mid + lo = 24 mantissa bits) and six bf16 MFMA products per 16-deep block are accumulated in fp32: the fp32 result to ~4e-6 of its
 * largest element at K = 20 736 (one fp32 accumulation chain; the exact path sits at ~1e-6), at ~1.7x the speed of the exact fp32 MFMA kernel.
 * Scope: the default B-spline spec (n_basis 8, order 3, SiLU) on 8x8 or 16x16 planes, 3x3 / stride 1 / pad 1, one group, C % 8 == 0, O % 128 == 0, even B on 8x8
 * planes, dense NCHW -- KAN-VGG11's 64 -> 128 @ 16x16, 128 -> 256 and 256 -> 256 @ 8x8 layers.  Computes what kan_conv_fwd computes for such a layer (kan_layers.py:199-200, 203-239) into ONE slab.
 *   kan_split_supported      1 if (geom, basis) is in scope
 *   kan_split_weight_bytes   size of the cut-weight buffer `wc` (0 if out of scope)
 *   kan_split_pack_weights   reference-layout weights (as kan_pack_weights takes them) -> wc; once per weight update
 *   kan_conv_fwd_split       z[B][O][H][W] = the conv stage */
int kan_split_supported(const KanGeom* geom, const KanBasis* basis);
long long kan_split_weight_bytes(const KanGeom* geom, const KanBasis* basis);
int kan_split_pack_weights(const float* w_base, const float* w_basis, void* wc, const KanGeom* geom, const KanBasis* basis, void* stream);
int kan_conv_fwd_split(const float* x, const void* wc, float* z, const KanGeom* geom, const KanBasis* basis, void* stream);

/* One AdamW step over a flat fp32 block of n elements, in place (p, m = exp_avg, v = exp_avg_sq; g is read only and
 * multiplied by grad_scale first).  Replaces the per-tensor update loop of torch.optim.AdamW as the reference builds it
 * (generic_train.py:24 `optim.AdamW(model.parameters(), lr, weight_decay)`, stepped once per batch: evaluations.py train()),
 * same formulas in the same order, amsgrad / maximize off:
 *   p *= 1 - lr*weight_decay;  m += (g - m)(1 - beta1);  v = v*beta2 + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` counts from 1.  The four blocks must be 16-byte aligned.  HBM-bound: 28 bytes per element. */
int kan_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int step, float grad_scale, void* stream);

/* The same step with the gradients left where autograd (or an all-reduce bucket) wrote them.  p / m / v are flat blocks
 * holding n_seg segments (one per parameter); segment s covers elements [seg_off[s], seg_off[s] + seg_n[s]) and reads its
 * gradient from the device address seg_grad[s] (0 = no gradient this step: the segment is skipped, as torch skips it).
 * Workgroup b updates elements [chunk_start[b], chunk_start[b] + chunk_elems) of segment chunk_seg[b]; the caller builds
 * the chunk table once per layout (seg_off and chunk_start multiples of 4 elements, chunk_elems a multiple of 4).  All
 * tables are DEVICE arrays; only seg_grad changes between steps.  torch counts steps PER PARAMETER (one that had no
 * gradient for a while is bias-corrected with its own count): seg_bias, if not NULL, holds per segment
 * { -lr / (1 - beta1^step_s), 1 / sqrt(1 - beta2^step_s) } and overrides the values derived from `step`. */
int kan_adamw_step_segments(float* p, float* m, float* v, const unsigned long long* seg_grad, const long long* seg_off, const int* seg_n,
                            const int* chunk_seg, const int* chunk_start, const float* seg_bias, int n_chunks, int chunk_elems, double lr,
                            double beta1, double beta2, double eps, double weight_decay, int step, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------- Wav-KAN wavelet stage
 * u[b, o, ho, wo] = sum_{c, r, t} w[o, c, r, t] * psi((x[b, c, hi, wi] - trans[o, c]) / scale[o, c]),  zero padding applied to the
 * wavelet values.  Replaces WaveletConvND / WaveletConvNDFast / WaveletConvNDFastPlusOne .forward up to (not including) the 1x1
 * `wavelet_out` conv (wav_kan_layers.py:186-217, 257-276, 318-338): the three versions are this arithmetic over differently shaped
 * weight tensors.  One group per call (pre-offset x / u by the group's first channel; x_bstride / u_bstride = elements between
 * images).  scale, trans: [O][C]; w: [O][C][kh][kw] (the 'fast' layout; 'base' and 'fast_plus_one' reshape to it).  fp32, direct
 * vector-ALU kernels (every (o, c) pair has its own wavelet: nothing to share through a GEMM). */
enum { KAN_WAV_MEXICAN_HAT = 0, KAN_WAV_MORLET = 1, KAN_WAV_DOG = 2, KAN_WAV_MEYER = 3, KAN_WAV_SHANNON = 4 };
typedef struct {
    int B, C, H, W, O, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
    int wavelet;                      /* KAN_WAV_* */
    long long x_bstride, u_bstride;
} KanWavGeom;
int kan_wav_fwd(const float* x, const float* scale, const float* trans, const float* w, float* u, const KanWavGeom* geom, void* stream);
/* dx = d loss / d x through the wavelets (autograd of the same lines); du: [B][O][Ho][Wo] with u_bstride */
int kan_wav_bwd_input(const float* du, const float* x, const float* scale, const float* trans, const float* w, float* dx,
                      const KanWavGeom* geom, void* stream);
/* dw [O][C][kh][kw], dscale / dtrans [O][C]; workspace: kan_wav_param_workspace(geom) floats of scratch (partial sums, added in a fixed order) */
long long kan_wav_param_workspace(const KanWavGeom* geom);
int kan_wav_bwd_params(const float* du, const float* x, const float* scale, const float* trans, const float* w, float* dw, float* dscale,
                       float* dtrans, float* workspace, const KanWavGeom* geom, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KANCONV_H */
