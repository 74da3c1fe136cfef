// Host-side interfaces between the translation units of libkanconv (not part of the C ABI; nothing here is exported).
//   kanconv.hip     C-ABI entry points, planning, the tap-major / halo / position-major GEMM kernels, layout and norm kernels
//   kan_direct.hip  band kernels: layers of few input channels (first layers: 3 -> 64) and layers whose output count fills no 128-wide
//                   tile (64 -> 192), any kernel size / stride / dilation / padding
#pragma once
#include "kanconv.h"

// sets the thread-local message behind kan_last_error() and returns -1 (defined in kanconv.hip)
int kan_fail_msg(const char* fmt, const char* a);

// ---------------------------------------------------------------------------------------------------------------- band kernels
// The tap-major kernels expand every input value once per TAP it is used under (9x for 3x3, 121x for 11x11) -- vector-ALU work the fp32
// MFMA shares its issue slots with; with 3 input channels the GEMM depth is so short (243 rows) that this staging IS the kernel (0.34 -
// 0.40 matrix-pipe busy).  The band kernels expand the inputs a pixel tile can touch ONCE per channel group into an LDS halo tile and let
// every tap read it through a shifted address (the idea of k_conv_fwd_halo), for any geometry:
//   * strided taps by PHASE: input row = sh*ho - ph + dh*r = sh*(ho + off_r) + a_r, so the taps of one (row phase a, column phase b)
//     read the sub-image x[sh*i + a][sw*j + b] at (i, j) = (ho + off_r, wo + off_t): a stride-1 pattern.  A stride-s layer is
//     s*s phases of stride-1 sub-convolutions over the same output tile;
//   * a pixel tile is TP CONSECUTIVE output pixels in (image, row, column) order (no row-band restriction, so 27- or 55-wide planes fill
//     their 128-pixel tiles); its halo holds one "virtual row" per touched output row plus span_r extra rows PER IMAGE, so that rows of
//     different images never alias:  cell(pixel) = vrow * HC + wo,  B operand of tap (r, t) = cell + (off_r - OR0) * HC + (off_t - OC0);
//   * the halo is plane-minor, sH[cell][NPS] with NPS odd: the planes of a k-pair are an IMMEDIATE offset apart (no address arithmetic in
//     the MFMA loop) and 32 consecutive pixels hit 32 banks;
//   * depth order: step = (phase, channel group, tap of the phase), NPLE = even(NG * P) rows per step (row = ch * P + p, a zero pad row
//     when NG * P is odd); weights are packed in that order (kan_pack_weights follows the plan) and arrive by LDS-DMA.
// tools/probe/band_emul.py is the numpy restatement of this index arithmetic, checked against conv2d on random geometries.
#define KAN_BAND_MAX_TAPS 128
#define KAN_BAND_MAX_PHASES 16
typedef struct KanBandCfg {
    int ok;                       /* 0: this geometry / basis does not take the band kernels */
    int fast, P, NG, NPLE, NPS;   /* compile-time spec, planes per channel, channels per group, rows per step, halo words per cell */
    int NGR;                      /* channel groups (ceil(C / NG)) */
    int WO, MO, WP, TO, TP, NT;   /* waves along outputs, 32-output MFMA blocks per wave, waves along pixels; tile TO = 32 MO WO x TP = 64 WP; threads */
    int tiles_o, tiles_p;
    int n_phase, n_taps, n_steps; /* n_steps = NGR * n_taps weight steps of NPLE rows */
    int HC, span_r, OR0, OC0;     /* halo row length in cells, extra rows per image, smallest row / column offset of any tap */
    int cells;                    /* halo cells allocated per tile (largest tile) */
    int slots;                    /* expansion units per thread: ceil(NG * cells / NT) */
    int TS;                       /* taps per barrier step of the forward (2 where one tap is under ~36 MFMAs per wave) */
    int lds_bytes, wgs_per_cu;
    int fwd_splits;               /* split-K over the (phase, group) list */
    /* weight gradient (k_band_bwd_weight): row tiles of TR = 32 WR rows inside one (phase, channel group), all TO = 32 NI outputs per wave */
    int bw_ok, bw_WR, bw_NI, bw_NT, bw_tiles_o, bw_row_tiles, bw_TPI, bw_ptiles, bw_cells, bw_slots, bw_lds_bytes, bw_splits;
    unsigned short bw_ph_rt0[KAN_BAND_MAX_PHASES + 1];                      /* first row tile of each phase (NGR * ceil(nt * NPLE / TR) tiles per phase) */
    unsigned char ph_a[KAN_BAND_MAX_PHASES], ph_b[KAN_BAND_MAX_PHASES];     /* row / column phase of phase i */
    short ph_tap0[KAN_BAND_MAX_PHASES + 1];                                 /* first entry of phase i in the tap lists below */
    unsigned short tap_shift[KAN_BAND_MAX_TAPS];                            /* phase-ordered: halo shift of the tap, in cells */
    unsigned short tap_rt[KAN_BAND_MAX_TAPS];                               /* phase-ordered: r << 8 | t */
    short tap_step[KAN_BAND_MAX_TAPS], tap_nt[KAN_BAND_MAX_TAPS];           /* natural tap r*kw + t -> step of (group 0, this tap), taps in its phase:
                                                                               step(c, tap) = tap_step[tap] + (c / NG) * tap_nt[tap] */
} KanBandCfg;

/* Fill cfg for (geom, basis); `fast` = the compile-time spec of the basis (0: none -> cfg->ok = 0).  Pure host arithmetic. */
void kan_band_cfg(const KanGeom* g, const KanBasis* b, int fast, KanBandCfg* cfg);
/* dwp: bw_splits slabs of G * Kpad * Opad floats in band order (rows as the packed forward weights: kan_unpack_wgrad follows the plan). */
int kan_band_bwd_weight_launch(const float* dz, const float* x, const float* xn, float* dwp, const KanGeom* g, const KanBasis* b,
                               const KanBandCfg* cfg, int splits, long long slab_elems, void* stream);
/* z slabs [fwd_splits][B][O_total][Ho][Wo] as kan_conv_fwd; wp in band order. */
int kan_band_fwd_launch(const float* x, const float* xn, const float* wp, float* z, const KanGeom* g, const KanBasis* b,
                        const KanBandCfg* cfg, int splits, long long slab_elems, void* stream);
