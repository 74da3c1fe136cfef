// libkanconv, band kernels (see kan_internal.h for the design): layers of few input channels (3 -> 64 first layers, any kernel / stride)
// and layers whose output count fills no 128-wide tile (64 -> 192).  gfx950 only; fp32 MFMA 32x32x2.
//
// Forward:  z[o][pixel] = sum over steps (phase, channel group, tap) of  Wb[step][row][o] * sH[cell(pixel) + shift(tap)][row]
// with the expanded halo tile sH built once per (phase, channel group) and pixel tile.  tools/probe/band_emul.py restates the index
// arithmetic in numpy and checks it against conv2d; the formulas below follow it line by line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include "kanconv.h"
#include "kan_device.h"
#include "kan_common.h"
#include "kan_internal.h"

#define BAND_READ4(r0, r1, r2, r3, addrA, addrB0, addrB1, oA0, oA1, oB)                                                 \
    asm volatile("ds_read_b32 %0, %4 offset:%7\n\tds_read_b32 %1, %4 offset:%8\n\tds_read_b32 %2, %5 offset:%9\n\tds_read_b32 %3, %6 offset:%9" \
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addrA), "v"(addrB0), "v"(addrB1), "n"(oA0), "n"(oA1), "n"(oB) : "memory")

#define BAND_READ3(r0, r1, r2, addrA, addrB0, addrB1, oA, oB)                                                             \
    asm volatile("ds_read_b32 %0, %3 offset:%6\n\tds_read_b32 %1, %4 offset:%7\n\tds_read_b32 %2, %5 offset:%7"           \
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(addrA), "v"(addrB0), "v"(addrB1), "n"(oA), "n"(oB) : "memory")
#define BAND_WAIT3(r0, r1, r2, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(r0), "+v"(r1), "+v"(r2) :: "memory")

namespace {

// Device view of KanBandCfg (by-value kernel argument).
struct BandTab {
    int n_phase, NGR, HC, span_r, OR0, OC0, cells, n_steps, nslots;   // nslots: expansion units per thread actually needed (<= SLOTS)
    FastDiv divHC, divBlk, divCells, divNGR, divHoWo, divWo;   // by HC, by Ho + span_r, by cells, by NGR, by Ho * Wo, by Wo
    unsigned ph_pack[KAN_BAND_MAX_PHASES];                     // a | b << 8 | first tap << 16 | taps << 24
    unsigned short tap_shift[KAN_BAND_MAX_TAPS];               // phase-ordered, in cells
};

// Workgroup: WO x WP waves, one (32 MO outputs) x (64 pixels) tile per wave (MO x 2 MFMA 32x32x2) => TO = 32 MO WO outputs x TP = 64 WP pixels.
// MO = 1 (64-output layers): 128-pixel tiles on four waves, so that the halo tile -- the LDS bill of this kernel -- covers 128 pixels, not 256.
// Dynamic LDS: [2 weight buffers | basis table | tap shifts | phase table | dump words | halo tile cells x NPS].
// A barrier step holds TS taps (TS * NPLE weight rows; TS = 2 where one tap is under ~36 MFMAs per wave: a barrier + DMA wait per 16 - 28
// MFMAs was the measured bound of the first version on the 5-plane ChebyKAN layers).
template <int KIND, int FAST, int NG, int WO, int MO, int WP, int SLOTS, int TS>
__global__ __launch_bounds__(WO * WP * 64, 2) void k_band_fwd(
    const float* __restrict__ x, const float* __restrict__ xn, const float* __restrict__ wp, float* __restrict__ z,
    DevGeom g, DevBasis bs, BandTab tb, int Opad, int groups_per_split, long long slab_elems, unsigned x_bytes, int tiles_o) {
    constexpr int P = fast_planes(FAST), NPL = NG * P, NPLE = NPL + (NPL & 1), NPS = NPLE + 1;
    constexpr int TO = WO * MO * 32, TP = WP * 64, NT = WO * WP * 64, NW = WO * WP;
    constexpr int WBUF = ((TS * NPLE * TO + 255) / 256) * 256;  // floats per weight buffer: whole 1-KiB wave copies
    constexpr int NQ = WBUF / 256, NQW = (NQ + NW - 1) / NW;    // wave copies per step / per wave
    static_assert(NT >= KAN_BAND_MAX_TAPS && NT >= KAN_MAX_TABLE, "table staging assumes one thread per entry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sW = smem;
    float* const sTab = sW + 2 * WBUF;
    int* const sShift = reinterpret_cast<int*>(sTab + KAN_MAX_TABLE);         // byte shift of the phase-ordered tap
    unsigned* const sPh = reinterpret_cast<unsigned*>(sShift + KAN_BAND_MAX_TAPS);
    float* const sDump = reinterpret_cast<float*>(sPh + KAN_BAND_MAX_PHASES);
    float* const sH = sDump + NT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_o = wave / WP, w_p = wave % WP;
    const int HoWo = g.Ho * g.Wo, HW = g.H * g.W, Mtot = g.B * HoWo;
    const BlockId blk = xcd_block_order(true);
    const int grp = blk.y / tiles_o;
    const int px_tile0 = blk.x * TP, o_tile0 = (blk.y - grp * tiles_o) * TO;
    {
        const size_t xo = (size_t)grp * g.C * HW;
        x += xo; xn += xo;
        z += (size_t)grp * g.O * HoWo;
        wp += (size_t)grp * tb.n_steps * NPLE * Opad;
    }
    if (tid < KAN_MAX_TABLE) sTab[tid] = bs.tab[tid];
    if (tid < KAN_BAND_MAX_TAPS) sShift[tid] = (int)tb.tap_shift[tid] * (NPS * 4);
    if (tid < KAN_BAND_MAX_PHASES) sPh[tid] = tb.ph_pack[tid];
    // every plane word of every cell is (re)written by each group's expansion, zeros included; what no expansion writes is the zero row
    // that pads an odd plane count: it is multiplied by zero weights, but must be finite.  (Round 3: the first version zero-filled the
    // whole tile -- 23 LDS stores per thread and tile -- and spent ~60 % of its 3.5 vector instructions per MFMA on runtime integer divisions,
    // unit decode of unused slots and that fill: PMC on 3 -> 64 @32x32.)
    if (NPL & 1)
        for (int i = tid; i < tb.cells; i += NT) sH[i * NPS + NPL] = 0.f;

    // ---- the tile: TP consecutive output pixels from px_tile0, in (image, row, column) order (band_emul.py: tile_layout)
    const int p_last = min(px_tile0 + TP, Mtot) - 1;
    const int b0 = fastdiv(px_tile0, tb.divHoWo), ho0 = fastdiv(px_tile0 - b0 * HoWo, tb.divWo);
    const int b1 = fastdiv(p_last, tb.divHoWo), ho1 = fastdiv(p_last - b1 * HoWo, tb.divWo);
    const int n0 = (b0 == b1 ? ho1 : g.Ho - 1) - ho0 + 1;                   // output rows of the first image
    const int blk0 = n0 + tb.span_r, blkN = g.Ho + tb.span_r;               // virtual rows of image 0 / of every later image

    // B-operand base of this lane's two pixels: cell * NPS words, the odd plane of a k-pair for lanes 32 - 63
    const int kh2 = lane >> 5;
    unsigned vb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int px = px_tile0 + w_p * 64 + q * 32 + (lane & 31);
        int cell = 0;
        if (px < Mtot) {
            const int b = fastdiv(px, tb.divHoWo), r = px - b * HoWo, ho = fastdiv(r, tb.divWo), wo = r - ho * g.Wo, k = b - b0;
            const int v = k == 0 ? ho - ho0 : blk0 + (k - 1) * blkN + ho;
            cell = v * tb.HC + wo;
        }
        vb[q] = lds_addr(sH) + (unsigned)(cell * NPS + kh2) * 4u;
    }

    // ---- expansion units of this thread (fixed per tile): unit = channel-of-group * cells + cell -> (image, sub-row i, sub-column j)
    int u_ij[SLOTS], u_base[SLOTS], u_dst[SLOTS]; unsigned u_ok = 0, u_img = 0;   // u_ij = i (low 16, signed) | j << 16;  u_dst = LDS word | ch << 24
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        if (k >= tb.nslots) continue;                        // uniform: slots this geometry does not need cost nothing
        const int u = tid + k * NT;
        const int ch = fastdiv(u, tb.divCells), cell = u - ch * tb.cells;
        const int v = fastdiv(cell, tb.divHC), jj = cell - v * tb.HC;
        int kimg = 0, lv = v, first = ho0;
        if (v >= blk0) { const int w2 = v - blk0, q2 = fastdiv(w2, tb.divBlk); kimg = 1 + q2; lv = w2 - q2 * blkN; first = 0; }
        const int i = first + lv + tb.OR0, j = jj + tb.OC0, img = b0 + kimg;
        u_ij[k] = (i & 0xffff) | (j << 16);
        u_base[k] = img * (int)g.xbs + ch * HW;
        u_dst[k] = (cell * NPS + ch * P) | (ch << 24);
        u_ok |= ((ch < NG) ? 1u : 0u) << k;                  // the unit exists: its planes are written on every expansion (zeros where there is no input)
        u_img |= ((img < g.B) ? 1u : 0u) << k;
    }
    const bool same_in = (KIND != KAN_BASIS_RBF && KIND != KAN_BASIS_POLY) || (x == xn);
    const kan_rsrc x_rs = make_rsrc(x, x_bytes), xn_rs = make_rsrc(same_in ? x : xn, x_bytes);
    float xa[SLOTS], xb[SLOTS]; unsigned inb_mask = 0; int s_cbase = 0;
    auto load_group = [&](int gi) {                          // request the inputs of (phase, group) gi
        const int ph = fastdiv(gi, tb.divNGR), gg = gi - ph * tb.NGR;
        const unsigned pk = __builtin_amdgcn_readfirstlane(sPh[ph]);
        const int pa = pk & 0xff, pb = (pk >> 8) & 0xff, cbase = gg * NG;
        inb_mask = 0; s_cbase = cbase;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            if (k >= tb.nslots) continue;
            const int i = (int)(short)(u_ij[k] & 0xffff), j = u_ij[k] >> 16, ch = u_dst[k] >> 24;
            const int row = g.sh * i + pa, col = g.sw * j + pb;
            const bool inb = ((u_ok & u_img) >> k) & 1u && cbase + ch < g.C && (unsigned)row < (unsigned)g.H && (unsigned)col < (unsigned)g.W;
            const unsigned off = inb ? (unsigned)(u_base[k] + cbase * HW + row * g.W + col) * 4u : KAN_OOB;
            xa[k] = buf_load(x_rs, off);
            xb[k] = same_in ? xa[k] : buf_load(xn_rs, off);
            inb_mask |= (inb ? 1u : 0u) << k;
        }
    };
    auto expand = [&]() {                                    // write the P planes of every unit (zeros outside the image / past the last channel)
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (k < tb.nslots && ((u_ok >> k) & 1u))
                stage_unit<KIND, FAST>(bs, sTab, (inb_mask >> k) & 1u, xa[k], xb[k], sH + (u_dst[k] & 0xffffff), 1, sDump + tid, s_cbase + (u_dst[k] >> 24));
    };

    // ---- weight steps: n_taps * NPLE rows x TO floats, a straight 2-D copy by LDS-DMA (16 bytes per lane, 1 KiB per wave instruction)
    unsigned w_off[NQW]; unsigned w_okm = 0, w_ok1 = 0;      // lanes inside TS taps' rows / inside one tap's rows (a phase's odd last tap)
#pragma unroll
    for (int j = 0; j < NQW; ++j) {
        const int q = j * NW + wave, f = q * 256 + 4 * lane, row = f / TO, col = f - row * TO;
        w_off[j] = (unsigned)(row * Opad + col) * 4u;
        w_okm |= ((q < NQ && row < TS * NPLE) ? 1u : 0u) << j;
        w_ok1 |= ((q < NQ && row < NPLE) ? 1u : 0u) << j;
    }
    auto issue_w = [&](int s, int buf, int n_taps) {          // taps s .. s + n_taps - 1 of the band order -> sW[buf]
        const char* wsrc = (const char*)(wp + (size_t)s * NPLE * Opad + o_tile0);
        float* dW = sW + buf * WBUF;
        const unsigned okm = n_taps == TS ? w_okm : w_ok1;
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const int q = j * NW + wave;                     // wave-uniform
            if (q < NQ) {
                if ((okm >> j) & 1u) glds16((const float*)(wsrc + w_off[j]), dW + q * 256);
            }
        }
    };

    f32x16 acc[MO][2];
#pragma unroll
    for (int a = 0; a < MO; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    __syncthreads();                                         // tables + zero fill visible
    const int NGI = tb.n_phase * tb.NGR;
    const int gi0 = blk.z * groups_per_split, gi1 = min(NGI, gi0 + groups_per_split);
    int s = 0, buf = 0;
    auto taps_of = [&](int gi) -> int { return (int)(__builtin_amdgcn_readfirstlane(sPh[fastdiv(gi, tb.divNGR)]) >> 24); };
    if (gi0 < gi1) {
        const int ph = fastdiv(gi0, tb.divNGR), gg = gi0 - ph * tb.NGR;
        const unsigned pk = __builtin_amdgcn_readfirstlane(sPh[ph]);
        s = tb.NGR * (int)((pk >> 16) & 0xff) + gg * (int)(pk >> 24);       // first weight step of (phase, group)
        issue_w(s, 0, min(TS, (int)(pk >> 24)));
        load_group(gi0);
    }
    const int ao = w_o * (MO * 32) + (lane & 31);
    for (int gi = gi0; gi < gi1; ++gi) {
        const int ph = fastdiv(gi, tb.divNGR);
        const unsigned pk = __builtin_amdgcn_readfirstlane(sPh[ph]);
        const int tap0 = (pk >> 16) & 0xff, nt = (int)(pk >> 24);
        const int nt_next = gi + 1 < gi1 ? taps_of(gi + 1) : 0;
        __syncthreads();                                     // every wave has finished reading the previous group's halo
        expand();
        const int n_bs = (nt + TS - 1) / TS;                 // barrier steps of this group
#pragma unroll 1
        for (int jb = 0; jb < n_bs; ++jb) {
            const int j0 = jb * TS, n_here = min(TS, nt - j0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own weight DMA of this step landed (the compiler does not tie LDS-DMA to the barrier)
            __syncthreads();                                 // ... everybody's; (jb == 0) halo visible; previous step's reads done
            const int n_next = jb + 1 < n_bs ? min(TS, nt - j0 - TS) : min(TS, nt_next);
            if (n_next > 0) issue_w(s + n_here, buf ^ 1, n_next);
            if (jb == (n_bs >> 1) && gi + 1 < gi1) load_group(gi + 1);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if (t < n_here) {                            // uniform
                    const unsigned sh = (unsigned)__builtin_amdgcn_readfirstlane(sShift[tap0 + j0 + t]);
                    const unsigned aw = lds_addr(sW + buf * WBUF + (t * NPLE + kh2) * TO + ao), ab0 = vb[0] + sh, ab1 = vb[1] + sh;
                    const unsigned ab[2] = {ab0, ab1};
                    float fa[2][MO], fb[2][2];
                    // operand fetch in explicit ISA (immediate offsets; the reads of k-pair kk + 1 are in flight during the MFMAs of kk)
#define BAND_RD(n_, KK)                                                                                                          \
                    do {                                                                                                         \
                        _Pragma("unroll") for (int mi = 0; mi < MO; ++mi)                                                        \
                            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[n_][mi]) : "v"(aw), "n"((2 * (KK)) * TO * 4 + mi * 128) : "memory"); \
                        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                            \
                            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[n_][q]) : "v"(ab[q]), "n"((2 * (KK)) * 4) : "memory"); \
                    } while (0)
                    BAND_RD(0, 0);
#pragma unroll
                    for (int kk = 0; kk < NPLE / 2; ++kk) {
                        const int c_ = kk & 1, n_ = c_ ^ 1;
                        if (kk + 1 < NPLE / 2) {
                            BAND_RD(n_, kk + 1);
                            if constexpr (MO == 1) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[c_][0]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                            else if constexpr (MO == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[c_][0]), "+v"(fa[c_][MO - 1]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                            else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[c_][0]), "+v"(fa[c_][1]), "+v"(fa[c_][MO - 1]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                        } else {
                            if constexpr (MO == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c_][0]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                            else if constexpr (MO == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c_][0]), "+v"(fa[c_][MO - 1]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c_][0]), "+v"(fa[c_][1]), "+v"(fa[c_][MO - 1]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                        }
#pragma unroll
                        for (int mi = 0; mi < MO; ++mi) {
                            acc[mi][0] = MFMA32(fa[c_][mi], fb[c_][0], acc[mi][0]);
                            acc[mi][1] = MFMA32(fa[c_][mi], fb[c_][1], acc[mi][1]);
                        }
                    }
#undef BAND_RD
                }
            }
            buf ^= 1; s += n_here;
        }
    }

    // ---- store: column (lane) = pixel => coalesced along the plane
    float* zs = z + (size_t)blk.z * slab_elems;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int px = px_tile0 + w_p * 64 + ni * 32 + (lane & 31);
        if (px >= Mtot) continue;
        const int b = fastdiv(px, tb.divHoWo), hw = px - b * HoWo;
        float* zb = zs + (size_t)b * g.ybs + hw;
#pragma unroll
        for (int mi = 0; mi < MO; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o_tile0 + w_o * (MO * 32) + mi * 32 + mfma_row(r, lane);
                if (o < g.O) zb[(size_t)o * HoWo] = acc[mi][ni][r];
            }
        }
    }
}

// ============================================================================ weight gradient, band order
// dWb[step][row][o] = sum over pixels of sH[cell(pixel) + shift(step)][row] * dz[o][pixel]  -- the forward's operands, transposed roles:
// the A operand of row (tap j, plane row q) at pixel p is ONE word of the halo tile, lane-constant base (tap shift + plane) plus the
// pixel's cell offset, so nothing is staged per tap.  Workgroup = WR waves, wave w owns rows [32 w, 32 w + 32) of a row tile of the
// (phase, channel group)'s nt * NPLE rows and all TO = 32 NI outputs (1 x NI MFMA blocks: ONE address add per k-pair serves NI MFMAs).
// Pixel tiles are 128 consecutive pixels of ONE image (per-image tiles: a 16-pixel step never straddles images, so dz arrives by
// LDS-DMA exactly as in k_conv_bwd_weight_halo, XOR-swizzled [o][16 px]); the pixels' cell offsets sit in a small LDS table.
// Per step and wave: 8 table reads + 8 address adds + 8 (1 + NI) operand reads + 8 NI MFMAs, TO / (4 WR) DMA instructions.
// Dynamic LDS: [2 dz buffers | basis table | tap shifts | phase table | pixel cell offsets | dump words | halo tile].
struct BandWTab {
    int n_phase, NGR, HC, span_r, OR0, OC0, cells, n_steps, TPI, n_ptiles;     // TPI: pixel tiles per image; n_ptiles = B * TPI
    FastDiv divHC, divCells, divTPI, divWo;
    unsigned ph_pack[KAN_BAND_MAX_PHASES];                     // a | b << 8 | first tap << 16 | taps << 24
    unsigned short ph_rt0[KAN_BAND_MAX_PHASES + 1];            // first row tile (grid x) of the phase: phase ph owns NGR * n_rt(ph) tiles
    unsigned short tap_shift[KAN_BAND_MAX_TAPS];
};

// (192 outputs: 96 accumulator registers per lane; the launch bound asks for three 4-wave workgroups per CU, i.e. <= 168 VGPRs -- at 169 the
//  64 -> 192 layer lost a workgroup per CU and ran 2.46 -> 2.91 ms)
template <int KIND, int FAST, int NG, int WR, int NI, int SLOTS>
__global__ __launch_bounds__(WR * 64, (NI == 6 && WR == 4) ? 3 : 2) void k_band_bwd_weight(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ xn, float* __restrict__ dwp,
    DevGeom g, DevBasis bs, BandWTab tb, int Kpad, int Opad, int ptiles_per_split, long long slab_elems, unsigned x_bytes, unsigned dz_bytes,
    int tiles_o) {
    constexpr int P = fast_planes(FAST), NPL = NG * P, NPLE = NPL + (NPL & 1), NPS = NPLE + 1;
    constexpr int TR = WR * 32, TO = NI * 32, NT = WR * 64, KPX = 16, TPX = 128, ZB = KPX * TO;
    constexpr int NZ = TO / 4, NZW = (NZ + WR - 1) / WR;        // dz DMA instructions (4 output rows x 16 pixels each) per step / per wave
    static_assert(NT >= KAN_BAND_MAX_TAPS && NT >= TPX, "table staging assumes one thread per entry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sZ = smem;                                     // 2 * ZB
    float* const sTab = sZ + 2 * ZB;
    int* const sShift = reinterpret_cast<int*>(sTab + KAN_MAX_TABLE);
    unsigned* const sPh = reinterpret_cast<unsigned*>(sShift + KAN_BAND_MAX_TAPS);
    int* const sCell = reinterpret_cast<int*>(sPh + KAN_BAND_MAX_PHASES);      // TPX byte offsets (cell * NPS * 4) of the tile's pixels
    float* const sDump = reinterpret_cast<float*>(sCell + TPX);
    float* const sH = sDump + NT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh2 = lane >> 5;
    const int HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int grp = (int)blockIdx.y / tiles_o;
    const int o_tile0 = ((int)blockIdx.y - grp * tiles_o) * TO;
    {
        const size_t xo = (size_t)grp * g.C * HW;
        x += xo; xn += xo;
        dz += (size_t)grp * g.O * HoWo;
        dwp += (size_t)grp * Kpad * Opad;
    }
    // ---- this workgroup's rows: grid x -> (phase, channel group, row tile)
    int ph = 0;
    while (ph + 1 < tb.n_phase && (int)blockIdx.x >= (int)tb.ph_rt0[ph + 1]) ++ph;
    const unsigned pk = tb.ph_pack[ph];
    const int pa = pk & 0xff, pb = (pk >> 8) & 0xff, tap0 = (pk >> 16) & 0xff, nt = (int)(pk >> 24);
    const int rows = nt * NPLE, n_rt = (rows + TR - 1) / TR;
    const int rel = (int)blockIdx.x - (int)tb.ph_rt0[ph], gg = rel / n_rt, rt = rel - gg * n_rt;
    const int cbase = gg * NG;
    const int row0 = (tb.NGR * tap0 + gg * nt) * NPLE + rt * TR;           // first row of this tile in the band order

    if (tid < KAN_MAX_TABLE) sTab[tid] = bs.tab[tid];
    if (tid < KAN_BAND_MAX_TAPS) sShift[tid] = (int)tb.tap_shift[tid] * (NPS * 4);
    if (NPL & 1)                                             // the zero row of an odd plane count (every other word is rewritten by each tile's expansion)
        for (int i = tid; i < tb.cells; i += NT) sH[i * NPS + NPL] = 0.f;
    __syncthreads();

    // A-operand base of this lane's row: tap shift + plane word (rows past the group's end: any valid address, discarded at the store)
    const int rho = rt * TR + wave * 32 + (lane & 31);
    unsigned aRow;
    {
        const int rr = rho < rows ? rho : 0, j = rr / NPLE, q = rr - j * NPLE;
        aRow = lds_addr(sH) + (unsigned)(sShift[tap0 + j] + q * 4);
    }
    const unsigned cellbase = lds_addr(reinterpret_cast<float*>(sCell)) + (unsigned)kh2 * 4u;
    // B-operand addresses: one per k-pair (XOR swizzle of k_conv_bwd_weight_halo); output block ni and the buffer are immediates
    unsigned bB[KPX / 2];
    {
        const int ol = lane & 31, sw = (ol >> 1) & 15;
#pragma unroll
        for (int kk = 0; kk < KPX / 2; ++kk) bB[kk] = lds_addr(sZ + ol * KPX + ((2 * kk + kh2) ^ sw));
    }
    // dz DMA: chunk m = 4 output rows x 16 pixels; lane -> (row, swizzled pixel)
    unsigned zoff[NZW]; int zq[NZW];
#pragma unroll
    for (int j = 0; j < NZW; ++j) {
        const int m = j * WR + wave, ol = 4 * m + (lane >> 4), q = (lane & 15) ^ ((ol >> 1) & 15);
        zoff[j] = (m < NZ && o_tile0 + ol < g.O) ? (unsigned)((o_tile0 + ol) * HoWo + q) * 4u : KAN_OOB;
        zq[j] = q;
    }
    const bool same_in = (KIND != KAN_BASIS_RBF && KIND != KAN_BASIS_POLY) || (x == xn);
    const kan_rsrc x_rs = make_rsrc(x, x_bytes), xn_rs = make_rsrc(same_in ? x : xn, x_bytes), dz_rs = make_rsrc(dz, dz_bytes);

    // expansion units (fixed): unit = channel-of-group * cells + cell -> (virtual row v, halo column jj)
    int u_vj[SLOTS], u_dst[SLOTS]; unsigned u_ok = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int u = tid + k * NT;
        const int ch = fastdiv(u, tb.divCells), cell = u - ch * tb.cells;
        const int v = fastdiv(cell, tb.divHC), jj = cell - v * tb.HC;
        u_vj[k] = v | (jj << 16);
        u_dst[k] = (cell * NPS + ch * P) | (ch << 24);
        u_ok |= ((ch < NG) ? 1u : 0u) << k;                  // the unit exists: written on every tile (zeros outside the image / past the last channel)
    }

    f32x16 acc[NI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int pt0 = (int)blockIdx.z * ptiles_per_split, pt1 = min(tb.n_ptiles, pt0 + ptiles_per_split);
    int last_ti = -1;
    for (int pt = pt0; pt < pt1; ++pt) {
        const int b = fastdiv(pt, tb.divTPI), ti = pt - b * tb.TPI;
        const int p0 = ti * TPX, npx = min(TPX, HoWo - p0);
        const int ho0 = fastdiv(p0, tb.divWo);
        __syncthreads();                                     // previous tile's reads of sH / sCell / sZ are done
        // ---- this tile's first dz step, its pixel -> cell table, its halo
        auto issue_dz = [&](int st, int zb) {
            const int valid = npx - st * KPX;                // pixels of this step inside the image (uniform)
            const int soff = __builtin_amdgcn_readfirstlane((b * (int)g.ybs + p0 + st * KPX) * 4);
            float* dst = sZ + zb * ZB;
#pragma unroll
            for (int j = 0; j < NZW; ++j) {
                const int m = j * WR + wave;                 // wave-uniform
                if (m < NZ) {
                    const unsigned vo = (valid >= KPX || zq[j] < valid) ? zoff[j] : KAN_OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(dz_rs, (__attribute__((address_space(3))) void*)(dst + m * 64), 4, (int)vo, soff, 0, 0);
                }
            }
        };
        issue_dz(0, 0);
        if (ti != last_ti) {
            if (tid < TPX) {
                const int p = p0 + min(tid, npx - 1), ho = fastdiv(p, tb.divWo), wo = p - ho * g.Wo;
                sCell[tid] = ((ho - ho0) * tb.HC + wo) * (NPS * 4);
            }
            last_ti = ti;
        }
        {
            float xa[SLOTS], xb[SLOTS]; unsigned inb_mask = 0;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                const int v = u_vj[k] & 0xffff, jj = u_vj[k] >> 16, ch = u_dst[k] >> 24;
                const int row = g.sh * (ho0 + v + tb.OR0) + pa, col = g.sw * (jj + tb.OC0) + pb;
                const bool inb = ((u_ok >> k) & 1u) && cbase + ch < g.C && (unsigned)row < (unsigned)g.H && (unsigned)col < (unsigned)g.W;
                const unsigned off = inb ? (unsigned)(b * (int)g.xbs + (cbase + ch) * HW + row * g.W + col) * 4u : KAN_OOB;
                xa[k] = buf_load(x_rs, off);
                xb[k] = same_in ? xa[k] : buf_load(xn_rs, off);
                inb_mask |= (inb ? 1u : 0u) << k;
            }
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                if ((u_ok >> k) & 1u)
                    stage_unit<KIND, FAST>(bs, sTab, (inb_mask >> k) & 1u, xa[k], xb[k], sH + (u_dst[k] & 0xffffff), 1, sDump + tid, cbase + (u_dst[k] >> 24));
        }
        const int n_st = (npx + KPX - 1) / KPX;
#pragma unroll
        for (int st = 0; st < TPX / KPX; ++st) {
            if (st < n_st) {                                 // uniform
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this step's dz landed (own DMAs)
                __syncthreads();                             // ... everybody's; (st == 0) halo + cell table visible; previous step's reads done
                if (st + 1 < n_st) issue_dz(st + 1, (st + 1) & 1);
                // cell offsets of the step's 8 pixel pairs -> A addresses
                int co[KPX / 2];
#pragma unroll
                for (int kk = 0; kk < KPX / 2; ++kk)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(co[kk]) : "v"(cellbase), "n"((st * KPX + 2 * kk) * 4) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(co[0]), "+v"(co[1]), "+v"(co[2]), "+v"(co[3]), "+v"(co[4]), "+v"(co[5]), "+v"(co[6]), "+v"(co[7]) :: "memory");
                float fa[2], fb[2][NI];
#define BANDW_READ(n_, kk_)                                                                                                  \
                do {                                                                                                         \
                    const unsigned aa_ = aRow + (unsigned)co[kk_];                                                           \
                    asm volatile("ds_read_b32 %0, %1" : "=v"(fa[n_]) : "v"(aa_) : "memory");                                 \
                    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                        \
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[n_][ni]) : "v"(bB[kk_]), "n"(((st & 1) * ZB + ni * 32 * KPX) * 4) : "memory"); \
                } while (0)
                BANDW_READ(0, 0);
#pragma unroll
                for (int kk = 0; kk < KPX / 2; ++kk) {
                    const int c_ = kk & 1, n_ = c_ ^ 1;
                    if (kk + 1 < KPX / 2) {
                        BANDW_READ(n_, kk + 1);
                        if constexpr (NI == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[c_]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                        else if constexpr (NI == 4) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[c_]), "+v"(fb[c_][0]), "+v"(fb[c_][1]), "+v"(fb[c_][2]), "+v"(fb[c_][3]) :: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fa[c_]), "+v"(fb[c_][0]), "+v"(fb[c_][1]), "+v"(fb[c_][2]), "+v"(fb[c_][3]), "+v"(fb[c_][4]), "+v"(fb[c_][NI - 1]) :: "memory");
                    } else {
                        if constexpr (NI == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c_]), "+v"(fb[c_][0]), "+v"(fb[c_][1]) :: "memory");
                        else if constexpr (NI == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c_]), "+v"(fb[c_][0]), "+v"(fb[c_][1]), "+v"(fb[c_][2]), "+v"(fb[c_][3]) :: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c_]), "+v"(fb[c_][0]), "+v"(fb[c_][1]), "+v"(fb[c_][2]), "+v"(fb[c_][3]), "+v"(fb[c_][4]), "+v"(fb[c_][NI - 1]) :: "memory");
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[ni] = MFMA32(fa[c_], fb[c_][ni], acc[ni]);
                }
#undef BANDW_READ
            }
        }
    }

    // ---- store: lane = output (column), registers = rows
    float* out = dwp + (size_t)blockIdx.z * slab_elems;
    const int rbase = rt * TR + wave * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = rbase + mfma_row(r, lane);
        if (rl >= rows) continue;
        float* orow = out + (size_t)(row0 - rt * TR + rl) * Opad + o_tile0 + (lane & 31);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) orow[ni * 32] = acc[ni][r];
    }
}

int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
int posmod(int a, int b) { return a - floordiv(a, b) * b; }

double band_slab_cost(double slab_bytes) { const double bw = slab_bytes / 4.0e6; return bw > 1.0 / 6.0 ? bw : 1.0 / 6.0; }

// Instantiated (compile-time spec, channels per group, tile) combinations.
bool band_has_kernel(int fast, int NG, int WO, int MO, int WP) {
    const bool tile = WO == 2 && ((WP == 2 && MO >= 1 && MO <= 3) || (WP == 4 && MO == 1));
    if (!tile || NG < 1 || NG > 3) return false;
    return fast == 1 || fast == 2 || fast == 3 || fast == 4 || fast == 5 || fast == 6;
}

}  // namespace

static void band_cfg_pass(const KanGeom* g, const KanBasis* b, int fast, KanBandCfg* c, bool wide_pixels);

// The halo tile is what a pixel tile costs: its cells are expanded once per (phase, channel group), whatever the tile's pixel count.  64-output
// layers whose halo is large next to the tile (strided or wide kernels: 3 -> 64 k11 s4 expands 580 cells per 128 pixels -- 3.2 vector
// instructions per MFMA, measured) take 256-pixel tiles on eight waves where that brings the cells per pixel down by a quarter or more.
void kan_band_cfg(const KanGeom* g, const KanBasis* b, int fast, KanBandCfg* c) {
    band_cfg_pass(g, b, fast, c, false);
    if (round_up(g->O, 64) % 128 != 0 && round_up(g->O, 64) != 192) {          // 64-output tiles only
        KanBandCfg w;
        band_cfg_pass(g, b, fast, &w, true);
        // (a third fewer cells per pixel at least: 3 -> 64 k11 s4 0.86 -> 0.80 ms; at 30 % fewer -- 3 -> 64 3x3 on 30x30 -- the eight-wave tile's
        //  66 KB of LDS cost more than the expansion saved: 0.113 -> 0.120 ms)
        if (w.ok && (!c->ok || 3LL * w.cells * c->TP <= 2LL * c->cells * w.TP)) *c = w;
    }
}

static void band_cfg_pass(const KanGeom* g, const KanBasis* b, int fast, KanBandCfg* c, bool wide_pixels) {
    memset(c, 0, sizeof(*c));
    if (!fast) return;
    const int T = g->kh * g->kw;
    if (T > KAN_BAND_MAX_TAPS || T > 255 || g->sh * g->sw > KAN_BAND_MAX_PHASES || g->sh > 255 || g->sw > 255) return;
    if (g->H > 16000 || g->W > 16000) return;                       // (sub-row / sub-column indices ride in 16 bits)
    c->fast = fast; c->P = fast_planes(fast);
    c->NG = g->C <= 3 ? g->C : 2;
    const int NPL = c->NG * c->P;
    c->NPLE = NPL + (NPL & 1); c->NPS = c->NPLE + 1;
    c->NGR = ceil_div(g->C, c->NG);
    const int Opad = round_up(g->O, 64);
    c->WO = 2; c->WP = 2;                                           // four waves: 2 along the outputs x 2 along the 128 pixels
    if (Opad % 128 == 0) c->MO = 2;                                  // 128 outputs: wave tile 64 x 64
    else if (Opad == 192) c->MO = 3;                                 // 192 outputs: wave tile 96 x 64 (six waves of 64 x 64 left two SIMDs with twice the
                                                                     // work of the others and the rest parked at the step barrier: 43 % of wave time, PMC)
    else c->MO = 1;                                                  // 64 outputs: wave tile 32 x 64
    if (wide_pixels) c->WP = 4;                                      // second pass (below): 256-pixel tiles, eight waves
    if (!band_has_kernel(fast, c->NG, c->WO, c->MO, c->WP)) return;
    c->TO = c->WO * c->MO * 32; c->TP = c->WP * 64; c->NT = c->WO * c->WP * 64;
    c->tiles_o = Opad / c->TO;
    const long long Mtot = (long long)g->B * g->Ho * g->Wo;
    c->tiles_p = ceil_div(Mtot, c->TP);
    // ---- taps by phase (band_emul.py: band_tables)
    int a_r[256], off_r[256], b_t[256], off_t[256];
    int OR0 = 1 << 30, OR1 = -(1 << 30), OC0 = 1 << 30, OC1 = -(1 << 30);
    for (int r = 0; r < g->kh; ++r) {
        const int e = g->dh * r - g->ph;
        a_r[r] = posmod(e, g->sh); off_r[r] = floordiv(e, g->sh);
        OR0 = off_r[r] < OR0 ? off_r[r] : OR0; OR1 = off_r[r] > OR1 ? off_r[r] : OR1;
    }
    for (int t = 0; t < g->kw; ++t) {
        const int e = g->dw * t - g->pw;
        b_t[t] = posmod(e, g->sw); off_t[t] = floordiv(e, g->sw);
        OC0 = off_t[t] < OC0 ? off_t[t] : OC0; OC1 = off_t[t] > OC1 ? off_t[t] : OC1;
    }
    c->OR0 = OR0; c->OC0 = OC0; c->span_r = OR1 - OR0;
    c->HC = g->Wo + (OC1 - OC0);
    if ((long long)c->span_r * c->HC + (OC1 - OC0) >= 65536) return;
    int n = 0;
    c->n_phase = 0;
    for (int pa = 0; pa < g->sh; ++pa)
        for (int pb = 0; pb < g->sw; ++pb) {
            const int first = n;
            for (int r = 0; r < g->kh; ++r)
                for (int t = 0; t < g->kw; ++t)
                    if (a_r[r] == pa && b_t[t] == pb) {
                        c->tap_shift[n] = (unsigned short)((off_r[r] - OR0) * c->HC + (off_t[t] - OC0));
                        c->tap_rt[n] = (unsigned short)((r << 8) | t);
                        ++n;
                    }
            if (n == first) continue;                               // no tap lands on this phase (e.g. stride > kernel)
            c->ph_a[c->n_phase] = (unsigned char)pa; c->ph_b[c->n_phase] = (unsigned char)pb;
            c->ph_tap0[c->n_phase] = (short)first;
            ++c->n_phase;
        }
    c->ph_tap0[c->n_phase] = (short)n;
    c->n_taps = n;                                                  // == T
    c->n_steps = c->NGR * n;
    for (int ph = 0; ph < c->n_phase; ++ph) {
        const int t0 = c->ph_tap0[ph], nt = c->ph_tap0[ph + 1] - t0;
        for (int j = 0; j < nt; ++j) {
            const int r = c->tap_rt[t0 + j] >> 8, t = c->tap_rt[t0 + j] & 0xff;
            c->tap_step[r * g->kw + t] = (short)(c->NGR * t0 + j);
            c->tap_nt[r * g->kw + t] = (short)nt;
        }
    }
    if ((long long)c->NGR * n >= 32000) return;                     // (tap_step is 16 bits)
    // ---- halo cells of the largest tile: virtual rows = rows touched + span_r per touched image (band_emul.py: tile_layout)
    const int HoWo = g->Ho * g->Wo;
    int vr_max = 0;
    const long long scan = c->tiles_p < (long long)HoWo + 1 ? c->tiles_p : (long long)HoWo + 1;      // the pattern repeats with p0 mod Ho*Wo
    for (long long i = 0; i < scan; ++i) {
        const long long p0 = i * c->TP, p1 = (p0 + c->TP < Mtot ? p0 + c->TP : Mtot) - 1;
        const int bb0 = (int)(p0 / HoWo), h0 = (int)(p0 % HoWo) / g->Wo, bb1 = (int)(p1 / HoWo), h1 = (int)(p1 % HoWo) / g->Wo;
        const int n0 = (bb0 == bb1 ? h1 : g->Ho - 1) - h0 + 1, nimg = bb1 - bb0 + 1;
        int vr = n0 + c->span_r;
        if (nimg > 2) vr += (nimg - 2) * (g->Ho + c->span_r);
        if (nimg > 1) vr += h1 + 1 + c->span_r;
        vr_max = vr > vr_max ? vr : vr_max;
    }
    c->cells = vr_max * c->HC;
    c->slots = ceil_div((long long)c->NG * c->cells, c->NT);
    if (c->slots > 6) return;                                       // (one instantiation holds up to six units per thread; unused ones are skipped at run time)
    c->TS = 1;      // taps per barrier step.  Two were measured on every target shape (3 -> 64 3x3: 0.083 -> 0.093 ms; 3 -> 64 k11 s4: 0.86 -> 0.94;
                    // 64 -> 192 k5: 3.22 -> 3.24): the barrier is not what bounds these steps, and the doubled weight buffer costs a workgroup per CU
    const int WBUF = ceil_div(c->TS * c->NPLE * c->TO, 256) * 256;
    c->lds_bytes = (2 * WBUF + KAN_MAX_TABLE + KAN_BAND_MAX_TAPS + KAN_BAND_MAX_PHASES + c->NT + c->cells * c->NPS + c->NPS) * 4;
    if (c->lds_bytes > 80 * 1024) return;                           // two workgroups per CU at least (above 64 KB: hipFuncSetAttribute at the launch)
    int wgs = 160 * 1024 / c->lds_bytes;
    const int by_threads = 2048 / c->NT, by_regs = c->MO == 3 ? 3 : c->NT <= 256 ? 4 : 2;
    wgs = wgs < by_threads ? wgs : by_threads; wgs = wgs < by_regs ? wgs : by_regs;
    c->wgs_per_cu = wgs < 1 ? 1 : wgs;
    // ---- split-K over the (phase, group) list: the round model of pick_splits (kanconv.hip)
    const int NGI = c->n_phase * c->NGR;
    const long long tiles = (long long)c->tiles_p * c->tiles_o * ngroups(g), SLOTS = 256ll * c->wgs_per_cu;
    const double steps_per_group = (double)c->n_steps / NGI, slab_bytes = 4.0 * g->B * g->O * g->Ho * g->Wo * ngroups(g);
    int best = 1; double best_cost = -1;
    for (int sp = 1; sp <= NGI && sp <= 64; ++sp) {
        const int gps = ceil_div(NGI, sp);
        if (ceil_div(NGI, gps) != sp || gps * steps_per_group < 8.0) continue;
        const long long rounds = (tiles * sp + SLOTS - 1) / SLOTS;
        const double cost = (double)rounds * (gps * steps_per_group + 6.0) + sp * band_slab_cost(slab_bytes);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sp; }
    }
    c->fwd_splits = best;
    c->ok = 1;
    // ---- weight gradient: per-image pixel tiles of 128, row tiles inside a (phase, group)
    c->bw_NI = Opad % 128 == 0 ? 4 : Opad == 192 ? 6 : 2;
    int rows_max = 0;
    for (int ph = 0; ph < c->n_phase; ++ph) { const int r = (c->ph_tap0[ph + 1] - c->ph_tap0[ph]) * c->NPLE; rows_max = r > rows_max ? r : rows_max; }
    c->bw_WR = (c->bw_NI == 6 || rows_max <= 128) ? 4 : 8;
    if (c->bw_NI == 4 && c->bw_WR == 8) c->bw_WR = 4;                // (instantiated: (4,2) (8,2) (4,4) (4,6))
    c->bw_NT = c->bw_WR * 64;
    const int TR = c->bw_WR * 32, TO = c->bw_NI * 32;
    c->bw_tiles_o = Opad / TO;
    int rt = 0;
    for (int ph = 0; ph < c->n_phase; ++ph) {
        c->bw_ph_rt0[ph] = (unsigned short)rt;
        rt += c->NGR * ceil_div((c->ph_tap0[ph + 1] - c->ph_tap0[ph]) * c->NPLE, TR);
    }
    c->bw_ph_rt0[c->n_phase] = (unsigned short)rt;
    c->bw_row_tiles = rt;
    c->bw_TPI = ceil_div(HoWo, 128);
    c->bw_ptiles = g->B * c->bw_TPI;
    int vrb = 0;
    for (int ti = 0; ti < c->bw_TPI; ++ti) {
        const int p0 = ti * 128, p1 = (p0 + 128 < HoWo ? p0 + 128 : HoWo) - 1;
        const int v = p1 / g->Wo - p0 / g->Wo + 1 + c->span_r;
        vrb = v > vrb ? v : vrb;
    }
    c->bw_cells = vrb * c->HC;
    const int need = ceil_div((long long)c->NG * c->bw_cells, c->bw_NT);
    c->bw_slots = need <= 3 ? 3 : 6;
    c->bw_lds_bytes = (2 * 16 * TO + KAN_MAX_TABLE + KAN_BAND_MAX_TAPS + KAN_BAND_MAX_PHASES + 128 + c->bw_NT + c->bw_cells * c->NPS + c->NPS) * 4;
    // rows of a (phase, group) fill row tiles of 32 WR rows; phases of a strided layer hold few taps each (3 -> 64 k11 s4: 144 / 96 / 64 rows),
    // and a half-empty tile is half-wasted matrix work: below 3/4 filled the tap-major kernel keeps the layer (measured there: 1.0 vs 1.8 ms)
    // (small launches -- under 2 GFLOP -- are latency-bound either way and keep the band kernel: one code path for a layer's forward and gradient)
    const double dense_flops = 2.0 * g->B * g->O * g->Ho * g->Wo * (double)g->C * c->P * T * ngroups(g);
    const bool bw_kernel = (fast >= 1 && fast <= 4) && (dense_flops < 2.0e9 || 4LL * c->n_steps * c->NPLE >= 3LL * rt * TR);
    if (bw_kernel && need <= 6 && c->bw_lds_bytes <= 80 * 1024 && rt < 65535 && (long long)c->bw_tiles_o * ngroups(g) <= 65535) {
        int wg = 160 * 1024 / c->bw_lds_bytes;
        const int by_thr = 2048 / c->bw_NT, by_reg = c->bw_NI == 6 ? (c->bw_NT <= 256 ? 3 : 1) : (c->bw_NT <= 256 ? 4 : 2);
        wg = wg < by_thr ? wg : by_thr; wg = wg < by_reg ? wg : by_reg; wg = wg < 1 ? 1 : wg;
        const long long tiles_w = (long long)rt * c->bw_tiles_o * ngroups(g), SL = 256ll * wg;
        const double slab_w = 4.0 * c->n_steps * c->NPLE * Opad * ngroups(g);
        int bestw = 1; double bestc = -1;
        for (int sp = 1; sp <= c->bw_ptiles && sp <= 1024; ++sp) {       // one pixel tile = 8 steps + its halo expansion (~4 step-equivalents)
            const int pps = ceil_div(c->bw_ptiles, sp);
            if (ceil_div(c->bw_ptiles, pps) != sp) continue;
            const long long rounds = (tiles_w * sp + SL - 1) / SL;
            const double cost = (double)rounds * (pps * 12.0 + 6.0) + sp * band_slab_cost(slab_w);
            if (bestc < 0 || cost < bestc) { bestc = cost; bestw = sp; }
        }
        c->bw_splits = bestw;
        c->bw_ok = 1;
    }
}

int kan_band_fwd_launch(const float* x, const float* xn, const float* wp, float* z, const KanGeom* g, const KanBasis* b,
                        const KanBandCfg* c, int splits, long long slab_elems, void* stream) {
    if (!c->ok) return kan_fail_msg("internal: band forward launched without a valid configuration%s", "");
    DevGeom dg = dev_geom(g); DevBasis db = dev_basis(b);
    BandTab tb;
    memset(&tb, 0, sizeof(tb));
    tb.n_phase = c->n_phase; tb.NGR = c->NGR; tb.HC = c->HC; tb.span_r = c->span_r; tb.OR0 = c->OR0; tb.OC0 = c->OC0; tb.cells = c->cells;
    tb.n_steps = c->n_steps; tb.nslots = c->slots;
    tb.divHC = make_fastdiv(c->HC); tb.divBlk = make_fastdiv(g->Ho + c->span_r); tb.divCells = make_fastdiv(c->cells); tb.divNGR = make_fastdiv(c->NGR);
    tb.divHoWo = make_fastdiv(g->Ho * g->Wo); tb.divWo = make_fastdiv(g->Wo);
    for (int ph = 0; ph < c->n_phase; ++ph)
        tb.ph_pack[ph] = (unsigned)c->ph_a[ph] | ((unsigned)c->ph_b[ph] << 8) | ((unsigned)c->ph_tap0[ph] << 16) |
                         ((unsigned)(c->ph_tap0[ph + 1] - c->ph_tap0[ph]) << 24);
    for (int i = 0; i < c->n_taps; ++i) tb.tap_shift[i] = c->tap_shift[i];
    if (splits < 1) return kan_fail_msg("internal: band forward needs at least one slab%s", "");
    const int Opad = round_up(g->O, 64), NGI = c->n_phase * c->NGR, gps = ceil_div(NGI, splits);        // grid.z = the PLAN's slab count (what the caller allocated)
    if ((long long)c->tiles_o * ngroups(g) > 65535) return kan_fail_msg("groups * output tiles exceed the grid limit%s", "");
    const dim3 grid(c->tiles_p, c->tiles_o * ngroups(g), splits);
    const unsigned x_bytes = (unsigned)((long long)g->B * g->x_bstride * 4);
    hipStream_t st = (hipStream_t)stream;
#define BAND_LAUNCH(KIND, F, NGV, MOV, WPV, SL)                                                                                         \
    do {                                                                                                                                \
        static int lds_raised = 0;      /* one-time kernel attribute setup: dynamic LDS above the 64 KB default */                         \
        if (c->lds_bytes > 64 * 1024 && lds_raised < c->lds_bytes) {                                                                        \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_fwd<KIND, F, NGV, 2, MOV, WPV, SL, 1>),                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)                                   \
                return kan_fail_msg("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed%s", "");                                        \
            lds_raised = 80 * 1024;                                                                                                         \
        }                                                                                                                                   \
        hipLaunchKernelGGL((k_band_fwd<KIND, F, NGV, 2, MOV, WPV, SL, 1>), grid, dim3(2 * WPV * 64), (size_t)c->lds_bytes, st, x, xn, wp, z, \
                           dg, db, tb, Opad, gps, slab_elems, x_bytes, c->tiles_o);                                                         \
    } while (0)
#define BAND_SLOTS(KIND, F, NGV, MOV, WPV) BAND_LAUNCH(KIND, F, NGV, MOV, WPV, 6)
#define BAND_TILE(KIND, F, NGV)                                                                \
    do {                                                                                       \
        if (c->MO == 1 && c->WP == 4) BAND_SLOTS(KIND, F, NGV, 1, 4);                           \
        else if (c->MO == 1) BAND_SLOTS(KIND, F, NGV, 1, 2);                                    \
        else if (c->MO == 2) BAND_SLOTS(KIND, F, NGV, 2, 2);                                    \
        else BAND_SLOTS(KIND, F, NGV, 3, 2);                                                    \
    } while (0)
#define BAND_NG(KIND, F)                                                                       \
    do {                                                                                       \
        if (c->NG == 1) BAND_TILE(KIND, F, 1);                                                  \
        else if (c->NG == 2) BAND_TILE(KIND, F, 2);                                             \
        else BAND_TILE(KIND, F, 3);                                                             \
    } while (0)
    switch (c->fast) {
        case 1: BAND_NG(KAN_BASIS_BSPLINE, 1); break;
        case 2: BAND_NG(KAN_BASIS_BSPLINE, 2); break;
        case 3: BAND_NG(KAN_BASIS_RBF, 3); break;
        case 4: BAND_NG(KAN_BASIS_CHEBY, 4); break;
        case 5: BAND_NG(KAN_BASIS_CHEBY, 5); break;
        case 6: BAND_NG(KAN_BASIS_POLY, 6); break;
        default: return kan_fail_msg("internal: no band forward kernel for this basis%s", "");
    }
#undef BAND_NG
#undef BAND_TILE
#undef BAND_SLOTS
#undef BAND_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { kan_fail_msg("launch failed: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

int kan_band_bwd_weight_launch(const float* dz, const float* x, const float* xn, float* dwp, const KanGeom* g, const KanBasis* b,
                               const KanBandCfg* c, int splits, long long slab_elems, void* stream) {
    if (!c->ok || !c->bw_ok) return kan_fail_msg("internal: band weight gradient launched without a valid configuration%s", "");
    DevGeom dg = dev_geom(g); DevBasis db = dev_basis(b);
    BandWTab tb;
    memset(&tb, 0, sizeof(tb));
    tb.n_phase = c->n_phase; tb.NGR = c->NGR; tb.HC = c->HC; tb.span_r = c->span_r; tb.OR0 = c->OR0; tb.OC0 = c->OC0; tb.cells = c->bw_cells;
    tb.n_steps = c->n_steps; tb.TPI = c->bw_TPI; tb.n_ptiles = c->bw_ptiles;
    tb.divHC = make_fastdiv(c->HC); tb.divCells = make_fastdiv(c->bw_cells); tb.divTPI = make_fastdiv(c->bw_TPI); tb.divWo = make_fastdiv(g->Wo);
    for (int ph = 0; ph < c->n_phase; ++ph)
        tb.ph_pack[ph] = (unsigned)c->ph_a[ph] | ((unsigned)c->ph_b[ph] << 8) | ((unsigned)c->ph_tap0[ph] << 16) |
                         ((unsigned)(c->ph_tap0[ph + 1] - c->ph_tap0[ph]) << 24);
    for (int ph = 0; ph <= c->n_phase; ++ph) tb.ph_rt0[ph] = c->bw_ph_rt0[ph];
    for (int i = 0; i < c->n_taps; ++i) tb.tap_shift[i] = c->tap_shift[i];
    if (splits < 1) return kan_fail_msg("internal: band weight gradient needs at least one slab%s", "");
    const int Opad = round_up(g->O, 64), Kpad = c->n_steps * c->NPLE, pps = ceil_div(c->bw_ptiles, splits);   // grid.z = the PLAN's slab count
    const dim3 grid(c->bw_row_tiles, c->bw_tiles_o * ngroups(g), splits);
    const unsigned x_bytes = (unsigned)((long long)g->B * g->x_bstride * 4), dz_bytes = (unsigned)((long long)g->B * g->y_bstride * 4);
    hipStream_t st = (hipStream_t)stream;
#define BANDW_LAUNCH(KIND, F, NGV, WRV, NIV, SL)                                                                                                   \
    do {                                                                                                                                           \
        static int lds_raised = 0;                                                                                                                 \
        if (c->bw_lds_bytes > 64 * 1024 && lds_raised < c->bw_lds_bytes) {                                                                         \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_bwd_weight<KIND, F, NGV, WRV, NIV, SL>),                                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)                                          \
                return kan_fail_msg("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed%s", "");                                               \
            lds_raised = 80 * 1024;                                                                                                                \
        }                                                                                                                                          \
        hipLaunchKernelGGL((k_band_bwd_weight<KIND, F, NGV, WRV, NIV, SL>), grid, dim3(WRV * 64), (size_t)c->bw_lds_bytes, st, dz, x, xn, dwp, dg, db, \
                           tb, Kpad, Opad, pps, slab_elems, x_bytes, dz_bytes, c->bw_tiles_o);                                                    \
    } while (0)
#define BANDW_SLOTS(KIND, F, NGV, WRV, NIV) do { if (c->bw_slots <= 3) BANDW_LAUNCH(KIND, F, NGV, WRV, NIV, 3); else BANDW_LAUNCH(KIND, F, NGV, WRV, NIV, 6); } while (0)
#define BANDW_TILE(KIND, F, NGV)                                                                \
    do {                                                                                        \
        if (c->bw_NI == 2 && c->bw_WR == 4) BANDW_SLOTS(KIND, F, NGV, 4, 2);                     \
        else if (c->bw_NI == 2) BANDW_SLOTS(KIND, F, NGV, 8, 2);                                 \
        else if (c->bw_NI == 4) BANDW_SLOTS(KIND, F, NGV, 4, 4);                                 \
        else BANDW_SLOTS(KIND, F, NGV, 4, 6);                                                    \
    } while (0)
#define BANDW_NG(KIND, F)                                                                       \
    do {                                                                                        \
        if (c->NG == 1) BANDW_TILE(KIND, F, 1);                                                  \
        else if (c->NG == 2) BANDW_TILE(KIND, F, 2);                                             \
        else BANDW_TILE(KIND, F, 3);                                                             \
    } while (0)
    switch (c->fast) {
        case 1: BANDW_NG(KAN_BASIS_BSPLINE, 1); break;
        case 2: BANDW_NG(KAN_BASIS_BSPLINE, 2); break;
        case 3: BANDW_NG(KAN_BASIS_RBF, 3); break;
        case 4: BANDW_NG(KAN_BASIS_CHEBY, 4); break;
        default: return kan_fail_msg("internal: no band weight-gradient kernel for this basis%s", "");
    }
#undef BANDW_NG
#undef BANDW_TILE
#undef BANDW_SLOTS
#undef BANDW_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { kan_fail_msg("launch failed: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
