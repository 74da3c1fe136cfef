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
    int n_phase, NGR, HC, span_r, OR0, OC0, cells, n_steps;
    FastDiv divHC, divBlk, divCells, divNGR;                   // by HC, by Ho + span_r, by cells, by NGR
    unsigned ph_pack[KAN_BAND_MAX_PHASES];                     // a | b << 8 | first tap << 16 | taps << 24
    unsigned short tap_shift[KAN_BAND_MAX_TAPS];               // phase-ordered, in cells
};

// Workgroup: WO x WP waves, one (32 MO outputs) x (64 pixels) tile per wave (MO x 2 MFMA 32x32x2) => TO = 32 MO WO outputs x TP = 64 WP pixels.
// MO = 1 (64-output layers): 128-pixel tiles on four waves, so that the halo tile -- the LDS bill of this kernel -- covers 128 pixels, not 256.
// Dynamic LDS: [2 weight buffers | basis table | tap shifts | phase table | dump words | halo tile cells x NPS].
template <int KIND, int FAST, int NG, int WO, int MO, int WP, int SLOTS>
__global__ __launch_bounds__(WO * WP * 64, 2) void k_band_fwd(
    const float* __restrict__ x, const float* __restrict__ xn, const float* __restrict__ wp, float* __restrict__ z,
    DevGeom g, DevBasis bs, BandTab tb, int Opad, int groups_per_split, long long slab_elems, unsigned x_bytes, int tiles_o) {
    constexpr int P = fast_planes(FAST), NPL = NG * P, NPLE = NPL + (NPL & 1), NPS = NPLE + 1;
    constexpr int TO = WO * MO * 32, TP = WP * 64, NT = WO * WP * 64, NW = WO * WP;
    constexpr int WBUF = ((NPLE * TO + 255) / 256) * 256;       // floats per weight buffer: whole 1-KiB wave copies
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
    for (int i = tid; i < tb.cells * NPS + NPS; i += NT) sH[i] = 0.f;      // pad rows, pad words, out-of-image cells: zero for good

    // ---- the tile: TP consecutive output pixels from px_tile0, in (image, row, column) order (band_emul.py: tile_layout)
    const int p_last = min(px_tile0 + TP, Mtot) - 1;
    const int b0 = px_tile0 / HoWo, ho0 = (px_tile0 - b0 * HoWo) / g.Wo;
    const int b1 = p_last / HoWo, ho1 = (p_last - b1 * HoWo) / g.Wo;
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
            const int b = px / HoWo, r = px - b * HoWo, ho = r / g.Wo, wo = r - ho * g.Wo, k = b - b0;
            const int v = k == 0 ? ho - ho0 : blk0 + (k - 1) * blkN + ho;
            cell = v * tb.HC + wo;
        }
        vb[q] = lds_addr(sH) + (unsigned)(cell * NPS + kh2) * 4u;
    }

    // ---- expansion units of this thread (fixed per tile): unit = channel-of-group * cells + cell -> (image, sub-row i, sub-column j)
    int u_ij[SLOTS], u_base[SLOTS], u_dst[SLOTS]; unsigned u_ok = 0;       // u_ij = i (low 16, signed) | j << 16;  u_dst = LDS word | ch << 24
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int u = tid + k * NT;
        const int ch = fastdiv(u, tb.divCells), cell = u - ch * tb.cells;
        const int v = fastdiv(cell, tb.divHC), jj = cell - v * tb.HC;
        int kimg = 0, lv = v, first = ho0;
        if (v >= blk0) { const int w2 = v - blk0, q2 = fastdiv(w2, tb.divBlk); kimg = 1 + q2; lv = w2 - q2 * blkN; first = 0; }
        const int i = first + lv + tb.OR0, j = jj + tb.OC0, img = b0 + kimg;
        u_ij[k] = (i & 0xffff) | (j << 16);
        u_base[k] = img * (int)g.xbs + ch * HW;
        u_dst[k] = (cell * NPS + ch * P) | (ch << 24);
        u_ok |= ((ch < NG && img < g.B) ? 1u : 0u) << k;
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
            const int i = (int)(short)(u_ij[k] & 0xffff), j = u_ij[k] >> 16, ch = u_dst[k] >> 24;
            const int row = g.sh * i + pa, col = g.sw * j + pb;
            const bool inb = ((u_ok >> k) & 1u) && cbase + ch < g.C && (unsigned)row < (unsigned)g.H && (unsigned)col < (unsigned)g.W;
            const unsigned off = inb ? (unsigned)(u_base[k] + cbase * HW + row * g.W + col) * 4u : KAN_OOB;
            xa[k] = buf_load(x_rs, off);
            xb[k] = same_in ? xa[k] : buf_load(xn_rs, off);
            inb_mask |= (inb ? 1u : 0u) << k;
        }
    };
    auto expand = [&]() {                                    // write the P planes of every unit (zeros outside the image / past the last channel)
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if ((u_ok >> k) & 1u)
                stage_unit<KIND, FAST>(bs, sTab, (inb_mask >> k) & 1u, xa[k], xb[k], sH + (u_dst[k] & 0xffffff), 1, sDump + tid, s_cbase + (u_dst[k] >> 24));
    };

    // ---- weight steps: NPLE rows x TO floats, a straight 2-D copy by LDS-DMA (16 bytes per lane, 1 KiB per wave instruction)
    unsigned w_off[NQW]; unsigned w_okm = 0;
#pragma unroll
    for (int j = 0; j < NQW; ++j) {
        const int q = j * NW + wave, f = q * 256 + 4 * lane, row = f / TO, col = f - row * TO;
        w_off[j] = (unsigned)(row * Opad + col) * 4u;
        w_okm |= ((q < NQ && row < NPLE) ? 1u : 0u) << j;
    }
    auto issue_w = [&](int s, int buf) {
        const char* wsrc = (const char*)(wp + (size_t)s * NPLE * Opad + o_tile0);
        float* dW = sW + buf * WBUF;
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const int q = j * NW + wave;                     // wave-uniform
            if (q < NQ) {
                if ((w_okm >> j) & 1u) glds16((const float*)(wsrc + w_off[j]), dW + q * 256);
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
    if (gi0 < gi1) {
        const int ph = fastdiv(gi0, tb.divNGR), gg = gi0 - ph * tb.NGR;
        const unsigned pk = __builtin_amdgcn_readfirstlane(sPh[ph]);
        s = tb.NGR * (int)((pk >> 16) & 0xff) + gg * (int)(pk >> 24);       // first weight step of (phase, group)
        issue_w(s, 0);
        load_group(gi0);
    }
    const int ao = w_o * (MO * 32) + (lane & 31);
    for (int gi = gi0; gi < gi1; ++gi) {
        const int ph = fastdiv(gi, tb.divNGR);
        const unsigned pk = __builtin_amdgcn_readfirstlane(sPh[ph]);
        const int tap0 = (pk >> 16) & 0xff, nt = (int)(pk >> 24);
        __syncthreads();                                     // every wave has finished reading the previous group's halo
        expand();
#pragma unroll 1
        for (int j = 0; j < nt; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own weight DMA of this step landed (the compiler does not tie LDS-DMA to the barrier)
            __syncthreads();                                 // ... everybody's; (j == 0) halo visible; previous step's reads done
            if (j + 1 < nt || gi + 1 < gi1) issue_w(s + 1, buf ^ 1);
            if (j == (nt >> 1) && gi + 1 < gi1) load_group(gi + 1);
            const unsigned sh = (unsigned)__builtin_amdgcn_readfirstlane(sShift[tap0 + j]);
            const unsigned aw = lds_addr(sW + buf * WBUF + kh2 * TO + ao), ab0 = vb[0] + sh, ab1 = vb[1] + sh;
            if constexpr (MO == 2) {
                float fa[2][2], fb[2][2];
                BAND_READ4(fa[0][0], fa[0][1], fb[0][0], fb[0][1], aw, ab0, ab1, 0, 32 * 4, 0);
#pragma unroll
                for (int kk = 0; kk < NPLE / 2; ++kk) {
                    const int c_ = kk & 1, n_ = c_ ^ 1;
                    if (kk + 1 < NPLE / 2) {
                        BAND_READ4(fa[n_][0], fa[n_][1], fb[n_][0], fb[n_][1], aw, ab0, ab1, (2 * (kk + 1)) * TO * 4, (2 * (kk + 1)) * TO * 4 + 128,
                                   (2 * (kk + 1)) * 4);
                        LDS_WAIT4(fa[c_][0], fa[c_][1], fb[c_][0], fb[c_][1], 4);
                    } else {
                        LDS_WAIT4(fa[c_][0], fa[c_][1], fb[c_][0], fb[c_][1], 0);
                    }
                    acc[0][0] = MFMA32(fa[c_][0], fb[c_][0], acc[0][0]);
                    acc[0][1] = MFMA32(fa[c_][0], fb[c_][1], acc[0][1]);
                    acc[MO - 1][0] = MFMA32(fa[c_][1], fb[c_][0], acc[MO - 1][0]);
                    acc[MO - 1][1] = MFMA32(fa[c_][1], fb[c_][1], acc[MO - 1][1]);
                }
            } else {
                float fa[2], fb[2][2];
                BAND_READ3(fa[0], fb[0][0], fb[0][1], aw, ab0, ab1, 0, 0);
#pragma unroll
                for (int kk = 0; kk < NPLE / 2; ++kk) {
                    const int c_ = kk & 1, n_ = c_ ^ 1;
                    if (kk + 1 < NPLE / 2) {
                        BAND_READ3(fa[n_], fb[n_][0], fb[n_][1], aw, ab0, ab1, (2 * (kk + 1)) * TO * 4, (2 * (kk + 1)) * 4);
                        BAND_WAIT3(fa[c_], fb[c_][0], fb[c_][1], 3);
                    } else {
                        BAND_WAIT3(fa[c_], fb[c_][0], fb[c_][1], 0);
                    }
                    acc[0][0] = MFMA32(fa[c_], fb[c_][0], acc[0][0]);
                    acc[0][1] = MFMA32(fa[c_], fb[c_][1], acc[0][1]);
                }
            }
            buf ^= 1; ++s;
        }
    }

    // ---- store: column (lane) = pixel => coalesced along the plane
    float* zs = z + (size_t)blk.z * slab_elems;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int px = px_tile0 + w_p * 64 + ni * 32 + (lane & 31);
        if (px >= Mtot) continue;
        const int b = px / HoWo, hw = px - b * HoWo;
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

int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
int posmod(int a, int b) { return a - floordiv(a, b) * b; }

double band_slab_cost(double slab_bytes) { const double bw = slab_bytes / 4.0e6; return bw > 1.0 / 6.0 ? bw : 1.0 / 6.0; }

// Instantiated (compile-time spec, channels per group, tile) combinations.
bool band_has_kernel(int fast, int NG, int WO, int MO, int WP) {
    const bool tile = (WO == 2 && MO == 1 && WP == 2) || (WO == 2 && MO == 2 && WP == 2) || (WO == 3 && MO == 2 && WP == 2);
    if (!tile || NG < 1 || NG > 3) return false;
    return fast == 1 || fast == 2 || fast == 3 || fast == 4 || fast == 5 || fast == 6;
}

}  // namespace

void kan_band_cfg(const KanGeom* g, const KanBasis* b, int fast, KanBandCfg* c) {
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
    c->WP = 2;                                                      // 128-pixel tiles, 4 or 6 waves
    if (Opad % 128 == 0) { c->WO = 2; c->MO = 2; }                   // 128 outputs: wave tile 64 x 64
    else if (Opad == 192) { c->WO = 3; c->MO = 2; }                  // 192 outputs on six waves
    else { c->WO = 2; c->MO = 1; }                                   // 64 outputs: wave tile 32 x 64
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
    if (c->slots > 8) return;
    c->slots = c->slots <= 4 ? 4 : 8;
    const int WBUF = ceil_div(c->NPLE * c->TO, 256) * 256;
    c->lds_bytes = (2 * WBUF + KAN_MAX_TABLE + KAN_BAND_MAX_TAPS + KAN_BAND_MAX_PHASES + c->NT + c->cells * c->NPS + c->NPS) * 4;
    if (c->lds_bytes > 64 * 1024) return;
    int wgs = 160 * 1024 / c->lds_bytes;
    const int by_threads = 2048 / c->NT, by_regs = c->NT <= 256 ? 3 : 2;
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
}

int kan_band_fwd_launch(const float* x, const float* xn, const float* wp, float* z, const KanGeom* g, const KanBasis* b,
                        const KanBandCfg* c, long long slab_elems, void* stream) {
    if (!c->ok) return kan_fail_msg("internal: band forward launched without a valid configuration%s", "");
    DevGeom dg = dev_geom(g); DevBasis db = dev_basis(b);
    BandTab tb;
    memset(&tb, 0, sizeof(tb));
    tb.n_phase = c->n_phase; tb.NGR = c->NGR; tb.HC = c->HC; tb.span_r = c->span_r; tb.OR0 = c->OR0; tb.OC0 = c->OC0; tb.cells = c->cells;
    tb.n_steps = c->n_steps;
    tb.divHC = make_fastdiv(c->HC); tb.divBlk = make_fastdiv(g->Ho + c->span_r); tb.divCells = make_fastdiv(c->cells); tb.divNGR = make_fastdiv(c->NGR);
    for (int ph = 0; ph < c->n_phase; ++ph)
        tb.ph_pack[ph] = (unsigned)c->ph_a[ph] | ((unsigned)c->ph_b[ph] << 8) | ((unsigned)c->ph_tap0[ph] << 16) |
                         ((unsigned)(c->ph_tap0[ph + 1] - c->ph_tap0[ph]) << 24);
    for (int i = 0; i < c->n_taps; ++i) tb.tap_shift[i] = c->tap_shift[i];
    const int Opad = round_up(g->O, 64), NGI = c->n_phase * c->NGR, gps = ceil_div(NGI, c->fwd_splits);
    if ((long long)c->tiles_o * ngroups(g) > 65535) return kan_fail_msg("groups * output tiles exceed the grid limit%s", "");
    const dim3 grid(c->tiles_p, c->tiles_o * ngroups(g), c->fwd_splits);
    const unsigned x_bytes = (unsigned)((long long)g->B * g->x_bstride * 4);
    hipStream_t st = (hipStream_t)stream;
#define BAND_LAUNCH(KIND, F, NGV, WOV, MOV, SL)                                                                                         \
    hipLaunchKernelGGL((k_band_fwd<KIND, F, NGV, WOV, MOV, 2, SL>), grid, dim3(WOV * 2 * 64), (size_t)c->lds_bytes, st, x, xn, wp, z, dg, db, tb, \
                       Opad, gps, slab_elems, x_bytes, c->tiles_o)
#define BAND_SLOTS(KIND, F, NGV, WOV, MOV) do { if (c->slots <= 4) BAND_LAUNCH(KIND, F, NGV, WOV, MOV, 4); else BAND_LAUNCH(KIND, F, NGV, WOV, MOV, 8); } while (0)
#define BAND_TILE(KIND, F, NGV)                                                                \
    do {                                                                                       \
        if (c->WO == 2 && c->MO == 1) BAND_SLOTS(KIND, F, NGV, 2, 1);                           \
        else if (c->WO == 2) BAND_SLOTS(KIND, F, NGV, 2, 2);                                    \
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
