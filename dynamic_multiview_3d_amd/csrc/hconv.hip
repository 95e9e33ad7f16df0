// Halo-tile convolution kernel (img2feat / feat2img for the layers that carry the FLOPs).
//
// A workgroup owns a TH x TW tile (128 pixels) of the output's phase grid and ALL taps:
//   1. the input halo ((TH-1)*sa + taps) x ((TW-1)*sa + taps) x 32 channels is staged ONCE in LDS
//      (zero-filled outside the image) -- every input byte is read from L2/HBM ~1.3x instead of
//      once per filter tap (25x for 5x5), and the main loop needs no bounds checks;
//   2. the main loop has NO barriers: each wave multiplies its 32 pixels by the filter taps with
//      v_mfma_f32_32x32x2_f32, A operand = ds_read_b32 from the halo at (pixel + tap offset),
//      B operand = filter values prefetched one tap ahead straight from global memory (the
//      filters are <= 600 KB and L2-resident; all waves read the same lines);
//   3. for stride-2 transposed convs all four output phases are produced from the same halo
//      (4 accumulator sets per wave), so the 4 dense sub-convolutions share one staging pass.
// Pixel stride in LDS is 33 floats: lanes (= consecutive pixels) hit distinct banks.
// Channel order inside a 32-channel chunk is permuted (k-slot lh covers channels lh*16..lh*16+15)
// identically for A and B, which lets the N-major filters be fetched as 16-byte loads.
#include "conv_common.h"
#include <algorithm>
#include <stdlib.h>
#include <stdio.h>

namespace mv3d {

static inline bool kmajor_of(const IgemmParams& p) { return p.w_ns == 1; }


// One tap's operands in registers: A fragments (MT pixel groups x 16 k-steps) and B fragments
// (16 k-steps x NT channel groups).  Two sets ping-pong so that the set for tap t+1 is being filled
// (ds_read / global_load) while the MFMAs of tap t run; no register copies.
template <int MT, int NT>
struct Frags {
    float a[MT][16];
    float b[16][NT];
};

template <int NPH, int MT, int NT, bool KMAJOR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void hconv_kernel(const IgemmParams p, const HconvExtra x) {
    extern __shared__ __attribute__((aligned(16))) float halo[];
    constexpr int CS = 33;
    constexpr int NTHR = WAVES * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    int b = blockIdx.x;
    const int tw_i = b % x.tiles_w; b /= x.tiles_w;
    const int th_i = b % x.tiles_h;
    const int n = (b / x.tiles_h) * x.G;                 // first image of the tile (G whole images when G > 1)
    const int oh0 = th_i * x.TH, ow0 = tw_i * x.TW;
    const int n0 = blockIdx.y * 32 * NT;
    const int zks = (int)blockIdx.z % x.ksplit;            // chunk-split index; grid.z = phase * ksplit + zks

    int lane_base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pidx = (wave * MT + m) * 32 + li;
        const int g = pidx >> x.img_shift, pr = pidx & ((1 << x.img_shift) - 1);      // G == 1: img_shift covers the tile
        const int tr = pr >> x.tw_shift, tc = pr & (x.TW - 1);
        lane_base[m] = ((g * x.HRi + tr * p.sa_h) * x.HC + tc * p.sa_w) * CS + lh * 16;
    }

    f32x16 acc[NPH][MT][NT];
#pragma unroll
    for (int a = 0; a < NPH; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int y = 0; y < NT; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][m][y][r] = 0.f;

    Frags<MT, NT> f0, f1;
    // phase-split launches: this workgroup's taps are [tap_lo, tap_hi) of the flat list
    const int ph_z = x.phase_split ? (int)blockIdx.z / x.ksplit : 0;
    const int tap_lo = x.phase_split ? p.tap_begin[ph_z] : 0;
    const int ntaps_here = x.phase_split ? p.tap_begin[ph_z + 1] - tap_lo : x.ntaps_total;
    const int cc_begin = (x.chunks * zks) / x.ksplit, cc_end = (x.chunks * (zks + 1)) / x.ksplit;
    const int total_seq = (cc_end - cc_begin) * ntaps_here;

    // Filter fetch for one (chunk, tap): unconditional loads from clamped (always valid) addresses and
    // NO masking -- a conditional load makes hipcc branch around every load, and a select on the loaded
    // value makes it wait for the load right away, either of which kills the one-tap-ahead prefetch
    // (cdna_hip_programming.md, trap (c)).  Clamping is enough: output columns >= Cc are never
    // stored, and channels >= Ka multiply halo entries that were staged as zeros.
    auto load_b = [&](Frags<MT, NT>& f, int seq) {
        const int cq = seq / ntaps_here;
        const int cc = cc_begin + cq;
        const int t = tap_lo + seq - cq * ntaps_here;
        const float* wt = p.Wt + (int64_t)p.taps[t].widx * p.w_tap_stride;
        const int cbase = cc * 32 + lh * 16;
        const int cb = cbase < p.Ka ? cbase : 0;
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = n0 + y * 32 + li;
            const int colc = col < p.Cc ? col : p.Cc - 1;
            if constexpr (KMAJOR) {
                const float* src = wt + (int64_t)cb * p.w_ks + colc;
#pragma unroll
                for (int kp = 0; kp < 16; ++kp) {
                    f.b[kp][y] = *src;
                    src += p.w_ks;
                }
            } else {
                const float4* src = reinterpret_cast<const float4*>(wt + (int64_t)colc * p.w_ns + cb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = src[j];
                    f.b[4 * j][y] = v.x; f.b[4 * j + 1][y] = v.y; f.b[4 * j + 2][y] = v.z; f.b[4 * j + 3][y] = v.w;
                }
            }
        }
    };
    auto load_a = [&](Frags<MT, NT>& f, int t) {
        const IgemmTap tap = p.taps[t];
        const int off = ((tap.dh - x.dh_min) * x.HC + (tap.dw - x.dw_min)) * CS;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* ap = halo + lane_base[m] + off;
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) f.a[m][kp] = ap[kp];
        }
    };

    load_b(f1, 0);
    const int ih0 = oh0 * p.sa_h + x.dh_min, iw0 = ow0 * p.sa_w + x.dw_min;
    const int halo_pix = x.HR * x.HC;
    int seq = 0;
    for (int cc = cc_begin; cc < cc_end; ++cc) {
        if (cc > cc_begin) __syncthreads();
        // halo staging in batches of 8 independent 16-byte loads per thread (all in flight
        // together), then the LDS stores; out-of-image pixels load a valid dummy address and are zeroed
        for (int base = 0; base < halo_pix * 8; base += NTHR * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NTHR + tid;
                const int pix = idx >> 3, c4 = idx & 7;
                const int hrv = pix / x.HC, hc = pix - hrv * x.HC;
                const int g = hrv / x.HRi, hr = hrv - g * x.HRi;              // G == 1: g = 0
                const int ih = ih0 + hr, iw = iw0 + hc;
                const int ch = cc * 32 + c4 * 4;
                const bool ok = !(x.dbg & 1) && idx < halo_pix * 8 && n + g < p.N && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa && ch < p.Ka;
                const float* src = ok ? p.A + (int64_t)(((n + g) * p.Ha + ih) * p.Wa + iw) * p.a_ld + ch : p.A;
                const float4 t4 = *reinterpret_cast<const float4*>(src);
                v[u] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NTHR + tid;
                if (idx < halo_pix * 8) {
                    float* d = halo + (idx >> 3) * CS + (idx & 7) * 4;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
        __syncthreads();
        // f0 = operands of the current tap, f1 = operands of the next one (loads in flight while the
        // current tap's MFMAs run).  The hand-over f0 = f1 is register moves issued behind the last
        // MFMA; a branch-selected ping-pong was tried and makes hipcc bounce the accumulators between
        // AGPRs and VGPRs on every tap.
        load_a(f1, tap_lo);
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            const int tb = x.phase_split ? tap_lo : p.tap_begin[ph], te = (x.dbg & 8) ? tb : (x.phase_split ? tap_lo + ntaps_here : p.tap_begin[ph + 1]);
            for (int t = tb; t < te; ++t, ++seq) {
#pragma unroll
                for (int kp = 0; kp < 16; ++kp) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) f0.a[m][kp] = f1.a[m][kp];
#pragma unroll
                    for (int y = 0; y < NT; ++y) f0.b[kp][y] = f1.b[kp][y];
                }
                if (seq + 1 < total_seq && !(x.dbg & 2)) load_b(f1, seq + 1);
                if (t + 1 < tap_lo + ntaps_here && !(x.dbg & 4)) load_a(f1, t + 1);
#pragma unroll
                for (int kp = 0; kp < 16; ++kp)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int y = 0; y < NT; ++y)
                            acc[ph][m][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a[m][kp], f0.b[kp][y], acc[ph][m][y], 0, 0, 0);
            }
        }
    }

    const int Hp = p.Hp[0], Wp = p.Wp[0];
    const int64_t npix_total = (int64_t)p.N * p.Hc * p.Wc;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int phe = x.phase_split ? ph_z : ph;
        const int phh = phe / p.so_w, phw = phe % p.so_w;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int y = 0; y < NT; ++y) {
                const int col = n0 + y * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = (wave * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int g = q >> x.img_shift, qr = q & ((1 << x.img_shift) - 1);
                    const int ohp = oh0 + (qr >> x.tw_shift), owp = ow0 + (qr & (x.TW - 1));
                    if (n + g < p.N && ohp < Hp && owp < Wp && col < p.Cc) {
                        const int64_t pix = (int64_t)((n + g) * p.Hc + ohp * p.so_h + phh) * p.Wc + owp * p.so_w + phw;
                        if (x.ksplit > 1) p.Part[((int64_t)zks * npix_total + pix) * p.Cc + col] = acc[ph][m][y][r];
                        else p.Out[pix * p.c_ld + col] = epilogue_value(p, acc[ph][m][y][r], pix, col);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent variant for single-phase problems (conv fwd / stride-1 conv dgrad / deconv dgrad) with
// 5x5 or 3x3 filters -- the bulk of the FLOPs.  Same tile / halo / fragment scheme as hconv_kernel, plus:
//   * a workgroup walks a list of (tile, channel-chunk) units with TWO halo buffers in LDS; while it
//     multiplies unit u out of one buffer, unit u+1 streams into the other: every tap issues one
//     16-byte global load per thread and, two taps later, writes it to LDS (a 3-deep register ring,
//     the out-of-image mask is applied at the write, never at the load).  No s_waitcnt of the tap loop
//     has to cover a burst of older loads (vmcnt retires in order), nothing is serial at the unit
//     boundary except one barrier;
//   * the tap loop is fully unrolled (NTAPS is a template parameter): the two operand register sets
//     ping-pong by a compile-time index, there are no register copies and no per-tap branches.
template <int NTAPS, int MT, int NT, bool KMAJOR, int PF, int WAVES, int B1, int B2, int B3>
__global__ __launch_bounds__(WAVES * 64) void hconvp_kernel(const IgemmParams p, const HconvExtra x) {
    extern __shared__ __attribute__((aligned(16))) float halo_all[];
    constexpr int CS = 33;
    constexpr int NTHR = WAVES * 64;
    constexpr int NPH = (B1 >= NTAPS) ? 1 : 4;         // taps [0,B1) phase 0, [B1,B2) phase 1, [B2,B3) 2, [B3,NTAPS) 3
    constexpr int LPT = (PF + NTAPS - 3) / (NTAPS - 2);       // halo loads issued per tap (all committed by the last tap)
    constexpr int DLY = 2, RING = DLY + 1;
    static_assert(((PF + LPT - 1) / LPT) + DLY <= NTAPS, "halo prefetch does not fit in the tap loop");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * 32 * NT;
    const int tiles_per_img = x.tiles_h * x.tiles_w;
    const int total_tiles = p.N * tiles_per_img;
    const int halo_f4 = x.HR * x.HC * 8;
    const int buf_floats = x.HR * x.HC * CS;

    int lane_base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pidx = (wave * MT + m) * 32 + li;
        const int tr = pidx >> x.tw_shift, tc = pidx & (x.TW - 1);
        lane_base[m] = ((tr * p.sa_h) * x.HC + tc * p.sa_w) * CS + lh * 16;
    }

    f32x16 acc[NPH][MT][NT];
    Frags<MT, NT> f[2];
    float4 ring[RING][LPT];
    bool ring_ok[RING][LPT];

    // (tile, chunk) -> pointer of halo float4 slot `idx` (or null when outside the image / channel range)
    // halo slot j of this thread: (row, col, channel group) inside the halo, fixed for the whole kernel
    int slot_hr[PF], slot_hc[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int pix = (j * NTHR + tid) >> 3;
        slot_hr[j] = pix / x.HC;
        slot_hc[j] = pix - slot_hr[j] * x.HC;
    }
    const int c4 = (tid & 7) * 4;
    // per-unit origin (wave-uniform): image base row/col of the halo and the channel chunk
    struct Origin { int n, ih0, iw0, ch0; };
    auto origin_of = [&](int tile, int cc) {
        Origin o;
        o.n = tile / tiles_per_img;
        const int r = tile - o.n * tiles_per_img;
        const int th_i = r / x.tiles_w, tw_i = r - th_i * x.tiles_w;
        o.ih0 = th_i * x.TH * p.sa_h + x.dh_min;
        o.iw0 = tw_i * x.TW * p.sa_w + x.dw_min;
        o.ch0 = cc * 32;
        return o;
    };
    auto halo_src = [&](const Origin& o, int j) -> const float* {
        const int ih = o.ih0 + slot_hr[j], iw = o.iw0 + slot_hc[j], ch = o.ch0 + c4;
        const bool ok = (j * NTHR + tid) < halo_f4 && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa && ch < p.Ka;
        return ok ? p.A + (int64_t)((o.n * p.Ha + ih) * p.Wa + iw) * p.a_ld + ch : nullptr;
    };
    auto halo_store = [&](float* buf, int j, const float4& v, bool ok) {
        const int idx = j * NTHR + tid;
        if (idx < halo_f4) {
            float* d = buf + (idx >> 3) * CS + (idx & 7) * 4;
            d[0] = ok ? v.x : 0.f; d[1] = ok ? v.y : 0.f; d[2] = ok ? v.z : 0.f; d[3] = ok ? v.w : 0.f;
        }
    };
    auto load_b = [&](Frags<MT, NT>& fr, int cc, int t) {
        const float* wt = p.Wt + (int64_t)p.taps[t].widx * p.w_tap_stride;
        const int cbase = cc * 32 + lh * 16;
        const int cb = cbase < p.Ka ? cbase : 0;
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = n0 + y * 32 + li;
            const int colc = col < p.Cc ? col : p.Cc - 1;
            if constexpr (KMAJOR) {
                const float* src = wt + (int64_t)cb * p.w_ks + colc;
#pragma unroll
                for (int kp = 0; kp < 16; ++kp) { fr.b[kp][y] = *src; src += p.w_ks; }
            } else {
                const float4* src = reinterpret_cast<const float4*>(wt + (int64_t)colc * p.w_ns + cb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = src[j];
                    fr.b[4 * j][y] = v.x; fr.b[4 * j + 1][y] = v.y; fr.b[4 * j + 2][y] = v.z; fr.b[4 * j + 3][y] = v.w;
                }
            }
        }
    };
    auto load_a = [&](Frags<MT, NT>& fr, const float* buf, int t) {
        const IgemmTap tap = p.taps[t];
        const int off = ((tap.dh - x.dh_min) * x.HC + (tap.dw - x.dw_min)) * CS;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* ap = buf + lane_base[m] + off;
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) fr.a[m][kp] = ap[kp];
        }
    };

    int tile = blockIdx.x, cc = 0, cur = 0;
    if (tile >= total_tiles) return;
    // first unit: plain staging
    {
        const Origin o0 = origin_of(tile, 0);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const float* src = halo_src(o0, j);
            const float4 v = *reinterpret_cast<const float4*>(src ? src : p.A);
            halo_store(halo_all, j, v, src != nullptr);
        }
    }
    __syncthreads();
    while (true) {
        const float* buf = halo_all + cur * buf_floats;
        float* nbuf = halo_all + (cur ^ 1) * buf_floats;
        int ntile = tile, ncc = cc + 1;                        // next unit (wave-uniform)
        if (ncc == x.chunks) { ncc = 0; ntile = tile + gridDim.x; }
        const bool has_next = ntile < total_tiles;
        const Origin on = origin_of(has_next ? ntile : tile, ncc);
        if (cc == 0) {
#pragma unroll
            for (int a = 0; a < NPH; ++a)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int y = 0; y < NT; ++y)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][m][y][r] = 0.f;
        }
        load_b(f[0], cc, 0);
        load_a(f[0], buf, 0);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            if (has_next) {
                if (t >= DLY) {
#pragma unroll
                    for (int q = 0; q < LPT; ++q)
                        if ((t - DLY) * LPT + q < PF)
                            halo_store(nbuf, (t - DLY) * LPT + q, ring[(t - DLY) % RING][q], ring_ok[(t - DLY) % RING][q]);
                }
#pragma unroll
                for (int q = 0; q < LPT; ++q)
                    if (t * LPT + q < PF) {
                        const float* src = halo_src(on, t * LPT + q);
                        ring_ok[t % RING][q] = src != nullptr;
                        ring[t % RING][q] = *reinterpret_cast<const float4*>(src ? src : p.A);
                    }
            }
            if (t + 1 < NTAPS) {
                load_b(f[(t + 1) & 1], cc, t + 1);
                load_a(f[(t + 1) & 1], buf, t + 1);
            }
            // keep this tap's loads (operands of tap t+1, halo of the next unit) AHEAD of its MFMAs: left
            // alone, hipcc sinks them to the end of the block and the next tap opens with s_waitcnt vmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
            constexpr int dummy_ = 0; (void)dummy_;
            const int PH = (NPH == 1) ? 0 : ((t >= B1) + (t >= B2) + (t >= B3));      // compile-time after unrolling
#pragma unroll
            for (int kp = 0; kp < 16; ++kp)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int y = 0; y < NT; ++y)
                        acc[PH][m][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[t & 1].a[m][kp], f[t & 1].b[kp][y], acc[PH][m][y], 0, 0, 0);
        }
        if (cc == x.chunks - 1) {
            const int n = tile / tiles_per_img;
            const int r2 = tile - n * tiles_per_img;
            const int th_i = r2 / x.tiles_w, tw_i = r2 - th_i * x.tiles_w;
            const int oh0 = th_i * x.TH, ow0 = tw_i * x.TW;
#pragma unroll
            for (int a = 0; a < NPH; ++a) {
                const int phh = a / p.so_w, phw = a % p.so_w;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int y = 0; y < NT; ++y) {
                        const int col = n0 + y * 32 + li;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int q = (wave * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const int ohp = oh0 + (q >> x.tw_shift), owp = ow0 + (q & (x.TW - 1));
                            if (ohp < p.Hp[0] && owp < p.Wp[0] && col < p.Cc) {
                                const int64_t pix = (int64_t)(n * p.Hc + ohp * p.so_h + phh) * p.Wc + owp * p.so_w + phw;
                                p.Out[pix * p.c_ld + col] = epilogue_value(p, acc[a][m][y][r], pix, col);
                            }
                        }
                    }
            }
        }
        if (!has_next) break;
        __syncthreads();                 // next buffer complete, this one no longer read
        tile = ntile; cc = ncc; cur ^= 1;
    }
}

// [tap][C][K] -> [tap][K][C]: gives the forward convolution the same reduction-contiguous filter
// layout the backward-data kernels read natively (16-byte B-fragment loads instead of 16 strided
// 4-byte loads per tap).  Filters are <= 0.6 MB; one 32x32 LDS tile per workgroup.
__global__ __launch_bounds__(256) void transpose_filter_kernel(const float* __restrict__ w, float* __restrict__ wt, int C, int K) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = w + (int64_t)tap * C * K;
    float* dst = wt + (int64_t)tap * C * K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, k = k0 + tx;
        tile[ty + 8 * j][tx] = (c < C && k < K) ? src[(int64_t)c * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + 8 * j, c = c0 + tx;
        if (k < K && c < C) dst[(int64_t)k * C + c] = tile[tx][ty + 8 * j];
    }
}

template <int NTAPS, int MT, int NT, bool KMAJOR, int PF, int WAVES, int B1 = NTAPS, int B2 = NTAPS, int B3 = NTAPS>
static int launch_hconvp(const IgemmParams& p, const HconvExtra& x, dim3 grid, size_t lds, void* stream, const char* name,
                         const char* who, double flops, double bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvp_kernel<NTAPS, MT, NT, KMAJOR, PF, WAVES, B1, B2, B3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        hconvp_kernel<NTAPS, MT, NT, KMAJOR, PF, WAVES, B1, B2, B3><<<grid, WAVES * 64, lds, s>>>(p, x);
        return launched(who);
    });
}

template <int NPH, int MT, int NT, bool KMAJOR, int WAVES>
static int launch_hconv(const IgemmParams& p, const HconvExtra& x, dim3 grid, size_t lds, void* stream, const char* name,
                        const char* who, double flops, double bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hconv_kernel<NPH, MT, NT, KMAJOR, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        hconv_kernel<NPH, MT, NT, KMAJOR, WAVES><<<grid, WAVES * 64, lds, s>>>(p, x);
        return launched(who);
    });
}

// Tile search: TW in {8..64}, TH = PIX/TW; least padded MFMA work, halo size breaks ties.
static bool pick_tile(const IgemmParams& p, int PIX, int Hp, int Wp, int dh_span, int dw_span, size_t lds_cap, HconvExtra* out, bool b3 = false) {
    int64_t best_cost = -1;
    for (int sh = 3; sh <= 6; ++sh) {
        const int TW = 1 << sh, TH = PIX / TW;
        if (TH < 1 || TH > Hp * 2 || TW > Wp * 2) continue;
        const int th = cdiv(Hp, TH), tw = cdiv(Wp, TW);
        const int HR = (TH - 1) * p.sa_h + dh_span, HC = (TW - 1) * p.sa_w + dw_span;
        size_t lds = (size_t)HR * HC * 33 * sizeof(float);
        if (b3) { HconvExtra t = {}; t.HC = HC; t.HR = HR; t.TW = TW; bconv_set_rows(&t, p.sa_h); lds = (size_t)bconv_lds_bytes(t); }
        if (lds > lds_cap) continue;
        const int64_t cost = (int64_t)th * tw * (PIX * 64 + HR * HC);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            out->TH = TH; out->TW = TW; out->tw_shift = sh; out->tiles_h = th; out->tiles_w = tw; out->HR = HR; out->HC = HC;
            out->G = 1; out->HRi = HR; out->ksplit = 1;
            out->img_shift = (PIX == 256) ? 8 : (PIX == 128 ? 7 : 6);
        }
    }
    return best_cost >= 0;
}

int try_hconv(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes, IgemmParams* epi_out) {
    const int nph = p.so_h * p.so_w;
    if (disabled_paths() & 1) return 1;
    if (p.fold || (nph != 1 && nph != 4)) return 1;
    if (p.Ka % 16 != 0 || p.a_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15)) return 1;
    const bool kmajor_in = kmajor_of(p);
    const bool b3 = !(disabled_paths() & 4096);          // split-bf16 matrix-core path (bconv.hip); off = exact fp32 MFMA
    if (!b3 && !kmajor_in && ((p.w_ns % 4) != 0 || (reinterpret_cast<uintptr_t>(p.Wt) & 15))) return 1;
    const int Hp = p.Hp[0], Wp = p.Wp[0];
    if (nph == 4 && (p.Hp[1] != Hp || p.Wp[1] != Wp)) return 1;
    if (Hp * Wp < 16 || Wp < 4) return 1;
    if (p.Cc < 16) return 1;
    const int ntaps = p.tap_begin[nph];
    if (ntaps < 2) return 1;                      // 1x1 (fc layers): no halo reuse to exploit
    for (int ph = 0; ph < nph; ++ph) if (p.tap_begin[ph + 1] == p.tap_begin[ph]) return 1;
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int t = 0; t < ntaps; ++t) {
        dh_min = std::min<int>(dh_min, p.taps[t].dh); dh_max = std::max<int>(dh_max, p.taps[t].dh);
        dw_min = std::min<int>(dw_min, p.taps[t].dw); dw_max = std::max<int>(dw_max, p.taps[t].dw);
    }
    // ---- small images (phase grid of 16..64 pixels: the 8x8 / 4x4 layers around the bottleneck) --------
    // A tile is G whole images (128 pixels) with their halos stacked; the channel chunks are split over
    // grid.z so that ~512 workgroups exist (each image-row of the GEMM has only N*Hp*Wp = 1k..4k rows but
    // K = 576..2304); raw partial sums are finished by igemm_splitk_epilogue in a fixed order.
    {
        const int ipx = Hp * Wp;
        const bool pow2 = (ipx & (ipx - 1)) == 0 && (Wp & (Wp - 1)) == 0;
        if (ipx <= 64 && ipx >= 16 && pow2 && Wp >= 4 && !(disabled_paths() & 1024)) {
            if (b3 && !(disabled_paths() & 16777216)) {      // reduction split inside the workgroup: one launch (sconv.hip)
                const int rc = try_sconv(p, ws, ws_bytes, stream, who, flops, bytes);
                if (rc != 1) return rc;
            }
            HconvExtra x = {};
            x.G = 128 / ipx; x.TH = Hp; x.TW = Wp; x.tiles_h = 1; x.tiles_w = 1;
            x.tw_shift = 0; while ((1 << x.tw_shift) < Wp) ++x.tw_shift;
            x.img_shift = 0; while ((1 << x.img_shift) < ipx) ++x.img_shift;
            x.HRi = (Hp - 1) * p.sa_h + (dh_max - dh_min + 1);
            x.HC = (Wp - 1) * p.sa_w + (dw_max - dw_min + 1);
            x.HR = x.G * x.HRi;
            x.dh_min = dh_min; x.dw_min = dw_min;
            x.chunks = cdiv(p.Ka, 32); x.ntaps_total = ntaps;
            x.phase_split = (nph == 4) ? 1 : 0;
            x.dbg = 0;
            const size_t lds = (size_t)x.HR * x.HC * 33 * sizeof(float);
            const int tiles = cdiv(p.N, x.G), ny = cdiv(p.Cc, 32);
            const int zph = (nph == 4) ? 4 : 1;
            bconv_set_rows(&x, p.sa_h);
            const size_t wfb = bconv_filter_bytes(p, 1);
            if (b3 && bconv_lds_bytes(x) <= 150 * 1024 && ws && ws_bytes >= wfb && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
                IgemmParams q = p;
                const size_t used = (wfb + 255) & ~(size_t)255;
                static int si_blocks = -1;
                if (si_blocks < 0) { const char* e = getenv("MV3D_SI_BLOCKS"); si_blocks = e ? atoi(e) : 512; }
                int ksplit = std::max(1, std::min(x.chunks, si_blocks / std::max(1, tiles * ny * zph)));
                const size_t per_split = (size_t)p.N * p.Hc * p.Wc * p.Cc * sizeof(float);
                while (ksplit > 1 && used + (size_t)ksplit * per_split > ws_bytes) --ksplit;
                x.ksplit = ksplit;
                q.ksplit = ksplit;
                q.Part = ksplit > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + used) : nullptr;
                dim3 grid(tiles, ny, zph * ksplit);
                int rc = launch_bconv(q, x, 1, 1, 1, 4, grid, ws, stream, "bconv<small-img,128px,N32>", who, flops, bytes);
                if (rc != MV3D_OK) return rc;
                if (ksplit > 1) { *epi_out = q; return 2; }
                return MV3D_OK;
            }
            if (lds <= 150 * 1024) {
                IgemmParams q = p;
                // workspace: [transposed filter copy][split partials]
                size_t used = 0;
                bool use_t = false;
                if (kmajor_in && p.Ka % 4 == 0 && (disabled_paths() & 2048)) {     // opt-in: these launches are latency-bound, the copy costs more than it saves
                    const size_t wbytes = (size_t)ntaps * p.Ka * p.Cc * sizeof(float);
                    if (ws && ws_bytes >= wbytes && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
                        const float* wsrc = p.Wt; float* wdst = (float*)ws;
                        const int C = p.Ka, K = p.Cc;
                        dim3 tg(cdiv(K, 32), cdiv(C, 32), ntaps);
                        int rc = dispatch(stream, OpInfo{"transpose_filter", 0.0, 2.0 * wbytes}, [=](hipStream_t s) {
                            transpose_filter_kernel<<<tg, 256, 0, s>>>(wsrc, wdst, C, K);
                            return launched("transpose_filter_kernel");
                        });
                        if (rc != MV3D_OK) return rc;
                        q.Wt = (const float*)ws; q.w_ks = 1; q.w_ns = p.Ka;
                        use_t = true;
                        used = (wbytes + 255) & ~(size_t)255;
                    }
                }
                const bool km = use_t ? false : kmajor_in;
                int ksplit = std::max(1, std::min(x.chunks, 512 / std::max(1, tiles * ny * zph)));
                const size_t per_split = (size_t)p.N * p.Hc * p.Wc * p.Cc * sizeof(float);
                while (ksplit > 1 && (!ws || used + (size_t)ksplit * per_split > ws_bytes)) --ksplit;
                x.ksplit = ksplit;
                q.ksplit = ksplit;
                q.Part = ksplit > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + used) : nullptr;
                dim3 grid(tiles, ny, zph * ksplit);
                int rc = km ? launch_hconv<1, 1, 1, true, 4>(q, x, grid, lds, stream, "hconv<small-img,128px,N32,kmajorB>", who, flops, bytes)
                            : launch_hconv<1, 1, 1, false, 4>(q, x, grid, lds, stream, "hconv<small-img,128px,N32,nmajorB>", who, flops, bytes);
                if (rc != MV3D_OK) return rc;
                if (ksplit > 1) { *epi_out = q; return 2; }
                return MV3D_OK;
            }
        }
    }
    if (Hp * Wp < 64 || Wp < 8) return 1;
    if (b3 && nph == 4 && !(disabled_paths() & 16777216)) {      // stride-2 transposed convolutions: one phase per workgroup (sconv.hip, TILE)
        const int rc = try_sconv(p, ws, ws_bytes, stream, who, flops, bytes);
        if (rc != 1) return rc;
    }
    if (b3 && nph == 1) {                                         // 4 x 16-tile kernel: stride 2 with 32 channels, the small stride-1 layers (sconv.hip, s2conv)
        const int rc = try_s2conv(p, ws, ws_bytes, stream, who, flops, bytes);
        if (rc != 1) return rc;
    }
    if (b3) {
        // split-bf16 kernels: prefer 64 pixels x 32..64 columns per wave (operand reuse from registers); two
        // workgroups share a CU, so ask for >= 512 workgroups before settling on a tile size
        const int dh_span = dh_max - dh_min + 1, dw_span = dw_max - dw_min + 1;
        const size_t wfb_max = bconv_filter_bytes(p, 2);
        if (ws && ws_bytes >= wfb_max && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
            HconvExtra bx = {};
            int MT = 0, NT = 1, WAVES = 4, fused = nph;
            const char* name = nullptr;
            int ckw = 0; bool crev = false;
            static int cc_min = -1;
            if (cc_min < 0) { const char* e = getenv("MV3D_CC_MINTILES"); cc_min = e ? atoi(e) : 48; }
            const int ctiles = p.N * cdiv(Hp, 16) * cdiv(Wp, 16);
            if (nph == 1 && !(disabled_paths() & 1048576) && cconv_eligible(p, &ckw, &crev) && ctiles * cdiv(p.Cc, 32) >= cc_min) {
                // pipelined kernel (cconv.hip): 16 x 16 tiles x 32 filters, one persistent workgroup per CU
                static int th8_below = -1;       // fewer 16 x 16 workgroup tasks than this: 8 x 16 tiles (MV3D_CC_TH8_BELOW, 0 = never)
                if (th8_below < 0) { const char* e = getenv("MV3D_CC_TH8_BELOW"); th8_below = e ? atoi(e) : 512; }
                const int th = ctiles * cdiv(p.Cc, 32) < th8_below ? 8 : 16;
                bx.TH = th; bx.TW = 16; bx.tw_shift = 4; bx.img_shift = 8;
                bx.tiles_h = cdiv(Hp, th); bx.tiles_w = cdiv(Wp, 16);
                bx.HR = th - 1 + ckw; bx.HC = 20; bx.G = 1; bx.HRi = bx.HR; bx.ksplit = 1;
                MT = th / 8; NT = 1; WAVES = 8; name = "cconv";
            } else if (nph == 1) {
                struct Cand { int pix, MT, NT, WAVES; const char* name; };
                const int n2 = p.Cc > 32 ? 2 : 1;
                const Cand ladder[6] = {{256, 2, n2, 4, n2 == 2 ? "bconv<1ph,256px,N64>" : "bconv<1ph,256px,N32>"}, {256, 2, 1, 4, "bconv<1ph,256px,N32>"},
                                        {128, 1, n2, 4, n2 == 2 ? "bconv<1ph,128px,N64>" : "bconv<1ph,128px,N32>"}, {128, 1, 1, 4, "bconv<1ph,128px,N32>"},
                                        {64, 1, n2, 2, n2 == 2 ? "bconv<1ph,64px,N64>" : "bconv<1ph,64px,N32>"}, {64, 1, 1, 2, "bconv<1ph,64px,N32>"}};
                for (int c = 0; c < 6; ++c) {
                    const Cand& cd = ladder[c];
                    static int maxpix = -1;
                    if (maxpix < 0) { const char* e = getenv("MV3D_BC_MAXPIX"); maxpix = e ? atoi(e) : 256; }
                    if (cd.pix > maxpix) continue;
                    if (cd.pix == 256 && Hp * Wp < 256) continue;
                    HconvExtra x = {};
                    if (!pick_tile(p, cd.pix, Hp, Wp, dh_span, dw_span, 78 * 1024, &x, true)) continue;
                    const int64_t blocks = (int64_t)p.N * x.tiles_h * x.tiles_w * cdiv(p.Cc, 32 * cd.NT);
                    bx = x; MT = cd.MT; NT = cd.NT; WAVES = cd.WAVES; name = cd.name;
                    static int minblk = -1;
                    if (minblk < 0) { const char* e = getenv("MV3D_BC_MINBLK"); minblk = e ? atoi(e) : 512; }
                    if (blocks >= minblk) break;
                }
            } else {
                HconvExtra x = {};
                if (pick_tile(p, 128, Hp, Wp, dh_span, dw_span, 78 * 1024, &x, true)) {
                    const int64_t fused_blocks = (int64_t)p.N * x.tiles_h * x.tiles_w * cdiv(p.Cc, 32);
                    bx = x; MT = 1; NT = 1; WAVES = 4;
                    if (fused_blocks >= 256) { fused = 4; name = "bconv<4ph,128px,N32>"; }
                    else { fused = 1; bx.phase_split = 1; name = "bconv<phase-split,128px,N32>"; }
                    if (fused == 1 && fused_blocks * 4 < 256) {
                        HconvExtra x64 = {};
                        if (pick_tile(p, 64, Hp, Wp, dh_span, dw_span, 78 * 1024, &x64, true)) {
                            bx = x64; bx.phase_split = 1; WAVES = 2; name = "bconv<phase-split,64px,N32>";
                        }
                    }
                }
            }
            if (MT != 0) {
                bx.dh_min = dh_min; bx.dw_min = dw_min;
                bx.chunks = cdiv(p.Ka, 32);
                bx.ntaps_total = ntaps;
                { const char* e = getenv("MV3D_DBG"); bx.dbg = e ? atoi(e) : 0; }
                bconv_set_rows(&bx, p.sa_h);
                IgemmParams q = p;
                q.ksplit = 1;
                dim3 grid(p.N * bx.tiles_h * bx.tiles_w, cdiv(p.Cc, 32 * NT), bx.phase_split ? 4 : 1);
                if (getenv("MV3D_TRACE"))
                    fprintf(stderr, "[mv3d] %-22s %-30s N=%d in %dx%dx%d (stride %d) out %dx%dx%d taps=%d tile %dx%d halo %dx%d lds=%d grid=%dx%dx%d %.2f GFLOP\n",
                            who, name, p.N, p.Ha, p.Wa, p.Ka, p.sa_h, p.Hc, p.Wc, p.Cc, ntaps, bx.TH, bx.TW, bx.HR, bx.HC,
                            bconv_lds_bytes(bx), grid.x, grid.y, grid.z, flops * 1e-9);
                return launch_bconv(q, bx, fused, MT, NT, WAVES, grid, ws, stream, name, who, flops, bytes);
            }
        }
    }
    // Configuration ladder, biggest tile first; step down while the launch would leave CUs idle.
    //   256 px (2 pixel groups per wave): halves per-tap operand traffic; needs a halo <= 78 KB so
    //   that two workgroups still share a CU.   64 px (2 waves): for layers with few pixels.
    struct Cand { int pix, MT, NT, WAVES; size_t cap; };
    const int ntmax = (p.Cc > 32 && nph == 1) ? 2 : 1;
    const Cand ladder[4] = {{256, 2, 1, 4, 78 * 1024}, {128, 1, ntmax, 4, 150 * 1024}, {128, 1, 1, 4, 150 * 1024}, {64, 1, 1, 2, 150 * 1024}};
    HconvExtra best = {};
    int MT = 0, NT = 1, WAVES = 4;
    const int dh_span = dh_max - dh_min + 1, dw_span = dw_max - dw_min + 1;
    for (int c = 0; c < 4; ++c) {
        const Cand& cd = ladder[c];
        if (cd.pix == 64 && (disabled_paths() & 8)) continue;
        if (cd.pix == 256 && (nph != 1 || ntmax != 1 || Hp * Wp < 256)) continue;
        HconvExtra x = {};
        if (!pick_tile(p, cd.pix, Hp, Wp, dh_span, dw_span, cd.cap, &x)) continue;
        const int64_t blocks = (int64_t)p.N * x.tiles_h * x.tiles_w * cdiv(p.Cc, 32 * cd.NT);
        best = x; MT = cd.MT; NT = cd.NT; WAVES = cd.WAVES;
        if (blocks >= 256) break;                  // at least one workgroup per CU
    }
    if (MT == 0) return 1;
    // 4-phase problems with few tiles: one phase per workgroup (grid.z = 4) quadruples the workgroup count;
    // the small input halo is simply staged once per phase.
    int phase_split = 0;
    if (nph == 4) {
        const int64_t fused_blocks = (int64_t)p.N * best.tiles_h * best.tiles_w * cdiv(p.Cc, 32 * NT);
        if (fused_blocks < 256 || WAVES == 2) {
            HconvExtra x128 = {};
            if (pick_tile(p, 128, Hp, Wp, dh_span, dw_span, 150 * 1024, &x128) &&
                (int64_t)p.N * x128.tiles_h * x128.tiles_w * cdiv(p.Cc, 32) * 4 >= 256) {
                best = x128; MT = 1; NT = 1; WAVES = 4;
            } else {
                HconvExtra x64 = {};
                if (!pick_tile(p, 64, Hp, Wp, dh_span, dw_span, 150 * 1024, &x64)) return 1;
                best = x64; MT = 1; NT = 1; WAVES = 2;
            }
            phase_split = 1;
        }
    }
    {   // diagnostics: MV3D_HCONV_SKIP=i falls back to igemm for the i-th eligible call only
        static int counter = 0;
        const char* e = getenv("MV3D_HCONV_SKIP");
        const int idx = counter++;
        if (e && atoi(e) == idx) { fprintf(stderr, "[mv3d] hconv call %d skipped: %s Ha=%d Ca=%d Cc=%d nph=%d\n", idx, who, p.Ha, p.Ca, p.Cc, nph); return 1; }
    }
    best.dh_min = dh_min; best.dw_min = dw_min;
    best.chunks = cdiv(p.Ka, 32);
    best.ntaps_total = ntaps;
    best.phase_split = phase_split;
    { const char* e = getenv("MV3D_DBG"); best.dbg = e ? atoi(e) : 0; }
    const size_t lds = (size_t)best.HR * best.HC * 33 * sizeof(float);
    dim3 grid(p.N * best.tiles_h * best.tiles_w, cdiv(p.Cc, 32 * NT), 1);
    IgemmParams q = p;
    q.ksplit = 1;
    // Forward-direction filters ([tap][C][K], reduction index strided): make a [tap][K][C] copy in the
    // workspace so that B fragments are 16-byte loads like in the backward-data direction.
    bool use_t = false;
    if (kmajor_in && p.Ka % 4 == 0 && !(disabled_paths() & 256)) {
        const size_t wbytes = (size_t)ntaps * p.Ka * p.Cc * sizeof(float);
        if (ws && ws_bytes >= wbytes && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
            const float* wsrc = p.Wt; float* wdst = (float*)ws;
            const int C = p.Ka, K = p.Cc;
            dim3 tg(cdiv(K, 32), cdiv(C, 32), ntaps);
            int rc = dispatch(stream, OpInfo{"transpose_filter", 0.0, 2.0 * wbytes}, [=](hipStream_t s) {
                transpose_filter_kernel<<<tg, 256, 0, s>>>(wsrc, wdst, C, K);
                return launched("transpose_filter_kernel");
            });
            if (rc != MV3D_OK) return rc;
            q.Wt = (const float*)ws; q.w_ks = 1; q.w_ns = p.Ka;      // [tap][Cc][Ka]
            use_t = true;
        }
    }
    const bool kmajor = use_t ? false : kmajor_in;
    if (nph == 1 && WAVES == 4 && (ntaps == 25 || ntaps == 9) && p.so_h == 1 && p.so_w == 1 && 2 * lds <= 160 * 1024 &&
        !(disabled_paths() & 128)) {
        // persistent kernel, one workgroup per CU (two halo buffers in LDS)
        const int pf_need = cdiv(best.HR * best.HC * 8, 256);
        const int ny = cdiv(p.Cc, 32 * NT);
        const int tiles_total = p.N * best.tiles_h * best.tiles_w;
        dim3 pgrid(std::min(tiles_total, std::max(1, 256 / ny)), ny, 1);
        const size_t lds2 = 2 * lds;
#define MV3D_HCONVP(NTAPS_, MT_, NT_, KM_, PF_, W_, NAME) launch_hconvp<NTAPS_, MT_, NT_, KM_, PF_, W_>(q, best, pgrid, lds2, stream, NAME, who, flops, bytes)
        // 256-pixel tiles run as 8 waves x 32 pixels: two waves per SIMD cover each other's non-MFMA issue slots
        const int pf8 = cdiv(best.HR * best.HC * 8, 512);
        if (ntaps == 25) {
            if (MT == 2 && pf8 <= 9) return kmajor ? MV3D_HCONVP(25, 1, 1, true, 9, 8, "hconvp<5x5,256px,N32,kmajorB>") : MV3D_HCONVP(25, 1, 1, false, 9, 8, "hconvp<5x5,256px,N32,nmajorB>");
            if (MT == 1 && NT == 1 && pf_need <= 13) return kmajor ? MV3D_HCONVP(25, 1, 1, true, 13, 4, "hconvp<5x5,128px,N32,kmajorB>") : MV3D_HCONVP(25, 1, 1, false, 13, 4, "hconvp<5x5,128px,N32,nmajorB>");
            if (MT == 1 && NT == 2 && pf_need <= 13) return kmajor ? MV3D_HCONVP(25, 1, 2, true, 13, 4, "hconvp<5x5,128px,N64,kmajorB>") : MV3D_HCONVP(25, 1, 2, false, 13, 4, "hconvp<5x5,128px,N64,nmajorB>");
            if (MT == 1 && NT == 1 && pf_need <= 23) return kmajor ? MV3D_HCONVP(25, 1, 1, true, 23, 4, "hconvp<5x5,128px,N32,kmajorB,bighalo>") : MV3D_HCONVP(25, 1, 1, false, 23, 4, "hconvp<5x5,128px,N32,nmajorB,bighalo>");
        } else {
            if (MT == 2 && pf8 <= 7) return kmajor ? MV3D_HCONVP(9, 1, 1, true, 7, 8, "hconvp<3x3,256px,N32,kmajorB>") : MV3D_HCONVP(9, 1, 1, false, 7, 8, "hconvp<3x3,256px,N32,nmajorB>");
            if (MT == 1 && NT == 1 && pf_need <= 14) return kmajor ? MV3D_HCONVP(9, 1, 1, true, 14, 4, "hconvp<3x3,128px,N32,kmajorB>") : MV3D_HCONVP(9, 1, 1, false, 14, 4, "hconvp<3x3,128px,N32,nmajorB>");
            if (MT == 1 && NT == 2 && pf_need <= 14) return kmajor ? MV3D_HCONVP(9, 1, 2, true, 14, 4, "hconvp<3x3,128px,N64,kmajorB>") : MV3D_HCONVP(9, 1, 2, false, 14, 4, "hconvp<3x3,128px,N64,nmajorB>");
        }
#undef MV3D_HCONVP
    }
    if (nph == 4 && WAVES == 4 && MT == 1 && NT == 1 && !kmajor && (ntaps == 25 || ntaps == 9) && 2 * lds <= 160 * 1024 &&
        (disabled_paths() & 512)) {          // opt-in (MV3D_DISABLE bit 9): measured slower than two resident workgroups per CU
        // 4-phase persistent kernel (stride-2 transposed conv / conv dgrad): the phase of every tap is a
        // compile-time constant (taps are listed phase by phase: 4,6,6,9 for 5x5 and 4,2,2,1 for 3x3)
        bool layout_ok;
        if (ntaps == 25) layout_ok = p.tap_begin[1] == 4 && p.tap_begin[2] == 10 && p.tap_begin[3] == 16;
        else layout_ok = p.tap_begin[1] == 4 && p.tap_begin[2] == 6 && p.tap_begin[3] == 8;
        const int pf_need = cdiv(best.HR * best.HC * 8, 256);
        if (layout_ok && pf_need <= 7) {
            const int ny = cdiv(p.Cc, 32);
            const int tiles_total = p.N * best.tiles_h * best.tiles_w;
            dim3 pgrid(std::min(tiles_total, std::max(1, 256 / ny)), ny, 1);
            const size_t lds2 = 2 * lds;
            if (ntaps == 25) return launch_hconvp<25, 1, 1, false, 7, 4, 4, 10, 16>(q, best, pgrid, lds2, stream, "hconvp<4ph,5x5,128px,N32>", who, flops, bytes);
            return launch_hconvp<9, 1, 1, false, 7, 4, 4, 6, 8>(q, best, pgrid, lds2, stream, "hconvp<4ph,3x3,128px,N32>", who, flops, bytes);
        }
    }
    if (phase_split) {
        dim3 sgrid(p.N * best.tiles_h * best.tiles_w, cdiv(p.Cc, 32), 4);
        if (WAVES == 4) return kmajor ? launch_hconv<1, 1, 1, true, 4>(q, best, sgrid, lds, stream, "hconv<phase-split,128px,N32,kmajorB>", who, flops, bytes)
                                      : launch_hconv<1, 1, 1, false, 4>(q, best, sgrid, lds, stream, "hconv<phase-split,128px,N32,nmajorB>", who, flops, bytes);
        return kmajor ? launch_hconv<1, 1, 1, true, 2>(q, best, sgrid, lds, stream, "hconv<phase-split,64px,N32,kmajorB>", who, flops, bytes)
                      : launch_hconv<1, 1, 1, false, 2>(q, best, sgrid, lds, stream, "hconv<phase-split,64px,N32,nmajorB>", who, flops, bytes);
    }
#define MV3D_HCONV(NPH_, MT_, NT_, KM_, W_, NAME) launch_hconv<NPH_, MT_, NT_, KM_, W_>(q, best, grid, lds, stream, NAME, who, flops, bytes)
    if (nph == 1) {
        if (MT == 2) return kmajor ? MV3D_HCONV(1, 2, 1, true, 4, "hconv<1ph,256px,N32,kmajorB>") : MV3D_HCONV(1, 2, 1, false, 4, "hconv<1ph,256px,N32,nmajorB>");
        if (WAVES == 2) return kmajor ? MV3D_HCONV(1, 1, 1, true, 2, "hconv<1ph,64px,N32,kmajorB>") : MV3D_HCONV(1, 1, 1, false, 2, "hconv<1ph,64px,N32,nmajorB>");
        if (NT == 1) return kmajor ? MV3D_HCONV(1, 1, 1, true, 4, "hconv<1ph,128px,N32,kmajorB>") : MV3D_HCONV(1, 1, 1, false, 4, "hconv<1ph,128px,N32,nmajorB>");
        return kmajor ? MV3D_HCONV(1, 1, 2, true, 4, "hconv<1ph,128px,N64,kmajorB>") : MV3D_HCONV(1, 1, 2, false, 4, "hconv<1ph,128px,N64,nmajorB>");
    }
    if (WAVES == 2) return kmajor ? MV3D_HCONV(4, 1, 1, true, 2, "hconv<4ph,64px,N32,kmajorB>") : MV3D_HCONV(4, 1, 1, false, 2, "hconv<4ph,64px,N32,nmajorB>");
    return kmajor ? MV3D_HCONV(4, 1, 1, true, 4, "hconv<4ph,128px,N32,kmajorB>") : MV3D_HCONV(4, 1, 1, false, 4, "hconv<4ph,128px,N32,nmajorB>");
#undef MV3D_HCONV
}

}  // namespace mv3d
