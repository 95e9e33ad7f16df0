// Halo-tile convolution on the bf16 matrix cores at fp32-class accuracy ("split-bf16").
//
// gfx950's fp32 MFMA runs at 1/16 of the bf16 rate.  Every fp32 operand x is split exactly-enough into
//   x = hi + lo + r,   hi = bf16_rne(x),  lo = bf16_rne(x - hi),  |r| <= 2^-17 |x|
// and every product a*b is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (three
// v_mfma_f32_32x32x16_bf16, fp32 accumulation; bf16 x bf16 products are exact in fp32).  The dropped
// terms are below 2^-16 relative per product -- a few 1e-6 after accumulation, against the 1e-3 parity bar
// -- for 16/3 = 5.3x the arithmetic rate of the fp32 MFMA path.
//
// Structure (same tile scheme as hconv.hip, whose planner produces the HconvExtra used here):
//   * a workgroup owns a TH x TW tile of the output's phase grid for ALL taps; the input halo is staged in
//     LDS once per 32-channel chunk, converted on the way in: per halo pixel 32 hi-bf16 (64 B), 32 lo-bf16
//     (64 B), 16 B pad (pixel stride 144 B = 36 banks: sixteen consecutive pixels cover all 64 banks once
//     for ds_read_b128; the row stride is padded so that the lane groups of that instruction stay
//     conflict-free across tile rows);
//   * A fragments = ds_read_b128 of 8 consecutive channels of a lane's pixel (+ tap offset);
//   * B fragments = the filter, split once per launch by bconv_split_filter_kernel into MFMA fragment order:
//     a wave's load is one contiguous, fully coalesced 1 KiB line group from L2;
//   * tap loop without barriers, operands of tap t+1 in flight under the MFMAs of tap t (two register sets,
//     loop unrolled by two so no register copies), stride-2 transposed convs as 4 phases from one halo.
#include "conv_common.h"
#include <algorithm>
#include <mutex>
#include <string.h>
#include <vector>

namespace mv3d {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BC_PIXB = 144;        // LDS bytes per halo pixel

__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    bf16x4 h, l;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
    l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}


// Halo staging shared by the bconv kernels: fp32 global -> split -> LDS.  A thread owns one 4-channel group
// (c4 = tid & 7) and every (NTHR/8)-th halo pixel; the (row, column) of its pixels advance by a constant step, so
// the loop carries no divisions and only a handful of integer instructions per 16-byte piece (the address
// arithmetic of a naive index -> (row, column) decomposition costs more than the MFMAs of a small layer).
template <int NTHR>
__device__ __forceinline__ void bconv_stage_halo(const IgemmParams& p, const HconvExtra& x, unsigned char* halo, int cc, int n,
                                                 int ih0, int iw0, int tid) {
    constexpr int PS = NTHR / 8;                         // pixel step between a thread's pieces
    const int c4 = tid & 7;
    const int halo_pix = x.HR * x.HC;
    const int ch = cc * 32 + c4 * 4;
    const bool ch_ok = ch < p.Ka && !(x.dbg & 1);
    const int dq = PS / x.HC, dr = PS - dq * x.HC;       // wave-uniform (scalar) division, once per call
    int pix = tid >> 3;
    int hrv = (int)(((unsigned)pix * (unsigned)x.inv_hc) >> 20), hc = pix - hrv * x.HC;
    for (; pix < halo_pix;) {
        float4 v[8];
        int lofs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = x.G > 1 ? (int)(((unsigned)hrv * (unsigned)x.inv_hri) >> 20) : 0;
            const int hr = hrv - g * x.HRi;
            const int ih = ih0 + hr, iw = iw0 + hc;
            const bool ok = ch_ok && pix < halo_pix && n + g < p.N && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
            const float* src = ok ? p.A + (int64_t)(((n + g) * p.Ha + ih) * p.Wa + iw) * p.a_ld + ch : p.A;
            const float4 t4 = *reinterpret_cast<const float4*>(src);
            v[u] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            lofs[u] = pix < halo_pix ? hrv * x.row_bytes + hc * BC_PIXB + c4 * 8 : -1;
            pix += PS; hc += dr; hrv += dq;
            if (hc >= x.HC) { hc -= x.HC; ++hrv; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (lofs[u] >= 0) {
                uint2 hi, lo;
                split4(v[u], hi, lo);
                *reinterpret_cast<uint2*>(halo + lofs[u]) = hi;
                *reinterpret_cast<uint2*>(halo + lofs[u] + 64) = lo;
            }
        }
    }
}

// Output tile store shared by the bconv kernels.  acc rows: q = 8*(r>>2) + (r&3) + 4*lh inside a 32-pixel group; four
// consecutive r are four consecutive pixels of one tile row, so one address is computed per group of four.  Tiles
// that lie completely inside the output take a path without per-element bounds checks.
template <int MT, int NT>
__device__ __forceinline__ void bconv_store_tile(const IgemmParams& p, const HconvExtra& x, const f32x16 (&acc)[MT][NT], int wave, int lane,
                                                 int n, int oh0, int ow0, int n0, int phh, int phw, int zks) {
    const int li = lane & 31, lh = lane >> 5;
    const int Hp = p.Hp[0], Wp = p.Wp[0];
    const int64_t npix_total = (int64_t)p.N * p.Hc * p.Wc;
    const bool interior = n + x.G <= p.N && oh0 + x.TH <= Hp && ow0 + x.TW <= Wp && n0 + 32 * NT <= p.Cc && !(x.dbg & 16);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = n0 + y * 32 + li;
            const float bias = (p.bias && col < p.Cc) ? p.bias[col] : 0.f;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int q = (wave * MT + m) * 32 + 8 * grp + 4 * lh;
                const int g = q >> x.img_shift, qr = q & ((1 << x.img_shift) - 1);
                const int ohp = oh0 + (qr >> x.tw_shift), owp = ow0 + (qr & (x.TW - 1));
                const int64_t pix0 = (int64_t)((n + g) * p.Hc + ohp * p.so_h + phh) * p.Wc + owp * p.so_w + phw;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[m][y][4 * grp + j];
                    const int64_t pix = pix0 + j * p.so_w;
                    if (interior || (n + g < p.N && ohp < Hp && owp + j < Wp && col < p.Cc && !(x.dbg & 16))) {
                        if (x.ksplit > 1) p.Part[((int64_t)zks * npix_total + pix) * p.Cc + col] = a;
                        else {
                            float v = act_apply(a + bias, p.act, p.leak);
                            if (p.gact != MV3D_ACT_NONE) v *= act_grad_from_out(p.gref[pix * p.g_ld + col], p.gact, p.gleak);
                            p.Out[pix * p.c_ld + col] = v;
                        }
                    }
                }
            }
        }
}

// Filter -> fragment order.  Wf[((t * chunks + cc) * ntiles + nt) * 4 + (s * 2 + part)][lane] (16 bytes each):
// lane (li, lh) holds channels cc*32 + s*16 + lh*8 + 0..7 of output column nt*32 + li, part 0 = hi, 1 = lo.
// Channels >= Ka and columns >= Cc are zero, so the main kernel needs no clamping.
__global__ __launch_bounds__(256) void bconv_split_filter_kernel(const IgemmParams p, uint4* __restrict__ Wf, int ntaps, int chunks, int ntiles) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(gid & 63);
    int64_t r = gid >> 6;
    const int s = (int)(r & 1); r >>= 1;
    const int nt = (int)(r % ntiles); r /= ntiles;
    const int cc = (int)(r % chunks); r /= chunks;
    const int t = (int)r;
    if (t >= ntaps) return;
    const int li = lane & 31, lh = lane >> 5;
    const int col = nt * 32 + li;
    const int ch0 = cc * 32 + s * 16 + lh * 8;
    const float* wt = p.Wt + (int64_t)p.taps[t].widx * p.w_tap_stride;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = ch0 + j;
        v[j] = (col < p.Cc && ch < p.Ka) ? wt[(int64_t)ch * p.w_ks + (int64_t)col * p.w_ns] : 0.f;
    }
    uint2 h0, l0, h1, l1;
    split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    uint4* dst = Wf + ((((int64_t)t * chunks + cc) * ntiles + nt) * 4 + s * 2) * 64 + lane;
    dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

template <int MT, int NT>
struct BFrags {
    uint4 a[MT][2][2];      // [pixel group][k-step][hi, lo]
    uint4 b[NT][2][2];      // [column group][k-step][hi, lo]
};

template <int NPH, int MT, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bconv_kernel(const IgemmParams p, const HconvExtra x, const uint4* __restrict__ Wf, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char halo[];
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
    const int zks = (int)blockIdx.z % x.ksplit;          // chunk-split index; grid.z = phase * ksplit + zks

    int lane_base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pidx = (wave * MT + m) * 32 + li;
        const int g = pidx >> x.img_shift, pr = pidx & ((1 << x.img_shift) - 1);
        const int tr = pr >> x.tw_shift, tc = pr & (x.TW - 1);
        lane_base[m] = (g * x.HRi + tr * p.sa_h) * x.row_bytes + tc * p.sa_w * BC_PIXB + lh * 16;
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

    // phase-split launches: this workgroup's taps are [tap_lo, tap_lo + ntaps_here) of the flat list
    const int ph_z = x.phase_split ? (int)blockIdx.z / x.ksplit : 0;
    const int tap_lo = x.phase_split ? p.tap_begin[ph_z] : 0;
    const int ntaps_here = x.phase_split ? p.tap_begin[ph_z + 1] - tap_lo : x.ntaps_total;
    const int tap_hi = tap_lo + ntaps_here;
    const int cc_begin = (x.chunks * zks) / x.ksplit, cc_end = (x.chunks * (zks + 1)) / x.ksplit;
    const int total_seq = (cc_end - cc_begin) * ntaps_here;

    const uint4* wf_lane = Wf + (int64_t)(blockIdx.y * NT) * 256 + lane;
    auto load_b = [&](BFrags<MT, NT>& f, int seq) {
        seq = seq < total_seq ? seq : total_seq - 1;                    // look-ahead past the end re-reads the last tap
        const int cq = seq / ntaps_here;
        const int cc = cc_begin + cq;
        const int t = tap_lo + seq - cq * ntaps_here;
        const uint4* src = wf_lane + ((int64_t)t * x.chunks + cc) * ntiles * 256;
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f.b[y][s][0] = src[(y * 4 + s * 2) * 64];
                f.b[y][s][1] = src[(y * 4 + s * 2 + 1) * 64];
            }
    };
    // LDS byte offset of a tap, fetched one tap early
    // (lane t of every wave keeps tap t's offset; v_readlane with the wave-uniform tap index costs no memory access)
    int lane_off;
    {
        const IgemmTap tap = p.taps[lane < x.ntaps_total ? lane : 0];
        lane_off = (tap.dh - x.dh_min) * x.row_bytes + (tap.dw - x.dw_min) * BC_PIXB;
    }
    auto tap_off = [&](int t) {
        t = t < tap_hi ? t : tap_hi - 1;
        return __builtin_amdgcn_readlane(lane_off, t);
    };
    auto load_a = [&](BFrags<MT, NT>& f, int off) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned char* ap = halo + lane_base[m] + off;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f.a[m][s][0] = *reinterpret_cast<const uint4*>(ap + s * 32);
                f.a[m][s][1] = *reinterpret_cast<const uint4*>(ap + 64 + s * 32);
            }
        }
    };

    BFrags<MT, NT> f0, f1;
    load_b(f0, 0);
    const int ih0 = oh0 * p.sa_h + x.dh_min, iw0 = ow0 * p.sa_w + x.dw_min;
    const int halo_pix = x.HR * x.HC;
    int seq = 0;
    for (int cc = cc_begin; cc < cc_end; ++cc) {
        if (cc > cc_begin) __syncthreads();
        // halo staging: batches of 8 independent 16-byte loads per thread, then split + LDS stores;
        // out-of-image pixels load a valid dummy address and are zeroed
        bconv_stage_halo<NTHR>(p, x, halo, cc, n, ih0, iw0, tid);
        __syncthreads();
        load_a(f0, tap_off(tap_lo));
        int off1 = tap_off(tap_lo + 1);                  // offset of the tap after the one in f0
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            auto mma = [&](const BFrags<MT, NT>& f) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int y = 0; y < NT; ++y) {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, f.a[m][s][0]), al = __builtin_bit_cast(bf16x8, f.a[m][s][1]);
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, f.b[y][s][0]), bl = __builtin_bit_cast(bf16x8, f.b[y][s][1]);
                            acc[ph][m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ph][m][y], 0, 0, 0);
                            acc[ph][m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ph][m][y], 0, 0, 0);
                            acc[ph][m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ph][m][y], 0, 0, 0);
                        }
            };
            const int tb = x.phase_split ? tap_lo : p.tap_begin[ph], te = x.phase_split ? tap_hi : p.tap_begin[ph + 1];
            int t = tb;
            // invariant: f0 holds the operands of tap t
            if (x.dbg & 8) t = te;
            // every iteration issues the same loads unconditionally (clamped indices): a conditional load makes hipcc
            // fall back to s_waitcnt vmcnt(0) at the join, i.e. no look-ahead at all
            for (; t + 1 < te; t += 2, seq += 2) {
                const int off2 = tap_off(t + 2);
                load_b(f1, seq + 1); load_a(f1, off1);
                __builtin_amdgcn_sched_barrier(0);
                mma(f0);
                const int off3 = tap_off(t + 3);
                load_b(f0, seq + 2); load_a(f0, off2);
                __builtin_amdgcn_sched_barrier(0);
                mma(f1);
                off1 = off3;
            }
            if (t < te) {                  // odd tap count: one more, then hand the look-ahead set over
                const int off2 = tap_off(t + 2);
                load_b(f1, seq + 1); load_a(f1, off1);
                __builtin_amdgcn_sched_barrier(0);
                mma(f0);
                f0 = f1;
                off1 = off2;
                ++seq;
            }
        }
    }

#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int phe = x.phase_split ? ph_z : ph;
        bconv_store_tile<MT, NT>(p, x, acc[ph], wave, lane, n, oh0, ow0, n0, phe / p.so_w, phe % p.so_w, zks);
    }
}

// Single-phase variant with the tap loop fully unrolled (NTAPS = 25 or 9 is a template parameter):
//   * filter fragments travel in a ring of U register sets (U divides NTAPS, so the slot of a tap is a compile-time
//     constant in every chunk): the load for tap t+U-1 is issued when tap t starts -- U-1 taps (~1.5k cycles at 64
//     pixels per wave) of L2 latency cover instead of one;
//   * the activation fragments have ONE register set: each (pixel group, k-step) pair is refilled for the next tap
//     right behind the three MFMAs that consumed it, which leaves the rest of the tap to cover the LDS latency;
//   * tap offsets are compile-time indexed scalars: no scalar loads or waits inside the loop.
template <int NTAPS, int U, int MT, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2))) void bconvu_kernel(const IgemmParams p, const HconvExtra x, const uint4* __restrict__ Wf, int ntiles) {
    static_assert(NTAPS % U == 0, "ring slots must line up across chunks");
    // look-ahead distance in taps: the whole ring at 32 filters per wave, two taps at 64 (twice the MFMAs per tap and
    // twice the registers per ring slot; two waves per SIMD need <= 256 registers)
    constexpr int D = (NT == 1) ? U - 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char halo[];
    constexpr int NTHR = WAVES * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    const int n0 = blockIdx.y * 32 * NT;
    const int zks = (int)blockIdx.z;                     // chunk-split index (grid.z = ksplit)

    int lane_base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pidx = (wave * MT + m) * 32 + li;
        const int g = pidx >> x.img_shift, pr = pidx & ((1 << x.img_shift) - 1);
        const int tr = pr >> x.tw_shift, tc = pr & (x.TW - 1);
        lane_base[m] = (g * x.HRi + tr * p.sa_h) * x.row_bytes + tc * p.sa_w * BC_PIXB + lh * 16;
    }
    int toff[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
        toff[t] = (p.taps[t].dh - x.dh_min) * x.row_bytes + (p.taps[t].dw - x.dw_min) * BC_PIXB;

    f32x16 acc[MT][NT];

    const int cc_begin = (x.chunks * zks) / x.ksplit, cc_end = (x.chunks * (zks + 1)) / x.ksplit;

    struct BSet { uint4 b[NT][2][2]; };
    BSet ring[U];
    uint4 a[MT][2][2];
    // filter fragments of (tap t, chunk cc): wave-uniform base (scalar arithmetic) + lane index
    const uint4* wf_base = Wf + (int64_t)(blockIdx.y * NT) * 256;
    const int64_t tap_stride = (int64_t)x.chunks * ntiles * 256, chunk_stride = (int64_t)ntiles * 256;
    auto load_b = [&](BSet& f, int t, int cc) {
        const uint4* src = wf_base + t * tap_stride + cc * chunk_stride;
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f.b[y][s][0] = src[(y * 4 + s * 2) * 64 + lane];
                f.b[y][s][1] = src[(y * 4 + s * 2 + 1) * 64 + lane];
            }
    };
    // workgroups walk the tile list with stride gridDim.x (the host launches about two workgroups per CU): the
    // per-workgroup start-up cost is paid once, and co-resident workgroups drift out of phase so that one's
    // staging / stores overlap the other's MFMAs
    const int total_tiles = x.n_tiles;
    bool first = true;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int bq = tile;
    const int tw_i = bq % x.tiles_w; bq /= x.tiles_w;
    const int th_i = bq % x.tiles_h;
    const int n = (bq / x.tiles_h) * x.G;
    const int oh0 = th_i * x.TH, ow0 = tw_i * x.TW;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][y][r] = 0.f;
    if (tile == (int)blockIdx.x) {      // later tiles find their first taps in the ring already (look-ahead of the previous tile)
#pragma unroll
        for (int u = 0; u < D; ++u) load_b(ring[u], u, cc_begin);
    }

    const int ih0 = oh0 * p.sa_h + x.dh_min, iw0 = ow0 * p.sa_w + x.dw_min;
    for (int cc = cc_begin; cc < cc_end; ++cc) {
        if (cc > cc_begin || !first) __syncthreads();
        first = false;
        bconv_stage_halo<NTHR>(p, x, halo, cc, n, ih0, iw0, tid);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned char* ap = halo + lane_base[m] + toff[0] + s * 32;
                a[m][s][0] = *reinterpret_cast<const uint4*>(ap);
                a[m][s][1] = *reinterpret_cast<const uint4*>(ap + 64);
            }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            {   // look-ahead tap: same chunk, or the first taps of the next chunk (of the next tile after the last chunk)
                const int tn = t + D;
                const int ccn = cc + 1 < cc_end ? cc + 1 : cc_begin;
                load_b(ring[tn % U], tn < NTAPS ? tn : tn - NTAPS, tn < NTAPS ? cc : ccn);
            }
            // keep the look-ahead load HERE: left alone, hipcc sinks it down to its first use four taps later and the
            // ring degenerates into load-wait-multiply
            __builtin_amdgcn_sched_barrier(0);
            const BSet& f = ring[t % U];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, a[m][s][0]), al = __builtin_bit_cast(bf16x8, a[m][s][1]);
#pragma unroll
                    for (int y = 0; y < NT; ++y) {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, f.b[y][s][0]), bl = __builtin_bit_cast(bf16x8, f.b[y][s][1]);
                        acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m][y], 0, 0, 0);
                        acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m][y], 0, 0, 0);
                        acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][y], 0, 0, 0);
                    }
                    if (t + 1 < NTAPS) {
                        const unsigned char* ap = halo + lane_base[m] + toff[t + 1 < NTAPS ? t + 1 : t] + s * 32;
                        a[m][s][0] = *reinterpret_cast<const uint4*>(ap);
                        a[m][s][1] = *reinterpret_cast<const uint4*>(ap + 64);
                    }
                }
        }
    }

    bconv_store_tile<MT, NT>(p, x, acc, wave, lane, n, oh0, ow0, n0, 0, 0, zks);
    }
}


// LDS row stride: 16 consecutive lanes of a ds_read_b128 group must land on 16 distinct 4-bank slots.
// With 144-byte pixels that holds inside a tile row; across tile rows it needs the row stride
// = 0 (mod 256 B) for 16-pixel rows and = 128 (mod 256 B) for 8-pixel rows (MI355X_MICROARCH.md, LDS lane groups).
// Modelled cost of one wave's A-operand ds_read_b128 for a row pitch: the LDS serves such a read in four groups of 16 lanes
// (lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same two in the upper half); a group is conflict-free when its 16
// lanes hit 16 different 16-byte slots of the 256-byte bank row.  Returns the sum over the two lower groups of the worst slot
// multiplicity (2 = no conflict); the upper half only differs by a constant offset.
static int bconv_pitch_cost(const HconvExtra& x, int stride, int rb) {
    static const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int cost = 0;
    for (int g = 0; g < 2; ++g) {
        int cnt[16] = {};
        int worst = 0;
        for (int i = 0; i < 16; ++i) {
            const int pidx = groups[g][i];
            const int gi = pidx >> x.img_shift, pr = pidx & ((1 << x.img_shift) - 1);
            const int tr = pr >> x.tw_shift, tc = pr & (x.TW - 1);
            const long addr = (long)(gi * x.HRi + tr * stride) * rb + (long)tc * stride * BC_PIXB;
            worst = std::max(worst, ++cnt[(addr >> 4) & 15]);
        }
        cost += worst;
    }
    return cost;
}

void bconv_set_rows(HconvExtra* x, int stride_h) {
    // exact for dividends < 2^20 / divisor (halo pixel indices stay below a few thousand)
    x->inv_hc = ((1 << 20) + x->HC - 1) / x->HC;
    x->inv_hri = x->HRi > 0 ? ((1 << 20) + x->HRi - 1) / x->HRi : 0;
    int rb = x->HC * BC_PIXB;
    const int rb_min = rb;
    static const int s2rows = getenv("MV3D_BC_S2ROWS") ? atoi(getenv("MV3D_BC_S2ROWS")) : 1;
    if (x->TW == 16) rb = (rb + 255) & ~255;
    else if (x->TW == 8 && stride_h == 2 && s2rows) rb = ((rb + 63) & ~127) + 64;     // two halo rows per tile row: 2 * rb = 128 mod 256
    else if (x->TW == 8) rb = ((rb + 127) & ~255) + 128;
    // Pitch search: among the pitches between the bare row and the rule above, the smallest one with the fewest modelled bank
    // conflicts.  For TW >= 8 it never takes more LDS than the rule, so pick_tile's size estimate (which passes only HC / HR / TW
    // and therefore keeps the rule) is an upper bound of what is launched.  Tiles narrower than 8 pixels may take up to 240 bytes
    // per halo row MORE than the bare pitch: they only come from the small-image planner (hconv.hip), which calls this function
    // with the full geometry and sizes LDS, capacity checks and workgroups per CU from bconv_lds_bytes() of the result.
    static const int search = getenv("MV3D_BC_PITCHSEARCH") ? atoi(getenv("MV3D_BC_PITCHSEARCH")) : 1;
    const bool geom = x->TW > 0 && x->img_shift > 0 && (1 << x->tw_shift) == x->TW && (x->G <= 1 || x->HRi > 0);
    if (search && geom) {
        int best = rb, bestc = bconv_pitch_cost(*x, stride_h, rb);
        // tiles narrower than 8 pixels (4 x 4 and 2 x 2 maps, several images per tile) have no rule: look up to one bank row
        // beyond the bare pitch (these halos are small; the extra LDS is a few KB)
        const int hi = x->TW < 8 ? rb_min + 256 : rb;
        for (int c = (rb_min + 15) & ~15; c < hi; c += 16) {
            const int k = bconv_pitch_cost(*x, stride_h, c);
            if (k < bestc) { best = c; bestc = k; }
        }
        if (getenv("MV3D_TRACE")) fprintf(stderr, "[mv3d] halo row pitch %d (rule %d, bare %d): modelled read cost %d (2 = conflict-free)\n", best, rb, rb_min, bestc);
        rb = best;
    }
    x->row_bytes = rb;
}

int bconv_lds_bytes(const HconvExtra& x) { return x.HR * x.row_bytes; }

size_t bconv_filter_bytes(const IgemmParams& p, int NT) {
    const int nph = p.so_h * p.so_w;
    const int ntaps = p.tap_begin[nph];
    const int chunks = cdiv(p.Ka, 32), ntiles = cdiv(p.Cc, 32 * NT) * NT;
    return (size_t)ntaps * chunks * ntiles * 4096;
}

template <int NPH, int MT, int NT, int WAVES>
static int launch_bconv_t(const IgemmParams& p, const HconvExtra& x, dim3 grid, size_t lds, const uint4* wf, int ntiles, void* stream,
                          const char* name, const char* who, double flops, double bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bconv_kernel<NPH, MT, NT, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (x.ksplit > 1) name = intern_label("%s+ksplit", name);
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        bconv_kernel<NPH, MT, NT, WAVES><<<grid, WAVES * 64, lds, s>>>(p, x, wf, ntiles);
        return launched(who);
    });
}

template <int NTAPS, int U, int MT, int NT, int WAVES>
static int launch_bconvu_t(const IgemmParams& p, const HconvExtra& x, dim3 grid, size_t lds, const uint4* wf, int ntiles, void* stream,
                           const char* name, const char* who, double flops, double bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bconvu_kernel<NTAPS, U, MT, NT, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    HconvExtra xp = x;
    xp.n_tiles = (int)grid.x;
    // the planner's label names the tile plan ("bconv<1ph,256px,N32>"); this instance is the unrolled kernel for NTAPS taps
    name = intern_label("bconvu<%s,%s%dpx,N%d%s>", NTAPS == 25 ? "5x5" : "3x3", x.G > 1 ? "small-img," : "", 32 * MT * WAVES, 32 * NT,
                        x.ksplit > 1 ? ",ksplit" : "");
    static int wg_per_cu = -1;
    if (wg_per_cu < 0) { const char* e = getenv("MV3D_BC_PERSIST"); wg_per_cu = e ? atoi(e) : 3; }      // 3 where LDS allows (52 KB stride-2 tiles): +1 % on the step over 2
    dim3 pg = grid;
    if (wg_per_cu > 0) {
        const int by_lds = std::max(1, (int)((160 * 1024) / std::max<size_t>(lds, 1)));
        const int per_cu = std::min(wg_per_cu, by_lds);
        const int cap = std::max(1, 256 * per_cu / (int)(grid.y * grid.z));
        pg.x = std::min<int>((int)grid.x, cap);
    }
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        bconvu_kernel<NTAPS, U, MT, NT, WAVES><<<pg, WAVES * 64, lds, s>>>(p, xp, wf, ntiles);
        return launched(who);
    });
}

static int launch_bconv_cfg(const IgemmParams& p, const HconvExtra& x, int nph_fused, int MT, int NT, int WAVES, dim3 grid, size_t lds,
                            const uint4* wf, int ntiles, void* stream, const char* name, const char* who, double flops, double bytes) {
#define MV3D_BC(NPH_, MT_, NT_, W_) launch_bconv_t<NPH_, MT_, NT_, W_>(p, x, grid, lds, wf, ntiles, stream, name, who, flops, bytes)
    const int ntaps_all = p.tap_begin[p.so_h * p.so_w];
    if (WAVES == 8) return launch_cconv(p, x, wf, ntiles, stream, who, flops, bytes);      // pipelined kernel (cconv.hip); planned in hconv.hip
    if (nph_fused == 1 && !x.phase_split && p.so_h == 1 && p.so_w == 1 && (ntaps_all == 25 || ntaps_all == 9) && !(disabled_paths() & 8192)) {
#define MV3D_BCU(MT_, NT_, W_) (ntaps_all == 25 ? launch_bconvu_t<25, 5, MT_, NT_, W_>(p, x, grid, lds, wf, ntiles, stream, name, who, flops, bytes) \
                                                : launch_bconvu_t<9, 3, MT_, NT_, W_>(p, x, grid, lds, wf, ntiles, stream, name, who, flops, bytes))
        if (MT == 2 && NT == 1 && WAVES == 4) return MV3D_BCU(2, 1, 4);
        if (MT == 2 && NT == 2 && WAVES == 4) return MV3D_BCU(2, 2, 4);
        if (MT == 1 && NT == 2 && WAVES == 4) return MV3D_BCU(1, 2, 4);
        if (MT == 1 && NT == 1 && WAVES == 4) return MV3D_BCU(1, 1, 4);
        if (MT == 1 && NT == 1 && WAVES == 2) return MV3D_BCU(1, 1, 2);
        if (MT == 1 && NT == 2 && WAVES == 2) return MV3D_BCU(1, 2, 2);
#undef MV3D_BCU
    }
    if (nph_fused == 4) {
        if (MT == 1 && NT == 1 && WAVES == 4) return MV3D_BC(4, 1, 1, 4);
        if (MT == 1 && NT == 1 && WAVES == 2) return MV3D_BC(4, 1, 1, 2);
    } else {
        if (MT == 2 && NT == 1 && WAVES == 4) return MV3D_BC(1, 2, 1, 4);
        if (MT == 2 && NT == 2 && WAVES == 4) return MV3D_BC(1, 2, 2, 4);
        if (MT == 1 && NT == 2 && WAVES == 4) return MV3D_BC(1, 1, 2, 4);
        if (MT == 1 && NT == 1 && WAVES == 4) return MV3D_BC(1, 1, 1, 4);
        if (MT == 1 && NT == 1 && WAVES == 2) return MV3D_BC(1, 1, 1, 2);
        if (MT == 1 && NT == 2 && WAVES == 2) return MV3D_BC(1, 1, 2, 2);
    }
#undef MV3D_BC
    return fail(MV3D_E_UNSUPPORTED, "%s: no split-bf16 kernel for nph=%d MT=%d NT=%d waves=%d", who, nph_fused, MT, NT, WAVES);
}

const void* bconv_cache_lookup(const IgemmParams& p, int* ntiles_out);

int launch_bconv(const IgemmParams& p, const HconvExtra& x, int nph_fused, int MT, int NT, int WAVES, dim3 grid,
                 void* wfrag, void* stream, const char* name, const char* who, double flops, double bytes) {
    const int nph = p.so_h * p.so_w;
    const int ntaps = p.tap_begin[nph];
    const int chunks = x.chunks, ntiles = cdiv(p.Cc, 32 * NT) * NT;
    uint4* wf = reinterpret_cast<uint4*>(wfrag);
    int cached_tiles = 0;
    const void* cached = bconv_cache_lookup(p, &cached_tiles);
    if (cached && cached_tiles >= ntiles) {
        wf = reinterpret_cast<uint4*>(const_cast<void*>(cached));
        const size_t lds0 = (size_t)bconv_lds_bytes(x);
        return launch_bconv_cfg(p, x, nph_fused, MT, NT, WAVES, grid, lds0, wf, cached_tiles, stream, name, who, flops, bytes);
    }
    {
        const int64_t threads = (int64_t)ntaps * chunks * ntiles * 2 * 64;
        const int blocks = (int)cdiv64(threads, 256);
        const IgemmParams pc = p;
        int rc = dispatch(stream, OpInfo{"bconv_split_filter", 0.0, 2.0 * (double)ntaps * p.Ka * p.Cc * 4.0}, [=](hipStream_t s) {
            bconv_split_filter_kernel<<<blocks, 256, 0, s>>>(pc, wf, ntaps, chunks, ntiles);
            return launched("bconv_split_filter_kernel");
        });
        if (rc != MV3D_OK) return rc;
    }
    return launch_bconv_cfg(p, x, nph_fused, MT, NT, WAVES, grid, (size_t)bconv_lds_bytes(x), wf, ntiles, stream, name, who, flops, bytes);
}


// The prepared (fragment-ordered, split) copy of p's filter for kernels outside this file (sconv.hip): the bound copy of the
// filter cache, or -- for an unbound filter -- a copy split into `ws` by a launch dispatched here.  Returns null with *rc set
// when neither is possible (*rc = 1: the caller should fall back to another kernel).
const uint4* bconv_get_filter(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, int* ntiles_out, int* rc) {
    const int nph = p.so_h * p.so_w;
    const int ntaps = p.tap_begin[nph];
    const int chunks = cdiv(p.Ka, 32), ntiles = cdiv(p.Cc, 32);
    int cached_tiles = 0;
    const void* cached = bconv_cache_lookup(p, &cached_tiles);
    if (cached && cached_tiles >= ntiles) { *ntiles_out = cached_tiles; *rc = MV3D_OK; return reinterpret_cast<const uint4*>(cached); }
    const size_t need = bconv_filter_bytes(p, 1);
    if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 15)) { *rc = 1; return nullptr; }
    uint4* wf = reinterpret_cast<uint4*>(ws);
    const int64_t threads = (int64_t)ntaps * chunks * ntiles * 2 * 64;
    const int blocks = (int)cdiv64(threads, 256);
    const IgemmParams pc = p;
    *rc = dispatch(stream, OpInfo{"bconv_split_filter", 0.0, 2.0 * (double)ntaps * p.Ka * p.Cc * 4.0}, [=](hipStream_t s) {
        bconv_split_filter_kernel<<<blocks, 256, 0, s>>>(pc, wf, ntaps, chunks, ntiles);
        return launched("bconv_split_filter_kernel");
    });
    if (*rc != MV3D_OK) return nullptr;
    *ntiles_out = ntiles;
    return wf;
}

// ---- prepared-filter cache ------------------------------------------------------------------------
// A caller that knows a set of filters stays constant over several convolution calls (a training step:
// every filter is used by the forward pass and again, in the other orientation, by the backward-data pass)
// binds a caller-owned buffer per (filter, orientation); mv3d_filter_cache_refresh() then re-splits ALL bound
// filters in one launch and the convolution calls that find their filter in the cache skip their own split
// launch.  Unbound filters keep working (split per call into the workspace).
struct SplitJob {
    const float* Wt; uint4* Wf;
    int ntaps, chunks, ntiles, Ka, Cc, w_tap_stride, w_ks, w_ns;
    int first_block, nblocks;
    int16_t widx[36];
};

static std::mutex g_cache_mu;
static std::vector<SplitJob> g_jobs;
static const SplitJob* g_table_dev = nullptr;
static int g_table_jobs = 0, g_total_blocks = 0;

__global__ __launch_bounds__(256) void bconv_split_all_kernel(const SplitJob* __restrict__ jobs, int njobs) {
    // block -> job map behind the job array (one dependent load instead of a linear search over ~40 jobs)
    const int* block_job = reinterpret_cast<const int*>(jobs + njobs);
    const SplitJob& jb = jobs[block_job[blockIdx.x]];       // read in place: a local copy indexed by `t` would live in scratch
    const int64_t gid = (int64_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    const int lane = (int)(gid & 63);
    int64_t r = gid >> 6;
    const int s = (int)(r & 1); r >>= 1;
    const int nt = (int)(r % jb.ntiles); r /= jb.ntiles;
    const int cc = (int)(r % jb.chunks); r /= jb.chunks;
    const int t = (int)r;
    if (t >= jb.ntaps) return;
    const int li = lane & 31, lh = lane >> 5;
    const int col = nt * 32 + li;
    const int ch0 = cc * 32 + s * 16 + lh * 8;
    const float* wt = jb.Wt + (int64_t)jb.widx[t] * jb.w_tap_stride;
    float v[8];
    if (jb.w_ks == 1 && (jb.w_ns & 3) == 0 && ch0 + 8 <= jb.Ka && col < jb.Cc && ((reinterpret_cast<uintptr_t>(wt) & 15) == 0)) {
        // reduction index contiguous in memory (feature -> image direction): two 16-byte loads per lane
        const float4* src = reinterpret_cast<const float4*>(wt + (int64_t)col * jb.w_ns + ch0);
        const float4 u0 = src[0], u1 = src[1];
        v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = ch0 + e;
            v[e] = (col < jb.Cc && ch < jb.Ka) ? wt[(int64_t)ch * jb.w_ks + (int64_t)col * jb.w_ns] : 0.f;
        }
    }
    uint2 h0, l0, h1, l1;
    split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    uint4* dst = jb.Wf + ((((int64_t)t * jb.chunks + cc) * jb.ntiles + nt) * 4 + s * 2) * 64 + lane;
    dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

static bool job_matches(const SplitJob& j, const IgemmParams& p, int ntaps) {
    if (j.Wt != p.Wt || j.ntaps != ntaps || j.Ka != p.Ka || j.Cc != p.Cc || j.w_ks != p.w_ks || j.w_ns != p.w_ns ||
        j.w_tap_stride != p.w_tap_stride) return false;
    for (int t = 0; t < ntaps; ++t) if (j.widx[t] != p.taps[t].widx) return false;
    return true;
}

// prepared copy of p's filter, or null; *ntiles_out = column-tile count of its layout
const void* bconv_cache_lookup(const IgemmParams& p, int* ntiles_out) {
    const int ntaps = p.tap_begin[p.so_h * p.so_w];
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (!g_table_dev) return nullptr;                 // never committed: the buffers hold nothing yet
    for (int i = 0; i < g_table_jobs && i < (int)g_jobs.size(); ++i)
        if (job_matches(g_jobs[i], p, ntaps)) { *ntiles_out = g_jobs[i].ntiles; return g_jobs[i].Wf; }
    return nullptr;
}

size_t bconv_prepared_bytes(const IgemmParams& p) {
    if (p.fold || p.Ka % 16 != 0 || p.Cc < 16) return 0;
    const int ntaps = p.tap_begin[p.so_h * p.so_w];
    if (ntaps < 2 || ntaps > 36) return 0;
    return bconv_filter_bytes(p, 2);
}

int bconv_cache_bind(const IgemmParams& p, void* prepared, size_t bytes) {
    const size_t need = bconv_prepared_bytes(p);
    if (need == 0) return 1;                          // this operation never uses a prepared filter
    if (!prepared || bytes < need || (reinterpret_cast<uintptr_t>(prepared) & 15))
        return fail(MV3D_E_INVAL, "mv3d_filter_cache_bind: prepared buffer null, misaligned or %zu < %zu bytes", bytes, need);
    const int ntaps = p.tap_begin[p.so_h * p.so_w];
    SplitJob j = {};
    j.Wt = p.Wt; j.Wf = reinterpret_cast<uint4*>(prepared);
    j.ntaps = ntaps; j.chunks = cdiv(p.Ka, 32); j.ntiles = cdiv(p.Cc, 64) * 2; j.Ka = p.Ka; j.Cc = p.Cc;
    j.w_tap_stride = p.w_tap_stride; j.w_ks = p.w_ks; j.w_ns = p.w_ns;
    for (int t = 0; t < ntaps; ++t) j.widx[t] = p.taps[t].widx;
    j.nblocks = (int)cdiv64((int64_t)ntaps * j.chunks * j.ntiles * 2 * 64, 256);
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto& e : g_jobs)
        if (job_matches(e, p, ntaps)) { e.Wf = j.Wf; g_table_dev = nullptr; return MV3D_OK; }
    g_jobs.push_back(j);
    g_table_dev = nullptr;                            // table must be committed again
    return MV3D_OK;
}

}  // namespace mv3d

using namespace mv3d;

extern "C" {

static size_t cache_table_bytes_locked() {
    size_t blocks = 0;
    for (const auto& j : g_jobs) blocks += (size_t)j.nblocks;
    return g_jobs.size() * sizeof(SplitJob) + blocks * sizeof(int);
}

size_t mv3d_filter_cache_table_bytes(void) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    return cache_table_bytes_locked();
}

int mv3d_filter_cache_commit(void* table_dev, size_t table_bytes, void* stream) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    const size_t need = cache_table_bytes_locked();
    if (g_jobs.empty()) { g_table_dev = nullptr; g_table_jobs = 0; g_total_blocks = 0; return MV3D_OK; }
    if (!table_dev || table_bytes < need) return fail(MV3D_E_INVAL, "mv3d_filter_cache_commit: table %zu < %zu bytes", table_bytes, need);
    int first = 0;
    for (auto& j : g_jobs) { j.first_block = first; first += j.nblocks; }
    std::vector<unsigned char> host(need);
    memcpy(host.data(), g_jobs.data(), g_jobs.size() * sizeof(SplitJob));
    int* bj = reinterpret_cast<int*>(host.data() + g_jobs.size() * sizeof(SplitJob));
    for (size_t i = 0; i < g_jobs.size(); ++i)
        for (int b = 0; b < g_jobs[i].nblocks; ++b) bj[g_jobs[i].first_block + b] = (int)i;
    hipError_t e = hipMemcpyAsync(table_dev, host.data(), need, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));      // g_jobs may be edited right after
    if (e != hipSuccess) return fail(MV3D_E_HIP, "mv3d_filter_cache_commit: %s", hipGetErrorString(e));
    g_table_dev = reinterpret_cast<const SplitJob*>(table_dev);
    g_table_jobs = (int)g_jobs.size();
    g_total_blocks = first;
    return MV3D_OK;
}

int mv3d_filter_cache_refresh(void* stream) {
    const SplitJob* table; int njobs, blocks; double bytes = 0.0;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        table = g_table_dev; njobs = g_table_jobs; blocks = g_total_blocks;
        for (int i = 0; i < njobs; ++i) bytes += 8.0 * g_jobs[i].ntaps * (double)g_jobs[i].Ka * g_jobs[i].Cc;
    }
    if (!table || njobs == 0) return MV3D_OK;
    return dispatch(stream, OpInfo{"bconv_split_all", 0.0, bytes}, [=](hipStream_t s) {
        bconv_split_all_kernel<<<blocks, 256, 0, s>>>(table, njobs);
        return launched("bconv_split_all_kernel");
    });
}

int mv3d_filter_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_jobs.clear(); g_table_dev = nullptr; g_table_jobs = 0; g_total_blocks = 0;
    return MV3D_OK;
}

}  // extern "C"
