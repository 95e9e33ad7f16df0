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
    auto load_a = [&](BFrags<MT, NT>& f, int t) {
        t = t < tap_hi ? t : tap_hi - 1;
        const IgemmTap tap = p.taps[t];
        const int off = (tap.dh - x.dh_min) * x.row_bytes + (tap.dw - x.dw_min) * BC_PIXB;
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
        for (int base = 0; base < halo_pix * 8; base += NTHR * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NTHR + tid;
                const int pix = idx >> 3, c4 = idx & 7;
                const int hrv = pix / x.HC, hc = pix - hrv * x.HC;
                const int g = hrv / x.HRi, hr = hrv - g * x.HRi;
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
                    const int pix = idx >> 3, c4 = idx & 7;
                    const int hrv = pix / x.HC, hc = pix - hrv * x.HC;
                    unsigned char* d = halo + hrv * x.row_bytes + hc * BC_PIXB + c4 * 8;
                    uint2 hi, lo;
                    split4(v[u], hi, lo);
                    *reinterpret_cast<uint2*>(d) = hi;
                    *reinterpret_cast<uint2*>(d + 64) = lo;
                }
            }
        }
        __syncthreads();
        load_a(f0, tap_lo);
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
            for (; t + 1 < te; t += 2, seq += 2) {
                if (!(x.dbg & 2)) load_b(f1, seq + 1);
                if (!(x.dbg & 4)) load_a(f1, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(f0);
                if (!(x.dbg & 2)) load_b(f0, seq + 2);
                if (!(x.dbg & 4)) load_a(f0, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(f1);
            }
            if (t < te) {                  // odd tap count: one more, then hand the look-ahead set over
                load_b(f1, seq + 1); load_a(f1, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(f0);
                f0 = f1;
                ++seq;
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

// LDS row stride: 16 consecutive lanes of a ds_read_b128 group must land on 16 distinct 4-bank slots.
// With 144-byte pixels that holds inside a tile row; across tile rows it needs the row stride
// = 0 (mod 256 B) for 16-pixel rows and = 128 (mod 256 B) for 8-pixel rows (MI355X_MICROARCH.md, LDS lane groups).
void bconv_set_rows(HconvExtra* x) {
    int rb = x->HC * BC_PIXB;
    if (x->TW == 16) rb = (rb + 255) & ~255;
    else if (x->TW == 8) rb = ((rb + 127) & ~255) + 128;
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
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        bconv_kernel<NPH, MT, NT, WAVES><<<grid, WAVES * 64, lds, s>>>(p, x, wf, ntiles);
        return launched(who);
    });
}

int launch_bconv(const IgemmParams& p, const HconvExtra& x, int nph_fused, int MT, int NT, int WAVES, dim3 grid,
                 void* wfrag, void* stream, const char* name, const char* who, double flops, double bytes) {
    const int nph = p.so_h * p.so_w;
    const int ntaps = p.tap_begin[nph];
    const int chunks = x.chunks, ntiles = cdiv(p.Cc, 32 * NT) * NT;
    uint4* wf = reinterpret_cast<uint4*>(wfrag);
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
    const size_t lds = (size_t)bconv_lds_bytes(x);
#define MV3D_BC(NPH_, MT_, NT_, W_) launch_bconv_t<NPH_, MT_, NT_, W_>(p, x, grid, lds, wf, ntiles, stream, name, who, flops, bytes)
    if (nph_fused == 4) {
        if (MT == 1 && NT == 1 && WAVES == 4) return MV3D_BC(4, 1, 1, 4);
        if (MT == 1 && NT == 1 && WAVES == 2) return MV3D_BC(4, 1, 1, 2);
    } else {
        if (MT == 2 && NT == 1 && WAVES == 4) return MV3D_BC(1, 2, 1, 4);
        if (MT == 2 && NT == 2 && WAVES == 4) return MV3D_BC(1, 2, 2, 4);
        if (MT == 1 && NT == 2 && WAVES == 4) return MV3D_BC(1, 1, 2, 4);
        if (MT == 1 && NT == 1 && WAVES == 4) return MV3D_BC(1, 1, 1, 4);
        if (MT == 1 && NT == 1 && WAVES == 2) return MV3D_BC(1, 1, 1, 2);
    }
#undef MV3D_BC
    return fail(MV3D_E_UNSUPPORTED, "%s: no split-bf16 kernel for nph=%d MT=%d NT=%d waves=%d", who, nph_fused, MT, NT, WAVES);
}

}  // namespace mv3d
