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

struct HconvExtra {
    int TH, TW, tw_shift;
    int tiles_h, tiles_w;
    int HR, HC;
    int dh_min, dw_min;
    int chunks, ntaps_total;
    int dbg;   // MV3D_DBG diagnostics: 1 = no halo loads, 8 = skip the tap loop
};

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
    const int n = b / x.tiles_h;
    const int oh0 = th_i * x.TH, ow0 = tw_i * x.TW;
    const int n0 = blockIdx.y * 32 * NT;

    int lane_base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pidx = (wave * MT + m) * 32 + li;
        const int tr = pidx >> x.tw_shift, tc = pidx & (x.TW - 1);
        lane_base[m] = ((tr * p.sa_h) * x.HC + tc * p.sa_w) * CS + lh * 16;
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
    const int total_seq = x.chunks * x.ntaps_total;

    // Filter fetch for one (chunk, tap): unconditional loads from clamped (always valid) addresses and
    // NO masking -- a conditional load makes hipcc branch around every load, and a select on the loaded
    // value makes it wait for the load right away, either of which kills the one-tap-ahead prefetch
    // (cdna_hip_programming.md, trap (c)).  Clamping is enough: output columns >= Cc are never
    // stored, and channels >= Ka multiply halo entries that were staged as zeros.
    auto load_b = [&](Frags<MT, NT>& f, int seq) {
        const int cc = seq / x.ntaps_total;
        const int t = seq - cc * x.ntaps_total;
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
    for (int cc = 0; cc < x.chunks; ++cc) {
        if (cc) __syncthreads();
        // halo staging in batches of 8 independent 16-byte loads per thread (all in flight
        // together), then the LDS stores; out-of-image pixels load a valid dummy address and are zeroed
        for (int base = 0; base < halo_pix * 8; base += NTHR * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NTHR + tid;
                const int pix = idx >> 3, c4 = idx & 7;
                const int hr = pix / x.HC, hc = pix - hr * x.HC;
                const int ih = ih0 + hr, iw = iw0 + hc;
                const int ch = cc * 32 + c4 * 4;
                const bool ok = !(x.dbg & 1) && idx < halo_pix * 8 && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa && ch < p.Ka;
                const float* src = ok ? p.A + (int64_t)((n * p.Ha + ih) * p.Wa + iw) * p.a_ld + ch : p.A;
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
        load_a(f1, 0);
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            const int tb = p.tap_begin[ph], te = (x.dbg & 8) ? p.tap_begin[ph] : p.tap_begin[ph + 1];
            for (int t = tb; t < te; ++t, ++seq) {
#pragma unroll
                for (int kp = 0; kp < 16; ++kp) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) f0.a[m][kp] = f1.a[m][kp];
#pragma unroll
                    for (int y = 0; y < NT; ++y) f0.b[kp][y] = f1.b[kp][y];
                }
                if (seq + 1 < total_seq && !(x.dbg & 2)) load_b(f1, seq + 1);
                if (t + 1 < x.ntaps_total && !(x.dbg & 4)) load_a(f1, t + 1);
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
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int phh = ph / p.so_w, phw = ph % p.so_w;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int y = 0; y < NT; ++y) {
                const int col = n0 + y * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = (wave * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int ohp = oh0 + (q >> x.tw_shift), owp = ow0 + (q & (x.TW - 1));
                    if (ohp < Hp && owp < Wp && col < p.Cc) {
                        const int64_t pix = (int64_t)(n * p.Hc + ohp * p.so_h + phh) * p.Wc + owp * p.so_w + phw;
                        p.Out[pix * p.c_ld + col] = epilogue_value(p, acc[ph][m][y][r], pix, col);
                    }
                }
            }
    }
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
static bool pick_tile(const IgemmParams& p, int PIX, int Hp, int Wp, int dh_span, int dw_span, size_t lds_cap, HconvExtra* out) {
    int64_t best_cost = -1;
    for (int sh = 3; sh <= 6; ++sh) {
        const int TW = 1 << sh, TH = PIX / TW;
        if (TH < 1 || TH > Hp * 2 || TW > Wp * 2) continue;
        const int th = cdiv(Hp, TH), tw = cdiv(Wp, TW);
        const int HR = (TH - 1) * p.sa_h + dh_span, HC = (TW - 1) * p.sa_w + dw_span;
        const size_t lds = (size_t)HR * HC * 33 * sizeof(float);
        if (lds > lds_cap) continue;
        const int64_t cost = (int64_t)th * tw * (PIX * 64 + HR * HC);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            out->TH = TH; out->TW = TW; out->tw_shift = sh; out->tiles_h = th; out->tiles_w = tw; out->HR = HR; out->HC = HC;
        }
    }
    return best_cost >= 0;
}

int try_hconv(const IgemmParams& p, void* stream, const char* who, double flops, double bytes) {
    const int nph = p.so_h * p.so_w;
    if (disabled_paths() & 1) return 1;
    if (p.fold || (nph != 1 && nph != 4)) return 1;
    if (p.Ka % 16 != 0 || p.a_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15)) return 1;
    const bool kmajor = (p.w_ns == 1);
    if (!kmajor && ((p.w_ns % 4) != 0 || (reinterpret_cast<uintptr_t>(p.Wt) & 15))) return 1;
    const int Hp = p.Hp[0], Wp = p.Wp[0];
    if (nph == 4 && (p.Hp[1] != Hp || p.Wp[1] != Wp)) return 1;
    if (Hp * Wp < 64 || Wp < 8) return 1;
    if (p.Cc < 16) return 1;
    const int ntaps = p.tap_begin[nph];
    if (ntaps < 2) return 1;                      // 1x1 (fc layers): no halo reuse to exploit
    for (int ph = 0; ph < nph; ++ph) if (p.tap_begin[ph + 1] == p.tap_begin[ph]) return 1;
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int t = 0; t < ntaps; ++t) {
        dh_min = std::min<int>(dh_min, p.taps[t].dh); dh_max = std::max<int>(dh_max, p.taps[t].dh);
        dw_min = std::min<int>(dw_min, p.taps[t].dw); dw_max = std::max<int>(dw_max, p.taps[t].dw);
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
    {   // diagnostics: MV3D_HCONV_SKIP=i falls back to igemm for the i-th eligible call only
        static int counter = 0;
        const char* e = getenv("MV3D_HCONV_SKIP");
        const int idx = counter++;
        if (e && atoi(e) == idx) { fprintf(stderr, "[mv3d] hconv call %d skipped: %s Ha=%d Ca=%d Cc=%d nph=%d\n", idx, who, p.Ha, p.Ca, p.Cc, nph); return 1; }
    }
    best.dh_min = dh_min; best.dw_min = dw_min;
    best.chunks = cdiv(p.Ka, 32);
    best.ntaps_total = ntaps;
    { const char* e = getenv("MV3D_DBG"); best.dbg = e ? atoi(e) : 0; }
    const size_t lds = (size_t)best.HR * best.HC * 33 * sizeof(float);
    dim3 grid(p.N * best.tiles_h * best.tiles_w, cdiv(p.Cc, 32 * NT), 1);
    IgemmParams q = p;
    q.ksplit = 1;
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
