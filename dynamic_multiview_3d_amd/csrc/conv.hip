// conv2d / conv2d_transpose forward + backward for gfx950 as three implicit-GEMM primitives
// over (image side [N,H,W,C], feature side [N,Ho,Wo,K], filter [kh,kw,C,K]):
//
//   img2feat : feat[n,ho,wo,k]  = sum img[n,ho*s+p-pt,wo*s+q-pl,c] * f[p,q,c,k]   conv fwd, deconv dgrad
//   feat2img : img[n,u,v,c]     = sum feat[n,i,j,k] * f[u-i*s+pt, v-j*s+pl, c,k]  conv dgrad, deconv fwd
//   filtgrad : df[p,q,c,k]      = sum img[n,ho*s+p-pt,...,c] * feat[n,ho,wo,k]    conv/deconv wgrad
//
// TF-1.3 semantics being restated: tf.nn.conv2d SAME (tf_utils.py:81) and
// tf.nn.conv2d_transpose == Conv2DBackpropInput (tf_utils.py:96); SURVEY Appendix A.1/A.2.
//
// img2feat/feat2img share ONE kernel (igemm_kernel): the output pixels of a launch are split into
// stride phases; inside a phase every output pixel uses the same tap list
// (dh, dw, filter tap), and the input pixel is (oh'*sa + dh, ow'*sa + dw).  A strided transposed
// conv therefore becomes s*s dense sub-convolutions with no multiplications by structural zeros.
// Tiles are staged global -> registers -> LDS (next tile's loads in flight during the MFMAs) and
// multiplied with v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
#include "conv_common.h"
#include "reduce_common.h"
#include <algorithm>

namespace mv3d {

// WM waves along M (4/WM along N); each wave owns a (32*MT) x (32*NT) output tile.
template <int WM, int MT, int NT, bool VEC, bool BKMAJOR>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    constexpr int BK = 32;
    constexpr int WN = 4 / WM;
    constexpr int BM = WM * 32 * MT;
    constexpr int BN = WN * 32 * NT;
    constexpr int LDA = BK + 1;
    constexpr int LDB = BN + 1;
    constexpr int A_PASSES = VEC ? BM / 32 : BM / 8;
    constexpr int B_LOADS = BK * BN / 256;

    __shared__ float As[BM * LDA];
    __shared__ float Bs[BK * LDB];
    __shared__ int s_rowbase[BM], s_ih0[BM], s_iw0[BM], s_outpix[BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int ks = blockIdx.z % p.ksplit;
    const int ph = blockIdx.z / p.ksplit;
    const int phh = ph / p.so_w, phw = ph % p.so_w;
    const int Hp = p.Hp[phh], Wp = p.Wp[phw];
    const int Mp = p.N * Hp * Wp;
    const int m0 = blockIdx.x * BM;
    if (m0 >= Mp) return;
    const int n0 = blockIdx.y * BN;

    for (int r = tid; r < BM; r += 256) {
        int m = m0 + r;
        if (m < Mp) {
            int n = m / (Hp * Wp);
            int rem = m - n * (Hp * Wp);
            int ohp = rem / Wp, owp = rem - ohp * Wp;
            s_rowbase[r] = n * p.Ha * p.Wa;
            s_ih0[r] = ohp * p.sa_h;
            s_iw0[r] = owp * p.sa_w;
            s_outpix[r] = (n * p.Hc + ohp * p.so_h + phh) * p.Wc + owp * p.so_w + phw;
        } else {
            s_rowbase[r] = 0; s_ih0[r] = -(1 << 28); s_iw0[r] = 0; s_outpix[r] = -1;
        }
    }
    __syncthreads();

    const int tap0 = p.tap_begin[ph];
    const int ntap = p.tap_begin[ph + 1] - tap0;
    const int kchunks = (p.Ka + BK - 1) / BK;
    const int iters = ntap * kchunks;
    const int it_begin = (int)((int64_t)iters * ks / p.ksplit);
    const int it_end = (int)((int64_t)iters * (ks + 1) / p.ksplit);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    float4 ra4[VEC ? A_PASSES : 1];
    float ra1[VEC ? 1 : A_PASSES];
    float rb[B_LOADS];

    auto load_tile = [&](int it) {
        const int t = it / kchunks;
        const int c0 = (it - t * kchunks) * BK;
        const IgemmTap tap = p.taps[tap0 + t];
        if constexpr (VEC) {
            const int c = c0 + (tid & 7) * 4;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const int r = ps * 32 + (tid >> 3);
                const int ih = s_ih0[r] + tap.dh, iw = s_iw0[r] + tap.dw;
                const bool ok = (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa && c < p.Ka;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4*>(p.A + (int64_t)(s_rowbase[r] + ih * p.Wa + iw) * p.a_ld + c);
                ra4[ps] = v;
            }
        } else {
            const int e = c0 + (tid & 31);
            const int q = p.fold ? e / p.Ca : 0;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const int r = ps * 8 + (tid >> 5);
                const int ih = s_ih0[r] + tap.dh, iw0 = s_iw0[r] + tap.dw;
                const bool ok = (unsigned)ih < (unsigned)p.Ha && (unsigned)(iw0 + q) < (unsigned)p.Wa && e < p.Ka;
                float v = 0.f;
                if (ok) v = p.A[(int64_t)(s_rowbase[r] + ih * p.Wa + iw0) * p.a_ld + e];
                ra1[ps] = v;
            }
        }
        const float* wt = p.Wt + (int64_t)tap.widx * p.w_tap_stride;
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            int kk, nn;
            if constexpr (BKMAJOR) { nn = tid % BN; kk = tid / BN + (256 / BN) * j; }
            else { kk = tid & 31; nn = (tid >> 5) + 8 * j; }
            const bool ok = (c0 + kk) < p.Ka && (n0 + nn) < p.Cc;
            rb[j] = ok ? wt[(int64_t)(c0 + kk) * p.w_ks + (int64_t)(n0 + nn) * p.w_ns] : 0.f;
        }
    };

    auto store_tile = [&]() {
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                float* d = &As[(ps * 32 + (tid >> 3)) * LDA + (tid & 7) * 4];
                d[0] = ra4[ps].x; d[1] = ra4[ps].y; d[2] = ra4[ps].z; d[3] = ra4[ps].w;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) As[(ps * 8 + (tid >> 5)) * LDA + (tid & 31)] = ra1[ps];
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            int kk, nn;
            if constexpr (BKMAJOR) { nn = tid % BN; kk = tid / BN + (256 / BN) * j; }
            else { kk = tid & 31; nn = (tid >> 5) + 8 * j; }
            Bs[kk * LDB + nn] = rb[j];
        }
    };

    if (it_begin < it_end) load_tile(it_begin);
    for (int it = it_begin; it < it_end; ++it) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (it + 1 < it_end) load_tile(it + 1);
        const float* a_base = &As[(wm * 32 * MT + li) * LDA + lh];
        const float* b_base = &Bs[lh * LDB + wn * 32 * NT + li];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[MT], b[NT];
#pragma unroll
            for (int x = 0; x < MT; ++x) a[x] = a_base[x * 32 * LDA + 2 * kp];
#pragma unroll
            for (int y = 0; y < NT; ++y) b[y] = b_base[2 * kp * LDB + y * 32];
#pragma unroll
            for (int x = 0; x < MT; ++x)
#pragma unroll
                for (int y = 0; y < NT; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int64_t npix_total = (int64_t)p.N * p.Hc * p.Wc;
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = n0 + wn * 32 * NT + y * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 * MT + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int pix = s_outpix[row];
                if (pix >= 0 && col < p.Cc) {
                    if (p.ksplit > 1) p.Part[((int64_t)ks * npix_total + pix) * p.Cc + col] = acc[x][y][r];
                    else p.Out[(int64_t)pix * p.c_ld + col] = epilogue_value(p, acc[x][y][r], pix, col);
                }
            }
        }
}

// split-K tail: sum the partial slabs (in slab order: the result does not depend on the launch shape), apply the
// epilogue, store with the output's pixel stride.  Eight slab rows are requested before the first add: with one load per
// trip of a rolled loop the sum costs ksplit serial memory round trips.  VEC = 4: four consecutive channels per thread.
template <int VEC>
__global__ __launch_bounds__(256) void igemm_splitk_epilogue(const IgemmParams p) {
    const int64_t npix = (int64_t)p.N * p.Hc * p.Wc;
    const int64_t total = npix * p.Cc;
    const int64_t nvec = total / VEC;
    for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nvec; iv += (int64_t)gridDim.x * 256) {
        const int64_t idx = iv * VEC;
        const int64_t pix = idx / p.Cc;
        const int col = (int)(idx - pix * p.Cc);
        float s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.f;
        const float* src = p.Part + idx;
        int k = 0;
        for (; k + 8 <= p.ksplit; k += 8) {
            float v[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)(k + u) * total);
                    v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
                } else {
                    v[u][0] = src[(int64_t)(k + u) * total];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[e] += v[u][e];
        }
        for (; k < p.ksplit; ++k) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)k * total);
                s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
            } else {
                s[0] += src[(int64_t)k * total];
            }
        }
        if constexpr (VEC == 4) {
            float4 o;
            o.x = epilogue_value(p, s[0], pix, col); o.y = epilogue_value(p, s[1], pix, col + 1);
            o.z = epilogue_value(p, s[2], pix, col + 2); o.w = epilogue_value(p, s[3], pix, col + 3);
            *reinterpret_cast<float4*>(p.Out + pix * p.c_ld + col) = o;
        } else {
            p.Out[pix * p.c_ld + col] = epilogue_value(p, s[0], pix, col);
        }
    }
}

static void launch_splitk_epilogue(const IgemmParams& e, int blocks, hipStream_t s) {
    const int64_t total = (int64_t)e.N * e.Hc * e.Wc * e.Cc;
    const bool vec = e.Cc % 4 == 0 && e.c_ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(e.Out) | reinterpret_cast<uintptr_t>(e.Part)) & 15) == 0;
    if (vec) {
        const int b4 = (int)std::min<int64_t>(cdiv64(total / 4, 256), 4096);
        igemm_splitk_epilogue<4><<<std::max(b4, 1), 256, 0, s>>>(e);
    } else {
        igemm_splitk_epilogue<1><<<blocks, 256, 0, s>>>(e);
    }
}

// Stride-2 transposed conv with a thin image side (flow field / rgb / depth / mask heads: 32 -> 1..4
// channels, tf_utils.py:96 via appearance_flow_model.py:125, main_model.py:77).  HBM/L2-bound
// (AI ~ 40 flop/B).  One thread owns one position of the input grid and produces the whole S x S
// block of output pixels from its 3x3 (k=5) / 2x2 (k=3) neighbourhood: every input pixel is loaded
// once per thread and feeds all phases that use it (tap geometry resolved at compile time), the
// filters sit in LDS and are read as broadcasts.
template <int KS, int CC>
__global__ __launch_bounds__(256) void thin_deconv_s2_kernel(const IgemmParams p) {
    constexpr int S = 2;
    constexpr int PT = (KS - S) / 2;                          // TF SAME pad_before for even sizes
    constexpr int DMIN = -((KS - 1 - PT) / S), DMAX = (S - 1 + PT) / S;
    extern __shared__ __attribute__((aligned(16))) float s_w[];      // [KS*KS][CC][Ka]
    const int Ka = p.Ka;
    for (int i = threadIdx.x; i < KS * KS * CC * Ka; i += 256) {
        const int t = i / (CC * Ka);
        const int rem = i - t * CC * Ka;
        const int c = rem / Ka, k = rem - c * Ka;
        s_w[i] = p.Wt[(int64_t)t * p.w_tap_stride + (int64_t)c * p.w_ns + (int64_t)k * p.w_ks];
    }
    __syncthreads();
    const int Hp = p.Hp[0], Wp = p.Wp[0];
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= p.N * Hp * Wp) return;
    const int n = m / (Hp * Wp);
    const int rem = m - n * (Hp * Wp);
    const int up = rem / Wp, vp = rem - up * Wp;
    float acc[S * S][CC];
#pragma unroll
    for (int a = 0; a < S * S; ++a)
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[a][c] = 0.f;
#pragma unroll
    for (int dh = DMIN; dh <= DMAX; ++dh)
#pragma unroll
        for (int dw = DMIN; dw <= DMAX; ++dw) {
            const int ih = up + dh, iw = vp + dw;
            if ((unsigned)ih >= (unsigned)p.Ha || (unsigned)iw >= (unsigned)p.Wa) continue;
            const float* src = p.A + (int64_t)((n * p.Ha + ih) * p.Wa + iw) * p.a_ld;
            for (int k = 0; k < Ka; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + k);
#pragma unroll
                for (int phh = 0; phh < S; ++phh)
#pragma unroll
                    for (int phw = 0; phw < S; ++phw) {
                        // output row u = S*up + phh reads input row i = up + dh through filter row P = phh + PT - S*dh
                        constexpr int dummy = 0; (void)dummy;
                        const int P = phh + PT - S * dh, Q = phw + PT - S * dw;
                        if (P < 0 || P >= KS || Q < 0 || Q >= KS) continue;      // folds at compile time (all constants)
                        const float* w = s_w + (P * KS + Q) * CC * Ka + k;
#pragma unroll
                        for (int c = 0; c < CC; ++c) {
                            const float4 wv = *reinterpret_cast<const float4*>(w + c * Ka);
                            float t = acc[phh * S + phw][c];
                            t = fmaf(v.x, wv.x, t); t = fmaf(v.y, wv.y, t); t = fmaf(v.z, wv.z, t); t = fmaf(v.w, wv.w, t);
                            acc[phh * S + phw][c] = t;
                        }
                    }
            }
        }
#pragma unroll
    for (int phh = 0; phh < S; ++phh)
#pragma unroll
        for (int phw = 0; phw < S; ++phw) {
            const int64_t pix = (int64_t)(n * p.Hc + up * S + phh) * p.Wc + vp * S + phw;
#pragma unroll
            for (int c = 0; c < CC; ++c) p.Out[pix * p.c_ld + c] = epilogue_value(p, acc[phh * S + phw][c], pix, c);
        }
}

// Tiled form for Ka = 32 (the flow / rgb / depth / mask heads): in the kernel above a lane walks its own 128-byte
// channel row, so every load instruction touches 64 different cache lines and the 3 x 3 neighbourhoods are fetched
// through L1 again and again (47 us for 42 MB).  Here a workgroup owns 16 x 16 positions of the input grid of one
// image: the 18 x 18 x 32 halo is staged once with coalesced 16-byte loads (144-byte pixel pitch: ds_read_b128 of 16
// consecutive pixels covers all banks), the filters sit next to it, and a thread computes its 2 x 2 x CC outputs from
// LDS with packed FMAs (two partial sums per output, combined at the end).
typedef float tdf2 __attribute__((ext_vector_type(2)));

template <int KS, int CC, bool SW>
__global__ __launch_bounds__(256) void thin_deconv_s2_tile_kernel(const IgemmParams p, int tiles_h, int tiles_w) {
    constexpr int S = 2, KA = 32, PITCH = KA + 4;
    constexpr int PT = (KS - S) / 2;
    constexpr int DMIN = -((KS - 1 - PT) / S), DMAX = (S - 1 + PT) / S;
    constexpr int HR = 16 + DMAX - DMIN, HPIX = HR * HR;
    constexpr int NLOAD = (HPIX * (KA / 4) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_w = s_mem;                                      // [KS*KS][CC][KA]  (unused when SW)
    float* s_a = s_mem + (SW ? 0 : KS * KS * CC * KA);       // [HPIX][PITCH]
    const float* __restrict__ wt_g = p.Wt;
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h;
    const int n = b / tiles_h;
    const int u0 = th * 16, v0 = tw * 16;
    // halo: every load is issued before the first LDS write (clamped addresses, zero outside the image)
    {
        float4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + 256 * i;
            const int pix = min(idx >> 3, HPIX - 1), c4 = idx & 7;
            const int hr = pix / HR, hc = pix - hr * HR;
            const int ih = u0 + DMIN + hr, iw = v0 + DMIN + hc;
            const bool ok = (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
            const int ihc = min(max(ih, 0), p.Ha - 1), iwc = min(max(iw, 0), p.Wa - 1);
            const float4 t = *reinterpret_cast<const float4*>(p.A + (int64_t)((n * p.Ha + ihc) * p.Wa + iwc) * p.a_ld + c4 * 4);
            v[i] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (!SW)
            for (int i = tid; i < KS * KS * CC * KA; i += 256) {
                const int t = i / (CC * KA);
                const int rem = i - t * CC * KA;
                const int c = rem / KA, k = rem - c * KA;
                s_w[i] = p.Wt[(int64_t)t * p.w_tap_stride + (int64_t)c * p.w_ns + (int64_t)k * p.w_ks];
            }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + 256 * i;
            if (idx < HPIX * 8) *reinterpret_cast<float4*>(s_a + (idx >> 3) * PITCH + (idx & 7) * 4) = v[i];
        }
    }
    __syncthreads();
    const int tu = tid >> 4, tv = tid & 15;
    const int up = u0 + tu, vp = v0 + tv;
    tdf2 acc[S * S][CC];
#pragma unroll
    for (int a = 0; a < S * S; ++a)
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[a][c] = (tdf2){0.f, 0.f};
#pragma unroll
    for (int dh = DMIN; dh <= DMAX; ++dh)
#pragma unroll
        for (int dw = DMIN; dw <= DMAX; ++dw) {
            const float* src = s_a + ((tu + dh - DMIN) * HR + (tv + dw - DMIN)) * PITCH;
            // two 4-channel groups per trip and a scheduling fence between trips: fully unrolled, hipcc hoists all 472
            // LDS reads of the thread to the top and spills 5 KB of them to scratch
#pragma unroll 2
            for (int k = 0; k < KA; k += 4) {
                __builtin_amdgcn_sched_barrier(0);
                const float4 v = *reinterpret_cast<const float4*>(src + k);
                const tdf2 vlo = {v.x, v.y}, vhi = {v.z, v.w};
#pragma unroll
                for (int phh = 0; phh < S; ++phh)
#pragma unroll
                    for (int phw = 0; phw < S; ++phw) {
                        const int P = phh + PT - S * dh, Q = phw + PT - S * dw;
                        if (P < 0 || P >= KS || Q < 0 || Q >= KS) continue;      // folds at compile time
                        const float* w = (SW ? wt_g + (int64_t)(P * KS + Q) * p.w_tap_stride : s_w + (P * KS + Q) * CC * KA) + k;
#pragma unroll
                        for (int c = 0; c < CC; ++c) {
                            // SW: the filter element is the same for every lane -> scalar loads, SGPR operands, no LDS traffic
                            const float4 wv = *reinterpret_cast<const float4*>(w + (SW ? (int64_t)c * p.w_ns : c * KA));
                            const tdf2 wlo = {wv.x, wv.y}, whi = {wv.z, wv.w};
                            acc[phh * S + phw][c] = __builtin_elementwise_fma(vlo, wlo, acc[phh * S + phw][c]);
                            acc[phh * S + phw][c] = __builtin_elementwise_fma(vhi, whi, acc[phh * S + phw][c]);
                        }
                    }
            }
        }
    if (up >= p.Ha || vp >= p.Wa) return;
#pragma unroll
    for (int phh = 0; phh < S; ++phh)
#pragma unroll
        for (int phw = 0; phw < S; ++phw) {
            const int64_t pix = (int64_t)(n * p.Hc + up * S + phh) * p.Wc + vp * S + phw;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const tdf2 a = acc[phh * S + phw][c];
                p.Out[pix * p.c_ld + c] = epilogue_value(p, a[0] + a[1], pix, c);
            }
        }
}

// Convolution with a tiny image side (C <= 4: the RGB / depth / mask input layers e0, and the input
// gradient of the flow / rgb heads): the filter row (kw*C <= 20 contiguous floats of an NHWC row) is
// folded into the GEMM reduction.  One wave = 32 consecutive output pixels x 32 output channels;
// both MFMA operands come straight from global memory (A: 4 bytes per lane at a 4*s*C-byte pitch,
// B: 128-byte filter rows, L1-resident), no LDS.  HBM-bound: the layer moves ~46 MB for 1.3 GFLOP.
struct SmallCParams {
    const float* X; const float* Wt; float* Y;
    int N, H, W, C, Ho, Wo, K, y_ld;
    int kh, kw, sh, sw, pt, pl;
    int wtiles;                       // 32-pixel groups per output row
    IgemmParams ep;                   // epilogue fields only
};

template <int NT>
__global__ __launch_bounds__(256) void smallc_img2feat_kernel(const SmallCParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int item = blockIdx.x * 4 + wave;
    if (item >= p.N * p.Ho * p.wtiles) return;
    const int wt = item % p.wtiles; item /= p.wtiles;
    const int ho = item % p.Ho;
    const int n = item / p.Ho;
    const int wo = wt * 32 + li;
    const int Cf = p.kw * p.C;                         // folded reduction length per filter row
    const int npair = (Cf + 1) >> 1;
    const int iw0 = wo * p.sw - p.pl;
    f32x16 acc[NT];
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;
    for (int pr = 0; pr < p.kh; ++pr) {
        const int ih = ho * p.sh + pr - p.pt;
        const bool row_ok = (unsigned)ih < (unsigned)p.H && wo < p.Wo;
        const float* xrow = p.X + (int64_t)((n * p.H + (row_ok ? ih : 0)) * p.W) * p.C;
        const float* wrow = p.Wt + (int64_t)pr * Cf * p.K;
        float a[10], b[10][NT];
#pragma unroll
        for (int kp = 0; kp < 10; ++kp) {
            if (kp < npair) {
                const int e = 2 * kp + lh;
                const int q = e / p.C;
                const bool ok = row_ok && e < Cf && (unsigned)(iw0 + q) < (unsigned)p.W;
                const float av = xrow[ok ? (int64_t)iw0 * p.C + e : 0];
                a[kp] = ok ? av : 0.f;
                const int ec = e < Cf ? e : Cf - 1;
#pragma unroll
                for (int y = 0; y < NT; ++y) {
                    const int col = y * 32 + li;
                    b[kp][y] = wrow[(int64_t)ec * p.K + (col < p.K ? col : p.K - 1)];
                }
            }
        }
#pragma unroll
        for (int kp = 0; kp < 10; ++kp)
            if (kp < npair) {
#pragma unroll
                for (int y = 0; y < NT; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], b[kp][y], acc[y], 0, 0, 0);
            }
    }
#pragma unroll
    for (int y = 0; y < NT; ++y) {
        const int col = y * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w2 = wt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (w2 < p.Wo && col < p.K) {
                const int64_t pix = (int64_t)(n * p.Ho + ho) * p.Wo + w2;
                p.Y[pix * p.y_ld + col] = epilogue_value(p.ep, acc[y][r], pix, col);
            }
        }
    }
}


// Split-bf16 version of the same layer (arithmetic: bconv.hip).  A folded filter row is kw*C <= 16 reduction
// elements = exactly one v_mfma_f32_32x32x16_bf16 k-step, so a 32-pixel x 32-filter block costs kh * 3 MFMAs instead of
// kh * ceil(kw*C/2) fp32 ones.  The filter rows are split once per wave into registers and reused over a persistent
// walk of (image, output row, 32-pixel group) items; a lane's A fragment is 8 consecutive floats of the NHWC input
// row (4-byte aligned only -- the pitch between output pixels is s*C floats -- hence scalar loads, L1-resident).
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void csplit8(const float (&v)[8], cbf16x8& hi, cbf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        hi[j] = h;
        lo[j] = (__bf16)(v[j] - (float)h);
    }
}

template <int NT, int KH>
__global__ __launch_bounds__(256) void smallc_b3_kernel(const SmallCParams p, int nitems) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int Cf = p.kw * p.C;                         // <= 16
    cbf16x8 bh[KH][NT], bl[KH][NT];
#pragma unroll
    for (int pr = 0; pr < KH; ++pr)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = y * 32 + li;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * lh + j;
                v[j] = (e < Cf && col < p.K && pr < p.kh) ? p.Wt[((int64_t)pr * Cf + e) * p.K + col] : 0.f;
            }
            csplit8(v, bh[pr][y], bl[pr][y]);
        }
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
        int it = item;
        const int wt = it % p.wtiles; it /= p.wtiles;
        const int ho = it % p.Ho;
        const int n = it / p.Ho;
        const int wo = wt * 32 + li;
        const int iw0 = wo * p.sw - p.pl;
        f32x16 acc[NT];
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;
        float a[KH][8];
#pragma unroll
        for (int pr = 0; pr < KH; ++pr) {
            const int ih = ho * p.sh + pr - p.pt;
            const bool row_ok = pr < p.kh && (unsigned)ih < (unsigned)p.H && wo < p.Wo;
            const float* xrow = p.X + (int64_t)((n * p.H + (row_ok ? ih : 0)) * p.W) * p.C + (int64_t)iw0 * p.C;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * lh + j;
                const int q = e / p.C;
                const bool ok = row_ok && e < Cf && (unsigned)(iw0 + q) < (unsigned)p.W;
                const float av = xrow[ok ? e : -(int64_t)iw0 * p.C];        // valid dummy: first element of the row
                a[pr][j] = ok ? av : 0.f;
            }
        }
#pragma unroll
        for (int pr = 0; pr < KH; ++pr) {
            cbf16x8 ah, al;
            csplit8(a[pr], ah, al);
#pragma unroll
            for (int y = 0; y < NT; ++y) {
                acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[pr][y], acc[y], 0, 0, 0);
                acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[pr][y], acc[y], 0, 0, 0);
                acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[pr][y], acc[y], 0, 0, 0);
            }
        }
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = y * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w2 = wt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (w2 < p.Wo && col < p.K) {
                    const int64_t pix = (int64_t)(n * p.Ho + ho) * p.Wo + w2;
                    p.Y[pix * p.y_ld + col] = epilogue_value(p.ep, acc[y][r], pix, col);
                }
            }
        }
    }
}

// The same arithmetic with the address arithmetic taken out (round 2; ablations of smallc_b3_kernel on the 128 x 128 x 3 -> 32
// layer at batch 64: 48 us, 37 without the input loads, 33 without the stores, 20 with neither -- each of the 56 memory
// instructions of an item carried its own 64-bit address computation, clamp and select).  Here the input and the output are
// addressed through buffer descriptors: one per-lane offset register, the row / pixel part in a scalar register, the element
// part in the instruction's immediate; image rows above / below the image come back as zeros from the descriptor's bounds check
// (a scalar select of the row offset), columns left / right of it are masked with one v_cndmask per element from eight per-item
// predicates; the four waves of a workgroup split the filter rows once between them (LDS) instead of once each.
// Requires Wo % 32 == 0, K == 32, power-of-two Wo / 32 and Ho (shifts instead of divisions), tensors below 2 GiB.
template <int KH>
__global__ __launch_bounds__(256) void smallc_b3s_kernel(const SmallCParams p, int nitems, int wt_shift, int ho_shift) {
    __shared__ __attribute__((aligned(16))) uint4 fsh[KH][2][64];          // [filter row][hi, lo][lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int Cf = p.kw * p.C;                         // <= 16
    for (int pr = wave; pr < KH; pr += 4) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = 8 * lh + j;
            v[j] = e < Cf ? p.Wt[((int64_t)pr * Cf + e) * p.K + li] : 0.f;
        }
        cbf16x8 h, l;
        csplit8(v, h, l);
        fsh[pr][0][lane] = __builtin_bit_cast(uint4, h);
        fsh[pr][1][lane] = __builtin_bit_cast(uint4, l);
    }
    __syncthreads();
    cbf16x8 bh[KH], bl[KH];
#pragma unroll
    for (int pr = 0; pr < KH; ++pr) {
        bh[pr] = __builtin_bit_cast(cbf16x8, fsh[pr][0][lane]);
        bl[pr] = __builtin_bit_cast(cbf16x8, fsh[pr][1][lane]);
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)((int64_t)p.N * p.H * p.W * p.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((int64_t)p.N * p.Ho * p.Wo * p.y_ld * 4), 0x00020000);
    const float bias = p.ep.bias ? p.ep.bias[li] : 0.f;
    const int out_lane = (4 * lh * p.y_ld + li) * 4;                       // bytes: accumulator row 4 lh (+ r & 3 + 8 (r >> 2)), column li
    const int row_bytes = p.W * p.C * 4;
    int qj[8];                                                             // input column of fragment element j, relative to iw0
#pragma unroll
    for (int j = 0; j < 8; ++j) qj[j] = (8 * lh + j) / p.C;
    // (a two-register-set prefetch of the next item's rows was measured slower at every grid size: 28.7 - 38 us against 24.8)
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
        const int wt = item & ((1 << wt_shift) - 1);
        const int rowi = item >> wt_shift;                                 // n * Ho + ho
        const int ho = rowi & ((1 << ho_shift) - 1), n = rowi >> ho_shift;
        const int iw0 = (wt * 32 + li) * p.sw - p.pl;
        const int xlane = (iw0 * p.C + 8 * lh) * 4;                        // bytes from the start of the input row (may be negative)
        bool okj[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) okj[j] = (unsigned)(iw0 + qj[j]) < (unsigned)p.W;
        float a[KH][8];
#pragma unroll
        for (int pr = 0; pr < KH; ++pr) {
            const int ih = ho * p.sh + pr - p.pt;                          // wave-uniform
            const int so = (unsigned)ih < (unsigned)p.H ? (n * p.H + ih) * row_bytes : (int)0x80000000;
            // a row outside the image: every lane's offset lands beyond num_records (lane offsets are far below 2^30) -> zeros
#pragma unroll
            for (int j = 0; j < 8; ++j)
                a[pr][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, xlane + 4 * j, so, 0));
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int pr = 0; pr < KH; ++pr) {
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = okj[j] ? a[pr][j] : 0.f;
            cbf16x8 ah, al;
            csplit8(m, ah, al);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[pr], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[pr], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[pr], acc, 0, 0, 0);
        }
        const int pix0 = rowi * p.Wo + wt * 32;                            // wave-uniform: first pixel of the item
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pq = pix0 + (q & 3) + 8 * (q >> 2);                  // + 4 lh in the lane part
            float v = act_apply(acc[q] + bias, p.ep.act, p.ep.leak);
            if (p.ep.gact != MV3D_ACT_NONE) v *= act_grad_from_out(p.ep.gref[(int64_t)(pq + 4 * lh) * p.ep.g_ld + li], p.ep.gact, p.ep.gleak);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, out_lane, pq * p.y_ld * 4, 0);
        }
    }
}

// feat2img with a thin image side (C <= 4: flow field, rgb / depth / mask heads): one thread per
// output pixel on the VALU, the phase's filter taps staged once per block in LDS.  These layers are
// HBM/L2-bound (AI ~ 40 flop/B): padding 2 channels to a 32-wide MFMA tile would multiply the work by 16.
template <int CC>
__global__ __launch_bounds__(256) void thin_feat2img_kernel(const IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];      // [ntap][CC][Ka]
    const int ph = blockIdx.z;
    const int phh = ph / p.so_w, phw = ph % p.so_w;
    const int Hp = p.Hp[phh], Wp = p.Wp[phw];
    const int Mp = p.N * Hp * Wp;
    const int tap0 = p.tap_begin[ph];
    const int ntap = p.tap_begin[ph + 1] - tap0;
    const int Ka = p.Ka;
    for (int i = threadIdx.x; i < ntap * CC * Ka; i += 256) {
        const int t = i / (CC * Ka);
        const int rem = i - t * CC * Ka;
        const int c = rem / Ka, k = rem - c * Ka;
        s_w[i] = p.Wt[(int64_t)p.taps[tap0 + t].widx * p.w_tap_stride + (int64_t)c * p.w_ns + (int64_t)k * p.w_ks];
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Mp) return;
    const int n = m / (Hp * Wp);
    const int rem = m - n * (Hp * Wp);
    const int ohp = rem / Wp, owp = rem - ohp * Wp;
    float acc[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = 0.f;
    for (int t = 0; t < ntap; ++t) {
        const IgemmTap tap = p.taps[tap0 + t];
        const int ih = ohp * p.sa_h + tap.dh, iw = owp * p.sa_w + tap.dw;
        if ((unsigned)ih >= (unsigned)p.Ha || (unsigned)iw >= (unsigned)p.Wa) continue;
        const float* src = p.A + (int64_t)((n * p.Ha + ih) * p.Wa + iw) * p.a_ld;
        const float* w = s_w + t * CC * Ka;
        for (int k = 0; k < Ka; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + k);
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float4 wv = *reinterpret_cast<const float4*>(w + c * Ka + k);
                acc[c] = fmaf(v.x, wv.x, acc[c]);
                acc[c] = fmaf(v.y, wv.y, acc[c]);
                acc[c] = fmaf(v.z, wv.z, acc[c]);
                acc[c] = fmaf(v.w, wv.w, acc[c]);
            }
        }
    }
    const int64_t pix = (int64_t)(n * p.Hc + ohp * p.so_h + phh) * p.Wc + owp * p.so_w + phw;
#pragma unroll
    for (int c = 0; c < CC; ++c) p.Out[pix * p.c_ld + c] = epilogue_value(p, acc[c], pix, c);
}

// ------------------------------------------------------------------------------------------------
// filtgrad: df[tap][c][k] = sum_m img[pix(m,tap)][c] * feat[m][k].  The reduction runs over output
// pixels, so both MFMA operands are read straight from global memory in operand layout
// (A[i=c][k=pixel], B[k=pixel][j=k]: 128-byte channel rows, coalesced); L1/L2 absorb the re-reads
// across the 4 waves of a block, which work on neighbouring taps of the same tile.
// Split over pixel slabs; partial filters are summed by reduce_slabs_kernel in a fixed order
// (no float atomics: results are run-to-run reproducible).
struct FiltgradParams {
    const float* img; const float* feat;
    float* out;          // [nslab][ntap][Cf][K]  (the final df when nslab == 1)
    float* bias_out;     // [nslab][K] or null
    int N, H, W, C, img_ld;
    int Ho, Wo, K, feat_ld;
    int kw, sh, sw, pt, pl;
    int fold, Cf, ntap;
    int ctiles, ktiles;
    int rows_total, rows_per_slab;
};

template <int NT, int U>
__global__ __launch_bounds__(256) void filtgrad_kernel(const FiltgradParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;                 // over [ctile][ktile][tap], tap fastest
    if (item >= p.ctiles * p.ktiles * p.ntap) return;
    const int tap = item % p.ntap;
    const int kt = (item / p.ntap) % p.ktiles;
    const int ct = item / (p.ntap * p.ktiles);
    const int c0 = ct * 32, k0 = kt * 32 * NT;
    const int slab = blockIdx.y;
    const int row_begin = slab * p.rows_per_slab;
    const int row_end = min(row_begin + p.rows_per_slab, p.rows_total);
    const int tp = p.fold ? tap : tap / p.kw;
    const int tq = p.fold ? 0 : tap % p.kw;
    const bool do_bias = p.bias_out != nullptr && tap == 0 && ct == 0;

    const int e = c0 + li;                                  // channel (or folded (q,c)) index of this lane
    const int eq = p.fold ? e / p.C : 0;
    const bool e_ok = e < p.Cf;

    f32x16 acc[NT];
    float sb[NT];
#pragma unroll
    for (int y = 0; y < NT; ++y) {
        sb[y] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;
    }
    bool k_ok[NT];
#pragma unroll
    for (int y = 0; y < NT; ++y) k_ok[y] = (k0 + y * 32 + li) < p.K;

    for (int row = row_begin; row < row_end; ++row) {
        const int n = row / p.Ho, ho = row - n * p.Ho;
        const int ih = ho * p.sh + tp - p.pt;
        const bool row_ok = (unsigned)ih < (unsigned)p.H;
        if (!row_ok && !do_bias) continue;
        const float* irow = p.img + (int64_t)((n * p.H + (row_ok ? ih : 0)) * p.W) * p.img_ld + (e_ok ? e : 0);
        const float* frow = p.feat + (int64_t)(row * p.Wo) * p.feat_ld + (k_ok[0] ? k0 + li : 0);
        for (int wo = 0; wo < p.Wo; wo += 2 * U) {
            // U pixel pairs per trip: all loads first (independent, clamped + masked), then the MFMAs
            float a[U], b[U][NT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int wp = wo + 2 * u + lh;
                const int iw0 = wp * p.sw + tq - p.pl;
                const bool a_ok = row_ok && e_ok && wp < p.Wo && (unsigned)(iw0 + eq) < (unsigned)p.W;
                const float av = irow[a_ok ? (int64_t)iw0 * p.img_ld : 0];
                a[u] = a_ok ? av : 0.f;
#pragma unroll
                for (int y = 0; y < NT; ++y) {
                    const bool b_ok = wp < p.Wo && k_ok[y];
                    const float bv = frow[b_ok ? (int64_t)wp * p.feat_ld + y * 32 : 0];
                    b[u][y] = b_ok ? bv : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int y = 0; y < NT; ++y) {
                    acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][y], acc[y], 0, 0, 0);
                    sb[y] += b[u][y];
                }
        }
    }
    float* out = p.out + ((int64_t)slab * p.ntap + tap) * p.Cf * p.K;
#pragma unroll
    for (int y = 0; y < NT; ++y) {
        const int k = k0 + y * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (c < p.Cf && k < p.K) out[(int64_t)c * p.K + k] = acc[y][r];
        }
        if (do_bias) {
            const float s = sb[y] + __shfl_xor(sb[y], 32);
            if (lh == 0 && k < p.K) p.bias_out[(int64_t)slab * p.K + k] = s;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ part, int nslab, int64_t count,
                                                          float* __restrict__ out, const float* __restrict__ part2,
                                                          int64_t count2, float* __restrict__ out2, int blocks1) {
    // two segments in one launch: the filter partials and (optionally) the bias partials
    __shared__ float s_sum[4][16][17];
    const int blk = blockIdx.x;
    if (blk >= blocks1) reduce_slabs_body<1>(part2, nslab, count2, blk - blocks1, s_sum, [&](int64_t i, float t) { out2[i] = t; });
    else reduce_slabs_body<VEC>(part, nslab, count, blk, s_sum, [&](int64_t i, float t) { out[i] = t; });
}


// ---- filter gradient when the image side has 1..4 channels (e0: rgb -> 32, the 2/3/1-channel heads) --------------
// df[tap][c][k] = sum_pixels img[pixel*s + tap][c] * feat[pixel][k] is a (taps*C <= 100) x (K <= 32 per workgroup column)
// matrix accumulated over N*Ho*Wo ~ 262k pixels: 1 GFLOP over 46 MB, i.e. a streaming problem.  A workgroup walks a slab
// of output-row blocks; per block the image rows and the feature rows are staged in LDS (fp32, exact); thread (kq, rg)
// owns filters 4kq..4kq+3 of up to four (tap, c) rows and does rank-1 updates from a 16-byte feature read and one
// (broadcast) image read per row.  Per-slab partials + reduce_slabs_kernel, like every other filter gradient.
struct ThinFgParams {
    const float* img; const float* feat;
    float* out; float* bias_out;
    int N, H, W, C, img_ld;
    int Ho, Wo, K, feat_ld;
    int kh, kw, sh, sw, pt, pl;
    int TR;                       // output rows per staged block
    int IR, IW;                   // staged image rows / columns: (TR-1)*sh + kh, (Wo-1)*sw + kw
    int blocks_per_img, blocks_total, blocks_per_slab;
    int nrows;                    // taps * C
};

typedef float tf2 __attribute__((ext_vector_type(2)));

// lane = (kq = lane & 3: filters 8kq..8kq+7 of the workgroup's 32, rgrp = lane >> 2: rows rgrp + 16j, j < NJ): a wave
// covers the whole (taps*C) x 32 block, 8*NJ accumulators per lane, 8*NJ FMAs per (2 + NJ) LDS reads; the four waves
// take every fourth pixel of a staged block and their partials are combined through LDS at the end (fixed order).
template <int NJ>
__global__ __launch_bounds__(256) void thin_filtgrad_kernel(const ThinFgParams p) {
    extern __shared__ __attribute__((aligned(16))) float tfs[];
    float* img_s = tfs;                                   // [IR][IW][C]
    float* feat_s = tfs + ((p.IR * p.IW * p.C + 3) & ~3); // [TR][Wo][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kq = lane & 3, rgrp = lane >> 2;
    const int k0 = blockIdx.x * 32;
    const int slab = blockIdx.y;
    int roff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = rgrp + 16 * j;
        const int rr = row < p.nrows ? row : 0;
        const int tap = rr / p.C, c = rr - tap * p.C;
        roff[j] = ((tap / p.kw) * p.IW + (tap % p.kw)) * p.C + c;
    }
    tf2 acc[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = tf2{0.f, 0.f};
    tf2 bsum[4] = {tf2{0.f, 0.f}, tf2{0.f, 0.f}, tf2{0.f, 0.f}, tf2{0.f, 0.f}};
    const int b_begin = slab * p.blocks_per_slab, b_end = min(b_begin + p.blocks_per_slab, p.blocks_total);
    const int row_elems = p.IW * p.C;
    const int npx = p.TR * p.Wo;
    const int inv_c = 65536 / p.C + 1;                    // e / C for e < 8192 (C <= 4)
    const int inv_wo = (1 << 20) / p.Wo + 1;              // px / Wo for px < 4096
    for (int blk = b_begin; blk < b_end; ++blk) {
        const int n = blk / p.blocks_per_img, oh0 = (blk - n * p.blocks_per_img) * p.TR;
        const int ih0 = oh0 * p.sh - p.pt, iw0 = -p.pl;
        if (blk > b_begin) __syncthreads();
        // Staging: every load of a batch is issued before the first LDS store of that batch, from clamped (always valid)
        // addresses with the bounds test applied to the value.  A load under a branch inside a rolled loop leaves one
        // request in flight per thread: the staging of a block then costs a dozen serial memory round trips.
        // image rows: one staged row per wave pass, lanes along the row (contiguous in global memory when img_ld == C)
        for (int r = wave; r < p.IR; r += 4) {
            const int ih = ih0 + r;
            const bool row_ok = (unsigned)ih < (unsigned)p.H;
            const float* src = p.img + (int64_t)((n * p.H + (row_ok ? ih : 0)) * p.W) * p.img_ld;
            for (int e0 = lane; e0 < row_elems; e0 += 64 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + 64 * u, row_elems - 1);
                    const int col = (int)(((unsigned)e * (unsigned)inv_c) >> 16), c = e - col * p.C;
                    const int iw = iw0 + col;
                    const bool ok = row_ok && (unsigned)iw < (unsigned)p.W;
                    const float t = src[(int64_t)(ok ? iw : 0) * p.img_ld + c];
                    v[u] = ok ? t : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + 64 * u < row_elems) img_s[r * row_elems + e0 + 64 * u] = v[u];
            }
        }
        // feature rows: TR x Wo x 32 floats as float4
        for (int i0 = tid; i0 < npx * 8; i0 += 256 * 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 256 * u, npx * 8 - 1);
                const int c4 = i & 7, px = i >> 3;
                const int r = (int)(((unsigned)px * (unsigned)inv_wo) >> 20), ow = px - r * p.Wo;
                const int oh = oh0 + r, kk = k0 + c4 * 4;
                const bool ok = oh < p.Ho && kk < p.K;
                const float4 t = *reinterpret_cast<const float4*>(p.feat + (int64_t)((n * p.Ho + (ok ? oh : 0)) * p.Wo + ow) * p.feat_ld + (ok ? kk : 0));
                v[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                if (i < npx * 8) *reinterpret_cast<float4*>(feat_s + (size_t)(i >> 3) * 32 + (i & 7) * 4) = v[u];
            }
        }
        __syncthreads();
        for (int r = 0; r < p.TR; ++r) {
            const float* frow = feat_s + (size_t)r * p.Wo * 32 + kq * 8;
            const float* irow = img_s + (size_t)(r * p.sh) * row_elems;
#pragma unroll 4
            for (int ow = wave; ow < p.Wo; ow += 4) {
                const float4 f0 = *reinterpret_cast<const float4*>(frow + ow * 32);
                const float4 f1 = *reinterpret_cast<const float4*>(frow + ow * 32 + 4);
                const tf2 fv[4] = {tf2{f0.x, f0.y}, tf2{f0.z, f0.w}, tf2{f1.x, f1.y}, tf2{f1.z, f1.w}};
                const float* ip = irow + ow * p.sw * p.C;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float a = ip[roff[j]];
                    const tf2 av = tf2{a, a};
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_elementwise_fma(av, fv[e], acc[j][e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[e] += fv[e];
            }
        }
    }
    // combine the four waves' partials through LDS (wave 0 adds waves 1..3 in order), then store
    __syncthreads();
    float* xch = tfs;                                      // [3][NJ*8 + 8][64]
    constexpr int NV = NJ * 8 + 8;
    if (wave > 0) {
        float* d = xch + (size_t)(wave - 1) * NV * 64 + lane;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[((j * 4 + e) * 2) * 64] = acc[j][e].x; d[((j * 4 + e) * 2 + 1) * 64] = acc[j][e].y; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[(NJ * 8 + e * 2) * 64] = bsum[e].x; d[(NJ * 8 + e * 2 + 1) * 64] = bsum[e].y; }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < 3; ++w) {
            const float* d = xch + (size_t)w * NV * 64 + lane;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[j][e].x += d[((j * 4 + e) * 2) * 64]; acc[j][e].y += d[((j * 4 + e) * 2 + 1) * 64]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { bsum[e].x += d[(NJ * 8 + e * 2) * 64]; bsum[e].y += d[(NJ * 8 + e * 2 + 1) * 64]; }
        }
        const int64_t fcount = (int64_t)p.kh * p.kw * p.C * p.K;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = rgrp + 16 * j;                // = tap * C + c
            if (row >= p.nrows) continue;
            float* o = p.out + (int64_t)slab * fcount + (int64_t)row * p.K + k0 + kq * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k0 + kq * 8 + 2 * e < p.K) o[2 * e] = acc[j][e].x;
                if (k0 + kq * 8 + 2 * e + 1 < p.K) o[2 * e + 1] = acc[j][e].y;
            }
        }
        if (p.bias_out && rgrp == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k0 + kq * 8 + 2 * e < p.K) p.bias_out[(int64_t)slab * p.K + k0 + kq * 8 + 2 * e] = bsum[e].x;
                if (k0 + kq * 8 + 2 * e + 1 < p.K) p.bias_out[(int64_t)slab * p.K + k0 + kq * 8 + 2 * e + 1] = bsum[e].y;
            }
        }
    }
}

static bool thin_filtgrad_plan(const mv3d_conv_geom* g, ThinFgParams& p, int* nslab_out, size_t* lds_out) {
    if (disabled_paths() & 16384) return false;
    if (g->C > 4 || g->kh * g->kw * g->C > 112 || g->K % 4 != 0 || g->feat_ld % 4 != 0 || g->Wo < 8) return false;
    p = ThinFgParams{};
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.img_ld = g->img_ld;
    p.Ho = g->Ho; p.Wo = g->Wo; p.K = g->K; p.feat_ld = g->feat_ld;
    p.kh = g->kh; p.kw = g->kw; p.sh = g->sh; p.sw = g->sw;
    int ho, wo;
    same_pad(g->H, g->kh, g->sh, &ho, &p.pt);
    same_pad(g->W, g->kw, g->sw, &wo, &p.pl);
    p.nrows = g->kh * g->kw * g->C;
    p.IW = (g->Wo - 1) * g->sw + g->kw;
    size_t lds = 0;
    for (int tr = 4; tr >= 1; tr >>= 1) {
        p.TR = std::min(tr, g->Ho);
        p.IR = (p.TR - 1) * g->sh + g->kh;
        lds = (size_t)(((p.IR * p.IW * p.C + 3) & ~3) + p.TR * g->Wo * 32) * sizeof(float);
        if (lds <= 52 * 1024) break;
    }
    {   // the end-of-kernel exchange of the four waves' partials reuses the staging area: [3][NJ*8 + 8][64] floats
        const int nj = cdiv(p.nrows, 16);
        const int NJ = nj <= 2 ? 2 : (nj <= 4 ? 4 : (nj <= 5 ? 5 : 7));
        lds = std::max(lds, (size_t)3 * (NJ * 8 + 8) * 64 * sizeof(float));
    }
    if (lds > 64 * 1024) return false;
    p.blocks_per_img = cdiv(g->Ho, p.TR);
    p.blocks_total = g->N * p.blocks_per_img;
    const int kcols = cdiv(g->K, 32);
    int nslab = std::max(1, std::min(p.blocks_total, 768 / kcols));
    p.blocks_per_slab = cdiv(p.blocks_total, nslab);
    nslab = cdiv(p.blocks_total, p.blocks_per_slab);
    *nslab_out = nslab;
    *lds_out = lds;
    return true;
}

static int dispatch_reduce(void* stream, const float* part, int nslab, int64_t fcount, float* df, const float* bpart, int K, float* db) {
    if (finalize_collecting()) {          // one batched reduction (+ optimiser) at the end of the pass: mv3d_grad_finalize_commit
        finalize_push(part, nslab, fcount, df);
        if (db && bpart) finalize_push(bpart, nslab, K, db);
        return MV3D_OK;
    }
    const bool vec = fcount % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
    const int blocks1 = (int)cdiv64(fcount, vec ? 64 : 16);
    const int blocks2 = (db && bpart) ? cdiv(K, 16) : 0;
    return dispatch(stream, OpInfo{"reduce_slabs", 0.0, 4.0 * ((double)fcount + (blocks2 ? K : 0)) * (nslab + 1)}, [=](hipStream_t s) {
        if (vec) reduce_slabs_kernel<4><<<blocks1 + blocks2, 256, 0, s>>>(part, nslab, fcount, df, bpart, K, db, blocks1);
        else reduce_slabs_kernel<1><<<blocks1 + blocks2, 256, 0, s>>>(part, nslab, fcount, df, bpart, K, db, blocks1);
        return launched("reduce_slabs_kernel");
    });
}

// ---- tiny linear layers (the angle MLP a0/a1/a2: 2->64->64->64, tf_utils.py:54-67): one thread per
// output element; an MFMA tile pipeline costs ~15 us of pure latency for ~0.5 MFLOP.
__global__ __launch_bounds__(256) void small_fc_fwd_kernel(int B, int in, int out, const float* x, int x_ld, const float* M,
                                                          float* y, int y_ld, const IgemmParams ep) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * out) return;
    const int b = idx / out, o = idx - b * out;
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < in; ++i) acc = fmaf(x[(int64_t)b * x_ld + i], M[(int64_t)i * out + o], acc);
    y[(int64_t)b * y_ld + o] = epilogue_value(ep, acc, b, o);
}
__global__ __launch_bounds__(256) void small_fc_dgrad_kernel(int B, int in, int out, const float* dy, int dy_ld, const float* M,
                                                            float* dx, int dx_ld, const IgemmParams ep) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * in) return;
    const int b = idx / in, i = idx - b * in;
    float acc = 0.f;
#pragma unroll 8
    for (int o = 0; o < out; ++o) acc = fmaf(dy[(int64_t)b * dy_ld + o], M[(int64_t)i * out + o], acc);
    dx[(int64_t)b * dx_ld + i] = epilogue_value(ep, acc, b, i);
}
__global__ __launch_bounds__(256) void small_fc_wgrad_kernel(int B, int in, int out, const float* x, int x_ld, const float* dy,
                                                            int dy_ld, float* dM, float* db) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (in + 1) * out) return;
    const int i = idx / out, o = idx - i * out;
    float acc = 0.f;
    if (i < in) {
#pragma unroll 8
        for (int b = 0; b < B; ++b) acc = fmaf(x[(int64_t)b * x_ld + i], dy[(int64_t)b * dy_ld + o], acc);
        dM[(int64_t)i * out + o] = acc;
    } else if (db) {
#pragma unroll 8
        for (int b = 0; b < B; ++b) acc += dy[(int64_t)b * dy_ld + o];
        db[o] = acc;
    }
}
// wgrad and dgrad of one small layer in one launch: blocks [0, wblocks) do small_fc_wgrad_kernel's elements, the rest
// small_fc_dgrad_kernel's -- same products, same order -- but with both operands of a block staged in LDS by ONE round of loads
// (B, in, out <= 64..128: <= 64 KiB): the per-layer kernels read them from L2 inside the loop, eight rounds of latency per element.
__global__ __launch_bounds__(256) void small_fc_bwd_kernel(int B, int in, int out, const float* x, int x_ld, const float* dy, int dy_ld,
                                                          const float* M, float* dM, float* db, float* dx, int dx_ld, const IgemmParams ep,
                                                          int wblocks) {
    extern __shared__ __attribute__((aligned(16))) float sb[];
    const int tid = threadIdx.x;
    float* s_dy = sb;                                   // [B][out]
    for (int idx = tid; idx < B * out; idx += 256) { const int b = idx / out, o = idx - b * out; s_dy[idx] = dy[(int64_t)b * dy_ld + o]; }
    if ((int)blockIdx.x < wblocks) {
        float* s_x = sb + B * out;                      // [B][in]
        for (int idx = tid; idx < B * in; idx += 256) { const int b = idx / in, i = idx - b * in; s_x[idx] = x[(int64_t)b * x_ld + i]; }
        __syncthreads();
        const int idx = blockIdx.x * 256 + tid;
        if (idx >= (in + 1) * out) return;
        const int i = idx / out, o = idx - i * out;
        float acc = 0.f;
        if (i < in) {
#pragma unroll 8
            for (int b = 0; b < B; ++b) acc = fmaf(s_x[b * in + i], s_dy[b * out + o], acc);
            dM[(int64_t)i * out + o] = acc;
        } else if (db) {
#pragma unroll 8
            for (int b = 0; b < B; ++b) acc += s_dy[b * out + o];
            db[o] = acc;
        }
        return;
    }
    float* s_m = sb + B * out;                          // [in][out + 1]
    for (int idx = tid; idx < in * out; idx += 256) { const int i = idx / out, o = idx - i * out; s_m[i * (out + 1) + o] = M[idx]; }
    __syncthreads();
    const int idx = ((int)blockIdx.x - wblocks) * 256 + tid;
    if (idx >= B * in) return;
    const int b = idx / in, i = idx - b * in;
    float acc = 0.f;
#pragma unroll 8
    for (int o = 0; o < out; ++o) acc = fmaf(s_dy[b * out + o], s_m[i * (out + 1) + o], acc);
    dx[(int64_t)b * dx_ld + i] = epilogue_value(ep, acc, b, i);
}

// A chain of small layers, four batch rows per workgroup: the arithmetic of small_fc_fwd_kernel layer by layer (same products, same
// order), the activations of a row pass from layer to layer through LDS
struct FcChainDev {
    int B, nlayers, in, x_ld;
    const float* x;
    struct L { const float* M; const float* bias; float* y; int y_ld, out, act; float leak; } l[MV3D_FC_CHAIN_MAX];
};
__global__ __launch_bounds__(256) void fc_chain_fwd_kernel(const FcChainDev c) {
    extern __shared__ __attribute__((aligned(16))) float cw[];          // all layers' matrices, then the two activation buffers
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * 4;
    // every load of the workgroup goes out before the first product: the matrices (<= 16 KiB each) and the four input rows -- in the
    // per-layer kernels a thread's 64 products wait for eight rounds of L2 latency each, which is what a layer's launch costs
    int woff[MV3D_FC_CHAIN_MAX + 1];
    {
        int in = c.in, off = 0;
        for (int l = 0; l < c.nlayers; ++l) { woff[l] = off; off += in * c.l[l].out; in = c.l[l].out; }
        woff[c.nlayers] = off;
    }
    float* a = cw + ((woff[c.nlayers] + 3) & ~3);                       // [2][4 * 64]
    {
        int in = c.in;
        for (int l = 0; l < c.nlayers; ++l) {
            const int n = in * c.l[l].out;
            for (int idx = tid; idx < n; idx += 256) cw[woff[l] + idx] = c.l[l].M[idx];
            in = c.l[l].out;
        }
    }
    for (int idx = tid; idx < 4 * c.in; idx += 256) {
        const int r = idx / c.in, i = idx - r * c.in;
        a[r * 64 + i] = b0 + r < c.B ? c.x[(int64_t)(b0 + r) * c.x_ld + i] : 0.f;
    }
    __syncthreads();
    int in = c.in, cur = 0;
    for (int l = 0; l < c.nlayers; ++l) {
        const int out = c.l[l].out;
        const float* M = cw + woff[l];
        const float* ain = a + cur * 256;
        float* aout = a + (cur ^ 1) * 256;
        for (int idx = tid; idx < 4 * out; idx += 256) {
            const int r = idx / out, o = idx - r * out;
            float acc = 0.f;
#pragma unroll 8
            for (int i = 0; i < in; ++i) acc = fmaf(ain[r * 64 + i], M[i * out + o], acc);
            if (c.l[l].bias) acc += c.l[l].bias[o];
            const float v = act_apply(acc, c.l[l].act, c.l[l].leak);
            if (b0 + r < c.B) c.l[l].y[(int64_t)(b0 + r) * c.l[l].y_ld + o] = v;
            aout[r * 64 + o] = v;
        }
        __syncthreads();
        cur ^= 1;
        in = out;
    }
}

static inline bool is_small_fc(int B, int in, int out) { return !(disabled_paths() & 4) && in <= 256 && out <= 256 && (int64_t)B * (in + out) <= (1 << 16); }

// ------------------------------------------------------------------------------------------------ host side
static int check_geom(const mv3d_conv_geom* g, const char* who) {
    if (!g) return fail(MV3D_E_INVAL, "%s: null geometry", who);
    if (g->dtype != MV3D_F32) return fail(MV3D_E_UNSUPPORTED, "%s: dtype %d not supported (fp32 only)", who, g->dtype);
    if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->K <= 0) return fail(MV3D_E_INVAL, "%s: non-positive dimension", who);
    if (g->kh <= 0 || g->kw <= 0 || g->kh * g->kw > 36) return fail(MV3D_E_INVAL, "%s: filter %dx%d unsupported", who, g->kh, g->kw);
    if (g->sh < 1 || g->sh > 2 || g->sw < 1 || g->sw > 2) return fail(MV3D_E_UNSUPPORTED, "%s: stride %dx%d unsupported (1 or 2)", who, g->sh, g->sw);
    if (g->Ho != cdiv(g->H, g->sh) || g->Wo != cdiv(g->W, g->sw))
        return fail(MV3D_E_INVAL, "%s: feature size %dx%d is not ceil(%d/%d) x ceil(%d/%d) (SAME)", who, g->Ho, g->Wo, g->H, g->sh, g->W, g->sw);
    if (g->img_ld < g->C || g->feat_ld < g->K) return fail(MV3D_E_INVAL, "%s: pixel stride smaller than channel count", who);
    if ((int64_t)g->N * g->H * g->W >= (1ll << 31) || (int64_t)g->N * g->Ho * g->Wo * 1 >= (1ll << 31))
        return fail(MV3D_E_UNSUPPORTED, "%s: more than 2^31 pixels", who);
    return MV3D_OK;
}

static void fill_epilogue(IgemmParams& p, const mv3d_epilogue* e) {
    p.bias = nullptr; p.act = MV3D_ACT_NONE; p.leak = 0.2f;
    p.gact = MV3D_ACT_NONE; p.gleak = 0.2f; p.gref = nullptr; p.g_ld = 0;
    if (!e) return;
    p.bias = (const float*)e->bias; p.act = e->act; p.leak = e->leak;
    p.gact = e->gmask_act; p.gleak = e->gmask_leak; p.gref = (const float*)e->gmask_ref; p.g_ld = e->gmask_ld;
}

static int check_epilogue(const mv3d_epilogue* e, const char* who) {
    if (!e) return MV3D_OK;
    if (e->act < 0 || e->act > MV3D_ACT_TANH || e->gmask_act < 0 || e->gmask_act > MV3D_ACT_TANH)
        return fail(MV3D_E_INVAL, "%s: bad activation enum", who);
    if (e->gmask_act != MV3D_ACT_NONE && !e->gmask_ref) return fail(MV3D_E_INVAL, "%s: gmask_act without gmask_ref", who);
    return MV3D_OK;
}

template <int WM, int MT, int NT>
static void launch_igemm_cfg(const IgemmParams& p, bool vec, bool bkmajor, dim3 grid, hipStream_t s) {
    if (vec) {
        if (bkmajor) igemm_kernel<WM, MT, NT, true, true><<<grid, 256, 0, s>>>(p);
        else igemm_kernel<WM, MT, NT, true, false><<<grid, 256, 0, s>>>(p);
    } else {
        if (bkmajor) igemm_kernel<WM, MT, NT, false, true><<<grid, 256, 0, s>>>(p);
        else igemm_kernel<WM, MT, NT, false, false><<<grid, 256, 0, s>>>(p);
    }
}

static size_t igemm_plan(IgemmParams& p, int* cfg_out, dim3* grid_out) {
    // tile selection: wide N tiles when the output has the channels, small M tiles when there are few pixels
    int maxMp = p.N * p.Hp[0] * p.Wp[0];
    int nphase = p.so_h * p.so_w;
    int cfg, BM, BN;
    if (p.Cc > 32) { cfg = (maxMp >= 128 * 512) ? 1 : 2; }      // 128x64 or 64x64
    else cfg = 0;                                                 // 128x32
    if (cfg == 0 && maxMp < 128 * 256) cfg = 3;                   // 64x32 is not instantiated; use 64x64 anyway
    if (cfg == 3) cfg = 2;
    if (cfg == 0) { BM = 128; BN = 32; }
    else if (cfg == 1) { BM = 128; BN = 64; }
    else { BM = 64; BN = 64; }
    int gx = cdiv(maxMp, BM), gy = cdiv(p.Cc, BN);
    int blocks = gx * gy * nphase;
    int min_iters = 1 << 30;
    for (int ph = 0; ph < nphase; ++ph) {
        int it = (p.tap_begin[ph + 1] - p.tap_begin[ph]) * cdiv(p.Ka, 32);
        if (it < min_iters) min_iters = it;
    }
    int ksplit = 1;
    if (blocks < 384 && min_iters >= 8) {
        ksplit = cdiv(768, blocks);
        if (ksplit > min_iters / 4) ksplit = min_iters / 4;
        if (ksplit > 32) ksplit = 32;
        if (ksplit < 1) ksplit = 1;
    }
    p.ksplit = ksplit;
    *cfg_out = cfg;
    *grid_out = dim3(gx, gy, nphase * ksplit);
    return ksplit > 1 ? (size_t)ksplit * p.N * p.Hc * p.Wc * p.Cc * sizeof(float) : 0;
}

static const char* igemm_name(int cfg, bool vec, bool bkmajor) {
    static const char* names[3][2][2] = {
        {{"igemm<128x32,scalarA,nmajorB>", "igemm<128x32,scalarA,kmajorB>"}, {"igemm<128x32,vecA,nmajorB>", "igemm<128x32,vecA,kmajorB>"}},
        {{"igemm<128x64,scalarA,nmajorB>", "igemm<128x64,scalarA,kmajorB>"}, {"igemm<128x64,vecA,nmajorB>", "igemm<128x64,vecA,kmajorB>"}},
        {{"igemm<64x64,scalarA,nmajorB>", "igemm<64x64,scalarA,kmajorB>"}, {"igemm<64x64,vecA,nmajorB>", "igemm<64x64,vecA,kmajorB>"}}};
    return names[cfg][vec ? 1 : 0][bkmajor ? 1 : 0];
}

static int run_igemm(IgemmParams p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes) {
    const bool bkmajor = (p.w_ns == 1);
    const bool vec = !p.fold && (p.Ca % 4 == 0) && (p.a_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    if (!bkmajor && p.Cc <= 4 && p.so_h * p.so_w <= 4 && (p.Ka % 4 == 0) && (p.a_ld % 4 == 0) &&
        ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0) && p.Ka <= 512) {
        // thin image-side output -> VALU kernel
        int maxMp = p.N * p.Hp[0] * p.Wp[0];
        int maxtap = 0;
        for (int ph = 0; ph < p.so_h * p.so_w; ++ph) maxtap = std::max(maxtap, p.tap_begin[ph + 1] - p.tap_begin[ph]);
        size_t lds = (size_t)maxtap * p.Cc * p.Ka * sizeof(float);
        // stride-2 square-filter heads: all four phases per thread (thin_deconv_s2_kernel)
        const int ntaps_all = p.tap_begin[p.so_h * p.so_w];
        const bool s2 = p.so_h == 2 && p.so_w == 2 && (ntaps_all == 25 || ntaps_all == 9) && p.Hc == 2 * p.Ha && p.Wc == 2 * p.Wa &&
                        p.Hp[0] == p.Ha && p.Wp[0] == p.Wa && !(disabled_paths() & 64);
        if (s2) {
            const int KS = ntaps_all == 25 ? 5 : 3;
            const size_t lds2 = (size_t)KS * KS * p.Cc * p.Ka * sizeof(float);
            if (p.Ka == 32) {          // matrix-core form (thin.hip)
                const int hrc = try_thin_head(p, stream, who, flops, bytes);
                if (hrc != 1) return hrc;
            }
            if (p.Ka == 32 && !(disabled_paths() & 131072)) {
                const int HR = KS == 5 ? 18 : 17;
                // filter rows contiguous along the reduction and 16-byte aligned: read them with scalar loads
                // (measured 43 us against 37 us with the filters in LDS: scalar and LDS returns share one counter, so every
                // use waits for all of them -- kept behind MV3D_DISABLE bit 262144 for experiments)
                const bool sw = (disabled_paths() & 262144) && p.w_ks == 1 && p.w_ns % 4 == 0 && p.w_tap_stride % 4 == 0 &&
                                (reinterpret_cast<uintptr_t>(p.Wt) & 15) == 0;
                const size_t lds3 = (sw ? 0 : lds2) + (size_t)HR * HR * 36 * sizeof(float);
                const int tiles_h = cdiv(p.Ha, 16), tiles_w = cdiv(p.Wa, 16);
                const int blocks = p.N * tiles_h * tiles_w;
                p.ksplit = 1;
                static const char* tn3[4] = {"thin_deconv_s2<1>", "thin_deconv_s2<2>", "thin_deconv_s2<3>", "thin_deconv_s2<4>"};
                return dispatch(stream, OpInfo{tn3[p.Cc - 1], flops, bytes}, [=](hipStream_t s) {
#define MV3D_THINT(KS_, CC_) do { if (sw) thin_deconv_s2_tile_kernel<KS_, CC_, true><<<blocks, 256, lds3, s>>>(p, tiles_h, tiles_w); \
                                  else thin_deconv_s2_tile_kernel<KS_, CC_, false><<<blocks, 256, lds3, s>>>(p, tiles_h, tiles_w); } while (0)
                    if (KS == 5) { switch (p.Cc) { case 1: MV3D_THINT(5, 1); break; case 2: MV3D_THINT(5, 2); break; case 3: MV3D_THINT(5, 3); break; default: MV3D_THINT(5, 4); break; } }
                    else { switch (p.Cc) { case 1: MV3D_THINT(3, 1); break; case 2: MV3D_THINT(3, 2); break; case 3: MV3D_THINT(3, 3); break; default: MV3D_THINT(3, 4); break; } }
#undef MV3D_THINT
                    return launched("thin_deconv_s2_tile_kernel");
                });
            }
            if (lds2 <= 64 * 1024) {
                const int blocks = cdiv(p.N * p.Ha * p.Wa, 256);
                p.ksplit = 1;
                static const char* tn2[4] = {"thin_deconv_s2<1>", "thin_deconv_s2<2>", "thin_deconv_s2<3>", "thin_deconv_s2<4>"};
                return dispatch(stream, OpInfo{tn2[p.Cc - 1], flops, bytes}, [=](hipStream_t s) {
#define MV3D_THIN(KS_, CC_) thin_deconv_s2_kernel<KS_, CC_><<<blocks, 256, lds2, s>>>(p)
                    if (KS == 5) { switch (p.Cc) { case 1: MV3D_THIN(5, 1); break; case 2: MV3D_THIN(5, 2); break; case 3: MV3D_THIN(5, 3); break; default: MV3D_THIN(5, 4); break; } }
                    else { switch (p.Cc) { case 1: MV3D_THIN(3, 1); break; case 2: MV3D_THIN(3, 2); break; case 3: MV3D_THIN(3, 3); break; default: MV3D_THIN(3, 4); break; } }
#undef MV3D_THIN
                    return launched("thin_deconv_s2_kernel");
                });
            }
        }
        if (lds <= 64 * 1024) {
            dim3 grid(cdiv(maxMp, 256), 1, p.so_h * p.so_w);
            p.ksplit = 1;
            static const char* tn[4] = {"thin_feat2img<1>", "thin_feat2img<2>", "thin_feat2img<3>", "thin_feat2img<4>"};
            return dispatch(stream, OpInfo{tn[p.Cc - 1], flops, bytes}, [=](hipStream_t s) {
                switch (p.Cc) {
                    case 1: thin_feat2img_kernel<1><<<grid, 256, lds, s>>>(p); break;
                    case 2: thin_feat2img_kernel<2><<<grid, 256, lds, s>>>(p); break;
                    case 3: thin_feat2img_kernel<3><<<grid, 256, lds, s>>>(p); break;
                    default: thin_feat2img_kernel<4><<<grid, 256, lds, s>>>(p); break;
                }
                return launched("thin_feat2img_kernel");
            });
        }
    }
    {
        IgemmParams e = {};
        int hrc = try_hconv(p, ws, ws_bytes, stream, who, flops, bytes, &e);
        if (hrc == 2) {
            const int64_t total = (int64_t)e.N * e.Hc * e.Wc * e.Cc;
            const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
            return dispatch(stream, OpInfo{"igemm_splitk_epilogue", 0.0, (double)total * 4.0 * (e.ksplit + 1)}, [=](hipStream_t s) {
                launch_splitk_epilogue(e, blocks, s);
                return launched("igemm_splitk_epilogue");
            });
        }
        if (hrc != 1) return hrc;
    }
    int cfg; dim3 grid;
    size_t need = igemm_plan(p, &cfg, &grid);
    if (need > ws_bytes || (need && !ws)) {
        // not enough scratch for split-K: fall back to a single pass (still correct)
        p.ksplit = 1;
        grid.z = p.so_h * p.so_w;
        need = 0;
    }
    p.Part = (float*)ws;
    int rc = dispatch(stream, OpInfo{igemm_name(cfg, vec, bkmajor), flops, bytes}, [=](hipStream_t s) {
        if (cfg == 0) launch_igemm_cfg<4, 1, 1>(p, vec, bkmajor, grid, s);
        else if (cfg == 1) launch_igemm_cfg<4, 1, 2>(p, vec, bkmajor, grid, s);
        else launch_igemm_cfg<2, 1, 1>(p, vec, bkmajor, grid, s);
        return launched(who);
    });
    if (rc != MV3D_OK || p.ksplit == 1) return rc;
    const int64_t total = (int64_t)p.N * p.Hc * p.Wc * p.Cc;
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    return dispatch(stream, OpInfo{"igemm_splitk_epilogue", 0.0, (double)total * 4.0 * (p.ksplit + 1)}, [=](hipStream_t s) {
        launch_splitk_epilogue(p, blocks, s);
        return launched("igemm_splitk_epilogue");
    });
}

static inline double conv_flops(const mv3d_conv_geom* g) {
    return 2.0 * g->N * g->Ho * g->Wo * g->kh * g->kw * (double)g->C * g->K;
}
static inline double conv_bytes(const mv3d_conv_geom* g) {
    return 4.0 * ((double)g->N * g->H * g->W * g->C + (double)g->N * g->Ho * g->Wo * g->K + (double)g->kh * g->kw * g->C * g->K);
}

// image side -> feature side (conv fwd / deconv dgrad)
// Geometry / tap table / filter strides of the image -> feature direction (conv fwd, deconv dgrad); no pointers.
static void i2f_params(const mv3d_conv_geom* g, IgemmParams& p, int* pt_out, int* pl_out) {
    int ho, wo, pt, pl;
    same_pad(g->H, g->kh, g->sh, &ho, &pt);
    same_pad(g->W, g->kw, g->sw, &wo, &pl);
    p.N = g->N; p.Ha = g->H; p.Wa = g->W; p.Ca = g->C; p.a_ld = g->img_ld;
    p.Hc = g->Ho; p.Wc = g->Wo; p.Cc = g->K; p.c_ld = g->feat_ld;
    p.sa_h = g->sh; p.sa_w = g->sw; p.so_h = 1; p.so_w = 1;
    p.Hp[0] = g->Ho; p.Wp[0] = g->Wo; p.Hp[1] = 0; p.Wp[1] = 0;
    p.fold = (g->C < 16 && g->img_ld == g->C && g->kw > 1) ? 1 : 0;
    int nt = 0;
    if (p.fold) {
        for (int pp = 0; pp < g->kh; ++pp) { p.taps[nt].dh = (int8_t)(pp - pt); p.taps[nt].dw = (int8_t)(-pl); p.taps[nt].widx = (int16_t)(pp * g->kw); ++nt; }
        p.Ka = g->kw * g->C;
    } else {
        for (int pp = 0; pp < g->kh; ++pp)
            for (int q = 0; q < g->kw; ++q) { p.taps[nt].dh = (int8_t)(pp - pt); p.taps[nt].dw = (int8_t)(q - pl); p.taps[nt].widx = (int16_t)(pp * g->kw + q); ++nt; }
        p.Ka = g->C;
    }
    p.tap_begin[0] = 0; p.tap_begin[1] = nt;
    p.w_tap_stride = g->C * g->K; p.w_ks = g->K; p.w_ns = 1;
    *pt_out = pt; *pl_out = pl;
}

static int img2feat(const mv3d_conv_geom* g, const void* img, const void* w, void* feat, const mv3d_epilogue* epi,
                    void* ws, size_t ws_bytes, void* stream, const char* who) {
    int rc = check_geom(g, who);
    if (rc == MV3D_OK) rc = check_epilogue(epi, who);
    if (rc != MV3D_OK) return rc;
    if (!img || !w || !feat) return fail(MV3D_E_INVAL, "%s: null tensor pointer", who);
    IgemmParams p = {};
    int pt, pl;
    i2f_params(g, p, &pt, &pl);
    p.A = (const float*)img; p.Wt = (const float*)w; p.Out = (float*)feat;
    fill_epilogue(p, epi);
    if (p.fold && g->C <= 4 && !(disabled_paths() & 64)) {     // row-band kernel (thin.hip): stride 2, kw * C <= 16
        const int brc = try_smallc_band(g, p, pt, pl, img, w, feat, stream, who, conv_flops(g), conv_bytes(g));
        if (brc != 1) return brc;
    }
    if (p.fold && g->C <= 4 && g->kw * g->C <= 20 && g->K <= 64 && !(disabled_paths() & 64)) {
        SmallCParams q = {};
        q.X = (const float*)img; q.Wt = (const float*)w; q.Y = (float*)feat;
        q.N = g->N; q.H = g->H; q.W = g->W; q.C = g->C; q.Ho = g->Ho; q.Wo = g->Wo; q.K = g->K; q.y_ld = g->feat_ld;
        q.kh = g->kh; q.kw = g->kw; q.sh = g->sh; q.sw = g->sw; q.pt = pt; q.pl = pl;
        q.wtiles = cdiv(g->Wo, 32);
        q.ep = p;
        const int items = g->N * g->Ho * q.wtiles;
        const bool two = g->K > 32;
        return dispatch(stream, OpInfo{two ? "smallc_img2feat<N64>" : "smallc_img2feat<N32>", conv_flops(g), conv_bytes(g)}, [=](hipStream_t s) {
            const bool b3 = !(disabled_paths() & 4096) && q.kw * q.C <= 16 && (q.kh == 5 || q.kh == 3) && !two;
            if (b3) {
                static int sc_blocks = -1;
                if (sc_blocks < 0) { const char* e = getenv("MV3D_SC_BLOCKS"); sc_blocks = e ? atoi(e) : 2048; }
                const int blocks = std::min(cdiv(items, 4), sc_blocks);     // persistent walk: the filter split is per wave
                auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
                // (forward only: with a gradient mask the saved-output loads keep their 64-bit addresses and the old kernel is faster, 26.9 vs 30.2 us)
                const bool lean = !(disabled_paths() & 8388608) && q.ep.gact == MV3D_ACT_NONE && q.Wo % 32 == 0 && q.K == 32 && pow2(q.wtiles) && pow2(q.Ho) &&
                                  (int64_t)q.N * q.H * q.W * q.C * 4 < 0x40000000 && (int64_t)q.N * q.Ho * q.Wo * q.y_ld * 4 < 0x7fffffff;
                if (lean) {
                    int ws_ = 0, hs_ = 0;
                    while ((1 << ws_) < q.wtiles) ++ws_;
                    while ((1 << hs_) < q.Ho) ++hs_;
                    if (q.kh == 5) smallc_b3s_kernel<5><<<blocks, 256, 0, s>>>(q, items, ws_, hs_);
                    else smallc_b3s_kernel<3><<<blocks, 256, 0, s>>>(q, items, ws_, hs_);
                } else if (q.kh == 5) smallc_b3_kernel<1, 5><<<blocks, 256, 0, s>>>(q, items);
                else smallc_b3_kernel<1, 3><<<blocks, 256, 0, s>>>(q, items);
            } else if (two) smallc_img2feat_kernel<2><<<cdiv(items, 4), 256, 0, s>>>(q);
            else smallc_img2feat_kernel<1><<<cdiv(items, 4), 256, 0, s>>>(q);
            return launched(who);
        });
    }
    return run_igemm(p, ws, ws_bytes, stream, who, conv_flops(g), conv_bytes(g));
}

// Same for the feature -> image direction (conv dgrad, deconv fwd): one dense sub-convolution per stride phase.
static void f2i_params(const mv3d_conv_geom* g, IgemmParams& p) {
    int ho, wo, pt, pl;
    same_pad(g->H, g->kh, g->sh, &ho, &pt);
    same_pad(g->W, g->kw, g->sw, &wo, &pl);
    p.N = g->N; p.Ha = g->Ho; p.Wa = g->Wo; p.Ca = g->K; p.a_ld = g->feat_ld;
    p.Hc = g->H; p.Wc = g->W; p.Cc = g->C; p.c_ld = g->img_ld;
    p.sa_h = 1; p.sa_w = 1; p.so_h = g->sh; p.so_w = g->sw;
    for (int a = 0; a < 2; ++a) {
        p.Hp[a] = a < g->sh ? (g->H - a + g->sh - 1) / g->sh : 0;
        p.Wp[a] = a < g->sw ? (g->W - a + g->sw - 1) / g->sw : 0;
    }
    int nt = 0;
    for (int phh = 0; phh < g->sh; ++phh)
        for (int phw = 0; phw < g->sw; ++phw) {
            p.tap_begin[phh * g->sw + phw] = nt;
            const int p0 = (phh + pt) % g->sh, q0 = (phw + pl) % g->sw;
            for (int pp = p0; pp < g->kh; pp += g->sh)
                for (int q = q0; q < g->kw; q += g->sw) {
                    // u = s*u' + ph, tap p: i = (u + pt - p)/s = u' + (ph + pt - p)/s  (exact division)
                    p.taps[nt].dh = (int8_t)((phh + pt - pp) / g->sh);
                    p.taps[nt].dw = (int8_t)((phw + pl - q) / g->sw);
                    p.taps[nt].widx = (int16_t)(pp * g->kw + q);
                    ++nt;
                }
        }
    p.tap_begin[g->sh * g->sw] = nt;
    p.fold = 0; p.Ka = g->K;
    p.w_tap_stride = g->C * g->K; p.w_ks = 1; p.w_ns = g->K;
}

// feature side -> image side (conv dgrad / deconv fwd), one dense sub-convolution per stride phase
static int feat2img(const mv3d_conv_geom* g, const void* feat, const void* w, void* img, const mv3d_epilogue* epi,
                    void* ws, size_t ws_bytes, void* stream, const char* who) {
    int rc = check_geom(g, who);
    if (rc == MV3D_OK) rc = check_epilogue(epi, who);
    if (rc != MV3D_OK) return rc;
    if (!img || !w || !feat) return fail(MV3D_E_INVAL, "%s: null tensor pointer", who);
    IgemmParams p = {};
    f2i_params(g, p);
    p.A = (const float*)feat; p.Wt = (const float*)w; p.Out = (float*)img;
    fill_epilogue(p, epi);
    return run_igemm(p, ws, ws_bytes, stream, who, conv_flops(g), conv_bytes(g));
}

struct WgTileParams;
bool wgrad_tile_plan(const mv3d_conv_geom* g, struct WgTileParams* out, int* nslab_out, int* cfg_out, size_t* lds_out);
int wgrad_tile_launch_erased(const mv3d_conv_geom* g, const void* img, const void* feat, void* out, void* bias_out, void* stream, const char* who, int* nslab_out);

static void filtgrad_plan(const mv3d_conv_geom* g, FiltgradParams& p, int* nt_out, int* nslab_out) {
    int ho, wo;
    same_pad(g->H, g->kh, g->sh, &ho, &p.pt);
    same_pad(g->W, g->kw, g->sw, &wo, &p.pl);
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.img_ld = g->img_ld;
    p.Ho = g->Ho; p.Wo = g->Wo; p.K = g->K; p.feat_ld = g->feat_ld;
    p.kw = g->kw; p.sh = g->sh; p.sw = g->sw;
    p.fold = (g->C < 16 && g->img_ld == g->C && g->kw > 1) ? 1 : 0;
    p.Cf = p.fold ? g->kw * g->C : g->C;
    p.ntap = p.fold ? g->kh : g->kh * g->kw;
    const int NT = (g->K > 32) ? 2 : 1;
    p.ctiles = cdiv(p.Cf, 32);
    p.ktiles = cdiv(g->K, 32 * NT);
    p.rows_total = g->N * g->Ho;
    const int items = p.ctiles * p.ktiles * p.ntap;
    const int blocks_x = cdiv(items, 4);
    int nslab = cdiv(2048, blocks_x);
    if (nslab > p.rows_total) nslab = p.rows_total;
    // do not split below ~64 pixel pairs of work per slab
    int min_rows = cdiv(128, g->Wo);
    if (nslab > cdiv(p.rows_total, min_rows)) nslab = cdiv(p.rows_total, min_rows);
    if (nslab < 1) nslab = 1;
    p.rows_per_slab = cdiv(p.rows_total, nslab);
    nslab = cdiv(p.rows_total, p.rows_per_slab);
    *nt_out = NT;
    *nslab_out = nslab;
}

int wgrad_tile_nslab(const mv3d_conv_geom* g);
int set_wgrad_cus(int cus);

static size_t filtgrad_ws_bytes(const mv3d_conv_geom* g) {
    {
        const int tw = thin_wgrad_slabs(g);
        if (tw > 0) return (size_t)tw * ((size_t)g->kh * g->kw * g->C * g->K + g->K) * sizeof(float);
    }
    {
        ThinFgParams tp; int ns; size_t lds;
        if (thin_filtgrad_plan(g, tp, &ns, &lds)) return (size_t)ns * ((size_t)g->kh * g->kw * g->C * g->K + g->K) * sizeof(float);
    }
    FiltgradParams p = {};
    int NT, nslab;
    filtgrad_plan(g, p, &NT, &nslab);
    const int tiled = wgrad_tile_nslab(g);
    if (tiled > 0) nslab = tiled;
    if (nslab == 1) return 0;
    return (size_t)nslab * ((size_t)g->kh * g->kw * g->C * g->K + g->K) * sizeof(float);
}

static int filtgrad(const mv3d_conv_geom* g, const void* img, const void* feat, void* df, void* db,
                    void* ws, size_t ws_bytes, void* stream, const char* who) {
    int rc = check_geom(g, who);
    if (rc != MV3D_OK) return rc;
    if (!img || !feat || !df) return fail(MV3D_E_INVAL, "%s: null tensor pointer", who);
    const int64_t fcount = (int64_t)g->kh * g->kw * g->C * g->K;
    float* dfp = (float*)df; float* dbp = (float*)db;
    const int K = g->K;
    {   // tiled persistent kernel for the layers that carry the FLOPs
        const int tiled = wgrad_tile_nslab(g);
        if (tiled > 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
            float* part = dfp; float* bpart = dbp;
            if (tiled > 1) {
                size_t need = (size_t)tiled * (fcount + K) * sizeof(float);
                if (!ws || ws_bytes < need) return fail(MV3D_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
                part = (float*)ws;
                bpart = db ? (float*)ws + (int64_t)tiled * fcount : nullptr;
            }
            int ns = 0;
            rc = wgrad_tile_launch_erased(g, img, feat, part, bpart, stream, who, &ns);
            if (rc == MV3D_OK && ns == 1 && finalize_collecting()) { finalize_push(nullptr, 0, fcount, dfp); if (dbp) finalize_push(nullptr, 0, K, dbp); }
            if (rc != MV3D_OK || ns == 1) return rc;
            return dispatch_reduce(stream, part, ns, fcount, dfp, bpart, K, dbp);
        }
    }
    {   // 1..3 image-side channels, stride 2: matrix-core band kernel (thin.hip)
        const int tw = thin_wgrad_slabs(g);
        if (tw > 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0) {
            const size_t need = (size_t)tw * (fcount + K) * sizeof(float);
            if (!ws || ws_bytes < need) return fail(MV3D_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
            float* part = (float*)ws;
            float* bpart = db ? (float*)ws + (int64_t)tw * fcount : nullptr;
            rc = thin_wgrad_launch(g, img, feat, part, bpart, stream, who, conv_flops(g), conv_bytes(g));
            if (rc == MV3D_OK) return dispatch_reduce(stream, part, tw, fcount, dfp, bpart, K, dbp);
            if (rc != 1) return rc;
        }
    }
    {   // 1..4 image-side channels: streaming VALU kernel
        ThinFgParams tp; int ns; size_t lds;
        if (thin_filtgrad_plan(g, tp, &ns, &lds) && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
            const size_t need = (size_t)ns * (fcount + K) * sizeof(float);
            if (!ws || ws_bytes < need) return fail(MV3D_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
            tp.img = (const float*)img; tp.feat = (const float*)feat;
            tp.out = (float*)ws;
            tp.bias_out = db ? (float*)ws + (int64_t)ns * fcount : nullptr;
            dim3 grid(cdiv(g->K, 32), ns);
            const int nj = cdiv(tp.nrows, 16);
            rc = dispatch(stream, OpInfo{"thin_filtgrad", conv_flops(g), conv_bytes(g)}, [=](hipStream_t s) {
                if (nj <= 2) thin_filtgrad_kernel<2><<<grid, 256, lds, s>>>(tp);
                else if (nj <= 4) thin_filtgrad_kernel<4><<<grid, 256, lds, s>>>(tp);
                else if (nj <= 5) thin_filtgrad_kernel<5><<<grid, 256, lds, s>>>(tp);
                else thin_filtgrad_kernel<7><<<grid, 256, lds, s>>>(tp);
                return launched(who);
            });
            if (rc != MV3D_OK) return rc;
            return dispatch_reduce(stream, tp.out, ns, fcount, dfp, tp.bias_out, K, dbp);
        }
    }
    FiltgradParams p = {};
    int NT, nslab;
    filtgrad_plan(g, p, &NT, &nslab);
    p.img = (const float*)img; p.feat = (const float*)feat;
    if (nslab > 1) {
        size_t need = (size_t)nslab * (fcount + g->K) * sizeof(float);
        if (!ws || ws_bytes < need) return fail(MV3D_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
        p.out = (float*)ws;
        p.bias_out = db ? (float*)ws + (int64_t)nslab * fcount : nullptr;
    } else {
        p.out = (float*)df;
        p.bias_out = (float*)db;
    }
    dim3 grid(cdiv(p.ctiles * p.ktiles * p.ntap, 4), nslab);
    rc = dispatch(stream, OpInfo{NT == 2 ? "filtgrad<NT=2>" : "filtgrad<NT=1>", conv_flops(g), conv_bytes(g)}, [=](hipStream_t s) {
        const bool wide = p.Wo >= 16;
        if (NT == 2) { if (wide) filtgrad_kernel<2, 8><<<grid, 256, 0, s>>>(p); else filtgrad_kernel<2, 2><<<grid, 256, 0, s>>>(p); }
        else { if (wide) filtgrad_kernel<1, 8><<<grid, 256, 0, s>>>(p); else filtgrad_kernel<1, 2><<<grid, 256, 0, s>>>(p); }
        return launched(who);
    });
    if (rc == MV3D_OK && nslab == 1 && finalize_collecting()) { finalize_push(nullptr, 0, fcount, dfp); if (dbp) finalize_push(nullptr, 0, K, dbp); }
    if (rc != MV3D_OK || nslab == 1) return rc;
    return dispatch_reduce(stream, p.out, nslab, fcount, dfp, p.bias_out, K, dbp);
}

}  // namespace mv3d

using namespace mv3d;

extern "C" {

int mv3d_conv2d_fwd(const mv3d_conv_geom* g, const void* x, const void* w, void* y, const mv3d_epilogue* epi,
                    void* ws, size_t wsb, void* stream) {
    return img2feat(g, x, w, y, epi, ws, wsb, stream, "mv3d_conv2d_fwd");
}
int mv3d_conv2d_dgrad(const mv3d_conv_geom* g, const void* dy, const void* w, void* dx, const mv3d_epilogue* epi,
                      void* ws, size_t wsb, void* stream) {
    return feat2img(g, dy, w, dx, epi, ws, wsb, stream, "mv3d_conv2d_dgrad");
}
int mv3d_conv2d_wgrad(const mv3d_conv_geom* g, const void* x, const void* dy, void* dw, void* db,
                      void* ws, size_t wsb, void* stream) {
    return filtgrad(g, x, dy, dw, db, ws, wsb, stream, "mv3d_conv2d_wgrad");
}
int mv3d_deconv2d_fwd(const mv3d_conv_geom* g, const void* x, const void* w, void* y, const mv3d_epilogue* epi,
                      void* ws, size_t wsb, void* stream) {
    return feat2img(g, x, w, y, epi, ws, wsb, stream, "mv3d_deconv2d_fwd");
}
int mv3d_deconv2d_dgrad(const mv3d_conv_geom* g, const void* dy, const void* w, void* dx, const mv3d_epilogue* epi,
                        void* ws, size_t wsb, void* stream) {
    return img2feat(g, dy, w, dx, epi, ws, wsb, stream, "mv3d_deconv2d_dgrad");
}
int mv3d_deconv2d_wgrad(const mv3d_conv_geom* g, const void* x, const void* dy, void* dw, void* ws, size_t wsb, void* stream) {
    return filtgrad(g, dy, x, dw, nullptr, ws, wsb, stream, "mv3d_deconv2d_wgrad");
}

static bool filter_op_params(const mv3d_conv_geom* g, int op, const void* w, IgemmParams& p, const char* who) {
    if (!g || check_geom(g, who) != MV3D_OK) return false;
    if (op == MV3D_FILTER_CONV_FWD || op == MV3D_FILTER_DECONV_DGRAD) { int pt, pl; i2f_params(g, p, &pt, &pl); }
    else if (op == MV3D_FILTER_CONV_DGRAD || op == MV3D_FILTER_DECONV_FWD) f2i_params(g, p);
    else { fail(MV3D_E_INVAL, "%s: unknown filter operation %d", who, op); return false; }
    p.Wt = (const float*)w;
    return true;
}

size_t mv3d_filter_prepared_bytes(const mv3d_conv_geom* g, int op) {
    IgemmParams p = {};
    if (!filter_op_params(g, op, nullptr, p, "mv3d_filter_prepared_bytes")) return 0;
    if (disabled_paths() & (1 | 4096)) return 0;
    return bconv_prepared_bytes(p);
}

int mv3d_filter_cache_bind(const mv3d_conv_geom* g, int op, const void* w, void* prepared, size_t prepared_bytes) {
    IgemmParams p = {};
    if (!w) return fail(MV3D_E_INVAL, "mv3d_filter_cache_bind: null filter pointer");
    if (!filter_op_params(g, op, w, p, "mv3d_filter_cache_bind")) return MV3D_E_INVAL;
    int rc = bconv_cache_bind(p, prepared, prepared_bytes);
    if (rc == 1) return fail(MV3D_E_UNSUPPORTED, "mv3d_filter_cache_bind: this operation does not take a prepared filter");
    return rc;
}

int mv3d_set_wgrad_cus(int cus) { return set_wgrad_cus(cus); }

size_t mv3d_conv_wgrad_workspace_bytes(const mv3d_conv_geom* g) {
    if (!g || check_geom(g, "mv3d_conv_wgrad_workspace_bytes") != MV3D_OK) return 0;
    return filtgrad_ws_bytes(g);
}

size_t mv3d_conv_workspace_bytes(const mv3d_conv_geom* g) {
    if (!g || check_geom(g, "mv3d_conv_workspace_bytes") != MV3D_OK) return 0;
    // split-K partials of either direction: at most 32 slabs of the larger output
    size_t out_img = (size_t)g->N * g->H * g->W * g->C, out_feat = (size_t)g->N * g->Ho * g->Wo * g->K;
    size_t big = out_img > out_feat ? out_img : out_feat;
    size_t igemm = 0;
    {   // replay the planner for both directions to get the exact split
        IgemmParams p = {};
        int cfg; dim3 grid;
        p.N = g->N; p.Hc = g->Ho; p.Wc = g->Wo; p.Cc = g->K; p.so_h = p.so_w = 1; p.Hp[0] = g->Ho; p.Wp[0] = g->Wo;
        const bool fold = (g->C < 16 && g->img_ld == g->C && g->kw > 1);
        p.Ka = fold ? g->kw * g->C : g->C; p.tap_begin[0] = 0; p.tap_begin[1] = fold ? g->kh : g->kh * g->kw;
        size_t a = igemm_plan(p, &cfg, &grid);
        igemm = a;
        (void)big;
        IgemmParams q = {};
        q.N = g->N; q.Hc = g->H; q.Wc = g->W; q.Cc = g->C; q.so_h = g->sh; q.so_w = g->sw;
        for (int i = 0; i < 2; ++i) { q.Hp[i] = i < g->sh ? (g->H - i + g->sh - 1) / g->sh : 0; q.Wp[i] = i < g->sw ? (g->W - i + g->sw - 1) / g->sw : 0; }
        q.Ka = g->K;
        int ho, wo, pt, pl;
        same_pad(g->H, g->kh, g->sh, &ho, &pt);
        same_pad(g->W, g->kw, g->sw, &wo, &pl);
        int nt = 0;
        for (int phh = 0; phh < g->sh; ++phh)
            for (int phw = 0; phw < g->sw; ++phw) {
                q.tap_begin[phh * g->sw + phw] = nt;
                for (int pp = (phh + pt) % g->sh; pp < g->kh; pp += g->sh)
                    for (int qq = (phw + pl) % g->sw; qq < g->kw; qq += g->sw) ++nt;
            }
        q.tap_begin[g->sh * g->sw] = nt;
        size_t b = igemm_plan(q, &cfg, &grid);
        if (b > igemm) igemm = b;
    }
    size_t fg = filtgrad_ws_bytes(g);
    // transposed / fragment-ordered split filter copy of the halo kernels (channel counts padded to the MFMA tile)
    size_t filt = (size_t)g->kh * g->kw * ((g->C + 63) / 64 * 64) * ((g->K + 63) / 64 * 64) * sizeof(float);
    size_t m = igemm > fg ? igemm : fg;
    // small-image halo kernel: transposed filter copy + up to 16 chunk-split partial copies of an output
    const size_t big_out = out_img > out_feat ? out_img : out_feat;
    size_t small = 0;
    if ((size_t)g->Ho * g->Wo <= 64 || (size_t)g->H * g->W <= 64) small = ((filt + 255) & ~(size_t)255) + 16 * big_out * sizeof(float);
    m = m > small ? m : small;
    return m > filt ? m : filt;
}

// ---- linear layers as 1x1 "convolutions" over a 1 x B image (tf_utils.py:67) -------------------
static void fc_geom(mv3d_conv_geom& g, int B, int in, int out, int x_ld, int y_ld) {
    g.N = 1; g.H = 1; g.W = B; g.C = in; g.Ho = 1; g.Wo = B; g.K = out;
    g.kh = g.kw = g.sh = g.sw = 1; g.img_ld = x_ld; g.feat_ld = y_ld; g.dtype = MV3D_F32;
}
int mv3d_fc_fwd(int B, int in, int out, const void* x, int x_ld, const void* M, void* y, int y_ld,
                const mv3d_epilogue* epi, void* ws, size_t wsb, void* stream) {
    mv3d_conv_geom g; fc_geom(g, B, in, out, x_ld, y_ld);
    if (B > 0 && in > 0 && out > 0 && x && M && y && is_small_fc(B, in, out)) {
        int rc = check_epilogue(epi, "mv3d_fc_fwd");
        if (rc != MV3D_OK) return rc;
        IgemmParams ep = {}; fill_epilogue(ep, epi);
        return dispatch(stream, OpInfo{"small_fc_fwd", 2.0 * B * in * out, 4.0 * (B * in + in * out + B * out)}, [=](hipStream_t s) {
            small_fc_fwd_kernel<<<cdiv(B * out, 256), 256, 0, s>>>(B, in, out, (const float*)x, x_ld, (const float*)M, (float*)y, y_ld, ep);
            return launched("small_fc_fwd_kernel");
        });
    }
    if (B > 0 && in > 0 && out > 0 && x && M && y && x_ld >= in && y_ld >= out && check_epilogue(epi, "mv3d_fc_fwd") == MV3D_OK) {
        int rc = try_fc_stream(false, B, in, out, x, x_ld, M, y, y_ld, epi, ws, wsb, stream, "mv3d_fc_fwd", fill_epilogue, launch_splitk_epilogue);
        if (rc != 1) return rc;
    }
    return img2feat(&g, x, M, y, epi, ws, wsb, stream, "mv3d_fc_fwd");
}
int mv3d_fc_dgrad(int B, int in, int out, const void* dy, int dy_ld, const void* M, void* dx, int dx_ld,
                  const mv3d_epilogue* epi, void* ws, size_t wsb, void* stream) {
    mv3d_conv_geom g; fc_geom(g, B, in, out, dx_ld, dy_ld);
    if (B > 0 && in > 0 && out > 0 && dy && M && dx && is_small_fc(B, in, out)) {
        int rc = check_epilogue(epi, "mv3d_fc_dgrad");
        if (rc != MV3D_OK) return rc;
        IgemmParams ep = {}; fill_epilogue(ep, epi);
        return dispatch(stream, OpInfo{"small_fc_dgrad", 2.0 * B * in * out, 4.0 * (B * in + in * out + B * out)}, [=](hipStream_t s) {
            small_fc_dgrad_kernel<<<cdiv(B * in, 256), 256, 0, s>>>(B, in, out, (const float*)dy, dy_ld, (const float*)M, (float*)dx, dx_ld, ep);
            return launched("small_fc_dgrad_kernel");
        });
    }
    if (B > 0 && in > 0 && out > 0 && dy && M && dx && dy_ld >= out && dx_ld >= in && check_epilogue(epi, "mv3d_fc_dgrad") == MV3D_OK) {
        int rc = try_fc_stream(true, B, in, out, dy, dy_ld, M, dx, dx_ld, epi, ws, wsb, stream, "mv3d_fc_dgrad", fill_epilogue, launch_splitk_epilogue);
        if (rc != 1) return rc;
    }
    return feat2img(&g, dy, M, dx, epi, ws, wsb, stream, "mv3d_fc_dgrad");
}
int mv3d_fc_wgrad(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* dM, void* db,
                  void* ws, size_t wsb, void* stream) {
    mv3d_conv_geom g; fc_geom(g, B, in, out, x_ld, dy_ld);
    if (B > 0 && in > 0 && out > 0 && x && dy && dM && is_small_fc(B, in, out)) {
        return dispatch(stream, OpInfo{"small_fc_wgrad", 2.0 * B * in * out, 4.0 * (B * in + in * out + B * out)}, [=](hipStream_t s) {
            small_fc_wgrad_kernel<<<cdiv((in + 1) * out, 256), 256, 0, s>>>(B, in, out, (const float*)x, x_ld, (const float*)dy, dy_ld, (float*)dM, (float*)db);
            return launched("small_fc_wgrad_kernel");
        });
    }
    if (B > 0 && in > 0 && out > 0 && x && dy && dM && x_ld >= in && dy_ld >= out) {
        int rc = try_fc_wgrad(B, in, out, x, x_ld, dy, dy_ld, dM, db, stream, "mv3d_fc_wgrad");
        if (rc != 1) return rc;
    }
    return filtgrad(&g, x, dy, dM, db, ws, wsb, stream, "mv3d_fc_wgrad");
}
int mv3d_fc_wgrad_dgrad(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, const void* M,
                        void* dM, void* db, void* dx, int dx_ld, const mv3d_epilogue* dx_epi, void* ws, size_t wsb, void* stream) {
    if (B > 0 && in > 0 && out > 0 && x && dy && M && dM && dx && is_small_fc(B, in, out)) {
        int rc = check_epilogue(dx_epi, "mv3d_fc_wgrad_dgrad");
        if (rc != MV3D_OK) return rc;
        IgemmParams ep = {}; fill_epilogue(ep, dx_epi);
        const int wblocks = cdiv((in + 1) * out, 256), dblocks = cdiv(B * in, 256);
        const size_t lds = ((size_t)B * out + std::max((size_t)B * in, (size_t)in * (out + 1))) * sizeof(float);
        if (lds <= 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_fc_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipGetLastError();        // recording without a device: the attribute call fails and nothing is launched
        }
        if (lds <= 64 * 1024)
        return dispatch(stream, OpInfo{"small_fc_bwd", 4.0 * B * in * out, 4.0 * (2.0 * B * in + 2.0 * in * out + 2.0 * B * out)}, [=](hipStream_t s) {
            small_fc_bwd_kernel<<<wblocks + dblocks, 256, lds, s>>>(B, in, out, (const float*)x, x_ld, (const float*)dy, dy_ld, (const float*)M,
                                                                  (float*)dM, (float*)db, (float*)dx, dx_ld, ep, wblocks);
            return launched("small_fc_bwd_kernel");
        });
    }
    int rc = mv3d_fc_wgrad(B, in, out, x, x_ld, dy, dy_ld, dM, db, ws, wsb, stream);
    if (rc != MV3D_OK) return rc;
    return mv3d_fc_dgrad(B, in, out, dy, dy_ld, M, dx, dx_ld, dx_epi, ws, wsb, stream);
}
int mv3d_fc_chain_fwd(const mv3d_fc_chain* c, void* stream) {
    if (!c || c->nlayers < 2 || c->nlayers > MV3D_FC_CHAIN_MAX || c->B <= 0 || c->in <= 0 || c->in > 64 || !c->x || c->x_ld < c->in)
        return fail(MV3D_E_INVAL, "mv3d_fc_chain_fwd: bad chain (2..%d layers, widths <= 64)", MV3D_FC_CHAIN_MAX);
    if (disabled_paths() & 4) return fail(MV3D_E_UNSUPPORTED, "mv3d_fc_chain_fwd: small fc kernels are disabled (MV3D_DISABLE bit 4)");
    FcChainDev d = {};
    d.B = c->B; d.nlayers = c->nlayers; d.in = c->in; d.x_ld = c->x_ld; d.x = (const float*)c->x;
    double flops = 0.0, bytes = 4.0 * c->B * c->in;
    int in = c->in;
    for (int l = 0; l < c->nlayers; ++l) {
        const mv3d_fc_chain_layer& s = c->l[l];
        if (!s.M || !s.y || s.out <= 0 || s.out > 64 || s.y_ld < s.out || s.act < 0 || s.act > MV3D_ACT_TANH)
            return fail(MV3D_E_INVAL, "mv3d_fc_chain_fwd: bad layer %d", l);
        d.l[l].M = (const float*)s.M; d.l[l].bias = (const float*)s.bias; d.l[l].y = (float*)s.y;
        d.l[l].y_ld = s.y_ld; d.l[l].out = s.out; d.l[l].act = s.act; d.l[l].leak = s.leak;
        flops += 2.0 * c->B * in * s.out; bytes += 4.0 * ((double)in * s.out + (double)c->B * s.out);
        in = s.out;
    }
    const int blocks = cdiv(c->B, 4);
    size_t welems = 0;
    { int k = c->in; for (int l = 0; l < c->nlayers; ++l) { welems += (size_t)k * c->l[l].out; k = c->l[l].out; } }
    const size_t lds = (((welems + 3) & ~(size_t)3) + 2 * 256) * sizeof(float);       // <= 66 KiB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_chain_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipGetLastError();
    return dispatch(stream, OpInfo{"small_fc_chain_fwd", flops, bytes}, [=](hipStream_t s) {
        fc_chain_fwd_kernel<<<blocks, 256, lds, s>>>(d);
        return launched("fc_chain_fwd_kernel");
    });
}
int mv3d_fc_wgrad_adam(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* M, void* adam_m, void* adam_v,
                       void* db, const void* adam_state, void* stream) {
    if (B <= 0 || in <= 0 || out <= 0 || !x || !dy || !M || !adam_m || !adam_v || !adam_state || x_ld < in || dy_ld < out)
        return fail(MV3D_E_INVAL, "mv3d_fc_wgrad_adam: bad arguments");
    // one predicate with mv3d_fc_wgrad_adam_supported (plus the pointer alignment only this call can see)
    int rc = mv3d_fc_wgrad_adam_supported(B, in, out, x_ld, dy_ld) ? try_fc_wgrad_adam(B, in, out, x, x_ld, dy, dy_ld, M, adam_m, adam_v, db, adam_state, stream, "mv3d_fc_wgrad_adam") : 1;
    if (rc == 1) return fail(MV3D_E_UNSUPPORTED, "mv3d_fc_wgrad_adam: %d x %d x %d is not a layer of the fused kernel (use mv3d_fc_wgrad + mv3d_adam_step_dev)", B, in, out);
    return rc;
}
int mv3d_fc_wgrad_adam_supported(int B, int in, int out, int x_ld, int dy_ld) {
    if (is_small_fc(B, in, out)) return 0;
    return !(B < 2 || in < 64 || out < 64 || (disabled_paths() & (16 | 4096)) || in % 4 || out % 4 || x_ld % 4 || dy_ld % 4);
}
size_t mv3d_fc_workspace_bytes(int B, int in, int out) {
    mv3d_conv_geom g; fc_geom(g, B, in, out, in, out);
    size_t a = mv3d_conv_workspace_bytes(&g);
    a = std::max(a, fc_stream_ws_bytes(B, in, out, false));
    a = std::max(a, fc_stream_ws_bytes(B, in, out, true));
    return a;
}

}  // extern "C"
