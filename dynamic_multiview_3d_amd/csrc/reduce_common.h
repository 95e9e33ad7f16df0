// Fixed-order sum of per-slab partial filters, shared by reduce_slabs_kernel (conv.hip) and grad_finalize_kernel (elem.hip).
#pragma once
#include "common.h"

namespace mv3d {

// Sums the per-slab partial filters in a fixed order (deterministic).  256 threads = 16 element lanes x 16 slab
// lanes: each thread adds every 16th slab with independent loads in flight, the 16 lanes of an element are combined
// through LDS; fin(index, sum) finishes an element (store, or optimiser update).  VEC=4: an element lane owns four consecutive floats (16-byte loads, 256 contiguous bytes per slab row
// and wave quarter); VEC=1 is the scalar form for counts that are not a multiple of four and for the bias segment.
template <int VEC, class FIN>
__device__ __forceinline__ void reduce_slabs_body(const float* __restrict__ part, int nslab, int64_t count,
                                                  int blk, float (*s_sum)[16][17], FIN fin) {
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = ((int64_t)blk * 16 + e) * VEC;
    float acc[8][VEC];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[u][v] = 0.f;
    if (i < count) {
        const float* p = part + (int64_t)sl * count + i;
        const int64_t step = 16 * count;
        int k = sl;
        for (; k + 112 < nslab; k += 128, p += 8 * step) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p + u * step);
                    acc[u][0] += t.x; acc[u][1] += t.y; acc[u][2] += t.z; acc[u][3] += t.w;
                } else {
                    acc[u][0] += p[u * step];
                }
            }
        }
        for (; k < nslab; k += 16, p += step) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                acc[0][0] += t.x; acc[0][1] += t.y; acc[0][2] += t.z; acc[0][3] += t.w;
            } else {
                acc[0][0] += p[0];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        s_sum[v][sl][e] = ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) + ((acc[4][v] + acc[5][v]) + (acc[6][v] + acc[7][v]));
    __syncthreads();
    if (sl < VEC && i < count) {                       // thread (sl = v, e) finishes component v of element lane e
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += s_sum[sl][j][e];
        fin(i + sl, t);                                 // element index inside the segment, its sum
    }
}

}  // namespace mv3d
