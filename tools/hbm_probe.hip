// Read-rate probe for the fc weight stream (GPU box only):  hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe
// Patterns over a [R][N] fp32 matrix (4096 x 4096 by default), every element read once per launch:
//   linear : grid-stride float4, 8 loads in flight per lane
//   colrow : the fc_stream forward pattern -- workgroup = 128 columns x R/nsplit rows, lane (li, lh) reads 16 rows
//            (one dword each, row stride N) of column n0 + li, DEPTH chunk groups of 16 loads in flight
//   rowvec : the dgrad pattern -- lane owns one row of the transposed view and reads 64 contiguous bytes (4 x float4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void linear_kernel(const float4* __restrict__ w, size_t n4, float* out) {
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { const float4 v = w[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void colrow_kernel(const float* __restrict__ w, int R, int N, int nsplit, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int col = blockIdx.x * 128 + wave * 32 + li;
    const int chunks = R / 32, c0 = chunks * blockIdx.y / nsplit, c1 = chunks * (blockIdx.y + 1) / nsplit;
    float b[DEPTH][16];
    float acc = 0.f;
    auto load = [&](float (&d)[16], int c) {
        const float* src = w + (size_t)(c * 32 + lh * 16) * N + col;
#pragma unroll
        for (int k = 0; k < 16; ++k) { d[k] = *src; src += N; }
    };
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) load(b[u], c0 + u < c1 ? c0 + u : c1 - 1);
    for (int c = c0; c < c1; c += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (c + u < c1) {
                const int cn = c + u + DEPTH - 1 < c1 ? c + u + DEPTH - 1 : c1 - 1;
                load(b[(u + DEPTH - 1) % DEPTH], cn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc += b[u][k];
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void rowvec_kernel(const float* __restrict__ w, int R, int N, int nsplit, float* out) {
    // transposed use: output index = matrix row; reduction along the row (contiguous)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + li;
    const int chunks = N / 32, c0 = chunks * blockIdx.y / nsplit, c1 = chunks * (blockIdx.y + 1) / nsplit;
    float4 b[DEPTH][4];
    float acc = 0.f;
    auto load = [&](float4 (&d)[4], int c) {
        const float4* src = reinterpret_cast<const float4*>(w + (size_t)row * N + c * 32 + lh * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = src[k];
    };
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) load(b[u], c0 + u < c1 ? c0 + u : c1 - 1);
    for (int c = c0; c < c1; c += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (c + u < c1) {
                const int cn = c + u + DEPTH - 1 < c1 ? c + u + DEPTH - 1 : c1 - 1;
                load(b[(u + DEPTH - 1) % DEPTH], cn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += b[u][k].x + b[u][k].y + b[u][k].z + b[u][k].w;
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int R = 4096, N = 4096, NMAT = argc > 1 ? atoi(argv[1]) : 8;       // NMAT matrices rotate so the 256 MB cache cannot hold them
    const size_t elems = (size_t)R * N;
    float* w; float* out;
    CK(hipMalloc(&w, elems * NMAT * sizeof(float)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(w, 0, elems * NMAT * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < NMAT; ++i) launch(w + elems * i);
        hipEventRecord(e0);
        const int reps = 5 * NMAT;
        for (int i = 0; i < reps; ++i) launch(w + elems * (i % NMAT));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, elems * 4.0 * reps / ms / 1e9);
    };
    timeit("linear 2048 wg", [&](float* m) { linear_kernel<<<2048, 256>>>((const float4*)m, elems / 4, out); });
    timeit("linear 512 wg", [&](float* m) { linear_kernel<<<512, 256>>>((const float4*)m, elems / 4, out); });
    timeit("linear 256 wg", [&](float* m) { linear_kernel<<<256, 256>>>((const float4*)m, elems / 4, out); });
    timeit("colrow depth6 nsplit8", [&](float* m) { colrow_kernel<6><<<dim3(N / 128, 8), 256>>>(m, R, N, 8, out); });
    timeit("colrow depth3 nsplit16", [&](float* m) { colrow_kernel<3><<<dim3(N / 128, 16), 256>>>(m, R, N, 16, out); });
    timeit("colrow depth6 nsplit16", [&](float* m) { colrow_kernel<6><<<dim3(N / 128, 16), 256>>>(m, R, N, 16, out); });
    timeit("colrow depth6 nsplit32", [&](float* m) { colrow_kernel<6><<<dim3(N / 128, 32), 256>>>(m, R, N, 32, out); });
    timeit("rowvec depth6 nsplit8", [&](float* m) { rowvec_kernel<6><<<dim3(R / 128, 8), 256>>>(m, R, N, 8, out); });
    timeit("rowvec depth6 nsplit16", [&](float* m) { rowvec_kernel<6><<<dim3(R / 128, 16), 256>>>(m, R, N, 16, out); });
    timeit("rowvec depth6 nsplit32", [&](float* m) { rowvec_kernel<6><<<dim3(R / 128, 32), 256>>>(m, R, N, 32, out); });
    return 0;
}
