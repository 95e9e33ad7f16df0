// Probe: buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor): lane -> LDS mapping, out-of-range lanes, LDS bases
// beyond 64 KiB.  hipcc --offload-arch=gfx950 tools/dma_probe.hip -o tools/dma_probe.bin && tools/dma_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* a, int nbytes, float* out, int lds_base, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) ((float*)lds)[i] = -1.0f;        // 160 KiB of -1
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
    int off = lane * 16;
    if (lane % 5 == 3) off = (int)0x80000000;                               // out of range
    if (mode == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + lds_base), 16, off, 0, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds + lds_base), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = ((float*)(lds + lds_base))[i];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    float *a, *o;
    hipMalloc(&a, 1024); hipMalloc(&o, 1024);
    hipMemcpy(a, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 2; ++mode)
        for (int base : {0, 4096, 70000 / 16 * 16, 102400, 150000 / 16 * 16}) {
            k<<<1, 64, 160 * 1024>>>(a, 1024, o, base, mode);
            std::vector<float> r(256);
            hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
            int ok = 0, zero = 0, stale = 0, other = 0;
            for (int i = 0; i < 256; ++i) {
                const int lane = i / 4;
                const bool oob = mode == 0 && lane % 5 == 3;
                if (!oob && r[i] == h[i]) ++ok; else if (r[i] == 0.f) ++zero; else if (r[i] == -1.f) ++stale; else ++other;
            }
            printf("mode %d (%s) lds base %6d: %3d correct, %3d zero, %3d stale(-1), %3d other; hip: %s\n", mode, mode ? "global_load_lds" : "buffer_load lds",
                   base, ok, zero, stale, other, hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
