// Measurement aid: does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) write ZEROS for lanes whose offset is out of range, and does soffset take part in the range check?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out, const unsigned* offs, unsigned soff) {
    __shared__ unsigned lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(uintptr_t)lds, 16, offs[threadIdx.x], soff, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    const int N = 4096;
    std::vector<unsigned> h(N); for (int i = 0; i < N; ++i) h[i] = i;
    unsigned *d, *o, *of; hipMalloc(&d, N * 4); hipMalloc(&o, 512 * 4); hipMalloc(&of, 64 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> offs(64);
    for (int i = 0; i < 64; ++i) offs[i] = (i % 3 == 1) ? 0xFFFFFFF0u : (i % 3 == 2 ? 8192u - 32 + 16 * (i / 3 % 4) : 64u * i);
    hipMemcpy(of, offs.data(), 256, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 1024u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 8192u, o, of, soff);
        std::vector<unsigned> r(512); hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        printf("soffset %u (buffer = 8192 bytes of dword indices)\n", soff);
        for (int i = 0; i < 12; ++i) printf("  lane %2d voff %10u -> %08x %08x %08x %08x\n", i, offs[i], r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    }
    return 0;
}
