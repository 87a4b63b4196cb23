// Probe (MI355X): what does `buffer_load_dwordx4 ... lds` (LDS-DMA) write for lanes whose offset fails the buffer
// descriptor's range check -- zeros, or nothing?   hipcc --offload-arch=gfx950 -O2 lds_dma_oob_probe.hip -o /tmp/oob && /tmp/oob
// The attention kernels of mha_sh.hip rely on the answer for the rows of a ragged 64-row tile beyond its length.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))
__global__ void probe(const unsigned* src, unsigned* out, int valid_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned img[256];           // 1 KiB: one wave-instruction
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) img[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, valid_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)img, 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = lane; i < 256; i += 64) out[i] = img[i];
}
int main() {
    unsigned *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out, 40 * 16);                                  // lanes 0..39 in range, 40..63 out of range
    std::vector<unsigned> o(256);
    hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_out = 0, kept_out = 0, other = 0;
    for (int i = 0; i < 256; ++i) {
        if (i < 160) ok_in += o[i] == 0x1000u + i;
        else if (o[i] == 0) ++zero_out; else if (o[i] == 0xdeadbeefu) ++kept_out; else ++other;
    }
    printf("in-range dwords correct %d/160; out-of-range dwords: zero %d, untouched %d, other %d (of 96)\n", ok_in, zero_out, kept_out, other);
    return 0;
}
