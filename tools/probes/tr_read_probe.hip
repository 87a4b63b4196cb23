#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS tile: [R rows][C cols] of 16-bit ids; every lane passes its own address; dump what each lane gets.
__global__ void k(short* out, int pitch) {
  __shared__ __attribute__((aligned(16))) short t[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 64) t[i] = (short)i;   // value = linear index (row*pitch+col)
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, i16 = lane & 15;
  // per the guide: lane 4q+p of the 16-lane group supplies the address of block row q, columns 4p..4p+3
  const int q = i16 >> 2, p = i16 & 3;
  const int row0 = 8 * g;             // group g reads rows row0 .. row0+3
  const short* addr = &t[(row0 + q) * pitch + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d, 72);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) { if (l % 16 < 3 || l % 16 == 15) printf("lane %2d: %d %d %d %d  (row,col)=(%d,%d) (%d,%d)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], h[l*4]/72, h[l*4]%72, h[l*4+1]/72, h[l*4+1]%72); }
  return 0;
}
