#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4* lp;
__global__ void k(unsigned short* out, int pitch_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, t = l & 15, g = l >> 4;
  // group g reads rows 4g..4g+3, cols 0..15 of a row-major [64][pitch] image
  const int row = 4 * g + (t >> 2), col = 4 * (t & 3);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + row * pitch_elems + col));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  int pitches[2] = {16, 64};
  for (int pi = 0; pi < 2; ++pi) {
    const int P = pitches[pi];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, P);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pitch %d\n", P);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, g = l >> 4;
      for (int j = 0; j < 4; ++j) {
        const int expect = (4 * g + j) * P + i;   // column i of rows 4g+j
        if (h[l * 4 + j] != expect) ok = 0;
      }
      if (l < 20 || l % 16 == 0) printf("lane %2d: %d %d %d %d  (row,col)= (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3],
        h[l*4]/P, h[l*4]%P, h[l*4+1]/P, h[l*4+1]%P, h[l*4+2]/P, h[l*4+2]%P, h[l*4+3]/P, h[l*4+3]%P);
    }
    printf("matches 'lane i gets column i of the 4x16 block, elem j = row j': %s\n", ok ? "YES" : "NO");
  }
  return 0;
}
