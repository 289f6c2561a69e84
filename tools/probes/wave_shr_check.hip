// Probe (not part of the product): DPP wave_shr:1 / wave_shl:1 lane mapping on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  out[l] = __builtin_amdgcn_update_dpp(777u, l, 0x138, 0xf, 0xf, false);       // wave_shr:1
  out[64 + l] = __builtin_amdgcn_update_dpp(777u, l, 0x130, 0xf, 0xf, false);  // wave_shl:1
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 2; t++) { printf("%s", t ? "wave_shl1" : "wave_shr1"); for (int l = 0; l < 64; l++) printf(" %u", h[t * 64 + l]); printf("\n"); }
  return 0;
}
