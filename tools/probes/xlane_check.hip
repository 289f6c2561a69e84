// Probe (not part of the product): prints which source lane every cross-lane primitive reads on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto r2 = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
  out[128 + l] = r2[0]; out[192 + l] = r2[1];
  out[256 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x128, 0xf, 0xf, false);  // row_ror:8
  unsigned t = __builtin_amdgcn_update_dpp(999u, l, 0x104, 0xf, 0x5, false);    // row_shl:4 banks 0,2
  t = __builtin_amdgcn_update_dpp(t, l, 0x114, 0xf, 0xa, false);                // row_shr:4 banks 1,3
  out[320 + l] = t;
  out[384 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x4E, 0xf, 0xf, false);
  out[448 + l] = __builtin_amdgcn_update_dpp(999u, l, 0xB1, 0xf, 0xf, false);
}
int main() {
  unsigned* d; hipMalloc(&d, 512 * 4); k<<<1, 64>>>(d); unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"pl32.vdst", "pl32.vsrc", "pl16.vdst", "pl16.vsrc", "row_ror8", "xor4(shl/shr)", "quad[2301]", "quad[1032]"};
  for (int t = 0; t < 8; t++) { printf("%-14s", names[t]); for (int l = 0; l < 64; l++) printf(" %u", h[t * 64 + l]); printf("\n"); }
  return 0;
}
