// Hardware probe (round 3): the lane mapping of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950, which the register
// epilogue of vit_gemm_ws.h relies on.  Build: hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16 a'", "swap16 b'", "swap32 a'", "swap32 b'"};
  int bad = 0;
  for (int t = 0; t < 4; ++t) {
    printf("%s:", names[t]);
    for (int l = 0; l < 64; ++l) printf(" %u", h[t * 64 + l]);
    printf("\n");
  }
  for (int l = 0; l < 64; ++l) {          // the semantics vit_gemm_ws.h assumes (and tests/emu mirrors)
    const unsigned a16 = ((l >> 4) & 1) ? 100 + l - 16 : l, b16 = ((l >> 4) & 1) ? 100 + l : l + 16;
    const unsigned a32 = l >= 32 ? 100 + l - 32 : l, b32 = l >= 32 ? 100 + l : l + 32;
    bad += h[l] != a16 || h[64 + l] != b16 || h[128 + l] != a32 || h[192 + l] != b32;
  }
  printf(bad ? "MISMATCH with the assumed semantics in %d lanes\n" : "semantics as assumed (%d mismatches)\n", bad);
  return bad != 0;
}
