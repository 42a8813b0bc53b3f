// For element e of lane L of the first operand of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3): the
// lane whose scale byte multiplies it. The operand is zero but for that element (1.0), the second
// operand is all ones with unit scales, lane l carries the scale 2^l: D[row L & 15][*] = 2^x.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(int L, int e, float* out, int second) {
  const int l = threadIdx.x;
  i32x8 ones, x;
  for (int j = 0; j < 8; ++j) { ones[j] = 0x38383838; x[j] = 0; }
  if (l == L) x[e >> 2] = 0x38 << (8 * (e & 3));
  const int s = 127 + l;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (!second) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, ones, c, 0, 0, 0, s, 0, 0x7f);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, x, c, 0, 0, 0, 0x7f, 0, s);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

int main() {
  float* dout;
  float hout[256];
  (void)hipMalloc(&dout, sizeof(hout));
  for (int second = 0; second < 2; ++second)
    for (int L = 0; L < 64; L += 16 + (L == 16 ? -15 : 0) + (L == 17 ? 15 - 1 : 0)) {
      printf("operand %d lane %2d:", second, L);
      for (int e = 0; e < 32; ++e) {
        probe<<<1, 64>>>(L, e, dout, second);
        (void)hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        float v = 0;
        for (int i = 0; i < 256; ++i) v = fmaxf(v, hout[i]);
        printf(" %d", (int)lrintf(log2f(v)));
      }
      printf("\n");
    }
  return 0;
}
