// Which lane's scale byte multiplies which (row, k-block) of the first operand of
// v_mfma_scale_f32_16x16x128_f8f6f4? All data are 1.0 (e4m3 0x38), the second operand's scales are
// unit; in run (t, byte) the lanes of group t = l >> 4 carry 2^(l & 15) in byte `byte` of their
// scale register, every other byte of every lane 2^-100. D[row][col] / 32 then spells, in binary,
// which lanes of group t supply blocks of that row.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OA>
__global__ void probe(const int* sa, float* out, int which) {
  const int l = threadIdx.x;
  i32x8 ones;
  for (int j = 0; j < 8; ++j) ones[j] = 0x38383838;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (which == 0)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, c, 0, 0, OA, sa[l], 0, 0x7f7f7f7f);
  else
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, c, 0, 0, 0, 0x7f7f7f7f, OA, sa[l]);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

int main() {
  int hs[64];
  int* ds;
  float* dout;
  float hout[256];
  (void)hipMalloc(&ds, sizeof(hs));
  (void)hipMalloc(&dout, sizeof(hout));
  for (int which = 0; which < 2; ++which)
    for (int byte = 0; byte < 4; ++byte)
      for (int t = 0; t < 4; ++t) {
        for (int l = 0; l < 64; ++l) {
          hs[l] = 0x1b1b1b1b;                                   // 2^-100 everywhere
          if ((l >> 4) == t) hs[l] = (hs[l] & ~(0xff << (8 * byte))) | ((127 + (l & 15)) << (8 * byte));
        }
        (void)hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
        if (byte == 0) probe<0><<<1, 64>>>(ds, dout, which);
        if (byte == 1) probe<1><<<1, 64>>>(ds, dout, which);
        if (byte == 2) probe<2><<<1, 64>>>(ds, dout, which);
        if (byte == 3) probe<3><<<1, 64>>>(ds, dout, which);
        (void)hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        printf("operand %d opsel %d group %d:", which, byte, t);
        // D[row = 4 (l >> 4) + i][col = l & 15]; print per row (col 0) and per col (row 0)
        for (int row = 0; row < 16; ++row) printf(" %g", hout[((row >> 2) * 16 + 0) * 4 + (row & 3)] / 32);
        printf(" |");
        for (int col = 0; col < 16; ++col) printf(" %g", hout[col * 4] / 32);
        printf("\n");
      }
  return 0;
}
