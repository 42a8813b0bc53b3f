// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: which scale byte a lane's 32
// values are multiplied by. One wave, one MFMA, checked against a host model in which lane l of
// the first / second operand holds row (l & 15), k-block (l >> 4), and is scaled by byte `opsel`
// of ITS OWN scale register. Prints the maximum deviation per (opsel_a, opsel_b) pair.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OA, int OB>
__global__ void probe(const int* a, const int* b, const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = a[l * 8 + j]; bv[j] = b[l * 8 + j]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, OA, sa[l], OB, sb[l]);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

static float e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 3;   // bit 0: random A scales, bit 1: random B scales
  unsigned char ha[64 * 32], hb[64 * 32];
  int hsa[64], hsb[64];
  srand(5);
  for (int i = 0; i < 64 * 32; ++i) {
    do { ha[i] = rand() & 0xff; } while ((ha[i] & 0x7f) == 0x7f);
    do { hb[i] = rand() & 0xff; } while ((hb[i] & 0x7f) == 0x7f);
  }
  for (int l = 0; l < 64; ++l) {
    hsa[l] = hsb[l] = 0;
    for (int by = 0; by < 4; ++by) {
      hsa[l] |= ((mode & 1) ? (120 + rand() % 12) : 127) << (8 * by);
      hsb[l] |= ((mode & 2) ? (122 + rand() % 10) : 127) << (8 * by);
    }
  }
  int *da, *db, *dsa, *dsb;
  float* dout;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb));
  hipMalloc(&dsa, sizeof(hsa)); hipMalloc(&dsb, sizeof(hsb)); hipMalloc(&dout, 256 * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice);
  hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
  float hout[256];
  for (int combo = 0; combo < 4; ++combo) {
    const int oa = combo, ob = 3 - combo;
    if (combo == 0) probe<0, 3><<<1, 64>>>(da, db, dsa, dsb, dout);
    if (combo == 1) probe<1, 2><<<1, 64>>>(da, db, dsa, dsb, dout);
    if (combo == 2) probe<2, 1><<<1, 64>>>(da, db, dsa, dsb, dout);
    if (combo == 3) probe<3, 0><<<1, 64>>>(da, db, dsa, dsb, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        // D[row = 4 (l >> 4) + i][col = l & 15] = sum_k A[row][k] B[k][col]
        const int row = 4 * (l >> 4) + i, col = l & 15;
        double ref = 0;
        for (int q = 0; q < 4; ++q) {
          const int la = q * 16 + row, lb = q * 16 + col;      // lanes holding that k-block
          const double fa = ldexp(1.0, ((hsa[la] >> (8 * oa)) & 255) - 127);
          const double fb = ldexp(1.0, ((hsb[lb] >> (8 * ob)) & 255) - 127);
          double part = 0;
          for (int e = 0; e < 32; ++e) part += (double)e4m3(ha[la * 32 + e]) * e4m3(hb[lb * 32 + e]);
          ref += part * fa * fb;
        }
        worst = fmax(worst, fabs(ref - hout[l * 4 + i]));
        scale = fmax(scale, fabs(ref));
      }
    printf("opsel_a=%d opsel_b=%d  max|dev|=%g  max|ref|=%g\n", oa, ob, worst, scale);
  }
  return 0;
}
