// Probe for the "2 x f16" operand split (round 2): does v_mfma_f32_32x32x16_f16 honour f16 denormal inputs on gfx950, what do the
// conversions produce for small values, and how accurate is  a*b ~ a1*b1 + a1*b2 + a2*b1  (a = a1 + a2 in round-to-nearest f16) against
// the exact "3 x bf16" split (6 products) and a plain fp32 fmaf chain, all measured against an fp64 dot product.
//   hipcc --offload-arch=gfx950 -O3 -o f16x2_probe f16x2_probe.hip && ./f16x2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// A [32][K], B [32][K] fp32 (pre-scaled by the host); D[i][j] = sum_k A[i][k] B[j][k].  One wave.
// mode 0: 3 x bf16 truncation split, 6 products   mode 1: 2 x f16 RN split, 3 products   mode 2: 2 x f16, 4 products
// mode 3: fp32 fmaf chain on the VALU
__global__ void gemm_probe(const float* A, const float* B, float* D, int K, int mode) {
  const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if (mode == 3) {
    for (int e = 0; e < 16; ++e) {
      const int i = (e & 3) + 8 * (e >> 2) + 4 * g, j = r;
      float s = 0.f;
      for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s);
      D[i * 32 + j] = s;
    }
    return;
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[r * K + k0 + 8 * g + e]; b[e] = B[r * K + k0 + 8 * g + e]; }
    if (mode == 0) {
      bf16x8 ap[3], bp[3];
      for (int e = 0; e < 8; ++e) {
        float x = a[e];
        for (int p = 0; p < 3; ++p) {
          const float h = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
          ap[p][e] = (__bf16)h; x -= h;
        }
        x = b[e];
        for (int p = 0; p < 3; ++p) {
          const float h = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
          bp[p][e] = (__bf16)h; x -= h;
        }
      }
      const int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[q]], bp[PB[q]], acc, 0, 0, 0);
    } else {
      f16x8 ap[2], bp[2];
      for (int e = 0; e < 8; ++e) {
        ap[0][e] = (_Float16)a[e]; ap[1][e] = (_Float16)(a[e] - (float)ap[0][e]);
        bp[0][e] = (_Float16)b[e]; bp[1][e] = (_Float16)(b[e] - (float)bp[0][e]);
      }
      if (mode == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bp[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bp[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bp[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bp[0], acc, 0, 0, 0);
    }
  }
  for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * g) * 32 + r] = acc[e];
}

// denormal probe: A = value v in every element (as f16), B = 1 -> D = 16 * v if denormals are honoured
__global__ void denorm_probe(float* out, float v) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)v; b[e] = (_Float16)1.0f; }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}

// rate: register-only MFMA loop, f16 vs bf16
template <int F16>
__global__ __launch_bounds__(512) void rate_loop(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  f16x8 xh, yh; bf16x8 xb, yb;
  for (int e = 0; e < 8; ++e) {
    unsigned h1 = (lane * 8 + e + 1) * 2654435761u, h2 = (lane * 8 + e + 77) * 2246822519u;
    h1 ^= h1 >> 15; h1 *= 2246822519u; h1 ^= h1 >> 13; h2 ^= h2 >> 15; h2 *= 2654435761u; h2 ^= h2 >> 13;
    const float fx = ((int)(h1 >> 8) - (1 << 23)) * (1.0f / (1 << 23)), fy = ((int)(h2 >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.01f;
    xh[e] = (_Float16)fx; yh[e] = (_Float16)fy; xb[e] = (__bf16)fx; yb[e] = (__bf16)fy;
  }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (F16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[a], 0, 0, 0);
        else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[a], 0, 0, 0);
      }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
  float* dout;
  hipMalloc(&dout, 4096);
  const float vals[] = {6.1e-5f, 3.0e-5f, 1.0e-6f, 5.96e-8f, 1.2e-7f};
  for (float v : vals) {
    hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, dout, v);
    float h[2];
    hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost);
    printf("denorm v=%.4e  f16(v)=%.6e  mfma(16 terms)=%.6e  expected=%.6e\n", v, h[1], h[0], 16.0 * h[1]);
  }
  for (int f16 = 0; f16 < 2; ++f16) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (f16) hipLaunchKernelGGL(rate_loop<1>, dim3(1024), dim3(512), 0, 0, dout, 2000);
      else hipLaunchKernelGGL(rate_loop<0>, dim3(1024), dim3(512), 0, 0, dout, 2000);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double fl = 1024.0 * 8 * 2000 * 24 * 2.0 * 32 * 32 * 16;
    printf("rate %s: %.3f ms  %.1f TFLOP/s\n", f16 ? "f16 " : "bf16", ms, fl / ms * 1e-9);
  }
  const int K = 3456;
  std::vector<float> A(32 * K), B(32 * K), D(1024);
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
  // (activation magnitude, weight magnitude, kind) ; kind 1 = relu-like (half zeros), 2 = wide dynamic range (log-uniform over 2^-12..1)
  const double cases[][3] = {{1, 0.02, 0}, {1, 0.02, 1}, {1, 0.02, 2}, {0.01, 0.02, 0}, {1e-4, 0.02, 0}, {100, 0.02, 0}, {1000, 1, 0}, {1, 1e-4, 0}};
  const int ka_list[] = {0, 4};
  for (auto& c : cases)
    for (int ka : ka_list) {
      srand(1234);
      double wmax = 0;
      for (auto& x : B) { x = (float)(nrand() * c[1]); wmax = fmax(wmax, fabs(x)); }
      for (auto& x : A) {
        double v = nrand() * c[0];
        if (c[2] == 1 && v < 0) v = 0;
        if (c[2] == 2) v *= exp2(-12 * urand());
        x = (float)v;
      }
      int ew; frexp(wmax, &ew);                        // wmax in [2^(ew-1), 2^ew)
      const int kw = 14 - ew;                          // scaled max in [2^13, 2^14)
      std::vector<double> ref(1024);
      double refmax = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0;
          for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k];
          ref[i * 32 + j] = s; refmax = fmax(refmax, fabs(s));
        }
      printf("case act=%g w=%g kind=%g ka=%d kw=%d |ref|max=%.3e :", c[0], c[1], c[2], ka, kw, refmax);
      for (int mode = 0; mode < 4; ++mode) {
        std::vector<float> As = A, Bs = B;
        const bool sc = mode == 1 || mode == 2;
        if (sc) { for (auto& x : As) x = ldexpf(x, ka); for (auto& x : Bs) x = ldexpf(x, kw); }
        hipMemcpy(dA, As.data(), As.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, Bs.data(), Bs.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(gemm_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double emax = 0, erms = 0;
        for (int i = 0; i < 1024; ++i) {
          const double v = sc ? ldexp((double)D[i], -(ka + kw)) : D[i];
          const double e = fabs(v - ref[i]);
          emax = fmax(emax, e); erms += e * e;
        }
        printf("  m%d max %.2e rms %.2e", mode, emax / refmax, sqrt(erms / 1024) / refmax);
      }
      printf("\n");
    }
  return 0;
}
