// Probe: evaluates the SineGen interpolation formula on the device for samples [i0, i1) and prints the raw float bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ float sine_interp_raw(const float* tmp, int T, float scale, long long i) {
  const float src = __fmul_rn(scale, (float)i);
  int i0 = (int)src;
  const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
  float l1 = __fsub_rn(src, (float)i0); l1 = fminf(fmaxf(l1, 0.f), 1.f);
  const float l0 = __fsub_rn(1.f, l1);
  return __fmaf_rn(l0, tmp[i0], __fmul_rn(l1, tmp[i1]));
}
__global__ void k(const float* tmp, int T, float scale, long long a, int n, float* out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = sine_interp_raw(tmp, T, scale, a + j);
}
int main(int argc, char** argv) {
  // reads tmp (float32 binary) from argv[1], T = file size / 4, upp = 400
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  int T = sz / 4; std::vector<float> h(T); fread(h.data(), 4, T, f); fclose(f);
  long long N = (long long)T * 400; float scale = (float)(T - 1) / (float)(N - 1);
  float *d, *o; hipMalloc(&d, sz); hipMemcpy(d, h.data(), sz, hipMemcpyHostToDevice);
  hipMalloc(&o, N * 4);
  k<<<(N + 255) / 256, 256>>>(d, T, scale, 0, (int)N, o);
  std::vector<float> r(N); hipMemcpy(r.data(), o, N * 4, hipMemcpyDeviceToHost);
  FILE* g = fopen(argv[2], "wb"); fwrite(r.data(), 4, N, g); fclose(g);
  printf("wrote %lld values, scale %.9g\n", N, scale);
  return 0;
}
