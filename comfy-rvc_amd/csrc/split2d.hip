// Padded 2-D split-resident images (gfx950 only): the layout the split-resident GEMM kernel (conv_x3s.hip) runs RMVPE's 3 x 3 convolutions on
// (reference lib/rmvpe.py:233-428).  An H x W plane is stored with row pitch W + 2 - one zero column on either side - so that position
// p = h (W + 2) + w + 1 and a 3 x 3 tap (dh, dw) is the constant row offset dh (W + 2) + dw into the image: the horizontal zero padding is
// data, the vertical one the image's zero margins.  Every tensor exists as fp32 [C][H (W + 2)] (residuals, pooling) and / or as the bf16
// hi / lo image [16-channel chunk][hi | lo][8-channel half][margin + p][8 ch].  The kernels here are the level changes of the U-Net in that
// layout: AvgPool2d(2) (plain or padded input), the 2 x 2 phase interleave behind a ConvTranspose2d, and plain <-> padded conversion.
// One thread = one output position of one group of 8 channels = one 16-byte image row (+ 8 fp32 values); consecutive threads are
// consecutive positions.
#include "rvc_internal.h"
#include "ops.h"

namespace rvc {

typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8_store(const float (&v)[8], unsigned char* img, long long tp, int margin, int g, long long p) {
  u32x4p hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 ah = (__bf16)v[2 * j], bh = (__bf16)v[2 * j + 1];
    const __bf16 al = (__bf16)(v[2 * j] - (float)ah), bl = (__bf16)(v[2 * j + 1] - (float)bh);
    hi[j] = (unsigned)__builtin_bit_cast(unsigned short, ah) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
    lo[j] = (unsigned)__builtin_bit_cast(unsigned short, al) | ((unsigned)__builtin_bit_cast(unsigned short, bl) << 16);
  }
  unsigned char* row = img + (((long long)(g >> 1) * 4 + (g & 1)) * tp + margin + p) * 16;
  *reinterpret_cast<u32x4p*>(row) = hi;
  *reinterpret_cast<u32x4p*>(row + tp * 32) = lo;
}

// AvgPool2d(2, 2): x [C][H][xpitch] (xpitch = W: plain, W + 2: padded, first real column at xoff) -> padded level (H / 2) x (W / 2)
__global__ __launch_bounds__(256) void pool2_pad_kernel(const float* __restrict__ x, long long ldx, int xpitch, int xoff, int C, int Ho, int Wo,
                                                        float* __restrict__ y, long long ldy, unsigned char* __restrict__ img, long long tp, int margin) {
  const int Wp = Wo + 2;
  const long long P = (long long)Ho * Wp;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (p >= P) return;
  const int h = (int)(p / Wp), wq = (int)(p - (long long)h * Wp);
  float v[8];
  if (wq == 0 || wq == Wp - 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  } else {
    const int w = wq - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = g * 8 + j;
      const float* q = x + (long long)c * ldx + (long long)(2 * h) * xpitch + xoff + 2 * w;
      v[j] = c < C ? (q[0] + q[1] + q[xpitch] + q[xpitch + 1]) * 0.25f : 0.f;
    }
  }
  if (y) {
#pragma unroll
    for (int j = 0; j < 8; ++j) if (g * 8 + j < C) y[(long long)(g * 8 + j) * ldy + p] = v[j];
  }
  if (img) split8_store(v, img, tp, margin, g, p);
}
void pool2_pad_split(hipStream_t s, const float* x, long long ldx, bool x_padded, int C, int H, int W, float* y, long long ldy, unsigned char* img,
                     long long tp, int margin) {
  RVC_REQUIRE((C & 7) == 0 && (!img || (C & 15) == 0), "pool2_pad_split: channels must be a multiple of 16 for the image");
  const int Ho = H / 2, Wo = W / 2;
  const long long P = (long long)Ho * (Wo + 2);
  hipLaunchKernelGGL(pool2_pad_kernel, dim3((unsigned)((P + 255) / 256), C / 8), dim3(256), 0, s, x, ldx, x_padded ? W + 2 : W, x_padded ? 1 : 0, C, Ho, Wo,
                     y, ldy, img, tp, margin);
}

// out[c][2 h + a][2 w + b] = ph[(a 2 + b) Co + c][h][w]: ph padded (pitch W + 2) at level H x W; out at level 2 H x 2 W, padded fp32 (+ image) or plain fp32
__global__ __launch_bounds__(256) void interleave2_pad_kernel(const float* __restrict__ ph, long long ldp, int Co, int H, int W, float* __restrict__ y, long long ldy,
                                                              int ypad, unsigned char* __restrict__ img, long long tp, int margin) {
  const int W2 = 2 * W, Wp2 = W2 + 2, Wpi = W + 2;
  const long long P = (long long)(2 * H) * Wp2;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (p >= P) return;
  const int yy = (int)(p / Wp2), wq = (int)(p - (long long)yy * Wp2);
  const bool pad = wq == 0 || wq == Wp2 - 1;
  float v[8];
  if (pad) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  } else {
    const int xx = wq - 1;
    const int phase = (yy & 1) * 2 + (xx & 1);
    const long long src = (long long)(yy >> 1) * Wpi + (xx >> 1) + 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = g * 8 + j; v[j] = c < Co ? ph[((long long)phase * Co + c) * ldp + src] : 0.f; }
  }
  if (y) {
    if (ypad) {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (g * 8 + j < Co) y[(long long)(g * 8 + j) * ldy + p] = v[j];
    } else if (!pad) {
      const long long q = (long long)yy * W2 + (wq - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) if (g * 8 + j < Co) y[(long long)(g * 8 + j) * ldy + q] = v[j];
    }
  }
  if (img) split8_store(v, img, tp, margin, g, p);
}
void interleave2_pad_split(hipStream_t s, const float* ph, long long ldp, int Co, int H, int W, float* y, long long ldy, bool y_padded, unsigned char* img,
                           long long tp, int margin) {
  RVC_REQUIRE((Co & 7) == 0 && (!img || (Co & 15) == 0), "interleave2_pad_split: channels must be a multiple of 16 for the image");
  const long long P = (long long)(2 * H) * (2 * W + 2);
  hipLaunchKernelGGL(interleave2_pad_kernel, dim3((unsigned)((P + 255) / 256), Co / 8), dim3(256), 0, s, ph, ldp, Co, H, W, y, ldy, y_padded ? 1 : 0, img, tp, margin);
}

// plain [C][H][W] -> padded fp32 [C][H (W + 2)] and / or the image
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ x, long long ldx, int C, int H, int W, float* __restrict__ y, long long ldy,
                                                    unsigned char* __restrict__ img, long long tp, int margin) {
  const int Wp = W + 2;
  const long long P = (long long)H * Wp;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (p >= P) return;
  const int h = (int)(p / Wp), wq = (int)(p - (long long)h * Wp);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int c = g * 8 + j; v[j] = (wq == 0 || wq == Wp - 1 || c >= C) ? 0.f : x[(long long)c * ldx + (long long)h * W + wq - 1]; }
  if (y) {
#pragma unroll
    for (int j = 0; j < 8; ++j) if (g * 8 + j < C) y[(long long)(g * 8 + j) * ldy + p] = v[j];
  }
  if (img) split8_store(v, img, tp, margin, g, p);
}
void pad2d_split(hipStream_t s, const float* x, long long ldx, int C, int H, int W, float* y, long long ldy, unsigned char* img, long long tp, int margin) {
  RVC_REQUIRE((C & 7) == 0 && (!img || (C & 15) == 0), "pad2d_split: channels must be a multiple of 16 for the image");
  const long long P = (long long)H * (W + 2);
  hipLaunchKernelGGL(pad2d_kernel, dim3((unsigned)((P + 255) / 256), C / 8), dim3(256), 0, s, x, ldx, C, H, W, y, ldy, img, tp, margin);
}
// padded fp32 -> plain
__global__ __launch_bounds__(256) void unpad2d_kernel(const float* __restrict__ x, long long ldx, int C, int H, int W, float* __restrict__ y, long long ldy) {
  const long long n = (long long)H * W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= n) return;
  const int h = (int)(i / W), w = (int)(i - (long long)h * W);
  y[(long long)c * ldy + i] = x[(long long)c * ldx + (long long)h * (W + 2) + w + 1];
}
void unpad2d(hipStream_t s, const float* x, long long ldx, int C, int H, int W, float* y, long long ldy) {
  hipLaunchKernelGGL(unpad2d_kernel, dim3((unsigned)(((long long)H * W + 255) / 256), C), dim3(256), 0, s, x, ldx, C, H, W, y, ldy);
}

}  // namespace rvc
