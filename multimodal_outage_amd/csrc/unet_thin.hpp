// Streaming kernels for the UNet's thin pointwise layers (OutConv 4 -> out_ch, unet.py:86-92, at full image
// resolution): with <= 16 channels on either side these are pure HBM streams -- one pass over the input planes, one
// over the output planes -- and the implicit-GEMM engine (64x64 / 32x128 tiles padded 8..16x, split-K slabs) spent
// 2.5 ms of a 24 ms config-3 step on them.  A thread owns 4 neighbouring pixels (one 16-byte piece per plane).
#pragma once
#include "mo_common.h"
#include "unet_direct.hpp"     // ua_ld4 / ua_st4: fp32 or bf16 activation storage

struct UtArgs {
  const float* in; long is; int Ci;
  const float* sc; const float* sh; int relu, gsize;     // folded group-BatchNorm affine (+ ReLU) on load, or null
  const float* W;  const float* b;                      // (Co, Ci) row-major; transposed use: see kernels
  float* out; long os; int Co;
  long n_img; int HW;
  int bfi, bfo;                                          // storage of in / out: 0 fp32, 1 bf16
};

// out[img][co][p] = b[co] + sum_ci Wm[co][ci] * act(in[img][ci][p]);  TRANS: Wm[co][ci] = W[ci][co] (data gradient of
// the conv whose weight is W: `in` = dout with Ci = its Co, `out` = din)
template <int CO, bool TRANS>
__global__ __launch_bounds__(256) void ut_conv1x1_kernel(UtArgs a) {
  __shared__ float wsm[16][CO + 1];
  __shared__ float bsm[CO];
  const int tid = threadIdx.x;
  for (int i = tid; i < a.Ci * CO; i += 256) {
    const int ci = i / CO, co = i - ci * CO;
    float w = 0.f;
    if (co < a.Co) w = TRANS ? a.W[(long)ci * a.Co + co] : a.W[(long)co * a.Ci + ci];
    wsm[ci][co] = w;
  }
  if (tid < CO) bsm[tid] = (a.b && tid < a.Co) ? a.b[tid] : 0.f;
  __syncthreads();
  const long img = blockIdx.y;
  const long grp = img / a.gsize;
  for (int p = (blockIdx.x * 256 + tid) * 4; p < a.HW; p += gridDim.x * 1024) {
    float4 acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) { const float b = bsm[co]; acc[co] = make_float4(b, b, b, b); }
    for (int ci = 0; ci < a.Ci; ++ci) {
      float4 v = ua_ld4(a.in, img * a.is + (long)ci * a.HW + p, a.bfi);
      if (a.sc) {
        const float s = a.sc[grp * a.Ci + ci], t = a.sh[grp * a.Ci + ci];
        v.x = v.x * s + t; v.y = v.y * s + t; v.z = v.z * s + t; v.w = v.w * s + t;
      }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float w = wsm[ci][co];
        acc[co].x += w * v.x; acc[co].y += w * v.y; acc[co].z += w * v.z; acc[co].w += w * v.w;
      }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co)
      if (co < a.Co) ua_st4(a.out, img * a.os + (long)co * a.HW + p, acc[co], a.bfo);
  }
}

// dW[co][ci] = sum dout[img][co][p] * act(in[img][ci][p]);  db[co] = sum dout.  One slab row of Co*Ci + Co partial
// sums per workgroup (fixed order: thread-private sums over a strided pixel set, wave shuffles, the four waves through
// LDS); the caller's slab reduction adds the rows.
struct UtWgArgs {
  const float* dout; long dos; int Co;
  const float* in; long is; int Ci;
  const float* sc; const float* sh; int relu, gsize;
  float* slab;            // [gridDim.y * gridDim.x][Co*Ci + Co]
  long n_img; int HW, img_per_wg;
  int bfd, bfi;           // storage of dout / in
};
template <int CO, int CI>
__global__ __launch_bounds__(256) void ut_wgrad1x1_kernel(UtWgArgs a) {
  __shared__ float red[4][CO * CI + CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[CO][CI], accb[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) acc[co][ci] = 0.f;
  }
  const long img0 = (long)blockIdx.y * a.img_per_wg, img1 = min(img0 + a.img_per_wg, a.n_img);
  for (long img = img0; img < img1; ++img) {
    const long grp = img / a.gsize;
    for (int p = (blockIdx.x * 256 + tid) * 4; p < a.HW; p += gridDim.x * 1024) {
      float4 x[CI];
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < a.Ci) {
          v = ua_ld4(a.in, img * a.is + (long)ci * a.HW + p, a.bfi);
          if (a.sc) {
            const float s = a.sc[grp * a.Ci + ci], t = a.sh[grp * a.Ci + ci];
            v.x = v.x * s + t; v.y = v.y * s + t; v.z = v.z * s + t; v.w = v.w * s + t;
          }
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        x[ci] = v;
      }
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        if (co < a.Co) {
          const float4 d = ua_ld4(a.dout, img * a.dos + (long)co * a.HW + p, a.bfd);
          accb[co] += (d.x + d.y) + (d.z + d.w);
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) acc[co][ci] += d.x * x[ci].x + d.y * x[ci].y + d.z * x[ci].z + d.w * x[ci].w;
        }
      }
    }
  }
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int ci = 0; ci <= CI; ++ci) {
      float v = (ci < CI) ? acc[co][ci < CI ? ci : 0] : accb[co];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wave][ci < CI ? co * CI + ci : CO * CI + co] = v;
    }
  }
  __syncthreads();
  const int nout = a.Co * a.Ci + a.Co;
  float* row = a.slab + ((long)blockIdx.y * gridDim.x + blockIdx.x) * nout;
  for (int i = tid; i < nout; i += 256) {
    int src;
    if (i < a.Co * a.Ci) { const int co = i / a.Ci, ci = i - co * a.Ci; src = co * CI + ci; }
    else src = CO * CI + (i - a.Co * a.Ci);
    row[i] = (red[0][src] + red[1][src]) + (red[2][src] + red[3][src]);
  }
}
