// Direct 3x3 convolutions for the UNet's thin layers (unet.py:44-48: 13->4->4 at 256^2, 4->8->8 at 128^2, ...).
// At 4..32 channels these are not contractions worth an im2col GEMM: the implicit-GEMM form spent its time on
// per-element index arithmetic.  Here a workgroup owns a spatial tile of one image: the activated input halo
// tile (folded BatchNorm affine + ReLU applied on the way in, zero padding after activation) is staged in LDS
// eight channels at a time, each thread keeps 4 neighbouring pixels x all output channels in registers and
// walks the 9 taps with the weights read as LDS broadcasts.
#pragma once
#include "mo_common.h"

#define UD_CK 4          // input channels per LDS stage

// ---- activation storage: fp32, or bf16 ("bf16 mode" of BASELINE config 3: the raw conv outputs and their gradients at
// the large resolutions live in HBM as bf16, every kernel widens on load, computes in fp32 and narrows on store with
// round-to-nearest-even).  idx = element index; idx % 4 == 0 for the 4-wide forms.
__device__ __forceinline__ float ua_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float ua_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float4 ua_ld4(const void* base, long idx, int bf) {
  if (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(ua_lo(u.x), ua_hi(u.x), ua_lo(u.y), ua_hi(u.y));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
__device__ __forceinline__ float ua_ld1(const void* base, long idx, int bf) {
  if (bf) return __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(base)[idx]) << 16);
  return reinterpret_cast<const float*>(base)[idx];
}
__device__ __forceinline__ unsigned ua_pack2(float a, float b) {
  __bf16 ta = (__bf16)a, tb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ta) | ((unsigned)__builtin_bit_cast(unsigned short, tb) << 16);
}
__device__ __forceinline__ void ua_st4(void* base, long idx, float4 v, int bf) {
  if (bf) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = make_uint2(ua_pack2(v.x, v.y), ua_pack2(v.z, v.w));
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
}

struct UdConvArgs {
  const float* in0; const float* sc0; const float* sh0; long is0; int C0, relu0;
  const float* in1; const float* sc1; const float* sh1; long is1; int C1, relu1;
  const float* W;          // (Co, C0+C1, 3, 3)
  float* out; long os;
  int Co, H, Wd, gsize;
  float* stats;            // optional [img][tile][Co][2]: per-tile (sum, sum of squares) of the raw outputs
  int bf0, bf1, bfo;       // storage of in0 / in1 / out: 0 fp32, 1 bf16 (strides are in ELEMENTS either way)
  const long* off0 = nullptr;   // optional per-image ELEMENT offsets of in0 (image img at in0 + off0[img] instead of
                                // in0 + img * is0): the network input as the permuted batch view of lit.py:31, no copy
  int cosplit = 1;              // ux_conv3x3_mfma_kernel: workgroups per tile, each with its own 16*MB output channels
  long n_img = 0;               // (cosplit > 1: the grid is one-dimensional and padded; workgroups past the last tile leave)
};
template <class A> __device__ __forceinline__ long ud_base0(const A& a, long img) { return a.off0 ? a.off0[img] : img * a.is0; }

// activated value of channel c (of the concat) at (y, x) of image img; zero outside the image
__device__ __forceinline__ float ud_act(const UdConvArgs& a, long img, long grp, int c, int y, int x) {
  if ((unsigned)y >= (unsigned)a.H || (unsigned)x >= (unsigned)a.Wd) return 0.f;
  const bool first = c < a.C0;
  const int cc = first ? c : c - a.C0;
  const float* base = first ? a.in0 : a.in1;
  const long ib = first ? ud_base0(a, img) : img * a.is1;
  float v = ua_ld1(base, ib + ((long)cc * a.H + y) * a.Wd + x, first ? a.bf0 : a.bf1);
  const float* sc = first ? a.sc0 : a.sc1;
  if (sc) {
    const float* sh = first ? a.sh0 : a.sh1;
    const int C = first ? a.C0 : a.C1;
    const long gi = grp * C + cc;
    v = v * sc[gi] + sh[gi];
  }
  if (first ? a.relu0 : a.relu1) v = fmaxf(v, 0.f);
  return v;
}


// Stage the activated halo tile of channel `c` (of the concat) into LDS rows of stride LDT: the TW interior
// columns as aligned float4 (LDS column 4 + k), the left / right halo columns as scalars (LDS columns 3 and
// TW + 4); rows y0-1 .. y0+TH.  Wd % 4 == 0, so an interior quad is all inside or all outside the image.
template <int TH, int TW, int LDT>
__device__ __forceinline__ void ud_stage_halo(float* tile /* [nc][TH+2][LDT] */, const UdConvArgs& a, long img, long grp,
                                              int c_first, int nc, int y0, int x0, int tid) {
  constexpr int Q = TW / 4;                                     // float4 per row
  for (int idx = tid; idx < nc * (TH + 2) * Q; idx += 256) {
    const int c = idx / ((TH + 2) * Q), r = idx - c * ((TH + 2) * Q);
    const int yy = r / Q, q = r - yy * Q;
    const int y = y0 + yy - 1, x = x0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)y < (unsigned)a.H && x < a.Wd) {
      const int cg = c_first + c;
      const bool first = cg < a.C0;
      const int cc = first ? cg : cg - a.C0;
      const float* base = first ? a.in0 : a.in1;
      const long ib = first ? ud_base0(a, img) : img * a.is1;
      v = ua_ld4(base, ib + ((long)cc * a.H + y) * a.Wd + x, first ? a.bf0 : a.bf1);
      const float* sc = first ? a.sc0 : a.sc1;
      if (sc) {
        const float* sh = first ? a.sh0 : a.sh1;
        const long gi = grp * (first ? a.C0 : a.C1) + cc;
        const float s_ = sc[gi], t_ = sh[gi];
        v.x = v.x * s_ + t_; v.y = v.y * s_ + t_; v.z = v.z * s_ + t_; v.w = v.w * s_ + t_;
      }
      if (first ? a.relu0 : a.relu1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    *reinterpret_cast<float4*>(&tile[(c * (TH + 2) + yy) * LDT + 4 + 4 * q]) = v;
  }
  for (int idx = tid; idx < nc * (TH + 2) * 2; idx += 256) {
    const int c = idx / ((TH + 2) * 2), r = idx - c * ((TH + 2) * 2);
    const int yy = r >> 1, side = r & 1;
    tile[(c * (TH + 2) + yy) * LDT + (side ? TW + 4 : 3)] =
        ud_act(a, img, grp, c_first + c, y0 + yy - 1, side ? x0 + TW : x0 - 1);
  }
}

// TH x TW output pixels per workgroup (256 threads, 4 pixels along x each): (16,64) or (32,32)
template <int CO, int TH, int TW>
__global__ __launch_bounds__(256) void ud_conv3x3_kernel(UdConvArgs a) {
  constexpr int LDT = TW + 8;                                   // halo row: col 3 = x0-1, cols 4.. = interior (16-B aligned)
  __shared__ __attribute__((aligned(16))) float tile[UD_CK][TH + 2][LDT];
  __shared__ __attribute__((aligned(16))) float wsm[UD_CK][9][CO];
  const int tid = threadIdx.x;
  constexpr int GX = TW / 4;
  const int tx4 = (tid % GX) * 4, ty = tid / GX;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const long img = blockIdx.z;
  const long grp = img / a.gsize;
  const int Ci = a.C0 + a.C1;
  float acc[CO][4];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  for (int c0 = 0; c0 < Ci; c0 += UD_CK) {
    const int nc = min(UD_CK, Ci - c0);
    __syncthreads();                                            // previous stage fully consumed
    ud_stage_halo<TH, TW, LDT>(&tile[0][0][0], a, img, grp, c0, nc, y0, x0, tid);
    for (int idx = tid; idx < nc * 9 * CO; idx += 256) {
      const int c = idx / (9 * CO), r = idx - c * (9 * CO), tap = r / CO, co = r - tap * CO;
      wsm[c][tap][co] = (co < a.Co) ? a.W[((long)co * Ci + c0 + c) * 9 + tap] : 0.f;
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float seg[6];
        const float* row = &tile[c][ty + ky][tx4 + 3];
#pragma unroll
        for (int q = 0; q < 6; ++q) seg[q] = row[q];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float w = wsm[c][ky * 3 + kx][co];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[co][p] += w * seg[p + kx];
          }
        }
      }
    }
  }
  const int y = y0 + ty, x = x0 + tx4;
  const bool valid = y < a.H && x < a.Wd;                        // Wd % 4 == 0: a quad is all in or all out
  if (valid) {
#pragma unroll
    for (int co = 0; co < CO; ++co)
      if (co < a.Co)
        ua_st4(a.out, img * a.os + ((long)co * a.H + y) * a.Wd + x,
               make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]), a.bfo);
  }
  if (a.stats) {
    // BatchNorm statistics of this tile straight from the accumulators (fixed order: 4 pixels, wave shuffles, the four
    // waves through LDS) -- the separate pass over the conv output is gone
    __shared__ float red[4][2 * CO];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float s1 = valid ? (acc[co][0] + acc[co][1]) + (acc[co][2] + acc[co][3]) : 0.f;
      float s2 = valid ? (acc[co][0] * acc[co][0] + acc[co][1] * acc[co][1]) + (acc[co][2] * acc[co][2] + acc[co][3] * acc[co][3]) : 0.f;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      if (lane == 0) { red[wave][2 * co] = s1; red[wave][2 * co + 1] = s2; }
    }
    __syncthreads();
    if (tid < 2 * CO && (tid >> 1) < a.Co) {
      const long tile = (long)blockIdx.y * gridDim.x + blockIdx.x, ntile = (long)gridDim.x * gridDim.y;
      a.stats[((img * ntile + tile) * a.Co + (tid >> 1)) * 2 + (tid & 1)] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the thin 3x3 convs: dW[co][ci][tap] = sum_{img,y,x} dy[img][co][y][x] * act(in)[img][ci][y+ky-1][x+kx-1].
// A workgroup owns one spatial tile position, a range of images, 4 output channels and 4 input channels: every
// thread keeps the 4 x 4 x 9 partial sums of its own pixel quads in registers across the whole image range
// (dy quads straight from global, the activated input halo through LDS), and the 256 threads are folded once at
// the end (wave shuffles, then LDS).  One slab row per (tile, image range); uslab_reduce_kernel sums the rows.
// ------------------------------------------------------------------------------------------------
#define UD_WC 4          // output channels per workgroup
#define UD_WI 2          // input channels per workgroup (4 x 2 x 9 = 72 partial sums per thread)

struct UdWgradArgs {
  const float* dy; long dys;
  const float* in0; const float* sc0; const float* sh0; long is0; int C0, relu0;
  const float* in1; const float* sc1; const float* sh1; long is1; int C1, relu1;
  float* slab;             // [nz][Co*Ci*9]
  int Co, H, Wd, gsize;
  long n_img; int img_per_wg, n_cichunk;
  int bfd, bf0, bf1;       // storage of dy / in0 / in1 (0 fp32, 1 bf16); the VALU kernel below takes fp32 only
  const long* off0 = nullptr;   // optional per-image element offsets of in0 (as UdConvArgs::off0)
};

template <int TH, int TW>
__global__ __launch_bounds__(256) void ud_wgrad3x3_kernel(UdWgradArgs a) {
  constexpr int LDT = TW + 8;
  __shared__ __attribute__((aligned(16))) float tile[UD_WI][TH + 2][LDT];
  __shared__ float red[4][UD_WC * UD_WI * 9];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int GX = TW / 4;
  const int tx4 = (tid % GX) * 4, ty = tid / GX;
  const int tiles_x = (a.Wd + TW - 1) / TW;
  const int x0 = (blockIdx.x % tiles_x) * TW, y0 = (blockIdx.x / tiles_x) * TH;
  const long img0 = (long)blockIdx.y * a.img_per_wg;
  const long img1 = min(img0 + a.img_per_wg, a.n_img);
  const int ci0 = (blockIdx.z % a.n_cichunk) * UD_WI, co0 = (blockIdx.z / a.n_cichunk) * UD_WC;
  const int Ci = a.C0 + a.C1;
  // the conv arguments of ud_act
  UdConvArgs ca;
  ca.in0 = a.in0; ca.sc0 = a.sc0; ca.sh0 = a.sh0; ca.is0 = a.is0; ca.C0 = a.C0; ca.relu0 = a.relu0; ca.off0 = a.off0;
  ca.in1 = a.in1; ca.sc1 = a.sc1; ca.sh1 = a.sh1; ca.is1 = a.is1; ca.C1 = a.C1; ca.relu1 = a.relu1;
  ca.W = nullptr; ca.out = nullptr; ca.os = 0; ca.Co = a.Co; ca.H = a.H; ca.Wd = a.Wd; ca.gsize = a.gsize; ca.stats = nullptr;
  ca.bf0 = a.bf0; ca.bf1 = a.bf1; ca.bfo = 0;

  float acc[UD_WC][UD_WI][9];
#pragma unroll
  for (int co = 0; co < UD_WC; ++co)
#pragma unroll
    for (int c = 0; c < UD_WI; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[co][c][t] = 0.f;

  const int y = y0 + ty, x = x0 + tx4;
  const bool inside = (y < a.H) && (x < a.Wd);
  for (long img = img0; img < img1; ++img) {
    const long grp = img / a.gsize;
    __syncthreads();
    ud_stage_halo<TH, TW, LDT>(&tile[0][0][0], ca, img, grp, ci0, min(UD_WI, Ci - ci0), y0, x0, tid);
    float4 d[UD_WC];
#pragma unroll
    for (int co = 0; co < UD_WC; ++co)
      d[co] = (inside && co0 + co < a.Co)
                  ? *reinterpret_cast<const float4*>(&a.dy[img * a.dys + ((long)(co0 + co) * a.H + y) * a.Wd + x])
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < UD_WI; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float seg[6];
        const float* row = &tile[c][ty + ky][tx4 + 3];
#pragma unroll
        for (int q = 0; q < 6; ++q) seg[q] = row[q];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int co = 0; co < UD_WC; ++co)
            acc[co][c][ky * 3 + kx] += d[co].x * seg[kx] + d[co].y * seg[kx + 1] + d[co].z * seg[kx + 2] + d[co].w * seg[kx + 3];
      }
  }
  // fold the 256 threads: wave shuffles, then the four waves through LDS
#pragma unroll
  for (int co = 0; co < UD_WC; ++co)
#pragma unroll
    for (int c = 0; c < UD_WI; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v = acc[co][c][t];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][(co * UD_WI + c) * 9 + t] = v;
      }
  __syncthreads();
  if (tid < UD_WC * UD_WI * 9) {
    const int co = tid / (UD_WI * 9), r = tid - co * (UD_WI * 9), c = r / 9, t = r - c * 9;
    if (co0 + co < a.Co && ci0 + c < Ci) {
      const long z = (long)blockIdx.y * gridDim.x + blockIdx.x;
      a.slab[z * ((long)a.Co * Ci * 9) + ((long)(co0 + co) * Ci + ci0 + c) * 9 + t] =
          red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 convs on the fp32 matrix pipe (exact fp32: v_mfma_f32_16x16x4_f32 is bit for bit a
// k-ordered fmaf chain).  The reduction over pixels is what made the VALU form above expensive: every thread kept
// 72 partial sums and 4x2-channel blocks re-read dy and the halo per block.  Here the sum over pixels IS the MFMA's k:
//   D[m = (ci,tap)][j = co] += A[m][k] * B[k][j],   k = 4 neighbouring pixels of one row,
//   A[m][k] = act(in)[ci][y + ky - 1][x0 + k + kx - 1]   (a gather from the LDS halo tile: one ds_read_b32 per lane),
//   B[k][j] = dy[co][y][x0 + k]                          (LDS, planes padded so that the 16 channels hit 16 banks),
// so a wave accumulates all Co x 16 input channels x 9 taps in 4*MB*NB registers over every pixel it visits and the
// workgroup's partial result is folded once, at the end of its image range.  One workgroup = one 512-pixel tile
// position x a range of images x one chunk of 16 input channels; slab rows as for ud_wgrad3x3_kernel.
// ------------------------------------------------------------------------------------------------
typedef float uw_f32x4 __attribute__((ext_vector_type(4)));
// tiles of 512 pixels: 8 x 64, for narrower images 16 x 32 / 32 x 16 (template parameter TW)
#define UW_CIC 16                     // input channels per workgroup
template <int TW> struct UwGeo {
  static constexpr int TH = 512 / TW;
  static constexpr int LDT = TW + 8;                // halo row: col 3 = x0-1, cols 4.. = interior, col TW+4 = x0+TW
  static constexpr int PS = (TH + 2) * LDT + 8;     // halo plane stride (+8: planes spread over the banks)
  static constexpr int DPS = TH * TW + 4;           // dy plane stride (+4: channel j of the B fragment lands on bank 4j + k)
};

// dynamic LDS: max(cmax*UW_PS + Co*UW_DPS, 4*NB*MB*256) floats, cmax = min(Ci, UW_CIC) -- sized by the layer, so that
// the thin layers (20 KB) put 4..8 workgroups on a CU and hide their own staging latency
template <int TW>
static inline size_t uw_lds_bytes(int Ci, int Co, int NB, int MB) {
  const int cmax = Ci < UW_CIC ? Ci : UW_CIC;
  size_t a = (size_t)cmax * UwGeo<TW>::PS + (size_t)Co * UwGeo<TW>::DPS, b = (size_t)4 * NB * MB * 256;
  return (a > b ? a : b) * sizeof(float);
}
template <int NB, int MB, int TW = 64>     // Co <= 16*NB, chunk channels * 9 <= 16*MB, tile (512/TW) x TW
__global__ __launch_bounds__(256) void uw_wgrad_mfma_kernel(UdWgradArgs a) {
  constexpr int UW_TH_ = UwGeo<TW>::TH, UW_LDT = UwGeo<TW>::LDT, UW_PS = UwGeo<TW>::PS, UW_DPS = UwGeo<TW>::DPS;
  constexpr int QPR = TW / 4;                                        // pixel quads per tile row
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int cmax = min(a.C0 + a.C1, UW_CIC);
  float* xs = lds;
  float* dys = lds + cmax * UW_PS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = a.Wd / TW;
  const int x0 = (blockIdx.x % tiles_x) * TW, y0 = (blockIdx.x / tiles_x) * UW_TH_;
  const long img0 = (long)blockIdx.y * a.img_per_wg;
  const long img1 = min(img0 + a.img_per_wg, a.n_img);
  const int Ci = a.C0 + a.C1;
  const int ci0 = blockIdx.z * UW_CIC;
  const int cic = min(UW_CIC, Ci - ci0);
  UdConvArgs ca;
  ca.in0 = a.in0; ca.sc0 = a.sc0; ca.sh0 = a.sh0; ca.is0 = a.is0; ca.C0 = a.C0; ca.relu0 = a.relu0; ca.off0 = a.off0;
  ca.in1 = a.in1; ca.sc1 = a.sc1; ca.sh1 = a.sh1; ca.is1 = a.is1; ca.C1 = a.C1; ca.relu1 = a.relu1;
  ca.W = nullptr; ca.out = nullptr; ca.os = 0; ca.Co = a.Co; ca.H = a.H; ca.Wd = a.Wd; ca.gsize = a.gsize; ca.stats = nullptr;
  ca.bf0 = a.bf0; ca.bf1 = a.bf1; ca.bfo = 0;

  // per-lane fragment geometry
  int abase[MB]; float amask[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + (lane & 15);
    const bool ok = m < cic * 9;
    const int mm = ok ? m : 0;
    const int ci = mm / 9, tap = mm - 9 * ci, ky = tap / 3, kx = tap - 3 * ky;
    abase[mb] = ci * UW_PS + ky * UW_LDT + kx + 3 + (lane >> 4);
    amask[mb] = ok ? 1.f : 0.f;
  }
  int bbase[NB]; float bmask[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int co = nb * 16 + (lane & 15);
    bbase[nb] = (co < a.Co ? co : 0) * UW_DPS + (lane >> 4);
    bmask[nb] = co < a.Co ? 1.f : 0.f;
  }

  uw_f32x4 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (uw_f32x4){0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < cmax * UW_PS; i += 256) xs[i] = 0.f;            // (channels >= cic are masked, keep them finite)

  for (long img = img0; img < img1; ++img) {
    const long grp = img / a.gsize;
    __syncthreads();                                                   // previous tile fully consumed
    // activated halo tile: interior quads as aligned float4, the two halo columns as scalars
    for (int idx = tid; idx < cic * (UW_TH_ + 2) * QPR; idx += 256) {
      const int c = idx / ((UW_TH_ + 2) * QPR), r = idx - c * ((UW_TH_ + 2) * QPR);
      const int yy = r / QPR, q = r - yy * QPR;
      const int y = y0 + yy - 1, x = x0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)y < (unsigned)a.H) {
        const int cg = ci0 + c;
        const bool first = cg < a.C0;
        const int cc = first ? cg : cg - a.C0;
        const float* base = first ? a.in0 : a.in1;
        const long ib = first ? ud_base0(a, img) : img * a.is1;
        v = ua_ld4(base, ib + ((long)cc * a.H + y) * a.Wd + x, first ? a.bf0 : a.bf1);
        const float* sc = first ? a.sc0 : a.sc1;
        if (sc) {
          const float* sh = first ? a.sh0 : a.sh1;
          const long gi = grp * (first ? a.C0 : a.C1) + cc;
          const float s_ = sc[gi], t_ = sh[gi];
          v.x = v.x * s_ + t_; v.y = v.y * s_ + t_; v.z = v.z * s_ + t_; v.w = v.w * s_ + t_;
        }
        if (first ? a.relu0 : a.relu1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      *reinterpret_cast<float4*>(&xs[c * UW_PS + yy * UW_LDT + 4 + 4 * q]) = v;
    }
    for (int idx = tid; idx < cic * (UW_TH_ + 2) * 2; idx += 256) {
      const int c = idx / ((UW_TH_ + 2) * 2), r = idx - c * ((UW_TH_ + 2) * 2);
      const int yy = r >> 1, side = r & 1;
      xs[c * UW_PS + yy * UW_LDT + (side ? TW + 4 : 3)] =
          ud_act(ca, img, grp, ci0 + c, y0 + yy - 1, side ? x0 + TW : x0 - 1);
    }
    for (int idx = tid; idx < a.Co * UW_TH_ * QPR; idx += 256) {
      const int co = idx / (UW_TH_ * QPR), r = idx - co * (UW_TH_ * QPR);
      const int yy = r / QPR, q = r - yy * QPR;
      const float4 d = ua_ld4(a.dy, img * a.dys + ((long)co * a.H + y0 + yy) * a.Wd + x0 + 4 * q, a.bfd);
      *reinterpret_cast<float4*>(&dys[co * UW_DPS + yy * TW + 4 * q]) = d;
    }
    __syncthreads();
    // this wave's 32 pixel quads (two rows of the 8 x 64 tile, 4 / 8 rows of the narrower ones)
#pragma unroll 2
    for (int it = 0; it < 32; ++it) {
      const int pq = wave * 32 + it;
      const int yy = pq / QPR, q = pq % QPR;
      const int aoff = yy * UW_LDT + 4 * q, boff = yy * TW + 4 * q;
      float bv[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bv[nb] = dys[bbase[nb] + boff] * bmask[nb];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float av = xs[abase[mb] + aoff] * amask[mb];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nb], acc[nb][mb], 0, 0, 0);
      }
    }
  }
  // fold the four waves (fixed order) and write this workgroup's slab row
  __syncthreads();
  float* red = lds;                                                    // [4][NB*MB*256]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * NB * MB + nb * MB + mb) * 256 + r * 64 + lane] = acc[nb][mb][r];
  __syncthreads();
  const long z = (long)blockIdx.y * gridDim.x + blockIdx.x;
  float* row = a.slab + z * ((long)a.Co * Ci * 9);
  for (int e = tid; e < NB * MB * 256; e += 256) {
    const int blk = e >> 8, r = (e >> 6) & 3, l = e & 63;
    const int nb = blk / MB, mb = blk - nb * MB;
    const int co = nb * 16 + (l & 15), m = mb * 16 + (l >> 4) * 4 + r;
    if (co < a.Co && m < cic * 9) {
      const float s = (red[e] + red[NB * MB * 256 + e]) + (red[2 * NB * MB * 256 + e] + red[3 * NB * MB * 256 + e]);
      row[((long)co * Ci + ci0) * 9 + m] = s;          // (ci0 + m/9)*9 + m%9 == ci0*9 + m
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 conv of the DEEP levels (16..64 channels at 32x32 / 16x16 pixels) on the fp32 matrix pipe with an LDS halo tile.
// These layers are real contractions (K = Ci*9 = 144..576), but as an im2col GEMM on the tile engine they were bound by
// the per-element index arithmetic of the gather.  Here the gather is an LDS address:
//   D[co][pixel] += A[co][k] * B[k][pixel],  k = (ci, tap) in chunks of 4 (v_mfma_f32_16x16x4_f32, exact fp32),
//   A = weights staged k-major in LDS ([k][Co]: lanes read consecutive addresses), B = the activated halo tile
//   (B[k][pixel] = xs[ci][y + ky][x + kx]: one ds_read_b32 per lane).
// Workgroup = one TH x TW tile (256 pixels = 16 blocks of 16 pixels of a row; 4 blocks per wave) of one image x all
// output channels (MB blocks of 16); input channels staged 8 at a time.  Optional BatchNorm statistics per tile from
// the accumulators, as in ud_conv3x3_kernel.  The data gradient is this kernel on dy with the flipped weights.
// ------------------------------------------------------------------------------------------------
#define UX_CIC 8
template <int MB, int TW, bool FULL = false>      // FULL: the input channel count is a multiple of 8 (no partial last chunk:
                                                  // the k loop needs no masks; as a run-time branch the second loop copy
                                                  // cost the 32-pixel instantiations 15 %)
__global__ __launch_bounds__(256) void ux_conv3x3_mfma_kernel(UdConvArgs a) {
  constexpr int TH = 256 / TW;
  constexpr int LDT = TW + 8;
  constexpr int PS = (TH + 2) * LDT;
  constexpr int CP = 16 * MB;
  constexpr int BPR = TW / 16;                                   // 16-pixel blocks per tile row
  __shared__ __attribute__((aligned(16))) float xs[UX_CIC * PS];
  __shared__ __attribute__((aligned(16))) float wsm[UX_CIC * 9 * CP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // cosplit workgroups share a tile, each with 16 * MB of the output channels: 134 images of 16 x 16 pixels are 134
  // workgroups of one wave per SIMD on 256 compute units -- nothing overlapped the staging of a chunk (90 us for 35 us of
  // MFMA issue at 64 -> 64 channels).  Their grid is one-dimensional: workgroup L serves tile 8 (L / 8S) + L % 8, split
  // (L % 8S) / 8 -- the S workgroups of a tile are 8 apart, i.e. on the SAME XCD (round-robin dispatch), and the tile's halo
  // comes from memory once, not once per L2.
  int bx = blockIdx.x, by = blockIdx.y, co0 = 0;
  long img = blockIdx.z;
  const int gx = a.Wd / TW, gy = a.H / TH;
  if (a.cosplit > 1) {
    const int S8 = 8 * a.cosplit;
    const long c = blockIdx.x / S8;
    const int r = (int)(blockIdx.x - c * S8);
    const long tg = c * 8 + (r & 7);
    if (tg >= (long)gx * gy * a.n_img) return;
    co0 = (r >> 3) * CP;
    img = tg / (gx * gy);
    const int t = (int)(tg - img * (gx * gy));
    by = t / gx; bx = t - by * gx;
  }
  const int x0 = bx * TW, y0 = by * TH;
  const long grp = img / a.gsize;
  const int Ci = a.C0 + a.C1;
  const int lj = lane & 15, lk = lane >> 4;
  uw_f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (uw_f32x4){0.f, 0.f, 0.f, 0.f};
  int poff[4];                                                   // LDS offset of this lane's pixel in each of its blocks
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int blk = 4 * wave + nb;
    poff[nb] = (blk / BPR) * LDT + (blk % BPR) * 16 + lj + 3;
  }
  int kstep_b[UX_CIC * 9 / 4], kstep_w[UX_CIC * 9 / 4];
#pragma unroll
  for (int i = 0; i < UX_CIC * 9 / 4; ++i) {
    const int k = 4 * i + lk, c = k / 9, tap = k - 9 * c, ky = tap / 3, kx = tap - 3 * ky;
    kstep_b[i] = c * PS + ky * LDT + kx;
    kstep_w[i] = k * CP + lj;
  }
  for (int c0 = 0; c0 < Ci; c0 += UX_CIC) {
    const int nc = min(UX_CIC, Ci - c0);
    __syncthreads();
    ud_stage_halo<TH, TW, LDT>(xs, a, img, grp, c0, nc, y0, x0, tid);
    for (int idx = tid; idx < nc * 9 * CP; idx += 256) {
      const int k = idx / CP, co = idx - k * CP;
      const int c = k / 9, tap = k - 9 * c;
      wsm[idx] = (co0 + co < a.Co) ? a.W[((long)(co0 + co) * Ci + c0 + c) * 9 + tap] : 0.f;
    }
    __syncthreads();
    const int nk = nc * 9;
    // the 18 k-steps of a chunk: the tile / weight offsets of this lane's k = 4 i + lk do not depend on the chunk -- formed once
    // per kernel (kstep_b / kstep_w), not by two integer divisions per step (the counters showed 7.3 VALU instructions per
    // MFMA in this loop)
    if constexpr (MB == 4 && TW == 32) {                          // (64 output channels at 32 x 32: the unrolled form measured
                                                                 //  102 us against 90 us for the rolled loop)
      for (int ks = 0; ks < nk; ks += 4) {
        const int k = ks + lk;
        const bool ok = k < nk;
        const int kk = ok ? k : 0;
        const int c = kk / 9, tap = kk - 9 * c, ky = tap / 3, kx = tap - 3 * ky;
        const int boff = c * PS + ky * LDT + kx;
        const float m = ok ? 1.f : 0.f;
        float af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[mb] = wsm[kk * CP + mb * 16 + lj] * m;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float bv = xs[boff + poff[nb]];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb], bv, acc[mb][nb], 0, 0, 0);
        }
      }
    } else if constexpr (FULL) {                                 // full chunks only: every lane's k is inside, nothing to mask
#pragma unroll
      for (int i = 0; i < UX_CIC * 9 / 4; ++i) {
        float af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[mb] = wsm[kstep_w[i] + mb * 16];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float bv = xs[kstep_b[i] + poff[nb]];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb], bv, acc[mb][nb], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < UX_CIC * 9 / 4; ++i) {
        if (4 * i >= nk) break;                                  // (a last chunk of fewer than 8 channels)
        // lanes past the chunk's last k (a last chunk of fewer than 8 channels) multiply by zero -- but they must read
        // FINITE values: the LDS beyond this chunk's data holds whatever an earlier kernel left there (a NaN pattern made
        // 0 * NaN = NaN in a 1-channel 16 x 16 case, only after other tests had run), so they re-read step 0's operands
        const bool ok = 4 * i + lk < nk;
        const float m = ok ? 1.f : 0.f;
        const int wo = ok ? kstep_w[i] : kstep_w[0], bo = ok ? kstep_b[i] : kstep_b[0];
        float af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[mb] = wsm[wo + mb * 16] * m;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float bv = xs[bo + poff[nb]];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb], bv, acc[mb][nb], 0, 0, 0);
        }
      }
    }
  }
  // D[i = co][j = pixel]: lane holds column j = lane & 15, rows 4 * (lane >> 4) + r
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + mb * 16 + lk * 4 + r;
      if (co < a.Co) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int blk = 4 * wave + nb;
          const int y = y0 + blk / BPR, x = x0 + (blk % BPR) * 16 + lj;
          const long o = img * a.os + ((long)co * a.H + y) * a.Wd + x;
          if (a.bfo) { __bf16 t = (__bf16)acc[mb][nb][r]; reinterpret_cast<unsigned short*>(a.out)[o] = __builtin_bit_cast(unsigned short, t); }
          else a.out[o] = acc[mb][nb][r];
        }
      }
    }
  if (a.stats) {
    __shared__ float red[4][2 * CP];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { const float v = acc[mb][nb][r]; s1 += v; s2 += v * v; }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }     // the 16 pixel lanes
        if (lj == 0) { const int co = mb * 16 + lk * 4 + r; red[wave][2 * co] = s1; red[wave][2 * co + 1] = s2; }
      }
    __syncthreads();
    if (tid < 2 * CP && co0 + (tid >> 1) < a.Co) {
      const long tile = (long)by * gx + bx, ntile = (long)gx * gy;
      a.stats[((img * ntile + tile) * a.Co + co0 + (tid >> 1)) * 2 + (tid & 1)] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
  }
}
