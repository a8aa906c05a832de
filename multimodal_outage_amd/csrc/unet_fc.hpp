// Few-row Linear layers against a large fp32 weight matrix (Encoder.fc1 of config 3: 134 rows x 16384 -> 4096, a 268 MB
// matrix) in the bf16 mode: "3 x bf16" products on the bf16 matrix pipe.
//   w = w_hi + w_lo + O(2^-17 |w|),  w_hi = bf16(w), w_lo = bf16(w - w_hi)     (likewise x)
//   x * w ~= x_hi * w_hi + x_hi * w_lo + x_lo * w_hi                            (the dropped x_lo * w_lo is 2^-18 relative)
// three v_mfma_f32_16x16x32_bf16 per tile step with fp32 accumulation: ~1.5e-5 relative per product instead of the 2^-9 of
// a plain bf16 product, at 1/5 of the exact-fp32 MFMA's time -- the exact kernels were compute bound (18 GFLOP per pass at
// 30 % of the 157 TFLOP/s fp32 peak: 380 us against the 54 us the 268 MB take from HBM).  The weight matrix is read ONCE as
// fp32 and split in registers; the few-row operand is split beforehand into two bf16 matrices (ufc_split_kernel) and staged
// through LDS, shared by the four waves of a workgroup.
//   WMODE 0 (forward):        out[p][n] = sum_k x[p][k]    * W[n][k]      a wave owns 16 rows n of W, streams k
//   WMODE 1 (data gradient):  din[p][c] = sum_n dout[p][n] * W[n][c]      a wave owns 16 columns c of W, streams rows n
// Split-K over gridDim.y: slab[ks][p][col] partial sums, folded (+ bias, ReLU) by ufc_reduce_kernel in a fixed order.
#pragma once
#include "unet_bf16.hpp"

#define UFC_KC 128                       // reduction elements per staged chunk (4 MFMA steps of 32)
#define UFC_LD (UFC_KC + 8)              // LDS row stride of the few-row operand (bf16 elements): 272 B, conflict-free b128 reads
#define UFC_MB 9                         // 16-row blocks of the few-row operand: P <= 144 (134 = 67 counties x 2 days)

struct UfcArgs {
  const unsigned short* ah;              // few-row operand, hi / lo halves: bf16 [P][R] (R = reduction length)
  const unsigned short* al;
  const float* W;                        // WMODE 0: [C][R] (row = output column);  WMODE 1: [R][C]
  float* slab;                           // [gridDim.y][P][C]
  int P, R, C;                           // rows, reduction length, output columns
  int chunks_per_split;                  // chunks of UFC_KC per workgroup
  int groups_in_grid;                    // 1: row group = blockIdx.z;  0: every workgroup walks all row groups
};

__global__ void ufc_split_kernel(const float* __restrict__ x, long n, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 v = *reinterpret_cast<const float4*>(x + i);
  const unsigned h0 = ub_pack2(v.x, v.y), h1 = ub_pack2(v.z, v.w);
  const unsigned l0 = ub_pack2(v.x - ua_lo(h0), v.y - ua_hi(h0)), l1 = ub_pack2(v.z - ua_lo(h1), v.w - ua_hi(h1));
  *reinterpret_cast<uint2*>(hi + i) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(lo + i) = make_uint2(l0, l1);
}

__global__ void ufc_reduce_kernel(const float* __restrict__ slab, long stride, int nz, const float* __restrict__ bias, int C,
                                  int relu, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = bias ? bias[i % C] : 0.f;
  for (int z = 0; z < nz; ++z) s += slab[(long)z * stride + i];
  out[i] = relu ? fmaxf(s, 0.f) : s;
}

// 8 fp32 -> hi / lo bf16 fragments
__device__ __forceinline__ void ufc_split8(const float (&w)[8], ub_bf8& hi, ub_bf8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = ub_pack2(w[2 * i], w[2 * i + 1]);
    l[i] = ub_pack2(w[2 * i] - ua_lo(h[i]), w[2 * i + 1] - ua_hi(h[i]));
  }
  hi = __builtin_bit_cast(ub_bf8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(ub_bf8, make_uint4(l[0], l[1], l[2], l[3]));
}

// NBW: 16-column blocks of W per wave (a workgroup owns 64 NBW columns), KC: reduction elements per staged chunk.
// <1, 128> was the first version: per (column, k) of W it stages 144 / 64 = 2.25 values of the few-row operand through
// L2 -> LDS and reads every staged fragment from LDS once per 3 MFMAs -- the kernel scaled with the ROW count, not with
// the weight bytes (134 rows 117 us, 536 rows 428 us on the 268 MB matrix).  <2, 64>: twice the columns per staged chunk
// (half the L2 -> LDS traffic and half the LDS reads per MFMA) at the same registers (the chunk is half as long).
template <int WMODE, int MB, int NBW, int KC>             // MB: 16-row blocks of the few-row operand (P <= 16 * MB)
__global__ __launch_bounds__(256, 2) void ufc_kernel(UfcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];      // [2][MB*16][UFC_LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, lg = lane >> 4;
  constexpr int ROWS = MB * 16;
  constexpr int NS = KC / 32;                                              // MFMA steps per chunk
  constexpr int NI = (ROWS * (KC / 8) + 255) / 256;                        // 16-byte pieces of a chunk per thread and half
  unsigned short* xh = xs;
  unsigned short* xl = xs + ROWS * UFC_LD;
  const int c0 = (blockIdx.x * 4 + wave) * 16 * NBW;                       // this wave's 16 NBW output columns
  const long r_begin = (long)blockIdx.y * a.chunks_per_split * KC;
  const int nchunk = (int)min((long)a.chunks_per_split, (a.R - r_begin + KC - 1) / KC);
  const __amdgpu_buffer_rsrc_t rh = ub_rsrc(a.ah, (long)a.P * a.R * 2), rl = ub_rsrc(a.al, (long)a.P * a.R * 2);
  const __amdgpu_buffer_rsrc_t rw = ub_rsrc(a.W, (long)a.R * a.C * 4);

  // More rows than 16 * MB (several windows per step): row groups of 16 * MB, one after the other INSIDE the workgroup -- its
  // weight panel (its columns x its share of the reduction, ~256 KB) is then re-read from L2 / the Infinity Cache instead of
  // the whole 268 MB matrix being streamed from HBM once per group by separate launches (groups_in_grid: A/B switch).
  const int ngroups = (a.P + ROWS - 1) / ROWS;
  const int rg_first = a.groups_in_grid ? (int)blockIdx.z : 0, rg_last = a.groups_in_grid ? (int)blockIdx.z + 1 : ngroups;
  for (int rg = rg_first; rg < rg_last; ++rg) {
  const int prow = rg * ROWS;
  ub_f4 acc[NBW][MB];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (ub_f4){0.f, 0.f, 0.f, 0.f};

  // software pipeline as in unet_bf16.hpp: chunk ch+1's loads (weight fragments: NS steps x 8 fp32 per lane and column
  // block; the few-row operand's 16-byte pieces) are issued in front of chunk ch's matrix work and wait in registers
  float wreg[NBW][NS][8];
  ub_u4 xrh[NI], xrl[NI];
  int xsrc[NI], xdst[NI];                                                  // element offset in the operand (chunk 0) / in LDS
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int i = tid + 256 * it;
    const int row = i / (KC / 8), seg = i - row * (KC / 8);
    const bool ok = i < ROWS * (KC / 8);
    xdst[it] = ok ? row * UFC_LD + 8 * seg : -1;
    xsrc[it] = (ok && row + prow < a.P) ? (row + prow) * a.R + 8 * seg : -1;
  }
  auto load_chunk = [&](const int ch) {
    const long r0 = r_begin + (long)ch * KC;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int col = c0 + 16 * nb + lp;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const long r = r0 + 32 * s + 8 * lg;                               // first of this lane's 8 reduction indices
        if (WMODE == 0) {                                                  // W[col][r .. r+7]: 32 contiguous bytes
          const bool ok = col < a.C && r < a.R;                            // (R % 8 == 0)
          const unsigned off = ok ? (unsigned)(((long)col * a.R + r) * 4) : UB_OOB;
          const ub_u4 u0 = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)off, 0, 0);
          const ub_u4 u1 = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(off + 16), 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) { wreg[nb][s][j] = __uint_as_float(u0[j]); wreg[nb][s][4 + j] = __uint_as_float(u1[j]); }
        } else {                                                           // W[r + j][col]: 8 rows of the matrix
          // lane part of the address (row 8 lg, its column) in ONE register per column block, the row of the step and j
          // as the wave-uniform scalar offset: per-load 64-bit address arithmetic made this variant spill (340 B / lane)
          const unsigned voff = (col < a.C && r < a.R) ? (unsigned)((8 * lg * a.C + col) * 4) : UB_OOB;   // (R % 8 == 0)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            wreg[nb][s][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rw, (int)voff, (int)((r0 + 32 * s + j) * a.C * 4), 0));
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {                                      // rows >= P and reduction indices >= R read zero
      const int seg8 = (xdst[it] >= 0) ? (xdst[it] % UFC_LD) : 0;
      const unsigned off = (xsrc[it] >= 0 && r0 + seg8 < a.R) ? (unsigned)(((long)xsrc[it] + r0) * 2) : UB_OOB;
      xrh[it] = __builtin_amdgcn_raw_buffer_load_b128(rh, (int)off, 0, 0);
      xrl[it] = __builtin_amdgcn_raw_buffer_load_b128(rl, (int)off, 0, 0);
    }
  };
  if (nchunk > 0) load_chunk(0);
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();                                                       // previous chunk's reads of xs are done
#pragma unroll
    for (int it = 0; it < NI; ++it)
      if (xdst[it] >= 0) {
        *reinterpret_cast<ub_u4*>(&xh[xdst[it]]) = xrh[it];
        *reinterpret_cast<ub_u4*>(&xl[xdst[it]]) = xrl[it];
      }
    ub_bf8 wh[NBW][NS], wl[NBW][NS];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int s = 0; s < NS; ++s) ufc_split8(wreg[nb][s], wh[nb][s], wl[nb][s]);
    __syncthreads();
    if (ch + 1 < nchunk) load_chunk(ch + 1);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int o = (mb * 16 + lp) * UFC_LD + 32 * s + 8 * lg;
        const ub_bf8 ah = __builtin_bit_cast(ub_bf8, *reinterpret_cast<const ub_u4*>(&xh[o]));
        const ub_bf8 al = __builtin_bit_cast(ub_bf8, *reinterpret_cast<const ub_u4*>(&xl[o]));
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[nb][s], acc[nb][mb], 0, 0, 0);
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[nb][s], acc[nb][mb], 0, 0, 0);
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[nb][s], acc[nb][mb], 0, 0, 0);
        }
      }
    }
  }
  // D[row = 4*lg + r][col = lp] of every block -> slab[ks][p][c]
  float* out = a.slab + (long)blockIdx.y * a.P * a.C;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int col = c0 + 16 * nb + lp;
    if (col < a.C) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = prow + mb * 16 + 4 * lg + r;
          if (p < a.P) out[(long)p * a.C + col] = acc[nb][mb][r];
        }
    }
  }
  __syncthreads();                                                         // (the next group restages the LDS operand)
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same layers: dW[n][c] = sum_p dout[p][n] * x[p][c]  (k = p: 134 rows, padded to 160), output
// bound on the 268 MB of dW.  Both operands are needed p-contiguous: ufc_tsplit_kernel writes the hi / lo halves of a
// [P][Q] fp32 matrix transposed, [Q][UFC_PP] bf16 (and the column sums: the bias gradient).  A wave keeps the fragments
// of its 16 columns c (x^T, 5 steps x hi/lo) in registers and sweeps the rows n; the dout^T blocks of 64 rows are staged
// in LDS for the four waves of a workgroup.
// ------------------------------------------------------------------------------------------------
#define UFC_PP 160                       // padded row count of the transposed halves (5 MFMA steps of 32)
#define UFC_NS 5
#define UFC_TLD (UFC_PP + 8)             // LDS row stride of a staged dout^T row (bf16)

// out_hi/out_lo[q][p] = split(in[p][q]); colsum[q] = sum_p in[p][q] (may be NULL).  One thread per column q.
__global__ void ufc_tsplit_kernel(const float* __restrict__ in, int P, int Q, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo, float* __restrict__ colsum) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  float s = 0.f;
  for (int p0 = 0; p0 < UFC_PP; p0 += 8) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = p0 + 2 * i;
      const float a = (p < P) ? in[(long)p * Q + q] : 0.f, b = (p + 1 < P) ? in[(long)(p + 1) * Q + q] : 0.f;
      s += a + b;
      h[i] = ub_pack2(a, b);
      l[i] = ub_pack2(a - ua_lo(h[i]), b - ua_hi(h[i]));
    }
    *reinterpret_cast<uint4*>(hi + (long)q * UFC_PP + p0) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(lo + (long)q * UFC_PP + p0) = make_uint4(l[0], l[1], l[2], l[3]);
  }
  if (colsum) colsum[q] = s;
}

struct UfcWArgs {
  const unsigned short* dh; const unsigned short* dl;     // dout^T halves [N][UFC_PP]
  const unsigned short* xh; const unsigned short* xl;     // x^T halves    [C][UFC_PP]
  float* dW;                                              // [N][C]
  int N, C;
  int nrows_per_wg;                                       // rows n swept by one workgroup (multiple of 64)
};

__global__ __launch_bounds__(256, 2) void ufc_wgrad_kernel(UfcWArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short ds[2][64 * UFC_TLD];      // hi / lo halves of 64 rows of dout^T
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, lg = lane >> 4;
  const int c0 = (blockIdx.x * 4 + wave) * 16;
  const int n_begin = blockIdx.y * a.nrows_per_wg;
  const int n_end = min(n_begin + a.nrows_per_wg, a.N);
  // B fragments: this lane's column c0 + lp, pixels... rows p = 32 s + 8 lg .. +7 of x^T
  ub_bf8 bh[UFC_NS], bl[UFC_NS];
  {
    const bool ok = c0 + lp < a.C;
    const long o = (long)min(c0 + lp, a.C - 1) * UFC_PP + 8 * lg;
#pragma unroll
    for (int s = 0; s < UFC_NS; ++s) {
      uint4 vh = *reinterpret_cast<const uint4*>(a.xh + o + 32 * s), vl = *reinterpret_cast<const uint4*>(a.xl + o + 32 * s);
      if (!ok) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
      bh[s] = __builtin_bit_cast(ub_bf8, vh); bl[s] = __builtin_bit_cast(ub_bf8, vl);
    }
  }
  // staging of a 64-row group of dout^T: 64 rows x 160 bf16 = 20 sixteen-byte pieces per row and half -> 5 per thread
  constexpr int NI = 64 * (UFC_PP / 8) / 256;
  uint4 rh[NI], rl[NI];
  auto load_group = [&](const int n0) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + 256 * it, row = i / (UFC_PP / 8), seg = i - row * (UFC_PP / 8);
      const bool ok = n0 + row < a.N;
      const long o = (long)min(n0 + row, a.N - 1) * UFC_PP + 8 * seg;
      rh[it] = *reinterpret_cast<const uint4*>(a.dh + o); rl[it] = *reinterpret_cast<const uint4*>(a.dl + o);
      if (!ok) { rh[it] = make_uint4(0u, 0u, 0u, 0u); rl[it] = rh[it]; }
    }
  };
  if (n_begin < n_end) load_group(n_begin);
#pragma unroll 1
  for (int n0 = n_begin; n0 < n_end; n0 += 64) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + 256 * it, row = i / (UFC_PP / 8), seg = i - row * (UFC_PP / 8);
      *reinterpret_cast<uint4*>(&ds[0][row * UFC_TLD + 8 * seg]) = rh[it];
      *reinterpret_cast<uint4*>(&ds[1][row * UFC_TLD + 8 * seg]) = rl[it];
    }
    __syncthreads();
    if (n0 + 64 < n_end) load_group(n0 + 64);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      ub_f4 acc = (ub_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < UFC_NS; ++s) {
        const int o = (nb * 16 + lp) * UFC_TLD + 32 * s + 8 * lg;
        const ub_bf8 ah = __builtin_bit_cast(ub_bf8, *reinterpret_cast<const uint4*>(&ds[0][o]));
        const ub_bf8 al = __builtin_bit_cast(ub_bf8, *reinterpret_cast<const uint4*>(&ds[1][o]));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[s], acc, 0, 0, 0);
      }
      // D[row n = 4*lg + r][col c = lp]
      if (c0 + lp < a.C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + nb * 16 + 4 * lg + r;
          if (n < a.N) a.dW[(long)n * a.C + c0 + lp] = acc[r];
        }
      }
    }
  }
}
