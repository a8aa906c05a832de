// bf16-operand / fp32-accumulate GEMM for the dense adaptive-adjacency node-axis products on gfx950.
//
//   D[M][N] (+)= sum_k A[m][k] * B(k, n)
//   A: bf16 [M][K], k contiguous (adp^T for the forward product, adp for its data gradient, X for dA)
//   B: bf16, either KROWS [K][N] (n contiguous: the nbtc activation matrix X[v][j] as it lies in HBM,
//      fragments fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16) or XROWS [N][K]
//   D: fp32 row-major.
// 128x128x32 block tile, 4 waves (2x2) of 64x64, v_mfma_f32_32x32x16_bf16, register-prefetched double
// buffered LDS.  LDS rows are padded so that the ds_read_b128 fragment reads (80-byte rows) and the
// transposing reads (320-byte rows) are bank-conflict free (MI355X_MICROARCH.md, LDS section).
#include <cstdlib>
#include <type_traits>
#include "mo_common.h"
#include "../../include/mo_hip.h"

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_ptr;

#define GB_BM 128
#define GB_BN 128
#define GB_BK 32
#define GB_LDA 40    // bf16 elements per LDS row of an [x][k] tile (64 B data + 16 B pad)
#define GB_LDB 160   // bf16 elements per LDS row of a  [k][n] tile (256 B data + 64 B pad)

__device__ __forceinline__ uint4 gb_load16(const short* base, long row, int ld, int col, int rows, int cols) {
  // 8 consecutive bf16 at (row, col..col+7); zero outside [rows) x [cols) (cols % 8 == 0 is required)
  if (row < rows && col < cols) return *reinterpret_cast<const uint4*>(base + row * (long)ld + col);
  return make_uint4(0u, 0u, 0u, 0u);
}

template <bool B_KROWS>
__global__ void __launch_bounds__(256)
gemm_bf16_kernel(const short* __restrict__ A, int lda, const short* __restrict__ B, int ldb, float* __restrict__ D,
                 int ldd, int M, int N, int K, int beta, unsigned short* __restrict__ Dbf,
                 const float* __restrict__ bias, int relu, const float* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) short As[2][GB_BM * GB_LDA];
  __shared__ __attribute__((aligned(16))) short Bs[2][B_KROWS ? GB_BK * GB_LDB : GB_BN * GB_LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  // XCD-aware tile placement (speed only, any placement is correct): workgroups are dealt round-robin
  // over the 8 XCDs, each with a private 4 MiB L2.  XCD x = b % 8 owns the (x % 2, x / 2) patch of a
  // 2 x 4 split of the tile grid and walks it m-fastest, so that the ~128 co-resident workgroups of an
  // XCD share A row-panels and B column-panels through its L2 instead of every XCD streaming all of B.
  const int gm = (M + GB_BM - 1) / GB_BM, gn = (N + GB_BN - 1) / GB_BN;
  const int pm = (gm + 1) / 2, pn = (gn + 3) / 4;
  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int lm = idx % pm, ln = idx / pm;
  const int tm = (xcd & 1) * pm + lm, tn = (xcd >> 1) * pn + ln;
  if (tm >= gm || tn >= gn || ln >= pn) return;     // padding of the patch grid (whole workgroup exits)
  const int m0 = tm * GB_BM, n0 = tn * GB_BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // tile loads: 512 16-byte chunks per operand, 2 per thread
  uint4 ra[2], rb[2];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      {  // A: [128 rows][4 chunks of 8 k]
        const int x = f >> 2, c = f & 3;
        ra[i] = gb_load16(A, m0 + x, lda, k0 + 8 * c, M, K);
      }
      if (B_KROWS) {  // B: [32 k rows][16 chunks of 8 n]
        const int k = f >> 4, c = f & 15;
        rb[i] = gb_load16(B, k0 + k, ldb, n0 + 8 * c, K, N);
      } else {
        const int x = f >> 2, c = f & 3;
        rb[i] = gb_load16(B, n0 + x, ldb, k0 + 8 * c, N, K);
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      {
        const int x = f >> 2, c = f & 3;
        *reinterpret_cast<uint4*>(&As[buf][x * GB_LDA + 8 * c]) = ra[i];
      }
      if (B_KROWS) {
        const int k = f >> 4, c = f & 15;
        *reinterpret_cast<uint4*>(&Bs[buf][k * GB_LDB + 8 * c]) = rb[i];
      } else {
        const int x = f >> 2, c = f & 3;
        *reinterpret_cast<uint4*>(&Bs[buf][x * GB_LDA + 8 * c]) = rb[i];
      }
    }
  };

  const int nk = (K + GB_BK - 1) / GB_BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int fr = lane & 31, fh = lane >> 5;
  // transposing-read geometry (16-lane groups): group g -> k half (g>>1), column half (g&1)
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * GB_BK);
    const short* Ac = As[cur];
    const short* Bc = Bs[cur];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v8s a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const v8s*>(&Ac[(wm0 + i * 32 + fr) * GB_LDA + 16 * s + 8 * fh]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (B_KROWS) {
          const int kr = 16 * s + 8 * (tg >> 1) + tq;
          const int nc = wn0 + j * 32 + 16 * (tg & 1) + 4 * tp;
          v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)&Bc[kr * GB_LDB + nc]);
          v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)&Bc[(kr + 4) * GB_LDB + nc]);
          b[j] = (v8s){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
          b[j] = *reinterpret_cast<const v8s*>(&Bc[(wn0 + j * 32 + fr) * GB_LDA + 16 * s + 8 * fh]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a[i]),
                                                               __builtin_bit_cast(v8bf, b[j]), acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn0 + j * 32 + fr;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (bias) v += bias[n];
        if (relu) v = fmaxf(v, 0.f);
        if (mask && !(mask[(long)m * ldd + n] > 0.f)) v = 0.f;       // ReLU backward: gated by the forward value
        if (D) {
          float* o = D + (long)m * ldd + n;
          if (beta) v += *o;
          *o = v;
        } else if (beta) {      // bf16-only output: accumulate onto the stored bf16 value
          v += __uint_as_float(((unsigned)Dbf[(long)m * ldd + n]) << 16);
        }
        if (Dbf) { __bf16 t = (__bf16)v; Dbf[(long)m * ldd + n] = __builtin_bit_cast(unsigned short, t); }
      }
    }
}

// fp32-result epilogue options as mo_gemm_bf16_256_ex (+ bias[n], ReLU, ReLU-backward gate).  Four workgroups per CU
// overlap one tile's stores with the others' k loops: for the head's output-bound products (K = 256 / 512 against
// 1.6 GB of fp32 result) this kernel beats the 256x256 ring (509 vs 835 us on the skip data gradient, tools/gemm_head.py).
extern "C" int mo_gemm_bf16_ex(const void* A, int lda, const void* B, int ldb, int b_krows, float* D, int ldd, int M,
                               int N, int K, int beta, void* D_bf16, const float* bias, int relu, const float* mask,
                               void* stream) {
  MO_CHECK_ARG(A && B && (D || D_bf16) && M > 0 && N > 0 && K > 0);
  MO_CHECK_ARG(D || (!bias && !relu && !mask));
  // 16-byte chunk loads: leading dimensions and the contiguous extents must be multiples of 8 elements
  MO_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (K % 8) == 0 && (!b_krows || (N % 8) == 0));
  MO_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0);
  const int gm = mo_cdiv(M, GB_BM), gn = mo_cdiv(N, GB_BN);
  dim3 grid(8 * ((gm + 1) / 2) * ((gn + 3) / 4));
  if (b_krows)
    hipLaunchKernelGGL(gemm_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const short*)A, lda,
                       (const short*)B, ldb, D, ldd, M, N, K, beta, (unsigned short*)D_bf16, bias, relu, mask);
  else
    hipLaunchKernelGGL(gemm_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const short*)A, lda,
                       (const short*)B, ldb, D, ldd, M, N, K, beta, (unsigned short*)D_bf16, bias, relu, mask);
  return mo_launch_status();
}
extern "C" int mo_gemm_bf16(const void* A, int lda, const void* B, int ldb, int b_krows, float* D, int ldd, int M,
                            int N, int K, int beta, void* D_bf16, void* stream) {
  return mo_gemm_bf16_ex(A, lda, B, ldb, b_krows, D, ldd, M, N, K, beta, D_bf16, nullptr, 0, nullptr, stream);
}

// fp32 -> bf16 (round to nearest even; plain cast so that NaNs stay NaNs), 8 elements per thread
__global__ void f32_to_bf16_kernel(const float4* __restrict__ x, uint4* __restrict__ y, long n8) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = x[2 * i], b = x[2 * i + 1];
  const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned short h[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { __bf16 t = (__bf16)f[q]; h[q] = __builtin_bit_cast(unsigned short, t); }
  uint4 o;
  o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
  o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
  y[i] = o;
}
extern "C" int mo_f32_to_bf16(const float* x, void* y, long n, void* stream) {
  MO_CHECK_ARG(x && y && n > 0 && (n % 8) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0);
  long n8 = n / 8;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(mo_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (uint4*)y, n8);
  return mo_launch_status();
}

// ================================================================================================
// 256x256x32 tile, 8 waves (2x4 of 128x64), 4-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4)
// behind counted vmcnt waits and one raw s_barrier per k-tile; the k loop is software-pipelined by half k-tiles
// (LDS reads of the next 16 k in flight under the 8 MFMAs of the current 16 k, hand-counted lgkmcnt).
// What bounds it (probes of round 1, J = 24576, bf16 result: DMA only 284 us, MFMA + LDS reads only 295 us, both
// 419 us): the L2 -> LDS fill of 32 KB per k-tile and CU (about 53 GB/s per CU alone, less at the clock the MFMAs
// leave) runs at the same pace as the 1024 MFMA cycles per k-tile -- and the chip is power-limited under this kernel:
// the shader clock drops from 2402 MHz (idle) to 1539 MHz beside it with random operands, 2174 MHz with an all-zero B
// (tools/clock_probe), so matrix pipe, LDS reads and fill share one power budget and the ceiling is the 1.6 PF of the
// sustained clock, not the 2.5 PF of the spec sheet.
//   throughput model of the 128^2 kernel above: (bytes in flight per CU) x (flop per byte of the tile)
//   / (L2/MALL latency) -- 64 KB x 64 flop/B / ~2.4 us x 256 CUs ~ 0.45 PF, which is what it measures.
//   This kernel has 96 KB in flight at 128 flop/B.
// LDS images are unpadded and XOR-swizzled through the per-lane SOURCE address (the DMA destination is
// wave-uniform base + lane*16):  [x][64 B] tiles: slot ^= (x>>2)&3 (conflict-free ds_read_b128);
// [k][512 B] tiles: slot16 ^= (k&3)<<2 (conflict-free ds_read_b64_tr_b16).
// Requirements (checked on the host): lda,ldb % 8 == 0; A readable and zero in columns [K, Kpad32) when
// K % 32 != 0 (B rows are clamped, so 0 x finite = 0); N % 8 == 0 for KROWS B.
// ================================================================================================
typedef __attribute__((address_space(3))) void* gb_lds_ptr;
typedef __attribute__((address_space(1))) const void* gb_g_ptr;

// LDS-DMA issued from inline asm so that hipcc neither counts it in its own vmcnt bookkeeping nor drains
// it (vmcnt(0)) in front of every LDS read of the ring; completion is tracked by the hand-counted
// s_waitcnt vmcnt(N) of the k loop.  M0 (LDS destination base) is saved and restored in the same statement.
__device__ __forceinline__ void gb_dma16(const void* gsrc, uint32_t lds_byte_off) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_byte_off)
               : "memory");
}

// LDS fragment reads from inline asm (immediate offsets), for the software-pipelined k loop: hipcc's own lgkmcnt
// bookkeeping drains every outstanding LDS read at the loop head (s_waitcnt lgkmcnt(0)), which would serialise the
// reads of the next half k-tile with the MFMAs of the current one; these are invisible to it and are awaited by
// hand-counted s_waitcnt lgkmcnt(N) (LDS returns in order; the loop holds no scalar loads).
template <int OFF>
__device__ __forceinline__ v8s g2_lds128(uint32_t addr) {
  v8s r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ v4s g2_ldstr(uint32_t addr) {
  v4s r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

#define G2_BM 256
#define G2_BN 256
#define G2_BK 32
#define G2_ST 4
#define G2_STAGE_SHORTS 16384   // 16 KB A + 16 KB B per stage, in bf16 elements

// MF16: the same tile on v_mfma_f32_16x16x32_bf16 (8 x 4 accumulator blocks of 16 x 16 per wave instead of 4 x 2 of
// 32 x 32; an MFMA takes a whole 32-wide k-tile, so the software pipeline splits a k-tile by ROWS: blocks 0-3, then
// 4-7, the B fragments double-buffered across k-tiles).  Same LDS bytes and MFMA cycles per k-tile; under the power
// limit the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS item 7).  The fragment reads differ, so
// the source swizzles do too: [x][64 B]: slot ^= {0,2,3,1}[(x>>2)&3]; [k][512 B]: slot16 ^= (k&3)<<2 | ((k>>3)&1)<<1.
template <bool MF16> __device__ __forceinline__ int g2_swz_x(int x) {
  const int rb = (x >> 2) & 3;
  return MF16 ? ((0x78 >> (2 * rb)) & 3) : rb;
}
template <bool MF16> __device__ __forceinline__ int g2_swz_k(int k) {
  return ((k & 3) << 2) | (MF16 ? (((k >> 3) & 1) << 1) : 0);
}
template <bool B_KROWS, bool MF16>
__global__ void __launch_bounds__(512)
gemm_bf16_256_kernel(const short* __restrict__ A, int lda, const short* __restrict__ B, int ldb, float* __restrict__ D,
                     int ldd, int M, int N, int K, int beta, unsigned short* __restrict__ Dbf,
                     const float* __restrict__ bias, int relu, const float* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) short lds[G2_ST * G2_STAGE_SHORTS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

  const int gm = (M + G2_BM - 1) / G2_BM, gn = (N + G2_BN - 1) / G2_BN;
  const int pm = (gm + 1) / 2, pn = (gn + 3) / 4;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int lm = idx % pm, ln = idx / pm;
  const int tmi = (xcd & 1) * pm + lm, tni = (xcd >> 1) * pn + ln;
  if (tmi >= gm || tni >= gn || ln >= pn) return;
  const int m0 = tmi * G2_BM, n0 = tni * G2_BN;

  // per-lane DMA source offsets (constant over k): two A and two B instructions per wave per stage
  long a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 128 + i * 64 + lane;            // 16-byte slot index inside the 16 KB operand stage
    {
      const int x = p >> 2, pc = p & 3;
      const int c = pc ^ g2_swz_x<MF16>(x);
      const int gr = min(m0 + x, M - 1);
      a_off[i] = (long)gr * lda + 8 * c;
    }
    if (B_KROWS) {
      const int kr = p >> 5, ph = p & 31;
      const int lg = ph ^ g2_swz_k<MF16>(kr);
      const int gc = min(n0 + 8 * lg, N - 8);
      b_off[i] = gc;                                      // + (k0 + kr) * ldb at issue time (row clamp)
    } else {
      const int x = p >> 2, pc = p & 3;
      const int c = pc ^ g2_swz_x<MF16>(x);
      const int gr = min(n0 + x, N - 1);
      b_off[i] = (long)gr * ldb + 8 * c;
    }
  }

  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) short*)lds;
  auto issue = [&](int kt) {
    const int k0 = kt * G2_BK;
    const uint32_t st = lds_base + (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);   // byte offset of the stage
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int slot0 = wave * 128 + i * 64;              // wave-uniform LDS destination (16-byte slots)
      const uint32_t dstA = __builtin_amdgcn_readfirstlane(st + slot0 * 16);
      const uint32_t dstB = __builtin_amdgcn_readfirstlane(st + 16384 + slot0 * 16);
      gb_dma16(A + a_off[i] + k0, dstA);
      if (B_KROWS) {
        const int kr = (slot0 + lane) >> 5;
        const int gk = min(k0 + kr, K - 1);
        gb_dma16(B + (long)gk * ldb + b_off[i], dstB);
      } else {
        gb_dma16(B + b_off[i] + k0, dstB);
      }
    }
  };

  constexpr int NI = MF16 ? 8 : 4, NJ = MF16 ? 4 : 2, NR = MF16 ? 4 : 16;   // accumulator blocks of the wave's 128 x 64
  typedef typename std::conditional<MF16, f32x4, f32x16>::type acc_t;
  acc_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + G2_BK - 1) / G2_BK;
  for (int s = 0; s < G2_ST - 1 && s < nk; ++s) issue(s);

  const int fr = lane & 31, fh = lane >> 5;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  // accumulator element (i, j, r) -> row / column inside the wave's 128 x 64
  auto erow = [&](int i, int r) { return MF16 ? i * 16 + 4 * tg + r : i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh; };
  auto ecol = [&](int j) { return MF16 ? j * 16 + (lane & 15) : j * 32 + fr; };
  if constexpr (MF16) {
    uint32_t aA, aB[4];
    {
      const int x = wm0 + (lane & 15);                         // + 16 i rows = + 1024 i bytes (swizzle unchanged)
      aA = lds_base + (uint32_t)(x * 64 + ((tg ^ g2_swz_x<true>(x)) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (B_KROWS) {
          const int kr = 8 * tg + tq;                           // second read: + 4 k rows = + 2048 bytes
          const int nc = wn0 + 16 * j + 4 * tp;
          const int ph = (nc >> 3) ^ g2_swz_k<true>(kr);
          aB[j] = lds_base + 16384u + (uint32_t)(2 * (kr * 256 + ph * 8 + (nc & 7)));
        } else {
          const int xb = wn0 + (lane & 15);                     // + 16 j rows = + 1024 j bytes
          aB[j] = lds_base + 16384u + (uint32_t)(xb * 64 + ((tg ^ g2_swz_x<true>(xb)) << 4)) + 1024u * j;
        }
      }
    }
    auto rdAL = [&](int kt, v8s (&a)[4]) {                     // row blocks 0-3 of stage kt
      const uint32_t pa = aA + (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
      a[0] = g2_lds128<0>(pa); a[1] = g2_lds128<1024>(pa); a[2] = g2_lds128<2048>(pa); a[3] = g2_lds128<3072>(pa);
    };
    auto rdAH = [&](int kt, v8s (&a)[4]) {                     // row blocks 4-7
      const uint32_t pa = aA + (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
      a[0] = g2_lds128<4096>(pa); a[1] = g2_lds128<5120>(pa); a[2] = g2_lds128<6144>(pa); a[3] = g2_lds128<7168>(pa);
    };
    auto rdB = [&](auto half, int kt, v8s (&b)[2]) {           // column blocks 2 half, 2 half + 1
      constexpr int J0 = decltype(half)::value * 2;
      const uint32_t so = (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (B_KROWS) {
          const v4s l = g2_ldstr<0>(aB[J0 + j] + so), h = g2_ldstr<2048>(aB[J0 + j] + so);
          b[j] = (v8s){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
        } else {
          b[j] = g2_lds128<0>(aB[J0 + j] + so);
        }
      }
    };
    auto quad = [&](auto ih, auto jh, const v8s (&a)[4], const v8s (&b)[2]) {
      constexpr int I0 = decltype(ih)::value * 4, J0 = decltype(jh)::value * 2;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[i]),
                                                                         __builtin_bit_cast(v8bf, b[j]), acc[I0 + i][J0 + j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // The wave's 8 x 4 blocks as four quadrants (rows lo / hi x columns lo / hi) of 8 MFMAs; one register set per operand
    // half (aL, aH, bL, bH) and the fragments of the NEXT k-tile are read into a set as soon as its last quadrant has
    // issued, a quadrant or more ahead of their first use.  The order that makes this work alternates between k-tiles:
    //   X: LL LH | HH HL     reads after LH: aL', after HH: bH', after HL: aH' bL'
    //   Y: LH LL | HL HH     reads after LL: aL', after HL: bL', after HH: aH' bH'     ( | = vmcnt + barrier + DMA issue)
    // LDS returns in order: the first quadrant of a k-tile waits for the two oldest groups, the second for everything.
    typedef std::integral_constant<int, 0> LO;
    typedef std::integral_constant<int, 1> HI;
#define G2_WAIT_FIRST() do { if (B_KROWS) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); \
                             else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); } while (0)
#define G2_WAIT_ALL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    v8s aL[4], aH[4], bL[2], bH[2];
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rdAL(0, aL); rdB(LO{}, 0, bL); rdAH(0, aH); rdB(HI{}, 0, bH);
    auto mid = [&](int kt) {                       // stage kt+1 has landed everywhere; refill the stage of k-tile kt-1
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + G2_ST - 1 < nk) issue(kt + G2_ST - 1);
    };
    int kt = 0;
    for (; kt + G2_ST < nk; kt += 2) {             // branch-free steady state: both k-tiles have a successor and a refill
      G2_WAIT_FIRST(); quad(LO{}, LO{}, aL, bL);
      G2_WAIT_ALL();   quad(LO{}, HI{}, aL, bH);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(kt + G2_ST - 1);
      rdAL(kt + 1, aL);
      quad(HI{}, HI{}, aH, bH);
      rdB(HI{}, kt + 1, bH);
      quad(HI{}, LO{}, aH, bL);
      rdAH(kt + 1, aH); rdB(LO{}, kt + 1, bL);
      G2_WAIT_FIRST(); quad(LO{}, HI{}, aL, bH);
      G2_WAIT_ALL();   quad(LO{}, LO{}, aL, bL);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(kt + G2_ST);
      rdAL(kt + 2, aL);
      quad(HI{}, LO{}, aH, bL);
      rdB(LO{}, kt + 2, bL);
      quad(HI{}, HI{}, aH, bH);
      rdAH(kt + 2, aH); rdB(HI{}, kt + 2, bH);
    }
    // (fragment reads are in flight here; the wait keeps any register shuffling between the two loops behind them)
    G2_WAIT_ALL();
    for (;; kt += 2) {
      {                                            // X
        const bool next = kt + 1 < nk;
        G2_WAIT_FIRST(); quad(LO{}, LO{}, aL, bL);
        G2_WAIT_ALL();   quad(LO{}, HI{}, aL, bH);
        if (next) { mid(kt); rdAL(kt + 1, aL); }
        quad(HI{}, HI{}, aH, bH);
        if (next) rdB(HI{}, kt + 1, bH);
        quad(HI{}, LO{}, aH, bL);
        if (!next) break;
        rdAH(kt + 1, aH); rdB(LO{}, kt + 1, bL);
      }
      {                                            // Y
        const int k1 = kt + 1;
        const bool next = k1 + 1 < nk;
        G2_WAIT_FIRST(); quad(LO{}, HI{}, aL, bH);
        G2_WAIT_ALL();   quad(LO{}, LO{}, aL, bL);
        if (next) { mid(k1); rdAL(k1 + 1, aL); }
        quad(HI{}, LO{}, aH, bL);
        if (next) rdB(LO{}, k1 + 1, bL);
        quad(HI{}, HI{}, aH, bH);
        if (!next) break;
        rdAH(k1 + 1, aH); rdB(HI{}, k1 + 1, bH);
      }
    }
#undef G2_WAIT_FIRST
#undef G2_WAIT_ALL
  } else {
  auto mma = [&](const v8s (&a)[4], const v8s (&b)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a[i]),
                                                             __builtin_bit_cast(v8bf, b[j]), acc[i][j], 0, 0, 0);
  };
    // software-pipelined: the LDS reads of the next half k-tile are in flight under the MFMAs of the current one.
    // Stage kt+1 is awaited (own DMAs: vmcnt; everybody's: barrier) in the MIDDLE of k-tile kt, where the refill of
    // the stage consumed in k-tile kt-1 is issued too: two k-tiles of DMA in flight, one landed, one being read.
    // per-lane byte addresses inside a stage (the swizzle makes s / j an XOR, not an add, where noted)
    uint32_t aA[2], aB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int x = wm0 + fr;                                  // + 32 i rows = + 2048 i bytes (swizzle unchanged)
      aA[q] = lds_base + (uint32_t)(x * 64 + (((2 * q + fh) ^ ((x >> 2) & 3)) << 4));
      if (B_KROWS) {                                           // q = j; s adds 16 k rows = 8192 bytes
        const int kr = 8 * (tg >> 1) + tq;
        const int nc = wn0 + q * 32 + 16 * (tg & 1) + 4 * tp;
        const int ph = (nc >> 3) ^ ((kr & 3) << 2);
        aB[q] = lds_base + 16384u + (uint32_t)(2 * (kr * 256 + ph * 8 + (nc & 7)));
      } else {                                                 // q = s; j adds 32 rows = 2048 bytes
        const int xb = wn0 + fr;
        aB[q] = lds_base + 16384u + (uint32_t)(xb * 64 + (((2 * q + fh) ^ ((xb >> 2) & 3)) << 4));
      }
    }
    auto rd0 = [&](int kt, v8s (&a)[4], v8s (&b)[2]) {           // first half (k 0..15) of stage kt
      const uint32_t so = (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
      const uint32_t pa = aA[0] + so;
      a[0] = g2_lds128<0>(pa); a[1] = g2_lds128<2048>(pa); a[2] = g2_lds128<4096>(pa); a[3] = g2_lds128<6144>(pa);
      if (B_KROWS) {
        const uint32_t p0 = aB[0] + so, p1 = aB[1] + so;
        v4s l0 = g2_ldstr<0>(p0), h0 = g2_ldstr<2048>(p0), l1 = g2_ldstr<0>(p1), h1 = g2_ldstr<2048>(p1);
        b[0] = (v8s){l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
        b[1] = (v8s){l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
      } else {
        const uint32_t pb = aB[0] + so;
        b[0] = g2_lds128<0>(pb); b[1] = g2_lds128<2048>(pb);
      }
    };
    auto rd1 = [&](int kt, v8s (&a)[4], v8s (&b)[2]) {           // second half (k 16..31)
      const uint32_t so = (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
      const uint32_t pa = aA[1] + so;
      a[0] = g2_lds128<0>(pa); a[1] = g2_lds128<2048>(pa); a[2] = g2_lds128<4096>(pa); a[3] = g2_lds128<6144>(pa);
      if (B_KROWS) {
        const uint32_t p0 = aB[0] + so, p1 = aB[1] + so;
        v4s l0 = g2_ldstr<8192>(p0), h0 = g2_ldstr<10240>(p0), l1 = g2_ldstr<8192>(p1), h1 = g2_ldstr<10240>(p1);
        b[0] = (v8s){l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
        b[1] = (v8s){l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
      } else {
        const uint32_t pb = aB[1] + so;
        b[0] = g2_lds128<0>(pb); b[1] = g2_lds128<2048>(pb);
      }
    };
    // LDS reads per half k-tile and wave: 4 A + (4 transposing | 2 plain) B
#define G2_WAIT_PREV() do { if (B_KROWS) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); \
                            else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); } while (0)
    v8s a0[4], b0[2], a1[4], b1[2];
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd0(0, a0, b0);
    int kt = 0;
    for (; kt + G2_ST - 1 < nk; ++kt) {          // branch-free steady state
      rd1(kt, a1, b1);
      G2_WAIT_PREV();                            // a0/b0 have landed; a1/b1 stay in flight under the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(kt + G2_ST - 1);
      rd0(kt + 1, a0, b0);
      G2_WAIT_PREV();
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; kt < nk; ++kt) {                      // last three k-tiles: nothing left to issue
      rd1(kt, a1, b1);
      G2_WAIT_PREV();
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rd0(kt + 1, a0, b0);
        G2_WAIT_PREV();
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef G2_WAIT_PREV
  }

  if (!D) {
    // bf16-only result: the tile goes through the (now idle) ring LDS as [256][256] bf16 so that the global
    // stores are whole 16-byte chunks of 512-byte rows instead of 2-byte scatters.  Neighbouring lanes swap one
    // value (DPP quad_perm [1,0,3,2]): even lanes pack row r, odd lanes row r+1 -> one ds_write_b32 per pair.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned* stage = reinterpret_cast<unsigned*>(lds);            // [256 rows][128 dwords]
    const bool odd = fr & 1;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < NR; r += 2) {
          const float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
          const float send = odd ? v0 : v1;
          const float recv = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0xB1, 0xf, 0xf, true));
          __bf16 tl = (__bf16)(odd ? recv : v0), th = (__bf16)(odd ? v1 : recv);
          const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, tl) |
                              ((unsigned)__builtin_bit_cast(unsigned short, th) << 16);
          const int row = wm0 + erow(i, r) + (odd ? 1 : 0);
          const int cd = (wn0 + (ecol(j) & ~1)) >> 1;              // dword column
          stage[row * 128 + (cd ^ ((row & 7) << 2))] = pk;          // 16-byte chunks XOR-swizzled by row
        }
    __builtin_amdgcn_s_barrier();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int id = it * 512 + tid;
      const int row = id >> 5, ch = id & 31;
      const int m = m0 + row, n = n0 + ch * 8;
      if (m >= M || n >= N) continue;
      uint4 v = *reinterpret_cast<const uint4*>(&stage[row * 128 + ((ch ^ (row & 7)) << 2)]);
      unsigned short* dst = Dbf + (long)m * ldd + n;
      if (n + 8 <= N && ((ldd & 7) == 0)) {
        if (beta) {
          const uint4 o = *reinterpret_cast<const uint4*>(dst);
          const unsigned ov[4] = {o.x, o.y, o.z, o.w};
          unsigned nv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(nv[q] << 16) + __uint_as_float(ov[q] << 16);
            const float hi = __uint_as_float(nv[q] & 0xffff0000u) + __uint_as_float(ov[q] & 0xffff0000u);
            __bf16 tl = (__bf16)lo, th = (__bf16)hi;
            nv[q] = (unsigned)__builtin_bit_cast(unsigned short, tl) | ((unsigned)__builtin_bit_cast(unsigned short, th) << 16);
          }
          v = make_uint4(nv[0], nv[1], nv[2], nv[3]);
        }
        *reinterpret_cast<uint4*>(dst) = v;
      } else {
        const unsigned nv[4] = {v.x, v.y, v.z, v.w};
        for (int q = 0; q < 8 && n + q < N; ++q) {
          float x = __uint_as_float((q & 1) ? (nv[q >> 1] & 0xffff0000u) : (nv[q >> 1] << 16));
          if (beta) x += __uint_as_float(((unsigned)dst[q]) << 16);
          __bf16 t = (__bf16)x;
          dst[q] = __builtin_bit_cast(unsigned short, t);
        }
      }
    }
    return;
  }
  // fp32 result.  Output-bound shapes (the head's K = 256 / 512 products write 1.6 GB per launch) need whole 16-byte
  // pieces of 1-KB rows in flight, not one 4-byte store per lane and row: the tile leaves in two halves of 128 rows
  // through the (now idle) 128-KB ring -- accumulator layout in (bias / ReLU applied), rows out; the ReLU gate and the
  // beta read-modify-write happen on the way out with the same 16-byte pieces, and so does the optional bf16 copy.
  if ((ldd & 3) == 0 && (((uintptr_t)D) & 15) == 0 && (!mask || (((uintptr_t)mask) & 15) == 0) &&
      (!Dbf || (((uintptr_t)Dbf) & 7) == 0) && (N & 3) == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* stage = reinterpret_cast<float*>(lds);                  // [128 rows][256 floats], 32-float halves XOR-swizzled
    for (int half = 0; half < 2; ++half) {
      if ((wave >> 2) == half) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int col = wn0 + ecol(j);
            const int n = n0 + col;
            const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              const int row = erow(i, r);
              float v = acc[i][j][r] + bv;
              if (relu) v = fmaxf(v, 0.f);
              stage[row * 256 + (col ^ (((row >> 2) & 1) << 5))] = v;      // the two lane halves (rows +4) on other banks
            }
          }
      }
      __builtin_amdgcn_s_barrier();
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int id = it * 512 + tid;
        const int row = id >> 6, c4 = id & 63;
        const int m = m0 + half * 128 + row, n = n0 + 4 * c4;
        if (m >= M || n >= N) continue;
        float4 v = *reinterpret_cast<const float4*>(&stage[row * 256 + ((4 * c4) ^ (((row >> 2) & 1) << 5))]);
        const long o = (long)m * ldd + n;
        if (mask) {
          const float4 mk = *reinterpret_cast<const float4*>(mask + o);
          if (!(mk.x > 0.f)) v.x = 0.f;
          if (!(mk.y > 0.f)) v.y = 0.f;
          if (!(mk.z > 0.f)) v.z = 0.f;
          if (!(mk.w > 0.f)) v.w = 0.f;
        }
        if (beta) {
          const float4 ov = *reinterpret_cast<const float4*>(D + o);
          v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
        }
        *reinterpret_cast<float4*>(D + o) = v;
        if (Dbf) {
          __bf16 t0 = (__bf16)v.x, t1 = (__bf16)v.y, t2 = (__bf16)v.z, t3 = (__bf16)v.w;
          *reinterpret_cast<uint2*>(Dbf + o) =
              make_uint2((unsigned)__builtin_bit_cast(unsigned short, t0) | ((unsigned)__builtin_bit_cast(unsigned short, t1) << 16),
                         (unsigned)__builtin_bit_cast(unsigned short, t2) | ((unsigned)__builtin_bit_cast(unsigned short, t3) << 16));
        }
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn0 + ecol(j);
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int m = m0 + wm0 + erow(i, r);
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (bias) v += bias[n];
        if (relu) v = fmaxf(v, 0.f);
        if (mask && !(mask[(long)m * ldd + n] > 0.f)) v = 0.f;       // ReLU backward: gradient gated by the forward value
        if (D) {
          float* o = D + (long)m * ldd + n;
          if (beta) v += *o;
          *o = v;
        } else if (beta) {      // bf16-only output: accumulate onto the stored bf16 value
          v += __uint_as_float(((unsigned)Dbf[(long)m * ldd + n]) << 16);
        }
        if (Dbf) { __bf16 t = (__bf16)v; Dbf[(long)m * ldd + n] = __builtin_bit_cast(unsigned short, t); }
      }
    }
}

// MFMA shape of the ring kernel: MO_GEMM_MFMA=32 selects v_mfma_f32_32x32x16_bf16 (A/B switch, read once)
static bool g2_mfma16() {
  static const bool v = [] { const char* e = getenv("MO_GEMM_MFMA"); return !(e && atoi(e) == 32); }();
  return v;
}
// a_kpad: number of readable columns of A (>= K rounded up to 32, zero beyond K)
// fp32-result epilogue options (the head of the throughput mode): + bias[n], ReLU, and a ReLU-backward gate
// (result zeroed where mask[m][n] <= 0; mask has D's shape and leading dimension)
extern "C" int mo_gemm_bf16_256_ex(const void* A, int lda, int a_kpad, const void* B, int ldb, int b_krows, float* D,
                                   int ldd, int M, int N, int K, int beta, void* D_bf16, const float* bias, int relu,
                                   const float* mask, void* stream) {
  MO_CHECK_ARG(A && B && (D || D_bf16) && M > 0 && N >= 8 && K > 0);
  MO_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (!b_krows || (N % 8) == 0));
  MO_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0);
  MO_CHECK_ARG(D || (!bias && !relu && !mask));      // the options live on the fp32-result path
  const int kpad = (K + 31) / 32 * 32;
  MO_CHECK_ARG(a_kpad >= kpad && lda >= kpad);
  MO_CHECK_ARG(b_krows || (K % 32) == 0);          // XROWS B has no zero-padded operand to mask a K tail
  const int gm = mo_cdiv(M, G2_BM), gn = mo_cdiv(N, G2_BN);
  dim3 grid(8 * ((gm + 1) / 2) * ((gn + 3) / 4));
#define G2_LAUNCH(KR, MF) hipLaunchKernelGGL((gemm_bf16_256_kernel<KR, MF>), grid, dim3(512), 0, (hipStream_t)stream, \
    (const short*)A, lda, (const short*)B, ldb, D, ldd, M, N, K, beta, (unsigned short*)D_bf16, bias, relu, mask)
  if (g2_mfma16()) { if (b_krows) G2_LAUNCH(true, true); else G2_LAUNCH(false, true); }
  else { if (b_krows) G2_LAUNCH(true, false); else G2_LAUNCH(false, false); }
#undef G2_LAUNCH
  return mo_launch_status();
}
extern "C" int mo_gemm_bf16_256(const void* A, int lda, int a_kpad, const void* B, int ldb, int b_krows, float* D,
                                int ldd, int M, int N, int K, int beta, void* D_bf16, void* stream) {
  return mo_gemm_bf16_256_ex(A, lda, a_kpad, B, ldb, b_krows, D, ldd, M, N, K, beta, D_bf16, nullptr, 0, nullptr, stream);
}

// fp32 [rows][cols] -> bf16 [rows][ld_out] with zero fill of columns [cols, ld_out)
// ================================================================================================
// Weight gradient of a wide 1x1 conv as a split-K ring GEMM with BOTH operands k-major as they lie in HBM:
//   dW[m][n] = sum_k A[k][m] * B[k][n]      A: bf16 [K][M] (output gradient rows), B: bf16 [K][N] (input rows)
// Same 256x256x32 tile / 4-stage LDS-DMA ring / half-k-tile software pipeline as gemm_bf16_256_kernel; both LDS
// images are [k][512 B] (slot16 ^= (k&3)<<2) and both fragment kinds come from ds_read_b64_tr_b16.
// grid = (tiles, splits): split z owns k-tiles [z*kts, min((z+1)*kts, K/32)) and stores its fp32 tile to
// slab[z][M][N]; kk_reduce_kernel sums the slabs in a fixed order (deterministic).  K % 32 == 0, M, N % 8 == 0.
// ================================================================================================
__global__ void __launch_bounds__(512)
gemm_bf16_kk_kernel(const short* __restrict__ A, int lda, const short* __restrict__ B, int ldb, float* __restrict__ slab,
                    int M, int N, int nkt, int kts) {
  __shared__ __attribute__((aligned(16))) short lds[G2_ST * G2_STAGE_SHORTS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;
  const int gn = (N + G2_BN - 1) / G2_BN;
  const int m0 = (blockIdx.x / gn) * G2_BM, n0 = (blockIdx.x % gn) * G2_BN;
  const int kt0 = blockIdx.y * kts;
  const int nk = min(kts, nkt - kt0);
  float* __restrict__ D = slab + (long)blockIdx.y * M * N;

  // per-lane DMA sources: two A and two B instructions per wave per stage, each lane one 16-byte slot of a k row
  long a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 128 + i * 64 + lane;
    const int kr = p >> 5, ph = p & 31;
    const int lg = ph ^ ((kr & 3) << 2);
    a_off[i] = (long)(kt0 * G2_BK + kr) * lda + min(m0 + 8 * lg, M - 8);
    b_off[i] = (long)(kt0 * G2_BK + kr) * ldb + min(n0 + 8 * lg, N - 8);
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) short*)lds;
  auto issue = [&](int kt) {
    const uint32_t st = lds_base + (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int slot0 = wave * 128 + i * 64;
      const uint32_t dstA = __builtin_amdgcn_readfirstlane(st + slot0 * 16);
      const uint32_t dstB = __builtin_amdgcn_readfirstlane(st + 16384 + slot0 * 16);
      gb_dma16(A + a_off[i] + (long)kt * G2_BK * lda, dstA);
      gb_dma16(B + b_off[i] + (long)kt * G2_BK * ldb, dstB);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nk <= 0) return;
  for (int s = 0; s < G2_ST - 1 && s < nk; ++s) issue(s);

  const int fr = lane & 31, fh = lane >> 5;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  uint32_t aA[4], aB[2];
  {
    const int kr = 8 * (tg >> 1) + tq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nc = wm0 + q * 32 + 16 * (tg & 1) + 4 * tp;
      aA[q] = lds_base + (uint32_t)(2 * (kr * 256 + (((nc >> 3) ^ ((kr & 3) << 2)) << 3) + (nc & 7)));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nc = wn0 + q * 32 + 16 * (tg & 1) + 4 * tp;
      aB[q] = lds_base + 16384u + (uint32_t)(2 * (kr * 256 + (((nc >> 3) ^ ((kr & 3) << 2)) << 3) + (nc & 7)));
    }
  }
#define KK_FRAG(dst, addr, O) do { v4s lo_ = g2_ldstr<O>(addr), hi_ = g2_ldstr<O + 2048>(addr); \
    dst = (v8s){lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]}; } while (0)
  auto rd0 = [&](int kt, v8s (&a)[4], v8s (&b)[2]) {
    const uint32_t so = (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
    KK_FRAG(a[0], aA[0] + so, 0); KK_FRAG(a[1], aA[1] + so, 0); KK_FRAG(a[2], aA[2] + so, 0); KK_FRAG(a[3], aA[3] + so, 0);
    KK_FRAG(b[0], aB[0] + so, 0); KK_FRAG(b[1], aB[1] + so, 0);
  };
  auto rd1 = [&](int kt, v8s (&a)[4], v8s (&b)[2]) {
    const uint32_t so = (uint32_t)(kt % G2_ST) * (G2_STAGE_SHORTS * 2);
    KK_FRAG(a[0], aA[0] + so, 8192); KK_FRAG(a[1], aA[1] + so, 8192); KK_FRAG(a[2], aA[2] + so, 8192);
    KK_FRAG(a[3], aA[3] + so, 8192);
    KK_FRAG(b[0], aB[0] + so, 8192); KK_FRAG(b[1], aB[1] + so, 8192);
  };
#undef KK_FRAG
  auto mma = [&](const v8s (&a)[4], const v8s (&b)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a[i]),
                                                             __builtin_bit_cast(v8bf, b[j]), acc[i][j], 0, 0, 0);
  };
  // 12 transposing reads per half k-tile and wave
  v8s a0[4], b0[2], a1[4], b1[2];
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  rd0(0, a0, b0);
  int kt = 0;
  for (; kt + G2_ST - 1 < nk; ++kt) {
    rd1(kt, a1, b1);
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + G2_ST - 1);
    rd0(kt + 1, a0, b0);
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; kt < nk; ++kt) {
    rd1(kt, a1, b1);
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      rd0(kt + 1, a0, b0);
      asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn0 + j * 32 + fr;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m < M) D[(long)m * N + n] = acc[i][j][r];
      }
    }
}

__global__ void kk_reduce_kernel(const float4* __restrict__ slab, long n4, int ns, float4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = slab[i];
  for (int z = 1; z < ns; ++z) {
    const float4 v = slab[(long)z * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[i] = s;
}

static void kk_plan(int M, int N, long K, int& ns, int& kts) {
  const int tiles = mo_cdiv(M, G2_BM) * mo_cdiv(N, G2_BN);
  const int nkt = (int)(K / G2_BK);
  ns = mo_cdiv(256, tiles);                           // one round of the 256 CUs
  if (ns > nkt / 16) ns = nkt / 16 > 0 ? nkt / 16 : 1;  // at least 16 k-tiles per split
  kts = mo_cdiv(nkt, ns);
  ns = mo_cdiv(nkt, kts);
}
extern "C" long mo_wgrad_bf16_kk_ws_floats(int M, int N, long K) {
  int ns, kts;
  kk_plan(M, N, K, ns, kts);
  return (long)ns * M * N;
}
// dW[M][N] = A^T B over K rows (fp32 result); ws: mo_wgrad_bf16_kk_ws_floats(M, N, K) floats
extern "C" int mo_wgrad_bf16_kk(const void* A, int lda, const void* B, int ldb, long K, int M, int N, float* dW,
                                float* ws, void* stream) {
  MO_CHECK_ARG(A && B && dW && ws && M >= 8 && N >= 8 && K >= 32);
  MO_CHECK_ARG((K % G2_BK) == 0 && (M % 8) == 0 && (N % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0);
  MO_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)dW % 16) == 0 && ((long)M * N) % 4 == 0);
  MO_CHECK_ARG(K / G2_BK < (1L << 30));
  int ns, kts;
  kk_plan(M, N, K, ns, kts);
  dim3 grid(mo_cdiv(M, G2_BM) * mo_cdiv(N, G2_BN), ns);
  hipLaunchKernelGGL(gemm_bf16_kk_kernel, grid, dim3(512), 0, (hipStream_t)stream, (const short*)A, lda, (const short*)B,
                     ldb, ws, M, N, (int)(K / G2_BK), kts);
  const long n4 = (long)M * N / 4;
  hipLaunchKernelGGL(kk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)ws, n4, ns, (float4*)dW);
  return mo_launch_status();
}

__global__ void f32_to_bf16_pad_kernel(const float* __restrict__ x, int rows, int cols, unsigned short* __restrict__ y,
                                       int ld_out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * ld_out) return;
  const int r = (int)(i / ld_out), c = (int)(i - (long)r * ld_out);
  __bf16 t = (__bf16)(c < cols ? x[(long)r * cols + c] : 0.f);
  y[i] = __builtin_bit_cast(unsigned short, t);
}
extern "C" int mo_f32_to_bf16_padded(const float* x, int rows, int cols, void* y, int ld_out, void* stream) {
  MO_CHECK_ARG(x && y && rows > 0 && cols > 0 && ld_out >= cols);
  long n = (long)rows * ld_out;
  hipLaunchKernelGGL(f32_to_bf16_pad_kernel, dim3(mo_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, cols,
                     (unsigned short*)y, ld_out);
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Skip-path weight gradients of ALL layers as one k-major product (throughput mode): the cropped rows of every layer's
// gated output g_i that reach the head (graph_wavenet.py:230-236: the last Tf steps) are gathered once into a bf16
// matrix gcat[G*Tf][32*nl]; dW_all[Cs][32*nl] = dskip^T gcat is then ONE mo_wgrad_bf16_kk launch instead of nl passes over
// the fp32 dskip (786 MB each at the benchmark size), and mo_skip_wsplit hands layer i its 32 columns.
// ------------------------------------------------------------------------------------------------------------------
struct SkipGatherArgs { const float* g[8]; int Tout[8]; int nl; long G; int Tf; };
__global__ void skip_gather_bf16_kernel(SkipGatherArgs a, uint4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one 8-channel piece
  const int per_row = 4 * a.nl;
  const long row = i / per_row;
  if (row >= a.G * a.Tf) return;
  const int piece = (int)(i - row * per_row), layer = piece >> 2, c8 = piece & 3;
  const long grp = row / a.Tf; const int t = (int)(row - grp * a.Tf);
  const int To = a.Tout[layer];
  const float4* src = reinterpret_cast<const float4*>(a.g[layer] + ((grp * To + To - a.Tf + t) * 32 + 8 * c8));
  const float4 x = src[0], y = src[1];
  const float f[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
  unsigned short h[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { __bf16 tq = (__bf16)f[q]; h[q] = __builtin_bit_cast(unsigned short, tq); }
  out[i] = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16),
                      h[6] | ((unsigned)h[7] << 16));
}
extern "C" int mo_skip_gather_bf16(const float* const* g, const int* Tout, int nl, long G, int Tf, void* out_bf16,
                                   void* stream) {
  MO_CHECK_ARG(g && Tout && out_bf16 && nl >= 1 && nl <= 8 && G > 0 && Tf > 0 && ((uintptr_t)out_bf16 % 16) == 0);
  SkipGatherArgs a; a.nl = nl; a.G = G; a.Tf = Tf;
  for (int i = 0; i < 8; ++i) { a.g[i] = i < nl ? g[i] : nullptr; a.Tout[i] = i < nl ? Tout[i] : 1; }
  for (int i = 0; i < nl; ++i) MO_CHECK_ARG(g[i] && Tout[i] >= Tf && ((uintptr_t)g[i] % 16) == 0);
  const long n = G * Tf * 4L * nl;
  hipLaunchKernelGGL(skip_gather_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                     (uint4*)out_bf16);
  return mo_launch_status();
}
struct SkipSplitArgs { float* dW[8]; };
__global__ void skip_wsplit_kernel(const float* __restrict__ all, int Cs, int nl, SkipSplitArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cs * 32 * nl) return;
  const int co = i / (32 * nl), r = i - co * (32 * nl), layer = r >> 5, c = r & 31;
  a.dW[layer][co * 32 + c] = all[i];
}
extern "C" int mo_skip_wsplit(const float* dW_all, int Cs, int nl, float* const* dW, void* stream) {
  MO_CHECK_ARG(dW_all && dW && Cs > 0 && nl >= 1 && nl <= 8);
  SkipSplitArgs a;
  for (int i = 0; i < 8; ++i) a.dW[i] = i < nl ? dW[i] : nullptr;
  for (int i = 0; i < nl; ++i) MO_CHECK_ARG(dW[i]);
  hipLaunchKernelGGL(skip_wsplit_kernel, dim3(mo_cdiv((long)Cs * 32 * nl, 256)), dim3(256), 0, (hipStream_t)stream, dW_all,
                     Cs, nl, a);
  return mo_launch_status();
}
