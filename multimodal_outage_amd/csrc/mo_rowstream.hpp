// Row-streaming kernels for the [P][32]-channel tensors of the gwnet block (gfx950).
//
// The nbtc activations are tall matrices of 128-byte rows.  Every kernel here lets each wave stream rows
// on its own -- no workgroup barriers in the main loop, tens of loads in flight per wave, 8 waves per CU --
// which is what an HBM-bound pass needs on this chip (bytes in flight per CU / latency), and feeds the fp32
// MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products) straight from registers where the operand layout allows.
#pragma once
#include "mo_gemm.hpp"

// ------------------------------------------------------------------------------------------------
// Weight gradients: slab[z][m][n] = sum_{p in rows of workgroup z} a(p, m) * b(p, n)
//   a: one segment [P][32*MA] (dropout mask regenerated on load), b: NB segments [.][32] with per-segment
//   row map / folded BatchNorm affine / ReLU.  The contraction index is the ROW, so a half-wave's 128-byte
//   row read IS the MFMA operand (lane = column, lane>>5 = k): global_load_dword -> v_mfma, no LDS.
//   cs[z][m] = column sums of a (the bias gradient) when cs != null.
// ------------------------------------------------------------------------------------------------
// row pairs in flight per wave: (MA+NB)*U loads <= 56 (the vmcnt counter is 6 bits), at most 12 pairs
__host__ __device__ constexpr int rsw_u(int MA, int NB) {
  return 48 / (MA + NB) > 12 ? 12 : (48 / (MA + NB) < 2 ? 2 : 48 / (MA + NB));
}

template <int MA, int NB, bool MAPPED, bool BBF = false>     // BBF: b segments 1.. are stored as bf16
__global__ __launch_bounds__(256, 2) void rs_wgrad_kernel(MoOperand A, MoOperand B, float* __restrict__ slab,
                                                         float* __restrict__ cs, long P, int post_b) {
  extern __shared__ float rs_sm[];
  constexpr int RSW_U = rsw_u(MA, NB), RSW_ROWS = 2 * RSW_U;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  const long nwaves = (long)gridDim.x * 4;
  const long w = (long)wave * gridDim.x + blockIdx.x;      // neighbouring workgroups stream neighbouring chunks
  const long nchunk = (P + RSW_ROWS - 1) / RSW_ROWS;
  constexpr int LDA = 32 * MA;

  f32x16 acc[MA][NB];
#pragma unroll
  for (int i = 0; i < MA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float csum[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) csum[i] = 0.f;

  // per-lane constants of the b segments
  float bsc[NB], bsh[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    bsc[j] = B.seg[j].scale ? mo_gload(B.seg[j].scale + c) : 1.f;
    bsh[j] = B.seg[j].scale ? mo_gload(B.seg[j].shift + c) : 0.f;
  }
  const float* aptr = A.seg[0].ptr;
  const uint32_t dseed = A.seg[0].drop_seed, dthresh = A.seg[0].drop_thresh;
  const float dscale = A.seg[0].drop_scale;
  // row map p -> (g, t): one real division per wave, then incremental (chunks advance by a fixed stride);
  // the in-chunk offsets use a 16-bit reciprocal (exact for To < 200, checked by the host)
  const int To = MAPPED ? B.seg[0].To : 1;
  const unsigned inv16 = 65536u / (unsigned)To + 1u;
  const unsigned stride = (unsigned)(nwaves * RSW_ROWS);
  const unsigned dq = stride / (unsigned)To, dr = stride - dq * (unsigned)To;
  unsigned g0 = 0, t0 = 0;
  if (MAPPED) { const unsigned p0 = (unsigned)(w * RSW_ROWS); g0 = p0 / (unsigned)To; t0 = p0 - g0 * (unsigned)To; }

  float ra[RSW_U][MA], rb[RSW_U][NB];
  uint32_t aidx[RSW_U];          // element index of ra[u][0] (dropout) or ~0u when the row is past P
  uint32_t bok[RSW_U];           // bit j: b segment j row is in range

  auto issue = [&](long chunk, int u) {
    const long p = chunk * RSW_ROWS + 2 * u + half;
    const bool ok = p < P;
    const long pc = ok ? p : 0;
#pragma unroll
    for (int i = 0; i < MA; ++i) ra[u][i] = mo_gload(aptr + pc * LDA + i * 32 + c);
    aidx[u] = ok ? (uint32_t)(pc * LDA + c) : 0xffffffffu;
    uint32_t okm = 0;
    long g = pc; int t = 0;
    if (MAPPED) {
      const unsigned x = t0 + (unsigned)(2 * u + half);
      const unsigned q = (x * inv16) >> 16;
      t = (int)(x - q * (unsigned)To);
      g = (long)(g0 + q);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      long srow = pc; bool v = ok;
      if (MAPPED) {
        const int tt = t + B.seg[j].off;
        v = ok & ((unsigned)tt < (unsigned)B.seg[j].Ti);
        srow = v ? g * B.seg[j].Ti + tt : 0;
      }
      if (BBF && j > 0) {
        const unsigned short hv = *(const __attribute__((address_space(1))) unsigned short*)(uintptr_t)(
            reinterpret_cast<const unsigned short*>(B.seg[j].ptr) + srow * 32 + c);
        rb[u][j] = __uint_as_float(((unsigned)hv) << 16);
      } else {
        rb[u][j] = mo_gload(B.seg[j].ptr + srow * 32 + c);
      }
      okm |= (v ? 1u : 0u) << j;
    }
    bok[u] = okm;
  };

  auto consume = [&](int u) {
    float av[MA], bv[NB];
#pragma unroll
    for (int i = 0; i < MA; ++i) {
      float v = ra[u][i];
      if (dthresh) {
        const uint32_t h = mo_hash32(dseed, aidx[u] + 32u * i);
        v = (h >= dthresh) ? v * dscale : 0.f;
      }
      if (aidx[u] == 0xffffffffu) v = 0.f;
      av[i] = v;
      csum[i] += v;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float v = rb[u][j];
      if (post_b & 1) v = v * bsc[j] + bsh[j];
      if (post_b & 2) v = fmaxf(v, 0.f);
      if (!((bok[u] >> j) & 1)) v = 0.f;
      bv[j] = v;
    }
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
  };

  // Straight-line steady state (no branch around the refill, or the compiler's vmcnt bookkeeping turns
  // conservative and drains the pipeline once per chunk): the last iteration re-reads its own chunk.
  if (w < nchunk) {
#pragma unroll
    for (int u = 0; u < RSW_U; ++u) issue(w, u);
    for (long chunk = w; chunk < nchunk; chunk += nwaves) {
      const bool more = chunk + nwaves < nchunk;
      const long nx = more ? chunk + nwaves : chunk;
      if (MAPPED && more) { t0 += dr; g0 += dq; if (t0 >= (unsigned)To) { t0 -= (unsigned)To; g0 += 1; } }
#pragma unroll
      for (int u = 0; u < RSW_U; ++u) {
        consume(u);
        issue(nx, u);                   // refill the slot just consumed: RSW_U-1 pairs stay in flight
      }
    }
  }

  // ---- workgroup reduction of the four waves' partial tiles through LDS, then one slab per workgroup
  constexpr int TILE = MA * NB * 16 * 64;            // floats per wave
  float* buf0 = rs_sm;                               // [TILE] x 2
  float* buf1 = rs_sm + TILE;
  auto put = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((i * NB + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * NB + j) * 16 + r) * 64 + lane];
  };
  if (wave == 2) put(buf0);
  if (wave == 3) put(buf1);
  __syncthreads();
  if (wave == 0) add(buf0);
  if (wave == 1) add(buf1);
  __syncthreads();
  if (wave == 1) put(buf0);
  __syncthreads();
  if (wave == 0) {
    add(buf0);
    float* out = slab + (long)blockIdx.x * (32 * MA) * (32 * NB);
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          out[(long)m * (32 * NB) + j * 32 + c] = acc[i][j][r];
        }
  }
  if (cs) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MA; ++i) rs_sm[(wave * MA + i) * 64 + lane] = csum[i];
    __syncthreads();
    if (threadIdx.x < 32 * MA) {
      const int i = threadIdx.x >> 5, cc = threadIdx.x & 31;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) s += rs_sm[(q * MA + i) * 64 + cc] + rs_sm[(q * MA + i) * 64 + 32 + cc];
      cs[(long)blockIdx.x * (32 * MA) + threadIdx.x] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row contractions over the CHANNEL axis (gcn mlp forward / data gradient): out[p][n] = sum_k x[p][k] w[k][n].
// The MFMA wants lane = row for x, the coalesced load gives lane = column, so every 32x32 block takes one
// trip through a wave-private LDS tile (ds_write_b128 rows in, ds_read_b128 rows out, stride 36 floats: both
// conflict-free); no workgroup barrier after the weights are staged.  Rows are dealt to waves in runs of
// 128 (= one BatchNorm partial row), neighbouring workgroups on neighbouring runs.
// ------------------------------------------------------------------------------------------------
#define RS_LDX 36
#define RS_LDW 33

struct RsMlpArgs {
  const float* src[MO_MAX_SEG];   // fwd: the ns sources; bwd: src[0] = dh
  float* out[MO_MAX_SEG];         // fwd: out[0] = h; bwd: the ns source gradients
  unsigned short* out_bf;         // bwd: optional bf16 copy of out[ns-1]
  const float* W;                 // reference mlp weight [32][32*ns]
  const float* bias;              // fwd
  const float* res; const float* rscale; const float* rshift;   // fwd residual (row map Tout -> Tin, offset)
  float* partial;                 // fwd: BatchNorm partial sums [ceil(P/128)][64]
  long P;
  int Tout, Tin;
  uint32_t drop_seed, drop_thresh; float drop_scale;
  int s0bf;                       // fwd, bf16 sources: source 0 (the gated TCN output) is read from its bf16 copy too
};

// Bounds-checked stores through a buffer descriptor (num_records = the tensor's bytes): rows past P are
// dropped by the hardware, so the streaming loops stay branch-free (a branch around a store makes the
// compiler's vmcnt bookkeeping conservative and drains the load pipeline).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ void rs_store_f32(__amdgpu_buffer_rsrc_t r, long elem, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)(unsigned)(elem * 4), 0, 0);
}
__device__ __forceinline__ void rs_store_bf16(__amdgpu_buffer_rsrc_t r, long elem, float v) {
  __bf16 tb = (__bf16)v;
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, tb), r, (int)(unsigned)(elem * 2), 0, 0);
}

typedef unsigned rs_v4u __attribute__((ext_vector_type(4)));
// Loads go through buffer descriptors too: one 32-bit byte offset per lane serves every source tensor of the
// same shape (the descriptor carries the base), rows past the end read as zero.
__device__ __forceinline__ float4 rs_load4(__amdgpu_buffer_rsrc_t r, unsigned byteoff) {
  const rs_v4u x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byteoff, 0, 0);
  const unsigned x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];   // (bit_cast of a vector-element lvalue reads element 0)
  return make_float4(__uint_as_float(x0), __uint_as_float(x1), __uint_as_float(x2), __uint_as_float(x3));
}
__device__ __forceinline__ float rs_load1(__amdgpu_buffer_rsrc_t r, unsigned byteoff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byteoff, 0, 0));
}
// coalesced 32-row block load of [.][32] rows: load j covers rows 8j + (lane>>3), columns 4*(lane&7)..+3;
// lane_off = (lane>>3)*128 + (lane&7)*16
__device__ __forceinline__ void rs_issue_block(float4 (&v)[4], __amdgpu_buffer_rsrc_t r, long row0, unsigned lane_off) {
  const unsigned base = (unsigned)(row0 * 128) + lane_off;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = rs_load4(r, base + 1024u * j);
}
// bf16-stored [.][32] rows (64 B): two 16-byte loads cover the 32-row tile, load j = rows 16j + (lane>>2), eight
// columns from 8*(lane&3); lane_off_bf = (lane>>2)*64 + (lane&3)*16.  The raw bits wait in v[0..1].
__device__ __forceinline__ void rs_issue_block_bf(float4 (&v)[4], __amdgpu_buffer_rsrc_t r, long row0, unsigned lane_off_bf) {
  const unsigned base = (unsigned)(row0 * 64) + lane_off_bf;
#pragma unroll
  for (int j = 0; j < 2; ++j) v[j] = rs_load4(r, base + 1024u * j);
}
__device__ __forceinline__ void rs_put_bf(float* X, const float4 (&v)[4], int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned u0 = __float_as_uint(v[j].x), u1 = __float_as_uint(v[j].y), u2 = __float_as_uint(v[j].z),
                   u3 = __float_as_uint(v[j].w);
    float* d = &X[(16 * j + (lane >> 2)) * RS_LDX + 8 * (lane & 3)];
    *reinterpret_cast<float4*>(d) = make_float4(__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u),
                                                __uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u));
    *reinterpret_cast<float4*>(d + 4) = make_float4(__uint_as_float(u2 << 16), __uint_as_float(u2 & 0xffff0000u),
                                                    __uint_as_float(u3 << 16), __uint_as_float(u3 & 0xffff0000u));
  }
}
// Two accumulator registers holding rows i, i+1 of a 32x32 tile (r even) as bf16: neighbouring lanes swap one
// value (DPP quad_perm [1,0,3,2]) so that every lane stores one packed dword -- even lanes row i, odd lanes
// row i+1 -- instead of two 2-byte stores.  elem0 = element index of (row i, column 0).
__device__ __forceinline__ void rs_store_bf16_pair(__amdgpu_buffer_rsrc_t r, long elem0, int n, float v0, float v1,
                                                   int row_stride = 32) {
  const bool odd = n & 1;
  const float send = odd ? v0 : v1;
  const float recv = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0xB1, 0xf, 0xf, true));
  const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
  __bf16 tl = (__bf16)lo, th = (__bf16)hi;
  const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, tl) | ((unsigned)__builtin_bit_cast(unsigned short, th) << 16);
  const long e = elem0 + (odd ? row_stride : 0) + (n & ~1);
  __builtin_amdgcn_raw_buffer_store_b32(pk, r, (int)(unsigned)(e * 2), 0, 0);
}
__device__ __forceinline__ void rs_put(float* X, const float4 (&v)[4], int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4*>(&X[(8 * j + (lane >> 3)) * RS_LDX + 4 * (lane & 7)]) = v[j];
}
// a[t] = x[row lane&31][k = 16*(lane>>5) + t]
__device__ __forceinline__ void rs_get(const float* X, float (&a)[16], int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(&X[(lane & 31) * RS_LDX + 16 * (lane >> 5) + 4 * q]);
    a[4 * q + 0] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
  }
}

// ---- bf16-MFMA path of the throughput mode (v_mfma_f32_32x32x16_bf16: 16x fewer MFMA cycles than the exact
// fp32 32x32x2): the wave-private LDS tile holds bf16 [32 rows][32 k], row stride 80 B (conflict-free b128 reads)
#define RS_LDXB 40
typedef short rs_v8s __attribute__((ext_vector_type(8)));
typedef __bf16 rs_v8bf __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned rs_pack_bf16(float a, float b) {
  __bf16 ta = (__bf16)a, tb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ta) | ((unsigned)__builtin_bit_cast(unsigned short, tb) << 16);
}
// fp32 tile registers (rs_issue_block layout) -> bf16 LDS tile
__device__ __forceinline__ void rs_put_f2b(short* Xb, const float4 (&v)[4], int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<uint2*>(&Xb[(8 * j + (lane >> 3)) * RS_LDXB + 4 * (lane & 7)]) =
        make_uint2(rs_pack_bf16(v[j].x, v[j].y), rs_pack_bf16(v[j].z, v[j].w));
}
// bf16 tile registers (rs_issue_block_bf layout) -> bf16 LDS tile: a straight copy
__device__ __forceinline__ void rs_put_b2b(short* Xb, const float4 (&v)[4], int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    *reinterpret_cast<float4*>(&Xb[(16 * j + (lane >> 2)) * RS_LDXB + 8 * (lane & 3)]) = v[j];
}
// A fragments of the two K=16 MFMAs of a 32-wide contraction: lane = row (lane&31), k = 16h + 8*(lane>>5) + 0..7
__device__ __forceinline__ void rs_get_b(const short* Xb, rs_v8s (&a)[2], int lane) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
    a[h] = *reinterpret_cast<const rs_v8s*>(&Xb[(lane & 31) * RS_LDXB + 16 * h + 8 * (lane >> 5)]);
}

template <int NS, bool DROP, bool SBF, bool S0BF = false>   // SBF: sources 1.. are stored as bf16 (source 0, the gated TCN
                                                            // output, fp32 -- or, S0BF, read from its bf16 copy as well)
__global__ __launch_bounds__(256, 2) void rs_mlp_fwd_kernel(RsMlpArgs a) {
  // SBF also selects the bf16 MFMA: weights as bf16 [n][k] rows (stride KT+8), tiles as bf16
  __shared__ __attribute__((aligned(16))) float Ws[SBF ? 16 * (32 * NS + 8) : NS * 32 * RS_LDW];
  __shared__ __attribute__((aligned(16))) float Xs[4][SBF ? 16 * RS_LDXB : 32 * RS_LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  constexpr int KT = 32 * NS, LWB = KT + 8;
  short* Wb = reinterpret_cast<short*>(Ws);
  for (int idx = tid; idx < 32 * KT; idx += 256) {
    const int co = idx / KT, k = idx - co * KT;
    if (SBF) { __bf16 t = (__bf16)a.W[idx]; Wb[co * LWB + k] = __builtin_bit_cast(short, t); }
    else Ws[k * RS_LDW + co] = a.W[idx];
  }
  __syncthreads();
  float* X = Xs[wave];
  short* Xb = reinterpret_cast<short*>(Xs[wave]);
  const long P = a.P;
  const long NG = (P + 127) >> 7;
  const long nwaves = (long)gridDim.x * 4;
  const long w = (long)wave * gridDim.x + blockIdx.x;
  if (w >= NG) return;
  const float bv = a.bias[n];
  const float asc = a.rscale ? a.rscale[n] : 1.f;
  const float ash = a.rscale ? a.rshift[n] : 0.f;
  const unsigned To = (unsigned)a.Tout, Ti = (unsigned)a.Tin, aoff = Ti - To;
  const unsigned inv16 = 65536u / To + 1u;
  const __amdgpu_buffer_rsrc_t hout = rs_rsrc(a.out[0], P * 128);
  const __amdgpu_buffer_rsrc_t resr = rs_rsrc(a.res, (P / To) * Ti * 128);
  __amdgpu_buffer_rsrc_t srcr[NS];
  // (source 0 from its bf16 copy: the fp32 values would be rounded to bf16 by rs_put_f2b on the way into the MFMA -- the
  //  same round-to-nearest-even that produced the copy -- so the product is bit-identical and 0.32 GB per layer stay home)
  constexpr bool s0bf = SBF && S0BF;       // (a template parameter: as a run-time flag both paths cost 272 B of scratch)
#pragma unroll
  for (int s = 0; s < NS; ++s) srcr[s] = rs_rsrc(a.src[s], P * ((SBF && (s > 0 || s0bf)) ? 64 : 128));
  const unsigned lane_off = (unsigned)((lane >> 3) * 128 + (lane & 7) * 16);
  const unsigned lane_off_bf = (unsigned)((lane >> 2) * 64 + (lane & 3) * 16);

  // Ring of RS_R (block, source) tiles in flight per wave, 4 KB each; step i = 4*NS*run + NS*jb + s lives in
  // slot i % RS_R (static: RS_R divides 4*NS) and is refilled with step i + RS_R as soon as it sits in LDS.
  constexpr int RS_R = 4;
  float4 ring[RS_R][4];
  auto step_row0 = [&](long gi, int i) -> long { return gi * 128 + (i / NS) * 32; };   // i in [0, 4*NS)
  auto issue_step = [&](float4 (&dst)[4], long gi, int i) {
    if (SBF && ((i % NS) > 0 || s0bf)) rs_issue_block_bf(dst, srcr[i % NS], step_row0(gi, i), lane_off_bf);
    else rs_issue_block(dst, srcr[i % NS], step_row0(gi, i), lane_off);
  };
#pragma unroll
  for (int i = 0; i < RS_R; ++i) issue_step(ring[i], w, i);

  for (long gi = w; gi < NG; gi += nwaves) {
    const long gnext = (gi + nwaves < NG) ? gi + nwaves : gi;     // tail: harmless re-read of the own run
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const long m0 = gi * 128 + jb * 32;
      // residual rows of this block: (g, t) of the first row by one division, the 32 offsets by reciprocal
      const unsigned um0 = (unsigned)(m0 < P ? m0 : 0);
      const unsigned g0 = um0 / To, t0 = um0 - g0 * To;
      float rres[16];       // issued first: by the epilogue they are the OLDEST loads in flight (no drain)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned ii = (r & 3) + 8 * (r >> 2) + 4 * half;
        const unsigned x = t0 + ii, q = (x * inv16) >> 16;
        const unsigned rr = (g0 + q) * Ti + (x - q * To) + aoff;
        rres[r] = rs_load1(resr, rr * 128u + 4u * n);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int i = jb * NS + s;                 // step within the run
        const int slot = i % RS_R;
        if (SBF) { if (s > 0 || s0bf) rs_put_b2b(Xb, ring[slot], lane); else rs_put_f2b(Xb, ring[slot], lane); }
        else rs_put(X, ring[slot], lane);
        {
          const int in = i + RS_R;                 // the step this slot serves next
          issue_step(ring[slot], (in < 4 * NS) ? gi : gnext, in % (4 * NS));
        }
        if (SBF) {
          rs_v8s av[2];
          rs_get_b(Xb, av, lane);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const rs_v8s bw = *reinterpret_cast<const rs_v8s*>(&Wb[n * LWB + s * 32 + 16 * h + 8 * half]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rs_v8bf, av[h]),
                                                          __builtin_bit_cast(rs_v8bf, bw), acc, 0, 0, 0);
          }
        } else {
          float av[16];
          rs_get(X, av, lane);
#pragma unroll
          for (int t = 0; t < 16; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], Ws[(s * 32 + 16 * half + t) * RS_LDW + n], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from hoisting later segments' LDS traffic
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r] + bv;
        if (DROP) {
          const uint32_t h = mo_hash32(a.drop_seed, (uint32_t)(m * 32 + n));
          v = (h >= a.drop_thresh) ? v * a.drop_scale : 0.f;
        }
        v += rres[r] * asc + ash;
        rs_store_f32(hout, m * 32 + n, v);
        const float vm = (m < P) ? v : 0.f;
        s1 += vm; s2 += vm * vm;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    a.partial[gi * 64 + lane] = half ? s2 : s1;
  }
}

// data gradient: dsrc[s][p][:] = dm[p][:] @ W[:, 32s:32s+32], dm = dropout-masked dh
template <int NS, bool DROP, bool OBF>      // OBF: gradients of sources 1.. are bf16 tensors (source 0 fp32)
__global__ __launch_bounds__(256, 2) void rs_mlp_bwd_kernel(RsMlpArgs a) {
  // fp32: [k = co][n], row stride 32*NS+1.  OBF (bf16 MFMA): bf16 [n][k = co], row stride RS_LDXB
  __shared__ __attribute__((aligned(16))) float Ws[OBF ? 16 * NS * RS_LDXB : 32 * (NS * 32 + 1)];
  __shared__ __attribute__((aligned(16))) float Xs[4][OBF ? 16 * RS_LDXB : 32 * RS_LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  constexpr int NT = 32 * NS, LW = NT + 1;
  short* Wb = reinterpret_cast<short*>(Ws);
  for (int idx = tid; idx < 32 * NT; idx += 256) {
    const int co = idx / NT, c = idx - co * NT;
    if (OBF) { __bf16 t = (__bf16)a.W[idx]; Wb[c * RS_LDXB + co] = __builtin_bit_cast(short, t); }
    else Ws[co * LW + c] = a.W[idx];
  }
  __syncthreads();
  float* X = Xs[wave];
  short* Xb = reinterpret_cast<short*>(Xs[wave]);
  const long P = a.P;
  const long NG = (P + 127) >> 7;
  const long nwaves = (long)gridDim.x * 4;
  const long w = (long)wave * gridDim.x + blockIdx.x;
  if (w >= NG) return;
  const __amdgpu_buffer_rsrc_t srcr = rs_rsrc(a.src[0], P * 128);
  const unsigned lane_off = (unsigned)((lane >> 3) * 128 + (lane & 7) * 16);
  float4 ring[4][4];                              // the four blocks of a run in flight
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) rs_issue_block(ring[jb], srcr, w * 128 + jb * 32, lane_off);
  for (long gi = w; gi < NG; gi += nwaves) {
    const long gnext = (gi + nwaves < NG) ? gi + nwaves : gi;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const long m0 = gi * 128 + jb * 32;
      if (DROP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t e = (uint32_t)((m0 + 8 * j + (lane >> 3)) * 32 + 4 * (lane & 7));
          float* vp = &ring[jb][j].x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t h = mo_hash32(a.drop_seed, e + q);
            vp[q] = (h >= a.drop_thresh) ? vp[q] * a.drop_scale : 0.f;
          }
        }
      }
      if (OBF) rs_put_f2b(Xb, ring[jb], lane);
      else rs_put(X, ring[jb], lane);
      rs_issue_block(ring[jb], srcr, gnext * 128 + jb * 32, lane_off);
      float av[16];
      rs_v8s ab[2];
      if (OBF) rs_get_b(Xb, ab, lane);
      else rs_get(X, av, lane);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (OBF) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const rs_v8s bw = *reinterpret_cast<const rs_v8s*>(&Wb[(s * 32 + n) * RS_LDXB + 16 * h + 8 * half]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rs_v8bf, ab[h]),
                                                          __builtin_bit_cast(rs_v8bf, bw), acc, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 16; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], Ws[(16 * half + t) * LW + s * 32 + n], acc, 0, 0, 0);
        }
        if (OBF && s > 0) {
          const __amdgpu_buffer_rsrc_t o = rs_rsrc(a.out[s], P * 64);
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            rs_store_bf16_pair(o, m * 32, n, acc[r], acc[r + 1]);
          }
        } else {
          const __amdgpu_buffer_rsrc_t o = rs_rsrc(a.out[s], P * 128);
          const __amdgpu_buffer_rsrc_t ob = rs_rsrc(a.out_bf, a.out_bf ? P * 64 : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            rs_store_f32(o, m * 32 + n, acc[r]);
            if (s == NS - 1) rs_store_bf16(ob, m * 32 + n, acc[r]);     // num_records 0 when absent: dropped
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // one accumulator tile at a time (register budget)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Gated TCN (graph_wavenet.py:222-226) on the row-streaming scheme: the K taps of a 32-row output block are
// K row-mapped 32x32 tiles of h_prev (BatchNorm affine folded in front of the LDS trip), contracted with the
// packed weights into a filter and a gate tile; MODE 0 writes g = tanh(f) * sigmoid(g) (+ bf16 copy), MODE 1
// the pre-activation gradients dpre[p][0:32 | 32:64] from dg.  Unmapped / out-of-range rows are fetched with
// an out-of-range buffer offset and read as zero: no branches in the stream.
// ------------------------------------------------------------------------------------------------
#define RS_OOB 0xFFFFF000u

struct RsTcnArgs {
  const float* h_prev; const float* scale; const float* shift;   // [G*Tin][32], folded BatchNorm (may be null)
  const float* Wp;                  // packed [tau][co' (64)][ci (32)]
  const float* bf; const float* bg;
  const float* dg;                  // MODE 1: [G*Tout][32]
  float* out;                       // MODE 0: g [G*Tout][32]; MODE 1: dpre [G*Tout][64]
  unsigned short* out_bf;           // MODE 0: optional bf16 copy
  const float* dpre; const float* dres; float* du;   // data-gradient kernel
  long G; int Tin, Tout, dil;
  float* crop; int crop_tf;         // MODE 0: fp32 g of the LAST crop_tf steps only, compact [G*crop_tf][32] (out may be null)
};

// byte offsets of the four 16-byte loads of a row-mapped 32x32 tile: output rows m0 + 8j + (lane>>3) of a
// [.][To] row space map to source rows (g, t + shift) of a [.][Ti] space (RS_OOB when outside)
__device__ __forceinline__ void rs_tile_offsets(unsigned (&off)[4], long m0, long P, unsigned To, unsigned Ti,
                                                unsigned inv16, int shift, unsigned rowbytes, unsigned colbytes,
                                                int lane) {
  const unsigned um0 = (unsigned)(m0 < P ? m0 : 0);
  const unsigned g0 = um0 / To, t0 = um0 - g0 * To;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned o = 8 * j + (lane >> 3);
    const unsigned x = t0 + o, q = (x * inv16) >> 16;
    const int tt = (int)(x - q * To) + shift;
    const bool ok = ((unsigned)tt < Ti) & (m0 + o < P);
    off[j] = ok ? ((g0 + q) * Ti + (unsigned)tt) * rowbytes + colbytes : RS_OOB;
  }
}
__device__ __forceinline__ void rs_issue_tile(float4 (&v)[4], __amdgpu_buffer_rsrc_t r, const unsigned (&off)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = rs_load4(r, off[j]);
}

template <int K, int MODE, bool MF, bool CROP = false>     // MF: bf16 MFMA (throughput mode) instead of the exact fp32 MFMA;
                                                           // CROP (MODE 0): fp32 g of the last crop_tf steps only
__global__ __launch_bounds__(256, (K <= 2 && MODE == 0) ? 3 : 2) void rs_tcn_kernel(RsTcnArgs a) {
  constexpr int LW = 65;
  constexpr int LWB = 32 * K + 8;                    // MF: bf16 [co' (64)][tau*32 + ci] rows
  __shared__ __attribute__((aligned(16))) float Ws[MF ? 32 * LWB : K * 32 * LW];   // fp32: [tau*32 + ci][co']
  __shared__ __attribute__((aligned(16))) float Xs[4][MF ? 16 * RS_LDXB : 32 * RS_LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  short* Wb = reinterpret_cast<short*>(Ws);
  for (int idx = tid; idx < K * 64 * 32; idx += 256) {
    const int tau = idx / 2048, rem = idx - tau * 2048, cop = rem >> 5, ci = rem & 31;
    if (MF) { __bf16 t = (__bf16)a.Wp[idx]; Wb[cop * LWB + tau * 32 + ci] = __builtin_bit_cast(short, t); }
    else Ws[(tau * 32 + ci) * LW + cop] = a.Wp[idx];
  }
  __syncthreads();
  float* X = Xs[wave];
  short* Xb = reinterpret_cast<short*>(Xs[wave]);
  const unsigned To = (unsigned)a.Tout, Ti = (unsigned)a.Tin;
  const long P = a.G * To;
  const long NG = (P + 127) >> 7;
  const long nwaves = (long)gridDim.x * 4;
  const long w = (long)wave * gridDim.x + blockIdx.x;
  if (w >= NG) return;
  const unsigned inv16 = 65536u / To + 1u;
  const __amdgpu_buffer_rsrc_t hr = rs_rsrc(a.h_prev, a.G * Ti * 128);
  const __amdgpu_buffer_rsrc_t outr = rs_rsrc(a.out, a.out ? P * (MODE == 0 ? 128 : (MF ? 128 : 256)) : 0);
  const __amdgpu_buffer_rsrc_t cropr = rs_rsrc(a.crop, (MODE == 0 && CROP) ? a.G * a.crop_tf * 128 : 0);
  const __amdgpu_buffer_rsrc_t bfr = rs_rsrc(a.out_bf, (MODE == 0 && a.out_bf) ? P * 64 : 0);
  const __amdgpu_buffer_rsrc_t dgr = rs_rsrc(a.dg, MODE == 1 ? P * 128 : 0);
  const unsigned colb = (unsigned)((lane & 7) * 16);
  // folded BatchNorm affine of the lane's four columns
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.scale) {
    sc = *reinterpret_cast<const float4*>(a.scale + 4 * (lane & 7));
    sh = *reinterpret_cast<const float4*>(a.shift + 4 * (lane & 7));
  }
  const float bfv = a.bf[n], bgv = a.bg[n];

  constexpr int RS_R = 4, SPR = 4 * K;              // ring slots; steps per 128-row run
  float4 ring[RS_R][4];
  auto issue_step = [&](float4 (&dst)[4], long gi, int i) {   // step i of run gi: block i / K, tap i % K
    unsigned off[4];
    rs_tile_offsets(off, gi * 128 + (i / K) * 32, P, To, Ti, inv16, (i % K) * a.dil, 128u, colb, lane);
    rs_issue_tile(dst, hr, off);
  };
#pragma unroll
  for (int i = 0; i < RS_R; ++i) issue_step(ring[i], w, i);

  for (long gi = w; gi < NG; gi += nwaves) {
    const long gnext = (gi + nwaves < NG) ? gi + nwaves : gi;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const long m0 = gi * 128 + jb * 32;
      const unsigned cum0 = (unsigned)(m0 < P ? m0 : 0), cg0 = CROP ? cum0 / To : 0, ct0 = cum0 - cg0 * To;   // (crop store)
      float dgv[16];
      if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          dgv[r] = rs_load1(dgr, (unsigned)(m * 128 + 4 * n));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x16 accf, accg;
#pragma unroll
      for (int r = 0; r < 16; ++r) { accf[r] = 0.f; accg[r] = 0.f; }
#pragma unroll
      for (int tau = 0; tau < K; ++tau) {
        const int i = jb * K + tau, slot = i % RS_R;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4& v = ring[slot][j];
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        }
        if (MF) rs_put_f2b(Xb, ring[slot], lane);
        else rs_put(X, ring[slot], lane);
        {
          const int in = i + RS_R;
          issue_step(ring[slot], in < SPR ? gi : gnext, in % SPR);
        }
        if (MF) {
          rs_v8s ab[2];
          rs_get_b(Xb, ab, lane);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const rs_v8s wf = *reinterpret_cast<const rs_v8s*>(&Wb[n * LWB + tau * 32 + 16 * h + 8 * half]);
            const rs_v8s wg = *reinterpret_cast<const rs_v8s*>(&Wb[(32 + n) * LWB + tau * 32 + 16 * h + 8 * half]);
            accf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rs_v8bf, ab[h]),
                                                           __builtin_bit_cast(rs_v8bf, wf), accf, 0, 0, 0);
            accg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rs_v8bf, ab[h]),
                                                           __builtin_bit_cast(rs_v8bf, wg), accg, 0, 0, 0);
          }
        } else {
          float av[16];
          rs_get(X, av, lane);
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const float* wrow = &Ws[(tau * 32 + 16 * half + t) * LW];
            accf = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], wrow[n], accf, 0, 0, 0);
            accg = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], wrow[32 + n], accg, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;          // rows m (register r) and m + 1 (r + 1)
        const float f0 = mo_tanh(accf[r] + bfv), f1 = mo_tanh(accf[r + 1] + bfv);
        const float g0 = mo_sigmoid(accg[r] + bgv), g1 = mo_sigmoid(accg[r + 1] + bgv);
        if (MODE == 0) {
          if (!CROP) {
            rs_store_f32(outr, m * 32 + n, f0 * g0);
            rs_store_f32(outr, (m + 1) * 32 + n, f1 * g1);
          }
          rs_store_bf16_pair(bfr, m * 32, n, f0 * g0, f1 * g1);
          if (CROP) {
            // the skip path reads the last crop_tf steps of every (node, window) group in fp32: only those rows
            // ((group, step) of a row by the block's one division + the reciprocal, as rs_tile_offsets does)
            const unsigned o = (unsigned)((r & 3) + 8 * (r >> 2) + 4 * half);
            const unsigned x0 = ct0 + o, q0 = (x0 * inv16) >> 16, tt0 = x0 - q0 * To;
            const unsigned x1 = x0 + 1, q1 = (x1 * inv16) >> 16, tt1 = x1 - q1 * To;
            const unsigned tc = To - (unsigned)a.crop_tf;
            const long c0 = (m < P && tt0 >= tc) ? ((long)(cg0 + q0) * a.crop_tf + (tt0 - tc)) * 32 + n : 0x3FFFFFFFL;
            const long c1 = (m + 1 < P && tt1 >= tc) ? ((long)(cg0 + q1) * a.crop_tf + (tt1 - tc)) * 32 + n : 0x3FFFFFFFL;
            rs_store_f32(cropr, c0, f0 * g0);                           // (0x3FFFFFFF * 4: outside every descriptor)
            rs_store_f32(cropr, c1, f1 * g1);
          }
        } else if (MF) {        // throughput mode: the pre-activation gradients are a bf16 [P][64] tensor
          rs_store_bf16_pair(outr, m * 64, n, dgv[r] * g0 * (1.f - f0 * f0), dgv[r + 1] * g1 * (1.f - f1 * f1), 64);
          rs_store_bf16_pair(outr, m * 64 + 32, n, dgv[r] * f0 * g0 * (1.f - g0), dgv[r + 1] * f1 * g1 * (1.f - g1), 64);
        } else {
          rs_store_f32(outr, m * 64 + n, dgv[r] * g0 * (1.f - f0 * f0));
          rs_store_f32(outr, m * 64 + 32 + n, dgv[r] * f0 * g0 * (1.f - g0));
          rs_store_f32(outr, (m + 1) * 64 + n, dgv[r + 1] * g1 * (1.f - f1 * f1));
          rs_store_f32(outr, (m + 1) * 64 + 32 + n, dgv[r + 1] * f1 * g1 * (1.f - g1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// data gradient of the gated TCN: du[(g,t)][ci] = sum_tau dpre[(g, t - tau*d)][0:64] @ Wp[tau] (+ dres[(g, t - (Tin-Tout))])
template <int K, bool MF>
__global__ __launch_bounds__(256, 2) void rs_tcn_du_kernel(RsTcnArgs a) {
  constexpr int LWB = 64 * K + 8;                   // MF: bf16 [ci (32)][tau*64 + co'] rows
  __shared__ __attribute__((aligned(16))) float Ws[MF ? 16 * LWB : K * 64 * RS_LDW];   // fp32: [tau*64 + co'][ci]
  __shared__ __attribute__((aligned(16))) float Xs[4][MF ? 16 * RS_LDXB : 32 * RS_LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  short* Wb = reinterpret_cast<short*>(Ws);
  for (int idx = tid; idx < K * 64 * 32; idx += 256) {
    if (MF) { __bf16 t = (__bf16)a.Wp[idx]; Wb[(idx & 31) * LWB + (idx >> 5)] = __builtin_bit_cast(short, t); }
    else Ws[(idx >> 5) * RS_LDW + (idx & 31)] = a.Wp[idx];
  }
  __syncthreads();
  float* X = Xs[wave];
  short* Xb = reinterpret_cast<short*>(Xs[wave]);
  const unsigned To = (unsigned)a.Tout, Ti = (unsigned)a.Tin;
  const long P = a.G * Ti;                          // rows of du
  const long NG = (P + 127) >> 7;
  const long nwaves = (long)gridDim.x * 4;
  const long w = (long)wave * gridDim.x + blockIdx.x;
  if (w >= NG) return;
  const unsigned inv16 = 65536u / Ti + 1u;
  const __amdgpu_buffer_rsrc_t dpr = rs_rsrc(a.dpre, a.G * To * (MF ? 128 : 256));
  const __amdgpu_buffer_rsrc_t dur = rs_rsrc(a.du, P * 128);
  const __amdgpu_buffer_rsrc_t drr = rs_rsrc(a.dres, a.dres ? a.G * To * 128 : 0);
  const unsigned colb = (unsigned)((lane & 7) * 16);

  constexpr int RS_R = 4, SPB = 2 * K, SPR = 4 * SPB;   // steps: (tap, half) tiles of [32 rows][32 of 64 columns]
  float4 ring[RS_R][4];
  auto issue_step = [&](float4 (&dst)[4], long gi, int i) {
    const int st = i % SPB, tau = st >> 1, hh = st & 1;
    if (MF) {     // bf16 [.][64] rows: two 16-byte loads per tile, rows 16j + (lane>>2), eight columns from 8*(lane&3)
      const long m0 = gi * 128 + (i / SPB) * 32;
      const unsigned um0 = (unsigned)(m0 < P ? m0 : 0);
      const unsigned g0 = um0 / Ti, t0 = um0 - g0 * Ti;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned o = 16 * j + (lane >> 2);
        const unsigned x = t0 + o, q = (x * inv16) >> 16;
        const int tt = (int)(x - q * Ti) - tau * a.dil;
        const bool ok = ((unsigned)tt < To) & (m0 + o < P);
        dst[j] = rs_load4(dpr, ok ? ((g0 + q) * To + (unsigned)tt) * 128u + 64u * hh + 16u * (lane & 3) : RS_OOB);
      }
    } else {
      unsigned off[4];
      rs_tile_offsets(off, gi * 128 + (i / SPB) * 32, P, Ti, To, inv16, -tau * a.dil, 256u, colb + 128u * hh, lane);
      rs_issue_tile(dst, dpr, off);
    }
  };
#pragma unroll
  for (int i = 0; i < RS_R; ++i) issue_step(ring[i], w, i);

  for (long gi = w; gi < NG; gi += nwaves) {
    const long gnext = (gi + nwaves < NG) ? gi + nwaves : gi;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const long m0 = gi * 128 + jb * 32;
      // residual rows (g, t - (Tin - Tout)) of the [G*Tout][32] gradient, in the accumulator layout
      const unsigned um0 = (unsigned)(m0 < P ? m0 : 0);
      const unsigned g0 = um0 / Ti, t0 = um0 - g0 * Ti;
      float rres[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned ii = (r & 3) + 8 * (r >> 2) + 4 * half;
        const unsigned x = t0 + ii, q = (x * inv16) >> 16;
        const int tt = (int)(x - q * Ti) - (int)(Ti - To);
        const bool ok = (tt >= 0) & (m0 + ii < P);
        rres[r] = rs_load1(drr, ok ? ((g0 + q) * To + (unsigned)tt) * 128u + 4u * n : RS_OOB);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int st = 0; st < SPB; ++st) {
        const int i = jb * SPB + st, slot = i % RS_R;
        if (MF) rs_put_b2b(Xb, ring[slot], lane);
        else rs_put(X, ring[slot], lane);
        {
          const int in = i + RS_R;
          issue_step(ring[slot], in < SPR ? gi : gnext, in % SPR);
        }
        if (MF) {
          rs_v8s ab[2];
          rs_get_b(Xb, ab, lane);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const rs_v8s bw = *reinterpret_cast<const rs_v8s*>(
                &Wb[n * LWB + (st >> 1) * 64 + (st & 1) * 32 + 16 * h + 8 * half]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rs_v8bf, ab[h]),
                                                          __builtin_bit_cast(rs_v8bf, bw), acc, 0, 0, 0);
          }
        } else {
          float av[16];
          rs_get(X, av, lane);
#pragma unroll
          for (int t = 0; t < 16; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], Ws[((st >> 1) * 64 + (st & 1) * 32 + 16 * half + t) * RS_LDW + n],
                                                       acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        rs_store_f32(dur, m * 32 + n, acc[r] + rres[r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients on the bf16 MFMA (throughput mode).  The contraction index is the ROW, so the operands are
// needed k-major: every 32-row tile (coalesced 16-byte loads, fp32 tiles rounded to bf16 on the way) goes through
// the wave's LDS tile and comes back TRANSPOSED by ds_read_b64_tr_b16 -- two MFMAs per 32 rows and 32x32 block
// instead of sixteen fp32 ones, and eight times fewer load instructions than the dword-per-lane fp32 form.
// Same slab / column-sum contract as rs_wgrad_kernel.
// ------------------------------------------------------------------------------------------------
typedef short rs_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) rs_v4s* rs_lds_v4s_ptr;
// f[s]: lane (column lane&31) gets rows k = 16s + 8*(lane>>5) + 0..7 of the [32][32] bf16 tile
__device__ __forceinline__ void rs_get_tr(const short* Xb, rs_v8s (&f)[2], int lane) {
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kr = 16 * s + 8 * (tg >> 1) + tq, nc = 16 * (tg & 1) + 4 * tp;
    const rs_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rs_lds_v4s_ptr)&Xb[kr * RS_LDXB + nc]);
    const rs_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rs_lds_v4s_ptr)&Xb[(kr + 4) * RS_LDXB + nc]);
    f[s] = (rs_v8s){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

template <int MA, int NB, bool MAPPED, bool BBF, bool ABF = false, bool B0BF = false>    // ABF: a is stored as bf16 (the
                                                          // TCN's dpre); B0BF: segment 0 of b from its bf16 copy too
__global__ __launch_bounds__(256, 2) void rs_wgrad_bf_kernel(MoOperand A, MoOperand B, float* __restrict__ slab,
                                                            float* __restrict__ cs, long P, int post_b) {
  extern __shared__ float rs_sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, half = lane >> 5;
  short* Xb = reinterpret_cast<short*>(rs_sm) + wave * (32 * RS_LDXB);
  const long NG = (P + 127) >> 7;
  const long nwaves = (long)gridDim.x * 4;
  const long w = (long)wave * gridDim.x + blockIdx.x;
  constexpr int LDA = 32 * MA, S = MA + NB, SPR = 4 * S, RS_R = 4;

  f32x16 acc[MA][NB];
#pragma unroll
  for (int i = 0; i < MA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 csum[MA];
  float csum8[ABF ? MA : 1][8];        // ABF: a lane holds eight columns of its rows
#pragma unroll
  for (int i = 0; i < MA; ++i) csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < (ABF ? MA : 1); ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) csum8[i][q] = 0.f;

  const unsigned To = MAPPED ? (unsigned)B.seg[0].To : 1u;
  const unsigned inv16 = 65536u / To + 1u;
  constexpr bool b0bf = BBF && !MAPPED && B0BF;        // segment 0 from its bf16 copy too (bit-identical: see rs_mlp_fwd_kernel)
  const __amdgpu_buffer_rsrc_t ar = rs_rsrc(A.seg[0].ptr, P * LDA * (ABF ? 2 : 4));
  __amdgpu_buffer_rsrc_t br[NB];
  float4 bsc[NB], bsh[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const long rows = MAPPED ? (P / To) * B.seg[j].Ti : P;
    br[j] = rs_rsrc(B.seg[j].ptr, rows * ((BBF && (j > 0 || b0bf)) ? 64 : 128));
    bsc[j] = make_float4(1.f, 1.f, 1.f, 1.f); bsh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (B.seg[j].scale) {
      bsc[j] = *reinterpret_cast<const float4*>(B.seg[j].scale + 4 * (lane & 7));
      bsh[j] = *reinterpret_cast<const float4*>(B.seg[j].shift + 4 * (lane & 7));
    }
  }
  const unsigned lane_off = (unsigned)((lane >> 3) * 128 + (lane & 7) * 16);
  const unsigned lane_off_bf = (unsigned)((lane >> 2) * 64 + (lane & 3) * 16);
  const unsigned lane_off_a = (unsigned)((lane >> 3) * (LDA * 4) + (lane & 7) * 16);
  const uint32_t dseed = A.seg[0].drop_seed, dthresh = A.seg[0].drop_thresh;
  const float dscale = A.seg[0].drop_scale;

  float4 ring[RS_R][4];
  auto issue_step = [&](float4 (&dst)[4], long gi, int i) {      // step i of run gi: block i / S, operand tile i % S
    const long row0 = gi * 128 + (i / S) * 32;
    const int t = i % S;
    if (t < MA) {
      if (ABF) {
        const unsigned base = (unsigned)(row0 * (LDA * 2)) + (unsigned)((lane >> 2) * (LDA * 2) + (lane & 3) * 16) + 64u * t;
#pragma unroll
        for (int j = 0; j < 2; ++j) dst[j] = rs_load4(ar, base + (unsigned)(16 * LDA * 2) * j);
      } else {
        const unsigned base = (unsigned)(row0 * (LDA * 4)) + lane_off_a + 128u * t;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = rs_load4(ar, base + (unsigned)(8 * LDA * 4) * j);
      }
    } else {
      const int j = t - MA;
      if (MAPPED) {
        unsigned off[4];
        rs_tile_offsets(off, row0, P, To, (unsigned)B.seg[j].Ti, inv16, B.seg[j].off, 128u,
                        (unsigned)((lane & 7) * 16), lane);
        rs_issue_tile(dst, br[j], off);
      } else if (BBF && (j > 0 || b0bf)) {
        rs_issue_block_bf(dst, br[j], row0, lane_off_bf);
      } else {
        rs_issue_block(dst, br[j], row0, lane_off);
      }
    }
  };

  if (w < NG) {
#pragma unroll
    for (int i = 0; i < RS_R; ++i) issue_step(ring[i], w, i);
    for (long gi = w; gi < NG; gi += nwaves) {
      const long gnext = (gi + nwaves < NG) ? gi + nwaves : gi;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const long m0 = gi * 128 + jb * 32;
        rs_v8s af[MA][2];
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) {
          const int i = jb * S + ma, slot = i % RS_R;
          if (ABF) {
            // bf16 a tile: straight copy into the LDS tile; column sums from the widened values (inline-asm adds
            // behind the LDS stores, see below)
            rs_put_b2b(Xb, ring[slot], lane);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const unsigned u[4] = {__float_as_uint(ring[slot][j].x), __float_as_uint(ring[slot][j].y),
                                     __float_as_uint(ring[slot][j].z), __float_as_uint(ring[slot][j].w)};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float lo = __uint_as_float(u[q] << 16), hi = __uint_as_float(u[q] & 0xffff0000u);
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum8[ABF ? ma : 0][2 * q]) : "v"(lo) : "memory");
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum8[ABF ? ma : 0][2 * q + 1]) : "v"(hi) : "memory");
              }
            }
          } else {
          // a tile: regenerated dropout mask, column sums (the bias gradient) from the fp32 values
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* vp = &ring[slot][j].x;
              if (dthresh) {
                const uint32_t e = (uint32_t)((m0 + 8 * j + (lane >> 3)) * LDA + ma * 32 + 4 * (lane & 7));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const uint32_t h = mo_hash32(dseed, e + q);
                  vp[q] = (h >= dthresh) ? vp[q] * dscale : 0.f;
                }
              }
            }
            rs_put_f2b(Xb, ring[slot], lane);
            // Column sums on the VALU through inline asm: written as plain `csum += v` the compiler re-associated the
            // sums into many partials (~60 VGPRs in this kernel).  hipcc does NOT insert the vmcnt wait for a load
            // result that only an inline-asm operand consumes, so the adds sit BEHIND the tile's LDS stores (the
            // "memory" clobber keeps them there): the conversions feeding those stores have already waited.
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum[ma].x) : "v"(ring[slot][j].x) : "memory");
              asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum[ma].y) : "v"(ring[slot][j].y) : "memory");
              asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum[ma].z) : "v"(ring[slot][j].z) : "memory");
              asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum[ma].w) : "v"(ring[slot][j].w) : "memory");
            }
          }
          issue_step(ring[slot], (i + RS_R) < SPR ? gi : gnext, (i + RS_R) % SPR);
          rs_get_tr(Xb, af[ma], lane);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int i = jb * S + MA + j, slot = i % RS_R;
          if (BBF && (j > 0 || b0bf)) {
            rs_put_b2b(Xb, ring[slot], lane);
          } else {
            if (post_b) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float4& v = ring[slot][q];
                v.x = v.x * bsc[j].x + bsh[j].x; v.y = v.y * bsc[j].y + bsh[j].y;
                v.z = v.z * bsc[j].z + bsh[j].z; v.w = v.w * bsc[j].w + bsh[j].w;
                if (post_b & 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              }
            }
            rs_put_f2b(Xb, ring[slot], lane);
          }
          issue_step(ring[slot], (i + RS_R) < SPR ? gi : gnext, (i + RS_R) % SPR);
          rs_v8s bfr[2];
          rs_get_tr(Xb, bfr, lane);
#pragma unroll
          for (int ma = 0; ma < MA; ++ma)
#pragma unroll
            for (int s = 0; s < 2; ++s)
              acc[ma][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rs_v8bf, af[ma][s]),
                                                                   __builtin_bit_cast(rs_v8bf, bfr[s]), acc[ma][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  // ---- workgroup reduction of the four waves' partial tiles through LDS, then one slab per workgroup
  __syncthreads();                                   // the wave tiles at the start of rs_sm are done with
  constexpr int TILE = MA * NB * 16 * 64;
  float* buf0 = rs_sm;
  float* buf1 = rs_sm + TILE;
  auto put = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((i * NB + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * NB + j) * 16 + r) * 64 + lane];
  };
  if (wave == 2) put(buf0);
  if (wave == 3) put(buf1);
  __syncthreads();
  if (wave == 0) add(buf0);
  if (wave == 1) add(buf1);
  __syncthreads();
  if (wave == 1) put(buf0);
  __syncthreads();
  if (wave == 0) {
    add(buf0);
    float* out = slab + (long)blockIdx.x * (32 * MA) * (32 * NB);
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          out[(long)m * (32 * NB) + j * 32 + c] = acc[i][j][r];
        }
  }
  if (cs) {
    // column sums: lane holds columns 4*(lane&7)..+3 summed over its rows; fold the 8 row groups and 4 waves
    __syncthreads();
    if (ABF) {
      float* cb = rs_sm;                                          // [wave][MA][64 lanes][8]
#pragma unroll
      for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) cb[((wave * MA + i) * 64 + lane) * 8 + q] = csum8[ABF ? i : 0][q];
      __syncthreads();
      if (tid < 32 * MA) {
        const int i = tid >> 5, col = tid & 31;
        float s = 0.f;
        for (int q = 0; q < 4; ++q)
          for (int gq = 0; gq < 16; ++gq) s += cb[(((q * MA + i) * 64) + gq * 4 + (col >> 3)) * 8 + (col & 7)];
        cs[(long)blockIdx.x * (32 * MA) + tid] = s;
      }
      return;
    }
    float4* cbuf = reinterpret_cast<float4*>(rs_sm);              // [wave][MA][64]
#pragma unroll
    for (int i = 0; i < MA; ++i) cbuf[(wave * MA + i) * 64 + lane] = csum[i];
    __syncthreads();
    if (tid < 32 * MA) {
      const int i = tid >> 5, col = tid & 31;
      float s = 0.f;
      for (int q = 0; q < 4; ++q)
        for (int gq = 0; gq < 8; ++gq) {
          const float4 v = cbuf[(q * MA + i) * 64 + gq * 8 + (col >> 2)];
          s += (col & 3) == 0 ? v.x : (col & 3) == 1 ? v.y : (col & 3) == 2 ? v.z : v.w;
        }
      cs[(long)blockIdx.x * (32 * MA) + tid] = s;
    }
  }
}
