// Small-graph Graph WaveNet body for gfx950: the whole stack of layers (gated TCN -> diffusion hops -> mlp + dropout +
// residual -> BatchNorm statistics, graph_wavenet.py:214-250) of ONE forward call in ONE workgroup, forward and backward.
//
// Why: inside Modified_UNET the Graph WaveNet runs on the 67-county graph, one call per batch element
// (unet.py:221), each call a batch of ONE window with kernel_size 1: an activation is N*T*32 floats = 17 KB (H = 2) ..
// 60 KB (H = 7).  The general engine spends ~25 launches of 3..20 us per layer on it (300 per call, 1.7 ms of GPU time
// and 3 ms of launch-thread time per window, profiles/r02_unet_c3_kernel_stats.csv) -- the work itself is 37 MFLOP.
// Here a call is one workgroup of 16 waves: every contraction of a layer is cut into 16x16 tiles that the waves take in
// turn (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), the weights of the layer and the dense supports
// sit in LDS, activations live in the saved-for-backward tensors (L1/L2 resident, written once), phases are separated
// by workgroup barriers.  All B calls of a step are one launch (grid = B); BatchNorm statistics are per call -- exactly
// the (1, N, T) statistics of the reference's per-call BatchNorm2d (unet.py:221 -> graph_wavenet.py:250).
//
// Layout: the engine's "nbtc" rows r = (n*B + b)*T + t of 32 channels (include/mo_hip.h), so the start conv, the skip
// contraction, the head and the boundary transposes of the general engine are used unchanged around this kernel.
// Supports: the identity (the reference's default static support, graph_wavenet.py:13-32) is folded into the mlp
// weights of the layer (x1 = x2 = g); every other support, static or adaptive, is a dense N x N matrix in LDS.
#include "mo_common.h"
#include "../../include/mo_hip.h"

typedef float sg_f4 __attribute__((ext_vector_type(4)));
#define SG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SG_MAXL 16
#define SG_MAXS 3
#define SG_THREADS 768     // 12 waves: >= the 9..10 row strips of a phase at N*T = 134, and 170 VGPRs per lane (1024 threads: 128, spills)
#define SG_WAVES 12
#define SG_NR (SG_THREADS / 32)   // rows per sweep of the elementwise phases
#define SG_LDW 66          // LDS row stride of a [k][64] weight image: % 4 == 2 -> the four k-groups of a wave hit disjoint banks
#define SG_LDM 34          // ... of a [k][32] image

struct SgLayer {
  const float *Wf, *bf, *Wg, *bg;     // (32,32), (32)
  const float *Wm, *bm;               // (32, 32*(1+2S)), (32): reference column order [g, A1 g, A1^2 g, A2 g, ...]
  const float *gamma, *beta;          // BatchNorm affine
  float *rmean, *rvar;                // running statistics (eval mode reads them; training: sg_running_kernel)
};
struct SgArgs {
  int B, N, T, L, P;                  // calls, nodes, steps, layers, P = N*T positions per call
  int nsup, ndense;                   // supports in reference order; ndense of them are dense matrices
  int dense_of[SG_MAXS];              // support k -> index of its dense matrix, or -1 (identity)
  const float* adj[SG_MAXS];          // dense supports A_d [N][N] row-major (nconv applies A^T on the node axis)
  int adaptive_dense;                 // index of the adaptive support among the dense ones (its gradient is wanted), or -1
  SgLayer ly[SG_MAXL];
  const float* h0;                    // [rows][32] start-conv output
  float* gcat;                        // [rows][32 L]: gated TCN outputs of all layers side by side (the skip contraction's operand)
  float* hs;                          // [L][rows][32] pre-BatchNorm layer outputs
  float* xs;                          // [L][ndense][2][rows][32] diffusion hops x1, x2
  float* stats;                       // [B][L][6][32]: scale, shift, mean, rstd, unbiased variance, (unused)
  int training; float eps;
  uint32_t seed; uint32_t thresh; float dscale;     // dropout (seed of layer i: seed + 7919 i), 0 = off
  // backward only
  const float* dgskip;                // [rows][32 L]: d loss / d g_i through the skip path
  float* dxo;                         // [rows][32] gradient w.r.t. a layer's output (in: none; out: gradient w.r.t. h0)
  float* dh;                          // [rows][32] scratch: gradient w.r.t. the pre-BatchNorm output
  float* dg;                          // [rows][32] scratch
  float* dxs;                         // [ndense][2][rows][32] scratch: gradients of the hops
  float* dpre;                        // [rows][64] scratch: gradients of the TCN pre-activations
  float* slab;                        // [B][L][SG_SLAB]: per-call parameter gradients
  float* dA;                          // [B][N][N] per-call gradient of the adaptive support (or null)
  long rows;                          // N*B*T
};
// per-layer slab layout (floats)
#define SG_S_WF 0
#define SG_S_BF 1024
#define SG_S_WG 1056
#define SG_S_BG 2080
#define SG_S_BM 2112
#define SG_S_GA 2144
#define SG_S_BE 2176
#define SG_S_WM 2208                 // 32 * 32 * (1 + 2 * nsup) follow
static inline long sg_slab_floats(int nsup) { return SG_S_WM + 32L * 32 * (1 + 2 * nsup); }

// sigmoid / tanh on the hardware exp and reciprocal, as the general engine's gate epilogues (mo_gemm.hpp mo_sigmoid /
// mo_tanh: ~1e-6 relative; the libm tanhf is an order of magnitude more VALU instructions -- it was most of phase A)
__device__ __forceinline__ float sg_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float sg_tanh(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// One 16x16 output tile: acc += A_op[16][32 kc .. 32 kc + 31] * B_op[..][16] for kc in [0, kchunks); the loaders fill
// a[j] = A_op[m0 + lane%16][32 kc + 8 q + j], b[j] = B_op[32 kc + 8 q + j][n0 + lane%16] (q = lane / 16): the MFMA's four
// k slots of a step are the four q groups, so a lane's eight k are CONTIGUOUS in memory for a k-major operand.
template <class LA, class LB>
__device__ __forceinline__ sg_f4 sg_tile(int kchunks, LA la, LB lb) {
  sg_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < kchunks; ++kc) {
    float a[8], b[8];
    la(kc, a); lb(kc, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = SG_MFMA(a[j], b[j], acc);
  }
  return acc;
}

struct SgGeo {
  int B, T, P, N, b;
  __device__ __forceinline__ long row(int pl) const { const int n = pl / T; return (long)pl + (long)n * (B - 1) * T + (long)b * T; }
  __device__ __forceinline__ long nrow(int n, int t) const { return ((long)n * B + b) * T + t; }
};

// Strided operand rows through buffer descriptors: ONE 32-bit lane offset per job plus a wave-uniform SGPR offset per
// (k chunk, k step) -- 48 pointer-based loads in flight cost 96 VGPRs of addresses and the kernel spilled.  Rows past the
// tensor (nodes >= N) fall out of the descriptor's range and read 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_rsrc(const float* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float sg_bl(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void sg_ld8(const float* p, float* a) {
  const float4 u = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(p + 4);
  a[0] = u.x; a[1] = u.y; a[2] = u.z; a[3] = u.w; a[4] = v.x; a[5] = v.y; a[6] = v.z; a[7] = v.w;
}

// ------------------------------------------------------------------------------------------------ forward
// dynamic LDS: adj [ndense][N][LDA] | wt [32][SG_LDW] (filter|gate, k-major) | wm [32 ne][SG_LDM] (k-major) | biases etc.
// A wave's job is a strip of 16 rows with ALL its output columns (the A operand is loaded once) and every global load of a
// job is issued before the first MFMA: a phase costs one L2 round trip, not one per 32-wide k chunk.
// LR (LDS-resident): when N*T is small enough (134 positions at H = 2: 19 KB per [P][32] tensor) the layer input, the gated
// output and the hops are MIRRORED in LDS and every read of a tensor produced inside the kernel comes from there -- a
// dependent access through L2 costs ~2 us on this chip and a layer has five of them; the global copies (saved for
// backward) are write-only here.
#define SG_LDP 36          // LDS row stride of a mirrored [P][32] tensor (144 B: 16-byte aligned rows, bank-spread)
template <int ND, bool LR>
__global__ __launch_bounds__(SG_THREADS) void sg_fwd_kernel(SgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NE = 1 + 2 * ND;
  // LR: every tensor a later phase reads is an LDS mirror, so a phase boundary only has to order LDS traffic -- __syncthreads
  // also waits for the phase's global stores (the saved-for-backward copies; ~2 us each, five phases per layer)
#define SG_SYNC() do { if (LR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads(); } while (0)
  const int tid = threadIdx.x, lane = tid & 63, wave_ = tid >> 6, l16 = lane & 15, q = lane >> 4;
  const int N = a.N, T = a.T, P = a.P, LDA = N + 3 - ((N + 1) & 3);
  // LDA % 4 == 2 (bank spread of the four k groups): N + 3 - ((N + 1) & 3) is the smallest value >= N that is 2 (mod 4)
  float* adj_s = lds;
  float* wt = adj_s + ND * N * LDA;
  float* wm = wt + 32 * SG_LDW;
  float* bias = wm + 32 * NE * SG_LDM;          // bf[32] bg[32] bm[32]
  float* aff = bias + 96;                        // scale[32] shift[32] of the layer input
  float* part = aff + 64;                        // [SG_WAVES][64] BatchNorm partial sums
  float* mir = part + SG_WAVES * 64;             // LR: XL[2] | GL | X1L[ND] X2L[ND] ...  each [P][SG_LDP]
  const int MS = a.P * SG_LDP;
  SgGeo g; g.B = a.B; g.T = T; g.P = P; g.N = N; g.b = blockIdx.x;
  const long rows = a.rows;
  const int ldg = 32 * a.L;
  for (int d = 0; d < ND; ++d)
    for (int i = tid; i < N * N; i += SG_THREADS) adj_s[(d * N + i / N) * LDA + i % N] = a.adj[d][i];
  const int mtiles = (P + 15) >> 4;
  const int ntn = (N + 15) >> 4, kch = (N + 31) >> 5;     // N <= 80: kch <= 3
  // the epilogue rows of this wave's strip (one strip per wave when there are <= 12 of them): the position -> row
  // arithmetic (an integer division per row) once per kernel instead of in every epilogue -- the phases are bound by the
  // VALU issue of ONE compute unit, and that arithmetic was a third of an epilogue (tools/sg_timing.py)
  const bool one_strip = mtiles <= SG_WAVES;
  long erow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) erow[i] = g.row(min(wave_ * 16 + 4 * q + i, P - 1));
  if (LR) {
    for (int i = tid; i < P * 8; i += SG_THREADS) {       // the call's rows of h0 -> XL[0]
      const int pl = i >> 3, c4 = (i & 7) * 4;
      *reinterpret_cast<float4*>(mir + pl * SG_LDP + c4) = *reinterpret_cast<const float4*>(a.h0 + g.row(pl) * 32 + c4);
    }
  }

  for (int li = 0; li < a.L; ++li) {
    const SgLayer& Ly = a.ly[li];
    // (the job -> row arithmetic below depends on the wave index only; laundering it keeps the compiler from hoisting two
    //  dozen 64-bit row addresses out of the layer loop and holding them in VGPRs across every phase -- it spilled)
    int wave = wave_; asm volatile("" : "+v"(wave));
#ifdef SG_TIMING
    long long tt0 = wall_clock64(); long long cc0 = clock64(); float* tst = a.stats + ((long)blockIdx.x * a.L + li) * 6 * 32 + 160; int tk = 0;
#define SG_T() do { if (tid == 0) tst[tk] = (float)(wall_clock64() - tt0); ++tk; } while (0)
#else
#define SG_T() do {} while (0)
#endif
    // ---- weights of the layer into LDS (k-major images); identity supports folded into the g block of the mlp
    for (int i = tid; i < 32 * 32; i += SG_THREADS) {
      const int co = i >> 5, ci = i & 31;
      wt[ci * SG_LDW + co] = Ly.Wf[i];
      wt[ci * SG_LDW + 32 + co] = Ly.Wg[i];
      const float* wrow = Ly.Wm + (long)co * 32 * (1 + 2 * a.nsup);
      float w0 = wrow[ci];
      for (int k = 0; k < a.nsup; ++k) {
        const int d = a.dense_of[k];
        const float w1 = wrow[32 * (1 + 2 * k) + ci], w2 = wrow[32 * (2 + 2 * k) + ci];
        if (d < 0) w0 += w1 + w2;
        else { wm[(32 * (1 + 2 * d) + ci) * SG_LDM + co] = w1; wm[(32 * (2 + 2 * d) + ci) * SG_LDM + co] = w2; }
      }
      wm[ci * SG_LDM + co] = w0;
    }
    if (tid < 32) { bias[tid] = Ly.bf[tid]; bias[32 + tid] = Ly.bg[tid]; bias[64 + tid] = Ly.bm[tid]; }
    if (li == 0 && tid < 32) { aff[tid] = 1.f; aff[32 + tid] = 0.f; }
    SG_SYNC();
    SG_T();
    const float* xin = li == 0 ? a.h0 : a.hs + (long)(li - 1) * rows * 32;
    float* gout = a.gcat + li * 32;
    float* XLc = mir + (li & 1) * MS;             // LR mirrors: this layer's input, the next layer's input, g, hops
    float* XLn = mir + ((li & 1) ^ 1) * MS;
    float* GL = mir + 2 * MS;

    // ---- A: gated TCN (kernel_size 1: two 1x1 convs)  g = tanh(Wf x + bf) * sigmoid(Wg x + bg)
    for (int mt = wave; mt < mtiles; mt += SG_WAVES) {
#ifdef SG_TIMING
      long long ca = clock64();
#endif
      const int pl = min(mt * 16 + l16, P - 1);
      float av[8];
      sg_ld8(LR ? XLc + pl * SG_LDP + 8 * q : xin + g.row(pl) * 32 + 8 * q, av);
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = av[j] * aff[8 * q + j] + aff[32 + 8 * q + j];
#ifdef SG_TIMING
      asm volatile("" :: "v"(av[0]), "v"(av[7]));
      long long cb = clock64();
#endif
      sg_f4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* wr = wt + (8 * q + j) * SG_LDW + l16;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = SG_MFMA(av[j], wr[16 * u], acc[u]);
      }
#ifdef SG_TIMING
      asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][3]));
      long long cc = clock64();
      if (tid == 0) { tst[10] = (float)(cb - ca); tst[11] = (float)(cc - cb); tst[14] = (float)(ca - cc0); }
#endif
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const int c = nh * 16 + l16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int po = mt * 16 + 4 * q + i;
          if (po < P) {
            const float gv = sg_tanh(acc[nh][i] + bias[c]) * sg_sigmoid(acc[2 + nh][i] + bias[32 + c]);
            gout[(one_strip ? erow[i] : g.row(po)) * ldg + c] = gv;
            if (LR) GL[po * SG_LDP + c] = gv;
          }
        }
      }
#ifdef SG_TIMING
      if (tid == 0) tst[12] = (float)(clock64() - cc);
#endif
    }
#ifdef SG_TIMING
    { long long cw = clock64(); SG_SYNC(); if (tid == 0) tst[13] = (float)(clock64() - cw); }
#else
    SG_SYNC();
#endif
    SG_T();

    // ---- B: diffusion hops of the dense supports: x1 = A^T g, x2 = A^T x1 on the node axis ([N][T*32] matrices);
    //      a job = 16 nodes x the 32 channels of one time step
    for (int hop = 0; hop < 2 && ND > 0; ++hop) {
      for (int job = wave; job < ND * ntn * T; job += SG_WAVES) {
        const int d = job / (ntn * T), r2 = job - d * (ntn * T), mt = r2 / T, tt = r2 - mt * T;
        const float* As = adj_s + d * N * LDA;
        float* x1 = a.xs + (((long)li * ND + d) * 2) * rows * 32;
        const float* src = hop == 0 ? gout : x1;
        const int lds_ = hop == 0 ? ldg : 32;
        float* dst = hop == 0 ? x1 : x1 + rows * 32;
        const int wcl = min(mt * 16 + l16, N - 1);
        const __amdgpu_buffer_rsrc_t rs = sg_rsrc(src, (rows * lds_ - (hop == 0 ? li * 32 : 0)) * 4);
        const int voff = (int)((g.nrow(8 * q, tt) * lds_ + l16) * 4), sstep = a.B * T * lds_ * 4;
        float bv[3][2][8];
        const float* ML = hop == 0 ? GL : mir + (3 + 2 * d) * MS;       // LR: the hop's input mirror
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
          if (kc < kch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (LR) {
                const float* r = ML + (min(kc * 32 + 8 * q + j, N - 1) * T + tt) * SG_LDP + l16;
                bv[kc][0][j] = r[0]; bv[kc][1][j] = r[16];
              } else {
                bv[kc][0][j] = sg_bl(rs, voff, (kc * 32 + j) * sstep); bv[kc][1][j] = sg_bl(rs, voff + 64, (kc * 32 + j) * sstep);
              }
            }
          }
        float avv[3][8];                                  // (the adjacency operand up front too: one LDS latency per job)
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int v = kc * 32 + 8 * q + j; avv[kc][j] = (kc < kch && v < N) ? As[v * LDA + wcl] : 0.f; }
        sg_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
          if (kc < kch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc0 = SG_MFMA(avv[kc][j], bv[kc][0][j], acc0); acc1 = SG_MFMA(avv[kc][j], bv[kc][1][j], acc1); }
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int w = mt * 16 + 4 * q + i;
          if (w < N) {
            float* o = dst + g.nrow(w, tt) * 32 + l16; o[0] = acc0[i]; o[16] = acc1[i];
            if (LR) { float* m = mir + (3 + 2 * d + hop) * MS + (w * T + tt) * SG_LDP + l16; m[0] = acc0[i]; m[16] = acc1[i]; }
          }
        }
      }
      SG_SYNC();
      SG_T();
    }

    // ---- C: mlp over [g, x1_d, x2_d ...] + bias, dropout, residual; BatchNorm partial sums
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float* hout = a.hs + (long)li * rows * 32;
    const uint32_t lseed = a.seed + 7919u * (uint32_t)li;
    for (int mt = wave; mt < mtiles; mt += SG_WAVES) {
      const int pl = min(mt * 16 + l16, P - 1);
      const long r = g.row(pl);
      float av[NE][8];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float* s = LR ? mir + (2 + e) * MS + pl * SG_LDP
                            : (e == 0 ? gout + r * ldg : a.xs + (((long)li * ND) * 2 + (e - 1)) * rows * 32 + r * 32);
        sg_ld8(s + 8 * q, av[e]);
      }
      sg_f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* wr = wm + (e * 32 + 8 * q + j) * SG_LDM + l16;
          acc[0] = SG_MFMA(av[e][j], wr[0], acc[0]); acc[1] = SG_MFMA(av[e][j], wr[16], acc[1]);
        }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const int c = nh * 16 + l16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int po = mt * 16 + 4 * q + i;
          if (po < P) {
            const long ro = one_strip ? erow[i] : g.row(po);
            float v = acc[nh][i] + bias[64 + c];
            if (a.thresh) v = mo_hash32(lseed, (uint32_t)(ro * 32 + c)) < a.thresh ? 0.f : v * a.dscale;
            v += (LR ? XLc[po * SG_LDP + c] : xin[ro * 32 + c]) * aff[c] + aff[32 + c];
            hout[ro * 32 + c] = v;
            if (LR) XLn[po * SG_LDP + c] = v;
            s1[nh] += v; s2[nh] += v * v;
          }
        }
      }
    }
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      float t1 = s1[nh], t2 = s2[nh];
      t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
      if (q == 0) { part[wave * 64 + nh * 16 + l16] = t1; part[wave * 64 + 32 + nh * 16 + l16] = t2; }
    }
    SG_SYNC();
    SG_T();
    if (tid < 32) {
      float* st = a.stats + ((long)blockIdx.x * a.L + li) * 6 * 32;
      float sc, sh;
      if (a.training) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < SG_WAVES; ++w) { t1 += (double)part[w * 64 + tid]; t2 += (double)part[w * 64 + 32 + tid]; }
        const double mean = t1 / P;
        double var = t2 / P - mean * mean; if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        sc = Ly.gamma[tid] * rstd; sh = Ly.beta[tid] - (float)mean * sc;
        st[64 + tid] = (float)mean; st[96 + tid] = rstd; st[128 + tid] = (float)(P > 1 ? var * P / (P - 1) : var);
      } else {
        const float rstd = 1.f / sqrtf(Ly.rvar[tid] + a.eps);
        sc = Ly.gamma[tid] * rstd; sh = Ly.beta[tid] - Ly.rmean[tid] * sc;
        st[64 + tid] = Ly.rmean[tid]; st[96 + tid] = rstd; st[128 + tid] = Ly.rvar[tid];
      }
      st[tid] = sc; st[32 + tid] = sh;
      aff[tid] = sc; aff[32 + tid] = sh;
    }
    SG_SYNC();
    SG_T();
#ifdef SG_TIMING
    if (tid == 0) { tst[8] = (float)(clock64() - cc0); tst[9] = (float)(wall_clock64() - tt0); }
#endif
  }
}
#undef SG_SYNC

// running statistics: the B calls of a step update them one after the other (unet.py:221), momentum 0.1
__global__ void sg_running_kernel(SgArgs a, float momentum) {
  const int li = blockIdx.x, c = threadIdx.x;
  if (c >= 32) return;
  float rm = a.ly[li].rmean[c], rv = a.ly[li].rvar[c];
  for (int b = 0; b < a.B; ++b) {
    const float* st = a.stats + ((long)b * a.L + li) * 6 * 32;
    rm = (1.f - momentum) * rm + momentum * st[64 + c];
    rv = (1.f - momentum) * rv + momentum * st[128 + c];
  }
  a.ly[li].rmean[c] = rm; a.ly[li].rvar[c] = rv;
}

// ------------------------------------------------------------------------------------------------ backward
// dynamic LDS: adj | wt (k-major filter|gate, for the recompute) | wtn [64][SG_LDM] natural (data gradient of the TCN)
//            | wmn [32][32 ne + 2] natural (data gradient of the mlp) | bias | aff | small reduction scratch
// Jobs as in the forward (a strip of 16 rows x all columns, loads up front).  The weight-gradient tiles (k = the P
// positions of the call) are dealt to the waves BEHIND the data jobs of the same phase: wave w's first weight tile is
// (w - data jobs) mod 16, so the waves without a data job start on them at once.
template <int ND>
__global__ __launch_bounds__(SG_THREADS) void sg_bwd_kernel(SgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NE = 1 + 2 * ND, LDN = 32 * NE + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave_ = tid >> 6, l16 = lane & 15, q = lane >> 4;
  const int N = a.N, T = a.T, P = a.P, LDA = N + 3 - ((N + 1) & 3);
  float* adj_s = lds;
  float* wt = adj_s + ND * N * LDA;
  float* wtn = wt + 32 * SG_LDW;
  float* wmn = wtn + 64 * SG_LDM;
  float* bias = wmn + 32 * LDN;                 // bf | bg
  float* aff = bias + 64;                        // scale, shift of the layer input
  float* bnk = aff + 64;                         // gamma*rstd [32], k1 [32], k2 [32], mean [32], rstd [32]
  double* red = reinterpret_cast<double*>(bnk + 160);    // [2][32][33] doubles
  SgGeo g; g.B = a.B; g.T = T; g.P = P; g.N = N; g.b = blockIdx.x;
  const long rows = a.rows;
  const int ldg = 32 * a.L;
  const long slabL = SG_S_WM + 32L * 32 * (1 + 2 * a.nsup);
  for (int d = 0; d < ND; ++d)
    for (int i = tid; i < N * N; i += SG_THREADS) adj_s[(d * N + i / N) * LDA + i % N] = a.adj[d][i];
  const int mtiles = (P + 15) >> 4, pch = (P + 31) >> 5;
  const bool one_strip = mtiles <= SG_WAVES;          // (see sg_fwd_kernel: epilogue rows once per kernel)
  long erow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) erow[i] = g.row(min(wave_ * 16 + 4 * q + i, P - 1));
  const int ntn = (N + 15) >> 4, kch = (N + 31) >> 5;
  // persistent accumulators of the adaptive support's gradient: tile (mt, nt) of dA[v][w] owned by a fixed wave
  sg_f4 accA[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // 25 tiles at N = 67..80 / 12 waves
  const int nAt = (a.adaptive_dense >= 0 && a.dA) ? ntn * ntn : 0;

  // one weight-gradient tile: D[16 x 16] = sum over the call's positions of A[p][m0 + .] * B[p][n0 + .].  The k order is
  // free: k = (time step, node), so a lane's eight k are eight consecutive NODES at one time step -- a uniform row stride
  // (B*T rows) that goes into the SGPR offset of a buffer load; nodes >= N read 0 from both operands.  fa(value, row)
  // post-processes the A values (dropout mask), fb the B values (BatchNorm affine of the layer input).
  auto wtile = [&](const float* pa, int lda, int ca, const float* pb, int ldb, int cb, auto fa, auto fb) {
    sg_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t ra = sg_rsrc(pa, rows * lda * 4), rb = sg_rsrc(pb, rows * ldb * 4);
    const int sa = a.B * T * lda * 4, sb_ = a.B * T * ldb * 4;
    for (int t = 0; t < T; ++t) {
      const long r0 = g.nrow(8 * q, t);
      const int va = (int)((r0 * lda + ca) * 4), vb = (int)((r0 * ldb + cb) * 4);
#pragma unroll
      for (int kc = 0; kc < 3; ++kc)
        if (kc < kch) {
          float av[8], bv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { av[j] = sg_bl(ra, va, (kc * 32 + j) * sa); bv[j] = sg_bl(rb, vb, (kc * 32 + j) * sb_); }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const long r = r0 + (long)(kc * 32 + j) * a.B * T;
            acc = SG_MFMA(fa(av[j], r), fb(bv[j]), acc);
          }
        }
    }
    return acc;
  };

  for (int li = a.L - 1; li >= 0; --li) {
    const SgLayer& Ly = a.ly[li];
    int wave = wave_; asm volatile("" : "+v"(wave));      // (see sg_fwd_kernel)
    const int wrot = (wave + SG_WAVES - (mtiles % SG_WAVES)) % SG_WAVES;     // first weight tile of this wave
#ifdef SG_TIMING
    long long tb0 = wall_clock64(); float* tsb = a.stats + ((long)blockIdx.x * a.L + li) * 6 * 32 + 160 + 16; int tkb = 0;
#define SG_TB() do { if (tid == 0) tsb[tkb] = (float)(wall_clock64() - tb0); ++tkb; } while (0)
#else
#define SG_TB() do {} while (0)
#endif
    float* slab = a.slab + ((long)blockIdx.x * a.L + li) * slabL;
    const bool has_dh = li < a.L - 1;            // the last layer's gcn / bn output is dead (graph_wavenet.py:252)
    // ---- weights
    for (int i = tid; i < 32 * 32; i += SG_THREADS) {
      const int co = i >> 5, ci = i & 31;
      wt[ci * SG_LDW + co] = Ly.Wf[i]; wt[ci * SG_LDW + 32 + co] = Ly.Wg[i];
      wtn[co * SG_LDM + ci] = Ly.Wf[i]; wtn[(32 + co) * SG_LDM + ci] = Ly.Wg[i];
      const float* wrow = Ly.Wm + (long)co * 32 * (1 + 2 * a.nsup);
      float w0 = wrow[ci];
      for (int k = 0; k < a.nsup; ++k) {
        const int d = a.dense_of[k];
        const float w1 = wrow[32 * (1 + 2 * k) + ci], w2 = wrow[32 * (2 + 2 * k) + ci];
        if (d < 0) w0 += w1 + w2;
        else { wmn[co * LDN + 32 * (1 + 2 * d) + ci] = w1; wmn[co * LDN + 32 * (2 + 2 * d) + ci] = w2; }
      }
      wmn[co * LDN + ci] = w0;
    }
    if (tid < 32) {
      bias[tid] = Ly.bf[tid]; bias[32 + tid] = Ly.bg[tid];
      if (li == 0) { aff[tid] = 1.f; aff[32 + tid] = 0.f; }
      else { const float* sp = a.stats + ((long)blockIdx.x * a.L + li - 1) * 6 * 32; aff[tid] = sp[tid]; aff[32 + tid] = sp[32 + tid]; }
      const float* st = a.stats + ((long)blockIdx.x * a.L + li) * 6 * 32;
      bnk[96 + tid] = st[64 + tid]; bnk[128 + tid] = st[96 + tid];
    }
    __syncthreads(); SG_TB();
    const float* xin = li == 0 ? a.h0 : a.hs + (long)(li - 1) * rows * 32;
    const float* gl = a.gcat + li * 32;
    const float* hl = a.hs + (long)li * rows * 32;
    const uint32_t lseed = a.seed + 7919u * (uint32_t)li;
    const int tr = tid >> 5, tc = tid & 31;       // elementwise phases: 32 rows x 32 channels per sweep
    auto dhm = [&](long r, int c) {               // dropout-masked dh (the gradient of the mlp output)
      float v = a.dh[r * 32 + c];
      if (a.thresh) v = mo_hash32(lseed, (uint32_t)(r * 32 + c)) < a.thresh ? 0.f : v * a.dscale;
      return v;
    };

    if (has_dh) {
      // ---- R1: BatchNorm backward sums  s1 = sum dxo, s2 = sum dxo * xhat   (per channel over the call's P positions)
      {
        double s1 = 0.0, s2 = 0.0;
        const float mu = bnk[96 + tc], rs = bnk[128 + tc];
        for (int pl = tr; pl < P; pl += SG_NR) {
          const long r = g.row(pl);
          const float d = a.dxo[r * 32 + tc];
          s1 += (double)d; s2 += (double)d * (double)((hl[r * 32 + tc] - mu) * rs);
        }
        red[tr * 33 + tc] = s1; red[32 * 33 + tr * 33 + tc] = s2;
      }
      __syncthreads(); SG_TB();
      if (tid < 32) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < SG_NR; ++k) { s1 += red[k * 33 + tid]; s2 += red[32 * 33 + k * 33 + tid]; }
        slab[SG_S_GA + tid] = (float)s2; slab[SG_S_BE + tid] = (float)s1;
        bnk[tid] = Ly.gamma[tid] * bnk[128 + tid];
        bnk[32 + tid] = (float)(s1 / P); bnk[64 + tid] = (float)(s2 / P);
      }
      __syncthreads(); SG_TB();
      // ---- R2: dh = gamma rstd (dxo - k1 - xhat k2); bias gradient of the mlp = sum of the dropout-masked dh
      {
        double s3 = 0.0;
        const float mu = bnk[96 + tc], rs = bnk[128 + tc], gr = bnk[tc], k1 = bnk[32 + tc], k2 = bnk[64 + tc];
        for (int pl = tr; pl < P; pl += SG_NR) {
          const long r = g.row(pl);
          const float v = gr * (a.dxo[r * 32 + tc] - k1 - (hl[r * 32 + tc] - mu) * rs * k2);
          a.dh[r * 32 + tc] = v;
          float m = v;
          if (a.thresh) m = mo_hash32(lseed, (uint32_t)(r * 32 + tc)) < a.thresh ? 0.f : v * a.dscale;
          s3 += (double)m;
        }
        red[tr * 33 + tc] = s3;
      }
      __syncthreads(); SG_TB();
      if (tid < 32) {
        double s3 = 0.0;
        for (int k = 0; k < SG_NR; ++k) s3 += red[k * 33 + tid];
        slab[SG_S_BM + tid] = (float)s3;
      }
      // ---- M: mlp backward.  data: dsrc_e = dhm @ Wm_eff[:, e]  (e = 0: dg = ... + skip-path gradient)
      for (int mt = wave; mt < mtiles; mt += SG_WAVES) {
        const int pl = min(mt * 16 + l16, P - 1);
        const long r = g.row(pl);
        float av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = dhm(r, 8 * q + j);
        sg_f4 acc[2 * NE];
#pragma unroll
        for (int u = 0; u < 2 * NE; ++u) acc[u] = (sg_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* wr = wmn + (8 * q + j) * LDN + l16;
#pragma unroll
          for (int u = 0; u < 2 * NE; ++u) acc[u] = SG_MFMA(av[j], wr[16 * u], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 2 * NE; ++u) {
          const int e = u >> 1, c = (u & 1) * 16 + l16;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int po = mt * 16 + 4 * q + i;
            if (po < P) {
              const long ro = one_strip ? erow[i] : g.row(po);
              if (e == 0) a.dg[ro * 32 + c] = acc[u][i] + a.dgskip[ro * ldg + li * 32 + c];
              else a.dxs[(long)(e - 1) * rows * 32 + ro * 32 + c] = acc[u][i];
            }
          }
        }
      }
#ifdef SG_TIMING
      if (tid == 0) tsb[12] = (float)(wall_clock64() - tb0);     // (wave 0: its strip of the mlp data gradient done)
#endif
      //      weights: dWm[co][e*32 + ci] = sum_p dhm[p][co] src_e[p][ci]
      for (int jw = wrot; jw < 2 * 2 * NE; jw += SG_WAVES) {
        const int ct = jw / (2 * NE), nt = jw - ct * (2 * NE);
        const int e = nt >> 1, ci = (nt & 1) * 16 + l16, co = ct * 16 + l16;
        const float* s = e == 0 ? gl : a.xs + (((long)li * ND) * 2 + (e - 1)) * rows * 32;
        const int lds_ = e == 0 ? ldg : 32;
        const sg_f4 acc = wtile(a.dh, 32, co, s, lds_, ci,
            [&](float v, long r) { return a.thresh ? (mo_hash32(lseed, (uint32_t)(r * 32 + co)) < a.thresh ? 0.f : v * a.dscale) : v; },
            [&](float v) { return v; });
        // rows co = ct*16 + 4q + i, column e*32 + (nt&1)*16 + l16; the g block also is the gradient of every identity block
        const int W = 32 * (1 + 2 * a.nsup);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float* drow = slab + SG_S_WM + (long)(ct * 16 + 4 * q + i) * W;
          if (e == 0) {
            drow[ci] = acc[i];
            for (int k = 0; k < a.nsup; ++k)
              if (a.dense_of[k] < 0) { drow[32 * (1 + 2 * k) + ci] = acc[i]; drow[32 * (2 + 2 * k) + ci] = acc[i]; }
          } else {
            int k = 0;
            for (; k < a.nsup - 1; ++k) if (a.dense_of[k] == ((e - 1) >> 1)) break;
            drow[32 * (1 + 2 * k + ((e - 1) & 1)) + ci] = acc[i];
          }
        }
      }
      __syncthreads(); SG_TB();
      if (ND > 0) {
        // ---- N1: dx1_d += A_d dx2_d   (a job = 16 nodes x the 32 channels of one time step)
        for (int job = wave; job < ND * ntn * T; job += SG_WAVES) {
          const int d = job / (ntn * T), r2 = job - d * (ntn * T), mt = r2 / T, tt = r2 - mt * T;
          const float* As = adj_s + d * N * LDA;
          float* dx1 = a.dxs + (long)(2 * d) * rows * 32;
          const float* dx2 = dx1 + rows * 32;
          const int vcl = min(mt * 16 + l16, N - 1);
          const __amdgpu_buffer_rsrc_t rs = sg_rsrc(dx2, rows * 32 * 4);
          const int voff = (int)((g.nrow(8 * q, tt) * 32 + l16) * 4), sstep = a.B * T * 32 * 4;
          float bv[3][2][8];
#pragma unroll
          for (int kc = 0; kc < 3; ++kc)
            if (kc < kch) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                bv[kc][0][j] = sg_bl(rs, voff, (kc * 32 + j) * sstep); bv[kc][1][j] = sg_bl(rs, voff + 64, (kc * 32 + j) * sstep);
              }
            }
          sg_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
          for (int kc = 0; kc < 3; ++kc)
            if (kc < kch) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int w = kc * 32 + 8 * q + j;
                const float av = w < N ? As[vcl * LDA + w] : 0.f;
                acc0 = SG_MFMA(av, bv[kc][0][j], acc0); acc1 = SG_MFMA(av, bv[kc][1][j], acc1);
              }
            }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int v = mt * 16 + 4 * q + i;
            if (v < N) { float* o = dx1 + g.nrow(v, tt) * 32 + l16; o[0] += acc0[i]; o[16] += acc1[i]; }
          }
        }
        __syncthreads(); SG_TB();
        // ---- N2: dg += sum_d A_d dx1_d ;  dA += x1^T-products of the adaptive support
        for (int job = wave; job < ntn * T; job += SG_WAVES) {
          const int mt = job / T, tt = job - mt * T;
          const int vcl = min(mt * 16 + l16, N - 1);
          sg_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            const float* As = adj_s + d * N * LDA;
            const float* dx1 = a.dxs + (long)(2 * d) * rows * 32;
            const __amdgpu_buffer_rsrc_t rs = sg_rsrc(dx1, rows * 32 * 4);
            const int voff = (int)((g.nrow(8 * q, tt) * 32 + l16) * 4), sstep = a.B * T * 32 * 4;
            float bv[3][2][8];
#pragma unroll
            for (int kc = 0; kc < 3; ++kc)
              if (kc < kch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  bv[kc][0][j] = sg_bl(rs, voff, (kc * 32 + j) * sstep); bv[kc][1][j] = sg_bl(rs, voff + 64, (kc * 32 + j) * sstep);
                }
              }
#pragma unroll
            for (int kc = 0; kc < 3; ++kc)
              if (kc < kch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int w = kc * 32 + 8 * q + j;
                  const float av = w < N ? As[vcl * LDA + w] : 0.f;
                  acc0 = SG_MFMA(av, bv[kc][0][j], acc0); acc1 = SG_MFMA(av, bv[kc][1][j], acc1);
                }
              }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int v = mt * 16 + 4 * q + i;
            if (v < N) { float* o = a.dg + g.nrow(v, tt) * 32 + l16; o[0] += acc0[i]; o[16] += acc1[i]; }
          }
        }
#ifdef SG_TIMING
        if (tid == 0) tsb[13] = (float)(wall_clock64() - tb0);   // (wave 0: its hop job done, dA tiles next)
#endif
        if (nAt) {
          const int d = a.adaptive_dense;
          const float* x1 = a.xs + (((long)li * ND + d) * 2) * rows * 32;
          const float* dx1 = a.dxs + (long)(2 * d) * rows * 32;
          const float* dx2 = dx1 + rows * 32;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int job = ((wave + 6) % SG_WAVES) + s * SG_WAVES;       // (the dg jobs above sit on the low waves)
            if (job < nAt) {
              const int mt = job / ntn, nt = job - mt * ntn;
              const int vcl = min(mt * 16 + l16, N - 1), wcl = min(nt * 16 + l16, N - 1);
              // dA[v][w] += sum_j x1[v][j] dx2[w][j] + g[v][j] dx1[w][j]:  k = j = (t, c), eight contiguous c per lane
              for (int t = 0; t < T; ++t) {
                float a0[8], b0[8], a1[8], b1[8];
                sg_ld8(x1 + g.nrow(vcl, t) * 32 + 8 * q, a0); sg_ld8(dx2 + g.nrow(wcl, t) * 32 + 8 * q, b0);
                sg_ld8(gl + g.nrow(vcl, t) * ldg + 8 * q, a1); sg_ld8(dx1 + g.nrow(wcl, t) * 32 + 8 * q, b1);
#pragma unroll
                for (int j = 0; j < 8; ++j) accA[s] = SG_MFMA(a0[j], b0[j], accA[s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) accA[s] = SG_MFMA(a1[j], b1[j], accA[s]);
              }
            }
          }
        }
        __syncthreads(); SG_TB();
      }
    } else {
      // last layer: only the skip path reaches g
      for (int pl = tr; pl < P; pl += SG_NR) { const long r = g.row(pl); a.dg[r * 32 + tc] = a.dgskip[r * ldg + li * 32 + tc]; }
      if (tid < 32) { slab[SG_S_GA + tid] = 0.f; slab[SG_S_BE + tid] = 0.f; slab[SG_S_BM + tid] = 0.f; }
      for (int i = tid; i < 32 * 32 * (1 + 2 * a.nsup); i += SG_THREADS) slab[SG_S_WM + i] = 0.f;
      __syncthreads(); SG_TB();
    }

    // ---- T1: recompute the pre-activations, form their gradients
    float sb[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int mt = wave; mt < mtiles; mt += SG_WAVES) {
      const int pl = min(mt * 16 + l16, P - 1);
      float av[8];
      sg_ld8(xin + g.row(pl) * 32 + 8 * q, av);
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = av[j] * aff[8 * q + j] + aff[32 + 8 * q + j];
      float dgv[2][4];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i) dgv[nh][i] = a.dg[(one_strip ? erow[i] : g.row(min(mt * 16 + 4 * q + i, P - 1))) * 32 + nh * 16 + l16];
      sg_f4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* wr = wt + (8 * q + j) * SG_LDW + l16;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = SG_MFMA(av[j], wr[16 * u], acc[u]);
      }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const int c = nh * 16 + l16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int po = mt * 16 + 4 * q + i;
          if (po < P) {
            const long ro = one_strip ? erow[i] : g.row(po);
            const float th = sg_tanh(acc[nh][i] + bias[c]), sg = sg_sigmoid(acc[2 + nh][i] + bias[32 + c]);
            const float df = dgv[nh][i] * sg * (1.f - th * th), dgt = dgv[nh][i] * th * sg * (1.f - sg);
            a.dpre[ro * 64 + c] = df; a.dpre[ro * 64 + 32 + c] = dgt;
            sb[nh][0] += df; sb[nh][1] += dgt;
          }
        }
      }
    }
    float* part = reinterpret_cast<float*>(red);           // [SG_WAVES][64]
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      float t1 = sb[nh][0], t2 = sb[nh][1];
      t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
      if (q == 0) { part[wave * 64 + nh * 16 + l16] = t1; part[wave * 64 + 32 + nh * 16 + l16] = t2; }
    }
    __syncthreads(); SG_TB();
    if (tid < 64) {
      double s = 0.0;
      for (int w = 0; w < SG_WAVES; ++w) s += (double)part[w * 64 + tid];
      slab[(tid < 32 ? SG_S_BF : SG_S_BG - 32) + tid] = (float)s;
    }
    // ---- T2: du = dpre @ [Wf; Wg] (+ dh through the residual) -> gradient w.r.t. the layer input (before its
    //      BatchNorm affine); weight gradients of the two convs: dW[c][ci] = sum_p dpre[p][c] xin[p][ci]
    for (int mt = wave; mt < mtiles; mt += SG_WAVES) {
      const int pl = min(mt * 16 + l16, P - 1);
      const long r = g.row(pl);
      float av[2][8];
      sg_ld8(a.dpre + r * 64 + 8 * q, av[0]); sg_ld8(a.dpre + r * 64 + 32 + 8 * q, av[1]);
      float dhv[2][4];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          dhv[nh][i] = has_dh ? a.dh[(one_strip ? erow[i] : g.row(min(mt * 16 + 4 * q + i, P - 1))) * 32 + nh * 16 + l16] : 0.f;
      sg_f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* wr = wtn + (kc * 32 + 8 * q + j) * SG_LDM + l16;
          acc[0] = SG_MFMA(av[kc][j], wr[0], acc[0]); acc[1] = SG_MFMA(av[kc][j], wr[16], acc[1]);
        }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int po = mt * 16 + 4 * q + i;
          if (po < P) a.dxo[(one_strip ? erow[i] : g.row(po)) * 32 + nh * 16 + l16] = acc[nh][i] + dhv[nh][i];
        }
    }
    for (int jw = wrot; jw < 8; jw += SG_WAVES) {
      const int ct = jw >> 1, nh = jw & 1;          // rows c of [Wf; Wg] (64), columns ci (32)
      const int cr = ct * 16 + l16, ci = nh * 16 + l16;
      const float sc = aff[ci], sh = aff[32 + ci];
      // (rows past the tensor read 0 from dpre, so the affine of the 0 read from xin there is multiplied away)
      const sg_f4 acc = wtile(a.dpre, 64, cr, xin, 32, ci, [&](float v, long) { return v; }, [&](float v) { return v * sc + sh; });
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = ct * 16 + 4 * q + i;
        slab[(c < 32 ? SG_S_WF + c * 32 : SG_S_WG + (c - 32) * 32) + ci] = acc[i];
      }
    }
    __syncthreads(); SG_TB();
    // the gradient w.r.t. the layer input x_li = BN_{li-1}(h_{li-1}) is now in dxo: the next iteration's R phases read it
  }
  if (nAt) {
    float* dA = a.dA + (long)blockIdx.x * N * N;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int job = ((wave_ + 6) % SG_WAVES) + s * SG_WAVES;
      if (job < nAt) {
        const int mt = job / ntn, nt = job - mt * ntn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int v = mt * 16 + 4 * q + i, w = nt * 16 + l16;
          if (v < N && w < N) dA[v * N + w] = accA[s][i];
        }
      }
    }
  }
}

// sum of the per-call slabs into the parameter gradients (fixed order over the calls)
struct SgDst { float *Wf, *bf, *Wg, *bg, *Wm, *bm, *gamma, *beta; };
struct SgReduceArgs { SgDst d[SG_MAXL]; const float* slab; int B, L, nsup; float* dA; const float* dAs; int NN; };
__global__ void sg_reduce_kernel(SgReduceArgs a) {
  const long slabL = SG_S_WM + 32L * 32 * (1 + 2 * a.nsup);
  if ((int)blockIdx.x == a.L) {                 // the adaptive support's gradient
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < a.NN; i += gridDim.y * blockDim.x) {
      float s = 0.f;
      for (int b = 0; b < a.B; ++b) s += a.dAs[(long)b * a.NN + i];
      a.dA[i] = s;
    }
    return;
  }
  const int li = blockIdx.x;
  const SgDst& d = a.d[li];
  for (long i = blockIdx.y * blockDim.x + threadIdx.x; i < slabL; i += gridDim.y * blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += a.slab[((long)b * a.L + li) * slabL + i];
    float* base; long o;
    if (i < SG_S_BF) { base = d.Wf; o = i; }
    else if (i < SG_S_WG) { base = d.bf; o = i - SG_S_BF; }
    else if (i < SG_S_BG) { base = d.Wg; o = i - SG_S_WG; }
    else if (i < SG_S_BM) { base = d.bg; o = i - SG_S_BG; }
    else if (i < SG_S_GA) { base = d.bm; o = i - SG_S_BM; }
    else if (i < SG_S_BE) { base = d.gamma; o = i - SG_S_GA; }
    else if (i < SG_S_WM) { base = d.beta; o = i - SG_S_BE; }
    else { base = d.Wm; o = i - SG_S_WM; }
    if (base) base[o] = s;                       // (null: a parameter without a gradient buffer)
  }
}

// ------------------------------------------------------------------------------------------------ C-ABI
static int sg_lda(int N) { return N + 3 - ((N + 1) & 3); }
static size_t sg_fwd_lds(int N, int ndense) {
  const int ne = 1 + 2 * ndense;
  return sizeof(float) * ((size_t)ndense * N * sg_lda(N) + 32 * SG_LDW + 32 * ne * SG_LDM + 96 + 64 + SG_WAVES * 64);
}
static size_t sg_fwd_mirrors(int P, int ndense) { return sizeof(float) * (size_t)(3 + 2 * ndense) * P * SG_LDP; }
static size_t sg_bwd_lds(int N, int ndense) {
  const int ne = 1 + 2 * ndense;
  return sizeof(float) * ((size_t)ndense * N * sg_lda(N) + 32 * SG_LDW + 64 * SG_LDM + 32 * (32 * ne + 2) + 64 + 64 + 160) +
         sizeof(double) * (2 * 32 * 33) + 64;
}
extern "C" int mo_gwnet_small_supported(int N, int T, int L, int nsup, int ndense) {
  if (N < 1 || T < 1 || L < 1 || L > SG_MAXL || nsup < 0 || nsup > SG_MAXS || ndense < 0 || ndense > nsup) return 0;
  if ((long)N * T > 8192 || N > 80) return 0;      // (the adaptive support's gradient: <= 25 tiles of 16 x 16, three per wave)
  return sg_bwd_lds(N, ndense) <= 160 * 1024 && sg_fwd_lds(N, ndense) <= 160 * 1024;
}
extern "C" long mo_gwnet_small_slab_floats(int nsup) { return sg_slab_floats(nsup); }

// params: L * 10 device pointers per layer in SgLayer order; dsts: L * 8 in SgDst order (backward)
static int sg_fill(SgArgs& a, int B, int N, int T, int L, int nsup, const int* dense_of, const float* const* adj,
                   int adaptive_dense, const void* const* params) {
  a.B = B; a.N = N; a.T = T; a.L = L; a.P = N * T; a.nsup = nsup; a.rows = (long)N * B * T;
  int nd = 0;
  for (int k = 0; k < SG_MAXS; ++k) { a.dense_of[k] = -1; a.adj[k] = nullptr; }
  for (int k = 0; k < nsup; ++k) { a.dense_of[k] = dense_of[k]; if (dense_of[k] >= 0) { if (dense_of[k] != nd) return MO_EINVAL; a.adj[nd++] = adj[dense_of[k]]; } }
  a.ndense = nd; a.adaptive_dense = adaptive_dense;
  for (int i = 0; i < L; ++i) {
    const void* const* p = params + (long)i * 10;
    SgLayer& y = a.ly[i];
    y.Wf = (const float*)p[0]; y.bf = (const float*)p[1]; y.Wg = (const float*)p[2]; y.bg = (const float*)p[3];
    y.Wm = (const float*)p[4]; y.bm = (const float*)p[5]; y.gamma = (const float*)p[6]; y.beta = (const float*)p[7];
    y.rmean = (float*)p[8]; y.rvar = (float*)p[9];
    for (int k = 0; k < 10; ++k) if (!p[k]) return MO_EINVAL;
  }
  return MO_OK;
}
extern "C" int mo_gwnet_small_fwd(int B, int N, int T, int L, int nsup, const int* dense_of, const float* const* adj,
                                  const void* const* params, const float* h0, float* gcat, float* hs, float* xs,
                                  float* stats, int training, float eps, float momentum, uint32_t seed,
                                  uint32_t thresh, float dscale, void* stream) {
  MO_CHECK_ARG(B > 0 && B <= 65535 && dense_of && params && h0 && gcat && hs && stats);
  SgArgs a = {};
  int rc = sg_fill(a, B, N, T, L, nsup, dense_of, adj, -1, params);
  if (rc) return rc;
  MO_CHECK_ARG(mo_gwnet_small_supported(N, T, L, nsup, a.ndense) && (a.ndense == 0 || xs));
  a.h0 = h0; a.gcat = gcat; a.hs = hs; a.xs = xs; a.stats = stats; a.training = training; a.eps = eps;
  a.seed = seed; a.thresh = training ? thresh : 0; a.dscale = dscale;
  hipStream_t st = (hipStream_t)stream;
  const bool lr = sg_fwd_lds(N, a.ndense) + sg_fwd_mirrors(a.P, a.ndense) <= 160 * 1024;
  const size_t lds = sg_fwd_lds(N, a.ndense) + (lr ? sg_fwd_mirrors(a.P, a.ndense) : 0);
  static bool attr = false;
  if (!attr) {
    const void* ks[] = {(const void*)sg_fwd_kernel<0, false>, (const void*)sg_fwd_kernel<1, false>, (const void*)sg_fwd_kernel<2, false>,
                        (const void*)sg_fwd_kernel<3, false>, (const void*)sg_fwd_kernel<0, true>, (const void*)sg_fwd_kernel<1, true>,
                        (const void*)sg_fwd_kernel<2, true>, (const void*)sg_fwd_kernel<3, true>, (const void*)sg_bwd_kernel<0>, (const void*)sg_bwd_kernel<1>,
                        (const void*)sg_bwd_kernel<2>, (const void*)sg_bwd_kernel<3>};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MO_ELAUNCH;
    attr = true;
  }
#define SG_FWD(ND) do { if (lr) hipLaunchKernelGGL((sg_fwd_kernel<ND, true>), dim3(B), dim3(SG_THREADS), lds, st, a); \
                        else hipLaunchKernelGGL((sg_fwd_kernel<ND, false>), dim3(B), dim3(SG_THREADS), lds, st, a); } while (0)
  switch (a.ndense) { case 0: SG_FWD(0); break; case 1: SG_FWD(1); break; case 2: SG_FWD(2); break; default: SG_FWD(3); break; }
#undef SG_FWD
  if (training) hipLaunchKernelGGL(sg_running_kernel, dim3(L), dim3(32), 0, st, a, momentum);
  return mo_launch_status();
}
extern "C" int mo_gwnet_small_bwd(int B, int N, int T, int L, int nsup, const int* dense_of, const float* const* adj,
                                  int adaptive_dense, const void* const* params, void* const* dsts, const float* h0,
                                  const float* gcat, const float* hs, const float* xs, const float* stats, float eps,
                                  uint32_t seed, uint32_t thresh, float dscale, const float* dgskip, float* dxo,
                                  float* ws /* mo_gwnet_small_bwd_ws_floats */, float* dA, void* stream) {
  MO_CHECK_ARG(B > 0 && B <= 65535 && dense_of && params && dsts && h0 && gcat && hs && stats && dgskip && dxo && ws);
  SgArgs a = {};
  int rc = sg_fill(a, B, N, T, L, nsup, dense_of, adj, adaptive_dense, params);
  if (rc) return rc;
  MO_CHECK_ARG(mo_gwnet_small_supported(N, T, L, nsup, a.ndense) && (a.ndense == 0 || xs));
  MO_CHECK_ARG(adaptive_dense < a.ndense && (adaptive_dense < 0 || dA));
  a.h0 = h0; a.gcat = const_cast<float*>(gcat); a.hs = const_cast<float*>(hs); a.xs = const_cast<float*>(xs);
  a.stats = const_cast<float*>(stats); a.training = 1; a.eps = eps; a.seed = seed; a.thresh = thresh; a.dscale = dscale;
  a.dgskip = dgskip; a.dxo = dxo;
  const long R = a.rows * 32;
  float* w = ws;
  a.dh = w; w += R; a.dg = w; w += R; a.dxs = w; w += (long)2 * a.ndense * R; a.dpre = w; w += 2 * R;
  const long slabL = sg_slab_floats(nsup);
  a.slab = w; w += (long)B * L * slabL;
  float* dAs = nullptr;
  if (adaptive_dense >= 0) { dAs = w; w += (long)B * N * N; }
  a.dA = dAs;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = sg_bwd_lds(N, a.ndense);
  switch (a.ndense) {
    case 0: hipLaunchKernelGGL(sg_bwd_kernel<0>, dim3(B), dim3(SG_THREADS), lds, st, a); break;
    case 1: hipLaunchKernelGGL(sg_bwd_kernel<1>, dim3(B), dim3(SG_THREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL(sg_bwd_kernel<2>, dim3(B), dim3(SG_THREADS), lds, st, a); break;
    default: hipLaunchKernelGGL(sg_bwd_kernel<3>, dim3(B), dim3(SG_THREADS), lds, st, a); break;
  }
  SgReduceArgs r = {};
  for (int i = 0; i < L; ++i) {
    void* const* p = dsts + (long)i * 8;
    r.d[i].Wf = (float*)p[0]; r.d[i].bf = (float*)p[1]; r.d[i].Wg = (float*)p[2]; r.d[i].bg = (float*)p[3];
    r.d[i].Wm = (float*)p[4]; r.d[i].bm = (float*)p[5]; r.d[i].gamma = (float*)p[6]; r.d[i].beta = (float*)p[7];
  }
  r.slab = a.slab; r.B = B; r.L = L; r.nsup = nsup; r.dA = dA; r.dAs = dAs; r.NN = N * N;
  hipLaunchKernelGGL(sg_reduce_kernel, dim3(L + (dAs ? 1 : 0), 8), dim3(256), 0, st, r);
  return mo_launch_status();
}
extern "C" long mo_gwnet_small_bwd_ws_floats(int B, int N, int T, int L, int nsup, int ndense) {
  const long R = (long)N * B * T * 32;
  return R * (4 + 2 * ndense) + (long)B * L * sg_slab_floats(nsup) + (long)B * N * N + 64;
}
