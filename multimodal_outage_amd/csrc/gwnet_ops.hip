// Graph-WaveNet hot path for gfx950: C-ABI entry points (include/mo_hip.h) + kernels.
// Reference semantics: /root/reference/models/graph_wavenet.py (cited per entry in mo_hip.h).
#include "mo_gemm.hpp"
#include "mo_rowstream.hpp"
#include "../../include/mo_hip.h"

#define ST(s) ((hipStream_t)(s))
#define RED_STAGE 256  // stage-A width of the two-stage [nblk][64] -> [64] reductions

// ------------------------------------------------------------------------------------------------
// GEMM launch helpers
// ------------------------------------------------------------------------------------------------
static void seg_init(MoSeg& s, const float* ptr, int ld) {
  s.ptr = ptr; s.scale = nullptr; s.shift = nullptr; s.ld = ld;
  s.To = 0; s.Ti = 0; s.off = 0; s.relu = 0; s.drop_seed = 0; s.drop_thresh = 0; s.drop_scale = 1.f;
  s.bf16 = 0;
}
static void op_init(MoOperand& o) {
  for (int i = 0; i < MO_MAX_SEG; ++i) seg_init(o.seg[i], nullptr, 0);
  o.nseg = 1; o.segw = 0; o.rows = 0; o.cols = 0;
}
static MoOperand op_simple(const float* ptr, int ld, long rows, long cols) {
  MoOperand o; op_init(o);
  seg_init(o.seg[0], ptr, ld);
  o.rows = (int)rows; o.cols = (int)cols;
  return o;
}
static void epi_init(MoEpi& e, float* out, int ldo) {
  for (int i = 0; i < MO_MAX_SEG; ++i) e.out[i] = nullptr;
  e.out[0] = out; e.nout = 1; e.osegw = 0; e.ldo = ldo;
  e.oTo = 0; e.oTi = 0; e.ooff = 0;
  e.bias = nullptr; e.bias2 = nullptr; e.relu = 0; e.beta = 0;
  e.mask = nullptr; e.ldmask = 0;
  e.add = nullptr; e.ldadd = 0; e.aTo = 0; e.aTi = 0; e.aoff = 0; e.ascale = nullptr; e.ashift = nullptr;
  e.aux = nullptr; e.ldaux = 0;
  e.drop_seed = 0; e.drop_thresh = 0; e.drop_scale = 1.f;
  e.partial = nullptr; e.slab_stride = 0; e.kchunk = 0; e.colsum = nullptr; e.out_bf = nullptr; e.bf_seg = 0;
}

// The branch-free vector loader applies when every segment is 16-byte loadable, segments are 32/64 wide,
// the row map shares one To, ReLU/dropout are operand-wide and the affine (if any) sits on 32-wide segments.
static bool op_fast_ok(const MoOperand& o) {
  if ((o.cols & 3) != 0) return false;
  if (o.nseg > 1 && o.segw != 32 && o.segw != 64) return false;
  for (int s = 0; s < o.nseg; ++s) {
    const MoSeg& g = o.seg[s];
    if (!g.ptr || (((uintptr_t)g.ptr) & 15) || (g.ld & 3)) return false;
    if (g.To != o.seg[0].To || g.relu != o.seg[0].relu) return false;
    if (g.scale && !(o.nseg == 1 ? o.cols <= 32 : o.segw == 32)) return false;
    if (g.drop_thresh && o.nseg != 1) return false;
    if (s > 0 && g.drop_thresh) return false;
  }
  return true;
}
// normalise the identity row map (To == 0) to To = Ti = 1, off = 0 for the fast loader
static MoOperand op_norm(const MoOperand& o) {
  MoOperand r = o;
  for (int s = 0; s < r.nseg; ++s)
    if (r.seg[s].To == 0) { r.seg[s].To = 1; r.seg[s].Ti = 1; r.seg[s].off = 0; }
  return r;
}

// resident-B capacity (k tiles) of the persistent kernel per tile configuration
template <int BN, int BK> struct PersistNK { static constexpr int value = (BN <= 32) ? 8 : 4; };
// Off by default: with one A tile in flight per workgroup and 2 workgroups per CU it measured 9 % SLOWER
// end to end than one workgroup per tile (3 per CU) -- fewer bytes in flight per CU.  It needs a deeper A ring
// before it pays; mo_set_option("persist", 1) enables it for A/B runs.
static int g_persist = 0;

template <int BM, int BN, int BK, int WM, int WN, int AM, int BMODE, int EPI>
static int launch(const MoOperand& A, const MoOperand& B, const MoEpi& E, long M, long N, int nz,
                  hipStream_t st) {
  if (M <= 0 || N <= 0) return MO_OK;
  dim3 grid(mo_cdiv(M, BM), mo_cdiv(N, BN), nz);
  dim3 block(WM * WN * 64);
  MoGeom G = {0, 0, 0, 1, 0, 0, -1, -1};
  if (op_fast_ok(A) && op_fast_ok(B)) {
    const MoOperand An = op_norm(A), Bn = op_norm(B);
    if constexpr (AM == MO_XROWS) {
      constexpr int NKRES = PersistNK<BN, BK>::value;
      const int K = An.cols;
      const int mt = mo_cdiv(M, BM);
      if (g_persist && nz == 1 && E.kchunk == 0 && mo_cdiv(K, BK) <= NKRES && mt >= 64) {
        dim3 pgrid(mt < 512 ? mt : 512, mo_cdiv(N, BN), 1);     // 2 workgroups per CU walk the M tiles
        hipLaunchKernelGGL((mo_conv_persist_kernel<BM, BN, BK, WM, WN, BMODE, EPI, NKRES>), pgrid, block, 0, st, An,
                           Bn, E, G, mt);
        return mo_launch_status();
      }
    }
    // skinny-K row contractions (tiles of 128 x <=64): single-buffered LDS, more workgroups per CU
    constexpr int NBUF = (AM == MO_XROWS && BN <= 64) ? 1 : 2;
    hipLaunchKernelGGL((mo_gemm_kernel<BM, BN, BK, WM, WN, AM, BMODE, EPI, MO_SRC_PLAIN, MO_SRC_PLAIN, 1, NBUF>), grid,
                       block, 0, st, An, Bn, E, G);
  } else {
    hipLaunchKernelGGL((mo_gemm_kernel<BM, BN, BK, WM, WN, AM, BMODE, EPI, MO_SRC_PLAIN, MO_SRC_PLAIN, 0>), grid,
                       block, 0, st, A, B, E, G);
  }
  return mo_launch_status();
}
static bool op_has_bf16(const MoOperand& o) {
  for (int q = 0; q < o.nseg && q < MO_MAX_SEG; ++q) if (o.seg[q].bf16) return true;
  return false;
}

extern "C" int mo_set_option(const char* name, int value) {
  if (!name) return MO_EINVAL;
  if (name[0] == 'p') { g_persist = value; return MO_OK; }   // "persist"
  return MO_EINVAL;
}

// split-K planning for weight gradients: slabs of [M][N], K = P rows
static void wgrad_plan(int M, int N, long P, int& nsplit, int& kchunk) {
  const int BK = 32;
  long tiles = (P + BK - 1) / BK;
  long tmn = (long)mo_cdiv(M, 64) * mo_cdiv(N, 64);
  long want = 768 / tmn;               // ~3 blocks per CU in flight; fewer, longer K-slices keep the slabs small
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  long per = (tiles + want - 1) / want;
  if (per < 4) per = 4;                // at least 128 rows per slice
  kchunk = (int)(per * BK);
  nsplit = (int)((P + kchunk - 1) / kchunk);
  if (nsplit < 1) nsplit = 1;
}

#define RSW_MAX_WG 768      // row-streaming weight gradient: up to 3 workgroups per CU, one slab each
extern "C" long mo_wgrad_ws_floats(int M, int N, long P) {
  int ns, kc; wgrad_plan(M, N, P, ns, kc);
  long a = (long)ns * M * N + (long)ns * M + 64;          // slabs + fused column-sum slabs
  long b = (long)mo_cdiv(P, 512) * (M > N ? M : N) + 64;  // fallback column-sum partials ...
  const long b2 = 1024L * (M > N ? M : N) + 64;           // ... which mo_colsum spreads over up to 1024 blocks (CSW_BLOCKS)
  if (b2 > b) b = b2;
  long c = (long)RSW_MAX_WG * ((long)M * N + M) + 64;     // row-streaming kernel slabs
  if ((M % 32) || (N % 32) || (M / 32) * (N / 32) > 8) c = 0;
  a = a > b ? a : b;
  return a > c ? a : c;
}

// out[i] (=) sum_z slab[z][i]: 32 outputs x 8 z-lanes per block, fixed summation order (deterministic)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, long stride, int nz, float* __restrict__ out,
                                   long n) {
  __shared__ float sm[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + tx;
  float s = 0.f;
  if (i < n)
    for (int z = ty; z < nz; z += 8) s += slab[(long)z * stride + i];
  sm[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sm[q][tx];
    out[i] = t;
  }
}
static inline dim3 slab_grid(long n) { return dim3(mo_cdiv(n, 32)); }

// TCN weight-gradient unpack: slab[z][co'][tau*32+ci] -> dWf/dWg[co][ci][tau]
__global__ void tcn_wgrad_reduce_kernel(const float* __restrict__ slab, long stride, int nz, int K,
                                        float* __restrict__ dWf, float* __restrict__ dWg) {
  __shared__ float sm[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + tx;  // over 64 * 32K
  const int n = 64 * 32 * K;
  float s = 0.f;
  if (i < n)
    for (int z = ty; z < nz; z += 8) s += slab[(long)z * stride + i];
  sm[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sm[q][tx];
    int cop = i / (32 * K), rem = i % (32 * K);
    int tau = rem / 32, ci = rem % 32;
    float* dst = (cop < 32) ? dWf : dWg;
    dst[((cop & 31) * 32 + ci) * K + tau] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// column sums: out[c] = sum_p X[p][c]   (bias gradients)
// ------------------------------------------------------------------------------------------------
#define CS_ROWS 512
__global__ void colsum_partial_kernel(const float* __restrict__ X, long P, int C, float* __restrict__ part, int rpb) {
  __shared__ float sm[256];
  const int CW = C < 256 ? C : 256;
  const int RY = 256 / CW;
  const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
  long r0 = (long)blockIdx.x * rpb;
  long r1 = r0 + rpb; if (r1 > P) r1 = P;
  for (int c0 = 0; c0 < C; c0 += CW) {
    int c = c0 + tx;
    float s = 0.f;
    if (ty < RY && c < C)
      for (long r = r0 + ty; r < r1; r += RY) s += X[r * C + c];
    sm[threadIdx.x] = (ty < RY) ? s : 0.f;
    __syncthreads();
    if (ty == 0 && c < C) {
      float t = 0.f;
      for (int y = 0; y < RY; ++y) t += sm[y * CW + tx];
      part[(long)blockIdx.x * C + c] = t;
    }
    __syncthreads();
  }
}
// wide rows (C % 4 == 0): CSW_BLOCKS blocks stride over 32-row chunks with 16-byte loads, four chunks in flight;
// the [CSW_BLOCKS][C] partials are summed by 32 slab lanes per column (fixed order => deterministic)
#define CSW_BLOCKS 1024
__global__ void __launch_bounds__(256) colsum_wide_kernel(const float4* __restrict__ X, long P, int C4,
                                                          float4* __restrict__ part) {
  __shared__ float4 sm[256];
  const int RY = 256 / C4;                       // rows in flight per block pass (C4 <= 256, divides 256)
  const int tx = threadIdx.x % C4, ty = threadIdx.x / C4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  const long step = (long)CSW_BLOCKS * RY;
  long r = (long)blockIdx.x * RY + ty;
  for (; r + 3 * step < P; r += 4 * step) {
    const float4 a = X[r * C4 + tx], b = X[(r + step) * C4 + tx], c = X[(r + 2 * step) * C4 + tx],
                 d = X[(r + 3 * step) * C4 + tx];
    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
    s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
  }
  for (; r < P; r += step) {
    const float4 a = X[r * C4 + tx];
    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
  }
  s0.x += s1.x + (s2.x + s3.x); s0.y += s1.y + (s2.y + s3.y); s0.z += s1.z + (s2.z + s3.z); s0.w += s1.w + (s2.w + s3.w);
  sm[threadIdx.x] = s0;
  __syncthreads();
  if (ty == 0) {
    float4 t = sm[tx];
    for (int y = 1; y < RY; ++y) { const float4 v = sm[y * C4 + tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    part[(long)blockIdx.x * C4 + tx] = t;
  }
}
__global__ void __launch_bounds__(1024) slab_reduce32_kernel(const float* __restrict__ slab, long stride, int nz,
                                                             float* __restrict__ out, long n) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + tx;
  float s = 0.f;
  if (i < n)
    for (int z = ty; z < nz; z += 32) s += slab[(long)z * stride + i];
  sm[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) t += sm[q][tx];
    out[i] = t;
  }
}
static bool colsum_wide_ok(const float* X, const float* ws, long P, int C) {
  return ((uintptr_t)ws % 16) == 0 && (C % 4) == 0 && C >= 4 && C <= 1024 && (256 % (C / 4)) == 0 && ((uintptr_t)X % 16) == 0 && P >= 4096;
}
extern "C" long mo_colsum_ws_floats(long P, int C) {
  const long a = (long)mo_cdiv(P, CS_ROWS) * C, b = (long)CSW_BLOCKS * C;
  return (a > b ? a : b) + 16;
}
extern "C" int mo_colsum(const float* X, long P, int C, float* out, float* ws, void* stream) {
  MO_CHECK_ARG(X && out && ws && P > 0 && C > 0);
  if (colsum_wide_ok(X, ws, P, C)) {
    hipLaunchKernelGGL(colsum_wide_kernel, dim3(CSW_BLOCKS), dim3(256), 0, ST(stream), (const float4*)X, P, C / 4,
                       (float4*)ws);
    hipLaunchKernelGGL(slab_reduce32_kernel, slab_grid(C), dim3(1024), 0, ST(stream), ws, (long)C, CSW_BLOCKS, out,
                       (long)C);
    return mo_launch_status();
  }
  // rows per block: few rows (the 67-node graph's 134 .. 1072 positions) are spread over up to CSW_BLOCKS blocks -- two
  // blocks of 512 rows walked one dependent load after the other took 139 us for 0.5 MB
  int rpb = (int)mo_cdiv(P, (long)CSW_BLOCKS);
  if (rpb < 4) rpb = 4;
  if (rpb > CS_ROWS) rpb = CS_ROWS;
  int nb = mo_cdiv(P, rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, ST(stream), X, P, C, ws, rpb);
  hipLaunchKernelGGL(slab_reduce_kernel, slab_grid(C), dim3(256), 0, ST(stream), ws, (long)C, nb, out,
                     (long)C);
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------
// layout conversion (B,C,N,T) <-> nbtc rows p=(n*B+b)*T+t of C channels
// ------------------------------------------------------------------------------------------------
// One block per (n-tile of 32 positions along the contiguous (n,t) run, b); transposes a [C-chunk 32][32 pos] tile.
__global__ void nchw_to_nbtc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int N, int T,
                                    const int* __restrict__ node_new) {
  __shared__ float tile[32][33];
  const long NT = (long)N * T;
  const int b = blockIdx.z;
  const long q0 = (long)blockIdx.x * 32;  // position index q = n*T + t within batch b
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int cy = ty; cy < 32; cy += 8) {
    int c = c0 + cy; long q = q0 + tx;
    tile[cy][tx] = (c < C && q < NT) ? x[((long)b * C + c) * NT + q] : 0.f;
  }
  __syncthreads();
  for (int qy = ty; qy < 32; qy += 8) {
    long q = q0 + qy; int c = c0 + tx;
    if (q < NT && c < C) {
      long n = q / T; int t = (int)(q - n * T);
      if (node_new) n = node_new[n];                       // row of the renumbered node space (graph clusters)
      y[((n * B + b) * T + t) * C + c] = tile[tx][qy];
    }
  }
}
__global__ void nbtc_to_nchw_kernel(const float* __restrict__ y, float* __restrict__ x, int B, int C, int N, int T,
                                    const int* __restrict__ node_new) {
  __shared__ float tile[32][33];
  const long NT = (long)N * T;
  const int b = blockIdx.z;
  const long q0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int qy = ty; qy < 32; qy += 8) {
    long q = q0 + qy; int c = c0 + tx;
    float v = 0.f;
    if (q < NT && c < C) {
      long n = q / T; int t = (int)(q - n * T);
      if (node_new) n = node_new[n];
      v = y[((n * B + b) * T + t) * C + c];
    }
    tile[qy][tx] = v;
  }
  __syncthreads();
  for (int cy = ty; cy < 32; cy += 8) {
    int c = c0 + cy; long q = q0 + tx;
    if (c < C && q < NT) x[((long)b * C + c) * NT + q] = tile[tx][cy];
  }
}
extern "C" int mo_nchw_to_nbtc(const float* x, float* y, int B, int C, int N, int T, const int32_t* node_new,
                               void* stream) {
  MO_CHECK_ARG(x && y && B > 0 && C > 0 && N > 0 && T > 0 && B < 65536);
  dim3 grid(mo_cdiv((long)N * T, 32), mo_cdiv(C, 32), B);
  hipLaunchKernelGGL(nchw_to_nbtc_kernel, grid, dim3(256), 0, ST(stream), x, y, B, C, N, T, node_new);
  return mo_launch_status();
}
extern "C" int mo_nbtc_to_nchw(const float* y, float* x, int B, int C, int N, int T, const int32_t* node_new,
                               void* stream) {
  MO_CHECK_ARG(x && y && B > 0 && C > 0 && N > 0 && T > 0 && B < 65536);
  dim3 grid(mo_cdiv((long)N * T, 32), mo_cdiv(C, 32), B);
  hipLaunchKernelGGL(nbtc_to_nchw_kernel, grid, dim3(256), 0, ST(stream), y, x, B, C, N, T, node_new);
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------
// 1x1 conv / Linear
// ------------------------------------------------------------------------------------------------
static int linear_bm(long P);
extern "C" int mo_conv1x1_fwd(const float* in, int Ci, int To, int Ti, int off, int in_relu, const float* W,
                              const float* b, int Co, float* out, long P_out, int out_relu, int beta,
                              void* stream) {
  MO_CHECK_ARG(in && W && out && Ci > 0 && Co > 0 && P_out > 0 && P_out < (1L << 31));
  MoOperand A = op_simple(in, Ci, P_out, Ci);
  A.seg[0].To = To; A.seg[0].Ti = Ti; A.seg[0].off = off; A.seg[0].relu = in_relu;
  MoOperand Bo = op_simple(W, Ci, Co, Ci);   // XROWS: rows = n = co, cols = k = ci
  MoEpi E; epi_init(E, out, Co);
  E.bias = b; E.relu = out_relu; E.beta = beta;
  if (Co <= 32)
    return launch<128, 32, 32, 4, 1, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P_out, Co, 1, ST(stream));
  if (linear_bm(P_out) == 32)      // few rows (the 67-county Graph WaveNet inside Modified_UNET, the fc layers): 32-row tiles
    return launch<32, 128, 32, 1, 4, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P_out, Co, 1, ST(stream));
  return launch<128, 128, 16, 2, 2, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P_out, Co, 1, ST(stream));
}

// Every layer's skip conv in one contraction over the concatenated channel axis (K = 32 * nl): the skip tensor
// is written once instead of being read-modified-written by each layer.
extern "C" int mo_skip_fwd(const float* const* g, const int* Tout, const float* const* W, int nl, const float* bias,
                           int Cs, long G, int Tf, float* skip, int beta, int relu, void* skip_bf16, void* stream) {
  MO_CHECK_ARG(g && Tout && W && skip && nl >= 1 && nl <= MO_MAX_SEG && Cs > 0 && G > 0 && Tf > 0);
  const long P = G * Tf;
  MO_CHECK_ARG(P < (1L << 31));
  MoOperand A; op_init(A);
  A.nseg = nl; A.segw = 32; A.rows = (int)P; A.cols = 32 * nl;
  MoOperand Bo; op_init(Bo);
  Bo.nseg = nl; Bo.segw = 32; Bo.rows = Cs; Bo.cols = 32 * nl;     // XROWS: rows = n = co, cols = k = (layer, ci)
  for (int i = 0; i < nl; ++i) {
    MO_CHECK_ARG(g[i] && W[i] && Tout[i] >= Tf);
    seg_init(A.seg[i], g[i], 32);
    A.seg[i].To = Tf; A.seg[i].Ti = Tout[i]; A.seg[i].off = Tout[i] - Tf;
    seg_init(Bo.seg[i], W[i], 32);
  }
  MoEpi E; epi_init(E, skip, Cs);
  E.bias = bias; E.beta = beta; E.relu = relu;                       // every consumer of skip applies ReLU first
  E.out_bf = (unsigned short*)skip_bf16; E.bf_seg = 0;
  if (Cs <= 32)
    return launch<128, 32, 32, 4, 1, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P, Cs, 1, ST(stream));
  return launch<128, 128, 16, 2, 2, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P, Cs, 1, ST(stream));
}

// Data gradient of a 1x1 conv with FEW output channels (end_conv_2: Co = 12): no contraction worth a tile kernel,
// one streaming pass -- din[p][j] = (mask[p][j] > 0) ? sum_c dout[p][c] * W[c][j] : 0, optional bf16 copy.
// A thread owns four columns j (its 4*Co weights stay in registers) and walks the rows.
template <int CO>
__global__ __launch_bounds__(256) void smallk_bwd_data_kernel(const float* __restrict__ dout, int Co, long P,
                                                             const float* __restrict__ W, int Ci,
                                                             const float* __restrict__ mask, float* __restrict__ din,
                                                             unsigned short* __restrict__ din_bf) {
  const int tpr = Ci / 4;                         // threads per row
  const int rpi = 256 / tpr;                      // rows per iteration of the block
  const int j4 = (threadIdx.x % tpr) * 4, rl = threadIdx.x / tpr;
  if (rl >= rpi) return;
  float4 w[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c)
    w[c] = (c < Co) ? *reinterpret_cast<const float4*>(&W[(long)c * Ci + j4]) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p = (long)blockIdx.x * rpi + rl; p < P; p += (long)gridDim.x * rpi) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float d = (c < Co) ? dout[p * Co + c] : 0.f;
      v.x += d * w[c].x; v.y += d * w[c].y; v.z += d * w[c].z; v.w += d * w[c].w;
    }
    if (mask) {
      const float4 m = *reinterpret_cast<const float4*>(&mask[p * Ci + j4]);
      v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4*>(&din[p * Ci + j4]) = v;
    if (din_bf) {
      __bf16 t0 = (__bf16)v.x, t1 = (__bf16)v.y, t2 = (__bf16)v.z, t3 = (__bf16)v.w;
      const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, t0) | ((unsigned)__builtin_bit_cast(unsigned short, t1) << 16);
      const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, t2) | ((unsigned)__builtin_bit_cast(unsigned short, t3) << 16);
      *reinterpret_cast<uint2*>(&din_bf[p * Ci + j4]) = make_uint2(lo, hi);
    }
  }
}
extern "C" int mo_conv1x1_bwd_data_smallk(const float* dout, int Co, long P, const float* W, int Ci, const float* mask,
                                          float* din, void* din_bf16, void* stream) {
  MO_CHECK_ARG(dout && W && din && Co >= 1 && Co <= 16 && P > 0 && Ci >= 4 && Ci <= 1024 && (Ci % 4) == 0);
  MO_CHECK_ARG((256 % (Ci / 4)) == 0 || Ci / 4 > 128);
  MO_CHECK_ARG((((uintptr_t)W) & 15) == 0 && (((uintptr_t)din) & 15) == 0 && (((uintptr_t)mask) & 15) == 0);
  const int rpi = 256 / (Ci / 4);
  long nb = (P + rpi - 1) / rpi;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL((smallk_bwd_data_kernel<16>), dim3((unsigned)nb), dim3(256), 0, ST(stream), dout, Co, P, W, Ci, mask,
                     din, (unsigned short*)din_bf16);
  return mo_launch_status();
}

extern "C" int mo_conv1x1_bwd_data(const float* dout, int Co, long P, const float* W, int Ci, float* din,
                                   int oTo, int oTi, int ooff, const float* mask, int beta, void* stream) {
  MO_CHECK_ARG(dout && W && din && Ci > 0 && Co > 0 && P > 0 && P < (1L << 31));
  MO_CHECK_ARG(!(mask && oTo));   // mask is indexed by the unmapped row
  MoOperand A = op_simple(dout, Co, P, Co);     // XROWS: rows = m = p, cols = k = co
  MoOperand Bo = op_simple(W, Ci, Co, Ci);      // KROWS: rows = k = co, cols = n = ci
  MoEpi E; epi_init(E, din, Ci);
  E.oTo = oTo; E.oTi = oTi; E.ooff = ooff; E.mask = mask; E.ldmask = Ci; E.beta = beta;
  if (Ci <= 32)
    return launch<128, 32, 32, 4, 1, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, Ci, 1, ST(stream));
  if (linear_bm(P) == 32)
    return launch<32, 128, 32, 1, 4, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, Ci, 1, ST(stream));
  return launch<128, 128, 16, 2, 2, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, Ci, 1, ST(stream));
}

// Few-row Linear layers against long K (the UNet Encoder/Decoder fc layers: ~134 rows x K = 16384): the tile grid
// alone is 10..26 workgroups, so K is split over blockIdx.z into slabs and a second pass sums them (+ bias, ReLU).
__global__ void linear_splitk_reduce_kernel(const float* __restrict__ ws, long stride, int nz, const float* __restrict__ bias,
                                            int ld, int relu, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = bias ? bias[i % ld] : 0.f;
  for (int z = 0; z < nz; ++z) s += ws[(long)z * stride + i];
  out[i] = relu ? fmaxf(s, 0.f) : s;
}
// row tile of the few-row Linear kernels: 134 rows (67 counties x 2 days) on 128-row tiles are two tiles, the second
// 95 % padding -- the exact-fp32 MFMA then does twice the work and the layer is compute bound on padding; 32-row tiles
// pad 134 to 160
static int linear_bm(long P) { return (mo_cdiv(P, 32) * 32L * 5 <= mo_cdiv(P, 128) * 128L * 4) ? 32 : 128; }
static int linear_splitk_plan(long P, int N, int K, int& nz, int& kchunk) {
  const long tiles = (long)mo_cdiv(P, linear_bm(P)) * mo_cdiv(N, 128);
  long want = 768 / tiles; if (want < 1) want = 1;
  long maxz = K / 512; if (maxz < 1) maxz = 1;                 // at least 512 of K per slab
  if (want > maxz) want = maxz;
  if (want > 64) want = 64;
  kchunk = (int)(((K + want - 1) / want + 31) / 32 * 32);
  nz = (K + kchunk - 1) / kchunk;
  return nz;
}
// Skip path backward in the throughput mode: the data gradients of ALL layers' skip convs are one bf16 ring-GEMM
// launch (dskip [G*Tf][Cs] x [Cs][32*L], mo_gemm_bf16_256); this adds layer `col0/32`'s 32 columns of that result to
// the last Tf time steps of its dg rows (the crop of graph_wavenet.py:230-236 backwards).
__global__ void skip_dg_add_kernel(const float4* __restrict__ all, int ld4, int col4, long rows, int Tf, int Tout,
                                   float4* __restrict__ dg) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = i >> 3;
  if (r >= rows) return;
  const int c = (int)(i & 7);
  const long g = r / Tf;
  const int t = (int)(r - g * Tf);
  const float4 v = all[r * ld4 + col4 + c];
  float4* d = dg + (g * Tout + (Tout - Tf + t)) * 8 + c;
  float4 o = *d;
  o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
  *d = o;
}
extern "C" int mo_skip_bwd_add(const float* all, int ld, int col0, long G, int Tf, int Tout, float* dg, void* stream) {
  MO_CHECK_ARG(all && dg && G > 0 && Tf > 0 && Tout >= Tf && ld >= col0 + 32 && col0 >= 0);
  MO_CHECK_ARG((ld % 4) == 0 && (col0 % 4) == 0 && (((uintptr_t)all) & 15) == 0 && (((uintptr_t)dg) & 15) == 0);
  const long rows = G * Tf;
  hipLaunchKernelGGL(skip_dg_add_kernel, dim3((unsigned)mo_cdiv(rows * 8, 256)), dim3(256), 0, ST(stream),
                     (const float4*)all, ld / 4, col0 / 4, rows, Tf, Tout, (float4*)dg);
  return mo_launch_status();
}

extern "C" long mo_linear_splitk_ws_floats(long P, int N, int K) {
  int nz, kc; linear_splitk_plan(P, N, K, nz, kc);
  return (long)nz * P * N + 64;
}
extern "C" int mo_conv1x1_fwd_splitk(const float* in, int Ci, const float* W, const float* b, int Co, float* out,
                                     long P, int out_relu, float* ws, void* stream) {
  MO_CHECK_ARG(in && W && out && ws && Ci > 0 && Co > 0 && P > 0 && P < (1L << 31));
  int nz, kchunk; linear_splitk_plan(P, Co, Ci, nz, kchunk);
  if (nz <= 1) return mo_conv1x1_fwd(in, Ci, 0, 0, 0, 0, W, b, Co, out, P, out_relu, 0, stream);
  MoOperand A = op_simple(in, Ci, P, Ci);
  MoOperand Bo = op_simple(W, Ci, Co, Ci);     // XROWS: rows = n = co, cols = k = ci
  MoEpi E; epi_init(E, ws, Co);
  E.slab_stride = P * (long)Co; E.kchunk = kchunk;
  int rc = (linear_bm(P) == 32)
               ? launch<32, 128, 32, 1, 4, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P, Co, nz, ST(stream))
               : launch<128, 128, 16, 2, 2, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, P, Co, nz, ST(stream));
  if (rc) return rc;
  const long n = P * (long)Co;
  hipLaunchKernelGGL(linear_splitk_reduce_kernel, dim3(mo_cdiv(n, 256)), dim3(256), 0, ST(stream), ws, n, nz, b, Co,
                     out_relu, out, n);
  return mo_launch_status();
}
extern "C" int mo_conv1x1_bwd_data_splitk(const float* dout, int Co, long P, const float* W, int Ci, float* din,
                                          float* ws, void* stream) {
  MO_CHECK_ARG(dout && W && din && ws && Ci > 0 && Co > 0 && P > 0 && P < (1L << 31));
  int nz, kchunk; linear_splitk_plan(P, Ci, Co, nz, kchunk);
  if (nz <= 1) return mo_conv1x1_bwd_data(dout, Co, P, W, Ci, din, 0, 0, 0, nullptr, 0, stream);
  MoOperand A = op_simple(dout, Co, P, Co);     // XROWS: rows = m = p, cols = k = co
  MoOperand Bo = op_simple(W, Ci, Co, Ci);      // KROWS: rows = k = co, cols = n = ci
  MoEpi E; epi_init(E, ws, Ci);
  E.slab_stride = P * (long)Ci; E.kchunk = kchunk;
  int rc = (linear_bm(P) == 32)
               ? launch<32, 128, 32, 1, 4, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, Ci, nz, ST(stream))
               : launch<128, 128, 16, 2, 2, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, Ci, nz, ST(stream));
  if (rc) return rc;
  const long n = P * (long)Ci;
  hipLaunchKernelGGL(linear_splitk_reduce_kernel, dim3(mo_cdiv(n, 256)), dim3(256), 0, ST(stream), ws, n, nz,
                     (const float*)nullptr, Ci, 0, din, n);
  return mo_launch_status();
}

// generic weight-gradient: slab[z][M][N] = sum_{k in chunk z} A(k,m) B(k,n), then reduce.  When `db` is
// given, the column sums of A (the bias gradient) are produced by the same pass (fast loader) or by a
// separate column-sum kernel over `a_plain` [P][M] (generic loader; a_plain may be null if impossible).
// Row-streaming path (mo_rowstream.hpp): a is one unmapped [P][32*MA] segment, b is NB 32-wide segments.
static bool rsw_ok(const MoOperand& A, const MoOperand& B, long P, int M, int N) {
  if ((M % 32) || (N % 32)) return false;
  const int MA = M / 32, NB = N / 32;
  if (!(MA == 1 || MA == 2 || MA == 8) || MA * NB > 8) return false;
  if (A.nseg != 1 || A.cols != M || A.seg[0].ld != M || A.seg[0].To != 0 || A.seg[0].scale || A.seg[0].relu) return false;
  if (!A.seg[0].ptr || (((uintptr_t)A.seg[0].ptr) & 3)) return false;
  if (B.nseg != NB || B.cols != N || (NB > 1 && B.segw != 32)) return false;
  if (P * (long)M >= (1L << 32)) return false;
  for (int j = 0; j < NB; ++j) {
    const MoSeg& g = B.seg[j];
    if (!g.ptr || g.ld != 32 || g.drop_thresh) return false;
    if (g.To != B.seg[0].To || g.relu != B.seg[0].relu) return false;
    if (g.To < 0 || g.To >= 200) return false;
    // bf16 storage: all of segments 1.. or none (segment 0: fp32, or its bf16 copy when the others are bf16), identity row map
    if (j > 0 && g.bf16 != B.seg[1].bf16) return false;
    if (j == 0 && g.bf16 && !(NB > 1 && B.seg[1].bf16)) return false;
    if (g.bf16 && g.To) return false;
  }
  return true;
}
template <int MA, int NB>
static void rsw_launch2(const MoOperand& A, const MoOperand& B, float* slab, float* cs, long P, int post_b, int nwg,
                        bool mf, hipStream_t st) {
  size_t lds = (size_t)MA * NB * 16 * 64 * 2 * sizeof(float);
  if (mf && lds < 4 * 32 * RS_LDXB * sizeof(short)) lds = 4 * 32 * RS_LDXB * sizeof(short);   // the waves' LDS tiles
  if (mf && lds < (size_t)4 * MA * 64 * 8 * sizeof(float)) lds = (size_t)4 * MA * 64 * 8 * sizeof(float);   // bf16-a column sums
  if (mf) {      // bf16 MFMA form (throughput mode): workgroups walk 128-row runs
    const bool bbf = NB > 1 && B.seg[1].bf16;
    dim3 g(nwg), b(256);
    if (bbf && B.seg[0].bf16) hipLaunchKernelGGL((rs_wgrad_bf_kernel<MA, NB, false, true, false, true>), g, b, lds, st, A, B, slab, cs, P, post_b);
    else if (bbf) hipLaunchKernelGGL((rs_wgrad_bf_kernel<MA, NB, false, true>), g, b, lds, st, A, B, slab, cs, P, post_b);
    else if (A.seg[0].bf16) hipLaunchKernelGGL((rs_wgrad_bf_kernel<MA, NB, true, false, true>), g, b, lds, st, A, B, slab, cs, P, post_b);
    else hipLaunchKernelGGL((rs_wgrad_bf_kernel<MA, NB, true, false>), g, b, lds, st, A, B, slab, cs, P, post_b);
    return;
  }
  if (NB > 1 && B.seg[1].bf16)       // bf16-stored sources 1.. (the gcn mlp of the throughput mode; unmapped)
    hipLaunchKernelGGL((rs_wgrad_kernel<MA, NB, false, true>), dim3(nwg), dim3(256), lds, st, A, B, slab, cs, P, post_b);
  else if (B.seg[0].To)
    hipLaunchKernelGGL((rs_wgrad_kernel<MA, NB, true>), dim3(nwg), dim3(256), lds, st, A, B, slab, cs, P, post_b);
  else
    hipLaunchKernelGGL((rs_wgrad_kernel<MA, NB, false>), dim3(nwg), dim3(256), lds, st, A, B, slab, cs, P, post_b);
}
static void rsw_launch(int MA, int NB, const MoOperand& A, const MoOperand& B, float* slab, float* cs, long P,
                       int post_b, int nwg, bool mf, hipStream_t st) {
#define RSW_CASE(a, b) if (MA == a && NB == b) return rsw_launch2<a, b>(A, B, slab, cs, P, post_b, nwg, mf, st);
  RSW_CASE(1, 1) RSW_CASE(1, 2) RSW_CASE(1, 3) RSW_CASE(1, 4) RSW_CASE(1, 5) RSW_CASE(1, 6) RSW_CASE(1, 7)
  RSW_CASE(1, 8) RSW_CASE(2, 1) RSW_CASE(2, 2) RSW_CASE(2, 3) RSW_CASE(2, 4) RSW_CASE(8, 1)
#undef RSW_CASE
}

static int wgrad_run(const MoOperand& A, const MoOperand& Bo, long P, int M, int N, float* ws, float* dW, float* db,
                     hipStream_t st, bool* db_done = nullptr, int* nslab = nullptr, bool mf = false) {
  if (rsw_ok(A, Bo, P, M, N)) {
    // (the mapped / fp32-source instances of the bf16 form ran out of registers: tcn and skip stay on the fp32 form)
    mf = mf && ((Bo.nseg > 1 && Bo.seg[1].bf16) || (Bo.seg[0].To != 0 && M == 64)) && P * (long)M * 4 < 0xFFFFF000L &&
         (((uintptr_t)A.seg[0].ptr) & 15) == 0;      // the gcn mlp and the TCN instances
    for (int j = 0; j < Bo.nseg; ++j) mf = mf && (((uintptr_t)Bo.seg[j].ptr) & 15) == 0;
    const int crows = mf ? 128 / 4 : 2 * rsw_u(M / 32, N / 32);     // rows per wave-chunk (bf16 form: 4 waves x 128-row runs)
    const long nchunk = (P + crows - 1) / crows;
    long nwg = (nchunk + 3) / 4;
    const long cap = 512;                                      // 2 workgroups per CU (register budget)
    if (nwg > cap) nwg = cap;
    float* cs = ws + nwg * (long)M * N;
    int post_b = 0;
    for (int j = 0; j < Bo.nseg; ++j) if (Bo.seg[j].scale) post_b |= 1;
    if (Bo.seg[0].relu) post_b |= 2;
    rsw_launch(M / 32, N / 32, A, Bo, ws, db ? cs : nullptr, P, post_b, (int)nwg, mf, st);
    if (dW) {
      long n = (long)M * N;
      hipLaunchKernelGGL(slab_reduce_kernel, slab_grid(n), dim3(256), 0, st, ws, n, (int)nwg, dW, n);
    }
    if (db) hipLaunchKernelGGL(slab_reduce_kernel, slab_grid(M), dim3(256), 0, st, cs, (long)M, (int)nwg, db, (long)M);
    if (db_done) *db_done = true;
    if (nslab) *nslab = (int)nwg;
    return mo_launch_status();
  }
  int nsplit, kchunk; wgrad_plan(M, N, P, nsplit, kchunk);
  if (nslab) *nslab = nsplit;
  // one K slice (few rows against a wide weight matrix: the Encoder / Decoder fc layers, 134 rows x 4096 x 16384):
  // the tile goes straight to dW -- a slab + reduction would write and re-read the whole 268 MB gradient once more
  const bool direct = nsplit == 1 && dW != nullptr;
  MoEpi E; epi_init(E, direct ? dW : ws, N);
  E.slab_stride = (long)M * N; E.kchunk = kchunk;
  const bool fused = db && op_fast_ok(A) && op_fast_ok(Bo);
  float* cs = ws + (direct ? 0 : (long)nsplit * M * N);
  if (fused) E.colsum = cs;
  int rc = launch<64, 64, 32, 2, 2, MO_KROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, M, N, nsplit, st);
  if (rc) return rc;
  if (dW && !direct) {
    long n = (long)M * N;
    hipLaunchKernelGGL(slab_reduce_kernel, slab_grid(n), dim3(256), 0, st, ws, n, nsplit, dW, n);
  }
  if (fused) hipLaunchKernelGGL(slab_reduce_kernel, slab_grid(M), dim3(256), 0, st, cs, (long)M, nsplit, db, (long)M);
  if (db_done) *db_done = fused;
  return mo_launch_status();
}

extern "C" int mo_conv1x1_bwd_weight(const float* dout, int Co, long P, const float* in, int Ci, int To, int Ti,
                                     int off, int in_relu, float* dW, float* db, float* ws, void* stream) {
  MO_CHECK_ARG(dout && in && dW && ws && Ci > 0 && Co > 0 && P > 0 && P < (1L << 31));
  MoOperand A = op_simple(dout, Co, P, Co);   // KROWS: rows = k = p, cols = m = co
  MoOperand Bo = op_simple(in, Ci, P, Ci);    // KROWS: rows = k = p (mapped), cols = n = ci
  Bo.seg[0].To = To; Bo.seg[0].Ti = Ti; Bo.seg[0].off = off; Bo.seg[0].relu = in_relu;
  bool done = false;
  int rc = wgrad_run(A, Bo, P, Co, Ci, ws, dW, db, ST(stream), &done);
  if (rc) return rc;
  if (db && !done) return mo_colsum(dout, P, Co, db, ws, stream);   // ws reuse is stream-ordered after the reduce
  return MO_OK;
}

// ------------------------------------------------------------------------------------------------
// adaptive adjacency
// ------------------------------------------------------------------------------------------------
__global__ void adp_fwd_kernel(const float* __restrict__ E1, const float* __restrict__ E2, int N, int R,
                               float* __restrict__ adp) {
  __shared__ float red[256];
  __shared__ float e1[64];
  const int v = blockIdx.x;
  if (threadIdx.x < R) e1[threadIdx.x] = E1[(long)v * R + threadIdx.x];
  __syncthreads();
  float mx = -3.0e38f;
  for (int w = threadIdx.x; w < N; w += blockDim.x) {
    float z = 0.f;
    for (int r = 0; r < R; ++r) z += e1[r] * E2[(long)r * N + w];
    z = fmaxf(z, 0.f);
    adp[(long)v * N + w] = z;
    mx = fmaxf(mx, z);
  }
  red[threadIdx.x] = mx; __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
  mx = red[0]; __syncthreads();
  float sum = 0.f;
  for (int w = threadIdx.x; w < N; w += blockDim.x) {
    float e = expf(adp[(long)v * N + w] - mx);
    adp[(long)v * N + w] = e;
    sum += e;
  }
  red[threadIdx.x] = sum; __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  const float inv = 1.f / red[0];
  for (int w = threadIdx.x; w < N; w += blockDim.x) adp[(long)v * N + w] *= inv;
}
__global__ void transpose2d_kernel(const float* __restrict__ a, float* __restrict__ at, int R, int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int y = ty; y < 32; y += 8) {
    int r = r0 + y, c = c0 + tx;
    tile[y][tx] = (r < R && c < C) ? a[(long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int y = ty; y < 32; y += 8) {
    int c = c0 + y, r = r0 + tx;
    if (c < C && r < R) at[(long)c * R + r] = tile[tx][y];
  }
}
extern "C" int mo_adp_fwd(const float* E1, const float* E2, int N, int R, float* adp, float* adpT, void* stream) {
  MO_CHECK_ARG(E1 && E2 && adp && N > 0 && R > 0 && R <= 64);
  hipLaunchKernelGGL(adp_fwd_kernel, dim3(N), dim3(256), 0, ST(stream), E1, E2, N, R, adp);
  if (adpT)
    hipLaunchKernelGGL(transpose2d_kernel, dim3(mo_cdiv(N, 32), mo_cdiv(N, 32)), dim3(256), 0, ST(stream), adp, adpT, N, N);
  return mo_launch_status();
}

// softmax/relu backward per row v: dz[w] = [z>0] * adp[v][w] * (dA[v][w] - sum_w' dA[v][w'] adp[v][w']); dE1[v][r]
__global__ void adp_bwd_rows_kernel(const float* __restrict__ E1, const float* __restrict__ E2,
                                    const float* __restrict__ adp, float* __restrict__ dA, int N, int R,
                                    float* __restrict__ dE1) {
  __shared__ float red[256];
  __shared__ float e1[64];
  __shared__ float acc1[64];
  const int v = blockIdx.x;
  if (threadIdx.x < R) { e1[threadIdx.x] = E1[(long)v * R + threadIdx.x]; }
  __syncthreads();
  float dot = 0.f;
  for (int w = threadIdx.x; w < N; w += blockDim.x) dot += dA[(long)v * N + w] * adp[(long)v * N + w];
  red[threadIdx.x] = dot; __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  dot = red[0]; __syncthreads();
  float part[16];
  // R <= 64 but register array limited: process R in chunks of 16
  for (int rc = 0; rc < R; rc += 16) {
#pragma unroll
    for (int q = 0; q < 16; ++q) part[q] = 0.f;
    for (int w = threadIdx.x; w < N; w += blockDim.x) {
      float z = 0.f;
      for (int r = 0; r < R; ++r) z += e1[r] * E2[(long)r * N + w];
      float a = adp[(long)v * N + w];
      float dz = (z > 0.f) ? a * (dA[(long)v * N + w] - dot) : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) if (rc + q < R) part[q] += dz * E2[(long)(rc + q) * N + w];
      if (rc + 16 >= R) dA[(long)v * N + w] = dz;   // overwrite on the last chunk
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      red[threadIdx.x] = part[q]; __syncthreads();
      for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
      if (threadIdx.x == 0 && rc + q < R) acc1[rc + q] = red[0];
      __syncthreads();
    }
  }
  if (threadIdx.x < R) dE1[(long)v * R + threadIdx.x] = acc1[threadIdx.x];
}
// dE2[r][w] partial over a v-range: part[z][r][w] = sum_{v in chunk z} E1[v][r] * dz[v][w]
#define ADP_VCHUNK 128
__global__ void adp_bwd_cols_kernel(const float* __restrict__ E1, const float* __restrict__ dz, int N, int R,
                                    float* __restrict__ part) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int v0 = blockIdx.y * ADP_VCHUNK;
  int v1 = v0 + ADP_VCHUNK; if (v1 > N) v1 = N;
  for (int rc = 0; rc < R; rc += 16) {
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if (w < N) {
      for (int v = v0; v < v1; ++v) {
        float d = dz[(long)v * N + w];
#pragma unroll
        for (int q = 0; q < 16; ++q) if (rc + q < R) acc[q] += E1[(long)v * R + rc + q] * d;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (rc + q < R) part[((long)blockIdx.y * R + rc + q) * N + w] = acc[q];
    }
  }
}
extern "C" int mo_adp_bwd(const float* E1, const float* E2, const float* adp, float* dA, int N, int R,
                          float* dE1, float* dE2, float* ws, long ws_floats, void* stream) {
  MO_CHECK_ARG(E1 && E2 && adp && dA && dE1 && dE2 && ws && N > 0 && R > 0 && R <= 64);
  int nz = mo_cdiv(N, ADP_VCHUNK);
  MO_CHECK_ARG(ws_floats >= (long)nz * R * N);
  hipLaunchKernelGGL(adp_bwd_rows_kernel, dim3(N), dim3(256), 0, ST(stream), E1, E2, adp, dA, N, R, dE1);
  hipLaunchKernelGGL(adp_bwd_cols_kernel, dim3(mo_cdiv(N, 256), nz), dim3(256), 0, ST(stream), E1, dA, N, R, ws);
  long n = (long)R * N;
  hipLaunchKernelGGL(slab_reduce_kernel, slab_grid(n), dim3(256), 0, ST(stream), ws, n, nz, dE2, n);
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------
// gated TCN
// ------------------------------------------------------------------------------------------------
__global__ void tcn_pack_kernel(const float* __restrict__ Wf, const float* __restrict__ Wg, int K,
                                float* __restrict__ Wp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // over K*64*32: [tau][co'][ci]
  if (i >= K * 64 * 32) return;
  int tau = i / (64 * 32), rem = i % (64 * 32);
  int cop = rem / 32, ci = rem % 32;
  const float* src = (cop < 32) ? Wf : Wg;
  Wp[i] = src[((cop & 31) * 32 + ci) * K + tau];
}
extern "C" int mo_tcn_pack_weights(const float* Wf, const float* Wg, int K, float* Wp, void* stream) {
  MO_CHECK_ARG(Wf && Wg && Wp && K >= 1 && K <= MO_MAX_SEG);
  hipLaunchKernelGGL(tcn_pack_kernel, dim3(mo_cdiv(K * 64 * 32, 256)), dim3(256), 0, ST(stream), Wf, Wg, K, Wp);
  return mo_launch_status();
}

static void tcn_operands(const float* h_prev, const float* scale, const float* shift, const float* Wp, int K,
                         int dil, long G, int Tin, int Tout, MoOperand& A, MoOperand& Bo) {
  op_init(A);
  A.nseg = K; A.segw = 32; A.rows = (int)(G * Tout); A.cols = 32 * K;
  for (int t = 0; t < K; ++t) {
    seg_init(A.seg[t], h_prev, 32);
    A.seg[t].To = Tout; A.seg[t].Ti = Tin; A.seg[t].off = t * dil;
    A.seg[t].scale = scale; A.seg[t].shift = shift;
  }
  op_init(Bo);   // XROWS: rows = n = co' (64), cols = k = tau*32+ci
  Bo.nseg = K; Bo.segw = 32; Bo.rows = 64; Bo.cols = 32 * K;
  for (int t = 0; t < K; ++t) seg_init(Bo.seg[t], Wp + (long)t * 64 * 32, 32);
}

// row-streaming TCN kernels (mo_rowstream.hpp)
static bool rs_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static bool rs_tcn_ok(int K, long G, int Tin, int Tout) {
  return K >= 1 && K <= 3 && Tin < 200 && Tout >= 1 && G * (long)Tin * 256 < 0xFFFFF000L;
}
static void rs_tcn_launch(int which, int K, const RsTcnArgs& a, long rows, bool mf, hipStream_t st) {
  const long NG = (rows + 127) / 128;
  long nwg = (NG + 3) / 4;
  const long cap = (K <= 2 && which == 0) ? 768 : 512;      // workgroups per CU the kernel's registers allow
  if (nwg > cap) nwg = cap;
#define RS_LAUNCH(KER) hipLaunchKernelGGL(KER, dim3((unsigned)nwg), dim3(256), 0, st, a)
#define RS_CASE2(k, M) \
    if (which == 0 && a.crop) RS_LAUNCH((rs_tcn_kernel<k, 0, M, true>)); \
    else if (which == 0) RS_LAUNCH((rs_tcn_kernel<k, 0, M>)); else if (which == 1) RS_LAUNCH((rs_tcn_kernel<k, 1, M>)); \
    else RS_LAUNCH((rs_tcn_du_kernel<k, M>));
#define RS_CASE(k) if (K == k) { if (mf) { RS_CASE2(k, true) } else { RS_CASE2(k, false) } return; }
  RS_CASE(1) RS_CASE(2) RS_CASE(3)
#undef RS_CASE
#undef RS_CASE2
#undef RS_LAUNCH
}

extern "C" int mo_tcn_fwd(const float* h_prev, const float* scale, const float* shift, const float* Wp,
                          const float* bf, const float* bg, int K, int dil, long G, int Tin, float* g_out,
                          void* g_bf16, int mfma_bf16, float* g_crop, int crop_tf, void* stream) {
  const int Tout = Tin - dil * (K - 1);
  MO_CHECK_ARG(h_prev && Wp && bf && bg && (g_out || (g_crop && g_bf16)) && K >= 1 && K <= MO_MAX_SEG && G > 0 && Tout > 0);
  MO_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  MO_CHECK_ARG(G * Tin < (1L << 31));
  MO_CHECK_ARG(!g_crop || (crop_tf >= 1 && crop_tf <= Tout && Tout < 65536 && G * (long)crop_tf * 128 < (1L << 32)));
  if (rs_tcn_ok(K, G, Tin, Tout) && rs_al16(h_prev) && rs_al16(g_out) && rs_al16(scale) && rs_al16(shift)) {
    RsTcnArgs a = {};
    a.h_prev = h_prev; a.scale = scale; a.shift = shift; a.Wp = Wp; a.bf = bf; a.bg = bg;
    a.out = g_out; a.out_bf = (unsigned short*)g_bf16; a.G = G; a.Tin = Tin; a.Tout = Tout; a.dil = dil;
    a.crop = g_crop; a.crop_tf = g_crop ? crop_tf : 0;
    rs_tcn_launch(0, K, a, G * Tout, mfma_bf16 != 0, ST(stream));
    return mo_launch_status();
  }
  if (mfma_bf16 || g_crop || !g_out) return MO_EUNSUPPORTED;
  MoOperand A, Bo; tcn_operands(h_prev, scale, shift, Wp, K, dil, G, Tin, Tout, A, Bo);
  MoEpi E; epi_init(E, g_out, 32);
  E.bias = bf; E.bias2 = bg; E.out_bf = (unsigned short*)g_bf16;
  return launch<128, 64, 32, 4, 1, MO_XROWS, MO_XROWS, MO_EPI_GATE>(A, Bo, E, G * Tout, 64, 1, ST(stream));
}

extern "C" int mo_tcn_bwd(const float* h_prev, const float* scale, const float* shift, const float* Wp,
                          const float* bf, const float* bg, int K, int dil, long G, int Tin, const float* dg,
                          const float* dres, float* du, float* dWf, float* dWg, float* dbf, float* dbg,
                          float* dpre_ws, float* ws2, int parts, int mfma_bf16, void* stream) {
  const int Tout = Tin - dil * (K - 1);
  MO_CHECK_ARG(h_prev && Wp && bf && bg && dg && dWf && dWg && dbf && dbg && dpre_ws && ws2);
  MO_CHECK_ARG(K >= 1 && K <= MO_MAX_SEG && G > 0 && Tout > 0 && G * Tin < (1L << 31));
  hipStream_t st = ST(stream);
  const long Pout = G * Tout, Pin = G * Tin;
  int rc = MO_OK;
  const bool rs = rs_tcn_ok(K, G, Tin, Tout) && rs_al16(h_prev) && rs_al16(dg) && rs_al16(dpre_ws) &&
                  rs_al16(scale) && rs_al16(shift) && rs_al16(du) && rs_al16(dres);
  if (rs && (parts & 1)) {
    RsTcnArgs a = {};
    a.h_prev = h_prev; a.scale = scale; a.shift = shift; a.Wp = Wp; a.bf = bf; a.bg = bg;
    a.dg = dg; a.out = dpre_ws; a.G = G; a.Tin = Tin; a.Tout = Tout; a.dil = dil;
    rs_tcn_launch(1, K, a, Pout, mfma_bf16 != 0, st);
    if (du) {
      a.dpre = dpre_ws; a.dres = dres; a.du = du;
      rs_tcn_launch(2, K, a, Pin, mfma_bf16 != 0, st);
    }
  }
  // 1) recompute pre-activations, dpre[p][0:32] = d/d(filter pre-act), [32:64] = d/d(gate pre-act)
  if (!rs && mfma_bf16) return MO_EUNSUPPORTED;
  if (!rs && (parts & 1)) {
    MoOperand A, Bo; tcn_operands(h_prev, scale, shift, Wp, K, dil, G, Tin, Tout, A, Bo);
    MoEpi E; epi_init(E, dpre_ws, 64);
    E.bias = bf; E.bias2 = bg; E.aux = dg; E.ldaux = 32;
    rc = launch<128, 64, 32, 4, 1, MO_XROWS, MO_XROWS, MO_EPI_GATE_BWD>(A, Bo, E, Pout, 64, 1, st);
    if (rc) return rc;
  }
  // 2) data gradient: du[(g,t)][ci] = sum_tau sum_co' dpre[(g,t-tau*d)][co'] Wp[tau][co'][ci]  (+ residual)
  if (!rs && (parts & 1) && du) {
    MoOperand A2; op_init(A2);
    A2.nseg = K; A2.segw = 64; A2.rows = (int)Pin; A2.cols = 64 * K;
    for (int t = 0; t < K; ++t) {
      seg_init(A2.seg[t], dpre_ws, 64);
      A2.seg[t].To = Tin; A2.seg[t].Ti = Tout; A2.seg[t].off = -t * dil;
    }
    MoOperand B2 = op_simple(Wp, 32, (long)K * 64, 32);   // KROWS: rows = k = tau*64+co', cols = ci
    MoEpi E2; epi_init(E2, du, 32);
    if (dres) { E2.add = dres; E2.ldadd = 32; E2.aTo = Tin; E2.aTi = Tout; E2.aoff = -(Tin - Tout); }
    rc = launch<128, 32, 32, 4, 1, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A2, B2, E2, Pin, 32, 1, st);
    if (rc) return rc;
  }
  // 3) weight gradients: slab[co'][tau*32+ci] = sum_p dpre[p][co'] * u[(g,t+tau*d)][ci]
  if (parts & 2) {
    MoOperand A3 = op_simple(dpre_ws, 64, Pout, 64);   // KROWS rows = p, cols = co'
    A3.seg[0].bf16 = mfma_bf16 ? 1 : 0;                // throughput mode: dpre is a bf16 tensor
    MoOperand B3; op_init(B3);
    B3.nseg = K; B3.segw = 32; B3.rows = (int)Pout; B3.cols = 32 * K;
    for (int t = 0; t < K; ++t) {
      seg_init(B3.seg[t], h_prev, 32);
      B3.seg[t].To = Tout; B3.seg[t].Ti = Tin; B3.seg[t].off = t * dil;
      B3.seg[t].scale = scale; B3.seg[t].shift = shift;
    }
    // bias gradients = column sums of dpre (64 columns), fused into the same pass: [dbf | dbg]
    float* db64 = ws2 + mo_wgrad_ws_floats(64, 32 * K, Pout) - 64;
    bool done = false;
    int nsplit = 0;
    rc = wgrad_run(A3, B3, Pout, 64, 32 * K, ws2, nullptr, db64, st, &done, &nsplit, mfma_bf16 != 0);
    if (rc) return rc;
    if (mfma_bf16 && !done) return MO_EUNSUPPORTED;      // bf16 dpre exists on the row-streaming path only
    hipLaunchKernelGGL(tcn_wgrad_reduce_kernel, dim3(mo_cdiv(64 * 32 * K, 32)), dim3(256), 0, st, ws2,
                       (long)64 * 32 * K, nsplit, K, dWf, dWg);
    if (done) {
      (void)hipMemcpyAsync(dbf, db64, 32 * sizeof(float), hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync(dbg, db64 + 32, 32 * sizeof(float), hipMemcpyDeviceToDevice, st);
    } else {
      int nb = mo_cdiv(Pout, CS_ROWS);
      hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, dpre_ws, Pout, 64, ws2, CS_ROWS);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, st, ws2, (long)64, nb, dbf, (long)32);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, st, ws2 + 32, (long)64, nb, dbg, (long)32);
    }
  }
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------
// node-axis products
// ------------------------------------------------------------------------------------------------
// Gather SpMM tuned for the 8-XCD cache hierarchy.  The row matrix is cut into column panels of 64 float4
// (1 KB per row): a panel's working set (n_rows KB, 3 MB at N=3000) fits one XCD's 4 MiB L2, so the ~6
// gathers per output row hit L2 instead of re-reading rows through the fabric.  Workgroups are dealt
// round-robin over the XCDs, so linear block id b -> (xcd = b % 8, i = b / 8) and XCD x walks panels
// x, x+8, ... row by row (placement affects speed only).  One wave per block, one float4 per lane; the edge
// loop is unrolled by 4 with clamped indices / zero weights (four independent 16-byte gathers in flight).
#define SPMM_PANEL 64
__global__ void spmm_csr_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                const float* __restrict__ vals, const float4* __restrict__ X, float4* __restrict__ Y,
                                long J4, int n_rows, int n_panels, int beta) {
  const long b = blockIdx.x;
  const int xcd = (int)(b & 7);
  const long i = b >> 3;
  const int panel = (int)(i / n_rows) * 8 + xcd;
  const int row = (int)(i % n_rows);
  if (panel >= n_panels) return;
  const long j = (long)panel * SPMM_PANEL + threadIdx.x;
  if (j >= J4) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  float4 acc = beta ? Y[(long)row * J4 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = e0; e < e1; e += 4) {
    float a[4]; float4 x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ee = min(e + q, e1 - 1);
      a[q] = (e + q < e1) ? vals[ee] : 0.f;
      x[q] = X[(long)colidx[ee] * J4 + j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc.x += a[q] * x[q].x; acc.y += a[q] * x[q].y; acc.z += a[q] * x[q].z; acc.w += a[q] * x[q].w;
    }
  }
  Y[(long)row * J4 + j] = acc;
}

// bf16-stored rows on either side (the throughput mode keeps the diffusion intermediates as bf16 tensors):
// eight columns per lane, so a bf16 row piece is one 16-byte load/store; fp32 accumulation.
struct Spmm8 { float v[8]; };
__device__ __forceinline__ Spmm8 spmm_ld8(const void* base, long idx8, bool bf) {
  Spmm8 r;
  if (bf) {
    const uint4 u = reinterpret_cast<const uint4*>(base)[idx8];
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
  } else {
    const float4 a = reinterpret_cast<const float4*>(base)[2 * idx8], c = reinterpret_cast<const float4*>(base)[2 * idx8 + 1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = c.x; r.v[5] = c.y; r.v[6] = c.z; r.v[7] = c.w;
  }
  return r;
}
__device__ __forceinline__ unsigned spmm_pack2(float a, float b) {
  __bf16 ta = (__bf16)a, tb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ta) | ((unsigned)__builtin_bit_cast(unsigned short, tb) << 16);
}
template <bool XBF, bool YBF>
__global__ void spmm_csr8_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                 const float* __restrict__ vals, const void* __restrict__ X, void* __restrict__ Y,
                                 long J8, int n_rows, int n_panels, int beta) {
  const long b = blockIdx.x;
  const int xcd = (int)(b & 7);
  const long i = b >> 3;
  const int panel = (int)(i / n_rows) * 8 + xcd;
  const int row = (int)(i % n_rows);
  if (panel >= n_panels) return;
  const long j = (long)panel * SPMM_PANEL + threadIdx.x;
  if (j >= J8) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  Spmm8 acc;
  if (beta) acc = spmm_ld8(Y, (long)row * J8 + j, YBF);
  else {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc.v[c] = 0.f;
  }
  for (int e = e0; e < e1; e += 4) {
    float a[4]; Spmm8 x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ee = min(e + q, e1 - 1);
      a[q] = (e + q < e1) ? vals[ee] : 0.f;
      x[q] = spmm_ld8(X, (long)colidx[ee] * J8 + j, XBF);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc.v[c] += a[q] * x[q].v[c];
  }
  if (YBF) {
    reinterpret_cast<uint4*>(Y)[(long)row * J8 + j] =
        make_uint4(spmm_pack2(acc.v[0], acc.v[1]), spmm_pack2(acc.v[2], acc.v[3]), spmm_pack2(acc.v[4], acc.v[5]),
                   spmm_pack2(acc.v[6], acc.v[7]));
  } else {
    float4* y = reinterpret_cast<float4*>(Y) + 2 * ((long)row * J8 + j);
    y[0] = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
    y[1] = make_float4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
  }
}
extern "C" int mo_spmm_csr(const int32_t* rowptr, const int32_t* colidx, const float* vals, int n_rows,
                           const void* X, void* Y, long J, int beta, int x_bf16, int y_bf16, void* stream) {
  MO_CHECK_ARG(rowptr && colidx && vals && X && Y && n_rows > 0 && J > 0 && (J % 4) == 0);
  MO_CHECK_ARG((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0);
  hipStream_t st = ST(stream);
  if (!x_bf16 && !y_bf16) {
    const long J4 = J / 4;
    const int n_panels = mo_cdiv(J4, SPMM_PANEL);
    const long nblk = 8L * n_rows * mo_cdiv(n_panels, 8);
    MO_CHECK_ARG(nblk < (1L << 31));
    hipLaunchKernelGGL(spmm_csr_kernel, dim3((unsigned)nblk), dim3(SPMM_PANEL), 0, st, rowptr, colidx, vals,
                       (const float4*)X, (float4*)Y, J4, n_rows, n_panels, beta);
    return mo_launch_status();
  }
  MO_CHECK_ARG((J % 8) == 0);
  const long J8 = J / 8;
  const int n_panels = mo_cdiv(J8, SPMM_PANEL);
  const long nblk = 8L * n_rows * mo_cdiv(n_panels, 8);
  MO_CHECK_ARG(nblk < (1L << 31));
  dim3 grid((unsigned)nblk), block(SPMM_PANEL);
  if (x_bf16 && y_bf16)
    hipLaunchKernelGGL((spmm_csr8_kernel<true, true>), grid, block, 0, st, rowptr, colidx, vals, X, Y, J8, n_rows, n_panels, beta);
  else if (x_bf16)
    hipLaunchKernelGGL((spmm_csr8_kernel<true, false>), grid, block, 0, st, rowptr, colidx, vals, X, Y, J8, n_rows, n_panels, beta);
  else
    hipLaunchKernelGGL((spmm_csr8_kernel<false, true>), grid, block, 0, st, rowptr, colidx, vals, X, Y, J8, n_rows, n_panels, beta);
  return mo_launch_status();
}

// ---- blocked CSR with per-block neighbour unions (throughput mode, bf16 rows) ---------------------------------
// The plain CSR kernels above fetch every neighbour row of every output row from L2 (6 reads + 1 write of a row
// piece per output piece): they run at the L2 -> CU rate, not at HBM's.  With the nodes renumbered so that 16
// consecutive output rows are a compact cluster of the graph (host side: gwnet_engine.cluster_order), the 16 rows
// share most of their neighbours: the block's DISTINCT source rows (its union, ~2.4 per output row on the k-NN
// graphs of the benchmark) are staged once in LDS as 512-byte pieces and every output piece is combined from LDS.
//   rowptr/vals: the CSR of the (renumbered) matrix; lcol[e]: position of entry e's column in its block's union;
//   uptr[nb+1], usrc[]: the unions.  One workgroup = one block of SB_R rows x one 256-column piece.
#define SB_R 16
#define SB_UMAX 64
__device__ __forceinline__ float sb_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float sb_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
template <bool YBF>
__global__ void __launch_bounds__(256) spmm_blk_kernel(const int* __restrict__ rowptr, const int* __restrict__ lcol,
                                                       const float* __restrict__ vals, const int* __restrict__ uptr,
                                                       const int* __restrict__ usrc, int n, int nb,
                                                       const unsigned short* __restrict__ X, void* __restrict__ Yv,
                                                       long J, int nc, int nsplit, int beta) {
  __shared__ uint4 tile[SB_UMAX * 32];                         // [union row][512 B]
  const long bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const long idx = bid >> 3;
  const int blk = (int)(idx % nb);
  const int split = (int)(idx / nb);
  // this workgroup's column pieces: xcd + 8*(split + nsplit*k) -- a piece stays on one XCD's L2, and the block's
  // metadata (union, entries) is read once for all of them
  const int cstride = 8 * nsplit;
  int chunk = xcd + 8 * split;
  if (chunk >= nc) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  const int u0 = uptr[blk], U = uptr[blk + 1] - u0;
  const int r0 = blk * SB_R + wave * 4;                        // this wave's 4 output rows
  const int rp = rowptr[min(r0 + min(lane, 4), n)];            // lanes 0..4 hold the 5 row pointers
  int src[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int uu = wave * 2 + half + 8 * k;
    src[k] = (uu < U) ? usrc[u0 + uu] : -1;
  }
  const int eb = __builtin_amdgcn_readlane(rp, 0), ee = __builtin_amdgcn_readlane(rp, 4);
  const int myl0 = (eb + lane < ee) ? lcol[eb + lane] : 0;
  const float myw0 = (eb + lane < ee) ? vals[eb + lane] : 0.f;
  const uint2* t2 = reinterpret_cast<const uint2*>(tile);
  auto fetch = [&](int c, uint4 (&v)[8]) {                      // the union's pieces of column chunk c -> registers
    const long col8 = (long)c * 256 + 8 * l5;
    const bool cv = col8 < J;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = make_uint4(0u, 0u, 0u, 0u);
      if (src[k] >= 0 && cv) v[k] = *reinterpret_cast<const uint4*>(X + (long)src[k] * J + col8);
    }
  };
  uint4 v[8];
  fetch(chunk, v);
  for (; chunk < nc; chunk += cstride) {
    __syncthreads();                                           // everybody is done with the previous chunk's tile
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int uu = wave * 2 + half + 8 * k;
      if (src[k] >= 0) tile[uu * 32 + l5] = v[k];
    }
    __syncthreads();
    if (chunk + cstride < nc) fetch(chunk + cstride, v);       // in flight under this chunk's arithmetic
    if (r0 >= n) continue;
    const long col4 = (long)chunk * 256 + 4 * lane;
    const bool ov = col4 < J;
    float acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
      const int r = r0 + q;
      if (beta && r < n && ov) {
        if (YBF) {
          const uint2 o = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(Yv) + (long)r * J + col4);
          acc[q][0] = sb_lo(o.x); acc[q][1] = sb_hi(o.x); acc[q][2] = sb_lo(o.y); acc[q][3] = sb_hi(o.y);
        } else {
          const float4 o = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Yv) + (long)r * J + col4);
          acc[q][0] = o.x; acc[q][1] = o.y; acc[q][2] = o.z; acc[q][3] = o.w;
        }
      }
    }
    for (int base = eb; base < ee; base += 64) {               // the wave's entries, 64 at a time in registers
      int myl = myl0;
      float myw = myw0;
      if (base != eb) {
        const int e = base + lane;
        myl = (e < ee) ? lcol[e] : 0;
        myw = (e < ee) ? vals[e] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int q0 = max(__builtin_amdgcn_readlane(rp, q), base);
        const int q1 = min(__builtin_amdgcn_readlane(rp, q + 1), base + 64);
        for (int k = q0; k < q1; ++k) {
          const int li = __builtin_amdgcn_readlane(myl, k - base);
          const float w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(myw), k - base));
          const uint2 t = t2[li * 64 + lane];
          acc[q][0] += w * sb_lo(t.x); acc[q][1] += w * sb_hi(t.x);
          acc[q][2] += w * sb_lo(t.y); acc[q][3] += w * sb_hi(t.y);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q;
      if (r < n && ov) {
        if (YBF)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(Yv) + (long)r * J + col4) =
              make_uint2(spmm_pack2(acc[q][0], acc[q][1]), spmm_pack2(acc[q][2], acc[q][3]));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(Yv) + (long)r * J + col4) =
              make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
      }
    }
  }
}
extern "C" int mo_spmm_blk(const int32_t* rowptr, const int32_t* lcol, const float* vals, const int32_t* uptr,
                           const int32_t* usrc, int n_rows, int max_union, const void* X_bf16, void* Y, long J,
                           int beta, int y_bf16, void* stream) {
  MO_CHECK_ARG(rowptr && lcol && vals && uptr && usrc && X_bf16 && Y && n_rows > 0 && J > 0 && (J % 8) == 0);
  MO_CHECK_ARG(max_union >= 0 && max_union <= SB_UMAX);
  MO_CHECK_ARG((((uintptr_t)X_bf16) & 15) == 0 && (((uintptr_t)Y) & 15) == 0);
  const int nb = mo_cdiv(n_rows, SB_R);
  const int nc = (int)((J + 255) / 256);
  // column pieces per XCD, split so that the grid keeps >= ~16 workgroups per CU while a workgroup loops over
  // several pieces (its metadata is read once and the next piece is prefetched under the current one)
  const int per_xcd = mo_cdiv(nc, 8);
  int nsplit = (int)mo_cdiv(4096L, 8L * nb);
  if (nsplit > per_xcd) nsplit = per_xcd;
  if (nsplit < 1) nsplit = 1;
  const long nblk = 8L * nb * nsplit;
  MO_CHECK_ARG(nblk < (1L << 31));
  dim3 grid((unsigned)nblk), block(256);
  if (y_bf16)
    hipLaunchKernelGGL(spmm_blk_kernel<true>, grid, block, 0, ST(stream), rowptr, lcol, vals, uptr, usrc, n_rows, nb,
                       (const unsigned short*)X_bf16, Y, J, nc, nsplit, beta);
  else
    hipLaunchKernelGGL(spmm_blk_kernel<false>, grid, block, 0, ST(stream), rowptr, lcol, vals, uptr, usrc, n_rows, nb,
                       (const unsigned short*)X_bf16, Y, J, nc, nsplit, beta);
  return mo_launch_status();
}

// Two blocked products into ONE pass over Y:  Y (+)= S1 X1 + S2 X2  (the backward of a layer accumulates the two static
// supports' hops into the fp32 gradient of the gated output: as two launches each read and wrote its 0.64 GB; the sums are
// formed in the same order -- Y, then S1's entries, then S2's -- so the result is bit-identical to the two launches).
// Per column chunk the union tile is staged twice (matrix 1, then 2) through the same LDS and the same prefetch registers.
struct SbMat { const int* rowptr; const int* lcol; const float* vals; const int* uptr; const int* usrc; const unsigned short* X; };
template <bool YBF>
__global__ void __launch_bounds__(256) spmm_blk2_kernel(SbMat m0, SbMat m1, int n, int nb, void* __restrict__ Yv, long J,
                                                        int nc, int nsplit, int beta) {
  __shared__ uint4 tile[SB_UMAX * 32];
  const long bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const long idx = bid >> 3;
  const int blk = (int)(idx % nb);
  const int split = (int)(idx / nb);
  const int cstride = 8 * nsplit;
  int chunk = xcd + 8 * split;
  if (chunk >= nc) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  const int r0 = blk * SB_R + wave * 4;
  int src[2][8], rp[2], eb[2], ee[2], myl0[2]; float myw0[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const SbMat& M = m ? m1 : m0;
    const int u0 = M.uptr[blk], U = M.uptr[blk + 1] - u0;
    rp[m] = M.rowptr[min(r0 + min(lane, 4), n)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int uu = wave * 2 + half + 8 * k;
      src[m][k] = (uu < U) ? M.usrc[u0 + uu] : -1;
    }
    eb[m] = __builtin_amdgcn_readlane(rp[m], 0); ee[m] = __builtin_amdgcn_readlane(rp[m], 4);
    myl0[m] = (eb[m] + lane < ee[m]) ? M.lcol[eb[m] + lane] : 0;
    myw0[m] = (eb[m] + lane < ee[m]) ? M.vals[eb[m] + lane] : 0.f;
  }
  const uint2* t2 = reinterpret_cast<const uint2*>(tile);
  uint4 v[8];
  auto fetch = [&](const int m, const int c) {
    const unsigned short* X = m ? m1.X : m0.X;
    const long col8 = (long)c * 256 + 8 * l5;
    const bool cv = col8 < J;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = make_uint4(0u, 0u, 0u, 0u);
      if (src[m][k] >= 0 && cv) v[k] = *reinterpret_cast<const uint4*>(X + (long)src[m][k] * J + col8);
    }
  };
  fetch(0, chunk);
  for (; chunk < nc; chunk += cstride) {
    const long col4 = (long)chunk * 256 + 4 * lane;
    const bool ov = col4 < J;
    float acc[4][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      __syncthreads();                                         // everybody is done with the previous tile
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int uu = wave * 2 + half + 8 * k;
        if (src[m][k] >= 0) tile[uu * 32 + l5] = v[k];
      }
      __syncthreads();
      if (m == 0) fetch(1, chunk);                             // in flight under this matrix's arithmetic
      else if (chunk + cstride < nc) fetch(0, chunk + cstride);
      if (r0 >= n) continue;
      if (m == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
          const int r = r0 + q;
          if (beta && r < n && ov) {
            if (YBF) {
              const uint2 o = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(Yv) + (long)r * J + col4);
              acc[q][0] = sb_lo(o.x); acc[q][1] = sb_hi(o.x); acc[q][2] = sb_lo(o.y); acc[q][3] = sb_hi(o.y);
            } else {
              const float4 o = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Yv) + (long)r * J + col4);
              acc[q][0] = o.x; acc[q][1] = o.y; acc[q][2] = o.z; acc[q][3] = o.w;
            }
          }
        }
      }
      const SbMat& M = m ? m1 : m0;
      for (int base = eb[m]; base < ee[m]; base += 64) {
        int myl = myl0[m];
        float myw = myw0[m];
        if (base != eb[m]) {
          const int e = base + lane;
          myl = (e < ee[m]) ? M.lcol[e] : 0;
          myw = (e < ee[m]) ? M.vals[e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int q0 = max(__builtin_amdgcn_readlane(rp[m], q), base);
          const int q1 = min(__builtin_amdgcn_readlane(rp[m], q + 1), base + 64);
          for (int k = q0; k < q1; ++k) {
            const int li = __builtin_amdgcn_readlane(myl, k - base);
            const float w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(myw), k - base));
            const uint2 t = t2[li * 64 + lane];
            acc[q][0] += w * sb_lo(t.x); acc[q][1] += w * sb_hi(t.x);
            acc[q][2] += w * sb_lo(t.y); acc[q][3] += w * sb_hi(t.y);
          }
        }
      }
    }
    if (r0 >= n) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q;
      if (r < n && ov) {
        if (YBF)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(Yv) + (long)r * J + col4) =
              make_uint2(spmm_pack2(acc[q][0], acc[q][1]), spmm_pack2(acc[q][2], acc[q][3]));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(Yv) + (long)r * J + col4) =
              make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
      }
    }
  }
}
extern "C" int mo_spmm_blk2(const int32_t* rowptr1, const int32_t* lcol1, const float* vals1, const int32_t* uptr1,
                            const int32_t* usrc1, int max_union1, const void* X1_bf16,
                            const int32_t* rowptr2, const int32_t* lcol2, const float* vals2, const int32_t* uptr2,
                            const int32_t* usrc2, int max_union2, const void* X2_bf16,
                            int n_rows, void* Y, long J, int beta, int y_bf16, void* stream) {
  MO_CHECK_ARG(rowptr1 && lcol1 && vals1 && uptr1 && usrc1 && X1_bf16 && rowptr2 && lcol2 && vals2 && uptr2 && usrc2 && X2_bf16);
  MO_CHECK_ARG(Y && n_rows > 0 && J > 0 && (J % 8) == 0);
  MO_CHECK_ARG(max_union1 >= 0 && max_union1 <= SB_UMAX && max_union2 >= 0 && max_union2 <= SB_UMAX);
  MO_CHECK_ARG((((uintptr_t)X1_bf16) & 15) == 0 && (((uintptr_t)X2_bf16) & 15) == 0 && (((uintptr_t)Y) & 15) == 0);
  const int nb = mo_cdiv(n_rows, SB_R);
  const int nc = (int)((J + 255) / 256);
  const int per_xcd = mo_cdiv(nc, 8);
  int nsplit = (int)mo_cdiv(4096L, 8L * nb);
  if (nsplit > per_xcd) nsplit = per_xcd;
  if (nsplit < 1) nsplit = 1;
  const long nblk = 8L * nb * nsplit;
  MO_CHECK_ARG(nblk < (1L << 31));
  SbMat a = {rowptr1, lcol1, vals1, uptr1, usrc1, (const unsigned short*)X1_bf16};
  SbMat b = {rowptr2, lcol2, vals2, uptr2, usrc2, (const unsigned short*)X2_bf16};
  dim3 grid((unsigned)nblk), block(256);
  if (y_bf16) hipLaunchKernelGGL(spmm_blk2_kernel<true>, grid, block, 0, ST(stream), a, b, n_rows, nb, Y, J, nc, nsplit, beta);
  else hipLaunchKernelGGL(spmm_blk2_kernel<false>, grid, block, 0, ST(stream), a, b, n_rows, nb, Y, J, nc, nsplit, beta);
  return mo_launch_status();
}

// Small graphs (N <= 128: the 67-county Graph WaveNet inside Modified_UNET): the node-axis products are a few MFLOP, and
// a 128x128 tile-engine launch spends ~30 us on one padded tile row.  Direct kernels instead:
//   Y[m][j] (+)= sum_k A[k][m] * X[k][j]: one wave per output row and 256 columns, operands straight from L2;
//   dA[v][w] (+)= sum_j X[v][j] * dY[w][j]: a workgroup owns a 16 x 16 block of dA and walks j in panels of 64.
#define ADJS_MAXN 128
// one wave = one output row m x 256 columns (a float4 per lane): A[k][m] is a scalar load, X[k][j..j+3] a coalesced
// 16-byte load from L2; no LDS, no barrier.  J % 4 == 0.
__global__ __launch_bounds__(256) void adj_small_kernel(const float* __restrict__ A, int N, const float* __restrict__ X,
                                                        float* __restrict__ Y, long J, int beta) {
  const int lane = threadIdx.x & 63;
  const int m = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
  const long j = ((long)blockIdx.x * 64 + lane) * 4;
  if (m >= N || j >= J) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* xp = X + j;
  const float* ap = A + m;
#pragma unroll 8
  for (int k = 0; k < N; ++k) {
    const float a = ap[(long)k * N];
    const float4 x = *reinterpret_cast<const float4*>(xp + (long)k * J);
    acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y); acc.z = fmaf(a, x.z, acc.z); acc.w = fmaf(a, x.w, acc.w);
  }
  float4* y = reinterpret_cast<float4*>(Y + (long)m * J + j);
  if (beta) { const float4 o = *y; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
  *y = acc;
}
__global__ __launch_bounds__(256) void adj_grad_small_kernel(const float* __restrict__ X, const float* __restrict__ dY, int N,
                                                             long J, float* __restrict__ dA, int beta) {
  __shared__ float xs[16][65], ys[16][65];
  const int tid = threadIdx.x, tw = tid & 15, tv = tid >> 4;
  const int v0 = blockIdx.y * 16, w0 = blockIdx.x * 16;
  float acc = 0.f;
  for (long jc = 0; jc < J; jc += 64) {
    __syncthreads();
    for (int i = tid; i < 16 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      const bool jin = jc + c < J;
      xs[r][c] = (jin && v0 + r < N) ? X[(long)(v0 + r) * J + jc + c] : 0.f;
      ys[r][c] = (jin && w0 + r < N) ? dY[(long)(w0 + r) * J + jc + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 16
    for (int c = 0; c < 64; ++c) acc = fmaf(xs[tv][c], ys[tw][c], acc);
  }
  if (v0 + tv < N && w0 + tw < N) {
    float* d = dA + (long)(v0 + tv) * N + w0 + tw;
    *d = beta ? *d + acc : acc;
  }
}

extern "C" int mo_adj_gemm(const float* A_km, int N, const float* X, float* Y, long J, int beta, void* stream) {
  MO_CHECK_ARG(A_km && X && Y && N > 0 && J > 0 && J < (1L << 31));
  if (N <= ADJS_MAXN && (J & 3) == 0 && ((((uintptr_t)X) | ((uintptr_t)Y)) & 15) == 0) {
    hipLaunchKernelGGL(adj_small_kernel, dim3((unsigned)mo_cdiv(J, 256), (unsigned)mo_cdiv(N, 4)), dim3(256), 0, ST(stream),
                       A_km, N, X, Y, J, beta);
    return mo_launch_status();
  }
  MoOperand A = op_simple(A_km, N, N, N);        // KROWS: rows = k, cols = m
  MoOperand Bo = op_simple(X, (int)J, N, J);     // KROWS: rows = k, cols = n = j
  MoEpi E; epi_init(E, Y, (int)J);
  E.beta = beta;
  return launch<128, 128, 16, 2, 2, MO_KROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, N, J, 1, ST(stream));
}

extern "C" int mo_adj_grad(const float* X, const float* dY, int N, long J, float* dA, int beta, void* stream) {
  MO_CHECK_ARG(X && dY && dA && N > 0 && J > 0 && J < (1L << 31));
  if (N <= ADJS_MAXN) {
    hipLaunchKernelGGL(adj_grad_small_kernel, dim3(mo_cdiv(N, 16), mo_cdiv(N, 16)), dim3(256), 0, ST(stream), X, dY, N, J, dA, beta);
    return mo_launch_status();
  }
  MoOperand A = op_simple(X, (int)J, N, J);      // XROWS: rows = m = v, cols = k = j
  MoOperand Bo = op_simple(dY, (int)J, N, J);    // XROWS: rows = n = w, cols = k = j
  MoEpi E; epi_init(E, dA, N);
  E.beta = beta;
  return launch<128, 128, 16, 2, 2, MO_XROWS, MO_XROWS, MO_EPI_STORE>(A, Bo, E, N, N, 1, ST(stream));
}

// ------------------------------------------------------------------------------------------------
// gcn mlp + residual + BatchNorm
// ------------------------------------------------------------------------------------------------
extern "C" long mo_mlp_partial_floats(long P) { return ((long)mo_cdiv(P, 128) + RED_STAGE + 2) * 64; }

// row-streaming mlp kernels (mo_rowstream.hpp): 16-byte aligned [P][32] tensors, byte offsets within 32 bits
static bool rs_mlp_ok(const float* const* t, int n, long P) {
  if (P * 128 >= (1L << 32) || n < 1 || n > 7) return false;
  for (int s = 0; s < n; ++s) if (!t[s] || (((uintptr_t)t[s]) & 15)) return false;
  return true;
}
static void rs_mlp_launch(bool fwd, int ns, const RsMlpArgs& a, bool bf, hipStream_t st) {
  const long NG = (a.P + 127) / 128;
  long nwg = (NG + 3) / 4;
  if (nwg > 512) nwg = 512;                       // 2 workgroups per CU
  const bool drop = a.drop_thresh != 0;
#define RS_LAUNCH(K) hipLaunchKernelGGL(K, dim3((unsigned)nwg), dim3(256), 0, st, a)
#define RS_CASE2(k, B) \
    if (fwd && B && a.s0bf) { if (drop) RS_LAUNCH((rs_mlp_fwd_kernel<k, true, B, B>)); else RS_LAUNCH((rs_mlp_fwd_kernel<k, false, B, B>)); } \
    else if (fwd) { if (drop) RS_LAUNCH((rs_mlp_fwd_kernel<k, true, B>)); else RS_LAUNCH((rs_mlp_fwd_kernel<k, false, B>)); } \
    else { if (drop) RS_LAUNCH((rs_mlp_bwd_kernel<k, true, B>)); else RS_LAUNCH((rs_mlp_bwd_kernel<k, false, B>)); }
#define RS_CASE(k) if (ns == k) { if (bf) { RS_CASE2(k, true) } else { RS_CASE2(k, false) } return; }
  RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7)
#undef RS_LAUNCH
#undef RS_CASE
#undef RS_CASE2
}
// bf16-storage masks of the mlp entry points: none, or every source but the first
// (forward / weight gradient: source 0 may be read from its bf16 copy as well -- every bit set)
static bool rs_mask_ok(int mask, int ns) { return mask == 0 || (ns > 1 && (mask | 1) == ((1 << ns) - 1)); }

extern "C" int mo_gcn_mlp_fwd(const float* const* srcs, int ns, const float* W, const float* b, long G, int Tout,
                              int Tin, const float* res, const float* rscale, const float* rshift,
                              uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, float* h,
                              float* partial, int src_bf16_mask, void* stream) {
  MO_CHECK_ARG(srcs && W && b && res && h && partial && ns >= 1 && ns <= MO_MAX_SEG && G > 0);
  MO_CHECK_ARG(rs_mask_ok(src_bf16_mask, ns));
  MO_CHECK_ARG(Tout > 0 && Tin >= Tout && G * Tin < (1L << 31));
  MO_CHECK_ARG((rscale == nullptr) == (rshift == nullptr));
  const long P = G * Tout;
  for (int s = 0; s < ns; ++s) MO_CHECK_ARG(srcs[s]);
  if (rs_mlp_ok(srcs, ns, P) && Tout < 200 && !(((uintptr_t)res) & 15)) {
    RsMlpArgs a = {};
    for (int s = 0; s < ns; ++s) a.src[s] = srcs[s];
    a.out[0] = h; a.W = W; a.bias = b; a.res = res; a.rscale = rscale; a.rshift = rshift; a.partial = partial;
    a.P = P; a.Tout = Tout; a.Tin = Tin;
    a.drop_seed = drop_seed; a.drop_thresh = drop_thresh; a.drop_scale = drop_scale;
    a.s0bf = src_bf16_mask & 1;
    rs_mlp_launch(true, ns, a, src_bf16_mask != 0, ST(stream));
    return mo_launch_status();
  }
  if (src_bf16_mask) return MO_EUNSUPPORTED;     // bf16 storage exists on the row-streaming path only
  MoOperand A; op_init(A);
  A.nseg = ns; A.segw = 32; A.rows = (int)P; A.cols = 32 * ns;
  for (int s = 0; s < ns; ++s) { seg_init(A.seg[s], srcs[s], 32); }
  MoOperand Bo = op_simple(W, 32 * ns, 32, 32 * ns);   // XROWS rows = n = co, cols = k
  MoEpi E; epi_init(E, h, 32);
  E.bias = b; E.add = res; E.ldadd = 32; E.aTo = Tout; E.aTi = Tin; E.aoff = Tin - Tout;
  E.ascale = rscale; E.ashift = rshift;
  E.drop_seed = drop_seed; E.drop_thresh = drop_thresh; E.drop_scale = drop_scale;
  E.partial = partial;
  return launch<128, 32, 32, 4, 1, MO_XROWS, MO_XROWS, MO_EPI_MLP>(A, Bo, E, P, 32, 1, ST(stream));
}

// stage A of every [nblk][64] -> [64] reduction: 32 blocks each sum a strided share of the rows
// (coalesced 256-byte rows), fixed order => deterministic
__global__ void rows64_reduce_kernel(const float* __restrict__ part, long nblk, float* __restrict__ out /* [RED_STAGE][64] */) {
  __shared__ double sm[4][64];
  const int c = threadIdx.x & 63, y = threadIdx.x >> 6;   // 256 threads: 4 row lanes x 64 columns
  double s = 0.0;
  for (long b = (long)blockIdx.x * 4 + y; b < nblk; b += (long)RED_STAGE * 4) s += (double)part[b * 64 + c];
  sm[y][c] = s;
  __syncthreads();
  if (y == 0) out[blockIdx.x * 64 + c] = (float)(sm[0][c] + sm[1][c] + sm[2][c] + sm[3][c]);
}

__global__ void bn_finalize_kernel(const float* __restrict__ partial, long nblk, long count, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum,
                                   float eps, int training, float* scale, float* shift, float* mean_out,
                                   float* rstd_out) {
  __shared__ double sm[8][64];
  const int c = threadIdx.x & 63, y = threadIdx.x >> 6;   // 512 threads: 8 lanes x 64 (sum|sumsq) columns
  double s = 0.0;
  if (training)
    for (long b = y; b < nblk; b += 8) s += (double)partial[b * 64 + c];
  sm[y][c] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int ch = threadIdx.x;
    float mean, var;
    if (training) {
      double s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < 8; ++q) { s1 += sm[q][ch]; s2 += sm[q][32 + ch]; }
      double m = s1 / (double)count;
      double v = s2 / (double)count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m; var = (float)v;
      double unbiased = (count > 1) ? v * (double)count / (double)(count - 1) : v;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    } else {
      mean = running_mean[ch]; var = running_var[ch];
    }
    const float rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[ch] * rstd;
    scale[ch] = sc;
    shift[ch] = beta[ch] - mean * sc;
    mean_out[ch] = mean;
    rstd_out[ch] = rstd;
  }
}
extern "C" int mo_bn_finalize(const float* partial, long nblk, long count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int training,
                              float* scale, float* shift, float* mean, float* rstd, void* stream) {
  MO_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && mean && rstd && count > 0);
  MO_CHECK_ARG(!training || (partial && nblk > 0));
  if (training && nblk > 4 * RED_STAGE) {
    // the partial buffer has 64 spare floats... use its tail rows [nblk .. nblk+RED_STAGE) as stage-A output
    float* stage = const_cast<float*>(partial) + nblk * 64;
    hipLaunchKernelGGL(rows64_reduce_kernel, dim3(RED_STAGE), dim3(256), 0, ST(stream), partial, nblk, stage);
    partial = stage; nblk = RED_STAGE;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(512), 0, ST(stream), partial, nblk, count, gamma, beta,
                     running_mean, running_var, momentum, eps, training, scale, shift, mean, rstd);
  return mo_launch_status();
}

// BN backward, phase 1: per-block partial sums of dy and dy*xhat over 128-row chunks (C = 32)
__global__ void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ h, long P,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      float* __restrict__ part) {
  // a thread = 4 channels (one 16-byte load per tensor and row) x every 32nd row of the chunk: the one-float-per-lane form
  // ran at 3.2 TB/s of its 1.4 GB (439 us per launch at batch 256)
  __shared__ float sm[2][32][33];
  const int c4 = (threadIdx.x & 7) * 4, y = threadIdx.x >> 3;   // 32 row lanes
  const long r0 = (long)blockIdx.x * 128;
  long r1 = r0 + 128; if (r1 > P) r1 = P;
  const float4 mu = *reinterpret_cast<const float4*>(mean + c4), rs = *reinterpret_cast<const float4*>(rstd + c4);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (long r = r0 + y; r < r1; r += 32) {
    const float4 d = *reinterpret_cast<const float4*>(dy + r * 32 + c4);
    const float4 x = *reinterpret_cast<const float4*>(h + r * 32 + c4);
    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
    s2.x += d.x * ((x.x - mu.x) * rs.x); s2.y += d.y * ((x.y - mu.y) * rs.y);
    s2.z += d.z * ((x.z - mu.z) * rs.z); s2.w += d.w * ((x.w - mu.w) * rs.w);
  }
  sm[0][y][c4] = s1.x; sm[0][y][c4 + 1] = s1.y; sm[0][y][c4 + 2] = s1.z; sm[0][y][c4 + 3] = s1.w;
  sm[1][y][c4] = s2.x; sm[1][y][c4 + 1] = s2.y; sm[1][y][c4 + 2] = s2.z; sm[1][y][c4 + 3] = s2.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, ch = threadIdx.x & 31;
    float t = 0.f;
    for (int q = 0; q < 32; ++q) t += sm[which][q][ch];
    part[(long)blockIdx.x * 64 + threadIdx.x] = t;
  }
}
__global__ void bn_bwd_final_kernel(const float* __restrict__ part, long nblk, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, float* __restrict__ k12 /* [64] scratch */,
                                    long count) {
  __shared__ double sm[8][64];
  const int c = threadIdx.x & 63, y = threadIdx.x >> 6;
  double s = 0.0;
  for (long b = y; b < nblk; b += 8) s += (double)part[b * 64 + c];
  sm[y][c] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    double t = 0.0;
    for (int q = 0; q < 8; ++q) t += sm[q][threadIdx.x];
    if (threadIdx.x < 32) dbeta[threadIdx.x] = (float)t; else dgamma[threadIdx.x - 32] = (float)t;
    k12[threadIdx.x] = (float)(t / (double)count);
  }
}
__global__ void bn_bwd_apply_kernel(const float4* __restrict__ dy, const float4* __restrict__ h, long n4,
                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ k12,
                                    float4* __restrict__ dh) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i & 7) * 4;
  float4 d = dy[i], x = h[i], o;
  float* dp = &d.x; float* xp = &x.x; float* op = &o.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float rs = rstd[c + q];
    const float xh = (xp[q] - mean[c + q]) * rs;
    op[q] = gamma[c + q] * rs * (dp[q] - k12[c + q] - xh * k12[32 + c + q]);
  }
  dh[i] = o;
}
extern "C" int mo_bn_bwd(const float* dy, const float* h, long P, const float* gamma, const float* mean,
                         const float* rstd, float* dh, float* dgamma, float* dbeta, float* ws, void* stream) {
  MO_CHECK_ARG(dy && h && gamma && mean && rstd && dh && dgamma && dbeta && ws && P > 0);
  hipStream_t st = ST(stream);
  const long nblk = mo_cdiv(P, 128);
  float* k12 = ws + nblk * 64;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk), dim3(256), 0, st, dy, h, P, mean, rstd, ws);
  if (nblk > 4 * RED_STAGE) {
    float* stage = ws + (nblk + 1) * 64;
    hipLaunchKernelGGL(rows64_reduce_kernel, dim3(RED_STAGE), dim3(256), 0, st, ws, nblk, stage);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(1), dim3(512), 0, st, stage, (long)RED_STAGE, dgamma, dbeta, k12, P);
  } else {
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(1), dim3(512), 0, st, ws, nblk, dgamma, dbeta, k12, P);
  }
  const long n4 = P * 8;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(mo_cdiv(n4, 256)), dim3(256), 0, st, (const float4*)dy,
                     (const float4*)h, n4, gamma, mean, rstd, k12, (float4*)dh);
  return mo_launch_status();
}

__global__ void mlp_bias_grad_kernel(const float* __restrict__ dh, long P, uint32_t seed, uint32_t thresh,
                                     float dscale, float* __restrict__ part) {
  __shared__ float sm[256];
  const int c = threadIdx.x & 31, y = threadIdx.x >> 5;
  const long r0 = (long)blockIdx.x * CS_ROWS;
  long r1 = r0 + CS_ROWS; if (r1 > P) r1 = P;
  float s = 0.f;
  for (long r = r0 + y; r < r1; r += 8) {
    float v = dh[r * 32 + c];
    if (thresh) {
      uint32_t hsh = mo_hash32(seed, (uint32_t)(r * 32 + c));
      v = (hsh >= thresh) ? v * dscale : 0.f;
    }
    s += v;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    float t = 0.f;
    for (int q = 0; q < 8; ++q) t += sm[q * 32 + threadIdx.x];
    part[(long)blockIdx.x * 32 + threadIdx.x] = t;
  }
}

extern "C" int mo_gcn_mlp_bwd(const float* dh, const float* const* srcs, float* const* dsrcs, int ns,
                              const float* W, long P, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale,
                              float* dW, float* db, float* ws, void* dlast_bf16, int parts, int src_bf16_mask,
                              int dsrc_bf16_mask, void* stream) {
  MO_CHECK_ARG(dh && srcs && dsrcs && W && dW && db && ns >= 1 && ns <= MO_MAX_SEG && P > 0 && P < (1L << 31));
  MO_CHECK_ARG(ws || !(parts & 2));
  MO_CHECK_ARG(rs_mask_ok(src_bf16_mask, ns) && rs_mask_ok(dsrc_bf16_mask, ns) && !(dsrc_bf16_mask & 1));
  hipStream_t st = ST(stream);
  // data: dsrcs[s][p][c] = sum_co dm[p][co] W[co][s*32+c]
  MoOperand A = op_simple(dh, 32, P, 32);                 // XROWS rows = p, cols = k = co
  A.seg[0].drop_seed = drop_seed; A.seg[0].drop_thresh = drop_thresh; A.seg[0].drop_scale = drop_scale;
  MoOperand Bo = op_simple(W, 32 * ns, 32, 32 * ns);      // KROWS rows = k = co, cols = n
  MoEpi E; epi_init(E, dsrcs[0], 32);
  E.nout = ns; E.osegw = 32;
  E.out_bf = (unsigned short*)dlast_bf16; E.bf_seg = ns - 1;    // bf16 copy of the last source's gradient
  for (int s = 0; s < ns; ++s) { MO_CHECK_ARG(dsrcs[s] && srcs[s]); E.out[s] = dsrcs[s]; }
  int rc = MO_OK;
  if ((parts & 1) && rs_mlp_ok(&dh, 1, P) && rs_mlp_ok((const float* const*)dsrcs, ns, P)) {
    RsMlpArgs a = {};
    a.src[0] = dh; a.W = W; a.P = P; a.out_bf = (unsigned short*)dlast_bf16;
    for (int s = 0; s < ns; ++s) a.out[s] = dsrcs[s];
    a.drop_seed = drop_seed; a.drop_thresh = drop_thresh; a.drop_scale = drop_scale;
    rs_mlp_launch(false, ns, a, dsrc_bf16_mask != 0, st);
  } else if (parts & 1) {
    if (dsrc_bf16_mask) return MO_EUNSUPPORTED;
    if (ns == 1)
      rc = launch<128, 32, 32, 4, 1, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, 32, 1, st);
    else
      rc = launch<128, 128, 16, 2, 2, MO_XROWS, MO_KROWS, MO_EPI_STORE>(A, Bo, E, P, 32 * ns, 1, st);
    if (rc) return rc;
  }
  if (!(parts & 2)) return MO_OK;
  // weights: dW[co][s*32+c] = sum_p dm[p][co] srcs[s][p][c]
  MoOperand A2 = op_simple(dh, 32, P, 32);                // KROWS rows = k = p, cols = m = co
  A2.seg[0].drop_seed = drop_seed; A2.seg[0].drop_thresh = drop_thresh; A2.seg[0].drop_scale = drop_scale;
  MoOperand B2; op_init(B2);
  B2.nseg = ns; B2.segw = 32; B2.rows = (int)P; B2.cols = 32 * ns;
  for (int s = 0; s < ns; ++s) { seg_init(B2.seg[s], srcs[s], 32); B2.seg[s].bf16 = (src_bf16_mask >> s) & 1; }
  if (src_bf16_mask && !rsw_ok(A2, B2, P, 32, 32 * ns)) return MO_EUNSUPPORTED;
  bool done = false;
  rc = wgrad_run(A2, B2, P, 32, 32 * ns, ws, dW, db, st, &done, nullptr, src_bf16_mask != 0);
  if (rc) return rc;
  if (done) return MO_OK;
  // bias: db[co] = sum_p dm[p][co]  (dropout mask applies) -> reuse the GEMM with a ones column is
  // overkill; compute via colsum on dm materialised?  dm == dh when dropout is off; with dropout the
  // mask must be applied, so use a dedicated kernel below.
  int nb = mo_cdiv(P, CS_ROWS);
  hipLaunchKernelGGL(mlp_bias_grad_kernel, dim3(nb), dim3(256), 0, st, dh, P, drop_seed, drop_thresh, drop_scale, ws);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, st, ws, (long)32, nb, db, (long)32);
  return mo_launch_status();
}
// ------------------------------------------------------------------------------------------------
// loss + metrics
// ------------------------------------------------------------------------------------------------
#define MET_CHUNK 4096
extern "C" long mo_metrics_ws_floats(long n) { return (long)mo_cdiv(n, MET_CHUNK) * 4 + 16; }
__global__ void metrics_partial_kernel(const float* __restrict__ yh, const float* __restrict__ y, long n,
                                       float inv_n2, float* __restrict__ grad, float* __restrict__ part) {
  __shared__ float sm[3][256];
  const long i0 = (long)blockIdx.x * MET_CHUNK;
  long i1 = i0 + MET_CHUNK; if (i1 > n) i1 = n;
  float a = 0.f, b = 0.f, c = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = yh[i] - y[i];
    a += d * d; b += fabsf(d); c += fabsf(d) / fmaxf(fabsf(y[i]), 1.17e-06f);
    if (grad) grad[i] = d * inv_n2;
  }
  sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b; sm[2][threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
      sm[2][threadIdx.x] += sm[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) part[(long)blockIdx.x * 4 + threadIdx.x] = sm[threadIdx.x][0];
}
__global__ void metrics_final_kernel(const float* __restrict__ part, long nb, long n, float* __restrict__ sums) {
  __shared__ double sm[3][256];
  double a = 0, b = 0, c = 0;
  for (long i = threadIdx.x; i < nb; i += 256) { a += part[i * 4]; b += part[i * 4 + 1]; c += part[i * 4 + 2]; }
  sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b; sm[2][threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s]; sm[2][threadIdx.x] += sm[2][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x < 3) sums[threadIdx.x] = (float)sm[threadIdx.x][0];
  if (threadIdx.x == 3) sums[3] = (float)n;
}
extern "C" int mo_mse_metrics(const float* yhat, const float* y, long n, float* sums, float* grad, float* ws,
                              void* stream) {
  MO_CHECK_ARG(yhat && y && sums && ws && n > 0);
  long nb = mo_cdiv(n, MET_CHUNK);
  hipLaunchKernelGGL(metrics_partial_kernel, dim3(nb), dim3(256), 0, ST(stream), yhat, y, n, 2.0f / (float)n, grad, ws);
  hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(256), 0, ST(stream), ws, nb, n, sums);
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Date2Vec.encode
// ------------------------------------------------------------------------------------------------
__global__ void date2vec_kernel(const float* __restrict__ x, long n, const float* __restrict__ W1,
                                const float* __restrict__ b1, int k1, const float* __restrict__ W2,
                                const float* __restrict__ b2, int k2, float* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = k1 + k2;
  if (i >= n * kk) return;
  long r = i / kk; int j = (int)(i - r * kk);
  const float* xr = x + r * 6;
  if (j < k1) {
    float s = b1[j];
    for (int q = 0; q < 6; ++q) s += W1[j * 6 + q] * xr[q];
    out[i] = s;
  } else {
    int jj = j - k1;
    float s = b2[jj];
    for (int q = 0; q < 6; ++q) s += W2[jj * 6 + q] * xr[q];
    out[i] = sinf(s);
  }
}
extern "C" int mo_date2vec_encode(const float* x, long n, const float* W1, const float* b1, int k1,
                                  const float* W2, const float* b2, int k2, float* out, void* stream) {
  MO_CHECK_ARG(x && W1 && b1 && W2 && b2 && out && n > 0 && k1 > 0 && k2 > 0);
  long tot = n * (k1 + k2);
  hipLaunchKernelGGL(date2vec_kernel, dim3(mo_cdiv(tot, 256)), dim3(256), 0, ST(stream), x, n, W1, b1, k1, W2, b2, k2, out);
  return mo_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Adam on a flat parameter buffer (torch.optim.Adam semantics, lit.py:60)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2, float gscale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * mi / denom;
}
extern "C" int mo_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                            float beta2, float eps, float bias_c1, float bias_c2, float grad_scale, void* stream) {
  MO_CHECK_ARG(p && g && m && v && n > 0);
  hipLaunchKernelGGL(adam_kernel, dim3(mo_cdiv(n, 256)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2,
                     eps, bias_c1, bias_c2, grad_scale);
  return mo_launch_status();
}

extern "C" const char* mo_strerror(int code) {
  switch (code) {
    case MO_OK: return "ok";
    case MO_EINVAL: return "invalid argument";
    case MO_ELAUNCH: return "HIP launch/runtime error";
    case MO_EUNSUPPORTED: return "unsupported configuration";
    case MO_ECOMM: return "RCCL error";
    default: return "unknown error";
  }
}
extern "C" int mo_version(void) { return MO_ABI_VERSION; }
