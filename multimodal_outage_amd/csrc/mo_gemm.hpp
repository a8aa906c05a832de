// Tiled fp32 MFMA contraction engine for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma).
//
// One kernel template computes  D[m][n] = epilogue( sum_k A(m,k) * B(k,n) )  where each operand is a
// "virtual matrix" described at run time (MoOperand): a concatenation of column segments, each a
// row-major [rows][ld] fp32 array with an optional row map (time-crop / dilation tap / left pad),
// per-column affine (folded BatchNorm), ReLU and regenerated dropout mask applied on load.
// This is what lets the 1x1 convs, the dilated gated TCN, the gcn mlp over the never-materialised
// channel concat, their data/weight gradients and the dense adaptive-adjacency products all run on
// the same LDS-staged MFMA loop.
//
// Operand modes (which index enumerates the source rows):
//   KROWS: source rows = k, source columns = m (or n): tile copy is a straight row copy
//   XROWS: source rows = m (or n), source columns = k: tile copy transposes through LDS
// LDS tiles are always [BK][BX+pad] so that the MFMA fragment read (lane i=l&31, kk=l>>5 reads
// T[k+kk][x0+i]) is a conflict-free ds_read_b32.
#pragma once
#include "mo_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MO_MAX_SEG 12     // column segments of a virtual operand: 1 + 2 x 5 supports + 1 (graph_wavenet.py:79: (order*support_len+1)*c_in)
#define MO_KROWS 0
#define MO_XROWS 1

struct MoSeg {
  const float* ptr;
  const float* scale;  // optional per-column affine (index = column within segment)
  const float* shift;
  int ld;
  int To, Ti, off;     // row map r -> (g=r/To, t=r%To+off) -> g*Ti+t, valid iff 0<=t<Ti; To==0: identity
  int relu;
  uint32_t drop_seed, drop_thresh;  // drop_thresh==0: no dropout; element index = srow*ld + col
  float drop_scale;
  int bf16;            // 1: the segment is stored as bf16 (row-streaming kernels only; the tile engine rejects it)
};

struct MoOperand {
  MoSeg seg[MO_MAX_SEG];
  int nseg;   // segments concatenated along the column direction
  int segw;   // columns per segment (nseg>1); unused for nseg==1
  int rows;   // logical extent along the row direction
  int cols;   // logical extent along the column direction
};

enum { MO_EPI_STORE = 0, MO_EPI_GATE = 1, MO_EPI_GATE_BWD = 2, MO_EPI_MLP = 3, MO_EPI_NCHW = 4, MO_EPI_CONVT = 5 };

// Source kinds of an operand (what (row, col) of the virtual matrix means):
//   PLAIN : row-major segments as described by MoSeg (channels-last activations, weights)
//   IM2COL: rows q=(ci,tap) of a 3x3/pad-1 im2col matrix over NCHW images, cols p=(img,y,x);
//           up to two channel segments (the skip/up concat of unet.py:83) with a per-(group,channel)
//           folded BatchNorm affine + ReLU applied on load (seg.ld = image stride in floats)
//   NCHW  : rows = channel, cols p=(img,pix) of an NCHW tensor (same on-load activation)
//   CONVT : rows r=(co,ky,kx), cols p=(img,y,x) of the 2x up-sampled NCHW tensor gathered at
//           (2y+ky, 2x+kx)  (ConvTranspose2d k=2,s=2 backward, unet.py:71)
enum { MO_SRC_PLAIN = 0, MO_SRC_IM2COL = 1, MO_SRC_NCHW = 2, MO_SRC_CONVT = 3 };

struct MoGeom {
  int H, W, HW;     // spatial size of the (low-resolution, for CONVT) pixel grid
  int gsize;        // images per BatchNorm group (the reference normalises per county call: `horizon` images)
  int C0, C1;       // channels of segment 0 / 1
  int lw, lhw;      // log2(W), log2(HW) when powers of two (shift/mask pixel decomposition), else -1
};

struct MoEpi {
  float* out[MO_MAX_SEG];  // outputs, segmented along n when nout>1 (n -> out[n/osegw][.., n%osegw])
  int nout, osegw;
  int ldo;
  int oTo, oTi, ooff;      // output row map (oTo==0: identity); unmapped rows are skipped
  const float* bias;       // per n
  const float* bias2;      // GATE*: gate bias
  int relu;
  int beta;                // 1: accumulate into out
  const float* mask;       // multiply by (mask[m*ldmask+n] > 0)   (ReLU backward)
  int ldmask;
  const float* add;        // add source with its own row map and affine (residual / crop-add)
  int ldadd, aTo, aTi, aoff;
  const float* ascale;
  const float* ashift;
  const float* aux;        // GATE_BWD: upstream gradient dg[m][32]
  int ldaux;
  uint32_t drop_seed, drop_thresh;
  float drop_scale;
  float* partial;          // MLP: per-block BatchNorm partial sums [gridDim.x][64] (sum | sumsq)
  long slab_stride;        // split-K: slab z goes to out + z*slab_stride
  int kchunk;              // split-K chunk (multiple of BK); 0: whole K
  float* colsum;           // FAST, A in KROWS mode: per-slice column sums of A, [gridDim.z][M] (bias gradient)
  unsigned short* out_bf;  // optional bf16 copy: GATE: of g; STORE: of output segment bf_seg (same row stride)
  int bf_seg;
};

// Loads through descriptor pointers (which travel through LDS and lose their address space) must be
// emitted as global_load, not flat_load: flat loads also count on lgkmcnt, so every LDS wait of the MFMA
// phase would stall on the prefetch of the next tile.
typedef float mo_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float mo_gload(const float* p) {
  return *(const __attribute__((address_space(1))) float*)(uintptr_t)p;
}
__device__ __forceinline__ float4 mo_gload(const float4* p) {
  const mo_v4f v = *(const __attribute__((address_space(1))) mo_v4f*)(uintptr_t)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ bool mo_seg_row(const MoSeg& s, int r, long& srow) {
  if (s.To == 0) { srow = r; return true; }
  int g = r / s.To;
  int t = r - g * s.To + s.off;
  srow = (long)g * s.Ti + t;
  return (t >= 0) && (t < s.Ti);
}

__device__ __forceinline__ float mo_post(const MoSeg& s, float v, int cc, long srow) {
  if (s.scale) v = v * mo_gload(s.scale + cc) + mo_gload(s.shift + cc);
  if (s.relu) v = fmaxf(v, 0.f);
  if (s.drop_thresh) {
    uint32_t h = mo_hash32(s.drop_seed, (uint32_t)(srow * s.ld + cc));
    v = (h >= s.drop_thresh) ? v * s.drop_scale : 0.f;
  }
  return v;
}

// Four consecutive columns c..c+3 of logical row r (zero where out of range / unmapped).
__device__ __forceinline__ float4 mo_fetch4(const MoSeg* segs, int nseg, int segw, int rows, int cols,
                                            int r, int c) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r >= rows || c >= cols) return v;
  int s = 0, cc = c, w = cols;
  if (nseg > 1) { s = c / segw; cc = c - s * segw; w = segw; }
  const MoSeg& sg = segs[s];
  long srow;
  if (!mo_seg_row(sg, r, srow)) return v;
  const float* p = sg.ptr + srow * (long)sg.ld + cc;
  bool vec = ((sg.ld & 3) == 0) && ((((uintptr_t)sg.ptr) & 15) == 0) && (cc + 3 < w);
  if (vec) {
    v = mo_gload(reinterpret_cast<const float4*>(p));
  } else {
    if (cc + 0 < w) v.x = mo_gload(p + 0);
    if (cc + 1 < w) v.y = mo_gload(p + 1);
    if (cc + 2 < w) v.z = mo_gload(p + 2);
    if (cc + 3 < w) v.w = mo_gload(p + 3);
  }
  if (sg.scale || sg.relu || sg.drop_thresh) {
    if (cc + 0 < w) v.x = mo_post(sg, v.x, cc + 0, srow);
    if (cc + 1 < w) v.y = mo_post(sg, v.y, cc + 1, srow);
    if (cc + 2 < w) v.z = mo_post(sg, v.z, cc + 2, srow);
    if (cc + 3 < w) v.w = mo_post(sg, v.w, cc + 3, srow);
  }
  return v;
}

// The three image-source fetchers are branch-free: indices are clamped, every load is issued
// unconditionally and the result is masked, so the NV fetches of a tile stay in flight together.
// p -> (img, y, x): shifts and masks for power-of-two image sizes (all UNet levels), division otherwise
__device__ __forceinline__ void mo_pix(const MoGeom& g, int p, int& img, int& y, int& x) {
  if (g.lhw >= 0) {
    img = p >> g.lhw;
    const int pix = p & (g.HW - 1);
    y = pix >> g.lw; x = pix & (g.W - 1);
  } else {
    img = p / g.HW;
    const int pix = p - img * g.HW;
    y = pix / g.W; x = pix - y * g.W;
  }
}

__device__ __forceinline__ float4 mo_fetch4_im2col(const MoSeg* segs, const MoGeom& g, int rows, int cols, int q,
                                                   int p) {
  const bool inb = (q < rows) & (p < cols);
  const int qq = inb ? q : 0, pp = inb ? p : 0;
  const int ci = qq / 9, tap = qq - ci * 9;
  const int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
  const int s = (ci >= g.C0) ? 1 : 0;
  const int c = ci - (s ? g.C0 : 0);
  const int Cs = s ? g.C1 : g.C0;
  const MoSeg& sg = segs[s];
  const float* bptr = sg.ptr;
  const float* scp = sg.scale;
  const float* shp = sg.shift;
  const int sld = sg.ld, srelu = sg.relu;
  int img, y, x;
  mo_pix(g, pp, img, y, x);
  const int yy = y + ky;
  const bool rowok = inb & (yy >= 0) & (yy < g.H);
  const int yyc = min(max(yy, 0), g.H - 1);
  const float* base = bptr + (long)img * sld + (long)c * g.HW + yyc * g.W;
  const bool aff = scp != nullptr;
  const int grp = img / g.gsize;
  const float sc = aff ? mo_gload((aff ? scp : bptr) + (aff ? grp * Cs + c : 0)) : 1.f;
  const float sh = aff ? mo_gload((aff ? shp : bptr) + (aff ? grp * Cs + c : 0)) : 0.f;
  float t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int xx = x + j + kx;
    const int xc = min(max(xx, 0), g.W - 1);
    float u = mo_gload(base + xc) * sc + sh;
    u = srelu ? fmaxf(u, 0.f) : u;
    t[j] = (rowok & (xx >= 0) & (xx < g.W)) ? u : 0.f;
  }
  return make_float4(t[0], t[1], t[2], t[3]);
}

__device__ __forceinline__ float4 mo_fetch4_nchw(const MoSeg* segs, const MoGeom& g, int rows, int cols, int c,
                                                 int p) {
  const bool inb = (c < rows) & (p < cols);
  const int cq = inb ? c : 0, pp = inb ? p : 0;
  const MoSeg& sg = segs[0];
  const float* bptr = sg.ptr;
  const float* scp = sg.scale;
  const float* shp = sg.shift;
  const int sld = sg.ld, srelu = sg.relu;
  const int img = (g.lhw >= 0) ? (pp >> g.lhw) : (pp / g.HW);
  const int pix = pp - img * g.HW;
  float4 v = mo_gload(reinterpret_cast<const float4*>(bptr + (long)img * sld + (long)cq * g.HW + pix));
  const bool aff = scp != nullptr;
  const int grp = img / g.gsize;
  const float sc = aff ? mo_gload((aff ? scp : bptr) + (aff ? grp * g.C0 + cq : 0)) : 1.f;
  const float sh = aff ? mo_gload((aff ? shp : bptr) + (aff ? grp * g.C0 + cq : 0)) : 0.f;
  v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
  if (srelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
  return v;
}

__device__ __forceinline__ float4 mo_fetch4_convt(const MoSeg* segs, const MoGeom& g, int rows, int cols, int r,
                                                  int p) {
  const bool inb = (r < rows) & (p < cols);
  const int rq = inb ? r : 0, pp = inb ? p : 0;
  const MoSeg& sg = segs[0];
  const float* bptr = sg.ptr;
  const int sld = sg.ld;
  const int co = rq >> 2, ky = (rq >> 1) & 1, kx = rq & 1;
  int img, y, x;
  mo_pix(g, pp, img, y, x);
  const float* base = bptr + (long)img * sld + (long)co * 4 * g.HW + (long)(2 * y + ky) * 2 * g.W + 2 * x + kx;
  float4 v;
  v.x = mo_gload(base); v.y = mo_gload(base + 2); v.z = mo_gload(base + 4); v.w = mo_gload(base + 6);
  if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
  return v;
}

template <int SRC>
__device__ __forceinline__ float4 mo_fetch(const MoSeg* segs, int nseg, int segw, int rows, int cols,
                                           const MoGeom& g, int r, int c) {
  if (SRC == MO_SRC_IM2COL) return mo_fetch4_im2col(segs, g, rows, cols, r, c);
  if (SRC == MO_SRC_NCHW) return mo_fetch4_nchw(segs, g, rows, cols, r, c);
  if (SRC == MO_SRC_CONVT) return mo_fetch4_convt(segs, g, rows, cols, r, c);
  return mo_fetch4(segs, nseg, segw, rows, cols, r, c);
}

// Tile loader.  FAST (plain sources whose segments are all 16-byte vector-loadable; checked on the host):
// branch-free "issue" (address generation + unconditional clamped float4 loads, all in flight together)
// and "finish" (zero-fill, folded affine / ReLU / dropout, LDS store) run on either side of the MFMA
// phase, so a wave keeps NV independent loads outstanding instead of waiting for each one in turn.
// Otherwise (odd leading dimensions, im2col / NCHW / convT-gather sources): generic element fetch.
template <int BX, int BK, int NT, int MODE, int SRC, int FAST>
struct MoLoader {
  static constexpr int PAD = (MODE == MO_KROWS) ? 4 : 1;
  static constexpr int LD = BX + PAD;
  static constexpr int NV = (BK * BX / 4) / NT;  // float4 per thread per tile
  static_assert((BK * BX / 4) % NT == 0, "tile/thread mismatch");

  float4 reg[NV];
  int meta[NV];        // FAST: (seg << 8 | col-in-seg) or -1 when the element is out of range / unmapped
  uint32_t didx[NV];   // FAST: dropout element index (srow * ld + col)
  int rg[NV], rt[NV];  // FAST + XROWS: hoisted row decomposition (rows are fixed over the k loop)
  float4 cs[NV];       // FAST + KROWS: running column sums of the tile (fused bias gradient), post bit 3

  __device__ __forceinline__ void init(const MoOperand& op, int To, int x0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FAST && MODE == MO_XROWS) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int r = x0 + (tid + i * NT) / (BK / 4);
        const int g = r / To;
        rg[i] = g; rt[i] = r - g * To;
      }
    }
  }

  __device__ __forceinline__ void issue(const MoSeg* segs, const MoOperand& op, int To, int segshift,
                                        const MoGeom& G, int x0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      int r, c;
      if (MODE == MO_KROWS) { r = k0 + f / (BX / 4); c = x0 + 4 * (f % (BX / 4)); }
      else { r = x0 + f / (BK / 4); c = k0 + 4 * (f % (BK / 4)); }
      if (FAST) {
        int g, t;
        if (MODE == MO_XROWS) { g = rg[i]; t = rt[i]; }
        else { g = r / To; t = r - g * To; }
        // columns past the last segment (tile padding) are clamped onto it and masked by `valid`
        const int s = (op.nseg > 1) ? min(c >> segshift, op.nseg - 1) : 0;
        const int cc = (op.nseg > 1) ? (c & ((1 << segshift) - 1)) : c;
        const MoSeg& sg = segs[s];
        const float* bptr = sg.ptr;                 // read every descriptor field unconditionally:
        const int sld = sg.ld, sTi = sg.Ti, soff = sg.off;   // no short-circuit => straight-line code
        const int tt = t + soff;
        const bool valid = (r < op.rows) & (c < op.cols) & ((unsigned)tt < (unsigned)sTi);
        const long srow = (long)g * sTi + tt;
        const long eoff = valid ? srow * (long)sld + cc : 0;
        reg[i] = mo_gload(reinterpret_cast<const float4*>(bptr + eoff));
        meta[i] = valid ? ((s << 8) | cc) : -1;
        didx[i] = (uint32_t)eoff;
      } else {
        reg[i] = mo_fetch<SRC>(segs, op.nseg, op.segw, op.rows, op.cols, G, r, c);
      }
    }
  }

  // aff: LDS table [seg][3][32]: scale, shift (identity when absent), unused
  __device__ __forceinline__ void finish(float* T, const float* aff, const MoSeg* segs, int post, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      float4 v = reg[i];
      if (FAST) {
        const int m = meta[i];
        if (post & 1) {       // folded BatchNorm affine (32-wide segments)
          const int mm = m < 0 ? 0 : m;
          const int s = (mm >> 8) & 7, cc = mm & 31;
          const float4 sc = *reinterpret_cast<const float4*>(&aff[(s * 2 + 0) * 32 + cc]);
          const float4 sh = *reinterpret_cast<const float4*>(&aff[(s * 2 + 1) * 32 + cc]);
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        }
        if (post & 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (post & 4) {       // regenerated dropout mask (single-segment operands)
          const MoSeg& sg = segs[0];
          float* vp = &v.x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t h = mo_hash32(sg.drop_seed, didx[i] + q);
            vp[q] = (h >= sg.drop_thresh) ? vp[q] * sg.drop_scale : 0.f;
          }
        }
        if (m < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == MO_KROWS && (post & 8)) { cs[i].x += v.x; cs[i].y += v.y; cs[i].z += v.z; cs[i].w += v.w; }
      }
      if (MODE == MO_KROWS) {
        const int k = f / (BX / 4), x4 = f % (BX / 4);
        *reinterpret_cast<float4*>(&T[k * LD + 4 * x4]) = v;
      } else {
        const int x = f / (BK / 4), k4 = f % (BK / 4);
        T[(4 * k4 + 0) * LD + x] = v.x;
        T[(4 * k4 + 1) * LD + x] = v.y;
        T[(4 * k4 + 2) * LD + x] = v.z;
        T[(4 * k4 + 3) * LD + x] = v.w;
      }
    }
  }
};

// per-operand LDS affine table [seg][2][32] (scale | shift); identity for segments without an affine
__device__ __forceinline__ void mo_fill_aff(float* aff, const MoOperand& op, int tid, int nthreads) {
  for (int i = tid; i < MO_MAX_SEG * 64; i += nthreads) {
    const int s = i >> 6, which = (i >> 5) & 1, c = i & 31;
    float v = which ? 0.f : 1.f;
    if (s < op.nseg && op.seg[s].scale) v = which ? op.seg[s].shift[c] : op.seg[s].scale[c];
    aff[i] = v;
  }
}

// sigmoid / tanh on the hardware exp and reciprocal (v_exp_f32, v_rcp_f32): ~1e-6 relative error, and an
// order of magnitude fewer VALU instructions than the libm tanhf, which made the gate epilogues VALU-bound.
// (__builtin_amdgcn_rcpf IS v_rcp_f32, 1 ulp; __frcp_rn, used here until the end of round 3, is the correctly rounded
//  reciprocal -- the whole IEEE division sequence, v_div_scale x2 + v_rcp + 4 fma + v_div_fmas + v_div_fixup: a third of
//  the gated-TCN kernel's VALU instructions, in a kernel whose counters say VALU-bound.)
__device__ __forceinline__ float mo_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float mo_tanh(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

template <int TM, int TN, int WM, int WN, int AMODE, int BMODE, int EPI>
__device__ __forceinline__ void mo_epilogue(f32x16 (&acc)[TM][TN], const MoOperand& A, const MoOperand& B, const MoEpi& E,
                                            const MoGeom& G, int m0, int n0, int wm0, int wn0, int tile_x,
                                            float* red) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int fi = lane & 31, fk = lane >> 5;
  // ---------------------------------------------------------------- epilogue
  const int M = (AMODE == MO_KROWS) ? A.cols : A.rows;
  const int N = (BMODE == MO_KROWS) ? B.cols : B.rows;

  if (EPI == MO_EPI_STORE) {
    float* const obase_off = nullptr;
    (void)obase_off;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + fi;
        if (n >= N) continue;
        int os = 0, on = n;
        if (E.nout > 1) { os = n / E.osegw; on = n - os * E.osegw; }
        float* obase = E.out[os] + (long)blockIdx.z * E.slab_stride;
        const float bv = E.bias ? E.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
          if (m >= M) continue;
          float v = acc[i][j][r] + bv;
          if (E.add) {
            int g = m, t = 0; long arow = m; bool ok = true;
            if (E.aTo) { g = m / E.aTo; t = m - g * E.aTo + E.aoff; arow = (long)g * E.aTi + t; ok = (t >= 0) && (t < E.aTi); }
            if (ok) {
              float av = E.add[arow * E.ldadd + n];
              if (E.ascale) av = av * E.ascale[n] + E.ashift[n];
              v += av;
            }
          }
          if (E.relu) v = fmaxf(v, 0.f);
          if (E.mask) v = (E.mask[(long)m * E.ldmask + n] > 0.f) ? v : 0.f;
          long orow = m;
          if (E.oTo) {
            int g = m / E.oTo; int t = m - g * E.oTo + E.ooff;
            if (t < 0 || t >= E.oTi) continue;
            orow = (long)g * E.oTi + t;
          }
          float* o = obase + orow * E.ldo + on;
          if (E.beta) v += *o;
          *o = v;
          if (E.out_bf && os == E.bf_seg) {
            __bf16 tb = (__bf16)v;
            E.out_bf[orow * E.ldo + on] = __builtin_bit_cast(unsigned short, tb);
          }
        }
      }
    }
  } else if (EPI == MO_EPI_GATE || EPI == MO_EPI_GATE_BWD) {
    // TN == 2: acc[i][0] = filter pre-activation, acc[i][1] = gate pre-activation, same (row, col)
    const int c = fi;  // channel 0..31
    const float bf = E.bias[c], bg = E.bias2[c];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (m >= M) continue;
        const float f = mo_tanh(acc[i][0][r] + bf);
        const float g = mo_sigmoid(acc[i][TN - 1][r] + bg);
        if (EPI == MO_EPI_GATE) {
          E.out[0][(long)m * E.ldo + c] = f * g;
          if (E.out_bf) { __bf16 tb = (__bf16)(f * g); E.out_bf[(long)m * E.ldo + c] = __builtin_bit_cast(unsigned short, tb); }
        } else {
          const float dg = E.aux[(long)m * E.ldaux + c];
          E.out[0][(long)m * E.ldo + c] = dg * g * (1.f - f * f);
          E.out[0][(long)m * E.ldo + 32 + c] = dg * f * g * (1.f - g);
        }
      }
    }
  } else if (EPI == MO_EPI_NCHW || EPI == MO_EPI_CONVT) {
    // rows m = channel (or (co,ky,kx) for CONVT), cols n = pixel p=(img,y,x): NCHW store, coalesced along n
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int pcol = n0 + wn0 + j * 32 + fi;
      if (pcol >= N) continue;
      const int img = (G.lhw >= 0) ? (pcol >> G.lhw) : (pcol / G.HW);
      const int pix = pcol - img * G.HW;
      float* obase = E.out[0] + (long)img * E.ldo;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
          if (m >= M) continue;
          float v = acc[i][j][r];
          if (EPI == MO_EPI_NCHW) {
            if (E.bias) v += E.bias[m];
            if (E.relu) v = fmaxf(v, 0.f);
            float* o = obase + (long)(E.ooff + m) * G.HW + pix;
            if (E.beta) v += *o;
            *o = v;
          } else {
            const int co = m >> 2, ky = (m >> 1) & 1, kx = m & 1;
            const int y = pix / G.W, x = pix - y * G.W;
            if (E.bias) v += E.bias[co];
            obase[(long)(E.ooff + co) * 4 * G.HW + (long)(2 * y + ky) * 2 * G.W + 2 * x + kx] = v;
          }
        }
      }
    }
  } else if (EPI == MO_EPI_MLP) {
    // BN == 32 (TN == 1, WN == 1): bias + dropout + residual(affine, row map) -> h; BN partial sums.
    const int n = fi;
    const float bv = E.bias[n];
    const float asc = E.ascale ? E.ascale[n] : 1.f;
    const float ash = E.ashift ? E.ashift[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (m >= M) continue;
        float v = acc[i][0][r] + bv;
        if (E.drop_thresh) {
          uint32_t h = mo_hash32(E.drop_seed, (uint32_t)((long)m * E.ldo + n));
          v = (h >= E.drop_thresh) ? v * E.drop_scale : 0.f;
        }
        int g = m / E.aTo; int t = m - g * E.aTo + E.aoff;
        const float rv = E.add[((long)g * E.aTi + t) * E.ldadd + n] * asc + ash;
        v += rv;
        E.out[0][(long)m * E.ldo + n] = v;
        s1 += v;
        s2 += v * v;
      }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) { red[wave * 64 + lane] = s1; red[wave * 64 + 32 + lane] = s2; }
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM * WN; ++w) t += red[w * 64 + tid];
      E.partial[(long)tile_x * 64 + tid] = t;
    }
  }
}

// NBUF = 2: double-buffered LDS, one barrier per k-tile.  NBUF = 1: a single LDS tile pair and two barriers
// per k-tile -- half the LDS, so more workgroups per CU; for the skinny-K, latency-bound contractions the
// bytes in flight per CU (workgroups x one tile each) matter more than the extra barrier.
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, int EPI, int ASRC = MO_SRC_PLAIN,
          int BSRC = MO_SRC_PLAIN, int FAST = 0, int NBUF = 2>
__global__ void __launch_bounds__(WM* WN * 64)
mo_gemm_kernel(const MoOperand A, const MoOperand B, const MoEpi E, const MoGeom G) {
  constexpr int NT = WM * WN * 64;
  constexpr int SM = BM / WM, SN = BN / WN;
  constexpr int TM = SM / 32, TN = SN / 32;
  static_assert(SM % 32 == 0 && SN % 32 == 0, "wave tile must be a multiple of 32x32");
  using TA = MoLoader<BM, BK, NT, AMODE, ASRC, FAST>;
  using TB = MoLoader<BN, BK, NT, BMODE, BSRC, FAST>;

  __shared__ __attribute__((aligned(16))) float As[NBUF][BK * TA::LD];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK * TB::LD];
  __shared__ MoSeg sA[MO_MAX_SEG];
  __shared__ MoSeg sB[MO_MAX_SEG];
  __shared__ float red[(EPI == MO_EPI_MLP) ? WM * WN * 64 : 1];
  __shared__ __attribute__((aligned(16))) float affA[FAST ? MO_MAX_SEG * 64 : 4];
  __shared__ __attribute__((aligned(16))) float affB[FAST ? MO_MAX_SEG * 64 : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WN) * SM;
  const int wn0 = (wave % WN) * SN;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // stage operand descriptors in LDS (segment index may vary per lane)
  if (tid < MO_MAX_SEG) { sA[tid] = A.seg[tid]; sB[tid] = B.seg[tid]; }
  // FAST path, uniform per operand: bit0 affine present, bit1 ReLU, bit2 dropout
  int postA = 0, postB = 0;
  if (FAST) {
    for (int q = 0; q < MO_MAX_SEG; ++q) {
      if (q < A.nseg && A.seg[q].scale) postA |= 1;
      if (q < B.nseg && B.seg[q].scale) postB |= 1;
    }
    if (A.seg[0].relu) postA |= 2;
    if (B.seg[0].relu) postB |= 2;
    if (A.seg[0].drop_thresh) postA |= 4;
    if (AMODE == MO_KROWS && E.colsum && blockIdx.y == 0) postA |= 8;
    if (B.seg[0].drop_thresh) postB |= 4;
    if (postA & 1) mo_fill_aff(affA, A, tid, NT);
    if (postB & 1) mo_fill_aff(affB, B, tid, NT);
  }
  const int ToA = A.seg[0].To > 0 ? A.seg[0].To : 1, ToB = B.seg[0].To > 0 ? B.seg[0].To : 1;
  const int shA = (A.segw == 64) ? 6 : 5, shB = (B.segw == 64) ? 6 : 5;
  __syncthreads();

  // logical extents: M, N from the x-direction of each operand, K from the k-direction
  const int K = (AMODE == MO_KROWS) ? A.rows : A.cols;
  int kbeg = 0, kend = K;
  if (E.kchunk > 0) {
    kbeg = blockIdx.z * E.kchunk;
    kend = min(K, kbeg + E.kchunk);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TA la;
  TB lb;
  la.init(A, ToA, m0, tid);
  lb.init(B, ToB, n0, tid);
  const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  // Split-K chunks end on BK multiples except the last, which ends at K == rows/cols, so the operand's
  // own extent check bounds the k direction.
  if (nk > 0) {
    la.issue(sA, A, ToA, shA, G, m0, kbeg, tid);
    lb.issue(sB, B, ToB, shB, G, n0, kbeg, tid);
    la.finish(As[0], affA, sA, postA, tid);
    lb.finish(Bs[0], affB, sB, postB, tid);
  }
  __syncthreads();

  const int fi = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = (NBUF == 2) ? (kt & 1) : 0;
    if (kt + 1 < nk) {
      la.issue(sA, A, ToA, shA, G, m0, kbeg + (kt + 1) * BK, tid);
      lb.issue(sB, B, ToB, shB, G, n0, kbeg + (kt + 1) * BK, tid);
    }
    const float* Ac = As[cur];
    const float* Bc = Bs[cur];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ac[(2 * ks + fk) * TA::LD + wm0 + i * 32 + fi];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bc[(2 * ks + fk) * TB::LD + wn0 + j * 32 + fi];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (NBUF == 1) __syncthreads();      // everyone is done reading the single tile pair
    if (kt + 1 < nk) {
      la.finish(As[(NBUF == 2) ? (cur ^ 1) : 0], affA, sA, postA, tid);
      lb.finish(Bs[(NBUF == 2) ? (cur ^ 1) : 0], affB, sB, postB, tid);
    }
    __syncthreads();
  }

  if (FAST && AMODE == MO_KROWS && (postA & 8)) {
    // column sums of A over this block's K range: threads with equal x4 hold partial sums of the same
    // four columns; reduce the NT/(BM/4) k-lanes through LDS (the tile buffers are free now)
    constexpr int XG = BM / 4, KL = NT / XG;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TA::NV; ++i) { t.x += la.cs[i].x; t.y += la.cs[i].y; t.z += la.cs[i].z; t.w += la.cs[i].w; }
    float* red2 = &As[0][0];
    const int x4 = tid % XG, kl = tid / XG;
    *reinterpret_cast<float4*>(&red2[kl * BM + 4 * x4]) = t;
    __syncthreads();
    if (tid < BM) {
      float sacc = 0.f;
#pragma unroll 4
      for (int q = 0; q < KL; ++q) sacc += red2[q * BM + tid];
      const int Mtot = A.cols;
      if (m0 + tid < Mtot) E.colsum[(long)blockIdx.z * Mtot + m0 + tid] = sacc;
    }
    __syncthreads();
  }

  mo_epilogue<TM, TN, WM, WN, AMODE, BMODE, EPI>(acc, A, B, E, G, m0, n0, wm0, wn0, (int)blockIdx.x, red);
}

// ------------------------------------------------------------------------------------------------
// Persistent variant for the skinny-K channel contractions (1x1 convs, gated TCN, gcn mlp, their data
// gradients): K <= NKRES*BK, so the whole B operand (weights) stays resident in LDS and each workgroup
// walks M tiles grid-stride with the A prefetch running across tile boundaries.  The plain kernel pays
// its prologue (descriptor staging, affine table, first tile round trip) once per 128 rows, which for
// K = 32..224 is most of its life; here it is paid once per workgroup and the loads never drain.
// A is XROWS (rows = positions) and both operands use the FAST loader (checked on the host).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WM, int WN, int BMODE, int EPI, int NKRES>
__global__ void __launch_bounds__(WM* WN * 64)
mo_conv_persist_kernel(const MoOperand A, const MoOperand B, const MoEpi E, const MoGeom G, int num_mtiles) {
  constexpr int NT = WM * WN * 64;
  constexpr int SM = BM / WM, SN = BN / WN;
  constexpr int TM = SM / 32, TN = SN / 32;
  using TA = MoLoader<BM, BK, NT, MO_XROWS, MO_SRC_PLAIN, 1>;
  using TB = MoLoader<BN, BK, NT, BMODE, MO_SRC_PLAIN, 1>;

  __shared__ __attribute__((aligned(16))) float As[2][BK * TA::LD];
  __shared__ __attribute__((aligned(16))) float Bres[NKRES][BK * TB::LD];
  __shared__ MoSeg sA[MO_MAX_SEG];
  __shared__ MoSeg sB[MO_MAX_SEG];
  __shared__ float red[(EPI == MO_EPI_MLP) ? WM * WN * 64 : 1];
  __shared__ __attribute__((aligned(16))) float affA[MO_MAX_SEG * 64];
  __shared__ __attribute__((aligned(16))) float affB[MO_MAX_SEG * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WN) * SM;
  const int wn0 = (wave % WN) * SN;
  const int n0 = blockIdx.y * BN;

  if (tid < MO_MAX_SEG) { sA[tid] = A.seg[tid]; sB[tid] = B.seg[tid]; }
  int postA = 0, postB = 0;
  for (int q = 0; q < MO_MAX_SEG; ++q) {
    if (q < A.nseg && A.seg[q].scale) postA |= 1;
    if (q < B.nseg && B.seg[q].scale) postB |= 1;
  }
  if (A.seg[0].relu) postA |= 2;
  if (B.seg[0].relu) postB |= 2;
  if (A.seg[0].drop_thresh) postA |= 4;
  if (B.seg[0].drop_thresh) postB |= 4;
  if (postA & 1) mo_fill_aff(affA, A, tid, NT);
  if (postB & 1) mo_fill_aff(affB, B, tid, NT);
  const int ToA = A.seg[0].To > 0 ? A.seg[0].To : 1, ToB = B.seg[0].To > 0 ? B.seg[0].To : 1;
  const int shA = (A.segw == 64) ? 6 : 5, shB = (B.segw == 64) ? 6 : 5;
  __syncthreads();

  const int K = A.cols;
  const int nk = (K + BK - 1) / BK;     // <= NKRES (host)
  TA la;
  TB lb;
  lb.init(B, ToB, n0, tid);
  for (int kt = 0; kt < nk; ++kt) {     // resident B: loaded once per workgroup
    lb.issue(sB, B, ToB, shB, G, n0, kt * BK, tid);
    lb.finish(Bres[kt], affB, sB, postB, tid);
  }
  int mt = blockIdx.x;
  if (mt < num_mtiles) {
    la.init(A, ToA, mt * BM, tid);
    la.issue(sA, A, ToA, shA, G, mt * BM, 0, tid);
    la.finish(As[0], affA, sA, postA, tid);
  }
  __syncthreads();

  const int fi = lane & 31, fk = lane >> 5;
  int it = 0;
  for (; mt < num_mtiles; mt += gridDim.x) {
    const int m0 = mt * BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++it) {
      const int cur = it & 1;
      const bool last_k = (kt + 1 == nk);
      const int nmt = mt + gridDim.x;
      const bool has_next = !last_k || (nmt < num_mtiles);
      if (has_next) {
        if (last_k) {                   // first k-tile of this workgroup's next M tile
          la.init(A, ToA, nmt * BM, tid);
          la.issue(sA, A, ToA, shA, G, nmt * BM, 0, tid);
        } else {
          la.issue(sA, A, ToA, shA, G, m0, (kt + 1) * BK, tid);
        }
      }
      const float* Ac = As[cur];
      const float* Bc = Bres[kt];
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Ac[(2 * ks + fk) * TA::LD + wm0 + i * 32 + fi];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bc[(2 * ks + fk) * TB::LD + wn0 + j * 32 + fi];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (has_next) la.finish(As[cur ^ 1], affA, sA, postA, tid);
      __syncthreads();
    }
    mo_epilogue<TM, TN, WM, WN, MO_XROWS, BMODE, EPI>(acc, A, B, E, G, m0, n0, wm0, wn0, mt, red);
  }
}
