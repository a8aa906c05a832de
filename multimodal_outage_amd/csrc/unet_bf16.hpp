// 3x3 convolutions of the UNet on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulation) -- the arithmetic of
// the "bf16 mode" of BASELINE config 3 (MO_BF_MATH in `dtypes`): activations and weights enter the MFMA rounded to bf16
// (round to nearest even), every sum is fp32.  The fp32 kernels of unet_direct.hpp spend their time on the VALU (thin
// layers, 20..40 % of the 78 TFLOP/s scalar-FMA peak) or on a 4x-padded fp32 MFMA; on this pipe the arithmetic of a thin
// conv is a few percent of its memory time, so the kernels below are organised around the staging of the tile.
//
// Forward / data gradient (ub_conv3x3_kernel):  D[pixel][co] += A[pixel][k] * B[k][co],  k = (tap, ci)
//   * the activated input tile (folded BatchNorm affine + ReLU on the way in, zero padding AFTER activation) lies in LDS
//     CHANNELS-LAST as bf16: [row][col][CP], so the 8 consecutive k a lane feeds to one MFMA are 16 contiguous bytes
//     (CP >= 8: eight channels of one tap; CP == 4: four channels of two neighbouring taps of a row);
//   * rows of D are 16 consecutive pixels of an image row, so a lane ends up with 4 consecutive pixels of one output
//     channel (8 / 16 bytes of the NCHW result) and the BatchNorm statistics come straight from the accumulators;
//   * the weights are gathered once per workgroup from the fp32 (Co, Ci, 3, 3) tensor into B fragments held in registers
//     (the data gradient reads the same tensor transposed and flipped: no flipped copy is made); a workgroup walks a
//     range of images at one tile position.
#pragma once
#include "unet_direct.hpp"

typedef __bf16 ub_bf8 __attribute__((ext_vector_type(8)));
typedef float ub_f4 __attribute__((ext_vector_type(4)));
typedef unsigned ub_u2 __attribute__((ext_vector_type(2)));
typedef unsigned ub_u4 __attribute__((ext_vector_type(4)));

struct UbConvArgs {
  UdConvArgs c;
  int flip;                 // 1: W is the forward conv's (Ci, Co, 3, 3) tensor, use W[ci][co][8 - tap] (data gradient)
  int n_img, img_per_wg;
};

typedef float ub_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 ub_bf2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned ub_pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((ub_f2){lo, hi}, ub_bf2));
}
// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in dispatch order.  A (band, image range) grid
// dispatched as it stands puts vertically neighbouring bands -- which share two halo rows of every tile -- on different
// XCDs, so every halo row comes from HBM / the Infinity Cache twice.  Renumbered, XCD x works through a CONTIGUOUS range of
// (image range, band) pairs, band fastest: the bands of an image range run side by side on one L2.  (Bijective for any
// grid size; placement only matters for speed.)
__device__ __forceinline__ void ub_xcd_remap(int& band, int& chunk) {
  const unsigned nwg = gridDim.x * gridDim.y, l = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned xcd = l & 7, slot = l >> 3, q = nwg >> 3, r = nwg & 7;
  const unsigned w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  band = (int)(w % gridDim.x); chunk = (int)(w / gridDim.x);
}
template <bool B> struct UbBool { static constexpr bool value = B; };
template <int N> struct UbInt { static constexpr int value = N; };

#define UB_TW 64
#define UB_TH 16
#define UB_OOB 0x80000000u  // a byte offset no descriptor of this file covers (images are < 2 GiB): loads give 0, stores are dropped

// One descriptor per (tensor, image): a lane's 32-bit byte offset inside the image is computed ONCE per kernel (the
// staging pattern of a thread is the same for every image and band), rows outside the image / channels past the
// source's count / unused lanes read as zero through the range check -- the stage has no branches and no 64-bit
// address arithmetic.  (The first version computed 64-bit addresses and the BatchNorm affine per task and channel:
// ~400 VALU instructions per 16 values, 45 us of a 119 us launch with the global loads removed.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ub_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}

// Thin outputs (Co <= 4 / <= 8: every 256^2 and 128^2 layer of the UNet) with D[pixel][co] leave 12 / 8 of the MFMA's 16
// columns -- and of the epilogue's lanes -- idle.  DX = 4 / 2 packs DX neighbouring pixels of a row into the 16 rows of
// the product instead:  D[m = (co, dx)][n = pixel group] += A[m][k] * B[k][n],  k = (ky, column x_n - 1 + c, ci) over the
// DX + 2 columns the group's taps touch,  A[(co, dx)][(ky, c, ci)] = W[co][ci][ky][c - dx] (zero outside the 3 taps),
// B = the channels-last tile as it lies in LDS (the (DX + 2) * CP values of a group and row are contiguous).  One MFMA
// tile then covers 16 * DX pixels of a row for all output channels: 3 instead of 8 MFMAs (and LDS fragment reads) per 64
// pixels at Ci = Co = 4, a lane ends up with DX consecutive pixels of a channel, and every lane has an output.
template <int CP, int NB, int RB, bool TWO, int TW = UB_TW, int DX = 1, bool SPREAD = false>   // padded input channels (4, 8, 16, 32), blocks of 16 output
                                               // channels, rows per band, TWO: two views of CP/2 channels each (the skip /
                                               // up concat), TW: tile width (64; 32 for 32-pixel-wide images), DX: pixels
                                               // per MFMA row group (1; 4 / 2 for Co <= 4 / 8, NB = 1, TW = 64, CP <= 16)
__global__ __launch_bounds__(256, 2) void ub_conv3x3_kernel(UbConvArgs A) {
  static_assert(DX == 1 || (NB == 1 && TW == 64 && CP <= 16 && (DX == 2 || DX == 4)), "packed variant: thin outputs only");
  constexpr int TH = UB_TH, LDT = TW + 8;                 // col = x - x0 + 4: whole quads x0-4 .. x0+TW+3 are staged
  constexpr int CPC = CP >= 8 ? CP / 8 : 1;               // 16-byte chunks per pixel
  constexpr int CPK = (DX + 2) * CP / 8;                  // packed variant: k chunks of 8 per kernel row
  constexpr int NCH = DX > 1 ? 3 * CPK : (CP == 4 ? 6 : 9 * CPC);   // k chunks of 8
  constexpr int NM = (NCH + 3) / 4;                       // MFMAs per 16-pixel block (per 16 * DX pixels) and output block
  constexpr int NTB = TW / (16 * DX);                     // MFMA tiles per tile row
  constexpr int NR = RB + 2, QH = TW / 4 + 2;
  constexpr int CQS = TWO ? CP / 8 : CP / 4;              // channel quads per source
  constexpr int NTASK = CQS * NR * QH;                    // staging tasks per source and band (4 pixels x 4 channels each)
  constexpr int NTS = (NTASK + 255) / 256;                // ... per thread
  constexpr int NT = TWO ? 2 * NTS : NTS;                 // task t < NTS reads the first view, the others the second
  constexpr int WR = RB / 4;                              // rows of a band per wave
  __shared__ __attribute__((aligned(16))) unsigned short tile[NR * LDT * CP];
  __shared__ __attribute__((aligned(16))) float aff[2][CP];   // folded BatchNorm affine of the current image group
  __shared__ float red[4][32 * NB];
  const UdConvArgs& a = A.c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, lg = lane >> 4;
  // a workgroup owns a band of TH image rows and walks its tiles_x tiles left to right (then the next image): the
  // cache lines a tile shares with its neighbour (the halo quad of a 128-byte bf16 row segment is a whole extra line on
  // either side) are then re-read by the same CU a moment later instead of by another XCD from HBM
  const int tiles_x = a.Wd / TW;
  int bandx, chunky;
  ub_xcd_remap(bandx, chunky);
  const int y0 = bandx * TH;
  const int Ci = a.C0 + a.C1;
  const int es0 = a.bf0 ? 2 : 4, es1 = a.bf1 ? 2 : 4, eso = a.bfo ? 2 : 4;
  const int HW = a.H * a.Wd;

  // B fragments: lane (co = lp, chunk = 4m + lg) holds 8 consecutive k
  ub_bf8 wf[NM][NB];
  {
    float wr[NM][NB][8];                                  // branch-free gather: every load is in flight before the first use
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 16 + lp, c = 4 * m + lg;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          int ci, tap;
          bool ok = c < NCH && co < a.Co;
          if (DX > 1) {                                     // row m = lp = (co, dx); k = (ky, column, ci)
            const int e = (c % CPK) * 8 + j, col = e / CP, kx = col - (lp % DX);
            ci = e % CP; tap = (c / CPK) * 3 + kx;
            ok = c < NCH && (lp / DX) < a.Co && kx >= 0 && kx < 3;
            if (!ok) tap = 0;
          }
          else if (CP == 4) { const int kx = 2 * (c & 1) + (j >> 2); ci = j & 3; tap = (c >> 1) * 3 + kx; ok = ok && kx < 3; }
          else { tap = c / CPC; ci = (c % CPC) * 8 + j; }
          ok = ok && ci < Ci;
          const int cic = min(ci, Ci - 1), coc = min(DX > 1 ? lp / DX : co, a.Co - 1), tc = min(tap, 8);
          const float w = a.W[A.flip ? ((long)cic * a.Co + coc) * 9 + 8 - tc : ((long)coc * Ci + cic) * 9 + tc];
          wr[m][nb][j] = ok ? w : 0.f;
        }
      }
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[m][nb][j] = (__bf16)wr[m][nb][j];
  }
  // A fragments: LDS byte address of this lane's chunk in the first 16-pixel block of its first row
  int abase[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    int c = 4 * m + lg;
    if (c >= NCH) c = 0;                                  // (its weights are zero; any finite data will do)
    int o;
    if (DX > 1) o = (((c / CPK) * LDT + DX * lp + 3) * CP + (c % CPK) * 8) * 2;   // group lp's columns x_n - 1 .. of row ky
    else if (CP == 4) o = (((c >> 1) * LDT + lp + 2 * (c & 1) + 3) * CP) * 2;
    else { const int tap = c / CPC, ky = tap / 3, kx = tap - 3 * ky; o = ((ky * LDT + lp + kx + 3) * CP + (c % CPC) * 8) * 2; }
    abase[m] = o + wave * WR * LDT * CP * 2;
  }
  // staging tasks of this thread: the same for every image and band.  toff: byte offset inside the image of
  // (first channel of the quad, row y0 + r - 1, column x); trow: that row, or far outside for lanes without a task /
  // columns outside the image; tdst: LDS element of (row r, column, channel quad)
  // Lane order of the tasks (CP >= 8): channel quad fastest, then pixel quad, then row, and a lane writes its four pixels
  // in an order rotated by a few bits of its pixel-quad index (trot).  A pixel is CP * 2 = 16 / 32 / 64 bytes of LDS, a
  // pixel quad 64 / 128 / 256: with the pixel quad fastest and every lane writing pixel j in the j-th instruction the
  // lanes of a ds_write_b64 were 128 bytes apart at CP = 16 -- one bank pair for a whole lane group (the counters of the
  // 13-channel first conv: 324 M conflict cycles against 79 M LDS-active ones; the 4-channel layers, whose lanes write 32
  // contiguous bytes each, had 5 M).  A ds_write_b64 is served in four groups of 16 contiguous lanes over 32 banks (a
  // 128-byte row): now the 16 lanes of a group cover its 16 eight-byte slots -- (channel quad) x (rotated pixel) [x a low
  // pixel-quad bit at CP = 8].
  constexpr int NQB = CP == 8 ? 1 : 0;                    // low pixel-quad bits that still select a slot of the 128-byte row
  // A template variant (SPREAD) launched for an fp32 first view only -- the 13-channel network input: on the bf16-stored
  // layers the stage is bound by instruction issue and the selects of the rotation cost more than the conflicts (16 -> 16
  // at 64^2: 29 -> 31 us, and 33 us with the choice as a run-time flag; the first conv: 206 -> 189 us).
  static_assert(!SPREAD || CP >= 8, "the spread staging order is for >= 8 padded channels");
  constexpr bool spread = SPREAD;
  int toff[NTS], trow[NTS], tcol[NTS], tdst[NTS], trot[NTS];
#pragma unroll
  for (int t = 0; t < NTS; ++t) {
    const int idx = tid + t * 256;
    int cq, r, q;
    if (spread) { cq = idx % CQS; const int r2 = idx / CQS; r = r2 / QH; q = r2 - r * QH; }
    else { cq = idx / (NR * QH); const int r2 = idx - cq * (NR * QH); r = r2 / QH; q = r2 - r * QH; }
    trot[t] = spread ? (q >> NQB) & 3 : 0;
    tcol[t] = 4 * q - 4;                                  // column relative to the tile
    toff[t] = (cq * 4 * a.H + y0 + r - 1) * a.Wd + tcol[t];   // in ELEMENTS (the two views may differ in width), tile 0
    trow[t] = y0 + r - 1;
    tdst[t] = (idx < NTASK) ? (r * LDT + 4 * q) * CP + cq * 4 : -1;
  }
  // output: lane (co = lp + 16 nb, pixels 4 lg .. 4 lg + 3 of a block)
  constexpr int NVO = DX == 2 ? 2 : NB;                   // (packed: lane = channel lg and 4 pixels, or channels 2 lg, 2 lg + 1 and 2 pixels)
  unsigned vout[NVO];
#pragma unroll
  for (int nb = 0; nb < NVO; ++nb) {
    const int co = DX == 4 ? lg : (DX == 2 ? 2 * lg + nb : nb * 16 + lp);
    vout[nb] = (co < a.Co) ? (unsigned)(((co * a.H + y0 + wave * WR) * a.Wd + (DX > 1 ? DX * lp : 4 * lg)) * eso) : UB_OOB;
  }

  // ---- the workgroup's stages: (image, tile of the row band, band of RB rows) in order.  Software pipeline: the global loads of stage k+1 are
  // issued in front of the matrix phase of stage k and wait in registers, so their latency is covered by it:
  //   barrier | convert + write LDS (stage k) | barrier | issue loads (k+1) | matrix phase (k) | ...
  constexpr int NBI = TH / RB;                            // bands per image
  const long img0 = (long)chunky * A.img_per_wg;
  const long img1 = min(img0 + (long)A.img_per_wg, (long)A.n_img);
  const int SPI = NBI * tiles_x;                          // stages per image
  const int nstage = (int)(img1 - img0) * SPI;
  // prefetch distance in stages (raw[] is a ring of PF register sets).  Two stages ahead was measured for CP <= 8 and
  // changed nothing (4->4 at 256^2: 61 -> 64 us): the stage is bound by instruction issue, not by the load latency.
  constexpr int PF = 1;
  ub_u4 raw[PF][NT][4];
  auto inside = [&](const int tl, const int tx, const int b) {
    return tdst[tl] >= 0 && (unsigned)(trow[tl] + b * RB) < (unsigned)a.H && (unsigned)(tcol[tl] + tx * TW) < (unsigned)a.Wd;
  };

  auto load_view = [&](auto bfc, auto slotc, const int sec, const long img, const int tx, const int b) {
    constexpr bool BF = decltype(bfc)::value;
    constexpr int SL = decltype(slotc)::value;
    constexpr int es = BF ? 2 : 4;
    const __amdgpu_buffer_rsrc_t rs = sec ? ub_rsrc(reinterpret_cast<const char*>(a.in1) + img * a.is1 * es, (long)a.C1 * HW * es)
                                          : ub_rsrc(reinterpret_cast<const char*>(a.in0) + ud_base0(a, img) * es, (long)a.C0 * HW * es);
    const int pb = HW * es;                               // channels past the view's count fall out of the descriptor's range
#pragma unroll
    for (int tl = 0; tl < NTS; ++tl) {
      const unsigned off = inside(tl, tx, b) ? (unsigned)((toff[tl] + tx * TW + b * RB * a.Wd) * es) : UB_OOB;   // (nothing is fetched for the padding)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ub_u4& d = raw[SL][sec * NTS + tl][j];
        if (BF) { const ub_u2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(off + j * pb), 0, 0); d[0] = u[0]; d[1] = u[1]; }
        else d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + j * pb), 0, 0);
      }
    }
  };
  auto issue_loads = [&](const int k, auto slotc) {
    const long img = img0 + k / SPI;
    const int tx = (k / NBI) % tiles_x, b = k % NBI;
    if (a.bf0) load_view(UbBool<true>{}, slotc, 0, img, tx, b); else load_view(UbBool<false>{}, slotc, 0, img, tx, b);
    if (TWO) { if (a.bf1) load_view(UbBool<true>{}, slotc, 1, img, tx, b); else load_view(UbBool<false>{}, slotc, 1, img, tx, b); }
  };
  // activation (folded BatchNorm affine, ReLU), rounding to bf16, transposition to channels-last, LDS
  auto commit_view = [&](auto bfc, auto slotc, const int sec, const int tx, const int b) {
    constexpr bool BF = decltype(bfc)::value;
    constexpr int SL = decltype(slotc)::value;
    const float flo = (sec ? a.relu1 : a.relu0) ? 0.f : -__builtin_inff();
#pragma unroll
    for (int tl = 0; tl < NTS; ++tl) {
      const int dsta = tdst[tl] + (sec ? CP / 2 : 0);     // (channel position inside the pixel)
      const int ch = (dsta < 0 ? 0 : dsta) & (CP - 1);
      const float4 sc4 = *reinterpret_cast<const float4*>(&aff[0][ch]);
      const float4 sh4 = *reinterpret_cast<const float4*>(&aff[1][ch]);
      const float scj[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shj[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      // zero padding comes AFTER the activation: rows / columns outside the image are zero, not relu(shift); what the
      // loads brought from there (a neighbouring row or plane, or the range check's zero) is multiplied away
      const float keep = inside(tl, tx, b) ? 1.f : 0.f;
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const ub_u4& d = raw[SL][sec * NTS + tl][j];
        const unsigned ux = d[0], uy = d[1];
        if (BF) v[j] = make_float4(ua_lo(ux), ua_hi(ux), ua_lo(uy), ua_hi(uy));
        else { const unsigned uz = d[2], uw = d[3];
               v[j] = make_float4(__uint_as_float(ux), __uint_as_float(uy), __uint_as_float(uz), __uint_as_float(uw)); }
        const float s_ = scj[j] * keep, t_ = shj[j] * keep;
        v[j].x = fmaxf(v[j].x * s_ + t_, flo); v[j].y = fmaxf(v[j].y * s_ + t_, flo);
        v[j].z = fmaxf(v[j].z * s_ + t_, flo); v[j].w = fmaxf(v[j].w * s_ + t_, flo);
      }
      const uint2 p0 = make_uint2(ub_pack2(v[0].x, v[1].x), ub_pack2(v[2].x, v[3].x));
      const uint2 p1 = make_uint2(ub_pack2(v[0].y, v[1].y), ub_pack2(v[2].y, v[3].y));
      const uint2 p2 = make_uint2(ub_pack2(v[0].z, v[1].z), ub_pack2(v[2].z, v[3].z));
      const uint2 p3 = make_uint2(ub_pack2(v[0].w, v[1].w), ub_pack2(v[2].w, v[3].w));
      if (tdst[tl] >= 0) {
        unsigned short* dst = &tile[dsta];
        if (CP == 4) {
          *reinterpret_cast<uint4*>(dst) = make_uint4(p0.x, p0.y, p1.x, p1.y);
          *reinterpret_cast<uint4*>(dst + 8) = make_uint4(p2.x, p2.y, p3.x, p3.y);
        } else {
          if constexpr (!spread) {
            *reinterpret_cast<uint2*>(dst) = p0; *reinterpret_cast<uint2*>(dst + CP) = p1;
            *reinterpret_cast<uint2*>(dst + 2 * CP) = p2; *reinterpret_cast<uint2*>(dst + 3 * CP) = p3;
          } else {
            // instruction n writes pixel (n + trot) & 3 of the lane's quad
            const int rot = trot[tl];
            const bool r1 = rot & 1, r2 = rot & 2;
            const uint2 a0 = r1 ? p1 : p0, a1 = r1 ? p2 : p1, a2 = r1 ? p3 : p2, a3 = r1 ? p0 : p3;     // rotated by rot & 1
            const uint2 w0 = r2 ? a2 : a0, w1 = r2 ? a3 : a1, w2 = r2 ? a0 : a2, w3 = r2 ? a1 : a3;     // ... and by rot & 2
            *reinterpret_cast<uint2*>(dst + ((rot + 0) & 3) * CP) = w0; *reinterpret_cast<uint2*>(dst + ((rot + 1) & 3) * CP) = w1;
            *reinterpret_cast<uint2*>(dst + ((rot + 2) & 3) * CP) = w2; *reinterpret_cast<uint2*>(dst + ((rot + 3) & 3) * CP) = w3;
          }
        }
      }
    }
  };
  float s1[NVO], s2[NVO];
#pragma unroll
  for (int nb = 0; nb < NVO; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; }
  auto matrix_phase = [&](auto bfoc, const long img, const int tx, const int b) {
    constexpr bool BFO = decltype(bfoc)::value;
    constexpr int eo = BFO ? 2 : 4;
    const __amdgpu_buffer_rsrc_t ro = ub_rsrc(reinterpret_cast<char*>(a.out) + img * a.os * eo, (long)a.Co * HW * eo);
    const char* lds = reinterpret_cast<const char*>(tile);
    if constexpr (DX > 1) {
#pragma unroll 1
      for (int rr = 0; rr < WR; ++rr) {
        unsigned vo[NVO];
#pragma unroll
        for (int nb = 0; nb < NVO; ++nb) vo[nb] = vout[nb] + (unsigned)(((b * RB + rr) * a.Wd + tx * TW) * eo);
        ub_f4 acc[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[tb] = (ub_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
          for (int tb = 0; tb < NTB; ++tb) {
            const char* ap = lds + (abase[m] + rr * (LDT * CP * 2)) + tb * 16 * DX * CP * 2;
            ub_bf8 pf;
            if (CP == 4) {                                  // 8-byte aligned only
              // TWO ds_read_b64, not one ds_read2_b64 (the second offset is hidden from the compiler): the lanes of a pixel
              // group row are 32 bytes apart -- over the 32 banks of a ds_read2_b64 a 4-way conflict on either half
              // (conflict / active cycles 5.6, a quarter of the kernel's instruction waits), over the 64 banks of a
              // ds_read_b64 2-way
              int o2 = abase[m] + rr * (LDT * CP * 2) + tb * 16 * DX * CP * 2 + 8;
              asm volatile("" : "+v"(o2));
              const uint2 lo = *reinterpret_cast<const uint2*>(ap);
              const uint2 hi = *reinterpret_cast<const uint2*>(lds + o2);
              pf = __builtin_bit_cast(ub_bf8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            } else {
              pf = __builtin_bit_cast(ub_bf8, *reinterpret_cast<const uint4*>(ap));
            }
            acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m][0], pf, acc[tb], 0, 0, 0);
          }
        // D[m = 4 lg + r][n = lp]: DX = 4: channel lg, pixels 4 lp + r;  DX = 2: channels 2 lg + (r >> 1), pixels 2 lp + (r & 1)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
          const float c0 = acc[tb][0], c1 = acc[tb][1], c2 = acc[tb][2], c3 = acc[tb][3];
          if (DX == 4) {
            if (BFO) {
              const ub_u2 pk = {ub_pack2(c0, c1), ub_pack2(c2, c3)};
              __builtin_amdgcn_raw_buffer_store_b64(pk, ro, (int)vo[0], 0, 0);
            } else {
              const ub_u4 pk = {__float_as_uint(c0), __float_as_uint(c1), __float_as_uint(c2), __float_as_uint(c3)};
              __builtin_amdgcn_raw_buffer_store_b128(pk, ro, (int)vo[0], 0, 0);
            }
            s1[0] += (c0 + c1) + (c2 + c3);
            s2[0] += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
          } else {
            if (BFO) {
              __builtin_amdgcn_raw_buffer_store_b32(ub_pack2(c0, c1), ro, (int)(vo[0] + tb * 64), 0, 0);
              __builtin_amdgcn_raw_buffer_store_b32(ub_pack2(c2, c3), ro, (int)(vo[1] + tb * 64), 0, 0);
            } else {
              const ub_u2 p0 = {__float_as_uint(c0), __float_as_uint(c1)}, p1 = {__float_as_uint(c2), __float_as_uint(c3)};
              __builtin_amdgcn_raw_buffer_store_b64(p0, ro, (int)(vo[0] + tb * 128), 0, 0);
              __builtin_amdgcn_raw_buffer_store_b64(p1, ro, (int)(vo[1] + tb * 128), 0, 0);
            }
            s1[0] += c0 + c1; s2[0] += c0 * c0 + c1 * c1;
            s1[1] += c2 + c3; s2[1] += c2 * c2 + c3 * c3;
          }
        }
      }
      return;
    }
#pragma unroll 1
    for (int rr = 0; rr < WR; ++rr) {
      unsigned vo[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) vo[nb] = vout[nb] + (unsigned)(((b * RB + rr) * a.Wd + tx * TW) * eo);   // (UB_OOB stays out of range)
      // the four 16-pixel blocks of the row as independent accumulation chains: all MFMAs are issued before the first
      // epilogue reads an accumulator (a block at a time, 34 % of the wave cycles were issue stalls on MFMA results)
      ub_f4 acc[TW / 16][NB];
#pragma unroll
      for (int cb = 0; cb < TW / 16; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = (ub_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int cb = 0; cb < TW / 16; ++cb) {
          const char* ap = lds + (abase[m] + rr * (LDT * CP * 2)) + cb * 16 * CP * 2;
          ub_bf8 af;
          if (CP == 4) {                                  // 8-byte aligned only
            const uint2 lo = *reinterpret_cast<const uint2*>(ap);
            const uint2 hi = *reinterpret_cast<const uint2*>(ap + 8);
            af = __builtin_bit_cast(ub_bf8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          } else {
            af = __builtin_bit_cast(ub_bf8, *reinterpret_cast<const uint4*>(ap));
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[m][nb], acc[cb][nb], 0, 0, 0);
        }
      // D[pixel 4*lg + r][co = lp]
#pragma unroll
      for (int cb = 0; cb < TW / 16; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float c0 = acc[cb][nb][0], c1 = acc[cb][nb][1], c2 = acc[cb][nb][2], c3 = acc[cb][nb][3];
          if (BFO) {
            const ub_u2 pk = {ub_pack2(c0, c1), ub_pack2(c2, c3)};
            __builtin_amdgcn_raw_buffer_store_b64(pk, ro, (int)(vo[nb] + cb * 32), 0, 0);
          } else {
            const ub_u4 pk = {__float_as_uint(c0), __float_as_uint(c1), __float_as_uint(c2), __float_as_uint(c3)};
            __builtin_amdgcn_raw_buffer_store_b128(pk, ro, (int)(vo[nb] + cb * 64), 0, 0);
          }
          s1[nb] += (c0 + c1) + (c2 + c3);
          s2[nb] += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
        }
    }
  };
  auto load_affine = [&](const long img) {                // folded BatchNorm affine of the image's group -> LDS
    if (tid < CP) {
      const long grp = img / a.gsize;
      float s_ = 0.f, t_ = 0.f;
      const bool sec = TWO && tid >= CP / 2;
      const int cc = sec ? tid - CP / 2 : tid;
      if (cc < (sec ? a.C1 : a.C0)) {
        const float* sc = sec ? a.sc1 : a.sc0;
        const float* sh = sec ? a.sh1 : a.sh0;
        const long gi = grp * (sec ? a.C1 : a.C0) + cc;
        s_ = sc ? sc[gi] : 1.f; t_ = sc ? sh[gi] : 0.f;
      }
      aff[0][tid] = s_; aff[1][tid] = t_;
    }
  };

  auto stage = [&](const int k, auto slotc) {
    const long img = img0 + k / SPI;
    const int tx = (k / NBI) % tiles_x, b = k % NBI;
    __syncthreads();                                      // stage k-1's matrix phase is done with the tile; aff is in place
    if (a.bf0) commit_view(UbBool<true>{}, slotc, 0, tx, b); else commit_view(UbBool<false>{}, slotc, 0, tx, b);
    if (TWO) { if (a.bf1) commit_view(UbBool<true>{}, slotc, 1, tx, b); else commit_view(UbBool<false>{}, slotc, 1, tx, b); }
    __syncthreads();
    if (k + PF < nstage) issue_loads(k + PF, slotc);      // (this stage's registers are free again)
    if (k + 1 < nstage && (k + 1) % SPI == 0 && (img + 1) % a.gsize == 0) load_affine(img + 1);   // (every reader of aff is past the barrier)
    if (a.bfo) matrix_phase(UbBool<true>{}, img, tx, b); else matrix_phase(UbBool<false>{}, img, tx, b);
    if (b == NBI - 1 && tx == tiles_x - 1) {              // the band of this image is complete
      if (a.stats) {                                      // per-BAND BatchNorm statistics from the accumulators (one row
                                                          // per image and 16-row band: the finalize kernel walks 4x fewer
                                                          // rows than with one per tile)
#pragma unroll
        for (int nb = 0; nb < NVO; ++nb) {
          float t1 = s1[nb], t2 = s2[nb];
          if (DX > 1) {                                     // a channel's sums lie in the 16 lanes of a group
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
            const int co = DX == 4 ? lg : 2 * lg + nb;
            if (lp == 0) { red[wave][2 * co] = t1; red[wave][2 * co + 1] = t2; }
          } else {
            t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
            t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
            if (lg == 0) { red[wave][2 * (nb * 16 + lp)] = t1; red[wave][2 * (nb * 16 + lp) + 1] = t2; }
          }
        }
        __syncthreads();
        if (tid < 32 * NB && (tid >> 1) < a.Co) {
          a.stats[((img * (long)gridDim.x + (long)bandx) * a.Co + (tid >> 1)) * 2 + (tid & 1)] =
              (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < NVO; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; }
    }
  };
  if (nstage > 0) { load_affine(img0); issue_loads(0, UbInt<0>{}); }
  if (PF > 1 && nstage > 1) issue_loads(1, UbInt<PF - 1>{});
#pragma unroll 1
  for (int k0 = 0; k0 < nstage; k0 += PF) {
    stage(k0, UbInt<0>{});
    if (PF > 1 && k0 + 1 < nstage) stage(k0 + 1, UbInt<PF - 1>{});
  }
}

// Weight gradient on the bf16 matrix pipe (ub_wgrad3x3_kernel):
//   dW[co][(ci,ky),kx] = sum_{img,pixel} dy[co][pixel] * act(x)[ci][pixel + (ky-1, kx-1)]
//   D[m = (ci,ky)][n = co] += A[m][k] * B[k][n],   k = 32 consecutive pixels of an image row, one MFMA per kx:
//   * B: a lane's 8 consecutive pixels of dy[co] are 16 contiguous bytes of the NCHW tensor -- loaded straight from
//     global memory into the fragment register (no LDS), one stage ahead;
//   * A: the activated input tile lies in LDS PLANAR as bf16 ([ci][row][col], as in HBM); a lane reads the 16 aligned
//     bytes of its 8 pixels plus the dword on either side and forms the kx = 0 / 1 / 2 fragments with v_alignbit
//     (the shift by one pixel is a 16-bit funnel shift; all lanes of an MFMA share kx, so the shift is a constant);
//   * the four waves of a workgroup split the tile's pixels (k), each keeps the whole D in registers over every image
//     of its range; one fold at the end, slab rows as for uw_wgrad_mfma_kernel (summed by uslab_reduce_kernel).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ub_align16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

template <int CI, int NB, int RB, bool TWO>    // padded input channels, blocks of 16 output channels, rows per stage, two views
__global__ __launch_bounds__(256, 2) void ub_wgrad3x3_kernel(UdWgradArgs a) {
  constexpr int TW = UB_TW, TH = UB_TH, LDW = TW + 16;    // col = x - x0 + 8 (interior 16-byte aligned; x0-4 .. x0+TW+3 staged)
  constexpr int NR = RB + 2, QH = TW / 4 + 2;
  constexpr int PS = NR * LDW + 8;                        // plane stride (elements)
  constexpr int CS = TWO ? CI / 2 : CI;                   // channels per view
  constexpr int TPP = 2 * NR * QH;                        // staging tasks (4 pixels of one channel) per channel pair
  constexpr int NP = (TPP + 255) / 256;                   // ... per thread
  constexpr int NCP = CS / 2;                             // channel pairs per view
  constexpr int NTV = NCP * NP;                           // tasks per thread and view
  constexpr int NV = TWO ? 2 : 1;
  constexpr int NPAIR = CI * 3;                           // rows of D: (ci, ky)
  constexpr int NJ = (NPAIR + 15) / 16;
  constexpr int KS = RB / 2;                              // k steps (32 pixels) per wave and stage
  constexpr int NBI = TH / RB;
  constexpr int XS_BYTES = CI * PS * 2, FOLD_BYTES = NJ * 3 * NB * 1024;
  __shared__ __attribute__((aligned(16))) char smem[XS_BYTES > FOLD_BYTES ? XS_BYTES : FOLD_BYTES];
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
  float* fold = reinterpret_cast<float*>(smem);            // (after the last stage)
  __shared__ float aff[2][CI];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, lg = lane >> 4;
  const int tiles_x = a.Wd / TW;                         // (a workgroup walks the tiles of its row band left to right,
  int bandx, chunky;
  ub_xcd_remap(bandx, chunky);                            // (bands of an image range on one XCD, see above)
  const int y0 = bandx * TH;                              //  as ub_conv3x3_kernel does)
  const int Ci = a.C0 + a.C1;
  const int HW = a.H * a.Wd;

  // staging pattern of one channel pair (the same for every pair, view, image and stage)
  int toff[NP], tdst[NP], trow[NP], tcol[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = tid + p * 256;
    const int cl = idx / (NR * QH), r2 = idx - cl * (NR * QH), r = r2 / QH, q = r2 - r * QH;
    tcol[p] = 4 * q - 4;
    toff[p] = (cl * a.H + y0 + r - 1) * a.Wd + tcol[p];
    tdst[p] = (idx < TPP) ? cl * PS + r * LDW + 4 * q + 4 : -1;
    trow[p] = y0 + r - 1;
  }
  // A fragments: LDS byte address of this lane's 8 pixels (kx = 1) in the first row / half of the wave
  int abase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pi = min(j * 16 + lp, NPAIR - 1), ci = pi / 3, ky = pi - 3 * ci;
    abase[j] = (ci * PS + (wave * (RB / 4) + ky) * LDW + 8 * lg + 8) * 2;
  }
  // B fragments: element offset of this lane's 8 pixels of dy[co] in the first row / half of the wave
  unsigned dyo[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    dyo[nb] = (nb * 16 + lp < a.Co) ? (unsigned)((((nb * 16 + lp) * a.H + y0 + wave * (RB / 4)) * a.Wd + 8 * lg) * 2) : UB_OOB;

  ub_f4 acc[NJ][3][NB];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[j][kx][nb] = (ub_f4){0.f, 0.f, 0.f, 0.f};

  const long img0 = (long)chunky * a.img_per_wg;
  const long img1 = min(img0 + (long)a.img_per_wg, a.n_img);
  const int SPI = NBI * tiles_x;                          // stages per image
  const int nstage = (int)(img1 - img0) * SPI;
  ub_u4 raw[NV * NTV];
  auto inside = [&](const int p, const int tx, const int b) {
    return tdst[p] >= 0 && (unsigned)(trow[p] + b * RB) < (unsigned)a.H && (unsigned)(tcol[p] + tx * TW) < (unsigned)a.Wd;
  };                                    // (bf16 views use two dwords of each)
  ub_u4 dyr[KS][NB];

  auto load_view = [&](auto bfc, const int sec, const long img, const int tx, const int b) {
    constexpr bool BF = decltype(bfc)::value;
    constexpr int es = BF ? 2 : 4;
    const __amdgpu_buffer_rsrc_t rs = sec ? ub_rsrc(reinterpret_cast<const char*>(a.in1) + img * a.is1 * es, (long)a.C1 * HW * es)
                                          : ub_rsrc(reinterpret_cast<const char*>(a.in0) + ud_base0(a, img) * es, (long)a.C0 * HW * es);
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned off = inside(p, tx, b) ? (unsigned)((toff[p] + cp * 2 * HW + tx * TW + b * RB * a.Wd) * es) : UB_OOB;
        ub_u4& d = raw[sec * NTV + cp * NP + p];
        if (BF) { const ub_u2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); d[0] = u[0]; d[1] = u[1]; }
        else d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
      }
  };
  auto issue_loads = [&](const int k) {
    const long img = img0 + k / SPI;
    const int tx = (k / NBI) % tiles_x, b = k % NBI;
    if (a.bf0) load_view(UbBool<true>{}, 0, img, tx, b); else load_view(UbBool<false>{}, 0, img, tx, b);
    if (TWO) { if (a.bf1) load_view(UbBool<true>{}, 1, img, tx, b); else load_view(UbBool<false>{}, 1, img, tx, b); }
    const __amdgpu_buffer_rsrc_t rd = ub_rsrc(reinterpret_cast<const char*>(a.dy) + img * a.dys * 2, (long)a.Co * HW * 2);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        dyr[s][nb] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)(dyo[nb] + (unsigned)(((b * RB + (s >> 1)) * a.Wd + tx * TW + 32 * (s & 1)) * 2)), 0, 0);
  };
  auto commit_view = [&](auto bfc, const int sec, const int tx, const int b) {
    constexpr bool BF = decltype(bfc)::value;
    const float flo = (sec ? a.relu1 : a.relu0) ? 0.f : -__builtin_inff();
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int cbase = sec * CS + cp * 2;
        const int dst = tdst[p] + cbase * PS;
        const int ch = cbase + ((tdst[p] >= PS) ? 1 : 0);
        const float keep = inside(p, tx, b) ? 1.f : 0.f;
        const float s_ = aff[0][ch] * keep, t_ = aff[1][ch] * keep;
        const ub_u4& d = raw[sec * NTV + cp * NP + p];
        const unsigned ux = d[0], uy = d[1];
        float4 v;
        if (BF) v = make_float4(ua_lo(ux), ua_hi(ux), ua_lo(uy), ua_hi(uy));
        else { const unsigned uz = d[2], uw = d[3];
               v = make_float4(__uint_as_float(ux), __uint_as_float(uy), __uint_as_float(uz), __uint_as_float(uw)); }
        v.x = fmaxf(v.x * s_ + t_, flo); v.y = fmaxf(v.y * s_ + t_, flo);
        v.z = fmaxf(v.z * s_ + t_, flo); v.w = fmaxf(v.w * s_ + t_, flo);
        if (tdst[p] >= 0) *reinterpret_cast<uint2*>(&xs[dst]) = make_uint2(ub_pack2(v.x, v.y), ub_pack2(v.z, v.w));
      }
  };
  auto load_affine = [&](const long img) {
    if (tid < CI) {
      const long grp = img / a.gsize;
      float s_ = 0.f, t_ = 0.f;
      const bool sec = TWO && tid >= CS;
      const int cc = sec ? tid - CS : tid;
      if (cc < (sec ? a.C1 : a.C0)) {
        const float* sc = sec ? a.sc1 : a.sc0;
        const float* sh = sec ? a.sh1 : a.sh0;
        const long gi = grp * (sec ? a.C1 : a.C0) + cc;
        s_ = sc ? sc[gi] : 1.f; t_ = sc ? sh[gi] : 0.f;
      }
      aff[0][tid] = s_; aff[1][tid] = t_;
    }
  };

  if (nstage > 0) { load_affine(img0); issue_loads(0); }
#pragma unroll 1
  for (int k = 0; k < nstage; ++k) {
    const long img = img0 + k / SPI;
    const int tx = (k / NBI) % tiles_x, b = k % NBI;
    __syncthreads();                                      // stage k-1's matrix phase is done with the tile; aff is in place
    if (a.bf0) commit_view(UbBool<true>{}, 0, tx, b); else commit_view(UbBool<false>{}, 0, tx, b);
    if (TWO) { if (a.bf1) commit_view(UbBool<true>{}, 1, tx, b); else commit_view(UbBool<false>{}, 1, tx, b); }
    ub_u4 dyc[KS][NB];                                    // this stage's dy fragments (the registers are refilled below)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dyc[s][nb] = dyr[s][nb];
    __syncthreads();
    if (k + 1 < nstage) {
      issue_loads(k + 1);
      if ((k + 1) % SPI == 0 && (img + 1) % a.gsize == 0) load_affine(img + 1);
    }
    const char* lds = reinterpret_cast<const char*>(xs);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ub_bf8 bfr[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bfr[nb] = __builtin_bit_cast(ub_bf8, dyc[s][nb]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const char* ap = lds + abase[j] + ((s >> 1) * LDW + 32 * (s & 1)) * 2;
        const unsigned dm = *reinterpret_cast<const unsigned*>(ap - 4);
        const uint4 dc = *reinterpret_cast<const uint4*>(ap);
        const unsigned dn = *reinterpret_cast<const unsigned*>(ap + 16);
        const ub_u4 f0 = {ub_align16(dc.x, dm), ub_align16(dc.y, dc.x), ub_align16(dc.z, dc.y), ub_align16(dc.w, dc.z)};
        const ub_u4 f1 = {dc.x, dc.y, dc.z, dc.w};
        const ub_u4 f2 = {ub_align16(dc.y, dc.x), ub_align16(dc.z, dc.y), ub_align16(dc.w, dc.z), ub_align16(dn, dc.w)};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc[j][0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ub_bf8, f0), bfr[nb], acc[j][0][nb], 0, 0, 0);
          acc[j][1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ub_bf8, f1), bfr[nb], acc[j][1][nb], 0, 0, 0);
          acc[j][2][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ub_bf8, f2), bfr[nb], acc[j][2][nb], 0, 0, 0);
        }
      }
    }
  }
  // fold the four waves in a fixed order (wave 0 stores, 1..3 add in turn) and write this workgroup's slab row
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            float4* f = reinterpret_cast<float4*>(&fold[(((j * 3 + kx) * NB + nb) * 64 + lane) * 4]);
            float4 v = make_float4(acc[j][kx][nb][0], acc[j][kx][nb][1], acc[j][kx][nb][2], acc[j][kx][nb][3]);
            if (w > 0) { const float4 o = *f; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *f = v;
          }
    }
  }
  __syncthreads();
  const long z = (long)chunky * gridDim.x + bandx;
  float* row = a.slab + z * ((long)a.Co * Ci * 9);
  for (int e = tid; e < NJ * 3 * NB * 256; e += 256) {
    const int blk = e >> 8, l = (e >> 2) & 63, r = e & 3;
    const int nb = blk % NB, kx = (blk / NB) % 3, j = blk / (3 * NB);
    const int co = nb * 16 + (l & 15), pi = j * 16 + (l >> 4) * 4 + r;
    const int ci = pi / 3, ky = pi - 3 * ci;
    if (co < a.Co && ci < Ci) row[((long)co * Ci + ci) * 9 + ky * 3 + kx] = fold[e];
  }
}
