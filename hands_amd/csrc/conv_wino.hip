// conv_wino.hip -- 3x3 / stride 1 / pad 1 NHWC convolution as Winograd F(2x2, 3x3) on the gfx950 fp32 matrix cores.
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; d = 4x4 input patch, Y = 2x2 output pixels)
//
// 16 multiplications per 2x2 outputs and (cin, cout) pair instead of 36: the direct implicit GEMM (conv_igemm.hip)
// runs these layers at its k-loop ceiling (131-134 TFLOP/s), so the only way to make them faster is to execute
// fewer matrix-core FLOPs.  The 3x3 / stride-1 convolutions are 36 % of a hands_light forward.
//
// Per frequency f = (xi, nu) the layer is a GEMM   M_f[o, t] = sum_c U_f[o, c] * V_f[t, c]   (t = 2x2 output tile):
//   * U = G g G^T is computed ONCE on the host in fp64 (hands_pack_conv3x3_winograd_f64) and stored in MFMA-A operand
//     order: a wave's 16-byte-per-lane load IS its fragment (1 KB contiguous per instruction, no LDS, no shuffles);
//   * V = B^T d B is computed on the fly: the workgroup keeps the RAW input patch of its 32 tiles in LDS (16 channels
//     per stage, filled by LDS-DMA `buffer_load_dwordx4 ... lds`: no staging registers, out-of-image pixels arrive as
//     the buffer unit's zeros = the convolution's zero padding), wave xi combines the two patch rows B^T selects
//     (xi: rows (0,-2) (1,+2) (2,-1) (1,-3)), then the four column combinations nu -- 2 LDS reads + 2 adds per V value;
//   * wave xi accumulates its four nu in 4 x 16 accumulator registers (32 tiles x 32 output channels each), applies
//     the nu half of A^T . A in registers, and the xi half goes through LDS (32 KB) once per workgroup.
// Workgroup = 4 waves = 32 tiles (128 output pixels) x 32 output channels (x several channel blocks, one after the other
// on the same patch), 16 MFMAs (v_mfma_f32_32x32x2_f32) per wave and 8-channel step.  Tiles: flattened tile rows
// R = b * nh + ty; a block is NR rows x D columns (D = 4 or 8, "rect") or 32 consecutive tiles of the row-major order
// (D = 7, "linear": 7-tile-wide maps, and 14-tile-wide ones as two virtual half rows per tile row: no idle lanes).
//
// Numerics: every product and sum is fp32, in a fixed order that depends on the layer only (batch-size invariant,
// run-to-run deterministic).  Winograd re-associates the 3x3 sum, so results differ from the direct kernel by fp32
// rounding (tools/winograd_parity.py: the end-to-end vertex error against an fp64 forward is the same 1e-7 m as the
// direct algorithm's).  Replaces F.conv2d(3x3, s1, p1) + eval BatchNorm2d (folded) + ReLU of
// src/nets/backbone/resnet.py:140-142 (conv2 / bn2 / relu of every stride-1 Bottleneck).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct WinoArgs {
  const float* __restrict__ in;
  const float* __restrict__ u;      // [Cout/32][Cin/8][xi 4][nu 4][lane 64][4]
  const float* __restrict__ bias;
  float* out;
  int B, H, W, Cin, Cout, nh, nw;
  int in_ps, out_ps, act;
  int nblk_m, nblk_n, nseg;
  int rows;                         // B * nh flattened tile rows
  int nbw, ngrp;                    // channel blocks per workgroup (divides nblk_n), groups = nblk_n / nbw
  int sgs;                          // groups per pass of the tile order (divides ngrp): their weights fit an XCD's L2
  uint32_t nh_mul, nh_sh, sgs_mul, sgs_sh, nbm_mul, nbm_sh, nseg_mul, nseg_sh;   // magic numbers: x / nh, / sgs, / nblk_m, / nseg
};

// x / d for 0 <= x < 2^31 with M = ceil(2^(31 + l) / d), l = ceil(log2 d) >= 1 (Granlund-Montgomery: exact, M < 2^32);
// d == 1 is encoded as M = 0.  A runtime integer division costs ~35 vector instructions and the tile setup has a dozen:
// measured 9-11 k cycles of a workgroup's 38-123 k (tools/prof_wino.py)
__device__ __forceinline__ int fastdiv(int x, uint32_t mul, uint32_t sh) {
  return mul == 0 ? x : (int)(__umulhi((uint32_t)x, mul) >> sh);
}

__device__ __forceinline__ int wino_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float4 f4(const u32x4& v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float f4e(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 fma4(float s, const float4& a, const float4& b) {
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));   // s = +-1: exact add / sub
}

// LDS-DMA: 64 lanes x 16 B from (descriptor base + voff + soff) to the wave-uniform LDS address dst + lane * 16.  An
// out-of-range voff (0x80000000) writes ZEROS (probed on gfx950: tools/wino_probe/lds_dma_oob.hip).  (A plain device
// function: with the builtin written inside the kernel template hipcc's host pass silently drops the kernel's stub.)
__device__ __forceinline__ void wino_dma16(__amdgpu_buffer_rsrc_t rsrc, float* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, voff, soff, 0, 0);
}

template <int D, bool LINEAR>
struct WinoGeom {
  static constexpr int PW = 2 * D + 2;                          // patch pixels per input row
  static constexpr int PWP = PW;                                // row pitch (pixels)
  static constexpr int RP = 4 * PWP + 1;                        // pitch of a tile row's 4 input rows: odd, so consecutive
                                                                // tile rows start in different bank classes
  static constexpr int NR = LINEAR ? (D - 1 + 32 + D - 1) / D : 32 / D;   // tile rows a block can touch
  // LDS-DMA pieces: one wave instruction = 16 pixels x 64 B of ONE input row (lane = pixel * 4 + 16-byte slot), so the
  // row part of every source address is wave-uniform (scalar unit) and the lane part is computed once per workgroup
  static constexpr int NP = (PWP + 15) / 16;                    // pieces per input row
  static constexpr int NPIECE = NR * 4 * NP;
  static constexpr int NJ = (NPIECE + 3) / 4;                   // per wave
  static_assert(NJ <= 8, "WINO_STEP issues the next stage's DMA pieces as J = 2t, 2t + 1 for t = 0..3: at most 8 per wave");
  static constexpr int BUF_FLOATS = ((NR * RP * 16 + 63) / 64) * 64;
};

#ifndef WINO_WAVES
#define WINO_WAVES 3
#endif
constexpr int ZROUND_FLOATS = 4 * 32 * 32;                      // epilogue exchange, one round: [xi][tile 32][32 channels]

// One workgroup: 32 tiles (128 output pixels) x `nbw` blocks of 32 output channels, one after the other on the same
// patch (the tile setup and the first fill's latency are paid once; the stages of consecutive channel blocks form one
// software pipeline).
template <int D, bool LINEAR, int VSH = 0>   // VSH: a tile row is 1 << VSH "virtual rows" of D tiles (linear)
__global__ void __launch_bounds__(256, WINO_WAVES) conv_wino_f32_kernel(WinoArgs a) {
  using G = WinoGeom<D, LINEAR>;
  // (at least 41 KB: 3 workgroups per CU is what the ~150 registers allow anyway, and hipcc then schedules for that)
  constexpr int LDS_FLOATS = 2 * G::BUF_FLOATS > 10496 ? 2 * G::BUF_FLOATS : 10496;
  static_assert(G::BUF_FLOATS >= ZROUND_FLOATS, "the epilogue exchange reuses one patch buffer");
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = frequency row
  const int l31 = lane & 31, half = lane >> 5;

  // Tile order (each XCD takes a contiguous eighth of it): passes of `sgs` channel-block groups whose weights (<= 2 MB) stay in
  // the XCD's 4 MB L2, tile blocks inside a pass, the groups of one patch next to each other.  With all channel blocks in
  // one pass the 4-17 MB weight sets of the 256- / 512-channel layers were re-streamed from the Infinity Cache by every
  // workgroup (fabric reads 0.69 / 0.96 GB per launch against 0.10 / 0.03 GB of input).  Any bijection gives the same bits.
  const int tile = wino_xcd_remap(blockIdx.x, a.nblk_m * a.ngrp);
  const int tq = fastdiv(tile, a.sgs_mul, a.sgs_sh), gi = tile - tq * a.sgs;
  const int pass = fastdiv(tq, a.nbm_mul, a.nbm_sh), mb = tq - pass * a.nblk_m;
  const int nb0 = (pass * a.sgs + gi) * a.nbw;

  // R0: first (virtual) tile row of the block.  Linear blocks of maps 2^VSH * D tiles wide (the 14-tile rows of layer2 as
  // two half rows of 7) count in virtual rows V = R * 2^VSH + part: 32 consecutive tiles of the row-major order either way,
  // each virtual row with its own patch rows (columns 2 * D * part - 1 ...), so no lane idles on a 14-wide map either
  static_assert(VSH == 0 || LINEAR, "virtual rows exist in the linear block order only");
  constexpr int VM = (1 << VSH) - 1;
  int R0, s0, tx0;                                              // (all 32-bit: B * nh * nw < 2^31 is checked by the host)
  if constexpr (LINEAR) {
    const int t0 = mb * 32;
    R0 = t0 / D; s0 = t0 - R0 * D; tx0 = 0;
  } else {
    const int rb = fastdiv(mb, a.nseg_mul, a.nseg_sh), seg = mb - rb * a.nseg;
    R0 = rb * G::NR; s0 = 0; tx0 = seg * D;
  }
  const int R0r = R0 >> VSH;                                    // real tile row
  const int b_first = fastdiv(R0r, a.nh_mul, a.nh_sh);          // first image this block touches (wave-uniform)
  const int ty_first = R0r - b_first * a.nh;

  // ---- weights: MFMA-A fragments straight from L2, one 16-byte load per lane, frequency and 8-channel step; the steps of
  //      consecutive channel blocks are contiguous, so one running scalar offset walks all of this workgroup's stages ----
  const int nc8 = a.Cin >> 3;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u) + (size_t)nb0 * nc8 * 4096, 0, (int)0x80000000u, 0x00020000);
  const int w_off = (xi * 1024 + lane * 4) * 4;
  float4 wr[2][4];
#define WINO_LOADW(SET, STEP)                                                                       \
  do {                                                                                              \
    _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                                \
      wr[SET][nu] = f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off + nu * 1024, (STEP) * 16384, 0)); \
  } while (0)
  WINO_LOADW(0, 0);

  // ---- LDS-DMA fill: piece i = j * 4 + wave  ->  input row (i / NP) = (tile row Rl, row a of its 4), pixels 16 (i % NP) .. + 15.
  //      Lane part (once): pixel x, source quad q = slot ^ ((x >> 1) & 3) (source-side swizzle: the DMA destination is
  //      lane-linear), column validity.  Row part (scalar, per piece): image, input row, validity.  ----------------------
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in) + (size_t)b_first * a.H * a.W * a.in_ps, 0, (int)0x80000000u, 0x00020000);
  const int piece = (G::NP == 1) ? 0 : (xi % G::NP);            // (4 % NP == 0: a wave always gets the same piece of a row)
  static_assert(4 % G::NP == 0, "pieces per row must divide the wave count");
  const int fx = piece * 16 + (lane >> 2);                      // patch pixel of this lane
  const bool f_lane = fx < G::PWP;                              // lanes past the row's end write nothing (exec mask)
  int f_colv[VM + 1];                                           // lane part of the source offset, per part of a tile row
#pragma unroll
  for (int pv = 0; pv <= VM; ++pv) {
    const int fwx = 2 * (tx0 + pv * D) - 1 + fx;
    f_colv[pv] = ((unsigned)fwx < (unsigned)a.W) ? (fwx * a.in_ps + (((lane & 3) ^ ((fx >> 1) & 3)) * 4)) * 4 : (int)0x80000000u;
  }
  int f_off[G::NJ];
  int f_dst[G::NJ];                                             // wave-uniform LDS float offset of the piece (-1: none)
#pragma unroll
  for (int j = 0; j < G::NJ; ++j) {
    const int i = j * 4 + xi;
    const int r = i / G::NP, fRl = r >> 2, ar = r & 3;
    const int Rr = (R0 + fRl) >> VSH, pv = (R0 + fRl) & VM;                  // real tile row and part of it
    const int t = ty_first + (Rr - R0r);
    const int db = fastdiv(t, a.nh_mul, a.nh_sh), ty = t - db * a.nh;        // image b_first + db
    const int hy = 2 * ty - 1 + ar;
    const bool rowok = Rr < a.rows && (unsigned)hy < (unsigned)a.H;
    const int rowoff = (db * a.H + hy) * a.W * a.in_ps * 4;
    int f_col = f_colv[0];
#pragma unroll
    for (int k = 1; k <= VM; ++k) f_col = pv == k ? f_colv[k] : f_col;
    f_off[j] = (rowok && f_col >= 0) ? rowoff + f_col : (int)0x80000000u;
    f_dst[j] = i < G::NPIECE ? (fRl * G::RP + ar * G::PWP + piece * 16) * 16 : -1;
  }
#define WINO_FILL1(J, BUF_OFF, CH)                                                                  \
  do {                                                                                              \
    if ((J) < G::NJ && f_lane && (((J) + 1) * 4 <= G::NPIECE || f_dst[(J) < G::NJ ? (J) : 0] >= 0))  \
      wino_dma16(x_rsrc, lds + (BUF_OFF) + f_dst[(J) < G::NJ ? (J) : 0], f_off[(J) < G::NJ ? (J) : 0], (CH) * 64); \
  } while (0)
#define WINO_FILL(BUF_OFF, CH)                                                                      \
  do {                                                                                              \
    if (f_lane) {                                                                                   \
      _Pragma("unroll") for (int j = 0; j < G::NJ; ++j) {                                           \
        if ((j + 1) * 4 <= G::NPIECE || f_dst[j] >= 0)           /* only the last one can be missing */ \
          wino_dma16(x_rsrc, lds + (BUF_OFF) + f_dst[j], f_off[j], (CH) * 64);                      \
      }                                                                                             \
    }                                                                                               \
  } while (0)
  WINO_FILL(0, 0);

  // ---- this lane's tile and its patch read addresses ---------------------------------------------------------------
  const int qq = s0 + l31;
  const int Rl = qq / D, col = qq - Rl * D;
  // rows B^T selects for frequency row xi: r = d[a1] + sg * d[a2]
  const int a1 = (xi == 0) ? 0 : (xi == 2 ? 2 : 1);
  const int a2 = (xi == 0) ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sg = (xi == 1) ? 1.f : -1.f;
  // byte address of (row a, pixel 2 col + b, quad 2 s + half): pixel (Rl RP + a PWP + 2 col + b) * 64 + ((2 s + half) ^ swz) * 16,
  // swz = (col + (b >> 1)) & 3
  const int pix0 = Rl * G::RP + 2 * col;
  int rd[2][2];       // [row a1 / a2][b >> 1], s = 0; s = 1 is the address ^ 32
#pragma unroll
  for (int bc = 0; bc < 2; ++bc) {
    const int sw = (col + bc) & 3;
    rd[0][bc] = (pix0 + a1 * G::PWP) * 64 + ((half ^ sw) * 16);
    rd[1][bc] = (pix0 + a2 * G::PWP) * 64 + ((half ^ sw) * 16);
  }
  const char* ldsb = reinterpret_cast<const char*>(lds);

  // ---- the output side of this thread: tile tl, channels 4 cq .. 4 cq + 3 of a block -------------------------------
  const int tl = tid >> 3, cq = tid & 7;
  const int q2 = s0 + tl;
  const int Rl2 = q2 / D, col2 = q2 - Rl2 * D;
  const int oR = (R0 + Rl2) >> VSH, otx = tx0 + ((R0 + Rl2) & VM) * D + col2;
  const int ob = fastdiv(oR, a.nh_mul, a.nh_sh), oty = oR - ob * a.nh;
  const bool o_ok = oR < a.rows && otx < a.nw;
  const bool o_row1 = 2 * oty + 1 < a.H, o_col1 = 2 * otx + 1 < a.W;
  float* const o_base = a.out + (((size_t)ob * a.H + 2 * oty) * a.W + 2 * otx) * (size_t)a.out_ps + nb0 * 32 + cq * 4;
  const int zpos = (tl * 8 + (cq ^ (tl & 7))) * 4;              // exchange buffer: quad cq of tile t at position cq ^ (t & 7)
  const int zwr = (l31 * 8) * 4;

  f32x16 acc[4];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;

#define WINO_STEP(BOFF, S, SET, DO_FILL, FOFF, CHN)                                                  \
  do {                                                                                              \
    float4 rr[4];                                                                                   \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                 \
      const float4 d1 = *reinterpret_cast<const float4*>(ldsb + (((rd[0][b >> 1] + (BOFF)) ^ ((S) * 32)) + b * 64)); \
      const float4 d2 = *reinterpret_cast<const float4*>(ldsb + (((rd[1][b >> 1] + (BOFF)) ^ ((S) * 32)) + b * 64)); \
      rr[b] = fma4(sg, d2, d1);                                                                     \
    }                                                                                               \
    float4 v[4];                                                                                    \
    v[0] = sub4(rr[0], rr[2]); v[1] = add4(rr[1], rr[2]); v[2] = sub4(rr[2], rr[1]); v[3] = sub4(rr[1], rr[3]); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                 \
      _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                              \
        acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wr[SET][nu], t), f4e(v[nu], t), acc[nu], 0, 0, 0); \
      /* the next stage's DMA pieces are issued two at a time behind the four MFMA groups of step 0 (a DMA issue costs  */ \
      /* the wave ~100 cycles: all of them at the stage's top stalled its own MFMA stream: -1 %) and have step 1 to land */ \
      if ((S) == 0 && (DO_FILL)) { __builtin_amdgcn_sched_barrier(0); WINO_FILL1(2 * t, FOFF, CHN); WINO_FILL1(2 * t + 1, FOFF, CHN); __builtin_amdgcn_sched_barrier(0); } \
    }                                                                                               \
  } while (0)

  // Stage s = (channel block s / nch, 16 channels s % nch) lives in buffer s & 1; its DMA and the weights of its first
  // step were issued one stage ahead -- also across the end of a channel block.  One barrier per stage (32 MFMAs per wave).
#ifdef HANDS_WINO_FULL_BARRIERS      // A/B switch: the round-5 barriers
#define WINO_LDS_BARRIER() __syncthreads()
#else
#define WINO_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
#endif
  const int nch = a.Cin >> 4;
  const int nst = nch * a.nbw;
  __syncthreads();                       // (the compiler's fence waits for this wave's DMA)
  int ch = 0, nbi = 0;
  for (int s = 0; s < nst; ++s) {
    const int boff = (s & 1) * (G::BUF_FLOATS * 4);            // bytes (a multiple of 256: the ^ 32 of step 1 is unaffected)
    const int foff = ((s & 1) ^ 1) * G::BUF_FLOATS;
    const int chn = ch + 1 == nch ? 0 : ch + 1;
    const bool fillnext = s + 1 < nst;
    WINO_LOADW(1, 2 * s + 1);
    __builtin_amdgcn_sched_barrier(0);                         // (hipcc otherwise sinks the loads to their first use)
    WINO_STEP(boff, 0, 0, fillnext, foff, chn);
    WINO_LOADW(0, s + 1 < nst ? 2 * s + 2 : 2 * s + 1);        // (the last stage re-loads a valid step: no branch)
    __builtin_amdgcn_sched_barrier(0);
    WINO_STEP(boff, 1, 1, fillnext, foff, chn);
    __syncthreads();
    ch = chn;
    if (ch != 0) continue;

    // ---- end of a channel block: nu half of A^T . A in registers, xi half through this stage's (now idle) patch buffer in
    //      two rounds (output columns j = 0, 1), bias + activation, 16-byte NHWC stores.  Accumulator register r of a lane:
    //      channel 8 (r >> 2) + 4 half + (r & 3) of tile l31.  The next stage's patch is already arriving in the other buffer.
    float* sZ = lds + (s & 1) * G::BUF_FLOATS;
    const int n = (nb0 + nbi) * 32 + cq * 4;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 z;
        if (j == 0) {
          z.x = (acc[0][4 * g + 0] + acc[1][4 * g + 0]) + acc[2][4 * g + 0];
          z.y = (acc[0][4 * g + 1] + acc[1][4 * g + 1]) + acc[2][4 * g + 1];
          z.z = (acc[0][4 * g + 2] + acc[1][4 * g + 2]) + acc[2][4 * g + 2];
          z.w = (acc[0][4 * g + 3] + acc[1][4 * g + 3]) + acc[2][4 * g + 3];
        } else {
          z.x = (acc[1][4 * g + 0] - acc[2][4 * g + 0]) - acc[3][4 * g + 0];
          z.y = (acc[1][4 * g + 1] - acc[2][4 * g + 1]) - acc[3][4 * g + 1];
          z.z = (acc[1][4 * g + 2] - acc[2][4 * g + 2]) - acc[3][4 * g + 2];
          z.w = (acc[1][4 * g + 3] - acc[2][4 * g + 3]) - acc[3][4 * g + 3];
        }
        *reinterpret_cast<float4*>(sZ + xi * 1024 + zwr + (((2 * g + half) ^ (l31 & 7)) * 4)) = z;
      }
      // LDS-only barriers in the two rounds (round 6): a __syncthreads() here also waits for the previous round's global stores and
      // for the next stage's weights / DMA to land (vmcnt(0)) -- the stage barrier above and below is where those are needed
      WINO_LDS_BARRIER();
      if (o_ok && (j == 0 || o_col1)) {
        const float4 z0 = *reinterpret_cast<const float4*>(sZ + 0 * 1024 + zpos);
        const float4 z1 = *reinterpret_cast<const float4*>(sZ + 1 * 1024 + zpos);
        const float4 z2 = *reinterpret_cast<const float4*>(sZ + 2 * 1024 + zpos);
        const float4 z3 = *reinterpret_cast<const float4*>(sZ + 3 * 1024 + zpos);
        float4 y0 = add4(add4(add4(z0, z1), z2), bv);
        float4 y1 = add4(sub4(sub4(z1, z2), z3), bv);
        if (a.act == HANDS_ACT_RELU) {
          y0.x = fmaxf(y0.x, 0.f); y0.y = fmaxf(y0.y, 0.f); y0.z = fmaxf(y0.z, 0.f); y0.w = fmaxf(y0.w, 0.f);
          y1.x = fmaxf(y1.x, 0.f); y1.y = fmaxf(y1.y, 0.f); y1.z = fmaxf(y1.z, 0.f); y1.w = fmaxf(y1.w, 0.f);
        } else if (a.act == HANDS_ACT_LEAKY_RELU) {
          y0.x = y0.x > 0.f ? y0.x : 0.01f * y0.x; y0.y = y0.y > 0.f ? y0.y : 0.01f * y0.y;
          y0.z = y0.z > 0.f ? y0.z : 0.01f * y0.z; y0.w = y0.w > 0.f ? y0.w : 0.01f * y0.w;
          y1.x = y1.x > 0.f ? y1.x : 0.01f * y1.x; y1.y = y1.y > 0.f ? y1.y : 0.01f * y1.y;
          y1.z = y1.z > 0.f ? y1.z : 0.01f * y1.z; y1.w = y1.w > 0.f ? y1.w : 0.01f * y1.w;
        }
        float* o = o_base + (size_t)nbi * 32 + (size_t)j * a.out_ps;
        *reinterpret_cast<float4*>(o) = y0;
        if (o_row1) *reinterpret_cast<float4*>(o + (size_t)a.W * a.out_ps) = y1;
      }
      WINO_LDS_BARRIER();                 // round 1 reuses the region; after round 1 the next stage's DMA may overwrite it
    }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
    ++nbi;
  }
#undef WINO_LDS_BARRIER
#undef WINO_STEP
#undef WINO_LOADW
#undef WINO_FILL
}

}  // namespace

// CU count of the CURRENT device (a launch goes to the device its stream belongs to = the caller's current device), cached per
// device ordinal; relaxed atomics: concurrent first calls store the same value.
static int wino_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Developer overrides (tests sweep them): read once per process.
static int wino_env_int(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}
static int wino_env_nbw() { static const int v = wino_env_int("HANDS_WINO_NBW"); return v; }
static int wino_env_sgs() { static const int v = wino_env_int("HANDS_WINO_SGS"); return v; }

static void wino_magic(int d, uint32_t& mul, uint32_t& sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  int l = 0;
  while ((1LL << l) < d) ++l;                                   // l = ceil(log2 d) >= 1
  mul = (uint32_t)(((1ULL << (31 + l)) + (uint64_t)d - 1) / (uint64_t)d);
  sh = (uint32_t)(l - 1);                                       // (x * mul) >> (31 + l) = umulhi(x, mul) >> (l - 1)
}

template <int D, bool LINEAR, int VSH = 0>
static int wino_launch(WinoArgs& a, hipStream_t stream) {
  using G = WinoGeom<D, LINEAR>;
  const long long rows = a.rows;
  if (rows * a.nw >= 0x7fffff00LL) return HANDS_EINVAL;
  long long nblk_m;
  if (LINEAR) {
    nblk_m = (rows * (D << VSH) + 31) / 32;
    a.nseg = 1;
  } else {
    a.nseg = (a.nw + D - 1) / D;
    nblk_m = (rows + G::NR - 1) / G::NR * a.nseg;
  }
  a.nblk_n = a.Cout / 32;
  if (nblk_m <= 0 || nblk_m * a.nblk_n > 0x7fffffffLL) return HANDS_EINVAL;
  a.nblk_m = (int)nblk_m;
  // Channel blocks per workgroup: a workgroup pays ~4 k cycles of tile setup once and ~4 k per channel block (epilogue)
  // next to ~6.9 k per 16-channel stage (tools/prof_wino.py), so short-K layers want several channel blocks per
  // workgroup -- but fewer, longer workgroups quantise worse on the chip's 3 x CUs slots, and a workgroup that walks several
  // channel blocks re-reads its patch from the Infinity Cache for each (the 128-channel layer: equal time, 2.2x the fabric
  // reads with 4 blocks per workgroup).  Cheapest divisor of nblk_n under that model, the smaller one on ties (a function of
  // the launch geometry only: the arithmetic and its order never depend on it).
  const long long slots = 3LL * wino_device_cus();
  const double nch = a.Cin / 16;
  double best = 0.0;
  a.nbw = 1;
  for (int w = 1; w <= a.nblk_n; ++w) {
    if (a.nblk_n % w) continue;
    const long long wgs = nblk_m * (a.nblk_n / w);
    const double cost = (double)((wgs + slots - 1) / slots) * (4.0 + w * (6.9 * nch + 4.0));
    if (w == 1 || cost < 0.99 * best) { best = cost; a.nbw = w; }
  }
  if (const int w = wino_env_nbw(); w >= 1 && a.nblk_n % w == 0) a.nbw = w;   // developer override (must divide Cout / 32)
  const int ngrp = a.ngrp = a.nblk_n / a.nbw;
  const long long nwg = nblk_m * ngrp;
  wino_magic(a.nh, a.nh_mul, a.nh_sh);
  a.sgs = 1;
  for (int g = 1; g <= ngrp; ++g)                               // largest divisor of ngrp whose weights are <= 2 MB
    if (ngrp % g == 0 && (long long)g * a.nbw * 2048 * a.Cin <= (2LL << 20)) a.sgs = g;
  if (const int g = wino_env_sgs(); g >= 1 && ngrp % g == 0) a.sgs = g;       // developer override (must divide the group count)
  wino_magic(a.sgs, a.sgs_mul, a.sgs_sh);
  wino_magic(a.nblk_m, a.nbm_mul, a.nbm_sh);
  wino_magic(a.nseg, a.nseg_mul, a.nseg_sh);
  // 32-bit byte offsets from the first image a block touches
  const long long imgs = G::NR / a.nh + 2;
  if (imgs * a.H * a.W * a.in_ps * 4 >= 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL((conv_wino_f32_kernel<D, LINEAR, VSH>), dim3((unsigned)nwg), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

// The 32-bit offset / grid limits of wino_launch for ANY of its block geometries (NR <= 8 tile rows per block): a layer
// that passes here launches; one that does not takes the direct kernel (hands_conv3x3_winograd_supported() == 0).
static bool wino_sizes_ok(const hands_conv_desc* d) {
  const long long nh = (d->H + 1) / 2, nw = (d->W + 1) / 2, rows = (long long)d->B * nh;
  if (rows >= 0x7fffff00LL || rows * nw >= 0x7fffff00LL) return false;
  const long long nblk_m_max = (rows * 14 + 31) / 32 + rows * ((nw + 3) / 4);      // >= every geometry's block count
  if (nblk_m_max * (d->Cout / 32) > 0x7fffffffLL) return false;
  const long long imgs = 8 / nh + 2;
  const long long ps = d->in_pix_stride > d->out_pix_stride ? d->in_pix_stride : d->out_pix_stride;
  return imgs * d->H * d->W * ps * 4 < 0x7fffffffLL;
}

static bool wino_ok(const hands_conv_desc* d) {
  const int act = d->act & HANDS_ACT_MASK;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return false;       // before wino_sizes_ok(): it divides by the tile-row count
  return wino_sizes_ok(d) && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->H && d->Wo == d->W &&
         d->Cin >= 16 && d->Cin % 16 == 0 && d->Cout >= 32 && d->Cout % 32 == 0 && d->in_pix_stride >= d->Cin &&
         d->out_pix_stride >= d->Cout && d->in_pix_stride % 4 == 0 && d->out_pix_stride % 4 == 0 &&
         (act == HANDS_ACT_NONE || act == HANDS_ACT_RELU || act == HANDS_ACT_LEAKY_RELU) && !(d->act & HANDS_MATH_BF16X3);
}

extern "C" int hands_conv3x3_winograd_supported(const hands_conv_desc* d) { return d && wino_ok(d) ? 1 : 0; }

// Multiply-accumulates the matrix cores really execute for this layer (idle tile lanes included): 16 frequencies x 32-tile
// blocks x Cin x Cout.  The algorithmic count of the layer is 9 * H * W * Cin * Cout per image: bench.py reports both.
extern "C" long long hands_conv3x3_winograd_executed_macs(const hands_conv_desc* d) {
  if (!d || !wino_ok(d)) return 0;
  const long long nh = (d->H + 1) / 2, nw = (d->W + 1) / 2, rows = (long long)d->B * nh;
  long long nblk_m;
  if (nw == 7 || nw == 14) nblk_m = (rows * nw + 31) / 32;
  else if (nw % 4 == 0 || nw < 8) nblk_m = (rows + 7) / 8 * ((nw + 3) / 4);
  else nblk_m = (rows + 3) / 4 * ((nw + 7) / 8);
  return 16LL * nblk_m * 32 * d->Cin * d->Cout;
}

extern "C" int hands_conv3x3_winograd_f32(const hands_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                                          float* out, hands_stream_t stream) {
  if (!d || !in || !u_packed || !bias || !out || !wino_ok(d)) return HANDS_EINVAL;
  if ((((uintptr_t)in) | ((uintptr_t)u_packed) | ((uintptr_t)bias) | ((uintptr_t)out)) & 15) return HANDS_EINVAL;   // 16-byte accesses
  WinoArgs a;
  a.in = in; a.u = u_packed; a.bias = bias; a.out = out;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
  a.nh = (d->H + 1) / 2; a.nw = (d->W + 1) / 2;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.act = d->act & HANDS_ACT_MASK;
  if ((long long)d->B * a.nh >= 0x7fffff00LL) return HANDS_EINVAL;
  a.rows = d->B * a.nh;
  a.nblk_m = a.nblk_n = a.nseg = a.nbw = a.ngrp = a.sgs = 0;
  hipStream_t s = (hipStream_t)stream;
  if (a.nw == 7) return wino_launch<7, true>(a, s);
  if (a.nw == 14) return wino_launch<7, true, 1>(a, s);
  if (a.nw % 4 == 0 || a.nw < 8) return wino_launch<4, false>(a, s);
  return wino_launch<8, false>(a, s);
}
