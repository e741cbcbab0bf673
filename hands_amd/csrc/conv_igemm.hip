// conv_igemm.hip -- NHWC convolution / linear layer as an implicit GEMM on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// GEMM view:  D[n, m] = sum_k W[n, k] * X[m, k]      (n = output channel, m = output pixel)
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so the
// accumulator of a lane holds 4 CONSECUTIVE output channels of one pixel -> 16-byte NHWC stores.
//
// Block tile = (64*WAVES_M pixels) x (64*WAVES_N channels), 4 waves, each wave a 64x64 tile made of
// 2x2 MFMA 32x32 blocks (64 accumulator registers).  K is walked in steps of 16 floats (64 B per
// row); tiles are staged global -> registers -> LDS with a two-deep LDS ring, the global loads of
// step t+1 in flight under the 32 MFMAs (2048 matrix-pipe cycles) of step t.  LDS rows are one
// k-step (16 floats, unpadded) with the four 16-byte chunks of row r XOR-swizzled by (r >> 2) & 3, which keeps
// both the ds_write_b128 staging writes and the ds_read_b128 fragment reads bank-conflict free.  Within a 16-wide k-step the k index is permuted (lane half h reads k =
// 8*kk + 4*h + t for MFMA t): both operands use the same permutation, so the sum is unchanged.
//
// Replaces: F.conv2d + eval BatchNorm2d (folded) + ReLU + residual add of
// src/nets/backbone/resnet.py:134-154,264-280; feature_conv (src/models/hands_light/model.py:91-101);
// every nn.Linear of the heads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// nn.GELU() (exact, erf form): x * 0.5 * (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float4 apply_act(float4 v, int act) {
  if (act == HANDS_ACT_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  } else if (act == HANDS_ACT_GELU) {
    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
  } else if (act == HANDS_ACT_LEAKY_RELU) {
    v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
    v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
  }
  return v;
}

__device__ __forceinline__ float f4elem(const float4& v, int t) {
  return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w));
}

constexpr int BK = 16;         // floats per k-step
constexpr int LDS_ROW = 16;    // LDS row (floats) = one k-step, unpadded; the 16-byte chunk c of row r sits at c ^ ((r >> 2) & 3):
                               // ds_write_b128 (8 contiguous lanes = 2 whole rows) and ds_read_b128 (its 16-lane groups of
                               // rows {0-3,12-15,20-27} / {4-11,16-19,28-31}) are both conflict-free, and the 256 x 64 tile's
                               // ring is 40 KB (4 workgroups per CU) instead of 50
constexpr int LDS_ROW_B3 = 24; // bf16x3 mode: three bf16 planes of 16 k (3 x 32 B), no pad: 2-way read conflicts, but 49 KB
                               // per block = 3 blocks per CU (measured +8 % over the padded 112-B rows at 2 blocks per CU)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 = b0 + b1 + b2 EXACTLY, each a bf16 (8 significant bits): truncation takes the top 8 bits, the exact
// remainder has at most 16, then 8 significant bits.  (inf / NaN inputs give NaN.)
__device__ __forceinline__ void split3(float x, uint32_t& u0, uint32_t& u1, uint32_t& u2) {
  u0 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(u0);
  u1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(u1);
  u2 = __float_as_uint(r2);          // <= 8 significant bits: its low half is zero already
}

// four consecutive k of one row -> three planes of 4 bf16 (8 B each) at dst + plane * 8 floats
__device__ __forceinline__ void store_split4(float* dst, float4 v) {
  uint32_t a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
  split3(v.x, a0, a1, a2); split3(v.y, b0, b1, b2); split3(v.z, c0, c1, c2); split3(v.w, d0, d1, d2);
  *reinterpret_cast<uint2*>(dst) = make_uint2((a0 >> 16) | b0, (c0 >> 16) | d0);
  *reinterpret_cast<uint2*>(dst + 8) = make_uint2((a1 >> 16) | b1, (c1 >> 16) | d1);
  *reinterpret_cast<uint2*>(dst + 16) = make_uint2((a2 >> 16) | (b2 & 0xffff0000u), (c2 >> 16) | (d2 & 0xffff0000u));
}

struct ConvArgs {
  const float* __restrict__ in;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  const float* res;   // may alias out (in-place residual update of the HMR state)
  float* out;
  int M, N, Kpad;
  int H, W, Cin, Ho, Wo, KH, KW, stride, pad;
  int in_ps, out_ps, res_ps;
  int relu;
  int nblk_m, nblk_n;
  const float* __restrict__ in2;   // MODE 2: second source, 1x1 sampled with stride2
  int K0, H2, W2, stride2, in2_ps;
  int ksplit;         // > 1: deterministic split-K, grid = tiles * ksplit, raw partial sums to `partial`
  float* partial;     // [ksplit][M][part_ps]
  int part_ps;
  // PRE instantiations (pointwise layers): the activation operand is leaky_relu(x * pre_scale[k] + pre_shift[k], 0.01) --
  // the eval BatchNorm -> LeakyReLU that PRECEDES the first convolution of a pre-activation residual unit
  // (handoccnet_light/hand_head.py:131-133,170-172), applied on the way into LDS instead of by a launch of its own
  const float* pre_scale;
  const float* pre_shift;
  // fused split-K reduction (round 6): per-tile arrival counters (zero before the launch, left zero by it).  The LAST slice of a
  // tile to arrive adds the partial sums in ascending slice order and runs the epilogue -- no second launch; nullptr = the
  // reduce kernel does it.  fin_res / fin_act: the epilogue's residual and activation (a.res / a.relu are off in split-K slices)
  int* counters = nullptr;
  const float* fin_res = nullptr;
  int fin_act = 0;
};

// XCD-aware block remap: blocks are dispatched round-robin over the 8 XCDs (private L2 each);
// give every XCD a contiguous range of tiles so that the n-tiles of one m-tile share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Epilogue of a FULL tile (no row / channel predicates) with act in {NONE, RELU}: the common case of every trunk
// layer, as straight-line code -- 16 LDS writes, then per output row one LDS read, packed bias (+ residual) adds,
// the clamp, one 64-bit address add and the store.  The general path below (runtime activation switch, per-row
// predicates) executes ~4x the instructions, and a workgroup in its epilogue holds a wave slot per SIMD without
// feeding the matrix pipe: with K = 256 a tile spent 12 of its 72 us there (tools/prof_tile.py).
// Same arithmetic as the general path: (acc + bias) + residual, then max(., 0).
constexpr int EROW = 68;                            // 64 channels + 4 pad floats (17 slots: odd)
constexpr int EPI_FLOATS = 4 * 32 * EROW;           // the epilogue's transposition buffers (one per wave) reuse the ring
template <int ACT, bool RES>
__device__ __forceinline__ void epilogue_full(const f32x16 (&acc)[2][2], float* sE, int lane, const float* bias_p,
                                              const char* res_u, int res_ps, char* out_u, int out_ps) {
  // res_u / out_u: wave-uniform byte address of this wave's (pixel 0, channel 0); the lane part is one 32-bit
  // offset, so every access is "scalar base + vector offset" and the row stepping costs no vector instructions
  const int half = lane >> 5;
  float* wr = sE + (lane & 31) * EROW + half * 4;
  const float* rd = sE + (lane >> 4) * EROW + (lane & 15) * 4;
  const uint32_t o_off = (uint32_t)((lane >> 4) * out_ps + (lane & 15) * 4) * 4u;
  const uint32_t r_off = (uint32_t)((lane >> 4) * res_ps + (lane & 15) * 4) * 4u;
  const float4 bv4 = *reinterpret_cast<const float4*>(bias_p);
#ifndef HANDS_EPI_SCALAR_ADDS
  const f32x2 b01 = {bv4.x, bv4.y}, b23 = {bv4.z, bv4.w};
#endif
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(wr + i * 32 + q * 8) =
            make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
    float4 rv[RES ? 8 : 1];
    if constexpr (RES) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        rv[r] = *reinterpret_cast<const float4*>(res_u + (size_t)(j * 32 + 4 * r) * (size_t)res_ps * 4 + r_off);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 t = *reinterpret_cast<const float4*>(rd + 4 * r * EROW);
#ifdef HANDS_EPI_SCALAR_ADDS
      // A/B variant (tools/build_variant.sh -DHANDS_EPI_SCALAR_ADDS): four scalar v_add_f32 per operand instead of two
      // v_pk_add_f32 (MI355X_MICROARCH.md lists packed f32 VALU beside MFMAs as an anti-lever); same arithmetic
      float4 v = make_float4(t.x + bv4.x, t.y + bv4.y, t.z + bv4.z, t.w + bv4.w);
      if constexpr (RES) { v.x += rv[r].x; v.y += rv[r].y; v.z += rv[r].z; v.w += rv[r].w; }
      asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));     // keep the SLP vectoriser from re-packing them
#else
      f32x2 v01 = {t.x, t.y}, v23 = {t.z, t.w};
      v01 = v01 + b01;
      v23 = v23 + b23;
      if constexpr (RES) {
        const f32x2 r01 = {rv[r].x, rv[r].y}, r23 = {rv[r].z, rv[r].w};
        v01 = v01 + r01;
        v23 = v23 + r23;
      }
      float4 v = make_float4(v01.x, v01.y, v23.x, v23.y);
#endif
      if constexpr (ACT != HANDS_ACT_NONE) v = apply_act(v, ACT);       // compile-time activation: no runtime switch
      *reinterpret_cast<float4*>(out_u + (size_t)(j * 32 + 4 * r) * (size_t)out_ps * 4 + o_off) = v;
      // rows in pairs: without the fence the scheduler hoists all eight LDS reads (+32 live registers on top of
      // the 64 accumulators and 32 residual values: 160+ VGPRs, 3 waves per SIMD instead of 4)
      if (r & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Split-K without the second launch: called by every slice of a tile after its partial sums are stored.  The slice that arrives
// last (agent-scope counter; release before the count, acquire after it -- the stream-K hand-off's protocol) reduces the tile:
// partial sums in ASCENDING slice order whoever arrives last, then bias, residual, activation -- operation for operation what
// splitk_reduce_kernel / splitk_reduce_f64_kernel do, so the output bits are those of the two-launch form.
template <int BM, int BN, bool F64>
__device__ __forceinline__ void splitk_fused_tail(const ConvArgs& a, float* lds, int tile, int m0, int n0, int wave_u) {
  // thread id from the wave index (an SGPR the caller saved before the tile) and mbcnt: reading threadIdx.x here would keep v0
  // alive through the whole tile -- one register too many for the 256 x 64 tile's four workgroups per CU
  const int tid = wave_u * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  // (the flag lives in the first word of the staging ring, idle by now: a __shared__ int of its own made the 256 x 64 tile's
  //  40 KB ring 40 964 bytes = three workgroups per CU instead of four)
  volatile int& s_last = *reinterpret_cast<volatile int*>(lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's partial stores have left
  __syncthreads();                                      // ... and every wave is done with the transposition buffers
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int old = __hip_atomic_fetch_add(a.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == a.ksplit - 1;
    if (last) __hip_atomic_store(a.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const size_t slice = (size_t)a.M * a.part_ps;
  if constexpr (F64) {
    const double* part = reinterpret_cast<const double*>(a.partial);
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int m = m0 + idx / BN, n = n0 + idx % BN;
      if (m >= a.M || n >= a.N) continue;
      const double* p = part + (size_t)m * a.part_ps + n;
      double v = p[0];
      for (int sl = 1; sl < a.ksplit; ++sl) v += p[sl * slice];
      float f = (float)(v + (double)a.bias[n]);
      if (a.fin_res) f += a.fin_res[(size_t)m * a.res_ps + n];
      if (a.fin_act == HANDS_ACT_RELU) f = fmaxf(f, 0.f);
      else if (a.fin_act == HANDS_ACT_GELU) f = gelu_erf(f);
      else if (a.fin_act == HANDS_ACT_LEAKY_RELU) f = f > 0.f ? f : 0.01f * f;
      a.out[(size_t)m * a.out_ps + n] = f;
    }
  } else {
    constexpr int C4 = BN / 4;
    for (int idx = tid; idx < BM * C4; idx += 256) {
      const int m = m0 + idx / C4, n = n0 + (idx % C4) * 4;
      if (m >= a.M || n >= a.N) continue;
      const float* p = a.partial + (size_t)m * a.part_ps + n;
      float4 v = *reinterpret_cast<const float4*>(p);
      for (int sl = 1; sl < a.ksplit; ++sl) {
        const float4 q = *reinterpret_cast<const float4*>(p + sl * slice);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      if (a.fin_res) {
        const float4 r = *reinterpret_cast<const float4*>(a.fin_res + (size_t)m * a.res_ps + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ps + n) = apply_act(v, a.fin_act);
    }
  }
}

// MODE 0: any convolution with Cin % 16 == 0 (k-steps ordered channel chunk, kh, kw); 1: the stem (Cin == 4, one filter tap per 16-byte chunk);
// 2: two 1x1 convolutions summed into one output (k < K0 from `in`, the rest from `in2` sampled with
//    its own stride) -- the last conv of a bottleneck fused with the block's downsample branch.
constexpr int SK_SLOT_FLOATS = 4 * 64 * 64;   // one workgroup's accumulators: 4 waves x 64 registers x 64 lanes

// One output tile (BM pixels x BN channels), k-steps [kt0, kt1).  acc_in != nullptr: the accumulators continue a
// chain another workgroup started (stream-K hand-off, register layout); acc_out != nullptr: dump the raw
// accumulators there instead of running the epilogue.  `split` is the slice index of the split-K form.
// PREC 0: exact fp32 (v_mfma_f32_32x32x2_f32).  PREC 1 ("bf16x3", a separately reported mode, never the default):
// both operands are split on the way into LDS into three bf16 planes whose sum is the fp32 value exactly, and a
// k-16 step is six v_mfma_f32_32x32x16_bf16 (the products b_i * b_j with i + j <= 2, fp32 accumulation): the
// dropped terms are <= 2^-24 of a product -- fp32-grade results at 3/8 of the matrix-pipe time.
// PREC 2 (HANDS_ACC_F64): the same fp32 operands and staging, but the fragments are widened to fp64 on their way out of LDS and
// the products are accumulated by v_mfma_f64_16x16x4_f64 (a wave's 64 x 64 tile = 4 x 4 blocks, 128 accumulator registers, two
// workgroups per CU, half the fp32 matrix rate): products of fp32 values are exact in fp64 and the sum carries 53 bits, so the
// stored output is the correctly rounded fp32 of (sum + bias) -- no accumulation error at all.  For the few layers whose rounding
// the network amplifies (handoccnet_light's heat-map head: DESIGN.md "Conditioning note"), not for throughput.
// BLK > 0 (8 or 4 k-steps = 128 / 64 floats): blocked summation -- after every BLK k-steps the accumulators are added to a
// second set and cleared, so no fp32 FMA chain is longer than the block (HANDS_SUM_BLOCK*; 64 more registers: these
// instantiations run two workgroups per CU instead of four).
__device__ __forceinline__ void tile_coords(const ConvArgs& a, int tile, int& mt, int& nt) {
  constexpr int RASTER_GM = 8;
  if (a.nblk_n >= 8 && a.nblk_m >= RASTER_GM) {
    const int per = RASTER_GM * a.nblk_n;
    const int gidx = tile / per, idx = tile - gidx * per;
    const int gm = min(RASTER_GM, a.nblk_m - gidx * RASTER_GM);
    nt = idx / gm; mt = gidx * RASTER_GM + (idx - nt * gm);
  } else { mt = tile / a.nblk_n; nt = tile - mt * a.nblk_n; }
}

template <int WAVES_M, int WAVES_N, int MODE, int PREC = 0, bool PRE = false, int BLK = 0>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, float* lds, int tile, int split, int kt0, int kt1,
                                          const float* acc_in, float* acc_out) {
  constexpr int ROW = PREC == 1 ? LDS_ROW_B3 : LDS_ROW;
  constexpr bool F64 = PREC == 2;
  constexpr bool STEM = MODE == 1;
  constexpr bool DUAL = MODE == 2;
  constexpr int BM = 64 * WAVES_M;
  constexpr int BN = 64 * WAVES_N;
  constexpr int A_ROWS = BM / 64;   // activation rows staged per thread
  constexpr int W_ROWS = BN / 64;   // weight rows staged per thread
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");

  float* sX = lds;                          // [2][BM][ROW]
  float* sW = lds + 2 * BM * ROW;       // [2][BN][ROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;            // wave row (pixels)
  const int wn = wave % WAVES_N;            // wave col (channels)

  // Tile order inside an XCD's contiguous range.  Few n-tiles: n fastest (consecutive tiles share the pixel rows).  Wide
  // layers (>= 8 n-tiles: the ViT GEMMs, N = 1280-5120, and the 1024 / 2048-channel expand convolutions): bands of
  // RASTER_GM = 8 m-tiles with m fastest inside a band, so the ~128 tiles resident on an XCD at one time form an
  // 8 x 16 block of the output (24 operand panels) instead of 4 m-rows x all n-tiles (34+): measured on hamer_light
  // bz=64: fabric reads per GEMM launch 2.50 -> 2.04 GB, speed unchanged (the Infinity Cache serves the re-reads
  // either way); hands_light / handoccnet_light neutral.  Any bijection gives the same output bits.
  int mt, nt;
  tile_coords(a, tile, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging assignment: row = (tid>>2) + 64*i, 16-byte chunk = tid&3 ------------
  const int srow = tid >> 2;
  const int chunk = tid & 3;
  const int swz_chunk = chunk ^ ((srow >> 2) & 3);     // PREC 0 ring position of this thread's 16-byte chunk (LDS_ROW)
  int x_base[A_ROWS];     // float offset of pixel (b, ho*s-pad, wo*s-pad) channel 0 (may be "negative")
  int x_hi0[A_ROWS], x_wi0[A_ROWS];
  bool x_ok[A_ROWS];
  int x_base2[DUAL ? A_ROWS : 1];
  // MODE 0 (padded convolution): per row ONE validity word (bit kh: row hi0 + kh inside the image, bit 16 + kw:
  // column wi0 + kw inside; 0 for rows past M) and a byte offset relative to a wave-uniform base.  A k-step then
  // costs and / compare / add / select per row, and the load is a raw-buffer load whose out-of-range offset
  // (0x80000000 >= num_records) returns zeros in hardware: no predicated branches, no zero fill in registers
  uint32_t x_mask[MODE == 0 ? A_ROWS : 1];
  int x_offb[MODE == 0 ? A_ROWS : 1];
  int x_b[A_ROWS];
  const int HoWo = a.Ho * a.Wo;
  // plain pointwise layer (one source, stride 1): the input pixel IS the output pixel, no (b, ho, wo) decomposition
  const bool same_pixel = DUAL && a.K0 >= a.Kpad && a.stride == 1 && a.H == a.Ho && a.W == a.Wo;
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + srow + 64 * i;
    x_ok[i] = m < a.M;
    const int mm = x_ok[i] ? m : 0;
    if (same_pixel) {
      x_b[i] = 0; x_hi0[i] = 0; x_wi0[i] = 0;
      x_base[i] = mm * a.in_ps;
      if constexpr (DUAL) x_base2[i] = 0;
      continue;
    }
    const int b = mm / HoWo;
    const int rem = mm - b * HoWo;
    const int ho = rem / a.Wo;
    const int wo = rem - ho * a.Wo;
    x_b[i] = b;
    x_hi0[i] = ho * a.stride - a.pad;
    x_wi0[i] = wo * a.stride - a.pad;
    x_base[i] = ((b * a.H + x_hi0[i]) * a.W + x_wi0[i]) * a.in_ps;
    if constexpr (DUAL) {
      x_base2[i] = ((b * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.in2_ps;
    }
  }
  __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)0x80000000u, 0x00020000);
  if constexpr (MODE == 0) {
    // base = first image this wave stages (its lane 0, i = 0 holds the smallest row); rows past M sit on image 0 and
    // have an empty mask, so their (possibly negative) offsets are never used
    const int b_w = __builtin_amdgcn_readfirstlane(x_b[0]);
    x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) + (size_t)b_w * a.H * a.W * a.in_ps, 0,
                                               (int)0x80000000u, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int hb = max(0, -x_hi0[i]), he = min(a.KH, a.H - x_hi0[i]);
      const int wb = max(0, -x_wi0[i]), we = min(a.KW, a.W - x_wi0[i]);
      const uint32_t hm = he > hb ? ((1u << he) - 1u) & ~((1u << hb) - 1u) : 0u;
      const uint32_t wm = we > wb ? ((1u << we) - 1u) & ~((1u << wb) - 1u) : 0u;
      x_mask[i] = (x_ok[i] && hm != 0u && wm != 0u) ? (hm | (wm << 16)) : 0u;
      x_offb[i] = (((((x_b[i] - b_w) * a.H + x_hi0[i]) * a.W + x_wi0[i]) * a.in_ps) + chunk * 4) * 4;
    }
  }
  // MODE 2 (pointwise / two-source): byte offsets relative to the first row this wave stages, one descriptor per
  // source; the k position is the instruction's scalar offset, so a k-step has no address arithmetic at all
  int x_off1[DUAL ? A_ROWS : 1], x_off2[DUAL ? A_ROWS : 1];
  __amdgpu_buffer_rsrc_t x_rsrc2 = x_rsrc;
  if constexpr (DUAL) {
    const int base1 = __builtin_amdgcn_readfirstlane(x_base[0]);
    const int base2 = __builtin_amdgcn_readfirstlane(x_base2[0]);
    x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) + base1, 0, (int)0x80000000u, 0x00020000);
    x_rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2) + base2, 0, (int)0x80000000u, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {      // rows past M read the wave's first row (their outputs are never stored)
      x_off1[i] = x_ok[i] ? (x_base[i] - base1 + chunk * 4) * 4 : chunk * 16;
      x_off2[i] = x_ok[i] ? (x_base2[i] - base2 + chunk * 4) * 4 : chunk * 16;
    }
  }
  // weights: rows of this tile relative to its first row, k position as the scalar offset
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w) + (size_t)n0 * a.Kpad, 0, (int)0x80000000u, 0x00020000);
  int w_offb[W_ROWS];
#pragma unroll
  for (int i = 0; i < W_ROWS; ++i) w_offb[i] = ((srow + 64 * i) * a.Kpad + chunk * 4) * 4;

  // staging registers (explicit scalars-of-float4: arrays captured by lambdas ended up in scratch)
  float4 xr[A_ROWS], wr[W_ROWS];
  static_assert(!PRE || (MODE == 2 && PREC != 1), "the input affine + LeakyReLU exists for exact-fp32 pointwise layers");
  float4 psr = make_float4(1.f, 1.f, 1.f, 1.f), pbr = make_float4(0.f, 0.f, 0.f, 0.f);   // PRE: scale / shift of the staged k chunk
  // k-step state (wave-uniform for the regular path)
  int kh = 0, kw = 0, c0 = 0, woff = 0;
  const int ntaps = a.KH * a.KW;
  const int cw = (a.Cin & 31) == 0 ? 2 * BK : BK;   // channel chunk of the MODE 0 k order: one 128-B line when possible

#define LOAD_TILES(KT)                                                                              \
  do {                                                                                              \
    if constexpr (STEM) {                                                                           \
      /* Cin == 4: every 16-byte chunk is its own filter tap */                                    \
      const int tap = (KT) * 4 + chunk;                                                             \
      const int tkh = tap / a.KW;                                                                   \
      const int tkw = tap - tkh * a.KW;                                                             \
      const int toff = (tkh * a.W + tkw) * a.in_ps;                                                 \
      _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                          \
        const bool ok = x_ok[i] && tap < ntaps && (unsigned)(x_hi0[i] + tkh) < (unsigned)a.H &&     \
                        (unsigned)(x_wi0[i] + tkw) < (unsigned)a.W;                                 \
        xr[i] = ok ? *reinterpret_cast<const float4*>(a.in + (x_base[i] + toff))                    \
                   : make_float4(0.f, 0.f, 0.f, 0.f);                                               \
      }                                                                                             \
    } else if constexpr (DUAL) {                                                                    \
      /* wave-uniform source select (k < K0: `in`, else `in2` sampled with its own stride).  Rows   */ \
      /* past M were given the address of row 0 above: load unconditionally (their outputs are     */ \
      /* never stored), which keeps the k-loop free of branches                                    */ \
      const bool second = (KT) * BK >= a.K0;                                                        \
      const int soff = ((KT) * BK - (second ? a.K0 : 0)) * 4;                                       \
      _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                          \
        const u32x4 ld = second ? __builtin_amdgcn_raw_buffer_load_b128(x_rsrc2, x_off2[i], soff, 0) \
                                : __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, x_off1[i], soff, 0);  \
        xr[i] = make_float4(__uint_as_float(ld.x), __uint_as_float(ld.y), __uint_as_float(ld.z), __uint_as_float(ld.w)); \
      }                                                                                             \
    } else {                                                                                        \
      /* k order (chunk of cw channels, kh, kw, 16-channel step): the taps of one chunk re-read the  */ \
      /* same 64-128 B of every pixel within a few k-steps, while the lines are still in the XCD's  */ \
      /* L2; with taps outermost every tap pass streamed the whole tile again from HBM (8.6x)       */ \
      const bool kvalid = c0 < a.Cin;          /* false only in caller-added zero padding of K */    \
      const uint32_t sel = kvalid ? ((1u << kh) | (0x10000u << kw)) : 0x80000000u;  /* bit 31 is never in a mask */ \
      const int toffb = ((kh * a.W + kw) * a.in_ps + c0) * 4;                                       \
      _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                          \
        const bool ok = (x_mask[i] & sel) == sel;                                                   \
        const u32x4 ld = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? x_offb[i] + toffb : (int)0x80000000u, 0, 0); \
        xr[i] = make_float4(__uint_as_float(ld.x), __uint_as_float(ld.y), __uint_as_float(ld.z), __uint_as_float(ld.w)); \
      }                                                                                             \
      woff = kvalid ? (kh * a.KW + kw) * a.Cin + c0 : (KT) * BK;                                    \
      c0 += BK;                                                                                     \
      if ((c0 & (cw - 1)) == 0) {            /* chunk of cw channels done for this tap: next tap */  \
        c0 -= cw;                                                                                   \
        if (++kw == a.KW) { kw = 0; if (++kh == a.KH) { kh = 0; c0 += cw; } }                       \
      }                                                                                             \
    }                                                                                               \
    if constexpr (MODE != 0) woff = (KT) * BK;                                                      \
    if constexpr (PRE) {                                                                            \
      psr = *reinterpret_cast<const float4*>(a.pre_scale + (KT) * BK + chunk * 4);                  \
      pbr = *reinterpret_cast<const float4*>(a.pre_shift + (KT) * BK + chunk * 4);                  \
    }                                                                                               \
    _Pragma("unroll") for (int i = 0; i < W_ROWS; ++i) {                                            \
      const u32x4 ld = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_offb[i], woff * 4, 0);        \
      wr[i] = make_float4(__uint_as_float(ld.x), __uint_as_float(ld.y), __uint_as_float(ld.z), __uint_as_float(ld.w)); \
    }                                                                                               \
  } while (0)

#define STORE_TILES(BUF)                                                                            \
  do {                                                                                              \
    if constexpr (PRE) {                                                                            \
      _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                          \
        float4 v = make_float4(xr[i].x * psr.x + pbr.x, xr[i].y * psr.y + pbr.y, xr[i].z * psr.z + pbr.z, xr[i].w * psr.w + pbr.w); \
        v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;                   \
        v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;                   \
        xr[i] = v;                                                                                  \
      }                                                                                             \
    }                                                                                               \
    if constexpr (PREC != 1) {                                                                      \
      float* dx = sX + (BUF) * BM * ROW + srow * ROW + swz_chunk * 4;                               \
      float* dw = sW + (BUF) * BN * ROW + srow * ROW + swz_chunk * 4;                               \
      _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i)                                            \
        *reinterpret_cast<float4*>(dx + 64 * i * ROW) = xr[i];                                      \
      _Pragma("unroll") for (int i = 0; i < W_ROWS; ++i)                                            \
        *reinterpret_cast<float4*>(dw + 64 * i * ROW) = wr[i];                                      \
    } else {                                                                                        \
      float* dx = sX + (BUF) * BM * ROW + srow * ROW + chunk * 2;                                   \
      float* dw = sW + (BUF) * BN * ROW + srow * ROW + chunk * 2;                                   \
      _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) store_split4(dx + 64 * i * ROW, xr[i]);    \
      _Pragma("unroll") for (int i = 0; i < W_ROWS; ++i) store_split4(dw + 64 * i * ROW, wr[i]);    \
    }                                                                                               \
  } while (0)

  f64x4 accd[F64 ? 4 : 1][F64 ? 4 : 1];    // PREC 2: [channel block][pixel block], D row = channel (lane >> 4) + 4 r, col = pixel lane & 15
  if constexpr (F64) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) accd[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  }
  f32x16 acc[2][2];
  if (acc_in != nullptr) {        // continue the FMA chain another workgroup started: same bits as an unsplit tile
    const float4* src = reinterpret_cast<const float4*>(acc_in) + (wave * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = src[((i * 2 + j) * 4 + q) * 64];
          acc[i][j][q * 4 + 0] = v.x; acc[i][j][q * 4 + 1] = v.y; acc[i][j][q * 4 + 2] = v.z; acc[i][j][q * 4 + 3] = v.w;
        }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // running tap / source state at the first k-step of this segment
  if constexpr (MODE == 0) {
    if (kt0 > 0) {                  // k-step kt = (chunk * ntaps + tap) * (cw / 16) + sub
      const int spt = cw / BK;
      const int ch0 = kt0 / (ntaps * spt);
      const int r0 = kt0 - ch0 * ntaps * spt;
      const int tap0 = r0 / spt;
      c0 = ch0 * cw + (r0 - tap0 * spt) * BK;
      kh = tap0 / a.KW;
      kw = tap0 - kh * a.KW;
    }
  }
#ifdef HANDS_PRO_PRIO        // A/B variant: the same for the prologue (first loads out as early as possible), back to 0 for the k-loop
  __builtin_amdgcn_s_setprio(HANDS_PRO_PRIO);
#endif
  LOAD_TILES(kt0);
  STORE_TILES(0);
  __syncthreads();
#ifdef HANDS_PRO_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif

  // fragment read offsets: row = lane&31, k half = lane>>5
  // PREC 0: chunk (2 kk + half) of row r is stored at chunk ^ ((r >> 2) & 3); the kk = 1 fragment is the kk = 0 address ^ 32 B
  // PREC 2: row = lane & 15 of a 16-row block, 16-byte chunk lane >> 4 (MFMA t of a k-step takes k = 4 (lane >> 4) + t from both operands)
  const int frag = PREC == 1 ? (lane & 31) * ROW + (lane >> 5) * 4
                   : F64     ? (lane & 15) * ROW + (((lane >> 4) ^ ((lane >> 2) & 3)) * 4)
                             : (lane & 31) * ROW + (((lane >> 5) ^ ((lane >> 2) & 3)) * 4);
  const float* fw = sW + (wn * 64) * ROW + frag;
  const float* fx = sX + (wm * 64) * ROW + frag;
  const float* fw1 = sW + (wn * 64) * ROW + (frag ^ 8);
  const float* fx1 = sX + (wm * 64) * ROW + (frag ^ 8);

#define COMPUTE_STEP(BUF)                                                                           \
  do {                                                                                              \
    if constexpr (F64) {                                                                            \
      float4 wf[4], xf[4]; /* [16-row block]: four consecutive k of the lane's row */                \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        wf[i] = *reinterpret_cast<const float4*>(fw + (BUF) * BN * ROW + i * 16 * ROW);             \
        xf[i] = *reinterpret_cast<const float4*>(fx + (BUF) * BM * ROW + i * 16 * ROW);             \
      }                                                                                             \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                               \
        double wd[4], xd[4];                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
          wd[i] = (double)f4elem(wf[i], t);                                                         \
          xd[i] = (double)f4elem(xf[i], t);                                                         \
        }                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            accd[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(wd[i], xd[j], accd[i][j], 0, 0, 0);   \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
    } else if constexpr (PREC == 0) {                                                               \
      float4 wf[2][2], xf[2][2]; /* [32-row block][kk] */                                           \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
          wf[i][kk] = *reinterpret_cast<const float4*>((kk ? fw1 : fw) + (BUF) * BN * ROW + i * 32 * ROW); \
          xf[i][kk] = *reinterpret_cast<const float4*>((kk ? fx1 : fx) + (BUF) * BM * ROW + i * 32 * ROW); \
        }                                                                                           \
      }                                                                                             \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                             \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                           \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                         \
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4elem(wf[i][kk], t), f4elem(xf[j][kk], t), \
                                                               acc[i][j], 0, 0, 0);                  \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
    } else {                                                                                        \
      /* lane (row = lane & 31, half = lane >> 5) holds k = 8 * half .. + 7 of its row, per plane */  \
      bf16x8 wf[2][3], xf[2][3]; /* [32-row block][plane] */                                        \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
          wf[i][p] = *reinterpret_cast<const bf16x8*>(fw + (BUF) * BN * ROW + i * 32 * ROW + p * 8); \
          xf[i][p] = *reinterpret_cast<const bf16x8*>(fx + (BUF) * BM * ROW + i * 32 * ROW + p * 8); \
        }                                                                                           \
      }                                                                                             \
      /* products b_pa(w) * b_pb(x) with pa + pb <= 2, smallest terms first */                       \
      _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                               \
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};                                                   \
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][PA[q]], xf[j][PB[q]], acc[i][j], 0, 0, 0); \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
  } while (0)

  static_assert(BLK == 0 || ((BLK & (BLK - 1)) == 0 && PREC == 0), "block length: a power of two k-steps, exact fp32 only");
  static_assert(!F64 || (WAVES_M == 2 && WAVES_N == 2), "fp64 accumulation: the 128 x 128 tile only");
  f32x16 tot[BLK ? 2 : 1][BLK ? 2 : 1];
  if constexpr (BLK > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  }
  // steady state: loads of step kt+1 are in flight under the 32 MFMAs of step kt
  for (int kt = kt0; kt + 1 < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    LOAD_TILES(kt + 1);
    COMPUTE_STEP(buf);
    if constexpr (BLK > 0) {
      if (((kt - kt0 + 1) & (BLK - 1)) == 0) {          // (uniform) end of a block: its sum joins the total, in block order
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
      }
    }
    STORE_TILES(buf ^ 1);
    __syncthreads();
  }
  COMPUTE_STEP((kt1 - 1 - kt0) & 1);
#ifdef HANDS_EPI_PRIO        // A/B variant (tools/build_variant.sh -DHANDS_EPI_PRIO=n): a wave in its epilogue asks for issue priority n over
  __builtin_amdgcn_s_setprio(HANDS_EPI_PRIO);   // the k-loop waves of the other workgroups on its SIMD (it holds a slot without feeding the pipe)
#endif
  if constexpr (BLK > 0) {                                // the last (possibly partial) block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += tot[i][j][r];
  }
#undef COMPUTE_STEP

  // ---- epilogue -------------------------------------------------------------------------------------
  // D row = channel = 8*q + 4*(lane>>5) + e, D col = pixel = lane&31.  Each wave transposes its
  // 32-pixel x 64-channel half tile through its own slice of the (now idle) staging LDS so that the
  // residual loads and the output stores run 256 B contiguous per pixel row (16 lanes x 16 B) instead
  // of 32 B: the expand convolutions of layer1/layer2 are HBM-bound and store-transaction bound.
  __syncthreads();                                  // every wave is done reading the k-loop tiles
  if (acc_out != nullptr) {                         // stream-K hand-off: raw accumulators, register layout
    float4* dst = reinterpret_cast<float4*>(acc_out) + (wave * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dst[((i * 2 + j) * 4 + q) * 64] =
              make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
    return;
  }
  float* sE = lds + wave * (32 * EROW);
  const int half = lane >> 5;
  const int c4 = lane & 15;                         // this lane's 4-channel group in the read-back
  const int n_lane = n0 + wn * 64 + c4 * 4;
  const bool n_ok = n_lane < a.N;
  const bool part = a.ksplit > 1;                   // split-K: raw partial sums, reduced by splitk_reduce_kernel
#ifdef HANDS_ABL_NO_EPI      // timing-only ablation (tools/build_variant.sh): no epilogue at all (one impossible store keeps the MFMAs)
  if (!part && m0 + BM <= a.M && n0 + BN <= a.N) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sacc += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
    if (sacc == 123456.789f) a.out[(size_t)m0 * a.out_ps + n0 + lane] = sacc;
    return;
  }
#endif
  if (!F64 && !part && m0 + BM <= a.M && n0 + BN <= a.N) {      // (uniform) full tile: straight-line epilogue, activation compiled in
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int mw = m0 + (wave_u / WAVES_N) * 64, nw = n0 + (wave_u % WAVES_N) * 64;
    char* ou = reinterpret_cast<char*>(a.out) + ((size_t)mw * a.out_ps + nw) * 4;
    const char* ru = reinterpret_cast<const char*>(a.res) + ((size_t)mw * a.res_ps + nw) * 4;
    const float* bp = a.bias + n_lane;
#define HANDS_EPI(ACT)                                                                            \
    if (a.res != nullptr) epilogue_full<ACT, true>(acc, sE, lane, bp, ru, a.res_ps, ou, a.out_ps); \
    else                  epilogue_full<ACT, false>(acc, sE, lane, bp, ru, a.res_ps, ou, a.out_ps)
    if (a.relu == HANDS_ACT_RELU) { HANDS_EPI(HANDS_ACT_RELU); }
    else if (a.relu == HANDS_ACT_NONE) { HANDS_EPI(HANDS_ACT_NONE); }
    else if (a.relu == HANDS_ACT_LEAKY_RELU) { HANDS_EPI(HANDS_ACT_LEAKY_RELU); }
    else { HANDS_EPI(HANDS_ACT_GELU); }
#undef HANDS_EPI
    return;
  }
  // PREC 2: the bias joins the fp64 sum before its one rounding (below), not the rounded value
  const float4 bv = (part || F64) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(a.bias + n_lane);
  double bd[F64 ? 4 : 1][F64 ? 4 : 1];
  if constexpr (F64) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = n0 + wn * 64 + 16 * i + (lane >> 4) + 4 * r;
        bd[i][r] = (!part && ch < a.N) ? (double)a.bias[ch] : 0.0;
      }
    if (part) {       // split-K slice of an fp64 launch: the partial sums stay fp64 ([ksplit][M][part_ps] doubles), reduced in fp64
      double* pd = reinterpret_cast<double*>(a.partial) + (size_t)split * a.M * a.part_ps;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ch = n0 + wn * 64 + 16 * i + (lane >> 4) + 4 * r;
            const int m = m0 + wm * 64 + 16 * jb + (lane & 15);
            if (m < a.M && ch < a.N) pd[(size_t)m * a.part_ps + ch] = accd[i][jb][r];
          }
      return;
    }
  }
  const bool has_res = !part && a.res != nullptr;
  const int act = part ? HANDS_ACT_NONE : a.relu;
  float* const obase = part ? a.partial + (size_t)split * a.M * a.part_ps : a.out;
  const int ops = part ? a.part_ps : a.out_ps;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (F64) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sE[(jj * 16 + (lane & 15)) * EROW + 16 * i + (lane >> 4) + 4 * r] = (float)(accd[i][2 * j + jj][r] + bd[i][r]);
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(sE + (lane & 31) * EROW + i * 32 + q * 8 + half * 4) =
            make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
    }
    const int mbase = m0 + wm * 64 + j * 32 + (lane >> 4);
    float4 rv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mbase + 4 * r;
      const int mc = m < a.M ? m : a.M - 1;
      rv[r] = has_res ? *reinterpret_cast<const float4*>(a.res + (size_t)mc * a.res_ps + (n_ok ? n_lane : 0))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mbase + 4 * r;
      const float4 t = *reinterpret_cast<const float4*>(sE + ((lane >> 4) + 4 * r) * EROW + c4 * 4);
      float4 v;
      v.x = t.x + bv.x + rv[r].x;
      v.y = t.y + bv.y + rv[r].y;
      v.z = t.z + bv.z + rv[r].z;
      v.w = t.w + bv.w + rv[r].w;
      v = apply_act(v, act);
      if (m < a.M && n_ok) *reinterpret_cast<float4*>(obase + (size_t)m * ops + n_lane) = v;
    }
  }
#undef LOAD_TILES
#undef STORE_TILES
}

template <int WAVES_M, int WAVES_N, int MODE, int PREC = 0, bool PRE = false, int BLK = 0>
__global__ void __launch_bounds__(256, 2) conv_igemm_f32_kernel(ConvArgs a) {
  constexpr int RING = 2 * (64 * WAVES_M + 64 * WAVES_N) * (PREC == 1 ? LDS_ROW_B3 : LDS_ROW);
  __shared__ __attribute__((aligned(16))) float lds[RING > EPI_FLOATS ? RING : EPI_FLOATS];
  const int ntiles = a.nblk_m * a.nblk_n;
  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;
  const int tile = xcd_remap(a.ksplit > 1 ? blockIdx.x - split * ntiles : blockIdx.x, ntiles);
  // k-step range of this block: the whole K unless split-K
  const int nk_all = a.Kpad / BK;
  const int kt0 = a.ksplit > 1 ? (int)((long long)split * nk_all / a.ksplit) : 0;
  const int kt1 = a.ksplit > 1 ? (int)((long long)(split + 1) * nk_all / a.ksplit) : nk_all;
  const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  conv_tile<WAVES_M, WAVES_N, MODE, PREC, PRE, BLK>(a, lds, tile, split, kt0, kt1, nullptr, nullptr);
  if (a.ksplit > 1 && a.counters != nullptr) {          // fused split-K reduction: outside conv_tile, nothing of the tile is live here
    int mt, nt;
    tile_coords(a, tile, mt, nt);
    splitk_fused_tail<64 * WAVES_M, 64 * WAVES_N, PREC == 2>(a, lds, tile, mt * 64 * WAVES_M, nt * 64 * WAVES_N, wave_u);
  }
}

// ---- grouped launch: up to GROUP_MAX independent pointwise layers of ONE kernel instantiation in one launch ----------------
// handoccnet_light at 32 samples per GPU runs many small independent launches next to each other in the graph -- the q / k / v
// (/ q2 / k2) projections of a FIT / SET block, the four FPN laterals, the two branches of an hourglass level, conv1 and the
// downsample of a stage's first bottleneck -- each a fraction of the chip wide.  Here they are ONE grid: problem p owns the
// blocks [start[p], start[p + 1]) (counts rounded up to the 8 XCDs so that the XCD-aware tile order of every problem is the
// one its own launch would use); every tile runs conv_tile exactly as in the plain kernel: same bits.
constexpr int GROUP_MAX = 8;
struct GroupArgs {
  ConvArgs a[GROUP_MAX];
  int start[GROUP_MAX + 1];
  int n;
};

template <int WAVES_M, int WAVES_N, int MODE, bool PRE, int BLK>
__global__ void __launch_bounds__(256, 2) conv_igemm_group_f32_kernel(GroupArgs g) {
  constexpr int RING = 2 * (64 * WAVES_M + 64 * WAVES_N) * LDS_ROW;
  __shared__ __attribute__((aligned(16))) float lds[RING > EPI_FLOATS ? RING : EPI_FLOATS];
  int p = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
  const ConvArgs& a = g.a[p];
  const int ntiles = a.nblk_m * a.nblk_n;
  const int local = (int)blockIdx.x - g.start[p];
  // the padding blocks of a problem whose tile count is not a multiple of 8: XCD x owns q + (x < r) tiles
  if ((local >> 3) >= (ntiles >> 3) + ((local & 7) < (ntiles & 7) ? 1 : 0)) return;
  conv_tile<WAVES_M, WAVES_N, MODE, 0, PRE, BLK>(a, lds, xcd_remap(local, ntiles), 0, 0, a.Kpad / BK, nullptr, nullptr);
}

// ---- stream-K: persistent workgroups with equal shares of (tile, k-step) units -------------------------------
// A launch of T tiles on 256 CUs leaves the CUs that got one tile fewer idle for a tile time (400-1600 tile
// launches: 76-88 % efficiency).  Here G = 256 * j workgroups (j per CU, G <= T) each take the unit range
// [g U/G, (g+1) U/G) of the U = T * nk (tile, k-step) units, tiles in the same XCD-aware order.  A range starts in
// the middle of tile ta (its last k-steps) and ends in the middle of tile tb (its first k-steps):
//   1. FIRST the head part of tb: k-steps [0, kb), accumulators dumped to this workgroup's slot, flag published;
//   2. the full tiles in between, each exactly as in the plain kernel;
//   3. LAST the tail part of ta: its accumulators start from the dump of workgroup g-1 (written at the start of
//      that workgroup's life, i.e. long ago) and CONTINUE the k-ordered FMA chain, then the normal epilogue.
// Every output is therefore produced by the same chain of fp32 FMAs as in the plain kernel -- bit-identical,
// independent of G and of the batch size.  If the flag is not up after a bounded wait (the producer was not
// resident yet: dispatch order is not guaranteed), the consumer recomputes the head part itself.
struct SkArgs {
  float* slots;          // [G][SK_SLOT_FLOATS]
  int* flags;            // [G], compared against epoch
  int epoch;             // differs from the previous use of this workspace
};

template <int WAVES_M, int WAVES_N, int MODE>
__global__ void __launch_bounds__(256, 2) conv_igemm_sk_f32_kernel(ConvArgs a, SkArgs sk) {
  constexpr int RING = 2 * (64 * WAVES_M + 64 * WAVES_N) * LDS_ROW;
  __shared__ __attribute__((aligned(16))) float lds[RING > EPI_FLOATS ? RING : EPI_FLOATS];
  __shared__ int s_ready;
  const int G = gridDim.x;
  const int g = xcd_remap(blockIdx.x, G);
  const int ntiles = a.nblk_m * a.nblk_n;
  const int nk = a.Kpad / BK;
  const long long total = (long long)ntiles * nk;
  const long long u0 = total * g / G, u1 = total * (g + 1) / G;
  const int ta = (int)(u0 / nk), ka = (int)(u0 - (long long)ta * nk);
  const int tb = (int)(u1 / nk), kb = (int)(u1 - (long long)tb * nk);

  // G > tiles (1-2 tiles per CU, two workgroups per CU): a range may lie INSIDE one tile -- k-steps [ka, kb) of tile
  // ta, continuing workgroup g-1's chain AND handing its own accumulators on.  The chain of a tile is then up to
  // three workgroups long; every link waits (bounded) for the previous one and otherwise recomputes from k = 0
  const bool inside = ta == tb;                       // kb > ka > = 0 then: u1 > u0
  const int first_full = ka > 0 ? ta + 1 : ta;
  const int nfull = inside ? 0 : tb - first_full;
  const int head = (!inside && kb > 0) ? 1 : 0;
  const int nseg = inside ? 1 : head + nfull + (ka > 0 ? 1 : 0);
  for (int sgi = 0; sgi < nseg; ++sgi) {              // one call site: the tile code is instantiated once
    int tile, k0 = 0, k1 = nk;
    const float* ain = nullptr;
    float* aout = nullptr;
    if (sgi < head) {                                 // 1. head part of the tile the range ends in
      tile = tb; k1 = kb;
      aout = sk.slots + (size_t)g * SK_SLOT_FLOATS;
    } else if (sgi - head < nfull) {                  // 2. full tiles
      tile = first_full + (sgi - head);
    } else if (inside && ka == 0) {                   // 1'. the range is the first part of one tile
      tile = ta; k1 = kb;
      aout = sk.slots + (size_t)g * SK_SLOT_FLOATS;
    } else {                                          // 3. tail part of the tile the range starts in (or 3'. a middle part)
      if (threadIdx.x == 0) {
        int ok = 0;
        for (int spin = 0; spin < 4096 && !ok; ++spin) {
          ok = __hip_atomic_load(sk.flags + (g - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sk.epoch;
          if (!ok) __builtin_amdgcn_s_sleep(8);
        }
        if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ready = ok;
      }
      __syncthreads();
      tile = ta;
      if (s_ready != 0) { k0 = ka; ain = sk.slots + (size_t)(g - 1) * SK_SLOT_FLOATS; }
      if (inside) { k1 = kb; aout = sk.slots + (size_t)g * SK_SLOT_FLOATS; }
    }
    __syncthreads();                                  // the staging LDS of the previous segment is free
    conv_tile<WAVES_M, WAVES_N, MODE>(a, lds, tile, 0, k0, k1, ain, aout);
    if (aout != nullptr) {                            // publish the dump: release at agent scope, then the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (a negative epoch is the test hook for the fallback: nobody publishes, every consumer times out and recomputes)
        if (sk.epoch > 0) __hip_atomic_store(sk.flags + g, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <int WAVES_M, int WAVES_N, int MODE, int PREC = 0, bool PRE = false, int BLK = 0>
int launch(ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N;
  a.nblk_m = (a.M + BM - 1) / BM;
  a.nblk_n = (a.N + BN - 1) / BN;
  const long long nwg = (long long)a.nblk_m * a.nblk_n * (a.ksplit > 1 ? a.ksplit : 1);
  if (nwg <= 0 || nwg > 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL((conv_igemm_f32_kernel<WAVES_M, WAVES_N, MODE, PREC, PRE, BLK>), dim3((unsigned)nwg), dim3(256), 0,
                     stream, a);
  return (int)hipGetLastError();
}

// exact-fp32 launch of MODE (0 any convolution, 2 pointwise) with the summation block the descriptor asks for
template <int MODE, bool PRE = false>
int launch_fp32(const hands_conv_desc* d, ConvArgs& a, hipStream_t s) {
  const bool narrow = d->Cout <= 64;
  if (d->act & HANDS_ACC_F64) return launch<2, 2, MODE, 2, PRE>(a, s);
  if (d->act & HANDS_SUM_BLOCK128) return narrow ? launch<4, 1, MODE, 0, PRE, 8>(a, s) : launch<2, 2, MODE, 0, PRE, 8>(a, s);
  if (d->act & HANDS_SUM_BLOCK64) return narrow ? launch<4, 1, MODE, 0, PRE, 4>(a, s) : launch<2, 2, MODE, 0, PRE, 4>(a, s);
  return narrow ? launch<4, 1, MODE, 0, PRE>(a, s) : launch<2, 2, MODE, 0, PRE>(a, s);
}

constexpr int SK_MAX_G = 1024;

// device facts the stream-K policy needs (immutable properties, looked up once per process)
int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

template <int WAVES_M, int WAVES_N, int MODE>
int sk_blocks_per_cu() {
  static int occ = 0;
  if (occ == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_sk_f32_kernel<WAVES_M, WAVES_N, MODE>, 256, 0) != hipSuccess || n < 1)
      n = 1;
    occ = n > 4 ? 4 : n;
  }
  return occ;
}

// Persistent grid size for a launch of `ntiles` tiles x `nk` k-steps, or 0 when the plain launch is at least as good:
// stream-K pays one 64 KB accumulator hand-off per workgroup, so it is taken only when the plain launch would
// leave > 5 % of the chip idle (tile quantisation: 2-8 tiles per CU) and a tile is long enough (>= 16 k-steps).
int streamk_grid(long long ntiles, int nk, int cus, int max_per_cu) {
  // measured (profiles/README.md): 784- and 1568-tile launches gain 8-22 %; short-K pointwise layers with thousands
  // of tiles lose (the hand-off is a fixed 64 KB write + read per workgroup)
  if (ntiles > cus + cus / 8 && ntiles < 2LL * cus && nk >= 128 && max_per_cu >= 2) {
    // 1.1-2 tiles per CU with a long K (layer4's 3x3 at 256 images, feature_conv: 392-400 tiles, 288-576 k-steps): the
    // plain launch runs two workgroups on some CUs and one on the others; 2 per CU with equal shares is 0.77 of that
    return 2 * cus;
  }
  if (nk < 16 || ntiles < 2LL * cus || ntiles > 8LL * cus) return 0;
  const double per_cu = (double)ntiles / cus;
  const long long rounds = (ntiles + cus - 1) / cus;
  if (per_cu / (double)rounds > 0.95) return 0;
  long long j = ntiles / cus;                       // G <= T: every range holds at least one whole tile of work
  if (j > max_per_cu) j = max_per_cu;
  long long G = j * cus;
  if (G > SK_MAX_G) G = SK_MAX_G / cus * cus;
  return G >= cus ? (int)G : 0;
}

template <int WAVES_M, int WAVES_N, int MODE>
int launch_streamk(ConvArgs& a, hipStream_t stream, void* workspace, long long workspace_bytes, int epoch) {
  constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N;
  a.nblk_m = (a.M + BM - 1) / BM;
  a.nblk_n = (a.N + BN - 1) / BN;
  const long long ntiles = (long long)a.nblk_m * a.nblk_n;
  const int G = (workspace && epoch != 0 && a.ksplit <= 1)
                    ? streamk_grid(ntiles, a.Kpad / BK, device_cus(), sk_blocks_per_cu<WAVES_M, WAVES_N, MODE>()) : 0;
  const long long need = (long long)G * SK_SLOT_FLOATS * 4 + (long long)SK_MAX_G * 4;
  if (G == 0 || workspace_bytes < need) return launch<WAVES_M, WAVES_N, MODE>(a, stream);
  SkArgs sk;
  sk.flags = reinterpret_cast<int*>(workspace);                       // [SK_MAX_G] ints first, then the slots
  sk.slots = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + SK_MAX_G * 4);
  sk.epoch = epoch;
  hipLaunchKernelGGL((conv_igemm_sk_f32_kernel<WAVES_M, WAVES_N, MODE>), dim3((unsigned)G), dim3(256), 0, stream, a, sk);
  return (int)hipGetLastError();
}

// out[m][n] = act(bias[n] + sum_s partial[s][m][n] (+ res[m][n])), s in ascending order (deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int S, int M, int N4, int part_ps,
                                     const float* __restrict__ bias, const float* res, int res_ps, float* out, int out_ps,
                                     int act) {
  const long long total = (long long)M * N4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N4) * 4;
    const long long m = i / N4;
    float4 v = *reinterpret_cast<const float4*>(partial + m * part_ps + n);
    for (int s = 1; s < S; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(partial + ((long long)s * M + m) * part_ps + n);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + m * res_ps + n);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(out + m * out_ps + n) = apply_act(v, act);
  }
}

// the same for an fp64 launch (HANDS_ACC_F64): fp64 partial sums, bias added in fp64, ONE rounding, then residual + activation in fp32
__global__ void splitk_reduce_f64_kernel(const double* __restrict__ partial, int S, int M, int N, int part_ps,
                                         const float* __restrict__ bias, const float* res, int res_ps, float* out, int out_ps,
                                         int act) {
  const long long total = (long long)M * N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const long long m = i / N;
    double v = partial[m * part_ps + n];
    for (int s = 1; s < S; ++s) v += partial[((long long)s * M + m) * part_ps + n];
    float f = (float)(v + (double)bias[n]);
    if (res) f += res[m * res_ps + n];
    if (act == HANDS_ACT_RELU) f = fmaxf(f, 0.f);
    else if (act == HANDS_ACT_GELU) f = gelu_erf(f);
    else if (act == HANDS_ACT_LEAKY_RELU) f = f > 0.f ? f : 0.01f * f;
    out[m * out_ps + n] = f;
  }
}

// split-K policy: depends on the layer (N, K) and on M only through a fixed threshold, so that the
// summation order -- hence every output bit -- is the same for every batch size below the threshold
int splitk_factor(const hands_conv_desc* d) {
  // linear layers only (one row per sample): a convolution's M = B*Ho*Wo would cross the threshold
  // between batch sizes and change the summation order with it
  if (d->KH != 1 || d->KW != 1 || d->H != 1 || d->W != 1 || d->stride != 1 || d->pad != 0 || d->Cin == 4) return 1;
  const long long M = d->B;
  if (M > 2048 || d->Kpad < 512) return 1;
  int s = d->Kpad / 256;
  return s > 8 ? 8 : s;
}

// geometry checks shared by the entry points: offsets are 32-bit inside the kernel and the output map must be
// the one the (H, W, K, stride, pad) geometry produces, or the k-loop would read outside the input
bool conv_geometry_ok(const hands_conv_desc* d) {
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 ||
      d->pad < 0)
    return false;
  if ((d->act & HANDS_ACT_MASK) > HANDS_ACT_LEAKY_RELU || (d->act & ~(HANDS_ACT_MASK | HANDS_MATH_BF16X3 | HANDS_SUM_BLOCK128 | HANDS_SUM_BLOCK64 | HANDS_ACC_F64))) return false;   // unknown code
  if ((d->act & HANDS_ACC_F64) && ((d->act & (HANDS_MATH_BF16X3 | HANDS_SUM_BLOCK128 | HANDS_SUM_BLOCK64)) || d->Cin == 4)) return false;
  if ((d->act & HANDS_SUM_BLOCK128) && (d->act & (HANDS_SUM_BLOCK64 | HANDS_MATH_BF16X3))) return false;          // one summation form
  if ((d->act & HANDS_SUM_BLOCK64) && (d->act & HANDS_MATH_BF16X3)) return false;
  if (d->Cin % 4 || d->Cout % 4 || d->Kpad % BK || d->Kpad < d->KH * d->KW * d->Cin) return false;
  if (d->Cin != 4 && d->Cin % 16) return false;
  if (d->in_pix_stride < d->Cin || d->out_pix_stride < d->Cout) return false;
  // the output map may not be larger than the one the geometry produces (the padded-convolution k-loop
  // checks every tap against the image; a larger map would still index pixels of the next image / past the end)
  if (d->H + 2 * d->pad < d->KH || d->W + 2 * d->pad < d->KW) return false;
  if (d->Ho > (d->H + 2 * d->pad - d->KH) / d->stride + 1 || d->Wo > (d->W + 2 * d->pad - d->KW) / d->stride + 1) return false;
  if ((long long)d->B * d->H * d->W * d->in_pix_stride >= (1LL << 31)) return false;
  if (d->Cin != 4) {
    // 32-bit byte offsets relative to the first image / row a wave stages (its 256 rows reach at most
    // 256 / (Ho * Wo) + 1 images further) and to the first weight row of a tile
    const long long imgs = 256 / ((long long)d->Ho * d->Wo) + 2;
    if (imgs * d->H * d->W * d->in_pix_stride * 4 >= (1LL << 31)) return false;
    if (256LL * d->Kpad * 4 >= (1LL << 31)) return false;
    // padded-convolution k-loop: tap validity is a 15 + 15 bit word per row
    if (!(d->KH == 1 && d->KW == 1 && d->pad == 0) && (d->KH > 15 || d->KW > 15)) return false;
  }
  return true;
}

// the branch-free pointwise k-loop (MODE 2 with its source switch out of reach) has no bounds checks: it
// needs every tap inside the image and the weight row to be exactly the pixel's channels
bool pointwise_route_ok(const hands_conv_desc* d) {
  return d->KH == 1 && d->KW == 1 && d->pad == 0 && d->Cin != 4 && d->Kpad == d->Cin &&
         (long long)(d->Ho - 1) * d->stride < d->H && (long long)(d->Wo - 1) * d->stride < d->W;
}

void fill_plain_args(const hands_conv_desc* d, const float* in, const float* w_packed, const float* bias, const float* residual,
                     float* out, const float* pre_scale, const float* pre_shift, ConvArgs& a) {
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Kpad = d->Kpad;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.res_ps = d->res_pix_stride;
  a.relu = d->act & HANDS_ACT_MASK;
  a.ksplit = 1; a.partial = nullptr; a.part_ps = 0; a.pre_scale = pre_scale; a.pre_shift = pre_shift;
  a.in2 = in; a.K0 = 1 << 30; a.H2 = a.W2 = a.stride2 = a.in2_ps = 0;
  a.nblk_m = a.nblk_n = 0;
}

// The kernel instantiation a POINTWISE layer runs (the unit a grouped launch is homogeneous in), or -1 when the layer cannot
// be grouped: bit 0 narrow tile (256 x 64), bit 1 blocked summation (64 floats), bit 2 operand affine + LeakyReLU.
int group_class(const hands_conv_desc* d, bool pre) {
  if (!d || !conv_geometry_ok(d) || !pointwise_route_ok(d)) return -1;
  if (d->act & (HANDS_MATH_BF16X3 | HANDS_SUM_BLOCK128 | HANDS_ACC_F64)) return -1;
  return (d->Cout <= 64 ? 1 : 0) | ((d->act & HANDS_SUM_BLOCK64) ? 2 : 0) | (pre ? 4 : 0);
}

template <int WAVES_M, int WAVES_N, bool PRE, int BLK>
int launch_group(GroupArgs& g, hipStream_t stream) {
  constexpr int BM = 64 * WAVES_M, BN = 64 * WAVES_N;
  long long total = 0;
  for (int p = 0; p < g.n; ++p) {
    ConvArgs& a = g.a[p];
    a.nblk_m = (a.M + BM - 1) / BM;
    a.nblk_n = (a.N + BN - 1) / BN;
    g.start[p] = (int)total;
    total += ((long long)a.nblk_m * a.nblk_n + 7) / 8 * 8;
    if (total > 0x7fffffffLL) return HANDS_EINVAL;
  }
  for (int p = g.n; p <= GROUP_MAX; ++p) g.start[p] = (int)total;
  hipLaunchKernelGGL((conv_igemm_group_f32_kernel<WAVES_M, WAVES_N, 2, PRE, BLK>), dim3((unsigned)total), dim3(256), 0, stream, g);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int hands_conv2d_group_class(const hands_conv_desc* d, int pre) { return group_class(d, pre != 0); }

extern "C" int hands_conv2d_group_f32(const hands_conv_job* jobs, int n, hands_stream_t stream) {
  if (!jobs || n < 1 || n > GROUP_MAX) return HANDS_EINVAL;
  GroupArgs g;
  g.n = n;
  int cls = -2;
  for (int p = 0; p < n; ++p) {
    const hands_conv_job& j = jobs[p];
    if (!j.desc || !j.in || !j.w_packed || !j.bias || !j.out || ((j.pre_scale != nullptr) != (j.pre_shift != nullptr))) return HANDS_EINVAL;
    const int c = group_class(j.desc, j.pre_scale != nullptr);
    if (c < 0 || (cls != -2 && c != cls)) return HANDS_EINVAL;        // one instantiation per launch
    cls = c;
    fill_plain_args(j.desc, j.in, j.w_packed, j.bias, j.residual, j.out, j.pre_scale, j.pre_shift, g.a[p]);
  }
  hipStream_t s = (hipStream_t)stream;
  switch (cls) {
    case 0: return launch_group<2, 2, false, 0>(g, s);
    case 1: return launch_group<4, 1, false, 0>(g, s);
    case 2: return launch_group<2, 2, false, 4>(g, s);
    case 3: return launch_group<4, 1, false, 4>(g, s);
    case 4: return launch_group<2, 2, true, 0>(g, s);
    case 5: return launch_group<4, 1, true, 0>(g, s);
    case 6: return launch_group<2, 2, true, 4>(g, s);
    case 7: return launch_group<4, 1, true, 4>(g, s);
  }
  return HANDS_EINVAL;
}

extern "C" int hands_conv2d_splitk_factor(const hands_conv_desc* d) { return d ? splitk_factor(d) : 0; }

extern "C" long long hands_conv2d_workspace_floats(const hands_conv_desc* d, int S) {
  if (!d || !conv_geometry_ok(d)) return -1;
  if (S <= 0) S = splitk_factor(d);
  if (S > d->Kpad / BK) S = d->Kpad / BK;
  if (S <= 1) return 0;
  return (long long)S * d->B * d->Ho * d->Wo * d->Cout * ((d->act & HANDS_ACC_F64) ? 2 : 1);   // fp64 partial sums
}

extern "C" int hands_conv2d_nhwc_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                     const float* bias, const float* residual, float* out,
                                     hands_stream_t stream) {
  if (!d || !in || !w_packed || !bias || !out) return HANDS_EINVAL;
  if (!conv_geometry_ok(d)) return HANDS_EINVAL;
  const bool stem = d->Cin == 4;
  ConvArgs a;
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Kpad = d->Kpad;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.res_ps = d->res_pix_stride;
  a.relu = d->act & HANDS_ACT_MASK;
  a.ksplit = 1; a.partial = nullptr; a.part_ps = 0; a.pre_scale = nullptr; a.pre_shift = nullptr;
  a.in2 = in; a.K0 = 1 << 30; a.H2 = a.W2 = a.stride2 = a.in2_ps = 0;
  hipStream_t s = (hipStream_t)stream;
  if (stem) return (d->Cout <= 64) ? launch<4, 1, 1>(a, s) : launch<2, 2, 1>(a, s);
  if (d->act & HANDS_MATH_BF16X3) {
    if (pointwise_route_ok(d)) return (d->Cout <= 64) ? launch<4, 1, 2, 1>(a, s) : launch<2, 2, 2, 1>(a, s);
    return (d->Cout <= 64) ? launch<4, 1, 0, 1>(a, s) : launch<2, 2, 0, 1>(a, s);
  }
  // pointwise layers (1x1, no padding; any stride) take the two-source instantiation with the switch
  // point out of reach: no tap state and no bounds checks in the k-loop
  if (pointwise_route_ok(d)) return launch_fp32<2>(d, a, s);
  return launch_fp32<0>(d, a, s);
}

extern "C" long long hands_conv2d_streamk_workspace_bytes(void) {
  return (long long)SK_MAX_G * SK_SLOT_FLOATS * 4 + (long long)SK_MAX_G * 4;
}

extern "C" int hands_conv2d_streamk_grid(const hands_conv_desc* d) {
  if (!d || !conv_geometry_ok(d) || d->Cin == 4) return 0;
  const bool narrow = d->Cout <= 64;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  const long long ntiles = narrow ? ((M + 255) / 256) * ((d->Cout + 63) / 64) : ((M + 127) / 128) * ((d->Cout + 127) / 128);
  return streamk_grid(ntiles, d->Kpad / BK, device_cus(), 4);
}

extern "C" int hands_conv2d_nhwc_streamk_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                             const float* bias, const float* residual, float* out, void* workspace,
                                             long long workspace_bytes, int epoch, hands_stream_t stream) {
  if (!d || !in || !w_packed || !bias || !out) return HANDS_EINVAL;
  if (!conv_geometry_ok(d)) return HANDS_EINVAL;
  if (d->Cin == 4) return hands_conv2d_nhwc_f32(d, in, w_packed, bias, residual, out, stream);   // the stem has its own kernel
  ConvArgs a;
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Kpad = d->Kpad;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.res_ps = d->res_pix_stride;
  if (d->act & (HANDS_MATH_BF16X3 | HANDS_SUM_BLOCK128 | HANDS_SUM_BLOCK64 | HANDS_ACC_F64))
    return hands_conv2d_nhwc_f32(d, in, w_packed, bias, residual, out, stream);
  a.relu = d->act & HANDS_ACT_MASK;
  a.ksplit = 1; a.partial = nullptr; a.part_ps = 0; a.pre_scale = nullptr; a.pre_shift = nullptr;
  a.in2 = in; a.K0 = 1 << 30; a.H2 = a.W2 = a.stride2 = a.in2_ps = 0;
  hipStream_t s = (hipStream_t)stream;
  if (pointwise_route_ok(d))
    return (d->Cout <= 64) ? launch_streamk<4, 1, 2>(a, s, workspace, workspace_bytes, epoch)
                           : launch_streamk<2, 2, 2>(a, s, workspace, workspace_bytes, epoch);
  return (d->Cout <= 64) ? launch_streamk<4, 1, 0>(a, s, workspace, workspace_bytes, epoch)
                         : launch_streamk<2, 2, 0>(a, s, workspace, workspace_bytes, epoch);
}

extern "C" int hands_conv1x1_dual_nhwc_f32(const hands_conv_desc* d, const float* in, const float* in2, int Cin2, int H2,
                                           int W2, int stride2, int in2_pix_stride, const float* w_packed,
                                           const float* bias, float* out, hands_stream_t stream) {
  if (!d || !in || !in2 || !w_packed || !bias || !out) return HANDS_EINVAL;
  if (d->B <= 0 || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->H != d->Ho || d->W != d->Wo)
    return HANDS_EINVAL;
  if (d->Cin % 16 || Cin2 <= 0 || Cin2 % 16 || d->Cout % 4 || d->Kpad != d->Cin + Cin2) return HANDS_EINVAL;
  if (d->in_pix_stride < d->Cin || in2_pix_stride < Cin2 || d->out_pix_stride < d->Cout || stride2 < 1) return HANDS_EINVAL;
  if ((d->Ho - 1) * stride2 >= H2 || (d->Wo - 1) * stride2 >= W2) return HANDS_EINVAL;
  if ((long long)d->B * d->H * d->W * d->in_pix_stride >= (1LL << 31) ||
      (long long)d->B * H2 * W2 * in2_pix_stride >= (1LL << 31))
    return HANDS_EINVAL;
  {   // 32-bit byte offsets relative to the first row a wave stages (see conv_geometry_ok)
    const long long imgs = 256 / ((long long)d->Ho * d->Wo) + 2;
    if (imgs * d->H * d->W * d->in_pix_stride * 4 >= (1LL << 31) || imgs * H2 * W2 * in2_pix_stride * 4 >= (1LL << 31) ||
        256LL * d->Kpad * 4 >= (1LL << 31))
      return HANDS_EINVAL;
  }
  ConvArgs a;
  a.in = in; a.w = w_packed; a.bias = bias; a.res = nullptr; a.out = out;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Kpad = d->Kpad;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.res_ps = 0;
  a.relu = d->act & HANDS_ACT_MASK;
  a.ksplit = 1; a.partial = nullptr; a.part_ps = 0; a.pre_scale = nullptr; a.pre_shift = nullptr;
  a.in2 = in2; a.K0 = d->Cin; a.H2 = H2; a.W2 = W2; a.stride2 = stride2; a.in2_ps = in2_pix_stride;
  hipStream_t s = (hipStream_t)stream;
  if (d->act & HANDS_MATH_BF16X3) return (d->Cout <= 64) ? launch<4, 1, 2, 1>(a, s) : launch<2, 2, 2, 1>(a, s);
  return (d->Cout <= 64) ? launch<4, 1, 2>(a, s) : launch<2, 2, 2>(a, s);
}

namespace {
int pre_launch(const hands_conv_desc* d, const float* in, const float* pre_scale, const float* pre_shift,
               const float* w_packed, const float* bias, const float* residual, float* out, hands_stream_t stream);

int splitk_launch(const hands_conv_desc* d, const float* in, const float* w_packed, const float* bias, const float* residual,
                  float* out, int S, float* workspace, long long workspace_floats,
                  hands_stream_t stream, const float* pre_scale = nullptr, const float* pre_shift = nullptr,
                  int* counters = nullptr, long long n_counters = 0) {
  if (!d) return HANDS_EINVAL;
  const bool pre = pre_scale != nullptr;
  if (pre && (!pre_shift || !pointwise_route_ok(d) || (d->act & HANDS_MATH_BF16X3))) return HANDS_EINVAL;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  const int part_ps = d->Cout;                          // Cout % 4 == 0
  if (S > d->Kpad / BK) S = d->Kpad / BK;
  const bool f64 = (d->act & HANDS_ACC_F64) != 0;       // fp64 accumulation: the partial sums are doubles (twice the workspace)
  if (S <= 1 || !workspace || workspace_floats < (long long)S * M * part_ps * (f64 ? 2 : 1) || (f64 && (((uintptr_t)workspace) & 7)))
    return pre ? pre_launch(d, in, pre_scale, pre_shift, w_packed, bias, residual, out, stream)
               : hands_conv2d_nhwc_f32(d, in, w_packed, bias, residual, out, stream);
  if (!in || !w_packed || !bias || !out || !conv_geometry_ok(d) || S > 64) return HANDS_EINVAL;
  const bool stem = d->Cin == 4;
  ConvArgs a;
  a.pre_scale = pre_scale; a.pre_shift = pre_shift;
  a.in = in; a.w = w_packed; a.bias = bias; a.res = nullptr; a.out = out;
  a.M = (int)M; a.N = d->Cout; a.Kpad = d->Kpad;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.res_ps = d->res_pix_stride;
  a.relu = HANDS_ACT_NONE;
  a.ksplit = S; a.partial = workspace; a.part_ps = part_ps;
  a.in2 = in; a.K0 = 1 << 30; a.H2 = a.W2 = a.stride2 = a.in2_ps = 0;
  {
    // fused reduction when the caller's counter array covers the tiles of the instantiation this launch takes
    const bool narrow = d->Cout <= 64 && !f64;
    const long long ntiles = narrow ? ((M + 255) / 256) * ((d->Cout + 63) / 64) : ((M + 127) / 128) * ((d->Cout + 127) / 128);
    if (counters && n_counters >= ntiles) { a.counters = counters; a.fin_res = residual; a.fin_act = d->act & HANDS_ACT_MASK; }
  }
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (pre) rc = launch_fp32<2, true>(d, a, s);
  else if (stem) rc = (d->Cout <= 64) ? launch<4, 1, 1>(a, s) : launch<2, 2, 1>(a, s);
  else if (pointwise_route_ok(d)) rc = launch_fp32<2>(d, a, s);
  else rc = launch_fp32<0>(d, a, s);
  if (rc) return rc;
  if (a.counters != nullptr) return 0;                  // the last slice of every tile has reduced it
  if (f64) {
    hipLaunchKernelGGL(splitk_reduce_f64_kernel, dim3(hands_grid_1d(M * d->Cout, 256)), dim3(256), 0, s,
                       reinterpret_cast<const double*>(workspace), S, (int)M, d->Cout, part_ps, bias, residual, d->res_pix_stride,
                       out, d->out_pix_stride, d->act & HANDS_ACT_MASK);
    return (int)hipGetLastError();
  }
  const long long total = M * (d->Cout / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(hands_grid_1d(total, 256)), dim3(256), 0, s, workspace, S, (int)M,
                     d->Cout / 4, part_ps, bias, residual, d->res_pix_stride, out, d->out_pix_stride, d->act & HANDS_ACT_MASK);
  return (int)hipGetLastError();
}
int pre_launch(const hands_conv_desc* d, const float* in, const float* pre_scale, const float* pre_shift,
               const float* w_packed, const float* bias, const float* residual, float* out, hands_stream_t stream) {
  if (!d || !in || !pre_scale || !pre_shift || !w_packed || !bias || !out) return HANDS_EINVAL;
  if (!conv_geometry_ok(d) || !pointwise_route_ok(d) || (d->act & HANDS_MATH_BF16X3)) return HANDS_EINVAL;
  ConvArgs a;
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Kpad = d->Kpad;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.res_ps = d->res_pix_stride;
  a.relu = d->act & HANDS_ACT_MASK;
  a.ksplit = 1; a.partial = nullptr; a.part_ps = 0;
  a.pre_scale = pre_scale; a.pre_shift = pre_shift;
  a.in2 = in; a.K0 = 1 << 30; a.H2 = a.W2 = a.stride2 = a.in2_ps = 0;
  hipStream_t s = (hipStream_t)stream;
  return launch_fp32<2, true>(d, a, s);
}
}  // namespace

extern "C" int hands_conv2d_nhwc_pre_f32(const hands_conv_desc* d, const float* in, const float* pre_scale,
                                         const float* pre_shift, const float* w_packed, const float* bias,
                                         const float* residual, float* out, int S, float* workspace,
                                         long long workspace_floats, hands_stream_t stream) {
  if (S > 1) return splitk_launch(d, in, w_packed, bias, residual, out, S, workspace, workspace_floats, stream,
                                  pre_scale, pre_shift);
  return pre_launch(d, in, pre_scale, pre_shift, w_packed, bias, residual, out, stream);
}

extern "C" int hands_conv2d_nhwc_splitk_n_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                              const float* bias, const float* residual, float* out, int S,
                                              float* workspace, long long workspace_floats, hands_stream_t stream) {
  return splitk_launch(d, in, w_packed, bias, residual, out, S, workspace, workspace_floats, stream);
}

extern "C" int hands_conv2d_nhwc_splitk_fused_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                                  const float* bias, const float* residual, float* out, int S,
                                                  float* workspace, long long workspace_floats, int* counters,
                                                  long long n_counters, hands_stream_t stream) {
  return splitk_launch(d, in, w_packed, bias, residual, out, S, workspace, workspace_floats, stream, nullptr, nullptr, counters,
                       n_counters);
}

extern "C" int hands_conv2d_nhwc_splitk_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                            const float* bias, const float* residual, float* out, float* workspace,
                                            long long workspace_floats, hands_stream_t stream) {
  if (!d) return HANDS_EINVAL;
  return hands_conv2d_nhwc_splitk_n_f32(d, in, w_packed, bias, residual, out, splitk_factor(d), workspace,
                                        workspace_floats, stream);
}
