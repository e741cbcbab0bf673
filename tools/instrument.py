#!/usr/bin/env python3
"""Build timestamp-instrumented copies of the hot kernels into build_ab/ (dev tool; the shipped library is never
instrumented).  The copies are made by textual patches of the CURRENT sources, so a patch that no longer applies fails
loudly here instead of silently measuring something else.

  python tools/instrument.py tile    -> build_ab/prof_tile.so    conv_igemm: per workgroup s_memrealtime at entry, after the
                                         tile setup, after the first barrier, after the k-loop, at exit  (tools/prof_tile.py)
  python tools/instrument.py clock   -> build_ab/prof_clock.so   conv_igemm: s_memrealtime + s_memtime at entry / exit of every
                                         workgroup (tools/prof_clock.py)
  python tools/instrument.py ring    -> build_ab/prof_ring.so    conv_igemm: the middle workgroup of every plain launch stamps
                                         both clocks into a ring (tools/prof_ring.py)
  python tools/instrument.py mano    -> build_ab/prof_mano.so    mano_heads_kernel: pose / blend / skinning / write-back time per wave (tools/prof_mano.py)
  python tools/instrument.py attn    -> build_ab/prof_attn.so    attention_kernel: seven phase stamps per wave (tools/prof_attn.py)
  python tools/instrument.py stem    -> build_ab/prof_stem.so    stem_pool_planar_kernel: entry / after the patch fill / after
                                         the GEMM / exit (tools/stem_prof.py)
Then e.g.:  HANDS_HIP_LIB=build_ab/prof_tile.so python tools/prof_tile.py 256,256,32,256,1,1,0,0
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV = os.path.join(ROOT, "hands_amd", "csrc", "conv_igemm.hip")
STEM = os.path.join(ROOT, "hands_amd", "csrc", "stem_pool.hip")
WINO = os.path.join(ROOT, "hands_amd", "csrc", "conv_wino.hip")
TRANS = os.path.join(ROOT, "hands_amd", "csrc", "transformer.hip")
MANO = os.path.join(ROOT, "hands_amd", "csrc", "mano_lbs.hip")


def sub(s, old, new, what):
    assert s.count(old) == 1, f"patch anchor not found exactly once ({what}): {old[:60]!r}"
    return s.replace(old, new, 1)


def stamp(array, slot, cond="prof"):
    return (f"__builtin_amdgcn_sched_barrier(0); if ({cond}) {array}[blockIdx.x * 8 + {slot}] = "
            f"__builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0);")


def conv_tile_timeline():
    s = open(CONV).read()
    s = sub(s, "namespace {\n\ntypedef float f32x16", """__device__ unsigned long long g_prof[16384 * 8];
__device__ unsigned long long g_steps[64 * 256];
extern "C" int hands_debug_prof(void* dst, void* dst2) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16384 * 8);
  hipMemcpyFromSymbol(dst2, HIP_SYMBOL(g_steps), sizeof(unsigned long long) * 64 * 256);
  return 0;
}
namespace {

typedef float f32x16""", "globals")
    s = sub(s, "  LOAD_TILES(kt0);\n  STORE_TILES(0);\n  __syncthreads();\n",
            "  const bool prof = threadIdx.x == 0 && blockIdx.x < 16384;\n  " + stamp("g_prof", 1) +
            "\n  LOAD_TILES(kt0);\n  STORE_TILES(0);\n  __syncthreads();\n  " + stamp("g_prof", 2) + "\n", "prologue")
    s = sub(s, "  COMPUTE_STEP((kt1 - 1 - kt0) & 1);\n#ifdef HANDS_EPI_PRIO",
            "  COMPUTE_STEP((kt1 - 1 - kt0) & 1);\n  " + stamp("g_prof", 3) + "\n#ifdef HANDS_EPI_PRIO", "loop end")
    s = sub(s, "    return;\n  }\n  // PREC 2: the bias joins the fp64 sum", "    " + stamp("g_prof", 4) +
            "\n    return;\n  }\n  // PREC 2: the bias joins the fp64 sum", "lean epilogue end")
    s = sub(s, "#undef LOAD_TILES\n#undef STORE_TILES\n}", "  " + stamp("g_prof", 4) + "\n#undef LOAD_TILES\n#undef STORE_TILES\n}",
            "general epilogue end")
    s = sub(s, "  const int ntiles = a.nblk_m * a.nblk_n;\n  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;\n",
            "  if (threadIdx.x == 0 && blockIdx.x < 16384) g_prof[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();\n"
            "  const int ntiles = a.nblk_m * a.nblk_n;\n  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;\n", "kernel entry")
    return s, "conv_igemm.hip"


def conv_clock():
    s = open(CONV).read()
    s = sub(s, "namespace {\n\ntypedef float f32x16", """__device__ unsigned long long g_prof[32768 * 4];
extern "C" int hands_debug_prof(void* dst, void* dst2) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 32768 * 4);
  return 0;
}
namespace {

typedef float f32x16""", "globals")
    s = sub(s, "  const int ntiles = a.nblk_m * a.nblk_n;\n  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;\n",
            "  if (threadIdx.x == 0 && blockIdx.x < 32768) { g_prof[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime(); "
            "g_prof[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); }\n"
            "  const int ntiles = a.nblk_m * a.nblk_n;\n  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;\n", "kernel entry")
    s = sub(s, "  conv_tile<WAVES_M, WAVES_N, MODE, PREC>(a, lds, tile, split, kt0, kt1, nullptr, nullptr);\n}",
            "  conv_tile<WAVES_M, WAVES_N, MODE, PREC>(a, lds, tile, split, kt0, kt1, nullptr, nullptr);\n"
            "  if (threadIdx.x == 0 && blockIdx.x < 32768) { g_prof[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime(); "
            "g_prof[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime(); }\n}", "kernel exit")
    return s, "conv_igemm.hip"


def conv_ring():
    s = open(CONV).read()
    s = sub(s, "namespace {\n\ntypedef float f32x16", """__device__ unsigned long long g_ring[8192 * 4];
__device__ unsigned int g_n;
extern "C" int hands_debug_ring(void* dst, unsigned int* n, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ring), sizeof(unsigned long long) * 8192 * 4);
  hipMemcpyFromSymbol(n, HIP_SYMBOL(g_n), sizeof(unsigned int));
  if (reset) { unsigned int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_n), &z, sizeof(z)); }
  return 0;
}
namespace {

typedef float f32x16""", "globals")
    s = sub(s, "  const int ntiles = a.nblk_m * a.nblk_n;\n  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;\n",
            "  unsigned long long rt0 = 0, mt0 = 0;\n  const bool stamp = threadIdx.x == 0 && blockIdx.x == gridDim.x / 2;\n"
            "  if (stamp) { rt0 = __builtin_amdgcn_s_memrealtime(); mt0 = __builtin_amdgcn_s_memtime(); }\n"
            "  const int ntiles = a.nblk_m * a.nblk_n;\n  const int split = a.ksplit > 1 ? blockIdx.x / ntiles : 0;\n", "kernel entry")
    s = sub(s, "  conv_tile<WAVES_M, WAVES_N, MODE, PREC>(a, lds, tile, split, kt0, kt1, nullptr, nullptr);\n}",
            "  conv_tile<WAVES_M, WAVES_N, MODE, PREC>(a, lds, tile, split, kt0, kt1, nullptr, nullptr);\n"
            "  if (stamp) {\n    const unsigned i = atomicAdd(&g_n, 1u) & 8191u;\n    g_ring[i * 4 + 0] = rt0; g_ring[i * 4 + 1] = mt0;\n"
            "    g_ring[i * 4 + 2] = __builtin_amdgcn_s_memrealtime(); g_ring[i * 4 + 3] = __builtin_amdgcn_s_memtime();\n  }\n}",
            "kernel exit")
    return s, "conv_igemm.hip"


def stem_phases():
    s = open(STEM).read()
    s = sub(s, "namespace {\n", "__device__ unsigned long long g_sprof[8192 * 4];\n"
            "extern \"C\" int hands_debug_sprof(void* dst) { hipDeviceSynchronize(); hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sprof), "
            "sizeof(unsigned long long) * 8192 * 4); return 0; }\nnamespace {\n", "globals")
    i = s.index("stem_pool_planar_kernel(")
    head, body = s[:i], s[i:]
    st = ("__builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 8192) g_sprof[blockIdx.x * 4 + %d] = "
          "__builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0);")
    body = sub(body, "  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n",
               "  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n  " + st % 0 + "\n", "entry")
    body = sub(body, "  // ---- 256 x 64 x 160 GEMM, 64 rows per wave", "  " + st % 1 + "\n  // ---- 256 x 64 x 160 GEMM, 64 rows per wave", "gemm")
    body = sub(body, "  // ---- pool through LDS, 16 channels per round (same max order",
               "  " + st % 2 + "\n  // ---- pool through LDS, 16 channels per round (same max order", "pool")
    body = sub(body, "\n}\n\n}  // namespace", "\n  " + st % 3 + "\n}\n\n}  // namespace", "exit")
    return head + body, "stem_pool.hip"


def wino_phases():
    """conv_wino: s_memtime (shader cycles) of wave 0 at entry / before the first barrier / after it / at exit, plus the
    summed duration of the per-channel-block epilogues (tools/prof_wino.py)"""
    s = open(WINO).read()
    s = sub(s, "namespace {\n\ntypedef float f32x16", """__device__ unsigned long long g_wprof[32768 * 8];
extern "C" int hands_debug_wprof(void* dst) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wprof), sizeof(unsigned long long) * 32768 * 8);
  hipMemset((void*)0, 0, 0);
  return 0;
}
extern "C" int hands_debug_wprof_clear() {
  static unsigned long long z[32768 * 8];
  hipDeviceSynchronize();
  hipMemcpyToSymbol(HIP_SYMBOL(g_wprof), z, sizeof(z));
  return 0;
}
namespace {

typedef float f32x16""", "globals")
    st = ("__builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 32768) g_wprof[blockIdx.x * 8 + %d] = "
          "__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);")
    s = sub(s, "  const int tid = threadIdx.x;\n  const int lane = tid & 63;\n  const int xi =", "  " + st % 0 +
            "\n  unsigned long long epi_t = 0, epi_0 = 0;\n  const int tid = threadIdx.x;\n  const int lane = tid & 63;\n  const int xi =", "entry")
    s = sub(s, "  __syncthreads();                       // (the compiler's fence waits for this wave's DMA)\n",
            "  " + st % 1 + "\n  __syncthreads();\n  " + st % 2 + "\n", "prologue")
    s = sub(s, "    float* sZ = lds + (s & 1) * G::BUF_FLOATS;\n",
            "    __builtin_amdgcn_sched_barrier(0); epi_0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);\n"
            "    float* sZ = lds + (s & 1) * G::BUF_FLOATS;\n", "epilogue start")
    s = sub(s, "    ++nbi;\n  }\n", "    ++nbi;\n    __builtin_amdgcn_sched_barrier(0); epi_t += __builtin_amdgcn_s_memtime() - epi_0; "
            "__builtin_amdgcn_sched_barrier(0);\n  }\n  " + st % 5 +
            "\n  if (threadIdx.x == 0 && blockIdx.x < 32768) g_wprof[blockIdx.x * 8 + 4] = epi_t;\n", "exit")
    return s, "conv_wino.hip"


def attn_phases():
    """attention_kernel<12,80> (ViT self-attention, csrc/transformer.hip): s_memrealtime of lane 0 of EVERY wave at entry / after
    the K fill's barrier / after Q.K^T / after the V^T write / after the softmax / after the barrier / after P.V + stores;
    slot 7 = HW_ID | XCC_ID << 32 (tools/prof_attn.py)"""
    s = open(TRANS).read()
    s = sub(s, "namespace {\n", """__device__ unsigned long long g_aprof[32768 * 8];
extern "C" int hands_debug_aprof(void* dst) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_aprof), sizeof(unsigned long long) * 32768 * 8);
  return 0;
}
namespace {
""", "globals")
    i = s.index("attention_kernel(const float* __restrict__ qkv")
    head, body = s[:i], s[i:]
    st = ("__builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0 && pslot < 32768) g_aprof[pslot * 8 + %d] = "
          "__builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0);")
    body = sub(body, "  const float* base = qkv + (long long)b * T * 3 * C + h * D;\n",
               "  const float* base = qkv + (long long)b * T * 3 * C + h * D;\n"
               "  const int pslot = (blockIdx.y * gridDim.x + blockIdx.x) * TW + (threadIdx.x >> 6);\n  " + st % 0 + "\n", "entry")
    body = sub(body, "  __syncthreads();\n  // V: requested now", "  __syncthreads();\n  " + st % 1 + "\n  // V: requested now", "K fill")
    body = sub(body, "  __syncthreads();   // every wave is done with K\n", "  " + st % 2 + "\n  __syncthreads();   // every wave is done with K\n", "QK")
    body = sub(body, "  // softmax over the keys of this lane's query", "  " + st % 3 + "\n  // softmax over the keys of this lane's query", "V write")
    body = sub(body, "  __syncthreads();   // V^T complete\n", "  " + st % 4 + "\n  __syncthreads();   // V^T complete\n  " + st % 5 + "\n", "softmax")
    body = sub(body, "    *reinterpret_cast<float4*>(orow + db * 16) = make_float4(o[db][0], o[db][1], o[db][2], o[db][3]);\n}\n",
               "    *reinterpret_cast<float4*>(orow + db * 16) = make_float4(o[db][0], o[db][1], o[db][2], o[db][3]);\n  " + st % 6 +
               "\n  if ((threadIdx.x & 63) == 0 && pslot < 32768) g_aprof[pslot * 8 + 7] = "
               "(unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);\n}\n", "exit")
    return head + body, "transformer.hip"


def mano_phases():
    """mano_heads_kernel: s_memrealtime of lane 0 of every wave at entry / after the pose + FK phase / accumulated over the
    chunks: blend product (+ its barrier), skinning product, write-back (+ barrier) / exit; slot 7 = HW_ID | XCC_ID << 32
    (tools/prof_mano.py)"""
    s = open(MANO).read()
    s = sub(s, "namespace {\n", """__device__ unsigned long long g_mprof[16384 * 8];
extern "C" int hands_debug_mprof(void* dst) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mprof), sizeof(unsigned long long) * 16384 * 8);
  return 0;
}
namespace {
""", "globals")
    # the stamps cost ~6 registers (124 -> 130 = 3 instead of 4 waves per SIMD): hold the instrumented copy at 128 (a few spills)
    # so that it has the production kernel's residency
    s = sub(s, "__launch_bounds__(256, 2) mano_heads_kernel", "__launch_bounds__(256, 4) mano_heads_kernel", "bounds")
    i = s.index("mano_heads_kernel(ManoHeadsArgs a) {")
    head, body = s[:i], s[i:]
    now = "(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__builtin_amdgcn_s_memrealtime())"   # uniform 32-bit ticks: scalar registers
    fence = "__builtin_amdgcn_sched_barrier(0);"
    body = sub(body, "  const int B = a.B;\n", "  const int B = a.B;\n  unsigned t_bl = 0, t_sk = 0, t_wb = 0, t_x = 0;\n"
               f"  {fence} const unsigned t_in = {now}; {fence}\n", "entry")
    body = sub(body, "  // ---- phase 2 + 3 per chunk", f"  {fence} const unsigned t_pose = {now}; t_x = t_pose; {fence}\n  // ---- phase 2 + 3 per chunk", "pose")
    body = sub(body, "    __syncthreads();\n#endif\n#if !(defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 2)",
               f"    __syncthreads();\n    {fence} {{ const unsigned t = {now}; t_bl += t - t_x; t_x = t; }} {fence}\n#endif\n#if !(defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 2)", "blend")
    body = sub(body, "#if !(defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 4)   // timing-only ablation 4: no vertex write-back",
               f"    {fence} {{ const unsigned t = {now}; t_sk += t - t_x; t_x = t; }} {fence}\n#if !(defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 4)   // timing-only ablation 4: no vertex write-back", "skin")
    body = sub(body, "    __syncthreads();      // the stage is rewritten by the next chunk's GEMM\n",
               f"    __syncthreads();      // the stage is rewritten by the next chunk's GEMM\n    {fence} {{ const unsigned t = {now}; t_wb += t - t_x; t_x = t; }} {fence}\n", "write-back")
    body = sub(body, "      d2[1] = 2.0f * (hy / hz) / a.img_res - 1.0f;\n    }\n  }\n}\n\n}  // namespace",
               "      d2[1] = 2.0f * (hy / hz) / a.img_res - 1.0f;\n    }\n  }\n"
               "  const int pslot = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);\n"
               f"  {fence} if ((threadIdx.x & 63) == 0 && pslot < 16384) {{ unsigned long long* o = g_mprof + pslot * 8; o[0] = t_in; o[1] = t_pose; o[2] = t_bl; o[3] = t_sk; o[4] = t_wb; o[5] = {now}; o[6] = c1 - c0;\n"
               "    o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); }\n}\n\n}  // namespace", "exit")
    return head + body, "mano_lbs.hip"


KINDS = {"mano": (mano_phases, "prof_mano"), "attn": (attn_phases, "prof_attn"), "wino": (wino_phases, "prof_wino"), "tile": (conv_tile_timeline, "prof_tile"), "clock": (conv_clock, "prof_clock"), "ring": (conv_ring, "prof_ring"),
         "stem": (stem_phases, "prof_stem")}

if __name__ == "__main__":
    kinds = sys.argv[1:] or list(KINDS)
    for k in kinds:
        fn, name = KINDS[k]
        src, replaces = fn()
        with tempfile.NamedTemporaryFile("w", suffix=".hip", delete=False) as f:
            f.write(src)
        subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), name, "-", f.name, replaces])
        os.unlink(f.name)
