// gemm_common.h -- shared device code of the fused-denoiser GEMM kernels (engine.hip, gemm_xs.hip): tile constants, the
// argument block, lane helpers and the common epilogue (bias, per-point term, ReLU, GroupNorm, t-/class-embedding add,
// residual, stores).  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/slide_engine.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct GemmArgs {
  const void *X;
  const void *W;
  const SlideEpi *epi;
  const float *in_scale, *in_shift;
  // deferred normalisation of the module-level path (ring kernels with the input affine): the producer stored the RAW
  // convolution output, this GEMM applies x' = relu(x * scale + shift) + add while the fragments go from LDS to MFMA.
  // aff_relu: ReLU after the affine; in_add [sample][add_bs] (first add_n channels) or NULL; aff_tps: 256-row tiles per
  // sample (the sample of tile t is t / aff_tps; 1 in the DDPM plans)
  const float *in_add;
  int aff_relu, add_bs, add_n, aff_tps;
  int rows, x_ld, k_pad, n_cob, in_bs;
  int w_cm;                 // LDS-DMA ring kernels: W is CHUNK-MAJOR [k_pad / 32][n_cob * 32][32] (X is chunk-major
                            // [k / 32][rows][32] when x_ld == 32): a wave's 16 rows x 64 B of one LDS-DMA instruction are then
                            // 1 KB of consecutive memory -- 2.2x the L2 -> LDS rate of 64-byte pieces of strided rows
                            // (tools/lds_fill.hip: 33.6 vs 15.5 TB/s)
  int shm_bytes;            // dynamic LDS of the launch (its last 16 bytes hold the persistent mode's tile index)
  const SlideGnFin *gn_fin; // small-launch affine GEMM: finalise the GroupNorm statistics here (include/slide_engine.h)
  const void *gfeat;        // gather mode (GAT kernels): point-feature table [B*16][g_ldf]; the first g_nsplit K chunks of X row
  const int *gidx;          //   (b, p, k) are read from its row b*16 + gidx[(b*16 + p)*16 + k], the rest from X (x_ld = its own ld)
  int g_ldf, g_nsplit, g_klog2;
  int *sched;               // persistent-mode tile counters (9 ints, zero), or nullptr
  int stagger;              // start delay (10 ns units) of the workgroups in odd wave slots, 0 = none
  unsigned long long *dbg;  // optional per-workgroup timeline (tools/ab/gemm_timeline.py): 16 x 100 MHz stamps per workgroup
  // pair decomposition (gemm_gx.hip; SLIDE_OP_GEMM_GX in include/slide_engine.h): per-point tables the X operand is generated
  // from, the per-slot scalars of group_knn and their coefficient vectors; gx_d2 / gx_w also serve the PAIR_NBR residual
  const void *gx_ta, *gx_tb;
  const float *gx_d2, *gx_w, *gx_vv;
  const int *gx_add_idx;
  int gx_ld, gx_mode, gx_add_idx_stride, gx_vbs;
  // split generated-X GEMM with a CHAINED second layer (gemm_gxs.hip, round 5: second_mlp -> rest_mlp of an SA block in one launch,
  // the middle activation in registers): float row-major weights [ch_n_cob*32][ch_k_pad] and epilogue descriptors of that layer
  const void *ch_W;
  const SlideEpi *ch_epi;
  int ch_n_cob, ch_k_pad;
  // PACKED VECTORS (round 6, SLIDE_EPI_PACKED_VECS in include/slide_engine.h): [n_cob][bias 32 | gamma 32 | beta 32] fp32 behind the
  // descriptor array, or NULL (the kernels then read the vectors through the descriptors' pointers); resolve_epi() sets them
  const float *vecs, *ch_vecs;
  // ATTEND epilogue (round 6, module-level path; SLIDE_OP_GEMM_ATTEND): this GEMM's output is the score map of an AttentionModule --
  // instead of being stored it is soft-maxed over the K neighbour rows of each point and contracted with the value rows
  const void *at_V;          // values [rows][at_ldv] fp16
  void *at_out;              // out [rows / K][at_ldo] fp16
  const float *at_vss;       // deferred normalisation of the values: [sample][scale | shift][at_ldv] fp32, or NULL
  const int *at_counts;      // [rows / K] or NULL: only the first max(1, count) neighbour slots of a point take part
  int at_ldv, at_ldo, at_klog2, at_pps, at_vrelu, at_C;
};

// the descriptor pointers as the ops carry them -> array + packed vectors (call once n_cob / ch_n_cob are set)
static inline void resolve_epi(GemmArgs &a) {
  a.vecs = a.ch_vecs = nullptr;
  if ((uintptr_t)a.epi & SLIDE_EPI_PACKED_VECS) {
    a.epi = reinterpret_cast<const SlideEpi *>((uintptr_t)a.epi & ~SLIDE_EPI_PACKED_VECS);
    a.vecs = reinterpret_cast<const float *>(a.epi + a.n_cob);
  }
  if ((uintptr_t)a.ch_epi & SLIDE_EPI_PACKED_VECS) {
    a.ch_epi = reinterpret_cast<const SlideEpi *>((uintptr_t)a.ch_epi & ~SLIDE_EPI_PACKED_VECS);
    a.ch_vecs = reinterpret_cast<const float *>(a.ch_epi + a.ch_n_cob);
  }
}

// SLIDE_OP_PAIR_FIRST (pair_first_kernel, engine.hip): what the pair-table epilogue reads beside the GEMM arguments
struct PairArgs {
  const float *xyz;          // coordinates [B*16][3]
  const float *wa, *wb;      // coordinate coefficients of the a / b tables, [ld][4] per pair channel
  _Float16 *ta, *tb;         // the tables [B*16][ld]
  const int *nbr;            // FP blocks: neighbour / squared-distance / weight slots [B*16][16] (first 8 of a row)
  const float *d2t, *wt;
  const float *vv_in;        // FP blocks: coefficient vectors of the two per-slot scalars [2][ld]
  float *vv_out;             //            and their per-sample scaled copies [B][2][ld]
  int pair_cob0, ld;         // first 32-channel block of the pair segments; table row length
};

namespace {

constexpr int TM = 256;  // rows per workgroup (whole samples: 1 x 256, 2 x 128 or 16 x 16 rows)
constexpr int BK = 32;   // K chunk staged through LDS
constexpr float GN_EPS = 1e-5f;

// PREC selects BOTH the MFMA operand type and the storage type of every activation matrix in HBM:
//   fp32: float activations,   v_mfma_f32_32x32x2_f32   (exact; parity mode)
//   fp16: _Float16 activations, v_mfma_f32_32x32x16_f16 (fp32 accumulate and fp32 epilogue math; throughput mode)
template <int PREC> struct TileT;
template <> struct TileT<SLIDE_PREC_F32> { using T = float; static constexpr int LDK = 36; static constexpr int EPL = 4; };
template <> struct TileT<SLIDE_PREC_F16> { using T = _Float16; static constexpr int LDK = 40; static constexpr int EPL = 8; };
// split mode: float activations in HBM (T), two fp16 planes (hi, scaled lo) per operand tile in LDS (rows of LDK = 40 halves)
template <> struct TileT<SLIDE_PREC_SPLIT> { using T = float; static constexpr int LDK = 40; static constexpr int EPL = 4; };

#ifdef SLIDE_TIMELINE  // instrumented build only (tools/ab/gemm_timeline.py); the product library carries no stamps
#define SLIDE_STAMP(a, k)                                                                     \
  do {                                                                                        \
    if ((a).dbg && threadIdx.x == 0) (a).dbg[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); \
  } while (0)
#else
#define SLIDE_STAMP(a, k) do { } while (0)
#endif

// sum over aligned groups of W lanes (16 or 32); every lane of the group gets the total
template <int W>
__device__ __forceinline__ float lane_group_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // xor 1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // xor 2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  if (W == 32) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));      // xor 16
  return v;
}

// Epilogue descriptors are staged in LDS and read back through readfirstlane: every field lands in an SGPR, every
// branch on it is a scalar branch, and the pointers are known GLOBAL (address space 1) so the compiler neither
// re-loads descriptor fields after each store (aliasing) nor falls back to flat accesses.
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ GLOBAL_AS T *gptr(uint64_t v) { return (GLOBAL_AS T *)v; }
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const GLOBAL_AS float *p) {
  const f32x4v v = *reinterpret_cast<const GLOBAL_AS f32x4v *>(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 gload4(const GLOBAL_AS _Float16 *p) {
  const f16x4 h = *reinterpret_cast<const GLOBAL_AS f16x4 *>(p);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ void gstore4(GLOBAL_AS float *p, float4 v) {
  f32x4v o; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  *reinterpret_cast<GLOBAL_AS f32x4v *>(p) = o;
}
__device__ __forceinline__ void gstore4(GLOBAL_AS _Float16 *p, float4 v) {
  f16x4 h;
  h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
  *reinterpret_cast<GLOBAL_AS f16x4 *>(p) = h;
}

// v_permlane32_swap: a[lanes 32..63] <-> b[lanes 0..31].  Written as asm: the builtin's second result was folded into
// the first by this toolchain when both fed conversions (residual path), silently corrupting quads 2p+1.
__device__ __forceinline__ void lane32_swap(uint32_t &a, uint32_t &b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

template <typename T> __device__ __forceinline__ float4 load4(const T *p);
template <> __device__ __forceinline__ float4 load4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 load4<_Float16>(const _Float16 *p) {
  const f16x4 h = *reinterpret_cast<const f16x4 *>(p);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename T> __device__ __forceinline__ void store4(T *p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void store4<_Float16>(_Float16 *p, float4 v) {
  f16x4 h;
  h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
  *reinterpret_cast<f16x4 *>(p) = h;
}

// CBW = 32-channel output blocks per wave; the workgroup tile is 256 rows x (32*CBW) channels: the four waves
// split the rows (64 each) and share the W panel, so every X element fetched from L2/HBM feeds 32*CBW MACs.
constexpr int EPI_DW = (int)(sizeof(SlideEpi) / 4);
static_assert(sizeof(SlideEpi) == 160, "descriptor layout is read by dword index in gemm_epilogue");

// copies the CBW epilogue descriptors of this workgroup and their per-channel vectors [cb][bias | gamma | beta][32]
// into LDS (visible after the caller's next barrier)
// With packed vectors (a.vecs) both go by LDS-DMA, four bytes per lane, straight from the descriptor buffer: no register staging,
// no pointer chase (descriptor -> bias / gamma / beta pointers -> values: two dependent round trips in every workgroup's prologue).
// The DMAs are complete after the issuing wave's next vmcnt wait that covers them -- every caller drains its VM counter and passes
// a workgroup barrier before its epilogue reads the tables.  Blocks past n_cob copy the last block (process() never reads them).
template <int CBW, int NT = 256>
__device__ __forceinline__ void stage_epilogue_tables(const GemmArgs &a, int cob0, int tid, uint32_t *epi_lds,
                                                      float *vec_lds) {
  if (a.vecs) {  // (uniform)
    const int lane = tid & 63, w0 = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    for (int i0 = w0; i0 < CBW * EPI_DW; i0 += NT) {
      const int i = i0 + lane;
      if (i < CBW * EPI_DW) {
        int cobi = cob0 + i / EPI_DW;
        cobi = cobi < a.n_cob ? cobi : a.n_cob - 1;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(reinterpret_cast<const uint32_t *>(a.epi + cobi) + i % EPI_DW),
                                         (__attribute__((address_space(3))) void *)(epi_lds + i0), 4, 0, 0);
      }
    }
    for (int i0 = w0; i0 < CBW * 96; i0 += NT) {
      const int i = i0 + lane;
      if (i < CBW * 96) {
        int cobi = cob0 + i / 96;
        cobi = cobi < a.n_cob ? cobi : a.n_cob - 1;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(a.vecs + (size_t)cobi * 96 + i % 96),
                                         (__attribute__((address_space(3))) void *)(vec_lds + i0), 4, 0, 0);
      }
    }
    return;
  }
  for (int i = tid; i < CBW * EPI_DW; i += NT) {
    const int cobi = cob0 + i / EPI_DW;
    epi_lds[i] = cobi < a.n_cob ? reinterpret_cast<const uint32_t *>(a.epi + cobi)[i % EPI_DW] : 0u;
  }
  for (int i = tid; i < CBW * 96; i += NT) {
    const int cobi = cob0 + i / 96, which = (i % 96) >> 5, c = i & 31;
    float val = 0.f;
    if (cobi < a.n_cob) {
      const SlideEpi *ed = a.epi + cobi;
      const float *src = which == 0 ? ed->bias : (which == 1 ? ed->gamma : ed->beta);
      if (src) val = src[c];
    }
    vec_lds[i] = val;
  }
}

// ATTEND epilogue (round 6; SLIDE_OP_GEMM_ATTEND, module-level path): the accumulators are the SCORE map of an AttentionModule
// (weight_conv's last convolution; reference pointnet2_ops attention.py:86-95: softmax(dim=-1) over the K neighbours, weighted sum of
// the values).  Per 32-channel block the workgroup's 256 x 32 scores (+ bias) go through LDS (the dead ring: [row][33] floats) so that
// a thread owns (point, channel) and walks the point's K = 4 .. 32 neighbour rows -- the arithmetic of rows_attn_kernel (rows_ops.hip:
// maximum, exponentials, normalised weighted sum over the first max(1, count) slots; the values' deferred GroupNorm + ReLU applied on
// load) on fp32 scores that never reach memory: the unfused path stored the score map in fp16 and read it back beside the values,
// 2 x rows x C x 2 B per attention block.  (A first version reduced over the lanes of the C layout with DPP: 15 cross-lane steps per
// channel register made the launch 2.4x the plain GEMM's time.)
template <int CBW>
__device__ __forceinline__ void attend_epilogue(const GemmArgs &a, f32x16 (&acc)[CBW][2], int row0, int cob0, int wave, int half,
                                                int col, const float *vec_lds, float *stile) {
  using T = _Float16;
  const int lg = a.at_klog2, K = 1 << lg, tid = threadIdx.x;
  const GLOBAL_AS T *V = gptr<const T>((uint64_t)a.at_V);
  GLOBAL_AS T *out = gptr<T>((uint64_t)a.at_out);
  const int c = tid & 31;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int cobi = cob0 + cb;
    if (cobi >= a.n_cob) break;  // uniform per workgroup
    if (cb) __syncthreads();     // the previous block's reads are done
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
        stile[(wave * 64 + rb * 32 + col) * 33 + ch] = acc[cb][rb][r] + vec_lds[cb * 96 + ch];
      }
    __syncthreads();
    const int cg = cobi * 32 + c;  // this thread's channel
    for (int p = tid >> 5; p < (TM >> lg); p += 8) {
      const int r0 = row0 + (p << lg);
      if (r0 >= a.rows) break;
      const int pt = r0 >> lg;
      int kk = K;
      if (a.at_counts) { kk = a.at_counts[pt]; kk = kk < 1 ? 1 : (kk > K ? K : kk); }
      float vs = 1.f, vh = 0.f;
      if (a.at_vss) {
        const int smp = pt / a.at_pps;
        vs = a.at_vss[((size_t)smp * 2 + 0) * a.at_ldv + cg];
        vh = a.at_vss[((size_t)smp * 2 + 1) * a.at_ldv + cg];
      }
      const float *sp = stile + (p << lg) * 33 + c;
      const GLOBAL_AS T *vp = V + (size_t)r0 * a.at_ldv + cg;
      float m = -INFINITY;
      for (int k = 0; k < kk; ++k) m = fmaxf(m, sp[k * 33]);
      float l = 0.f, n = 0.f;
#pragma unroll 4
      for (int k = 0; k < kk; ++k) {
        float v = (float)vp[(size_t)k * a.at_ldv];
        if (a.at_vss) {
          v = v * vs + vh;
          if (a.at_vrelu) v = fmaxf(v, 0.f);
        }
        const float e = __expf(sp[k * 33] - m);
        l += e;
        n = fmaf(e, v, n);
      }
      out[(size_t)pt * a.at_ldo + cg] = (T)(cg < a.at_C ? n / l : 0.f);
    }
  }
}

// Shared epilogue of the GEMM kernels: acc[cb][rb] holds D[co][row] in the 32x32 MFMA C layout (lane: row = lane & 31,
// reg r: co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  `red` = 4*CBW*2*16*2 floats of LDS scratch (the dead tiles).
// RB = 32-row blocks per wave (2; 1 for the split-K small-launch kernel, whose waves own one block each).
// PAIRRES: the instantiation also serves the PAIR residual (SLIDE_F_RES_PAIR / _NBR: two table rows instead of one stored row) --
// a template parameter because its address arithmetic costs every instantiation registers, also where no plan uses it
// KEEP: the finished values (normalised, ReLU, + add vector) are written back to `acc` instead of memory -- the caller chains the
// next layer on them (gemm_gxs.hip)
// NOADDV: LEAN instantiation for a layer that is known to be "GroupNorm epilogue, no add vector, no per-point pre-activation term"
// (the chained rest_mlp of gemm_gxs.hip): the STATS / RAW paths and the add-vector row registers are compiled out
// FMOK: the instantiation can store FRAGMENT-major blocks (SLIDE_F_OUT_FM; the generated-X kernels of gemm_gx.hip, whose outputs u / mo
// the register-X attention tail reads) -- a template parameter for the same reason as PAIRRES
template <int PREC, int NPXL, int CBW, int RB = 2, bool PAIRRES = false, bool KEEP = false, bool NOADDV = false, bool FMOK = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x16 (&acc)[CBW][RB], int row0, int cob0, int wave,
                                              int half, int col, const uint32_t *epi_lds, const float *vec_lds,
                                              float *red) {
  using T = typename TileT<PREC>::T;
  constexpr int NPX = 1 << NPXL;
  float *const gsh = red + 256 * CBW;  // [cb][sample][scale | shift][32], behind the partial sums
  // PH = 0: everything for channel block cb.  Samples spanning several waves (NPX >= 128) exchange their statistics
  // through LDS: PH = 1 (bias, partial sums -> LDS) for all blocks, ONE workgroup barrier, then PH = 2 (totals,
  // normalisation, stores) -- instead of a barrier per channel block.
  // PAIR_NBR residual (8-neighbour samples): the row's neighbour slot -- table row of the neighbour, the two per-slot scalars --
  // is the same for every channel block: looked up ONCE here, under phase 1, instead of at the top of every block's phase 2
  // (one dependent L2 round trip per block less: 11.3 -> 6 us of store phase on the FP1 layer of the feature net)
  int nbr_row[RB];
  f16x2 nbr_sc[RB];  // (d2, w)
  float nbr_d2f[RB], nbr_wf[RB];  // float activations (split mode, round 5): the two per-slot scalars in fp32
  if constexpr (PAIRRES && NPXL == 7) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      nbr_row[rb] = 0;
      nbr_sc[rb] = f16x2{(_Float16)0.f, (_Float16)0.f};
      nbr_d2f[rb] = nbr_wf[rb] = 0.f;
      const int row = row0 + wave * 64 + rb * 32 + col;
      if (a.gidx && a.gx_d2 && a.gx_w && row < a.rows) {
        const int smp = row >> NPXL, pxl = row & (NPX - 1);
        const int slot = (smp * 16 + (pxl >> 3)) * 16 + (pxl & 7);
        nbr_row[rb] = smp * 16 + a.gidx[slot];
        // (the squared distance is clamped to the fp16 range before the conversion -- ADVICE r3: beyond 65504 it became inf.  The
        //  residual itself stays in packed fp16, like the generated-X fragments of the same block: evaluating it in fp32 -- tried in
        //  round 4 -- left single forwards unchanged and made 1000-step position chains deviate 5x MORE from the fp32 mode
        //  (median per-shape distance 6.8e-4 vs 1.4e-4, tools/ab/chain_dev.py): the block's two uses of (d2, w) then round differently)
        nbr_d2f[rb] = a.gx_d2[slot]; nbr_wf[rb] = a.gx_w[slot];
        nbr_sc[rb] = f16x2{(_Float16)fminf(nbr_d2f[rb], 65504.f), (_Float16)nbr_wf[rb]};
      }
    }
  }
  auto process = [&](const int cb, auto ph_tag) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    const int cobi = cob0 + cb;
    if (cobi >= a.n_cob) return;  // uniform per workgroup
    auto rd = [&](int k) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)epi_lds[cb * EPI_DW + k]); };
    auto rdp = [&](int k) { return (uint64_t)rd(k) | ((uint64_t)rd(k + 1) << 32); };
    const int mode = NOADDV ? (int)SLIDE_EPI_NORM : (int)rd(0), flags = (int)rd(1), e_gs = (int)rd(2), e_n_norm = (int)rd(3);
    const float e_inv_count = __uint_as_float(rd(4)), e_stats_scale = __uint_as_float(rd(5));
    const int e_out_ld = (int)rd(6), e_res_ld = (int)rd(7), e_addvec_bs = (int)rd(8), e_stats_bs = (int)rd(9),
              e_pre_ld = (int)rd(10), e_pre_shift = (int)rd(11), e_idx_stride = (int)rd(12);
    const GLOBAL_AS float *e_addvec = gptr<const float>(rdp(20));
    const float *v_bias = vec_lds + cb * 96, *v_gamma = v_bias + 32, *v_beta = v_bias + 64;
    const GLOBAL_AS int *e_addvec_idx = gptr<const int>(rdp(22));
    const GLOBAL_AS T *resid = gptr<const T>(rdp(24)), *pre = NOADDV ? nullptr : gptr<const T>(rdp(26));
    const uint64_t e_out = rdp(28);
    GLOBAL_AS float *e_stats_sum = gptr<float>(rdp(30)), *e_stats_sq = gptr<float>(rdp(32));
    if (PH == 2 && cb == 0) SLIDE_STAMP(a, 8);
    // global reads of the store phase, issued first so that their latency overlaps the statistics / normalisation work
    constexpr bool kHalf = std::is_same<T, _Float16>::value;
    const bool wide16 = kHalf && !(flags & SLIDE_F_OUT_F32);
    const bool pair = PAIRRES && (flags & (SLIDE_F_RES_PAIR | SLIDE_F_RES_PAIR_NBR)) != 0;  // (fp16 rows, 128- / 256-row samples only)
    const GLOBAL_AS float *addv = NOADDV ? nullptr : e_addvec;
    static_assert(RB == 2 || NPXL < 6, "one row block per wave only for samples of at most 32 rows");
    constexpr int NA = NPXL >= 6 ? 1 : RB;  // a wave's 64 rows belong to one sample when NPX >= 64
    float4 apre[NA][4];
    u32x4 rpre[RB][2];
    int pr_a[RB], pr_b[RB];  // float rows with a PAIR residual: the two table rows of the lane's row (read in the store phase)
    if (PH != 1) {
      if (addv && e_addvec_idx) addv += (size_t)e_addvec_idx[0] * e_idx_stride;  // row t of a per-timestep table
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int row = row0 + wave * 64 + rb * 32 + col;
        const bool ok = row < a.rows;
        if (rb < NA) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            apre[rb][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (addv && ok) apre[rb][q] = gload4(addv + (size_t)(row >> NPXL) * e_addvec_bs + 8 * q + 4 * half);
          }
        }
        // pair residual: the row's value is the sum of two per-point table rows (+ the two per-slot terms of group_knn)
        size_t ra_row = (size_t)row, rb_row = 0;
        _Float16 sd2 = (_Float16)0.f, sw = (_Float16)0.f;
        pr_a[rb] = row; pr_b[rb] = 0;
        if constexpr (!kHalf && NPXL >= 7 && PAIRRES) {
          if (pair && ok) {
            const int smp = row >> NPXL, pxl = row & (NPX - 1);
            if (flags & SLIDE_F_RES_PAIR) {
              pr_a[rb] = smp * 16 + (pxl & 15); pr_b[rb] = row >> 4;
            } else if constexpr (NPXL == 7) {
              pr_a[rb] = nbr_row[rb]; pr_b[rb] = row >> 3;
            }
          }
        }
        if constexpr (kHalf && NPXL >= 7 && PAIRRES) {
          if (pair && ok) {
            const int smp = row >> NPXL, pxl = row & (NPX - 1);
            if (flags & SLIDE_F_RES_PAIR) {
              ra_row = (size_t)(smp * 16 + (pxl & 15)); rb_row = (size_t)(row >> 4);
            } else if constexpr (NPXL == 7) {
              ra_row = (size_t)nbr_row[rb]; rb_row = (size_t)(row >> 3);
              sd2 = nbr_sc[rb][0]; sw = nbr_sc[rb][1];
            }
          }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          rpre[rb][p] = u32x4{0u, 0u, 0u, 0u};
          if constexpr (kHalf)
            if (wide16 && resid && ok) {
              rpre[rb][p] = *(const GLOBAL_AS u32x4 *)(resid + ra_row * e_res_ld + 16 * p + 8 * half);
              if constexpr (NPXL >= 7 && PAIRRES) {
                if (pair) {
                  const u32x4 tb = *(const GLOBAL_AS u32x4 *)(gptr<const T>(rdp(34)) + rb_row * e_res_ld + 16 * p + 8 * half);
                  f16x8 r8 = __builtin_bit_cast(f16x8, rpre[rb][p]) + __builtin_bit_cast(f16x8, tb);
                  if (flags & SLIDE_F_RES_PAIR_NBR) {
                    const GLOBAL_AS float *vd = gptr<const float>(rdp(36)) + 16 * p + 8 * half;
                    const GLOBAL_AS float *vw = gptr<const float>(rdp(38)) + 16 * p + 8 * half;
                    const float4 d0 = gload4(vd), d1 = gload4(vd + 4), w0 = gload4(vw), w1 = gload4(vw + 4);
                    const f16x8 vd8 = {(_Float16)d0.x, (_Float16)d0.y, (_Float16)d0.z, (_Float16)d0.w,
                                       (_Float16)d1.x, (_Float16)d1.y, (_Float16)d1.z, (_Float16)d1.w};
                    const f16x8 vw8 = {(_Float16)w0.x, (_Float16)w0.y, (_Float16)w0.z, (_Float16)w0.w,
                                       (_Float16)w1.x, (_Float16)w1.y, (_Float16)w1.z, (_Float16)w1.w};
                    const f16x8 s8 = {sd2, sd2, sd2, sd2, sd2, sd2, sd2, sd2}, t8 = {sw, sw, sw, sw, sw, sw, sw, sw};
                    r8 = __builtin_elementwise_fma(s8, vd8, r8);
                    r8 = __builtin_elementwise_fma(t8, vw8, r8);
                  }
                  rpre[rb][p] = __builtin_bit_cast(u32x4, r8);
                }
              }
            }
        }
      }
    }
    // v[rb][i] = channel pair i of the lane: channels cpair(i) = 8 (i >> 1) + 4 half + 2 (i & 1) and +1.  Everything
    // below is written on pairs so that it compiles to packed fp32 VALU ops (v_pk_add/mul/fma_f32): the epilogue's VALU
    // instruction count, not MFMA, bounds the small-K launches (rocprofv3 SQ_INSTS_VALU vs SQ_INSTS_MFMA).
    f32x2 v[RB][8];
    if (PH == 2) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[rb][i] = f32x2{acc[cb][rb][2 * i], acc[cb][rb][2 * i + 1]};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bia = *reinterpret_cast<const float4 *>(v_bias + 8 * q + 4 * half);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          v[rb][2 * q] = f32x2{acc[cb][rb][4 * q], acc[cb][rb][4 * q + 1]} + f32x2{bia.x, bia.y};
          v[rb][2 * q + 1] = f32x2{acc[cb][rb][4 * q + 2], acc[cb][rb][4 * q + 3]} + f32x2{bia.z, bia.w};
        }
      }
      if (pre) {  // per-point term shared by the K neighbours of a point (query half of attention weight_conv.2)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int row = row0 + wave * 64 + rb * 32 + col;
          if (row < a.rows) {
            // shift >= 0: the row's point (row >> shift).  shift < 0: the row's NEIGHBOUR point, looked up in the sorted
            // neighbour table (K = 2^-shift): per-point partial products of a block's first layer, gathered
            size_t prow = (size_t)(row >> e_pre_shift);
            if (e_pre_shift < 0) {
              const int kl = -e_pre_shift, smp = row >> NPXL, pxl = row & (NPX - 1);
              prow = (size_t)(smp * 16 + a.gidx[(smp * 16 + (pxl >> kl)) * 16 + (pxl & ((1 << kl) - 1))]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 u = gload4(pre + prow * e_pre_ld + 8 * q + 4 * half);
              v[rb][2 * q] += f32x2{u.x, u.y};
              v[rb][2 * q + 1] += f32x2{u.z, u.w};
            }
          }
        }
      }
      if (flags & SLIDE_F_PRE_RELU) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[rb][i] = __builtin_elementwise_max(v[rb][i], f32x2{0.f, 0.f});
      }
      if (PH == 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int i = 0; i < 8; ++i) { acc[cb][rb][2 * i] = v[rb][i][0]; acc[cb][rb][2 * i + 1] = v[rb][i][1]; }
      }
    }
    // NSCOPE = number of independent sample scopes per wave (NPX=16: one per 32-row block, two samples each)
    constexpr int NSCOPE = (NPXL >= 6) ? 1 : RB;
    constexpr int LG = NPX < 32 ? NPX : 32;  // lanes (rows) of one sample inside a row block
    constexpr int WPS = NPXL >= 7 ? NPX / 64 : 1;  // waves per sample when a sample spans waves (2 or 4)
    // sums `NV` per-lane partials over the rows of the lane's sample: lanes -> (row blocks) -> waves via LDS.
    // XH: also fold the other lane half in (groups wider than one half's quad).
    auto reduce_rows = [&](auto nv_tag, float *s, float *ss, bool xh) __attribute__((always_inline)) {
      constexpr int NV = decltype(nv_tag)::value;
      if (PH != 2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          s[i] = lane_group_sum<LG>(s[i]);
          ss[i] = lane_group_sum<LG>(ss[i]);
        }
      }
      if (NPXL >= 7) {  // fixed summation order -> deterministic
        if (PH != 2 && col == 0) {
#pragma unroll
          for (int i = 0; i < NV; ++i)
            *reinterpret_cast<f32x2 *>(red + (((wave * CBW + cb) * 2 + half) * 16 + i) * 2) = f32x2{s[i], ss[i]};
        }
        if (PH == 0) __syncthreads();
        if (PH == 1) return;
        const int w0 = (wave / WPS) * WPS;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          f32x2 t = {0.f, 0.f};
#pragma unroll
          for (int w = 0; w < WPS; ++w)
            t += *reinterpret_cast<const f32x2 *>(red + ((((w0 + w) * CBW + cb) * 2 + half) * 16 + i) * 2);
          if (xh) {
#pragma unroll
            for (int w = 0; w < WPS; ++w)
              t += *reinterpret_cast<const f32x2 *>(red + ((((w0 + w) * CBW + cb) * 2 + (half ^ 1)) * 16 + i) * 2);
          }
          s[i] = t[0]; ss[i] = t[1];
        }
      } else if (xh) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {  // other lane half: v_permlane32_swap of (s, ss) -- pure VALU, no LDS round trip
          uint32_t ta = __float_as_uint(s[i]), tb = __float_as_uint(ss[i]);
          // after the swap: ta = [s.lo | ss.lo -> hi lanes], tb = [s.hi -> lo lanes | ss.hi]
          lane32_swap(ta, tb);
          const float mine_s = s[i], mine_ss = ss[i];
          // lo lanes: other half's s is in tb;   hi lanes: other half's ss is in ta
          // second swap of the pair (ss, s) gives the remaining two
          uint32_t tc = __float_as_uint(mine_ss), td = __float_as_uint(mine_s);
          lane32_swap(tc, td);
          // tc = [ss.lo | s.lo -> hi lanes], td = [ss.hi -> lo lanes | s.hi]
          s[i] = mine_s + (half ? __uint_as_float(tc) : __uint_as_float(tb));
          ss[i] = mine_ss + (half ? __uint_as_float(ta) : __uint_as_float(td));
        }
      }
    };
    if (mode == SLIDE_EPI_STATS && PH == 2) {
      // sums already written by finalize_stats; the raw values are stored below
    } else if (mode == SLIDE_EPI_STATS) {
#pragma unroll
      for (int sc = 0; sc < NSCOPE; ++sc) {
        float s[16], ss[16];
        if (PH != 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            f32x2 t, tt;
            if constexpr (NSCOPE == 1 && RB == 2) { t = v[0][i] + v[1][i]; tt = __builtin_elementwise_fma(v[0][i], v[0][i], v[1][i] * v[1][i]); }
            else { t = v[sc][i]; tt = v[sc][i] * v[sc][i]; }
            s[2 * i] = t[0]; s[2 * i + 1] = t[1]; ss[2 * i] = tt[0]; ss[2 * i + 1] = tt[1];
          }
        }
        reduce_rows(std::integral_constant<int, 16>(), s, ss, false);
        if (PH == 1) continue;
        const int row = row0 + wave * 64 + ((NSCOPE == 1) ? 0 : sc) * 32 + col;
        const bool writer = (NPXL >= 7) ? ((wave % WPS) == 0 && col == 0) : ((col & (LG - 1)) == 0);
        if (writer && row < a.rows) {
          const size_t b = (size_t)(row >> NPXL);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            e_stats_sum[b * e_stats_bs + c] = s[r] * e_stats_scale;
            e_stats_sq[b * e_stats_bs + c] = ss[r] * e_stats_scale;
          }
        }
      }
    } else if (mode == SLIDE_EPI_NORM && PH == 2) {
      // per-(sample, channel) scale g = gamma * rstd and shift beta - mean * g, prepared by finalize_stats
      const float *gp = gsh + ((cb * 2 + (WPS == 2 ? (wave >> 1) : 0)) * 2) * 32 + 4 * half;
      float4 g4[4], b4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        g4[q] = *reinterpret_cast<const float4 *>(gp + 8 * q);
        b4[q] = *reinterpret_cast<const float4 *>(gp + 32 + 8 * q);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          v[rb][2 * q] = __builtin_elementwise_fma(v[rb][2 * q], f32x2{g4[q].x, g4[q].y}, f32x2{b4[q].x, b4[q].y});
          v[rb][2 * q + 1] = __builtin_elementwise_fma(v[rb][2 * q + 1], f32x2{g4[q].z, g4[q].w}, f32x2{b4[q].z, b4[q].w});
        }
    } else if (mode == SLIDE_EPI_NORM) {
      // GroupNorm: groups of gs PHYSICAL channels (gs | 32).  Fold the lane's 16 channels into its groups BEFORE the
      // cross-lane reduction: SH = log2(channels of one group held by this lane) -> 16 >> SH values to reduce.
      auto norm_path = [&](auto sh_tag) __attribute__((always_inline)) {
        constexpr int SH = decltype(sh_tag)::value;
        constexpr int NV = 16 >> SH;
#pragma unroll
        for (int sc = 0; sc < NSCOPE; ++sc) {
          float s[NV], ss[NV];
          if constexpr (PH == 2) {
          } else if constexpr (SH == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x2 t, tt;
              if constexpr (NSCOPE == 1 && RB == 2) {
                t = (v[0][2 * q] + v[1][2 * q]) + (v[0][2 * q + 1] + v[1][2 * q + 1]);
                tt = v[0][2 * q] * v[0][2 * q];
                tt = __builtin_elementwise_fma(v[1][2 * q], v[1][2 * q], tt);
                tt = __builtin_elementwise_fma(v[0][2 * q + 1], v[0][2 * q + 1], tt);
                tt = __builtin_elementwise_fma(v[1][2 * q + 1], v[1][2 * q + 1], tt);
              } else {
                t = v[sc][2 * q] + v[sc][2 * q + 1];
                tt = __builtin_elementwise_fma(v[sc][2 * q], v[sc][2 * q], v[sc][2 * q + 1] * v[sc][2 * q + 1]);
              }
              s[q] = t[0] + t[1]; ss[q] = tt[0] + tt[1];
            }
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              f32x2 t, tt;
              if constexpr (NSCOPE == 1 && RB == 2) { t = v[0][i] + v[1][i]; tt = __builtin_elementwise_fma(v[0][i], v[0][i], v[1][i] * v[1][i]); }
              else { t = v[sc][i]; tt = v[sc][i] * v[sc][i]; }
              if constexpr (SH == 1) { s[i] = t[0] + t[1]; ss[i] = tt[0] + tt[1]; }
              else { s[(2 * i) >> SH] = t[0]; s[(2 * i + 1) >> SH] = t[1]; ss[(2 * i) >> SH] = tt[0]; ss[(2 * i + 1) >> SH] = tt[1]; }
            }
          }
          reduce_rows(std::integral_constant<int, NV>(), s, ss, SH == 2 && e_gs >= 8);
          if (PH == 1) continue;
          if (PH == 2 && cb == 0) SLIDE_STAMP(a, 11);
          if (SH == 2 && e_gs >= 16) {  // groups wider than both halves of a quad: fold quads
            if (e_gs == 16) {
              const float p0 = s[0] + s[1], p1 = s[2] + s[3], q0_ = ss[0] + ss[1], q1_ = ss[2] + ss[3];
              s[0] = s[1] = p0; s[2] = s[3] = p1; ss[0] = ss[1] = q0_; ss[2] = ss[3] = q1_;
            } else {
              const float p = (s[0] + s[1]) + (s[2] + s[3]), q_ = (ss[0] + ss[1]) + (ss[2] + ss[3]);
              s[0] = s[1] = s[2] = s[3] = p; ss[0] = ss[1] = ss[2] = ss[3] = q_;
            }
          }
          float mean[NV], rstd[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            mean[i] = s[i] * e_inv_count;
            const float var = fmaxf(ss[i] * e_inv_count - mean[i] * mean[i], 0.f);
            rstd[i] = __builtin_amdgcn_rsqf(var + GN_EPS);
          }
          if (PH == 2 && cb == 0) SLIDE_STAMP(a, 12);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int c0 = 8 * (i >> 1) + 4 * half + 2 * (i & 1);
            const f32x2 gam = *reinterpret_cast<const f32x2 *>(v_gamma + c0);
            const f32x2 bet = *reinterpret_cast<const f32x2 *>(v_beta + c0);
            const f32x2 rs = {rstd[(2 * i) >> SH], rstd[(2 * i + 1) >> SH]};
            const f32x2 mu = {mean[(2 * i) >> SH], mean[(2 * i + 1) >> SH]};
            f32x2 g = gam * rs;
            f32x2 bt = __builtin_elementwise_fma(-mu, g, bet);
            if (e_n_norm < 32) {  // MyGroupNorm leaves the last C % G channels as they are
              if (c0 >= e_n_norm) { g[0] = 1.f; bt[0] = 0.f; }
              if (c0 + 1 >= e_n_norm) { g[1] = 1.f; bt[1] = 0.f; }
            }
            if constexpr (NSCOPE == 1 && RB == 2) {
              v[0][i] = __builtin_elementwise_fma(v[0][i], g, bt);
              v[1][i] = __builtin_elementwise_fma(v[1][i], g, bt);
            } else {
              v[sc][i] = __builtin_elementwise_fma(v[sc][i], g, bt);
            }
          }
        }
      };
      if (e_gs >= 4) norm_path(std::integral_constant<int, 2>());
      else if (e_gs == 2) norm_path(std::integral_constant<int, 1>());
      else norm_path(std::integral_constant<int, 0>());
    }
    if (PH == 1) return;
    if (PH == 2 && cb == 0) SLIDE_STAMP(a, 9);
    // store.  A lane holds 4 consecutive channels per quad q (channels 8q + 4*half).  fp32 rows go out as they are
    // (16 B per lane).  For fp16 rows an 8-byte store per lane would touch only 16 B of every row per instruction,
    // which the memory system writes at half the rate of wider row pieces (tools/ab/store_pattern.hip: 3.1 vs 5.2 TB/s):
    // v_permlane32_swap trades quads 2p+1 / 2p between the lane halves so that lane (col, half) owns the 8 channels
    // 16p + 8*half .. +7 and issues 16-byte stores (32 B per row and instruction).  The residual is read the same way.
    // Pass 1 consumes every value that came from a global load (t-embedding rows, residual); pass 2 only converts and
    // stores.  Kept apart -- and compiled without the loaded operands when a block has none -- because any wait for a
    // load placed between stores is a vmcnt(0): it would also wait for the stores issued so far, one full write round
    // trip per row block.
    const float relu_lo = (flags & SLIDE_F_POST_RELU) ? 0.f : -3.0e38f;
    auto store_phase = [&](auto addv_tag, auto res_tag) __attribute__((always_inline)) {
      constexpr bool HA = decltype(addv_tag)::value, HR = decltype(res_tag)::value;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 lo = v[rb][2 * q], hi = v[rb][2 * q + 1];
          // ReLU without a branch per quad: clamp from below by 0 or by -FLT_MAX (v_med3_f32, one op per value)
          lo[0] = __builtin_amdgcn_fmed3f(lo[0], relu_lo, 3.0e38f); lo[1] = __builtin_amdgcn_fmed3f(lo[1], relu_lo, 3.0e38f);
          hi[0] = __builtin_amdgcn_fmed3f(hi[0], relu_lo, 3.0e38f); hi[1] = __builtin_amdgcn_fmed3f(hi[1], relu_lo, 3.0e38f);
          if constexpr (HA) {
            const float4 t = apre[rb < NA ? rb : 0][q];
            lo += f32x2{t.x, t.y}; hi += f32x2{t.z, t.w};
          }
          v[rb][2 * q] = lo; v[rb][2 * q + 1] = hi;
        }
        if constexpr (KEEP) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { acc[cb][rb][2 * i] = v[rb][i][0]; acc[cb][rb][2 * i + 1] = v[rb][i][1]; }
        }
        if constexpr (HR && kHalf) {
          if (wide16) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const u32x4 w = rpre[rb][p];
              uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
              lane32_swap(w0, w2);
              lane32_swap(w1, w3);
              const f16x2 a0 = __builtin_bit_cast(f16x2, w0), a1 = __builtin_bit_cast(f16x2, w1);
              const f16x2 b0 = __builtin_bit_cast(f16x2, w2), b1 = __builtin_bit_cast(f16x2, w3);
              v[rb][4 * p] += f32x2{(float)a0[0], (float)a0[1]};
              v[rb][4 * p + 1] += f32x2{(float)a1[0], (float)a1[1]};
              v[rb][4 * p + 2] += f32x2{(float)b0[0], (float)b0[1]};
              v[rb][4 * p + 3] += f32x2{(float)b1[0], (float)b1[1]};
            }
          }
        }
      }
      if constexpr (KEEP) return;
      // PAIR residual on float rows (split plans, 128- / 256-row samples): the table rows of a 16-byte piece pair (row block rb, quads
      // 2p, 2p + 1) are requested ONE PIECE PAIR AHEAD of the stores that consume the previous pair, the coefficient vectors of
      // group_knn's scalars once per block -- a load cannot move above a store the compiler must assume aliases it, so the plain
      // loop below paid one dependent memory round trip per quad: 9.2 of a 17.5 us workgroup on the FP blocks' Mlp layers of the
      // position plan (tools/ab/op_timeline.py)
      if constexpr (HR && !kHalf && NPXL == 7 && PAIRRES && RB == 2) {
        if (pair) {
          const GLOBAL_AS T *resb = gptr<const T>(rdp(34));
          const bool nbr = NPXL == 7 && (flags & SLIDE_F_RES_PAIR_NBR) != 0;
          bool okr[2];
          size_t orow[2];
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const int row = row0 + wave * 64 + rb * 32 + col;
            okr[rb] = row < a.rows;
            orow[rb] = (size_t)row * e_out_ld;
          }
          float4 vd[4], vw[4];
          if (nbr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              vd[q] = gload4(gptr<const float>(rdp(36)) + 8 * q + 4 * half);
              vw[q] = gload4(gptr<const float>(rdp(38)) + 8 * q + 4 * half);
            }
          }
          float4 A[2][2], B[2][2];
          auto fetch = [&](int idx, float4 (&A_)[2], float4 (&B_)[2]) __attribute__((always_inline)) {
            const int p = idx >> 1, rb = idx & 1;
            if (!okr[rb]) return;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c0 = 8 * (2 * p + j) + 4 * half;
              A_[j] = gload4(resid + (size_t)pr_a[rb] * e_res_ld + c0);
              B_[j] = gload4(resb + (size_t)pr_b[rb] * e_res_ld + c0);
            }
          };
          fetch(0, A[0], B[0]);
#pragma unroll
          for (int idx = 0; idx < 4; ++idx) {
            const int p = idx >> 1, rb = idx & 1;
            if (idx + 1 < 4) fetch(idx + 1, A[(idx + 1) & 1], B[(idx + 1) & 1]);
            if (!okr[rb]) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int q = 2 * p + j, c0 = 8 * q + 4 * half;
              float4 y = make_float4(v[rb][2 * q][0], v[rb][2 * q][1], v[rb][2 * q + 1][0], v[rb][2 * q + 1][1]);
              float4 t = A[idx & 1][j];
              const float4 t2 = B[idx & 1][j];
              t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w;
              if (nbr) {
                const float d2 = nbr_d2f[rb], w_ = nbr_wf[rb];
                t.x = fmaf(w_, vw[q].x, fmaf(d2, vd[q].x, t.x)); t.y = fmaf(w_, vw[q].y, fmaf(d2, vd[q].y, t.y));
                t.z = fmaf(w_, vw[q].z, fmaf(d2, vd[q].z, t.z)); t.w = fmaf(w_, vw[q].w, fmaf(d2, vd[q].w, t.w));
              }
              y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
              gstore4(gptr<T>(e_out) + orow[rb] + c0, y);
            }
          }
          return;
        }
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int row = row0 + wave * 64 + rb * 32 + col;
        const bool ok = row < a.rows;  // identical in both lane halves
        // 16-byte piece (p, half) of this lane's row: row-major / chunk-major rows, or FRAGMENT-major (SLIDE_F_OUT_FM: the 32-row
        // group's 2 KB as [p][half][row][8 halves] -- the 64 lanes of one store instruction write 1 KB of consecutive memory)
        // (row = a wave-uniform multiple of 32 + col: the group's base is SCALAR arithmetic, the lane's part one register for both rb)
        const bool fm = FMOK && (flags & SLIDE_F_OUT_FM) != 0;
        const int rbase = __builtin_amdgcn_readfirstlane(row0 + wave * 64) + rb * 32;
        const size_t o16 = (size_t)rbase * (fm ? 32 : e_out_ld) + (fm ? half * 256 + col * 8 : col * e_out_ld + 8 * half);
        const int o16p = fm ? 512 : 16;
#pragma unroll
        for (int p = 0; p < 2; ++p) {  // quads 2p and 2p+1
          if constexpr (kHalf) if (wide16) {
            f16x2 a0 = __builtin_convertvector(v[rb][4 * p], f16x2), a1 = __builtin_convertvector(v[rb][4 * p + 1], f16x2);
            f16x2 b0 = __builtin_convertvector(v[rb][4 * p + 2], f16x2), b1 = __builtin_convertvector(v[rb][4 * p + 3], f16x2);
            uint32_t ua0 = __builtin_bit_cast(uint32_t, a0), ua1 = __builtin_bit_cast(uint32_t, a1);
            uint32_t ub0 = __builtin_bit_cast(uint32_t, b0), ub1 = __builtin_bit_cast(uint32_t, b1);
            lane32_swap(ua0, ub0);
            lane32_swap(ua1, ub1);
            u32x4 o = {ua0, ua1, ub0, ub1};
#if !defined(SLIDE_ABL) || SLIDE_ABL != 1
            if (ok) *(GLOBAL_AS u32x4 *)(gptr<_Float16>(e_out) + o16 + o16p * p) = o;
#else
            if (ok && o[0] == 0x12345678u) *(GLOBAL_AS u32x4 *)(gptr<_Float16>(e_out) + o16 + o16p * p) = o;
#endif
            continue;
          }
          if (ok) {
            // PAIR residual on float rows: the table rows (and group_knn's coefficient vectors) of BOTH quads of this piece are
            // requested together, ahead of the piece's stores -- a load cannot move above a store the compiler must assume aliases
            // it, so one quad at a time was a dependent memory round trip per quad: 9.2 of a 17.5 us workgroup on the FP blocks'
            // Mlp layers of the position plan (tools/ab/op_timeline.py), 6 with two quads per trip
            float4 pa[2], pb[2], pvd[2], pvw[2];
            if constexpr (HR && !kHalf && NPXL >= 7 && PAIRRES) {
              if (pair) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  const int c0 = 8 * (2 * p + j) + 4 * half;
                  pa[j] = gload4(resid + (size_t)pr_a[rb] * e_res_ld + c0);
                  pb[j] = gload4(gptr<const T>(rdp(34)) + (size_t)pr_b[rb] * e_res_ld + c0);
                  if constexpr (NPXL == 7) {
                    if (flags & SLIDE_F_RES_PAIR_NBR) {
                      pvd[j] = gload4(gptr<const float>(rdp(36)) + c0);
                      pvw[j] = gload4(gptr<const float>(rdp(38)) + c0);
                    }
                  }
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int q = 2 * p + j, c0 = 8 * q + 4 * half;
              float4 y = make_float4(v[rb][2 * q][0], v[rb][2 * q][1], v[rb][2 * q + 1][0], v[rb][2 * q + 1][1]);
              if constexpr (HR) {
                float4 t;
                bool plain = true;
                if constexpr (!kHalf && NPXL >= 7 && PAIRRES) {
                  if (pair) {  // PAIR residual on float rows (split mode): ta[q] + tb[p] (+ d2 vd + w vw), evaluated in fp32
                    plain = false;
                    t = pa[j];
                    const float4 t2 = pb[j];
                    t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w;
                    if constexpr (NPXL == 7) {
                      if (flags & SLIDE_F_RES_PAIR_NBR) {
                        const float4 vd = pvd[j], vw = pvw[j];
                        const float d2 = nbr_d2f[rb], w_ = nbr_wf[rb];
                        t.x = fmaf(w_, vw.x, fmaf(d2, vd.x, t.x)); t.y = fmaf(w_, vw.y, fmaf(d2, vd.y, t.y));
                        t.z = fmaf(w_, vw.z, fmaf(d2, vd.z, t.z)); t.w = fmaf(w_, vw.w, fmaf(d2, vd.w, t.w));
                      }
                    }
                  }
                }
                if (plain) t = gload4(resid + (size_t)row * e_res_ld + c0);
                y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
                // (float PAIR instantiations: keep the compiler from hoisting all sixteen table loads of the store phase above the
                //  first store -- 64 registers of loads in flight push the chained kernel of gemm_gxs.hip over its 256)
                if constexpr (!kHalf && NPXL >= 7 && PAIRRES && NOADDV) asm volatile("" ::: "memory");
              }
              if (flags & SLIDE_F_OUT_F32)
                gstore4(gptr<float>(e_out) + (size_t)row * e_out_ld + c0, y);
              else
                gstore4(gptr<T>(e_out) + (size_t)row * e_out_ld + c0, y);
            }
          }
        }
      }
    };
    using TT = std::true_type;
    using FF = std::false_type;
    if (addv) { if (resid) store_phase(TT(), TT()); else store_phase(TT(), FF()); }
    else { if (resid) store_phase(FF(), TT()); else store_phase(FF(), FF()); }
    if (PH == 2 && cb == 0) SLIDE_STAMP(a, 10);
  };
  // Between the phases ONE wave per channel block (wave == cb) turns the partial sums of the sample's waves into what
  // phase 2 needs, one channel per lane: STATS -> the per-(sample, channel) sums in global memory; NORM -> scale and
  // shift per (sample, channel) in LDS.  (Fixed summation order: deterministic.)
  auto finalize_stats = [&](const int cb) __attribute__((always_inline)) {
    constexpr int WPSF = NPXL >= 7 ? (1 << NPXL) / 64 : 1;
    const int cobi = cob0 + cb;
    if (cobi >= a.n_cob) return;
    auto rd = [&](int k) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)epi_lds[cb * EPI_DW + k]); };
    auto rdp = [&](int k) { return (uint64_t)rd(k) | ((uint64_t)rd(k + 1) << 32); };
    const int mode = NOADDV ? (int)SLIDE_EPI_NORM : (int)rd(0);
    if (mode == SLIDE_EPI_RAW) return;
    if (WPSF == 4 && half) return;                    // one sample per workgroup: the upper lane half has nothing to do
    const int smp = WPSF == 2 ? half : 0, w0 = smp * WPSF;
    const int c = col, hh = (c >> 2) & 1, q = c >> 3, j = c & 3;
    auto part = [&](int w, int h2, int slot) {
      return *reinterpret_cast<const f32x2 *>(red + ((((w0 + w) * CBW + cb) * 2 + h2) * 16 + slot) * 2);
    };
    if (mode == SLIDE_EPI_STATS) {
      f32x2 t = {0.f, 0.f};
#pragma unroll
      for (int w = 0; w < WPSF; ++w) t += part(w, hh, 4 * q + j);
      const int row = row0 + smp * NPX;
      if (row < a.rows) {
        const float scale = __uint_as_float(rd(5));
        const size_t o = (size_t)(row >> NPXL) * (int)rd(9) + c;
        gptr<float>(rdp(30))[o] = t[0] * scale;
        gptr<float>(rdp(32))[o] = t[1] * scale;
      }
      return;
    }
    const int gs = (int)rd(2), n_norm = (int)rd(3);
    const float inv_count = __uint_as_float(rd(4));
    int slot0, nslot = 1, nh = 1, h0 = hh;
    if (gs >= 4) {
      slot0 = q;
      if (gs >= 8) { nh = 2; h0 = 0; }
      if (gs >= 16) { nslot = gs >> 3; slot0 = (q / nslot) * nslot; }
    } else if (gs == 2) {
      slot0 = 2 * q + (j >> 1);
    } else {
      slot0 = 4 * q + j;
    }
    // fully unrolled per group shape so that the LDS reads of a lane issue back to back (a runtime-bounded loop
    // serialises them behind one another's latency)
    auto sum_parts = [&](auto nh_tag, auto ns_tag) __attribute__((always_inline)) {
      constexpr int NH = decltype(nh_tag)::value, NS = decltype(ns_tag)::value;
      f32x2 acc2 = {0.f, 0.f};
#pragma unroll
      for (int w = 0; w < WPSF; ++w)
#pragma unroll
        for (int h2 = 0; h2 < NH; ++h2)
#pragma unroll
          for (int sl = 0; sl < NS; ++sl) acc2 += part(w, h0 + h2, slot0 + sl);
      return acc2;
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    f32x2 t;
    if (nh == 1) t = sum_parts(I1(), I1());
    else if (nslot == 1) t = sum_parts(I2(), I1());
    else if (nslot == 2) t = sum_parts(I2(), I2());
    else t = sum_parts(I2(), I4());
    const float mean = t[0] * inv_count;
    const float var = fmaxf(t[1] * inv_count - mean * mean, 0.f);
    float g = vec_lds[cb * 96 + 32 + c] * __builtin_amdgcn_rsqf(var + GN_EPS);
    float bt = vec_lds[cb * 96 + 64 + c] - mean * g;
    if (c >= n_norm) { g = 1.f; bt = 0.f; }  // MyGroupNorm leaves the last C % G channels as they are
    gsh[((cb * 2 + smp) * 2 + 0) * 32 + c] = g;
    gsh[((cb * 2 + smp) * 2 + 1) * 32 + c] = bt;
  };
#if defined(SLIDE_ABL) && SLIDE_ABL == 3
  if (NPXL >= 7 && a.rows > 0) return;
#endif
  if (NPXL >= 7) {
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) process(cb, std::integral_constant<int, 1>());
    SLIDE_STAMP(a, 3);
    __syncthreads();
    SLIDE_STAMP(a, 4);
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
      if (wave == cb) finalize_stats(cb);
    __syncthreads();
#if defined(SLIDE_ABL) && SLIDE_ABL == 2
    if (a.rows > 0) return;
#endif
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) process(cb, std::integral_constant<int, 2>());
  } else {
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) process(cb, std::integral_constant<int, 0>());
  }
}

}  // namespace
