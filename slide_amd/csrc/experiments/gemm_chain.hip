// gemm_chain.hip -- consecutive PER-POINT layers of the denoiser (16 rows per sample) as ONE launch (SLIDE_OP_GEMM_CHAIN).
//
// Between the K-expanded bodies of two blocks the plan runs short chains of 16-row GEMMs: the second Mlp_plus_t_emb of an FP
// block (first_mlp | res_connect, then second_mlp + residual: pointnet2_modules.py:842-873) and, after the last block, the
// output head (fc_lyaer, pointnet2_with_pcld_condition.py:480-483).  Each was a launch of the split-K small kernel
// (engine.hip): 6 .. 13 us of mostly launch / prologue / drain for a few MFLOP per sample, and -- four chains being in flight
// in bench.py's arrangement -- several hundred workgroups each on the CUs the other chains' large kernels want.
//
// Here a workgroup owns 64 rows (four samples) and walks the chain's layers itself: a layer's GroupNorm is per sample and
// per-point layers have no neighbours, so nothing crosses workgroups.  Per layer: the input rows are staged into LDS once
// ([64][k_pad + 8] fp16: conflict-free ds_read_b128 for the B fragments), wave w computes the 32-channel blocks w, w + 8, ..
// for both 32-row blocks -- weights go straight from L2 into the A-fragment registers (row-major [n][k_pad], 32 bytes per lane
// and 32-deep chunk, prefetched four chunks ahead; every weight byte is read once per workgroup) -- and each block leaves
// through the COMMON epilogue (gemm_common.h: bias, embedding rows, GroupNorm, ReLU, residual, stores), so a fused chain
// computes exactly what its launches computed, up to the summation order of the fp32 accumulators (one wave per block
// instead of four K slices).  Layer outputs go through memory (L2) as before; a workgroup barrier after the stores orders
// them for the next layer's staging (same CU: the vector L1 holds no line of them before that).
#include "gemm_common.h"

namespace {

constexpr int CHAIN_MAX_LAYERS = 6;
constexpr int SLIDE_MAX_DEVICES_C = 64;

struct ChainLayer {
  const void *X, *W;
  const SlideEpi *epi;
  int x_ld, k_pad, n_cob, pad;
};

struct ChainArgs {
  ChainLayer l[CHAIN_MAX_LAYERS];
  int n_layers, rows;
  unsigned long long *dbg;
};

__global__ __launch_bounds__(512) void gemm_chain_kernel(ChainArgs c) {
  using T = _Float16;
  constexpr int NPXL = 4, PD = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int row0 = blockIdx.x * 64;
  uint32_t *const epi_w = reinterpret_cast<uint32_t *>(smem_raw) + wave * (EPI_DW + 96);  // this wave's descriptor + vectors
  float *const vec_w = reinterpret_cast<float *>(epi_w + EPI_DW);
  T *const xs = reinterpret_cast<T *>(smem_raw + 8 * (EPI_DW + 96) * 4);
  GemmArgs ga = GemmArgs();
  ga.rows = c.rows;
  ga.dbg = nullptr;
#pragma unroll 1
  for (int li = 0; li < c.n_layers; ++li) {
    const ChainLayer L = c.l[li];
    const int k_pad = L.k_pad, ldx = k_pad + 8, nk = k_pad >> 5;
    ga.n_cob = L.n_cob;
    // ---- stage the 64 input rows
    {
      const int ppr = k_pad >> 3;  // 16-byte pieces per row
      for (int i = tid; i < 64 * ppr; i += 512) {
        const int r = i / ppr, pc = i - r * ppr;
        int grow = row0 + r;
        grow = grow < c.rows ? grow : c.rows - 1;
        const u32x4 v = *(const GLOBAL_AS u32x4 *)(gptr<const T>((uint64_t)L.X) + (size_t)grow * L.x_ld + pc * 8);
        *reinterpret_cast<u32x4 *>(xs + (size_t)r * ldx + pc * 8) = v;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int cob = wave; cob < L.n_cob; cob += 8) {
      // descriptor + per-channel vectors of this block into the wave's own LDS area (read back by the epilogue)
      if (lane < EPI_DW) epi_w[lane] = reinterpret_cast<const uint32_t *>(L.epi + cob)[lane];
      const GLOBAL_AS T *wp = gptr<const T>((uint64_t)L.W) + (size_t)(cob * 32 + col) * k_pad + half * 16;
      f16x8 wq[PD][2];
#pragma unroll
      for (int j = 0; j < PD; ++j)
        if (j < nk) {
          wq[j][0] = *(const GLOBAL_AS f16x8 *)(wp + j * 32);
          wq[j][1] = *(const GLOBAL_AS f16x8 *)(wp + j * 32 + 8);
        }
      {
        const SlideEpi *ed = L.epi + cob;
        const float *b_ = ed->bias, *g_ = ed->gamma, *t_ = ed->beta;
        if (lane < 32) {
          vec_w[lane] = b_ ? b_[lane] : 0.f;
          vec_w[32 + lane] = g_ ? g_[lane] : 0.f;
          vec_w[64 + lane] = t_ ? t_[lane] : 0.f;
        }
      }
      f32x16 acc[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
      const T *xb0 = xs + (size_t)col * ldx + half * 16, *xb1 = xb0 + (size_t)32 * ldx;
#pragma unroll 1
      for (int kc0 = 0; kc0 < nk; kc0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
          const int kc = kc0 + j;
          if (kc < nk) {
            const f16x8 a0 = wq[j][0], a1 = wq[j][1];
            const f16x8 b00 = *reinterpret_cast<const f16x8 *>(xb0 + kc * 32), b01 = *reinterpret_cast<const f16x8 *>(xb0 + kc * 32 + 8);
            const f16x8 b10 = *reinterpret_cast<const f16x8 *>(xb1 + kc * 32), b11 = *reinterpret_cast<const f16x8 *>(xb1 + kc * 32 + 8);
            if (kc + PD < nk) {
              wq[j][0] = *(const GLOBAL_AS f16x8 *)(wp + (kc + PD) * 32);
              wq[j][1] = *(const GLOBAL_AS f16x8 *)(wp + (kc + PD) * 32 + 8);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b00, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b10, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b01, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b11, acc[1], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's descriptor / vectors are in LDS
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        f32x16 one[1][1];
        one[0][0] = acc[rb];
        gemm_epilogue<SLIDE_PREC_F16, NPXL, 1, 1>(ga, one, row0 + rb * 32, cob, 0, half, col, epi_w, vec_w, nullptr);
      }
    }
    // the layer's stores are complete and visible to the workgroup before the next layer stages its input (and this layer's
    // staged rows are dead)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

}  // namespace

// SLIDE_OP_GEMM_CHAIN (include/slide_engine.h): p[0] = HOST pointer to the layer table (SlideChainLayer[n]), i[0] = rows,
// i[1] = n layers
int slide_launch_gemm_chain(const SlideOp &o, hipStream_t s) {
  const SlideChainLayer *hl = (const SlideChainLayer *)o.p[0];
  ChainArgs c;
  c.rows = o.i[0]; c.n_layers = o.i[1];
  c.dbg = nullptr;
  if (!hl || c.rows <= 0 || c.rows % 16 || c.n_layers < 1 || c.n_layers > CHAIN_MAX_LAYERS) return -3;
  int kmax = 0;
  for (int i = 0; i < c.n_layers; ++i) {
    c.l[i].X = hl[i].X; c.l[i].W = hl[i].W; c.l[i].epi = (const SlideEpi *)hl[i].epi;
    c.l[i].x_ld = hl[i].x_ld; c.l[i].k_pad = hl[i].k_pad; c.l[i].n_cob = hl[i].n_cob; c.l[i].pad = 0;
    if (!hl[i].X || !hl[i].W || !hl[i].epi || hl[i].k_pad % 32 || hl[i].k_pad <= 0 || hl[i].x_ld % 8 || hl[i].x_ld < hl[i].k_pad ||
        hl[i].n_cob <= 0)
      return -3;
    kmax = hl[i].k_pad > kmax ? hl[i].k_pad : kmax;
  }
  const size_t shm = (size_t)8 * (EPI_DW + 96) * 4 + (size_t)64 * (kmax + 8) * 2;
  if (shm > 160 * 1024) return -8;
  static bool attr_done[SLIDE_MAX_DEVICES_C] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool &attr_set = attr_done[dev >= 0 && dev < SLIDE_MAX_DEVICES_C ? dev : 0];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_chain_kernel, dim3((c.rows + 63) / 64), dim3(512), shm, s, c);
  return (int)hipGetLastError();
}
