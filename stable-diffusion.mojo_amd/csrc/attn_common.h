// attn_common.h - what the two flash-attention kernel files share (device code only): the kernel argument block, the fragment types and
// the constants of the lazy-reference softmax.  kernels_attn.hip holds flash_attn_kernel<D,QB> (4-wave workgroups, every head dim) and
// the host dispatch; kernels_attn8.hip holds flash_attn8_kernel<D> (8 waves in two groups one phase apart, the long d = 40 key loops).
#pragma once
#include "common.h"
#include "lds_dma.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct AttnK {
  const half_t* Q; const half_t* K; const half_t* Vt; half_t* O; const half_t* zeros; const half_t* ones;
  long long sQ, sK, sVt, sO;
  int ldq, ldk, ldvt, ldo;
  int H, Sq, Sk, Skv;  // Skv: number of valid V^T columns (Sk rounded up to 8)
  float c;             // scale * log2(e)
  int diag;            // self-attention (Sq == Sk): the optimistic reference also covers each query's own 32-key block
  int xcd_map;         // 1: (head, query tile) remapped so that every XCD (dispatch id % 8) owns whole heads - its L2 then pulls a head's
                       // K / V^T once instead of every XCD pulling every head's (needs B*H % 8 == 0)
  int* exact_ctr;      // the context's count of workgroups that had to run the exact pass (tsd_debug_attn_exact_passes)
};

// Largest score (log2 units, relative to the running reference) a tile may reach before the reference is moved.
// P = exp2(s - ref) is then at most 2^12 - far inside fp16 (65504) and harmless for the fp32 O / row-sum accumulators.
#define TSD_ATTN_LAZY 12.0f
// Optimistic pass: ref = (row maximum of tile 0) + this, so a later score may exceed tile 0's maximum by 16 + 4 = 20 log2 units
// (13.9 nats) before an fp16 P overflows and the exact pass has to run; the largest P of tile 0 is then 2^-4, still 2^10
// above the smallest normal fp16.
#define TSD_ATTN_HEADROOM 4.0f
#ifndef TSD_ATTN_CHECK_EVERY
#define TSD_ATTN_CHECK_EVERY 8  // key tiles between two looks at the row sums in the optimistic softmax pass (early abort)
#endif

// An (empty) use of a 16-register block.  A device function, so the host pass never sees the "v" constraint.
__device__ __forceinline__ void keep_alive(const f16v& v) { asm volatile("" ::"v"(v)); }

// kernels_attn8.hip: the 8-wave two-group kernel (d = 40, Sk >= 512); variant = timing / ablation build selector (0 = shipped)
int launch_flash_attention8(tsd_ctx* ctx, const AttnK& k, int B, int H, int Sq, int d, int variant);
