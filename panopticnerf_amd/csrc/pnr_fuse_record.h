// What the fused MLP epilogue (pnr_mlp_fuse.h) hands to k_composite_combine (pnr_composite.hip): record layout and the
// fixed-point scale of the bbox-prior histograms.
#pragma once

#define PNR_FUSE_FIX_SCALE 1073741824.0f      /* 2^30: fixed-point bins of the bbox-prior histograms (a ray's weights sum to <= 1) */

// per-tile record (floats): [0] Q  [1 .. 1+C) semantic  [1+C .. 1+C+K) instance logit sums; padded to a multiple of 4.
// per sample: one float4 (lw, r, g, b).
__host__ __device__ static inline int pnr_fuse_record_floats(int C, int K) { return (1 + C + K + 3) & ~3; }
#define PNR_FUSE_REC_LOGITS 1                  /* first logit column of a record */
