/*
 * pnr_bench.h -- C-ABI of libpnr_bench.so: measurement helpers for bench.py and tools/ only.  NOT part of the product
 * library: every entry point here times launches with hipEvents and therefore SYNCHRONISES its stream (libpnr.so, include/pnr.h,
 * never does).  The product entry points that are timed are handed over by address (pnrb_bind), so the numbers are those of
 * the libpnr.so the caller loaded.  int return: 0 = ok, negative = PNR_E* (pnr.h); pnrb_last_error() gives the text.
 */
#ifndef PNR_BENCH_H
#define PNR_BENCH_H
#include <stdint.h>

#include "pnr.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* pnrb_last_error(void);
/* Addresses of pnr_mlp_forward and pnr_mlp_forward_tiles of the loaded libpnr.so. */
int pnrb_bind(void* mlp_forward, void* mlp_forward_tiles);

/* Mean milliseconds per launch of `iters` launches of pnr_mlp_forward (hipEvents recorded on `stream`, the stream the
 * launches go to) and the mean SHADER CLOCK during the last one (s_memtime / s_memrealtime of workgroup 0's first wave).
 * scratch: >= 16 device bytes.  ms / mhz: host floats. */
int pnrb_time_mlp_forward(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z, int64_t n_rays,
                          int n_samples, float* raw, int64_t raw_stride_s, int64_t raw_stride_c, int iters, void* scratch,
                          float* ms_out_host, float* mhz_out_host, void* stream);
/* The same for pnr_mlp_forward_tiles: the fused inference MLP launch alone (what pnr_mlp_forward_composite runs before its
 * small combine kernel).  workspace: pnr_mlp_forward_composite_workspace_bytes device bytes. */
int pnrb_time_mlp_forward_tiles(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z, int64_t n_rays,
                                int n_samples, void* workspace, int iters, void* scratch, float* ms_out_host,
                                float* mhz_out_host, void* stream);
/* What the matrix pipe of this device SUSTAINS: a register-only bf16 MFMA loop on every SIMD with constant operands
 * (random_operands = 0) or with pseudo-random operands that change from MFMA to MFMA (1: the toggle rate of real data -- on
 * MI355X the clock then drops from ~2.37 to ~1.83 GHz and the rate from ~2.46 to ~1.83 PFLOP/s).  scratch: >= 32 device bytes. */
int pnrb_probe_mfma_peak(int random_operands, int iters, void* scratch, float* tflops_out_host, float* mhz_out_host, void* stream);

/* The same loop on random operands with the ORDER of operand changes as the variable (0: both change every MFMA; 1: the two-tile
 * kernel's (b0,t0) (b0,t1) (b1,t0) (b1,t1); 2: the snake (b0,t0) (b0,t1) (b1,t1) (b1,t0); 3: only A changes; 4: only B changes):
 * what the part's power limit makes of operand toggling.  Synchronises the stream. */
int pnrb_probe_mfma_order(int pattern, int iters, void* scratch, float* tflops_out_host, float* mhz_out_host, void* stream);
/* What HBM delivers for k_composite's own access pattern with no arithmetic: a pure read of the channel-major raw image, per
 * wave the 8 channel rows of a batch of one ray, 8 loads in flight.  scratch: >= 1 KiB. */
int pnrb_probe_raw_read(const float* raw, int64_t raw_stride_c, int64_t n_rays, int n_samples, int n_channels, int iters,
                        void* scratch, float* gbs_out_host, void* stream);

/* The same for any candidate mapping of a ray onto lanes: lanes_per_ray (8 | 16 | 32 | 64) x n_samples / lanes_per_ray consecutive
 * samples per lane, 64 / lanes_per_ray rays per wave, rows_in_flight (4 | 8 | 16) channel rows requested at once, a grid of
 * waves_per_simd x 4 waves per CU at most (tools/composite_patterns.py: the table of profiles/r06/r06q). */
int pnrb_probe_raw_read_pattern(const float* raw, int64_t raw_stride_c, int64_t n_rays, int n_samples, int n_channels, int lanes_per_ray,
                                int rows_in_flight, int waves_per_simd, int iters, void* scratch, float* gbs_out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNR_BENCH_H */
