// Kernel arguments of k_mlp_tt (csrc/asm/gen_mlp_tt.py reads them with s_load at these byte offsets) and its launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PnrTTArgs {
    const void* image;          //  0  fragment stream of the plan-2 image (packed + data_off)
    const float* rays;          //  8
    const float* z;             // 16
    int32_t S, N;               // 24  samples, samples per ray
    uint32_t n_magic;           // 32  x / N = (x * n_magic) >> n_shift (pnr_set_div_magic)
    int32_t n_shift;            // 36
    int32_t n_groups, n_wg;     // 40  256-sample groups (filled by the launcher), workgroups (in: a cap, 0 = every compute unit; out: the grid)
    float* rec;                 // 48  per-tile records
    int32_t rec_floats, pad0;   // 56
    void* ps;                   // 64  per-sample (lw, r, g, b)
    int32_t n_sem, n_inst;      // 72
    unsigned long long* clk;    // 80  optional {shader cycles, 100 MHz ticks} of workgroup 0's first wave
    const void* aux;            // 88  per-ray table of k_ray_aux (pnr_mlp.hip): [ray][half-wave]{8 packed gamma(d) registers, |d|, 7 pad}
    void* save;                 // 96  null; A/B builds of the training-forward prototype (PNR_TT_SAVE=1 kernels): 352 KiB per (group, wave)
};
static_assert(sizeof(PnrTTArgs) == 104, "k_mlp_tt reads its arguments at fixed offsets");

// trace: the debug build of the kernel (EXTRA_TT=trace | abl builds only): 64 per-unit s_memtime stamps of workgroup 0's first wave
// to clk[0..63] and every workgroup's cycles to clk[64 + workgroup], 4 bytes each -- `clk` must hold (64 + n_wg) * 4 bytes
// (tools/tt_trace.py allocates 1280 for 256 workgroups); the plain kernels write 16 bytes.
// load the code object on the current device if that has not happened yet (not capturable: called from the device packer)
int pnr_mlp_tt_prepare(void);
// the same, best effort and silent, from the CPU-side packing entry points (no device / a capturing thread: nothing happens)
void pnr_mlp_tt_prepare_quiet(void);
int pnr_mlp_tt_launch(const PnrTTArgs& a, int nbs, int nbi, int head_depth, int head_tap, bool softmax, hipStream_t stream, bool trace = false,
                      int trace_abl = 0);
