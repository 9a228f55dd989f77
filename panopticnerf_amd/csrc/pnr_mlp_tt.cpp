// Launcher of k_mlp_tt, the two-tile assembly form of the fused inference MLP (csrc/asm/gen_mlp_tt.py): the code object is
// assembled at build time, embedded in libpnr.so (build/obj/pnr_mlp_tt_co.inc) and loaded once per device with
// hipModuleLoadData -- the "once-initialised module / kernel table" include/pnr.h allows next to the otherwise stateless library.
// The first call on a device must therefore happen OUTSIDE a stream capture (module loading is not capturable); later calls are
// plain kernel launches and capture like every other entry point.
#include <mutex>
#include <string>

#include "pnr_common.h"
#include "pnr_mlp_tt.h"

static const unsigned char k_co[] = {
#include "pnr_mlp_tt_co.inc"
};

namespace {
struct DevTable {
    hipModule_t mod = nullptr;
    hipFunction_t fn[8][4][2] = {};     // [variant = 4 head_tap + 2 softmax + (head_depth == 1)][nbs][nbi]; without heads: variant 0
    hipFunction_t fn_trace[8] = {};     // [ablation]: 0 = the trace build; 1, 2, 3, 4, 7 exist only in PNR_TT_ABL=1 builds of the library
    bool tried = false, ok = false;
};
DevTable g_tab[64];
std::mutex g_mu;
}   // namespace

// capturing != 0: the caller's stream is being captured -- a module that is not loaded yet must not be loaded now (hipModuleLoadData
// under a capture fails and, in the global capture mode, invalidates the capture): report it instead
static int tt_table(DevTable*& out, int capturing = 0)
{
    int dev = 0;
    PNR_HIP(hipGetDevice(&dev));
    PNR_REQUIRE(dev >= 0 && dev < 64, "pnr_mlp_tt: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_mu);
    DevTable& t = g_tab[dev];
    if (!t.tried) {
        PNR_REQUIRE(!capturing, "pnr_mlp_tt: the two-tile code object is not loaded on device %d yet and the stream is capturing: make the "
                    "first plan-2 call (or pnr_mlp_pack_device / pnr_mlp_packed_bytes with plan = 2) outside the capture", dev);
        t.tried = true;
        hipError_t e = hipModuleLoadData(&t.mod, k_co);
        const bool loaded = e == hipSuccess;
        static const int geo[6][2] = {{1, 1}, {2, 1}, {0, 0}, {1, 0}, {2, 0}, {3, 0}};      // (semantic, instance) logit blocks of the generated kernels
        for (int k = 0; k < 6 && e == hipSuccess; ++k) {
            // names as csrc/asm/gen_mlp_tt.py::variant_name spells them: k_mlp_tt_[f][d1][sm]_s<n>i<m>
            for (int v = 0; v < (geo[k][0] ? 8 : 1) && e == hipSuccess; ++v) {
                char tag[16], nm[64];
                snprintf(tag, sizeof(tag), "%s%s%s", (v & 4) ? "f" : "", (v & 1) ? "d1" : "", (v & 2) ? "sm" : "");
                snprintf(nm, sizeof(nm), "k_mlp_tt_%s%ss%di%d", tag, tag[0] ? "_" : "", geo[k][0], geo[k][1]);
                e = hipModuleGetFunction(&t.fn[v][geo[k][0]][geo[k][1]], t.mod, nm);
            }
        }
        // diagnostics kernels: only in `make EXTRA_TT=trace | abl` builds of the library
        for (int a = 0; a < 8 && e == hipSuccess; ++a) {
            char nm[64];
            if (a) snprintf(nm, sizeof(nm), "k_mlp_tt_s2i1_trace_a%d", a); else snprintf(nm, sizeof(nm), "k_mlp_tt_s2i1_trace");
            if (hipModuleGetFunction(&t.fn_trace[a], t.mod, nm) != hipSuccess) { t.fn_trace[a] = nullptr; (void)hipGetLastError(); }
        }
        if (e != hipSuccess) {
            pnr_set_error("pnr_mlp_tt: loading the two-tile code object failed: %s (the first call on a device must not be inside a "
                          "stream capture)", hipGetErrorString(e));
            if (loaded) (void)hipModuleUnload(t.mod);       // a partial failure must not leak the module across the retry
            t = DevTable();         // let the next call (outside a capture) try again
            (void)hipGetLastError();
            return PNR_EHIP;
        }
        t.ok = true;
    }
    PNR_REQUIRE(t.ok, "pnr_mlp_tt: the two-tile code object is not loaded");
    out = &t;
    return PNR_OK;
}

int pnr_mlp_tt_prepare(void)
{
    DevTable* t = nullptr;
    return tt_table(t);
}

// Best effort, for the entry points that are pure CPU work (pnr_mlp_pack, pnr_mlp_packed_bytes: they also run where there is no
// device): if a device is present, load the code object now, so that an image packed on the host and first launched inside a
// stream capture finds it loaded (the launch itself refuses to load under a capture).  Never fails, keeps pnr_last_error().
void pnr_mlp_tt_prepare_quiet(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return; }
    const std::string keep = pnr_last_error();
    DevTable* t = nullptr;
    if (tt_table(t) != PNR_OK) pnr_set_error("%s", keep.c_str());
}

int pnr_mlp_tt_launch(const PnrTTArgs& a, int nbs, int nbi, int head_depth, int head_tap, bool softmax, hipStream_t stream, bool trace, int trace_abl)
{
    PNR_REQUIRE(nbs >= 0 && nbs <= 3 && nbi >= 0 && nbi <= (nbs == 1 || nbs == 2 ? 1 : 0), "pnr_mlp_forward_composite: no two-tile kernel for %d + %d logit blocks", nbs, nbi);
    PNR_REQUIRE(a.S >= 1 && a.S < (1 << 28), "pnr_mlp_forward_composite: the two-tile kernel takes R*N < 2^28 samples per launch (got %d): "
                "render in chunks", a.S);
    DevTable* t = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    PNR_HIP(hipStreamIsCapturing(stream, &cs));
    int rc = tt_table(t, cs != hipStreamCaptureStatusNone);
    if (rc != PNR_OK) return rc;
    hipFunction_t fn = t->fn[!nbs ? 0 : (head_tap ? 4 : 0) + (softmax ? 2 : 0) + (head_depth == 1 ? 1 : 0)][nbs][nbi];
    PNR_REQUIRE(fn, "pnr_mlp_tt: no kernel for %d + %d logit blocks at head_depth %d", nbs, nbi, head_depth);
    if (trace) {
        PNR_REQUIRE(head_depth != 1 && !softmax && !head_tap, "pnr_mlp_tt: the trace build exists for head_depth 2, head_tap 0, logits compositing");
        PNR_REQUIRE(nbs == 2 && nbi == 1 && a.clk, "pnr_mlp_tt: the trace build exists for 2 + 1 logit blocks and needs the clock buffer");
        fn = t->fn_trace[trace_abl & 7];
        PNR_REQUIRE(fn, "pnr_mlp_tt: PNR_MLP_TRACE%s%d is not in this library: the trace kernels are diagnostics builds "
                    "(make EXTRA_TT=trace, ablations EXTRA_TT=abl)", trace_abl & 7 ? " ablation " : " ", trace_abl & 7);
    }
    PnrTTArgs ka = a;
    const int cus = pnr_cu_count();
    ka.n_groups = (a.S + 255) / 256;
    const int cap = (a.n_wg > 0 && a.n_wg < cus) ? a.n_wg : cus;       // PNR_MLP_WG_CAP: a share of the device for a launch that runs beside another
    ka.n_wg = ka.n_groups < cap ? ka.n_groups : cap;
    size_t size = sizeof(ka);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    PNR_HIP(hipModuleLaunchKernel(fn, (unsigned)ka.n_wg, 1, 1, 256, 1, 1, 0, stream, nullptr, extra));
    return PNR_OK;
}
