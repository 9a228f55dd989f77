#!/usr/bin/env python3
"""Generator of k_mlp_tt: the fused inference MLP (NeRF trunk + heads + compositing epilogue) as
HAND-PLACED gfx950 assembly in the two-tile form -- ONE wave per SIMD (4 waves per workgroup, 512 registers per lane), TWO
32-sample tiles per wave, every 1 KiB weight fragment read from the LDS feeds two v_mfma_f32_32x32x16_bf16.

Why assembly: the form needs the activations in AGPRs (MFMA B operand), the accumulators in VGPRs (the pack / ReLU reads them
without a move) and every filler instruction placed in a specific MFMA gap; hipcc's schedule of the same loop runs 46 cycles per
MFMA where the hand-placed one runs 35.4 and the 8-wave ping-pong kernel 39.2 (profiles/README.md, round 5: r05a).

Same arithmetic as k_mlp_pp<256, false, FUSE, plan 1> (pnr_mlp.hip, pnr_mlp_fuse.h), operation for operation and in the same
order per value: outputs (per-tile records, per-sample quadruples) are BIT-IDENTICAL (tests/test_gpu_stages.py).  Consumes the
plan-2 image (pnr_mlp_plan.h): plan 1's chunk order, every chunk <= 33 fragments, sem1 / inst1 one chunk each.

Geometry (fixed): D = 8, W = 256, skip = 4, xyz_L = 10, dir_L = 4, head_W = 128, head_tap 0; NBS = 0 | 1 | 2 semantic logit blocks, with
NBS >= 1 also NBI = 0 | 1 instance logit blocks -> five kernels k_mlp_tt_s<NBS>i<NBI> (s0i0: no heads; s<n>i0: no instance head), and
four more for head_depth = 1 (one Linear W -> n per head, read from h: k_mlp_tt_d1_s<NBS>i<NBI>).

Time structure per 256-sample group (one workgroup; tile = (wave, t), record index grp * 8 + wave * 2 + t):
  image chunk c lives in LDS slot c % 4 (33 KiB each); during chunk c every wave issues its LDS-DMA pieces of chunk c + 3; at
  the end of chunk c: s_waitcnt vmcnt (own pieces of chunk c + 2 landed) + ONE s_barrier, so chunk c + 2 is readable during chunk
  c + 1.  MFMAs are grouped in UNITS (<= 2 output blocks x all k-steps x 2 tiles, one accumulator per (block, tile)); a unit's
  accumulators are packed / reduced while the NEXT unit's MFMAs run, the unit after that finds its bias already in its
  accumulators.  Weight fragments pass through a ring of 4 register quads, 3 fragments ahead, counted lgkmcnt.
  Side work (input fetch and gamma(x) of the NEXT group, the compositing epilogue) is a queue of instructions drained a few
  per MFMA gap; what depends on the RAY alone (|d|, gamma(d / |d|)) is loaded from the table k_ray_aux (pnr_mlp.hip) wrote."""
import struct
import sys
from contextlib import contextmanager

import os
# Cache policy of the LDS-DMA weight pieces: the DEFAULT one.  With " nt" (what k_mlp_pp uses, -1 % there) this kernel's weight stream
# missed the L2 for 20 % of its 69 GB per fine-level launch -- 13.8 GB of fabric traffic per launch (rocprofv3 FETCH_SIZE; k_mlp_pp: 2.1 GB,
# the default policy: 0.09 GB) -- and the part paid for it in clock: 11.5 ms at 1773-1793 MHz with nt, 10.66 ms at 1863-1877 MHz without
# (same box, profiles/r05/r05p).  tools/build_tt_variant.sh builds A/B variants (PNR_TT_DMA_POLICY=" nt" | " sc0" | " sc1").
DMA_POLICY = os.environ.get("PNR_TT_DMA_POLICY", "")
STORE_NT = ""           # cache policy of the record / quadruple stores (" nt": measured +-0, round 5)
PIECE_FRAC = 1.0        # the pieces of a chunk go out in this first fraction of its gaps (0.5: measured slower)
WAIT2 = os.environ.get("PNR_TT_WAIT2", "0") != "0"             # (A/B builds) one s_waitcnt lgkmcnt per two fragments instead of one per fragment
# (A/B builds, tools/build_tt_variant.sh) the TRAINING-FORWARD PROTOTYPE: every packed activation block (trunk, feature, views, head hidden layers:
# 5.4 KB per sample, what k_mlp_fused<TRAIN> saves for the backward pass) is also stored to a scratch region -- 1 KiB per store, fully
# coalesced, layout arbitrary: the kernel's results stay valid, the stores are what is being timed (profiles/r06/r06p).  The library must be
# told too: PNR_TT_SAVE_PROTO=1 in the environment makes the launcher allocate and pass the region.
SAVE_PROTO = os.environ.get("PNR_TT_SAVE", "0") != "0"
ABL_ALL = int(os.environ.get("PNR_TT_ABL", "0"))        # (A/B builds; results INVALID) the timing-only ablations in every kernel: 1 = no weight pieces
SAVE_REGION = 352 * 1024                # bytes per wave and group: 2 tiles x 32 samples x 5.5 KiB
# The pack / ReLU / v_accvgpr_write triples as a three-stage software pipeline over the packed registers of a unit (closure j: write of pair
# j - 2, ReLU of pair j - 1, convert of pair j; three rotating temporaries): no instruction of a closure depends on another one of the same
# closure, where the plain form is three DEPENDENT instructions back to back (A/B builds: PNR_TT_PACK_PIPE=0)
PACK_PIPE = os.environ.get("PNR_TT_PACK_PIPE", "1") != "0"
V_T2 = 15                               # third pack temporary (V_TRACE's register: trace / save-prototype builds keep the plain form)
SHARE_BIAS = os.environ.get("PNR_TT_SHARE_BIAS", "1") != "0"     # one armed accumulator per block: tile 1's first MFMA reads tile 0's (A/B builds)
NSLOT, SLOT = 4, int(os.environ.get("PNR_TT_SLOT_KIB", "33")) * 1024      # (A/B builds: the slot stride, tools/build_tt_variant.sh)
ACC_PERM = [int(x) for x in os.environ.get("PNR_TT_ACC_PERM", "0,1,2,3,4,5,6,7").split(",")]     # register block of accumulator i (A/B builds)
SLOT_POS = [int(x) for x in os.environ.get("PNR_TT_SLOT_ORDER", "0,1,2,3").split(",")]      # physical position of logical slot c % 4


def slot_base(sl):
    return SLOT_POS[sl] * SLOT
P = 4                                   # fragment ring (quads)
D, SKIP = 8, 4

# ---- VGPR map
V_TID, V_LANE16, V_FRAG, V_BIAS, V_T0, V_DMA, V_LB4, V_TRACE = 0, 1, 2, 6, 10, 11, 14, 15
# V_DMA + m: lane * 16 + wave * 4 KiB + m * 16 KiB: the LDS-DMA pieces' address offsets.  A wave copies the fragments
# 16 m + 4 wave + i (i < 4: the instruction's immediate offset i KiB) of a chunk -- runs of four, round-robin over the waves.
V_T1 = 1    # second pack temporary (lane * 16 is only needed to build the bases): one temporary for consecutive packs let the
            # v_accvgpr_write of a pack see the NEXT pack's value (measured: records differed)
V_RING, V_ACC = 16, 32                  # ring: v[16:31]; accumulators 0..7: v[32 + 16 i : +16]
V_EX, V_ED = 160, 192                   # gamma(x) [2 tiles][16], gamma(d) [2][8]
V_G = 208                               # g blocks 0, 1 of both tiles [2][16]  (views' first two blocks; the other two live in A0)
V_ZZ, V_ZN, V_DN, V_LW = 240, 242, 244, 246      # [2] each: per-tile compositing state of the CURRENT group
V_IN = 224                              # next group's inputs [2][8]: ox oy oz dx dy dz zz zn   (v[224:239])
V_TMP = 208                             # encoder / epilogue temporaries share the g area once g is dead: v[208:223]
V_PTMP = 248                            # 8 more temporaries v[248:255]
V_AUXA = V_PTMP + 6                     # [2 tiles]: the lane's address in the per-ray table between the fetch of a tile and its point (v[254:255])
V_LWA = V_IN + 2                        # address of the local-weight table reads in a group's last unit (the next group's o_z: dead since its point)
# ---- AGPR map: two activation arrays of [2 tiles][64]
A0, A1 = 0, 128
# ---- SGPR map
S_IMG, S_RAYS, S_Z, S_S, S_N, S_MAGIC, S_SHIFT, S_NGRP, S_NWG, S_REC, S_RECF, S_PS, S_NSEM, S_NINST, S_CLK = 4, 6, 8, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 23, 24
S_WAVE, S_W4K, S_GRP, S_PIECE, S_T0, S_T1 = 26, 27, 28, 30, 34, 35
S_VALID, S_LAST = 36, 40                # [2 tiles] x 64-bit masks of the CURRENT group: s[36:39], s[40:43]
S_NVALID, S_NLAST = 44, 48              # the same of the NEXT group (filled at fetch time): s[44:47], s[48:51]
S_LO32, S_N0, S_HI0 = 52, 54, 56        # constants: lanes 0..31, lanes {0, 32}, lanes with hi == 0 (= lanes 0..31)
S_SAVE, S_REC_T = 58, 60                # saved exec; the two tiles' record pointers s[60:61], s[62:63]
S_MSK = 64                              # store masks of the (up to) three logit blocks: s[64:69] = lanes with hi == 0 and channel < n_out
S_REC_I = 70                            # the two tiles' record pointers + 4 n_sem (the instance columns): s[70:73]
S_AUX = 82                              # s[82:83]: the per-ray table k_ray_aux wrote (gamma(d) of both half-waves and |d|: 128 B per ray)
S_SV, S_SVBASE, S_SVT = 84, 88, 86      # save prototype: running store address s[84:85], the region's base s[88:89], a temporary
V_SV = 15                               # save prototype: lane * 16 (V_TRACE's register: the trace builds have no save prototype)
S_VMSK = 76                             # softmax kernels: s[76:77] / s[78:79] = ALL lanes whose channel of the semantic / instance head's last block exists
S_LWW, S_LWR = 74, 75                   # LDS addresses of this wave's local-weight table: + 4 (lane & 31) to write, + (V_BIAS[0] = slot 0 + 16 hi) to read
LW_BASE = NSLOT * SLOT                  # [4 waves][2 tiles][32 floats] behind the weight slots (1 KiB)
S_CLK0 = 96                             # s[96:99] clocks at start
S_K = 100                               # s100: literal constants that VOP3 cannot carry
S_Q = 19                                # the tile's Q (v_readlane) between the scan and its store
S_GRP2 = 29                             # the group whose inputs are being prefetched


def f32(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1) if n > 1 else "v%d" % lo


def ar(lo, n):
    return "a[%d:%d]" % (lo, lo + n - 1) if n > 1 else "a%d" % lo


def sr(lo, n=2):
    return "s[%d:%d]" % (lo, lo + n - 1) if n > 1 else "s%d" % lo


class Loc:
    """where 4 consecutive packed-bf16 registers (one k-step of B operand) or one packed register lives"""
    def __init__(self, kind, base):
        self.kind, self.base = kind, base       # kind: 'a' | 'v'

    def reg(self, i, n=1):
        return (ar if self.kind == "a" else vr)(self.base + i, n)


class Sim:
    """in-order counters: every issued operation is appended; wait(tag) = "at most N outstanding" with N = the CERTAIN operations
    issued after `tag` (operations some waves skip do not count: conservative)."""
    def __init__(self, name, cap):
        self.name, self.cap, self.q, self.n = name, cap, [], 0

    def push(self, certain=True):
        self.n += 1
        self.q.append((self.n, certain))
        return self.n

    def wait_count(self, tag):
        if tag is None or not any(t == tag for t, _ in self.q):
            return None                         # already covered by an earlier wait
        idx = [t for t, _ in self.q].index(tag)
        n = sum(1 for _, c in self.q[idx + 1:] if c)
        n = min(n, self.cap)
        # everything up to tag is known complete once the wait passes -- but only if counts were exact; drop them anyway: a later
        # wait on an older tag returns None (covered)
        self.q = self.q[idx + 1:]
        return n

    def drain(self):
        self.q = []

    def state(self):
        return [c for _, c in self.q]


class Gen:
    def __init__(self, nbs, nbi, name, trace=False, abl=0, depth=2, softmax=False, tap=0):
        # abl (trace builds only; results invalid): 1 = no LDS-DMA pieces in the loop, 2 = no chunk hand-over (vmcnt + barrier),
        # 4 = no pack / ReLU of the hidden layers
        # depth: pnr_mlp_desc.head_depth -- 2: heads W -> W/2 -> n (sem0 | inst0 | sem1 | inst1); 1: one Linear W -> n per head, straight from
        # the trunk output h (round 6: SURVEY.md 9 item 4 as a kernel variant, k_mlp_tt_d1_s<n>i<m>)
        self.depth = depth
        # tap: pnr_mlp_desc.head_tap -- 0: the heads read the trunk output h; 1: the feature_linear output F (round 6: k_mlp_tt_f..).  F must
        # then outlive the views layer: the last g block goes to the gamma(d) registers instead of F[0:15] (g_block), and the head
        # hidden layers' outputs to h -- dead once the rgb / sigma unit is issued -- instead of F (shs_block, shi_block)
        self.tap = tap
        # softmax: semantic_activation = softmax (PNR_MLP_SOFTMAX) -- the learned fields composite softmax(logits) over their channels;
        # the tail normalises each head's transposed logit blocks per sample before the weighted sums (tail_softmax), operation for
        # operation fuse_softmax_t of the ping-pong kernel (pnr_mlp_fuse.h): k_mlp_tt_sm_s<n>i<m>
        self.softmax = softmax
        self.nbs, self.nbi, self.name, self.trace, self.abl = nbs, nbi, name, trace, abl or ABL_ALL
        self.nstamp = 0
        self.o = []
        self.nlabel = 0
        self.lgkm = Sim("lgkm", 15)
        self.vm = Sim("vm", 63)
        self.side, self.side_lo, self.outbox, self.capture = [], [], [], None
        self.ring_tags = [None] * P
        self.piece_tag = {}
        self.pending_pack = None
        self.g2_acc = self.park_acc = None
        self.m0 = None
        self.logit_units = []
        self.acc_free = list(range(8))
        self.lwr_acc = self.lwr_tag = None
        self.aux_tags = {}
        self.build_plan()
        self.build_units()
        self.stream = []
        for u in self.units:
            u["frag0"] = len(self.stream)
            self.stream += [(sl, of) for sl, of, _, _ in self.unit_frags(u)]

    # ------------------------------------------------------------------ plan (mirrors pnr_build_plan, plan 2)
    def build_plan(self):
        L = []

        def add(name, nblk, segs, fbc, mode):
            nks = sum(n for _, n in segs)
            if fbc * nks + 1 > 33:
                fbc = 1
            L.append(dict(name=name, nblk=nblk, segs=segs, nks=nks, fbc=fbc, mode=mode))
        for i in range(D):
            if i == 0:
                add("L0", 8, [("ex", 4)], 8, "relu")
            elif i - 1 == SKIP:
                add("L%d" % i, 8, [("ex", 4), ("h", 16)], 2, "relu")
            else:
                add("L%d" % i, 8, [("h", 16)], 2, "relu")
        add("feature", 8, [("h", 16)], 2, "linear")
        add("views", 4, [("F", 16), ("ed", 2)], 2, "relu")
        add("rgbs", 1, [("g", 8), ("hh", 16)], 1, "rgbs")
        # heads: sem0 | inst0 | sem1 | inst1 -- a logit layer never follows its own hidden layer directly (the hidden layer's last
        # blocks would have to be packed inside the logit unit's first MFMA gaps), sem1's reduction runs beside inst1
        if self.depth == 1:             # one Linear per head: the logit layers read h itself (16 k-steps)
            if self.nbs:
                add("sem1", self.nbs, [("tap", 16)], self.nbs, "logits")
            if self.nbi:
                add("inst1", 1, [("tap", 16)], 1, "logits")
        else:
            if self.nbs:
                add("sem0", 4, [("tap", 16)], 2, "relu")
            if self.nbi:
                add("inst0", 4, [("tap", 16)], 2, "relu")
            if self.nbs:
                add("sem1", self.nbs, [("shs", 8)], self.nbs, "logits")
            if self.nbi:
                add("inst1", 1, [("shi", 8)], 1, "logits")
        self.layers = L
        self.chunks = []
        off = 0
        for li, l in enumerate(L):
            for fb in range(0, l["nblk"], l["fbc"]):
                nfrag = l["fbc"] * l["nks"] + 1
                self.chunks.append(dict(layer=li, fb=fb, nfb=l["fbc"], off=off, nfrag=nfrag))
                off += nfrag
        self.total_frags = off
        # slot = chunk % 4 is static, so a group is a multiple of four chunks: geometries whose image has 42 (no heads) or 45 (no
        # instance head) chunks get DUMMY chunks at the group's end -- one piece of image fragment 0 into the slot, a hand-over,
        # no unit.  The image itself is unchanged (pnr_build_plan knows nothing of them)
        while len(self.chunks) % NSLOT:
            self.chunks.append(dict(layer=None, fb=0, nfb=0, off=0, nfrag=1, dummy=True))
        self.NC = len(self.chunks)
        assert self.NC % NSLOT == 0, self.NC
        assert max(c["nfrag"] for c in self.chunks) <= 33

    # ------------------------------------------------------------------ emission helpers
    # Side work is written as closures; when one is taken off the queue it runs in CAPTURE mode: plain instructions become strings
    # in the outbox, operations the counters must see (LDS reads, vector-memory operations, waits, accumulator releases) become
    # deferred calls that run when the outbox entry is actually emitted -- so the counters see every operation in PROGRAM order.
    class Holder:
        def __init__(self, v=None):
            self.v = v

    def e(self, s):
        if self.capture is not None:
            self.capture.append("\t" + s)
        else:
            self.o.append("\t" + s)

    def defer(self, fn):
        if self.capture is not None:
            self.capture.append(fn)
        else:
            fn()

    @contextmanager
    def atomic(self):
        """side-work instructions that must not be separated by main-stream instructions: a changed EXEC mask (the MFMA stream's
        LDS reads, packs and LDS-DMA pieces must run on all lanes) or a dependence on SCC (the pieces' s_add_u32 clobbers it)"""
        if self.capture is None:
            yield
            return
        outer, self.capture = self.capture, []
        try:
            yield
        finally:
            grp, self.capture = self.capture, outer
            outer.append(grp)

    def label(self):
        self.nlabel += 1
        return ".L%s_%d" % (self.name, self.nlabel)

    def lds_read(self, instr):
        h = Gen.Holder()

        def now():
            self.o.append("\t" + instr)
            h.v = self.lgkm.push()
        self.defer(now)
        return h

    def vm_op(self, instr, certain=True):
        h = Gen.Holder()

        def now():
            self.o.append("\t" + instr)
            h.v = self.vm.push(certain)
        self.defer(now)
        return h

    def wait_lgkm(self, h):
        def now():
            n = self.lgkm.wait_count(h.v if isinstance(h, Gen.Holder) else h)
            if n is not None:
                self.o.append("\ts_waitcnt lgkmcnt(%d)" % n)
        if h is not None:
            self.defer(now)

    def wait_vm(self, h):
        def now():
            n = self.vm.wait_count(h.v if isinstance(h, Gen.Holder) else h)
            if n is not None:
                self.o.append("\ts_waitcnt vmcnt(%d)" % n)
        if h is not None:
            self.defer(now)

    def lit(self, sreg, val):
        """s_mov of a 32-bit literal (VOP3 instructions cannot carry one on gfx9)"""
        self.e("s_mov_b32 s%d, 0x%x" % (sreg, val & 0xffffffff))

    def q(self, cost, fn, low=False):
        """queue side work; low: work that holds no accumulator (the encoders) -- it waits for everything else"""
        (self.side_lo if low else self.side).append(fn)

    def side_busy(self):
        return bool(self.side or self.side_lo or self.outbox)

    def drain_side(self, budget=None):
        """emit queued side work: all of it (budget None) or `budget` instructions"""
        spent = 0
        while self.side_busy() and (budget is None or spent < budget):
            if not self.outbox:
                fn = self.side.pop(0) if self.side else self.side_lo.pop(0)
                self.capture = []
                fn()
                self.outbox, self.capture = self.capture, None
                continue
            x = self.outbox.pop(0)
            for y in (x if isinstance(x, list) else [x]):
                if callable(y):
                    y()
                else:
                    self.o.append(y)
                spent += 1

    # ------------------------------------------------------------------ LDS-DMA
    def chunk_base(self, chunk):
        """s[S_PIECE : S_PIECE + 1] = address of image chunk `chunk` (once per chunk, in front of its first piece)"""
        self.e("s_add_u32 s%d, s%d, 0x%x" % (S_PIECE, S_IMG, self.chunks[chunk]["off"] * 1024))
        self.e("s_addc_u32 s%d, s%d, 0" % (S_PIECE + 1, S_IMG + 1))

    def piece_list(self, chunk):
        """(m, i, waves) of the LDS-DMA pieces of a chunk: fragment 16 m + 4 wave + i, issued by waves 0 .. waves - 1"""
        n = self.chunks[chunk]["nfrag"]
        out = []
        for m in range((n + 15) // 16):
            for i in range(4):
                nw = max(0, min(4, -(-(n - 16 * m - i) // 4)))
                if nw:
                    out.append((m, i, nw))
        return out

    def piece(self, chunk, j):
        """piece j of image chunk `chunk` -> its LDS slot: M0 and the load (the address is the chunk's base + this lane's constant
        offset register + an immediate; the first version's per-piece 64-bit SALU add cost a third of a piece)"""
        m, i, nw = self.piece_list(chunk)[j]
        slot = chunk % NSLOT
        skip = None
        certain = nw >= 4
        if not certain:
            skip = self.label()
            self.e("s_cmp_ge_u32 s%d, %d" % (S_WAVE, nw))
            self.e("s_cbranch_scc1 %s" % skip)
        # the instruction's immediate offset moves BOTH addresses (memory and LDS): M0 carries the run's base only -- and stays
        # what it is for the four pieces of a run (nothing else in the loop writes M0; trace builds do)
        m0 = slot_base(slot) + 16 * m * 1024
        if self.m0 != m0 or skip or self.trace:
            self.e("s_add_u32 m0, s%d, 0x%x" % (S_W4K, m0))
            self.e("s_nop 0")                   # SALU write of M0 -> LDS-DMA: one wait state
        self.m0 = None if skip else m0          # (a guarded piece: the waves that skipped it did not write M0)
        tag = self.vm_op("global_load_lds_dwordx4 v%d, s[%d:%d] offset:%d%s" % (V_DMA + m, S_PIECE, S_PIECE + 1, i * 1024, DMA_POLICY), certain)
        if skip:
            self.o.append(skip + ":")
        return tag

    def pieces_of(self, chunk):
        return len(self.piece_list(chunk))

    # ------------------------------------------------------------------ operand locations of the layer inputs
    def loc_of(self, seg, layer_index, t, ks):
        """Loc of the 4 registers of k-step `ks` of segment `seg` for tile t"""
        if seg == "ex":
            return Loc("v", V_EX + 16 * t + 4 * ks)
        if seg == "ed":
            return Loc("v", V_ED + 8 * t + 4 * ks)
        if seg == "h":                  # the trunk's ping-pong: L1 reads A0 ... (layer i reads what layer i - 1 wrote)
            src = self.trunk_out(layer_index - 1)
            return Loc("a", src + 64 * t + 4 * ks)
        if seg == "tap":                # what the heads read: h, or the feature (head_tap 1)
            return Loc("a", (self.F_base() if self.tap else self.trunk_out(D - 1)) + 64 * t + 4 * ks)
        if seg == "hh":                 # the trunk output h = what L7 wrote
            return Loc("a", self.trunk_out(D - 1) + 64 * t + 4 * ks)
        if seg == "F":
            return Loc("a", self.F_base() + 64 * t + 4 * ks)
        if seg == "g":                  # 4 blocks of 8 registers: blocks 0, 1 in VGPRs, 2, 3 in the F array (dead after views)
            blk, r = ks // 2, (ks % 2) * 4
            return self.g_block(blk, t, r)
        if seg == "shs":
            blk, r = ks // 2, (ks % 2) * 4
            return self.shs_block(blk, t, r)
        if seg == "shi":
            blk, r = ks // 2, (ks % 2) * 4
            return self.shi_block(blk, t, r)
        raise KeyError(seg)

    @staticmethod
    def trunk_out(i):
        """array that trunk layer i writes: L0 -> A0, L1 -> A1, ..."""
        return A0 if i % 2 == 0 else A1

    def F_base(self):
        return A0 if self.trunk_out(D - 1) == A1 else A1        # feature: h (A1) -> F (A0)

    def g_block(self, blk, t, r=0):
        # a block's registers may be written while LATER views units still read F, so only the LAST block may live in F
        if blk < 2:
            return Loc("v", V_G + 16 * t + 8 * blk + r)
        if blk == 2:
            return Loc("v", self.acc_reg(self.g2_acc) + 8 * t + r)                  # an idle accumulator (views / rgbs units take two)
        if self.tap:
            # head_tap 1: F is read by the heads.  gamma(d) of THIS group is dead once the views layer's MFMAs are issued (the pack runs
            # during the rgb / sigma unit) and the next group's is encoded behind that unit: exactly [2 tiles][8] registers
            return Loc("v", V_ED + 8 * t + r)
        return Loc("a", self.F_base() + 8 * t + r)                                  # F[0:15] (packed during the rgb / sigma unit)

    def shs_block(self, blk, t, r=0):
        if self.tap:
            return Loc("a", self.trunk_out(D - 1) + 32 * t + 8 * blk + r)           # h[0:63]: dead since the rgb / sigma unit
        return Loc("a", self.F_base() + 32 + 32 * t + 8 * blk + r)                  # F[32:95]

    def shi_block(self, blk, t, r=0):
        if self.tap:
            return Loc("a", self.trunk_out(D - 1) + 64 + 32 * t + 8 * blk + r)      # h[64:127]
        # the last inst0 unit is packed while sem1's MFMAs still read shs: its blocks go to h (dead once inst0's MFMAs are issued)
        if blk < 2:
            return Loc("a", self.F_base() + 96 + 16 * t + 8 * blk + r)              # F[96:127]
        return Loc("a", self.trunk_out(D - 1) + 16 * t + 8 * (blk - 2) + r)         # h[0:31]

    def out_block(self, layer, blk, t):
        """Loc of the 8 packed registers of output block `blk` of `layer` for tile t"""
        n = layer["name"]
        if n.startswith("L"):
            return Loc("a", self.trunk_out(int(n[1:])) + 64 * t + 8 * blk)
        if n == "feature":
            return Loc("a", self.F_base() + 64 * t + 8 * blk)
        if n == "views":
            return self.g_block(blk, t)
        if n == "sem0":
            return self.shs_block(blk, t)
        if n == "inst0":
            return self.shi_block(blk, t)
        raise KeyError(n)

    # ------------------------------------------------------------------ units
    def build_units(self):
        """the MFMA stream of one group: list of units, each = (chunk, layer, blocks, ...)"""
        U = []
        for ci, c in enumerate(self.chunks):
            if c.get("dummy"):
                continue
            l = self.layers[c["layer"]]
            step = 2 if l["mode"] != "logits" else min(c["nfb"], 2)        # (a third semantic block, nbs = 3: units [0, 1] and [2])
            if l["name"] == "rgbs":
                step = 1
            blocks = list(range(c["fb"], c["fb"] + c["nfb"]))
            first = True
            for i in range(0, len(blocks), step):
                U.append(dict(chunk=ci, layer=c["layer"], blocks=blocks[i:i + step], first_of_chunk=first,
                              last_of_chunk=(i + step >= len(blocks))))
                first = False
        self.units = U

    def unit_frags(self, u):
        """fragments of a unit in consumption order: (lds slot, byte offset in slot, ks, b index in unit)"""
        c = self.chunks[u["chunk"]]
        l = self.layers[u["layer"]]
        out = []
        for ks in range(l["nks"]):
            for bi, blk in enumerate(u["blocks"]):
                out.append((u["chunk"] % NSLOT, ((blk - c["fb"]) * l["nks"] + ks) * 1024, ks, bi))
        return out

    def seg_of_ks(self, layer, ks):
        for seg, n in layer["segs"]:
            if ks < n:
                return seg, ks
            ks -= n
        raise IndexError

    # ------------------------------------------------------------------ the pieces of side work
    # ---- arithmetic building blocks (all scalar fp32, one value per lane; temporaries named by register number)
    def div(self, res, num, den, t):
        """res = num / den, correctly rounded (the sequence hipcc emits for -fhip-fp32-correctly-rounded-divide-sqrt); t: 5 temps"""
        a, r, b, q_, x = t[:5]
        e = self.e
        e("v_div_scale_f32 v%d, s[%d:%d], v%d, v%d, v%d" % (a, S_T0 + 0, S_T0 + 1, den, den, num))
        e("v_rcp_f32 v%d, v%d" % (r, a))
        e("v_div_scale_f32 v%d, vcc, v%d, v%d, v%d" % (b, num, den, num))
        e("v_fma_f32 v%d, -v%d, v%d, 1.0" % (x, a, r))
        e("v_fmac_f32 v%d, v%d, v%d" % (r, x, r))
        e("v_mul_f32 v%d, v%d, v%d" % (q_, b, r))
        e("v_fma_f32 v%d, -v%d, v%d, v%d" % (x, a, q_, b))
        e("v_fmac_f32 v%d, v%d, v%d" % (q_, x, r))
        e("v_fma_f32 v%d, -v%d, v%d, v%d" % (a, a, q_, b))
        e("v_div_fmas_f32 v%d, v%d, v%d, v%d" % (a, a, r, q_))
        e("v_div_fixup_f32 v%d, v%d, v%d, v%d" % (res, a, den, num))

    def sqrt(self, res, x, t):
        """res = sqrtf(x), correctly rounded (hipcc's sequence); x is clobbered; t: 3 temps"""
        y, ym, tt = t[:3]
        e = self.e
        self.lit(S_K, 0x0f800000)
        e("v_mul_f32 v%d, 0x4f800000, v%d" % (tt, x))
        e("v_cmp_gt_f32 vcc, s%d, v%d" % (S_K, x))
        e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (x, x, tt))
        e("v_sqrt_f32 v%d, v%d" % (y, x))
        e("s_nop 0")
        e("v_add_u32 v%d, -1, v%d" % (ym, y))
        e("v_fma_f32 v%d, -v%d, v%d, v%d" % (tt, ym, y, x))
        e("v_cmp_ge_f32 s[%d:%d], 0, v%d" % (S_T0, S_T0 + 1, tt))
        e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (ym, y, ym, S_T0, S_T0 + 1))
        e("v_add_u32 v%d, 1, v%d" % (tt, y))
        e("v_fma_f32 v%d, -v%d, v%d, v%d" % (y, tt, y, x))
        e("v_cmp_lt_f32 s[%d:%d], 0, v%d" % (S_T0, S_T0 + 1, y))
        e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (ym, ym, tt, S_T0, S_T0 + 1))
        e("v_mul_f32 v%d, 0x37800000, v%d" % (tt, ym))
        e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (ym, ym, tt))
        e("v_mov_b32 v%d, 0x260" % tt)
        e("v_cmp_class_f32 vcc, v%d, v%d" % (x, tt))
        e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (res, ym, x))

    def sincos(self, s_out, c_out, x, t):
        """sincos_cw (pnr_mlp.hip): s_out, c_out = sin, cos of v[x]; t: 6 temps"""
        kf, r, r2, a, b, ki = t[:6]
        e = self.e
        C = dict(twoopi=f32(0.636619772367581343), c1=f32(1.57079637050628662109375), c2=f32(-4.371139000186241e-08),
                 s3=f32(-1.9515295891e-4), s2=f32(8.3321608736e-3), s1=f32(-1.6666654611e-1),
                 c4=f32(2.443315711809948e-5), c3=f32(-1.388731625493765e-3), c2b=f32(4.166664568298827e-2))
        assert C["twoopi"] == 0x3f22f983 and C["c1"] == 0x3fc90fdb and C["c2"] == 0xb33bbd2e and C["s3"] == 0xb94ca1f9
        assert C["s2"] == 0x3c08839e and C["s1"] == 0xbe2aaaa3 and C["c4"] == 0x37ccf5ce and C["c3"] == 0xbab6061a and C["c2b"] == 0x3d2aaaa5
        e("v_mul_f32 v%d, 0x%x, v%d" % (kf, C["twoopi"], x))
        e("v_rndne_f32 v%d, v%d" % (kf, kf))
        e("v_mov_b32 v%d, v%d" % (r, x))
        e("v_fmac_f32 v%d, 0x%x, v%d" % (r, C["c1"] ^ 0x80000000, kf))              # r = fma(kf, -C1, x) = fma(-kf, C1, x)
        e("v_fmac_f32 v%d, 0x%x, v%d" % (r, C["c2"] ^ 0x80000000, kf))              # r = fma(-kf, C2, r)
        e("v_cvt_i32_f32 v%d, v%d" % (ki, kf))
        e("v_mul_f32 v%d, v%d, v%d" % (r2, r, r))
        # sine polynomial
        e("v_mov_b32 v%d, 0x%x" % (a, C["s2"]))
        e("v_fmac_f32 v%d, 0x%x, v%d" % (a, C["s3"], r2))                           # sp = fma(r2, s3, s2)
        e("v_fmaak_f32 v%d, v%d, v%d, 0x%x" % (a, r2, a, C["s1"]))                  # sp = fma(r2, sp, s1)
        e("v_mul_f32 v%d, v%d, v%d" % (b, r, r2))                                   # r * r2
        e("v_fma_f32 v%d, v%d, v%d, v%d" % (a, b, a, r))                            # sn = fma(r * r2, sp, r)
        # cosine polynomial
        e("v_mov_b32 v%d, 0x%x" % (b, C["c3"]))
        e("v_fmac_f32 v%d, 0x%x, v%d" % (b, C["c4"], r2))                           # cp = fma(r2, c4, c3)
        e("v_fmaak_f32 v%d, v%d, v%d, 0x%x" % (b, r2, b, C["c2b"]))                 # cp = fma(r2, cp, c2)
        e("v_fma_f32 v%d, v%d, v%d, -0.5" % (b, r2, b))                             # cp = fma(r2, cp, -0.5)
        e("v_fma_f32 v%d, v%d, v%d, 1.0" % (b, r2, b))                              # cs = fma(r2, cp, 1)
        # quadrant
        e("v_and_b32 v%d, 1, v%d" % (r, ki))
        e("v_cmp_eq_u32 vcc, 0, v%d" % r)
        e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (r2, b, a))                          # so = (k & 1) ? cs : sn
        e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (kf, a, b))                          # co = (k & 1) ? sn : cs
        e("v_and_b32 v%d, 2, v%d" % (a, ki))
        e("v_lshlrev_b32 v%d, 30, v%d" % (a, a))
        e("v_xor_b32 v%d, v%d, v%d" % (s_out, r2, a))                               # s = so ^ ((k & 2) << 30)
        e("v_add_u32 v%d, 1, v%d" % (a, ki))
        e("v_and_b32 v%d, 2, v%d" % (a, a))
        e("v_lshlrev_b32 v%d, 30, v%d" % (a, a))
        e("v_xor_b32 v%d, v%d, v%d" % (c_out, kf, a))                               # c = co ^ (((k + 1) & 2) << 30)

    def band_pack(self, dst3, s, c):
        """three packed registers of one band: (s_x s_y) (s_z c_x) (c_y c_z)"""
        e = self.e
        e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst3, s[0], s[1]))
        e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst3 + 1, s[2], c[0]))
        e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst3 + 2, c[1], c[2]))

    def band_next(self, s, c, t):
        """double-angle step, in place: s2 = (2 s) c, c2 = fma(-2 s, s, 1)"""
        e = self.e
        for a in range(3):
            e("v_add_f32 v%d, v%d, v%d" % (t, s[a], s[a]))                          # 2 s (exact)
            e("v_mul_f32 v%d, v%d, v%d" % (t, t, c[a]))                             # s2 = (2 s) * c
            e("v_mul_f32 v%d, -2.0, v%d" % (c[a], s[a]))                            # -2 s
            e("v_fma_f32 v%d, v%d, v%d, 1.0" % (c[a], c[a], s[a]))                  # c2 = fma(-2 s, s, 1)
            e("v_mov_b32 v%d, v%d" % (s[a], t))

    # ---- next group's inputs: address, loads (tile t), masks
    def fetch_tile(self, t, grp_sreg):
        """issue the loads of tile t of group s[grp_sreg]: ox oy oz dx | dy dz | zz | zn -> V_IN + 8 t; masks -> S_NVALID / S_NLAST"""
        e = self.e
        vi = V_IN + 8 * t
        s_, sl, ray, a64 = V_PTMP, V_PTMP + 1, V_PTMP + 2, V_PTMP + 4       # a64: 2 registers
        # s = (grp * 8 + wave * 2 + t) * 32 + n
        e("s_lshl_b32 s%d, s%d, 3" % (S_T0, grp_sreg))
        e("s_lshl_b32 s%d, s%d, 1" % (S_T1, S_WAVE))
        e("s_add_u32 s%d, s%d, s%d" % (S_T0, S_T0, S_T1))
        e("s_add_u32 s%d, s%d, %d" % (S_T0, S_T0, t))
        e("s_lshl_b32 s%d, s%d, 5" % (S_T0, S_T0))
        e("v_and_b32 v%d, 31, v%d" % (s_, V_TID))
        e("v_or_b32 v%d, s%d, v%d" % (s_, S_T0, s_))
        e("v_cmp_gt_i32 s[%d:%d], s%d, v%d" % (S_NVALID + 2 * t, S_NVALID + 2 * t + 1, S_S, s_))        # valid: s < S
        e("s_add_i32 s%d, s%d, -1" % (S_T1, S_S))
        e("v_min_i32 v%d, s%d, v%d" % (sl, S_T1, s_))                                   # sl = min(s, S - 1)
        e("v_mul_hi_u32 v%d, v%d, s%d" % (ray, sl, S_MAGIC))                            # ray = (sl * magic) >> shift, shift >= 32 ...
        e("v_mul_lo_u32 v%d, v%d, s%d" % (a64, sl, S_MAGIC))
        e("v_mov_b32 v%d, v%d" % (a64 + 1, ray))
        e("v_lshrrev_b64 v[%d:%d], s%d, v[%d:%d]" % (a64, a64 + 1, S_SHIFT, a64, a64 + 1))
        e("v_mov_b32 v%d, v%d" % (ray, a64))
        # last = (sl - ray * N + 1 == N)
        e("v_mul_lo_u32 v%d, v%d, s%d" % (a64, ray, S_N))
        e("v_sub_u32 v%d, v%d, v%d" % (a64, sl, a64))
        e("v_add_u32 v%d, 1, v%d" % (a64, a64))
        e("v_cmp_eq_u32 s[%d:%d], s%d, v%d" % (S_NLAST + 2 * t, S_NLAST + 2 * t + 1, S_N, a64))
        # this lane's line of the per-ray table: ray * 128 + hi * 64 (kept until the point is formed: aux_x)
        e("v_lshlrev_b32 v%d, 7, v%d" % (V_AUXA + t, ray))
        e("v_and_b32 v%d, 32, v%d" % (a64, V_TID))
        e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (V_AUXA + t, a64, V_AUXA + t))
        # loads
        e("v_lshlrev_b32 v%d, 5, v%d" % (a64, ray))                                     # ray * 32 bytes (R * 32 < 2^32: R * N < 2^31, N >= 32)
        self.vm_op("global_load_dwordx4 %s, v%d, s[%d:%d]" % (vr(vi, 4), a64, S_RAYS, S_RAYS + 1))
        self.vm_op("global_load_dwordx2 %s, v%d, s[%d:%d] offset:16" % (vr(vi + 4, 2), a64, S_RAYS, S_RAYS + 1))
        e("v_lshlrev_b32 v%d, 2, v%d" % (a64, sl))
        self.vm_op("global_load_dword v%d, v%d, s[%d:%d]" % (vi + 6, a64, S_Z, S_Z + 1))
        e("v_add_u32 v%d, 1, v%d" % (a64 + 1, sl))
        e("v_cmp_gt_i32 vcc, s%d, v%d" % (S_S, a64 + 1))
        e("v_cndmask_b32 v%d, v%d, v%d, vcc" % (a64 + 1, sl, a64 + 1))                  # min(sl + 1, S - 1) as HEAD: (sl + 1 < S) ? sl + 1 : sl
        e("v_lshlrev_b32 v%d, 2, v%d" % (a64 + 1, a64 + 1))
        return self.vm_op("global_load_dword v%d, v%d, s[%d:%d]" % (vi + 7, a64 + 1, S_Z, S_Z + 1))

    # ---- gamma(x), gamma(d), |d| of tile t from V_IN (results: V_EX / V_ED, and zz zn dn of the NEXT group parked in V_IN + {6, 7, 3})
    def encode_tile(self, t, part):
        """closures (side-queue pieces) that encode tile t.  part "x": point, |d|, gamma(x), q = d / |d| (into V_IN + 3..5);
        part "d": gamma(d) from q"""
        vi = V_IN + 8 * t
        T = list(range(V_TMP, V_TMP + 16))          # 16 temporaries
        s = T[0:3]
        c = T[3:6]
        q = T[6:9]                                  # the 3-vector being encoded
        w = T[9:15]                                 # sincos temps (6)
        tt = T[15]
        e = self.e
        ex, ed = V_EX + 16 * t, V_ED + 8 * t
        out = []

        def hi_select(dst, v_hi1, v_hi0):           # dst = hi ? v_hi1 : v_hi0   (S_HI0: lanes with hi == 0)
            e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (dst, v_hi1, v_hi0, S_HI0, S_HI0 + 1))

        def points():
            # p = o + d * z (separate multiply and add)
            for a in range(3):
                e("v_mul_f32 v%d, v%d, v%d" % (q[a], vi + 3 + a, vi + 6))
                e("v_add_f32 v%d, v%d, v%d" % (q[a], vi + a, q[a]))
        if part == "x":
            out.append((6, points))

        def aux_x():
            # |d| = sqrt((dx dx + dy dy) + dz dz) and gamma(d / |d|) are PER-RAY values: k_ray_aux (pnr_mlp.hip) computed them once per
            # ray, with the operations this kernel used to spend on them per sample (sqrt, three divisions, three sincos, a band step:
            # 188 instructions per tile).  Here: |d| into the ray record's `ox` slot, the table address into the `dx` slot -- both are
            # parked across the views layer (PARK) --, gamma(d) is loaded behind the rgb / sigma unit (part "d")
            self.aux_tags[t] = self.vm_op("global_load_dword v%d, v%d, s[%d:%d] offset:32" % (vi + 0, V_AUXA + t, S_AUX, S_AUX + 1))
            e("v_mov_b32 v%d, v%d" % (vi + 3, V_AUXA + t))
        if part == "x":
            out.append((2, aux_x))

        def xyz_reg():
            # reg 0 of gamma(x): pack(hi ? pz : px, hi ? 0 : py)
            hi_select(w[0], q[2], q[0])
            e("v_mov_b32 v%d, 0" % w[1])
            hi_select(w[1], w[1], q[1])
            e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (ex, w[0], w[1]))
            # base band of the half-wave: 2^5 for hi = 1
            e("v_mov_b32 v%d, 0x42000000" % tt)
            e("v_mov_b32 v%d, 1.0" % w[1])
            hi_select(tt, tt, w[1])
            for a in range(3):
                e("v_mul_f32 v%d, v%d, v%d" % (q[a], q[a], tt))         # p * base
        if part == "x":
            out.append((12, xyz_reg))
            for a in range(3):
                out.append((30, (lambda a=a: self.sincos(s[a], c[a], q[a], w))))
            for fp in range(5):
                out.append((3, (lambda fp=fp: self.band_pack(ex + 1 + 3 * fp, s, c))))
                if fp < 4:
                    out.append((15, (lambda: self.band_next(s, c, tt))))

        if part == "x":
            out.append((1, (lambda: self.wait_vm(self.aux_tags[t]))))       # |d| has landed before anything (park) copies it
            return out

        def d_load():
            # this lane's eight packed registers of gamma(d): [x|y or z|0] [band 0 triple] [band 1 triple] [0], as embed_lane writes them
            self.vm_op("global_load_dwordx4 %s, v%d, s[%d:%d]" % (vr(ed, 4), vi + 3, S_AUX, S_AUX + 1))
            self.aux_tags[2 + t] = self.vm_op("global_load_dwordx4 %s, v%d, s[%d:%d] offset:16" % (vr(ed + 4, 4), vi + 3, S_AUX, S_AUX + 1))
        out.append((2, d_load))
        out.append((1, (lambda: self.wait_vm(self.aux_tags[2 + t]))))
        return out

    # ---- the compositing epilogue of the rgb / sigma block (fuse_rgbs) of tile t; acc = VGPR base of its accumulator
    def rgbs_epilogue(self, t, acc):
        e = self.e
        T = list(range(V_PTMP, V_PTMP + 8))
        d, x, y, n, al, f = T[0], T[1], T[2], T[3], T[4], T[5]
        VAL, LAST = S_VALID + 2 * t, S_LAST + 2 * t

        def alpha():
            # dist = last ? 1e10 : (zn - zz); x = (dist * -|d|) * relu(sigma)
            e("v_sub_f32 v%d, v%d, v%d" % (d, V_ZN + t, V_ZZ + t))
            e("v_mov_b32 v%d, 0x501502f9" % x)
            e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (d, d, x, LAST, LAST + 1))
            e("v_mul_f32 v%d, v%d, -v%d" % (d, d, V_DN + t))
            e("v_max_f32 v%d, v%d, v%d" % (x, acc + 3, acc + 3))
            e("v_max_f32 v%d, 0, v%d" % (x, x))
            e("v_mul_f32 v%d, v%d, v%d" % (x, d, x))
            # expf(x) (OCML's sequence as hipcc inlines it)
            self.lit(S_K, 0x3fb8aa3b)
            e("v_mul_f32 v%d, 0x3fb8aa3b, v%d" % (y, x))
            e("v_fma_f32 v%d, v%d, s%d, -v%d" % (d, x, S_K, y))
            e("v_rndne_f32 v%d, v%d" % (n, y))
            e("v_fmac_f32 v%d, 0x32a5705f, v%d" % (d, x))
            e("v_sub_f32 v%d, v%d, v%d" % (y, y, n))
            e("v_add_f32 v%d, v%d, v%d" % (y, y, d))
            e("v_exp_f32 v%d, v%d" % (y, y))
            e("v_cvt_i32_f32 v%d, v%d" % (n, n))
            e("s_nop 0")
            e("v_ldexp_f32 v%d, v%d, v%d" % (y, y, n))
            self.lit(S_K, 0xc2ce8ed0)
            e("v_cmp_ngt_f32 s[%d:%d], s%d, v%d" % (S_T0, S_T0 + 1, S_K, x))
            e("v_cndmask_b32 v%d, 0, v%d, s[%d:%d]" % (y, y, S_T0, S_T0 + 1))
            self.lit(S_K, 0x42b17218)
            e("v_cmp_nlt_f32 s[%d:%d], s%d, v%d" % (S_T0, S_T0 + 1, S_K, x))
            e("v_mov_b32 v%d, 0x7f800000" % n)
            e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (y, n, y, S_T0, S_T0 + 1))
            e("v_sub_f32 v%d, 1.0, v%d" % (al, y))                                   # alpha = 1 - exp
            e("v_cndmask_b32 v%d, 0, v%d, s[%d:%d]" % (al, al, VAL, VAL + 1))        # invalid lanes: 0
            # f = valid ? (1 - alpha) + 1e-10 : 1
            e("v_sub_f32 v%d, 1.0, v%d" % (f, al))
            e("v_add_f32 v%d, 0x2edbe6ff, v%d" % (f, f))
            e("v_mov_b32 v%d, 1.0" % n)
            e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (f, n, f, VAL, VAL + 1))
        self.q(36, alpha)

        def scan():
            # inclusive product over lanes 0..31 of each half (tile_scan): row_shr 1, 2, 4, 8, row_bcast15 into rows 1 and 3
            for ctrl in ("row_shr:1", "row_shr:2", "row_shr:4", "row_shr:8"):
                e("v_mov_b32 v%d, 1.0" % n)
                e("s_nop 1")
                e("v_mov_b32_dpp v%d, v%d %s row_mask:0xf bank_mask:0xf" % (n, f, ctrl))
                e("v_mul_f32 v%d, v%d, v%d" % (f, f, n))
            e("v_mov_b32 v%d, 1.0" % n)
            e("s_nop 1")
            e("v_mov_b32_dpp v%d, v%d row_bcast:15 row_mask:0xa bank_mask:0xf" % (n, f))
            e("v_mul_f32 v%d, v%d, v%d" % (f, f, n))
            e("v_mov_b32 v%d, 1.0" % n)
            e("s_nop 1")
            e("v_readlane_b32 s%d, v%d, 31" % (S_Q, f))                               # q = total of the tile
            e("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (n, f))
            e("v_cndmask_b32 v%d, v%d, 1.0, s[%d:%d]" % (n, n, S_N0, S_N0 + 1))        # lane n == 0: 1
            e("v_mul_f32 v%d, v%d, v%d" % (V_LW + t, al, n))                          # lw = alpha * exclusive product
            if self.nbs:
                # ... and into this wave's LDS table (lanes 0..31: sample n's weight at 4 n): the logit tail reads its per-register weights
                # lwr[r] = lw[row(r, hi)] back with ds_read_b64 (round 6; 64 v_readlane + 64 v_mov per group before: 864 cycles in the open)
                with self.atomic():
                    e("v_add_u32 v%d, s%d, v%d" % (n, S_LWW, V_LB4))
                    e("s_mov_b64 exec, s[%d:%d]" % (S_HI0, S_HI0 + 1))
                    self.lds_read("ds_write_b32 v%d, v%d offset:%d" % (n, V_LW + t, 128 * t))
                    e("s_mov_b64 exec, -1")
        self.q(26 + (4 if self.nbs else 0), scan)

        def stores():
          with self.atomic():
            # rec pointer of this tile: rec + (grp * 8 + wave * 2 + t) * rec_floats * 4
            e("s_lshl_b32 s%d, s%d, 3" % (S_T0, S_GRP))
            e("s_lshl_b32 s%d, s%d, 1" % (S_SAVE, S_WAVE))
            e("s_add_u32 s%d, s%d, s%d" % (S_T0, S_T0, S_SAVE))
            e("s_add_u32 s%d, s%d, %d" % (S_T0, S_T0, t))
            e("s_mul_i32 s%d, s%d, s%d" % (S_SAVE, S_T0, S_RECF))
            e("s_mul_hi_i32 s%d, s%d, s%d" % (S_SAVE + 1, S_T0, S_RECF))
            e("s_lshl_b64 s[%d:%d], s[%d:%d], 2" % (S_SAVE, S_SAVE + 1, S_SAVE, S_SAVE + 1))
            e("s_add_u32 s%d, s%d, s%d" % (S_REC_T + 2 * t, S_REC, S_SAVE))
            e("s_addc_u32 s%d, s%d, s%d" % (S_REC_T + 2 * t + 1, S_REC + 1, S_SAVE + 1))
            e("s_lshl_b32 s%d, s%d, 2" % (S_SAVE, S_NSEM))                            # ... and of its instance columns
            e("s_add_u32 s%d, s%d, s%d" % (S_REC_I + 2 * t, S_REC_T + 2 * t, S_SAVE))
            e("s_addc_u32 s%d, s%d, 0" % (S_REC_I + 2 * t + 1, S_REC_T + 2 * t + 1))
            # ps[samp] = (lw, r, g, b): lanes hi == 0 and valid; byte offset (s0 * 16) fits 32 bits (S < 2^27 ... checked by the launcher)
            e("s_lshl_b32 s%d, s%d, 5" % (S_T0, S_T0))
            e("v_and_b32 v%d, 31, v%d" % (d, V_TID))
            e("v_or_b32 v%d, s%d, v%d" % (d, S_T0, d))
            e("v_lshlrev_b32 v%d, 4, v%d" % (d, d))
            e("v_mov_b32 v%d, v%d" % (T[4], V_LW + t))
            e("v_mov_b32 v%d, v%d" % (T[5], acc + 0))
            e("v_mov_b32 v%d, v%d" % (T[6], acc + 1))
            e("v_mov_b32 v%d, v%d" % (T[7], acc + 2))
            e("s_and_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (S_SAVE, S_SAVE + 1, VAL, VAL + 1, S_HI0, S_HI0 + 1))
            e("s_mov_b64 exec, s[%d:%d]" % (S_SAVE, S_SAVE + 1))
            self.vm_op("global_store_dwordx4 v%d, %s, s[%d:%d]%s" % (d, vr(T[4], 4), S_PS, S_PS + 1, STORE_NT))
            e("s_mov_b64 exec, 1")                                                   # lane 0: rec[0] = q
            e("v_mov_b32 v%d, s%d" % (x, S_Q))
            e("v_mov_b32 v%d, 0" % d)
            self.vm_op("global_store_dword v%d, v%d, s[%d:%d]%s" % (d, x, S_REC_T + 2 * t, S_REC_T + 2 * t + 1, STORE_NT))
            e("s_mov_b64 exec, -1")
        self.q(28, stores)

    # ------------------------------------------------------------------ accumulators
    def acc_reg(self, i):
        return V_ACC + 16 * ACC_PERM[i]

    def acc_take(self, n):
        assert len(self.acc_free) >= n, ("accumulators exhausted", self.acc_free)
        got, self.acc_free = self.acc_free[:n], self.acc_free[n:]
        return got

    def acc_release(self, ids):
        self.acc_free = sorted(set(self.acc_free) | set(ids))

    # ------------------------------------------------------------------ arming: the bias of a unit into freshly taken accumulators
    def arm_plan(self, u):
        """per accumulator of unit u (block-major, tile-minor): a list of closures, one LDS read each, that put the bias into it;
        the first closure of an accumulator TAKES it (u['accs'] fills as the closures run)"""
        c = self.chunks[u["chunk"]]
        l = self.layers[u["layer"]]
        slot = u["chunk"] % NSLOT
        bias_off = c["nfb"] * l["nks"] * 1024
        u["accs"], u["arm_tags"] = {}, []
        plans = []
        for bi, blk in enumerate(u["blocks"]):
            b_in_chunk = blk - c["fb"]
            for t in range(2):
                key = (bi, t)
                ops = []

                def take(key=key):
                    while not self.acc_free and self.side_busy():       # a queued epilogue still holds accumulators: run it now
                        self.drain_side(8)
                    u["accs"][key] = self.acc_take(1)[0]
                if t == 1 and SHARE_BIAS:
                    # the two tiles of a block start from the SAME bias: tile 1's first MFMA takes tile 0's armed accumulator as its
                    # C operand (emit_unit issues it in front of tile 0's) -- its own accumulator is only taken, never armed: half the
                    # arming reads (408 of the 816 LDS reads per group), same values
                    plans.append([take])
                    continue
                if l["mode"] == "logits":       # transposed product: every register = bias of channel lane & 31
                    def first(key=key, slot=slot, take=take):           # the address lives in the accumulator's first register, read last
                        take()
                        self.e("v_add_u32 v%d, 0x%x, v%d" % (self.acc_reg(u["accs"][key]), slot_base(slot), V_LB4))
                    ops.append(first)
                    for r in list(range(1, 16)) + [0]:
                        ops.append(lambda key=key, r=r, off=bias_off + b_in_chunk * 128: u["arm_tags"].append(
                            self.lds_read("ds_read_b32 v%d, v%d offset:%d" % (self.acc_reg(u["accs"][key]) + r, self.acc_reg(u["accs"][key]), off))))
                else:
                    for m in range(4):
                        def rd(key=key, m=m, off=bias_off + b_in_chunk * 128 + m * 32, take=take):
                            if m == 0:
                                take()
                            a = self.acc_reg(u["accs"][key])
                            u["arm_tags"].append(self.lds_read("ds_read_b128 %s, v%d offset:%d" % (vr(a + 4 * m, 4), V_BIAS + slot, off)))
                        ops.append(rd)
                plans.append(ops)
        return plans

    # ------------------------------------------------------------------ pack / ReLU of a unit's accumulators -> list of (dst, closure)
    def pack_ops(self, u):
        l = self.layers[u["layer"]]
        if PACK_PIPE and not self.trace and not SAVE_PROTO:
            return self.pack_ops_pipelined(u)
        out = []
        for bi, blk in enumerate(u["blocks"]):
            for t in range(2):
                a = self.acc_reg(u["accs"][(bi, t)])
                dst = self.out_block(l, blk, t)
                for p in range(8):
                    def fn(a=a, p=p, dst=dst, relu=(l["mode"] == "relu"), k=len(out)):
                        tmp = V_T0 if (k & 1) else V_T1
                        if dst.kind == "v" and relu:
                            self.e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (tmp, a + 2 * p, a + 2 * p + 1))
                            self.e("v_pk_max_i16 v%d, v%d, 0" % (dst.base + p, tmp))
                        elif dst.kind == "v":
                            self.e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst.base + p, a + 2 * p, a + 2 * p + 1))
                        else:
                            self.e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (tmp, a + 2 * p, a + 2 * p + 1))
                            if relu:
                                self.e("v_pk_max_i16 v%d, v%d, 0" % (tmp, tmp))
                            self.e("v_accvgpr_write_b32 a%d, v%d" % (dst.base + p, tmp))
                    out.append(((dst.kind, dst.base + p), fn, u["accs"][(bi, t)]))
        return out

    def pack_ops_pipelined(self, u):
        """-> [([destinations final after this closure], closure, accumulator)]: see PACK_PIPE"""
        l = self.layers[u["layer"]]
        relu = l["mode"] == "relu"
        pairs = []
        for bi, blk in enumerate(u["blocks"]):
            for t in range(2):
                a = self.acc_reg(u["accs"][(bi, t)])
                dst = self.out_block(l, blk, t)
                for p in range(8):
                    pairs.append((a + 2 * p, dst.kind, dst.base + p, u["accs"][(bi, t)]))
        T = (V_T0, V_T1, V_T2)
        n = len(pairs)

        def convert(k):
            src, kind, d, _ = pairs[k]
            self.e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (d if (kind == "v" and not relu) else T[k % 3], src, src + 1))

        def activate(k):
            _, kind, d, _ = pairs[k]
            if relu:
                self.e("v_pk_max_i16 v%d, v%d, 0" % (d if kind == "v" else T[k % 3], T[k % 3]))

        def store(k):
            _, kind, d, _ = pairs[k]
            if kind == "a":
                self.e("v_accvgpr_write_b32 a%d, v%d" % (d, T[k % 3]))

        def final_at(k):
            kind = pairs[k][1]
            return k + 2 if kind == "a" else (k + 1 if relu else k)
        out = []
        for j in range(n + 2):
            def fn(j=j):
                if 0 <= j - 2 < n:
                    store(j - 2)
                if 0 <= j - 1 < n:
                    activate(j - 1)
                if j < n:
                    convert(j)
            done = [(pairs[k][1], pairs[k][2]) for k in range(max(0, j - 2), min(n, j + 1)) if final_at(k) == j]
            out.append((done, fn, pairs[min(j, n - 1)][3]))
        return out

    # ------------------------------------------------------------------ one unit of MFMAs with its fillers
    @staticmethod
    def tile_at(ks, t):
        """the tile of the t-th MFMA of a fragment: (0, 1), except at a block's first k-step with SHARE_BIAS: (1, 0)"""
        return 1 - t if (SHARE_BIAS and ks == 0) else t

    def b_operand(self, u, ks, t):
        l = self.layers[u["layer"]]
        seg, k = self.seg_of_ks(l, ks)
        li = u["layer"]
        return self.loc_of(seg, li, t, k)

    def emit_unit(self, ui, side_budget=3):
        U = self.units
        u = U[ui]
        l = self.layers[u["layer"]]
        c = self.chunks[u["chunk"]]
        frags = self.unit_frags(u)
        nm = 2 * len(frags)
        swap = l["mode"] == "logits"
        e = self.e
        e("; ==== unit %d: %s blocks %s (chunk %d, slot %d), %d MFMAs" % (ui, l["name"], u["blocks"], u["chunk"], u["chunk"] % NSLOT, nm))
        self.stamp(ui)
        if l["name"] == "L%d" % (SKIP + 2) and u["blocks"][0] == 0:
            self.queue_inputs_early()
        if l["name"] == "views" and u["blocks"] == [0]:
            self.drain_side()                       # (nothing left normally) the early side work uses the g area
        if l["name"] == "views" and u["blocks"] == [1]:
            self.park()
        if l["name"] == (("sem0" if self.nbs else "inst0") if self.depth == 2 else "sem1") and u["blocks"][0] == 0:
            self.unpark()
        if l["name"] == "views" and u["blocks"] == [2]:
            self.g2_acc = self.acc_take(1)[0]       # home of g block 2 (packed during the next unit) until the rgb / sigma unit is issued
        # ---- what has to happen inside this unit
        # (1) the previous unit's pack / ReLU, each before the first MFMA of THIS unit that reads its destination
        prev = self.pending_pack
        self.pending_pack = None
        packs = prev["ops"] if prev else []
        reads = {}
        for i in range(nm):
            f, t = i // 2, i % 2
            loc = self.b_operand(u, frags[f][2], self.tile_at(frags[f][2], t))
            for r in range(4):
                reads.setdefault((loc.kind, loc.base + r), i)
        # ... and, pipelined with it, the NEXT unit's bias: as soon as an accumulator is packed it is released and re-armed.  One list
        # of "musts" in program order, spread evenly over the unit's gaps (the first version packed in the first 60 % of the gaps
        # and armed in the last 25 %: up to seven fillers per gap where five hide, and nothing to do in between)
        nxt = U[ui + 1] if ui + 1 < len(U) else None
        plans = self.arm_plan(nxt) if nxt is not None else []
        musts, dls, pack_fns = [], [], set()
        by_acc = {}
        for dst, fn, a in packs:
            by_acc.setdefault(a, []).append((dst, fn))
        k = 0
        for a, lst in by_acc.items():
            for dst, fn in lst:
                musts.append(fn if not self.abl & 4 else (lambda: None))
                pack_fns.add(musts[-1])
                if isinstance(dst, list):           # pipelined packs: every destination that becomes final in this closure
                    dd = [reads[d] for d in dst if d in reads]
                    dls.append(min(dd) if dd else None)
                else:
                    dls.append(reads.get(dst))
            musts.append(lambda a=a: self.acc_release([a]))
            dls.append(None)
            if k < len(plans):
                for op in plans[k]:
                    musts.append(op)
                    dls.append(None)
                k += 1
        for kk in range(k, len(plans)):
            for op in plans[kk]:
                musts.append(op)
                dls.append(None)
        lwr_from = None
        if nxt is None and (self.logit_units or l["mode"] == "logits"):
            # the group's last unit: the logit tail's per-register weights lwr[r] = lw[row(r, hi)] of both tiles come back from the LDS
            # table (rgbs_epilogue: scan) into the two accumulators nobody holds any more -- behind the packs that release them
            lwr_from = len(musts)
            for op in self.lwr_ops():
                musts.append(op)
                dls.append(None)
        n = len(musts)
        pos = []
        late = not packs and n > 0 and nxt is not None  # nothing to pack (the previous unit's results go through the side queue and hold
        for j in range(n):                              # their accumulators until then): arm in the unit's last third
            spread = 1 + (j * max(1, nm - 6)) // max(1, n) if not late else (2 * nm) // 3 + (j * max(1, nm // 3 - 5)) // max(1, n)
            pos.append(spread if dls[j] is None else min(spread, max(0, dls[j] - 2)))
        if lwr_from is not None:                        # the local-weight reads: in the unit's second half (the side queue drains first)
            nl = n - lwr_from
            for j in range(lwr_from, n):
                pos[j] = max(pos[j], min(nm - 2, nm // 2 + ((j - lwr_from) * max(1, nm // 2 - 3)) // nl))
        for j in range(n - 2, -1, -1):                  # program order: a must never after a later must's position
            pos[j] = min(pos[j], pos[j + 1])
        must_at = {}
        for j, g in enumerate(pos):
            must_at.setdefault(g, []).append(j)
        # (2) LDS-DMA pieces of chunk + 3, spread over the chunk's units
        piece_at = {}
        cu = [k for k, x in enumerate(U) if x["chunk"] == u["chunk"]]
        c3 = (u["chunk"] + 3) % self.NC
        npieces = self.pieces_of(c3)
        k_in = cu.index(ui)
        mine = [j for j in range(npieces) if (j * len(cu)) // npieces == k_in]
        cost = [3 * sum(1 for j in must_at.get(g, []) if j < n and dls[j] is not None or (j < n and musts[j] in pack_fns)) +
                sum(1 for j in must_at.get(g, [])) for g in range(nm)]
        for idx, j in enumerate(mine):
            span = max(len(mine) + 1, int(PIECE_FRAC * (nm - 3)))
            lo = 1 + (idx * span) // max(1, len(mine))
            hi = max(lo + 1, 1 + ((idx + 1) * span) // max(1, len(mine)))
            g = min(range(lo, min(hi, nm - 1)), key=lambda x: (cost[x], x))
            cost[g] += 2
            piece_at.setdefault(g, []).append(j)
        # ---- the MFMAs
        first_wait_done = False
        for i in range(nm):
            f, t = i // 2, i % 2
            slot, off, ks, bi = frags[f]
            gi = u["frag0"] + f                                    # index in the group's fragment stream
            if t == 0:
                if not first_wait_done:
                    for tg in u["arm_tags"][-1:]:
                        self.wait_lgkm(tg)                          # the unit's bias has landed (in-order: the last read covers all)
                    first_wait_done = True
                if WAIT2 and f + 1 < len(frags):
                    if f % 2 == 0:                                  # one counted wait per TWO fragments: the younger of the pair covers both
                        self.wait_lgkm(self.ring_tags[(gi + 1) % P])
                else:
                    self.wait_lgkm(self.ring_tags[gi % P])
            tile = self.tile_at(ks, t)
            a = self.acc_reg(u["accs"][(bi, tile)])
            # a block's first k-step (SHARE_BIAS): tile 1 first, reading the bias out of tile 0's accumulator, then tile 0 in place
            c_in = self.acc_reg(u["accs"][(bi, 0)]) if (SHARE_BIAS and ks == 0) else a
            bloc = self.b_operand(u, ks, tile)
            ring = vr(V_RING + 4 * (gi % P), 4)
            if swap:
                e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(a, 16), bloc.reg(0, 4), ring, vr(c_in, 16)))
            else:
                e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(a, 16), ring, bloc.reg(0, 4), vr(c_in, 16)))
            # ---- fillers of gap i
            if t == 1:
                gn = gi + P - 1
                if gn < len(self.stream):
                    sl, of = self.stream[gn]
                    self.ring_tags[gn % P] = self.lds_read("ds_read_b128 %s, v%d offset:%d" % (vr(V_RING + 4 * (gn % P), 4), V_FRAG + sl, of))
            for j in must_at.get(i, []):
                musts[j]()
            for j in ([] if self.abl & 1 else piece_at.get(i, [])):
                if j == 0:
                    self.chunk_base(c3)
                tg = self.piece(c3, j)
                self.piece_tag[c3] = tg
            self.drain_side(1 if l["name"].startswith("L") or l["name"] in ("feature", "views") else 3 if late else 2)
        for g in sorted(must_at):                       # (positions beyond the last gap: none by construction)
            assert g < nm, (g, nm)
        if SAVE_PROTO and packs:
            self.queue_saves([dst for dst, _, _ in packs])
        # ---- this unit's own results: packed during the next unit, or reduced by the side queue
        if l["mode"] in ("relu", "linear"):
            self.pending_pack = dict(ops=self.pack_ops(u), accs=list(u["accs"].values()))
        elif l["mode"] == "rgbs":
            for t in range(2):
                self.rgbs_epilogue(t, self.acc_reg(u["accs"][(0, t)]))
            ids = list(u["accs"].values())
            self.q(0, lambda: self.defer(lambda: self.acc_release(ids)))
            self.queue_inputs_late()
            self.acc_release([self.g2_acc])         # g is dead: every MFMA of the rgb / sigma unit has been issued
            self.g2_acc = None
            if not self.nbs:                        # no head follows (its first unit would do this): the parked inputs go home
                self.unpark()
        else:
            # logit blocks keep their accumulators until the group's tail (tail_logits): all blocks of a tile are reduced TOGETHER there,
            # three independent FMA chains beside each other -- one block at a time is a chain of 16 dependent FMAs and two LDS
            # round trips with nothing to overlap them (the first version's tail: 2650 cycles)
            self.logit_units.append(u)
        # ---- chunk hand-over
        if u["last_of_chunk"]:
            c2 = (u["chunk"] + 2) % self.NC
            if not self.abl & 2:
                self.wait_vm(self.piece_tag.get(c2))
                e("s_barrier")
            # dummy chunks behind the group's last real chunk: their share of the weight stream (the pieces of chunk + 3) and their
            # hand-over, with whatever side work is queued between the pieces
            d = u["chunk"] + 1
            while d < self.NC and self.chunks[d].get("dummy"):
                e("; ==== dummy chunk %d (slot %d)" % (d, d % NSLOT))
                c3 = (d + 3) % self.NC
                for j in ([] if self.abl & 1 else range(self.pieces_of(c3))):
                    if j == 0:
                        self.chunk_base(c3)
                    self.piece_tag[c3] = self.piece(c3, j)
                    self.drain_side(2)
                if not self.abl & 2:
                    self.wait_vm(self.piece_tag.get((d + 2) % self.NC))
                    e("s_barrier")
                d += 1

    def tail_softmax(self, blocks, lwr_reg, tag_lwr):
        """semantic_activation = softmax: per head and tile, IN PLACE -- the head's logit accumulators become e = exp(x - max) and the
        tile's lwr registers become lwr / denominator, so that the FMA chains that follow (the logits tail, unchanged) form
        sum_r (lwr[r] / den[r]) e[r] = the record column of softmax(logits).  Lane = channel, register r = sample row(r, hi): maximum
        and denominator of a sample are reductions over the 32 lanes of each half-wave -- quad_perm / row_half_mirror / row_mirror
        DPP steps and v_permlane16_swap for the xor-16 step, all sixteen registers of both tiles per step back to back (independent).
        Operation for operation fuse_softmax_t (pnr_mlp_fuse.h): max_raw chain, xor_max 1 2 4 8 16, mm = -m log2e, e = exp2(fma(x, log2e,
        mm)), d = e_0 (+ e_1), xor_add 1 2 4 8, swap-add 16, lwr * rcp(d).  Channels past the head's last are set to -inf first (the
        ping-pong kernel starts their accumulators there): exp makes them 0.  The lwr registers are consumed head by head: the
        semantic head's FMA chains are issued before the instance head rescales them."""
        e = self.e
        M = (V_TMP, V_RING)                         # m / d of the tile: 16 registers each (g area; the fragment ring is idle at a group's end)
        TMP = [V_IN + 10, V_IN + 11, V_IN + 12, V_IN + 13]  # temporaries of the xor-16 steps: the next group's o_z and q of tile 1 (dead: its
                                                            # gamma(d) is encoded) -- NOT the logits tail's sums / oth registers (V_IN + 1..5, 9)
        ninf = V_PTMP + 6
        heads = []
        for inst in (False, True):
            hb = [(u, bi, blk) for (u, bi, blk, i2) in blocks if i2 == inst]
            if hb:
                heads.append((inst, hb))
        for t in (0, 1):
            self.wait_lgkm(tag_lwr[t])
        e("v_mov_b32 v%d, 0xff800000" % ninf)

        def xor16(op):
            # x op x[lane ^ 16] through v_permlane16_swap on two copies (one ends up with rows {0, 0, 2, 2}, the other {1, 1, 3, 3}); four
            # registers at a time: a VALU write needs two wait states before v_permlane16_swap reads it, and so does the swap's result
            regs = [M[t] + r for t in (0, 1) for r in range(16)]
            for g in range(0, 32, 4):
                for i in range(4):
                    e("v_mov_b32 v%d, v%d" % (TMP[i], regs[g + i]))
                for i in range(4):
                    e("v_permlane16_swap_b32 v%d, v%d" % (regs[g + i], TMP[i]))
                for i in range(4):
                    e("%s v%d, v%d, v%d" % (op, regs[g + i], regs[g + i], TMP[i]))
        for inst, hb in heads:
            accs = [[self.acc_reg(u["accs"][(bi, t)]) for (u, bi, blk) in hb] for t in (0, 1)]      # [tile][block] -> first register
            NB = len(hb)
            vm = S_VMSK + 2 * (1 if inst else 0)
            for t in (0, 1):                        # the head's LAST block: channels >= n_out -> -inf
                for r in range(16):
                    e("v_cndmask_b32 v%d, v%d, v%d, s[%d:%d]" % (accs[t][NB - 1] + r, ninf, accs[t][NB - 1] + r, vm, vm + 1))
            # m[r] = max over the head's blocks (max_raw chain), then the 32-lane butterfly
            first_src = None
            if NB > 1:
                for t in (0, 1):
                    for r in range(16):
                        e("v_max_f32 v%d, v%d, v%d" % (M[t] + r, accs[t][0] + r, accs[t][1] + r))
                        for b in range(2, NB):
                            e("v_max_f32 v%d, v%d, v%d" % (M[t] + r, M[t] + r, accs[t][b] + r))
            else:
                first_src = [accs[t][0] for t in (0, 1)]
            for k, ctrl in enumerate(("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror")):
                for t in (0, 1):
                    for r in range(16):
                        src = (first_src[t] + r) if (k == 0 and first_src) else (M[t] + r)
                        e("v_max_f32_dpp v%d, v%d, v%d %s row_mask:0xf bank_mask:0xf bound_ctrl:1" % (M[t] + r, src, src, ctrl))
            xor16("v_max_f32")
            # mm = -m log2(e);  e = exp2(fma(x, log2 e, mm));  d = e_0 + e_1 ...
            for t in (0, 1):
                for r in range(16):
                    e("v_mul_f32 v%d, 0xbfb8aa3b, v%d" % (M[t] + r, M[t] + r))
            for t in (0, 1):
                for r in range(16):
                    for b in range(NB):
                        e("v_fmamk_f32 v%d, v%d, 0x3fb8aa3b, v%d" % (accs[t][b] + r, accs[t][b] + r, M[t] + r))
                        e("v_exp_f32 v%d, v%d" % (accs[t][b] + r, accs[t][b] + r))
            first_src = None
            if NB > 1:
                for t in (0, 1):
                    for r in range(16):
                        e("v_add_f32 v%d, v%d, v%d" % (M[t] + r, accs[t][0] + r, accs[t][1] + r))
                        for b in range(2, NB):
                            e("v_add_f32 v%d, v%d, v%d" % (M[t] + r, M[t] + r, accs[t][b] + r))
            else:
                first_src = [accs[t][0] for t in (0, 1)]
            for k, ctrl in enumerate(("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror")):
                for t in (0, 1):
                    for r in range(16):
                        src = (first_src[t] + r) if (k == 0 and first_src) else (M[t] + r)
                        e("v_add_f32_dpp v%d, v%d, v%d %s row_mask:0xf bank_mask:0xf bound_ctrl:1" % (M[t] + r, src, src, ctrl))
            xor16("v_add_f32")
            # wr[r] = lwr[r] * rcp(d[r]): kept in M (the lwr registers stay what they are for the next head)
            for t in (0, 1):
                for r in range(16):
                    e("v_rcp_f32 v%d, v%d" % (M[t] + r, M[t] + r))
            for t in (0, 1):
                for r in range(16):
                    e("v_mul_f32 v%d, v%d, v%d" % (M[t] + r, lwr_reg(t, r), M[t] + r))
            # this head's FMA chains against wr (fmas() is told to skip them): s = sum_r wr[r] e[r]
            for r in range(16):
                for t in (0, 1):
                    for (u, bi, blk) in hb:
                        kk = [i for i, (u2, bi2, blk2, i2) in enumerate(blocks) if u2 is u and bi2 == bi][0]
                        a = self.acc_reg(u["accs"][(bi, t)])
                        dst = self.sm_sums[t][kk]
                        if r == 0:
                            e("v_fma_f32 v%d, v%d, v%d, 0" % (dst, M[t], a))
                        else:
                            e("v_fmac_f32 v%d, v%d, v%d" % (dst, M[t] + r, a + r))

    def lwr_ops(self):
        """closures (one instruction each) that load lwr of both tiles: 8 ds_read_b64 per tile.  lw[8 q + 4 hi + j], j = 0..3, is
        register 4 q + ((j + 2) & 3) of the tile's block -- two positions round its quad, so that the tail's FMA (lwr[r] x register r
        of a 16-aligned accumulator) reads its two operands from different VGPR banks (register number mod 4)"""
        ops = []

        def first():
            # the table is written by the rgb / sigma unit's epilogue (scan), which is side work: everything of it that is still queued
            # goes out NOW, in front of the reads (head_depth 1 without an instance head: the last unit follows the rgb / sigma unit
            # directly -- the reads returned the previous group's weights), and with it the accumulators that epilogue holds
            while self.side or self.outbox or (len(self.acc_free) < 2 and self.side_busy()):
                self.drain_side(8)
            assert len(self.acc_free) >= 2, ("no free accumulators for the local weights", self.acc_free)
            self.lwr_acc = self.acc_take(2)
            self.e("v_add_u32 v%d, s%d, v%d" % (V_LWA, S_LWR, V_BIAS + 0))
        ops.append(first)
        for t in range(2):
            for q in range(4):
                for h2 in range(2):
                    def rd(t=t, q=q, h2=h2):
                        dst = self.acc_reg(self.lwr_acc[t]) + 4 * q + ((2 * h2 + 2) & 3)
                        self.lwr_tag = self.lds_read("ds_read_b64 %s, v%d offset:%d" % (vr(dst, 2), V_LWA, 128 * t + 32 * q + 8 * h2))
                    ops.append(rd)
        return ops

    def logit_blocks(self):
        return [(b, False) for b in range(self.nbs)] + [(0, True)] * self.nbi

    def tail_logits(self):
        """the group's logit blocks -> record columns (fuse_logits_t).  Both tiles pipelined: lwr of tile 0 and tile 1 (the fragment
        ring is idle at a group's end: its registers hold tile 1's), tile 0's FMAs -- every block's chain side by side --, its
        exchanges between the half-waves, tile 1's FMAs beside them, then the stores.  Masks and the instance columns' pointers are
        precomputed (S_MSK, S_REC_I)."""
        e = self.e
        if not self.logit_units:
            return
        blocks = []
        for u in self.logit_units:
            inst = self.layers[u["layer"]]["name"] == "inst1"
            for bi, blk in enumerate(u["blocks"]):
                blocks.append((u, bi, blk, inst))
        nb = len(blocks)
        lwr = tuple(self.acc_reg(a) for a in self.lwr_acc)

        def lwr_reg(t, r):
            return lwr[t] + 4 * (r >> 2) + (((r & 3) + 2) & 3)            # see lwr_ops
        sums = ([V_PTMP + k for k in range(nb)], [V_IN + 1 + k for k in range(nb)])
        oth = ([V_PTMP + 3 + k for k in range(nb)], [V_IN + 4, V_IN + 5, V_IN + 9][:nb])
        adr = V_PTMP + 7
        tag_lwr = [self.lwr_tag, self.lwr_tag]          # (in order: the last read covers all sixteen)
        e("v_xor_b32 v%d, 32, v%d" % (adr, V_TID))
        e("v_and_b32 v%d, 63, v%d" % (adr, adr))
        e("v_lshlrev_b32 v%d, 2, v%d" % (adr, adr))
        tags = [None, None]

        self.sm_sums = sums

        def fmas(ts):
            # the chains of the tiles in `ts` side by side: a chain is 16 DEPENDENT FMAs, and three of them interleaved still ran at
            # ~10 cycles per instruction (one wave per SIMD: nobody else fills the result latency); six run at the issue rate
            for t in ts:
                self.wait_lgkm(tag_lwr[t])
            for r in ([] if self.softmax else range(16)):
                for t in ts:
                    for k, (u, bi, blk, inst) in enumerate(blocks):
                        a = self.acc_reg(u["accs"][(bi, t)])
                        if r == 0:
                            e("v_fma_f32 v%d, v%d, v%d, 0" % (sums[t][k], lwr_reg(t, 0), a))
                        else:
                            e("v_fmac_f32 v%d, v%d, v%d" % (sums[t][k], lwr_reg(t, r), a + r))
            for t in ts:
                tags[t] = [self.lds_read("ds_bpermute_b32 v%d, v%d, v%d" % (oth[t][k], adr, sums[t][k])) for k in range(nb)]

        def stores(t):
            for k in range(nb):                     # the sums under the FULL exec mask, then the masked stores
                self.wait_lgkm(tags[t][k])
                e("v_add_f32 v%d, v%d, v%d" % (oth[t][k], sums[t][k], oth[t][k]))
            for k, (u, bi, blk, inst) in enumerate(blocks):
                e("s_mov_b64 exec, s[%d:%d]" % (S_MSK + 2 * k, S_MSK + 2 * k + 1))
                base = (S_REC_I if inst else S_REC_T) + 2 * t
                self.vm_op("global_store_dword v%d, v%d, s[%d:%d] offset:%d%s" % (V_LB4, oth[t][k], base, base + 1, 4 + 128 * blk, STORE_NT))
            e("s_mov_b64 exec, -1")
        self.stamp(len(self.units) + 4)
        if self.softmax:
            self.tail_softmax(blocks, lwr_reg, tag_lwr)
        fmas((0, 1))
        self.stamp(len(self.units) + 5)
        stores(0)
        stores(1)
        for u in self.logit_units:
            self.acc_release(list(u["accs"].values()))
        self.acc_release(self.lwr_acc)
        self.lwr_acc = None
        self.logit_units = []

    def queue_inputs_early(self):
        """side work from the first unit behind the skip layer on (gamma(x) of this group is dead): the next group's inputs, its
        point, gamma(x), and |d| from the per-ray table.  Temporaries and inputs live in the g area (free until views' first pack)."""
        def which():
            # g2 = grp + n_wg < n_groups ? grp + n_wg : grp
            with self.atomic():
                self.e("s_add_u32 s%d, s%d, s%d" % (S_GRP2, S_GRP, S_NWG))
                self.e("s_cmp_lt_i32 s%d, s%d" % (S_GRP2, S_NGRP))
                self.e("s_cselect_b32 s%d, s%d, s%d" % (S_GRP2, S_GRP2, S_GRP))
        self.q(3, which)
        tags = {}
        for t in range(2):
            self.q(34, (lambda t=t: tags.__setitem__(t, self.fetch_tile(t, S_GRP2))))
        for t in range(2):
            self.q(1, (lambda t=t: self.wait_vm(tags[t])))      # (deferred: the holder is filled when the load is emitted)
            for cost, fn in self.encode_tile(t, "x"):
                self.q(cost, fn)

    PARK = (3, 6, 7, 0)             # what of V_IN outlives gamma(x): the per-ray table address, z, z_next, |d|

    def park(self):
        """views' packs of tile 1 land on V_IN: its six live values per tile wait in an idle accumulator until the rgb / sigma unit"""
        self.park_acc = self.acc_take(1)[0]
        for t in range(2):
            for i, k in enumerate(Gen.PARK):
                self.e("v_mov_b32 v%d, v%d" % (self.acc_reg(self.park_acc) + len(Gen.PARK) * t + i, V_IN + 8 * t + k))

    def unpark(self):
        for t in range(2):
            for i, k in enumerate(Gen.PARK):
                self.e("v_mov_b32 v%d, v%d" % (V_IN + 8 * t + k, self.acc_reg(self.park_acc) + len(Gen.PARK) * t + i))
        self.acc_release([self.park_acc])
        self.park_acc = None

    def queue_inputs_late(self):
        """behind the rgb / sigma unit (gamma(d) of this group is dead): gamma(d) of the next group -- since the per-ray table two
        16-byte loads per tile; before it ~250 instructions of division, sincos and packing per tile.  (Round 6 tried the skip layer's
        one-block units of the SAME group instead -- the emptiest gaps of the loop: bit-identical, and no faster: the skip layer's units
        grew by what the head units shrank, profiles/r06/r06f.  What the kernel pays for is the NUMBER of non-MFMA instructions, not
        where they stand.)"""
        for t in range(2):
            for cost, fn in self.encode_tile(t, "d"):
                self.q(cost, fn, low=True)

    # ------------------------------------------------------------------ trace builds (k_mlp_tt_*_trace): s_memtime per unit
    def stamp(self, idx):
        """lane idx of v13 := low word of s_memtime here (written one stamp later: the scalar load has returned by then); the
        clock output of a trace build is the 64 stamps of workgroup 0's first wave in its LAST group"""
        if not self.trace:
            return
        pair = S_CLK0 + 2 * (self.nstamp & 1)
        prev = S_CLK0 + 2 * ((self.nstamp + 1) & 1)
        if self.last_stamp is not None:
            self.o.append("\tv_writelane_b32 v%d, s%d, %d" % (V_TRACE, prev, self.last_stamp))
        self.o.append("\ts_memtime s[%d:%d]" % (pair, pair + 1))
        self.last_stamp = idx
        self.nstamp += 1
        if idx == len(self.units) + 3:
            # the group's end also goes to lane 56 + (group counter & 7): the durations of the last eight groups
            skip = self.label()
            self.o += ["\ts_waitcnt lgkmcnt(0)", "\ts_and_b32 s%d, s101, 7" % S_T0, "\ts_add_u32 s%d, s%d, 56" % (S_T0, S_T0),
                       "\ts_mov_b32 m0, s%d" % S_T0, "\ts_nop 1",
                       "\tv_writelane_b32 v%d, s%d, m0" % (V_TRACE, pair),
                       # every 32nd group's end (groups 0, 32, .., 160) to lanes 48..53: the average group time per window
                       "\ts_and_b32 s%d, s101, 31" % S_T0, "\ts_cmp_eq_u32 s%d, 0" % S_T0, "\ts_cbranch_scc0 %s" % skip,
                       "\ts_lshr_b32 s%d, s101, 5" % S_T0, "\ts_add_u32 s%d, s%d, 48" % (S_T0, S_T0), "\ts_mov_b32 m0, s%d" % S_T0,
                       "\ts_nop 1", "\tv_writelane_b32 v%d, s%d, m0" % (V_TRACE, pair), skip + ":",
                       "\ts_add_u32 s101, s101, 1"]

    # ------------------------------------------------------------------ group boundary
    def prefetch_first_unit(self):
        """bias of unit 0 into fresh accumulators + the ring's first three fragments (the only LGKM operations in flight at a
        group's start)"""
        u0 = self.units[0]
        for plan in self.arm_plan(u0):
            for op in plan:
                op()
        for g in range(P - 1):
            sl, of = self.stream[g]
            self.ring_tags[g % P] = self.lds_read("ds_read_b128 %s, v%d offset:%d" % (vr(V_RING + 4 * (g % P), 4), V_FRAG + sl, of))

    def advance_group_state(self):
        """the prefetched group becomes the current one: z, z_next, |d| and the masks"""
        e = self.e
        for t in range(2):
            vi = V_IN + 8 * t
            e("v_mov_b32 v%d, v%d" % (V_ZZ + t, vi + 6))
            e("v_mov_b32 v%d, v%d" % (V_ZN + t, vi + 7))
            e("v_mov_b32 v%d, v%d" % (V_DN + t, vi + 0))
            e("s_mov_b64 s[%d:%d], s[%d:%d]" % (S_VALID + 2 * t, S_VALID + 2 * t + 1, S_NVALID + 2 * t, S_NVALID + 2 * t + 1))
            e("s_mov_b64 s[%d:%d], s[%d:%d]" % (S_LAST + 2 * t, S_LAST + 2 * t + 1, S_NLAST + 2 * t, S_NLAST + 2 * t + 1))

    def save_base(self):
        """s[S_SV : S_SV + 1] = region base + (4 group + wave) * SAVE_REGION"""
        e = self.e
        e("s_lshl_b32 s%d, s%d, 2" % (S_SVT, S_GRP))
        e("s_add_u32 s%d, s%d, s%d" % (S_SVT, S_SVT, S_WAVE))
        self.lit(S_K, SAVE_REGION)
        e("s_mul_hi_u32 s%d, s%d, s%d" % (S_SV + 1, S_SVT, S_K))
        e("s_mul_i32 s%d, s%d, s%d" % (S_SV, S_SVT, S_K))
        e("s_add_u32 s%d, s%d, s%d" % (S_SV, S_SV, S_SVBASE))
        e("s_addc_u32 s%d, s%d, s%d" % (S_SV + 1, S_SV + 1, S_SVBASE + 1))
        self.save_n = 0

    def queue_saves(self, regs):
        """side work: 16-byte-per-lane stores of the packed registers `regs` [(kind, index)] -- quads of consecutive registers"""
        regs = sorted(regs)
        assert len(regs) % 4 == 0
        for i in range(0, len(regs), 4):
            k, r = regs[i]
            assert [x for x in regs[i:i + 4]] == [(k, r + j) for j in range(4)], regs[i:i + 4]

            def fn(k=k, r=r):
                self.vm_op("global_store_dwordx4 v%d, %s, s[%d:%d] offset:%d" % (V_SV, (ar if k == "a" else vr)(r, 4), S_SV, S_SV + 1,
                                                                                   1024 * (self.save_n % 4)))
                self.save_n += 1
                if self.save_n % 4 == 0:
                    with self.atomic():         # (the carry: no other scalar instruction between the two)
                        self.e("s_add_u32 s%d, s%d, 0x1000" % (S_SV, S_SV))
                        self.e("s_addc_u32 s%d, s%d, 0" % (S_SV + 1, S_SV + 1))
            self.q(1, fn)

    def group_body(self):
        self.m0 = None
        if SAVE_PROTO:
            self.save_base()
        assert self.acc_free == [i for i in range(8) if i not in self.units[0]["accs"].values()], self.acc_free
        self.last_stamp = None
        for ui in range(len(self.units)):
            self.emit_unit(ui)
        self.stamp(len(self.units))
        self.drain_side()                               # whatever the last units could not cover
        self.tail_logits()
        self.stamp(len(self.units) + 1)
        assert self.acc_free == list(range(8)), self.acc_free
        self.e("s_waitcnt lgkmcnt(0)")
        self.lgkm.drain()
        self.advance_group_state()
        self.stamp(len(self.units) + 2)
        if self.trace:
            self.o.append("\ts_waitcnt lgkmcnt(0)")
            self.stamp(len(self.units) + 3)

    # ------------------------------------------------------------------ the kernel
    def kernel(self):
        e, name = self.e, self.name
        self.o += ["\t.text", "\t.globl\t%s" % name, "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
        for dst, off, n in ((S_IMG, 0x0, 2), (S_RAYS, 0x8, 2), (S_Z, 0x10, 2), (S_S, 0x18, 2), (S_MAGIC, 0x20, 2), (S_NGRP, 0x28, 2),
                            (S_REC, 0x30, 2), (S_RECF, 0x38, 1), (S_PS, 0x40, 2), (S_NSEM, 0x48, 2), (S_CLK, 0x50, 2), (S_AUX, 0x58, 2)) + (((S_SVBASE, 0x60, 2),) if SAVE_PROTO else ()):
            e("s_load_dword%s %s, s[0:1], 0x%x" % ("x2" if n == 2 else "", sr(dst, n), off))
        # wave id from v0 itself (never written): a v_readfirstlane of a register that is re-used a few instructions later was
        # observed to return the LATER value while scalar-load data was returning (tools/probe/gen_two_tile_asm.py)
        e("v_readfirstlane_b32 s%d, v0" % S_WAVE)
        e("s_nop 4")
        e("s_lshr_b32 s%d, s%d, 6" % (S_WAVE, S_WAVE))
        e("s_lshl_b32 s%d, s%d, 12" % (S_W4K, S_WAVE))                    # wave * 4 KiB
        e("v_and_b32 v%d, 63, v0" % V_T0)
        e("v_lshlrev_b32 v%d, 4, v%d" % (V_LANE16, V_T0))
        e("v_lshrrev_b32 v%d, 5, v0" % V_LB4)
        e("v_and_b32 v%d, 1, v%d" % (V_LB4, V_LB4))
        e("v_lshlrev_b32 v%d, 4, v%d" % (V_LB4, V_LB4))                  # hi * 16
        for sl in range(NSLOT):
            self.lit(S_T0, slot_base(sl))
            e("v_add_u32 v%d, s%d, v%d" % (V_FRAG + sl, S_T0, V_LANE16))
            e("v_add_u32 v%d, s%d, v%d" % (V_BIAS + sl, S_T0, V_LB4))
        for k in range(3):
            e("v_add_u32 v%d, s%d, v%d" % (V_DMA + k, S_W4K, V_LANE16))
            if k:
                e("v_add_u32 v%d, 0x%x, v%d" % (V_DMA + k, 16384 * k, V_DMA + k))
        e("v_and_b32 v%d, 31, v0" % V_LB4)
        e("v_lshlrev_b32 v%d, 2, v%d" % (V_LB4, V_LB4))
        if SAVE_PROTO:
            assert not self.trace
            e("v_and_b32 v%d, 63, v0" % V_SV)
            e("v_lshlrev_b32 v%d, 4, v%d" % (V_SV, V_SV))
        e("s_mov_b32 s%d, -1" % S_LO32)
        e("s_mov_b32 s%d, 0" % (S_LO32 + 1))
        e("s_mov_b32 s%d, 1" % S_N0)
        e("s_mov_b32 s%d, 1" % (S_N0 + 1))
        e("s_mov_b32 s%d, -1" % S_HI0)
        e("s_mov_b32 s%d, 0" % (S_HI0 + 1))
        e("s_lshr_b32 s%d, s%d, 4" % (S_LWW, S_W4K))                      # wave * 256: this wave's [2 tiles][32] local-weight table
        e("s_add_u32 s%d, s%d, 0x%x" % (S_LWW, S_LWW, LW_BASE))
        e("s_sub_u32 s%d, s%d, 0x%x" % (S_LWR, S_LWW, slot_base(0)))      # + V_BIAS[0] (= slot 0's base + 16 hi) = table + 16 hi
        e("s_waitcnt lgkmcnt(0)")
        # store masks of the logit blocks (constant): channel = 32 blk + (lane & 31) < n_out, lanes 0..31 only
        for k, (blk, inst) in enumerate(self.logit_blocks()):
            e("v_and_b32 v%d, 31, v0" % V_T0)
            if blk:
                e("v_add_u32 v%d, %d, v%d" % (V_T0, 32 * blk, V_T0))
            e("v_cmp_gt_i32 vcc, s%d, v%d" % (S_NINST if inst else S_NSEM, V_T0))
            e("s_and_b64 s[%d:%d], vcc, s[%d:%d]" % (S_MSK + 2 * k, S_MSK + 2 * k + 1, S_HI0, S_HI0 + 1))
        if self.softmax:
            # ALL lanes (both halves) whose channel of the head's last block exists: 32 (blocks - 1) + (lane & 31) < n_out
            for k, (nblk, sreg) in enumerate(((self.nbs, S_NSEM), (self.nbi, S_NINST))):
                if nblk:
                    e("v_and_b32 v%d, 31, v0" % V_T0)
                    if nblk > 1:
                        e("v_add_u32 v%d, %d, v%d" % (V_T0, 32 * (nblk - 1), V_T0))
                    e("v_cmp_gt_i32 s[%d:%d], s%d, v%d" % (S_VMSK + 2 * k, S_VMSK + 2 * k + 1, sreg, V_T0))
        lend, lloop, lfin = self.label(), self.label(), self.label()
        e("s_mov_b32 s%d, s2" % S_GRP)
        e("s_cmp_ge_i32 s%d, s%d" % (S_GRP, S_NGRP))
        e("s_cbranch_scc1 %s" % lend)
        # chunks 0, 1, 2 -> slots 0, 1, 2
        for c in range(3):
            self.chunk_base(c)
            for j in range(self.pieces_of(c)):
                self.piece_tag[c] = self.piece(c, j)
        e("s_waitcnt vmcnt(0)")
        self.vm.drain()
        e("s_barrier")
        # the first group's inputs and encodings (exposed, once per workgroup)
        e("s_mov_b32 s%d, s%d" % (S_GRP2, S_GRP))
        tags = [self.fetch_tile(t, S_GRP2) for t in range(2)]
        for t in range(2):
            self.wait_vm(tags[t])
            for part in ("x", "d"):
                for _, fn in self.encode_tile(t, part):
                    fn()
        self.advance_group_state()
        if not self.trace:
            e("s_memtime s[%d:%d]" % (S_CLK0, S_CLK0 + 1))
            e("s_memrealtime s[%d:%d]" % (S_CLK0 + 2, S_CLK0 + 3))
        else:
            e("v_mov_b32 v%d, 0" % V_TRACE)
            e("s_mov_b32 s101, 0")
            e("s_memtime s[90:91]")                                  # the workgroup's start
            e("s_waitcnt lgkmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        self.lgkm.drain()
        self.prefetch_first_unit()
        # pass 1 (discarded): brings the counters to their steady state at the loop's head
        keep = self.o
        self.o = []
        self.group_body()
        self.prefetch_first_unit()
        self.o = keep
        # pass 2: the loop body
        self.o.append(lloop + ":")
        self.group_body()
        e("s_add_u32 s%d, s%d, s%d" % (S_GRP, S_GRP, S_NWG))
        e("s_cmp_ge_i32 s%d, s%d" % (S_GRP, S_NGRP))
        e("s_cbranch_scc1 %s" % lend)
        self.prefetch_first_unit()
        e("s_branch %s" % lloop)
        self.o.append(lend + ":")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_cmp_eq_u64 s[%d:%d], 0" % (S_CLK, S_CLK + 1))
        e("s_cbranch_scc1 %s" % lfin)
        if self.trace:              # every workgroup: cycles of its first wave from start to end -> clk[64 + workgroup]
            lw = self.label()
            e("s_cmp_lg_u32 s%d, 0" % S_WAVE)
            e("s_cbranch_scc1 %s" % lw)
            e("s_memtime s[%d:%d]" % (S_CLK0, S_CLK0 + 1))
            e("s_waitcnt lgkmcnt(0)")
            e("s_sub_u32 s%d, s%d, s90" % (S_CLK0, S_CLK0))
            e("s_lshl_b32 s%d, s2, 2" % S_T0)
            e("s_add_u32 s%d, s%d, 256" % (S_T0, S_T0))
            e("v_mov_b32 v%d, s%d" % (V_T0, S_T0))
            e("v_mov_b32 v%d, s%d" % (V_T1, S_CLK0))
            e("s_mov_b64 exec, 1")
            e("global_store_dword v%d, v%d, s[%d:%d]" % (V_T0, V_T1, S_CLK, S_CLK + 1))
            e("s_mov_b64 exec, -1")
            e("s_waitcnt vmcnt(0)")
            self.o.append(lw + ":")
        e("s_cmp_lg_u32 s2, 0")
        e("s_cbranch_scc1 %s" % lfin)
        if self.trace:
            e("s_cmp_lg_u32 s%d, 0" % S_WAVE)
            e("s_cbranch_scc1 %s" % lfin)
            e("v_and_b32 v%d, 63, v0" % V_T0)
            e("v_lshlrev_b32 v%d, 2, v%d" % (V_T0, V_T0))
            e("global_store_dword v%d, v%d, s[%d:%d]" % (V_T0, V_TRACE, S_CLK, S_CLK + 1))
            e("s_waitcnt vmcnt(0)")
            e("s_branch %s" % lfin)
        e("s_memtime s[%d:%d]" % (S_T0 - 0, S_T0 + 1))
        e("s_memrealtime s[%d:%d]" % (S_SAVE, S_SAVE + 1))
        e("s_waitcnt lgkmcnt(0)")
        e("s_sub_u32 s%d, s%d, s%d" % (S_T0, S_T0, S_CLK0))
        e("s_subb_u32 s%d, s%d, s%d" % (S_T1, S_T1, S_CLK0 + 1))
        e("s_sub_u32 s%d, s%d, s%d" % (S_SAVE, S_SAVE, S_CLK0 + 2))
        e("s_subb_u32 s%d, s%d, s%d" % (S_SAVE + 1, S_SAVE + 1, S_CLK0 + 3))
        e("v_cmp_eq_u32 vcc, 0, v0")
        e("s_and_saveexec_b64 s[%d:%d], vcc" % (S_VALID, S_VALID + 1))
        e("s_cbranch_execz %s" % lfin)
        e("v_mov_b32 v20, s%d" % S_T0)
        e("v_mov_b32 v21, s%d" % S_T1)
        e("v_mov_b32 v22, s%d" % S_SAVE)
        e("v_mov_b32 v23, s%d" % (S_SAVE + 1))
        e("v_mov_b32 v%d, 0" % V_T0)
        e("global_store_dwordx4 v%d, v[20:23], s[%d:%d]" % (V_T0, S_CLK, S_CLK + 1))
        e("s_waitcnt vmcnt(0)")
        self.o.append(lfin + ":")
        e("s_endpgm")
        self.o += [".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name), ""]
        self.o += ["\t.rodata", "\t.p2align\t6", "\t.amdhsa_kernel %s" % name,
                   "\t\t.amdhsa_group_segment_fixed_size %d" % (NSLOT * SLOT + 1024),
                   "\t\t.amdhsa_private_segment_fixed_size 0", "\t\t.amdhsa_kernarg_size %d" % KERNARG_BYTES, "\t\t.amdhsa_user_sgpr_count 2",
                   "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1", "\t\t.amdhsa_system_sgpr_workgroup_id_x 1",
                   "\t\t.amdhsa_system_vgpr_workitem_id 0", "\t\t.amdhsa_next_free_vgpr 512", "\t\t.amdhsa_next_free_sgpr 102",
                   "\t\t.amdhsa_accum_offset 256", "\t\t.amdhsa_reserve_vcc 1", "\t\t.amdhsa_float_denorm_mode_32 3",
                   "\t\t.amdhsa_float_denorm_mode_16_64 3", "\t\t.amdhsa_dx10_clamp 1", "\t\t.amdhsa_ieee_mode 1",
                   "\t.end_amdhsa_kernel", ""]
        return "\n".join(self.o)


KERNARG_BYTES = 104


def metadata(names):
    o = ["\t.amdgpu_metadata", "---", "amdhsa.kernels:"]
    for n in names:
        o += ["  - .agpr_count:     256", "    .args:", "      - .offset:         0", "        .size:           %d" % KERNARG_BYTES,
              "        .value_kind:     by_value", "    .group_segment_fixed_size: %d" % (NSLOT * SLOT + 1024),
              "    .kernarg_segment_align: 8", "    .kernarg_segment_size: %d" % KERNARG_BYTES, "    .max_flat_workgroup_size: 256",
              "    .name:           %s" % n, "    .private_segment_fixed_size: 0", "    .sgpr_count:     108",
              "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % n, "    .uniform_work_group_size: 1",
              "    .uses_dynamic_stack: false", "    .vgpr_count:     512", "    .vgpr_spill_count: 0", "    .wavefront_size: 64"]
    o += ["amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", "\t.end_amdgpu_metadata", ""]
    return "\n".join(o)


def variant_name(tap, depth, sm, nbs, nbi):
    tag = ("f" if tap else "") + ("d1" if depth == 1 else "") + ("sm" if sm else "")
    return "k_mlp_tt_%ss%di%d" % (tag + "_" if tag else "", nbs, nbi)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    parts = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 5", ""]
    names = []
    for nbs, nbi in ((1, 1), (2, 1), (0, 0), (1, 0), (2, 0), (3, 0)):
        n = "k_mlp_tt_s%di%d" % (nbs, nbi)
        names.append(n)
        parts.append(Gen(nbs, nbi, n).kernel())
    # the variants of the networks WITH heads: head_tap (f) x head_depth 1 (d1) x softmax compositing (sm): k_mlp_tt[_f][_d1][sm]... as
    # variant_name() spells them (pnr_mlp_tt.cpp builds the same names)
    for tap in (0, 1):
        for depth in (2, 1):
            for sm in (False, True):
                if (tap, depth, sm) == (0, 2, False):
                    continue
                for nbs, nbi in ((1, 1), (2, 1), (1, 0), (2, 0), (3, 0)):
                    n = variant_name(tap, depth, sm, nbs, nbi)
                    names.append(n)
                    parts.append(Gen(nbs, nbi, n, depth=depth, softmax=sm, tap=tap).kernel())
    # diagnostics builds only (make EXTRA_TT=trace | abl): the production library carries no kernel that writes (64 + n_wg) * 4 bytes
    # to the clock buffer (a 16-byte clk_probe buffer under PNR_MLP_TRACE was an out-of-bounds device write)
    extra = sys.argv[3] if len(sys.argv) > 3 else ""
    if extra in ("trace", "abl"):
        names.append("k_mlp_tt_s2i1_trace")     # debug: per-unit s_memtime stamps instead of the clock pair (tools/tt_trace.py)
        parts.append(Gen(2, 1, "k_mlp_tt_s2i1_trace", trace=True).kernel())
    if extra == "abl":                          # timing-only ablations of the trace build (results invalid)
        for abl in (1, 2, 3, 4, 7):
            names.append("k_mlp_tt_s2i1_trace_a%d" % abl)
            parts.append(Gen(2, 1, names[-1], trace=True, abl=abl).kernel())
    unit_names = ["%s %s" % (g.layers[u["layer"]]["name"], u["blocks"]) for g in [Gen(2, 1, "x")] for u in g.units]
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("\n".join(unit_names) + "\n")
    parts.append(metadata(names))
    with open(out, "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
