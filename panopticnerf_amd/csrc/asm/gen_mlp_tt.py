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

Geometry (fixed): D = 8, W = 256, skip = 4, xyz_L = 10, dir_L = 4, head_W = 128, head_tap 0; NBS = 1 | 2 semantic and NBI = 0 | 1
instance logit blocks -> four kernels k_mlp_tt_s<NBS>i<NBI>.

Time structure per 256-sample group (one workgroup; tile = (wave, t), record index grp * 8 + wave * 2 + t):
  image chunk c lives in LDS slot c % 4 (33 KiB each); during chunk c every wave issues its LDS-DMA pieces of chunk c + 3; at
  the end of chunk c: s_waitcnt vmcnt (own pieces of chunk c + 2 landed) + ONE s_barrier, so chunk c + 2 is readable during chunk
  c + 1.  MFMAs are grouped in UNITS (<= 2 output blocks x all k-steps x 2 tiles, one accumulator per (block, tile)); a unit's
  accumulators are packed / reduced while the NEXT unit's MFMAs run, the unit after that finds its bias already in its
  accumulators.  Weight fragments pass through a ring of 4 register quads, 3 fragments ahead, counted lgkmcnt.
  Side work (input fetch, gamma(x) / gamma(d) of the NEXT group, the compositing epilogue) is a queue of instructions drained a few
  per MFMA gap."""
import struct
import sys
from contextlib import contextmanager

NSLOT, SLOT = 4, 33 * 1024
P = 4                                   # fragment ring (quads)
D, SKIP = 8, 4

# ---- VGPR map
V_TID, V_LANE16, V_FRAG, V_BIAS, V_T0, V_T1, V_T2, V_T3, V_LB4, V_ZERO = 0, 1, 2, 6, 10, 11, 12, 13, 14, 15
V_RING, V_ACC = 16, 32                  # ring: v[16:31]; accumulators 0..7: v[32 + 16 i : +16]
V_EX, V_ED = 160, 192                   # gamma(x) [2 tiles][16], gamma(d) [2][8]
V_G = 208                               # g blocks 0, 1 of both tiles [2][16]  (views' first two blocks; the other two live in A0)
V_ZZ, V_ZN, V_DN, V_LW = 240, 242, 244, 246      # [2] each: per-tile compositing state of the CURRENT group
V_IN = 224                              # next group's inputs [2][8]: ox oy oz dx dy dz zz zn   (v[224:239])
V_TMP = 208                             # encoder / epilogue temporaries share the g area once g is dead: v[208:223]
V_PTMP = 248                            # 8 more temporaries v[248:255]
# ---- AGPR map: two activation arrays of [2 tiles][64]
A0, A1 = 0, 128
# ---- SGPR map
S_IMG, S_RAYS, S_Z, S_S, S_N, S_MAGIC, S_SHIFT, S_NGRP, S_NWG, S_REC, S_RECF, S_PS, S_NSEM, S_NINST, S_CLK = 4, 6, 8, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 23, 24
S_WAVE, S_W1K, S_GRP, S_PIECE, S_IMGW, S_T0, S_T1 = 26, 27, 28, 30, 32, 34, 35
S_VALID, S_LAST = 36, 40                # [2 tiles] x 64-bit masks of the CURRENT group: s[36:39], s[40:43]
S_NVALID, S_NLAST = 44, 48              # the same of the NEXT group (filled at fetch time): s[44:47], s[48:51]
S_LO32, S_N0, S_HI0 = 52, 54, 56        # constants: lanes 0..31, lanes {0, 32}, lanes with hi == 0 (= lanes 0..31)
S_SAVE, S_REC_T = 58, 60                # saved exec; the two tiles' record pointers s[60:61], s[62:63]
S_W = 64                                # 32 local weights of a tile (v_readlane): s[64:95]
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
    def __init__(self, nbs, nbi, name):
        self.nbs, self.nbi, self.name = nbs, nbi, name
        self.o = []
        self.nlabel = 0
        self.lgkm = Sim("lgkm", 15)
        self.vm = Sim("vm", 63)
        self.side, self.outbox, self.capture = [], [], None
        self.ring_tags = [None] * P
        self.piece_tag = {}
        self.pending_pack = None
        self.acc_free = list(range(8))
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
        add("sem0", 4, [("hh", 16)], 2, "relu")
        add("sem1", self.nbs, [("shs", 8)], self.nbs, "logits")
        if self.nbi:
            add("inst0", 4, [("hh", 16)], 2, "relu")
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

    def q(self, cost, fn):
        self.side.append(fn)

    def side_busy(self):
        return bool(self.side or self.outbox)

    def drain_side(self, budget=None):
        """emit queued side work: all of it (budget None) or `budget` instructions"""
        spent = 0
        while self.side_busy() and (budget is None or spent < budget):
            if not self.outbox:
                fn = self.side.pop(0)
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
    def piece(self, chunk, j, guard_nfrag=None):
        """fragment wave + 4 j of image chunk `chunk` -> its LDS slot"""
        c = self.chunks[chunk]
        slot = chunk % NSLOT
        skip = None
        certain = True
        if c["nfrag"] - 4 * j < 4:              # only waves < nfrag - 4 j hold such a fragment
            skip = self.label()
            certain = False
            self.e("s_cmp_ge_u32 s%d, %d" % (S_WAVE, c["nfrag"] - 4 * j))
            self.e("s_cbranch_scc1 %s" % skip)
        self.e("s_add_u32 m0, s%d, 0x%x" % (S_W1K, slot * SLOT + j * 4096))
        self.e("s_add_u32 s%d, s%d, 0x%x" % (S_PIECE, S_IMGW, (c["off"] + 4 * j) * 1024))
        self.e("s_addc_u32 s%d, s%d, 0" % (S_PIECE + 1, S_IMGW + 1))
        tag = self.vm_op("global_load_lds_dwordx4 v%d, s[%d:%d] nt" % (V_LANE16, S_PIECE, S_PIECE + 1), certain)
        if skip:
            self.o.append(skip + ":")
        return tag

    def pieces_of(self, chunk):
        n = self.chunks[chunk]["nfrag"]
        return (n + 3) // 4

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
            return Loc("v", V_EX + 16 * t + r)                                      # gamma(x) is dead after the skip layer
        return Loc("a", self.F_base() + 8 * t + r)                                  # F[0:15] (packed during the rgb / sigma unit)

    def shs_block(self, blk, t, r=0):
        return Loc("a", self.F_base() + 32 + 32 * t + 8 * blk + r)                  # F[32:95]

    def shi_block(self, blk, t, r=0):
        # inst0 runs after sem1: sem0's outputs are dead, but the LAST inst0 unit is packed while inst1's MFMAs read the first
        if blk < 2:
            return Loc("a", self.F_base() + 96 + 16 * t + 8 * blk + r)              # F[96:127]
        return Loc("a", self.F_base() + 32 + 16 * t + 8 * (blk - 2) + r)            # F[32:63] (shs is dead)

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
            l = self.layers[c["layer"]]
            step = 2 if l["mode"] != "logits" else c["nfb"]
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
    def encode_tile(self, t, chunks):
        """list of closures (side-queue pieces) that encode tile t"""
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
            # p = o + d * z (separate multiply and add), |d| = sqrt((dx dx + dy dy) + dz dz)
            for a in range(3):
                e("v_mul_f32 v%d, v%d, v%d" % (q[a], vi + 3 + a, vi + 6))
                e("v_add_f32 v%d, v%d, v%d" % (q[a], vi + a, q[a]))
            e("v_mul_f32 v%d, v%d, v%d" % (w[0], vi + 3, vi + 3))
            e("v_mul_f32 v%d, v%d, v%d" % (w[1], vi + 4, vi + 4))
            e("v_add_f32 v%d, v%d, v%d" % (w[0], w[0], w[1]))
            e("v_mul_f32 v%d, v%d, v%d" % (w[1], vi + 5, vi + 5))
            e("v_add_f32 v%d, v%d, v%d" % (w[0], w[0], w[1]))
        out.append((11, points))

        def norm():
            self.sqrt(w[5], w[0], [w[1], w[2], w[3]])
            # park |d| in the ray record's unused `ox` slot?  no: ox is still needed for nothing after points() -- q holds the point
            e("v_mov_b32 v%d, v%d" % (vi + 0, w[5]))        # V_IN + 0 := |d| of the next group
        out.append((22, norm))

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
        out.append((12, xyz_reg))
        for a in range(3):
            out.append((30, (lambda a=a: self.sincos(s[a], c[a], q[a], w))))
        for fp in range(5):
            out.append((3, (lambda fp=fp: self.band_pack(ex + 1 + 3 * fp, s, c))))
            if fp < 4:
                out.append((15, (lambda: self.band_next(s, c, tt))))

        # gamma(d): q = d / |d|
        def dirs():
            for a in range(3):
                self.div(q[a], vi + 3 + a, vi + 0, [w[0], w[1], w[2], w[3], w[4]])
                e("s_nop 0")
        out.append((36, dirs))

        def d_reg():
            hi_select(w[0], q[2], q[0])
            e("v_mov_b32 v%d, 0" % w[1])
            hi_select(w[1], w[1], q[1])
            e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (ed, w[0], w[1]))
            e("v_mov_b32 v%d, 4.0" % tt)
            e("v_mov_b32 v%d, 1.0" % w[1])
            hi_select(tt, tt, w[1])
            for a in range(3):
                e("v_mul_f32 v%d, v%d, v%d" % (q[a], q[a], tt))
        out.append((12, d_reg))
        for a in range(3):
            out.append((30, (lambda a=a: self.sincos(s[a], c[a], q[a], w))))
        out.append((3, (lambda: self.band_pack(ed + 1, s, c))))
        out.append((15, (lambda: self.band_next(s, c, tt))))
        out.append((3, (lambda: self.band_pack(ed + 4, s, c))))
        out.append((1, (lambda: e("v_mov_b32 v%d, 0" % (ed + 7)))))
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
        self.q(26, scan)

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
            self.vm_op("global_store_dwordx4 v%d, %s, s[%d:%d]" % (d, vr(T[4], 4), S_PS, S_PS + 1))
            e("s_mov_b64 exec, 1")                                                   # lane 0: rec[0] = q
            e("v_mov_b32 v%d, s%d" % (x, S_Q))
            self.vm_op("global_store_dword v%d, v%d, s[%d:%d]" % (V_ZERO, x, S_REC_T + 2 * t, S_REC_T + 2 * t + 1))
            e("s_mov_b64 exec, -1")
        self.q(28, stores)

    # ---- a transposed 32-channel logit block of tile t (fuse_logits_t): rec[rec_base + ch] = sum_r lwr[r] acc[r] (+ other half)
    def lwr_load(self, t):
        """the tile's 32 local weights -> s[S_W : S_W + 32) (lane order)"""
        e = self.e
        e("s_nop 1")
        for lane in range(32):
            e("v_readlane_b32 s%d, v%d, %d" % (S_W + lane, V_LW + t, lane))

    def logits_epilogue(self, t, acc, blk, inst):
        e = self.e
        T = list(range(V_TMP, V_TMP + 16))
        s_, o, adr = V_PTMP, V_PTMP + 1, V_PTMP + 2

        def body():
            # lwr[r] (lwr_build): v[V_TMP + r] = hi ? lw[row(r, 1)] : lw[row(r, 0)]
            e("v_fma_f32 v%d, v%d, v%d, 0" % (s_, T[0], acc))
            for r in range(1, 16):
                e("v_fmac_f32 v%d, v%d, v%d" % (s_, T[r], acc + r))
            # s += s of the other half-wave (lane ^ 32)
            e("v_xor_b32 v%d, 32, v%d" % (adr, V_TID))
            e("v_and_b32 v%d, 63, v%d" % (adr, adr))
            e("v_lshlrev_b32 v%d, 2, v%d" % (adr, adr))
            tag = self.lds_read("ds_bpermute_b32 v%d, v%d, v%d" % (o, adr, s_))
            # channel ch = blk * 32 + (lane & 31); store for hi == 0 and ch < n_out
            e("v_and_b32 v%d, 31, v%d" % (adr, V_TID))
            e("v_add_u32 v%d, %d, v%d" % (adr, blk * 32, adr))
            e("v_cmp_gt_i32 vcc, s%d, v%d" % (S_NINST if inst else S_NSEM, adr))
            e("s_and_b64 s[%d:%d], vcc, s[%d:%d]" % (S_SAVE, S_SAVE + 1, S_HI0, S_HI0 + 1))
            e("v_lshlrev_b32 v%d, 2, v%d" % (adr, adr))
            self.wait_lgkm(tag)
            e("v_add_f32 v%d, v%d, v%d" % (o, s_, o))
            self._logits_store(t, adr, o, inst)
        return body

    def _logits_store(self, t, adr, o, inst):
        e = self.e
        with self.atomic():
            e("s_mov_b64 exec, s[%d:%d]" % (S_SAVE, S_SAVE + 1))
            if inst:                # record column 1 + n_sem + ch
                e("s_lshl_b32 s%d, s%d, 2" % (S_T0, S_NSEM))
                e("s_add_u32 s%d, s%d, s%d" % (S_T0, S_REC_T + 2 * t, S_T0))
                e("s_addc_u32 s%d, s%d, 0" % (S_T1, S_REC_T + 2 * t + 1))
                self.vm_op("global_store_dword v%d, v%d, s[%d:%d] offset:4" % (adr, o, S_T0, S_T1))
            else:                   # record column 1 + ch
                self.vm_op("global_store_dword v%d, v%d, s[%d:%d] offset:4" % (adr, o, S_REC_T + 2 * t, S_REC_T + 2 * t + 1))
            e("s_mov_b64 exec, -1")

    # ------------------------------------------------------------------ accumulators
    def acc_reg(self, i):
        return V_ACC + 16 * i

    def acc_take(self, n):
        assert len(self.acc_free) >= n, ("accumulators exhausted", self.acc_free)
        got, self.acc_free = self.acc_free[:n], self.acc_free[n:]
        return got

    def acc_release(self, ids):
        self.acc_free = sorted(set(self.acc_free) | set(ids))

    # ------------------------------------------------------------------ arming: the bias of a unit into freshly taken accumulators
    def arm_ops(self, u):
        """list of closures, one LDS read each: bias of unit u -> its accumulators (u['accs'] is set here)"""
        c = self.chunks[u["chunk"]]
        l = self.layers[u["layer"]]
        slot = u["chunk"] % NSLOT
        bias_off = c["nfb"] * l["nks"] * 1024
        nb = len(u["blocks"])
        ids = self.acc_take(2 * nb)
        u["accs"] = {(bi, t): ids[bi * 2 + t] for bi in range(nb) for t in range(2)}
        u["arm_tags"] = []
        ops = []
        for bi, blk in enumerate(u["blocks"]):
            b_in_chunk = blk - c["fb"]
            for t in range(2):
                a = self.acc_reg(u["accs"][(bi, t)])
                if l["mode"] == "logits":       # transposed product: every register = bias of channel lane & 31
                    def mk_addr(slot=slot):
                        self.e("v_add_u32 v%d, 0x%x, v%d" % (V_T2, slot * SLOT, V_LB4))
                    ops.append(mk_addr)
                    for r in range(16):
                        ops.append(lambda a=a, r=r, off=bias_off + b_in_chunk * 128: u["arm_tags"].append(
                            self.lds_read("ds_read_b32 v%d, v%d offset:%d" % (a + r, V_T2, off))))
                else:
                    for m in range(4):
                        ops.append(lambda a=a, m=m, off=bias_off + b_in_chunk * 128 + m * 32: u["arm_tags"].append(
                            self.lds_read("ds_read_b128 %s, v%d offset:%d" % (vr(a + 4 * m, 4), V_BIAS + slot, off))))
        return ops

    # ------------------------------------------------------------------ pack / ReLU of a unit's accumulators -> list of (dst, closure)
    def pack_ops(self, u):
        l = self.layers[u["layer"]]
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
                    out.append(((dst.kind, dst.base + p), fn))
        return out

    # ------------------------------------------------------------------ one unit of MFMAs with its fillers
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
        # ---- what has to happen inside this unit
        # (1) the previous unit's pack / ReLU, each before the first MFMA of THIS unit that reads its destination
        prev = self.pending_pack
        self.pending_pack = None
        packs = prev["ops"] if prev else []
        reads = {}
        for i in range(nm):
            f, t = i // 2, i % 2
            loc = self.b_operand(u, frags[f][2], t)
            for r in range(4):
                reads.setdefault((loc.kind, loc.base + r), i)
        n = len(packs)
        pos = []
        for j, (dst, _) in enumerate(packs):
            spread = 1 + (j * max(1, int(0.62 * nm) - 1)) // max(1, n)
            dl = reads.get(dst)
            pos.append(spread if dl is None else min(spread, max(0, dl - 2)))
        for j in range(n - 2, -1, -1):                  # program order: a pack never after a later pack's position
            pos[j] = min(pos[j], pos[j + 1])
        pack_at = {}
        for j, g in enumerate(pos):
            pack_at.setdefault(g, []).append(j)
        # (2) LDS-DMA pieces of chunk + 3, spread over the chunk's units
        piece_at = {}
        cu = [k for k, x in enumerate(U) if x["chunk"] == u["chunk"]]
        c3 = (u["chunk"] + 3) % self.NC
        npieces = self.pieces_of(c3)
        k_in = cu.index(ui)
        mine = [j for j in range(npieces) if (j * len(cu)) // npieces == k_in]
        for idx, j in enumerate(mine):
            g = 2 + (idx * (nm - 4)) // max(1, len(mine))
            piece_at.setdefault(g, []).append(j)
        # (3) the next unit's bias: armed in this unit's tail, once its accumulators are free
        nxt = U[ui + 1] if ui + 1 < len(U) else None
        n_arm = 0 if nxt is None else 2 * len(nxt["blocks"]) * (17 if self.layers[nxt["layer"]]["mode"] == "logits" else 4)
        arm_from = max(nm - 5 - n_arm, (max(pos) + 1) if pos else 0, 0)      # done ~4 gaps before the unit ends: the ring's reads stay the youngest
        arm = None
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
                self.wait_lgkm(self.ring_tags[gi % P])
            a = self.acc_reg(u["accs"][(bi, t)])
            bloc = self.b_operand(u, ks, t)
            ring = vr(V_RING + 4 * (gi % P), 4)
            if swap:
                e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(a, 16), bloc.reg(0, 4), ring, vr(a, 16)))
            else:
                e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(a, 16), ring, bloc.reg(0, 4), vr(a, 16)))
            # ---- fillers of gap i
            if t == 1:
                gn = gi + P - 1
                if gn < len(self.stream):
                    sl, of = self.stream[gn]
                    self.ring_tags[gn % P] = self.lds_read("ds_read_b128 %s, v%d offset:%d" % (vr(V_RING + 4 * (gn % P), 4), V_FRAG + sl, of))
            for j in pack_at.get(i, []):
                packs[j][1]()
            if prev and i == (max(pos) if pos else 0):
                self.acc_release(prev["accs"])
                prev = None
            for j in piece_at.get(i, []):
                tg = self.piece(c3, j)
                self.piece_tag[c3] = tg
            if nxt is not None and i >= arm_from:
                if arm is None:
                    # the accumulators must be there: if a queued epilogue still holds some, run it now
                    need = 2 * len(nxt["blocks"])
                    while len(self.acc_free) < need and self.side_busy():
                        self.drain_side(8)
                    arm = self.arm_ops(nxt)
                per = -(-len(arm) // max(1, nm - 4 - i))
                for _ in range(min(per, len(arm))):
                    arm.pop(0)()
            self.drain_side(side_budget)
        if prev:
            self.acc_release(prev["accs"])
        if nxt is not None:
            if arm is None:
                while len(self.acc_free) < 2 * len(nxt["blocks"]) and self.side_busy():
                    self.drain_side(8)
                arm = self.arm_ops(nxt)
            while arm:
                arm.pop(0)()
        # ---- this unit's own results: packed during the next unit, or reduced by the side queue
        if l["mode"] in ("relu", "linear"):
            self.pending_pack = dict(ops=self.pack_ops(u), accs=list(u["accs"].values()))
        elif l["mode"] == "rgbs":
            for t in range(2):
                self.rgbs_epilogue(t, self.acc_reg(u["accs"][(0, t)]))
            ids = list(u["accs"].values())
            self.q(0, lambda: self.defer(lambda: self.acc_release(ids)))
            self.queue_next_group_inputs()
        else:
            inst = l["name"] == "inst1"
            ids = list(u["accs"].values())
            for t in range(2):
                self.q(66, (lambda t=t: self.lwr_build(t)))
                for bi, blk in enumerate(u["blocks"]):
                    self.q(32, self.logits_epilogue(t, self.acc_reg(u["accs"][(bi, t)]), blk, inst))
            self.q(0, lambda: self.defer(lambda: self.acc_release(ids)))
        # ---- chunk hand-over
        if u["last_of_chunk"]:
            c2 = (u["chunk"] + 2) % self.NC
            self.wait_vm(self.piece_tag.get(c2))
            e("s_barrier")

    def lwr_build(self, t):
        """lwr[r] of tile t -> v[V_TMP + r]: 32 readlanes, then two moves per register (full exec: the hi = 1 value; lanes 0..31: hi = 0)"""
        e = self.e
        self.lwr_load(t)
        for r in range(16):
            e("v_mov_b32 v%d, s%d" % (V_TMP + r, S_W + (r & 3) + 8 * (r >> 2) + 4))
        with self.atomic():
            e("s_mov_b64 exec, s[%d:%d]" % (S_HI0, S_HI0 + 1))
            for r in range(16):
                e("v_mov_b32 v%d, s%d" % (V_TMP + r, S_W + (r & 3) + 8 * (r >> 2)))
            e("s_mov_b64 exec, -1")

    def queue_next_group_inputs(self):
        """side work: the next group's inputs, encodings, |d| (V_IN, V_EX, V_ED)"""
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
            for cost, fn in self.encode_tile(t, None):
                self.q(cost, fn)

    # ------------------------------------------------------------------ group boundary
    def prefetch_first_unit(self):
        """bias of unit 0 into fresh accumulators + the ring's first three fragments (the only LGKM operations in flight at a
        group's start)"""
        u0 = self.units[0]
        for op in self.arm_ops(u0):
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

    def group_body(self):
        assert self.acc_free == [i for i in range(8) if i not in self.units[0]["accs"].values()], self.acc_free
        for ui in range(len(self.units)):
            self.emit_unit(ui)
        self.drain_side()                               # whatever the last units could not cover (the last logit block's reduction)
        assert self.acc_free == list(range(8)), self.acc_free
        self.e("s_waitcnt lgkmcnt(0)")
        self.lgkm.drain()
        self.advance_group_state()

    # ------------------------------------------------------------------ the kernel
    def kernel(self):
        e, name = self.e, self.name
        self.o += ["\t.text", "\t.globl\t%s" % name, "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
        for dst, off, n in ((S_IMG, 0x0, 2), (S_RAYS, 0x8, 2), (S_Z, 0x10, 2), (S_S, 0x18, 2), (S_MAGIC, 0x20, 2), (S_NGRP, 0x28, 2),
                            (S_REC, 0x30, 2), (S_RECF, 0x38, 1), (S_PS, 0x40, 2), (S_NSEM, 0x48, 2), (S_CLK, 0x50, 2)):
            e("s_load_dword%s %s, s[0:1], 0x%x" % ("x2" if n == 2 else "", sr(dst, n), off))
        # wave id from v0 itself (never written): a v_readfirstlane of a register that is re-used a few instructions later was
        # observed to return the LATER value while scalar-load data was returning (tools/probe/gen_two_tile_asm.py)
        e("v_readfirstlane_b32 s%d, v0" % S_WAVE)
        e("s_nop 4")
        e("s_lshr_b32 s%d, s%d, 6" % (S_WAVE, S_WAVE))
        e("s_lshl_b32 s%d, s%d, 10" % (S_W1K, S_WAVE))
        e("v_and_b32 v%d, 63, v0" % V_T0)
        e("v_lshlrev_b32 v%d, 4, v%d" % (V_LANE16, V_T0))
        e("v_lshrrev_b32 v%d, 5, v0" % V_T1)
        e("v_and_b32 v%d, 1, v%d" % (V_T1, V_T1))
        e("v_lshlrev_b32 v%d, 4, v%d" % (V_T1, V_T1))                    # hi * 16
        for sl in range(NSLOT):
            self.lit(S_T0, sl * SLOT)
            e("v_add_u32 v%d, s%d, v%d" % (V_FRAG + sl, S_T0, V_LANE16))
            e("v_add_u32 v%d, s%d, v%d" % (V_BIAS + sl, S_T0, V_T1))
        e("v_and_b32 v%d, 31, v0" % V_LB4)
        e("v_lshlrev_b32 v%d, 2, v%d" % (V_LB4, V_LB4))
        e("v_mov_b32 v%d, 0" % V_ZERO)
        e("s_mov_b32 s%d, -1" % S_LO32)
        e("s_mov_b32 s%d, 0" % (S_LO32 + 1))
        e("s_mov_b32 s%d, 1" % S_N0)
        e("s_mov_b32 s%d, 1" % (S_N0 + 1))
        e("s_mov_b32 s%d, -1" % S_HI0)
        e("s_mov_b32 s%d, 0" % (S_HI0 + 1))
        e("s_waitcnt lgkmcnt(0)")
        e("s_add_u32 s%d, s%d, s%d" % (S_IMGW, S_IMG, S_W1K))
        e("s_addc_u32 s%d, s%d, 0" % (S_IMGW + 1, S_IMG + 1))
        lend, lloop, lfin = self.label(), self.label(), self.label()
        e("s_mov_b32 s%d, s2" % S_GRP)
        e("s_cmp_ge_i32 s%d, s%d" % (S_GRP, S_NGRP))
        e("s_cbranch_scc1 %s" % lend)
        # chunks 0, 1, 2 -> slots 0, 1, 2
        for c in range(3):
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
            for _, fn in self.encode_tile(t, None):
                fn()
        self.advance_group_state()
        e("s_memtime s[%d:%d]" % (S_CLK0, S_CLK0 + 1))
        e("s_memrealtime s[%d:%d]" % (S_CLK0 + 2, S_CLK0 + 3))
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
        e("s_cmp_lg_u32 s2, 0")
        e("s_cbranch_scc1 %s" % lfin)
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
        e("v_mov_b32 v%d, 0" % V_ZERO)
        e("global_store_dwordx4 v%d, v[20:23], s[%d:%d]" % (V_ZERO, S_CLK, S_CLK + 1))
        e("s_waitcnt vmcnt(0)")
        self.o.append(lfin + ":")
        e("s_endpgm")
        self.o += [".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name), ""]
        self.o += ["\t.rodata", "\t.p2align\t6", "\t.amdhsa_kernel %s" % name,
                   "\t\t.amdhsa_group_segment_fixed_size %d" % (NSLOT * SLOT),
                   "\t\t.amdhsa_private_segment_fixed_size 0", "\t\t.amdhsa_kernarg_size %d" % KERNARG_BYTES, "\t\t.amdhsa_user_sgpr_count 2",
                   "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1", "\t\t.amdhsa_system_sgpr_workgroup_id_x 1",
                   "\t\t.amdhsa_system_vgpr_workitem_id 0", "\t\t.amdhsa_next_free_vgpr 512", "\t\t.amdhsa_next_free_sgpr 102",
                   "\t\t.amdhsa_accum_offset 256", "\t\t.amdhsa_reserve_vcc 1", "\t\t.amdhsa_float_denorm_mode_32 3",
                   "\t\t.amdhsa_float_denorm_mode_16_64 3", "\t\t.amdhsa_dx10_clamp 1", "\t\t.amdhsa_ieee_mode 1",
                   "\t.end_amdhsa_kernel", ""]
        return "\n".join(self.o)


KERNARG_BYTES = 88


def metadata(names):
    o = ["\t.amdgpu_metadata", "---", "amdhsa.kernels:"]
    for n in names:
        o += ["  - .agpr_count:     256", "    .args:", "      - .offset:         0", "        .size:           %d" % KERNARG_BYTES,
              "        .value_kind:     by_value", "    .group_segment_fixed_size: %d" % (NSLOT * SLOT),
              "    .kernarg_segment_align: 8", "    .kernarg_segment_size: %d" % KERNARG_BYTES, "    .max_flat_workgroup_size: 256",
              "    .name:           %s" % n, "    .private_segment_fixed_size: 0", "    .sgpr_count:     108",
              "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % n, "    .uniform_work_group_size: 1",
              "    .uses_dynamic_stack: false", "    .vgpr_count:     512", "    .vgpr_spill_count: 0", "    .wavefront_size: 64"]
    o += ["amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", "\t.end_amdgpu_metadata", ""]
    return "\n".join(o)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    parts = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 5", ""]
    names = []
    for nbs, nbi in ((1, 1), (2, 1)):
        n = "k_mlp_tt_s%di%d" % (nbs, nbi)
        names.append(n)
        parts.append(Gen(nbs, nbi, n).kernel())
    parts.append(metadata(names))
    with open(out, "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
