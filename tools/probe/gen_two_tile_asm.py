#!/usr/bin/env python3
"""Generator of the HAND-PLACED two-tile prototype (gfx950 assembly, timing only; round-4 verdict item 1).

One wave per SIMD (4 waves per workgroup, 512 registers per lane), TWO 32-sample tiles per wave: every 1 KiB weight fragment read
from the LDS feeds two v_mfma_f32_32x32x16_bf16 (tile 0, tile 1).  The same steady hidden-layer loop as
panopticnerf_amd/csrc/bench/pnr_proto_two_tile.hip (which hipcc schedules: its accumulators end up in AGPRs, every packed output
costs two v_accvgpr_read, one bias tuple bounces through VGPRs) -- here every register and every issue slot is chosen by hand:

  registers   activations (MFMA B operand): AGPRs   cur/nxt x 2 tiles x 64 packed bf16x2 = a[0:255]
              accumulators: VGPRs, two sets X, Y of 4 x 16 (2 output blocks x 2 tiles) = v[32:159] -- the pack / ReLU reads them
              without a move; one v_accvgpr_write per packed output is the price of the AGPR activations
              fragment ring: 4 quads v[16:31]; per-slot LDS bases v[2:9]
  per MFMA gap (32 cycles = 8 issue slots, <= 5 fillers hide: MI355X_MICROARCH.md) in a 64-MFMA chunk:
              0.5 s_waitcnt + 0.5 ds_read_b128 (fragment) ; gaps 0..42: 3 of 4 gaps pack one output register of the PREVIOUS chunk
              (v_cvt_pk_bf16_f32, v_pk_max_i16, v_accvgpr_write) ; gaps 44..59: one bias quad of the NEXT chunk straight into the
              vacated accumulator ; every 7th gap one LDS-DMA piece (m0 + 2 SALU + global_load_lds_dwordx4 nt)
  weight stream  4 LDS slots x 33 KiB; during chunk c the wave issues its 8-9 pieces of chunk c+3; at the end of chunk c
              `s_waitcnt vmcnt(8)` (own pieces of chunk c+2 landed) + ONE s_barrier; the tail of chunk c+1 may then read chunk c+2:
              the fragment ring never drains
  layers      10 x 256-wide per 256-sample group (40 chunks of 2 blocks x 16 k-steps: 1280 MFMAs per tile; the real fused plan
              runs 1336), two layers per loop trip (in -> out, out -> in): no hand-over copies

FLAGS: 1 = LDS-DMA pieces, 2 = fragment / bias reads, 4 = pack / ReLU epilogue; the file holds one kernel per flag set.
Nothing here computes a network: results are not checked (pnrb_proto_two_tile_asm only times)."""
import sys

P, NFRAG, CF, NSLOT, NCHUNK = 4, 32, 33, 4, 40
SLOT = CF * 1024
BIAS0, BIAS1 = 44, 60
TOPWAIT = (64 - BIAS1) // 2
LAYER_PAIRS = NCHUNK // 8

# VGPR map
V_TID, V_LANE16, V_FRAG, V_BIAS, V_T0, V_T1, V_HASH, V_ZERO, V_RING, V_ACC = 0, 1, 2, 6, 10, 11, 12, 13, 16, 32
# SGPR map
S_IMG, S_NGRP, S_NWG, S_SINK, S_CLK, S_WAVE, S_GRP, S_W1K, S_C3, S_SRC, S_SRCW, S_PIECE, S_TRIP = 4, 6, 7, 8, 10, 12, 13, 14, 15, 16, 18, 20, 22
S_K = 40        # s40.. constants of the hash


def frag_off(f):
    return ((f & 1) * 16 + (f >> 1)) * 1024


def younger(f):
    return 2 + sum(1 for g in range(2 * f - 5, 2 * f) if BIAS0 <= g < BIAS1)


def epi_reg(g):
    return -1 if (g & 3) == 3 else (g >> 2) * 3 + (g & 3)


def acc(st, b, t):
    return V_ACC + st * 64 + (b * 2 + t) * 16


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1)


def ar(lo, n):
    return "a[%d:%d]" % (lo, lo + n - 1)


class Gen:
    def __init__(self, flags, name, stop=0):
        self.flags, self.name, self.o, self.nlabel, self.stop = flags, name, [], 0, stop       # stop: debug builds end after stage `stop`

    def e(self, s):
        self.o.append("\t" + s)

    def label(self):
        self.nlabel += 1
        return ".L%s_%d" % (self.name, self.nlabel)

    def piece(self, src_lo, src_hi, lds_imm, j, guard_wave0):
        """one 1 KiB LDS-DMA piece: fragment wave + 4 j of the chunk at s[src_lo:src_hi] (+ wave KiB folded in) -> LDS lds_imm + wave KiB"""
        skip = None
        if guard_wave0:
            skip = self.label()
            self.e("s_cmp_lg_u32 s%d, 0" % S_WAVE)
            self.e("s_cbranch_scc1 %s" % skip)
        self.e("s_add_u32 m0, s%d, 0x%x" % (S_W1K, lds_imm + j * 4096))
        self.e("s_add_u32 s%d, s%d, 0x%x" % (S_PIECE, src_lo, j * 4096))
        self.e("s_addc_u32 s%d, s%d, 0" % (S_PIECE + 1, src_hi))
        self.e("global_load_lds_dwordx4 v%d, s[%d:%d] nt" % (V_LANE16, S_PIECE, S_PIECE + 1))
        if skip:
            self.o.append(skip + ":")

    def chunk(self, LP, CB):
        DMA, READS, EPI = self.flags & 1, self.flags & 2, self.flags & 4
        cur, prv = CB & 1, 1 - (CB & 1)
        inb = lambda t: (0 if LP == 0 else 128) + 64 * t
        outb = lambda t: (128 if LP == 0 else 0) + 64 * t
        dstb, PB = (inb, 6) if CB == 0 else (outb, 2 * (CB - 1))
        slot, slotn, slotr = CB, (CB + 1) % NSLOT, (CB + 3) % NSLOT
        self.e("; ---- layer parity %d chunk %d" % (LP, CB))
        if DMA:     # source of chunk cg + 3: image + c3 * SLOT (+ wave KiB); c3 advances with wrap
            self.e("s_mul_i32 s%d, s%d, 0x%x" % (S_PIECE, S_C3, SLOT))
            self.e("s_add_u32 s%d, s%d, s%d" % (S_SRC, S_IMG, S_PIECE))
            self.e("s_addc_u32 s%d, s%d, 0" % (S_SRC + 1, S_IMG + 1))
            self.e("s_add_u32 s%d, s%d, s%d" % (S_SRCW, S_SRC, S_W1K))
            self.e("s_addc_u32 s%d, s%d, 0" % (S_SRCW + 1, S_SRC + 1))
            self.e("s_add_u32 s%d, s%d, 1" % (S_C3, S_C3))
            self.e("s_cmp_eq_u32 s%d, %d" % (S_C3, NCHUNK))
            self.e("s_cselect_b32 s%d, 0, s%d" % (S_C3, S_C3))
        if READS:
            self.e("s_waitcnt lgkmcnt(%d)" % TOPWAIT)          # this chunk's bias quads (requested in the previous chunk's tail)
        for i in range(64):
            ks, b, t = i >> 2, (i >> 1) & 1, i & 1
            f = ks * 2 + b
            if t == 0 and READS:
                self.e("s_waitcnt lgkmcnt(%d)" % younger(f))
            a = acc(cur, b, t)
            self.e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(a, 16), vr(V_RING + 4 * (f % P), 4), ar(inb(t) + 4 * ks, 4), vr(a, 16)))
            # ---- fillers of gap i
            if t == 1 and READS:
                fn = f + P - 1
                if fn < NFRAG:
                    self.e("ds_read_b128 %s, v%d offset:%d" % (vr(V_RING + 4 * (fn % P), 4), V_FRAG + slot, frag_off(fn)))
                else:
                    self.e("ds_read_b128 %s, v%d offset:%d" % (vr(V_RING + 4 * (fn % P), 4), V_FRAG + slotn, frag_off(fn - NFRAG)))
            if EPI and 0 <= epi_reg(i) < 32:
                q = epi_reg(i)
                bb, tt, p = q >> 4, (q >> 3) & 1, q & 7
                src = acc(prv, bb, tt) + 2 * p
                tmp = V_T0 if (q & 1) else V_T1
                self.e("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (tmp, src, src + 1))
                self.e("v_pk_max_i16 v%d, v%d, 0" % (tmp, tmp))
                self.e("v_accvgpr_write_b32 a%d, v%d" % (dstb(tt) + (PB + bb) * 8 + p, tmp))
            if READS and BIAS0 <= i < BIAS1:
                j = i - BIAS0
                tup, m = j >> 2, j & 3
                self.e("ds_read_b128 %s, v%d offset:%d" % (vr(acc(prv, tup >> 1, tup & 1) + 4 * m, 4), V_BIAS + slotn,
                                                           NFRAG * 1024 + (tup >> 1) * 128 + m * 32))
            if DMA and i % 7 == 2:
                j = i // 7
                self.piece(S_SRCW, S_SRCW + 1, slotr * SLOT, j, guard_wave0=(j == 8))
        if DMA:
            self.e("s_waitcnt vmcnt(8)")
        self.e("s_barrier")

    def kernel(self):
        e, name = self.e, self.name
        self.o += ["\t.text", "\t.globl\t%s" % name, "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
        e("s_load_dwordx2 s[%d:%d], s[0:1], 0x0" % (S_IMG, S_IMG + 1))
        e("s_load_dwordx2 s[%d:%d], s[0:1], 0x8" % (S_NGRP, S_NGRP + 1))
        e("s_load_dwordx2 s[%d:%d], s[0:1], 0x10" % (S_SINK, S_SINK + 1))
        e("s_load_dwordx2 s[%d:%d], s[0:1], 0x18" % (S_CLK, S_CLK + 1))
        # wave id from v0 itself: v0 is never written.  (Measured: `v_lshrrev v3, 6, v0; v_readfirstlane s12, v3` with v3 re-used ten
        # instructions later returned the LATER value of v3 in two of four waves -- the read of a v_readfirstlane's source is not
        # ordered against younger VALU writes of it while scalar-load data is returning.)
        e("v_readfirstlane_b32 s%d, v0" % S_WAVE)
        e("s_nop 4")
        e("s_lshr_b32 s%d, s%d, 6" % (S_WAVE, S_WAVE))
        e("s_lshl_b32 s%d, s%d, 10" % (S_W1K, S_WAVE))
        e("v_and_b32 v1, 63, v0")
        e("v_lshlrev_b32 v%d, 4, v1" % V_LANE16)
        e("v_lshrrev_b32 v10, 5, v0")
        e("v_and_b32 v10, 1, v10")
        e("v_lshlrev_b32 v10, 4, v10")
        for sl in range(NSLOT):
            e("s_mov_b32 s%d, 0x%x" % (S_PIECE, sl * SLOT))
            e("v_add_u32 v%d, s%d, v%d" % (V_FRAG + sl, S_PIECE, V_LANE16))
            e("v_add_u32 v%d, s%d, v10" % (V_BIAS + sl, S_PIECE))
        for k, c in enumerate((1664525, 1013904223, 0x007f007f, 0x3f003f00, 2654435761, 40503)):
            e("s_mov_b32 s%d, 0x%x" % (S_K + k, c))
        e("s_waitcnt lgkmcnt(0)")
        lend = self.label()
        if self.stop == 1:
            e("s_branch %s" % lend)
        if self.stop == 8:            # debug: dump s4, s5 (image), s12 (wave), s14 to the clock buffer
            ldump = self.label()
            e("s_cmp_lg_u32 s2, 0")
            e("s_cbranch_scc1 %s" % ldump)
            e("v_mov_b32 v20, s4")
            e("v_mov_b32 v21, s5")
            e("v_mov_b32 v22, s12")
            e("v_mov_b32 v23, v0")
            e("v_mov_b32 v13, 0")
            e("v_cmp_eq_u32 vcc, 0, v1")
            e("s_and_saveexec_b64 s[38:39], vcc")
            e("global_store_dwordx4 v13, v[20:23], s[10:11]")
            e("s_waitcnt vmcnt(0)")
            self.o.append(ldump + ":")
            e("s_endpgm")
        # chunks 0, 1, 2 -> slots 0, 1, 2
        for c in range(3):
            e("s_add_u32 s%d, s%d, 0x%x" % (S_SRC, S_IMG, c * SLOT))
            e("s_addc_u32 s%d, s%d, 0" % (S_SRC + 1, S_IMG + 1))
            e("s_add_u32 s%d, s%d, s%d" % (S_SRCW, S_SRC, S_W1K))
            e("s_addc_u32 s%d, s%d, 0" % (S_SRCW + 1, S_SRC + 1))
            for j in range(9):
                if self.stop == 7 and c == 0 and j == 0:       # debug: the same first piece as a plain register load (no LDS)
                    e("s_add_u32 s%d, s%d, 0" % (S_PIECE, S_SRCW))
                    e("s_addc_u32 s%d, s%d, 0" % (S_PIECE + 1, S_SRCW + 1))
                    e("global_load_dwordx4 v[20:23], v%d, s[%d:%d]" % (V_LANE16, S_PIECE, S_PIECE + 1))
                    break
                self.piece(S_SRCW, S_SRCW + 1, c * SLOT, j, guard_wave0=(j == 8))
                if self.stop == 5 and c == 0 and j == 0:
                    break
            if self.stop in (5, 6, 7) and c == 0:
                break
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        if self.stop in (5, 6, 7):
            e("s_branch %s" % lend)
        e("s_mov_b32 s%d, 3" % S_C3)
        if self.stop == 2:
            e("s_branch %s" % lend)
        # activations: post-ReLU-like bf16 pairs in [0.5, 1) from a per-lane LCG
        e("v_mul_lo_u32 v%d, v0, s%d" % (V_HASH, S_K + 4))
        e("s_mul_i32 s%d, s2, s%d" % (S_PIECE, S_K + 5))
        e("v_add_u32 v%d, s%d, v%d" % (V_HASH, S_PIECE, V_HASH))

        def rnd(areg):
            e("v_mul_lo_u32 v%d, v%d, s%d" % (V_HASH, V_HASH, S_K))
            e("v_add_u32 v%d, s%d, v%d" % (V_HASH, S_K + 1, V_HASH))
            e("v_lshrrev_b32 v%d, 8, v%d" % (V_T0, V_HASH))
            e("v_and_b32 v%d, s%d, v%d" % (V_T0, S_K + 2, V_T0))
            e("v_or_b32 v%d, s%d, v%d" % (V_T0, S_K + 3, V_T0))
            e("v_accvgpr_write_b32 a%d, v%d" % (areg, V_T0))
        for r in range(256):
            rnd(r)
        for r in range(64):                      # Y = 0 (its "previous results" are packed during the first chunk)
            e("v_mov_b32 v%d, 0" % (acc(1, 0, 0) + r))
        e("v_mov_b32 v%d, 0" % V_ZERO)
        if self.stop == 3:
            e("s_branch %s" % lend)
        e("s_memtime s[30:31]")
        e("s_memrealtime s[32:33]")
        e("s_waitcnt lgkmcnt(0)")
        # chunk 0's bias quads into X, then the ring's first three fragments (the youngest LGKM operations, as in steady state)
        for j in range(16):
            tup, m = j >> 2, j & 3
            e("ds_read_b128 %s, v%d offset:%d" % (vr(acc(0, tup >> 1, tup & 1) + 4 * m, 4), V_BIAS, NFRAG * 1024 + (tup >> 1) * 128 + m * 32))
        for f in range(P - 1):
            e("ds_read_b128 %s, v%d offset:%d" % (vr(V_RING + 4 * f, 4), V_FRAG, frag_off(f)))
        if not (self.flags & 2):
            e("s_waitcnt lgkmcnt(0)")
        if self.stop == 4:
            e("s_branch %s" % lend)
        e("s_mov_b32 s%d, s2" % S_GRP)
        lgrp, llayer, lfin = self.label(), self.label(), self.label()
        self.o.append(lgrp + ":")
        e("s_cmp_ge_i32 s%d, s%d" % (S_GRP, S_NGRP))
        e("s_cbranch_scc1 %s" % lend)
        for t in range(2):                       # this group's "inputs": block 0 of both tiles
            for r in range(8):
                rnd(64 * t + r)
        e("s_mov_b32 s%d, %d" % (S_TRIP, LAYER_PAIRS))
        self.o.append(llayer + ":")
        for LP in range(2):
            for CB in range(4):
                self.chunk(LP, CB)
        e("s_sub_u32 s%d, s%d, 1" % (S_TRIP, S_TRIP))
        e("s_cmp_lg_u32 s%d, 0" % S_TRIP)
        e("s_cbranch_scc1 %s" % llayer)
        e("s_add_u32 s%d, s%d, s%d" % (S_GRP, S_GRP, S_NWG))
        e("s_branch %s" % lgrp)
        self.o.append(lend + ":")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_memtime s[34:35]")
        e("s_memrealtime s[36:37]")
        e("s_waitcnt lgkmcnt(0)")
        e("s_cmp_lg_u32 s2, 0")
        e("s_cbranch_scc1 %s" % lfin)
        e("v_cmp_eq_u32 vcc, 0, v0")
        e("s_and_saveexec_b64 s[38:39], vcc")
        e("s_cbranch_execz %s" % lfin)
        e("s_sub_u32 s34, s34, s30")
        e("s_subb_u32 s35, s35, s31")
        e("s_sub_u32 s36, s36, s32")
        e("s_subb_u32 s37, s37, s33")
        for k in range(4):
            e("v_mov_b32 v%d, s%d" % (20 + k, 34 + k))
        e("v_mov_b32 v%d, 0" % V_ZERO)
        e("global_store_dwordx4 v%d, v[20:23], s[%d:%d]" % (V_ZERO, S_CLK, S_CLK + 1))
        e("s_waitcnt vmcnt(0)")
        self.o.append(lfin + ":")
        e("s_endpgm")
        self.o += [".Lend_%s:" % name, "\t.size\t%s, .Lend_%s-%s" % (name, name, name), ""]
        self.o += ["\t.rodata", "\t.p2align\t6", "\t.amdhsa_kernel %s" % name,
                   "\t\t.amdhsa_group_segment_fixed_size %d" % (NSLOT * SLOT),
                   "\t\t.amdhsa_private_segment_fixed_size 0", "\t\t.amdhsa_kernarg_size 32", "\t\t.amdhsa_user_sgpr_count 2",
                   "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1", "\t\t.amdhsa_system_sgpr_workgroup_id_x 1",
                   "\t\t.amdhsa_system_vgpr_workitem_id 0", "\t\t.amdhsa_next_free_vgpr 512", "\t\t.amdhsa_next_free_sgpr 48",
                   "\t\t.amdhsa_accum_offset 256", "\t\t.amdhsa_reserve_vcc 1", "\t\t.amdhsa_float_denorm_mode_32 3",
                   "\t\t.amdhsa_float_denorm_mode_16_64 3", "\t\t.amdhsa_dx10_clamp 1", "\t\t.amdhsa_ieee_mode 1",
                   "\t.end_amdhsa_kernel", ""]
        return "\n".join(self.o)


def metadata(names):
    o = ["\t.amdgpu_metadata", "---", "amdhsa.kernels:"]
    for n in names:
        o += ["  - .agpr_count:     256", "    .args:", "      - .offset:         0", "        .size:           32",
              "        .value_kind:     by_value", "    .group_segment_fixed_size: %d" % (NSLOT * SLOT),
              "    .kernarg_segment_align: 8", "    .kernarg_segment_size: 32", "    .max_flat_workgroup_size: 256",
              "    .name:           %s" % n, "    .private_segment_fixed_size: 0", "    .sgpr_count:     56",
              "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % n, "    .uniform_work_group_size: 1",
              "    .uses_dynamic_stack: false", "    .vgpr_count:     512", "    .vgpr_spill_count: 0", "    .wavefront_size: 64"]
    o += ["amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", "\t.end_amdgpu_metadata", ""]
    return "\n".join(o)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    flag_sets = (7, 6, 3, 2, 0)
    parts = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 5", ""]
    names = []
    for fl in flag_sets:
        n = "k_two_tile_asm_f%d" % fl
        names.append(n)
        parts.append(Gen(fl, n).kernel())
    for stop in (1, 2, 3, 4, 5, 6, 7, 8):          # debug: the full kernel cut short after stage `stop` (pnrb_proto_two_tile_asm flags 100 + stop)
        n = "k_two_tile_asm_f%d" % (100 + stop)
        names.append(n)
        parts.append(Gen(7, n, stop).kernel())
    parts.append(metadata(names))
    with open(out, "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
