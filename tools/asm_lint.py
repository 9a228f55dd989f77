"""Build-time lint of the gfx950 assembly of libpnr.so's objects: the three things in the kernels that rest on
what the COMPILER emitted rather than on what the source says, checked on the assembly of the very compile that produced the
object (`hipcc -save-temps=obj`; panopticnerf_amd/csrc/Makefile runs this after every .hip object and fails the build):

  1. pending LDS destinations (check_lds_pending.py): no instruction reads or overwrites the destination of an inline-asm
     ds_read the hand-counted lgkmcnt waits have not covered, along every path through the kernel's branches;
  2. the vmcnt markers of k_mlp_pp (pnr_mlp.hip): between PNR_FETCH_BEGIN and PNR_FETCH_END <N> there are exactly N plain
     global loads, no LDS-DMA piece and no branch -- N is the constant the following m_done() waits with;
  3. M0: pnr_dma_piece (pnr_common.h) sets M0 by hand; every access to M0 in the object must be that helper's own
     `s_mov_b32 m0, sN` directly followed by `s_nop` and a scalar-base `global_load_lds_dwordx4 vN, s[a:b]`, and no per-lane
     64-bit LDS-DMA address may exist.

usage: python tools/asm_lint.py file.s [file.s ...]      exit status 1 and one line per finding on failure.
tests/test_asm_lint.py runs the same functions (and keeps the unit tests of the checker itself)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_lds_pending as lds  # noqa: E402


def fetch_markers(text):
    """-> list of findings for check 2; [] when the listing carries no marker."""
    out = []
    begins = [i for i, l in enumerate(text) if "PNR_FETCH_BEGIN" in l]
    for b in begins:
        e = next((i for i in range(b + 1, len(text)) if "PNR_FETCH_END" in text[i]), None)
        if e is None:
            out.append("line %d: PNR_FETCH_BEGIN without PNR_FETCH_END" % (b + 1))
            continue
        want = int(re.search(r"PNR_FETCH_END (\d+)", text[e]).group(1))
        body = [l.split()[0] for l in text[b + 1:e] if l.strip() and not l.strip().startswith(";")]
        vmem = [op for op in body if op.startswith(("global_", "buffer_", "flat_", "scratch_"))]
        if not all(op.startswith("global_load_dword") and "lds" not in op for op in vmem):
            out.append("line %d: something other than plain global loads between the fetch markers: %s" % (b + 1, vmem))
        if len(vmem) != want:
            out.append("line %d: %d vector-memory instructions between the fetch markers, the wait assumes %d" % (b + 1, len(vmem), want))
        if any(op.startswith("s_cbranch") or op.startswith("s_branch") for op in body):
            out.append("line %d: a branch between the fetch markers" % (b + 1))
    return out


def m0_accesses(text):
    """(lines that write M0 with pnr_dma_piece's own s_mov, every other line that mentions m0)."""
    code = [l.split(";")[0].strip() for l in text]
    mine = [l for l in code if re.match(r"s_mov_b32\s+m0,\s*(s\d+|vcc_lo|vcc_hi)$", l)]
    other = [l for l in code if re.search(r"\bm0\b", l) and l not in mine]
    return mine, other


def m0_rule(text):
    out = []
    mine, other = m0_accesses(text)
    out += ["M0 touched outside pnr_dma_piece: %s" % l for l in other[:5]]
    code = [l.split(";")[0].strip() for l in text if l.split(";")[0].strip()]
    for i, l in enumerate(code):
        if l in mine and not (i + 2 < len(code) and code[i + 1].startswith("s_nop")
                              and re.match(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", code[i + 2])):
            out.append("s_mov m0 not followed by s_nop + scalar-base LDS-DMA: %s" % " | ".join(code[i:i + 3]))
    if any(l.startswith("global_load_lds_dwordx4 v[") for l in code):
        out.append("an LDS-DMA piece with a per-lane 64-bit address")
    return out


def has_inline_lds_reads(text):
    """Does the listing carry ds_read instructions inside inline-asm blocks (;;#ASMSTART .. ;;#ASMEND)?  Check 1 is for those:
    the compiler's own LDS reads come with the compiler's own (correct) waits, whose counts also cover ds_write and scalar
    loads the checker does not model."""
    inside = False
    for l in text:
        if "#ASMSTART" in l:
            inside = True
        elif "#ASMEND" in l:
            inside = False
        elif inside and l.strip().startswith("ds_read"):
            return True
    return False


def lint_file(path):
    text = open(path).read().split("\n")
    findings = []
    if has_inline_lds_reads(text):
        findings = ["%s:%d: %s (%s; ds_read at line %d)" % (os.path.basename(path), f[0], f[2], f[1], f[3])
                    for f in lds.check_cfg(text)[:10]]
    findings += ["%s: %s" % (os.path.basename(path), m) for m in fetch_markers(text) + m0_rule(text)]
    return findings


def main(argv):
    bad = []
    for p in argv:
        bad += lint_file(p)
    for b in bad:
        print("asm_lint: " + b, file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
