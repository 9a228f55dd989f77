"""numpy emulation of the fused MLP kernel's DATAFLOW (panopticnerf_amd/csrc/pnr_mlp.hip) on
the packed weight image, using the documented gfx950 MFMA fragment layouts
(cdna_hip_programming.md section 3):  lane l = (n = l & 31, hi = l >> 5);
D[i][n] += sum_{hi,j} A_lane(i,hi)[j] * B_lane(n,hi)[j];  accumulator register r of lane
(n,hi) holds D[(r&3) + 8*(r>>2) + 4*hi][n].

This is a CPU check of the host packer + slot maps + chunk order (a `not gpu` test of the
host logic): if the emulation matches the dense oracle MLP, the only things left for the GPU
to prove are the MFMA builtin's own layout and the kernel's indexing.
"""
import numpy as np
import torch


def _row(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def _bf(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _embed_lane(p, hi, nf, nv):
    """p (n,3) -> lane-vector share (n, nv) of half-wave hi, mirroring embed_lane<> in the kernel."""
    n = p.shape[0]
    v = np.zeros((n, nv), np.float32)
    v[:, 0] = p[:, 2] if hi else p[:, 0]
    v[:, 1] = 0.0 if hi else p[:, 1]
    for fp in range(nf):
        sc = np.float32(2.0 ** ((nf + fp) if hi else fp))
        for a in range(3):
            arg = (p[:, a] * sc).astype(np.float32)
            v[:, 2 + 6 * fp + a] = np.sin(arg.astype(np.float64)).astype(np.float32)
            v[:, 2 + 6 * fp + 3 + a] = np.cos(arg.astype(np.float64)).astype(np.float32)
    return v


class PackedImage:
    def __init__(self, img_u8):
        b = img_u8.numpy().tobytes() if hasattr(img_u8, "numpy") else bytes(img_u8)
        self.b = b
        h = np.frombuffer(b[:128], np.uint32)
        assert h[0] == 0x504E5231
        self.n_chunks, self.max_frags, self.table_off, self.data_off = int(h[2]), int(h[3]), int(h[4]), int(h[5])
        self.desc = np.frombuffer(b[32:32 + 64], np.int32)
        self.table = np.frombuffer(b[self.table_off:self.table_off + 8 * self.n_chunks], np.uint32).reshape(-1, 2)
        self.bf16 = int(self.desc[8]) == 0
        self.kpl = 8 if self.bf16 else 4

    def chunk(self, ci):
        """-> (A fragments (nfrag-1, 64 lanes, kpl), bias floats (256)) of chunk ci."""
        off, nfrag = int(self.table[ci, 0]), int(self.table[ci, 1])
        raw = self.b[self.data_off + off * 1024: self.data_off + (off + nfrag) * 1024]
        nw = nfrag - 1
        if self.bf16:
            u = np.frombuffer(raw[:nw * 1024], np.uint16).astype(np.uint32) << 16
            A = u.view(np.float32).reshape(nw, 64, 8)
        else:
            A = np.frombuffer(raw[:nw * 1024], np.float32).reshape(nw, 64, 4)
        bias = np.frombuffer(raw[nw * 1024: nw * 1024 + 1024], np.float32)
        return A, bias


def emulate(img_u8, pts, viewdirs):
    """pts, viewdirs (n,3) float32 (n a multiple of 32 not required) -> raw (n, 4+C+K) float32."""
    im = PackedImage(img_u8)
    D, W, skip, n_sem, n_inst = (int(im.desc[i]) for i in (0, 1, 2, 5, 6))
    kpl, bf = im.kpl, im.bf16
    q = _bf if bf else (lambda x: np.asarray(x, np.float32))
    n = pts.shape[0]
    state = {"ci": 0}

    def layer(segs, n_out, relu, to_regs=True):
        """segs: list of [V_hi0, V_hi1] each (n, VL).  Returns lane vectors or dense (n, n_out)."""
        nfb = (n_out + 31) // 32
        nks = sum(V[0].shape[1] // kpl for V in segs)
        dense = np.zeros((n, nfb * 32), np.float32)
        fb = 0
        while fb < nfb:
            A, bias = im.chunk(state["ci"])       # a chunk = fbc consecutive 32-row blocks, then the bias fragment
            state["ci"] += 1
            assert A.shape[0] % nks == 0, (A.shape, nks)
            fbc = A.shape[0] // nks
            for b in range(fbc):
                Dm = np.tile(bias[None, b * 32:(b + 1) * 32], (n, 1)).astype(np.float32)        # (n, 32 rows)
                ks = b * nks
                for V in segs:
                    for k in range(V[0].shape[1] // kpl):
                        for hi in (0, 1):
                            a = A[ks][hi * 32:(hi + 1) * 32]                     # (32 rows i, kpl)
                            bvals = V[hi][:, k * kpl:(k + 1) * kpl]             # (n, kpl)
                            Dm += bvals @ a.T
                        ks += 1
                dense[:, (fb + b) * 32:(fb + b + 1) * 32] = Dm
            fb += fbc
        if relu:
            dense = np.maximum(dense, 0.0)
        if not to_regs:
            return dense[:, :n_out]
        dense = q(dense)
        out = [np.zeros((n, nfb * 16), np.float32) for _ in (0, 1)]
        for hi in (0, 1):
            for fb in range(nfb):
                for r in range(16):
                    out[hi][:, fb * 16 + r] = dense[:, fb * 32 + _row(r, hi)]
        return out

    ex = [q(_embed_lane(pts, hi, 5, 32)) for hi in (0, 1)]
    ed = [q(_embed_lane(viewdirs, hi, 2, 16)) for hi in (0, 1)]
    h = layer([ex], W, True)
    for l in range(1, D):
        h = layer([ex, h], W, True) if l - 1 == skip else layer([h], W, True)
    raw = np.zeros((n, 4 + n_sem + n_inst), np.float32)
    f = layer([h], W, False)                      # plan order: appearance branch first, panoptic heads last
    g = layer([f, ed], W // 2, True)
    raw[:, 0:4] = layer([g, h], 4, False, to_regs=False)
    if n_sem:
        sh = layer([h], W // 2, True)
        raw[:, 4:4 + n_sem] = layer([sh], n_sem, False, to_regs=False)
    if n_inst:
        sh = layer([h], W // 2, True)
        raw[:, 4 + n_sem:] = layer([sh], n_inst, False, to_regs=False)
    assert state["ci"] == im.n_chunks, (state["ci"], im.n_chunks)
    return raw
