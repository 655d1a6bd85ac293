"""Lane-level emulation (numpy, CPU) of the data movement of gemm4_mfma_ps_kernel (bitsandbytes_amd/csrc/gemm4_mfma_ps.hip).

The kernel's correctness rests on index algebra that cannot run in a GPU-less container: the LDS-DMA instructions whose XOR
swizzles are applied on the SOURCE side (lane l of an instruction always writes LDS bytes [16 l, 16 l + 16) of its 1-KiB
piece and fetches the global piece that belongs there), the ring slots of the weight / scale / activation stages, the
ds_read_b128 + v_permlane32_swap that hands every lane the packed weights of its two column tiles, the k order of the
32x32x16 MFMA steps (lane half h of K quarter q: k = 32 q + 16 h + 8 s + 0..7), the four-way split of the accumulator
registers among the K quarters in the epilogue and the K slices. This script replays exactly those formulas per lane
against the HARDWARE semantics (what a 32x32x16 MFMA sums, which lanes one ds_read_b128 pass serves, little-endian byte
order of a dword, v_permlane32_swap) and compares the result with a plain matrix product.

    python tests/checks/emulate_ps_mfma.py        (also imported by tests/test_host_logic.py)
"""
import numpy as np

LUT = 65536
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
READ_GROUPS = READ_GROUPS + [[l + 32 for l in g_] for g_ in READ_GROUPS]


def _check_read_b128(addr):
    """ds_read_b128: the fixed 16-lane groups over 64 banks (16 positions of 16 B)."""
    for grp in READ_GROUPS:
        assert len(set((addr[grp] // 16) % 16)) == 16, "ds_read_b128 bank conflict"


def lds_map(MT, nested=False):
    DA, DW = (2, 3) if MT >= 4 else (3, 4)
    ASB, WSB, SSB = 32 * MT * 256, 128 * 64, 2048 if nested else 1024
    ABase = LUT
    WBase = ABase + DA * ASB
    SBase = WBase + DW * WSB
    Code2 = SBase + DW * SSB
    assert Code2 + 1024 <= 163840
    return dict(DA=DA, DW=DW, ASB=ASB, WSB=WSB, SSB=SSB, ABase=ABase, WBase=WBase, SBase=SBase)


def emulate(M=40, K=768, MT=2, seed=0, sps=None, N=128):
    rng = np.random.default_rng(seed)
    L = lds_map(MT)
    DA, DW = L["DA"], L["DW"]
    code = rng.standard_normal(16)
    nib = rng.integers(0, 16, size=(N, K))
    packed = (nib[:, 0::2] << 4 | nib[:, 1::2]).astype(np.uint8)   # element 2i in the HIGH nibble (reference default/ops.py:256)
    scale = rng.standard_normal((N, K // 64)).astype(np.float32)
    A = rng.standard_normal((M, K))
    a_bytes = np.arange(M * K * 2, dtype=np.int64).reshape(M, K * 2)  # the activation matrix as byte ADDRESSES (2 B / element)

    lanes = np.arange(64)
    n, h = lanes & 31, lanes >> 5
    stages_total = K // 128
    sps = sps or stages_total
    out = np.zeros((32 * MT, N))
    for sb in range(0, stages_total, sps):                       # K slices (added in slice order by the finalize kernel)
        ns = min(sps, stages_total - sb)
        lds_w = {}   # LDS byte address -> (kind, payload): weights: packed byte value; activations: global byte address
        acc = np.zeros((8, 2, MT, 64, 16))                        # [wave][nt][mt][lane][register]

        def issue_w(j, slot):
            sa = sb + min(j, ns - 1)
            for q in range(4):                                    # the four weight loaders
                for t in range(2):
                    row = 16 * (2 * q + t) + (lanes >> 2)
                    col = np.minimum(row, N - 1)
                    src = col * (K // 2) + (((lanes & 3) ^ ((row >> 2) & 3)) << 4) + sa * 64
                    dst = L["WBase"] + slot * L["WSB"] + (2 * q + t) * 1024 + lanes * 16
                    for l in range(64):
                        for b in range(16):
                            lds_w[dst[l] + b] = ("w", packed.reshape(-1)[src[l] + b], sa)
                d = 64 * q + lanes
                col = np.minimum(d >> 1, N - 1)
                blk = (col * K + 64 * (d & 1) + sa * 128) >> 6
                dst = L["SBase"] + slot * L["SSB"] + q * 256 + lanes * 4
                for l in range(64):
                    lds_w[dst[l]] = ("s", scale.reshape(-1)[blk[l]], sa)

        def issue_a(j, slot):
            sa = sb + min(j, ns - 1)
            for q in range(4):
                for t in range(2 * MT):
                    row = 4 * (q + 4 * t) + (lanes >> 4)
                    m = np.minimum(row, M - 1)
                    src = m * K * 2 + (((lanes & 15) ^ (row & 15)) << 4) + sa * 256
                    dst = L["ABase"] + slot * L["ASB"] + (q + 4 * t) * 1024 + lanes * 16
                    for l in range(64):
                        for b in range(16):
                            lds_w[dst[l] + b] = ("a", src[l] + b, sa)

        def read_w(stage, wslot):
            """Iteration `stage`: the packed weights and scales of the stage leave the ring slot; per wavefront the B operand
            values [nt][s][lane][8] they decode to (the kernel spreads look-ups and multiplies over the next two iterations)."""
            res = {}
            sa = sb + min(stage, ns - 1)
            for wave in range(8):
                c, q = wave >> 2, wave & 3
                w_rd = L["WBase"] + (64 * c + lanes) * 64 + ((q ^ ((lanes >> 2) & 3)) << 4) + wslot * L["WSB"]
                _check_read_b128(w_rd)
                w = np.zeros((64, 16), dtype=np.int64)
                for l in range(64):
                    for b in range(16):
                        kind, v, tag = lds_w[w_rd[l] + b]
                        assert kind == "w" and tag == sa, "weight ring slot holds another stage"
                        w[l, b] = v
                # v_permlane32_swap(vdst = dwords 0 / 1, src0 = dwords 2 / 3): vdst of lanes 32-63 <-> src0 of lanes 0-31
                X, Y = w[:, 0:8].copy(), w[:, 8:16].copy()
                X2, Y2 = X.copy(), Y.copy()
                X2[32:] = Y[:32]
                Y2[:32] = X[32:]
                wt = [X2, Y2]                                      # wt[nt][lane] = 8 bytes: dword s = step s
                sc = np.zeros((2, 64))
                for nt in range(2):
                    s_rd = L["SBase"] + ((64 * c + 32 * nt + n) * 2 + (q >> 1)) * 4 + wslot * L["SSB"]
                    for l in range(64):
                        kind, v, tag = lds_w[s_rd[l]]
                        assert kind == "s" and tag == sa
                        sc[nt, l] = v
                bvals = np.zeros((2, 2, 64, 8))
                for nt in range(2):
                    for s in range(2):
                        for b in range(4):                         # byte b of dword s: (hi nibble, lo nibble) = elements 2b, 2b + 1
                            byte = wt[nt][:, 4 * s + b]
                            bvals[nt, s, :, 2 * b] = code[byte >> 4] * sc[nt]
                            bvals[nt, s, :, 2 * b + 1] = code[byte & 15] * sc[nt]
                res[wave] = bvals
            return res

        def read_a(stage, aslot):
            res = {}
            sa = sb + min(stage, ns - 1)
            for wave in range(8):
                q = wave & 3
                avals = np.zeros((2, MT, 64, 8))
                for s in range(2):
                    a_rd = L["ABase"] + n * 256 + (((4 * q + 2 * h + s) ^ (n & 15)) << 4) + aslot * L["ASB"]
                    for mt in range(MT):
                        addr = a_rd + mt * 8192
                        _check_read_b128(addr)
                        for l in range(64):
                            for e in range(8):
                                kind, g0, tag = lds_w[addr[l] + 2 * e]
                                assert kind == "a" and tag == sa, "activation ring slot holds another stage"
                                mm, kk = divmod(g0 // 2, K)
                                avals[s, mt, l, e] = A[mm, kk]
                res[wave] = avals
            return res

        def mfma(wave, bvals, avals):
            for s in range(2):
                for nt in range(2):
                    for mt in range(MT):
                        # v_mfma_f32_32x32x16: C[row, col] += sum over h, e of A[lane (row, h)][e] * B[lane (col, h)][e];
                        # register i of lane (col, h') holds row (i & 3) + 8 (i >> 2) + 4 h'
                        Am = avals[s, mt].reshape(2, 32, 8)        # [h][row][e]
                        Bm = bvals[nt, s].reshape(2, 32, 8)        # [h][col][e]
                        C = np.einsum("hre,hce->rc", Am, Bm)
                        for l in range(64):
                            for i in range(16):
                                acc[wave, nt, mt, l, i] += C[(i & 3) + 8 * (i >> 2) + 4 * (l >> 5), l & 31]

        # ---- start-up: the weight loaders request their first DW - 1 stages
        for j in range(DW - 1):
            issue_w(j, j)
        # ---- the pipeline: iteration i reads the weights of stage i and the activations of stage i - 2 from the LDS and
        # multiplies stage i - 3; ring slots: weight stage i in wslot, activation stage i - 2 in aslot
        wslot, aslot = 0, DA - 2
        bq, aq = {}, {}
        for i in range(ns + 3):
            if i - 1 + DW < ns:
                issue_w(i - 1 + DW, (wslot - 1) % DW)
            ja = i - 3 + DA
            if 0 <= ja < ns:
                issue_a(ja, (aslot - 1) % DA)
            if i >= 3:
                for wave in range(8):
                    mfma(wave, bq[i - 3][wave], aq[i - 3][wave])
                del bq[i - 3], aq[i - 3]
            if i < ns:
                bq[i] = read_w(i, wslot)
            if 2 <= i <= ns + 1:
                aq[i - 2] = read_a(i - 2, aslot)
            wslot, aslot = (wslot + 1) % DW, (aslot + 1) % DA

        # ---- epilogue: wavefront (c, o) owns flat registers [8 MT o, 8 MT (o + 1)) of the column group and adds the four
        # K quarters in the order q = 0, 1, 2, 3
        RS = 8 * MT
        for c in range(2):
            for o in range(4):
                for f in range(o * RS, (o + 1) * RS):
                    nt, mt, i = (f // 16) // MT, (f // 16) % MT, f % 16
                    for l in range(64):
                        v = 0.0
                        for qq in range(4):
                            v += acc[4 * c + qq, nt, mt, l, i]
                        m = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5)
                        out[m, 64 * c + 32 * nt + (l & 31)] += v

    W = code[nib] * np.repeat(scale, 64, axis=1)
    ref = A @ W.T
    return float(np.abs(out[:M] - ref).max() / np.abs(ref).max())


if __name__ == "__main__":
    for (M, K, MT, sps) in ((40, 768, 2, None), (64, 1024, 2, 3), (17, 512, 2, None), (100, 640, 4, 2)):
        print(M, K, MT, sps, emulate(M=M, K=K, MT=MT, seed=M, sps=sps))
