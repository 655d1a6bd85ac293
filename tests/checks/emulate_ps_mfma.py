"""Lane-level emulation (numpy, CPU) of the data movement of gemm4_mfma_ps_kernel (bitsandbytes_amd/csrc/gemm4_mfma_ps.hip).

The kernel's correctness rests on index algebra that cannot run in a GPU-less container: coalesced weight loads in the
"lane 4r + p" shape, the transposition through the wavefront-private LDS tile, the XOR-swizzled activation chunk buffers (one per K
half, written and read by the half's 4 column-group wavefronts), the k order of the 32x32x16 MFMA steps and the accumulator layout
of the epilogue. This script replays exactly those formulas per lane against the HARDWARE semantics (what a 32x32x16 MFMA sums,
which lanes one ds_write_b128 / ds_read_b128 pass serves, little-endian byte order of a dword) and compares the result with a
plain matrix product.

    python tests/checks/emulate_ps_mfma.py        (also imported by tests/test_host_logic.py)
"""
import numpy as np

A_BASE = 65536
A_BUF = 16384
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
READ_GROUPS = READ_GROUPS + [[l + 32 for l in g_] for g_ in READ_GROUPS]


def _check_write_b128(addr):
    """ds_write_b128: 8 CONTIGUOUS lanes per pass over 32 banks (8 positions of 16 B)."""
    for i in range(0, 64, 8):
        assert len(set((addr[i:i + 8] // 16) % 8)) == 8, "ds_write_b128 bank conflict"


def _check_read_b128(addr):
    """ds_read_b128: the fixed 16-lane groups over 64 banks (16 positions of 16 B)."""
    for grp in READ_GROUPS:
        assert len(set((addr[grp] // 16) % 16)) == 16, "ds_read_b128 bank conflict"


def emulate(M=40, K=768, MT=2, seed=0, sps=None):
    rng = np.random.default_rng(seed)
    N = 128
    code = rng.standard_normal(16)
    nib = rng.integers(0, 16, size=(N, K))
    packed = (nib[:, 0::2] << 4 | nib[:, 1::2]).astype(np.uint8)   # element 2i in the HIGH nibble (reference default/ops.py:256)
    scale = rng.standard_normal((N, K // 64))
    A = rng.standard_normal((M, K))

    lanes = np.arange(64)
    r, pp = lanes >> 2, lanes & 3
    n, h = lanes & 31, lanes >> 5
    arow, apiece = lanes >> 4, lanes & 15
    AI = 2 * MT

    stages_total = K // 256
    sps = sps or stages_total
    out = np.zeros((32 * MT, N))
    for sb in range(0, stages_total, sps):                       # K slices (added in slice order by the finalize kernel)
        ns = min(sps, stages_total - sb)
        acc = np.zeros((8, MT, 64, 16))                           # [wave][mt][lane][register]
        for j in range(ns):
            lds = {}                                              # the two activation buffers, 16-byte granules by byte address
            # ---- decode phase: every wavefront writes its share of its K half's chunk
            for wave in range(8):
                g, q = wave & 3, wave >> 2
                kq = 256 * sb + q * 128 * ns
                for i in range(AI):
                    row_local = 8 * MT * g + 4 * i + arow
                    addr = A_BASE + q * A_BUF + row_local * 256 + ((apiece ^ (row_local & 15)) << 4)
                    _check_write_b128(addr)
                    for l in lanes:
                        m = min(int(row_local[l]), M - 1)
                        k0 = kq + 128 * j + 8 * apiece[l]
                        assert addr[l] not in lds
                        lds[int(addr[l])] = (m, k0)
            assert len(lds) == 2 * 32 * MT * 16
            # ---- (2..5) per wavefront: transposition, decode, MFMA steps
            for wave in range(8):
                g, q = wave & 3, wave >> 2
                kq = 256 * sb + q * 128 * ns
                col0 = 32 * g
                tile = {}
                for i in range(2):
                    row = 16 * i + r
                    t_wr = (row * 4 + (pp ^ ((row >> 2) & 3))) * 16
                    _check_write_b128(t_wr)
                    for l in lanes:
                        kbyte = (kq + 128 * j) // 2 + 16 * pp[l]
                        tile[int(t_wr[l])] = (col0 + int(row[l]), kbyte)      # 16 bytes of row, from byte kbyte
                assert len(tile) == 128
                d = []                                            # d[s][lane] = (weight row, first byte) of dword s
                for i in range(2):
                    t_rd = (n * 4 + ((2 * h + i) ^ ((n >> 2) & 3))) * 16
                    _check_read_b128(t_rd)
                    for dw in range(4):
                        d.append([(tile[int(t_rd[l])][0], tile[int(t_rd[l])][1] + 4 * dw) for l in lanes])
                a_rd = A_BASE + q * A_BUF + n * 256 + (((8 * h) ^ (n & 15)) << 4)
                for s in range(8):
                    # provenance: lane (n, h), step s must hold k [64 h + 8 s, + 8) of column col0 + n
                    for l in lanes:
                        wrow, kbyte = d[s][l]
                        assert wrow == col0 + n[l]
                        assert kbyte * 2 == kq + 128 * j + 64 * h[l] + 8 * s, "weight dword is not the expected k range"
                    Bop = np.zeros((64, 8))
                    for l in lanes:
                        wrow, kbyte = d[s][l]
                        k0 = kbyte * 2
                        byts = packed[wrow, kbyte:kbyte + 4]
                        # one scale per lane and chunk: block of k kq + 128 j + 64 h
                        sc = scale[wrow, (kq + 128 * j + 64 * h[l]) // 64]
                        assert k0 // 64 == (kq + 128 * j + 64 * h[l]) // 64, "MFMA step leaves the lane's quantization block"
                        Bop[l] = [sc * (code[b >> 4] if e == 0 else code[b & 15]) for b in byts for e in (0, 1)]
                    for mt in range(MT):
                        addr = (a_rd ^ (s << 4)) + mt * 8192
                        _check_read_b128(addr)
                        Aop = np.zeros((64, 8))
                        for l in lanes:
                            m, k0 = lds[int(addr[l])]
                            assert m == min(32 * mt + n[l], M - 1), "activation fragment row"
                            assert k0 == kq + 128 * j + 64 * h[l] + 8 * s, "activation fragment k"
                            Aop[l] = A[m, k0:k0 + 8]
                        # MFMA 32x32x16: D[i][c] += sum over halves hh and elements e of Aop[i + 32 hh][e] * Bop[c + 32 hh][e];
                        # register reg of lane (c, hh2) = row (reg & 3) + 8 (reg >> 2) + 4 hh2
                        D = sum(Aop[32 * hh:32 * hh + 32] @ Bop[32 * hh:32 * hh + 32].T for hh in range(2))   # [row i][col c]
                        for l in lanes:
                            for reg in range(16):
                                row = (reg & 3) + 8 * (reg >> 2) + 4 * h[l]
                                acc[wave, mt, l, reg] += D[row, n[l]]
        # ---- epilogue: K half 0 + K half 1 of every column group, register -> (row, column)
        for g in range(4):
            for mt in range(MT):
                for l in lanes:
                    for reg in range(16):
                        v = acc[g, mt, l, reg] + acc[g + 4, mt, l, reg]
                        m = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * h[l]
                        out[m, 32 * g + n[l]] += v
    W = code[nib] * np.repeat(scale, 64, axis=1)
    ref = A @ W.T
    return np.abs(out[:M] - ref).max() / np.abs(ref).max()


if __name__ == "__main__":
    for (M, K, MT, sps) in ((40, 768, 2, None), (64, 1024, 2, 2), (17, 512, 1, None), (32, 1280, 1, 3)):
        e = emulate(M=M, K=K, MT=MT, seed=M, sps=sps)
        print(f"M={M} K={K} MT={MT} sps={sps}: max rel err vs plain product {e:.2e}")
        assert e < 1e-12
    print("ok")
