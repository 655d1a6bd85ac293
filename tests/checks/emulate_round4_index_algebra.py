"""CPU replay of two pieces of index algebra added in round 4 (no GPU, numpy only; imported by tests/test_host_logic.py):

1. The K-quarter kernel's accumulator-layout slabs (csrc/gemm4_mfma_kq.hip, epilogue) against gemm4_finalize_kq_kernel's
   decoding of them: every (row, column) of a workgroup tile must be stored exactly once, and the finalize thread that reads a
   16-byte piece must attribute its four values to the rows / column the storing lane held.
2. The peer chain's transport (csrc/gemv4_stream.hip, kPeer): producer slot of an output row -> granule stream -> the consumer's
   16-byte fetch of round r / lane -> the 8-byte half-chunk it writes into the swizzled activation image -> the element a
   decoding lane reads back in load_slice. Every k of x must arrive at the lane and register that multiplies weight k.
"""
import numpy as np


# ----------------------------------------------------------------------------------------------- 1. K-quarter slabs
def kq_store_map(MT):
    """What the kernel stores: {16-byte piece index within the workgroup tile: [(row, col) x 4]}.
    32x32 accumulator layout: register i of lane (n, h) = row (i & 3) + 8 (i >> 2) + 4 h, column n. The wavefront (c, o) owns the
    registers f = o * RS + ch * 4 + k of its 2 * MT tiles (flat: tile f / 16 = nt * MT + mt, register f % 16), RS = 8 MT."""
    RS, CH = 8 * MT, 2 * MT
    pieces = {}
    for c in range(2):
        for o in range(4):
            for ch in range(CH):
                f0 = o * RS + ch * 4
                t, j = f0 // 16, (f0 % 16) // 4
                for lane in range(64):
                    n, h = lane & 31, lane >> 5
                    vals = []
                    for k in range(4):
                        f = f0 + k
                        nt, mt, i = (f // 16) // MT, (f // 16) % MT, f % 16
                        assert (f // 16) == t and (i >> 2) == j
                        vals.append((32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h, 64 * c + 32 * nt + n))
                    # byte offset: soffset c * 8192 MT + voffset lane * 16 + (t * 4 + j) * 1024
                    off = c * 8192 * MT + lane * 16 + (t * 4 + j) * 1024
                    assert off % 16 == 0 and off // 16 not in pieces
                    pieces[off // 16] = vals
    return pieces


def kq_finalize_map(MT, idx):
    """What gemm4_finalize_kq_kernel makes of piece `idx` of workgroup tile 0: [(row, col) x 4]."""
    lane = idx & 63
    r = idx >> 6
    j = r & 3
    r >>= 2
    t = r % (2 * MT)
    r //= 2 * MT
    c = r & 1
    assert r >> 1 == 0
    col = 64 * c + 32 * (t // MT) + (lane & 31)
    row0 = 32 * (t % MT) + 8 * j + 4 * (lane >> 5)
    return [(row0 + k, col) for k in range(4)]


def check_kq_slabs():
    for MT in (1, 2):
        st = kq_store_map(MT)
        assert sorted(st) == list(range(1024 * MT)), "every 16-byte piece of the 16 KiB x MT tile written once"
        seen = set()
        for idx, vals in st.items():
            assert vals == kq_finalize_map(MT, idx), (MT, idx)
            seen.update(vals)
        assert seen == {(r, c) for r in range(32 * MT) for c in range(128)}
    return True


# ----------------------------------------------------------------------------------------------- 2. peer chain transport
def check_peer_chain(K, world, ns, builders=8, waves=16):
    """bf16 / fp16 (2-byte values): CH = 4 16-byte chunks of 8 values per lane and 2048-value segment, swz(l) = (l >> 2) & 3."""
    CH, EPC = 4, 8
    assert world * ns == K and ns % 4 == 0
    swz = lambda l: (l >> 2) & 3  # noqa: E731
    # producer: value v = rank * ns + row sits in granule v >> 1, half v & 1; granules are 8 bytes {2 values, tag}
    granule_of = lambda v: (v >> 1, v & 1)  # noqa: E731
    # consumer: round r, fetching wavefront w (0 .. waves - builders - 1), lane: byte offset of a 16-byte fetch = two granules = four values
    image = {}   # LDS byte address (relative to the image) of an 8-byte write -> the 4 value indices it holds
    for r in range(8):
        for w in range(waves - builders):
            for lane in range(64):
                off = ((r * (waves - builders) + w) * 64 + lane) * 16
                if off >= K * 4:
                    continue
                g0 = off // 8                       # first granule of the fetch
                values = [2 * g0, 2 * g0 + 1, 2 * g0 + 2, 2 * g0 + 3]
                for v in values:
                    assert granule_of(v)[0] in (g0, g0 + 1)
                c, half = off >> 5, (off >> 4) & 1  # 16-byte chunk of x (8 values), which half of it
                sg, cs = c >> 8, c & 255
                lp, q = cs // CH, cs % CH
                addr = ((sg * CH * 64 + CH * lp + (q ^ swz(lp))) * 16 + half * 8)
                assert addr not in image
                image[addr] = values
    assert len(image) == K // 4, "every value written exactly once"
    # decode side (load_slice): wavefront column sw = segment, lane l reads its q-th chunk at slot CH l + (q ^ swz(l)); the lane owns
    # k = seg * 2048 + 32 l + 8 q .. + 8
    S = (K + 2047) // 2048
    for seg in range(S):
        for l in range(64):
            for q in range(CH):
                k0 = seg * 2048 + 32 * l + 8 * q
                if k0 >= K:
                    continue
                slot = CH * l + (q ^ swz(l))
                base = (seg * CH * 64 + slot) * 16
                got = image[base] + image[base + 8]
                assert got == list(range(k0, k0 + 8)), (K, seg, l, q, got[:2], k0)
    return True


if __name__ == "__main__":
    assert check_kq_slabs()
    for (K, world, ns) in ((4096, 1, 4096), (8192, 2, 4096), (11008, 8, 1376), (16384, 4, 4096), (4096, 8, 512)):
        assert check_peer_chain(K, world, ns)
    print("ok")
