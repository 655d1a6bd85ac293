"""Index-algebra check of the 'diagonal MFMA' decode for the dot kernel (numpy, no GPU)."""
import numpy as np
rng = np.random.default_rng(0)
L = 64
def permlane32_swap(a, b):
    a, b = a.copy(), b.copy()
    t = a[32:].copy(); a[32:] = b[:32]; b[:32] = t
    return a, b
def permlane16_swap(a, b):
    # odd rows of a <-> even rows of b
    a, b = a.copy(), b.copy()
    for ra, rb in ((1, 0), (3, 2)):
        t = a[16*ra:16*ra+16].copy(); a[16*ra:16*ra+16] = b[16*rb:16*rb+16]; b[16*rb:16*rb+16] = t
    return a, b
def transpose4(w):  # w: [4][64] (reg d, lane)
    w0, w1, w2, w3 = w
    w0, w2 = permlane32_swap(w0, w2)
    w1, w3 = permlane32_swap(w1, w3)
    w0, w1 = permlane16_swap(w0, w1)
    w2, w3 = permlane16_swap(w2, w3)
    return [w0, w1, w2, w3]
# tag every (lane, dword) with an id
w = [np.array([lane * 4 + d for lane in range(L)]) for d in range(4)]
wt = transpose4(w)
ok = True
for dp in range(4):
    for lane in range(L):
        i, g = lane & 15, lane >> 4
        want = (i + 16 * dp) * 4 + g   # dword g of original lane i + 16 dp
        ok &= wt[dp][lane] == want
print("transpose maps (i,g) reg d' -> dword g of lane i+16d':", ok)

def mfma(A, B):  # A,B: [64][8]; returns D regs [64][4]
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(L):
        Am[l % 16, 8 * (l // 16):8 * (l // 16) + 8] = A[l]
        Bm[8 * (l // 16):8 * (l // 16) + 8, l % 16] = B[l]
    Dm = Am @ Bm
    D = np.zeros((L, 4))
    for l in range(L):
        for r in range(4):
            D[l, r] = Dm[4 * (l // 16) + r, l % 16]
    return D
# full check: one 2048-k segment, one row, M=1
K = 2048
wts = rng.standard_normal(K)          # decoded weights (code values)
x = rng.standard_normal(K)
scale_lane = rng.standard_normal(L)   # one scale per lane's 32-element run
ref = sum(scale_lane[l] * np.dot(wts[32*l:32*l+32], x[32*l:32*l+32]) for l in range(L))
# lane l owns elements 32l..32l+31 as 4 dwords of 8
own = [[wts[32*l + 8*d: 32*l + 8*d + 8] for l in range(L)] for d in range(4)]  # [d][lane] -> 8 values
# transposition acts on dwords; emulate by moving the 8-vectors with the id map
acc = np.zeros((L, 4))
for dp in range(4):
    A = np.zeros((L, 8)); B = np.zeros((L, 8)); s = np.zeros(L)
    for lane in range(L):
        i, g = lane & 15, lane >> 4
        src_lane, src_d = divmod(int(wt[dp][lane]), 4)
        A[lane] = own[src_d][src_lane]
        Lorig = i + 16 * dp
        B[lane] = x[32 * Lorig + 8 * g: 32 * Lorig + 8 * g + 8]   # x chunk index 4*Lorig + g
        s[lane] = scale_lane[(lane & 15) + 16 * dp]                 # bpermute: scale of lane (lane&15)+16dp
    D = mfma(A, B)
    acc += s[:, None] * D
# diagonal extraction: lane (j,g) useful iff g == j>>2, reg j&3
tot = 0.0
for lane in range(L):
    j, g = lane & 15, lane >> 4
    if g == (j >> 2):
        tot += acc[lane, j & 3]
print("diag-MFMA result", tot, "reference", ref, "ok", np.isclose(tot, ref))
