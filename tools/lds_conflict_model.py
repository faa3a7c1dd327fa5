"""LDS bank-conflict model of the one-wavefront stiffness kernel (stiffness_wave_eo_element, d4est_hip_wave.h) under the gfx950 banking
rules of MI355X_MICROARCH.md (LDS table):
  ds_read_b64 : 2 lane groups of 32 (lanes 0-31, 32-63), bank = (byte address / 4) mod 64, an access covers 2 consecutive banks
  ds_write_b64: 4 lane groups of 16 contiguous lanes,    bank = (byte address / 4) mod 32
Per group: cycles = the largest number of DISTINCT addresses that share a bank (identical addresses broadcast); conflict cycles =
cycles - 1.  usage: python tools/lds_conflict_model.py [N]   (N = NQ = deg + 1 <= 8; one element per 64 lanes)"""
import sys
from collections import defaultdict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NQ, PN, PQ = N, N | 1, N | 1
lanes = [(l % NQ, l // NQ) for l in range(NQ * NQ)]


def cycles(addrs, write):
    groups = [range(g * 16, g * 16 + 16) for g in range(4)] if write else [range(0, 32), range(32, 64)]
    mod = 32 if write else 64
    tot = extra = 0
    for g in groups:
        bank = defaultdict(set)
        for l in g:
            if l < len(addrs) and addrs[l] is not None:
                for d in (0, 1):
                    bank[(2 * addrs[l] + d) % mod].add(addrs[l])
        c = max((len(v) for v in bank.values()), default=0)
        tot += c
        extra += max(c - 1, 0)
    return tot, extra


stages = []   # (name, write?, address(a, b, r) for r in range(count), count)
img = lambda a, b, t: (a + NQ * b + 64 * t) % N + PN * (((a + NQ * b + 64 * t) // N) % N + N * ((a + NQ * b + 64 * t) // (N * N)))
stages.append(("element image: u -> R0[i + PN (j + N k)]", True, img, (N ** 3 + 63) // 64))
stages.append(("S1 read  x[i] = R0[i + PN (a + N b)]", False, lambda a, b, i: i + PN * (a + N * b), N))
stages.append(("S1 write R0/R1[a + PN (iq + NQ b)]  (x2 fields)", True, lambda a, b, q: a + PN * (q + NQ * b), 2 * NQ))
stages.append(("S2 read  R0/R1[j + PN (a + NQ b)]  (x2)", False, lambda a, b, j: (j % N) + PN * (a + NQ * b), 2 * N))
stages.append(("S2 write R0/R1[b + PN (a + NQ jq)]  (x2)", True, lambda a, b, q: b + PN * (a + NQ * (q % NQ)), 2 * NQ))
stages.append(("S3 read  R0/R1[k + PN (a + NQ b)]  (x3 fields)", False, lambda a, b, k: (k % N) + PN * (a + NQ * b), 3 * N))
stages.append(("S3 write R0[b + PN (a + NQ jq)]", True, lambda a, b, q: b + PN * (a + NQ * q), NQ))
stages.append(("S5 write R0/R1[b + PQ (a + NQ k)]  (x2)", True, lambda a, b, k: b + PQ * (a + NQ * (k % N)), 2 * N))
stages.append(("S6 read  R0/R1[jq + PQ (a + NQ b)]  (x3)", False, lambda a, b, q: (q % NQ) + PQ * (a + NQ * b), 3 * NQ))
stages.append(("S6 write R0[b + PQ (a + NQ k)]  (third field)", True, lambda a, b, k: b + PQ * (a + NQ * k), N))
stages.append(("S6 write R0/R1[a + PQ (j + N b)]  (x2)", True, lambda a, b, j: a + PQ * ((j % N) + N * b), 2 * N))
stages.append(("S7 read  R0/R1[iq + PQ (a + N b)]  (x2)", False, lambda a, b, q: (q % NQ) + PQ * (a + N * b), 2 * NQ))
stages.append(("S7 write R0[i + PN (a + N b)]", True, lambda a, b, i: i + PN * (a + N * b), N))
stages.append(("element image: R0 -> A u", False, img, (N ** 3 + 63) // 64))
tr = tw = er = ew = nr = nw = 0
print("N = NQ = %d, padded line length %d doubles; per wavefront and element" % (N, PN))
print("%-52s %6s %12s %16s" % ("access", "instr", "LDS cycles", "conflict cycles"))
for name, wr, f, cnt in stages:
    c = e = 0
    for r in range(cnt):
        t, x = cycles([f(a, b, r) for a, b in lanes], wr)
        c += t; e += x
    print("%-52s %6d %12d %16d" % (name, cnt, c, e))
    if wr: tw += c; ew += e; nw += cnt
    else: tr += c; er += e; nr += cnt
print("reads : %d instructions, %d LDS cycles, %d of them conflict cycles" % (nr, tr, er))
print("writes: %d instructions, %d LDS cycles, %d of them conflict cycles" % (nw, tw, ew))
print("total conflict cycles per wavefront: %d  (x 4096 wavefronts at config 2 = %.2f M per launch)" % (er + ew, (er + ew) * 4096 / 1e6))
