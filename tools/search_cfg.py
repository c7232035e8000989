#!/usr/bin/env python3
"""Enumerate (P, XG, Q, CH, NT) kernel shapes for a window and rank them by a simple
critical-path cost model (pk_fma slots per CU per window).  Tuning aid only."""
import itertools, math, sys
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 257
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 257
L = int(sys.argv[3]) if len(sys.argv) > 3 else 65
SYM = 0.75  # row-pass cost factor if the symmetric-tap form is used (98/130)
rows = []
for P, XG, Q, CH, NT in itertools.product(range(4, 17), (2, 4, 6, 8, 10, 12, 16), (4, 8, 12, 16, 24, 32), (8, 16, 24, 32, 48, 64), (128, 256, 512)):
    if CH % Q and Q % CH: pass
    if CH % 4 or Q % 4 or Q > CH: continue
    TW = P * XG
    if TW > 160: continue
    slack = 0 if (CH % Q == 0 and (L - 1) % Q == 0) else Q - 1
    RR = -(-(CH + L - 1 + slack) // 4) * 4
    lds = CH * ((TW + L - 1) | 1) * 4 + RR * (TW | 1) * 8
    if lds > 160 * 1024: continue
    NW = NT // 64
    wgs = min(160 * 1024 // lds, 32 // NW, 8)
    waves = wgs * NW
    if waves < 8: continue
    nstrips = -(-n2 // TW)
    NA = n1 + L - 1
    nch = -(-NA // CH)
    row_wt = -(-(CH * XG) // 64)            # wave-tasks per chunk
    ws = [min(TW, n2 - s * TW) for s in range(nstrips)]
    col_wt = -(-((CH // Q) * TW) // 64)
    row_rounds = -(-row_wt // NW)
    col_rounds = -(-col_wt // NW)
    # per-WG critical path per chunk (pk_fma issue slots per wave)
    crit = row_rounds * P * L + col_rounds * Q * L
    # total VALU slots per chunk (all waves) — what the CU must execute
    tot = row_wt * P * L + col_wt * Q * L
    col_chunks = -(-n1 // CH)
    per_win_tot = nstrips * (nch * row_wt * P * L + col_chunks * col_wt * Q * L)
    per_win_crit = nstrips * (nch * row_rounds * P * L + col_chunks * col_rounds * Q * L)
    # CU throughput bound: tot / 4 SIMDs ; latency bound: crit * NW / waves-per-SIMD ... take max
    t_cu = per_win_tot / 4.0
    t_lat = per_win_crit * NW / 4.0 / max(1, waves // 4) * (waves / 4.0) / max(1, waves / 4.0)
    ideal = (NA * n2 + n1 * n2) * L / 64 / 4.0
    rows.append((max(t_cu, per_win_crit / max(1, wgs) * 1.0), t_cu, per_win_crit / wgs, ideal / t_cu, P, XG, TW, Q, CH, NT, lds // 1024, wgs, waves, nstrips, row_wt, col_wt))
rows.sort()
print("cost  t_cu  crit/wg  eff   P XG TW  Q CH NT ldsKB wgs waves strips row_wt col_wt")
for r in rows[:40]:
    print("%7.0f %7.0f %7.0f %.3f  %2d %2d %3d %2d %2d %3d %3d %d %2d %d %d %d" % r)
