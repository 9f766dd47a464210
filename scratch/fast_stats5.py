"""Per-cell simulation of the survivor-filter strategies of k_fast_cells on the benchmark scene (CPU, oracle pyramid, iniThFAST pass):
scoring rounds and an instruction estimate per cell for: no filter; diagonal test when n > 64 (the kernel); + the two odd ring quads as a
third pass when the list is still just above a multiple of 64."""
import sys, os, math, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
img, _ = pkg.synth.frame_pair(640, 480, 0, 0)
orc = O.OrbOracle(1000, 1.2, 8, 20, 7); orc.extract(img)
ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
th = 20
SCORE, FILT, FILT2 = 110, 21, 30
tot = dict(cells=0, s0=0, s1=0, s2=0, r0=0, r1=0, r2=0)
band = {T: 0 for T in (88, 92, 96, 100, 104, 108, 112)}       # policy: no diagonal test for T < n <= 128
for l in range(8):
    v = orc.level_image(l).astype(np.int32); H, W = v.shape
    c = v[3:-3, 3:-3]
    R = [v[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in ring]
    def quad(i):
        a, b, cc, d = R[i], R[i + 4], R[i + 8], R[i + 12]
        return (np.minimum(np.maximum(a, cc), np.maximum(b, d)) > c + th) | (np.maximum(np.minimum(a, cc), np.minimum(b, d)) < c - th)
    q0, q1, q2, q3 = quad(0), quad(1), quad(2), quad(3)
    A = q0 & q2; C = A & q1 & q3
    # cells like ORBextractor.cc:789-806: detection area [16-3, W-16+3) etc. in level coordinates; arrays above are offset by 3
    minx, miny, maxx, maxy = 16 - 3, 16 - 3, W - 16 + 3, H - 16 + 3
    wid, hei = maxx - minx, maxy - miny
    ncols, nrows = wid // 30, hei // 30
    wc, hc = math.ceil(wid / ncols), math.ceil(hei / nrows)
    for i in range(nrows):
        y0 = miny + i * hc; y1 = min(y0 + hc + 6, maxy)
        if y0 >= maxy - 3: continue
        for j in range(ncols):
            x0 = minx + j * wc; x1 = min(x0 + wc + 6, maxx)
            if x0 >= maxx - 6: continue
            sl = (slice(y0, y1 - 6), slice(x0, x1 - 6))          # detection pixels of the cell (3-px ring inside the ROI), in the offset-3 arrays
            n0, nA, nC = int(q0[sl].sum()), int(A[sl].sum()), int(C[sl].sum())
            r0 = math.ceil(n0 / 64)
            n1 = nA if n0 > 64 else n0
            r1 = math.ceil(n1 / 64); c1 = (FILT * r0 if n0 > 64 else 0)
            # third pass when it can drop a round: 52 / 66 of the entries stay
            n2, c2 = n1, c1
            if n0 > 64 and n1 > 64 and (n1 - 1) % 64 < 17:
                n2 = nC; c2 += FILT2 * r1
            r2 = math.ceil(n2 / 64)
            for T in band: band[T] += (SCORE * r0 if (n0 <= 64 or T < n0 <= 128) else SCORE * math.ceil(nA / 64) + FILT * r0)
            tot["cells"] += 1; tot["r0"] += r0; tot["r1"] += r1; tot["r2"] += r2
            tot["s0"] += SCORE * r0; tot["s1"] += SCORE * r1 + c1; tot["s2"] += SCORE * r2 + c2
n = tot["cells"]
print("cells %d; scoring rounds per cell: none %.2f, diagonal %.2f, + odd quads %.2f; filter + scoring instructions per cell: %.0f / %.0f / %.0f"
      % (n, tot["r0"] / n, tot["r1"] / n, tot["r2"] / n, tot["s0"] / n, tot["s1"] / n, tot["s2"] / n))
print("no diagonal test for T < n <= 128:", {T: round(v / n, 1) for T, v in band.items()})
