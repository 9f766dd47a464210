# validate the per-lane perm rule for the blur's reflected edge columns
def reflect(p, n):
    if n == 1: return 0
    while p < 0 or p >= n: p = -p if p < 0 else 2*n-2-p
    return p
def check(w, pitch):
    import random
    row = [random.randrange(256) for _ in range(pitch)]
    for x0 in range(0, w, 4):
        interior = x0 >= 4 and x0 + 7 <= w - 1
        ofs = 4 if x0 == 0 else 0
        start = x0 - 4 + ofs
        L = []
        for t in range(3):
            a = start + 4*t
            if a + 4 <= pitch and a >= 0: L.append(row[a:a+4])
            else: L.append([None]*4)          # not loaded (predicated to 0)
        D = []
        for t in range(3):
            qs = []
            for k in range(4):
                p = x0 - 4 + 4*t + k
                s = reflect(p, w)
                q = s - start
                qs.append(q if 0 <= q <= 11 else None)
            mapped = [q for q in qs if q is not None]
            useA = (min(mapped) <= 3) if mapped else True
            out = []
            for q in qs:
                if q is None: out.append(0); continue
                if useA:
                    out.append((L[0]+L[1])[q] if q <= 7 else 0)
                else:
                    out.append((L[1]+L[2])[q-4] if q >= 4 else 0)
            D.append(out)
        win = D[0]+D[1]+D[2]
        for i in range(4):
            x = x0 + i
            if x >= w: continue
            for tap in range(-3, 4):
                want = row[reflect(x+tap, w)]
                got = win[4 + i + tap]
                if got is None or got != want:
                    return (w, x0, x, tap, got, want)
    return None
bad = 0
for w in range(16, 700):
    for pitch in {((w+63)//64)*64, w if w % 4 == 0 else ((w+3)//4)*4}:
        r = check(w, pitch)
        if r: bad += 1; print("FAIL", r, pitch)
        if bad > 5: raise SystemExit
print("ok")
