"""Executable model of bellman_amd/csrc/fft.hip's pass/tile index math (ntt_pass_kernel +
plan_passes + ntt_run), over any prime field with a 2^k-th root of unity.  Lets the CPU-only
test-suite check the decomposition against the restated serial_fft before a GPU is involved."""


def plan_passes(log_n, log_tile=10, max_r=8):
    if log_n <= log_tile:
        return [log_n]
    l = -(-log_n // max_r)
    q, rem = divmod(log_n, l)
    return [q + (1 if i < rem else 0) for i in range(l)]


def brev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def ntt_model(data, mod, omega, log_n, inverse=False, pre=None, post=None, post_const=None,
              log_tile=10, max_r=8):
    n = 1 << log_n
    tw = [pow(omega, i, mod) for i in range(n)]
    r = plan_passes(log_n, log_tile, max_r)
    L = len(r)
    assert L <= 4
    cur = list(data)
    scratch = [None] * n
    s = 0
    for p in range(L):
        last = p == L - 1
        src = cur if p == 0 else scratch
        dst = cur if last else scratch
        if L == 1:
            dst = cur
        rp = r[p]
        R = 1 << rp
        log_c = log_tile - rp
        if L == 1:
            log_c = 0
        elif last:
            log_c = min(log_c, r[0])
        else:
            log_c = min(log_c, log_n - s - rp)
        C = 1 << log_c
        tiles = n >> (rp + log_c)
        out_writes = {}
        for t in range(tiles):
            jp0 = 0
            if not last:
                logM = log_n - s - rp
                tpb = (1 << logM) >> log_c
                kprefix, rem = divmod(t, tpb)
                jp0 = rem << log_c
                base = (kprefix << (log_n - s)) + jp0
                irs, ics = 1 << logM, 1
                ob, ors, ocs = base, irs, 1
            elif L == 1:
                base, irs, ics = 0, 1, 0
                ob, ors, ocs = 0, 1, 0
            else:
                groups = (1 << r[0]) >> log_c
                mid, g = divmod(t, groups)
                k00 = g << log_c
                M0 = n >> r[0]
                base = k00 * M0 + (mid << rp)
                irs, ics = 1, M0
                rev = 0
                if L == 3:
                    rev = mid
                elif L == 4:
                    k1, k2 = mid >> r[2], mid & ((1 << r[2]) - 1)
                    rev = k1 + (k2 << r[1])
                ob = k00 + (rev << r[0])
                ors, ocs = 1 << (log_n - rp), 1
            twl = []
            for i in range(R >> 1):
                e = i << (log_n - rp)
                if inverse:
                    e = (n - e) & (n - 1)
                twl.append(tw[e])
            tile = [[0] * R for _ in range(C)]
            for e in range(R << log_c):
                if not last:
                    row, col = e >> log_c, e & (C - 1)
                else:
                    col, row = e >> rp, e & (R - 1)
                g = base + row * irs + col * ics
                v = src[g]
                if pre is not None and p == 0:
                    v = v * pre[g] % mod
                tile[col][brev(row, rp)] = v
            for st in range(rp):
                m = 1 << st
                for b in range((R >> 1) << log_c):
                    col, bb = b >> (rp - 1), b & ((R >> 1) - 1)
                    j, k = bb & (m - 1), bb >> st
                    r1 = (k << (st + 1)) | j
                    r2 = r1 + m
                    x, y = tile[col][r1], tile[col][r2] * twl[j << (rp - 1 - st)] % mod
                    tile[col][r1], tile[col][r2] = (x + y) % mod, (x - y) % mod
            for e in range(R << log_c):
                row, col = e >> log_c, e & (C - 1)
                v = tile[col][row]
                g = ob + row * ors + col * ocs
                if not last:
                    ex = (((jp0 + col) * row) << s) & (n - 1)
                    if inverse:
                        ex = (n - ex) & (n - 1)
                    v = v * tw[ex] % mod
                elif post is not None:
                    v = v * post[g] % mod
                elif post_const is not None:
                    v = v * post_const % mod
                assert g not in out_writes
                out_writes[g] = v
        assert len(out_writes) == n
        for g, v in out_writes.items():
            dst[g] = v
        s += rp
    return cur
