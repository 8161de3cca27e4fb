"""Executable model of bellman_amd/csrc/fft.hip (ntt_pass_kernel + ntt_step + plan_passes + ntt_run) over
any prime field with a 2^k-th root of unity: the pass / tile index math, the in-tile radix-8/4/2 register steps
with their master-table twiddle indices, the XCD-aware tile permutation and the two-level tables.  Lets the
CPU-only test-suite check the decomposition against a plain DFT before a GPU is involved; `log_tile` / `max_r`
shrink the tile so that multi-pass plans are reachable at sizes a Python loop can afford."""


def plan_passes(log_n, max_r=11):
    l = 1 if log_n <= max_r else -(-log_n // max_r)
    q, rem = divmod(log_n, l)
    return [q + (1 if i < rem else 0) for i in range(l)]


def brev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def step_groups(r, gmax=3):
    """the stage groups of an r-stage sub-FFT: radix-8 steps, 4 = 2 + 2, then what is left (the 256-thread kernel);
    radix-4 steps, then what is left (gmax = 2: the 512-thread kernel of the one-level tables)"""
    out, left = [], r
    while left:
        if gmax == 2:
            g = 2 if left >= 2 else 1
        else:
            g = 2 if left == 4 else (3 if left >= 3 else left)
        out.append(g)
        left -= g
    return out


def step_is_wave_local(total, log_tile, g, s, threads):
    """fft.hip: a step with one task per thread over a full tile whose task span 2^(s+g) fits a wavefront's block of
    tile / waves positions"""
    waves = max(1, threads // 64)
    log_waves = waves.bit_length() - 1
    return total == (1 << log_tile) and (total >> g) == threads and log_tile >= g + s + log_waves


def ntt_step(tile, total, r, s, g, first, master, master_bits, mod, threads, by_wave=None):
    """ntt_step<G, FIRST>: tile is the flat LDS array indexed col * R + pos.  by_wave: dict wavefront -> set of the
    positions it touched (filled for the barrier check of ntt_model)"""
    m = 1 << s
    ntasks = total >> g
    hi_bits = r - s - g
    touched = set()
    for tid in range(threads):
        task = tid
        while task < ntasks:
            lo = 0 if first else (task & (m - 1))
            rest = task >> s
            hi, col = rest & ((1 << hi_bits) - 1), rest >> hi_bits
            pos0 = (col << r) + (hi << (s + g)) + lo
            idxs = [pos0 + (t << s) for t in range(1 << g)]
            assert not (touched & set(idxs)), "two tasks own one element"
            touched |= set(idxs)
            if by_wave is not None:
                by_wave.setdefault(tid >> 6, set()).update(idxs)
            e = [tile[i] for i in idxs]
            for j in range(g):
                for t in range(1 << g):
                    if t & (1 << j):
                        continue
                    tl = t & ((1 << j) - 1)
                    y = e[t + (1 << j)]
                    if not (first and tl == 0):
                        idx = (tl * m + lo) * ((1 << (master_bits - 1)) >> (s + j))
                        assert 0 <= idx < (1 << (master_bits - 1)), "master table index out of range"
                        y = y * master[idx] % mod
                    e[t], e[t + (1 << j)] = (e[t] + y) % mod, (e[t] - y) % mod
            for i, v in zip(idxs, e):
                tile[i] = v
            task += threads
    assert len(touched) == total


def ntt_model(data, mod, omega, log_n, inverse=False, pre_g=None, post_g=None, post_scale=1, post_const=None,
              log_tile=11, max_r=11, threads=256, gmax=3, barriers_skipped=None):
    """omega: primitive 2^log_n-th root.  pre_g: multiply input i by pre_g^i (coset_fft); post_g / post_scale:
    multiply output k by post_scale * post_g^k (icoset_fft); post_const: multiply every output (ifft's 1/n)."""
    n = 1 << log_n
    r = plan_passes(log_n, max_r)
    L = len(r)
    assert L <= 3 and max(r) <= max_r <= log_tile
    w = pow(omega, mod - 2, mod) if inverse else omega
    # master table: w_{2^log_tile}^i, i < 2^(log_tile-1), derived from the domain root of unity of that size
    master_bits = log_tile
    if log_n >= master_bits:
        wm = pow(w, 1 << (log_n - master_bits), mod)
    else:   # the master root is a 2^master_bits-th root whose 2^(master_bits - log_n)-th power is w
        wm = None
    lb = (log_n + 1) // 2
    mask = (1 << lb) - 1

    def two_level(base, scale=1):
        lo = [scale * pow(base, i, mod) % mod for i in range(1 << lb)]
        step = pow(base, 1 << lb, mod)
        hi = [pow(step, i, mod) for i in range(1 << (log_n - lb))]
        return lo, hi

    tw_lo, tw_hi = two_level(w) if L > 1 else (None, None)
    pre = two_level(pre_g) if pre_g is not None else None
    post = two_level(post_g, post_scale) if post_g is not None else None
    cur = list(data)
    scratch = [None] * n
    s = 0
    for p in range(L):
        last = p == L - 1
        src = cur if p == 0 else scratch
        dst = cur if last else scratch
        rp = r[p]
        R = 1 << rp
        # in-tile twiddles: entry i of the master table is w_R^(i * R / 2^master_bits) - only multiples are used
        if wm is not None:
            master = [pow(wm, i, mod) for i in range(1 << (master_bits - 1))]
        else:
            wr = pow(w, 1 << (log_n - rp), mod) if rp else 1   # w_R
            scale_ = 1 << (master_bits - rp)
            master = [pow(wr, i // scale_, mod) if i % scale_ == 0 else None for i in range(1 << (master_bits - 1))]
        log_c = log_tile - rp
        if L == 1:
            log_c = 0
        elif last:
            log_c = min(log_c, r[0])
        else:
            log_c = min(log_c, log_n - s - rp)
        C = 1 << log_c
        tiles = n >> (rp + log_c)
        swizzle = tiles >= 64 and tiles % 8 == 0
        seen_tiles = set()
        out_writes = {}
        for b in range(tiles):
            t = (b & 7) * (tiles >> 3) + (b >> 3) if swizzle else b
            assert t not in seen_tiles
            seen_tiles.add(t)
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
                mid, k00 = divmod(t, groups)
                k00 <<= log_c
                M0 = n >> r[0]
                base = k00 * M0 + (mid << rp)
                irs, ics = 1, M0
                ob = k00 + (mid << r[0])
                ors, ocs = 1 << (log_n - rp), 1
            total = R << log_c
            tile = [None] * total
            for e in range(total):
                if not last:
                    row, col = e >> log_c, e & (C - 1)
                else:
                    col, row = e >> rp, e & (R - 1)
                g = base + row * irs + col * ics
                v = src[g]
                if p == 0 and pre is not None:
                    v = v * pre[1][g >> lb] % mod * pre[0][g & mask] % mod
                tile[(col << rp) + brev(row, rp)] = v
            sbits, first = 0, True
            prev_local, prev_g, prev_sets = False, 0, None
            for g_ in step_groups(rp, gmax):
                local = step_is_wave_local(total, log_tile, g_, sbits, threads)
                sets = {}
                ntt_step(tile, total, rp, sbits, g_, first, master, master_bits, mod, threads, by_wave=sets)
                if local and prev_local and g_ == prev_g:
                    # the kernel has no s_barrier between these two steps: every wavefront must stay inside what it
                    # alone touched in the previous step
                    assert sets == prev_sets, "a step without a barrier crosses wavefronts"
                    if barriers_skipped is not None:
                        barriers_skipped.append((rp, sbits))
                sbits += g_
                first = False
                prev_local, prev_g, prev_sets = local, g_, sets
            for e in range(total):
                row, col = e >> log_c, e & (C - 1)
                v = tile[(col << rp) + row]
                g = ob + row * ors + col * ocs
                if not last:
                    ex = (((jp0 + col) * row) << s) & (n - 1)
                    if ex:
                        v = v * tw_hi[ex >> lb] % mod * tw_lo[ex & mask] % mod
                elif post is not None:
                    v = v * post[1][g >> lb] % mod * post[0][g & mask] % mod
                elif post_const is not None:
                    v = v * post_const % mod
                assert g not in out_writes
                out_writes[g] = v
        assert len(out_writes) == n
        for g, v in out_writes.items():
            dst[g] = v
        s += rp
    return cur
