"""Analytic LDS bank-conflict check of the ds_read_b128 fragment reads (and the attention V^T ds_write_b32 stores) of the GEMM and attention kernels (no GPU needed): lane groups and
banking as documented for gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, LDS table: ds_read_b128 is serviced in four 16-lane groups
{0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; bank = (addr / 4) mod 64, so a 16-byte access occupies
one of 16 slots of a 256-byte bank row).  Prints LDS cycles per wave instruction: 4 = conflict-free.  python tools/lds_conflicts.py"""
import itertools

GROUPS = [[*range(0, 4), *range(12, 16), *range(20, 28)], [*range(4, 12), *range(16, 20), *range(28, 32)],
          [*range(32, 36), *range(44, 48), *range(52, 60)], [*range(36, 44), *range(48, 52), *range(60, 64)]]


def cycles(addr):
    total = 0
    for g in GROUPS:
        slots = {}
        for lane in g:
            a = addr(lane)
            slots.setdefault((a // 16) % 16, set()).add(a)
        total += max(len(v) for v in slots.values())
    return total


def store_degree(addr, lanes=64):
    """ds_write_b32: two 32-lane groups, bank = (addr / 4) mod 32 -> (worst, mean) lanes per bank over the groups"""
    worst, tot, n = 0, 0, 0
    for g0 in range(0, lanes, 32):
        banks = {}
        for lane in range(g0, g0 + 32):
            a = addr(lane)
            if a is not None:
                banks.setdefault((a // 4) % 32, set()).add(a)
        if banks:
            d = max(len(v) for v in banks.values())
            worst, tot, n = max(worst, d), tot + d, n + 1
    return worst, tot / max(n, 1)


def v_swz2(kb2, vsw2, row, chunk):
    """attention.hip v_swz2"""
    if kb2 == 128:
        return row * 256 + ((chunk ^ (row & 15) ^ (((row >> 3) & 7) if vsw2 else 0)) << 4)
    return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (((row >> 4) & 7) if vsw2 else 0)) << 4)


def vt_report(d, kb2, vsw2):
    """V^T tile of attention.hip: fragment-read cycles, transposing-store conflict degree, and the consistency of the read-side
    shortcut  v_swz2(l16, chunk) ^ VX(i) + i*16*VROW == v_swz2(16 i + l16, chunk)."""
    dc, dvf, vrow = d // 8, (d + 15) // 16, kb2 * 2
    reads = []
    for i in range(dvf):
        for s2 in range(kb2 // 32):
            reads.append(cycles(lambda l: v_swz2(kb2, vsw2, i * 16 + (l & 15), s2 * 4 + (l >> 4))))
            vx = 0 if not vsw2 else (((2 * i) & 7) << 4 if kb2 == 128 else (i & 7) << 4)
            for l in range(64):
                assert (v_swz2(kb2, vsw2, l & 15, s2 * 4 + (l >> 4)) ^ vx) + i * 16 * vrow == v_swz2(kb2, vsw2, i * 16 + (l & 15), s2 * 4 + (l >> 4))
    tasks = (kb2 // 2) * dc
    worst, mean, n = 0, 0.0, 0
    for w0 in range(0, tasks, 64):
        for e in range(8):
            def addr(lane):
                task = w0 + lane
                if task >= tasks:
                    return None
                pair, c = divmod(task, dc)
                key = 2 * pair
                pos = 32 * (key >> 5) + 8 * ((key >> 2) & 3) + 4 * ((key >> 4) & 1) + (key & 3)
                return v_swz2(kb2, vsw2, c * 8 + e, pos >> 3) + (pos & 7) * 2
            w, m = store_degree(addr)
            worst, mean, n = max(worst, w), mean + m, n + 1
    return max(reads), worst, mean / n


def frag(row_bytes, perm):
    """fragment read: lane -> row (lane & 15), 16-byte chunk ks * 4 + (lane >> 4) permuted by perm(row, chunk)"""
    return lambda ks: cycles(lambda l: (l & 15) * row_bytes + (perm(l & 15, ks * 4 + (l >> 4)) << 4))


if __name__ == '__main__':
    gemm = frag(128, lambda r, c: c ^ ((r >> 1) & 7))
    print('GEMM / conv A, W fragments (gemm.hip, gemm_big.hip; 128-byte rows):', [gemm(ks) for ks in range(2)])
    print('attention K, d = 40 / 64 (128-byte rows):', [gemm(ks) for ks in range(2)])
    vt = frag(256, lambda r, c: c ^ (r & 15))
    print('attention V^T, 128-key fills (256-byte rows):', [vt(s) for s in range(4)])
    for d, kb2 in ((40, 128), (64, 128), (80, 64), (160, 64)):
        for vsw2 in (False, True):
            r, w, m = vt_report(d, kb2, vsw2)
            print(f'attention V^T d = {d}, {kb2}-key fills, {"variant 4 (VSW2)" if vsw2 else "default"}: worst read {r} cycles; '
                  f'transposing ds_write_b32 stores {w}-way worst, {m:.2f}-way mean')
    for dp in (96, 160):
        old = frag(dp * 2, lambda r, c: c ^ ((r >> 2) & 3))
        new = frag(dp * 2, lambda r, c: c ^ (((r >> 3) & 1) << 1))
        print(f'attention K, DP = {dp} ({dp * 2}-byte rows): default {[old(k) for k in range(dp // 32)]}  variant 2 {[new(k) for k in range(dp // 32)]}')
        best = min(itertools.product(range(4), repeat=4),
                   key=lambda t: sum(frag(dp * 2, lambda r, c: c ^ t[(r >> 2) & 3])(k) for k in range(dp // 32)))
        print(f'    best per-4-row XOR table found by exhaustive search: {best}')
