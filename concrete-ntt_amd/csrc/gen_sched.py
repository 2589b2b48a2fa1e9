#!/usr/bin/env python3
"""Pass schedules of the LDS-resident NTT kernels (single source of truth).

A transform of N = 2^LOGN points is executed by TPP = N / E threads (E = 2^LOGE coefficients per
thread) as a short list of *passes*.  In a pass every thread holds the E coefficients whose index
bits listed in RMASK vary (the other LOGN-LOGE index bits are the thread's id inside the
polynomial), and performs the butterfly stages whose index bit is in GMASK (a subset of RMASK)
entirely in registers.  Between passes the polynomial is exchanged through LDS.  The first pass
reads global memory and the last pass writes it directly, so RMASK of the first/last pass decides
how well those accesses coalesce.

Stage <-> index bit: the forward (Cooley-Tukey, src/prime64/shoup.rs:544-615) stage s pairs
coefficients that differ in bit b = LOGN-1-s and runs s = 0..LOGN-1 (b descending); the inverse
(Gentleman-Sande, src/prime64/shoup.rs:1306-1377) visits the same bits ascending.  In both, the
butterfly of element e on bit b uses table entry (1 << (LOGN-1-b)) + (e >> (b+1)).

This file emits sched_gen.inc (C++ specialisations) and, with --check, replays every schedule in
pure Python against a direct evaluation of the transform, so the index logic is validated without
a GPU.  It also scores LDS bank conflicts of each exchange for a candidate XOR swizzle using the
per-instruction lane groups of MI355X_MICROARCH.md (LDS section).
"""
import argparse
import itertools
import os
import random
import sys

MAX_LDS_BYTES = 128 * 1024  # one polynomial per workgroup must fit (160 KiB LDS per CU)


def bits_of(mask):
    return [b for b in range(32) if (mask >> b) & 1]


def pdep(x, mask):
    out, k = 0, 0
    for b in bits_of(mask):
        out |= ((x >> k) & 1) << b
        k += 1
    return out


class Sched:
    def __init__(self, bits, logn, inv, loge, passes, swz, block, pad=(0, 0)):
        self.bits, self.logn, self.inv, self.loge = bits, logn, inv, loge
        self.passes = passes  # list of (rmask, gmask)
        self.swz = swz        # list of (shift, mask, lshift) XOR terms applied to the element index
        self.pad = pad        # (shift, words): e + (e >> shift) * words -- padded layout instead of a swizzle (family 2)
        self.tpp = 1 << (logn - loge)
        self.block = block
        self.ppb = max(1, block // self.tpp)

    def phys(self, e):
        out = e
        for sh, m, l in self.swz:
            out ^= ((e >> sh) & m) << l
        if self.pad[1]:
            out += (e >> self.pad[0]) * self.pad[1]
        return out


def default_loge(bits, logn):
    loge = 4
    while (1 << (logn - loge)) > 1024:
        loge += 1
    return min(loge, logn)


def make_passes(bits, logn, inv, loge, first_x=None):
    """Forward pass list (top bits first); the inverse is its mirror."""
    full = (1 << logn) - 1
    if logn <= loge:
        return [(full, full)]
    v = {64: 1, 32: 2}[bits]
    npass = -(-logn // loge)
    slack = npass * loge - logn
    if first_x is None:
        first_x = min(v, slack)
    passes = []
    hi = logn  # exclusive upper bit of the not-yet-done stage bits
    for k in range(npass):
        if k == 0:
            g = loge - first_x
            gm = ((1 << g) - 1) << (hi - g)
            rm = gm | ((1 << first_x) - 1)
        elif k == npass - 1:
            g = hi
            gm = (1 << g) - 1
            x = loge - g
            rm = gm | (((1 << x) - 1) << (logn - x))
        else:
            g = loge
            gm = ((1 << g) - 1) << (hi - g)
            rm = gm
        assert bin(rm).count("1") == loge and g > 0, (bits, logn, k, bin(rm))
        passes.append((rm, gm))
        hi -= g
    assert hi == 0
    return passes[::-1] if inv else passes


# hand-tuned overrides: (bits, logn, inv) -> dict(first_x=.., swz=[..], block=..)
OVERRIDES = {
    # found by an exhaustive search over two-term XOR swizzles (score(): every exchange of the three passes
    # plus the I/O transpose of the persistent kernel is conflict-free except one 2-way ds_write_b128)
    (64, 10, False): {"swz": [(3, 3, 1), (6, 7, 2)]},
    (64, 10, True): {"swz": [(3, 3, 1), (6, 7, 2)]},
    # (round 3 tried the padded layout of family 2 -- {"pad": (5, 2)} -- for (64, 10) and (64, 11): 9.5 % fewer VALU
    # instructions and 117 instead of 162 VGPRs in the N = 1024 forward kernel, and no change in its speed on the same box
    # (271.4 / 273.8 vs 272.6 / 273.7 us, inverse 0.4 % slower): the package power cap does not care about v_xor.  Not adopted.)
    # u32 at the native64 / native_binary64 sizes: 32 coefficients per thread (same 32 data VGPRs as u64 x 16),
    # so N=2048 lives in one wavefront and N=4096 in two, with 16-byte coalesced first/last accesses
    (32, 11, False): {"loge": 5},
    (32, 11, True): {"loge": 5},
    (32, 12, False): {"loge": 5},
    (32, 12, True): {"loge": 5},
}


def default_swz(bits, logn, loge, passes):
    """XOR the thread-id bits that sit above bit 6 into bits [2..5): decorrelates the strided
    exchange reads (see score())."""
    if logn <= 6:
        return []
    terms = []
    hi_bits = min(3, logn - 6)
    terms.append((6, (1 << hi_bits) - 1, 2))
    return terms


# FAM 2 ("wave blocks", 64-bit words, N = 4096 ... 16384): the top LOGN-10 stages run in ONE pass on registers whose
# twiddles are all wave-uniform, then the polynomial crosses LDS once (the only exchange that needs a workgroup barrier)
# and every wavefront owns a contiguous 1024-point block on which it runs the N = 1024 schedule with wave-private
# exchanges.  The inverse is the mirror image.
BLK_LOGN = 10
BLK_PASSES = [(0x381, 0x380), (0x78, 0x78), (0x207, 0x7)]   # the (64, 10) forward schedule
# LDS layout of family 2: PADDED, e + (e >> 5) * 2 (two words behind every 32: 6.25 % more LDS, 136 KiB at N = 16384).
# Unlike an XOR swizzle it is additive -- phys(thread part | register part) = phys(thread part) + phys(register part) -- so
# the per-register part of every exchange address is a compile-time constant in the DS instruction's immediate offset
# instead of one v_xor per access (-140 of 2640 VALU instructions per thread in the N = 16384 kernel; measured +2 %,
# profiles/r03_blk_lab_ablation.txt).  Conflict score (score() below): three 2-way conflicts over the ten access sets of a
# transform against one with the (64, 10) swizzle [(3, 3, 1), (6, 7, 2)]; found by exhaustive search over one- and two-level
# paddings with at most 20 % overhead.
BLK_PAD = (5, 2)


# 32-bit words (round 4), N = 8192 ... 32768: 2048-word blocks, 32 coefficients per thread -- the (32, 11) schedule inside a
# wavefront, the top LOGN - 11 stages before it.  Padding (7, 4): four words behind every 128 (3 % more LDS: 132 KiB at N = 32768),
# a multiple of four words so that 16-byte accesses stay aligned; score() 13 over the twelve access sets of both directions.
BLK32_LOGN = 11
BLK32_PASSES = [(0x703, 0x700), (0xf8, 0xf8), (0x607, 0x7)]   # the (32, 11) forward schedule
BLK32_PAD = (7, 4)


def fam2_supported(bits, logn):
    return (bits == 64 and 12 <= logn <= 14) or (bits == 32 and 13 <= logn <= 15)


def make_sched_fam2(bits, logn, inv):
    assert fam2_supported(bits, logn)
    if bits == 32:
        top = logn - BLK32_LOGN
        gm0 = ((1 << top) - 1) << BLK32_LOGN
        # the other 5 - top register bits: bits 0, 1 (16-byte accesses), then the bit just below the block boundary
        extra = [0, 1, 10][:5 - top]
        rm0 = gm0 | sum(1 << b for b in extra)
        passes = [(rm0, gm0)] + BLK32_PASSES
        if inv:
            passes = passes[::-1]
        return Sched(bits, logn, inv, 5, passes, [], 1 << (logn - 5), BLK32_PAD)
    top = logn - BLK_LOGN                       # stages of the first pass
    gm0 = ((1 << top) - 1) << BLK_LOGN
    # the other 4 - top register bits: bit 0 (16-byte accesses), then the bits just below the block boundary
    extra = [0, 9][:4 - top]
    rm0 = gm0 | sum(1 << b for b in extra)
    passes = [(rm0, gm0)] + BLK_PASSES
    if inv:
        passes = passes[::-1]
    tpp = 1 << (logn - 4)
    return Sched(bits, logn, inv, 4, passes, [], tpp, BLK_PAD)


# FAM 3 (32-bit words, round 4): the passes of family 1 (16 coefficients per thread wherever the size allows) with a PADDED
# exchange layout instead of the XOR swizzle -- for the whole-product kernels of the native plans (native_fused.hpp), which
# are bound by their VALU instruction count: the swizzled address of every 4-byte LDS access costs a shift / xor / or
# triple (1950 of 15500 instructions per thread in the N = 4096 native64 product), the padded one is one base per pass plus
# the DS instruction's immediate offset.  Paddings by exhaustive search over (shift, words) with score() (total over both
# directions, next to the swizzle's): logn 5..8 conflict-free like the swizzle; 9: 10 vs 9; 10: 10 vs 9; 11, 12: 9 vs 9;
# 13, 14: 14 vs 13; 15: 8 vs 9.  Every padding is a multiple of four words behind a multiple of four: 16-byte accesses stay aligned.
FAM3_PAD = {5: (4, 4), 6: (4, 4), 7: (4, 4), 8: (4, 4), 9: (7, 8), 10: (6, 4), 11: (6, 4), 12: (6, 4), 13: (7, 8), 14: (6, 4),
            15: (5, 4)}


def fam3_supported(bits, logn):
    return bits == 32 and logn in FAM3_PAD


def make_sched(bits, logn, inv, fam=0):
    """fam 0: the tuned schedule; fam 1: the 16-coefficients-per-thread schedule where fam 0 overrides LOGE
    (the whole-polymul kernels keep K residue tiles in registers and cannot afford 32 per thread);
    fam 2: wave blocks (make_sched_fam2); fam 3: family 1 with a padded exchange layout."""
    if fam == 2:
        return make_sched_fam2(bits, logn, inv)
    if fam == 3:
        assert fam3_supported(bits, logn)
        b = make_sched(bits, logn, inv, 1)
        return Sched(bits, logn, inv, b.loge, b.passes, [], b.block, FAM3_PAD[logn])
    ov = OVERRIDES.get((bits, logn, inv), {})
    if fam == 1:
        ov = {k: v for k, v in ov.items() if k != "loge"}
    loge = ov.get("loge", default_loge(bits, logn))
    passes = ov.get("passes") or make_passes(bits, logn, inv, loge, ov.get("first_x"))
    pad = ov.get("pad", (0, 0))
    swz = [] if pad[1] else ov.get("swz", default_swz(bits, logn, loge, passes))
    tpp = 1 << (logn - loge)
    block = ov.get("block", max(tpp, 256 if tpp <= 256 else tpp))
    return Sched(bits, logn, inv, loge, passes, swz, block, pad)


def supported(bits, logn):
    lo = 4 if bits == 64 else 5  # try_new gates: src/prime64.rs:709, src/prime32.rs:635
    return lo <= logn and (1 << logn) * (bits // 8) <= MAX_LDS_BYTES


# --------------------------------------------------------------------------------------------
# pure-Python replay (exact arithmetic) -- validates the schedule/index logic
# --------------------------------------------------------------------------------------------
def replay(s, a, p, table):
    """table[i]: twid (fwd) or inv_twid (inv), laid out as the reference."""
    n = 1 << s.logn
    data = list(a)
    for (rm, gm) in s.passes:
        cm = ((1 << s.logn) - 1) & ~rm
        new = list(data)
        for tid in range(s.tpp):
            ebase = pdep(tid, cm)
            E = 1 << s.loge
            regs = [data[ebase | pdep(j, rm)] for j in range(E)]
            gbits = bits_of(gm)
            order = gbits if s.inv else gbits[::-1]
            rbits = bits_of(rm)
            for b in order:
                k = rbits.index(b)
                for j in range(E):
                    if (j >> k) & 1:
                        continue
                    j1 = j | (1 << k)
                    e = ebase | pdep(j, rm)
                    w = table[(1 << (s.logn - 1 - b)) + (e >> (b + 1))]
                    x, y = regs[j], regs[j1]
                    if s.inv:
                        regs[j], regs[j1] = (x + y) % p, (x - y) * w % p
                    else:
                        t = y * w % p
                        regs[j], regs[j1] = (x + t) % p, (x - t) % p
            for j in range(E):
                new[ebase | pdep(j, rm)] = regs[j]
        data = new
    return data


def direct(logn, inv, a, p, table):
    n = 1 << logn
    a = list(a)
    if not inv:
        t, m = n, 1
        while m < n:
            t //= 2
            for i in range(m):
                w = table[m + i]
                for j in range(2 * i * t, 2 * i * t + t):
                    u, v = a[j], a[j + t] * w % p
                    a[j], a[j + t] = (u + v) % p, (u - v) % p
            m *= 2
    else:
        t, m = 1, n
        while m > 1:
            m //= 2
            for i in range(m):
                w = table[m + i]
                for j in range(2 * i * t, 2 * i * t + t):
                    u, v = a[j], a[j + t]
                    a[j], a[j + t] = (u + v) % p, (u - v) * w % p
            t *= 2
    return a


# --------------------------------------------------------------------------------------------
# LDS bank-conflict score of an exchange (MI355X_MICROARCH.md, LDS table)
# --------------------------------------------------------------------------------------------
B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def lane_groups(kind, nbytes):
    if kind == "read":
        if nbytes <= 8:
            return [list(range(0, 32)), list(range(32, 64))], (32 if nbytes == 4 else 64)
        return B128_GROUPS, 64
    if nbytes == 4:
        return [list(range(0, 32)), list(range(32, 64))], 32
    if nbytes == 8:
        return [list(range(16 * i, 16 * i + 16)) for i in range(4)], 32
    return [list(range(8 * i, 8 * i + 8)) for i in range(8)], 32


def vec_bits(rm, bits):
    v, cap = 0, {64: 1, 32: 2}[bits]
    while v < cap and (rm >> v) & 1:
        v += 1
    return v


def score(s, rm, kind):
    """Average LDS cycles per wave-instruction relative to conflict-free (1.0 = clean)."""
    cm = ((1 << s.logn) - 1) & ~rm
    v = vec_bits(rm, s.bits)
    nbytes = (s.bits // 8) << v
    groups, nbanks = lane_groups(kind, nbytes)
    tot, cnt = 0.0, 0
    E = 1 << s.loge
    wave_tids = min(64, s.tpp)
    for j in range(0, E, 1 << v):
        for grp in groups:
            per_bank = {}
            for lane in grp:
                tid = lane % wave_tids  # several polynomials per wave: separate LDS regions, same pattern
                poly = lane // wave_tids
                e = pdep(tid, cm) | pdep(j, rm)
                addr = (s.phys(e) + poly * (1 << s.logn)) * (s.bits // 8)
                for d in range(nbytes // 4):
                    bank = ((addr // 4) + d) % nbanks
                    per_bank.setdefault(bank, set()).add((addr // 4) + d)
            worst = max(len(x) for x in per_bank.values())
            tot += worst
            cnt += 1
    return tot / cnt


def report(s):
    out = []
    for k in range(len(s.passes) - 1):
        w = score(s, s.passes[k][0], "write")
        r = score(s, s.passes[k + 1][0], "read")
        out.append((w, r))
    return out


# --------------------------------------------------------------------------------------------
def emit(path):
    lines = ["// GENERATED by gen_sched.py -- do not edit; re-run `python3 gen_sched.py`.",
             "// Sched<BITS, LOGN, INV, FAM>: pass masks and LDS swizzle of the LDS-resident NTT kernels.",
             "// FAM 1 = 16 coefficients per thread wherever FAM 0 uses 32 (falls back to FAM 0 elsewhere).", ""]
    keys = []
    for bits in (64, 32):
        for logn in range(4, 16):
            if not supported(bits, logn):
                continue
            for inv in (False, True):
                keys.append((bits, logn, inv, 0))
                if "loge" in OVERRIDES.get((bits, logn, inv), {}):
                    keys.append((bits, logn, inv, 1))
                if fam2_supported(bits, logn):
                    keys.append((bits, logn, inv, 2))
                if fam3_supported(bits, logn):
                    keys.append((bits, logn, inv, 3))
    for bits, logn, inv, fam in keys:
            if True:
                s = make_sched(bits, logn, inv, fam)
                np_ = len(s.passes)
                swz = list(s.swz) + [(0, 0, 0)] * (2 - len(s.swz))
                lines.append("template <> struct Sched<%d, %d, %s%s> {" % (bits, logn, "true" if inv else "false",
                                                                         ", %d" % fam if fam else ""))
                lines.append("    static constexpr int LOGE = %d, NPASS = %d, BLOCK = %d;" % (s.loge, np_, s.block))
                lines.append("    static constexpr uint32_t RMASK[%d] = {%s};" %
                             (np_, ", ".join("0x%xu" % r for r, _ in s.passes)))
                lines.append("    static constexpr uint32_t GMASK[%d] = {%s};" %
                             (np_, ", ".join("0x%xu" % g for _, g in s.passes)))
                lines.append("    static constexpr uint32_t SWZ_SH0 = %d, SWZ_M0 = 0x%xu, SWZ_L0 = %d;" % swz[0])
                lines.append("    static constexpr uint32_t SWZ_SH1 = %d, SWZ_M1 = 0x%xu, SWZ_L1 = %d;" % swz[1])
                lines.append("    static constexpr uint32_t PAD_SH = %d, PAD_MUL = %d;" % s.pad)
                lines.append("};")
    lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


def check():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))
    import gen_golden as gg
    rnd = random.Random(1)
    for bits, p in ((64, 4611686018427322369), (32, 1062862849)):
        for logn in range(4, 16):
            if not supported(bits, logn):
                continue
            n = 1 << logn
            if n > 4096 and logn not in (14, 15):
                pass
            pl = gg.Plan(n, p, bits)
            a = [rnd.randrange(p) for _ in range(n)]
            for inv, fam in ((False, 0), (True, 0), (False, 1), (True, 1), (False, 2), (True, 2), (False, 3), (True, 3)):
                if fam == 1 and "loge" not in OVERRIDES.get((bits, logn, inv), {}):
                    continue
                if fam == 2 and not fam2_supported(bits, logn):
                    continue
                if fam == 3 and not fam3_supported(bits, logn):
                    continue
                s = make_sched(bits, logn, inv, fam)
                table = pl.inv_twid if inv else pl.twid
                got = replay(s, a, p, table)
                want = direct(logn, inv, a, p, table)
                assert got == want, (bits, logn, inv, fam)
                print("ok bits=%d logn=%2d inv=%d fam=%d loge=%d tpp=%4d block=%4d passes=%s lds(w,r)=%s" % (
                    bits, logn, inv, fam, s.loge, s.tpp, s.block,
                    ["%x/%x" % (r, g) for r, g in s.passes],
                    ["%.1f/%.1f" % wr for wr in report(s)]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    emit(os.path.join(here, "sched_gen.inc"))
    if args.check:
        check()
