#!/usr/bin/env python3
"""Writes kbmod_amd/csrc/search_lds_asm.h: the hand-scheduled summing loops of kb_search_lds as inline-asm statements.

Two families, each for chunks of 8 and of 16 candidates (C), with and without observation counts (COUNT / FAST), for waves
that copy a piece of every staged slab and waves that do not (NP1 / NP0):

  LOOP    one group of epochs (search_lds.h, asm_run): two epochs per trip, the table words of epoch e + 2 fetched BEHIND the
          wait for epoch e's LDS reads, into the registers those reads just consumed -- scalar loads share the LDS counter and
          return out of order, so any wait for them waits for everything: this way they have a whole epoch to land in --,
          two slabs in flight, nothing the compiler schedules while anything is in flight.
  STREAM  a run of whole groups inside one chunk (every group full, staging a full group of the same chunk, an even number
          of epochs per group): the same trip, with the group change -- LDS writes landed, barrier, the read pointer to the
          other buffer, the write pointer back to slot 0 of the one just read -- inside the statement, so that slab loads
          and table words stay in flight across it.

Registers inside a statement (all named in its clobber list, the compiler keeps nothing in them across it):

                                   C = 8        C = 16
  offsets of the even epochs (A)   s[68:75]     s[36:51]      origin/bytes of the slab requested in even epochs (gA)  s[84:87]  s[68:71]
  offsets of the odd epochs (B)    s[76:83]     s[52:67]      ... in odd epochs (gB)                                  s[92:95]  s[72:75]
  address of the slab requested    s[88:89]     s[76:77]      running byte offsets into the two tables               s90, s91  s78, s79
  v[100:103] / v[104:107]  piece of the slab delivered in even / odd epochs (va / vb)
  v[108:123]               eight (psi, phi) samples: a chunk of 16 is summed in two batches of eight
  v124                     address / count temporary
  COUNT only: one register per candidate (v[92:99], C = 16: v[84:99]) as a shift register of the samples' NO_DATA bits: a
      sample costs ONE v_alignbit_b32 (bits = bits << 1 | sign(phi); NO_DATA is (+0, -0) and no valid phi is negative).  The
      registers are emptied -- their population counts subtracted from the packed observation counts, to which the caller
      adds the epochs of the statement -- at its end and at least every 32 epochs inside it.

A slab is as tall as ITS epoch's shift box (SlabRef::bytes), not as the chunk's tallest: a wave whose 1 KiB piece starts
behind the slab's end (%[wp] >= bytes) reads the statement's first slab instead -- lines its neighbours hold in cache -- and
writes that where the piece would have gone, behind every row the epoch's offsets reach; the counters of the two copies in
flight stay the same for every wave and epoch.

Every read-write operand is early-clobber: the statements write them while inputs are still being read.

Since the end of round 3 the STREAM statements stage by LDS-DMA (stream_dma below; KB_GEN_NO_DMA=1 regenerates the
register-staged form described above, which the LOOP statements keep): a piece is requested with
global_load_lds_dwordx4 <lane offset>, <slab address> with M0 = the LDS address of the wave's piece in the slot requested next
(scalar s80 / s96, advanced by the slot stride per request; the caller's M0 is kept in s81 / s97), so v[100:107] and the
second-piece registers are free: the count-free statements for chunks of 16 read all sixteen samples of an epoch into
v[92:123] before one wait (WIDE_BATCH).  The request of an odd epoch sits at the END of its trip, behind the group change
when there was one (slot 0 of the group after next lies in the buffer everybody reads until that barrier), and its slab
reference is fetched in the even half of the same trip; %[gb] therefore counts from one entry earlier (KB_LDS_DMA in the
generated header tells search_lds.h).  `s_waitcnt vmcnt(0)` stands in front of every group's barrier.

Round 5: ONE more statement, STREAM_FAST_C32 (stream32 below), for chunks of 32 candidates on arrays beyond the Infinity
Cache, where the kernel is bound by the bytes that cross the fabric and those fall with the candidates a staged slab
serves.  It is the two-epochs-per-trip plan read differently: a trip is ONE epoch, its "even half" sums candidates 0-15 with
the A offsets (a0-a15), its "odd half" candidates 16-31 of the SAME epoch with the B offsets (a16-a31) -- a table row is 32
offsets, A and B its two halves --, so the scalar registers are those of C = 16.  One slab is requested per trip, at its end
(behind the barrier when the group changed), its reference fetched in the trip's first half; groups may hold any number of
epochs.  64 accumulator registers leave room for eight samples in flight (v[108:123]), not sixteen.  Count-free only: the
host gives this instance stacks without NO_DATA pixels and edge tables for the border tiles (search_kernels.hip).
"""
import os

TRIM = os.environ.get("KB_GEN_NO_TRIM") is None  # (timing experiment: first pieces copied whatever the slab's height)
# timing experiments (wrong results): statements without the slab loads / the LDS writes / the barrier / the table refills;
# KB_GEN_OUT = where the header goes (a variant build includes it through -DKB_ASM_HEADER)
NO_GLOAD = os.environ.get("KB_GEN_NO_GLOAD") is not None
NO_LWRITE = os.environ.get("KB_GEN_NO_LWRITE") is not None
NO_BARRIER = os.environ.get("KB_GEN_NO_BARRIER") is not None
NO_REFILL = os.environ.get("KB_GEN_NO_REFILL") is not None
# STREAM statements that stage by LDS-DMA (global_load_lds_dwordx4, M0 = destination): no staging registers, no ds_write_b128,
# no vmcnt wait inside an epoch.  A slab's bytes land whenever they arrive, so the request that used to be in flight across
# a group change (slot 0 of the group after next: its buffer is still being read) is issued BEHIND the barrier: the request of
# an odd epoch moves to the end of its trip.  The header then defines KB_LDS_DMA (search_lds.h passes the slab references
# from one entry earlier: the reference of an odd epoch is now fetched in the even half of its own trip).
DMA = os.environ.get("KB_GEN_NO_DMA") is None  # (KB_GEN_NO_DMA: the register-staged STREAM statements, for comparisons)
BURST_REGS = DMA and os.environ.get("KB_GEN_BURST") is not None  # (two more scalar registers in the clobber lists: see BURST)
# Timing experiment (KB_GEN_STAGGER="<bit>:<sleep>"): behind every group barrier the waves whose %[wp] (1024 x wave number) has
# that bit set sleep <sleep> x 64 cycles, so that the waves of a SIMD are not all in the read phase (then all in the add phase)
# of an epoch at the same moment.
STAGGER = os.environ.get("KB_GEN_STAGGER")


class Plan:
    def __init__(self, C):
        self.C = C
        if C == 8:
            self.A, self.B, self.gA, self.gB, self.addr, self.o1, self.o2 = 68, 76, 84, 92, 88, 90, 91
            self.sregs = range(68, (100 if BURST_REGS else 98) if DMA else 96)
            self.ws, self.m0s = 96, 97           # DMA: LDS address of the wave's piece in the slot requested next; M0 of the caller
            self.bc, self.bo = 98, 99            # BURST: requests left in a burst (pairs), running offset into the slab references
            self.bits = [92 + c for c in range(8)]
        else:   # (C = 32 runs on the plan of 16: a trip is one epoch in two halves of 16 candidates, see stream32)
            self.A, self.B, self.gA, self.gB, self.addr, self.o1, self.o2 = 36, 52, 68, 72, 76, 78, 79
            self.sregs = range(36, (84 if BURST_REGS else 82) if DMA else 80)   # (s32 - s35 are the stack registers of a kernel that has scratch memory)
            self.ws, self.m0s = 80, 81
            self.bc, self.bo = 82, 83
            self.bits = [84 + c for c in range(8)] * 2   # candidate k and k + 8 share a register (alternate bits)
        self.row = 4 * C                    # bytes of a table row (one epoch's offsets)
        self.loadx = f"s_load_dwordx{min(C, 16)}"
        self.raw = [108 + 2 * c for c in range(8)]

    def nbatch(self, fast, np_):
        # samples per batch: eight, or four where the second piece of a slab takes the upper half of the sample registers
        return 4 if (np_ == 2 and not fast) else 8

    def second(self, fast):
        # registers of a wave's second piece of the slabs delivered in even / odd epochs
        return ("v[92:95]", "v[96:99]") if fast else ("v[116:119]", "v[120:123]")

    def bit_regs(self):
        return sorted(set(self.bits))


def ln(text):
    return f'"{text}\\n\\t"\n'


def reads(p, base, batch, nb=8):
    return "".join(ln(f"v_add_u32 v124, s{base + nb * batch + c}, %[rb]\\n\\tds_read_b64 v[{p.raw[c]}:{p.raw[c] + 1}], v124") for c in range(nb))


def adds(p, fast, batch, nb=8):
    out = ""
    for c in range(nb):
        k = nb * batch + c
        out += ln(f"v_pk_add_f32 %[a{k}], %[a{k}], v[{p.raw[c]}:{p.raw[c] + 1}]"
                  + ("" if fast else f"\\n\\tv_alignbit_b32 v{p.bits[k]}, v{p.bits[k]}, v{p.raw[c] + 1}, 31"))
    return out


def zero_bits(p):
    return "".join(ln(f"v_mov_b32 v{b}, 0") for b in p.bit_regs())


def flush_bits(p, zero):
    # candidate 2j in the low half of count j, 2j + 1 in the high half
    out = ""
    if p.C == 8:
        for j in range(4):
            out += ln(f"v_bcnt_u32_b32 v124, v{p.bits[2 * j + 1]}, 0\\n\\tv_lshlrev_b32 v124, 16, v124\\n\\t"
                      f"v_bcnt_u32_b32 v124, v{p.bits[2 * j]}, v124\\n\\tv_sub_u32 %[c{j}], %[c{j}], v124")
    else:
        # register r holds candidate r in its odd bits and candidate r + 8 in its even bits (the later one of an epoch);
        # v108 / v109 are free here (no sample is in flight at a tally)
        for j in range(4):
            lo, hi = p.bits[2 * j], p.bits[2 * j + 1]
            for mask, cnt in (("0xaaaaaaaa", j), ("0x55555555", j + 4)):
                out += ln(f"v_and_b32 v124, {mask}, v{lo}\\n\\tv_and_b32 v108, {mask}, v{hi}\\n\\t"
                          f"v_bcnt_u32_b32 v124, v124, 0\\n\\tv_bcnt_u32_b32 v108, v108, 0\\n\\t"
                          f"v_lshl_add_u32 v124, v108, 16, v124\\n\\tv_sub_u32 %[c{cnt}], %[c{cnt}], v124")
    return out + (zero_bits(p) if zero else "")


def half(p, which, fast, np_, request, refill):
    base = p.A if which == "A" else p.B
    g = p.gA if which == "A" else p.gB
    mine, other = ("v[100:103]", "v[104:107]") if which == "A" else ("v[104:107]", "v[100:103]")
    sa, sb = p.second(fast)
    mine2, other2 = (sa, sb) if which == "A" else (sb, sa)
    imm1, imm2 = ("", "") if which == "A" else (f" offset:{hex(p.row)}", " offset:0x10")
    s = ""
    if request and np_:
        for piece in range(np_):
            wp, go, dst = (("%[wp]", "%[go]", other), ("%[wq]", "%[gq]", other2))[piece]
            if TRIM or piece == 1:
                s += ln(f"s_cmp_lt_u32 {wp}, s{g + 2}\\n\\ts_cselect_b32 s{p.addr}, s{g}, %[dl]\\n\\ts_cselect_b32 s{p.addr + 1}, s{g + 1}, %[dh]")
                s += ln(f"s_add_u32 s{p.addr}, %[tl], s{p.addr}\\n\\ts_addc_u32 s{p.addr + 1}, %[th], s{p.addr + 1}")
            else:
                s += ln(f"s_add_u32 s{p.addr}, %[tl], s{g}\\n\\ts_addc_u32 s{p.addr + 1}, %[th], s{g + 1}")
            if not NO_GLOAD:
                s += ln(f"global_load_dwordx4 {dst}, {go}, s[{p.addr}:{p.addr + 1}]")
    nb = p.nbatch(fast, np_)
    batches = p.C // nb
    for b in range(batches):
        s += reads(p, base, b, nb)
        s += ln("s_waitcnt lgkmcnt(0)")
        if b == batches - 1 and refill and not NO_REFILL:
            s += ln(f"{p.loadx} s[{base}:{base + p.C - 1}], %[ob], s{p.o1}{imm1}")
            if np_:
                s += ln(f"s_load_dwordx4 s[{g}:{g + 3}], %[gb], s{p.o2}{imm2}")
        s += adds(p, fast, b, nb)
    if np_:
        s += ln(f"s_waitcnt vmcnt({np_ if request else 0})")
        if not NO_LWRITE:
            s += ln(f"ds_write_b128 %[wd], {mine}")
            if np_ == 2:
                s += ln(f"ds_write_b128 %[wd], {mine2} offset:%[so]")
    s += ln("v_add_u32 %[wd], %[st], %[wd]")
    return s


def prologue(p, fast, np_):
    s = "" if fast else zero_bits(p)
    s += ln(f"{p.loadx} s[{p.A}:{p.A + p.C - 1}], %[ob], 0x0")
    s += ln(f"{p.loadx} s[{p.B}:{p.B + p.C - 1}], %[ob], {hex(p.row)}")
    if np_:
        s += ln(f"s_load_dwordx4 s[{p.gA}:{p.gA + 3}], %[gb], 0x0")
        s += ln(f"s_load_dwordx4 s[{p.gB}:{p.gB + 3}], %[gb], 0x10")
    s += ln(f"s_mov_b32 s{p.o1}, {hex(2 * p.row)}\\n\\ts_mov_b32 s{p.o2}, 0x20")
    if np_:
        s += ln("global_load_dwordx4 v[100:103], %[go], %[b0]")
    if np_ == 2:
        s += ln(f"global_load_dwordx4 {p.second(fast)[0]}, %[gq], %[b0]")
    s += ln("s_waitcnt lgkmcnt(0)")
    return s


def advance(p):
    return ln(f"s_add_u32 s{p.o1}, s{p.o1}, {hex(2 * p.row)}\\n\\ts_add_u32 s{p.o2}, s{p.o2}, 0x20")


def loop(p, fast, np_):
    s = prologue(p, fast, np_)
    s += f'"s_cmp_eq_u32 %[np], 0\\n\\ts_cbranch_scc1 kb_tail_%=_{np_}\\n"\n'
    s += f'"kb_loop_%=_{np_}:\\n\\t"\n'
    s += half(p, "A", fast, np_, True, True)
    s += half(p, "B", fast, np_, True, True)
    s += advance(p)
    s += f'"s_sub_u32 %[np], %[np], 1\\n\\ts_cmp_lg_u32 %[np], 0\\n\\ts_cbranch_scc1 kb_loop_%=_{np_}\\n"\n'
    s += f'"kb_tail_%=_{np_}:\\n\\t"\n'
    s += ln(f"s_cmp_eq_u32 %[od], 0\\n\\ts_cbranch_scc1 kb_end_%=_{np_}")
    s += half(p, "A", fast, np_, False, False)
    s += f'"kb_end_%=_{np_}:\\n\\t"\n'
    return s


def stream(p, fast, np_):
    """%[np]: pairs of epochs in total (>= 1), %[gc]: pairs until the next barrier, %[pg]: pairs per group, %[dr]: distance
    from the buffer being read to the other one, %[es]: bytes of a group of staged slabs; COUNT: %[fc] groups until the
    shift registers are emptied, %[fg] groups per 32 epochs."""
    s = prologue(p, fast, np_)
    s += f'"s_cmp_eq_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sfin_%=_{np_}\\n"\n'
    s += f'"kb_sloop_%=_{np_}:\\n\\t"\n'
    s += half(p, "A", fast, np_, True, True)
    s += half(p, "B", fast, np_, True, True)
    s += advance(p)
    s += ln(f"s_sub_u32 %[gc], %[gc], 1\\n\\ts_cmp_lg_u32 %[gc], 0\\n\\ts_cbranch_scc1 kb_snb_%=_{np_}")
    s += ln("s_waitcnt lgkmcnt(0)" + ("" if NO_BARRIER else "\\n\\ts_barrier"))
    s += ln("v_add_u32 %[rb], %[dr], %[rb]\\n\\tv_subrev_u32 %[wd], %[es], %[wd]\\n\\tv_subrev_u32 %[wd], %[dr], %[wd]")
    s += ln("s_sub_u32 %[dr], 0, %[dr]\\n\\ts_mov_b32 %[gc], %[pg]")
    if not fast:
        s += ln(f"s_sub_u32 %[fc], %[fc], 1\\n\\ts_cmp_lg_u32 %[fc], 0\\n\\ts_cbranch_scc1 kb_snb_%=_{np_}")
        s += flush_bits(p, True)
        s += ln("s_mov_b32 %[fc], %[fg]")
    s += '"\\n"\n'
    s += f'"kb_snb_%=_{np_}:\\n\\t"\n'
    s += f'"s_sub_u32 %[np], %[np], 1\\n\\ts_cmp_lg_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sloop_%=_{np_}\\n"\n'
    s += f'"kb_sfin_%=_{np_}:\\n\\t"\n'
    s += half(p, "A", fast, np_, True, False)
    s += half(p, "B", fast, np_, False, False)
    return s


def dma_request(p, which, np_):
    """The slab an even / odd epoch requests, straight into the wave's piece(s) of the slot requested next."""
    g = p.gA if which == "A" else p.gB
    s = ""
    for piece in range(np_):
        wp, go = (("%[wp]", "%[go]"), ("%[wq]", "%[gq]"))[piece]
        if TRIM or piece == 1:
            s += ln(f"s_cmp_lt_u32 {wp}, s{g + 2}\\n\\ts_cselect_b32 s{p.addr}, s{g}, %[dl]\\n\\ts_cselect_b32 s{p.addr + 1}, s{g + 1}, %[dh]")
            s += ln(f"s_add_u32 s{p.addr}, %[tl], s{p.addr}\\n\\ts_addc_u32 s{p.addr + 1}, %[th], s{p.addr + 1}")
        else:
            s += ln(f"s_add_u32 s{p.addr}, %[tl], s{g}\\n\\ts_addc_u32 s{p.addr + 1}, %[th], s{g + 1}")
        s += ln(f"s_mov_b32 m0, s{p.ws}" if piece == 0 else f"s_add_u32 m0, s{p.ws}, %[so]")
        s += ln(f"s_nop 0\\n\\tglobal_load_lds_dwordx4 {go}, s[{p.addr}:{p.addr + 1}]")
    if np_:
        s += ln(f"s_add_u32 s{p.ws}, s{p.ws}, %[st]")
    return s


# all 16 samples of an epoch read before ONE wait in the count-free statements for chunks of 16 (the registers the staging no
# longer needs hold them): 64 x 2048 x 2048 37.8 -> 36.6 ms, cfg2 2.41 -> 2.39 ms; KB_GEN_NO_WIDE_BATCH: two batches of eight
WIDE_BATCH = os.environ.get("KB_GEN_NO_WIDE_BATCH") is None
# ... and in the counting statements too (v[92:99] join their clobbers: two registers of the headline instance go to scratch
# memory, outside the loops): a stack with 1 % masked pixels 2.71 -> 2.64 ms at cfg2, 41.3 -> 39.9 ms on 2048 x 2048; KB_GEN_NO_WIDE_COUNT
WIDE_COUNT = os.environ.get("KB_GEN_NO_WIDE_COUNT") is None
# (per generated statement: the counting STREAM statement for chunks of 16 exists in both forms, COUNT and COUNTW; the pooled-list
# instance, which is short of registers, keeps COUNT -- search_lds.h: KB_LDS_WIDE_COUNT)
CUR = {"wide_count": False}


# Timing experiment (KB_GEN_BURST=1; measured slower, not the default): a whole group's slabs requested in ONE burst right behind
# the group change (every slot of the buffer just read is free from that barrier on) instead of one request per epoch, so that
# the last request of a group has a whole group to land in and a CU holds a group's worth of bytes in flight.  Same bits
# (`bench.py --verify` clean on cfg2, a masked stack, cfg3 and a 128-epoch stack), but cfg2 2.37 -> 2.56 ms and the cfg4 share
# 21.5 -> 23.3 ms (tools/ab.sh "noburst burst"): the references of a burst come by scalar loads that every wave of the tile waits
# for at the one moment when nobody else has work (right behind the barrier) -- and the HBM-resident configuration gained
# nothing relative to the cache-resident one, i.e. bytes in flight are not what binds it.
BURST = DMA and os.environ.get("KB_GEN_BURST") is not None


def dma_burst(p, np_, first):
    """Requests for the %[pg] * 2 slabs of the group staged next; their references are read in pairs from %[gb] + s{o2} - 0x20
    (+ 0x0 in the prologue: `first`), where the running offset stands one trip ahead of the epochs done."""
    if not np_:
        return ""
    tag = f"kb_burst{'p' if first else 'g'}_%=_{np_}"
    s = ln(f"s_mov_b32 s{p.bc}, %[pg]")
    s += ln(f"s_mov_b32 s{p.bo}, 0x0" if first else f"s_sub_u32 s{p.bo}, s{p.o2}, 0x20")
    s += '"\\n"\n'
    s += f'"{tag}:\\n\\t"\n'
    s += ln(f"s_load_dwordx4 s[{p.gA}:{p.gA + 3}], %[gb], s{p.bo}")
    s += ln(f"s_load_dwordx4 s[{p.gB}:{p.gB + 3}], %[gb], s{p.bo} offset:0x10")
    s += ln("s_waitcnt lgkmcnt(0)")
    s += dma_request(p, "A", np_)
    s += dma_request(p, "B", np_)
    s += ln(f"s_add_u32 s{p.bo}, s{p.bo}, 0x20\\n\\ts_sub_u32 s{p.bc}, s{p.bc}, 1\\n\\ts_cmp_lg_u32 s{p.bc}, 0\\n\\ts_cbranch_scc1 {tag}")
    return s


def dma_half(p, which, fast, np_, request, refill):
    base = p.A if which == "A" else p.B
    imm1 = "" if which == "A" else f" offset:{hex(p.row)}"
    s = dma_request(p, "A", np_) if (request and which == "A" and not BURST) else ""
    nb = 16 if (WIDE_BATCH and (fast or CUR["wide_count"]) and p.C == 16) else 8
    saved = p.raw
    if nb == 16:
        p.raw = [92 + 2 * c for c in range(16)]  # (no staging registers: v[92:107] are free in the count-free statements)
    for b in range(p.C // nb):
        s += reads(p, base, b, nb)
        s += ln("s_waitcnt lgkmcnt(0)")
        if b == p.C // nb - 1 and refill and not NO_REFILL:
            s += ln(f"{p.loadx} s[{base}:{base + p.C - 1}], %[ob], s{p.o1}{imm1}")
            if np_ and which == "A" and not BURST:
                # the references of the slabs the NEXT even epoch and THIS trip's odd epoch request (the base is one entry early)
                s += ln(f"s_load_dwordx4 s[{p.gA}:{p.gA + 3}], %[gb], s{p.o2} offset:0x10")
                s += ln(f"s_load_dwordx4 s[{p.gB}:{p.gB + 3}], %[gb], s{p.o2}")
        s += adds(p, fast, b, nb)
    p.raw = saved
    return s


def stream_dma(p, fast, np_):
    s = "" if fast else zero_bits(p)
    s += ln(f"{p.loadx} s[{p.A}:{p.A + p.C - 1}], %[ob], 0x0")
    s += ln(f"{p.loadx} s[{p.B}:{p.B + p.C - 1}], %[ob], {hex(p.row)}")
    if np_ and not BURST:
        s += ln(f"s_load_dwordx4 s[{p.gA}:{p.gA + 3}], %[gb], 0x10")
        s += ln(f"s_load_dwordx4 s[{p.gB}:{p.gB + 3}], %[gb], 0x20")
    s += ln(f"s_mov_b32 s{p.o1}, {hex(2 * p.row)}\\n\\ts_mov_b32 s{p.o2}, 0x20")
    s += ln(f"v_readfirstlane_b32 s{p.ws}, %[wd]")
    if BURST:
        s += dma_burst(p, np_, True)  # the whole first staged group (the references start AT %[gb] in this form)
    else:
        if np_:
            s += ln(f"s_mov_b32 m0, s{p.ws}\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %[go], %[b0]")
        if np_ == 2:
            s += ln(f"s_add_u32 m0, s{p.ws}, %[so]\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %[gq], %[b0]")
        s += ln(f"s_add_u32 s{p.ws}, s{p.ws}, %[st]")
    s += ln("s_waitcnt lgkmcnt(0)")
    s += f'"s_cmp_eq_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sfin_%=_{np_}\\n"\n'
    s += f'"kb_sloop_%=_{np_}:\\n\\t"\n'
    s += dma_half(p, "A", fast, np_, True, True)
    s += dma_half(p, "B", fast, np_, False, True)
    s += advance(p)
    s += ln(f"s_sub_u32 %[gc], %[gc], 1\\n\\ts_cmp_lg_u32 %[gc], 0\\n\\ts_cbranch_scc1 kb_snb_%=_{np_}")
    s += ln("s_waitcnt vmcnt(0) lgkmcnt(0)" + ("" if NO_BARRIER else "\\n\\ts_barrier"))
    if STAGGER:
        bit, sleep = STAGGER.split(":")
        s += ln(f"s_bitcmp0_b32 %[wp], {bit}\\n\\ts_cbranch_scc1 kb_nosleep_%=_{np_}\\n\\ts_sleep {sleep}")
        s += '"\\n"\n'
        s += f'"kb_nosleep_%=_{np_}:\\n\\t"\n'
    s += ln(f"v_add_u32 %[rb], %[dr], %[rb]\\n\\ts_sub_u32 s{p.ws}, s{p.ws}, %[es]\\n\\ts_sub_u32 s{p.ws}, s{p.ws}, %[dr]")
    s += ln("s_sub_u32 %[dr], 0, %[dr]\\n\\ts_mov_b32 %[gc], %[pg]")
    if BURST:
        s += dma_burst(p, np_, False)  # the group after next, into the buffer everybody has just left
    if not fast:
        s += ln(f"s_sub_u32 %[fc], %[fc], 1\\n\\ts_cmp_lg_u32 %[fc], 0\\n\\ts_cbranch_scc1 kb_snb_%=_{np_}")
        s += flush_bits(p, True)
        s += ln("s_mov_b32 %[fc], %[fg]")
    s += '"\\n"\n'
    s += f'"kb_snb_%=_{np_}:\\n\\t"\n'
    if not BURST:
        s += dma_request(p, "B", np_)  # (the odd epoch's request: behind the barrier when the group changed)
    s += f'"s_sub_u32 %[np], %[np], 1\\n\\ts_cmp_lg_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sloop_%=_{np_}\\n"\n'
    s += f'"kb_sfin_%=_{np_}:\\n\\t"\n'
    s += dma_half(p, "A", fast, np_, not BURST, False)
    s += dma_half(p, "B", fast, np_, False, False)
    return s


def half32(p, which, np_, refill):
    """Candidates 0-15 (A) or 16-31 (B) of one epoch: two batches of eight reads; behind the second wait the half's offsets of
    the NEXT epoch are fetched (and, in the first half, the reference of the slab this trip requests at its end)."""
    base, acc0, imm = (p.A, 0, "") if which == "A" else (p.B, 16, " offset:0x40")
    s = ""
    for b in range(2):
        s += reads(p, base, b, 8)
        s += ln("s_waitcnt lgkmcnt(0)")
        if b == 1 and refill and not NO_REFILL:
            s += ln(f"s_load_dwordx16 s[{base}:{base + 15}], %[ob], s{p.o1}{imm}")
            if np_ and which == "A":
                s += ln(f"s_load_dwordx4 s[{p.gB}:{p.gB + 3}], %[gb], s{p.o2}")
        for c in range(8):
            k = acc0 + 8 * b + c
            s += ln(f"v_pk_add_f32 %[a{k}], %[a{k}], v[{p.raw[c]}:{p.raw[c] + 1}]")
    return s


def stream32(p, fast, np_):
    """%[np]: epochs in total (>= 1), %[gc]: epochs until the next barrier, %[pg]: epochs per group; the rest as stream_dma.
    Requests: the prologue asks for the first slab of the group staged next (%[b0]), every trip but the last for the one after
    (its reference sits one entry behind the trip's own: %[gb] counts from one entry earlier, like the DMA statements of 16)."""
    assert fast
    s = ln(f"s_load_dwordx16 s[{p.A}:{p.A + 15}], %[ob], 0x0")
    s += ln(f"s_load_dwordx16 s[{p.B}:{p.B + 15}], %[ob], 0x40")
    s += ln(f"s_mov_b32 s{p.o1}, 0x80\\n\\ts_mov_b32 s{p.o2}, 0x10")
    s += ln(f"v_readfirstlane_b32 s{p.ws}, %[wd]")
    if np_:
        s += ln(f"s_mov_b32 m0, s{p.ws}\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %[go], %[b0]")
    if np_ == 2:
        s += ln(f"s_add_u32 m0, s{p.ws}, %[so]\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %[gq], %[b0]")
    s += ln(f"s_add_u32 s{p.ws}, s{p.ws}, %[st]")
    s += ln("s_waitcnt lgkmcnt(0)")
    s += f'"s_cmp_eq_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sfin_%=_{np_}\\n"\n'
    s += f'"kb_sloop_%=_{np_}:\\n\\t"\n'
    s += half32(p, "A", np_, True)
    s += half32(p, "B", np_, True)
    s += ln(f"s_add_u32 s{p.o1}, s{p.o1}, 0x80\\n\\ts_add_u32 s{p.o2}, s{p.o2}, 0x10")
    s += ln(f"s_sub_u32 %[gc], %[gc], 1\\n\\ts_cmp_lg_u32 %[gc], 0\\n\\ts_cbranch_scc1 kb_snb_%=_{np_}")
    s += ln("s_waitcnt vmcnt(0) lgkmcnt(0)" + ("" if NO_BARRIER else "\\n\\ts_barrier"))
    s += ln(f"v_add_u32 %[rb], %[dr], %[rb]\\n\\ts_sub_u32 s{p.ws}, s{p.ws}, %[es]\\n\\ts_sub_u32 s{p.ws}, s{p.ws}, %[dr]")
    s += ln("s_sub_u32 %[dr], 0, %[dr]\\n\\ts_mov_b32 %[gc], %[pg]")
    s += '"\\n"\n'
    s += f'"kb_snb_%=_{np_}:\\n\\t"\n'
    s += dma_request(p, "B", np_)
    s += f'"s_sub_u32 %[np], %[np], 1\\n\\ts_cmp_lg_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sloop_%=_{np_}\\n"\n'
    s += f'"kb_sfin_%=_{np_}:\\n\\t"\n'
    s += half32(p, "A", np_, False)
    s += half32(p, "B", np_, False)
    return s


def combined(p, body, fast):
    """One statement for every wave: %[nq] (0, 1 or 2: the pieces of each slab this wave copies) picks the body."""
    dma = body is stream_dma or body is stream32
    s = ln(f"s_mov_b32 s{p.m0s}, m0") if dma else ""
    s += ln("s_cmp_eq_u32 %[nq], 2\\n\\ts_cbranch_scc1 kb_two_%=\\n\\ts_cmp_eq_u32 %[nq], 1\\n\\ts_cbranch_scc1 kb_one_%=")
    s += body(p, fast, 0)
    s += '"s_branch kb_done_%=\\n"\n"kb_one_%=:\\n\\t"\n'
    s += body(p, fast, 1)
    s += '"s_branch kb_done_%=\\n"\n"kb_two_%=:\\n\\t"\n'
    s += body(p, fast, 2)
    s += '"\\n"\n"kb_done_%=:\\n\\t"\n'
    if not fast:
        s += flush_bits(p, False)
    if dma:
        s += ln(f"s_mov_b32 m0, s{p.m0s}")
    s += '"s_waitcnt vmcnt(0) lgkmcnt(0)"\n'
    return s


def operands(p, family, fast):
    outs = [f'[a{c}] "+&v"(acc[{c}])' for c in range(p.C)]
    if not fast:
        outs += [f'[c{j}] "+&v"(cntp[{j}])' for j in range(p.C // 2)]
    outs.append('[wd] "+&v"(wd)')
    ins = []
    if family == "STREAM":
        outs += ['[rb] "+&v"(rb)', '[np] "+&s"(pairs)', '[gc] "+&s"(gc)', '[dr] "+&s"(dr)']
        if not fast:
            outs.append('[fc] "+&s"(fc)')
    else:
        outs.append('[np] "+&s"(pairs)')
        ins.append('[rb] "v"(rb)')
    ins += ['[go] "v"(go)', '[gq] "v"(gq)', '[nq] "s"(nq)', '[ob] "s"(ob)', '[gb] "s"(gb)', '[b0] "s"(b0)', '[tl] "s"(tl)',
            '[th] "s"(th)', '[wp] "s"(wp)', '[dl] "s"(dl)', '[dh] "s"(dh)', '[wq] "s"(wq)', '[so] "i"(stage_round(ROWS))',
            '[st] "s"(st)']
    if family == "STREAM":
        ins += ['[pg] "s"(pg)', '[es] "s"(es)']
        if not fast:
            ins.append('[fg] "s"(fg)')
    else:
        ins.append('[od] "s"(odd)')
    return ", ".join(outs), ", ".join(ins)


def clobbers(p, fast):
    if p.C == 32:  # (64 accumulator registers: eight samples in flight + the address temporary, nothing else)
        return ", ".join([f'"v{i}"' for i in range(108, 125)] + [f'"s{i}"' for i in p.sregs] + ['"vcc"', '"scc"', '"memory"'])
    v = [f'"v{i}"' for i in range(92, 100)] if fast else [f'"v{b}"' for b in p.bit_regs()]
    if not fast and CUR["wide_count"] and DMA and p.C == 16:
        v += [f'"v{i}"' for i in range(92, 100)]
    v += [f'"v{i}"' for i in range(100, 125)]
    return ", ".join(v + [f'"s{i}"' for i in p.sregs] + ['"vcc"', '"scc"', '"memory"'])


def select_tree(n):
    """KB_SELECT_TREE_<n>(P, F, SEL, ACC): the (psi, phi) sums of candidate SEL (per lane, 0 .. n - 1) out of the n accumulator
    pairs ACC[0 .. n - 1] (.x, .y), halving by the bits of SEL like select_by_bits (search_device.h) -- written out because next
    to 64 live sums the compiler's own tree keeps a level's worth of intermediate values alive and puts the sums into scratch
    memory.  Depth first: one temporary per level, the two fields one after the other through the same temporaries."""
    levels = n.bit_length() - 1
    text = []
    for k in range(levels):  # m<k>: lanes whose SEL has bit k set
        text.append(f"v_and_b32 %[t0], {1 << k}, %[sel]\\n\\tv_cmp_ne_u32 %[m{k}], 0, %[t0]")

    def field(prefix, dest):
        free = [f"%[t{i}]" for i in range(levels + 1)]

        def sub(lo, size, out=None):
            if size == 1:
                return f"%[{prefix}{lo}]", False
            half = size // 2
            a, a_tmp = sub(lo, half)
            b, b_tmp = sub(lo + half, half)
            d = out if out is not None else (a if a_tmp else free.pop())
            text.append(f"v_cndmask_b32 {d}, {a}, {b}, %[m{half.bit_length() - 1}]")
            if b_tmp:
                free.append(b)
            if a_tmp and d != a:
                free.append(a)
            return d, True

        sub(0, n, dest)

    field("a", "%[p]")
    field("b", "%[f]")
    outs = ['[p] "=&v"(P)', '[f] "=&v"(F)'] + [f'[t{i}] "=&v"(kb_t{i})' for i in range(levels + 1)] + \
           [f'[m{k}] "=&s"(kb_m{k})' for k in range(levels)]
    ins = ['[sel] "v"(SEL)'] + [f'[a{c}] "v"((ACC)[{c}].x)' for c in range(n)] + [f'[b{c}] "v"((ACC)[{c}].y)' for c in range(n)]
    out = [f'#define KB_SELECT_TREE_{n}(P, F, SEL, ACC) \\', '    { \\',
           '        float ' + ", ".join(f"kb_t{i}" for i in range(levels + 1)) + '; \\',
           '        unsigned long long ' + ", ".join(f"kb_m{k}" for k in range(levels)) + '; \\',
           '        asm( \\']
    out += [f'            "{t}\\n\\t" \\' for t in text]
    out += [f'            : {", ".join(outs)} \\', f'            : {", ".join(ins)}); \\', '    }', '']
    return out


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = ['// GENERATED by tools/gen_lds_loop.py -- the hand-scheduled summing loops of kb_search_lds (search_lds.h: asm_run and the',
           '// run of whole groups at the top of the group loop).  Register plan, schedule and the reasons: the generator\'s header.',
           '// KB_LDS_RUN_<LOOP|STREAM> expand, inside lds_search_tile, to the statement for its C and FAST; a statement holds the',
           '// bodies for waves that copy no, one or two pieces of every slab and picks by nq; it names the variables of its call',
           '// site (acc, cntp, wd, rb, pairs, gc, dr, fc, go, gq, nq, ob, gb, b0, tl, th, wp, wq, dl, dh, st, pg, es, fg, odd).',
           '#ifndef KB_SEARCH_LDS_ASM_H_', '#define KB_SEARCH_LDS_ASM_H_', ''] + (['#define KB_LDS_DMA 1', ''] if DMA else []) + (['#define KB_LDS_BURST 1', ''] if BURST else [])
    for C in (8, 16):
        p = Plan(C)
        for family, body in (("LOOP", loop), ("STREAM", stream_dma if DMA else stream)):
            kinds = [(True, "FAST", False), (False, "COUNT", False)]
            if family == "STREAM" and C == 16 and DMA and WIDE_BATCH and WIDE_COUNT:
                kinds.append((False, "COUNTW", True))
            for fast, kind, wide_count in kinds:
                CUR["wide_count"] = wide_count
                outs, ins = operands(p, family, fast)
                lines = combined(p, body, fast).rstrip("\n").split("\n")
                out.append(f'#define KB_LDS_{family}_{kind}_C{C} \\')
                out.append("    asm volatile( \\")
                out.append(" \\\n".join("        " + x for x in lines) + " \\")
                out.append(f"        : {outs} \\")
                out.append(f"        : {ins} \\")
                out.append(f"        : {clobbers(p, fast)});")
                out.append('')
    if DMA:
        p = Plan(32)
        outs, ins = operands(p, "STREAM", True)
        lines = combined(p, stream32, True).rstrip("\n").split("\n")
        out.append('#define KB_LDS_STREAM_FAST_C32 \\')
        out.append("    asm volatile( \\")
        out.append(" \\\n".join("        " + x for x in lines) + " \\")
        out.append(f"        : {outs} \\")
        out.append(f"        : {ins} \\")
        out.append(f"        : {clobbers(p, True)});")
        out.append('')
    for family in ("LOOP", "STREAM"):
        out.append(f'#define KB_LDS_RUN_{family} \\')
        if family == "STREAM" and DMA:
            out.append('    if constexpr (C == 32) { \\')
            out.append('        if constexpr (FAST) { KB_LDS_STREAM_FAST_C32 } \\')
            out.append('    } else \\')
        out.append('    if constexpr (C == 8) { \\')
        out.append(f'        if constexpr (FAST) {{ KB_LDS_{family}_FAST_C8 }} else {{ KB_LDS_{family}_COUNT_C8 }} \\')
        out.append('    } else { \\')
        if family == "STREAM" and DMA and WIDE_BATCH and WIDE_COUNT:
            out.append(f'        if constexpr (FAST) {{ KB_LDS_{family}_FAST_C16 }} else if constexpr (KB_LDS_WIDE_COUNT) {{ KB_LDS_{family}_COUNTW_C16 }} else {{ KB_LDS_{family}_COUNT_C16 }} \\')
        else:
            out.append(f'        if constexpr (FAST) {{ KB_LDS_{family}_FAST_C16 }} else {{ KB_LDS_{family}_COUNT_C16 }} \\')
        out.append('    }')
        out.append('')
    out += select_tree(32)
    out.append('#endif')
    with open(os.environ.get("KB_GEN_OUT", os.path.join(root, "kbmod_amd", "csrc", "search_lds_asm.h")), "w") as fh:
        fh.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
