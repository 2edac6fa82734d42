#!/usr/bin/env python3
"""Writes kbmod_amd/csrc/search_lds_asm.h: the hand-scheduled summing loop of kb_search_lds as inline-asm strings.

Register plan inside the asm statement (all named in its clobber list, so the compiler keeps nothing in them across it):
  s[68:75]  offsets of the even epochs (set A)      s[76:83]  offsets of the odd epochs (set B)
  s[84:85]  origin of the slab requested in even epochs (gA)   s[86:87]  ... in odd epochs (gB)
  s[88:89]  address of the slab being requested     s90 / s91  running byte offsets into the offset / slab-reference tables
  v[100:103] piece of the slab delivered in even epochs (va)   v[104:107] ... in odd epochs (vb)
  v[108:123] the eight (psi, phi) samples of the epoch         v124  address / count temporary
"""
import os

A, B = 68, 76
RAW = [108 + 2 * c for c in range(8)]


def reads(base):
    return "".join(f'"v_add_u32 v124, s{base + c}, %[rb]\\n\\tds_read_b64 v[{RAW[c]}:{RAW[c] + 1}], v124\\n\\t"\n' for c in range(8))


def adds_fast():
    return "".join(f'"v_pk_add_f32 %[a{c}], %[a{c}], v[{RAW[c]}:{RAW[c] + 1}]\\n\\t"\n' for c in range(8))


def adds_count():
    # per pair j: candidate 2j counts in the low half of cnt j (carry-in), 2j + 1 in the high half (0x10000 through a select);
    # two instructions between a compare and the reader of its vcc where packed adds are left to fill the gap, s_nop 1 where not
    out = []
    pk = [f'"v_pk_add_f32 %[a{c}], %[a{c}], v[{RAW[c]}:{RAW[c] + 1}]\\n\\t"\n' for c in range(8)]
    fill = [pk[0] + pk[1], pk[2] + pk[3], pk[4] + pk[5], pk[6] + pk[7]] + ['"s_nop 1\\n\\t"\n'] * 4
    f = 0
    for j in range(4):
        hi_y, lo_y = RAW[2 * j + 1] + 1, RAW[2 * j] + 1
        out.append(f'"v_cmp_ne_u32 vcc, 0x80000000, v{hi_y}\\n\\t"\n')
        out.append(fill[f]); f += 1
        out.append('"v_cndmask_b32 v124, 0, %[k64], vcc\\n\\t"\n')
        out.append(f'"v_cmp_ne_u32 vcc, 0x80000000, v{lo_y}\\n\\t"\n')
        out.append(fill[f]); f += 1
        out.append(f'"v_addc_co_u32 %[c{j}], vcc, v124, %[c{j}], vcc\\n\\t"\n')
    return "".join(out)


def half(which, fast, np_, request, refill):
    base = A if which == "A" else B
    g = 84 if which == "A" else 86
    mine, other = ("v[100:103]", "v[104:107]") if which == "A" else ("v[104:107]", "v[100:103]")
    imm8, imm2 = ("", "") if which == "A" else (" offset:0x20", " offset:0x10")
    s = ""
    if request and np_:
        s += f'"s_add_u32 s88, %[tl], s{g}\\n\\ts_addc_u32 s89, %[th], s{g + 1}\\n\\t"\n'
        s += f'"global_load_dwordx4 {other}, %[go], s[88:89]\\n\\t"\n'
    s += reads(base)
    s += '"s_waitcnt lgkmcnt(0)\\n\\t"\n'
    if refill:
        s += f'"s_load_dwordx8 s[{base}:{base + 7}], %[ob], s90{imm8}\\n\\t"\n'
        s += f'"s_load_dwordx2 s[{g}:{g + 1}], %[gb], s91{imm2}\\n\\t"\n'
    s += adds_fast() if fast else adds_count()
    if np_:
        s += f'"s_waitcnt vmcnt({1 if request else 0})\\n\\t"\n'
        s += f'"ds_write_b128 %[wd], {mine}\\n\\t"\n'
    s += '"v_add_u32 %[wd], %[st], %[wd]\\n\\t"\n'
    return s


def body(fast, np_):
    s = ""
    # prologue: both sets of table words, the first slab's piece
    s += f'"s_load_dwordx8 s[{A}:{A + 7}], %[ob], 0x0\\n\\t"\n'
    s += f'"s_load_dwordx8 s[{B}:{B + 7}], %[ob], 0x20\\n\\t"\n'
    s += '"s_load_dwordx2 s[84:85], %[gb], 0x0\\n\\t"\n'
    s += '"s_load_dwordx2 s[86:87], %[gb], 0x10\\n\\t"\n'
    s += '"s_mov_b32 s90, 0x40\\n\\ts_mov_b32 s91, 0x20\\n\\t"\n'
    if np_:
        s += '"global_load_dwordx4 v[100:103], %[go], %[b0]\\n\\t"\n'
    s += '"s_waitcnt lgkmcnt(0)\\n\\t"\n'
    s += '"s_cmp_eq_u32 %[np], 0\\n\\ts_cbranch_scc1 kb_tail_%=\\n"\n'
    s += '"kb_loop_%=:\\n\\t"\n'
    s += half("A", fast, np_, True, True)
    s += half("B", fast, np_, True, True)
    s += '"s_add_u32 s90, s90, 0x40\\n\\ts_add_u32 s91, s91, 0x20\\n\\t"\n'
    s += '"s_sub_u32 %[np], %[np], 1\\n\\ts_cmp_lg_u32 %[np], 0\\n\\ts_cbranch_scc1 kb_loop_%=\\n"\n'
    s += '"kb_tail_%=:\\n\\t"\n'
    s += '"s_cmp_eq_u32 %[od], 0\\n\\ts_cbranch_scc1 kb_end_%=\\n\\t"\n'
    s += half("A", fast, np_, False, False)
    s += '"kb_end_%=:\\n\\t"\n'
    s += '"s_waitcnt vmcnt(0) lgkmcnt(0)"\n'
    return s


def stream(fast, np_):
    """A run of whole groups inside one chunk (every group full, staging a full group of the same chunk, an even number of
    epochs per group): the same two-epoch trip, with the group change -- LDS writes landed, barrier, the read pointer to the
    other buffer, the write pointer back to slot 0 of the one just read -- inside the statement, so the slab loads and the
    table words stay in flight across it.  %[np]: pairs in total (>= 1), %[gc]: pairs until the next barrier, %[pg]: pairs per
    group, %[dr]: distance from the buffer being read to the other one, %[es]: bytes of a group of staged slabs."""
    s = ""
    s += f'"s_load_dwordx8 s[{A}:{A + 7}], %[ob], 0x0\\n\\t"\n'
    s += f'"s_load_dwordx8 s[{B}:{B + 7}], %[ob], 0x20\\n\\t"\n'
    s += '"s_load_dwordx2 s[84:85], %[gb], 0x0\\n\\t"\n'
    s += '"s_load_dwordx2 s[86:87], %[gb], 0x10\\n\\t"\n'
    s += '"s_mov_b32 s90, 0x40\\n\\ts_mov_b32 s91, 0x20\\n\\t"\n'
    if np_:
        s += '"global_load_dwordx4 v[100:103], %[go], %[b0]\\n\\t"\n'
    s += '"s_waitcnt lgkmcnt(0)\\n\\t"\n'
    s += '"s_cmp_eq_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sfin_%=\\n"\n'
    s += '"kb_sloop_%=:\\n\\t"\n'
    s += half("A", fast, np_, True, True)
    s += half("B", fast, np_, True, True)
    s += '"s_add_u32 s90, s90, 0x40\\n\\ts_add_u32 s91, s91, 0x20\\n\\t"\n'
    s += '"s_sub_u32 %[gc], %[gc], 1\\n\\ts_cmp_lg_u32 %[gc], 0\\n\\ts_cbranch_scc1 kb_snb_%=\\n\\t"\n'
    s += '"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\\n\\t"\n'
    s += '"v_add_u32 %[rb], %[dr], %[rb]\\n\\tv_subrev_u32 %[wd], %[es], %[wd]\\n\\tv_subrev_u32 %[wd], %[dr], %[wd]\\n\\t"\n'
    s += '"s_sub_u32 %[dr], 0, %[dr]\\n\\ts_mov_b32 %[gc], %[pg]\\n"\n'
    s += '"kb_snb_%=:\\n\\t"\n'
    s += '"s_sub_u32 %[np], %[np], 1\\n\\ts_cmp_lg_u32 %[np], 1\\n\\ts_cbranch_scc1 kb_sloop_%=\\n"\n'
    s += '"kb_sfin_%=:\\n\\t"\n'
    s += half("A", fast, np_, True, False)
    s += half("B", fast, np_, False, False)
    s += '"s_waitcnt vmcnt(0) lgkmcnt(0)"\n'
    return s


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = ['// GENERATED by tools/gen_lds_loop.py -- the hand-scheduled summing loop of kb_search_lds (search_lds.h, streamlined run).',
           '// One asm statement runs a whole group: two epochs per trip, table words of epoch e + 2 fetched BEHIND the wait for',
           '// epoch e\'s LDS reads (a whole epoch for the scalar loads to land in), two slabs in flight, no compiler-scheduled',
           '// code while anything is in flight.  Register plan: see the generator.',
           '#ifndef KB_SEARCH_LDS_ASM_H_', '#define KB_SEARCH_LDS_ASM_H_', '']
    for fast in (True, False):
        for np_ in (0, 1):
            out.append(f'#define KB_LDS_LOOP_{"FAST" if fast else "COUNT"}_NP{np_} \\')
            lines = body(fast, np_).rstrip("\n").split("\n")
            out.append(" \\\n".join("    " + ln for ln in lines))
            out.append('')
            out.append(f'#define KB_LDS_STREAM_{"FAST" if fast else "COUNT"}_NP{np_} \\')
            lines = stream(fast, np_).rstrip("\n").split("\n")
            out.append(" \\\n".join("    " + ln for ln in lines))
            out.append('')
    out.append('#define KB_LDS_LOOP_CLOBBERS \\')
    regs = [f'"v{i}"' for i in range(100, 125)] + [f'"s{i}"' for i in range(68, 92)] + ['"vcc"', '"scc"', '"memory"']
    out.append("    " + ", ".join(regs))
    out.append('')
    out.append('#endif')
    with open(os.path.join(root, "kbmod_amd", "csrc", "search_lds_asm.h"), "w") as fh:
        fh.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
