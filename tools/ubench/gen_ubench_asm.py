#!/usr/bin/env python3
"""ubench_asm.h: asm-exact loop bodies for the issue model (the compiler fuses or splits the C++ forms: ds_read2_b64, a wait per
read).  Each macro is ONE asm statement holding the whole timed loop with fixed registers:
  v[32:95] accumulators (pairs), v[96:127] samples, v[16:31] addresses; %0 = lane base (v), %1 = result, %2 = iterations (s), %3.. offsets (s).
KB_UB_TODAY   16 candidates x 1 row:  16 v_add_u32, 16 ds_read_b64, wait, 16 v_pk_add_f32            (3.0 instructions / evaluation)
KB_UB_ROWS2   8 candidates x 2 rows:  8 v_add_u32, 16 ds_read_b64 (row 2 at offset:704), wait, 16 pk_add   (2.5)
KB_UB_ROWS2W  16 candidates x 2 rows, two batches of 8 candidates: 64 accumulator registers              (2.5)
KB_UB_ROWS4   4 rows x 8 candidates in two batches of (8 cand x 2 rows) sharing the 8 addresses: 8 v_add per 32 reads (2.25)
"""


def body(n_cand, rows, batches, shared_addr=False):
    out = []
    per = n_cand // batches if not shared_addr else n_cand
    acc = 32
    for b in range(batches):
        if not (shared_addr and b > 0):
            for c in range(per):
                out.append(f"v_add_u32 v{16 + c}, s{{off{(b * per + c) % 16}}}, %0")
        k = 0
        for c in range(per):
            for r in range(rows):
                row = r + (b * rows if shared_addr else 0)
                off = f" offset:{704 * row}" if row else ""
                out.append(f"ds_read_b64 v[{96 + 2 * k}:{97 + 2 * k}], v{16 + c}{off}")
                k += 1
        out.append("s_waitcnt lgkmcnt(0)")
        for j in range(k):
            out.append(f"v_pk_add_f32 v[{acc}:{acc + 1}], v[{acc}:{acc + 1}], v[{96 + 2 * j}:{97 + 2 * j}]")
            acc += 2
    return out, acc - 32


def macro(name, n_cand, rows, batches, shared_addr=False):
    ins, n_acc = body(n_cand, rows, batches, shared_addr)
    lines = ["s_mov_b32 s60, %2"]
    lines += [f"v_mov_b32 v{32 + i}, 0" for i in range(n_acc)]
    lines.append("1:")
    for i in ins:
        for c in range(16):
            i = i.replace(f"s{{off{c}}}", f"%{3 + c}")
        lines.append(i)
    lines += ["s_sub_u32 s60, s60, 1", "s_cmp_lg_u32 s60, 0", "s_cbranch_scc1 1b"]
    # fold the accumulators into v32 (so that the work is observable)
    for i in range(1, n_acc):
        lines.append(f"v_add_f32 v32, v32, v{32 + i}")
    lines.append("v_mov_b32 %1, v32")
    text = " \\\n    ".join('"' + l + '\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in list(range(32, 32 + max(n_acc, 1))) + list(range(96, 128)) + list(range(16, 32)))
    return (f"#define {name}(base, iters, o, result) \\\n  asm volatile( \\\n    {text} \\\n    : \"+v\"(base), \"=&v\"(result) : \"s\"(iters), "
            + ", ".join(f'"s"(o[{c}])' for c in range(16)) + f" : {clob}, \"s60\", \"scc\", \"memory\");\n"
            f"// {name}: {len(ins)} instructions per iteration, {n_cand * rows} evaluations\n")


def class_macro(name, jumps=True):
    """Prefix-sharing model: per iteration only n = 1 + (iteration & 15) of the 16 (address add, read, add) units run (mean 8.5),
    entered by two computed jumps (s_setpc_b64) into code laid out for unit 15 down to 0; `jumps` False: the same unit counts
    cannot be had without jumps, so the variant runs all 16 (control: the jump overhead alone shows against KB_UB_TODAY)."""
    lines = ["s_mov_b32 s60, %2"] + [f"v_mov_b32 v{32 + i}, 0" for i in range(32)]
    # the two entry tables' addresses, once (s[66:67] reads, s[68:69] adds)
    lines += ["s_getpc_b64 s[66:67]", "kb_ub_here_%=:", "s_add_u32 s66, s66, kb_ub_rd_%=-kb_ub_here_%=", "s_addc_u32 s67, s67, 0",
              "s_getpc_b64 s[68:69]", "kb_ub_here2_%=:", "s_add_u32 s68, s68, kb_ub_ad_%=-kb_ub_here2_%=", "s_addc_u32 s69, s69, 0", "1:"]
    lines += ["s_and_b32 s62, s60, 15" if jumps else "s_mov_b32 s62, 15", "s_sub_u32 s62, 15, s62", "s_mul_i32 s63, s62, 12",
              "s_add_u32 s64, s66, s63", "s_addc_u32 s65, s67, 0", "s_setpc_b64 s[64:65]", "kb_ub_rd_%=:"]
    for j in range(15, -1, -1):
        lines += [f"v_add_u32_e32 v124, %{3 + j}, %0", f"ds_read_b64 v[{96 + 2 * j}:{97 + 2 * j}], v124"]
    lines += ["s_waitcnt lgkmcnt(0)", "s_lshl_b32 s63, s62, 3", "s_add_u32 s64, s68, s63", "s_addc_u32 s65, s69, 0", "s_setpc_b64 s[64:65]", "kb_ub_ad_%=:"]
    for j in range(15, -1, -1):
        lines.append(f"v_pk_add_f32 v[{32 + 2 * j}:{33 + 2 * j}], v[{32 + 2 * j}:{33 + 2 * j}], v[{96 + 2 * j}:{97 + 2 * j}]")
    lines += ["s_sub_u32 s60, s60, 1", "s_cmp_lg_u32 s60, 0", "s_cbranch_scc1 1b"]
    for i in range(1, 32):
        lines.append(f"v_add_f32 v32, v32, v{32 + i}")
    lines.append("v_mov_b32 %1, v32")
    text = " \\\n    ".join('"' + l + '\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in list(range(32, 64)) + list(range(96, 128)))
    return (f"#define {name}(base, iters, o, result) \\\n  asm volatile( \\\n    {text} \\\n    : \"+v\"(base), \"=&v\"(result) : \"s\"(iters), "
            + ", ".join(f'"s"(o[{c}])' for c in range(16)) + f" : {clob}, \"s60\", \"s62\", \"s63\", \"s64\", \"s65\", \"s66\", \"s67\", \"s68\", \"s69\", \"scc\", \"memory\");\n")


def fixed_macro(name, n_updates):
    """The addresses of the 16 candidates live in v[16:31] across the loop (set up once); an iteration updates n_updates of
    them (a candidate whose shift changed between two epochs) and reads every sample at address + immediate offset: what a
    statement costs when the slot of an epoch is an immediate and most candidates keep their shift from one epoch to the next."""
    lines = ["s_mov_b32 s60, %2"] + [f"v_mov_b32 v{32 + i}, 0" for i in range(32)]
    lines += [f"v_add_u32 v{16 + c}, %{3 + c}, %0" for c in range(16)]
    lines.append("1:")
    for u in range(n_updates):
        lines.append(f"v_add_u32 v{16 + (3 * u) % 16}, %{3 + u}, %0")
    for c in range(16):
        lines.append(f"ds_read_b64 v[{96 + 2 * c}:{97 + 2 * c}], v{16 + c} offset:{64 * (c % 4)}")
    lines.append("s_waitcnt lgkmcnt(0)")
    for c in range(16):
        lines.append(f"v_pk_add_f32 v[{32 + 2 * c}:{33 + 2 * c}], v[{32 + 2 * c}:{33 + 2 * c}], v[{96 + 2 * c}:{97 + 2 * c}]")
    lines += ["s_sub_u32 s60, s60, 1", "s_cmp_lg_u32 s60, 0", "s_cbranch_scc1 1b"]
    for i in range(1, 32):
        lines.append(f"v_add_f32 v32, v32, v{32 + i}")
    lines.append("v_mov_b32 %1, v32")
    text = " \\\n    ".join('"' + l + '\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in list(range(32, 64)) + list(range(96, 128)) + list(range(16, 32)))
    return (f"#define {name}(base, iters, o, result) \\\n  asm volatile( \\\n    {text} \\\n    : \"+v\"(base), \"=&v\"(result) : \"s\"(iters), "
            + ", ".join(f'"s"(o[{c}])' for c in range(16)) + f" : {clob}, \"s60\", \"scc\", \"memory\");\n")


def branchy_macro(name, mask):
    """KB_UB_FIXED* with the updates chosen per candidate at run time: bit c of a scalar mask says whether candidate c's address
    changes this epoch; the update sits out of line (s_cbranch_scc1 there, s_branch back): what the kept-address statement
    costs with its control flow.  The mask alternates between `mask` and its rotation, so that no branch is always (not) taken."""
    lines = ["s_mov_b32 s60, %2", f"s_mov_b32 s61, {mask}"] + [f"v_mov_b32 v{32 + i}, 0" for i in range(32)]
    lines += [f"v_add_u32 v{16 + c}, %{3 + c}, %0" for c in range(16)]
    lines.append("1:")
    for c in range(16):
        lines += [f"s_bitcmp1_b32 s61, {c}", f"s_cbranch_scc1 kb_ub_u{c}_%=", f"kb_ub_b{c}_%=:"]
    for c in range(16):
        lines.append(f"ds_read_b64 v[{96 + 2 * c}:{97 + 2 * c}], v{16 + c} offset:{64 * (c % 4)}")
    lines.append("s_waitcnt lgkmcnt(0)")
    for c in range(16):
        lines.append(f"v_pk_add_f32 v[{32 + 2 * c}:{33 + 2 * c}], v[{32 + 2 * c}:{33 + 2 * c}], v[{96 + 2 * c}:{97 + 2 * c}]")
    lines += ["s_lshl_b32 s62, s61, 3", "s_lshr_b32 s61, s61, 13", "s_or_b32 s61, s61, s62", "s_and_b32 s61, s61, 0xffff"]  # rotate the 16-bit mask by 3
    lines += ["s_sub_u32 s60, s60, 1", "s_cmp_lg_u32 s60, 0", "s_cbranch_scc1 1b", "s_branch kb_ub_end_%="]
    for c in range(16):
        lines += [f"kb_ub_u{c}_%=:", f"v_add_u32 v{16 + c}, %{3 + c}, %0", f"s_branch kb_ub_b{c}_%="]
    lines.append("kb_ub_end_%=:")
    for i in range(1, 32):
        lines.append(f"v_add_f32 v32, v32, v{32 + i}")
    lines.append("v_mov_b32 %1, v32")
    text = " \\\n    ".join('"' + l + '\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in list(range(32, 64)) + list(range(96, 128)) + list(range(16, 32)))
    return (f"#define {name}(base, iters, o, result) \\\n  asm volatile( \\\n    {text} \\\n    : \"+v\"(base), \"=&v\"(result) : \"s\"(iters), "
            + ", ".join(f'"s"(o[{c}])' for c in range(16)) + f" : {clob}, \"s60\", \"s61\", \"s62\", \"scc\", \"memory\");\n")


print("// generated by gen_ubench_asm.py -- do not edit")
print(macro("KB_UB_TODAY", 16, 1, 1))
print(macro("KB_UB_ROWS2", 8, 2, 1))
print(macro("KB_UB_ROWS2W", 16, 2, 2))
print(macro("KB_UB_ROWS4", 8, 2, 2, shared_addr=True))
print(class_macro("KB_UB_CLASS"))
print(class_macro("KB_UB_CLASS16", jumps=False))
print(fixed_macro("KB_UB_FIXED0", 0))
print(fixed_macro("KB_UB_FIXED5", 5))
print(fixed_macro("KB_UB_FIXED16", 16))
print(branchy_macro("KB_UB_BRANCHY5", 0x1249))
print(branchy_macro("KB_UB_BRANCHY0", 0))
