#!/usr/bin/env python3
"""Generates instr_cost2.hip: one pinned stream per instruction form ({i} = one of 8 in/out vector registers, {p} = one of 8 register pairs, {B} = another
vector register, {S} = a scalar register, {L} = an LDS address, masks in s[20:35]); see instr_cost.hip for the method."""
import sys

FORMS = [
    # (name, template for ONE instruction; {i} -> %0..%7, {B} -> %8, {S} -> %9, {m} -> s[20+2i:21+2i], {d} -> s(20+i))
    ("v_add_f32 v,v,v", "v_add_f32 {i}, {i}, {B}"),
    ("v_sub_f32", "v_sub_f32 {i}, {i}, {B}"),
    ("v_subrev_f32", "v_subrev_f32 {i}, {i}, {B}"),
    ("v_min_f32", "v_min_f32 {i}, {i}, {B}"),
    ("v_max_f32", "v_max_f32 {i}, {i}, {B}"),
    ("v_mul_f32 v,s,v", "v_mul_f32 {i}, {S}, {i}"),
    ("v_or_b32", "v_or_b32 {i}, {i}, {B}"),
    ("v_xor_b32", "v_xor_b32 {i}, {i}, {B}"),
    ("v_sub_u32", "v_sub_u32 {i}, {i}, {B}"),
    ("v_lshrrev_b32 v,5,v", "v_lshrrev_b32 {i}, 5, {i}"),
    ("v_lshrrev_b32 v,v,v", "v_lshrrev_b32 {i}, {B}, {i}"),
    ("v_ashrrev_i32 v,31,v", "v_ashrrev_i32 {i}, 31, {i}"),
    ("v_lshlrev_b32 v,v,v", "v_lshlrev_b32 {i}, {B}, {i}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {i}, {i}, {B}"),
    ("v_mul_i32_i24", "v_mul_i32_i24 {i}, {i}, {B}"),
    ("v_min_i32", "v_min_i32 {i}, {i}, {B}"),
    ("v_max_i32", "v_max_i32 {i}, {i}, {B}"),
    ("v_min_u32", "v_min_u32 {i}, {i}, {B}"),
    ("v_max_u32", "v_max_u32 {i}, {i}, {B}"),
    ("v_mov_b32 v,v", "v_mov_b32 {i}, {B}"),
    ("v_mov_b32 v,const", "v_mov_b32 {i}, 1.0"),
    ("v_mov_b32 v,literal", "v_mov_b32 {i}, 0x12345"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 {i}, {i}"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {i}, {i}"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {i}, {i}"),
    ("v_trunc_f32", "v_trunc_f32 {i}, {i}"),
    ("v_floor_f32", "v_floor_f32 {i}, {i}"),
    ("v_fract_f32", "v_fract_f32 {i}, {i}"),
    ("v_not_b32", "v_not_b32 {i}, {i}"),
    ("v_sqrt_f32", "v_sqrt_f32 {i}, {i}"),
    ("v_rsq_f32", "v_rsq_f32 {i}, {i}"),
    ("v_add3_u32", "v_add3_u32 {i}, {i}, {B}, {B}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {i}, {i}, 3, {B}"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {i}, {i}, {B}, 2"),
    ("v_and_or_b32", "v_and_or_b32 {i}, {i}, {B}, {B}"),
    ("v_or3_b32", "v_or3_b32 {i}, {i}, {B}, {B}"),
    ("v_xad_u32", "v_xad_u32 {i}, {i}, {B}, {B}"),
    ("v_max3_f32", "v_max3_f32 {i}, {i}, {B}, {B}"),
    ("v_mad_u32_u24 v,v,const,v", "v_mad_u32_u24 {i}, {i}, 32, {B}"),
    ("v_alignbit_b32", "v_alignbit_b32 {i}, {i}, {B}, 7"),
    ("v_perm_b32", "v_perm_b32 {i}, {i}, {B}, {B}"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 {i}, -1, {i}"),
    ("v_div_scale_f32", "v_div_scale_f32 {i}, vcc, {i}, {B}, {i}"),
    ("v_div_fmas_f32", "v_div_fmas_f32 {i}, {i}, {B}, {B}"),
    ("v_div_fixup_f32", "v_div_fixup_f32 {i}, {i}, {B}, {B}"),
    ("v_ldexp_f32", "v_ldexp_f32 {i}, {i}, 2"),
    ("v_mul_f32 e64 neg", "v_mul_f32 {i}, -{i}, {B}"),
    ("v_add_f32 e64 clamp", "v_add_f32 {i}, {i}, {B} clamp"),
    ("v_add_f32 e64 mul:2", "v_add_f32 {i}, {i}, {B} mul:2"),
    ("v_cmp_class_f32 s", "v_cmp_class_f32 {m}, {i}, {B}"),
    ("v_cmp_eq_u32 vcc,const", "v_cmp_eq_u32 vcc, 3, {i}"),
    ("v_cmp_lt_f32 s + s_and chain (2 instr)", "v_cmp_lt_f32 {m}, {i}, {B}\n s_and_b64 s[36:37], s[36:37], {m}"),
    ("v_cndmask e64 consts", "v_cndmask_b32 {i}, 0, 1.0, {m}"),
    ("v_cndmask vcc after v_cmp (2 instr, indep)", "v_cmp_lt_f32 vcc, {B}, {i}\n v_cndmask_b32 {i}, {i}, {B}, vcc"),
    ("v_lshlrev_b64", "v_lshlrev_b64 {p}, 3, {p}"),
    ("v_pk_add_f32", "v_pk_add_f32 {p}, {p}, {p}"),
    ("v_pk_mul_f32", "v_pk_mul_f32 {p}, {p}, {p}"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {p}, {p}, {p}, {p}"),
    ("v_pk_add_u16", "v_pk_add_u16 {i}, {i}, {B}"),
    ("v_pk_sub_i16", "v_pk_sub_i16 {i}, {i}, {B}"),
    ("v_pk_min_u16", "v_pk_min_u16 {i}, {i}, {B}"),
    ("v_pk_max_i16", "v_pk_max_i16 {i}, {i}, {B}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {i}, {i}, {B}, {B}"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {i}, 3, {i}"),
    ("v_pk_mul_lo_u16", "v_pk_mul_lo_u16 {i}, {i}, {B}"),
    ("v_add_f64", "v_add_f64 {p}, {p}, {p}"),
    ("v_mul_f64", "v_mul_f64 {p}, {p}, {p}"),
    ("v_cvt_f64_f32", "v_cvt_f64_f32 {p}, {i}"),
    ("v_cvt_f32_f64", "v_cvt_f32_f64 {i}, {p}"),
    ("v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp {i}, {i} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_u32 sdwa", "v_add_u32_sdwa {i}, {i}, {B} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 {d}, {i}"),
    ("v_writelane_b32", "v_writelane_b32 {i}, {S}, 3"),
    ("fast+slow alternating (v_add_f32, v_med3)", "v_add_f32 {i}, {i}, {B}\n v_med3_i32 {i}, {i}, {B}, {B}"),
    ("fast+SALU alternating (v_add_f32, s_and_b64)", "v_add_f32 {i}, {i}, {B}\n s_and_b64 {m}, {m}, s[36:37]"),
    ("slow+SALU alternating (v_med3, s_and_b64)", "v_med3_i32 {i}, {i}, {B}, {B}\n s_and_b64 {m}, {m}, s[36:37]"),
    ("fast,fast,SALU (2:1)", "v_add_f32 {i}, {i}, {B}\n v_mul_f32 {i}, {i}, {B}\n s_and_b64 {m}, {m}, s[36:37]"),
    ("s_add_u32", "s_add_u32 {d}, {d}, 3"),
    ("s_bcnt1_i32_b64", "s_bcnt1_i32_b64 {d}, {m}"),
    ("s_cmp + s_cselect", "s_cmp_lt_u32 {d}, 77\n s_cselect_b32 {d}, {d}, 5"),
    ("s_nop 0", "s_nop 0"),
    ("ds_write_b32 (8, then wait)", "ds_write_b32 {L}, {i} offset:{o}"),
    ("ds_read_b64 (8, then wait)", "ds_read_b64 {p}, {L} offset:{o}"),
    ("ds_bpermute_b32 (8, then wait)", "ds_bpermute_b32 {i}, {L}, {i}"),
]
WAIT = {"ds_write_b32 (8, then wait)", "ds_read_b64 (8, then wait)", "ds_bpermute_b32 (8, then wait)"}


def body(tmpl, name):
    out = []
    for i in range(8):
        t = tmpl.replace("{i}", "%%%d" % i).replace("{m}", "s[%d:%d]" % (20 + 2 * i, 21 + 2 * i)).replace("{d}", "s%d" % (20 + i))
        t = t.replace("{p}", "%%%d" % (8 + i)).replace("{o}", str(256 * i))
        t = t.replace("{B}", "%16").replace("{S}", "%17").replace("{L}", "%18")
        out.append(t)
    s = "\\n ".join(x.replace("\n", "\\n") for x in out)
    if name in WAIT:
        s += "\\n s_waitcnt lgkmcnt(0)"
    return s


def count(tmpl):
    return 8 * (tmpl.count("\n") + 1)


src = open(sys.argv[1]).read() if len(sys.argv) > 1 else None
print("// generated by gen_instr_cost.py -- do not edit")
print("#define STREAMS(X) \\")
for n, (name, tmpl) in enumerate(FORMS):
    # careful: the B/S/L substitution must not touch mnemonics: templates use them only as standalone operands
    print('    X(%d, "%s", %d, "%s") \\' % (n, name, count(tmpl), body(tmpl, name)))
print("")
