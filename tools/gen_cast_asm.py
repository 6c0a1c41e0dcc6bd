#!/usr/bin/env python3
"""Emit csrc/rt_cast_asm.h: the hand-scheduled gfx950 triangle loop of World::cast (main.rs:183-262).

The loop body is written once below, in terms of symbolic register names, and instantiated for the two SGPR
buffers (A/B) that let triangle i+1 be fetched while triangle i is tested.  Everything about the arithmetic is
fixed by parity: IEEE binary32 mul/add/sub in the reference's order, the compiler's own correctly rounded
division sequence (v_div_scale / v_rcp / fma refinement / v_div_fmas / v_div_fixup), literal NaN behaviour of
the comparisons.  What is hand-made is the control and data flow around it:

  * one 112-byte scalar fetch per triangle (s_load_dwordx16 + x8 + x4, base + one offset register + immediates) into a
    buffer of 28 SGPRs, issued one triangle ahead; the other buffer is being tested meanwhile;
  * lane predicates live in SGPR pairs and are combined with s_and/s_andn2/s_or; EXEC is never narrowed for the
    arithmetic, so there is no save/restore nesting — wave-level exits (after culling, after t <= 0, after
    exclusion + nearest, after the bounding-sphere test, after each signed area) branch on SCC straight to the next triangle;
  * PLANE SHARING: a triangle flagged FOLLOWS (same n and d, bit for bit, as its predecessor: the second half of every
    square(), main.rs:741-746) skips the plane part — n.d, the divide, the plane point and the culling / t > 0 mask are
    those of the predecessor, still in registers; only exclusion, nearest, the bounding-sphere test and the signed areas
    are the triangle's own.  The reference scene's 28 non-dodecahedron triangles are 14 such pairs;
  * the accept is the only EXEC switch: six moves for the lanes that found a nearer hit.

Hazards handled by construction (gfx9 family): >= 4 instructions between the VCC-writing v_div_scale and
v_div_fmas; s_nop after v_rcp_f32 before its consumer (trans-use hazard on gfx940+); VALU-written masks are
only consumed by SALU (interlocked) and branches use SCC, never VCCZ/EXECZ.

    python tools/gen_cast_asm.py > homework-18-graphics-raytracer_amd/csrc/rt_cast_asm.h
    python tools/gen_cast_asm.py --count-stages > homework-18-graphics-raytracer_amd/csrc/rt_cast_asm_diag.h
"""

# ---- operand map of the asm statement (see cast_asm() in rt_cast.h) ----
# outputs: the nearest accepted hit so far: t, triangle index, and what the reference keeps of it — n.d (its sign is the
# backface flag) and the three signed areas (the barycentric numerators), exactly the values the accept was decided on
OUT = {"best_t": 0, "best_prim": 1, "best_nd": 2, "best_a0": 3, "best_a1": 4, "best_a2": 5}
TEMPS = {f"r{k}": 6 + k for k in range(14)}
_IN_NAMES = ["ox", "oy", "oz", "dx", "dy", "dz", "exid", "keep_back", "keep_front", "ex_if_back", "ex_if_front", "ptr", "n", "filter_ok"]
IN = {name: 20 + k for k, name in enumerate(_IN_NAMES)}

import sys
COUNT = "--count-stages" in sys.argv  # diagnostic build (RT_DIAG_STAGES): eight extra "+v" operands count how far each triangle test got
if COUNT:
    TEMPS.update({f"c{k}": 20 + k for k in range(8)})
    IN = {k: v + 8 for k, v in IN.items()}
SPHERE_FILTER = "--no-sphere-filter" not in sys.argv  # conservative per-triangle bounding-sphere rejection of the plane hit point
SLOW = {}

# fixed scalar registers (all in the clobber list)
# (s32/s33 are the ABI stack/frame pointers and s100/s101 are reserved by the compiler: stay inside s34..s99)
S_OFF, S_I = "s34", "s35"  # byte offset of the record pair being fetched (the base stays in the input operand); loop index
S_PLANE = "s[36:37]"       # lanes that passed culling and t > 0 on the current plane
S_ALIVE, S_T1, S_T2 = "s[38:39]", "s[96:97]", "s[98:99]"
S_EXSAVE = S_T2

BUF = {
    "A": {"x16": "s[40:55]", "x8": "s[56:63]", "x4": "s[64:67]", "base": 40},
    "B": {"x16": "s[68:83]", "x8": "s[84:91]", "x4": "s[92:95]", "base": 68},
}


def regs(buf):
    b = BUF[buf]["base"]
    names = ["nx", "ny", "nz", "d", "v0x", "v0y", "v0z", "obj", "v1x", "v1y", "v1z", "area", "v2x", "v2y", "v2z", "bq",
             "e0x", "e0y", "e0z", "bcx", "e1x", "e1y", "e1z", "bcy", "e2x", "e2y", "e2z", "bcz"]
    return {n: f"s{b + k}" for k, n in enumerate(names)}


def op(name):
    if name in OUT:
        return f"%{OUT[name]}"
    if name in TEMPS:
        return f"%{TEMPS[name]}"
    return f"%{IN[name]}"


def loads(buf, offset):
    b = BUF[buf]
    imm = lambda x: f" offset:{hex(x)}" if x else ""
    return [
        f"s_load_dwordx16 {b['x16']}, {op('ptr')}, {S_OFF}{imm(offset)}",
        f"s_load_dwordx8 {b['x8']}, {op('ptr')}, {S_OFF}{imm(offset + 0x40)}",
        f"s_load_dwordx4 {b['x4']}, {op('ptr')}, {S_OFF}{imm(offset + 0x60)}",
    ]


def dot_sv(dst, tmp, s3, v3):
    """dst = (s.x*v.x + s.y*v.y) + s.z*v.z   (cgmath dot, left to right)"""
    return [
        f"v_mul_f32 {dst}, {s3[0]}, {v3[0]}",
        f"v_mul_f32 {tmp}, {s3[1]}, {v3[1]}",
        f"v_add_f32 {dst}, {dst}, {tmp}",
        f"v_mul_f32 {tmp}, {s3[2]}, {v3[2]}",
        f"v_add_f32 {dst}, {dst}, {tmp}",
    ]


def area(dst, e, vtx, p, w, c, tmp, n):
    """dst = dot(cross(e, p - vtx), n)   (main.rs:219-221)"""
    out = [f"v_subrev_f32 {w[k]}, {vtx[k]}, {p[k]}" for k in range(3)]  # w = p - vtx
    out += [
        f"v_mul_f32 {c[0]}, {e[1]}, {w[2]}", f"v_mul_f32 {tmp}, {e[2]}, {w[1]}", f"v_sub_f32 {c[0]}, {c[0]}, {tmp}",
        f"v_mul_f32 {c[1]}, {e[2]}, {w[0]}", f"v_mul_f32 {tmp}, {e[0]}, {w[2]}", f"v_sub_f32 {c[1]}, {c[1]}, {tmp}",
        f"v_mul_f32 {c[2]}, {e[0]}, {w[1]}", f"v_mul_f32 {tmp}, {e[1]}, {w[0]}", f"v_sub_f32 {c[2]}, {c[2]}, {tmp}",
        f"v_mul_f32 {c[0]}, {n[0]}, {c[0]}", f"v_mul_f32 {c[1]}, {n[1]}, {c[1]}", f"v_add_f32 {c[0]}, {c[0]}, {c[1]}",
        f"v_mul_f32 {c[2]}, {n[2]}, {c[2]}", f"v_add_f32 {dst}, {c[0]}, {c[2]}",
    ]
    return out


def test(buf, label_next):
    """One triangle.  A triangle whose record says FOLLOWS (bit 31 of its object word: rt_scene_create sets it when the
    triangle's plane — n and d, bit for bit — is the previous triangle's, e.g. the second half of a square()) skips the
    plane part: nd, t, p and the mask of lanes that passed culling and t > 0 (S_PLANE) are still in registers."""
    T = regs(buf)
    n = (T["nx"], T["ny"], T["nz"])
    d = (op("dx"), op("dy"), op("dz"))
    o = (op("ox"), op("oy"), op("oz"))
    r = [op(f"r{k}") for k in range(14)]
    nd, num = r[13], r[12]  # num becomes t; nd, t and p stay live for the triangles that follow on the same plane
    t = num
    p = (r[0], r[1], r[2])
    L = []
    cnt = (lambda k: [f"v_add_u32 {op('c%d' % k)}, 1, {op('c%d' % k)}"]) if COUNT else (lambda k: [])
    L += [f"s_bitcmp1_b32 {T['obj']}, 31", f"s_cbranch_scc1 .Lcast_follow{buf}_%=", f".Lcast_lead{buf}_%=:"]
    # ---- the plane (main.rs:184-188, 202-205, 210): shared by every triangle that follows on it ----
    L += cnt(0)
    # nd = n . d ; bf = nd > 0
    L += dot_sv(nd, r[9], n, d)
    L += [f"v_cmp_lt_f32 vcc, 0, {nd}"]
    # S_PLANE = bf ? keep_back : keep_front          (culling, main.rs:185-188)
    L += [f"s_and_b64 {S_PLANE}, vcc, {op('keep_back')}", f"s_andn2_b64 {S_T1}, {op('keep_front')}, vcc", f"s_or_b64 {S_PLANE}, {S_PLANE}, {S_T1}",
          f"s_and_b64 {S_PLANE}, {S_PLANE}, exec", f"s_cbranch_scc0 {label_next}"]
    # num = d - n . o                              (main.rs:203-204)
    L += cnt(1)
    L += dot_sv(r[9], r[10], n, o)
    L += [f"v_sub_f32 {num}, {T['d']}, {r[9]}"]
    # t = num / nd, correctly rounded (the sequence hipcc emits for an IEEE f32 divide)
    q0, rc, e, q1, q = r[0], r[1], r[2], r[3], r[4]
    L += [f"v_div_scale_f32 {q0}, {S_T1}, {nd}, {nd}, {num}",
          f"v_rcp_f32 {rc}, {q0}",
          "s_nop 0",
          f"v_fma_f32 {e}, -{q0}, {rc}, 1.0",
          f"v_fmac_f32 {rc}, {e}, {rc}",
          f"v_div_scale_f32 {q1}, vcc, {num}, {nd}, {num}",
          f"v_mul_f32 {q}, {q1}, {rc}",
          f"v_fma_f32 {e}, -{q0}, {q}, {q1}",
          f"v_fmac_f32 {q}, {e}, {rc}",
          f"v_fma_f32 {q0}, -{q0}, {q}, {q1}",
          f"v_div_fmas_f32 {q0}, {q0}, {rc}, {q}",
          f"v_div_fixup_f32 {t}, {q0}, {nd}, {num}"]
    # S_PLANE &= !(t <= 0)   (NaN passes, main.rs:205)
    L += [f"v_cmp_nge_f32 vcc, 0, {t}", f"s_and_b64 {S_PLANE}, {S_PLANE}, vcc", f"s_cbranch_scc0 {label_next}"]
    # p = o + d * t
    L += cnt(2)
    for k in range(3):
        L += [f"v_mul_f32 {p[k]}, {d[k]}, {t}", f"v_add_f32 {p[k]}, {o[k]}, {p[k]}"]
    # ---- this triangle (leaders fall in; followers enter here with the plane's registers) ----
    L += [f".Lcast_tri{buf}_%=:"]
    # exclusion (main.rs:190-200): same primitive and (bf ? ex_if_back : ex_if_front).  Almost every triangle is excluded
    # by no lane at all: test that first and keep the mask algebra out of line.
    # nearest (main.rs:229-233): a lane that already holds a nearer hit cannot accept this triangle whatever its areas are
    # (one more `continue`; the order of the rejections is immaterial); best_t = NaN while None, equal t passes (the later
    # primitive wins a tie)
    L += [f"v_cmp_eq_u32_e64 {S_T2}, {S_I}, {op('exid')}",
          f"v_cmp_nlt_f32 vcc, {op('best_t')}, {t}",
          f"s_cmp_lg_u64 {S_T2}, 0", f"s_cbranch_scc1 .Lcast_excl{buf}_%=",
          f"s_and_b64 {S_ALIVE}, {S_PLANE}, vcc",
          f".Lcast_excl{buf}_back_%=:",
          f"s_cbranch_scc0 {label_next}"]
    SLOW[buf] = [f".Lcast_excl{buf}_%=:",
                 f"s_and_b64 {S_ALIVE}, {S_PLANE}, vcc",
                 f"v_cmp_lt_f32 vcc, 0, {nd}",
                 f"s_and_b64 {S_T1}, vcc, {op('ex_if_back')}", f"s_and_b64 {S_T1}, {S_T1}, {S_T2}", f"s_andn2_b64 {S_ALIVE}, {S_ALIVE}, {S_T1}",
                 f"s_andn2_b64 {S_T1}, {op('ex_if_front')}, vcc", f"s_and_b64 {S_T1}, {S_T1}, {S_T2}", f"s_andn2_b64 {S_ALIVE}, {S_ALIVE}, {S_T1}",
                 f"s_branch .Lcast_excl{buf}_back_%=",
                 # A triangle on the previous triangle's plane.  Bit 30: its n and d equal the predecessor's only up to the
                 # signs of zero components ((-0, 1, 0) after (0, 1, 0): the two halves of most square()s).  Every product and sum
                 # the plane part forms is then the same too, except where ALL its terms are zeros: n.d = +-0 (t = +-inf or NaN, which
                 # the reference keeps apart) — or num = +-0, i.e. t = +-0, rejected by `t <= 0` either way.  So if any lane has
                 # n.d == 0 the triangle evaluates its own plane after all; otherwise the predecessor's values ARE its values.
                 f".Lcast_follow{buf}_%=:",
                 f"s_bitcmp1_b32 {T['obj']}, 30", f"s_cbranch_scc0 .Lcast_share{buf}_%=",
                 f"v_cmp_eq_f32 vcc, 0, {nd}", f"s_and_b64 {S_T1}, vcc, exec", f"s_cbranch_scc1 .Lcast_lead{buf}_%=",
                 # nothing to do unless some lane passed culling and t > 0 on the plane
                 f".Lcast_share{buf}_%=:",
                 f"s_cmp_lg_u64 {S_PLANE}, 0", f"s_cbranch_scc1 .Lcast_tri{buf}_%=", f"s_branch {label_next}"]
    w = (r[3], r[4], r[5])
    L += cnt(7) if COUNT else []
    if SPHERE_FILTER:
        # Conservative rejection (rt_device_scene.h "bounding sphere"): the plane hit point p of a lane lies outside the
        # triangle for sure when q = |p - c|^2 exceeds bq = 1.05 R^2 (c, R: the triangle's enclosing circle) — at least one
        # signed area is then negative by a margin far above its rounding error.  q must also be finite and moderate
        # (< 1e30): a non-finite p makes the areas NaN, which the reference ACCEPTS.  NaN q fails both compares (kept).
        # bq = +inf switches the filter off for a triangle (degenerate, sliver or huge coordinates) — skipped outright.
        L += [f"s_cmp_eq_u32 {T['bq']}, 0x7f800000", f"s_cbranch_scc1 .Lcast_nofilter{buf}_%="]
        L += [f"v_subrev_f32 {w[0]}, {T['bcx']}, {p[0]}", f"v_subrev_f32 {w[1]}, {T['bcy']}, {p[1]}", f"v_subrev_f32 {w[2]}, {T['bcz']}, {p[2]}",
              f"v_mul_f32 {w[0]}, {w[0]}, {w[0]}", f"v_mul_f32 {w[1]}, {w[1]}, {w[1]}", f"v_add_f32 {w[0]}, {w[0]}, {w[1]}",
              f"v_mul_f32 {w[2]}, {w[2]}, {w[2]}", f"v_add_f32 {w[0]}, {w[0]}, {w[2]}",
              f"v_cmp_lt_f32_e64 {S_T1}, {T['bq']}, {w[0]}", f"v_cmp_gt_f32 vcc, 0x7149f2ca, {w[0]}",
              f"s_and_b64 {S_T1}, {S_T1}, vcc", f"s_and_b64 {S_T1}, {S_T1}, {op('filter_ok')}",
              f"s_andn2_b64 {S_ALIVE}, {S_ALIVE}, {S_T1}", f"s_cbranch_scc0 {label_next}",
              f".Lcast_nofilter{buf}_%=:"]
    e0 = (T["e0x"], T["e0y"], T["e0z"]); e1 = (T["e1x"], T["e1y"], T["e1z"]); e2 = (T["e2x"], T["e2y"], T["e2z"])
    v0 = (T["v0x"], T["v0y"], T["v0z"]); v1 = (T["v1x"], T["v1y"], T["v1z"]); v2 = (T["v2x"], T["v2y"], T["v2z"])
    # the three signed areas, each followed by its own wave-level exit: a lane stays alive unless its area < 0
    # (v_cmp_ngt 0, a  ==  !(a < 0): NaN passes, main.rs:224); coherent waves usually leave after the first
    L += cnt(3)
    L += area(r[6], e0, v1, p, w, (r[6], r[9], r[10]), r[11], n)
    L += [f"v_cmp_ngt_f32 vcc, 0, {r[6]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
    L += cnt(4)
    L += area(r[7], e1, v2, p, w, (r[7], r[9], r[10]), r[11], n)
    L += [f"v_cmp_ngt_f32 vcc, 0, {r[7]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
    L += cnt(5)
    L += area(r[8], e2, v0, p, w, (r[8], r[9], r[10]), r[11], n)
    L += [f"v_cmp_ngt_f32 vcc, 0, {r[8]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
    # accept for the lanes in alive
    L += cnt(6)
    L += [f"s_mov_b64 {S_EXSAVE}, exec", f"s_mov_b64 exec, {S_ALIVE}",
          f"v_mov_b32 {op('best_t')}, {t}", f"v_mov_b32 {op('best_prim')}, {S_I}", f"v_mov_b32 {op('best_nd')}, {nd}",
          f"v_mov_b32 {op('best_a0')}, {r[6]}", f"v_mov_b32 {op('best_a1')}, {r[7]}", f"v_mov_b32 {op('best_a2')}, {r[8]}",
          f"s_mov_b64 exec, {S_EXSAVE}"]
    return L


def main():
    L = []
    N = op("n")
    L += [f"s_mov_b32 {S_OFF}, 0", f"s_mov_b32 {S_I}, 0", f"s_mov_b64 {S_PLANE}, 0",
          f"s_cmp_eq_u32 {N}, 0", "s_cbranch_scc1 .Lcast_done_%="]
    L += loads("A", 0)
    # whatever its record says, the first triangle of a call has no predecessor in registers: it is a leader
    L += ["s_waitcnt lgkmcnt(0)", f"s_bitset0_b32 {regs('A')['obj']}, 31"]
    L += [".Lcast_loop_%=:", "s_waitcnt lgkmcnt(0)"]
    L += loads("B", 0x80)
    L += test("A", ".Lcast_nextA_%=")
    L += [".Lcast_nextA_%=:", f"s_add_u32 {S_I}, {S_I}, 1", f"s_cmp_ge_u32 {S_I}, {N}", "s_cbranch_scc1 .Lcast_done_%=",
          "s_waitcnt lgkmcnt(0)", f"s_add_u32 {S_OFF}, {S_OFF}, 0x100"]
    L += loads("A", 0)
    L += test("B", ".Lcast_nextB_%=")
    L += [".Lcast_nextB_%=:", f"s_add_u32 {S_I}, {S_I}, 1", f"s_cmp_lt_u32 {S_I}, {N}", "s_cbranch_scc1 .Lcast_loop_%=",
          "s_branch .Lcast_done_%="]
    for b in ("A", "B"):
        L += SLOW.get(b, [])
    L += [".Lcast_done_%=:", "s_waitcnt lgkmcnt(0)"]

    clobbers = [f"s{k}" for k in range(34, 100)] + ["vcc", "scc"]
    print("/* GENERATED by tools/gen_cast_asm.py — do not edit; edit the generator. */")
    print("#ifndef RT_CAST_ASM_H")
    print("#define RT_CAST_ASM_H")
    print("#define RT_CAST_ASM_TEXT \\")
    for line in L:
        print(f'    "{line}\\n\\t" \\')
    print('    ""')
    print("#define RT_CAST_ASM_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers))
    print(f"#define RT_CAST_ASM_INSTRUCTIONS {len([l for l in L if not l.endswith(':')])}")
    print("#endif")


if __name__ == "__main__":
    main()
