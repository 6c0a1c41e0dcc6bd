#!/usr/bin/env python3
"""Emit csrc/rt_cast_asm.h: the hand-scheduled gfx950 triangle loop of World::cast (main.rs:183-262).

The loop body is written once below, in terms of symbolic register names, and instantiated for the two SGPR
buffers (A/B) that let triangle i+1 be fetched while triangle i is tested.  Everything about the arithmetic is
fixed by parity: IEEE binary32 mul/add/sub in the reference's order, the compiler's own correctly rounded
division sequence (v_div_scale / v_rcp / fma refinement / v_div_fmas / v_div_fixup), literal NaN behaviour of
the comparisons.  What is hand-made is the control and data flow around it:

  * one 112-byte scalar fetch per triangle (s_load_dwordx16 + x8 + x4) into a buffer of 28 SGPRs, issued one
    triangle ahead; the other buffer is being tested meanwhile;
  * lane predicates live in SGPR pairs and are combined with s_and/s_andn2/s_or; EXEC is never narrowed for the
    arithmetic, so there is no save/restore nesting — three wave-level exits (after culling/exclusion, after
    t <= 0, after inside+nearest) branch on SCC straight to the next triangle;
  * the accept is the only EXEC switch: five instructions for the lanes that found a nearer hit.

Hazards handled by construction (gfx9 family): >= 4 instructions between the VCC-writing v_div_scale and
v_div_fmas; s_nop after v_rcp_f32 before its consumer (trans-use hazard on gfx940+); VALU-written masks are
only consumed by SALU (interlocked) and branches use SCC, never VCCZ/EXECZ.

    python tools/gen_cast_asm.py > homework-18-graphics-raytracer_amd/csrc/rt_cast_asm.h
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
EARLY_AREA_EXITS = "--no-early-area-exits" not in sys.argv
OUT_OF_LINE_EXCLUSION = "--inline-exclusion" not in sys.argv
SPHERE_FILTER = "--no-sphere-filter" not in sys.argv  # conservative per-triangle bounding-sphere rejection of the plane hit point
NEAREST_EARLY = "--nearest-late" not in sys.argv      # test !(best_t < t) right after t > 0 instead of after the areas
ILP = "--ilp" in sys.argv  # interleave the three signed-area chains (and p) for a lone wave; one exit after all three
XT = [f"v{k}" for k in range(148, 168)]  # extra temporaries of the ILP form: fixed VGPRs, named in the clobber list
SLOW = {}

# fixed scalar registers (all in the clobber list)
# (s32/s33 are the ABI stack/frame pointers and s100/s101 are reserved by the compiler: stay inside s34..s99)
S_PTR = "s[34:35]"
S_PTR_LO, S_PTR_HI = "s34", "s35"
S_I, S_N = "s36", "s37"
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
    return [
        f"s_load_dwordx16 {b['x16']}, {S_PTR}, {hex(offset)}",
        f"s_load_dwordx8 {b['x8']}, {S_PTR}, {hex(offset + 0x40)}",
        f"s_load_dwordx4 {b['x4']}, {S_PTR}, {hex(offset + 0x60)}",
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
    T = regs(buf)
    n = (T["nx"], T["ny"], T["nz"])
    d = (op("dx"), op("dy"), op("dz"))
    o = (op("ox"), op("oy"), op("oz"))
    r = [op(f"r{k}") for k in range(14)]
    nd, num = r[13], r[12]  # num becomes t; nd stays live to the accept
    t = num
    L = []
    cnt = (lambda k: [f"v_add_u32 {op('c%d' % k)}, 1, {op('c%d' % k)}"]) if COUNT else (lambda k: [])
    L += cnt(0)
    # nd = n . d ; bf = nd > 0
    L += dot_sv(nd, r[9], n, d)
    L += [f"v_cmp_lt_f32 vcc, 0, {nd}"]
    # alive = bf ? keep_back : keep_front          (culling, main.rs:185-188)
    L += [f"s_and_b64 {S_ALIVE}, vcc, {op('keep_back')}", f"s_andn2_b64 {S_T1}, {op('keep_front')}, vcc", f"s_or_b64 {S_ALIVE}, {S_ALIVE}, {S_T1}"]
    # exclusion: same primitive and (bf ? ex_if_back : ex_if_front)   (main.rs:190-200)
    # Almost every triangle is excluded by no lane at all: test that first and keep the mask algebra out of line.
    excl = [f"s_and_b64 {S_T1}, vcc, {op('ex_if_back')}", f"s_and_b64 {S_T1}, {S_T1}, {S_T2}", f"s_andn2_b64 {S_ALIVE}, {S_ALIVE}, {S_T1}",
            f"s_andn2_b64 {S_T1}, {op('ex_if_front')}, vcc", f"s_and_b64 {S_T1}, {S_T1}, {S_T2}", f"s_andn2_b64 {S_ALIVE}, {S_ALIVE}, {S_T1}"]
    L += [f"v_cmp_eq_u32_e64 {S_T2}, {S_I}, {op('exid')}"]
    if OUT_OF_LINE_EXCLUSION:
        L += [f"s_cmp_lg_u64 {S_T2}, 0", f"s_cbranch_scc1 .Lcast_excl{buf}_%=", f".Lcast_excl{buf}_back_%=:"]
        SLOW[buf] = [f".Lcast_excl{buf}_%=:"] + excl + [f"s_branch .Lcast_excl{buf}_back_%="]
    else:
        L += excl
    L += [f"s_and_b64 {S_ALIVE}, {S_ALIVE}, exec", f"s_cbranch_scc0 {label_next}"]
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
    # alive &= !(t <= 0)   (NaN passes, main.rs:205)
    if NEAREST_EARLY:
        # ... and alive &= !(best_t < t): a lane that already holds a nearer hit cannot accept this triangle whatever its
        # areas are (main.rs:229-233 is one more `continue`; the order of the rejections is immaterial), best_t = NaN while None
        L += [f"v_cmp_nge_f32 vcc, 0, {t}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc",
              f"v_cmp_nlt_f32 vcc, {op('best_t')}, {t}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
    else:
        L += [f"v_cmp_nge_f32 vcc, 0, {t}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
    # p = o + d * t
    L += cnt(2)
    p = (r[0], r[1], r[2])
    if ILP:
        L += [f"v_mul_f32 {p[k]}, {d[k]}, {t}" for k in range(3)] + [f"v_add_f32 {p[k]}, {o[k]}, {p[k]}" for k in range(3)]
    else:
        for k in range(3):
            L += [f"v_mul_f32 {p[k]}, {d[k]}, {t}", f"v_add_f32 {p[k]}, {o[k]}, {p[k]}"]
    w = (r[3], r[4], r[5])
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
    if ILP:
        # p = o + d*t was emitted component by component above; the three areas are independent chains: give each
        # its own temporaries and issue them round-robin so a lone wave always has an independent instruction next
        chains = [
            area(r[6], e0, v1, p, (XT[0], XT[1], XT[2]), (r[6], XT[3], XT[4]), XT[5], n),
            area(r[7], e1, v2, p, (XT[6], XT[7], XT[8]), (r[7], XT[9], XT[10]), XT[11], n),
            area(r[8], e2, v0, p, (XT[12], XT[13], XT[14]), (r[8], XT[15], XT[16]), XT[17], n),
        ]
        for k in range(len(chains[0])):
            for c in chains:
                L.append(c[k])
        L += [f"v_min3_f32 {r[9]}, {r[6]}, {r[7]}, {r[8]}", f"v_cmp_ngt_f32 vcc, 0, {r[9]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc"]
    else:
        # the three signed areas, each followed by its own wave-level exit: a lane stays alive unless its area < 0
        # (v_cmp_ngt 0, a  ==  !(a < 0): NaN passes, main.rs:224); coherent waves usually leave after the first
        early = EARLY_AREA_EXITS
        L += cnt(3)
        L += area(r[6], e0, v1, p, w, (r[6], r[9], r[10]), r[11], n)
        if early:
            L += [f"v_cmp_ngt_f32 vcc, 0, {r[6]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
        L += cnt(4)
        L += area(r[7], e1, v2, p, w, (r[7], r[9], r[10]), r[11], n)
        if early:
            L += [f"v_cmp_ngt_f32 vcc, 0, {r[7]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
        L += cnt(5)
        L += area(r[8], e2, v0, p, w, (r[8], r[9], r[10]), r[11], n)
        if early:
            L += [f"v_cmp_ngt_f32 vcc, 0, {r[8]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc"]
        else:
            L += [f"v_min3_f32 {r[9]}, {r[6]}, {r[7]}, {r[8]}", f"v_cmp_ngt_f32 vcc, 0, {r[9]}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc"]
    if NEAREST_EARLY:
        L += [f"s_cbranch_scc0 {label_next}"]
    else:
        # nearest: !(best_t < t), best_t = NaN while None (main.rs:229-233)
        L += [f"v_cmp_nlt_f32 vcc, {op('best_t')}, {t}", f"s_and_b64 {S_ALIVE}, {S_ALIVE}, vcc", f"s_cbranch_scc0 {label_next}"]
    # accept for the lanes in alive
    L += cnt(6)
    L += [f"s_mov_b64 {S_EXSAVE}, exec", f"s_mov_b64 exec, {S_ALIVE}",
          f"v_mov_b32 {op('best_t')}, {t}", f"v_mov_b32 {op('best_prim')}, {S_I}", f"v_mov_b32 {op('best_nd')}, {nd}",
          f"v_mov_b32 {op('best_a0')}, {r[6]}", f"v_mov_b32 {op('best_a1')}, {r[7]}", f"v_mov_b32 {op('best_a2')}, {r[8]}",
          f"s_mov_b64 exec, {S_EXSAVE}"]
    return L


def main():
    L = []
    L += [f"s_mov_b64 {S_PTR}, {op('ptr')}", f"s_mov_b32 {S_N}, {op('n')}", f"s_mov_b32 {S_I}, 0",
          f"s_cmp_eq_u32 {S_N}, 0", "s_cbranch_scc1 .Lcast_done_%="]
    L += loads("A", 0)
    L += [".Lcast_loop_%=:", "s_waitcnt lgkmcnt(0)"]
    L += loads("B", 0x80)
    L += test("A", ".Lcast_nextA_%=")
    L += [".Lcast_nextA_%=:", f"s_add_u32 {S_I}, {S_I}, 1", f"s_cmp_ge_u32 {S_I}, {S_N}", "s_cbranch_scc1 .Lcast_done_%=",
          "s_waitcnt lgkmcnt(0)", f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 0x100", f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0"]
    L += loads("A", 0)
    L += test("B", ".Lcast_nextB_%=")
    L += [".Lcast_nextB_%=:", f"s_add_u32 {S_I}, {S_I}, 1", f"s_cmp_lt_u32 {S_I}, {S_N}", "s_cbranch_scc1 .Lcast_loop_%=",
          "s_branch .Lcast_done_%="]
    for b in ("A", "B"):
        L += SLOW.get(b, [])
    L += [".Lcast_done_%=:", "s_waitcnt lgkmcnt(0)"]

    clobbers = [f"s{k}" for k in range(34, 100)] + ["vcc", "scc"] + (XT[:18] if ILP else [])
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
