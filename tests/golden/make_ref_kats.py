#!/usr/bin/env python3
"""Transcribe the reference's own known-answer vectors into tests/golden/ref_kats.json.

Runs ONLY in the build container (it reads /root/reference, which does not exist on the GPU box).
It extracts DATA -- the matrix literals that the reference's unit tests feed to and expect from the
deterministic helpers on the preimage-sampling path -- never source text.
Each record names the reference test it comes from (file:line of the `fn`).

usage: python tests/golden/make_ref_kats.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
U64_MAX = 2**64 - 1


def rust_string_literals(body):
    """All "..." literals of a Rust snippet with `\\<newline><indent>` continuations folded."""
    out = []
    for mt in re.finditer(r'"((?:[^"\\]|\\.|\\\n)*)"', body, flags=re.S):
        s = re.sub(r"\\\n\s*", "", mt.group(1))
        out.append(s)
    return out


def fn_body(text, name):
    mt = re.search(r"fn\s+" + re.escape(name) + r"\s*\(", text)
    if not mt:
        raise KeyError(name)
    i = text.index("{", mt.end())
    depth, j = 0, i
    while True:
        c = text[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    line = text.count("\n", 0, mt.start()) + 1
    return text[i:j + 1], line


def parse_matz(s):
    """'[[1, 2],[3, 4]] mod 8' -> (rows, modulus|None); '{}' placeholders -> u64::MAX."""
    s = s.replace("{}", str(U64_MAX))
    mod = None
    if " mod " in s:
        s, modtxt = s.split(" mod ")
        mod = int(modtxt.strip())
    rows = []
    for row in re.findall(r"\[([^\[\]]*)\]", s):
        rows.append([int(x) for x in row.split(",") if x.strip() != ""])
    return rows, mod


def parse_matpoly(s):
    """'[[4  -1 7 6 -8, 0, 1  1]]' -> rows of coefficient lists (constant term first)."""
    rows = []
    for row in re.findall(r"\[([^\[\]]*)\]", s):
        polys = []
        for ent in row.split(","):
            toks = ent.split()
            if not toks or toks == ["0"]:
                polys.append([])
            else:
                n = int(toks[0])
                coeffs = [int(t) for t in toks[1:]]
                assert len(coeffs) == n, (ent, toks)
                polys.append(coeffs)
        rows.append(polys)
    return rows


def main():
    kats = {"_about": "data transcribed by tests/golden/make_ref_kats.py from the reference's unit tests "
                      "(qfall/tools @ /root/reference); file:line = the #[test] fn each record comes from"}

    def load(rel):
        with open(os.path.join(REF, rel)) as fh:
            return fh.read()

    # ---- gadget_classical.rs
    rel = "src/sample/g_trapdoor/gadget_classical.rs"
    t = load(rel)
    for name, (k, base) in {"correctness_base_2": (5, 2), "correctness_base_5": (4, 5)}.items():
        body, line = fn_body(t, name)
        rows, _ = parse_matz(rust_string_literals(body)[0])
        kats[f"gen_gadget_vec/{name}"] = {"src": f"{rel}:{line}", "k": k, "base": base, "expect": [r[0] for r in rows]}
    for name, (n, k, base) in {"correctness_base_2_3x3": (3, 3, 2), "correctness_base_3_2x5": (2, 5, 3)}.items():
        body, line = fn_body(t, name)
        rows, _ = parse_matz(rust_string_literals(body)[0])
        kats[f"gen_gadget_mat/{name}"] = {"src": f"{rel}:{line}", "n": n, "k": k, "base": base, "expect": rows}
    body, line = fn_body(t, "returns_correct_solution_vec")
    kats["find_solution_gadget_vec/returns_correct_solution_vec"] = {
        "src": f"{rel}:{line}", "k": 5, "base": 3, "q": 125, "values": list(range(0, 124)),
        "property": "gen_gadget_vec(k,base)^t * sol == least non-negative residue of value"}
    body, line = fn_body(t, "returns_correct_solution_mat")
    rows, mod = parse_matz(rust_string_literals(body)[0])
    kats["find_solution_gadget_mat/returns_correct_solution_mat"] = {
        "src": f"{rel}:{line}", "k": 5, "base": 3, "q": mod, "value": rows,
        "property": "gen_gadget_mat(3,k,base) * sol == value"}
    sb = {"base_2_power_two": dict(n=2, q=16, k=4, base=2),
          "base_2_arbitrary": dict(n=1, q=0b1100110, k=7, base=2),
          "base_5_power_5": dict(n=1, q=625, k=4, base=5),
          "base_5_arbitrary": dict(n=1, q=int("4123", 5), k=4, base=5)}
    # the four tests of `mod test_short_basis_gadget` (function names repeat in short_basis_ring.rs)
    mod_start = t.index("mod test_short_basis_gadget")
    for name, prm in sb.items():
        body, line = fn_body(t[mod_start:], name)
        line += t.count("\n", 0, mod_start)
        rows, _ = parse_matz(rust_string_literals(body)[-1])
        kats[f"short_basis_gadget/{name}"] = {"src": f"{rel}:{line}", **prm, "expect": rows}

    # ---- short_basis_classical.rs: fixed (A, R), n = 2, q = 8
    rel = "src/sample/g_trapdoor/short_basis_classical.rs"
    t = load(rel)
    body, line = fn_body(t, "get_fixed_trapdoor_for_tag_identity")
    lits = rust_string_literals(body)
    a_rows, a_mod = parse_matz(lits[0])
    r_rows, _ = parse_matz(lits[1])
    fixed = {"n": 2, "q": a_mod, "A": a_rows, "R": r_rows}
    body, line_l = fn_body(t, "working_sa_l")
    sa_l, _ = parse_matz(rust_string_literals(body)[0])
    body, line_r = fn_body(t, "working_sa_r_identity")
    sa_r, _ = parse_matz(rust_string_literals(body)[0])
    kats["short_basis_classical/working_sa_l"] = {"src": f"{rel}:{line_l}", **fixed, "expect": sa_l}
    kats["short_basis_classical/working_sa_r_identity"] = {"src": f"{rel}:{line_r}", **fixed, "expect": sa_r}
    body, line = fn_body(t, "working_example_tag_identity")
    a_rows2, a_mod2 = parse_matz(rust_string_literals(body)[0])
    kats["short_basis_classical/compute_w_working_example_tag_identity"] = {
        "src": f"{rel}:{line}", "n": 2, "q": a_mod2, "A": a_rows2, "property": "G W == -A[I|0] mod q"}

    # ---- short_basis_ring.rs: fixed (a, r, e), n = 4, q = 16
    rel = "src/sample/g_trapdoor/short_basis_ring.rs"
    t = load(rel)
    body, line = fn_body(t, "get_fixed_trapdoor")
    lits = rust_string_literals(body)
    fixed = {"n": 4, "q": 16, "a": parse_matpoly(lits[0])[0], "r": parse_matpoly(lits[1])[0],
             "e": parse_matpoly(lits[2])[0]}
    gen_mod = t.index("mod test_gen_sa")
    body, line_l = fn_body(t[gen_mod:], "working_sa_l")
    line_l += t.count("\n", 0, gen_mod)
    kats["short_basis_ring/working_sa_l"] = {
        "src": f"{rel}:{line_l}", **fixed,
        "note": "the reference test calls gen_sa_l(&r, &e) although the signature is gen_sa_l(e, r): "
                "the expected first row therefore carries the fixture's r, the second its e",
        "expect": parse_matpoly(rust_string_literals(body)[0])}
    body, line_r = fn_body(t[gen_mod:], "working_sa_r")
    line_r += t.count("\n", 0, gen_mod)
    rows, _ = parse_matz(rust_string_literals(body)[0])
    kats["short_basis_ring/working_sa_r"] = {"src": f"{rel}:{line_r}", **fixed,
                                             "expect_coefficient_embedding": rows}
    cs_mod = t.index("mod test_compute_s")
    for name, prm in sb.items():
        body, line = fn_body(t[cs_mod:], name)
        line += t.count("\n", 0, cs_mod)
        prm = dict(prm)
        if name == "base_2_power_two":
            prm["n"] = 8
        kats[f"short_basis_ring/compute_s/{name}"] = {"src": f"{rel}:{line}", **prm,
                                                      "expect": parse_matpoly(rust_string_literals(body)[-1])}

    # ---- rotation_matrix.rs
    rel = "src/utils/rotation_matrix.rs"
    t = load(rel)
    body, line = fn_body(t, "correct_rotation_matrix_vec")
    lits = rust_string_literals(body)
    kats["rot_minus/correct_rotation_matrix_vec"] = {
        "src": f"{rel}:{line}", "vec": [r[0] for r in parse_matz(lits[0])[0]], "expect": parse_matz(lits[2])[0]}
    body, line = fn_body(t, "correct_rotation_matrix_mat")
    lits = rust_string_literals(body)
    kats["rot_minus_matrix/correct_rotation_matrix_mat"] = {
        "src": f"{rel}:{line}", "mat": parse_matz(lits[0])[0], "expect": parse_matz(lits[1])[0],
        "note": "{} placeholders of the format! calls are u64::MAX"}

    # ---- parameter formulas (gadget_parameters.rs default_unchanged, gadget_default.rs correct_default_dimensions)
    rel = "src/sample/g_trapdoor/gadget_parameters.rs"
    _, line = fn_body(load(rel), "default_unchanged")
    cases = []
    for n in [5, 10, 50, 100]:
        for k in [5, 10, 25]:
            ln = (n - 1).bit_length()
            cases.append({"n": n, "q": 2**k, "base": 2, "k": k, "m_bar": n * k + ln * ln})
    kats["gadget_parameters/default_unchanged"] = {"src": f"{rel}:{line}", "cases": cases}
    rel = "src/sample/g_trapdoor/gadget_default.rs"
    _, line = fn_body(load(rel), "correct_default_dimensions")
    cases = []
    for n in [5, 10, 50]:
        for k in [5, 10]:
            ln = (n - 1).bit_length()
            mb = n * k + ln * ln
            cases.append({"n": n, "q": 2**k, "a_rows": n, "a_cols": mb + n * k, "r_rows": mb, "r_cols": n * k})
    kats["gadget_default/correct_default_dimensions"] = {"src": f"{rel}:{line}", "cases": cases}

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.json")
    with open(out, "w") as fh:
        json.dump(kats, fh, indent=1, sort_keys=True)
    print(f"wrote {out}: {len(kats) - 1} records")


if __name__ == "__main__":
    main()
