"""Build-container-only script: extracts NUMBERS the reference tabulates into JSON fixtures (data, not source text).

  reference_nodes_weights.json         Gauss-Legendre and Gauss-Lobatto abscissas / weights for n = 1..20, the values the reference's
                                       operator tables start from (src/dGMath/GL_and_GLL_nodes_and_weights.h:6-4080 gauss, :4082-4652
                                       lobatto; consumed by d4est_operators.c:727-741, :790-805).  Each `x[i] = <expr>;` is evaluated
                                       as double arithmetic, exactly what the C compiler does with it.
  cubed_sphere_7tree_connectivity.json tree_to_tree / tree_to_face of d4est_connectivity_new_sphere_7tree
                                       (src/Geometry/d4est_connectivity_cubed_sphere.c:41-58).

  p8est_tables.json                    the integer face / corner tables of p4est 2.8 (third_party/p4est-2.8.tar.gz,
                                       src/p8est_connectivity.c:29-63, :145-152: p8est_face_corners, _face_dual, _face_permutations,
                                       _face_permutation_sets, _face_permutation_refs, _corner_faces) and the reference's own copies
                                       (src/dGMath/d4est_reference.c:3-12: d4est_reference_p8est_FToF_code, _code_to_perm,
                                       _perm_to_order), read out of the initialisers as integers.

Run from the repo root:  python tests/golden/make_reference_tables.py   (needs /root/reference; the fixtures are committed).
"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def numeric(expr):
    """value of a C numeric initialiser made of float literals, + - * / and parentheses -- parsed, never evaluated as code (the text
    comes from the untrusted reference tree)"""
    import ast
    import operator
    expr = expr.strip()
    if not re.fullmatch(r"[0-9eE+\-*/(). ]+", expr):
        raise ValueError("not a numeric initialiser: %r" % expr)
    ops = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv}

    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return float(node.value)
        if isinstance(node, ast.BinOp) and type(node.op) in ops:
            return ops[type(node.op)](ev(node.left), ev(node.right))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            return -ev(node.operand) if isinstance(node.op, ast.USub) else ev(node.operand)
        raise ValueError("not a numeric initialiser: %r" % expr)
    return ev(ast.parse(expr, mode="eval"))


def parse_rule(text, xname, wname, nmax=20):
    out = {}
    blocks = re.split(r"if\s*\(\s*n\s*==\s*(\d+)\s*\)", text)
    for k in range(1, len(blocks) - 1, 2):
        n = int(blocks[k])
        if n > nmax or n in out:
            continue
        body = blocks[k + 1]
        x, w = {}, {}
        for name, idx, expr in re.findall(r"(\w+)\[(\d+)\]\s*=\s*([^;]+);", body):
            if name not in (xname, wname):
                continue
            val = numeric(expr)
            (x if name == xname else w)[int(idx)] = val
        if len(x) == n and len(w) == n:
            out[n] = {"x": [x[i] for i in range(n)], "w": [w[i] for i in range(n)]}
    return out


def main():
    src = open(os.path.join(REF, "dGMath", "GL_and_GLL_nodes_and_weights.h")).read()
    cut = src.index("d4est_operators_lobatto_nodes_and_weights")
    gauss = parse_rule(src[:cut], "x", "w")
    lobatto = parse_rule(src[cut:], "xtab", "weight")
    assert sorted(gauss) == list(range(1, 21)), sorted(gauss)
    assert sorted(lobatto) == list(range(1, 21)), sorted(lobatto)
    with open(os.path.join(HERE, "reference_nodes_weights.json"), "w") as fh:
        json.dump({"source": "src/dGMath/GL_and_GLL_nodes_and_weights.h (numeric values only)",
                   "gauss": {str(n): gauss[n] for n in sorted(gauss)},
                   "lobatto": {str(n): lobatto[n] for n in sorted(lobatto)}}, fh, indent=0)

    csrc = open(os.path.join(REF, "Geometry", "d4est_connectivity_cubed_sphere.c")).read()
    fn = csrc[csrc.index("d4est_connectivity_new_sphere_7tree"):csrc.index("d4est_connectivity_new_sphere_innerouter_shell")]

    def table(name):
        body = re.search(name + r"\[[^\]]*\]\s*=\s*\{([^}]*)\}", fn).group(1)
        body = re.sub(r"//[^\n]*", "", body)
        return [int(v) for v in re.findall(r"-?\d+", body)]

    ttt, ttf = table("tree_to_tree"), table("tree_to_face")
    assert len(ttt) == 42 and len(ttf) == 42
    with open(os.path.join(HERE, "cubed_sphere_7tree_connectivity.json"), "w") as fh:
        json.dump({"source": "src/Geometry/d4est_connectivity_cubed_sphere.c:41-58 (d4est_connectivity_new_sphere_7tree)",
                   "num_trees": 7, "tree_to_tree": ttt, "tree_to_face": ttf}, fh)
    # ---- integer topology tables: the initialisers' integers, in order
    import tarfile

    def int_table(text, name):
        m = re.search(r"\b" + re.escape(name) + r"\s*((?:\[\s*\d*\s*\])+)\s*=\s*\{(.*?)\}\s*;", text, flags=re.S)
        if not m:
            raise ValueError("table %s not found" % name)
        dims = [int(d) for d in re.findall(r"\[\s*(\d+)\s*\]", m.group(1))]
        body = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
        vals = [int(v) for v in re.findall(r"-?\d+", body)]
        n = 1
        for d in dims:
            n *= d
        assert len(vals) == n, (name, dims, len(vals))
        return {"shape": dims, "values": vals}

    with tarfile.open("/root/reference/third_party/p4est-2.8.tar.gz") as tf:
        member = [m for m in tf.getmembers() if m.name.endswith("/src/p8est_connectivity.c")][0]
        p8 = tf.extractfile(member).read().decode()
    dref = open(os.path.join(REF, "dGMath", "d4est_reference.c")).read()
    tables = {"source": "third_party/p4est-2.8.tar.gz: src/p8est_connectivity.c:29-63, :145-152; src/dGMath/d4est_reference.c:3-12 (integers only)"}
    for name in ("p8est_face_corners", "p8est_face_dual", "p8est_face_permutations", "p8est_face_permutation_sets",
                 "p8est_face_permutation_refs", "p8est_corner_faces"):
        tables[name] = int_table(p8, name)
    for name in ("d4est_reference_p8est_FToF_code", "d4est_reference_p8est_code_to_perm", "d4est_reference_p8est_perm_to_order"):
        tables[name] = int_table(dref, name)
    with open(os.path.join(HERE, "p8est_tables.json"), "w") as fh:
        json.dump(tables, fh)
    print("wrote p8est_tables.json")
    print("wrote reference_nodes_weights.json (n = 1..20) and cubed_sphere_7tree_connectivity.json")


if __name__ == "__main__":
    main()
