#!/usr/bin/env python3
"""Extract the numeric quadrature constants of the reference as DATA.

Reads the constant tables of math-bem/src/core/integration/gauss.rs:134-400
(Gauss-Legendre n in {1..8,10,12,16,20}; triangle rules 1/4/7/13 points) and
writes them, in this repo's own flat layout, to

  oracle/ma_oracle_tables.h           (used only by the CPU oracle)
  math_audio_amd/csrc/ma_tables.h     (used only by the HIP product code)

Only the numbers travel; this script is run once in the authoring container
(the reference is not present on the GPU box) and its outputs are committed.
"""
import re, sys, pathlib

REF = pathlib.Path("/root/reference/math-bem/src/core/integration/gauss.rs")
ROOT = pathlib.Path(__file__).resolve().parent.parent

src = REF.read_text()
num = r"-?\d+\.\d+(?:[eE]-?\d+)?"

def flat(name):
    m = re.search(r"static %s: \[[^=]*=\s*\[(.*?)\];" % name, src, re.S)
    assert m, name
    return re.findall(num, m.group(1))

GL_ORDERS = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20]
gl = {n: (flat("GL%d_X" % n), flat("GL%d_W" % n)) for n in GL_ORDERS}
for n, (x, w) in gl.items():
    assert len(x) == n and len(w) == n, n
tri = {n: flat("GAUCORWEI_TR%d" % n) for n in (1, 4, 7, 13)}
for n, v in tri.items():
    assert len(v) == 3 * n, n

def emit(path, prefix, qual, banner):
    out = []
    out.append("/* %s */" % banner)
    out.append("/* Numeric constants only; source of the values: reference")
    out.append("   math-bem/src/core/integration/gauss.rs:134-400 (extracted by")
    out.append("   tools/extract_quadrature_tables.py). Layout is this repo's own. */")
    out.append("#pragma once")
    # one packed abscissa/weight pool + offset table indexed by order
    pool_x, pool_w, offs = [], [], {}
    for n in GL_ORDERS:
        offs[n] = len(pool_x)
        pool_x += gl[n][0]
        pool_w += gl[n][1]
    out.append("#define %s_GL_POOL %d" % (prefix, len(pool_x)))
    out.append("%s double %s_gl_x[%d] = {" % (qual, prefix.lower(), len(pool_x)))
    out.append("  " + ",\n  ".join(pool_x))
    out.append("};")
    out.append("%s double %s_gl_w[%d] = {" % (qual, prefix.lower(), len(pool_w)))
    out.append("  " + ",\n  ".join(pool_w))
    out.append("};")
    # order -> (offset, count) with the reference's nearest-table fallback
    # (gauss.rs:41-58): n<=2->2, <=4->4, <=6->6, <=8->8, <=12->12, <=16->16, else 20
    def table_for(n):
        if n in gl: return n
        if n <= 2: return 2
        if n <= 4: return 4
        if n <= 6: return 6
        if n <= 8: return 8
        if n <= 12: return 12
        if n <= 16: return 16
        return 20
    out.append("/* index = requested order 0..20 (0 unused); value = {pool offset, point count} */")
    out.append("%s int %s_gl_index[21][2] = {" % (qual, prefix.lower()))
    rows = ["{0,0}"]
    for n in range(1, 21):
        t = table_for(n)
        rows.append("{%d,%d}" % (offs[t], t))
    out.append("  " + ", ".join(rows))
    out.append("};")
    for n in (1, 4, 7, 13):
        v = tri[n]
        out.append("/* triangle rule, %d points: xi, eta, raw weight (sum 1; callers scale by 0.5) */" % n)
        out.append("%s double %s_tri%d[%d][3] = {" % (qual, prefix.lower(), n, n))
        out.append(",\n".join("  {%s, %s, %s}" % (v[3*i], v[3*i+1], v[3*i+2]) for i in range(n)))
        out.append("};")
    path.write_text("\n".join(out) + "\n")
    print("wrote", path)

emit(ROOT / "oracle/ma_oracle_tables.h", "MAO", "static const",
     "CPU-oracle copy of the quadrature tables (test infrastructure)")
emit(ROOT / "math_audio_amd/csrc/ma_tables.h", "MAT", "static const",
     "Product copy of the quadrature tables (host side; device copies are uploaded to __constant__)")
