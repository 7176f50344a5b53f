#!/usr/bin/env python3
"""Generate tests/golden/reproj_golden.json — known answers for ReprojectionFactor::Evaluate.

Runs ONLY in the build container (needs /root/reference as text).  It reads the reference's
src/reprojection_factor.cpp, extracts the 18 scalar Jacobian assignment expressions
(`jacobians[a][b] = ...;`, lines 63-83) and evaluates them with Python floats (IEEE-754 double, the
same arithmetic the C++ performs).  The residual is stored twice:
  r       the Eigen expression of lines 24-38, restated here operation by operation (it is executable code in the
          reference, but Eigen is not available to run it);
  r_text  the reference's own scalar residual expressions — the commented MATLAB output at lines 53-54 — evaluated as
          text exactly like the Jacobians.  Those expressions omit the 1/|q|^2 factor of line 33 (they were generated
          for a unit quaternion), so r_text is stored only for the unit-quaternion cases (|q| = 1 to rounding), where
          the two forms agree to ~1e-13 relative; null elsewhere.
The output file holds inputs and expected outputs only (no reference text).
"""
import json
import math
import os
import random
import re
import sys

REF = "/root/reference/src/reprojection_factor.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reproj_golden.json")


def load_exprs():
    src = open(REF).read()
    pat = re.compile(r"jacobians\[(\d)\]\[(\d+)\]\s*=\s*(.*?);", re.S)
    exprs = {}
    for m in pat.finditer(src):
        a, b, e = int(m.group(1)), int(m.group(2)), m.group(3)
        e = e.replace("camera_info.focal", "focal")
        exprs[(a, b)] = compile(e, f"<J{a}_{b}>", "eval")
    assert len(exprs) == 18, len(exprs)
    return exprs


def load_residual_exprs():
    """residuals[0|1] = ...; inside the comment block at src/reprojection_factor.cpp:52-55, as text."""
    src = open(REF).read()
    out = {}
    for m in re.finditer(r"residuals\[(\d)\]\s*=\s*(.*?);", src, re.S):
        e = m.group(2)
        e = e.replace("camera_info.focal", "focal").replace("camera_info.cx", "cx").replace("camera_info.cy", "cy")
        e = e.replace("obs(0)", "obs0").replace("obs(1)", "obs1")
        out[int(m.group(1))] = compile(e, f"<r{m.group(1)}>", "eval")
    assert sorted(out) == [0, 1], sorted(out)
    return out


def residual(q, t, p, obs, focal, cx, cy):
    # src/reprojection_factor.cpp:24-38 (Eigen expression order)
    w, x, y, z = q
    ss = [[0, -z, y], [z, 0, -x], [-y, x, 0]]
    ll = [[ss[i][j] + (w if i == j else 0.0) for j in range(3)] for i in range(3)]
    v = [x, y, z]
    LR = [[v[i] * v[j] + sum(ll[i][k] * ll[k][j] for k in range(3)) for j in range(3)] for i in range(3)]
    nn = w * w + x * x + y * y + z * z
    g = [sum(LR[i][k] * p[k] for k in range(3)) / nn + t[i] for i in range(3)]
    psi = 1.0 / g[2]
    K = [[focal, 0.0, cx], [0.0, focal, cy]]
    r = [sum(K[i][k] * psi * g[k] for k in range(3)) - obs[i] for i in range(2)]
    return r


def main():
    exprs = load_exprs()
    rexprs = load_residual_exprs()
    rng = random.Random(0x5EED0011)
    cams = [(718.856, 607.1928, 185.2157), (385.7544860839844, 323.1204833984375, 236.7432098388672)]
    cases = []
    for i in range(256):
        focal, cx, cy = cams[i % 2]
        # quaternion: near identity, random, and deliberately non-unit
        kind = i % 4
        if kind == 0:
            q = [1.0, rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05)]
        else:
            q = [rng.gauss(0, 1) for _ in range(4)]
        nrm = math.sqrt(sum(c * c for c in q))
        scale = 1.0 if kind in (0, 1) else rng.uniform(0.3, 3.0)
        q = [c / nrm * scale for c in q]
        t = [rng.uniform(-5, 5), rng.uniform(-2, 2), rng.uniform(-5, 5)]
        # choose p so that gamma_z is positive: sample gamma, back-rotate approximately via R^T
        depth = [0.05, 1.0, 8.0, 60.0][(i // 4) % 4] * rng.uniform(0.8, 1.25)
        g = [rng.uniform(-0.8, 0.8) * depth, rng.uniform(-0.3, 0.3) * depth, depth]
        w, x, y, z = [c / scale for c in q]
        R = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
             [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
             [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
        d = [g[k] - t[k] for k in range(3)]
        p = [sum(R[k][j] * d[k] for k in range(3)) for j in range(3)]
        obs = [rng.uniform(0, 1241), rng.uniform(0, 376)]
        env = dict(pow=pow, focal=focal,
                   q1=q[0], q2=q[1], q3=q[2], q4=q[3], t1=t[0], t2=t[1], t3=t[2],
                   p1=p[0], p2=p[1], p3=p[2])
        jp = [0.0] * 14
        jx = [0.0] * 6
        for (a, b), code in exprs.items():
            val = eval(code, {"__builtins__": {}}, env)
            if a == 0:
                jp[b] = val
            else:
                jx[b] = val
        r = residual(q, t, p, obs, focal, cx, cy)
        r_text = None
        if scale == 1.0:  # unit quaternion: the commented scalar residuals (no 1/|q|^2) apply
            renv = dict(env, cx=cx, cy=cy, obs0=obs[0], obs1=obs[1])
            r_text = [eval(rexprs[0], {"__builtins__": {}}, renv), eval(rexprs[1], {"__builtins__": {}}, renv)]
        cases.append(dict(focal=focal, cx=cx, cy=cy, pose=q + t, point=p, obs=obs,
                          r=r, r_text=r_text, jpose=jp, jpoint=jx))
    with open(OUT, "w") as f:
        json.dump(dict(source="src/reprojection_factor.cpp:24-38 (r, restated), :53-54 (r_text, evaluated as text), :63-83 (Jacobians, evaluated as text)",
                       cases=cases), f)
    print("wrote", OUT, len(cases))


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not mounted; fixture is generated in the build container only")
    main()
