"""Names for nearest-hit mismatches, in the vocabulary of SURVEY.md 9.1 (test infrastructure).

  G2  both sides hit at the bit-identical t: an equal-t tie resolved differently (Ray.cu:199,216)
  G3  the reference's OWN tree disagrees with its own brute-force loop over all primitives on this ray: its slab test
      (Ray.cu:54-104) rejected a box whose primitive is hit -- a box false-negative, decided by re-running the box-free
      search (orc.Scene.trace_brute) on the ray
  G1  everything else: a predicate within rounding of its threshold; the margins of the disputed primitive's predicates
      (Ray.cu:114,122,126,148,151,158), recomputed in double, say which one

None is tolerated by the tests; this only words the failure message."""
import numpy as np


def _tri_margins(p, o, d):
    v1, v2, v3 = (np.asarray(p[k], np.float64) for k in ("v1", "v2", "v3"))
    ng = np.cross(v2 - v1, v3 - v1)
    ln = np.linalg.norm(ng)
    if not ln > 0:
        return {"degenerate": 0.0}
    ng /= ln
    den = float(ng @ d)
    out = {"|Ng.d| - 1e-4": abs(den) - 1e-4}
    if den != 0:
        t = float((v1 - o) @ ng) / den
        x = o + t * d
        out["t + 1e-4"] = t + 1e-4
        out["edge1"] = float(np.cross(v2 - v1, x - v1) @ ng)
        out["edge2"] = float(np.cross(v3 - v2, x - v2) @ ng)
        out["edge3"] = float(np.cross(v1 - v3, x - v3) @ ng)
    return out


def _sph_margins(p, o, d):
    c = np.asarray(p["position"], np.float64); r = float(p["radius"])
    a = float(d @ d); b = 2 * float(d @ (o - c)); cc = float(c @ c + o @ o - 2 * (c @ o) - r * r)
    disc = b * b - 4 * a * cc
    out = {"disc - 1e-9": disc - 1e-9}
    if disc > 0:
        out["t2 - 1e-9"] = (-b + disc ** 0.5) / (2 * a) - 1e-9
        out["t - 1e-9"] = (-b - disc ** 0.5) / (2 * a) - 1e-9
    return out


def margins(prim, ray6):
    """the predicates of the hit test of `prim` for the ray, as (value - threshold) in double: the smallest |margin| is the grazing one"""
    o, d = np.asarray(ray6[:3], np.float64), np.asarray(ray6[3:], np.float64)
    return _sph_margins(prim, o, d) if int(prim["type"]) == 0 else _tri_margins(prim, o, d)


def classify(ids_g, ts_g, ids_o, ts_o, bad, sc=None, rays=None, prims=None, limit=8):
    """one line per mismatch (the first `limit`): which class, and the evidence.  sc: the orc.Scene that produced ids_o (for the
    G3 decision); rays: the (n, 6) rays; prims: the scene (for the G1 margins)"""
    out = []
    for i in np.flatnonzero(bad)[:limit]:
        head = f"ray {i}: gpu ({ids_g[i]}, {ts_g[i]!r}) oracle ({ids_o[i]}, {ts_o[i]!r})"
        if ids_g[i] >= 0 and ids_o[i] >= 0 and np.float32(ts_g[i]).view(np.uint32) == np.float32(ts_o[i]).view(np.uint32):
            out.append(head + " -> G2 equal-t tie")
            continue
        if sc is not None and rays is not None:
            ib, tb = sc.trace_brute(np.asarray(rays[i], np.float32).reshape(1, 6), nthreads=1)
            same = ib[0] == ids_o[i] and (ib[0] < 0 or np.float32(tb[0]).view(np.uint32) == np.float32(ts_o[i]).view(np.uint32))
            if not same:
                agrees = "the gpu agrees with the brute-force result" if ib[0] == ids_g[i] else "the gpu agrees with neither"
                out.append(head + f" -> G3 box false-negative in the reference's slab test: its brute-force loop gives ({ib[0]}, {tb[0]!r}); {agrees}")
                continue
        line = head + " -> G1 threshold grazing"
        if prims is not None and rays is not None:
            for who, pid in (("gpu's", ids_g[i]), ("oracle's", ids_o[i])):
                if pid >= 0:
                    m = margins(prims[pid], rays[i])
                    k = min(m, key=lambda q: abs(m[q]))
                    line += f"; {who} primitive {pid}: nearest predicate {k} = {m[k]:.3e}"
        out.append(line)
    return "; ".join(out)
