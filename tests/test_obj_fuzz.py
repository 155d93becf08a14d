"""Randomised OBJ files through three loaders: the reference's own loadOBJ (objloader.cpp compiled where it lies into
oracle/_ref/ref_driver -- only where /root/reference exists), the oracle's restatement and the product's crt_load_obj.
loadOBJ is a token loop over fscanf (objloader.cpp:41-109), so what it accepts is what fscanf accepts: any number format
of %f, any white space, CRLF, unknown tokens that eat their line, lines longer than its 1000-byte buffer, a missing final
newline.  Files whose face indices are out of range are left out (the reference indexes past its arrays there)."""
import os
import subprocess

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.skipif(not orc.ref_driver_available(), reason="oracle/_ref not built (no /root/reference here)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def number(rng):
    x = float(rng.normal() * 10 ** rng.uniform(-4, 4))
    k = int(rng.integers(0, 9))
    if k == 0: return f"{x:.6e}"
    if k == 1: return f"{x:+.3f}"
    if k == 2: return str(int(x))
    if k == 3: return f"{x:.9g}"
    if k == 4: return f"{abs(x) % 1:.4f}"[1:]           # ".1234"
    if k == 5: return f"{int(x)}."
    if k == 6: return f"{x:.2E}"
    if k == 7: return "0x1.8p3" if rng.integers(0, 2) else "-0"
    return f"{x:.17g}"


def random_obj(rng):
    nv = int(rng.integers(3, 40))
    sep = lambda: rng.choice([" ", "  ", "\t", " \t "])
    eol = rng.choice(["\n", "\r\n", "\n\n", " \n"])
    lines = []
    kinds = ["v"] * nv + ["vn"] * nv
    extra = int(rng.integers(0, 25))
    for _ in range(extra):
        kinds.append(rng.choice(["#", "vt", "g", "s", "usemtl", "o", "long", "blank", "mtllib"]))
    order = rng.permutation(len(kinds))
    for i in order:
        k = kinds[i]
        if k in ("v", "vn"):
            lines.append(k + sep() + sep().join(number(rng) for _ in range(3)))
        elif k == "vt":
            lines.append("vt" + sep() + number(rng) + sep() + number(rng))
        elif k == "#":
            lines.append("# " + " ".join(rng.choice(["v", "f", "1", "2", "x", "vn"]) for _ in range(int(rng.integers(0, 8)))))
        elif k == "long":
            lines.append("# " + "x" * int(rng.integers(990, 1100)))           # straddles objloader.cpp's 1000-byte fgets buffer
        elif k == "blank":
            lines.append("")
        else:
            lines.append(k + " " + rng.choice(["a", "teapot", "1", "off", "mat 2"]))
    nf = int(rng.integers(0, 30))
    for _ in range(nf):
        a, b, c = (int(rng.integers(1, nv + 1)) for _ in range(3))
        tail = rng.choice(["", "", "", " ", f" {int(rng.integers(1, nv + 1))}"])      # a quad: the fourth index becomes a token of its own
        lines.insert(int(rng.integers(0, len(lines) + 1)), f"f{sep()}{a}{sep()}{b}{sep()}{c}{tail}")
    # faces may come before the vertices they name: the loader resolves indices after the whole file is read
    text = eol.join(lines)
    if rng.integers(0, 2):
        text += eol
    return text


def run_reference(path, tmp):
    out = os.path.join(tmp, "ref.bin")
    p = subprocess.run([orc.REF_DRIVER, "obj", path, out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
    if p.returncode != 0:
        return None
    raw = np.fromfile(out, np.uint8)
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    f = np.frombuffer(raw[4:], np.float32)
    return f[:3 * n].reshape(n, 3).copy(), f[3 * n:6 * n].reshape(n, 3).copy()


def test_random_obj_files_load_as_the_reference_loads_them(tmp_path):
    import cudaraytracer_amd as crt
    rng = np.random.default_rng(77 + int(os.environ.get("CRT_FUZZ_SEED_BASE", "0")))      # a soak: any other base draws other files
    loaded = failed = 0
    for k in range(300):
        text = random_obj(rng)
        if k % 25 == 24:
            text += "\nf 1/1/1 2/2/2 3/3/3\n"          # objloader.cpp:78-82: not "f a b c" -> loadOBJ fails
        p = tmp_path / f"m{k}.obj"
        p.write_bytes(text.encode())
        ref = run_reference(str(p), str(tmp_path))
        if ref is None:
            failed += 1
            with pytest.raises(crt.CrtError):
                crt.load_obj(str(p))
            continue
        loaded += 1
        rv, rn = ref
        ov, on = orc.load_obj(str(p))
        pv, pn = crt.load_obj(str(p))
        assert ov.shape == rv.shape == pv.shape, f"file {k}: corners {ov.shape} (oracle) {rv.shape} (reference) {pv.shape} (product)\n{text[:400]}"
        assert np.array_equal(bits(ov), bits(rv)) and np.array_equal(bits(on), bits(rn)), f"file {k}: oracle differs from the reference"
        assert np.array_equal(bits(pv), bits(rv)) and np.array_equal(bits(pn), bits(rn)), f"file {k}: product differs from the reference"
    assert loaded >= 250 and failed >= 10, (loaded, failed)
