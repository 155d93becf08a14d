"""Host-side layer of the product (no GPU): OBJ loader, scene builders, animation step and the two BVH
builders, checked against the oracle and the reference-derived golden vectors."""
import os

import numpy as np
import pytest

import orc
from sphere_pad import sphere_box_half

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def crt():
    import cudaraytracer_amd as m
    return m


def test_load_obj_equals_reference_golden(crt):
    z = np.load(os.path.join(G, "teapot_obj.npz"))
    v, n = crt.load_obj(crt.TEAPOT_OBJ)
    assert v.tobytes() == z["verts"].tobytes() and n.tobytes() == z["normals"].tobytes()


def test_load_obj_errors(crt, tmp_path):
    with pytest.raises(crt.CrtError):
        crt.load_obj(str(tmp_path / "missing.obj"))
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nvn 0 0 1\nf 1/1/1 2/2/2 3/3/3\n")      # objloader.cpp:78-82: not "f a b c" -> failure
    with pytest.raises(crt.CrtError):
        crt.load_obj(str(p))
    p2 = tmp_path / "oob.obj"
    p2.write_text("v 0 0 0\nvn 0 0 1\nf 1 2 3\n")                  # index past the vertex list: error, not a crash
    with pytest.raises(crt.CrtError):
        crt.load_obj(str(p2))


def test_teapot_scene_literals(crt):
    prims = crt.scene_teapot(with_sphere=True)
    assert len(prims) == 1025 and int((prims["type"] == 1).sum()) == 1024
    t = prims[5]
    assert t["objNum"] == 15 and t["shininess"] == 64                     # main.cu:240 objNum = corner index
    assert np.allclose(t["specular"], [0.9, 0.4, 0]) and np.allclose(t["diffuse"], [0.8, 0.3, 0.1])
    assert np.allclose(np.linalg.norm(prims["v1Norm"][:1024], axis=1), 1, atol=1e-6)   # main.cu:249 normalises
    s = prims[-1]
    assert s["type"] == 0 and s["radius"] == 20 and list(s["position"]) == [-40, 40, 0] and s["objNum"] == 1
    # unseeded glibc rand(): main.cu:306-310
    assert np.allclose(s["direction"], [0.6803754, -0.2112341, 0.5661984], atol=1e-6)
    assert abs(float(s["speed"]) - (0.01 + 0.7984400 * 0.3)) < 1e-6
    assert len(crt.scene_teapot(with_sphere=False)) == 1024
    lights = crt.default_lights()
    assert list(lights[0]["direction"]) == [2, -1, -1] and list(lights[1]["direction"]) == [-2, -1, 1]
    assert not lights["point"].any()


def test_synthetic_scenes_are_deterministic_and_in_range(crt):
    a, b = crt.scene_soup(5000, seed=1), crt.scene_soup(5000, seed=1)
    assert a.tobytes() == b.tobytes() and a.tobytes() != crt.scene_soup(5000, seed=2).tobytes()
    c = (a["v1"] + a["v2"] + a["v3"]) / 3
    assert c[:, 0].min() > -182 and c[:, 0].max() < 182 and c[:, 2].min() > -282 and c[:, 2].max() < 202
    assert np.abs(a["v1"] - a["v2"]).max() <= 3.0
    assert np.allclose(np.linalg.norm(a["v1Norm"], axis=1), 1, atol=1e-5)
    s = crt.scene_spheres(4096, seed=1)
    assert s["radius"].min() >= 2 and s["radius"].max() <= 8
    assert np.all(np.abs(s["position"][:, 0]) + s["radius"] <= 200.001) and np.all(np.abs(s["position"][:, 2]) + s["radius"] <= 300.001)
    assert s["speed"].min() >= 0.01 and s["speed"].max() <= 0.31


def test_animation_step_equals_oracle_bitwise(crt):
    """main.cu:545-573: rotation of triangles (positions and normals), sphere bounce and advance."""
    prims = np.concatenate([crt.scene_teapot(), crt.scene_spheres(300, seed=4)])
    prims["speed"][-300:] *= 40          # make spheres reach the walls within the test
    sc = orc.Scene(prims)
    mine = prims.copy()
    for _ in range(40):
        crt.animate_host(mine)
        sc.animate()
    assert mine.tobytes() == sc.prims().tobytes()
    flipped = np.sign(mine["direction"][-300:]) != np.sign(prims["direction"][-300:])
    assert flipped.any(), "no sphere bounced off a wall: the reflect branches were not exercised"


def walk_oracle(d):
    """(boxes, leaf membership) of the oracle's pointer tree in DFS order"""
    leaves = [d["leaf_prims"][f:f + c] for f, c, l in zip(d["first"], d["count"], d["isleaf"]) if l]
    return leaves


@pytest.mark.parametrize("scene", ["teapot", "spheres", "soup", "small", "soup_threaded"])
def test_reference_topology_equals_oracle_tree(crt, scene):
    """A12: the product's index-based restatement of newBuild/constructTree gives the oracle's tree:
    same leaf membership in DFS order (=> same tie-break ranks) and bit-identical boxes."""
    prims = {"teapot": lambda: crt.scene_teapot(), "spheres": lambda: crt.scene_spheres(3000, seed=9),
             "soup": lambda: crt.scene_soup(20000, seed=5), "small": lambda: crt.scene_spheres(40, seed=1),
             # large enough for the builders to hand subtrees to other threads (ranges above max(n/64, 8192) primitives)
             "soup_threaded": lambda: crt.scene_soup(120000, seed=6)}[scene]()
    compare_with_oracle_tree(crt, prims)


def compare_with_oracle_tree(crt, prims):
    sc = orc.Scene(prims)
    sc.build()
    d = sc.dump()
    sc.close()
    h = crt.host_bvh(prims, 1)
    order = np.concatenate([l for l in walk_oracle(d)]) if len(prims) else np.zeros(0, np.int32)
    assert np.array_equal(h["slot_to_scene"], order.astype(np.uint32))
    rank = np.empty(len(prims), np.uint32)
    rank[order] = np.arange(len(prims), dtype=np.uint32)
    assert np.array_equal(h["rank_of_scene"], rank)
    assert np.array_equal(crt.host_bvh(prims, 0)["rank_of_scene"], rank)      # the wide tree carries the same ranks
    # boxes: walk both trees together
    boxes, links = h["boxes"], h["links"]

    def rec(oi, box, link):
        assert np.array_equal(d["boxes"][oi], box), f"box of oracle node {oi}"
        if d["isleaf"][oi]:
            if d["count"][oi] > 0:
                assert link & 0x80000000 and (link & 0x1FFFFFF) == d["first"][oi] and ((link >> 25) & 63) + 1 == d["count"][oi]
            else:
                assert link == 0xFFFFFFFF
            return
        assert not (link & 0x80000000)
        rec(d["left"][oi], boxes[link, 0], int(links[link, 0]))
        rec(d["right"][oi], boxes[link, 1], int(links[link, 1]))
    if d["isleaf"][0]:
        rec(0, boxes[0, 0], int(links[0, 0]))
    else:
        rec(0, h["root"], 0)


@pytest.mark.parametrize("seed", [int(os.environ.get("CRT_FUZZ_SEED_BASE", "0")) + k for k in range(24)])
def test_random_scenes_build_the_oracle_s_tree(crt, seed):
    """the generators of tests/test_gpu_fuzz.py (triangles and spheres interleaved, stretched, on a lattice with many equal
    centroid coordinates and boxes of zero thickness): newBuild/constructTree restated in the product gives the oracle's tree,
    leaf for leaf and box for box; where the reference recurses forever both sides say so; the wide tree is a valid hierarchy
    with exact leaf boxes either way"""
    from test_gpu_fuzz import lattice_scene, random_scene
    rng = np.random.default_rng(31000 + seed)
    prims = lattice_scene(crt, rng) if seed % 3 == 2 else random_scene(crt, rng)
    if seed % 4 == 1:          # everything below the origin: the reference's maxima start at 0 (main.cu:35, tests/test_gpu_robustness.py)
        for f in ("position", "v1", "v2", "v3"):
            prims[f] = prims[f] - np.float32(400)
    check_wide_hierarchy(crt, prims)
    sc = orc.Scene(prims)
    try:
        try:
            sc.build()
            builds = True
        except RuntimeError:
            builds = False
    finally:
        sc.close()
    if builds:
        compare_with_oracle_tree(crt, prims)
    else:
        with pytest.raises(crt.CrtError):
            crt.host_bvh(prims, 1)


def check_wide_hierarchy(crt, prims):
    h = crt.host_bvh(prims, 0)
    boxes, links, s2s = h["boxes"], h["links"], h["slot_to_scene"]
    assert sorted(s2s.tolist()) == list(range(len(prims)))                 # every primitive in exactly one leaf
    half = sphere_box_half(prims)
    lo = np.where(prims["type"][:, None] == 0, prims["position"] - half[:, None],
                  np.minimum(np.minimum(prims["v1"], prims["v2"]), prims["v3"]))
    hi = np.where(prims["type"][:, None] == 0, prims["position"] + half[:, None],
                  np.maximum(np.maximum(prims["v1"], prims["v2"]), prims["v3"]))
    seen = np.zeros(len(prims), bool)
    depth_seen = 0
    stack = [(0, h["root"], 1)]
    while stack:
        node, pbox, depth = stack.pop()
        depth_seen = max(depth_seen, depth)
        for k in range(links.shape[1]):
            link = int(links[node, k])
            if link == 0xFFFFFFFF:
                continue
            b = boxes[node, k]
            assert np.all(b[0::2] >= pbox[0::2]) and np.all(b[1::2] <= pbox[1::2])     # child inside parent
            if link & 0x80000000:
                start, cnt = link & 0x1FFFFFF, ((link >> 25) & 63) + 1
                assert cnt <= 4
                idx = s2s[start:start + cnt]
                assert not seen[idx].any()
                seen[idx] = True
                assert np.all(lo[idx] >= b[0::2]) and np.all(hi[idx] <= b[1::2])        # exact bounds: triangles as they are, spheres with sphere_box_half
                assert np.array_equal(b[0::2], lo[idx].min(axis=0)) and np.array_equal(b[1::2], hi[idx].max(axis=0))
            else:
                assert tuple(h["parents"][link]) == (node, k)
                stack.append((link, b, depth + 1))
    assert seen.all() and depth_seen == h["depth"]
    return h


@pytest.mark.parametrize("mode", [0, 1])
def test_threaded_host_build_equals_the_serial_one(crt, mode, monkeypatch):
    """the host builders hand subtrees over disjoint ranges to other threads; node numbering, slots, ranks and boxes
    are those of the build on one thread (CRT_BUILD_THREADS=1), byte for byte"""
    prims = np.concatenate([crt.scene_soup(90000, seed=3), crt.scene_spheres(20000, seed=4)])
    threaded = crt.host_bvh(prims, mode)
    monkeypatch.setenv("CRT_BUILD_THREADS", "1")
    serial = crt.host_bvh(prims, mode)
    assert threaded["depth"] == serial["depth"]
    for k in ("root", "boxes", "links", "parents", "slot_to_scene", "rank_of_scene"):
        assert np.array_equal(threaded[k], serial[k]), k


def test_wide_bvh_is_a_valid_hierarchy(crt):
    check_wide_hierarchy(crt, np.concatenate([crt.scene_soup(6000, seed=3), crt.scene_spheres(500, seed=3)]))


def test_coincident_centroids_do_not_hang(crt):
    """bvh.cu:14-24,58-68: more than 64 primitives with one centroid recurse forever in the reference;
    here the build reports CRT_ERR_BUILD for the reference topology.  The wide builder splits by count."""
    prims = crt.scene_spheres(100, seed=1)
    prims["position"] = (1, 2, 3)
    with pytest.raises(crt.CrtError) as e:
        crt.host_bvh(prims, 1)
    assert e.value.code == -6
    h = check_wide_hierarchy(crt, prims)
    assert np.array_equal(h["rank_of_scene"], np.arange(100))       # no reference order: ties resolve by scene index


def teapot_grid(crt):
    """64 small teapots below y = 0: the reference's box maxima never fall below 0 (main.cu:35), so for a group of
    neighbouring teapots y looks like the longest axis, and the flat rings of the mesh (hundreds of triangles with
    one centroid y) can never be split: constructTree recurses forever on this perfectly ordinary scene."""
    tp = crt.scene_teapot(with_sphere=False)
    grid = []
    for i in range(8):
        for j in range(8):
            q = tp.copy()
            for f in ("v1", "v2", "v3"):
                q[f] = (q[f] * np.float32(0.4) + np.array([(i - 3.5) * 45, -100 + (j % 3) * 60, (j - 3.5) * 60], np.float32)).astype(np.float32)
            grid.append(q)
    return np.concatenate(grid)


def test_scene_the_reference_cannot_build(crt):
    prims = teapot_grid(crt)
    sc = orc.Scene(prims)
    with pytest.raises(RuntimeError):
        sc.build()                                   # the oracle restates the runaway recursion (bounded)
    sc.close()
    with pytest.raises(crt.CrtError):
        crt.host_bvh(prims, 1)
    check_wide_hierarchy(crt, prims)


def test_area_light_sample_table_equals_oracle(crt):
    """the fixed draws of the area lights (include/crt.h: crt_area_table) are the oracle's, bit for bit, and lie
    on the unit disk"""
    a, b = crt.area_table(), orc.area_table()
    assert a.tobytes() == b.tobytes()
    assert (a[:, 0] >= 0).all() and (a[:, 0] < 1).all()
    assert np.allclose(a[:, 1] ** 2 + a[:, 2] ** 2, 1, atol=1e-6)
    assert len(np.unique(a[:, 0])) > 1000


def test_tiles_partition_the_frame(crt):
    for (w, h, r) in ((720, 360, 1), (1920, 1080, 8), (333, 205, 3), (3840, 2160, 8)):
        total = 0
        seen = set()
        for rank in range(r):
            p = crt.default_params(width=w, height=h, rank=rank, nranks=r)
            ids, npx = crt.tiles_owned(p)
            assert not (seen & set(ids.tolist()))
            seen |= set(ids.tolist())
            total += npx
            assert all(crt.tile_owner(int(t), r) == rank for t in ids)
        assert total == w * h and len(seen) == ((w + 31) // 32) * ((h + 31) // 32)
    # static balance at 4K over 8 ranks: within 2 % of the mean
    counts = [crt.tiles_owned(crt.default_params(width=3840, height=2160, rank=k, nranks=8))[1] for k in range(8)]
    assert max(counts) / (sum(counts) / 8) < 1.02


def test_write_bmp_roundtrip(crt, tmp_path):
    from PIL import Image
    img = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)          # BGR, row 0 = bottom
    path = str(tmp_path / "t.bmp")
    crt.write_bmp(path, img, 7, 5)
    rgb = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(rgb[::-1, :, ::-1], img)
