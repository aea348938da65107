"""Scene ingestion (kl_load_mesh): OBJ and PLY files into kl_tri arrays. Host code only: runs without a GPU."""
import struct

import numpy as np
import pytest

import klrender as kl

OBJ = """# a unit square (quad, fan-triangulated), a triangle with relative indices, two materials
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
usemtl floor
f 1/1/1 2/2/1 3/3/1 4/4/1
v 0 0 1
v 1 0 1
v 0 1 1
usemtl glass
f -3 -2 -1
usemtl floor
f 5//1 6//1 7//1
"""


def test_obj_fans_uvs_relative_indices_and_materials(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text(OBJ)
    t = kl.load_mesh(str(p), material=2, n_materials=8)
    assert len(t) == 4
    assert np.array_equal(t["v0"][0], [0, 0, 0]) and np.array_equal(t["v1"][0], [1, 0, 0]) and np.array_equal(t["v2"][0], [1, 1, 0])
    assert np.array_equal(t["v0"][1], [0, 0, 0]) and np.array_equal(t["v1"][1], [1, 1, 0]) and np.array_equal(t["v2"][1], [0, 1, 0])
    assert np.array_equal(t["uv1"][0], [1, 0]) and np.array_equal(t["uv2"][1], [0, 1])
    assert np.array_equal(t["v0"][2], [0, 0, 1]) and np.array_equal(t["v2"][2], [0, 1, 1]) and np.array_equal(t["uv0"][2], [0, 0])
    assert list(t["material"]) == [2, 2, 3, 2]          # floor -> material, glass -> material + 1
    assert list(kl.load_mesh(str(p), material=7, n_materials=8)["material"]) == [7, 7, 7, 7]   # clamped to the last one


def test_obj_errors(tmp_path):
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(kl.KlError, match="bad face corner"):
        kl.load_mesh(str(p))
    with pytest.raises(kl.KlError, match="cannot read"):
        kl.load_mesh(str(tmp_path / "missing.obj"))
    with pytest.raises(kl.KlError, match="extension"):
        kl.load_mesh(str(tmp_path / "mesh.stl"))
    e = tmp_path / "empty.obj"
    e.write_text("# nothing\n")
    assert len(kl.load_mesh(str(e))) == 0


def _ply_header(fmt, nv, nf, uv=True):
    h = "ply\nformat %s 1.0\ncomment made by the test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % (fmt, nv)
    if uv:
        h += "property float s\nproperty float t\n"
    h += "property uchar red\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % nf
    return h


def test_ply_ascii_and_binary_agree(tmp_path):
    verts = [(0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (1, 1, 0, 1, 1), (0, 1, 0, 0, 1), (0.5, 0.5, 1, 0.5, 0.5)]
    faces = [(0, 1, 2, 3), (0, 1, 4), (1, 2, 4)]
    a = tmp_path / "a.ply"
    a.write_text(_ply_header("ascii", 5, 3) + "".join("%g %g %g %g %g 255\n" % v for v in verts)
                 + "".join("%d %s\n" % (len(f), " ".join(map(str, f))) for f in faces))
    b = tmp_path / "b.ply"
    blob = _ply_header("binary_little_endian", 5, 3).encode()
    for v in verts:
        blob += struct.pack("<5fB", *v, 255)
    for f in faces:
        blob += struct.pack("<B%di" % len(f), len(f), *f)
    b.write_bytes(blob)
    ta, tb = kl.load_mesh(str(a), material=1), kl.load_mesh(str(b), material=1)
    assert len(ta) == 4 and ta.tobytes() == tb.tobytes()
    assert np.array_equal(ta["v2"][3], [0.5, 0.5, 1]) and np.array_equal(ta["uv2"][3], [0.5, 0.5]) and set(ta["material"]) == {1}
    c = tmp_path / "c.ply"
    c.write_text("ply\nformat binary_big_endian 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(kl.KlError, match="not supported"):
        kl.load_mesh(str(c))
    d = tmp_path / "d.ply"
    d.write_bytes(blob[:-6])
    with pytest.raises(kl.KlError, match="truncated"):
        kl.load_mesh(str(d))


def _expect_refused(path, needle=None):
    with pytest.raises(kl.KlError) as ei:
        kl.load_mesh(str(path))
    if needle:
        assert needle in str(ei.value)


def test_ply_header_counts_are_checked_against_the_file(tmp_path):
    """a header may claim any count: the loader must answer with a status, not loop or exhaust memory"""
    p = tmp_path / "huge.ply"
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 1000000000000000000\nend_header\n")
    _expect_refused(p, "no properties")
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 1000000000000000000\nproperty float x\nend_header\n\0\0\0\0")
    _expect_refused(p, "exceeds the size")
    p.write_bytes(b"ply\nformat ascii 1.0\nelement vertex -5\nproperty float x\nend_header\n")
    _expect_refused(p, "bad PLY element")
    # face index that is NaN / far out of range
    p.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                  b"element face 1\nproperty list uchar float vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 nan 2\n")
    _expect_refused(p)            # the stream refuses "nan" as a number: reported as truncated data
    p.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                  b"element face 1\nproperty list uchar double vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1e300 2\n")
    _expect_refused(p, "out of range")


def test_garbage_and_truncated_files_are_refused(tmp_path):
    rng = np.random.default_rng(11)
    for k in range(20):
        p = tmp_path / ("g%d.ply" % k)
        p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 40\nproperty float x\nproperty float y\nproperty float z\n"
                      b"element face 30\nproperty list uchar int vertex_indices\nend_header\n" + rng.bytes(int(rng.integers(0, 900))))
        try:
            kl.load_mesh(str(p))
        except kl.KlError:
            pass
    for k in range(20):
        p = tmp_path / ("g%d.obj" % k)
        p.write_bytes(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n" + rng.bytes(int(rng.integers(0, 400))))
        try:
            kl.load_mesh(str(p))
        except kl.KlError:
            pass
