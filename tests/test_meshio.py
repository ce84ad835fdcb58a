"""The harness's PLY / OBJ ingestion (bench.py --config C2 --mesh ...): SURVEY.md §8(c)/(d) asks C2 to accept a
user-supplied Stanford bunny path."""
import struct

import numpy as np

from nanort_amd import meshio

V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0.5]], dtype=np.float32)
F = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)


def test_obj_with_polygons_and_negative_indices(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0.5\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -2\n")
    v, f = meshio.load_mesh(str(p))
    assert np.array_equal(v, V)
    assert f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]


def test_ply_ascii_as_bun_zipper_is_laid_out(tmp_path):
    p = tmp_path / "m.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment zipper output\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                 "property float confidence\nproperty float intensity\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n"
                 + "".join("%g %g %g 0.5 0.5\n" % tuple(r) for r in V) + "3 0 1 2\n3 0 2 3\n")
    v, f = meshio.load_mesh(str(p))
    assert np.array_equal(v, V) and np.array_equal(f, F)


def test_ply_binary_little_endian(tmp_path):
    p = tmp_path / "m.ply"
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
            "element face 2\nproperty list uchar uint vertex_indices\nend_header\n").encode()
    body = V.astype("<f4").tobytes() + b"".join(struct.pack("<B3I", 3, *row) for row in F.tolist())
    p.write_bytes(head + body)
    v, f = meshio.load_mesh(str(p))
    assert np.array_equal(v, V) and np.array_equal(f, F)
