"""Triangle-mesh ingestion for the bench / test harness: Stanford PLY (ascii and binary_little_endian, as
`bun_zipper.ply` ships) and Wavefront OBJ (`v` / `f` lines, polygons fanned).  SURVEY.md §8(c)/(d): config C2 is
the Stanford Bunny when the user supplies its path (`bench.py --config C2 --mesh bun_zipper.ply`) and the
procedural lumpy sphere otherwise — the reference tree ships no bunny and there is no network.

Returns (vertices float32 (n, 3), faces uint32 (m, 3)), the layout nanort::TriangleMesh takes
(reference nanort.h:922-930).  Harness code: the product (C ABI) only ever sees the flat arrays.
"""
import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def _fan(polys):
    tris = []
    for p in polys:
        for k in range(1, len(p) - 1):
            tris.append((p[0], p[k], p[k + 1]))
    return np.asarray(tris, dtype=np.uint32).reshape(-1, 3)


def load_obj(path):
    verts, polys = [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "f":
                idx = [int(w.split("/")[0]) for w in t[1:]]
                polys.append([i - 1 if i > 0 else len(verts) + i for i in idx])
    return np.asarray(verts, dtype=np.float32).reshape(-1, 3), _fan(polys)


def load_ply(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:  # EOF before end_header: a truncated file
                raise ValueError("%s: PLY header has no end_header" % path)
            t = line.decode("ascii", "replace").split()
            if not t or t[0] == "comment" or t[0] == "obj_info":
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                elements.append((t[1], int(t[2]), []))
            elif t[0] == "property":
                elements[-1][2].append(t[1:])
            elif t[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError("%s: PLY format %r not supported" % (path, fmt))
        verts, polys = None, []
        for name, count, props in elements:
            is_list = [p[0] == "list" for p in props]
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                if name == "vertex":
                    cols = [p[-1] for p in props]
                    ix = [cols.index(c) for c in ("x", "y", "z")]
                    verts = np.asarray([[float(r[i]) for i in ix] for r in rows], dtype=np.float32)
                elif name == "face":
                    polys = [[int(w) for w in r[1:1 + int(r[0])]] for r in rows]
            elif not any(is_list):
                rec = np.dtype([(p[-1], "<" + _PLY_TYPES[p[0]]) for p in props])
                data = np.frombuffer(f.read(count * rec.itemsize), dtype=rec, count=count)
                if name == "vertex":
                    verts = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
            else:  # records with list properties (faces): walk them
                for _ in range(count):
                    for p in props:
                        if p[0] == "list":
                            n = int(np.frombuffer(f.read(np.dtype(_PLY_TYPES[p[1]]).itemsize), dtype="<" + _PLY_TYPES[p[1]])[0])
                            it = np.dtype("<" + _PLY_TYPES[p[2]])
                            vals = np.frombuffer(f.read(n * it.itemsize), dtype=it)
                            if name == "face" and p[-1] in ("vertex_indices", "vertex_index"):
                                polys.append([int(v) for v in vals])
                        else:
                            f.read(np.dtype(_PLY_TYPES[p[0]]).itemsize)
    if verts is None:
        raise ValueError("%s: no vertex element" % path)
    return np.ascontiguousarray(verts), _fan(polys)


def load_mesh(path):
    """Dispatch on the extension (.ply / .obj)."""
    low = path.lower()
    if low.endswith(".ply"):
        return load_ply(path)
    if low.endswith(".obj"):
        return load_obj(path)
    raise ValueError("%s: expected a .ply or .obj file" % path)
