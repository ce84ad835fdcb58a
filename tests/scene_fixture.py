"""A small instanced scene shared by the two-level (nanosg) tests: a displaced plane + 4 transformed copies of
one lumpy sphere (rotation x non-uniform scale x translation), nanosg matrix convention (row 3 = translation)."""
import numpy as np

from nanort_amd import scenes


def xform(scale=(1, 1, 1), rot_z=0.0, rot_x=0.0, trans=(0, 0, 0)):
    S = np.diag([scale[0], scale[1], scale[2], 1.0])
    cz, sz = np.cos(rot_z), np.sin(rot_z)
    cx, sx = np.cos(rot_x), np.sin(rot_x)
    Rz = np.array([[cz, sz, 0, 0], [-sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    Rx = np.array([[1, 0, 0, 0], [0, cx, sx, 0], [0, -sx, cx, 0], [0, 0, 0, 1.0]])
    M = S @ Rz @ Rx
    M[3, :3] = trans
    return M.astype(np.float32)


def instances(sphere_res=(48, 24), plane_res=(60, 30)):
    sv, sf = scenes.sphere(*sphere_res)
    sv = sv - np.array([0, 5, 0], dtype=np.float32)
    pv, pf = scenes.plane(*plane_res)
    return [
        (pv, pf, xform()),
        (sv, sf, xform((0.3, 0.3, 0.3), 0.3, 0.2, (-4, 6, 4))),
        (sv, sf, xform((0.2, 0.4, 0.2), 1.0, -0.5, (3, 4, 5))),
        (sv, sf, xform((0.25, 0.25, 0.5), 2.0, 0.7, (0.5, 8, 6))),
        (sv, sf, xform((0.15, 0.15, 0.15), 0, 0, (0.5, 8, 6.2))),
    ]
