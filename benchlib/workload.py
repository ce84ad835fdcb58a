"""Workload set-up shared by the headline run, the other configs and the counter sub-run: mesh, GPU-built BVH, the two ray waves
resident in HBM; SURVEY 8(d)'s byte formulas."""
import os

import numpy as np

CONFIGS = {
    # name: mesh generator, precision, image, scaling when N > 1
    "C2": {"mesh": "sphere", "real": "f32", "w": 1920, "h": 1080, "scaling": "weak",
           "text": "C2 stand-in: closed lumpy sphere 264x132 = 69,168 triangles fp32 (Stanford bun_zipper.ply when --mesh is given)"},
    "C3": {"mesh": ("plane", 1000, 500), "real": "f32", "w": 1920, "h": 1080, "scaling": "weak",
           "text": "C3: Plane(1000,500) = 1,000,000 triangles fp32"},
    "C4": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "strong",
           "text": "C4: Plane(2500,2000) = 10,000,000 triangles fp32, fixed 4096x4096 frame"},
    "C4tile": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "weak", "tile_of": 8,
               "text": "C4 tile: Plane(2500,2000) = 10,000,000 triangles fp32, one GPU's 4096x512 share (rows y = 0 mod 8) of the 4096x4096 frame"},
    # (probe sizes between the tile and the frame: work-distribution sweeps, tools/tune_probe.py)
    "C3a": {"mesh": ("plane", 1000, 500), "real": "f32", "w": 1600, "h": 960, "scaling": "weak", "text": "C3 mesh, 1600x960 (1.5 M rays)"},
    "C3b": {"mesh": ("plane", 1000, 500), "real": "f32", "w": 2048, "h": 1184, "scaling": "weak", "text": "C3 mesh, 2048x1184 (2.4 M rays)"},
    "C3c": {"mesh": ("plane", 1000, 500), "real": "f32", "w": 2560, "h": 1440, "scaling": "weak", "text": "C3 mesh, 2560x1440 (3.7 M rays)"},
    "C2c": {"mesh": "sphere", "real": "f32", "w": 2560, "h": 1440, "scaling": "weak", "text": "C2 stand-in, 2560x1440 (3.7 M rays)"},
    "C4sixth": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "weak", "tile_of": 6,
                "text": "C4 sixth: Plane(2500,2000) = 10,000,000 triangles fp32, 4096x682 rows (y = 0 mod 6) of the 4096x4096 frame"},
    "C4quarter": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "weak", "tile_of": 4,
                  "text": "C4 quarter: Plane(2500,2000) = 10,000,000 triangles fp32, 4096x1024 rows (y = 0 mod 4) of the 4096x4096 frame"},
    "C4half": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "weak", "tile_of": 2,
               "text": "C4 half: Plane(2500,2000) = 10,000,000 triangles fp32, 4096x2048 rows (y = 0 mod 2) of the 4096x4096 frame"},
    "C5": {"mesh": ("plane", 1000, 500), "real": "f64", "w": 1920, "h": 1080, "scaling": "weak",
           "text": "C5: Plane(1000,500) = 1,000,000 triangles, fp64 build + traversal"},
}


def algorithmic_bytes(counters, real_bytes=4):
    """per ray sizeof(Ray)+sizeof(Hit) + 40 B per node visit + 52 B per triangle test (fp32)."""
    if real_bytes == 4:
        return 52 * counters["num_rays"] + 40 * counters["nodes_visited"] + 52 * counters["tris_tested"]
    return 104 * counters["num_rays"] + 64 * counters["nodes_visited"] + 88 * counters["tris_tested"]


def build_bytes(num_tris, num_nodes, real_bytes=4):
    """read the mesh once + write the tree once = N*(12 + 9*sizeof(T)) + nodes*sizeof(BVHNode) + 4N."""
    return num_tris * (12 + 9 * real_bytes) + num_nodes * (40 if real_bytes == 4 else 64) + 4 * num_tris


# ---------------------------------------------------------------------------
# workload set-up (shared by the headline run, the `configs` extras and the PMC child)
# ---------------------------------------------------------------------------
def make_mesh(cfg, mesh_path=None):
    from nanort_amd import scenes

    if cfg["mesh"] == "sphere":
        if mesh_path:
            from nanort_amd import meshio

            v, f = meshio.load_mesh(mesh_path)
            # the C2 camera looks at (0, 5, 0) from z = 20: bring a user mesh into that frame (uniform scale to a 15-unit box)
            lo, hi = v.min(axis=0), v.max(axis=0)
            s = np.float32(15.0 / float((hi - lo).max()))
            v = ((v - (lo + hi) * np.float32(0.5)) * s + np.array([0, 5, 0], np.float32)).astype(np.float32)
            return np.ascontiguousarray(v), np.ascontiguousarray(f), "user mesh %s (%d triangles)" % (os.path.basename(mesh_path), f.shape[0])
        v, f = scenes.sphere()
        return v, f, None
    _, nx, ny = cfg["mesh"]
    v, f = scenes.plane(nx, ny)
    return v, f, None


class Workload:
    """One config on one rank: mesh, GPU-built BVH, wave 1 and wave 2 resident in HBM."""

    def __init__(self, name, rank=0, world=1, device=0, builds=5, mesh_path=None):
        import torch

        from nanort_amd import BVHAccel, TriangleMesh, scenes
        from nanort_amd.wire import hit_dtype, ray_dtype, widen_rays

        cfg = CONFIGS[name]
        self.name, self.cfg, self.rank, self.world, self.torch = name, cfg, rank, world, torch
        self.real = np.float32 if cfg["real"] == "f32" else np.float64
        self.rb = 4 if cfg["real"] == "f32" else 8
        self.RAY, self.HIT = ray_dtype(self.real), hit_dtype(self.real)
        v32, self.faces, self.mesh_note = make_mesh(cfg, mesh_path)
        self.verts32 = v32
        self.verts = v32 if self.real == np.float32 else v32.astype(np.float64)
        mesh = TriangleMesh(self.verts, self.faces)
        self.accel = BVHAccel(self.real, device=device)
        self.build_ms = []
        for _ in range(max(1, builds)):
            assert self.accel.Build(mesh.num_faces, mesh)
            self.build_ms.append(self.accel.LastBuildMs())
        self.stats = self.accel.GetStatistics()
        self.num_nodes = int(self.stats["num_leaf_nodes"] + self.stats["num_branch_nodes"])
        # image rows of this rank: interleaved; weak scaling grows the image, strong scaling cuts a fixed one
        W, H = cfg["w"], cfg["h"]
        self.width = W
        if "tile_of" in cfg:  # one GPU's share of the C4 frame
            t = cfg["tile_of"]
            self.h_glob, y0, step, rows = H, rank, t * world, H // (t * world)
        elif cfg["scaling"] == "strong":
            if H % world:
                raise SystemExit("--config %s: %d rows do not split into %d equal tiles" % (name, H, world))
            self.h_glob, y0, step, rows = H, rank, world, H // world
        else:
            self.h_glob, y0, step, rows = H * world, rank, world, H
        self.rows = rows
        rays1_f32 = scenes.camera_rays_rows(W, self.h_glob, y0, step, rows)
        self.rays1 = rays1_f32 if self.real == np.float32 else widen_rays(rays1_f32)
        self.n1 = self.rays1.shape[0]
        cuda = torch.device("cuda", device)
        self.d_rays1 = torch.from_numpy(self.rays1.view(np.uint8)).to(cuda)
        self.d_hits1 = torch.empty(self.n1 * self.HIT.itemsize, dtype=torch.uint8, device=cuda)
        self.d_mask1 = torch.empty(self.n1, dtype=torch.uint8, device=cuda)
        self.accel.TraverseBatchDevice(self.d_rays1, self.d_hits1, self.d_mask1)
        torch.cuda.synchronize()
        self.hits1 = self.d_hits1.cpu().numpy().view(self.HIT)
        self.mask1 = self.d_mask1.cpu().numpy()
        # wave 2 is generated on the host from the wave-1 hits, in fp32 as SURVEY 8(d) defines it (widened for C5);
        # pixel index of ray i in the global image: row (y0 + step * (i // W)), column i % W
        from nanort_amd.wire import HIT_F32

        h32 = self.hits1
        if self.real != np.float32:
            h32 = np.zeros(self.n1, dtype=HIT_F32)
            for k in ("t", "u", "v"):
                h32[k] = self.hits1[k].astype(np.float32)
            h32["prim_id"] = self.hits1["prim_id"]
        self.kind2 = "bounce"
        rays2_f32 = scenes.secondary_rays("bounce", v32, self.faces, rays1_f32, h32, self.mask1, pixel_base=rank * self.n1)
        self.rays1_f32, self.hits1_f32 = rays1_f32, h32
        self.rays2 = rays2_f32 if self.real == np.float32 else widen_rays(rays2_f32)
        self.n2 = self.rays2.shape[0]
        self.d_rays2 = torch.from_numpy(self.rays2.view(np.uint8)).to(cuda)
        # wave-2 records are padded to n1 so that every rank's gather slice has the same size
        self.d_hits2 = torch.empty(max(1, self.n1) * self.HIT.itemsize, dtype=torch.uint8, device=cuda)
        self.d_mask2 = torch.empty(max(1, self.n1), dtype=torch.uint8, device=cuda)

    def counters(self):
        c1 = self.accel.TraverseCountDevice(self.d_rays1)
        c2 = self.accel.TraverseCountDevice(self.d_rays2) if self.n2 else {"num_rays": 0, "nodes_visited": 0, "tris_tested": 0}
        return c1, c2

    def results(self):
        self.torch.cuda.synchronize()
        return (self.d_hits1.cpu().numpy().view(self.HIT), self.d_mask1.cpu().numpy(),
                self.d_hits2.cpu().numpy().view(self.HIT)[:self.n2], self.d_mask2.cpu().numpy()[:self.n2])

    def describe(self):
        return "%s; %dx%d objrender-camera primaries + 1 cosine bounce per hit (%d + %d rays per GPU per step)" % (
            self.mesh_note or self.cfg["text"], self.width, self.rows, self.n1, self.n2)



def per_wave_counts(wl, c1, c2, ms1, ms2):
    """Work per ray of the two waves (the counting pass of the literal kernel: identical to the CPU oracle's counts)."""
    return {"primary": {"ms": round(ms1, 4), "rays": wl.n1, "nodes_per_ray": round(c1["nodes_visited"] / max(1, wl.n1), 2),
                        "tris_per_ray": round(c1["tris_tested"] / max(1, wl.n1), 2), "algorithmic_bytes": int(algorithmic_bytes(c1, wl.rb))},
            "bounce": {"ms": round(ms2, 4), "rays": wl.n2, "nodes_per_ray": round(c2["nodes_visited"] / max(1, wl.n2), 2),
                       "tris_per_ray": round(c2["tris_tested"] / max(1, wl.n2), 2), "algorithmic_bytes": int(algorithmic_bytes(c2, wl.rb))}}

