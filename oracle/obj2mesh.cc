// oracle/obj2mesh.cc — TEST INFRASTRUCTURE (golden-fixture generation only).
//
// Flattens an OBJ into vertices[]/faces[] the way the reference's
// examples/objrender does (LoadObj, examples/objrender/main.cc:380-460): all
// shapes concatenated, per-shape vertex offsets added to the face indices.
// The OBJ parser/triangulator itself is the reference's own
// examples/objrender/tiny_obj_loader.{h,cc}, compiled from the reference tree
// by oracle/Makefile (never copied here).
//
// usage: obj2mesh in.obj out.bin     (run from the OBJ's directory so the
//                                      .mtl is found, as the example does)
// out.bin: u32 num_vertices, u32 num_faces, f32 xyz * nv, u32 ijk * nf
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "tiny_obj_loader.h"

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s in.obj out.bin\n", argv[0]);
    return 2;
  }
  std::vector<tinyobj::shape_t> shapes;
  std::vector<tinyobj::material_t> materials;
  std::string err = tinyobj::LoadObj(shapes, materials, argv[1]);
  if (!err.empty()) {
    fprintf(stderr, "%s\n", err.c_str());
    return 1;
  }
  std::vector<float> verts;
  std::vector<uint32_t> faces;
  for (size_t s = 0; s < shapes.size(); s++) {
    uint32_t base = (uint32_t)(verts.size() / 3);
    const tinyobj::mesh_t &m = shapes[s].mesh;
    for (size_t i = 0; i < m.indices.size(); i++) faces.push_back(base + m.indices[i]);
    for (size_t i = 0; i < m.positions.size(); i++) verts.push_back(m.positions[i]);
  }
  uint32_t nv = (uint32_t)(verts.size() / 3), nf = (uint32_t)(faces.size() / 3);
  FILE *fp = fopen(argv[2], "wb");
  if (!fp) return 1;
  fwrite(&nv, 4, 1, fp);
  fwrite(&nf, 4, 1, fp);
  fwrite(verts.data(), 4, verts.size(), fp);
  fwrite(faces.data(), 4, faces.size(), fp);
  fclose(fp);
  printf("%u vertices, %u faces\n", nv, nf);
  return 0;
}
