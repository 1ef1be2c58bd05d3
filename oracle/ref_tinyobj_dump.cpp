// ref_tinyobj_dump.cpp - builds the reference's own OBJ parser (the vendored
// tiny_obj_loader.h, included from where it lies under the reference checkout;
// nothing is copied) and dumps the per-corner vertex stream exactly as the
// reference's LoadObjFile (src/Renderers/ModelLoader.cpp:292-330) walks it.
// TEST INFRASTRUCTURE: used only to pin oracle/rt_oracle.c's OBJ reader.
//
// usage: ref_tinyobj_dump in.obj out.bin
// out.bin: int64 corner count, then per corner 6 floats (pos xyz, normal xyz;
// normal = 0 when the file has no normals).
#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.obj out.bin\n", argv[0]); return 2; }
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, argv[1])) {
        std::fprintf(stderr, "LoadObj failed: %s\n", err.c_str());
        return 1;
    }
    std::vector<float> out;
    for (const auto& shape : shapes) {
        for (const auto& index : shape.mesh.indices) {
            for (int k = 0; k < 3; ++k) out.push_back(attrib.vertices[3 * index.vertex_index + k]);
            for (int k = 0; k < 3; ++k)
                out.push_back(attrib.normals.size() != 0 && index.normal_index >= 0 ? attrib.normals[3 * index.normal_index + k] : 0.0f);
        }
    }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    int64_t n = (int64_t)(out.size() / 6);
    std::fwrite(&n, sizeof(n), 1, f);
    std::fwrite(out.data(), sizeof(float), out.size(), f);
    std::fclose(f);
    return 0;
}
