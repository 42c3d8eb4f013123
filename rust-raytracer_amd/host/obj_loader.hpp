// OBJ -> indexed triangle mesh, the way the reference's Scene::load does it
// (src/scene.rs:29-91) with its TriangleMeshBuilder (src/mesh.rs:136-195):
// polygons are fanned into triangles (0, i+1, i+2), positions and normals are
// multiplied by the scene's 3x3 transform, triangles whose cross product has
// zero squared length are skipped, vertices are shared when position, texture
// coordinate and normal are bit-identical, and the geometric normal is the
// normalized cross product flipped to the side of the first vertex normal.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace host {

struct Mesh {
    std::vector<float> positions;    // 3 per vertex
    std::vector<float> texcoords;    // 2 per vertex
    std::vector<float> normals;      // 3 per vertex
    std::vector<uint32_t> indices;   // 3 per triangle
    std::vector<float> geom_normals; // 3 per triangle
    uint32_t skipped_zero_area = 0;
    uint32_t num_vertices() const { return (uint32_t)(positions.size() / 3); }
    uint32_t num_triangles() const { return (uint32_t)(indices.size() / 3); }
};

// transform: row-major 3x3.  Faces may omit vt and vn (the reference requires
// both, src/scene.rs:46-51; a face without vn keeps the unflipped geometric
// normal).  Returns false and fills `error` when the file cannot be read or a
// face refers to a missing element.
bool load_obj(const std::string &path, const float transform[9], Mesh &out, std::string &error);

} // namespace host
