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

// The same mesh -- every array equal element for element, the same error for the same file -- built
// by `threads` threads (0 = up to 16): chunked parsing, then vertex sharing split by key hash with
// vertices numbered in order of first appearance (obj_loader_mt.cpp).  For the reference's
// gigabyte-sized scene files.  A thread gets at least min_chunk_bytes of text.
bool load_obj_parallel(const std::string &path, const float transform[9], unsigned threads, Mesh &out,
                       std::string &error, size_t min_chunk_bytes = 65536);

// FNV-1a over the five arrays of a mesh: equal meshes, equal fingerprints
uint64_t mesh_fingerprint(const Mesh &m);

} // namespace host
