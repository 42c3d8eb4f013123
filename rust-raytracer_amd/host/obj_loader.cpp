#include "obj_loader.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace host {

namespace {

struct Key {
    uint32_t d[8];
    bool operator==(const Key &o) const { return memcmp(d, o.d, sizeof d) == 0; }
};
struct KeyHash {
    size_t operator()(const Key &k) const
    {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t v : k.d) {
            h ^= v;
            h *= 1099511628211ull;
        }
        return (size_t)h;
    }
};

uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

void mul3(const float m[9], const float v[3], float out[3])
{
    for (int r = 0; r < 3; r++) out[r] = (m[3 * r] * v[0] + m[3 * r + 1] * v[1]) + m[3 * r + 2] * v[2];
}

// one "v/vt/vn" reference; 0 = absent, negative = relative to the end
bool parse_ref(const char *&p, long counts[3], long out[3])
{
    for (int k = 0; k < 3; k++) out[k] = 0;
    for (int k = 0; k < 3; k++) {
        if (*p != '/' || k == 0) {
            char *end;
            long v = strtol(p, &end, 10);
            if (end != p) {
                out[k] = v < 0 ? counts[k] + v + 1 : v;
                p = end;
            } else if (k == 0) {
                return false;
            }
        }
        if (*p == '/') {
            p++;
            if (k == 2) return false;
        } else {
            break;
        }
    }
    return true;
}

} // namespace

bool load_obj(const std::string &path, const float transform[9], Mesh &out, std::string &error)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        error = "cannot open " + path;
        return false;
    }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> text((size_t)size + 1);
    if (size > 0 && fread(text.data(), 1, (size_t)size, f) != (size_t)size) {
        fclose(f);
        error = "short read on " + path;
        return false;
    }
    fclose(f);
    text[(size_t)size] = 0;

    std::vector<float> vp, vt, vn;
    std::unordered_map<Key, uint32_t, KeyHash> index_of;
    std::vector<long> poly; // 3 longs per corner
    out = Mesh();

    auto vertex = [&](const float p[3], const float t[2], const float n[3]) -> uint32_t {
        Key k{{bits(p[0]), bits(p[1]), bits(p[2]), bits(t[0]), bits(t[1]), bits(n[0]), bits(n[1]), bits(n[2])}};
        auto it = index_of.find(k);
        if (it != index_of.end()) return it->second;
        uint32_t idx = out.num_vertices();
        out.positions.insert(out.positions.end(), p, p + 3);
        out.texcoords.insert(out.texcoords.end(), t, t + 2);
        out.normals.insert(out.normals.end(), n, n + 3);
        index_of.emplace(k, idx);
        return idx;
    };

    char *line = text.data();
    long line_no = 0;
    while (*line) {
        char *eol = strchr(line, '\n');
        if (eol) *eol = 0;
        line_no++;
        const char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *e;
            p += 2;
            for (int k = 0; k < 3; k++) {
                vp.push_back(strtof(p, &e));
                p = e;
            }
        } else if (p[0] == 'v' && p[1] == 't') {
            char *e;
            p += 2;
            for (int k = 0; k < 2; k++) {
                vt.push_back(strtof(p, &e));
                p = e;
            }
        } else if (p[0] == 'v' && p[1] == 'n') {
            char *e;
            p += 2;
            for (int k = 0; k < 3; k++) {
                vn.push_back(strtof(p, &e));
                p = e;
            }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            poly.clear();
            long counts[3] = {(long)(vp.size() / 3), (long)(vt.size() / 2), (long)(vn.size() / 3)};
            while (true) {
                while (*p == ' ' || *p == '\t' || *p == '\r') p++;
                if (!*p) break;
                long r[3];
                if (!parse_ref(p, counts, r) || r[0] < 1 || r[0] > counts[0] || r[1] < 0 || r[1] > counts[1] ||
                    r[2] < 0 || r[2] > counts[2]) {
                    error = path + ": bad face reference on line " + std::to_string(line_no);
                    return false;
                }
                poly.insert(poly.end(), r, r + 3);
            }
            const size_t corners = poly.size() / 3;
            if (corners < 3) {
                error = path + ": face with fewer than 3 vertices on line " + std::to_string(line_no);
                return false;
            }
            for (size_t i = 0; i + 2 < corners; i++) { // fan, src/scene.rs:43-44
                const size_t c[3] = {0, i + 1, i + 2};
                float pos[3][3], tex[3][2], nor[3][3];
                bool has_normal = true;
                for (int k = 0; k < 3; k++) {
                    const long *r = &poly[3 * c[k]];
                    mul3(transform, &vp[3 * (size_t)(r[0] - 1)], pos[k]);
                    tex[k][0] = r[1] ? vt[2 * (size_t)(r[1] - 1)] : 0.f;
                    tex[k][1] = r[1] ? vt[2 * (size_t)(r[1] - 1) + 1] : 0.f;
                    if (r[2]) {
                        mul3(transform, &vn[3 * (size_t)(r[2] - 1)], nor[k]);
                    } else {
                        nor[k][0] = nor[k][1] = nor[k][2] = 0.f;
                        has_normal = false;
                    }
                }
                float e1[3], e2[3];
                for (int k = 0; k < 3; k++) {
                    e1[k] = pos[1][k] - pos[0][k];
                    e2[k] = pos[2][k] - pos[0][k];
                }
                const float cx = e1[1] * e2[2] - e1[2] * e2[1];
                const float cy = e1[2] * e2[0] - e1[0] * e2[2];
                const float cz = e1[0] * e2[1] - e1[1] * e2[0];
                const float magsq = (cx * cx + cy * cy) + cz * cz;
                if (magsq == 0.f || !std::isfinite(magsq)) { // src/scene.rs:53-59
                    out.skipped_zero_area++;
                    continue;
                }
                for (int k = 0; k < 3; k++) out.indices.push_back(vertex(pos[k], tex[k], nor[k]));
                const float len = sqrtf(magsq);
                float ng[3] = {cx / len, cy / len, cz / len};
                if (has_normal && (ng[0] * nor[0][0] + ng[1] * nor[0][1]) + ng[2] * nor[0][2] < 0.f) // src/mesh.rs:183-186
                    for (int k = 0; k < 3; k++) ng[k] = -ng[k];
                out.geom_normals.insert(out.geom_normals.end(), ng, ng + 3);
            }
        }
        if (!eol) break;
        line = eol + 1;
    }
    return true;
}

} // namespace host
