// Multi-threaded OBJ loading: the same mesh as load_obj (the reference's
// Scene::load, src/scene.rs:29-91, with TriangleMeshBuilder's vertex sharing,
// src/mesh.rs:136-195), bit for bit and index for index, for files of a
// gigabyte and more (the reference's san-miguel.obj is 1.14 GB).
//
//   A  the text is cut into one chunk per thread at line ends; every thread
//      parses its lines into local v / vt / vn arrays and a list of faces
//      (raw references, plus how many v / vt / vn lines of the chunk precede
//      the face, which is what relative references and range checks need)
//   B  prefix sums over the chunks give global element numbers; the element
//      arrays are concatenated
//   C  every thread resolves, checks, fans out and transforms its faces and
//      emits, per kept triangle, three vertex keys and the geometric normal
//   D  vertex sharing: keys are split by hash over the threads; each thread
//      finds, for its keys, the first corner (in file order) that holds the
//      same key; a prefix sum over the "is a first corner" flags numbers the
//      vertices in order of first appearance, which is the order the serial
//      builder numbers them in.
#include "obj_loader.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace host {

namespace {

struct Key {
    uint32_t d[8]; // position, texture coordinate, normal as bit patterns
};

inline uint64_t key_hash(const Key &k)
{
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (uint32_t v : k.d) {
        h ^= v;
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 29;
    }
    return h;
}

inline uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float unbits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline void mul3(const float m[9], const float v[3], float out[3])
{
    for (int r = 0; r < 3; r++) out[r] = (m[3 * r] * v[0] + m[3 * r + 1] * v[1]) + m[3 * r + 2] * v[2];
}

// white space inside a line (what strtof / strtol skip, minus the line end)
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// strtof for the decimal forms OBJ files hold, with the same result bit for bit.  A mantissa below 2^53
// and a power of ten up to 10^22 are exact doubles, so their product or quotient is the correctly
// rounded double of the decimal; rounding that to float equals rounding the decimal itself except
// when the double sits exactly on a float rounding boundary -- that case, subnormal results and every
// unusual spelling go to strtof.  Does not cross a line end; returns 0 without moving p when no
// number starts at p (as strtof does).
float parse_f32(const char *&p)
{
    const char *s = p;
    while (is_blank(*s)) s++;
    const char *start = s;
    bool neg = false;
    if (*s == '+' || *s == '-') {
        neg = (*s == '-');
        s++;
    }
    uint64_t m = 0;
    int exp10 = 0;
    bool any = false, dropped = false;
    if (s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) { // hexadecimal float
        char *e;
        const float f = strtof(start, &e);
        p = e;
        return f;
    }
    while (*s >= '0' && *s <= '9') {
        if (m < 100000000000000000ull)
            m = m * 10 + (uint64_t)(*s - '0');
        else {
            exp10++;
            dropped = true;
        }
        s++;
        any = true;
    }
    if (*s == '.') {
        s++;
        while (*s >= '0' && *s <= '9') {
            if (m < 100000000000000000ull) {
                m = m * 10 + (uint64_t)(*s - '0');
                exp10--;
            } else {
                dropped = true;
            }
            s++;
            any = true;
        }
    }
    if (!any) {
        if (*start == '\n' || *start == '\0') return 0.f; // nothing on this line: strtof would not convert either
        char *e;
        const float f = strtof(start, &e); // "inf", "nan", hexadecimal, or no number at all
        if (e != start) p = e;
        return f;
    }
    if (*s == 'e' || *s == 'E') {
        const char *t = s + 1;
        bool eneg = false;
        if (*t == '+' || *t == '-') {
            eneg = (*t == '-');
            t++;
        }
        if (*t >= '0' && *t <= '9') {
            int ev = 0;
            while (*t >= '0' && *t <= '9') {
                if (ev < 10000) ev = ev * 10 + (*t - '0');
                t++;
            }
            exp10 += eneg ? -ev : ev;
            s = t;
        }
    }
    if (!dropped && m < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
        double d = (double)m;
        d = exp10 < 0 ? d / kPow10[-exp10] : d * kPow10[exp10];
        uint64_t db;
        memcpy(&db, &d, 8);
        const bool on_boundary = (db & 0x1FFFFFFFull) == 0x10000000ull;
        if (!on_boundary && (d == 0.0 || (d > 1e-37 && d < 3e38))) {
            p = s;
            const float f = (float)d;
            return neg ? -f : f;
        }
    }
    char *e;
    const float f = strtof(start, &e);
    p = e;
    return f;
}

// strtol in base 10 that does not cross a line end; ok = false when no digits follow
long parse_long(const char *&p, bool &ok)
{
    const char *s = p;
    while (is_blank(*s)) s++;
    bool neg = false;
    if (*s == '+' || *s == '-') {
        neg = (*s == '-');
        s++;
    }
    if (!(*s >= '0' && *s <= '9')) {
        ok = false;
        return 0;
    }
    long v = 0;
    while (*s >= '0' && *s <= '9') {
        if (v < (1l << 52)) v = v * 10 + (*s - '0');
        s++;
    }
    p = s;
    ok = true;
    return neg ? -v : v;
}

// one "v/vt/vn" reference as written (0 = absent); the same grammar as the serial loader's
bool parse_ref_raw(const char *&p, long out[3])
{
    for (int k = 0; k < 3; k++) out[k] = 0;
    for (int k = 0; k < 3; k++) {
        if (*p != '/' || k == 0) {
            bool ok;
            const long v = parse_long(p, ok);
            if (ok)
                out[k] = v;
            else if (k == 0)
                return false;
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

struct Chunk {
    const char *begin = nullptr, *end = nullptr;
    std::vector<float> vp, vt, vn;
    std::vector<long> refs;           // 3 per corner, as written
    std::vector<uint32_t> face_corners; // corners per face
    std::vector<uint32_t> face_counts;  // 3 per face: v / vt / vn lines of this chunk before the face
    std::vector<uint32_t> face_line;    // line of the face inside the chunk (0-based)
    std::vector<uint8_t> face_bad;      // a reference of the face did not parse
    uint32_t lines = 0;
    // phase C
    std::vector<Key> keys;    // 3 per kept triangle
    std::vector<float> ng;    // 3 per kept triangle
    uint32_t skipped = 0;
    long err_line = -1;       // first failing line of the chunk (0-based inside the chunk)
    bool err_few = false;     // ... and whether it is "fewer than 3 vertices"
    size_t first_line = 0, v0 = 0, t0 = 0, n0 = 0, corner0 = 0;
};

void parse_chunk(Chunk &c)
{
    const char *line = c.begin;
    while (line < c.end) {
        const char *eol = (const char *)memchr(line, '\n', (size_t)(c.end - line));
        if (!eol) eol = c.end;
        const char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            for (int k = 0; k < 3; k++) c.vp.push_back(parse_f32(p));
        } else if (p[0] == 'v' && p[1] == 't') {
            p += 2;
            for (int k = 0; k < 2; k++) c.vt.push_back(parse_f32(p));
        } else if (p[0] == 'v' && p[1] == 'n') {
            p += 2;
            for (int k = 0; k < 3; k++) c.vn.push_back(parse_f32(p));
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            uint32_t corners = 0;
            bool bad = false;
            while (true) {
                while (*p == ' ' || *p == '\t' || *p == '\r') p++;
                if (*p == '\n' || *p == '\0' || p >= eol) break;
                long r[3];
                if (!parse_ref_raw(p, r)) {
                    bad = true;
                    break;
                }
                c.refs.insert(c.refs.end(), r, r + 3);
                corners++;
            }
            c.face_corners.push_back(corners);
            c.face_counts.push_back((uint32_t)(c.vp.size() / 3));
            c.face_counts.push_back((uint32_t)(c.vt.size() / 2));
            c.face_counts.push_back((uint32_t)(c.vn.size() / 3));
            c.face_line.push_back(c.lines);
            c.face_bad.push_back(bad ? 1 : 0);
        }
        c.lines++;
        line = eol + 1;
    }
}

void build_triangles(Chunk &c, const float *VP, const float *VT, const float *VN, const float transform[9])
{
    size_t ref_at = 0;
    for (size_t f = 0; f < c.face_corners.size(); f++) {
        const uint32_t corners = c.face_corners[f];
        const long *refs = c.refs.data() + ref_at;
        ref_at += (size_t)corners * 3;
        if (c.err_line >= 0) continue; // the file is rejected; nothing after the first error matters
        const long counts[3] = {(long)(c.v0 + c.face_counts[3 * f]), (long)(c.t0 + c.face_counts[3 * f + 1]),
                                (long)(c.n0 + c.face_counts[3 * f + 2])};
        bool bad = false;
        long res[3 * 64];
        std::vector<long> big;
        long *r = res;
        if (corners > 64) {
            big.resize((size_t)corners * 3);
            r = big.data();
        }
        for (uint32_t k = 0; k < corners && !bad; k++)
            for (int e = 0; e < 3; e++) {
                const long raw = refs[3 * k + e];
                const long v = raw < 0 ? counts[e] + raw + 1 : raw;
                r[3 * k + e] = v;
                if (v < (e == 0 ? 1 : 0) || v > counts[e]) bad = true;
            }
        if (bad || c.face_bad[f]) {
            c.err_line = (long)c.face_line[f];
            c.err_few = false;
            continue;
        }
        if (corners < 3) {
            c.err_line = (long)c.face_line[f];
            c.err_few = true;
            continue;
        }
        for (uint32_t i = 0; i + 2 < corners; i++) { // fan, src/scene.rs:43-44
            const uint32_t cs[3] = {0, i + 1, i + 2};
            float pos[3][3], tex[3][2], nor[3][3];
            bool has_normal = true;
            for (int k = 0; k < 3; k++) {
                const long *q = r + 3 * cs[k];
                mul3(transform, VP + 3 * (size_t)(q[0] - 1), pos[k]);
                tex[k][0] = q[1] ? VT[2 * (size_t)(q[1] - 1)] : 0.f;
                tex[k][1] = q[1] ? VT[2 * (size_t)(q[1] - 1) + 1] : 0.f;
                if (q[2]) {
                    mul3(transform, VN + 3 * (size_t)(q[2] - 1), nor[k]);
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
                c.skipped++;
                continue;
            }
            for (int k = 0; k < 3; k++)
                c.keys.push_back(Key{{bits(pos[k][0]), bits(pos[k][1]), bits(pos[k][2]), bits(tex[k][0]), bits(tex[k][1]),
                                      bits(nor[k][0]), bits(nor[k][1]), bits(nor[k][2])}});
            const float len = sqrtf(magsq);
            float ng[3] = {cx / len, cy / len, cz / len};
            if (has_normal && (ng[0] * nor[0][0] + ng[1] * nor[0][1]) + ng[2] * nor[0][2] < 0.f) // src/mesh.rs:183-186
                for (int k = 0; k < 3; k++) ng[k] = -ng[k];
            c.ng.insert(c.ng.end(), ng, ng + 3);
        }
    }
}

template <class F>
void run_threads(unsigned n, F fn)
{
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n; t++) pool.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto &th : pool) th.join();
}

} // namespace

bool load_obj_parallel(const std::string &path, const float transform[9], unsigned threads, Mesh &out,
                       std::string &error, size_t min_chunk_bytes)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        error = "cannot open " + path;
        return false;
    }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> text((size_t)size + 1);
    if (size > 0 && fread(text.data(), 1, (size_t)size, f) != (size_t)size) {
        fclose(f);
        error = "short read on " + path;
        return false;
    }
    fclose(f);
    text[(size_t)size] = 0;
    // a NUL inside the file ends the text there, as it does for the serial loader's C-string walk
    const size_t len = strlen(text.data());

    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    threads = std::min(threads, 64u);
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, len / std::max<size_t>(1, min_chunk_bytes) + 1));

    // ---- A: chunks at line ends
    std::vector<Chunk> chunks(T);
    {
        const char *base = text.data();
        size_t at = 0;
        for (unsigned t = 0; t < T; t++) {
            size_t stop = (t + 1 == T) ? len : std::max(at, len * (t + 1) / T);
            if (stop < len) {
                const char *nl = (const char *)memchr(base + stop, '\n', len - stop);
                stop = nl ? (size_t)(nl - base) + 1 : len;
            }
            chunks[t].begin = base + at;
            chunks[t].end = base + stop;
            at = stop;
        }
    }
    run_threads(T, [&](unsigned t) { parse_chunk(chunks[t]); });

    // ---- B: global numbering of the elements
    size_t nv = 0, nt = 0, nn = 0, nl = 0;
    for (Chunk &c : chunks) {
        c.v0 = nv;
        c.t0 = nt;
        c.n0 = nn;
        c.first_line = nl;
        nv += c.vp.size() / 3;
        nt += c.vt.size() / 2;
        nn += c.vn.size() / 3;
        nl += c.lines;
    }
    std::vector<float> VP(nv * 3 + 3), VT(nt * 2 + 2), VN(nn * 3 + 3);
    run_threads(T, [&](unsigned t) {
        Chunk &c = chunks[t];
        if (!c.vp.empty()) memcpy(VP.data() + 3 * c.v0, c.vp.data(), c.vp.size() * sizeof(float));
        if (!c.vt.empty()) memcpy(VT.data() + 2 * c.t0, c.vt.data(), c.vt.size() * sizeof(float));
        if (!c.vn.empty()) memcpy(VN.data() + 3 * c.n0, c.vn.data(), c.vn.size() * sizeof(float));
        std::vector<float>().swap(c.vp);
        std::vector<float>().swap(c.vt);
        std::vector<float>().swap(c.vn);
    });

    // ---- C: faces -> triangles
    run_threads(T, [&](unsigned t) { build_triangles(chunks[t], VP.data(), VT.data(), VN.data(), transform); });
    for (const Chunk &c : chunks)
        if (c.err_line >= 0) { // the first failing line of the file
            const std::string line = std::to_string(c.first_line + (size_t)c.err_line + 1);
            error = path + (c.err_few ? ": face with fewer than 3 vertices on line " : ": bad face reference on line ") + line;
            return false;
        }
    std::vector<float>().swap(VP);
    std::vector<float>().swap(VT);
    std::vector<float>().swap(VN);

    out = Mesh();
    size_t corners = 0;
    for (Chunk &c : chunks) {
        c.corner0 = corners;
        corners += c.keys.size();
        out.skipped_zero_area += c.skipped;
        std::vector<long>().swap(c.refs);
    }
    if (corners >= 0xFFFFFFFFull) {
        error = path + ": more than 2^32 triangle corners";
        return false;
    }
    std::vector<Key> keys(corners);
    out.geom_normals.resize(corners); // 3 floats per triangle = 1 per corner
    run_threads(T, [&](unsigned t) {
        Chunk &c = chunks[t];
        if (!c.keys.empty()) memcpy(keys.data() + c.corner0, c.keys.data(), c.keys.size() * sizeof(Key));
        if (!c.ng.empty()) memcpy(out.geom_normals.data() + c.corner0, c.ng.data(), c.ng.size() * sizeof(float));
        std::vector<Key>().swap(c.keys);
        std::vector<float>().swap(c.ng);
    });

    // ---- D: vertex sharing.  rep[c] = first corner with the key of corner c
    std::vector<uint32_t> rep(corners);
    std::vector<uint8_t> shard_of(corners);
    run_threads(T, [&](unsigned t) {
        const size_t lo = corners * t / T, hi = corners * (t + 1) / T;
        for (size_t c = lo; c < hi; c++) shard_of[c] = (uint8_t)((key_hash(keys[c]) >> 40) % T);
    });
    run_threads(T, [&](unsigned t) {
        size_t mine = 0;
        for (size_t c = 0; c < corners; c++) mine += (shard_of[c] == t);
        size_t cap = 64;
        while (cap < mine * 2) cap <<= 1;
        std::vector<uint32_t> table(cap, 0xFFFFFFFFu);
        for (size_t c = 0; c < corners; c++) {
            if (shard_of[c] != t) continue;
            const Key &k = keys[c];
            size_t slot = (size_t)key_hash(k) & (cap - 1);
            while (true) {
                const uint32_t other = table[slot];
                if (other == 0xFFFFFFFFu) {
                    table[slot] = (uint32_t)c;
                    rep[c] = (uint32_t)c;
                    break;
                }
                if (memcmp(keys[other].d, k.d, sizeof k.d) == 0) {
                    rep[c] = other;
                    break;
                }
                slot = (slot + 1) & (cap - 1);
            }
        }
    });
    // number the first corners in file order
    std::vector<size_t> part(T + 1, 0);
    run_threads(T, [&](unsigned t) {
        const size_t lo = corners * t / T, hi = corners * (t + 1) / T;
        size_t n = 0;
        for (size_t c = lo; c < hi; c++) n += (rep[c] == c);
        part[t + 1] = n;
    });
    for (unsigned t = 0; t < T; t++) part[t + 1] += part[t];
    const size_t num_vertices = part[T];
    std::vector<uint32_t> number(corners);
    out.positions.resize(num_vertices * 3);
    out.texcoords.resize(num_vertices * 2);
    out.normals.resize(num_vertices * 3);
    run_threads(T, [&](unsigned t) {
        const size_t lo = corners * t / T, hi = corners * (t + 1) / T;
        size_t n = part[t];
        for (size_t c = lo; c < hi; c++)
            if (rep[c] == c) {
                const Key &k = keys[c];
                for (int e = 0; e < 3; e++) out.positions[3 * n + e] = unbits(k.d[e]);
                for (int e = 0; e < 2; e++) out.texcoords[2 * n + e] = unbits(k.d[3 + e]);
                for (int e = 0; e < 3; e++) out.normals[3 * n + e] = unbits(k.d[5 + e]);
                number[c] = (uint32_t)n++;
            }
    });
    out.indices.resize(corners);
    run_threads(T, [&](unsigned t) {
        const size_t lo = corners * t / T, hi = corners * (t + 1) / T;
        for (size_t c = lo; c < hi; c++) out.indices[c] = number[rep[c]];
    });
    return true;
}

uint64_t mesh_fingerprint(const Mesh &m)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < bytes; i++) {
            h ^= b[i];
            h *= 1099511628211ull;
        }
        h ^= bytes;
        h *= 1099511628211ull;
    };
    mix(m.positions.data(), m.positions.size() * 4);
    mix(m.texcoords.data(), m.texcoords.size() * 4);
    mix(m.normals.data(), m.normals.size() * 4);
    mix(m.indices.data(), m.indices.size() * 4);
    mix(m.geom_normals.data(), m.geom_normals.size() * 4);
    return h;
}

} // namespace host
