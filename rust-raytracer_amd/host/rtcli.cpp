// Command-line host of the MI355X ray-tracing core: the reference's main()
// (src/main.rs:100-189) and argument set (src/args.rs:19-44) over librtcore's C
// ABI.  The asian-dragon and san-miguel scene definitions are the reference's
// (src/scene.rs:94-129); cornell-box and sponza name the other two assets the
// reference ships (scenes/CornellBox, scenes/sponza) with a camera and light of
// this host's own choosing, since the reference defines none for them.  Any
// other OBJ file can be given with its own camera and light.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../../include/rtcore.h"
#include "obj_loader.hpp"
#include "png_writer.hpp"

namespace {

struct Args {
    std::string scene = "san-miguel"; // src/args.rs:21-22
    uint32_t samples = 128;           // -n
    uint32_t max_bounces = 2;         // -b
    uint32_t tile_size = 16;          // -t
    std::string mode = "stream-shadow-immediate";
    uint64_t seed = 1235468;
    bool singlethread = false;
    bool progress = false;
    // not in the reference
    uint32_t width = 1920, height = 1080; // src/main.rs:103-104 hard-codes these
    int gpu = 0;
    bool exact = false, visit_stats = false, load_only = false;
    std::string output;
    uint32_t samples_per_pass = 0;
    std::string checkpoint;      // accumulation buffer saved after every pass
    bool resume = false;         // continue from --checkpoint when it matches this frame
    bool preview = false;        // rewrite the output image after every pass
    std::string loader = "parallel"; // OBJ loader: parallel (default) or serial, same mesh
    unsigned threads = 0;            // loader threads, 0 = up to 16
    size_t loader_chunk = 65536;     // least text per loader thread
    uint32_t stop_after = 0;     // stop once this many samples/px are done (0 = all): for testing --resume
    bool have_camera = false, have_light = false;
    rt_camera camera{};
    float light_pos[3] = {0, 0, 0}, light_power = 1.f;
    std::string transform = "identity";
};

const char *kModes[] = {"stream", "stream-shadow-immediate", "stream-camera-only", "single-ray", "packet4", "packet8"};

void usage()
{
    puts("Usage: rtcli [OPTIONS]\n"
         "  -s, --scene <asian-dragon|san-miguel|cornell-box|sponza|FILE.obj>   [default: san-miguel]\n"
         "  -n, --samples <N>            [default: 128]\n"
         "  -b, --max-bounces <N>        [default: 2]\n"
         "  -t, --tile-size <N>          [default: 16]\n"
         "  -m, --mode <stream|stream-shadow-immediate|stream-camera-only|single-ray|packet4|packet8>\n"
         "                               accepted for compatibility: every mode runs the same estimator,\n"
         "                               here as one wavefront integrator on the GPU\n"
         "      --seed <N>               [default: 1235468]\n"
         "      --singlethread           accepted, no effect\n"
         "  -p, --progress               print after every render pass\n"
         "additions:\n"
         "      --width <N> --height <N> [default: 1920 1080]\n"
         "      --gpu <ID>               HIP device [default: 0]\n"
         "      --exact                  walk the reference's binary tree in the reference's order\n"
         "      --visit-stats            fill the per-visit TraceStats counters (implies the exact walk)\n"
         "      --samples-per-pass <N>   [default: as many as fit the path workspace]\n"
         "      --output <FILE.png|FILE.ppm>   [default: output/YYYYmmdd_HHMMSS.png]\n"
         "      --load-only              load the scene, build the tree on the host, print its statistics\n"
         "      --checkpoint <FILE>      save the float4 accumulation buffer after every pass (implies passes\n"
         "                               of --samples-per-pass samples, default 8)\n"
         "      --resume                 continue from --checkpoint if it holds the same frame (size, seed,\n"
         "                               bounces); the finished image is the one an uninterrupted run gives\n"
         "      --preview                rewrite the output image after every pass\n"
         "      --stop-after <N>         stop when N samples/px are done\n"
         "      --loader <parallel|serial>   OBJ loader; both give the same mesh [default: parallel]\n"
         "      --threads <N>            threads of the parallel loader [default: up to 16]\n"
         "  for FILE.obj scenes: --eye x,y,z --target x,y,z --up x,y,z --fov DEG\n"
         "                       --light-pos x,y,z --light-power P\n"
         "                       --transform <identity|asian-dragon|san-miguel|cornell-box|sponza>");
}

bool parse3(const char *s, float out[3]) { return sscanf(s, "%f,%f,%f", &out[0], &out[1], &out[2]) == 3; }

bool parse_args(int argc, char **argv, Args &a)
{
    auto need = [&](int &i) -> const char * {
        if (i + 1 >= argc) {
            fprintf(stderr, "error: a value is required for '%s'\n", argv[i]);
            return nullptr;
        }
        return argv[++i];
    };
    bool eye = false, target = false;
    a.camera.up[2] = 1.f;
    a.camera.fov_deg = 60.f;
    for (int i = 1; i < argc; i++) {
        const std::string o = argv[i];
        const char *v = nullptr;
        if (o == "-h" || o == "--help") {
            usage();
            exit(0);
        } else if (o == "-s" || o == "--scene") {
            if (!(v = need(i))) return false;
            a.scene = v;
        } else if (o == "-n" || o == "--samples") {
            if (!(v = need(i))) return false;
            a.samples = (uint32_t)strtoul(v, nullptr, 10);
        } else if (o == "-b" || o == "--max-bounces") {
            if (!(v = need(i))) return false;
            unsigned long b = strtoul(v, nullptr, 10);
            if (b > 255) { // u8 in the reference
                fprintf(stderr, "error: invalid value '%s' for '--max-bounces': not in 0..=255\n", v);
                return false;
            }
            a.max_bounces = (uint32_t)b;
        } else if (o == "-t" || o == "--tile-size") {
            if (!(v = need(i))) return false;
            a.tile_size = (uint32_t)strtoul(v, nullptr, 10);
        } else if (o == "-m" || o == "--mode") {
            if (!(v = need(i))) return false;
            bool known = false;
            for (const char *m : kModes) known |= (strcmp(m, v) == 0);
            if (!known) {
                fprintf(stderr, "error: invalid value '%s' for '--mode'\n", v);
                return false;
            }
            a.mode = v;
        } else if (o == "--seed") {
            if (!(v = need(i))) return false;
            a.seed = strtoull(v, nullptr, 10);
        } else if (o == "--singlethread") {
            a.singlethread = true;
        } else if (o == "-p" || o == "--progress") {
            a.progress = true;
        } else if (o == "--width") {
            if (!(v = need(i))) return false;
            a.width = (uint32_t)strtoul(v, nullptr, 10);
        } else if (o == "--height") {
            if (!(v = need(i))) return false;
            a.height = (uint32_t)strtoul(v, nullptr, 10);
        } else if (o == "--gpu") {
            if (!(v = need(i))) return false;
            a.gpu = atoi(v);
        } else if (o == "--exact") {
            a.exact = true;
        } else if (o == "--visit-stats") {
            a.visit_stats = true;
        } else if (o == "--load-only") {
            a.load_only = true;
        } else if (o == "--samples-per-pass") {
            if (!(v = need(i))) return false;
            a.samples_per_pass = (uint32_t)strtoul(v, nullptr, 10);
        } else if (o == "--output") {
            if (!(v = need(i))) return false;
            a.output = v;
        } else if (o == "--checkpoint") {
            if (!(v = need(i))) return false;
            a.checkpoint = v;
        } else if (o == "--loader") {
            if (!(v = need(i))) return false;
            a.loader = v;
            if (a.loader != "parallel" && a.loader != "serial") {
                fprintf(stderr, "error: invalid value '%s' for '--loader'\n", v);
                return false;
            }
        } else if (o == "--threads") {
            if (!(v = need(i))) return false;
            a.threads = (unsigned)strtoul(v, nullptr, 10);
        } else if (o == "--loader-chunk-bytes") { // testing: cut small files into many chunks
            if (!(v = need(i))) return false;
            a.loader_chunk = (size_t)strtoull(v, nullptr, 10);
        } else if (o == "--resume") {
            a.resume = true;
        } else if (o == "--preview") {
            a.preview = true;
        } else if (o == "--stop-after") {
            if (!(v = need(i))) return false;
            a.stop_after = (uint32_t)strtoul(v, nullptr, 10);
        } else if (o == "--eye") {
            if (!(v = need(i)) || !parse3(v, a.camera.eye)) return false;
            eye = true;
        } else if (o == "--target") {
            if (!(v = need(i)) || !parse3(v, a.camera.target)) return false;
            target = true;
        } else if (o == "--up") {
            if (!(v = need(i)) || !parse3(v, a.camera.up)) return false;
        } else if (o == "--fov") {
            if (!(v = need(i))) return false;
            a.camera.fov_deg = strtof(v, nullptr);
        } else if (o == "--light-pos") {
            if (!(v = need(i)) || !parse3(v, a.light_pos)) return false;
            a.have_light = true;
        } else if (o == "--light-power") {
            if (!(v = need(i))) return false;
            a.light_power = strtof(v, nullptr);
        } else if (o == "--transform") {
            if (!(v = need(i))) return false;
            a.transform = v;
        } else {
            fprintf(stderr, "error: unexpected argument '%s'\n", argv[i]);
            return false;
        }
    }
    a.have_camera = eye && target;
    if (a.resume && a.checkpoint.empty()) {
        fprintf(stderr, "error: --resume needs --checkpoint <FILE>\n");
        return false;
    }
    if (a.samples == 0 || a.tile_size == 0 || a.width < 2 || a.height < 2) {
        fprintf(stderr, "error: samples, tile size and image size must be positive (image at least 2x2)\n");
        return false;
    }
    return true;
}

struct SceneDef {
    std::string obj_path;
    float transform[9];
    rt_camera camera;
    rt_light light;
};

void set_transform(const std::string &name, float m[9])
{
    const float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    // rows of the matrices of src/scene.rs:97-102 and 113-118 (built from columns, then transposed)
    const float dragon[9] = {1.f / 1000.f, 0, 0, 0, 0, 1.f / 1000.f, 0, 1.f / 1000.f, 0};
    const float miguel[9] = {1, 0, 0, 0, 0, -1, 0, 1, 0};
    // Y-up assets turned Z-up like San Miguel; Sponza is modelled in centimetres
    const float sponza[9] = {0.01f, 0, 0, 0, 0, -0.01f, 0, 0.01f, 0};
    const float *pick = name == "asian-dragon" ? dragon
                        : (name == "san-miguel" || name == "cornell-box") ? miguel
                        : name == "sponza" ? sponza : id;
    memcpy(m, pick, sizeof id);
}

bool scene_definition(const Args &a, SceneDef &d)
{
    if (a.scene == "asian-dragon") { // src/scene.rs:95-110
        d.obj_path = "./scenes/asian_dragon_obj/asian_dragon.obj";
        set_transform("asian-dragon", d.transform);
        const float ex = 0.6f, ey = -1.f, ez = 0.25f;
        const float len = sqrtf((ex * ex + ey * ey) + ez * ez);
        d.camera = rt_camera{{ex / len * 2.5f, ey / len * 2.5f, ez / len * 2.5f}, {0.f, 0.f, 0.35f}, {0.f, 0.f, 1.f}, 60.f};
        d.light = rt_light{{5.f, -10.f, 5.f}, {300.f, 300.f, 300.f}};
    } else if (a.scene == "san-miguel") { // src/scene.rs:111-126
        d.obj_path = "./scenes/San_Miguel/san-miguel.obj";
        set_transform("san-miguel", d.transform);
        d.camera = rt_camera{{28.f, -1.65f, 1.8f}, {27.f, -1.65f, 1.8f}, {0.f, 0.f, 1.f}, 60.f};
        d.light = rt_light{{20.f, -2.5f, 3.f}, {20.f, 20.f, 20.f}};
    } else if (a.scene == "cornell-box") { // no definition in the reference: camera and light chosen here
        d.obj_path = "./scenes/CornellBox/CornellBox-Original.obj";
        set_transform("cornell-box", d.transform);
        d.camera = rt_camera{{0.f, -3.2f, 0.8f}, {0.f, 0.f, 0.8f}, {0.f, 0.f, 1.f}, 40.f};
        d.light = rt_light{{0.f, 0.f, 1.4f}, {3.f, 3.f, 3.f}};
    } else if (a.scene == "sponza") { // no definition in the reference: camera and light chosen here
        d.obj_path = "./scenes/sponza/sponza.obj";
        set_transform("sponza", d.transform);
        d.camera = rt_camera{{-12.f, 0.5f, 2.5f}, {0.f, 0.f, 3.f}, {0.f, 0.f, 1.f}, 60.f};
        d.light = rt_light{{0.f, 0.f, 12.f}, {600.f, 600.f, 600.f}};
    } else {
        d.obj_path = a.scene;
        set_transform(a.transform, d.transform);
        if (!a.load_only && (!a.have_camera || !a.have_light)) {
            fprintf(stderr, "error: an OBJ scene needs --eye, --target, --light-pos (and --light-power)\n");
            return false;
        }
        d.camera = a.camera;
        d.light = rt_light{{a.light_pos[0], a.light_pos[1], a.light_pos[2]}, {a.light_power, a.light_power, a.light_power}};
    }
    return true;
}

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Rust's {:.2?} on a Duration picks the unit by magnitude
std::string fmt_duration(double s)
{
    char b[64];
    if (s >= 1.0)
        snprintf(b, sizeof b, "%.2fs", s);
    else if (s >= 1e-3)
        snprintf(b, sizeof b, "%.2fms", s * 1e3);
    else
        snprintf(b, sizeof b, "%.2fµs", s * 1e6);
    return b;
}

// Checkpoint of a frame in progress: the float4 accumulation buffer (sums of r, g, b and of sample
// counts, src/main.rs:108) with the frame parameters the sums depend on.  Random numbers are keyed by
// (seed, pixel, sample), so continuing at `samples_done` gives the frame an uninterrupted run gives.
// `frame_hash` covers what else decides a sample's value: the mesh (positions after the scene
// transform, indices), camera, light and whether the exact traversal forms are used.
struct CheckpointHeader {
    char magic[8]; // "RTACCUM2"
    uint32_t width, height, samples_done, max_bounces, tile_size, exact;
    uint64_t seed;
    uint64_t frame_hash;
};

uint64_t fnv1a(uint64_t h, const void *data, size_t bytes)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < bytes; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

uint64_t frame_hash_of(uint64_t mesh_fingerprint, const rt_camera &cam, const rt_light &light)
{
    uint64_t h = 0xCBF29CE484222325ull;
    h = fnv1a(h, &mesh_fingerprint, sizeof mesh_fingerprint);
    h = fnv1a(h, cam.eye, sizeof cam.eye);
    h = fnv1a(h, cam.target, sizeof cam.target);
    h = fnv1a(h, cam.up, sizeof cam.up);
    h = fnv1a(h, &cam.fov_deg, sizeof cam.fov_deg);
    h = fnv1a(h, light.pos, sizeof light.pos);
    h = fnv1a(h, light.intensity, sizeof light.intensity);
    return h;
}

bool write_checkpoint(const std::string &path, const CheckpointHeader &h, const std::vector<float> &accum)
{
    const std::string tmp = path + ".tmp"; // rename is atomic: a killed run never leaves half a file
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(accum.data(), sizeof(float), accum.size(), f) == accum.size();
    ok = (fclose(f) == 0) && ok;
    return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

// 1 = loaded, 0 = no file, -1 = a file that does not belong to this frame, -2 = a checkpoint of this
// frame that holds more samples than `max_samples`
int read_checkpoint(const std::string &path, const CheckpointHeader &want, uint32_t max_samples, std::vector<float> &accum,
                    uint32_t &samples_done)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    CheckpointHeader h{};
    std::vector<float> data(accum.size());
    bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, want.magic, 8) == 0 && h.width == want.width &&
              h.height == want.height && h.max_bounces == want.max_bounces && h.tile_size == want.tile_size &&
              h.exact == want.exact && h.seed == want.seed && h.frame_hash == want.frame_hash &&
              fread(data.data(), sizeof(float), data.size(), f) == data.size();
    fclose(f);
    if (!ok) return -1;
    if (h.samples_done > max_samples) return -2;
    accum.swap(data);
    samples_done = h.samples_done;
    return 1;
}

int fail_rt(const char *what)
{
    fprintf(stderr, "error: %s: %s\n", what, rt_last_error());
    return 1;
}

} // namespace

int main(int argc, char **argv)
{
    Args args;
    if (!parse_args(argc, argv, args)) {
        fprintf(stderr, "\nFor more information, try '--help'.\n");
        return 2;
    }
    SceneDef def;
    if (!scene_definition(args, def)) return 2;

    // Scene::load, src/scene.rs:29-91
    auto t_load = std::chrono::steady_clock::now();
    host::Mesh mesh;
    std::string error;
    const bool loaded = args.loader == "serial" ? host::load_obj(def.obj_path, def.transform, mesh, error)
                                                : host::load_obj_parallel(def.obj_path, def.transform, args.threads, mesh, error,
                                                                          args.loader_chunk);
    if (!loaded) {
        fprintf(stderr, "error: %s\n", error.c_str());
        return 1;
    }
    printf("loaded %s, %u triangles in %s\n", def.obj_path.c_str(), mesh.num_triangles(),
           fmt_duration(seconds_since(t_load)).c_str());
    if (mesh.skipped_zero_area > 0) printf("skipped %u triangles with zero area\n", mesh.skipped_zero_area);
    if (mesh.num_triangles() == 0) {
        fprintf(stderr, "error: the scene has no triangles\n");
        return 1;
    }

    if (args.load_only) {
        auto t_bvh = std::chrono::steady_clock::now();
        std::vector<rt_flat_node> nodes(mesh.num_triangles());
        std::vector<uint32_t> order(mesh.num_triangles());
        rt_scene_info info{};
        rt_bvh_stats st{};
        if (rt_bvh_build_host(mesh.positions.data(), mesh.num_vertices(), mesh.indices.data(), mesh.num_triangles(), 0,
                              nodes.data(), order.data(), &info, &st) != RT_OK)
            return fail_rt("rt_bvh_build_host");
        printf("BvhStats { num_leaves: %u, num_objs: %u, max_depth: %u, max_objs: %u }\n", st.num_leaves, st.num_objs,
               st.max_depth, st.max_objs);
        printf("bvh build %s\n", fmt_duration(seconds_since(t_bvh)).c_str());
        printf("%u vertices, %u inner nodes\n", mesh.num_vertices(), info.num_inner_nodes);
        printf("mesh fingerprint %016llx\n", (unsigned long long)host::mesh_fingerprint(mesh));
        return 0;
    }

    rt_ctx *ctx = nullptr;
    if (rt_ctx_create(args.gpu, &ctx) != RT_OK) return fail_rt("rt_ctx_create");
    if (args.exact) {
        rt_ctx_set_option(ctx, "traversal_depth0", 0);
        rt_ctx_set_option(ctx, "traversal_deep", 1);
        rt_ctx_set_option(ctx, "coop_steps", 0);
        rt_ctx_set_option(ctx, "fast_tree_buckets", 0); // no second tree: nothing walks it
    }
    auto t_bvh = std::chrono::steady_clock::now();
    rt_scene *scene = nullptr;
    if (rt_scene_create(ctx, mesh.positions.data(), mesh.num_vertices(), mesh.indices.data(), mesh.num_triangles(),
                        mesh.geom_normals.data(), &scene) != RT_OK)
        return fail_rt("rt_scene_create");
    rt_bvh_stats st{};
    rt_scene_stats(scene, &st);
    printf("BvhStats { num_leaves: %u, num_objs: %u, max_depth: %u, max_objs: %u }\n", st.num_leaves, st.num_objs,
           st.max_depth, st.max_objs);
    printf("bvh build %s\n", fmt_duration(seconds_since(t_bvh)).c_str());

    const uint32_t tiles_x = (args.width + args.tile_size - 1) / args.tile_size;
    const uint32_t tiles_y = (args.height + args.tile_size - 1) / args.tile_size;
    const uint32_t num_tiles = tiles_x * tiles_y;
    // src/main.rs:117-127
    printf("%u %ux%u tiles, %u pixels/tile, %u samples/pixel, %u total samples/tile, max bounces %u, mode %s (GPU wavefront)\n",
           num_tiles, args.tile_size, args.tile_size, args.tile_size * args.tile_size, args.samples,
           args.tile_size * args.tile_size * args.samples, args.max_bounces, args.mode.c_str());

    std::vector<float> accum((size_t)args.width * args.height * 4, 0.f);
    rt_trace_stats stats{};
    CheckpointHeader ck{{'R', 'T', 'A', 'C', 'C', 'U', 'M', '2'}, args.width, args.height, 0, args.max_bounces,
                        args.tile_size, args.exact ? 1u : 0u, args.seed,
                        frame_hash_of(host::mesh_fingerprint(mesh), def.camera, def.light)};
    uint32_t start = 0;
    if (args.resume) {
        const int r = read_checkpoint(args.checkpoint, ck, args.samples, accum, start);
        if (r == -2) {
            fprintf(stderr, "error: %s already holds more than the %u samples asked for\n", args.checkpoint.c_str(),
                    args.samples);
            return 1;
        }
        if (r < 0) {
            fprintf(stderr,
                    "error: %s is not a checkpoint of this frame (size, bounces, tile size, seed, mesh, camera, light "
                    "or traversal mode differ)\n",
                    args.checkpoint.c_str());
            return 1;
        }
        if (r > 0) printf("resumed %s at %u/%u samples\n", args.checkpoint.c_str(), start, args.samples);
    }
    std::string out = args.output;
    if (out.empty()) { // src/main.rs:184-188
        mkdir("output", 0777);
        char name[64];
        time_t now = time(nullptr);
        strftime(name, sizeof name, "output/%Y%m%d_%H%M%S.png", localtime(&now));
        out = name;
    }
    std::vector<uint8_t> rgb((size_t)args.width * args.height * 3);
    auto write_image = [&]() -> bool {
        if (rt_resolve(ctx, accum.data(), args.width, args.height, rgb.data()) != RT_OK) {
            fail_rt("rt_resolve");
            return false;
        }
        const bool ppm = out.size() > 4 && out.compare(out.size() - 4, 4, ".ppm") == 0;
        if (!(ppm ? host::write_ppm(out, rgb.data(), args.width, args.height)
                  : host::write_png(out, rgb.data(), args.width, args.height))) {
            fprintf(stderr, "error: cannot write %s\n", out.c_str());
            return false;
        }
        return true;
    };

    auto t_render = std::chrono::steady_clock::now();
    // one call per frame; with --progress, --checkpoint or --preview one call per pass of the frame's samples
    const bool passes = args.progress || args.preview || !args.checkpoint.empty();
    const uint32_t chunk = passes ? (args.samples_per_pass ? args.samples_per_pass : 8u) : args.samples;
    const uint32_t end = args.stop_after ? std::min(args.stop_after, args.samples) : args.samples;
    const uint32_t first = std::min(start, args.samples);
    uint32_t done = first;
    for (uint32_t s0 = first; s0 < end; s0 += chunk) {
        rt_render_params p{};
        p.width = args.width;
        p.height = args.height;
        p.samples = std::min(chunk, end - s0);
        p.sample_offset = s0;
        p.max_bounces = args.max_bounces;
        p.tile_size = args.tile_size;
        p.shard_index = 0;
        p.shard_count = 1;
        p.samples_per_pass = args.samples_per_pass;
        p.flags = args.visit_stats ? RT_RENDER_VISIT_STATS : 0;
        p.seed = args.seed;
        if (rt_render(scene, &def.camera, &def.light, &p, accum.data(), &stats) != RT_OK) return fail_rt("rt_render");
        done = s0 + p.samples;
        if (args.progress) printf("%u/%u samples completed\n", done, args.samples);
        if (!args.checkpoint.empty()) {
            ck.samples_done = done;
            if (!write_checkpoint(args.checkpoint, ck, accum)) {
                fprintf(stderr, "error: cannot write %s\n", args.checkpoint.c_str());
                return 1;
            }
        }
        if (args.preview && done < end && !write_image()) return 1;
    }
    const double elapsed = seconds_since(t_render);

    // TraceStats::print, src/trace_stats.rs:61-77
    printf("TraceStats { total_rays: %llu, inner_visit: %llu, inner_rays: %llu, leaf_visit: %llu, leaf_rays: %llu, "
           "leaf_objs: %llu, obj_intersect: %llu }\n",
           (unsigned long long)stats.total_rays, (unsigned long long)stats.inner_visit,
           (unsigned long long)stats.inner_rays, (unsigned long long)stats.leaf_visit,
           (unsigned long long)stats.leaf_rays, (unsigned long long)stats.leaf_objs,
           (unsigned long long)stats.obj_intersect);
    if (args.visit_stats) {
        printf("avg %.1f rays per inner visit\n", (float)stats.inner_rays / (float)stats.inner_visit);
        printf("avg %.1f rays %.1f objs per leaf visit\n", (float)stats.leaf_rays / (float)stats.leaf_visit,
               (float)stats.leaf_objs / (float)stats.leaf_visit);
        printf("avg %.1f intersects per ray\n", (float)stats.obj_intersect / (float)stats.total_rays);
    } else {
        printf("per-visit counters not collected (--visit-stats)\n");
    }
    // src/main.rs:170-179, over the samples this run rendered
    const uint32_t rendered = done - first;
    printf("%s for %u samples/px, %.2f samples/px/sec, %.2f MRays/sec, %.2f MSamples/sec\n",
           fmt_duration(elapsed).c_str(), rendered, (float)rendered / (float)elapsed,
           (float)stats.total_rays / 1000000.f / (float)elapsed,
           (float)((double)args.width * args.height * rendered) / 1000000.f / (float)elapsed);

    if (!write_image()) return 1;
    printf("wrote %s\n", out.c_str());
    rt_scene_destroy(scene);
    rt_ctx_destroy(ctx);
    return 0;
}
