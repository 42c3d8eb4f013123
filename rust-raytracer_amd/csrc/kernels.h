// Launch interface between the C-ABI layer (rtcore.hip) and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

#ifndef RTC_TRACE_BLOCK
#define RTC_TRACE_BLOCK 256
#endif
#ifndef RTC_LDS_STACK_DEPTH
#define RTC_LDS_STACK_DEPTH 24
#endif

// occupancy the trace kernels are compiled for (waves per SIMD): bounds the
// VGPR allocation; together with the LDS stack it sets resident blocks per CU
// any-hit launches on the wide tree: visit hit children nearest first (1) or in slot order (0)
#ifndef RTC_ANYHIT_SORTED
#define RTC_ANYHIT_SORTED 0
#endif
// 1: the coherent depth-0 kernels are built for more resident waves than the others (TraceTune)
#ifndef RTC_TRACE_TUNED
#define RTC_TRACE_TUNED 1
#endif
// exact forms without visit counters: the while-while form (0) gains 2.5 % at 6 waves per SIMD, the
// stepwise form (1) loses 10 % there to spills
#ifndef RTC_TUNE_EXACT0_WAVES
#define RTC_TUNE_EXACT0_WAVES 6
#endif
#ifndef RTC_TUNE_EXACT1_WAVES
#define RTC_TUNE_EXACT1_WAVES RTC_TRACE_WAVES_PER_SIMD
#endif
#ifndef RTC_TUNE_PRIMARY_WAVES
#define RTC_TUNE_PRIMARY_WAVES 6
#endif
// 1: the kernel keeps a 16-bit entry distance beside every LDS stack entry (6 bytes per entry: 16
// entries in the LDS that 24 plain ones took) and drops deferred nodes a found hit has put out of
// reach when it pops them.  On for the while-while closest-hit kernel of depth 0 (-3 % on all three
// workloads); off for the stepwise kernel of the deeper launches, where the longer push and pop
// paths cost more than the skipped items save (+4 % San-Miguel-class, +9 % dragon-class).
#ifndef RTC_TUNE_PRIMARY_DIST
#define RTC_TUNE_PRIMARY_DIST 1
#endif
#ifndef RTC_TUNE_EXT2_DIST
#define RTC_TUNE_EXT2_DIST 0
#endif
#ifndef RTC_TUNE_PRIMARY_LDS
#define RTC_TUNE_PRIMARY_LDS (RTC_TUNE_PRIMARY_DIST ? 16 : RTC_LDS_STACK_DEPTH)
#endif
#ifndef RTC_TUNE_ANY3_WAVES
#define RTC_TUNE_ANY3_WAVES 7
#endif
#ifndef RTC_TUNE_ANY3_LDS
#define RTC_TUNE_ANY3_LDS 16
#endif
#ifndef RTC_TUNE_ANY2_WAVES
#define RTC_TUNE_ANY2_WAVES 7
#endif
#ifndef RTC_TUNE_ANY2_LDS
#define RTC_TUNE_ANY2_LDS 16
#endif
#ifndef RTC_TUNE_EXT2_WAVES
#define RTC_TUNE_EXT2_WAVES 6
#endif
#ifndef RTC_TUNE_EXT2_LDS
#define RTC_TUNE_EXT2_LDS (RTC_TUNE_EXT2_DIST ? 16 : RTC_LDS_STACK_DEPTH)
#endif
#ifndef RTC_TUNE_EXT6_WAVES
#define RTC_TUNE_EXT6_WAVES 5
#endif
#ifndef RTC_TUNE_EXT6_LDS
#define RTC_TUNE_EXT6_LDS RTC_LDS_STACK_DEPTH
#endif
#ifndef RTC_TUNE_ANY6_WAVES
#define RTC_TUNE_ANY6_WAVES 6
#endif
#ifndef RTC_TUNE_ANY6_LDS
#define RTC_TUNE_ANY6_LDS 16
#endif
#ifndef RTC_TUNE_EXT8_WAVES
#define RTC_TUNE_EXT8_WAVES 5
#endif
#ifndef RTC_TUNE_EXT8_LDS
#define RTC_TUNE_EXT8_LDS RTC_LDS_STACK_DEPTH
#endif
#ifndef RTC_TUNE_ANY8_WAVES
#define RTC_TUNE_ANY8_WAVES 6
#endif
#ifndef RTC_TUNE_ANY8_LDS
#define RTC_TUNE_ANY8_LDS 16
#endif
#ifndef RTC_TUNE_EXT7_WAVES
#define RTC_TUNE_EXT7_WAVES 6
#endif
#ifndef RTC_TUNE_EXT7_LDS
#define RTC_TUNE_EXT7_LDS 19
#endif
#ifndef RTC_TUNE_ANY7_WAVES
#define RTC_TUNE_ANY7_WAVES 7
#endif
#ifndef RTC_TUNE_ANY7_LDS
#define RTC_TUNE_ANY7_LDS 16
#endif
// While-while form on the wide tree: the node loop ends when fewer lanes than this are still at a
// node, instead of running until the last lane has reached its leaf while the others wait there.
// Measured on one GPU (depth-0 extend launch, 1 = the plain form): San-Miguel-class 39.2 ms -> 37.0 at 8,
// 36.4 at 24, 35.6 at 40, 35.4 at 48; dragon-class 16.4 -> 15.4; Sponza-class unchanged.
#ifndef RTC_WW_NODE_MIN
#define RTC_WW_NODE_MIN 40
#endif
#ifndef RTC_WW_NODE_RELATIVE
#define RTC_WW_NODE_RELATIVE 1
#endif
// Fewest LDS stack entries any trace kernel keeps per lane: the global overflow area holds the
// entries beyond that, so it is sized from this value.
#define RTC_MIN2_(a, b) ((a) < (b) ? (a) : (b))
#if RTC_TRACE_TUNED
#define RTC_MIN_LDS_STACK_DEPTH \
    RTC_MIN2_(RTC_MIN2_(RTC_MIN2_(RTC_LDS_STACK_DEPTH, RTC_TUNE_EXT2_LDS), RTC_TUNE_PRIMARY_LDS), \
              RTC_MIN2_(RTC_MIN2_(RTC_MIN2_(RTC_TUNE_ANY3_LDS, RTC_TUNE_ANY2_LDS), RTC_MIN2_(RTC_TUNE_EXT6_LDS, RTC_TUNE_ANY6_LDS)), \
                        RTC_MIN2_(RTC_MIN2_(RTC_TUNE_EXT7_LDS, RTC_TUNE_ANY7_LDS), RTC_MIN2_(RTC_TUNE_EXT8_LDS, RTC_TUNE_ANY8_LDS))))
#else
#define RTC_MIN_LDS_STACK_DEPTH RTC_LDS_STACK_DEPTH
#endif
// the two-rays-per-lane form: waves per SIMD it is built for, LDS stack entries per ray
#ifndef RTC_DUAL_WAVES_PER_SIMD
#define RTC_DUAL_WAVES_PER_SIMD 4
#endif
#ifndef RTC_DUAL_LDS_STACK_DEPTH
#define RTC_DUAL_LDS_STACK_DEPTH 12
#endif
#ifndef RTC_TRACE_WAVES_PER_SIMD
#define RTC_TRACE_WAVES_PER_SIMD 5
#endif

namespace rtc {

// Ray-fetch counters of one trace launch: kFetchRanges counters, each alone in
// a 128-byte line.
#ifndef RTC_FETCH_RANGES
#define RTC_FETCH_RANGES 32
#endif
constexpr unsigned kFetchRanges = RTC_FETCH_RANGES;
constexpr unsigned kFetchStride = 32; // uints between counters
constexpr unsigned kFetchUints = kFetchRanges * kFetchStride;

struct TraceLaunch {
    unsigned *fetch;     // kFetchUints uints, zeroed before the launch
    unsigned *ovf;       // traversal stack overflow area
    StatsDev *stats;
    unsigned min_active; // a wave refills when fewer lanes than this are traversing (1..64)
    unsigned mode;       // 0 = while-while loop, 1 = one item (node or triangle) per iteration,
                         // 2 = one item per iteration on the quantized 4-wide tree, 3 = while-while loop on it
                         // (no visit counters in 2 and 3), 4 = form 1 in the visit order of the reference's
                         // single-ray closest hit (stream entry points only), 5 = two rays per lane, 6 = node walk
                         // and triangle tests side by side, 7 = triangle tests through a wave-level job queue
    const unsigned *perm; // optional: process ray perm[i] at position i (queue sorted by key), or null
    unsigned perm_n;      // perm covers (and permutes) queue entries 0..perm_n-1
    // stragglers: a ray that has used coop_steps loop iterations is finished by its
    // whole wave on a wave-private work stack of coop_cap 8-byte entries (coop + wave*cap)
    void *coop;
    unsigned coop_cap;
    unsigned coop_steps; // 0xFFFFFFFF = never
    unsigned depth;      // path depth the launch belongs to (slot of the per-depth counters)
    unsigned job_units_ok; // the wide tree's unit indices fit a triangle job of form 7 (else form 2 runs)
};

// StreamRay (src/ray.rs:6-12) as three float4.
struct alignas(16) rt_stream_ray_dev {
    float4 origin_far;
    float4 dir_recip_near;
    float4 direction;
};

// Camera frame derived on the host (src/camera.rs:25-37, 63-75).
struct CameraDev {
    float eye[3];
    float right[3];
    float up[3];
    float top_left[3];
    float step_x[3];
    float step_y[3];
    float pixel_size;
};

// One render pass: a contiguous run of the shard's tiles times up to
// samples_per_pass samples.
struct PassDev {
    uint32_t width, height;
    uint32_t tile_size, tiles_x;
    uint32_t shard_index, shard_count;
    uint32_t tile_begin, tile_count; // in the shard's own tile numbering
    uint32_t samples_per_pass;       // slot stride per pixel
    uint32_t samples_this_pass;      // <= samples_per_pass
    uint32_t sample_base;            // global index of sample 0 of this pass
    uint32_t num_slots;              // tile_count * tile_size^2 * samples_per_pass
    uint32_t seed_key;
    // derived (fill_pass_shifts): log2 of samples_per_pass, tile_size and tile_size^2 where these
    // are powers of two, else 0xFFFFFFFF; num_slots / samples_per_pass
    uint32_t k_shift, ts_shift, ts2_shift, pixel_slots;
};

inline uint32_t pow2_shift(uint32_t v)
{
    if (v == 0 || (v & (v - 1u)) != 0) return 0xFFFFFFFFu;
    uint32_t s = 0;
    while ((1u << s) != v) s++;
    return s;
}
inline void fill_pass_shifts(PassDev &ps)
{
    ps.k_shift = pow2_shift(ps.samples_per_pass);
    ps.ts_shift = pow2_shift(ps.tile_size);
    ps.ts2_shift = ps.ts_shift == 0xFFFFFFFFu ? 0xFFFFFFFFu : 2u * ps.ts_shift;
    ps.pixel_slots = ps.samples_per_pass ? ps.num_slots / ps.samples_per_pass : 0u;
}

struct ShadeArgs {
    // depth 0 without queues (primary != 0): ray and PathInfo of entry i are rebuilt from pass slot
    // i, its hit distance is hit_t[i], and the entry count is n_in_val (n_in null)
    unsigned primary;
    unsigned depth; // bounces the paths of this queue have behind them
    unsigned n_in_val;
    const float *hit_t;
    CameraDev cam;
    PassDev ps;
    const QRay *rays_in;
    const PathInfo *infos_in;
    const uint32_t *hit_ids;
    const uint32_t *n_in;
    const float *normals;
    QRay *rays_out;
    PathInfo *infos_out;
    uint32_t *n_out;
    QRay *shadow_rays;
    ShadowInfo *shadow_infos;
    uint32_t *n_shadow;
    float light_pos[3];
    float light_intensity[3];
    uint32_t max_bounces;
    uint32_t seed_key;
    // optional sort keys (Morton code of the ray origin), null = not wanted
    uint32_t *shadow_keys;
    uint32_t *bounce_keys;
    float key_lo[3];
    float key_scale[3]; // 1024 / extent per axis
    uint32_t key_mask;  // high bits of the 30-bit code that take part in the order
    uint32_t key_dir_shift; // bounce rays: the direction's octant takes the three lowest ordered bits, at this
                            // position (32: the key is the origin's code alone)
};

// sorts (key, index) pairs by bits [begin_bit, end_bit) of the key; indices are
// 0..n-1.  temp must hold sort_temp_bytes(n) bytes.
size_t sort_temp_bytes(unsigned n, unsigned begin_bit, unsigned end_bit); // 0 on error, else >= 256
hipError_t sort_indices_by_key(void *temp, size_t temp_bytes, const unsigned *keys_in, unsigned *keys_out,
                               unsigned *perm_out, unsigned n, unsigned begin_bit, unsigned end_bit,
                               hipStream_t stream);

void launch_trace_stream(bool anyhit, bool stats, const SceneDev &sc, rt_stream_ray_dev *rays, unsigned *hits,
                         unsigned char *occluded, unsigned n, const TraceLaunch &tl, unsigned max_blocks,
                         hipStream_t stream);
void launch_extend(bool stats, const SceneDev &sc, QRay *rays, unsigned *hit_ids, const unsigned *n_ptr,
                   unsigned n_upper, const TraceLaunch &tl, unsigned max_blocks, hipStream_t stream);
void launch_connect(bool stats, const SceneDev &sc, const QRay *rays, const ShadowInfo *infos,
                    const unsigned *n_ptr, unsigned n_upper, float *rad_r, float *rad_g, float *rad_b,
                    const TraceLaunch &tl, unsigned max_blocks, hipStream_t stream);
void launch_extend_primary(const SceneDev &sc, const CameraDev &cam, const PassDev &ps, unsigned *hit_ids,
                           float *hit_t, const TraceLaunch &tl, unsigned max_blocks, hipStream_t stream);
void launch_set_count(unsigned *dst, unsigned value, hipStream_t stream);
void launch_count_rays(const unsigned *n_ptr, StatsDev *dstats, unsigned depth, bool add_total, hipStream_t stream);
void launch_generate(const CameraDev &cam, const PassDev &ps, QRay *rays, PathInfo *infos, unsigned *n_out,
                     float *rad_r, float *rad_g, float *rad_b, hipStream_t stream);
void launch_gather_far(const rt_stream_ray_dev *rays, float *far_t, unsigned n, hipStream_t stream);
void launch_generate_stream(const CameraDev &cam, const PassDev &ps, rt_stream_ray_dev *rays,
                            unsigned long long total, hipStream_t stream);
void launch_shade(const ShadeArgs &a, unsigned n_upper, hipStream_t stream);
void launch_accumulate(const PassDev &ps, const float *rad_r, const float *rad_g, const float *rad_b,
                       float *accum, hipStream_t stream);
void launch_resolve(const float *accum, unsigned long long npix, unsigned char *rgb, hipStream_t stream);
int trace_blocks_per_cu();

} // namespace rtc
