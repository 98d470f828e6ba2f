// frame.hpp — headless frame driver: the per-frame pass order of the reference's main loop
// (src/main.cpp:261-274 construction, :345-391 passes, :416-420 history remaps) around the
// rendergraph mirror, plus a flat C interface (vkrh_*) so Python launchers (bench.py, tests,
// the multi-GPU driver) can run it one stage at a time and interleave RCCL exchanges.
#ifndef VKR_HOST_FRAME_HPP_INCLUDED
#define VKR_HOST_FRAME_HPP_INCLUDED
#include <stdint.h>
#include "../../include/vkr_postfx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vkrh_config {
  uint32_t full_width, full_height;  /* whole frame                                        */
  int32_t  origin_x, origin_y;       /* window of the frame held by this process           */
  uint32_t width, height;            /* window extent (== full on one GPU)                 */
  uint32_t tiled;                    /* 1: SSR reads the gathered whole-frame images       */
  void*    stream;                   /* HIP stream all passes are recorded on              */
} vkrh_config;

/* views / projection as 16 floats, column-major (glm layout) */
typedef struct vkrh_camera {
  float view[16], prev_view[16], projection[16];
  float fovy, aspect, znear, zfar;
} vkrh_camera;

enum {
  VKRH_STAGE_LUT        = 1u << 0,  /* ssr.preintegrate_pdf (main.cpp:269)                              */
  VKRH_STAGE_GBUFFER    = 1u << 1,  /* synthetic G-buffer for the current camera (replaces main.cpp:345) */
  VKRH_STAGE_PREV_DEPTH = 1u << 2,  /* prev_depth mip 0 from the previous camera + its Hi-Z mips        */
  VKRH_STAGE_DOWNSAMPLE = 1u << 3,  /* downsample_pass.run (main.cpp:347)                               */
  VKRH_STAGE_HIZ_TAIL   = 1u << 4,  /* tiled only: coarse mips of the gathered whole-frame pyramid       */
  VKRH_STAGE_SSR        = 1u << 5,  /* ssr.run: trace, filter, blur (main.cpp:375)                       */
  VKRH_STAGE_GTAO       = 1u << 6,  /* gtao main, filter, accumulate (main.cpp:384-388)                  */
  VKRH_STAGE_TAA        = 1u << 7,  /* taa_pass.run (main.cpp:391)                                       */
  VKRH_STAGE_SHADING    = 1u << 8,  /* shading_pass.draw -> color_out_tex, which TAA then resolves (main.cpp:390) */
  VKRH_STAGE_BRDF_LUT   = 1u << 9,  /* ssr.preintegrate_brdf (main.cpp:270)                             */
  VKRH_STAGE_GTAO_MAIN_ONLY = 1u << 10, /* gtao.add_main_pass alone (BASELINE configs[0])                 */
  /* passes the reference ships but never records                           */
  VKRH_STAGE_GTAO_GRAPHICS      = 1u << 11, /* gtao.add_main_pass_graphics, add_filter_pass, add_reprojection_pass */
  VKRH_STAGE_GTAO_DEINTERLEAVED = 1u << 12, /* gtao.deinterleave_depth, add_main_pass_deinterleaved                */
  VKRH_STAGE_SCREEN_TRACE       = 1u << 13, /* ScreenSpaceTrace main, filter, accumulate                           */
  VKRH_STAGE_SSR_CLASSIFIED     = 1u << 14, /* ssr.run with tile classification + indirect trace (advanced_ssr.cpp:547-550) */
  VKRH_STAGE_SSR_TRACE          = 1u << 15, /* first half of ssr.run: the trace (needs the Hi-Z pyramid)                    */
  VKRH_STAGE_SSR_RESOLVE        = 1u << 16, /* second half of ssr.run: filter + blur (needs albedo at the hit positions)      */
  VKRH_STAGE_RASTER             = 1u << 17, /* scene_renderer.draw_taa on the loaded scene (main.cpp:345) instead of the generator */
  /* tiled, hit normals by request: the trace as two stages around the arrival of the gathered pyramid — the head marches on the
   * window's own levels 1..gathered_mips and parks what needs more, the resume finishes the parked rays on the whole-frame
   * pyramid (after VKRH_STAGE_HIZ_TAIL).  HEAD + RESUME leave what VKRH_STAGE_SSR_TRACE leaves.                          */
  VKRH_STAGE_SSR_TRACE_HEAD     = 1u << 18,
  VKRH_STAGE_SSR_TRACE_RESUME   = 1u << 19,
  VKRH_STAGE_DOWNSAMPLE_NEXT    = 1u << 20,  /* the downsample of the NEXT frame into the G-buffer's second set (pipelined tiled frame) */
  VKRH_STAGE_CHAIN      = (1u << 3) | (1u << 5) | (1u << 6) | (1u << 7)
};

typedef void* (*vkrh_alloc_fn)(uint64_t bytes, void* user);
typedef void (*vkrh_free_fn)(void* ptr, void* user);

/* replace the device allocator used for every graph image (call before vkrh_create) */
void vkrh_set_allocator(vkrh_alloc_fn alloc, vkrh_free_fn free_fn, void* user);

void* vkrh_create(const vkrh_config* cfg);
void  vkrh_destroy(void* frame);
const char* vkrh_last_error(void);

int vkrh_set_camera(void* frame, const vkrh_camera* cam);
/* pin the host-side randoms of the reference (gtao.cpp:109-111 rand(), advanced_ssr.cpp:168-171 counter) */
int vkrh_pin_randoms(void* frame, float gtao_angle_jitter, uint32_t gtao_frame_count, uint32_t ssr_counter);
/* Scene for VKRH_STAGE_RASTER (replaces scene::load_tinygltf_scene, main.cpp:250): draw i renders
 * indices [index_offset, +index_count) with base vertex vertex_offset under `transform` (16 floats,
 * glm layout); texture index 0xFFFFFFFF = none.  Textures: RGBA8 mip chains, level 0 first. */
typedef struct vkrh_scene_draw {
  float    transform[16];
  uint32_t vertex_offset, index_offset, index_count;
  uint32_t albedo_tex_index, metalic_roughness_index, clip_alpha;
} vkrh_scene_draw;
typedef struct vkrh_scene_texture {
  uint32_t width, height, mip_levels, reserved;
  const uint8_t* levels[16];
} vkrh_scene_texture;
int vkrh_load_scene(void* frame, const vkr_raster_vertex* vertices, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                    const vkrh_scene_draw* draws, uint32_t draw_count, const vkrh_scene_texture* textures, uint32_t texture_count);
/* pin ScreenSpaceTrace's per-frame randoms (screen_trace.cpp:49-53) */
int vkrh_pin_screen_trace(void* frame, float angle_jitter, float random_offset, uint32_t frame_count);
/* 1 when the program table of the host layer knows `name` (a name of src/shaders/config.json or one of this path's own) */
int vkrh_has_program(const char* name);
int vkrh_set_gtao_mode(void* frame, uint32_t use_mis, uint32_t two_directions);
/* material mode of the synthetic G-buffer (VKRH_STAGE_GBUFFER): 0, or VKR_SYNTH_TEXTURED_ROUGHNESS (include/vkr_postfx.h) */
int vkrh_set_synth_flags(void* frame, uint32_t flags);
/* tiled frames: whole-frame Hi-Z view mips 0..mips-1 (image mips 1..mips) arrive by all-gather (default 4; tiles whose
 * extent is only divisible by 8, like the 15360x1080 strips of config 4, gather 3); VKRH_STAGE_HIZ_TAIL rebuilds the rest */
int vkrh_set_gathered_mips(void* frame, uint32_t mips);
/* record the stages in `mask` (canonical order) and submit them on the stream */
int vkrh_run(void* frame, uint32_t stage_mask);
/* end-of-frame remaps (main.cpp:416-420); swap_depth = 0 keeps a static G-buffer */
int vkrh_end_frame(void* frame, uint32_t swap_depth);
/* current descriptor of a named image ("depth", "taa_target", ...) */
int vkrh_image(void* frame, const char* name, uint32_t base_mip, uint32_t mip_count, vkr_img* out);
/* copies a named device buffer ("reflective_tiles", "glossy_tiles", "reflective_indirect",
 * "glossy_indirect") to host memory after synchronising the stream; returns its size in *bytes */
int vkrh_read_buffer(void* frame, const char* name, void* dst, uint64_t capacity, uint64_t* bytes);
/* one layer of a named array image ("deinterleaved_depth") */
int vkrh_image_layer(void* frame, const char* name, uint32_t layer, vkr_img* out);
/* Reads `name` (mip) back through ReadBackSystem and writes it with the reference's capture writers
 * (main.cpp:118-176): kind 0 = depth CSV (24-bit hex), 1 = depth PNG, 2 = RGBA8 PNG (alpha 255). */
int vkrh_capture(void* frame, const char* name, uint32_t mip, uint32_t kind, const char* path);
/* CPU-only check of the capture writers: writes <dir>/depth.csv, depth.png, color.png from a
 * width x height pattern depth = (x * 65537 + y * 257 + 0xAB000000) (top byte = stencil, must be
 * masked), rgba = (x, y, x ^ y, 7).  No GPU is touched. */
int vkrh_selftest_writers(const char* dir, uint32_t width, uint32_t height);
/* per-task device timing with HIP events on the frame's stream */
int vkrh_enable_task_timing(void* frame, uint32_t on);
/* time only the task of that name (two event records per frame instead of two per pass) */
int vkrh_enable_task_timing_only(void* frame, const char* task);
/* synchronises and returns "name total_ms launches\n" lines accumulated since the last call */
const char* vkrh_collect_task_times(void* frame);
/* Exercises the rendergraph / pass error paths the reference signals with exceptions (no kernel is
 * launched; images come from the installed allocator).  Writes "case: message" lines into buf. */
int vkrh_selftest_errors(char* buf, uint32_t buf_size);
/* names of the tasks executed by the last vkrh_run, '\n'-separated */
const char* vkrh_last_tasks(void* frame);
/* the stream lane (0 = the frame's stream) each of those tasks was recorded on, space separated */
const char* vkrh_last_lanes(void* frame);
/* 0 (default): every task on the frame's stream; 1: independent tasks of one vkrh_run spread over up to three streams */
int vkrh_set_async(void* frame, uint32_t on);

/* ---- the tiled frame: one process per GPU, horizontal strips, RCCL over xGMI -----------------
 * Rank r of `world` owns rows [r * H / world, (r + 1) * H / world) of the full_width x full_height frame and holds
 * them plus `halo` rows above / below (clipped to the frame).  One vkrh_tiled_step() is one frame:
 *
 *   downsample | all-gather(depth mips 1..k) starts on the exchange stream; own albedo / normal rows go into the frame images
 *   TAA (needs neither)            | its halo refresh starts
 *   Hi-Z tail + SSR trace          (after the gather; rays that end on another rank's rows stay pending) | the requests are
 *                                    counted, the counts all-gathered, and — from the second frame on — the whole request /
 *                                    reply round for hit colours + hit normals and the deferred hit-normal test
 *                                    (vkr_sssr_validate) are enqueued on the exchange stream at once
 *   GTAO main, filter, accumulate  | its halo refresh starts
 *   SSR filter + blur              (after the replies are in place) | its halo refresh starts; history remaps
 *
 * Hit colours: the filter reads the albedo at the hit position of every valid ray, anywhere in the frame.  Each rank asks
 * the owners for the footprint rows outside its window (vkr_hit_requests / _reply / _scatter, 4-byte requests and 16-byte
 * replies moved with vkr_halo_exchange) instead of receiving the albedo of the whole frame.  The message sizes of a frame
 * come from the PREVIOUS frame's counts (segments of fixed room, identical on every rank), so the host is not in the path:
 * it looks at this frame's counts before it queues the filter and repeats the round exactly if a segment overflowed; only
 * the first frame waits for its counts (with GTAO queued).  albedo_by_gather = 1 restores the all-gather.
 *
 * Every exchange is one grouped RCCL launch (vkr_all_gather / vkr_halo_exchange) on the frame's own exchange stream,
 * ordered against the compute stream with events only: the host never blocks, and a halo refresh issued after the
 * pass that produces a surface is awaited right before the pass that consumes it in the NEXT frame.
 * comm == NULL builds the same object without a wire: a test harness then advances it phase by phase
 * (vkrh_tiled_phase) and moves the bytes between in-process ranks itself (vkrh_tiled_gather_parts / _halo_peers). */
typedef struct vkrh_tiled_config {
  uint32_t full_width, full_height;
  uint32_t rank, world;
  uint32_t halo;            /* full-res pixels, even, a multiple of 2^gathered_mips                          */
  uint32_t gathered_mips;   /* depth image-mips 1..k travel by all-gather (the tile extent must divide by 2^k) */
  uint32_t force_tiled;     /* world == 1: still run the gathers and the staged frame (rehearsal)              */
  uint32_t albedo_by_gather;/* 0 (default): hit colours AND hit normals by request / reply; 1: all-gather the albedo and the
                             * downsampled normals of the whole frame (round 2); 2: albedo by request, normals gathered      */
  void*    stream;          /* compute stream                                                                */
  vkr_comm* comm;           /* RCCL communicator of include/vkr_postfx.h, or NULL (no wire: lockstep harness) */
  /* NULL: world strips of full_height / world rows.  Otherwise world + 1 increasing row numbers, [0] = 0 and [world] =
   * full_height, identical on every rank: strip r is rows [row_bounds[r], row_bounds[r + 1]).  Every bound is a
   * multiple of 2^gathered_mips and of 2, every strip at least `halo` rows high.  Strips of different heights balance
   * ranks whose rows differ in cost (vkrh_balance_rows); their shares travel with vkr_all_gather_v.                  */
  const uint32_t* row_bounds;
} vkrh_tiled_config;
enum { VKRH_TILED_PHASES = 5, VKRH_GATHER_HIZ = 0, VKRH_GATHER_ALBEDO = 1, VKRH_HALO_TAA = 0, VKRH_HALO_AO = 1, VKRH_HALO_SSR = 2 };
void* vkrh_tiled_create(const vkrh_tiled_config* cfg);
void  vkrh_tiled_destroy(void* tiled);
void* vkrh_tiled_frame(void* tiled);                 /* the frame inside: every vkrh_* call above works on it       */
int   vkrh_tiled_step(void* tiled);                  /* one frame, exchanges included (comm != NULL or world == 1)  */
int   vkrh_tiled_flush(void* tiled);                 /* completes the halo refreshes the last frame left in flight  */
/* lockstep harness: phase p of the frame without the wire; what must cross ranks between phases is exposed below */
int   vkrh_tiled_phase(void* tiled, uint32_t phase);
int   vkrh_tiled_gather_parts(void* tiled, uint32_t which, vkr_gather_part* out, uint32_t capacity, uint32_t* count);
int   vkrh_tiled_halo_peers(void* tiled, uint32_t surface, vkr_halo_peer* out, uint32_t capacity, uint32_t* count);
/* The hit-colour request / reply (albedo_by_gather == 0), step by step for the lockstep harness — between phase 3 and phase 4
 * of every rank, the harness moving the bytes of each peer list exactly as vkr_halo_exchange would:
 *   vkrh_tiled_hit_counts    row[o] = the requests this rank has for owner o (counted on the device at the end of phase 2)
 *   vkrh_tiled_hit_requests  matrix[r * world + o] = rank r's count for owner o, identical on every rank: writes the
 *                            requests and returns who gets / sends which bytes
 *   vkrh_tiled_hit_replies   answers the requests that arrived and returns the peer list of the way back
 *   vkrh_tiled_hit_finish    writes the replies that arrived into the whole-frame albedo image                          */
int   vkrh_tiled_hit_counts(void* tiled, uint32_t* row);
int   vkrh_tiled_hit_requests(void* tiled, const uint32_t* matrix, vkr_halo_peer* peers, uint32_t capacity, uint32_t* count);
int   vkrh_tiled_hit_replies(void* tiled, vkr_halo_peer* peers, uint32_t capacity, uint32_t* count);
int   vkrh_tiled_hit_finish(void* tiled);
/* bytes this rank received over the wire for the hit colours in the last frame (requests in + replies in)              */
int   vkrh_tiled_hit_bytes(void* tiled, uint64_t* bytes);
/* native wire: how the hit-colour rounds of the frames so far went — [0] enqueued on the previous frame's capacities (no host
 * round trip), [1] exact rounds after the host had the counts (first frame), [2] rounds repeated because a segment overflowed */
int   vkrh_tiled_hit_rounds(void* tiled, uint64_t* rounds3);
/* The room of every rank-to-owner segment of the NEXT frame's hit round, from this frame's world x world counts (what every
 * rank derives for itself): count * percent / 100 + 64 in steps of 64; adjacent strips always keep a segment, a distant pair that
 * asked for nothing keeps none.                                                                                          */
int   vkrh_hit_capacities(const uint32_t* counts, uint32_t world, uint32_t percent, uint32_t* capacities);
/* Measurement (tools/wire_emulation.py): a frame of several ranks that the in-process harness has driven so far (comm NULL) —
 * its receive buffers hold what real peers send, the hit segments laid out by vkrh_hit_capacities(counts) — continues natively
 * (vkrh_tiled_step) on `comm`, made by vkr_comm_create_emulated: every exchange holds the exchange stream for its wire time and
 * delivers what is already there.  With a static scene that is what the peers would send again.  A second call swaps the
 * communicator (another link rate; counts ignored).                                                                          */
int   vkrh_tiled_emulate_wire(void* tiled, void* comm, const uint32_t* counts);
/* 1: two frames in flight (VKR_TILED_PIPELINE=1): vkrh_tiled_step downsamples the NEXT frame and starts its depth all-gather
 * right after this frame's trace; the TAA runs behind GTAO (host/frame.cpp: pipelined_step)                                 */
int   vkrh_tiled_pipelined(void* tiled);
/* 1: the trace runs in two stages around the depth all-gather (VKRH_STAGE_SSR_TRACE_HEAD / _RESUME) and the TAA after GTAO */
int   vkrh_tiled_local_first(void* tiled);
/* requests of the last frame that this rank could not answer from its window (0 unless the ranks' strips disagree); synchronises */
int   vkrh_tiled_hit_errors(void* tiled, uint32_t* errors);
/* Diagnostics: how long the compute stream stood still for each exchange.  vkrh_tiled_time_waits(on) brackets every wait
 * with an event pair (≈7 us of queue time per frame: for a calibration run, not for the timed one);
 * vkrh_tiled_wait_times returns the totals since the last call in ms — [0] Hi-Z gather, [1] albedo gather, [2] TAA halo,
 * [3] AO halo, [4] SSR halo — and synchronises the compute stream.                                                    */
int   vkrh_tiled_time_waits(void* tiled, uint32_t on);
int   vkrh_tiled_wait_times(void* tiled, float* ms5);
/* New strip bounds from the compute time every rank measured with the current ones (ms[r] over rows
 * [bounds_in[r], bounds_in[r + 1]); cost taken as uniform inside a strip): cuts the frame where the cumulative cost
 * reaches r / world of the total, rounded to `align` rows, no strip below `min_rows`.  Pure host arithmetic.        */
int   vkrh_balance_rows(const float* ms, const uint32_t* bounds_in, uint32_t world, uint32_t align, uint32_t min_rows, uint32_t* bounds_out);

#ifdef __cplusplus
}
#endif
#endif
