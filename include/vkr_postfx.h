/*
 * vkr_postfx.h — C-ABI of the MI355X post-process hot path (Hi-Z / SSR / GTAO / TAA).
 *
 * One entry point per reference compute/fragment *program* (src/shaders/config.json)
 * on the hot path.  Arguments follow the reference shader's binding order: POD image
 * views for sampled / storage images, a pointer to the exact UBO struct the reference
 * uploads, the exact push-constant struct, then the HIP stream.  Plain pointers and
 * sizes only; no torch / Vulkan / glm types.
 *
 * Conventions
 *   - all image memory is device (HBM) memory, pitch-linear, rows 256-B aligned,
 *     4 / 8 bytes per texel in the reference's storage format (vkr_format);
 *   - every call is asynchronous on `stream`; returns 0 or a non-zero hipError_t-style
 *     code (message via vkr_last_error()); nothing is allocated inside a call;
 *   - UBO / push-constant structs are read on the host at call time (they become
 *     kernel arguments), so they may live on the caller's stack;
 *   - a `vkr_img` is a *view*: mip 0 of the view is the first mip the reference binds
 *     (e.g. GTAO binds depth image-mip 1 as a 1-mip view, gtao.cpp:119);
 *   - multi-GPU tiling: an image may hold only a window of the frame.  `full_width/
 *     full_height` is the whole frame's extent at view-mip 0, `origin_x/origin_y` the
 *     window's position in it.  Single GPU: origin 0, full == width/height.  Output
 *     images define the set of pixels a kernel computes (their window).
 */
#ifndef VKR_POSTFX_H_INCLUDED
#define VKR_POSTFX_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKR_MAX_MIPS 16
#define VKR_HALTON_SEQ_SIZE 128  /* advanced_ssr.cpp:6, trace.comp:18 */

/* Storage formats used by the path (scene_renderer.cpp:13-43, gtao.cpp:26-47,
 * advanced_ssr.cpp:62-92, taa.cpp:6).  Values are ours, not VkFormat. */
typedef enum vkr_format {
  VKR_FMT_UNDEFINED      = 0,
  VKR_FMT_D24_UNORM_S8   = 1,  /* uint32: depth in bits 0..23, stencil 24..31      */
  VKR_FMT_RG16_UNORM     = 2,  /* 2 x uint16                                       */
  VKR_FMT_RG16_SFLOAT    = 3,  /* 2 x fp16                                         */
  VKR_FMT_RGBA8_SRGB     = 4,  /* 4 x uint8, rgb sRGB-encoded, a linear            */
  VKR_FMT_RGBA8_UNORM    = 5,  /* 4 x uint8                                        */
  VKR_FMT_RGBA16_UNORM   = 6,  /* 4 x uint16                                       */
  VKR_FMT_RGBA16_SFLOAT  = 7,  /* 4 x fp16                                         */
  VKR_FMT_R16_SFLOAT     = 8,  /* 1 x fp16                                         */
  VKR_FMT_R32_SFLOAT     = 9,  /* 1 x fp32                                         */
  VKR_FMT_R8_UNORM       = 10, /* 1 x uint8 (create_gtao_texture, gtao.cpp:10)     */
  VKR_FMT_RGBA32_SFLOAT  = 11  /* 4 x fp32: AdvancedSSR::tile_planes (advanced_ssr.cpp:85), allocated by the reference's
                                  constructor, bound by no program of this path       */
} vkr_format;

/* bytes per texel of a vkr_format (0 for unknown) */
uint32_t vkr_format_bytes(uint32_t format);

typedef struct vkr_img {
  void*    base;                       /* device pointer, start of view-mip 0       */
  uint32_t format;                     /* vkr_format                                */
  uint32_t mip_count;                  /* mips in this view (>=1)                   */
  uint32_t width, height;              /* extent of view-mip 0 held in memory       */
  uint32_t full_width, full_height;    /* extent of view-mip 0 of the whole frame   */
  int32_t  origin_x, origin_y;         /* window origin inside the frame, view-mip 0*/
  uint32_t pitch_bytes[VKR_MAX_MIPS];  /* row pitch of each mip: < 16 MiB, rows < 2^24, and pitch x rows of the mip < 4 GiB (the kernels
                                          address texels with 32-bit offsets; VKR_ERR_LAYOUT otherwise)  */
  uint64_t mip_offset[VKR_MAX_MIPS];   /* byte offset of each mip from `base`       */
} vkr_img;

/* 4x4 column-major float matrix, memory-compatible with glm::mat4 */
typedef struct vkr_mat4 { float m[16]; } vkr_mat4;

/* ---- UBO structs, byte-compatible with what the reference uploads ------------------ */

/* GTAOParams, gtao.hpp:12-18 / main.comp:7-13 */
typedef struct vkr_gtao_params {
  vkr_mat4 normal_mat;
  float fovy, aspect, znear, zfar;
} vkr_gtao_params;

/* push constants of gtao_compute_main, gtao.cpp:101-113 / main.comp:20-26 */
typedef struct vkr_gtao_push {
  float    angle_offset;
  float    weight_ratio;
  uint32_t use_mis;
  uint32_t two_directions;
  uint32_t reflections_only;
} vkr_gtao_push;

/* push constants of gtao_filter, gtao.cpp:210-215 / filter.comp:12-15 */
typedef struct vkr_gtao_filter_push { float znear, zfar; } vkr_gtao_filter_push;

/* AccumConstants, gtao.cpp:300-305 / accum.comp:16-21 */
typedef struct vkr_gtao_accum_params {
  vkr_mat4 inverse_camera;
  vkr_mat4 prev_inverse_camera;
  vkr_mat4 mvp;
  float    fovy_aspect_znear_zfar[4];
} vkr_gtao_accum_params;

typedef struct vkr_gtao_accum_push { uint32_t clear_history; } vkr_gtao_accum_push;

/* TraceParams, advanced_ssr.cpp:138-145 / trace.comp:9-16 (std140: mat4, uint, 4 floats) */
typedef struct vkr_trace_params {
  vkr_mat4 normal_mat;
  uint32_t frame_random;
  float fovy, aspect, znear, zfar;
} vkr_trace_params;

typedef struct vkr_trace_push  { float max_roughness; } vkr_trace_push;     /* trace.comp:24-26  */
typedef struct vkr_filter_push { uint32_t render_flags; } vkr_filter_push;  /* filter.comp:28-30 */
/* multi-GPU trace: `normal` holds frame rows [normal_row0, normal_row1) only when the march ends (vkr_sssr_trace_windowed) */
typedef struct vkr_trace_window_push { float max_roughness; uint32_t normal_row0, normal_row1; } vkr_trace_window_push;
#define VKR_NORMALIZE_REFLECTIONS  1u   /* filter.comp:22-24 */
#define VKR_ACCUMULATE_REFLECTIONS 2u
#define VKR_BILATERAL_FILTER       4u

/* blur.comp:16-20 / advanced_ssr.cpp:388-392 */
typedef struct vkr_blur_push {
  float    max_roughness;
  uint32_t accumulate;
  uint32_t disable_blur;
} vkr_blur_push;

/* ReprojectConsts (blur.comp:22-26) == TAAUniforms (resolve.comp:11-15, taa.cpp:24-28) */
typedef struct vkr_reproject_params {
  vkr_mat4 inverse_camera;
  vkr_mat4 prev_inverse_camera;
  float    fovy_aspect_znear_zfar[4];
} vkr_reproject_params;

/* SSRParams of the simple SSR pass, ssr.hpp:8-15 / ssr/shader.frag:9-16 */
typedef struct vkr_ssr_params {
  vkr_mat4 normal_mat;
  float fovy, aspect, znear, zfar;
} vkr_ssr_params;

/* ShaderConstants of the deferred-shading composite, defered_shading.cpp:4-12 / shader.frag:15-23 */
typedef struct vkr_shading_params {
  vkr_mat4 inverse_camera;
  vkr_mat4 camera;
  vkr_mat4 shadow_mvp;
  float fovy, aspect, znear, zfar;
} vkr_shading_params;

/* push constants, defered_shading.cpp:68-72 / shader.frag:30-33 (vec2 + uint) */
typedef struct vkr_shading_push {
  float    min_max_roughness[2];
  uint32_t show_ao;
} vkr_shading_push;

/* ---- tile-classified trace -------------------------------------------------- */
/* classification.comp:26-30 (advanced_ssr.cpp:463-471)                                             */
typedef struct vkr_classification_push { int32_t width, height; float max_roughness, glossy_value; } vkr_classification_push;
/* trace_indirect.comp:25-28 (advanced_ssr.cpp:231-236)                                             */
typedef struct vkr_trace_indirect_push { uint32_t reflection_type; float max_roughness; } vkr_trace_indirect_push;

/* ---- dormant GTAO variants ------------------------------------------- */
/* gtao/main.frag:17-19 push constants (gtao.cpp:355-357)                                           */
typedef struct vkr_gtao_gfx_push { float angle_offset; } vkr_gtao_gfx_push;
/* gtao/reproject.comp:11-17 == GTAOReprojection (gtao.hpp:28-34)                                   */
typedef struct vkr_gtao_reprojection {
  vkr_mat4 camera_to_prev_frame;
  float fovy, aspect, znear, zfar;
} vkr_gtao_reprojection;
/* gtao_opt/deinterleave.comp:6-8 (gtao.cpp:464)                                                    */
typedef struct vkr_deinterleave_push { int32_t pattern_step; } vkr_deinterleave_push;
/* gtao_opt/main_deinterleaved.comp:17-21 (gtao.cpp:474-478)                                        */
typedef struct vkr_gtao_deinterleaved_push { int32_t pattern_n; uint32_t layer; float angle_offset; } vkr_gtao_deinterleaved_push;

/* ---- ScreenSpaceTrace ------------------------------------------------- */
/* screen_trace/trace.comp:14-22 == GpuParams (screen_trace.cpp:30-38)                              */
typedef struct vkr_screen_trace_params {
  vkr_mat4 normal_mat;
  float random_offset, angle_offset, fovy, aspect, znear, zfar;
} vkr_screen_trace_params;
/* screen_trace/filter.comp:8-11 (screen_trace.cpp:103-106)                                         */
typedef struct vkr_screen_trace_filter_push { float znear, zfar; } vkr_screen_trace_filter_push;
/* screen_trace/accumulate.comp:7-12 (screen_trace.cpp:148-153)                                     */
typedef struct vkr_screen_trace_accum_push { float fovy, aspect, znear, zfar; } vkr_screen_trace_accum_push;

/* ---- G-buffer raster stage --------------------------------------------------- */
/* scene::Vertex, scene/scene.hpp:15-19                                                              */
typedef struct vkr_raster_vertex { float pos[3], norm[3], uv[2]; } vkr_raster_vertex;
/* Transform, gbuf/opaque_taa.vert:15-18                                                             */
typedef struct vkr_raster_transform { vkr_mat4 model, normal; } vkr_raster_transform;
/* one draw_indexed of scene_renderer.cpp:196-214 with its push constants (:132-137)                 */
typedef struct vkr_raster_draw {
  uint32_t transform_index, albedo_index, mr_index, flags;   /* PushData; 0xFFFFFFFF = no texture    */
  uint32_t index_offset, index_count, vertex_offset, reserved;  /* reserved: VKR_RASTER_DRAW_* hints */
} vkr_raster_draw;
/* opaque_taa.frag:32-34 discards fragments whose filtered albedo alpha is 0 (no depth, no colour written); the stage
 * evaluates that for every draw with an albedo texture.  Hint: the caller guarantees that no texel of any mip level of
 * the draw's albedo texture has alpha 0, so the discard cannot fire and the test is skipped (same result, cheaper). */
#define VKR_RASTER_DRAW_OPAQUE_ALBEDO 1u
/* GbufConst, scene_renderer.cpp:148-153 / opaque_taa.vert:7-12                                      */
typedef struct vkr_gbuf_const {
  vkr_mat4 view_projection, prev_view_projection;
  float    jitter[4];
  float    fovy_aspect_znear_zfar[4];
} vkr_gbuf_const;
/* vertices / indices / transforms: device memory; draws and textures: host arrays (launch params).
 * Textures are RGBA8_SRGB images with full mip chains (scene/images.cpp:32-49), sampled with the
 * scene sampler (scene_renderer.cpp:77-81: bilinear, linear mip, REPEAT).                          */
typedef struct vkr_raster_scene {
  const vkr_raster_vertex*    vertices;   uint32_t vertex_count;
  const uint32_t*             indices;    uint32_t index_count;
  const vkr_raster_transform* transforms; uint32_t transform_count;
  const vkr_raster_draw*      draws;      uint32_t draw_count;
  const vkr_img*              textures;   uint32_t texture_count;
} vkr_raster_scene;

/* Parameters of the synthetic G-buffer generator (replaces the raster stage
 * scene_renderer.cpp:140-220 + gbuf/opaque_taa.{vert,frag}). */
typedef struct vkr_synth_params {
  vkr_mat4 camera_to_world;   /* inverse view matrix of the frame being generated    */
  vkr_mat4 prev_mvp;          /* projection * previous view (velocity, opaque_taa.vert)*/
  vkr_mat4 mvp;               /* projection * current view                           */
  float    fovy, aspect, znear, zfar;
  uint32_t seed;              /* PCG32 stream for the checker / per-object material  */
  uint32_t flags;             /* VKR_SYNTH_*                                          */
} vkr_synth_params;
#define VKR_SYNTH_DEPTH_ONLY 1u          /* depth attachment only (used for prev_depth)                                        */
/* Second material mode: the roughness of an object is perturbed PER TEXEL (+-0.15, PCG hash of the pixel), like a roughness
 * texture: the blur's sigma (blur.comp:44-75) and the trace's lobe then vary inside every wavefront.  The default scene has
 * one roughness per object, which the SSR blur's wave-uniform-sigma path and empty-tile skips exploit.                    */
#define VKR_SYNTH_TEXTURED_ROUGHNESS 2u

/* ---- entry points ------------------------------------------------------------------- */

const char* vkr_version(void);
const char* vkr_last_error(void);
/* Version of the numeric contract the library was built with (csrc/vkr_device.hpp): 2 = the accumulation steps of the
 * shared helpers (dot, mix, mat4*vec4, cross, madd, texel coordinates ...) are fused multiply-adds, which GLSL without
 * `precise` allows every Vulkan driver to do; 1 = nothing fused (`make CONTRACT=1`).  Results of the two contracts
 * differ in the last bits; a checker must be built with the same one.                                              */
uint32_t vkr_numeric_contract(void);

/* program "downsample_gbuffer": downsample_pass.cpp:25-92 + downsample_gbuffer.frag:12-37.
 * depth: full image (view mip 0 = image mip 0, >=2 mips); writes depth mip 1.           */
int vkr_downsample_gbuffer(const vkr_img* depth, const vkr_img* normal, const vkr_img* velocity,
                           const vkr_img* out_normal, const vkr_img* out_velocity, void* stream);

/* program "depth_mips": downsample_pass.cpp:94-131 + depth_mips.frag:7-15.
 * Builds mips src_mip+1 .. mip_count-1, each the 2x2 min of its parent.                 */
int vkr_depth_mips(const vkr_img* depth, uint32_t src_mip, void* stream);

/* program "pdf_preintegrate": advanced_ssr.cpp:95-114 + preintegrate.comp:45-84          */
int vkr_pdf_preintegrate(const vkr_img* out_pdf, void* stream);

/* Fills the HaltonBuffer UBO of sssr_trace (host memory, `count` x vec4): xy = Halton(2,3) of index
 * i+1 exactly as advanced_ssr.cpp:8-34 builds it; zw — unused (0) in the reference — carry
 * (float)cos((double)phi), (float)sin((double)phi) with phi = (2*PI)*y, the two transcendentals
 * sampleGGXVNDF (brdf.glsl:146-148) needs per ray: 128 values evaluated once on the host instead
 * of per pixel on the device.  vkr_sssr_trace requires a buffer filled by this function.          */
void vkr_halton23_fill(float* host_vec4, uint32_t count);

/* program "sssr_trace": advanced_ssr.cpp:147-214 + trace.comp (bindings 0..7)            */
int vkr_sssr_trace(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                   const vkr_trace_params* params, const float* halton_vec4 /*device, 128 x vec4*/,
                   const vkr_img* out_ray, const vkr_img* out_occlusion, const vkr_img* pdf_tex,
                   const vkr_trace_push* push, void* stream);

/* "sssr_trace" in two launches, same images as vkr_sssr_trace bit for bit.  The head launch runs every ray's prologue, its 16
 * pinned steps and `park_after_rounds` (0..4) of the compacted 16-step rounds, and finishes the pixels whose rays have ended;
 * a ray that has not is PARKED: written, with what the rest of its march and its epilogue need (80 bytes), to a frame-wide
 * queue in `workspace`.  The resume launch gives every parked ray a lane of its own — the stragglers of many tiles side by side
 * in full waves — marches them to their end and writes their pixels.  workspace: device memory of at least
 * vkr_sssr_trace_workspace_bytes(rays width, rays height) bytes, 16-byte aligned, no initialisation needed; a workspace
 * belongs to one stream at a time.                                                                                      */
uint64_t vkr_sssr_trace_workspace_bytes(uint32_t rays_width, uint32_t rays_height);
int vkr_sssr_trace_split(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                         const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                         const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_trace_push* push,
                         void* workspace, uint64_t workspace_bytes, uint32_t park_after_rounds, void* stream);

/* Multi-GPU variant of "sssr_trace" (no reference counterpart; host/frame.hpp): `normal` is the whole-frame image but only
 * frame rows [normal_row0, normal_row1) of it are in memory when the march ends (the rank's window).  A ray that passes
 * every other validity test and whose hit-normal footprint (trace.comp:103-109) has a row outside them is stored as a
 * provisional hit: pending_mask (R8, the rays' extent; written for every pixel: 1 = pending) marks it and pending_data
 * (RGBA32F, twice the rays' width: two texels per pixel) keeps R and the hit uv.  Once the missing rows have arrived
 * (vkr_hit_requests / _reply / _scatter below) vkr_sssr_validate runs the deferred test and turns the rays that fail it
 * into misses — the stored rays are then bit-identical to vkr_sssr_trace's on a complete `normal`.                    */
int vkr_sssr_trace_windowed(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                            const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                            const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                            const vkr_img* pending_data, const vkr_trace_window_push* push, void* stream);
/* vkr_sssr_trace_windowed in two launches AROUND the arrival of the whole-frame pyramid (the depth all-gather of the tiled frame):
 * the head launch marches on `local_depth` alone — the levels of the pyramid this rank built itself, as the window image
 * holds them: full-width strips of the frame's levels, rows [origin_y, origin_y + height) each — and parks a ray in
 * `workspace` (vkr_sssr_trace_split above: same queue, same records) at its first fetch of a texel of the frame that is not
 * there: a row outside the strip, or a level the window image does not have.  A ray whose march has ended is parked as well
 * when its hit-depth sample (trace.comp:111-117) needs such a row; rays still marching after park_after_rounds compacted
 * rounds are parked as in the split.  `frame_depth` gives the head launch the frame's extents and level count only (not
 * read); once it is complete the resume launch finishes every parked ray on it.  Both launches take the arguments of
 * vkr_sssr_trace_windowed and leave, together, exactly its images.                                                      */
int vkr_sssr_trace_windowed_head(const vkr_img* local_depth, const vkr_img* frame_depth, const vkr_img* normal, const vkr_img* material,
                                 const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                 const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                                 const vkr_img* pending_data, const vkr_trace_window_push* push, void* workspace,
                                 uint64_t workspace_bytes, uint32_t park_after_rounds, void* stream);
int vkr_sssr_trace_windowed_resume(const vkr_img* frame_depth, const vkr_img* normal, const vkr_img* material,
                                   const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                   const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                                   const vkr_img* pending_data, const vkr_trace_window_push* push, void* workspace,
                                   uint64_t workspace_bytes, void* stream);
int vkr_sssr_validate(const vkr_img* rays, const vkr_img* pending_mask, const vkr_img* pending_data, const vkr_img* frame_normals,
                      const vkr_trace_params* params, void* stream);
/* the same, unless *skip_if_set (a device word, read by the launch) is non-zero: a round of requests that dropped some (see
 * vkr_hit_requests_bounded) has not delivered every hit normal, and the test must wait for the exact round               */
int vkr_sssr_validate_unless(const vkr_img* rays, const vkr_img* pending_mask, const vkr_img* pending_data, const vkr_img* frame_normals,
                             const vkr_trace_params* params, const uint32_t* skip_if_set, void* stream);

/* program "sssr_filter": advanced_ssr.cpp:308-369 + filter.comp (bindings 0..6)          */
int vkr_sssr_filter(const vkr_img* rays, const vkr_img* depth, const vkr_img* albedo,
                    const vkr_img* normal, const vkr_img* material, const vkr_img* out_reflections,
                    const vkr_trace_params* params, const vkr_filter_push* push, void* stream);

/* program "sssr_blur": advanced_ssr.cpp:371-438 + blur.comp (bindings 0..8)              */
int vkr_sssr_blur(const vkr_img* depth, const vkr_img* normal, const vkr_img* reflections,
                  const vkr_img* material, const vkr_img* history, const vkr_img* velocity,
                  const vkr_img* history_depth, const vkr_img* out_blurred,
                  const vkr_reproject_params* params, const vkr_blur_push* push, void* stream);


/* program "gtao_compute_main": gtao.cpp:84-148 + gtao/main.comp (bindings 0..5)          */
int vkr_gtao_main(const vkr_img* depth, const vkr_gtao_params* params, const vkr_img* normal,
                  const vkr_img* material, const vkr_img* pdf_tex, const vkr_img* gtao_inout,
                  const vkr_gtao_push* push, void* stream);

/* program "gtao_filter": gtao.cpp:198-239 + gtao/filter.comp (bindings 0..2)             */
int vkr_gtao_filter(const vkr_img* depth, const vkr_img* raw_gtao, const vkr_img* out_filtered,
                    const vkr_gtao_filter_push* push, void* stream);

/* program "gtao_accumulate": gtao.cpp:286-347 + gtao/accum.comp (bindings 0..6)          */
int vkr_gtao_accumulate(const vkr_img* depth, const vkr_img* prev_depth, const vkr_img* current_ao,
                        const vkr_img* out_accumulated, const vkr_img* velocity, const vkr_img* history,
                        const vkr_gtao_accum_params* params, const vkr_gtao_accum_push* push,
                        void* stream);

/* program "taa_resolve": taa.cpp:19-63 + taa/resolve.comp (bindings 0..6)                */
int vkr_taa_resolve(const vkr_img* history_color, const vkr_img* history_depth,
                    const vkr_img* current_depth, const vkr_img* velocity, const vkr_img* color,
                    const vkr_img* out_color, const vkr_reproject_params* params, void* stream);

/* program "ssr": ssr.cpp:10-73 + ssr/shader.frag:27-102 (bindings 0 normal, 1 depth [NEAREST,
 * clamp-to-border U], 2 frame colour, 3 SSRParams, 4 material; colour attachment RGBA8_UNORM).
 * depth: all mips of the full-res depth image.                                                  */
int vkr_ssr(const vkr_img* normal, const vkr_img* depth, const vkr_img* frame, const vkr_ssr_params* params,
            const vkr_img* material, const vkr_img* out, void* stream);

/* program "brdf_preintegrate": advanced_ssr.cpp:116-136 + preintegrate_ssr.comp:12-44 (split-sum LUT,
 * RG16F; consumed by the deferred-shading composite).  halton: the HaltonBuffer of vkr_sssr_trace.   */
int vkr_brdf_preintegrate(const float* halton_vec4 /*device*/, const vkr_img* out_brdf, void* stream);

/* program "defered_shading": defered_shading.cpp:47-118 + defered_shading/shader.frag:41-130
 * (bindings 0 albedo, 1 normal, 2 material, 3 depth [all mips], 4 Constants, 5 shadow map — bound by
 * the reference but never sampled, omitted here —, 6 occlusion, 7 brdf LUT, 8 reflections; colour
 * attachment RGBA8_SRGB).  SURVEY.md 8(f) #1.                                                        */
int vkr_defered_shading(const vkr_img* albedo, const vkr_img* normal, const vkr_img* material, const vkr_img* depth,
                        const vkr_shading_params* consts, const vkr_img* occlusion, const vkr_img* brdf,
                        const vkr_img* reflections, const vkr_img* out, const vkr_shading_push* push, void* stream);

/* SSSR_Clear (advanced_ssr.cpp:440-452): VkDispatchIndirectCommand{0,1,1} into both 3-word device
 * argument buffers.                                                                                */
int vkr_sssr_clear_indirect(uint32_t* reflective_args, uint32_t* glossy_args, void* stream);

/* program "sssr_classification": advanced_ssr.cpp:454-495 + classification.comp:38-98.  Appends
 * every 8x8 tile of the half-res extent (push->width/height = extent of `rays`) to the reflective
 * list when its mean roughness is below push->glossy_value, else to the glossy list; the tile
 * counts accumulate in word 0 of the argument buffers.  List order is unspecified (atomics), as in
 * the reference.  Bindings 0 material, 1/2 tile lists, 3/4 indirect arguments.                      */
int vkr_sssr_classification(const vkr_img* material, int32_t* reflective_tiles, int32_t* glossy_tiles,
                            uint32_t* reflective_args, uint32_t* glossy_args,
                            const vkr_classification_push* push, void* stream);

/* program "sssr_trace_indirect": advanced_ssr.cpp:216-302 + trace_indirect.comp:43-135, one
 * dispatch_indirect (push->reflection_type 0 = mirror: march from mip 0, <= 50 steps; 1 = glossy:
 * from mip 1, <= 25 steps).  Bindings 0 depth (view of mips 1..L-1), 1 normal, 2 material,
 * 3 TraceParams, 4 Halton, 5 rays, 6 tile list.  `indirect_args` word 0 is read on the device;
 * `max_tiles` = capacity of the tile list (launch bound).                                          */
int vkr_sssr_trace_indirect(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                            const vkr_trace_params* params, const float* halton_vec4 /*device*/, const vkr_img* out_rays,
                            const int32_t* tiles /*device*/, const uint32_t* indirect_args /*device*/, uint32_t max_tiles,
                            const vkr_trace_indirect_push* push, void* stream);

/* program "gtao_main" (graphics): gtao.cpp:349-413 + gtao/main.frag:45-48,164-196 — full-screen
 * triangle into `raw`; one slice, 20 samples, radius min(200/|P|, 32) px, sky -> 1.  Bindings 0 depth
 * (view mip depth_lod), 1 GTAOParams, 2 normal; colour attachment RGBA16F (only .r is written by the
 * shader; the unwritten components are frozen to 0).  The debug heat-map of trace_samples.glsl
 * (binding 7, compiled out on the host side by GTAO_TRACE_SAMPLES 0) is not reproduced.            */
int vkr_gtao_main_graphics(const vkr_img* depth, const vkr_gtao_params* params, const vkr_img* normal,
                           const vkr_img* out_raw, const vkr_gtao_gfx_push* push, void* stream);

/* program "gtao_reproject": gtao.cpp:241-284 + gtao/reproject.comp:27-66 (STATIC_REPROJECT mode:
 * same-pixel history, mix 0.05 when |z_prev - z_cur| < 1e-6).  Bindings 0 params, 1 depth, 2 prev
 * depth, 3 current ao (filtered), 4 previous ao, 5 out (R16F).  Floor dispatch (w/8, h/4).           */
int vkr_gtao_reproject(const vkr_gtao_reprojection* params, const vkr_img* depth, const vkr_img* prev_depth,
                       const vkr_img* current_ao, const vkr_img* prev_ao, const vkr_img* out, void* stream);

/* program "deinterleave_depth": gtao.cpp:445-470 + gtao_opt/deinterleave.comp:10-21.  `layers` is the
 * R32F array image, one descriptor per layer (layer = ((y & m) << step) + (x & m)).  The dispatch is
 * (layer_w/8, layer_h/4) groups exactly as the reference records it (gtao.cpp:468), so only source
 * texels below that extent are scattered.                                                          */
int vkr_deinterleave_depth(const vkr_img* depth, const vkr_img* layers, uint32_t layer_count,
                           const vkr_deinterleave_push* push, void* stream);

/* program "main_deinterleaved": gtao.cpp:472-526 + gtao_opt/main_deinterleaved.comp:38-124.  One
 * dispatch of (out_w/8, out_h/4) groups for push->layer; stores outside `out` are dropped.          */
int vkr_gtao_main_deinterleaved(const vkr_img* layers, uint32_t layer_count, const vkr_gtao_params* params,
                                const vkr_img* normal, const vkr_img* out_raw,
                                const vkr_gtao_deinterleaved_push* push, void* stream);

/* program "screen_trace_main": screen_trace.cpp:23-95 + screen_trace/trace.comp:27-37,230-343
 * (trace_tangent_space, 1 direction).  Bindings 0 depth (mip 0), 1 normal, 2 colour, 3 material,
 * 4 out RGBA16F, 5 Params.  Floor dispatch (w/8, h/8).  The 8x8 tile exchange through shared memory
 * is resolved as if `barrier()` followed the stores (trace.comp:310-311 has only a memory barrier);
 * slots of sky pixels, which return before writing theirs, read as "no hit".                       */
int vkr_screen_trace_main(const vkr_img* depth, const vkr_img* normal, const vkr_img* color, const vkr_img* material,
                          const vkr_img* out_raw, const vkr_screen_trace_params* params, void* stream);

/* program "screen_trace_filter": screen_trace.cpp:97-140 + screen_trace/filter.comp:13-39.           */
int vkr_screen_trace_filter(const vkr_img* raw, const vkr_img* depth, const vkr_img* out_filtered,
                            const vkr_screen_trace_filter_push* push, void* stream);

/* program "screen_trace_accumulate": screen_trace.cpp:142-181 + screen_trace/accumulate.comp:21-40
 * (accum is read and written in place).                                                            */
int vkr_screen_trace_accumulate(const vkr_img* depth, const vkr_img* prev_depth, const vkr_img* current,
                                const vkr_img* accum_inout, const vkr_screen_trace_accum_push* push, void* stream);

/* program "gbuf_opaque_taa": SceneRenderer::draw_taa (scene_renderer.cpp:140-220) + gbuf/opaque_taa.{vert,frag}
 * as a compute rasterizer: attachments cleared (colour 0, depth 1), every triangle of every draw
 * rasterised into a 64-bit visibility buffer (depth | triangle) with atomics, then resolved per pixel
 * (perspective-correct attributes, trilinear sRGB textures, octahedral normal, velocity).  Frozen
 * raster rules: pixel centres, top-left fill rule, 8 sub-pixel bits, cull none, depth LESS_OR_EQUAL
 * in D24 (gpu/pipelines.hpp:113-128), near-plane clipping in clip space, implicit LOD from forward
 * differences of the interpolated uv.  `scratch`: vkr_raster_scratch_bytes() of device memory (the
 * visibility buffer + two records per drawn triangle).                           */
uint64_t vkr_raster_scratch_bytes(uint32_t width, uint32_t height, uint32_t triangle_count /* summed over all draws */);
int vkr_raster_gbuffer(const vkr_raster_scene* scene, const vkr_gbuf_const* consts, const vkr_img* albedo,
                       const vkr_img* normal, const vkr_img* material, const vkr_img* velocity, const vkr_img* depth,
                       void* scratch, uint64_t scratch_bytes, void* stream);

/* synthetic G-buffer generator (no reference program).  Any of the
 * colour outputs may be NULL when VKR_SYNTH_DEPTH_ONLY is set.                          */
int vkr_synth_gbuffer(const vkr_img* depth, const vkr_img* normal, const vkr_img* albedo,
                      const vkr_img* material, const vkr_img* velocity,
                      const vkr_synth_params* params, void* stream);

/* Multi-GPU exchange helper (the reference is single-GPU, so there is no program this
 * replaces): copies `count` pitch-linear byte rectangles on `stream`, VKR_MAX_RECTS per launch.  Used to pack
 * a tile's surfaces for the all-gather, to scatter the gathered tiles into the whole-frame images and to
 * pack / unpack the halo rings of the history surfaces.  Addresses are device addresses; addresses, pitches
 * and row lengths must be multiples of 4 bytes (16 selects the wide path).  `rects` is host memory.       */
#define VKR_MAX_RECTS 64
typedef struct vkr_rect_copy {
  uint64_t src, dst;
  uint32_t src_pitch, dst_pitch; /* bytes between rows */
  uint32_t row_bytes, rows;
} vkr_rect_copy;
int vkr_copy_rects(const vkr_rect_copy* rects, uint32_t count, void* stream);

/* ---- the wire of the multi-GPU frame: RCCL over xGMI (SURVEY.md 8(b) "vkr_halo_exchange", 8(e)) -------------
 * One process per GPU.  Rank 0 makes an id (vkr_comm_unique_id) and hands its bytes to every rank out of band (a
 * file, a TCP store, MPI ...); every rank then calls vkr_comm_create, collectively, with its device current.  RCCL is
 * loaded with dlopen on first use (librccl.so.1, or $VKR_RCCL_LIBRARY): the library does not link against it.
 * Both exchanges are ONE grouped RCCL launch on `stream` and return at once; buffers are device memory.          */
#define VKR_COMM_ID_BYTES 128
typedef struct vkr_comm vkr_comm;
int vkr_comm_unique_id(uint8_t* id_bytes /* [VKR_COMM_ID_BYTES] */);
int vkr_comm_create(const uint8_t* id_bytes, int rank, int world, vkr_comm** out);
int vkr_comm_destroy(vkr_comm* comm);
int vkr_comm_rank(const vkr_comm* comm, int* rank, int* world);
/* 0 when RCCL can be loaded in this process (nothing collective happens); a launcher checks this on EVERY rank and
 * agrees on the result over its control plane before the collective vkr_comm_create, so that a rank without RCCL
 * cannot leave the others blocked inside ncclCommInitRank                                                          */
int vkr_comm_available(void);
/* A communicator that moves NOTHING and takes the time a fully connected xGMI node would (measurement facility, no RCCL
 * needed): every exchange enqueues, on the caller's stream, a one-wave kernel that holds the stream for
 * launch_us + (bytes on the busiest link) / link_gbps — each peer on a link of its own, full duplex: an all-gather costs
 * the largest share any peer sends, a point-to-point group the largest per-peer total — and copies the rank's own share of
 * an out-of-place all-gather into its slot.  What the peers would have sent is whatever the receive buffers already hold:
 * with a static scene and buffers filled once by real peers (the in-process lockstep harness) every frame receives what it
 * would have received, so the frame's stream / event schedule runs against realistic wire times on ONE GPU
 * (tools/wire_emulation.py, vkrh_tiled_emulate_wire).                                                                   */
int vkr_comm_create_emulated(int rank, int world, float link_gbps, float launch_us, vkr_comm** out);
/* Start-up check of a fresh communicator, collective: every rank gathers a rank-tagged pattern through vkr_all_gather
 * and vkr_all_gather_v (shares of different sizes, in place) and trades one with each neighbour rank through
 * vkr_halo_exchange, on `stream`, then verifies every byte it received (synchronises the stream).  0 = the wire delivers
 * what the tiled frame expects.  scratch: device memory of at least vkr_comm_selfcheck_bytes(world) bytes.          */
uint64_t vkr_comm_selfcheck_bytes(int world);
int vkr_comm_selfcheck(vkr_comm* comm, void* scratch, void* stream);
/* all-gather of several surfaces at once: recv receives [rank][bytes] from every rank's send (out of place; with
 * horizontal strips a tile's rows of a surface are contiguous, so recv can be the whole-frame image itself)       */
typedef struct vkr_gather_part { const void* send; void* recv; uint64_t bytes; } vkr_gather_part;
int vkr_all_gather(vkr_comm* comm, const vkr_gather_part* parts, uint32_t count, void* stream);
/* the same for shares of different sizes (strips balanced by cost): rank r's share of a surface lies at
 * recv + offsets[r] .. recv + offsets[r + 1] (offsets: world + 1 entries in host memory, identical on every rank; an
 * empty share is allowed); `send` is this rank's share where it lies now.  One grouped launch: every share sent straight
 * to each peer and received into place (ncclSend / ncclRecv: on a fully connected xGMI node a share crosses one link once;
 * RCCL fuses the point-to-point operations of a group), the own share copied on the stream.  VKR_GATHER_V_BROADCAST=1
 * (read at the first call) selects one ncclBroadcast per share, rooted at its owner, instead.                        */
typedef struct vkr_gather_v_part { const void* send; void* recv; const uint64_t* offsets; } vkr_gather_v_part;
int vkr_all_gather_v(vkr_comm* comm, const vkr_gather_v_part* parts, uint32_t count, void* stream);
/* halo refresh of one history surface: per neighbour the packed slice to send and the buffer to receive into
 * (either may be empty); pack / unpack with vkr_copy_rects around it                                               */
typedef struct vkr_halo_peer { int32_t peer; uint32_t reserved; const void* send; uint64_t send_bytes; void* recv; uint64_t recv_bytes; } vkr_halo_peer;
int vkr_halo_exchange(vkr_comm* comm, const vkr_halo_peer* peers, uint32_t count, void* stream);

/* ---- hit colours and hit normals by request / reply (multi-GPU; instead of all-gathering whole-frame surfaces) --------
 * filter.comp:112-134 reads the albedo bilinearly at the hit position of every valid ray, trace.comp:103-109 the downsampled
 * normal at the hit position of every candidate — anywhere in the frame.  A rank asks the owning ranks for exactly the
 * footprint rows it does not hold and writes the answers into its whole-frame images where an all-gather would have put
 * them (csrc/hit_exchange.hip; host/frame.hpp drives the steps and moves requests and replies with vkr_halo_exchange).
 * Strips: rank r owns full-res frame rows [row_bounds[r], row_bounds[r + 1]) (even numbers) and the half-res rows at half
 * of them.  A request is 4 bytes — bits 0..13 the frame row, 14..27 the left texel of the texel pair (<= width - 2),
 * VKR_HIT_BOTH_ROWS: also the row below (the two rows of a footprint usually have one owner), VKR_HIT_NORMAL: the
 * half-res normal image instead of the full-res albedo — and is answered with 16 bytes: the pair of the row and of the row
 * below it.  Frames up to 16384 x 16384.                                                                                */
typedef uint32_t vkr_hit_request;
#define VKR_HIT_BOTH_ROWS 0x10000000u
#define VKR_HIT_NORMAL    0x20000000u
typedef struct vkr_hit_sources {
  const vkr_img* rays;            /* RGBA16_UNORM, the rank's half-res window: albedo rows for every ray with w != 1           */
  uint32_t albedo_width, albedo_height;
  uint32_t window_row0, window_row1;   /* the full-res frame rows this rank holds                                             */
  const vkr_img* pending_mask;    /* or NULL: no normal requests.  R8 + RGBA32F of vkr_sssr_trace_windowed: normal rows for   */
  const vkr_img* pending_data;    /* every pending ray, outside half-res rows [normal_row0, normal_row1)                       */
  uint32_t normal_width, normal_height, normal_row0, normal_row1;
} vkr_hit_sources;
/* Two passes over the same `src` and bounds.  Pass 1 (out == NULL): counts[o] += number of requests this rank has for
 * owner o (counts: device, world entries, zeroed by the caller); with a workspace (device, VKR_HIT_WORKSPACE_WORDS uint32,
 * no initialisation needed) it also leaves there what pass 2 needs.  Pass 2 (out != NULL; workspace as pass 1 filled it):
 * writes the requests, owner o's from out[segments[o]] on (segments: host, world entries, the prefix sums of counts).    */
#define VKR_HIT_WORKSPACE_WORDS 4096u
int vkr_hit_requests(const vkr_hit_sources* src, const uint32_t* row_bounds, uint32_t world, uint32_t* counts, uint32_t* workspace,
                     const uint32_t* segments, vkr_hit_request* out, void* stream);
/* Pass 2 into segments of FIXED capacity (sized from the previous frame's counts, so that both ends of the exchange know the
 * message sizes without waiting for this frame's): owner o's requests from out[segments[o]] on, at most capacities[o] of them —
 * what does not fit is dropped (the caller compares this frame's counts with the capacities afterwards and repeats the round
 * exactly if any segment overflowed).  Slots a segment does not use must hold VKR_HIT_NO_REQUEST (fill `out` with 0xFF bytes
 * first): vkr_hit_reply answers such a slot with zeros and counts no error, vkr_hit_scatter skips it.                      */
#define VKR_HIT_NO_REQUEST 0xFFFFFFFFu
int vkr_hit_requests_bounded(const vkr_hit_sources* src, const uint32_t* row_bounds, uint32_t world, uint32_t* workspace,
                             const uint32_t* segments, const uint32_t* capacities, vkr_hit_request* out,
                             uint32_t* dropped_flag /* device word, zeroed by the caller: set to 1 when a request was dropped */, void* stream);
/* replies (device, 16 bytes each, 16-byte aligned): the texel pair(s) of requests[i] from this rank's albedo / downsampled-
 * normal window (normals may be NULL when no normal request can arrive); a request for texels the window does not hold
 * answers 0, increments error_counter[0] and leaves one such request and its index in error_counter[1], [2] (3 words)                                                                              */
#define VKR_HIT_REPLY_BYTES 16u
int vkr_hit_reply(const vkr_img* albedo, const vkr_img* normals, const vkr_hit_request* requests, uint32_t count, void* replies,
                  uint32_t* error_counter, void* stream);
int vkr_hit_scatter(const vkr_img* frame_albedo, const vkr_img* frame_normals, const vkr_hit_request* requests, const void* replies,
                    uint32_t count, void* stream);

/* Measurement switches — the library's only process-wide state.  The environment (VKR_BLUR_NO_SKIP, VKR_FILTER_NO_SKIP,
 * VKR_TAA_GENERIC, VKR_SHADING_GENERIC, VKR_BLUR_GENERIC, VKR_BLUR_LANE_LOOPS, VKR_TRACE_ONE_LAUNCH) is read once, at the first launch that asks; afterwards only vkr_set_switches
 * changes them.  NO_SKIP: evaluate every tap of the blur / filter even in tiles without a reflection / hit (a
 * content-independent time; the stored texels are the same wherever every weight is finite).  GENERIC: the TAA /
 * shading instantiations that do not assume equal window layouts.                                                  */
#define VKR_SWITCH_BLUR_NO_SKIP    1u
#define VKR_SWITCH_FILTER_NO_SKIP  2u
#define VKR_SWITCH_TAA_GENERIC     4u
#define VKR_SWITCH_SHADING_GENERIC 8u
#define VKR_SWITCH_BLUR_GENERIC    16u /* every wave of the blur on the per-lane Gaussian loop (no wave-uniform-sigma path) */
#define VKR_SWITCH_BLUR_LANE_LOOPS 64u /* waves that do not share one sigma on the per-lane tap loops of rounds 1-3 instead of the transposed packed rows (blur_rows) */
#define VKR_SWITCH_TRACE_ONE_LAUNCH 32u /* read by the HOST layer (host/gpu): program "sssr_trace" as one launch (vkr_sssr_trace) instead of
                                          * head + resume (vkr_sssr_trace_split); the library's entries do what their names say either way */
uint32_t vkr_get_switches(void);
void vkr_set_switches(uint32_t mask);

/* float4 streaming-read microbenchmark: the measured-roofline denominator of
 * SURVEY.md 8(d).  Reads `bytes` from `src`, writes one float per block to `sink`.      */
int vkr_stream_read(const void* src, uint64_t bytes, float* sink, uint32_t sink_len, void* stream);

/* Test hook: counts (into 5 device uint32, zeroed by the caller) where the kernels' cheap exact
 * arithmetic — normal-range division, UNORM decodes — disagrees with its IEEE definition.        */
int vkr_selftest_division(uint32_t* device_counters5, float znear, float zfar, void* stream);
/* Test hook: counts (into one device uint32, zeroed by the caller) the pixel centres g < size, size = 1..max_size (<= 65535),
 * whose uv = (g + 0.5) / size differs between the kernels' normal-range division and IEEE '/'.     */
int vkr_selftest_pixel_uv(uint32_t* device_counter, uint32_t max_size, void* stream);
/* Test hook: [0] counts the floats x in [2^-96, FLT_MAX] — all of them — whose cheap exact square root (vkr_device.hpp
 * sqrt_ieee) differs from sqrtf(x); [1] the floats s in [2^-48, 2^64] whose cheap exact reciprocal differs from 1.0f / s
 * (two device uint32, zeroed by the caller)                                                                       */
int vkr_selftest_sqrt(uint32_t* device_counters2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VKR_POSTFX_H_INCLUDED */
