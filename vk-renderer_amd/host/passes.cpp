// passes.cpp — the pass structs of passes.hpp.  Each method packs the constants the matching reference pass
// uploads (file:line given per method), lists its bindings in the shader's binding order and hands them to
// pass_recorder.hpp; the bound program (gpu/gpu.cpp program table) turns that into one C-ABI call.
#include "passes.hpp"
#include "imgui_pass.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "pass_recorder.hpp"

using rendergraph::ImageResourceId;
using rendergraph::RenderGraph;

namespace {

constexpr VkImageAspectFlags DEPTH = VK_IMAGE_ASPECT_DEPTH_BIT;
constexpr VkImageAspectFlags COLOR = VK_IMAGE_ASPECT_COLOR_BIT;

ImageResourceId make_image(RenderGraph &graph, VkFormat format, uint32_t w, uint32_t h, VkImageUsageFlags usage,
                           VkImageAspectFlags aspect = COLOR, uint32_t mips = 1, uint32_t layers = 1)
{
  return graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo {format, aspect, w, h, 1, mips, layers}, VK_IMAGE_TILING_OPTIMAL, usage);
}

VkSampler default_sampler() { return gpu::create_sampler(gpu::DEFAULT_SAMPLER); }

gpu::GraphicsPipeline fullscreen_pipeline(const char *program, const gpu::Registers &regs = {}) {
  auto p = gpu::create_graphics_pipeline();
  p.set_program(program);
  p.set_registers(regs);
  p.set_vertex_input({});
  return p;
}

void copy_mat(vkr_mat4 &dst, const glm::mat4 &src) { std::memcpy(dst.m, &src, sizeof(dst.m)); }

// {inverse(camera), inverse(prev_camera), fovy_aspect_znear_zfar}: TAAParams (taa.cpp:24-30), the blur's
// Params (advanced_ssr.cpp:388-401)
vkr_reproject_params reproject_params(const DrawTAAParams &p) {
  vkr_reproject_params r;
  copy_mat(r.inverse_camera, glm::inverse(p.camera));
  copy_mat(r.prev_inverse_camera, glm::inverse(p.prev_camera));
  std::memcpy(r.fovy_aspect_znear_zfar, &p.fovy_aspect_znear_zfar, sizeof(r.fovy_aspect_znear_zfar));
  return r;
}

constexpr uint32_t HALTON_SEQ_SIZE = 128;
enum : uint32_t { NORMALIZE_REFLECTIONS = 1, ACCUMULATE_REFLECTIONS = 2, BILATERAL_FILTER = 4 };  // advanced_ssr.cpp:304-306

}  // namespace

// ==== Gbuffer (scene_renderer.cpp:8-44) ================================================================
Gbuffer::Gbuffer(RenderGraph &graph, uint32_t width, uint32_t height) : w {width}, h {height} {
  const auto color_usage = VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  const auto depth_usage = VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  albedo = make_image(graph, VK_FORMAT_R8G8B8A8_SRGB, width, height, color_usage);
  normal = make_image(graph, VK_FORMAT_R16G16_UNORM, width, height, color_usage);
  velocity_vectors = make_image(graph, VK_FORMAT_R16G16_SFLOAT, width, height, color_usage);
  downsampled_normals = make_image(graph, VK_FORMAT_R16G16_UNORM, width/2, height/2, color_usage);
  downsampled_velocity_vectors = make_image(graph, VK_FORMAT_R16G16_SFLOAT, width/2, height/2, color_usage);
  material = make_image(graph, VK_FORMAT_R8G8B8A8_SRGB, width, height, color_usage);
  const uint32_t depth_mips = uint32_t(std::floor(std::log2(std::max(width, height)))) + 1;  // :13
  const auto ds = DEPTH|VK_IMAGE_ASPECT_STENCIL_BIT;
  depth = make_image(graph, VK_FORMAT_D24_UNORM_S8_UINT, width, height, depth_usage, ds, depth_mips);
  prev_depth = make_image(graph, VK_FORMAT_D24_UNORM_S8_UINT, width, height, depth_usage|VK_IMAGE_USAGE_TRANSFER_DST_BIT, ds, depth_mips);
  frame_hiz = depth;
  frame_normals = downsampled_normals;
  frame_albedo = albedo;
}

void Gbuffer::enable_tiling(RenderGraph &graph, uint32_t full_width, uint32_t full_height) {
  tiled = true;
  const uint32_t frame_mips = uint32_t(std::floor(std::log2(std::max(full_width, full_height)))) + 1;
  frame_hiz = graph.create_frame_image(gpu::ImageInfo {VK_FORMAT_D24_UNORM_S8_UINT, DEPTH, full_width/2, full_height/2, 1, frame_mips - 1, 1});
  frame_normals = graph.create_frame_image(gpu::ImageInfo {VK_FORMAT_R16G16_UNORM, COLOR, full_width/2, full_height/2});
  frame_albedo = graph.create_frame_image(gpu::ImageInfo {VK_FORMAT_R8G8B8A8_SRGB, COLOR, full_width, full_height});
}

void Gbuffer::enable_normal_requests(RenderGraph &graph, uint32_t row0, uint32_t row1) {
  if (!tiled) throw std::runtime_error {"Gbuffer::enable_normal_requests: the G-buffer is not tiled"};
  normals_by_request = true;
  normal_row0 = row0; normal_row1 = row1;
  const auto usage = VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_SAMPLED_BIT;
  pend_mask = graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo {VK_FORMAT_R8_UNORM, COLOR, w/2, h/2}, VK_IMAGE_TILING_OPTIMAL, usage);
  pend_data = graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo {VK_FORMAT_R32G32B32A32_SFLOAT, COLOR, 2 * (w/2), h/2}, VK_IMAGE_TILING_OPTIMAL, usage);
}

void Gbuffer::enable_pipelining(RenderGraph &graph) {
  if (pipelined) return;
  const auto color_usage = VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  const auto depth_usage = VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  const uint32_t depth_mips = uint32_t(std::floor(std::log2(std::max(w, h)))) + 1;
  depth_next = make_image(graph, VK_FORMAT_D24_UNORM_S8_UINT, w, h, depth_usage|VK_IMAGE_USAGE_TRANSFER_DST_BIT, DEPTH|VK_IMAGE_ASPECT_STENCIL_BIT, depth_mips);
  downsampled_normals_next = make_image(graph, VK_FORMAT_R16G16_UNORM, w/2, h/2, color_usage);
  downsampled_velocity_vectors_next = make_image(graph, VK_FORMAT_R16G16_SFLOAT, w/2, h/2, color_usage);
  pipelined = true;
}

void Gbuffer::swap_sets(RenderGraph &graph) {
  if (!pipelined) throw std::runtime_error {"Gbuffer::swap_sets: pipelining is not enabled"};
  graph.remap(depth, depth_next);  // (untiled, frame_hiz / frame_normals ARE these ids and follow the swap)
  graph.remap(downsampled_normals, downsampled_normals_next);
  graph.remap(downsampled_velocity_vectors, downsampled_velocity_vectors_next);
}

// ==== DownsamplePass (downsample_pass.cpp) ================================================================
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
DownsamplePass::DownsamplePass() : sampler {default_sampler()} {
  gpu::Registers always_write {};  // :6-9: depth test ALWAYS + depth write
  always_write.depth_stencil.depthTestEnable = VK_TRUE;
  always_write.depth_stencil.depthCompareOp = VK_COMPARE_OP_ALWAYS;
  always_write.depth_stencil.depthWriteEnable = VK_TRUE;
  downsample_gbuffer = fullscreen_pipeline("downsample_gbuffer", always_write);
  downsample_depth = fullscreen_pipeline("depth_mips", always_write);
}
#endif

// :25-92.  Size checks and their messages :37-50.
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void DownsamplePass::run_downsample_gbuff(RenderGraph &graph, ImageResourceId src_normals, ImageResourceId src_velocity, ImageResourceId depth,
  ImageResourceId out_normal, ImageResourceId out_velocity)
{
  const auto d = graph.get_descriptor(depth), n = graph.get_descriptor(out_normal), v = graph.get_descriptor(out_velocity);
  if (d.mip_levels < 2)
    throw std::runtime_error {"Can't downsample depth texture with 1 mip level"};
  const uint32_t half_w = std::max(1u, d.width/2), half_h = std::max(1u, d.height/2);
  if (half_w != n.width || half_h != n.height || v.width != n.width || v.height != n.height)
    throw std::runtime_error {"Output textures have different sizes"};
  downsample_gbuffer.set_rendersubpass({true, {n.format, v.format, d.format}});
  rec::fullscreen(graph, "DownsampleGbuffer", downsample_gbuffer,
    {rec::sampled_mips(0, depth, sampler, DEPTH, 0, 1), rec::sampled_mips(1, src_normals, sampler, COLOR, 0, 1),
     rec::sampled_mips(2, src_velocity, sampler, COLOR, 0, 1),
     rec::color_target(out_normal), rec::color_target(out_velocity), rec::depth_target(depth, 1)},
    rec::no_push(), half_w, half_h);
}
#endif

// :94-131 records one "DownsampleDepth" draw per mip (L-2 dependent passes).  MI355X-first: one task whose
// attachments are all remaining mips; the bound program reduces five levels per workgroup through LDS.
// Each mip is still the 2x2 min of its parent with extent max(1, W >> i) (:118-120).
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void DownsamplePass::run_downsample_depth(RenderGraph &graph, ImageResourceId depth, uint32_t src_mip) {
  const auto desc = graph.get_descriptor(depth);
  if (src_mip + 1 >= desc.mip_levels) return;
  downsample_depth.set_rendersubpass({true, {desc.format}});
  std::vector<rec::Binding> binds {rec::sampled_mips(0, depth, sampler, DEPTH, src_mip, 1)};
  for (uint32_t mip = src_mip + 1; mip < desc.mip_levels; mip++) binds.push_back(rec::depth_target(depth, mip));
  rec::fullscreen(graph, "DownsampleDepth", downsample_depth, binds, rec::no_push(),
                  std::max(desc.width >> (src_mip + 1), 1u), std::max(desc.height >> (src_mip + 1), 1u));
}
#endif

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void DownsamplePass::run(RenderGraph &graph, ImageResourceId src_normals, ImageResourceId src_velocity, ImageResourceId depth,
  ImageResourceId out_normals, ImageResourceId out_velocity)
{
  run_downsample_gbuff(graph, src_normals, src_velocity, depth, out_normals, out_velocity);  // :133-143
  run_downsample_depth(graph, depth, 1);
}
#endif

// ==== GTAO (gtao.cpp) ================================================================================
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
ImageResourceId create_gtao_texture(RenderGraph &graph, uint32_t width, uint32_t height) {  // :10-13
  return make_image(graph, VK_FORMAT_R8_UNORM, width, height, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT);
}
#endif

// resources :17-47, pipelines :49-81
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
GTAO::GTAO(RenderGraph &graph, uint32_t width, uint32_t height, bool use_ray_query, bool half_res, int pattern_n)
  : deinterleave_n {pattern_n}, pinned_jitter {std::numeric_limits<float>::quiet_NaN()}
{
  if (use_ray_query)
    throw std::runtime_error {"GTAO: ray-query AO needs an acceleration structure; not available on the HIP path"};
  if (half_res) {
    width /= 2;
    height /= 2;
    depth_lod = 1;
  }
  const auto usage = VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_SAMPLED_BIT;
  raw = make_image(graph, VK_FORMAT_R16G16B16A16_SFLOAT, width, height, usage|VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT);
  filtered = make_image(graph, VK_FORMAT_R16_SFLOAT, width, height, usage);
  prev_frame = make_image(graph, VK_FORMAT_R16_SFLOAT, width, height, usage);
  output = make_image(graph, VK_FORMAT_R16_SFLOAT, width, height, usage);
  accumulated_ao = make_image(graph, VK_FORMAT_R16G16_SFLOAT, width, height, usage);
  accumulated_history = make_image(graph, VK_FORMAT_R16G16_SFLOAT, width, height, usage);
  const uint32_t pattern_step = 1u << uint32_t(pattern_n);
  deinterleaved_depth = make_image(graph, VK_FORMAT_R32_SFLOAT, width/pattern_step, height/pattern_step, usage, COLOR, 1, pattern_step * pattern_step);

  main_pipeline = gpu::create_compute_pipeline("gtao_compute_main");
  filter_pipeline = gpu::create_compute_pipeline("gtao_filter");
  accumulate_pipeline = gpu::create_compute_pipeline("gtao_accumulate");
  reproject_pipeline = gpu::create_compute_pipeline("gtao_reproject");
  deinterleave_pipeline = gpu::create_compute_pipeline("deinterleave_depth");
  main_deinterleaved_pipeline = gpu::create_compute_pipeline("main_deinterleaved");
  main_pipeline_gfx = fullscreen_pipeline("gtao_main");
  main_pipeline_gfx.set_rendersubpass({false, {graph.get_descriptor(raw).format}});
  sampler = default_sampler();
}
#endif

// the 12-entry angle table + jitter every main-pass flavour uses (:109-111, :362-364, :487-489)
float GTAO::next_base_angle() {
  static const float table[12] {60.f, 300.f, 180.f, 240.f, 120.f, 0.f, 300.f, 60.f, 180.f, 120.f, 240.f, 0.f};
  const float jitter = std::isnan(pinned_jitter)? (rand()/float(RAND_MAX) - 0.5f) : pinned_jitter;
  return table[frame_count++ % 12]/360.f + jitter;
}

// :84-148; PushConsts :101-113; floor dispatch :145
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_main_pass(RenderGraph &graph, const GTAOParams &params, ImageResourceId depth, ImageResourceId normal,
  ImageResourceId material, ImageResourceId preintegrated_pdf)
{
  static_assert(sizeof(GTAOParams) == sizeof(vkr_gtao_params), "GTAOParams must match the C-ABI");
  const vkr_gtao_push pc {next_base_angle(), weight_ratio, mis_gtao? 1u : 0u, two_directions? 255u : 0u, only_reflections? 255u : 0u};
  rec::compute(graph, "GTAO_main", main_pipeline,
    {rec::sampled_mips(0, depth, sampler, DEPTH, depth_lod, 1), rec::uniform(1, params), rec::sampled(2, normal, sampler),
     rec::sampled(3, material, sampler), rec::sampled(4, preintegrated_pdf, sampler), rec::storage(5, raw)},
    rec::push(pc), rec::Grid {raw, 8, 4, rec::Floor});
}
#endif

// :198-239
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_filter_pass(RenderGraph &graph, const GTAOParams &params, ImageResourceId depth) {
  const vkr_gtao_filter_push pc {params.znear, params.zfar};
  rec::compute(graph, "GTAO_filter", filter_pipeline,
    {rec::sampled_mips(0, depth, sampler, DEPTH, depth_lod, 1), rec::sampled(1, raw, sampler), rec::storage(2, filtered)},
    rec::push(pc), rec::Grid {filtered, 8, 4, rec::Floor});
}
#endif

// :286-347; AccumConstants :300-305; clear_history is consumed once :313-315; ceil dispatch :345
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_accumulate_pass(RenderGraph &graph, const DrawTAAParams &params, const Gbuffer &gbuffer) {
  vkr_gtao_accum_params consts;
  copy_mat(consts.inverse_camera, glm::inverse(params.camera));
  copy_mat(consts.prev_inverse_camera, glm::inverse(params.prev_camera));
  copy_mat(consts.mvp, params.mvp);
  std::memcpy(consts.fovy_aspect_znear_zfar, &params.fovy_aspect_znear_zfar, sizeof(consts.fovy_aspect_znear_zfar));
  const vkr_gtao_accum_push pc {clear_history? 1u : 0u};
  clear_history = false;
  rec::compute(graph, "GTAO_accumulate", accumulate_pipeline,
    {rec::sampled_mips(0, gbuffer.depth, sampler, DEPTH, depth_lod, 1), rec::sampled_mips(1, gbuffer.prev_depth, sampler, DEPTH, depth_lod, 1),
     rec::sampled(2, filtered, sampler), rec::storage(3, accumulated_ao), rec::sampled(4, gbuffer.downsampled_velocity_vectors, sampler),
     rec::sampled(5, accumulated_history, sampler), rec::uniform(6, consts)},
    rec::push(pc), rec::Grid {accumulated_ao, 8, 4, rec::Ceil});
}
#endif

// ---- the variants the reference's frame loop never records ----
// :349-413: full-screen triangle into `raw`
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_main_pass_graphics(RenderGraph &graph, const GTAOParams &params, ImageResourceId depth, ImageResourceId normal) {
  const vkr_gtao_gfx_push pc {next_base_angle()};
  const auto ext = graph.get_descriptor(raw);
  rec::fullscreen(graph, "GTAO", main_pipeline_gfx,
    {rec::sampled_mips(0, depth, sampler, DEPTH, depth_lod, 1), rec::uniform(1, params), rec::sampled(2, normal, sampler), rec::color_target(raw)},
    rec::push(pc), ext.width, ext.height);
}
#endif

// :150-196 (VK_KHR_ray_query against the scene's TLAS): not part of this path
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_main_rt_pass(RenderGraph &, const GTAORTParams &, VkAccelerationStructureKHR, ImageResourceId, ImageResourceId) {
  throw std::runtime_error {"GTAO::add_main_rt_pass: the ray-query pass needs a scene acceleration structure (out of scope: SURVEY.md section 2b)"};
}
#endif

// :528-536, the panel as the reference draws it; the ImGui names are inert in this headless build (imgui_pass.hpp)
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::draw_ui() {
  ImGui::Begin("GTAO");
  ImGui::Checkbox("Enable MIS", &mis_gtao);
  ImGui::Checkbox("Use 2 directions", &two_directions);
  ImGui::Checkbox("Only reflections ao", &only_reflections);
  ImGui::SliderFloat("Weight ratio", &weight_ratio, 1.0, 5.0);
  clear_history = ImGui::Button("Clear history") || clear_history;  // a headless request_clear_history() stays pending
  ImGui::End();
}
#endif

// :241-284: filtered + prev_frame -> output
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_reprojection_pass(RenderGraph &graph, const GTAOReprojection &params, ImageResourceId depth, ImageResourceId prev_depth) {
  static_assert(sizeof(GTAOReprojection) == sizeof(vkr_gtao_reprojection), "GTAOReprojection must match the C-ABI");
  rec::compute(graph, "GTAO_reproject", reproject_pipeline,
    {rec::uniform(0, params), rec::sampled_mips(1, depth, sampler, DEPTH, depth_lod, 1), rec::sampled_mips(2, prev_depth, sampler, DEPTH, depth_lod, 1),
     rec::sampled(3, filtered, sampler), rec::sampled(4, prev_frame, sampler), rec::storage(5, output)},
    rec::no_push(), rec::Grid {output, 8, 4, rec::Floor});
}
#endif

// :445-470: the dispatch is sized by the *array* extent, as the reference records it (:468)
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::deinterleave_depth(RenderGraph &graph, ImageResourceId depth) {
  const vkr_deinterleave_push pc {deinterleave_n};
  rec::compute(graph, "GTAO_deinterleave", deinterleave_pipeline,
    {rec::sampled_mips(0, depth, sampler, DEPTH, depth_lod, 1), rec::storage_array(1, deinterleaved_depth)},
    rec::push(pc), rec::Grid {deinterleaved_depth, 8, 4, rec::Floor});
}
#endif

// :472-526: one dispatch per array layer of the *output* image (raw has one), exactly as the reference loops
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void GTAO::add_main_pass_deinterleaved(RenderGraph &graph, const GTAOParams &params, ImageResourceId normal) {
  const float base_angle = next_base_angle();
  const uint32_t layers = graph.get_descriptor(raw).array_layers? graph.get_descriptor(raw).array_layers : 1;
  for (uint32_t layer = 0; layer < layers; layer++) {
    const vkr_gtao_deinterleaved_push pc {deinterleave_n, layer, base_angle};
    rec::compute(graph, "GTAO_deinterleaved", main_deinterleaved_pipeline,
      {rec::sampled(0, deinterleaved_depth, sampler), rec::uniform(1, params), rec::sampled(2, normal, sampler), rec::storage(3, raw)},
      rec::push(pc), rec::Grid {raw, 8, 4, rec::Floor});
  }
}
#endif

// ==== AdvancedSSR (advanced_ssr.cpp) ========================================================================
// radical inverse with the reference's float-floor division (:8-20) kept as is
static float halton_elem(uint32_t index, uint32_t base) {
  float scale = 1.f, result = 0.f;
  for (uint32_t current = index;;) {
    scale = scale / float(base);
    result = result + scale * float(current % base);
    current = uint32_t(std::floor(float(current) / float(base)));
    if (current == 0) break;
  }
  return result;
}

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
std::vector<glm::vec4> halton23_seq(uint32_t count) {  // :22-34
  std::vector<glm::vec4> seq(count);
  for (uint32_t i = 0; i < count; i++)
    seq[i] = glm::vec4 {halton_elem(i + 1, 2), halton_elem(i + 1, 3), 0.f, 0.f};
  return seq;
}
#endif

// :36-93
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
AdvancedSSR::AdvancedSSR(RenderGraph &graph, uint32_t w, uint32_t h) {
  trace_pass = gpu::create_compute_pipeline("sssr_trace");
  filter_pass = gpu::create_compute_pipeline("sssr_filter");
  blur_pass = gpu::create_compute_pipeline("sssr_blur");
  preintegrate_pass = gpu::create_compute_pipeline("pdf_preintegrate");
  preintegrate_brdf_pass = gpu::create_compute_pipeline("brdf_preintegrate");
  classification_pass = gpu::create_compute_pipeline("sssr_classification");
  trace_indirect_pass = gpu::create_compute_pipeline("sssr_trace_indirect");

  // xy: the reference's table; zw: cos / sin of 2 PI y evaluated on the host for the HIP trace kernel
  // (vkr_halton23_fill, include/vkr_postfx.h) — zero in the reference
  auto halton_samples = halton23_seq(HALTON_SEQ_SIZE);
  std::vector<float> filled(4 * HALTON_SEQ_SIZE);
  vkr_halton23_fill(filled.data(), HALTON_SEQ_SIZE);
  for (uint32_t i = 0; i < HALTON_SEQ_SIZE; i++) {
    if (filled[4 * i] != halton_samples[i].x || filled[4 * i + 1] != halton_samples[i].y)
      throw std::runtime_error {"Halton table mismatch between host pass and C-ABI helper"};
    halton_samples[i].z = filled[4 * i + 2];
    halton_samples[i].w = filled[4 * i + 3];
  }
  const uint64_t bytes = sizeof(halton_samples[0]) * HALTON_SEQ_SIZE;
  halton_buffer = gpu::create_buffer(VMA_MEMORY_USAGE_CPU_TO_GPU, bytes, VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
  std::memcpy(halton_buffer->get_mapped_ptr(), halton_samples.data(), bytes);

  const auto usage = VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_STORAGE_BIT;
  rays = make_image(graph, VK_FORMAT_R16G16B16A16_UNORM, w/2, h/2, usage);
  rays_occlusion = make_image(graph, VK_FORMAT_R16_SFLOAT, w/2, h/2, usage);
  reflections = make_image(graph, VK_FORMAT_R8G8B8A8_UNORM, w/2, h/2, usage);
  blurred_reflection = make_image(graph, VK_FORMAT_R8G8B8A8_UNORM, w/2, h/2, usage);
  blurred_reflection_history = make_image(graph, VK_FORMAT_R8G8B8A8_UNORM, w/2, h/2, usage);
  preintegrated_pdf = make_image(graph, VK_FORMAT_R32_SFLOAT, 1024, 1024, usage);
  preintegrated_brdf = make_image(graph, VK_FORMAT_R16G16_SFLOAT, 1024, 1024, usage);
  sampler = default_sampler();

  // :77-83: indirect arguments and one tile-index slot per 8x8 block of the full-res frame
  const auto indirect_usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT|VK_BUFFER_USAGE_INDIRECT_BUFFER_BIT|VK_BUFFER_USAGE_TRANSFER_DST_BIT;
  reflective_indirect = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, sizeof(VkDispatchIndirectCommand), indirect_usage);
  glossy_indirect = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, sizeof(VkDispatchIndirectCommand), indirect_usage);
  const uint64_t tile_bytes = sizeof(int) * std::max<uint64_t>(1, uint64_t(w) * h/64);
  reflective_tiles = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, tile_bytes, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
  glossy_tiles = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, tile_bytes, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
}
#endif

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::preintegrate_pdf(RenderGraph &graph) {  // :95-114
  rec::compute(graph, "SSR_preintegrate", preintegrate_pass, {rec::storage(0, preintegrated_pdf)}, rec::no_push(),
               rec::Grid {preintegrated_pdf, 8, 4, rec::Ceil});
}
#endif

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::preintegrate_brdf(RenderGraph &graph) {  // :116-136
  rec::compute(graph, "BRDF_preintegrate", preintegrate_brdf_pass,
               {rec::uniform_buffer(0, halton_buffer), rec::storage(1, preintegrated_brdf)}, rec::no_push(),
               rec::Grid {preintegrated_brdf, 8, 4, rec::Ceil});
}
#endif

// TraceParams (:138-145) with the frame counter, which cycles modulo max_accumulated_rays (:168-171)
static vkr_trace_params trace_params(const AdvancedSSRParams &p, uint32_t counter) {
  vkr_trace_params t;
  copy_mat(t.normal_mat, p.normal_mat);
  t.frame_random = counter;
  t.fovy = p.fovy; t.aspect = p.aspect; t.znear = p.znear; t.zfar = p.zfar;
  return t;
}

void AdvancedSSR::advance_counter() {
  if (settings.update_random) counter = (counter + 1) % uint32_t(settings.max_accumulated_rays);
}

// :147-214.  Single GPU: the march reads image mips 1..L-1 of gbuff.depth (:186); tiled: the gathered
// whole-frame pyramid and normals.
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run_trace_pass(RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, ImageResourceId ssr_occlusion) {
  const vkr_trace_params config = trace_params(params, counter);
  const vkr_trace_push pc {settings.max_rougness};
  advance_counter();
  const auto hiz = gbuff.tiled? gbuff.frame_hiz : gbuff.depth;
  const uint32_t mips = graph.get_descriptor(hiz).mip_levels;
  if (gbuff.tiled && gbuff.normals_by_request) {  // multi-GPU: the hit-normal test of rays that end on another rank's rows is deferred
    if (!trace_windowed_pass.has_program()) trace_windowed_pass = gpu::create_compute_pipeline("sssr_trace_windowed");
    const vkr_trace_window_push wpc {settings.max_rougness, gbuff.normal_row0, gbuff.normal_row1};
    rec::compute(graph, "SSSR_trace", trace_windowed_pass,
      {rec::sampled_mips(0, hiz, sampler, DEPTH, 0, mips), rec::sampled(1, gbuff.frame_normals, sampler), rec::sampled(2, gbuff.material, sampler),
       rec::uniform(3, config), rec::uniform_buffer(4, halton_buffer), rec::storage(5, rays), rec::storage(6, ssr_occlusion),
       rec::sampled(7, preintegrated_pdf, sampler), rec::storage(8, gbuff.pend_mask), rec::storage(9, gbuff.pend_data)},
      rec::push(wpc), rec::Grid {rays, 8, 8, rec::Ceil});
    return;
  }
  rec::compute(graph, "SSSR_trace", trace_pass,
    {gbuff.tiled? rec::sampled_mips(0, hiz, sampler, DEPTH, 0, mips) : rec::sampled_mips(0, hiz, sampler, DEPTH, 1, mips - 1),
     rec::sampled(1, gbuff.tiled? gbuff.frame_normals : gbuff.downsampled_normals, sampler), rec::sampled(2, gbuff.material, sampler),
     rec::uniform(3, config), rec::uniform_buffer(4, halton_buffer), rec::storage(5, rays), rec::storage(6, ssr_occlusion),
     rec::sampled(7, preintegrated_pdf, sampler)},
    rec::push(pc), rec::Grid {rays, 8, 8, rec::Ceil});
}
#endif

// :308-369; flags :331-338; depth view = mips 0..9 (:342); tiled: the hit colour comes from the whole-frame albedo
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run_filter_pass(RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff) {
  const vkr_trace_params config = trace_params(params, counter);
  vkr_filter_push pc {0u};
  if (settings.normalize_reflections) pc.render_flags |= NORMALIZE_REFLECTIONS;
  if (settings.accumulate_reflections) pc.render_flags |= ACCUMULATE_REFLECTIONS;
  if (settings.bilateral_filter) pc.render_flags |= BILATERAL_FILTER;
  const uint32_t depth_mips = std::min(10u, graph.get_descriptor(gbuff.depth).mip_levels);
  rec::compute(graph, "SSSR_filter", filter_pass,
    {rec::sampled(0, rays, sampler), rec::sampled_mips(1, gbuff.depth, sampler, DEPTH, 0, depth_mips),
     rec::sampled(2, gbuff.tiled? gbuff.frame_albedo : gbuff.albedo, sampler), rec::sampled(3, gbuff.normal, sampler),
     rec::sampled(4, gbuff.material, sampler), rec::storage(5, reflections), rec::uniform(6, config)},
    rec::push(pc), rec::Grid {reflections, 8, 8, rec::Ceil});
}
#endif

// :371-438
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run_blur_pass(RenderGraph &graph, const AdvancedSSRParams &, const DrawTAAParams &taa_params, const Gbuffer &gbuff) {
  const vkr_blur_push pc {settings.max_rougness, settings.accumulate_reflections? 1u : 0u, settings.use_blur? 0u : 1u};
  const uint32_t depth_mips = std::min(10u, graph.get_descriptor(gbuff.depth).mip_levels);
  rec::compute(graph, "SSSR_blur", blur_pass,
    {rec::sampled_mips(0, gbuff.depth, sampler, DEPTH, 0, depth_mips), rec::sampled(1, gbuff.normal, sampler), rec::sampled(2, reflections, sampler),
     rec::sampled(3, gbuff.material, sampler), rec::sampled(4, blurred_reflection_history, sampler),
     rec::sampled(5, gbuff.downsampled_velocity_vectors, sampler), rec::sampled_mips(6, gbuff.prev_depth, sampler, DEPTH, 0, depth_mips),
     rec::storage(7, blurred_reflection), rec::uniform(8, reproject_params(taa_params))},
    rec::push(pc), rec::Grid {blurred_reflection, 8, 8, rec::Ceil});
}
#endif

// :440-452: VkDispatchIndirectCommand{0, 1, 1} into both argument buffers
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::clear_indirect_params(RenderGraph &graph) {
  struct Nothing {};
  const auto a = reflective_indirect, b = glossy_indirect;
  graph.add_task<Nothing>("SSSR_Clear",
    [&](Nothing &, rendergraph::RenderGraphBuilder &builder) {
      builder.transfer_write(a);
      builder.transfer_write(b);
    },
    [=](Nothing &, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      const VkDispatchIndirectCommand none {0, 1, 1};
      cmd.update_buffer(resources.get_buffer(a)->api_buffer(), 0, none);
      cmd.update_buffer(resources.get_buffer(b)->api_buffer(), 0, none);
    });
}
#endif

// :454-495
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run_classification_pass(RenderGraph &graph, const AdvancedSSRParams &, const Gbuffer &gbuff) {
  const auto extent = graph.get_descriptor(rays).extent2D();
  const vkr_classification_push pc {int(extent.width), int(extent.height), settings.max_rougness, settings.glossy_roughness_value};
  rec::compute(graph, "SSSR_Classification", classification_pass,
    {rec::sampled(0, gbuff.material, sampler), rec::storage_buffer(1, reflective_tiles, false), rec::storage_buffer(2, glossy_tiles, false),
     rec::storage_buffer(3, reflective_indirect, false), rec::storage_buffer(4, glossy_indirect, false)},
    rec::push(pc), rec::Grid {rays, 8, 8, rec::Ceil});
}
#endif

// :216-302: two indirect dispatches of one program, mirror tiles (reflection_type 0) then glossy tiles (1)
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run_trace_indirect_pass(RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff) {
  const vkr_trace_params config = trace_params(params, counter);
  const float max_roughness = settings.max_rougness;
  advance_counter();
  const auto hiz = gbuff.tiled? gbuff.frame_hiz : gbuff.depth;
  const uint32_t mips = graph.get_descriptor(hiz).mip_levels;
  const std::vector<rec::Binding> common {
    gbuff.tiled? rec::sampled_mips(0, hiz, sampler, DEPTH, 0, mips) : rec::sampled_mips(0, hiz, sampler, DEPTH, 1, mips - 1),
    rec::sampled(1, gbuff.tiled? gbuff.frame_normals : gbuff.downsampled_normals, sampler), rec::sampled(2, gbuff.material, sampler),
    rec::uniform(3, config), rec::uniform_buffer(4, halton_buffer), rec::storage(5, rays)};
  const rendergraph::BufferResourceId lists[2] {reflective_tiles, glossy_tiles}, arguments[2] {reflective_indirect, glossy_indirect};
  const auto pipeline = trace_indirect_pass;

  struct Data { std::vector<rec::Bound> bound; };
  graph.add_task<Data>("SSSR_trace",
    [&](Data &d, rendergraph::RenderGraphBuilder &builder) {
      d.bound = rec::declare(common, builder, VK_SHADER_STAGE_COMPUTE_BIT);
      for (int k = 0; k < 2; k++) {
        builder.use_indirect_buffer(arguments[k]);
        builder.use_storage_buffer(lists[k], VK_SHADER_STAGE_COMPUTE_BIT);
      }
    },
    [=](Data &d, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      cmd.bind_pipeline(pipeline);
      for (uint32_t kind = 0; kind < 2; kind++) {
        VkDescriptorSet set = rec::write(d.bound, resources, cmd);
        gpu::write_set(set, gpu::SSBOBinding {6, resources.get_buffer(lists[kind])});
        const vkr_trace_indirect_push pc {kind, max_roughness};
        cmd.bind_descriptors_compute(0, {set});
        cmd.push_constants_compute(0, sizeof(pc), &pc);
        cmd.dispatch_indirect(resources.get_buffer(arguments[kind])->api_buffer());
      }
    });
}
#endif

void AdvancedSSR::run_trace(RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, ImageResourceId ssr_occlusion) {
  if (settings.use_tile_classification) {  // the path advanced_ssr.cpp:547-550 keeps commented out
    clear_indirect_params(graph);
    run_classification_pass(graph, params, gbuff);
    run_trace_indirect_pass(graph, params, gbuff);
  } else {
    run_trace_pass(graph, params, gbuff, ssr_occlusion);
  }
}

void AdvancedSSR::run_trace_head(RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, ImageResourceId ssr_occlusion, uint32_t local_levels) {
  if (!gbuff.tiled || !gbuff.normals_by_request) throw std::runtime_error {"run_trace_head: only the tiled frame with hit normals by request traces in two tasks"};
  if (settings.use_tile_classification) throw std::runtime_error {"run_trace_head: not with tile classification"};
  head_config = trace_params(params, counter);
  advance_counter();
  if (!trace_head_pass.has_program()) trace_head_pass = gpu::create_compute_pipeline("sssr_trace_windowed_head");
  const uint32_t mips = graph.get_descriptor(gbuff.frame_hiz).mip_levels;
  const vkr_trace_window_push wpc {settings.max_rougness, gbuff.normal_row0, gbuff.normal_row1};
  rec::compute(graph, "SSSR_trace", trace_head_pass,
    {rec::sampled_mips(0, gbuff.depth, sampler, DEPTH, 1, local_levels), rec::sampled(1, gbuff.frame_normals, sampler), rec::sampled(2, gbuff.material, sampler),
     rec::uniform(3, head_config), rec::uniform_buffer(4, halton_buffer), rec::storage(5, rays), rec::storage(6, ssr_occlusion),
     rec::sampled(7, preintegrated_pdf, sampler), rec::storage(8, gbuff.pend_mask), rec::storage(9, gbuff.pend_data),
     rec::sampled_mips(10, gbuff.frame_hiz, sampler, DEPTH, 0, mips)},  // extents and level count only: its rows are still arriving
    rec::push(wpc), rec::Grid {rays, 8, 8, rec::Ceil});
}

void AdvancedSSR::run_trace_resume(RenderGraph &graph, const Gbuffer &gbuff, ImageResourceId ssr_occlusion) {
  if (!trace_resume_pass.has_program()) trace_resume_pass = gpu::create_compute_pipeline("sssr_trace_windowed_resume");
  const uint32_t mips = graph.get_descriptor(gbuff.frame_hiz).mip_levels;
  const vkr_trace_window_push wpc {settings.max_rougness, gbuff.normal_row0, gbuff.normal_row1};
  rec::compute(graph, "SSSR_trace_resume", trace_resume_pass,
    {rec::sampled_mips(0, gbuff.frame_hiz, sampler, DEPTH, 0, mips), rec::sampled(1, gbuff.frame_normals, sampler), rec::sampled(2, gbuff.material, sampler),
     rec::uniform(3, head_config), rec::uniform_buffer(4, halton_buffer), rec::storage(5, rays), rec::storage(6, ssr_occlusion),
     rec::sampled(7, preintegrated_pdf, sampler), rec::storage(8, gbuff.pend_mask), rec::storage(9, gbuff.pend_data)},
    rec::push(wpc), rec::Grid {rays, 8, 8, rec::Ceil});
}

void AdvancedSSR::run_resolve(RenderGraph &graph, const AdvancedSSRParams &params, const DrawTAAParams &taa_params, const Gbuffer &gbuff) {
  run_filter_pass(graph, params, gbuff);
  run_blur_pass(graph, params, taa_params, gbuff);
}

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run(RenderGraph &graph, const AdvancedSSRParams &params, const DrawTAAParams &taa_params, const Gbuffer &gbuff,
  ImageResourceId ssr_occlusion)
{
  run_trace(graph, params, gbuff, ssr_occlusion);  // :540-554
  run_resolve(graph, params, taa_params, gbuff);
}
#endif

// ==== TAA (taa.cpp) ========================================================================================
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
TAA::TAA(RenderGraph &graph, uint32_t w, uint32_t h) {  // :3-12
  pipeline = gpu::create_compute_pipeline("taa_resolve");
  const auto usage = VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  history = make_image(graph, VK_FORMAT_R16G16B16A16_SFLOAT, w, h, usage);
  target = make_image(graph, VK_FORMAT_R16G16B16A16_SFLOAT, w, h, usage);
  sampler = default_sampler();
}
#endif

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void TAA::run(RenderGraph &graph, const Gbuffer &gbuffer, ImageResourceId color, const DrawTAAParams &params) {  // :19-63
  rec::compute(graph, "TAA", pipeline,
    {rec::sampled(0, history, sampler), rec::sampled(1, gbuffer.prev_depth, sampler, DEPTH), rec::sampled(2, gbuffer.depth, sampler, DEPTH),
     rec::sampled(3, gbuffer.velocity_vectors, sampler), rec::sampled(4, color, sampler), rec::storage(5, target),
     rec::uniform(6, reproject_params(params))},
    rec::no_push(), rec::Grid {target, 8, 8, rec::Ceil});
}
#endif

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void TAA::remap_targets(RenderGraph &graph) { graph.remap(history, target); }  // :65-67

// ==== simple SSR (ssr.cpp) =====================================================================================
ImageResourceId create_ssr_tex(RenderGraph &graph, uint32_t w, uint32_t h) {  // :5-8
  return make_image(graph, VK_FORMAT_R8G8B8A8_UNORM, w, h, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_STORAGE_BIT);
}
#endif

// :10-73.  Depth is read through a NEAREST sampler with U / W clamp-to-border (:21-28).
void add_ssr_pass(RenderGraph &graph, ImageResourceId depth, ImageResourceId normal, ImageResourceId color, ImageResourceId material,
  ImageResourceId out, const SSRParams &params)
{
  static_assert(sizeof(SSRParams) == sizeof(vkr_ssr_params), "SSRParams must match the C-ABI");
  auto nearest = gpu::DEFAULT_SAMPLER;
  nearest.minFilter = nearest.magFilter = VK_FILTER_NEAREST;
  nearest.mipmapMode = VK_SAMPLER_MIPMAP_MODE_NEAREST;
  nearest.addressModeU = nearest.addressModeW = VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_BORDER;
  const auto sampler = default_sampler();
  auto pipeline = fullscreen_pipeline("ssr");
  pipeline.set_rendersubpass({false, {graph.get_descriptor(out).format}});
  const auto ext = graph.get_descriptor(out);
  rec::fullscreen(graph, "SSR", pipeline,
    {rec::sampled(0, normal, sampler), rec::sampled(1, depth, gpu::create_sampler(nearest), DEPTH), rec::sampled(2, color, sampler),
     rec::uniform(3, params), rec::sampled(4, material, sampler), rec::color_target(out)},
    rec::no_push(), ext.width, ext.height);
}

// ==== ScreenSpaceTrace (screen_trace.cpp) ========================================================================
ScreenSpaceTrace::ScreenSpaceTrace(RenderGraph &graph, uint32_t width, uint32_t height) {  // :3-21
  const auto usage = VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_SAMPLED_BIT;
  raw = make_image(graph, VK_FORMAT_R16G16B16A16_SFLOAT, width, height, usage);
  filtered = make_image(graph, VK_FORMAT_R16G16B16A16_SFLOAT, width, height, usage);
  accumulated = make_image(graph, VK_FORMAT_R16G16B16A16_SFLOAT, width, height, usage);
  trace_pipeline = gpu::create_compute_pipeline("screen_trace_main");
  filter_pipeline = gpu::create_compute_pipeline("screen_trace_filter");
  accum_pipeline = gpu::create_compute_pipeline("screen_trace_accumulate");
  sampler = default_sampler();
}

// :23-95; angle table + jitter and random_offset drawn per frame, always in this order (:47-53); dispatch w/8 x h/8
void ScreenSpaceTrace::add_main_pass(RenderGraph &graph, const ScreenTraceParams &params, ImageResourceId depth, ImageResourceId normal,
  ImageResourceId color, ImageResourceId material)
{
  static const float table[12] {60.f, 300.f, 180.f, 240.f, 120.f, 0.f, 300.f, 60.f, 180.f, 120.f, 240.f, 0.f};
  const float drawn_jitter = random_floats(generator) - 0.5f;
  const float drawn_offset = random_floats(generator);
  vkr_screen_trace_params ubo {};
  copy_mat(ubo.normal_mat, params.normal_mat);
  ubo.angle_offset = table[frame_count++ % 12]/360.f + (std::isnan(pinned_jitter)? drawn_jitter : pinned_jitter);
  ubo.random_offset = std::isnan(pinned_offset)? drawn_offset : pinned_offset;
  ubo.fovy = params.fovy; ubo.aspect = params.aspect; ubo.znear = params.znear; ubo.zfar = params.zfar;
  rec::compute(graph, "ScreenTrace", trace_pipeline,
    {rec::sampled_mips(0, depth, sampler, DEPTH, 0, 1), rec::sampled(1, normal, sampler), rec::sampled(2, color, sampler),
     rec::sampled(3, material, sampler), rec::storage(4, raw), rec::uniform(5, ubo)},
    rec::no_push(), rec::Grid {raw, 8, 8, rec::Floor});
}

void ScreenSpaceTrace::add_filter_pass(RenderGraph &graph, const ScreenTraceParams &params, ImageResourceId depth) {  // :97-140
  const vkr_screen_trace_filter_push pc {params.znear, params.zfar};
  rec::compute(graph, "ScreenTraceFilter", filter_pipeline,
    {rec::sampled(0, raw, sampler), rec::sampled_mips(1, depth, sampler, DEPTH, 0, 1), rec::storage(2, filtered)},
    rec::push(pc), rec::Grid {filtered, 8, 4, rec::Floor});
}

void ScreenSpaceTrace::add_accumulate_pass(RenderGraph &graph, const ScreenTraceParams &params, ImageResourceId depth, ImageResourceId prev_depth) {  // :142-181
  const vkr_screen_trace_accum_push pc {params.fovy, params.aspect, params.znear, params.zfar};
  rec::compute(graph, "ScreenTraceAccumulate", accum_pipeline,
    {rec::sampled_mips(0, depth, sampler, DEPTH, 0, 1), rec::sampled_mips(1, prev_depth, sampler, DEPTH, 0, 1), rec::sampled(2, filtered, sampler),
     rec::storage(3, accumulated)},
    rec::push(pc), rec::Grid {accumulated, 8, 4, rec::Floor});
}

// ==== DeferedShadingPass (defered_shading.cpp) ======================================================================
// advanced_ssr.cpp:497-545: the program "tile_regression" is not in the table (out of scope)
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::run_tile_regression_pass(RenderGraph &, const AdvancedSSRParams &, const Gbuffer &) {
  throw std::runtime_error {"AdvancedSSR::run_tile_regression_pass: tile_regression is not implemented on the post-process path"};
}
#endif

// advanced_ssr.cpp:556-567
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void AdvancedSSR::render_ui() {
  ImGui::Begin("SSSR");
  ImGui::SliderFloat("Max Roughness", &settings.max_rougness, 0.f, 1.f);
  ImGui::SliderFloat("Min glossy roughness", &settings.glossy_roughness_value, 0.f, 1.f);
  ImGui::SliderInt("Temporal rays", &settings.max_accumulated_rays, 1, 128);
  ImGui::Checkbox("Enable normalization", &settings.normalize_reflections);
  ImGui::Checkbox("Enable accumulation", &settings.accumulate_reflections);
  ImGui::Checkbox("Enable random rays", &settings.update_random);
  ImGui::Checkbox("Enable blur", &settings.use_blur);
  ImGui::Checkbox("Enable bilateral filter", &settings.bilateral_filter);
  ImGui::End();
}
#endif

// defered_shading.cpp:120-126
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void DeferedShadingPass::draw_ui() {
  ImGui::Begin("DeferedShading");
  ImGui::SliderFloat("Max Roughness", &min_max_roughness.y, min_max_roughness.x, 1.f);
  ImGui::SliderFloat("Min Roughness", &min_max_roughness.x, 0.f, min_max_roughness.y);
  ImGui::Checkbox("Show AO only", &only_ao);
  ImGui::End();
}
#endif

#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
DeferedShadingPass::DeferedShadingPass(RenderGraph &graph, SDL_Window *) {  // :14-31
  pipeline = fullscreen_pipeline("defered_shading");
  sampler = default_sampler();
  ubo_consts = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, sizeof(vkr_shading_params), VK_BUFFER_USAGE_TRANSFER_DST_BIT|VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
  graph_ref = &graph;
}
#endif

// :33-45: ShaderConstants {inverse(camera), camera, shadow_mvp, fovy, aspect, znear, zfar} written into the constant
// buffer (gpu_transfer::write_buffer in the reference; here the buffer keeps a host shadow the program reads)
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void DeferedShadingPass::update_params(const glm::mat4 &camera, const glm::mat4 &shadow, float fovy, float aspect, float znear, float zfar) {
  vkr_shading_params consts;
  copy_mat(consts.inverse_camera, glm::inverse(camera));
  copy_mat(consts.camera, camera);
  copy_mat(consts.shadow_mvp, shadow);
  consts.fovy = fovy; consts.aspect = aspect; consts.znear = znear; consts.zfar = zfar;
  std::memcpy(graph_ref->get_buffer(ubo_consts)->get_mapped_ptr(), &consts, sizeof(consts));
}
#endif

// :47-118.  Bindings 0-8 (:93-102); the shadow map (binding 5) is bound by the reference but never read by the
// shader, so it may be left out; push constants {vec2 min_max_roughness, uint show_ao} (:68-72).
#ifndef VKR_REFERENCE_PASSES  // defined by the reference's own source in the drop-in build (Makefile: refpasses)
void DeferedShadingPass::draw(RenderGraph &graph, const Gbuffer &gbuffer, ImageResourceId shadow, ImageResourceId ssao, ImageResourceId brdf_tex,
  ImageResourceId reflections, ImageResourceId out_image)
{
  vkr_shading_push pc {{min_max_roughness.x, min_max_roughness.y}, only_ao? 1u : 0u};
  pipeline.set_rendersubpass({false, {graph.get_descriptor(out_image).format}});
  std::vector<rec::Binding> binds {
    rec::sampled(0, gbuffer.albedo, sampler), rec::sampled(1, gbuffer.normal, sampler), rec::sampled(2, gbuffer.material, sampler),
    rec::sampled(3, gbuffer.depth, sampler, DEPTH), rec::uniform_buffer(4, ubo_consts), rec::sampled(6, ssao, sampler),
    rec::sampled(7, brdf_tex, sampler), rec::sampled(8, reflections, sampler), rec::color_target(out_image)};
  if (shadow.get_index() != ~0u) binds.push_back(rec::sampled_mips(5, shadow, sampler, DEPTH, 0, 1));
  const auto ext = graph.get_descriptor(out_image);
  rec::fullscreen(graph, "DeferedShading", pipeline, binds, rec::push(pc), ext.width, ext.height);
}
#endif

// ==== SyntheticGbuffer ===============================================================================================
SyntheticGbuffer::SyntheticGbuffer(uint32_t s) : seed {s} {
  pipeline = gpu::create_graphics_pipeline();
  pipeline.set_program("synthetic_gbuffer");
  pipeline.set_vertex_input({});
}

static vkr_synth_params synth_params(const glm::mat4 &camera, const glm::mat4 &mvp, const glm::mat4 &prev_mvp, const glm::vec4 &fazz, uint32_t seed, uint32_t flags) {
  vkr_synth_params p {};
  copy_mat(p.camera_to_world, glm::inverse(camera));
  copy_mat(p.prev_mvp, prev_mvp);
  copy_mat(p.mvp, mvp);
  p.fovy = fazz.x; p.aspect = fazz.y; p.znear = fazz.z; p.zfar = fazz.w;
  p.seed = seed;
  p.flags = flags;
  return p;
}

// same attachments, in the same order, as SceneRenderer::draw_taa
void SyntheticGbuffer::draw_taa(RenderGraph &graph, const Gbuffer &gbuffer, const DrawTAAParams &params) {
  rec::fullscreen(graph, "GbufferPass", pipeline,
    {rec::uniform(0, synth_params(params.camera, params.mvp, params.prev_mvp, params.fovy_aspect_znear_zfar, seed, material_flags & VKR_SYNTH_TEXTURED_ROUGHNESS)),
     rec::color_target(gbuffer.albedo), rec::color_target(gbuffer.normal), rec::color_target(gbuffer.material),
     rec::color_target(gbuffer.velocity_vectors), rec::depth_target(gbuffer.depth)},
    rec::no_push(), gbuffer.w, gbuffer.h);
}

void SyntheticGbuffer::draw_depth(RenderGraph &graph, ImageResourceId depth_target, const glm::mat4 &camera, const glm::mat4 &mvp, const glm::vec4 &fazz) {
  const auto desc = graph.get_descriptor(depth_target);
  rec::fullscreen(graph, "GbufferDepthOnly", pipeline,
    {rec::uniform(0, synth_params(camera, mvp, mvp, fazz, seed, VKR_SYNTH_DEPTH_ONLY)), rec::depth_target(depth_target)},
    rec::no_push(), desc.width, desc.height);
}

// ==== scene (scene/scene.cpp, scene/images.cpp) and SceneRenderer (scene_renderer.cpp:46-220) ===========================
namespace scene {

CompiledScene make_scene(const Vertex *vertices, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                         const FlatDraw *draws, uint32_t draw_count, const TextureData *textures, uint32_t texture_count)
{
  CompiledScene out;
  auto upload = [](const void *src, uint64_t bytes) {  // scene.cpp:285-296: one vertex and one index buffer for the whole file
    auto buf = gpu::create_buffer(VMA_MEMORY_USAGE_CPU_TO_GPU, std::max<uint64_t>(bytes, 4), VK_BUFFER_USAGE_TRANSFER_DST_BIT);
    if (bytes) std::memcpy(buf->get_mapped_ptr(), src, bytes);
    return buf;
  };
  out.vertex_buffer = upload(vertices, sizeof(Vertex) * uint64_t(vertex_count));
  out.index_buffer = upload(indices, sizeof(uint32_t) * uint64_t(index_count));
  auto repeat = gpu::DEFAULT_SAMPLER;  // scene_renderer.cpp:77-81
  repeat.addressModeU = repeat.addressModeV = VK_SAMPLER_ADDRESS_MODE_REPEAT;
  out.samplers.push_back(gpu::create_sampler(repeat));
  for (uint32_t i = 0; i < texture_count; i++) {  // images.cpp:32-49: RGBA8_SRGB with a full mip chain
    const TextureData &t = textures[i];
    auto img = std::make_shared<gpu::Image>(gpu::ImageInfo {VK_FORMAT_R8G8B8A8_SRGB, COLOR, t.width, t.height, 1, t.mip_levels, 1}, gpu::FrameWindow {});
    bool never_zero = true;
    for (uint32_t m = 0; m < t.mip_levels; m++) {
      img->upload_mip(m, t.levels[m]);
      const size_t texels = size_t(std::max(1u, t.width >> m)) * std::max(1u, t.height >> m);
      for (size_t k = 0; k < texels && never_zero; k++) never_zero = t.levels[m][4 * k + 3] != 0;
    }
    img->alpha_never_zero = never_zero;
    out.images.push_back(img);
    out.textures.push_back(Texture {i, 0});
  }
  for (uint32_t i = 0; i < draw_count; i++) {
    const FlatDraw &d = draws[i];
    Material mat;
    mat.albedo_tex_index = d.albedo_tex_index;
    mat.metalic_roughness_index = d.metalic_roughness_index;
    mat.clip_alpha = d.clip_alpha;
    out.materials.push_back(mat);
    BaseMesh mesh;
    mesh.primitives.push_back(Primitive {d.vertex_offset, d.index_offset, d.index_count, i});
    out.root_meshes.push_back(mesh);
    out.base_nodes.push_back(BaseNode {d.transform, {}, int(i)});
  }
  return out;
}

}  // namespace scene

void SceneRenderer::init_pipeline(RenderGraph &graph, const Gbuffer &) {  // :46-103
  owner = &graph;
  gpu::Registers regs {};
  regs.depth_stencil.depthTestEnable = VK_TRUE;
  regs.depth_stencil.depthWriteEnable = VK_TRUE;
  opaque_taa_pipeline = fullscreen_pipeline("gbuf_opaque_taa", regs);
  opaque_taa_pipeline.set_rendersubpass({true, {VK_FORMAT_R8G8B8A8_SRGB, VK_FORMAT_R16G16_UNORM, VK_FORMAT_R8G8B8A8_SRGB, VK_FORMAT_R16G16_SFLOAT}});
  auto repeat = gpu::DEFAULT_SAMPLER;
  repeat.addressModeU = repeat.addressModeV = VK_SAMPLER_ADDRESS_MODE_REPEAT;
  sampler = gpu::create_sampler(repeat);
  // host-visible here: the raster program premultiplies view_projection * model per draw on the host
  transform_buffer = graph.create_buffer(VMA_MEMORY_USAGE_CPU_TO_GPU, sizeof(glm::mat4) * 1000, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT|VK_BUFFER_USAGE_TRANSFER_DST_BIT);
  for (auto tex : target.textures) {
    auto &img = target.images[tex.image_index];
    texture_views.emplace_back(new gpu::ImageViewObject {img.get(), gpu::ImageViewRange {VK_IMAGE_VIEW_TYPE_2D, COLOR, 0, img->get_mip_levels(), 0, 1}});
    scene_textures.push_back({(VkImageView)texture_views.back().get(), target.samplers[tex.sampler_index]});
  }
  bindless_textures = gpu::allocate_descriptor_set(opaque_taa_pipeline.get_layout(1), {std::max<uint32_t>(1, uint32_t(scene_textures.size()))});
  if (!scene_textures.empty()) gpu::write_set(bindless_textures, gpu::ArrayOfImagesBinding {0, scene_textures});
}

// :105-131: depth-first walk, transforms = [model, transpose(inverse(model))] per drawn node
void SceneRenderer::update_scene() {
  if (!owner) throw std::runtime_error {"SceneRenderer::update_scene before init_pipeline"};
  std::vector<glm::mat4> transforms;
  draw_calls.clear();
  struct Walk {
    std::vector<glm::mat4> &transforms;
    std::vector<DrawCall> &calls;
    void node(const scene::BaseNode &n, const glm::mat4 &acc) {
      const glm::mat4 m = acc * n.transform;
      if (n.mesh_index >= 0) {
        calls.push_back(DrawCall {uint32_t(transforms.size()/2), uint32_t(n.mesh_index)});
        transforms.push_back(m);
        transforms.push_back(glm::transpose(glm::inverse(m)));
      }
      for (const auto &c : n.children) node(c, m);
    }
  } walk {transforms, draw_calls};
  for (const auto &n : target.base_nodes) walk.node(n, glm::mat4 {1.f});
  auto &buf = owner->get_buffer(transform_buffer);
  if (sizeof(glm::mat4) * transforms.size() > buf->get_size()) throw std::runtime_error {"Too many scene transforms"};
  std::memcpy(buf->get_mapped_ptr(), transforms.data(), sizeof(glm::mat4) * transforms.size());
}

// :140-220: clear, bind geometry + transforms + bindless textures, one draw_indexed per primitive with its PushData;
// CmdContext::end_renderpass hands the recorded draws to the raster program as one pass
void SceneRenderer::draw_taa(RenderGraph &graph, const Gbuffer &gbuffer, const DrawTAAParams &params) {
  static_assert(sizeof(vkr_gbuf_const) == 2 * sizeof(glm::mat4) + 2 * sizeof(glm::vec4), "GbufConst must match the C-ABI");
  vkr_gbuf_const consts {};
  copy_mat(consts.view_projection, params.mvp);
  copy_mat(consts.prev_view_projection, params.prev_mvp);
  std::memcpy(consts.jitter, &params.jitter, sizeof(consts.jitter));
  std::memcpy(consts.fovy_aspect_znear_zfar, &params.fovy_aspect_znear_zfar, sizeof(consts.fovy_aspect_znear_zfar));
  const std::vector<rec::Binding> binds {
    rec::uniform(0, consts), rec::storage_buffer(1, transform_buffer),
    rec::color_target(gbuffer.albedo), rec::color_target(gbuffer.normal), rec::color_target(gbuffer.material),
    rec::color_target(gbuffer.velocity_vectors), rec::depth_target(gbuffer.depth)};
  const uint32_t w = gbuffer.w, h = gbuffer.h;

  struct Data { std::vector<rec::Bound> bound; };
  graph.add_task<Data>("GbufferPass",
    [&](Data &d, rendergraph::RenderGraphBuilder &builder) { d.bound = rec::declare(binds, builder, VK_SHADER_STAGE_VERTEX_BIT); },
    [=](Data &d, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      std::vector<gpu::ImageViewObject> targets;
      for (const auto &r : d.bound)
        if (r.b.kind == rec::Binding::Color || r.b.kind == rec::Binding::Depth) targets.push_back(resources.get_image_range(r.view));
      cmd.set_framebuffer(w, h, targets);
      cmd.bind_pipeline(opaque_taa_pipeline);
      cmd.clear_color_attachments(0.f, 0.f, 0.f, 0.f);
      cmd.clear_depth_attachment(1.f);
      cmd.bind_viewport(0.f, 0.f, float(w), float(h), 0.f, 1.f);
      cmd.bind_scissors(0, 0, w, h);
      cmd.bind_vertex_buffers(0, {target.vertex_buffer->api_buffer()}, {0ul});
      cmd.bind_index_buffer(target.index_buffer->api_buffer(), 0, VK_INDEX_TYPE_UINT32);
      cmd.bind_descriptors_graphics(0, {rec::write(d.bound, resources, cmd)});
      cmd.bind_descriptors_graphics(1, {bindless_textures});
      for (const auto &call : draw_calls) {
        for (const auto &prim : target.root_meshes[call.mesh].primitives) {
          const auto &material = target.materials[prim.material_index];
          const uint32_t ntex = uint32_t(scene_textures.size());
          const uint32_t pc[4] {call.transform, material.albedo_tex_index < ntex? material.albedo_tex_index : scene::INVALID_TEXTURE,
                                material.metalic_roughness_index < ntex? material.metalic_roughness_index : scene::INVALID_TEXTURE,
                                material.clip_alpha? 0xffu : 0u};  // PushData :132-137
          cmd.push_constants_graphics(VK_SHADER_STAGE_VERTEX_BIT|VK_SHADER_STAGE_FRAGMENT_BIT, 0, sizeof(pc), pc);
          cmd.draw_indexed(prim.index_count, 1, prim.index_offset, int32_t(prim.vertex_offset), 0);
        }
      }
      cmd.end_renderpass();
    });
}
