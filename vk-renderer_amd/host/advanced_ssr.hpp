// advanced_ssr.hpp — stochastic Hi-Z screen-space reflections, public interface of
// src/advanced_ssr.hpp:7-113.  Implemented: run() = trace -> filter -> blur (advanced_ssr.cpp:551-553),
// preintegrate_pdf / preintegrate_brdf, remap_images, the getters, and the tile-classified trace the
// reference leaves commented out of run() (advanced_ssr.cpp:547-550):
// clear_indirect_params -> run_classification_pass -> run_trace_indirect_pass, selected with
// Settings::use_tile_classification (default off = the reference's behaviour).  The tile-regression
// experiment (advanced_ssr.cpp:497-538) stays out of scope.
#ifndef ADVANCED_SSR_HPP_INCLUDED
#define ADVANCED_SSR_HPP_INCLUDED

#include <vector>

#include "rendergraph/rendergraph.hpp"
#include "scene_renderer.hpp"

struct AdvancedSSRParams {
  glm::mat4 normal_mat;
  float fovy;
  float aspect;
  float znear;
  float zfar;
};

std::vector<glm::vec4> halton23_seq(uint32_t count);

struct AdvancedSSR {
  AdvancedSSR(rendergraph::RenderGraph &graph, uint32_t w, uint32_t h);
  void run(
    rendergraph::RenderGraph &graph,
    const AdvancedSSRParams &params,
    const DrawTAAParams &taa_params,
    const Gbuffer &gbuff,
    rendergraph::ImageResourceId ssr_occlusion);

  void preintegrate_pdf(rendergraph::RenderGraph &graph);
  void preintegrate_brdf(rendergraph::RenderGraph &graph);
  void remap_images(rendergraph::RenderGraph &graph) { graph.remap(blurred_reflection, blurred_reflection_history); }

  rendergraph::ImageResourceId get_ouput() const { return reflections; }
  rendergraph::ImageResourceId get_rays() const { return rays; }
  rendergraph::ImageResourceId get_blurred() const { return blurred_reflection; }
  rendergraph::ImageResourceId get_blurred_history() const { return blurred_reflection_history; }
  rendergraph::ImageResourceId get_occlusion() const { return rays_occlusion; }
  rendergraph::ImageResourceId get_preintegrated_pdf() const { return preintegrated_pdf; }
  rendergraph::ImageResourceId get_preintegrated_brdf() const { return preintegrated_brdf; }

  // headless equivalents of the ImGui controls (advanced_ssr.cpp:556-567)
  struct Settings {
    float max_rougness = 1.f;
    float glossy_roughness_value = 0.5f;
    bool normalize_reflections = true;
    bool accumulate_reflections = true;
    bool bilateral_filter = true;
    bool update_random = true;
    bool use_blur = true;
    int max_accumulated_rays = 16;
    bool use_tile_classification = false;  // run(): classification + indirect trace instead of run_trace_pass
  };
  Settings &get_settings() { return settings; }
  void set_counter(uint32_t c) { counter = c; }

  // the two halves of run(), for drivers that interleave an exchange between them (multi-GPU: the
  // trace needs the gathered Hi-Z pyramid, the filter the gathered albedo)
  void run_trace(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, rendergraph::ImageResourceId ssr_occlusion);
  void run_resolve(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const DrawTAAParams &taa_params, const Gbuffer &gbuff);

  // advanced_ssr.cpp:440-495,216-302 (private in the reference; public here so drivers can record them one by one)
  void clear_indirect_params(rendergraph::RenderGraph &graph);
  void run_classification_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff);
  void run_trace_indirect_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff);
  rendergraph::BufferResourceId get_reflective_tiles() const { return reflective_tiles; }
  rendergraph::BufferResourceId get_glossy_tiles() const { return glossy_tiles; }
  rendergraph::BufferResourceId get_reflective_indirect() const { return reflective_indirect; }
  rendergraph::BufferResourceId get_glossy_indirect() const { return glossy_indirect; }

private:
  gpu::BufferPtr halton_buffer;

  gpu::ComputePipeline trace_pass;
  gpu::ComputePipeline filter_pass;
  gpu::ComputePipeline blur_pass;
  gpu::ComputePipeline preintegrate_pass;
  gpu::ComputePipeline preintegrate_brdf_pass;
  gpu::ComputePipeline classification_pass;
  gpu::ComputePipeline trace_indirect_pass;

  rendergraph::BufferResourceId reflective_indirect;
  rendergraph::BufferResourceId glossy_indirect;
  rendergraph::BufferResourceId reflective_tiles;
  rendergraph::BufferResourceId glossy_tiles;

  VkSampler sampler;

  rendergraph::ImageResourceId rays;
  rendergraph::ImageResourceId reflections;
  rendergraph::ImageResourceId blurred_reflection;
  rendergraph::ImageResourceId blurred_reflection_history;
  rendergraph::ImageResourceId rays_occlusion;
  rendergraph::ImageResourceId preintegrated_pdf;
  rendergraph::ImageResourceId preintegrated_brdf;

  uint32_t counter {0u};
  Settings settings;

  void run_trace_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, rendergraph::ImageResourceId ssr_occlusion);
  void run_filter_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff);
  void run_blur_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const DrawTAAParams &taa_params, const Gbuffer &gbuff);
};

#endif
