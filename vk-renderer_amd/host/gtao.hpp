// gtao.hpp — ground-truth ambient occlusion pass, public interface of src/gtao.hpp:10-121.
// The passes the reference's frame loop runs (main.cpp:384-388) — add_main_pass, add_filter_pass,
// add_accumulate_pass, remap — over the C-ABI programs gtao_compute_main / gtao_filter /
// gtao_accumulate, and the variants it ships but never records: graphics
// main pass ("gtao_main"), static reprojection ("gtao_reproject"), deinterleaved depth + main pass
// ("deinterleave_depth", "main_deinterleaved").  The ray-query pass (add_main_rt_pass,
// gtao.cpp:150-196) needs a scene acceleration structure and is not part of this path: the
// constructor throws when use_ray_query is set.
#ifndef GTAO_HPP_INCLUDED
#define GTAO_HPP_INCLUDED

#include "glm_compat.hpp"
#include "rendergraph/rendergraph.hpp"
#include "scene_renderer.hpp"

rendergraph::ImageResourceId create_gtao_texture(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height);

struct GTAOParams {
  glm::mat4 normal_mat;
  float fovy;
  float aspect;
  float znear;
  float zfar;
};

struct GTAOReprojection {
  glm::mat4 camera_to_prev_frame;
  float fovy;
  float aspect;
  float znear;
  float zfar;
};

struct GTAO {
  GTAO(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height, bool use_ray_query, bool half_res = true, int pattern_n = 2);

  void add_main_pass(
    rendergraph::RenderGraph &graph,
    const GTAOParams &params,
    rendergraph::ImageResourceId depth,
    rendergraph::ImageResourceId normal,
    rendergraph::ImageResourceId material,
    rendergraph::ImageResourceId preintegrated_pdf);

  void add_main_pass_graphics(
    rendergraph::RenderGraph &graph,
    const GTAOParams &params,
    rendergraph::ImageResourceId depth,
    rendergraph::ImageResourceId normal);

  void add_filter_pass(
    rendergraph::RenderGraph &graph,
    const GTAOParams &params,
    rendergraph::ImageResourceId depth);

  void add_reprojection_pass(
    rendergraph::RenderGraph &graph,
    const GTAOReprojection &params,
    rendergraph::ImageResourceId depth,
    rendergraph::ImageResourceId prev_depth);

  void add_accumulate_pass(
    rendergraph::RenderGraph &graph,
    const DrawTAAParams &params,
    const Gbuffer &gbuffer);

  void deinterleave_depth(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId depth);
  void add_main_pass_deinterleaved(
    rendergraph::RenderGraph &graph,
    const GTAOParams &params,
    rendergraph::ImageResourceId normal);

  void remap(rendergraph::RenderGraph &graph) {
    graph.remap(accumulated_history, accumulated_ao);
  }

  rendergraph::ImageResourceId raw; //output of main pass
  rendergraph::ImageResourceId filtered; //output of filter pass
  rendergraph::ImageResourceId prev_frame; //previous frame
  rendergraph::ImageResourceId output; //final
  rendergraph::ImageResourceId accumulated_ao;
  rendergraph::ImageResourceId accumulated_history;
  rendergraph::ImageResourceId deinterleaved_depth;

  // ---- headless controls (ImGui toggles of gtao.cpp:528-536 in the reference) -------------------
  // The reference adds rand()/RAND_MAX - 0.5 to the per-frame angle (gtao.cpp:111); parity runs
  // pin it instead.  NaN = keep the reference behaviour.
  void pin_angle_jitter(float jitter) { pinned_jitter = jitter; }
  void set_mis(bool enabled) { mis_gtao = enabled; }
  void set_two_directions(bool enabled) { two_directions = enabled; }
  void set_only_reflections(bool enabled) { only_reflections = enabled; }
  void set_weight_ratio(float ratio) { weight_ratio = ratio; }
  void request_clear_history() { clear_history = true; }
  void set_frame_count(uint32_t n) { frame_count = n; }

private:
  float next_base_angle();

  gpu::GraphicsPipeline main_pipeline_gfx;
  gpu::ComputePipeline reproject_pipeline;
  gpu::ComputePipeline deinterleave_pipeline;
  gpu::ComputePipeline main_deinterleaved_pipeline;
  gpu::ComputePipeline main_pipeline;
  gpu::ComputePipeline filter_pipeline;
  gpu::ComputePipeline accumulate_pipeline;

  int deinterleave_n = 2;
  uint32_t depth_lod = 0;

  bool mis_gtao = true;
  bool two_directions = false;
  bool only_reflections = false;
  bool clear_history = false;
  float weight_ratio = 1.0;
  float pinned_jitter;

  uint32_t frame_count = 0;

  VkSampler sampler;
};

#endif
