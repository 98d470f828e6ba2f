// screen_trace.hpp — ScreenSpaceTrace, public interface of src/screen_trace.hpp:8-52 (SURVEY.md 8(a)
// row R2): a one-bounce screen-space radiance + horizon-AO tracer with a 4x4 depth-aware filter and
// a static-reprojection accumulator.  Not recorded by the reference's frame loop; kept as a drop-in.
#ifndef SCREEN_TRACE_HPP_INCLUDED
#define SCREEN_TRACE_HPP_INCLUDED

#include <limits>
#include <random>

#include "glm_compat.hpp"
#include "rendergraph/rendergraph.hpp"

struct ScreenTraceParams {
  glm::mat4 normal_mat;
  float fovy;
  float aspect;
  float znear;
  float zfar;
};

struct ScreenSpaceTrace {
  ScreenSpaceTrace(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height);

  void add_main_pass(
    rendergraph::RenderGraph &graph,
    const ScreenTraceParams &params,
    rendergraph::ImageResourceId depth,
    rendergraph::ImageResourceId normal,
    rendergraph::ImageResourceId color,
    rendergraph::ImageResourceId material);

  void add_filter_pass(
    rendergraph::RenderGraph &graph,
    const ScreenTraceParams &params,
    rendergraph::ImageResourceId depth);

  void add_accumulate_pass(
    rendergraph::RenderGraph &graph,
    const ScreenTraceParams &params,
    rendergraph::ImageResourceId depth,
    rendergraph::ImageResourceId prev_depth);

  rendergraph::ImageResourceId raw;
  rendergraph::ImageResourceId filtered;
  rendergraph::ImageResourceId accumulated;

  // headless control: the reference draws the angle jitter and random_offset from a
  // std::default_random_engine each frame (screen_trace.cpp:49-53); parity runs pin both.
  void pin_randoms(float angle_jitter, float random_offset) { pinned_jitter = angle_jitter; pinned_offset = random_offset; }
  void set_frame_count(uint32_t n) { frame_count = n; }

private:
  std::uniform_real_distribution<float> random_floats {0.0, 1.0};
  std::default_random_engine generator;
  float pinned_jitter = std::numeric_limits<float>::quiet_NaN();
  float pinned_offset = std::numeric_limits<float>::quiet_NaN();

  gpu::ComputePipeline trace_pipeline;
  gpu::ComputePipeline filter_pipeline;
  gpu::ComputePipeline accum_pipeline;

  uint32_t frame_count = 0;
  VkSampler sampler;
};

#endif
