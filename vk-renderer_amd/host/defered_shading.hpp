// defered_shading.hpp — deferred-shading composite, public interface of src/defered_shading.hpp:8-32
// (produces TAA's colour input, main.cpp:390-391).  The SDL window of the
// reference constructor only feeds ImGui; it is accepted and ignored.
#ifndef DEFFERED_SHADING_HPP_INCLUDED
#define DEFFERED_SHADING_HPP_INCLUDED

#include "rendergraph/rendergraph.hpp"
#include "scene_renderer.hpp"

struct SDL_Window;

struct DeferedShadingPass {
  DeferedShadingPass(rendergraph::RenderGraph &graph, SDL_Window *window);

  void update_params(const glm::mat4 &camera, const glm::mat4 &shadow, float fovy, float aspect, float znear, float zfar);

  void draw(rendergraph::RenderGraph &graph,
    const Gbuffer &gbuffer,
    rendergraph::ImageResourceId shadow,
    rendergraph::ImageResourceId ssao,
    rendergraph::ImageResourceId brdf_tex,
    rendergraph::ImageResourceId reflections,
    rendergraph::ImageResourceId out_image);

  // headless equivalents of the ImGui sliders (defered_shading.cpp:120-126)
  void set_roughness_range(float lo, float hi) { min_max_roughness = glm::vec2 {lo, hi}; }
  void set_only_ao(bool v) { only_ao = v; }

private:
  gpu::GraphicsPipeline pipeline;
  VkSampler sampler;
  rendergraph::BufferResourceId ubo_consts;

  glm::vec2 min_max_roughness {0.f, 1.f};
  bool only_ao = false;
  rendergraph::RenderGraph *graph_ref = nullptr;  // update_params writes the constant buffer the graph owns
};

#endif
