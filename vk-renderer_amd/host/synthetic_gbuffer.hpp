// synthetic_gbuffer.hpp — headless replacement of the G-buffer raster stage
// (SceneRenderer::draw_taa, src/scene_renderer.cpp:140-220 + shaders/gbuf/opaque_taa.*).
// Writes the same attachments of `Gbuffer` from an analytic scene so the post-process chain runs
// without Vulkan, geometry or textures.  Same call shape as draw_taa.
#ifndef SYNTHETIC_GBUFFER_HPP_INCLUDED
#define SYNTHETIC_GBUFFER_HPP_INCLUDED

#include "rendergraph/rendergraph.hpp"
#include "scene_renderer.hpp"

struct SyntheticGbuffer {
  explicit SyntheticGbuffer(uint32_t seed = 0x5EED0001u);

  // fills albedo / normal / material / velocity_vectors / depth (mip 0) for `params`
  void draw_taa(rendergraph::RenderGraph &graph, const Gbuffer &gbuffer, const DrawTAAParams &params);
  // fills mip 0 of `depth_target` only, as seen from `camera` (used to seed prev_depth)
  void draw_depth(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId depth_target, const glm::mat4 &camera, const glm::mat4 &mvp,
                  const glm::vec4 &fovy_aspect_znear_zfar);

private:
  gpu::GraphicsPipeline pipeline;
  uint32_t seed;
};

#endif
