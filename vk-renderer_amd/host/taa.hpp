// taa.hpp — temporal anti-aliasing resolve, public interface of src/taa.hpp:8-21.
#ifndef TAA_HPP_INCLUDED
#define TAA_HPP_INCLUDED

#include "glm_compat.hpp"
#include "rendergraph/rendergraph.hpp"
#include "scene_renderer.hpp"

struct TAA {
  TAA(rendergraph::RenderGraph &graph, uint32_t w, uint32_t h);

  void run(rendergraph::RenderGraph &graph, const Gbuffer &gbuffer, rendergraph::ImageResourceId color, const DrawTAAParams &params);
  void remap_targets(rendergraph::RenderGraph &graph);

  rendergraph::ImageResourceId get_output() const { return target; }
  rendergraph::ImageResourceId get_history() const { return history; }

private:
  rendergraph::ImageResourceId history;
  rendergraph::ImageResourceId target;
  gpu::ComputePipeline pipeline;
  VkSampler sampler;
};

#endif
