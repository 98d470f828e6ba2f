// scene_renderer.hpp — the G-buffer resource set and the per-frame camera block every hot-path
// pass consumes.  Mirrors `Gbuffer` (src/scene_renderer.hpp:11-24, ctor scene_renderer.cpp:8-44)
// and `DrawTAAParams` (scene_renderer.hpp:26-33), and `SceneRenderer` (scene_renderer.hpp:35-65): the raster
// stage that fills the G-buffer (draw_taa, scene_renderer.cpp:140-220) over the compute rasterizer bound
// as program "gbuf_opaque_taa".  SyntheticGbuffer (synthetic_gbuffer.hpp) fills the same attachments
// without geometry for the benchmark.
#ifndef SCENE_RENDERER_HPP_INCLUDED
#define SCENE_RENDERER_HPP_INCLUDED

#include "glm_compat.hpp"
#include "gpu/gpu.hpp"
#include "rendergraph/rendergraph.hpp"
#include "scene.hpp"

struct Gbuffer {
  Gbuffer(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height);

  rendergraph::ImageResourceId albedo;
  rendergraph::ImageResourceId normal;
  rendergraph::ImageResourceId downsampled_normals;
  rendergraph::ImageResourceId material;
  rendergraph::ImageResourceId depth;
  rendergraph::ImageResourceId prev_depth;
  rendergraph::ImageResourceId velocity_vectors;
  rendergraph::ImageResourceId downsampled_velocity_vectors;

  uint32_t w, h;

  // ---- multi-GPU tiling (not in the reference) ---------------------------------------------------
  // When the frame is tiled, reads with unbounded reach (Hi-Z march, hit normal, hit colour) go
  // to whole-frame copies assembled by the launcher over RCCL.  `frame_*` equal the window-local
  // images on a single GPU.
  bool tiled = false;
  rendergraph::ImageResourceId frame_hiz;      // D24, mips = image mips 1..L-1 of the whole frame
  rendergraph::ImageResourceId frame_normals;  // whole-frame downsampled_normals
  rendergraph::ImageResourceId frame_albedo;   // whole-frame albedo
  void enable_tiling(rendergraph::RenderGraph &graph, uint32_t full_width, uint32_t full_height);
};

struct DrawTAAParams {
  glm::mat4 mvp;
  glm::mat4 prev_mvp;
  glm::mat4 camera;
  glm::mat4 prev_camera;
  glm::vec4 jitter;
  glm::vec4 fovy_aspect_znear_zfar;
};

struct SceneRenderer {
  SceneRenderer(scene::CompiledScene &s) : target {s} {}

  void init_pipeline(rendergraph::RenderGraph &graph, const Gbuffer &buffer);
  void update_scene();
  void draw_taa(rendergraph::RenderGraph &graph, const Gbuffer &gbuffer, const DrawTAAParams &params);

  struct DrawCall {
    uint32_t transform;
    uint32_t mesh;
  };

  const std::vector<DrawCall> &get_drawcalls() const { return draw_calls; }
  rendergraph::BufferResourceId get_scene_transforms() const { return transform_buffer; }

private:
  scene::CompiledScene &target;
  rendergraph::RenderGraph *owner = nullptr;
  gpu::GraphicsPipeline opaque_taa_pipeline;
  VkSampler sampler;
  rendergraph::BufferResourceId transform_buffer;
  VkDescriptorSet bindless_textures {nullptr};
  std::vector<std::unique_ptr<gpu::ImageViewObject>> texture_views;
  std::vector<std::pair<VkImageView, VkSampler>> scene_textures;
  std::vector<DrawCall> draw_calls;
};

#endif
